// cdpr_kernels.hpp — step kernels of the batched CDPR engine (gfx950).
//
// Mapping "lane per robot" (LPR): one lane owns one robot for the whole step, so the
// 6xn structure matrix, the 6x6 normal matrix and the controller windows of that
// robot never leave its registers; a wavefront is 64 independent robots.  HBM sees
// only float4 struct-of-array slots: slot s of robot r lives at base[s * stride + r],
// so every wave-wide load/store is one contiguous 1 KiB dwordx4 access.
//
// State slots (float4 each):
//   0: px py pz qx   1: qy qz qw vx   2: vy vz wx wy   3: wz fkx fky fkz   4: fkqx fkqy fkqz fkqw (FK only)
//   P + 3c + {0,1,2}: cable c controller record: e0..e3 | e4..e7 | e8 e9 Ierr count
// Observable slots (float4 each), what publishJointStates / publishPlatformState carry
// (PLG.cpp:248-280):
//   0..2 as state slots 0..2 at the published step, 3: wz fk_residual fk_iterations td_infeasible
//   4 + {0..G-1}: joint position, +G: joint velocity, +2G: effort   (G = ceil(n/4) slots each)
#pragma once
#include "cdpr_device.hpp"

namespace cdpr {

enum StepFlags : uint32_t {
  kFlagFirstWorldStep = 1u << 0,  // JFC.cpp:61-66: stepTime <= 0 at t = 0 -> force 0, no PID call
  kFlagResetPid = 1u << 1,        // mode switch latched with this launch (JFC.cpp:101-103,113-115)
  kFlagActualIsVelocity = 1u << 2,  // velocity mode: Pid sees joint velocity (JFC.cpp:76), else position (JFC.cpp:88)
};

struct StepArgs {
  float4* state;
  float4* obs;
  const float* cmd;  // latched Joy.axes of the active mode, float[B][n]; nullptr -> desired 0 (state after Load)
  float* dbg;        // float[B][9] `pid` debug topic, or nullptr
  uint32_t batch;
  uint32_t stride;   // robots per slot row (batch rounded up to 64)
  int nsteps;        // world steps fused into this launch
  uint32_t flags;
  uint64_t publish_mask;  // bit k: publish observables at step k of this launch (PLG.cpp:236-242)
  StepConsts c;
  PidConsts pid;
};

__host__ __device__ constexpr int plat_slots(bool fk) { return fk ? 5 : 4; }
__host__ __device__ constexpr int joint_groups(int n) { return (n + 3) / 4; }
__host__ __device__ constexpr int state_slots(int n, bool fk) { return plat_slots(fk) + 3 * n; }
__host__ __device__ constexpr int obs_slots(int n) { return 4 + 3 * joint_groups(n); }

template <int N>
__device__ __forceinline__ void store_groups(float4* base, uint32_t stride, uint32_t r, const float (&v)[N]) {
  constexpr int G = joint_groups(N);
#pragma unroll
  for (int g = 0; g < G; ++g) {
    float4 o;
    o.x = v[4 * g];
    o.y = (4 * g + 1 < N) ? v[(4 * g + 1 < N) ? 4 * g + 1 : 0] : 0.f;
    o.z = (4 * g + 2 < N) ? v[(4 * g + 2 < N) ? 4 * g + 2 : 0] : 0.f;
    o.w = (4 * g + 3 < N) ? v[(4 * g + 3 < N) ? 4 * g + 3 : 0] : 0.f;
    base[(size_t)g * stride + r] = o;
  }
}

template <int N, bool FK, bool TD>
__global__ __launch_bounds__(64) void step_lane_per_robot(const StepArgs a) {
  const uint32_t r = blockIdx.x * 64u + threadIdx.x;
  if (r >= a.batch) return;
  const size_t st = a.stride;
  constexpr int P = plat_slots(FK);
  const StepConsts& c = a.c;

  // ---- load: everything is issued up front so the whole record is in flight at once
  float4* S = a.state + r;
  const float4 p0 = S[0 * st], p1 = S[1 * st], p2 = S[2 * st], p3 = S[3 * st];
  float4 p4 = make_float4(0.f, 0.f, 0.f, 1.f);
  if (FK) p4 = S[4 * st];
  float4 craw[N][3];
#pragma unroll
  for (int i = 0; i < N; ++i) {
#pragma unroll
    for (int k = 0; k < 3; ++k) craw[i][k] = S[(size_t)(P + 3 * i + k) * st];
  }
  float desired[N];
  if (a.cmd) {
    const float* cp = a.cmd + (size_t)r * N;
    if (N % 4 == 0) {
#pragma unroll
      for (int g = 0; g < N / 4; ++g) {
        const float4 v = reinterpret_cast<const float4*>(cp)[g];
        desired[4 * g] = v.x;
        desired[4 * g + 1] = v.y;
        desired[4 * g + 2] = v.z;
        desired[4 * g + 3] = v.w;
      }
    } else {
#pragma unroll
      for (int i = 0; i < N; ++i) desired[i] = cp[i];
    }
  } else {
#pragma unroll
    for (int i = 0; i < N; ++i) desired[i] = 0.f;
  }

  Platform s;
  s.px = p0.x; s.py = p0.y; s.pz = p0.z; s.qx = p0.w;
  s.qy = p1.x; s.qz = p1.y; s.qw = p1.z; s.vx = p1.w;
  s.vy = p2.x; s.vz = p2.y; s.wx = p2.z; s.wy = p2.w;
  s.wz = p3.x;
  float fkx = p3.y, fky = p3.z, fkz = p3.w, fkqx = p4.x, fkqy = p4.y, fkqz = p4.z, fkqw = p4.w;

  CableCtrl ctl[N];
  const bool reset = (a.flags & kFlagResetPid) != 0u;
#pragma unroll
  for (int i = 0; i < N; ++i) {
    ctl[i].e[0] = craw[i][0].x; ctl[i].e[1] = craw[i][0].y; ctl[i].e[2] = craw[i][0].z; ctl[i].e[3] = craw[i][0].w;
    ctl[i].e[4] = craw[i][1].x; ctl[i].e[5] = craw[i][1].y; ctl[i].e[6] = craw[i][1].z; ctl[i].e[7] = craw[i][1].w;
    ctl[i].e[8] = craw[i][2].x; ctl[i].e[9] = craw[i][2].y;
    ctl[i].ierr = craw[i][2].z;
    ctl[i].cnt = craw[i][2].w;
    if (reset) {  // Pid::reset, Pid.cpp:100-115
#pragma unroll
      for (int j = 0; j < kWin; ++j) ctl[i].e[j] = 0.f;
      ctl[i].ierr = 0.f;
      ctl[i].cnt = 0.f;
    }
  }

  const bool actual_is_vel = (a.flags & kFlagActualIsVelocity) != 0u;

  for (int step = 0; step < a.nsteps; ++step) {
    // ---- IK on the state at t_k (Joint::Position / GetVelocity)
    float len[N], jac[N][6], q[N], qd[N];
    ik_all<N>(c, s.px, s.py, s.pz, s.qx, s.qy, s.qz, s.qw, len, jac);
#pragma unroll
    for (int i = 0; i < N; ++i) {
      q[i] = c.l0[i] - len[i];
      qd[i] = -(jac[i][0] * s.vx + jac[i][1] * s.vy + jac[i][2] * s.vz + jac[i][3] * s.wx + jac[i][4] * s.wy +
                jac[i][5] * s.wz);
    }

    // ---- per-cable force (PLG.cpp:222-228 -> JFC.cpp:59-96 -> Pid.cpp:122-191)
    float f[N];
    const bool first = (step == 0) && (a.flags & kFlagFirstWorldStep);
    PidTerms t0;
    t0.wrote = false;
    t0.p = t0.i = t0.d = 0.f;
#pragma unroll
    for (int i = 0; i < N; ++i) {
      if (first) {
        f[i] = 0.f;
      } else {
        PidTerms t;
        f[i] = pid_update(a.pid, c.dt, ctl[i], desired[i], actual_is_vel ? qd[i] : q[i], t);
        if (i == 0) t0 = t;
      }
    }

    // ---- optional estimator + tension distribution
    float applied[N];
    float fk_res = 0.f;
    int fk_it = 0, td_flag = 0;
    if (FK) {
      float jest[N][6];
      fk_solve<N>(c, len, fkx, fky, fkz, fkqx, fkqy, fkqz, fkqw, jest, fk_res, fk_it);
      if (TD) td_flag = td_solve<N>(c, jest, f, applied);
    } else if (TD) {
      td_flag = td_solve<N>(c, jac, f, applied);
    }
    if (!TD) {
#pragma unroll
      for (int i = 0; i < N; ++i) applied[i] = f[i];
    }
    if (c.effort >= 0.f) {  // Joint::SetForce clamp (cube.sdf:438)
#pragma unroll
      for (int i = 0; i < N; ++i) applied[i] = fmaxf(fminf(applied[i], c.effort), -c.effort);
    }

    if (a.dbg) {  // `pid` topic, cable 0 only (PLG.cpp:223-227; Pid.cpp:139-142,158-168)
      float* d = a.dbg + (size_t)r * 9;
      if (t0.wrote) {
        d[0] = t0.p;
        d[1] = t0.i;
        d[2] = t0.d;
        d[3] = desired[0];
      }
      d[4] = applied[0];
    }

    // ---- observables of step t_k (PLG.cpp:236-242, 248-280)
    if ((a.publish_mask >> step) & 1ull) {
      float4* O = a.obs + r;
      O[0 * st] = make_float4(s.px, s.py, s.pz, s.qx);
      O[1 * st] = make_float4(s.qy, s.qz, s.qw, s.vx);
      O[2 * st] = make_float4(s.vy, s.vz, s.wx, s.wy);
      O[3 * st] = make_float4(s.wz, fk_res, (float)fk_it, (float)td_flag);
      constexpr int G = joint_groups(N);
      store_groups<N>(O + (size_t)4 * st, a.stride, 0, q);
      store_groups<N>(O + (size_t)(4 + G) * st, a.stride, 0, qd);
      store_groups<N>(O + (size_t)(4 + 2 * G) * st, a.stride, 0, applied);
    }

    // ---- world step to t_{k+1}
    dynamics_step<N>(c, s, jac, applied, qd);
  }

  // ---- store
  S[0 * st] = make_float4(s.px, s.py, s.pz, s.qx);
  S[1 * st] = make_float4(s.qy, s.qz, s.qw, s.vx);
  S[2 * st] = make_float4(s.vy, s.vz, s.wx, s.wy);
  S[3 * st] = make_float4(s.wz, fkx, fky, fkz);
  if (FK) S[4 * st] = make_float4(fkqx, fkqy, fkqz, fkqw);
#pragma unroll
  for (int i = 0; i < N; ++i) {
    S[(size_t)(P + 3 * i + 0) * st] = make_float4(ctl[i].e[0], ctl[i].e[1], ctl[i].e[2], ctl[i].e[3]);
    S[(size_t)(P + 3 * i + 1) * st] = make_float4(ctl[i].e[4], ctl[i].e[5], ctl[i].e[6], ctl[i].e[7]);
    S[(size_t)(P + 3 * i + 2) * st] = make_float4(ctl[i].e[8], ctl[i].e[9], ctl[i].ierr, ctl[i].cnt);
  }
}

}  // namespace cdpr
