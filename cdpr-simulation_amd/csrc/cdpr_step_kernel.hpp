// cdpr_step_kernel.hpp — the fused CDPR step kernel (gfx950, fp32), lane-per-robot mapping.
//
// One lane owns one robot for the whole step: the 6xn structure matrix, the 6x6 normal
// matrix, the controller windows never leave its registers; a wavefront is 64 independent
// robots and a workgroup is one wavefront (65 536 robots = 1 024 workgroups = one wave on
// every SIMD of the chip).
//
// Design points (evidence in profiles/ and DESIGN.md):
//   * cables are processed in PAIRS held in float2 registers so the per-cable math (IK rows,
//     FIR windows, J^T J / J^T r partial sums, wrench accumulation) issues as v_pk_fma_f32 /
//     v_pk_mul_f32: two cables per VALU instruction.  A packed f32 op holds the SIMD 4 cycles, so this halves
//     the ISSUE slots, not the ALU cycles — which is what counts with one wave per SIMD (4-cycle issue);
//   * every multiply-add is an explicit fma (the file is compiled with -ffp-contract=off), so
//     the one-step and the fused multi-step instantiations perform bit-identical arithmetic;
//   * the per-cable geometry (a_i, b_i, L0_i: 7n constants) is staged once per wave in LDS,
//     pair-interleaved, and re-read with broadcast ds_read_b128 wherever an IK evaluation needs
//     it; as kernel arguments these 56 values overflow the SGPR file and get spilled into VGPR
//     lanes (v_writelane/v_readlane: > 1 200 extra VALU instructions per step, measured);
//   * HBM sees only float4 struct-of-array slots (slot s of robot r at base[s*stride + r]): every
//     wave-wide access is one contiguous 1 KiB dwordx4 row, addressed as SGPR row base + one
//     shared VGPR offset;
//   * the Pid call counter (first call after reset returns 0; derivative window full after N
//     samples) is uniform over the batch, so it is a kernel argument and the branches on it
//     are scalar.
//
// State slots (float4 each):
//   0: px py pz qx   1: qy qz qw vx   2: vy vz wx wy   3: wz fkx fky fkz   4: fkqx fkqy fkqz fkqw (FK only)
//   controller records, per CABLE PAIR k (cables A = 2k, B = 2k+1), the derivative window as a RING of 10 slots:
//     P + 5k + m (m = 0..4): A[2m] B[2m] A[2m+1] B[2m+1]      (ring slots 2m and 2m+1 of both cables)
//     P + 5*NP + g:          Ierr of pairs 2g, 2g+1: A B A B   ("hot" rows, rewritten every step)
//   One step rewrites ONE ring row per pair (the slot the new error goes to) plus the hot rows: 6 rows at n = 8
//   instead of the 24 a shifted window would cost; the ring position is uniform over the batch, so it comes
//   from the kernel argument pid_calls and the FIR weights are looked up pre-rotated (StepArgs.wtab).
// Observable slots (PLG.cpp:248-280): 0..2 pose/twist at the published step,
//   3: wz fk_residual fk_iterations td_infeasible; 4..: joint position, velocity, effort (ceil(n/4) slots each)
//
// Reference paths (relative to src/cdpr_gazebo/): Pid.cpp, JFC.cpp = JointForceCalculator.cpp,
// PLG.cpp = CdprGazeboPlugin.cpp, gen = sdf/gen_cdpr.py.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace cdpr {

constexpr int kMaxCables = 8;
constexpr int kWin = 10;                // prior errors kept per cable (derivative window of up to 11 samples)
constexpr int kGeomFloatsPerPair = 16;  // [ax ay | az bx | by bz | l0 mask], each entry a (cable 2k, cable 2k+1) pair

typedef float v2f __attribute__((ext_vector_type(2)));

enum StepFlags : uint32_t {
  kFlagFirstWorldStep = 1u << 0,    // JFC.cpp:61-66: stepTime <= 0 at t = 0 -> force 0, no Pid call
  kFlagRolloutResetPid = 1u << 1,   // rollout entered from Position mode: setVelocityTarget resets the Pid (JFC.cpp:113-115)
  kFlagActualIsVelocity = 1u << 2,  // velocity mode: Pid sees joint velocity (JFC.cpp:76), else position (JFC.cpp:88)
  kFlagPublishAll = 1u << 4,        // every step of the launch is published (publishPeriod 0; world step 0 never is, PLG.cpp:237): launches
                                    // of more than 64 steps (cdpr_update_scheduled), where publish_mask has no bits left
  kFlagForceMode = 1u << 3,         // UpdateMode::Force (JFC.cpp:67-70): force = the commanded force, no Pid runs (uniform handles;
                                    // per-robot handles carry the mode per lane: meta mode bits 0)
};

// One Pid's parameters as the kernels take them (Pid.cpp:64-73); see the "active Pid" fields of StepArgs.
struct PidSet {
  float kf, kp, ki, kd, inv_ki, imax, imin, cmax, cmin;
  int clamp_cmd;
};

struct StepArgs {
  float4* state;
  float4* obs;
  const float* cmd;   // latched Joy.axes of the active mode, float[B][n] (all zeros until the first message: target 0 after Load)
  float* dbg;         // float[B][9] `pid` debug topic, or nullptr
  const float* geom;  // cable_pairs(n) * 16 floats, pair-interleaved cable geometry
  // ROLLOUT only: S sampled command sequences per robot over `nsteps` steps, nothing written but one cost per trajectory
  const float* roll_cmd;  // float[B][H][S][n]: per robot and step, a batch of S Joy.axes
  const float* roll_ref;  // float[B][3] reference position
  float* roll_cost;       // float[B][S]: sum over the horizon of |p(t_{k+1}) - p_ref|^2
  uint32_t roll_samples;
  unsigned long long* stamps;  // diagnostic builds only (CDPR_STAMPS)
  uint32_t batch;
  uint32_t stride;    // robots per slot row (batch rounded up to 64)
  int nsteps;         // world steps fused into this launch
  uint32_t flags;
  uint64_t publish_mask;  // bit k: publish observables at step k of this launch (PLG.cpp:236-242)
  size_t obs_step_stride; // float4 elements between the observable images of consecutive steps (0: every step
                          // overwrites the same image, as a topic does; > 0: a trajectory record keeps them all)
  int pid_calls;      // Pid::update calls since the last reset, before this launch (uniform over the batch; the host lets
                      // it saturate: only == 0 and >= nbuf matter)
  int ring_slot;      // ring slot the error of this launch's FIRST step goes to; every further step takes the next one.
                      // The ring position follows the WORLD step, (step + 8) % 10 (= the slot a handle that has run
                      // without a Pid reset since Load would use), not the time of the last Pid reset: a reset robot
                      // simply starts filling the ring wherever it stands, and derive() returns 0 until nbuf samples
                      // are in (Pid.cpp:200-203), by when every slot the weights reach has been overwritten.  So robots
                      // whose Pids were reset at different times (per-robot handles) share one pre-rotated weight row.
  // per-robot handles (PR instantiations; cdpr_config_t.per_robot_commands): every robot has its own JointForceCalculator
  // mode and Pid call count, as B independent plugin instances have (PLG.cpp:206-219).  meta[r]: bits 0-1 = mode
  // (1 Position, 2 Velocity, JFC.h:35-37), bits 2-7 = Pid::update calls since the robot's last Pid reset, saturating at
  // 63.  `cmd` is then the robot's ACTIVE target row (the latch kernel copies a Joy's row there when it reaches the
  // robot), the "active Pid" fields below hold the VELOCITY Pid and `alt` the POSITION Pid's gains and limits (the two
  // fit the same derivative window on such handles: one weight table, one nbuf).
  uint8_t* meta;
  PidSet alt;
  // world / body
  float dt, half_dt, dt_inv_mass, fgx, fgy, fgz;  // fg = m * g
  float ib[6], ibinv[6];                           // body inertia and inverse: xx yy zz xy xz yz
  float damping, effort;                           // joint damping; SetForce clamp (effort < 0: none)
  float vel_limit;                                 // SetForce velocity truncation (<= 0: none)
  int unilateral;                                  // cables cannot push
  // travel limits of the prismatic joints (cube.sdf:436-437): flag q outside [lo, hi] per cable (travel_on); PHYS
  // instantiations also model the stop (travel_stop, see apply_travel_stop)
  float travel_lo, travel_hi, inv_mass;
  int travel_on, travel_stop;
  int ph_lumped;                                   // PHYS instantiations: any lumped-leg term non-zero (else the plain world step)
  // lumped legs (PHYS instantiations only; cdpr_config_t.passive_damping ...): joint damping c of the passive
  // revolutes, inertia turning with a leg, mass sliding along the cable, point mass at each platform anchor,
  // n x the inertia each leg adds to the platform, gravity (for the point masses' weight)
  float ph_c, ph_jleg, ph_max, ph_mpt, ph_iadd_total, ph_mass, gx, gy, gz;
  // FK / TD ([NEW] stages)
  float fk_lambda, fk_tol;
  int fk_iters;
  float td_min, td_max, td_mid;
  // the active Pid (Pid.cpp:64-73)
  float kf, kp, ki, kd, inv_ki, imax, imin, cmax, cmin, inv_dt;
  // wtab[ws*12 + s], s < 10: end-point LS derivative weight of ring slot s when the new error goes to slot ws;
  // wtab[ws*12 + 10]: weight of the new error itself (closed form of Pid::derive, Pid.cpp:193-247).  A device
  // buffer, not a kernel-argument array: indexing a by-value argument with a run-time slot sends it to scratch.
  const float* wtab;
  // one-step launches (cdpr_onestep_kernel): the row of wtab for THIS launch's ring position, by value, so the
  // weights are kernel-argument scalar loads whatever the kernel has done to memory before the PID stage
  float wrow[kWin + 2];
  int nbuf, clamp_cmd;
  uint32_t split_swap;  // cdpr_split_kernel: workgroups whose index has odd parity under this mask swap the roles of their waves
  // Command schedule inside a launch of several steps (cdpr_update_scheduled): every sched_refresh steps the lanes take the
  // next Joy batch (batch j at cmd + j * sched_stride floats), after sched_ready[j] has become non-zero if a mailbox is
  // given (a host that fills the schedule while the launch runs).  sched_refresh = 0: one Joy for the whole launch.
  int sched_refresh;
  size_t sched_stride;
  const uint32_t* sched_ready;
  uint32_t* fault;  // with a mailbox: the handle's status word (device-mapped host memory); a wait that gives up ORs bit 0 into it
};

__host__ __device__ constexpr int plat_slots(bool fk) { return fk ? 5 : 4; }
__host__ __device__ constexpr int joint_groups(int n) { return (n + 3) / 4; }
__host__ __device__ constexpr int cable_pairs(int n) { return (n + 1) / 2; }
__host__ __device__ constexpr int ctrl_slots(int n) { return 5 * cable_pairs(n) + (cable_pairs(n) + 1) / 2; }
__host__ __device__ constexpr int state_slots(int n, bool fk) { return plat_slots(fk) + ctrl_slots(n); }
__host__ __device__ constexpr int obs_slots(int n) { return 4 + 3 * joint_groups(n); }

#define CDPR_DEV __device__ __forceinline__

// Is step `step` of this launch published (PLG.cpp:236-242)?
CDPR_DEV bool step_published(const StepArgs& a, int step) {
  if (a.flags & kFlagPublishAll) return !(step == 0 && (a.flags & kFlagFirstWorldStep));
  return ((a.publish_mask >> step) & 1ull) != 0ull;
}
// The mailbox of a scheduled launch: wait until Joy batch j is there (the host, or a producer kernel on another stream, sets
// the word after it has written the batch).  Bounded: ~2^23 polls of ~0.5 us; a producer that never delivers must not hang
// the GPU.  A wait that gives up raises the handle's status word (the host reports CDPR_ERR_DEVICE from cdpr_synchronize
// and the getters on) and every later wait of the launch returns at once: the launch ends in milliseconds, its results
// are declared invalid.
CDPR_DEV void mailbox_wait(const uint32_t* word, uint32_t* fault) {
  if (fault && __hip_atomic_load(fault, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0u) return;
  uint32_t spins = 0;
  while (__hip_atomic_load(word, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) == 0u) {
    if (++spins >= (1u << 23)) {
      if (fault) __hip_atomic_fetch_or(fault, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      return;
    }
    __builtin_amdgcn_s_sleep(16);
  }
}
CDPR_DEV void sched_wait(const StepArgs& a, int j) {
  if (a.sched_ready) mailbox_wait(a.sched_ready + j, a.fault);
}


// Diagnostic build only (-DCDPR_STAMPS, scripts/stamp_probe.py): per-wave s_memrealtime stamps (100 MHz) at the
// phase boundaries of the step kernel, written to a buffer of their own (StepArgs.stamps); never compiled into
// the shipped library.
#ifdef CDPR_STAMPS
#define CDPR_STAMP(i)                                                                           \
  do {                                                                                          \
    __builtin_amdgcn_sched_barrier(0);                                                          \
    if (a.stamps && threadIdx.x == 0) a.stamps[(size_t)blockIdx.x * 8 + (i)] = __builtin_amdgcn_s_memrealtime(); \
    __builtin_amdgcn_sched_barrier(0);                                                          \
  } while (0)
#else
#define CDPR_STAMP(i) do { } while (0)
#endif

// leaf helpers: without debug locations in the instruction-budget build (scripts/region_budget.py), so that their
// instructions count for the region that calls them
#ifdef CDPR_LEAF_NODEBUG
#define CDPR_LEAF __device__ __forceinline__ __attribute__((nodebug))
#else
#define CDPR_LEAF CDPR_DEV
#endif
CDPR_LEAF v2f splat(float s) { return (v2f){s, s}; }
CDPR_LEAF v2f fma2(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); }
CDPR_LEAF v2f fma2(float a, v2f b, v2f c) { return __builtin_elementwise_fma(splat(a), b, c); }
// x + y of a pair as ONE v_add_f32 on the two halves.  Written as `v.x + v.y` the backend pairs neighbouring sums up into
// v_pk_add_f32 and pays three v_mov_b32 per pair to line the halves up (12 instructions for six sums instead of 6).
CDPR_LEAF float hsum(v2f v) {
  float r;
  asm("v_add_f32_e32 %0, %1, %2" : "=v"(r) : "v"(v.x), "v"(v.y));
  return r;
}
CDPR_DEV v2f rsq2(v2f v) { return (v2f){__frsqrt_rn(v.x), __frsqrt_rn(v.y)}; }
CDPR_LEAF v2f min2(v2f a, v2f b) { return (v2f){fminf(a.x, b.x), fminf(a.y, b.y)}; }
CDPR_LEAF v2f max2(v2f a, v2f b) { return (v2f){fmaxf(a.x, b.x), fmaxf(a.y, b.y)}; }
CDPR_LEAF v2f abs2(v2f a) { return (v2f){fabsf(a.x), fabsf(a.y)}; }
CDPR_LEAF float comp4(const float4& v, int k) { return k == 0 ? v.x : (k == 1 ? v.y : (k == 2 ? v.z : v.w)); }

struct Rot {
  float r00, r01, r02, r10, r11, r12, r20, r21, r22;
};

CDPR_DEV Rot quat_to_rot(float x, float y, float z, float w) {
  const float x2 = x + x, y2 = y + y, z2 = z + z;
  const float xx = x * x2, yy = y * y2, zz = z * z2;
  const float xy = x * y2, xz = x * z2, yz = y * z2;
  const float wx = w * x2, wy = w * y2, wz = w * z2;
  Rot r;
  r.r00 = 1.f - (yy + zz);
  r.r01 = xy - wz;
  r.r02 = xz + wy;
  r.r10 = xy + wz;
  r.r11 = 1.f - (xx + zz);
  r.r12 = yz - wx;
  r.r20 = xz - wy;
  r.r21 = yz + wx;
  r.r22 = 1.f - (xx + yy);
  return r;
}

// q <- exp(theta / 2) (x) q, world-frame rotation increment, renormalised.
CDPR_DEV void quat_apply_rotvec(float& qx, float& qy, float& qz, float& qw, float tx, float ty, float tz) {
  const float a2 = fmaf(tz, tz, fmaf(ty, ty, tx * tx));
  // k = sin(|theta| / 2) / |theta|, cw = cos(|theta| / 2).  Round 5 (scripts/region_budget.py: this update was 73 vector
  // instructions per Newton iteration, 292 of a step's 3 174, against ~40 of arithmetic): |theta| = a2 * rsq(a2) and
  // k = s * rsq(a2) from ONE v_rsq_f32 instead of a correctly rounded sqrtf and a correctly rounded division (both expand to
  // ~10 instructions), and the small-angle series as a select instead of a divergent branch.  |theta| < 1e-4: series
  // (exact to fp32; also keeps rsq(0) = inf out of the result).
  const float inv_a = __frsqrt_rn(fmaxf(a2, 1e-30f));
  float s, c;
  __sincosf(0.5f * (a2 * inv_a), &s, &c);
  const bool tiny = a2 < 1e-8f;
  const float k = tiny ? fmaf(a2, -1.f / 48.f, 0.5f) : s * inv_a;
  const float cw = tiny ? fmaf(a2, -0.125f, 1.f) : c;
  const float dx = k * tx, dy = k * ty, dz = k * tz;
  const float nw = fmaf(-dz, qz, fmaf(-dy, qy, fmaf(-dx, qx, cw * qw)));
  const float nx = fmaf(-dz, qy, fmaf(dy, qz, fmaf(qw, dx, cw * qx)));
  const float ny = fmaf(-dx, qz, fmaf(dz, qx, fmaf(qw, dy, cw * qy)));
  const float nz = fmaf(-dy, qx, fmaf(dx, qy, fmaf(qw, dz, cw * qz)));
  const float inv = __frsqrt_rn(fmaf(nw, nw, fmaf(nz, nz, fmaf(ny, ny, nx * nx))));
  qx = nx * inv;
  qy = ny * inv;
  qz = nz * inv;
  qw = nw * inv;
}

// IK rows of all cable pairs (Joint::Position / GetVelocity restated; geometry statement
// gen:113-118): l = p + R b - a, L = |l|, u = l / L, J row = [u, (R b) x u].  Geometry from LDS.
template <int NP, bool ODD, bool WANT_L0>
CDPR_DEV void ik_rows(const float* lds, float px, float py, float pz, float qx, float qy, float qz, float qw,
                      v2f (&len)[NP], v2f (&jac)[NP][6], v2f (&l0)[NP]) {
  const Rot r = quat_to_rot(qx, qy, qz, qw);
#pragma unroll
  for (int k = 0; k < NP; ++k) {
    const float4 g0 = *reinterpret_cast<const float4*>(lds + k * kGeomFloatsPerPair);
    const float4 g1 = *reinterpret_cast<const float4*>(lds + k * kGeomFloatsPerPair + 4);
    const float4 g2 = *reinterpret_cast<const float4*>(lds + k * kGeomFloatsPerPair + 8);
    const v2f ax = {g0.x, g0.y}, ay = {g0.z, g0.w}, az = {g1.x, g1.y};
    const v2f bx = {g1.z, g1.w}, by = {g2.x, g2.y}, bz = {g2.z, g2.w};
    const v2f rbx = fma2(r.r02, bz, fma2(r.r01, by, splat(r.r00) * bx));
    const v2f rby = fma2(r.r12, bz, fma2(r.r11, by, splat(r.r10) * bx));
    const v2f rbz = fma2(r.r22, bz, fma2(r.r21, by, splat(r.r20) * bx));
    const v2f lx = (rbx - ax) + splat(px), ly = (rby - ay) + splat(py), lz = (rbz - az) + splat(pz);
    const v2f l2 = fma2(lz, lz, fma2(ly, ly, lx * lx));
    const v2f inv = rsq2(l2);
    len[k] = l2 * inv;
    const v2f ux = lx * inv, uy = ly * inv, uz = lz * inv;
    jac[k][0] = ux;
    jac[k][1] = uy;
    jac[k][2] = uz;
    jac[k][3] = fma2(rby, uz, -(rbz * uy));
    jac[k][4] = fma2(rbz, ux, -(rbx * uz));
    jac[k][5] = fma2(rbx, uy, -(rby * ux));
    if (WANT_L0 || (ODD && k == NP - 1)) {
      const float4 g3 = *reinterpret_cast<const float4*>(lds + k * kGeomFloatsPerPair + 12);
      l0[k] = (v2f){g3.x, g3.y};
      if (ODD && k == NP - 1) {  // odd cable count: the padding cable contributes nothing
        const v2f mask = {g3.z, g3.w};
        len[k] *= mask;
#pragma unroll
        for (int c = 0; c < 6; ++c) jac[k][c] *= mask;
      }
    }
  }
}

template <int N, bool WANT_L0>
CDPR_DEV void ik_pairs(const float* lds, float px, float py, float pz, float qx, float qy, float qz, float qw,
                       v2f (&len)[cable_pairs(N)], v2f (&jac)[cable_pairs(N)][6], v2f (&l0)[cable_pairs(N)]) {
  ik_rows<cable_pairs(N), (N & 1) != 0, WANT_L0>(lds, px, py, pz, qx, qy, qz, qw, len, jac, l0);
}

// g[c] = sum over cables of jac[.][c] * v[.]  (J^T v), pairs interleaved so the chains are independent
template <int NP>
CDPR_DEV void jt_partial(const v2f (&jac)[NP][6], const v2f (&v)[NP], v2f (&acc)[6]) {
#pragma unroll
  for (int c = 0; c < 6; ++c) acc[c] = jac[0][c] * v[0];
#pragma unroll
  for (int k = 1; k < NP; ++k) {
#pragma unroll
    for (int c = 0; c < 6; ++c) acc[c] = fma2(jac[k][c], v[k], acc[c]);
  }
}

template <int NP>
CDPR_DEV void jt_times(const v2f (&jac)[NP][6], const v2f (&v)[NP], float (&g)[6]) {
  v2f acc[6];
  jt_partial<NP>(jac, v, acc);
#pragma unroll
  for (int c = 0; c < 6; ++c) g[c] = hsum(acc[c]);
}

// Lower triangle of J^T J as 21 float2 partial sums over the cable pairs (entry e = a(a+1)/2 + b, b <= a).
template <int NP>
CDPR_DEV void gram_partial(const v2f (&jac)[NP][6], v2f (&acc)[21]) {
#pragma unroll
  for (int a = 0, e = 0; a < 6; ++a) {
#pragma unroll
    for (int b = 0; b <= a; ++b, ++e) acc[e] = jac[0][a] * jac[0][b];
  }
#pragma unroll
  for (int k = 1; k < NP; ++k) {
#pragma unroll
    for (int a = 0, e = 0; a < 6; ++a) {
#pragma unroll
      for (int b = 0; b <= a; ++b, ++e) acc[e] = fma2(jac[k][a], jac[k][b], acc[e]);
    }
  }
}

// In-place Cholesky solve of the SPD 6x6 system m x = g (lower triangle of m given, g -> x).
CDPR_DEV void chol_solve(float (&m)[6][6], float (&g)[6]) {
  float invd[6];
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    float d = m[j][j];
#pragma unroll
    for (int k = 0; k < j; ++k) d = fmaf(-m[j][k], m[j][k], d);
    invd[j] = __frsqrt_rn(d);
#pragma unroll
    for (int i = j + 1; i < 6; ++i) {
      float s = m[i][j];
#pragma unroll
      for (int k = 0; k < j; ++k) s = fmaf(-m[i][k], m[j][k], s);
      m[i][j] = s * invd[j];
    }
  }
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    float s = g[i];
#pragma unroll
    for (int k = 0; k < i; ++k) s = fmaf(-m[i][k], g[k], s);
    g[i] = s * invd[i];
  }
#pragma unroll
  for (int i = 5; i >= 0; --i) {
    float s = g[i];
#pragma unroll
    for (int k = i + 1; k < 6; ++k) s = fmaf(-m[k][i], g[k], s);
    g[i] = s * invd[i];
  }
}

#ifndef CDPR_CHOL_PACKED
#define CDPR_CHOL_PACKED 1  // 1: normal_solve uses the row-pair packed, right-looking factorization below; 0: chol_solve
#endif

// The same SPD 6x6 solve with the matrix held as ROW PAIRS: mp[c][p] = (m[2p][c], m[2p+1][c]), column c, only rows >= c
// meaningful.  Right-looking (outer-product) order: once column j is scaled, every remaining entry takes its
// -L[i][j] L[c][j] update at once, so (a) rows 2p, 2p+1 of a column update in ONE v_pk_fma_f32 with L[c][j] broadcast
// through op_sel (22 packed fmas instead of 35 plain ones, 12 packed scalings instead of 15, and the diagonal needs no
// chain of its own: it is the row-c half of the pair that holds it), and (b) the updates of one step are independent
// of each other: a single wave per SIMD pays every dependent-issue stall in full, and the left-looking form is one
// long chain per column.  Every entry still receives the same fmas in the same k order as chol_solve, so the
// results are bit-identical to it.
CDPR_DEV void chol_factor_pk(v2f (&mp)[6][3], float (&invd)[6]) {
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    const int p0 = j / 2;
    invd[j] = __frsqrt_rn((j & 1) ? mp[j][p0].y : mp[j][p0].x);
#pragma unroll
    for (int p = p0; p < 3; ++p) mp[j][p] = mp[j][p] * splat(invd[j]);  // L[.][j] (the diagonal half becomes sqrt(d): unused)
#pragma unroll
    for (int c = j + 1; c < 6; ++c) {
      const float lcj = (c & 1) ? mp[j][c / 2].y : mp[j][c / 2].x;  // L[c][j]
#pragma unroll
      for (int p = c / 2; p < 3; ++p) mp[c][p] = fma2(-lcj, mp[j][p], mp[c][p]);
    }
  }
}

// L L^T x = g with the factor of chol_factor_pk (g -> x)
CDPR_DEV void chol_apply_pk(const v2f (&mp)[6][3], const float (&invd)[6], float (&g)[6]) {
  // forward substitution, column oriented: y_j leaves as a scalar, the rows below take their update in pairs
  v2f gp[3] = {(v2f){g[0], g[1]}, (v2f){g[2], g[3]}, (v2f){g[4], g[5]}};
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    g[j] = ((j & 1) ? gp[j / 2].y : gp[j / 2].x) * invd[j];
#pragma unroll
    for (int p = (j + 1) / 2; p < 3; ++p) gp[p] = fma2(-g[j], mp[j][p], gp[p]);
  }
  // back substitution (L^T x = y): dot products along the columns of L, scalar
#pragma unroll
  for (int i = 5; i >= 0; --i) {
    float s = g[i];
#pragma unroll
    for (int k = i + 1; k < 6; ++k) s = fmaf(-((k & 1) ? mp[i][k / 2].y : mp[i][k / 2].x), g[k], s);
    g[i] = s * invd[i];
  }
}

CDPR_DEV void chol_solve_pk(v2f (&mp)[6][3], float (&g)[6]) {
  float invd[6];
  chol_factor_pk(mp, invd);
  chol_apply_pk(mp, invd, g);
}

// J^T J (+ lambda I) of a structure matrix held as cable pairs, as the row pairs chol_factor_pk takes
template <int NP, bool LAMBDA = true>
CDPR_DEV void normal_matrix_pk(const v2f (&jac)[NP][6], float lambda, v2f (&mp)[6][3]) {
  v2f acc[21];
  gram_partial<NP>(jac, acc);
#pragma unroll
  for (int a = 0, e = 0; a < 6; ++a) {
#pragma unroll
    for (int b = 0; b <= a; ++b, ++e) {
      const float v = (LAMBDA && a == b) ? hsum(acc[e]) + lambda : hsum(acc[e]);
      if (a & 1)
        mp[b][a / 2].y = v;
      else
        mp[b][a / 2].x = v;
    }
  }
  // the halves above the diagonal (row 2p < column 2p+1) are never read for a result; give them a value
  mp[1][0].x = 0.f;
  mp[3][1].x = 0.f;
  mp[5][2].x = 0.f;
}

// Solve (J^T J + lambda I) x = g in place (g -> x); J held as cable pairs.  LAMBDA = false: no damping term (the
// tension distribution), nothing is added to the diagonal.
template <int NP, bool LAMBDA = true>
CDPR_DEV void normal_solve(const v2f (&jac)[NP][6], float lambda, float (&g)[6]) {
#if CDPR_CHOL_PACKED
  v2f mp[6][3];
  normal_matrix_pk<NP, LAMBDA>(jac, lambda, mp);
  chol_solve_pk(mp, g);
#else
  v2f acc[21];
  gram_partial<NP>(jac, acc);
  float m[6][6];
#pragma unroll
  for (int a = 0, e = 0; a < 6; ++a) {
#pragma unroll
    for (int b = 0; b <= a; ++b, ++e) m[a][b] = (LAMBDA && a == b) ? hsum(acc[e]) + lambda : hsum(acc[e]);
  }
  chol_solve(m, g);
#endif
}

// The Pid a lane runs this step.  Uniform handles: the kernel arguments themselves (scalars).  PR: the velocity or the
// position Pid by the robot's own mode, one v_cndmask per coefficient.
struct PidCoef {
  float kf, kp, ki, kd, inv_ki, imax, imin, cmax, cmin;
  int nbuf;
  bool clamp;
};
template <bool PR>
CDPR_DEV PidCoef pid_coef(const StepArgs& a, bool is_vel) {
  PidCoef c;
  const bool alt = PR && !is_vel;
  c.kf = alt ? a.alt.kf : a.kf;
  c.kp = alt ? a.alt.kp : a.kp;
  c.ki = alt ? a.alt.ki : a.ki;
  c.kd = alt ? a.alt.kd : a.kd;
  c.inv_ki = alt ? a.alt.inv_ki : a.inv_ki;
  c.imax = alt ? a.alt.imax : a.imax;
  c.imin = alt ? a.alt.imin : a.imin;
  c.cmax = alt ? a.alt.cmax : a.cmax;
  c.cmin = alt ? a.alt.cmin : a.cmin;
  c.nbuf = a.nbuf;  // per-robot handles: the same window for both Pids
  c.clamp = (alt ? a.alt.clamp_cmd : a.clamp_cmd) != 0;
  return c;
}
constexpr uint32_t kMetaModeMask = 3u, kMetaCallShift = 2u, kMetaCallMax = 63u;
constexpr uint32_t kMetaForce = 0u, kMetaPosition = 1u, kMetaVelocity = 2u;  // JFC.h:35-37

struct Platform {
  float px, py, pz, qx, qy, qz, qw;
  float vx, vy, vz, wx, wy, wz;
};

// World step (Gazebo/ODE restated): semi-implicit Euler on the free platform
// under the wrench w (force, torque about the platform origin, world frame).
CDPR_DEV void integrate_pose(const StepArgs& a, Platform& s);
CDPR_DEV void integrate_velocity(const StepArgs& a, Platform& s, const float (&w)[6]) {
  const Rot r = quat_to_rot(s.qx, s.qy, s.qz, s.qw);
  s.vx = fmaf(a.dt_inv_mass, w[0], s.vx);
  s.vy = fmaf(a.dt_inv_mass, w[1], s.vy);
  s.vz = fmaf(a.dt_inv_mass, w[2], s.vz);
  // body frame: tau_b = R^T tau, w_b = R^T w
  float tbx = fmaf(r.r20, w[5], fmaf(r.r10, w[4], r.r00 * w[3]));
  float tby = fmaf(r.r21, w[5], fmaf(r.r11, w[4], r.r01 * w[3]));
  float tbz = fmaf(r.r22, w[5], fmaf(r.r12, w[4], r.r02 * w[3]));
  const float obx = fmaf(r.r20, s.wz, fmaf(r.r10, s.wy, r.r00 * s.wx));
  const float oby = fmaf(r.r21, s.wz, fmaf(r.r11, s.wy, r.r01 * s.wx));
  const float obz = fmaf(r.r22, s.wz, fmaf(r.r12, s.wy, r.r02 * s.wx));
  const float iox = fmaf(a.ib[4], obz, fmaf(a.ib[3], oby, a.ib[0] * obx));
  const float ioy = fmaf(a.ib[5], obz, fmaf(a.ib[1], oby, a.ib[3] * obx));
  const float ioz = fmaf(a.ib[2], obz, fmaf(a.ib[5], oby, a.ib[4] * obx));
  tbx -= fmaf(oby, ioz, -(obz * ioy));
  tby -= fmaf(obz, iox, -(obx * ioz));
  tbz -= fmaf(obx, ioy, -(oby * iox));
  const float abx = fmaf(a.ibinv[4], tbz, fmaf(a.ibinv[3], tby, a.ibinv[0] * tbx));
  const float aby = fmaf(a.ibinv[5], tbz, fmaf(a.ibinv[1], tby, a.ibinv[3] * tbx));
  const float abz = fmaf(a.ibinv[2], tbz, fmaf(a.ibinv[5], tby, a.ibinv[4] * tbx));
  s.wx = fmaf(a.dt, fmaf(r.r02, abz, fmaf(r.r01, aby, r.r00 * abx)), s.wx);
  s.wy = fmaf(a.dt, fmaf(r.r12, abz, fmaf(r.r11, aby, r.r10 * abx)), s.wy);
  s.wz = fmaf(a.dt, fmaf(r.r22, abz, fmaf(r.r21, aby, r.r20 * abx)), s.wz);
}

// second half of the world step: p+ = p + dt v+, q+ = normalize(q + dt/2 [w+, 0] (x) q)
CDPR_DEV void integrate_pose(const StepArgs& a, Platform& s) {
  s.px = fmaf(a.dt, s.vx, s.px);
  s.py = fmaf(a.dt, s.vy, s.py);
  s.pz = fmaf(a.dt, s.vz, s.pz);
  const float h = a.half_dt;
  const float nx = fmaf(h, fmaf(-s.wz, s.qy, fmaf(s.wy, s.qz, s.qw * s.wx)), s.qx);
  const float ny = fmaf(h, fmaf(-s.wx, s.qz, fmaf(s.wz, s.qx, s.qw * s.wy)), s.qy);
  const float nz = fmaf(h, fmaf(-s.wy, s.qx, fmaf(s.wx, s.qy, s.qw * s.wz)), s.qz);
  const float nw = fmaf(-h, fmaf(s.wz, s.qz, fmaf(s.wy, s.qy, s.wx * s.qx)), s.qw);
  const float inv = __frsqrt_rn(fmaf(nw, nw, fmaf(nz, nz, fmaf(ny, ny, nx * nx))));
  s.qx = nx * inv;
  s.qy = ny * inv;
  s.qz = nz * inv;
  s.qw = nw * inv;
}

CDPR_DEV void integrate(const StepArgs& a, Platform& s, const float (&w)[6]) {
  integrate_velocity(a, s, w);
  integrate_pose(a, s);
}

// Travel limits of the prismatic joints (cube.sdf:436-437): bit i set where joint i's position lies outside [lo, hi].
template <int N>
CDPR_DEV uint32_t travel_mask(const StepArgs& a, const v2f (&q)[cable_pairs(N)]) {
  uint32_t m = 0u;
  if (a.travel_on) {
#pragma unroll
    for (int k = 0; k < cable_pairs(N); ++k) {
      m |= (q[k].x < a.travel_lo || q[k].x > a.travel_hi) ? (1u << (2 * k)) : 0u;
      if (2 * k + 1 < N) m |= (q[k].y < a.travel_lo || q[k].y > a.travel_hi) ? (1u << (2 * k + 1)) : 0u;
    }
  }
  return m;
}
// observable slot 3, component w: tension-distribution flag in bit 0, travel-limit mask above it (<= 511: exact in a float)
CDPR_DEV float pack_flags(int td_flag, uint32_t limit_mask) { return (float)((uint32_t)td_flag | (limit_mask << 1)); }

// The joint stop itself ([EXT] Gazebo/ODE -> reduced; cdpr_config_t.travel_stop), between the velocity and the pose half
// of the world step: a joint at or beyond a limit that still moves outward takes the impulse that brings its rate to
// zero, lambda = qdot_i / (J_i M^-1 J_i^T), twist += M^-1 J_i^T lambda, M the platform's own mass and inertia; cables in
// index order, a.travel_stop sweeps (projected Gauss-Seidel; with several joints on their stops one sweep lets them creep).
// q and jac are those of the state at t_k (the pose has not moved yet).
template <int N>
CDPR_DEV void apply_travel_stop(const StepArgs& a, Platform& s, const v2f (&q)[cable_pairs(N)], const v2f (&jac)[cable_pairs(N)][6]) {
  const Rot r = quat_to_rot(s.qx, s.qy, s.qz, s.qw);
  // (the twist in locals over the sweeps.  Round 5 note: in most PHYS instantiations LLVM leaves the slice {vx, vy, vz, wx} of
  //  the Platform struct in scratch memory (16 B + 8 B per lane; with a stack object in the frame the spilled scalars go to
  //  memory as well).  Neither these locals nor -fno-slp-vectorize change that; the instantiation decides (n = 3, 4 with
  //  ROLLOUT and the general kernels at n = 4, 7, 8 are free of it).  Open: profiles/r05_resource_usage.txt.)
  float vx = s.vx, vy = s.vy, vz = s.vz, wx = s.wx, wy = s.wy, wz = s.wz;
  for (int sweep = 0; sweep < a.travel_stop; ++sweep)
#pragma unroll
  for (int i = 0; i < N; ++i) {
    const int k = i / 2;
    float j[6];
#pragma unroll
    for (int c = 0; c < 6; ++c) j[c] = (i & 1) ? jac[k][c].y : jac[k][c].x;
    const float qi = (i & 1) ? q[k].y : q[k].x;
    const float qdn = -fmaf(j[5], wz, fmaf(j[4], wy, fmaf(j[3], wx, fmaf(j[2], vz, fmaf(j[1], vy, j[0] * vx)))));
    const bool hit = (qi >= a.travel_hi && qdn > 0.f) || (qi <= a.travel_lo && qdn < 0.f);
    // Iw^-1 (rb x u) = R Ib^-1 R^T (rb x u)
    const float tbx = fmaf(r.r20, j[5], fmaf(r.r10, j[4], r.r00 * j[3]));
    const float tby = fmaf(r.r21, j[5], fmaf(r.r11, j[4], r.r01 * j[3]));
    const float tbz = fmaf(r.r22, j[5], fmaf(r.r12, j[4], r.r02 * j[3]));
    const float abx = fmaf(a.ibinv[4], tbz, fmaf(a.ibinv[3], tby, a.ibinv[0] * tbx));
    const float aby = fmaf(a.ibinv[5], tbz, fmaf(a.ibinv[1], tby, a.ibinv[3] * tbx));
    const float abz = fmaf(a.ibinv[2], tbz, fmaf(a.ibinv[5], tby, a.ibinv[4] * tbx));
    const float awx = fmaf(r.r02, abz, fmaf(r.r01, aby, r.r00 * abx));
    const float awy = fmaf(r.r12, abz, fmaf(r.r11, aby, r.r10 * abx));
    const float awz = fmaf(r.r22, abz, fmaf(r.r21, aby, r.r20 * abx));
    const float d = fmaf(j[5], awz, fmaf(j[4], awy, fmaf(j[3], awx, fmaf(j[2], j[2], fmaf(j[1], j[1], j[0] * j[0])) * a.inv_mass)));
    const float lam = hit ? qdn / d : 0.f;
    const float lm = lam * a.inv_mass;
    vx = fmaf(lm, j[0], vx);
    vy = fmaf(lm, j[1], vy);
    vz = fmaf(lm, j[2], vz);
    wx = fmaf(lam, awx, wx);
    wy = fmaf(lam, awy, wy);
    wz = fmaf(lam, awz, wz);
  }
  s.vx = vx, s.vy = vy, s.vz = vz, s.wx = wx, s.wy = wy, s.wz = wz;
}

// World step with the lumped legs ([EXT] -> reduced; closed forms in DESIGN.md section 1).  Leg i turns about its frame anchor with angular velocity (u x vP)/L, vP = v + omega x rb.  The
// passive joint dampers (universal pair at the frame, spherical triple at the platform, c each) act on the platform
// through a transverse force at the anchor, Fd = -(c/L)(2 vt/L - omega x u) (massless-leg torque balance), plus the
// spherical joint's torque c((u x vP)/L - omega); the links' inertia appears at the anchor as the apparent mass
// A = alpha I + beta u u^T (alpha = J_leg/L^2 + m_pt, beta = m_ax - J_leg/L^2), so the 6x6 mass matrix is
// M0 + sum alpha_i [[I, -[rb]x], [[rb]x, |rb|^2 I - rb rb^T]] + sum beta_i J_i J_i^T (J_i = [u, rb x u], the structure
// matrix row).  w comes in as the cable + gravity wrench and leaves untouched; velocity-product terms of the legs are
// neglected.  Two cables per instruction, like everything per-cable here.
template <int N>
CDPR_DEV void integrate_lumped_velocity(const StepArgs& a, const float* lds, Platform& s, const v2f (&jac)[cable_pairs(N)][6],
                                        const v2f (&len)[cable_pairs(N)], float (&w)[6]) {
  constexpr int NP = cable_pairs(N);
  const Rot r = quat_to_rot(s.qx, s.qy, s.qz, s.qw);
  v2f fx = splat(0.f), fy = splat(0.f), fz = splat(0.f), tx = splat(0.f), ty = splat(0.f), tz = splat(0.f);
  v2f sa = splat(0.f), sax = splat(0.f), say = splat(0.f), saz = splat(0.f);      // sum alpha, sum alpha rb
  v2f ixx = splat(0.f), iyy = splat(0.f), izz = splat(0.f), ixy = splat(0.f), ixz = splat(0.f), iyz = splat(0.f);  // sum alpha (|rb|^2 I - rb rb^T)
  v2f gram[21];
#pragma unroll
  for (int e = 0; e < 21; ++e) gram[e] = splat(0.f);
#pragma unroll
  for (int k = 0; k < NP; ++k) {
    const float4 g1 = *reinterpret_cast<const float4*>(lds + k * kGeomFloatsPerPair + 4);
    const float4 g2 = *reinterpret_cast<const float4*>(lds + k * kGeomFloatsPerPair + 8);
    const float4 g3 = *reinterpret_cast<const float4*>(lds + k * kGeomFloatsPerPair + 12);
    const v2f mask = ((N & 1) && k == NP - 1) ? (v2f){g3.z, g3.w} : splat(1.f);  // the padding cable of an odd count
    const v2f bx = {g1.z, g1.w}, by = {g2.x, g2.y}, bz = {g2.z, g2.w};
    const v2f rbx = fma2(r.r02, bz, fma2(r.r01, by, splat(r.r00) * bx));
    const v2f rby = fma2(r.r12, bz, fma2(r.r11, by, splat(r.r10) * bx));
    const v2f rbz = fma2(r.r22, bz, fma2(r.r21, by, splat(r.r20) * bx));
    const v2f ux = jac[k][0], uy = jac[k][1], uz = jac[k][2];
    const v2f l = (mask.x == 0.f || mask.y == 0.f) ? (v2f){len[k].x + (1.f - mask.x), len[k].y + (1.f - mask.y)} : len[k];  // padded lane: L = 1
    const v2f il = (v2f){__frcp_rn(l.x), __frcp_rn(l.y)};
    // anchor velocity vP = v + omega x rb, its part across the cable, omega x u, u x vP
    const v2f vpx = fma2(s.wy, rbz, fma2(-s.wz, rby, splat(s.vx)));
    const v2f vpy = fma2(s.wz, rbx, fma2(-s.wx, rbz, splat(s.vy)));
    const v2f vpz = fma2(s.wx, rby, fma2(-s.wy, rbx, splat(s.vz)));
    const v2f along = fma2(uz, vpz, fma2(uy, vpy, ux * vpx));
    const v2f vtx = fma2(-along, ux, vpx), vty = fma2(-along, uy, vpy), vtz = fma2(-along, uz, vpz);
    const v2f oux = fma2(s.wy, uz, -(splat(s.wz) * uy)), ouy = fma2(s.wz, ux, -(splat(s.wx) * uz)), ouz = fma2(s.wx, uy, -(splat(s.wy) * ux));
    const v2f uvx = fma2(uy, vpz, -(uz * vpy)), uvy = fma2(uz, vpx, -(ux * vpz)), uvz = fma2(ux, vpy, -(uy * vpx));
    const v2f cl = splat(-a.ph_c) * il * mask;  // -(c / L)
    const v2f two_il = il + il;
    const v2f fdx = cl * fma2(two_il, vtx, -oux), fdy = cl * fma2(two_il, vty, -ouy), fdz = cl * fma2(two_il, vtz, -ouz);
    // weight of the point masses at the anchor
    const v2f pm = splat(a.ph_mpt) * mask;
    const v2f fax = fma2(a.gx, pm, fdx), fay = fma2(a.gy, pm, fdy), faz = fma2(a.gz, pm, fdz);
    fx += fax;
    fy += fay;
    fz += faz;
    const v2f cm = splat(a.ph_c) * mask;
    tx += fma2(rby, faz, -(rbz * fay)) + cm * fma2(uvx, il, splat(-s.wx));
    ty += fma2(rbz, fax, -(rbx * faz)) + cm * fma2(uvy, il, splat(-s.wy));
    tz += fma2(rbx, fay, -(rby * fax)) + cm * fma2(uvz, il, splat(-s.wz));
    // apparent mass of the leg at the anchor
    const v2f mu = splat(a.ph_jleg) * il * il;
    const v2f alpha = (mu + splat(a.ph_mpt)) * mask, beta = (splat(a.ph_max) - mu) * mask;
    sa += alpha;
    sax = fma2(alpha, rbx, sax);
    say = fma2(alpha, rby, say);
    saz = fma2(alpha, rbz, saz);
    const v2f rb2 = fma2(rbz, rbz, fma2(rby, rby, rbx * rbx));
    ixx = fma2(alpha, fma2(-rbx, rbx, rb2), ixx);
    iyy = fma2(alpha, fma2(-rby, rby, rb2), iyy);
    izz = fma2(alpha, fma2(-rbz, rbz, rb2), izz);
    ixy = fma2(-alpha, rbx * rby, ixy);
    ixz = fma2(-alpha, rbx * rbz, ixz);
    iyz = fma2(-alpha, rby * rbz, iyz);
    v2f bj[6];
#pragma unroll
    for (int c = 0; c < 6; ++c) bj[c] = beta * jac[k][c];
#pragma unroll
    for (int p = 0, e = 0; p < 6; ++p) {
#pragma unroll
      for (int q = 0; q <= p; ++q, ++e) gram[e] = fma2(bj[p], jac[k][q], gram[e]);
    }
  }
  // world inertia R Ib R^T (+ the share of every leg that turns with the platform), gyroscopic torque
  const float b00 = a.ib[0], b11 = a.ib[1], b22 = a.ib[2], b01 = a.ib[3], b02 = a.ib[4], b12 = a.ib[5];
  const float c00 = fmaf(r.r02, b02, fmaf(r.r01, b01, r.r00 * b00)), c01 = fmaf(r.r02, b12, fmaf(r.r01, b11, r.r00 * b01)), c02 = fmaf(r.r02, b22, fmaf(r.r01, b12, r.r00 * b02));
  const float c10 = fmaf(r.r12, b02, fmaf(r.r11, b01, r.r10 * b00)), c11 = fmaf(r.r12, b12, fmaf(r.r11, b11, r.r10 * b01)), c12 = fmaf(r.r12, b22, fmaf(r.r11, b12, r.r10 * b02));
  const float c20 = fmaf(r.r22, b02, fmaf(r.r21, b01, r.r20 * b00)), c21 = fmaf(r.r22, b12, fmaf(r.r21, b11, r.r20 * b01)), c22 = fmaf(r.r22, b22, fmaf(r.r21, b12, r.r20 * b02));
  const float w00 = fmaf(c02, r.r02, fmaf(c01, r.r01, c00 * r.r00)) + a.ph_iadd_total, w11 = fmaf(c12, r.r12, fmaf(c11, r.r11, c10 * r.r10)) + a.ph_iadd_total,
              w22 = fmaf(c22, r.r22, fmaf(c21, r.r21, c20 * r.r20)) + a.ph_iadd_total;
  const float w01 = fmaf(c02, r.r12, fmaf(c01, r.r11, c00 * r.r10)), w02 = fmaf(c02, r.r22, fmaf(c01, r.r21, c00 * r.r20)),
              w12 = fmaf(c12, r.r22, fmaf(c11, r.r21, c10 * r.r20));
  const float iox = fmaf(w02, s.wz, fmaf(w01, s.wy, w00 * s.wx)), ioy = fmaf(w12, s.wz, fmaf(w11, s.wy, w01 * s.wx)), ioz = fmaf(w22, s.wz, fmaf(w12, s.wy, w02 * s.wx));
  float rhs[6] = {w[0] + hsum(fx), w[1] + hsum(fy), w[2] + hsum(fz), w[3] + hsum(tx) - fmaf(s.wy, ioz, -(s.wz * ioy)),
                  w[4] + hsum(ty) - fmaf(s.wz, iox, -(s.wx * ioz)), w[5] + hsum(tz) - fmaf(s.wx, ioy, -(s.wy * iox))};
  // mass matrix, lower triangle; [[rb]x]: rows (0,-z,y), (z,0,-x), (-y,x,0): lower-left block = +alpha [rb]x
  const float A = hsum(sa), ax = hsum(sax), ay = hsum(say), az = hsum(saz);
  const float mass = a.ph_mass;
  float m[6][6];
#pragma unroll
  for (int p = 0, e = 0; p < 6; ++p) {
#pragma unroll
    for (int q = 0; q <= p; ++q, ++e) m[p][q] = hsum(gram[e]);
  }
  m[0][0] += mass + A;
  m[1][1] += mass + A;
  m[2][2] += mass + A;
  m[3][1] += -az;  // row 3 of [rb]x summed: (0, -z, y)
  m[3][2] += ay;
  m[4][0] += az;   // (z, 0, -x)
  m[4][2] += -ax;
  m[5][0] += -ay;  // (-y, x, 0)
  m[5][1] += ax;
  m[3][3] += w00 + hsum(ixx);
  m[4][4] += w11 + hsum(iyy);
  m[5][5] += w22 + hsum(izz);
  m[4][3] += w01 + hsum(ixy);
  m[5][3] += w02 + hsum(ixz);
  m[5][4] += w12 + hsum(iyz);
  chol_solve(m, rhs);
  s.vx = fmaf(a.dt, rhs[0], s.vx);
  s.vy = fmaf(a.dt, rhs[1], s.vy);
  s.vz = fmaf(a.dt, rhs[2], s.vz);
  s.wx = fmaf(a.dt, rhs[3], s.wx);
  s.wy = fmaf(a.dt, rhs[4], s.wy);
  s.wz = fmaf(a.dt, rhs[5], s.wz);
}

// Slot row `slot` of robot at byte offset `off` (= 16 * robot, 32-bit): SGPR row base + one shared VGPR
// offset, so 30 rows cost one address register instead of 30 64-bit pointers.
CDPR_DEV float4 load_slot(const float4* base, size_t stride, int slot, uint32_t off) {
  return *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(base + (size_t)slot * stride) + off);
}
#ifndef CDPR_STORE_AUX
#define CDPR_STORE_AUX 2  // cache policy of the row stores: 0 plain, 2 nt, 16 sc1 (write-through), 17 sc0 sc1.
                          // Measured on MI355X, 65 536 x 8 cables, one launch per step: plain 15.84, nt 15.05,
                          // sc1 16.89, sc0 sc1 16.71 us/step -> nt
#endif
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#ifndef CDPR_STATE_STORE_PLAIN
#define CDPR_STATE_STORE_PLAIN 0  // experiment switch: 1 = state rows stored plain (kept in L2), observables keep
                                  // CDPR_STORE_AUX.  Measured: 13.52 vs 13.09 us/step (same box) -> off
#endif
CDPR_DEV void store_slot_plain(float4* base, size_t stride, int slot, uint32_t off, const float4& v) {
  *reinterpret_cast<float4*>(reinterpret_cast<char*>(base + (size_t)slot * stride) + off) = v;
}
CDPR_DEV void store_slot(float4* base, size_t stride, int slot, uint32_t off, const float4& v) {
#if CDPR_STORE_AUX == 0
  *reinterpret_cast<float4*>(reinterpret_cast<char*>(base + (size_t)slot * stride) + off) = v;
#else
  const u32x4 d = {__builtin_bit_cast(unsigned, v.x), __builtin_bit_cast(unsigned, v.y), __builtin_bit_cast(unsigned, v.z),
                   __builtin_bit_cast(unsigned, v.w)};
#ifdef CDPR_DESC_PER_BUFFER
  // diagnosis build (round 2's experiment, scripts/build_variants.sh descbuf): one descriptor per BUFFER, the row as the
  // scalar offset operand
  const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(base, 0, 0x7FFFFFFF, 0x00020000);
  __builtin_amdgcn_raw_buffer_store_b128(d, rsrc, off, (uint32_t)((size_t)slot * stride * sizeof(float4)), CDPR_STORE_AUX);
#else
  // one buffer descriptor per slot row (wave-uniform: SGPRs only), 32-bit lane offset, explicit cache policy
  const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(base + (size_t)slot * stride, 0, (int)(stride * sizeof(float4)), 0x00020000);
  __builtin_amdgcn_raw_buffer_store_b128(d, rsrc, off, 0, CDPR_STORE_AUX);
#endif
#endif
}

// Row store with the cache policy as a template argument (experiments on single rows; see CDPR_STORE_AUX for the values)
// store_slot for the lanes with `on` only, WITHOUT a branch: the row's buffer descriptor covers `stride` columns, an offset
// of all ones fails its range check and the store is dropped.  (Where a divergent `if` around the stores would be the only
// divergent branch of a region, LLVM structurizes the whole region because of it: see gen_lean_cold_tail.)
CDPR_DEV void store_slot_if(bool on, float4* base, size_t stride, int slot, uint32_t off, const float4& v) {
  const u32x4 d = {__builtin_bit_cast(unsigned, v.x), __builtin_bit_cast(unsigned, v.y), __builtin_bit_cast(unsigned, v.z),
                   __builtin_bit_cast(unsigned, v.w)};
  const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(base + (size_t)slot * stride, 0, (int)(stride * sizeof(float4)), 0x00020000);
  __builtin_amdgcn_raw_buffer_store_b128(d, rsrc, on ? off : 0xFFFFFFFFu, 0, CDPR_STORE_AUX);
}
template <int AUX>
CDPR_DEV void store_slot_aux(float4* base, size_t stride, int slot, uint32_t off, const float4& v) {
  if constexpr (AUX == 0) {
    *reinterpret_cast<float4*>(reinterpret_cast<char*>(base + (size_t)slot * stride) + off) = v;
  } else {
    const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(base + (size_t)slot * stride, 0, (int)(stride * sizeof(float4)), 0x00020000);
    const u32x4 d = {__builtin_bit_cast(unsigned, v.x), __builtin_bit_cast(unsigned, v.y), __builtin_bit_cast(unsigned, v.z),
                     __builtin_bit_cast(unsigned, v.w)};
    __builtin_amdgcn_raw_buffer_store_b128(d, rsrc, off, 0, AUX);
  }
}
#ifndef CDPR_SPLIT_PLAT_AUX
#define CDPR_SPLIT_PLAT_AUX CDPR_STORE_AUX  // cache policy of the role-split kernel's platform-row stores (the rows the NEXT launch reads first)
#endif

// win[k][slot] = e[k] for a wave-uniform ring slot, written as per-slot selects: a switch (or if-chain) over the
// slot gets merged by LLVM into ONE store through a run-time index into the register array, which sends the
// whole window to scratch memory (seen in the ISA: 256 B of scratch per lane); selects keep everything in VGPRs.
template <int NPX>
CDPR_DEV void ring_push(v2f (&win)[NPX][kWin], const v2f (&e)[NPX], int slot) {
#pragma unroll
  for (int j = 0; j < kWin; ++j) {
    const bool hit = (j == slot);
#pragma unroll
    for (int k = 0; k < NPX; ++k) {
      win[k][j].x = hit ? e[k].x : win[k][j].x;
      win[k][j].y = hit ? e[k].y : win[k][j].y;
    }
  }
}

// Ring row m of one cable pair as it lies in HBM: A[2m] B[2m] A[2m+1] B[2m+1] (each half of the float4 is one
// (A, B) register pair, so loads need no unpacking moves), with the new error e = (A, B) put into ring slot `slot`
// if that slot lies in this row.
CDPR_DEV float4 ring_row(const v2f (&w)[kWin], int m, v2f e, int slot) {
  const bool lo = (slot == 2 * m), hi = (slot == 2 * m + 1);
  return make_float4(lo ? e.x : w[2 * m].x, lo ? e.y : w[2 * m].y, hi ? e.x : w[2 * m + 1].x, hi ? e.y : w[2 * m + 1].y);
}

#ifndef CDPR_LDS_WINDOW
#define CDPR_LDS_WINDOW 1  // multi-step and rollout launches of the first-generation kernel: the derivative window of every
#endif                     // lane lives in LDS between the steps (20 KiB per wave at n = 8) instead of in 80 registers
#ifndef CDPR_EARLY_OBS
#define CDPR_EARLY_OBS 1  // first-generation kernel: pose / twist / joint position / joint velocity rows stored right after the IK
#endif
#ifndef CDPR_LPR_WAVES
#define CDPR_LPR_WAVES 1  // minimum waves per SIMD the lane-per-robot kernel is compiled for (register budget 512 / this)
#endif

// ROLLOUT = true: MPC fan-out (BASELINE config 5): one lane = one (robot, sampled command sequence); the robot's
// current state is the common start, commands change every step, state never leaves the chip.
#if CDPR_STATE_STORE_PLAIN
#define CDPR_STORE_STATE store_slot_plain
#else
#define CDPR_STORE_STATE store_slot
#endif

// The kernel's arguments as ONE iteration of a loop over blocks or steps sees them: re-read from the kernarg segment (the
// StepArgs struct is the kernel's first argument: offset 0) through a pointer the compiler cannot see through, so that
// neither the scalars nor what is computed from them once rides through the whole loop in registers.
template <bool PERSIST>
CDPR_DEV StepArgs block_args(const StepArgs& a_in) {
  if constexpr (PERSIST) {
    static_assert(sizeof(StepArgs) % 4 == 0, "StepArgs is read word by word");
    struct Words { uint32_t w[sizeof(StepArgs) / 4]; };
    typedef __attribute__((address_space(4))) const uint32_t* KArg;
    KArg kp = (KArg)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(kp));
    Words u;
#pragma unroll
    for (size_t i = 0; i < sizeof(StepArgs) / 4; ++i) u.w[i] = kp[i];
    return __builtin_bit_cast(StepArgs, u);
  } else {
    return a_in;
  }
}

// LOWREG = true (one-step kernel, large batches): fit two waves per SIMD (<= 256 registers) by NOT keeping what can be
// recomputed: the cable constants are re-read from LDS in every Newton iteration instead of being hoisted into 48
// registers, and the true structure matrix is rebuilt after the Newton stage instead of living through it.
// PHYS = true: the world step carries the lumped legs (integrate_lumped): handles created with any of
// cdpr_config_t.passive_damping / leg_inertia / cable_axial_mass / anchor_point_mass / anchor_inertia.
// PR = true: per-robot handles (StepArgs::meta): mode, Pid call count and therefore the Pid in use are per lane.
template <int N, bool FK, bool TD, bool SINGLE, bool ROLLOUT = false, bool LOWREG = false, bool PHYS = false, bool PR = false>
#ifndef CDPR_STEP_RELOAD_ARGS
#define CDPR_STEP_RELOAD_ARGS 0  // multi-step launches: re-read the kernel arguments per step (A/B)
#endif
__global__ __launch_bounds__(64, LOWREG ? 2 : CDPR_LPR_WAVES) void cdpr_step_kernel(const StepArgs a_in) {
  const StepArgs& a = a_in;  // (shadowed inside the step loop where CDPR_STEP_RELOAD_ARGS is set)
  constexpr int NP = cable_pairs(N);
  constexpr int P = plat_slots(FK);
  constexpr int G = joint_groups(N);
  __shared__ __attribute__((aligned(16))) float lds[NP * kGeomFloatsPerPair];

  const uint32_t lane = threadIdx.x;
  const uint32_t r = blockIdx.x * 64u + lane;  // robot, or trajectory index in a rollout
  const uint32_t units = ROLLOUT ? a.batch * a.roll_samples : a.batch;
  const uint32_t ru = (r < units) ? r : (units - 1u);  // tail lanes shadow the last unit, stores are masked
  const uint32_t rr = ROLLOUT ? ru / a.roll_samples : ru;  // robot whose record this lane reads
  const uint32_t sample = ROLLOUT ? ru - rr * a.roll_samples : 0u;
  const bool live = r < units;
  const size_t st = a.stride;

  // geometry load first (oldest outstanding load), then the robot's whole record: the LDS fill
  // below waits for the geometry only, the record stays in flight behind it
  CDPR_STAMP(0);
  const float gval = (lane < NP * kGeomFloatsPerPair) ? a.geom[lane] : 0.f;
  // (more than 8 cables: the geometry table is longer than the wave - 96 floats at n = 12 - and takes a second word per lane)
  float gval2 = 0.f;
  if constexpr (NP * kGeomFloatsPerPair > 64) gval2 = (lane + 64u < (uint32_t)(NP * kGeomFloatsPerPair)) ? a.geom[lane + 64u] : 0.f;

  const uint32_t off = rr * 16u;   // byte offset of this robot inside every slot row
  const uint32_t woff = r * 16u;   // same for stores (only used when live)
  const float4 p0 = load_slot(a.state, st, 0, off), p1 = load_slot(a.state, st, 1, off),
               p2 = load_slot(a.state, st, 2, off), p3 = load_slot(a.state, st, 3, off);
  float4 p4 = make_float4(0.f, 0.f, 0.f, 1.f);
  if (FK) p4 = load_slot(a.state, st, 4, off);
  constexpr int NH = (NP + 1) / 2;  // hot rows
  float4 wraw[NP][5], hraw[NH];
  {
#pragma unroll
    for (int k = 0; k < NP; ++k) {
#pragma unroll
      for (int m = 0; m < 5; ++m) wraw[k][m] = load_slot(a.state, st, P + 5 * k + m, off);
    }
#pragma unroll
    for (int g = 0; g < NH; ++g) hraw[g] = load_slot(a.state, st, P + 5 * NP + g, off);
  }
  v2f desired[NP];
#pragma unroll
  for (int k = 0; k < NP; ++k) desired[k] = splat(0.f);
  const float* vec_in = a.cmd;  // never null: before the first Joy the latched buffer holds zeros
  auto load_joy = [&](const float* cp) {  // one robot's Joy.axes (float[N]) as cable pairs
    if (N % 4 == 0) {
#pragma unroll
      for (int g = 0; g < N / 4; ++g) {
        const float4 v = reinterpret_cast<const float4*>(cp)[g];
        desired[2 * g] = (v2f){v.x, v.y};
        desired[2 * g + 1] = (v2f){v.z, v.w};
      }
    } else {
#pragma unroll
      for (int i = 0; i < N; ++i) {
        if (i & 1)
          desired[i / 2].y = cp[i];
        else
          desired[i / 2].x = cp[i];
      }
    }
  };
  if (!SINGLE && !ROLLOUT && a.sched_refresh > 0) sched_wait(a, 0);
  if (!ROLLOUT) load_joy(vec_in + (size_t)rr * N);

  if (lane < NP * kGeomFloatsPerPair) lds[lane] = gval;
  if constexpr (NP * kGeomFloatsPerPair > 64) {
    if (lane + 64u < (uint32_t)(NP * kGeomFloatsPerPair)) lds[lane + 64u] = gval2;
  }
  // single-wave workgroup: LDS operations of one wave execute in order, so the broadcast reads
  // below see the fill without an s_barrier (and without the vmcnt(0) a __syncthreads implies)
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

  Platform s;
  s.px = p0.x; s.py = p0.y; s.pz = p0.z; s.qx = p0.w;
  s.qy = p1.x; s.qz = p1.y; s.qw = p1.z; s.vx = p1.w;
  s.vy = p2.x; s.vz = p2.y; s.wx = p2.z; s.wy = p2.w;
  s.wz = p3.x;
  float fkx = p3.y, fky = p3.z, fkz = p3.w, fkqx = p4.x, fkqy = p4.y, fkqz = p4.z, fkqw = p4.w;

  // controller records as cable pairs: ring of the last 10 errors, integral
  v2f win[NP][kWin], ierr[NP];
#pragma unroll
  for (int k = 0; k < NP; ++k) {
#pragma unroll
    for (int m = 0; m < 5; ++m) {
      win[k][2 * m] = (v2f){wraw[k][m].x, wraw[k][m].y};      // register pairs as loaded: no moves
      win[k][2 * m + 1] = (v2f){wraw[k][m].z, wraw[k][m].w};
    }
    ierr[k] = (k & 1) ? (v2f){hraw[k / 2].z, hraw[k / 2].w} : (v2f){hraw[k / 2].x, hraw[k / 2].y};
  }
  // uniform handles: mode and call count are launch arguments; PR: this robot's own (per lane)
  uint32_t meta = 0u;
  if (PR) meta = a.meta[rr];
  bool actual_is_vel = PR ? ((meta & kMetaModeMask) == kMetaVelocity) : ((a.flags & kFlagActualIsVelocity) != 0u);
  int calls = PR ? (int)(meta >> kMetaCallShift) : a.pid_calls;
  float cost = 0.f, refx = 0.f, refy = 0.f, refz = 0.f;
  if (ROLLOUT) {
    refx = a.roll_ref[(size_t)rr * 3 + 0];
    refy = a.roll_ref[(size_t)rr * 3 + 1];
    refz = a.roll_ref[(size_t)rr * 3 + 2];
    if (PR) {  // a jointVelocities Joy reaches a robot in Position mode: its velocity Pid starts from reset (JFC.cpp:113-115)
      if (!actual_is_vel) {
#pragma unroll
        for (int k = 0; k < NP; ++k) ierr[k] = splat(0.f);
        calls = 0;
      }
      actual_is_vel = true;
    } else if (a.flags & kFlagRolloutResetPid) {
#pragma unroll
      for (int k = 0; k < NP; ++k) {
#pragma unroll
        for (int j = 0; j < kWin; ++j) win[k][j] = splat(0.f);
        ierr[k] = splat(0.f);
      }
      calls = 0;
    }
  }
  // Launches of several steps keep the window in LDS, one column per lane: the FIR reads it from there (20 ds_read2 per
  // step), the new error goes to its ring slot with one store per cable pair at a run-time address.  In registers the
  // window is 80 VGPRs that push the kernel past 256 (hundreds of v_accvgpr moves per step) and a ring push is 80
  // v_cndmask (the slot is a run-time value): every one of those is a vector instruction of the one wave that is
  // issue-bound, the LDS operations are not.  Same values either way.
  constexpr bool kLdsWin = CDPR_LDS_WINDOW && !SINGLE;
  __shared__ v2f lwin[kLdsWin ? kWin : 1][kLdsWin ? NP : 1][64];
  if (kLdsWin) {
#pragma unroll
    for (int j = 0; j < kWin; ++j) {
#pragma unroll
      for (int k = 0; k < NP; ++k) lwin[j][k][lane] = win[k][j];
    }
  }

  for (int step = 0; step < (SINGLE ? 1 : a_in.nsteps); ++step) {
    const StepArgs a = block_args<(CDPR_STEP_RELOAD_ARGS != 0) && !SINGLE>(a_in);
    if (!SINGLE && !ROLLOUT && a.sched_refresh > 0 && step > 0 && step % a.sched_refresh == 0) {
      // a launch over a command schedule (cdpr_update_scheduled): the next Joy batch at every refresh boundary
      const int j = step / a.sched_refresh;
      sched_wait(a, j);
      load_joy(a.cmd + (size_t)j * a.sched_stride + (size_t)rr * N);
    }
    if (ROLLOUT) {  // this step's Joy for this trajectory
      const float* cp = a.roll_cmd + (((size_t)rr * a.nsteps + step) * a.roll_samples + sample) * N;
      if (N % 4 == 0) {
#pragma unroll
        for (int g = 0; g < N / 4; ++g) {
          const float4 v = reinterpret_cast<const float4*>(cp)[g];
          desired[2 * g] = (v2f){v.x, v.y};
          desired[2 * g + 1] = (v2f){v.z, v.w};
        }
      } else {
#pragma unroll
        for (int i = 0; i < N; ++i) {
          if (i & 1)
            desired[i / 2].y = cp[i];
          else
            desired[i / 2].x = cp[i];
        }
      }
    }
    CDPR_STAMP(1);
    // ---- IK on the state at t_k
    v2f len[NP], jac[NP][6], l0[NP], q[NP], qd[NP];
    ik_pairs<N, true>(lds, s.px, s.py, s.pz, s.qx, s.qy, s.qz, s.qw, len, jac, l0);
#pragma unroll
    for (int k = 0; k < NP; ++k) {
      q[k] = l0[k] - len[k];
      qd[k] = -fma2(s.wz, jac[k][5], fma2(s.wy, jac[k][4], fma2(s.wx, jac[k][3],
                    fma2(s.vz, jac[k][2], fma2(s.vy, jac[k][1], splat(s.vx) * jac[k][0])))));
    }

#if CDPR_EARLY_OBS
    // the observables that are final already go out here (7 of 10 rows at n = 8): every wave of a launch reaches its
    // stores at the same moment, and ten rows per robot in one burst back the store path up when each step writes a NEW
    // image (trajectory record); spread over the step they drain under the Newton stage
    if (!ROLLOUT && step_published(a, step) && live) {
      float4* const obs = a.obs + (size_t)step * a.obs_step_stride;
      store_slot(obs, st, 0, woff, make_float4(s.px, s.py, s.pz, s.qx));
      store_slot(obs, st, 1, woff, make_float4(s.qy, s.qz, s.qw, s.vx));
      store_slot(obs, st, 2, woff, make_float4(s.vy, s.vz, s.wx, s.wy));
#pragma unroll
      for (int g = 0; g < G; ++g) {
        const int k0 = 2 * g, k1 = (2 * g + 1 < NP) ? 2 * g + 1 : 2 * g;
        const bool has = (2 * g + 1 < NP);
        store_slot(obs, st, 4 + g, woff, make_float4(q[k0].x, q[k0].y, has ? q[k1].x : 0.f, has ? q[k1].y : 0.f));
        store_slot(obs, st, 4 + G + g, woff, make_float4(qd[k0].x, qd[k0].y, has ? qd[k1].x : 0.f, has ? qd[k1].y : 0.f));
      }
    }
#endif
    CDPR_STAMP(2);
    // ---- per-cable force (PLG.cpp:222-228 -> JFC.cpp:59-96 -> Pid.cpp:122-191), two cables per instruction
    v2f f[NP], e_new[NP];
#pragma unroll
    for (int k = 0; k < NP; ++k) {
      f[k] = splat(0.f);
      e_new[k] = splat(0.f);
    }
    float dbg_p = 0.f, dbg_i = 0.f, dbg_d = 0.f;
    bool dbg_wrote = false;
    const bool first_world = (step == 0) && (a.flags & kFlagFirstWorldStep);
    int ring_slot = -1;  // ring slot this step's error goes to (-1: no sample taken)
    if (!first_world && !PR && (a.flags & kFlagForceMode)) {
      // UpdateMode::Force (JFC.cpp:67-70): the commanded force goes out as it is, no Pid is called
#pragma unroll
      for (int k = 0; k < NP; ++k) f[k] = desired[k];
    } else if (!first_world) {
      // PR: every lane runs the arithmetic, a robot whose Pid has just been reset (calls == 0) keeps force 0 and its integral
      if (PR || calls != 0) {  // not the first call since reset (Pid.cpp:123-126: that one returns 0)
        const PidCoef c = pid_coef<PR>(a, actual_is_vel);
        const bool is_force = PR && (meta & kMetaModeMask) == kMetaForce;  // this robot is driven open loop (JFC.cpp:67-70)
        const bool run = (!PR || calls != 0) && !is_force;
        const bool full = calls >= c.nbuf;  // derive(): 0 until the window holds nbuf samples (Pid.cpp:200-203)
        ring_slot = (a.ring_slot + step) % kWin;  // the oldest sample sits there and is overwritten below
        // weights pre-rotated for this ring position (scalar loads); per-robot handles: both Pids fit the same window
        // (cdpr_create sends anything else down the general path), so one row serves every lane whatever its mode
        const float* wt = a.wtab + ring_slot * (kWin + 2);
        v2f error[NP], acc[NP];
#pragma unroll
        for (int k = 0; k < NP; ++k) {
          error[k] = desired[k] - (actual_is_vel ? qd[k] : q[k]);
          acc[k] = splat(wt[kWin]) * error[k];
        }
        // closed-form end-point LS derivative (Pid.cpp:193-247); slot outer so the NP chains interleave
#pragma unroll
        for (int j = 0; j < kWin; ++j) {
#pragma unroll
          for (int k = 0; k < NP; ++k) acc[k] = fma2(wt[j], kLdsWin ? lwin[j][k][lane] : win[k][j], acc[k]);
        }
#pragma unroll
        for (int k = 0; k < NP; ++k) {
          const v2f p_term = splat(c.kp) * error[k];
          const v2f prev_ierr = ierr[k];
          v2f ie = fma2(a.dt, error[k], prev_ierr);
          const v2f i_term = splat(c.ki) * ie;
          const v2f i_cl = max2(min2(i_term, splat(c.imax)), splat(c.imin));  // Pid.cpp:143-152
          const v2f ie_cl = i_cl * splat(c.inv_ki);
          ie.x = (i_cl.x != i_term.x) ? ie_cl.x : ie.x;
          ie.y = (i_cl.y != i_term.y) ? ie_cl.y : ie.y;
          const v2f derived = full ? acc[k] * splat(a.inv_dt) : splat(0.f);
          const v2f d_term = splat(c.kd) * derived;
          const v2f cmd = fma2(c.kf, desired[k], p_term) + i_cl + d_term;
          v2f out = c.clamp ? max2(min2(cmd, splat(c.cmax)), splat(c.cmin)) : cmd;  // Pid.cpp:175-177
          const v2f bumped = fma2(splat(a.dt) * error[k], splat(c.ki), out);             // Pid.cpp:181-184
          ie.x = (out.x != cmd.x) ? prev_ierr.x : ie.x;
          ie.y = (out.y != cmd.y) ? prev_ierr.y : ie.y;
          out.x = (out.x != cmd.x) ? bumped.x : out.x;
          out.y = (out.y != cmd.y) ? bumped.y : out.y;
          ierr[k] = run ? ie : prev_ierr;
          f[k] = run ? out : (is_force ? desired[k] : splat(0.f));
          e_new[k] = error[k];
          if (k == 0) {
            dbg_p = p_term.x;
            dbg_i = i_term.x;
            dbg_d = d_term.x;
          }
        }
        dbg_wrote = run;
      }
      calls = PR ? min(calls + 1, (int)kMetaCallMax) : calls + 1;
    }

    if (!SINGLE && ring_slot >= 0) {
      if (kLdsWin) {
#pragma unroll
        for (int k = 0; k < NP; ++k) lwin[ring_slot][k][lane] = e_new[k];
      } else {
        ring_push<NP>(win, e_new, ring_slot);
      }
    }
    if (SINGLE && live) {
      // controller records are final: one ring row per cable pair (the one that takes the new error) + the hot rows
      if (ring_slot >= 0) {
#pragma unroll
        for (int m = 0; m < 5; ++m) {
          if (m == (ring_slot >> 1)) {
#pragma unroll
            for (int k = 0; k < NP; ++k) CDPR_STORE_STATE(a.state, st, P + 5 * k + m, woff, ring_row(win[k], m, e_new[k], ring_slot));
          }
        }
      }
#pragma unroll
      for (int g = 0; g < NH; ++g) {
        const int k1 = (2 * g + 1 < NP) ? 2 * g + 1 : 2 * g;
        CDPR_STORE_STATE(a.state, st, P + 5 * NP + g, woff, make_float4(ierr[2 * g].x, ierr[2 * g].y, ierr[k1].x, ierr[k1].y));
      }
    }

    CDPR_STAMP(3);
    // ---- optional estimator (Newton-Raphson FK, [NEW] SURVEY 8(a) row 14)
    v2f applied[NP];
    float fk_res = 0.f;
    int fk_it = 0, td_flag = 0;
    v2f jest[NP][6];
    if (FK) {
      v2f elen[NP], unused[NP];
      bool active = true;
      for (int it = 0; it < a.fk_iters; ++it) {
        uint32_t lds_off = 0;  // opaque zero offset per iteration: the geometry reads stay inside the loop (an opaque
        if (LOWREG) asm volatile("" : "+v"(lds_off));  // POINTER would lose the LDS address space and turn them into flat loads)
        const float* lds_it = lds + lds_off;
        ik_pairs<N, false>(lds_it, fkx, fky, fkz, fkqx, fkqy, fkqz, fkqw, elen, jest, unused);
        v2f res[NP];
        v2f rm = splat(0.f);
#pragma unroll
        for (int k = 0; k < NP; ++k) {
          res[k] = len[k] - elen[k];
          rm = max2(rm, abs2(res[k]));
        }
        active = active && !(fmaxf(rm.x, rm.y) < a.fk_tol);
        float g[6];
        jt_times<NP>(jest, res, g);
        normal_solve<NP>(jest, a.fk_lambda, g);
        if (active) {
          fkx += g[0];
          fky += g[1];
          fkz += g[2];
          quat_apply_rotvec(fkqx, fkqy, fkqz, fkqw, g[3], g[4], g[5]);
          ++fk_it;
        }
      }
      ik_pairs<N, false>(lds, fkx, fky, fkz, fkqx, fkqy, fkqz, fkqw, elen, jest, unused);
      v2f rm = splat(0.f);
#pragma unroll
      for (int k = 0; k < NP; ++k) rm = max2(rm, abs2(len[k] - elen[k]));
      fk_res = fmaxf(rm.x, rm.y);
    }

    CDPR_STAMP(4);
    // ---- optional tension distribution ([NEW] SURVEY 8(a) row 15):
    //      T = Tm 1 + J (J^T J)^-1 J^T (f - Tm 1), J at the FK estimate when there is one, then bounds
    if (TD) {
      v2f df[NP];
#pragma unroll
      for (int k = 0; k < NP; ++k) df[k] = f[k] - splat(a.td_mid);
      float g[6];
      if (FK) {
        jt_times<NP>(jest, df, g);
        normal_solve<NP, false>(jest, 0.f, g);
      } else {
        jt_times<NP>(jac, df, g);
        normal_solve<NP, false>(jac, 0.f, g);
      }
#pragma unroll
      for (int k = 0; k < NP; ++k) {
        v2f t = splat(a.td_mid);
#pragma unroll
        for (int c = 0; c < 6; ++c) t = fma2(g[c], FK ? jest[k][c] : jac[k][c], t);
        const v2f tc = max2(min2(t, splat(a.td_max)), splat(a.td_min));
        td_flag |= (tc.x != t.x) ? 1 : 0;
        if (2 * k + 1 < N) td_flag |= (tc.y != t.y) ? 1 : 0;
        applied[k] = tc;
      }
    } else {
#pragma unroll
      for (int k = 0; k < NP; ++k) applied[k] = f[k];
    }
    if (LOWREG && FK) {
      // rebuild the true structure matrix - and with it the joint positions and rates - HERE, behind the tension distribution:
      // none of the three is needed between the Pid and this point, so none rides through the Newton stage (q and qd were the
      // 16 registers that made this kernel spill 2-3 of its 256: 12-20 B of scratch per lane).  Same expressions on the same
      // inputs: same bits; the opaque LDS offset keeps the compiler from re-using (and so keeping alive) the first evaluation.
      uint32_t again = 0;
      asm volatile("" : "+v"(again));
      v2f len2[NP], l02[NP];
      ik_pairs<N, true>(lds + again, s.px, s.py, s.pz, s.qx, s.qy, s.qz, s.qw, len2, jac, l02);
#pragma unroll
      for (int k = 0; k < NP; ++k) {
        q[k] = l02[k] - len2[k];
        qd[k] = -fma2(s.wz, jac[k][5], fma2(s.wy, jac[k][4], fma2(s.wx, jac[k][3],
                      fma2(s.vz, jac[k][2], fma2(s.vy, jac[k][1], splat(s.vx) * jac[k][0])))));
      }
    }
    if (a.vel_limit > 0.f) {  // Joint::SetForce velocity truncation [EXT]: no pushing a runaway joint further out
#pragma unroll
      for (int k = 0; k < NP; ++k) {
        applied[k].x = (qd[k].x > a.vel_limit && applied[k].x > 0.f) || (qd[k].x < -a.vel_limit && applied[k].x < 0.f) ? 0.f : applied[k].x;
        applied[k].y = (qd[k].y > a.vel_limit && applied[k].y > 0.f) || (qd[k].y < -a.vel_limit && applied[k].y < 0.f) ? 0.f : applied[k].y;
      }
    }
    if (a.effort >= 0.f) {  // Joint::SetForce clamp (cube.sdf:438)
#pragma unroll
      for (int k = 0; k < NP; ++k) applied[k] = max2(min2(applied[k], splat(a.effort)), splat(-a.effort));
    }

    if (!ROLLOUT && a.dbg && live) {  // `pid` topic, cable 0 only (PLG.cpp:223-227; Pid.cpp:139-142,158-168)
      float* d = a.dbg + (size_t)r * 9;
      if (dbg_wrote) {
        d[0] = dbg_p;
        d[1] = dbg_i;
        d[2] = dbg_d;
        d[3] = desired[0].x;
      }
      d[4] = applied[0].x;
    }

    CDPR_STAMP(5);
    // ---- observables of step t_k (PLG.cpp:236-242, 248-280)
    if (!ROLLOUT && step_published(a, step) && live) {
      float4* const obs = a.obs + (size_t)step * a.obs_step_stride;
#if !CDPR_EARLY_OBS
      store_slot(obs, st, 0, woff, make_float4(s.px, s.py, s.pz, s.qx));
      store_slot(obs, st, 1, woff, make_float4(s.qy, s.qz, s.qw, s.vx));
      store_slot(obs, st, 2, woff, make_float4(s.vy, s.vz, s.wx, s.wy));
#endif
      store_slot(obs, st, 3, woff, make_float4(s.wz, fk_res, (float)fk_it, pack_flags(td_flag, travel_mask<N>(a, q))));
#pragma unroll
      for (int g = 0; g < G; ++g) {
        const int k0 = 2 * g, k1 = (2 * g + 1 < NP) ? 2 * g + 1 : 2 * g;
        const bool has = (2 * g + 1 < NP);
#if !CDPR_EARLY_OBS
        store_slot(obs, st, 4 + g, woff, make_float4(q[k0].x, q[k0].y, has ? q[k1].x : 0.f, has ? q[k1].y : 0.f));
        store_slot(obs, st, 4 + G + g, woff, make_float4(qd[k0].x, qd[k0].y, has ? qd[k1].x : 0.f, has ? qd[k1].y : 0.f));
#endif
        store_slot(obs, st, 4 + 2 * G + g, woff,
                   make_float4(applied[k0].x, applied[k0].y, has ? applied[k1].x : 0.f, has ? applied[k1].y : 0.f));
      }
    }

    // ---- world step to t_{k+1}: wrench = -J^T (applied - d qdot) + m g
    {
      v2f tens[NP];
#pragma unroll
      for (int k = 0; k < NP; ++k) {
        tens[k] = fma2(-a.damping, qd[k], applied[k]);
        if (a.unilateral) tens[k] = max2(tens[k], splat(0.f));  // [NEW] option: a cable cannot push
      }
      // (LOWREG: the true structure matrix was rebuilt behind the tension distribution, see there.  Rebuilding it was measured
      //  for multi-step and rollout launches too, with and without per-iteration geometry reads: 330 -> 33 v_accvgpr moves
      //  per step, yet 3-9 % slower: profiles/r02l_multistep_register_pressure_ab.txt)
      float w[6];
      jt_times<NP>(jac, tens, w);
      w[0] = a.fgx - w[0];
      w[1] = a.fgy - w[1];
      w[2] = a.fgz - w[2];
      w[3] = -w[3];
      w[4] = -w[4];
      w[5] = -w[5];
      if (PHYS) {  // lumped legs and / or the joint stop (wave-uniform choices)
        if (a.ph_lumped)
          integrate_lumped_velocity<N>(a, lds, s, jac, len, w);
        else
          integrate_velocity(a, s, w);
        if (a.travel_stop) apply_travel_stop<N>(a, s, q, jac);
        integrate_pose(a, s);
      } else {
        integrate(a, s, w);
      }
    }
    if (ROLLOUT) {
      const float ex = s.px - refx, ey = s.py - refy, ez = s.pz - refz;
      cost = fmaf(ez, ez, fmaf(ey, ey, fmaf(ex, ex, cost)));
    }
  }
  if (ROLLOUT) {
    if (live) a.roll_cost[r] = cost;
    return;
  }

  CDPR_STAMP(6);
  // ---- store
  if (PR && live) a.meta[r] = (uint8_t)((meta & kMetaModeMask) | ((uint32_t)calls << kMetaCallShift));
  if (live) {
    CDPR_STORE_STATE(a.state, st, 0, woff, make_float4(s.px, s.py, s.pz, s.qx));
    CDPR_STORE_STATE(a.state, st, 1, woff, make_float4(s.qy, s.qz, s.qw, s.vx));
    CDPR_STORE_STATE(a.state, st, 2, woff, make_float4(s.vy, s.vz, s.wx, s.wy));
    CDPR_STORE_STATE(a.state, st, 3, woff, make_float4(s.wz, fkx, fky, fkz));
    if (FK) CDPR_STORE_STATE(a.state, st, 4, woff, make_float4(fkqx, fkqy, fkqz, fkqw));
    if (!SINGLE) {
      if (kLdsWin) {
#pragma unroll
        for (int j = 0; j < kWin; ++j) {
#pragma unroll
          for (int k = 0; k < NP; ++k) win[k][j] = lwin[j][k][lane];
        }
      }
#pragma unroll
      for (int k = 0; k < NP; ++k) {
#pragma unroll
        for (int m = 0; m < 5; ++m) CDPR_STORE_STATE(a.state, st, P + 5 * k + m, woff, ring_row(win[k], m, splat(0.f), -1));
      }
#pragma unroll
      for (int g = 0; g < NH; ++g) {
        const int k1 = (2 * g + 1 < NP) ? 2 * g + 1 : 2 * g;
        CDPR_STORE_STATE(a.state, st, P + 5 * NP + g, woff, make_float4(ierr[2 * g].x, ierr[2 * g].y, ierr[k1].x, ierr[k1].y));
      }
    }
  }
  CDPR_STAMP(7);
}

}  // namespace cdpr
