// cdpr_device.hpp — per-robot device math of the batched CDPR step (gfx950, fp32).
//
// Every function works on one robot held entirely in the registers of one lane
// (lane-per-robot mapping) or on one cable of a robot (lane-per-cable mapping
// reuses the per-cable pieces).  All loops are compile-time bounded and fully
// unrolled so that the small arrays below live in VGPRs, never in scratch.
//
// Reference paths (relative to src/cdpr_gazebo/ of balazs-bamer/cdpr-simulation):
//   Pid.cpp = src/Pid.cpp, JFC.cpp = src/JointForceCalculator.cpp,
//   PLG.cpp = src/CdprGazeboPlugin.cpp, gen = sdf/gen_cdpr.py.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace cdpr {

constexpr int kMaxCables = 8;
constexpr int kWin = 10;  // prior errors kept per cable (derivative window of up to 11 samples)

// Constants shared by the whole batch; passed by value in the kernarg segment so
// they are fetched with scalar loads into SGPRs (zero HBM bytes per robot).
struct StepConsts {
  // geometry (cube.yaml:21-29; cube.sdf:310 for L0)
  float ax[kMaxCables], ay[kMaxCables], az[kMaxCables];  // frame anchors a_i
  float bx[kMaxCables], by[kMaxCables], bz[kMaxCables];  // platform anchors b_i (body frame)
  float l0[kMaxCables];                                  // cable length at joint position 0
  // world / body
  float dt, half_dt, inv_mass;
  float fgx, fgy, fgz;             // m * g, frame coords
  float ib[6];                     // body inertia  ixx iyy izz ixy ixz iyz
  float ibinv[6];                  // its inverse, same packing
  float damping, effort;           // joint damping, SetForce clamp (effort < 0: no clamp)
  // FK / TD ([NEW] stages)
  float fk_lambda, fk_tol;
  int fk_iters;
  float td_min, td_max, td_mid;
};

// The PID that is active for this launch (velocity or position mode is uniform
// over the batch: every robot receives its Joy on the same update, PLG.cpp:206-219).
struct PidConsts {
  float kf, kp, ki, kd;            // Pid.cpp:64-67
  float imax, imin, cmax, cmin;    // Pid.cpp:70-73
  float inv_dt;
  float w[kWin + 1];               // end-point LS derivative weights, oldest..newest, zero padded at the old end
  int nbuf;                        // mDbufferLength
  int clamp_cmd;                   // cmax > cmin (Pid.cpp:175)
};

struct Rot {
  float r00, r01, r02, r10, r11, r12, r20, r21, r22;
};

__device__ __forceinline__ Rot quat_to_rot(float x, float y, float z, float w) {
  Rot r;
  r.r00 = 1.f - 2.f * (y * y + z * z);
  r.r01 = 2.f * (x * y - z * w);
  r.r02 = 2.f * (x * z + y * w);
  r.r10 = 2.f * (x * y + z * w);
  r.r11 = 1.f - 2.f * (x * x + z * z);
  r.r12 = 2.f * (y * z - x * w);
  r.r20 = 2.f * (x * z - y * w);
  r.r21 = 2.f * (y * z + x * w);
  r.r22 = 1.f - 2.f * (x * x + y * y);
  return r;
}

// One cable of the IK stage (Joint::Position / GetVelocity restated; geometry
// statement gen:113-118): l = p + R b - a, L = |l|, u = l / L, J row = [u, (R b) x u].
__device__ __forceinline__ void ik_cable(const StepConsts& c, int i, float px, float py, float pz, const Rot& r,
                                         float& len, float (&row)[6]) {
  const float rbx = r.r00 * c.bx[i] + r.r01 * c.by[i] + r.r02 * c.bz[i];
  const float rby = r.r10 * c.bx[i] + r.r11 * c.by[i] + r.r12 * c.bz[i];
  const float rbz = r.r20 * c.bx[i] + r.r21 * c.by[i] + r.r22 * c.bz[i];
  const float lx = px + rbx - c.ax[i], ly = py + rby - c.ay[i], lz = pz + rbz - c.az[i];
  const float l2 = lx * lx + ly * ly + lz * lz;
  const float inv = __frsqrt_rn(l2);
  len = l2 * inv;
  const float ux = lx * inv, uy = ly * inv, uz = lz * inv;
  row[0] = ux;
  row[1] = uy;
  row[2] = uz;
  row[3] = rby * uz - rbz * uy;
  row[4] = rbz * ux - rbx * uz;
  row[5] = rbx * uy - rby * ux;
}

template <int N>
__device__ __forceinline__ void ik_all(const StepConsts& c, float px, float py, float pz, float qx, float qy, float qz,
                                       float qw, float (&len)[N], float (&jac)[N][6]) {
  const Rot r = quat_to_rot(qx, qy, qz, qw);
#pragma unroll
  for (int i = 0; i < N; ++i) ik_cable(c, i, px, py, pz, r, len[i], jac[i]);
}

// Solve (J^T J + lambda I) x = g in place (g -> x) by Cholesky; lower triangle in registers.
template <int N>
__device__ __forceinline__ void normal_solve(const float (&jac)[N][6], float lambda, float (&g)[6]) {
  float m[6][6];
#pragma unroll
  for (int a = 0; a < 6; ++a) {
#pragma unroll
    for (int b = 0; b <= a; ++b) {
      float s = (a == b) ? lambda : 0.f;
#pragma unroll
      for (int i = 0; i < N; ++i) s = fmaf(jac[i][a], jac[i][b], s);
      m[a][b] = s;
    }
  }
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

// q <- exp(theta / 2) (x) q, world-frame rotation increment, renormalised.
__device__ __forceinline__ void quat_apply_rotvec(float& qx, float& qy, float& qz, float& qw, float tx, float ty,
                                                  float tz) {
  const float a2 = tx * tx + ty * ty + tz * tz;
  float k, cw;
  if (a2 < 1e-8f) {  // |theta| < 1e-4: series (exact to fp32)
    k = 0.5f - a2 * (1.f / 48.f);
    cw = 1.f - a2 * 0.125f;
  } else {
    const float a = sqrtf(a2);
    float s;
    __sincosf(0.5f * a, &s, &cw);
    k = s / a;
  }
  const float dx = k * tx, dy = k * ty, dz = k * tz;
  const float nw = cw * qw - dx * qx - dy * qy - dz * qz;
  const float nx = cw * qx + qw * dx + dy * qz - dz * qy;
  const float ny = cw * qy + qw * dy + dz * qx - dx * qz;
  const float nz = cw * qz + qw * dz + dx * qy - dy * qx;
  const float inv = __frsqrt_rn(nx * nx + ny * ny + nz * nz + nw * nw);
  qx = nx * inv;
  qy = ny * inv;
  qz = nz * inv;
  qw = nw * inv;
}

// Newton-Raphson forward kinematics ([NEW]).  `meas` are the
// measured cable lengths; (px..qw) is the seed on entry and the estimate on exit.
// On exit jac/len hold the IK evaluation AT the estimate (reused by the tension stage).
template <int N>
__device__ __forceinline__ void fk_solve(const StepConsts& c, const float (&meas)[N], float& px, float& py, float& pz,
                                         float& qx, float& qy, float& qz, float& qw, float (&jac)[N][6],
                                         float& residual, int& iters) {
  float len[N];
  bool active = true;
  iters = 0;
  for (int it = 0; it < c.fk_iters; ++it) {
    ik_all<N>(c, px, py, pz, qx, qy, qz, qw, len, jac);
    float r[N];
    float rmax = 0.f;
#pragma unroll
    for (int i = 0; i < N; ++i) {
      r[i] = meas[i] - len[i];
      rmax = fmaxf(rmax, fabsf(r[i]));
    }
    active = active && !(rmax < c.fk_tol);
    float g[6];
#pragma unroll
    for (int a = 0; a < 6; ++a) {
      float s = 0.f;
#pragma unroll
      for (int i = 0; i < N; ++i) s = fmaf(jac[i][a], r[i], s);
      g[a] = s;
    }
    normal_solve<N>(jac, c.fk_lambda, g);
    if (active) {
      px += g[0];
      py += g[1];
      pz += g[2];
      quat_apply_rotvec(qx, qy, qz, qw, g[3], g[4], g[5]);
      ++iters;
    }
  }
  ik_all<N>(c, px, py, pz, qx, qy, qz, qw, len, jac);
  float rmax = 0.f;
#pragma unroll
  for (int i = 0; i < N; ++i) rmax = fmaxf(rmax, fabsf(meas[i] - len[i]));
  residual = rmax;
}

// Closed-form tension distribution ([NEW]) for the wrench the raw
// controller forces f would apply: T = Tm 1 + J (J^T J)^-1 J^T (f - Tm 1), then bounds.
template <int N>
__device__ __forceinline__ int td_solve(const StepConsts& c, const float (&jac)[N][6], const float (&f)[N],
                                        float (&tension)[N]) {
  float g[6];
#pragma unroll
  for (int a = 0; a < 6; ++a) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < N; ++i) s = fmaf(jac[i][a], f[i] - c.td_mid, s);
    g[a] = s;
  }
  normal_solve<N>(jac, 0.f, g);
  int flag = 0;
#pragma unroll
  for (int i = 0; i < N; ++i) {
    float t = c.td_mid;
#pragma unroll
    for (int a = 0; a < 6; ++a) t = fmaf(jac[i][a], g[a], t);
    if (t < c.td_min) {
      t = c.td_min;
      flag = 1;
    } else if (t > c.td_max) {
      t = c.td_max;
      flag = 1;
    }
    tension[i] = t;
  }
  return flag;
}

// Per-cable controller record of the fast path: the active Pid's integral, the
// count of update() calls since its last reset, and the 10 previous errors.
struct CableCtrl {
  float e[kWin];  // oldest .. newest
  float ierr;     // Pid::mIerr
  float cnt;      // update() calls since reset(): 0 => next call is the "first" one (Pid.cpp:123-126)
};

struct PidTerms {
  float p, i, d;
  bool wrote;  // false on the first call after a reset (the debug topic keeps its old values then)
};

// Pid::update (Pid.cpp:122-191) + derive (193-217) for a fixed step, with the
// polynomial fit folded into its closed-form FIR (fitPolynomial, 219-247, on a
// uniform grid).  Filters bypassed (cascade 0) on this path.
__device__ __forceinline__ float pid_update(const PidConsts& k, float dt, CableCtrl& s, float desired, float actual,
                                            PidTerms& terms) {
  float cmd_out;
  if (s.cnt == 0.f) {  // first call since reset: command 0, nothing else touched
    cmd_out = 0.f;
    terms.wrote = false;
    terms.p = terms.i = terms.d = 0.f;
  } else {
    const float error = desired - actual;
    const float p_term = k.kp * error;
    const float prev_ierr = s.ierr;
    float ierr = fmaf(dt, error, s.ierr);
    float i_term = k.ki * ierr;
    terms.p = p_term;
    terms.i = i_term;  // before the clamp, Pid.cpp:139-142
    if (i_term > k.imax) {
      i_term = k.imax;
      ierr = i_term / k.ki;
    } else if (i_term < k.imin) {
      i_term = k.imin;
      ierr = i_term / k.ki;
    }
    // derive(): window of the last nbuf errors, newest = this one
    float acc = k.w[kWin] * error;
#pragma unroll
    for (int j = 0; j < kWin; ++j) acc = fmaf(k.w[j], s.e[j], acc);
    const float derived = (s.cnt >= (float)k.nbuf) ? acc * k.inv_dt : 0.f;
#pragma unroll
    for (int j = 0; j + 1 < kWin; ++j) s.e[j] = s.e[j + 1];
    s.e[kWin - 1] = error;
    const float d_term = k.kd * derived;
    terms.d = d_term;
    terms.wrote = true;
    const float cmd = k.kf * desired + p_term + i_term + d_term;
    float out = k.clamp_cmd ? fmaxf(fminf(cmd, k.cmax), k.cmin) : cmd;
    if (out != cmd) {  // Pid.cpp:181-184 anti-windup
      ierr = prev_ierr;
      out = fmaf(dt * error, k.ki, out);
    }
    s.ierr = ierr;
    cmd_out = out;
  }
  s.cnt = fminf(s.cnt + 1.f, 1024.f);
  return cmd_out;
}

// Platform state of one robot.
struct Platform {
  float px, py, pz, qx, qy, qz, qw;
  float vx, vy, vz, wx, wy, wz;
};

// World step (Gazebo/ODE restated): wrench = -J^T T + m g with
// T_i = applied_i - d * qdot_i, semi-implicit Euler on the free platform.
template <int N>
__device__ __forceinline__ void dynamics_step(const StepConsts& c, Platform& s, const float (&jac)[N][6],
                                              const float (&applied)[N], const float (&qdot)[N]) {
  float w[6] = {c.fgx, c.fgy, c.fgz, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < N; ++i) {
    const float t = fmaf(-c.damping, qdot[i], applied[i]);
#pragma unroll
    for (int a = 0; a < 6; ++a) w[a] = fmaf(-jac[i][a], t, w[a]);
  }
  const Rot r = quat_to_rot(s.qx, s.qy, s.qz, s.qw);
  s.vx = fmaf(c.dt * c.inv_mass, w[0], s.vx);
  s.vy = fmaf(c.dt * c.inv_mass, w[1], s.vy);
  s.vz = fmaf(c.dt * c.inv_mass, w[2], s.vz);
  // body frame: tau_b = R^T tau, w_b = R^T w
  float tbx = r.r00 * w[3] + r.r10 * w[4] + r.r20 * w[5];
  float tby = r.r01 * w[3] + r.r11 * w[4] + r.r21 * w[5];
  float tbz = r.r02 * w[3] + r.r12 * w[4] + r.r22 * w[5];
  const float obx = r.r00 * s.wx + r.r10 * s.wy + r.r20 * s.wz;
  const float oby = r.r01 * s.wx + r.r11 * s.wy + r.r21 * s.wz;
  const float obz = r.r02 * s.wx + r.r12 * s.wy + r.r22 * s.wz;
  const float iox = c.ib[0] * obx + c.ib[3] * oby + c.ib[4] * obz;
  const float ioy = c.ib[3] * obx + c.ib[1] * oby + c.ib[5] * obz;
  const float ioz = c.ib[4] * obx + c.ib[5] * oby + c.ib[2] * obz;
  tbx -= oby * ioz - obz * ioy;
  tby -= obz * iox - obx * ioz;
  tbz -= obx * ioy - oby * iox;
  const float abx = c.ibinv[0] * tbx + c.ibinv[3] * tby + c.ibinv[4] * tbz;
  const float aby = c.ibinv[3] * tbx + c.ibinv[1] * tby + c.ibinv[5] * tbz;
  const float abz = c.ibinv[4] * tbx + c.ibinv[5] * tby + c.ibinv[2] * tbz;
  s.wx = fmaf(c.dt, r.r00 * abx + r.r01 * aby + r.r02 * abz, s.wx);
  s.wy = fmaf(c.dt, r.r10 * abx + r.r11 * aby + r.r12 * abz, s.wy);
  s.wz = fmaf(c.dt, r.r20 * abx + r.r21 * aby + r.r22 * abz, s.wz);
  s.px = fmaf(c.dt, s.vx, s.px);
  s.py = fmaf(c.dt, s.vy, s.py);
  s.pz = fmaf(c.dt, s.vz, s.pz);
  const float h = c.half_dt;
  const float nx = s.qx + h * (s.qw * s.wx + s.wy * s.qz - s.wz * s.qy);
  const float ny = s.qy + h * (s.qw * s.wy + s.wz * s.qx - s.wx * s.qz);
  const float nz = s.qz + h * (s.qw * s.wz + s.wx * s.qy - s.wy * s.qx);
  const float nw = s.qw - h * (s.wx * s.qx + s.wy * s.qy + s.wz * s.qz);
  const float inv = __frsqrt_rn(nx * nx + ny * ny + nz * nz + nw * nw);
  s.qx = nx * inv;
  s.qy = ny * inv;
  s.qz = nz * inv;
  s.qw = nw * inv;
}

}  // namespace cdpr
