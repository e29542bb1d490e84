// cdpr_step_kernel_f64.hpp — the step in the REFERENCE'S OWN PRECISION (gfx950, fp64): cdpr_config_t.precision = 64.
//
// The reference computes in double throughout (Pid.h members, Gazebo/ODE state); BASELINE.json's north-star allows fp32
// "to a stated tolerance", and the fp32 kernels are what the throughput figures are quoted on.  This kernel is the
// drop-in for callers who want the plugin's numbers instead: one robot (config 1) or a small batch, where the step is
// pure latency anyway and MI355X's vector fp64 issues at the plain-fp32 rate.  Same step semantics (DESIGN.md
// section 1), same Pid formulation as the fp32 fast path (the closed-form end-point least-squares derivative, weights in
// double; ring position from the world step), one lane per robot, plain scalar double arithmetic (there is no packed
// fp64), no role split, no LDS: clarity over speed.  It covers what the register-resident fp32 path covers for a
// uniform-mode handle (IK, Pid, optional Newton-Raphson FK and tension distribution, SetForce limits, velocity limit,
// unilateral cables, observables with publish decimation, travel-limit flags, world step, any number of steps per
// launch; round 5: the trajectory record and per-robot modes - PR instantiations: mode, Pid call count and so the Pid in use
// per lane, as in the fp32 PR kernels); everything else (general controller path, lumped legs, rollout) stays fp32-only and
// cdpr_create refuses the combination.
//
// HBM layout (doubles, struct of arrays over the batch, row r of the state at state[r * stride + robot]):
//   state: 0-6 pose (x y z qx qy qz qw) | 7-12 twist (v, w) | 13-19 FK estimate (pose7) | per cable i: 20 + 11 i + j,
//          j = 0..9 the ring of the last ten errors, j = 10 the integral
//   obs:   0-6 pose | 7-12 twist | 13 fk residual | 14 fk iterations | 15 flags (bit 0 TD infeasible, bits 1.. travel
//          limits) | 16 + i joint position | 16 + n + i joint velocity | 16 + 2n + i effort
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "cdpr_step_kernel.hpp"
#include "cdpr_general_step.hpp"  // gen_fit: the least-squares derivative on real stamps (the HOLD instantiations)

// Every multiply-add of this file is an explicit fma(), as in the fp32 kernels (the build runs with -ffp-contract=off so
// that all instantiations - here: the one-wave kernel in its LDS variants, its multi-step launches and the role-split
// kernel - give the same bits).  Written as a * b + c the accumulations cost twice the instructions: 165 v_mul_f64 +
// 100 v_add_f64 per cable in the Newton loop.  The CPU restatement is compiled without contraction: a few ulp apart,
// tolerances in tests/test_gpu_fp64.py.

namespace cdpr {

struct F64Args {
  double* state;
  double* obs;
  const float* cmd;     // latched Joy.axes of the active mode, float[B][n] (float32 on the wire: sensor_msgs/Joy)
  double* dbg;          // double[B][9] `pid` debug topic, or nullptr
  const double* geom;   // [n][7]: ax ay az bx by bz l0
  const double* wtab;   // [10][12]: weights pre-rotated per ring slot, as StepArgs::wtab
  uint32_t batch, stride;
  int nsteps;
  uint32_t flags;       // StepFlags
  uint64_t publish_mask;
  int pid_calls, ring_slot;
  int fk, td;           // stages
  double dt, half_dt, inv_mass, fgx, fgy, fgz;
  double ib[6], ibinv[6];
  double damping, effort, vel_limit;
  int unilateral, travel_on;
  double travel_lo, travel_hi;
  double fk_lambda, fk_tol;
  int fk_iters;
  double td_min, td_max, td_mid;
  double kf, kp, ki, kd, imax, imin, cmax, cmin, inv_dt;
  int nbuf, clamp_cmd;
  size_t obs_step_stride;  // doubles between the observable images of consecutive steps (0: every step overwrites the same image;
                           // > 0: a trajectory record keeps them all, cdpr_update_record)
  // per-robot handles (PR instantiations): meta[r] as StepArgs::meta (bits 0-1 mode, bits 2-7 Pid calls since the robot's last
  // reset, saturating); `cmd` is then the robot's ACTIVE target row, the fields above hold the VELOCITY Pid and these the
  // POSITION Pid's (the two share one derivative window on such handles: one weight table, one nbuf)
  uint8_t* meta;
  double alt_kf, alt_kp, alt_ki, alt_kd, alt_imax, alt_imin, alt_cmax, alt_cmin;
  int alt_clamp_cmd;
  // HOLD instantiations (round 5: the position-hold branch of JointForceCalculator::update, JFC.cpp:72-82, in double): mode of the
  // handle, velocityEpsilon, the two Pids' windows (the VELOCITY Pid in the primary fields, the POSITION Pid in alt_*)
  int hold_mode;  // 0 Force, 1 Position, 2 Velocity
  int step0;      // world step of the launch's first step (the Pids' stamps)
  double hold_eps;
  int degree, alt_nbuf, alt_degree;
  double hold_w[2][32];  // [position | velocity] Pid: the uniform-grid derivative weights by AGE of the sample (0: newest), in steps (kHoldWin, or kHoldWinLong of them)
  int travel_stop;             // TSTOP instantiations: sweeps of the joint stop (cdpr_config_t.travel_stop), 0 = flag only
  // TSTOP instantiations, the lumped legs (round 6; cdpr_config_t.passive_damping ...; the fp32 kernels' integrate_lumped_velocity in
  // double): any term non-zero, joint damping c of the passive revolutes, inertia turning with a leg, mass sliding along the cable,
  // point mass at each platform anchor, n x the inertia each leg adds to the platform, the platform's mass, gravity
  int ph_lumped;
  double ph_c, ph_jleg, ph_max, ph_mpt, ph_iadd_total, ph_mass, gx, gy, gz;
  // HOLD instantiations, the rest of Pid::update (round 5): the biquad cascades on the P and the D input (Pid.cpp:27-44 over
  // Filter.h:130-165; coefficients a0 a1 a2 b1 b2 of BiQuad::SetFc(relCutoff, 1, quality)) and cmd_limit = 0 (no clamp: the Pid
  // returns its stale mCmd member, Pid.cpp:175-184); [0] the POSITION Pid, [1] the VELOCITY Pid
  int any_cas, any_noclamp, max_cas;
  int pcas[2], dcas[2];
  double pcoef[2][5], dcoef[2][5];
  unsigned long long* stamps;  // diagnostic builds only (-DCDPR_STAMPS: scripts/stamp_probe_f64.py), 16 per workgroup
};

// phase boundaries of the role-split fp64 kernel (s_memrealtime, 100 MHz), lane 0 of each wave: 0.. the estimator wave's, 8.. the
// controller wave's; never compiled into the shipped library
#ifdef CDPR_STAMPS
#define CDPR_F64_STAMP(i)                                                                                             \
  do {                                                                                                                \
    __builtin_amdgcn_sched_barrier(0);                                                                                \
    if (a.stamps && lane == 0) a.stamps[(size_t)blockIdx.x * 16 + (i)] = __builtin_amdgcn_s_memrealtime();           \
    __builtin_amdgcn_sched_barrier(0);                                                                                \
  } while (0)
#define CDPR_F64_CLOCK(i)  /* the shader clock's own counter beside the 100 MHz one: cycles per microsecond over the estimator wave */ \
  do {                                                                                                                \
    __builtin_amdgcn_sched_barrier(0);                                                                                \
    if (a.stamps && lane == 0) a.stamps[(size_t)blockIdx.x * 16 + (i)] = __builtin_amdgcn_s_memtime();               \
    __builtin_amdgcn_sched_barrier(0);                                                                                \
  } while (0)
#ifdef CDPR_STAMPS_WAIT  // ... with the loads before a stamp landed (vmcnt(0)): what a phase waits for memory, at the price of the overlap
#define CDPR_F64_LANDED() __builtin_amdgcn_s_waitcnt(0x0F70)
#else
#define CDPR_F64_LANDED() do { } while (0)
#endif
#else
#define CDPR_F64_STAMP(i) do { } while (0)
#define CDPR_F64_CLOCK(i) do { } while (0)
#define CDPR_F64_LANDED() do { } while (0)
#endif

constexpr int kWinLong = 31;  // the ring of a derivative window of 12 .. 32 samples (CDPR_MAX_D_BUFFER - 1)
__host__ __device__ constexpr int f64_state_rows(int n, int w = kWin) { return 20 + (w + 1) * n; }
// HOLD handles keep BOTH Pids of every cable behind those rows: per cable mLastPosition (JFC.h:45), then per Pid (0 position,
// 1 velocity) one packed word (kHwHead ... kHwWas below) |
// mIerr | the window's values | the window's stamps (world steps, exact in a double) | mCmd | the biquads' states
constexpr int kHoldWin = kWin + 1;       // samples a HOLD record's window holds: 11 ...
constexpr int kHoldWinLong = 32;         // ... or 32 on handles with derivative windows of 12 .. 32 samples (later in round 6; one-wave kernel only)
constexpr int kHoldMaxCas = 4;                              // CDPR_MAX_CASCADE
__host__ __device__ constexpr int hold_cmd_row(int hw = kHoldWin) { return 2 + 2 * hw; }             // mCmd (read only where cmd_limit = 0 leaves it stale)
__host__ __device__ constexpr int hold_cas_row(int hw = kHoldWin) { return hold_cmd_row(hw) + 1; }   // x1 x2 y1 y2 of the P filter's stages, then of the D filter's (Filter.h:152-165)
__host__ __device__ constexpr int hold_pid_rows(int hw = kHoldWin) { return hold_cas_row(hw) + 8 * kHoldMaxCas; }
__host__ __device__ constexpr int hold_cable_rows(int hw = kHoldWin) { return 1 + 2 * hold_pid_rows(hw); }
constexpr int kHoldCmdRow = hold_cmd_row(), kHoldCasRow = hold_cas_row(), kHoldPidRows = hold_pid_rows(), kHoldCableRows = hold_cable_rows();
__host__ __device__ constexpr int f64_hold_row(int n, int cable, int pid, int hw = kHoldWin) { return f64_state_rows(n) + cable * hold_cable_rows(hw) + 1 + pid * hold_pid_rows(hw); }
__host__ __device__ constexpr int f64_hold_rows(int n, int hw = kHoldWin) { return n * hold_cable_rows(hw); }
// the packed word of a Pid record (the bits of a double, moved, never computed with): bits 0-31 mLastTime as a world step | 32-37 ring
// head | 38-43 samples in the window | 44-51 length of the newest run of consecutive steps, saturating | 52 mWasLastTime
constexpr int kHwHead = 32, kHwCount = 38, kHwRun = 44, kHwWas = 52;
__host__ __device__ constexpr int f64_obs_rows(int n) { return 16 + 3 * n; }

// 1 / sqrt(x) to double precision: v_rsq_f64 (about 26 good bits) + two Newton steps, y <- y + y e / 2 with e = 1 - x y^2
// (quadratic: 2^-26 -> 2^-51 -> rounding).  The library's sqrt() followed by a division is ~55 double instructions, this is
// 9 - and the step evaluates 9 structure matrices and 5 Cholesky factorizations per robot.  Agreement with the fp64 CPU restatement
// stays at the 1e-15 level (tests/test_gpu_fp64.py).
__device__ __forceinline__ double rsqrt64(double x) {
  // v_rsq_f64 is good to ~26 bits; ONE third-order step (y <- y + y e (1/2 + 3/8 e), e = 1 - x y^2: the error goes with e^3)
  // leaves it at rounding level in 5 operations, where two Newton steps take 8
  const double y = __builtin_amdgcn_rsq(x);
  const double e = fma(-x * y, y, 1.0);
  return fma(y * e, fma(0.375, e, 0.5), y);
}
// sqrt(x) = x * rsqrt(x), with one correction step on the product (s <- s + (x - s^2) y / 2)
__device__ __forceinline__ double sqrt_from_rsqrt64(double x, double y) {
  const double s0 = x * y;
  return fma(fma(-s0, s0, x), 0.5 * y, s0);
}

struct Rot64 {
  double r00, r01, r02, r10, r11, r12, r20, r21, r22;
};
__device__ __forceinline__ Rot64 quat_to_rot64(double x, double y, double z, double w) {
  Rot64 r;
  r.r00 = fma(-2.0, fma(y, y, z * z), 1.0);
  r.r01 = 2.0 * fma(x, y, -(z * w));
  r.r02 = 2.0 * fma(x, z, y * w);
  r.r10 = 2.0 * fma(x, y, z * w);
  r.r11 = fma(-2.0, fma(x, x, z * z), 1.0);
  r.r12 = 2.0 * fma(y, z, -(x * w));
  r.r20 = 2.0 * fma(x, z, -(y * w));
  r.r21 = 2.0 * fma(y, z, x * w);
  r.r22 = fma(-2.0, fma(x, x, y * y), 1.0);
  return r;
}

// sin(x)/x and cos(x) for the half angle of a rotation increment.  The library's double sin / cos carry a Payne-Hanek
// argument reduction that costs 2 KiB of scratch memory per lane whether it runs or not; a Newton step's half angle is
// tiny, so: halve x until it is below 0.8 (never, in practice), Taylor series in x^2 to x^20 / x^21 (remainder < 1e-21
// there), then undo the halvings with the double-angle formulas.
__device__ __forceinline__ void sinc_cos64(double x, double& sinc, double& c) {
  int halvings = 0;
  while (x > 0.8 && halvings < 16) {
    x *= 0.5;
    ++halvings;
  }
  const double z = x * x;
  // sin(x)/x = sum (-z)^k / (2k+1)!,  cos(x) = sum (-z)^k / (2k)!
  double sc = 1.0 / 51090942171709440000.0;  // 1/21!
  sc = fma(sc, -z, 1.0 / 121645100408832000.0);  // 1/19!
  sc = fma(sc, -z, 1.0 / 355687428096000.0);     // 1/17!
  sc = fma(sc, -z, 1.0 / 1307674368000.0);       // 1/15!
  sc = fma(sc, -z, 1.0 / 6227020800.0);          // 1/13!
  sc = fma(sc, -z, 1.0 / 39916800.0);            // 1/11!
  sc = fma(sc, -z, 1.0 / 362880.0);              // 1/9!
  sc = fma(sc, -z, 1.0 / 5040.0);                // 1/7!
  sc = fma(sc, -z, 1.0 / 120.0);                 // 1/5!
  sc = fma(sc, -z, 1.0 / 6.0);                   // 1/3!
  sc = fma(sc, -z, 1.0);
  double cc = 1.0 / 2432902008176640000.0;    // 1/20!
  cc = fma(cc, -z, 1.0 / 6402373705728000.0);    // 1/18!
  cc = fma(cc, -z, 1.0 / 20922789888000.0);      // 1/16!
  cc = fma(cc, -z, 1.0 / 87178291200.0);         // 1/14!
  cc = fma(cc, -z, 1.0 / 479001600.0);           // 1/12!
  cc = fma(cc, -z, 1.0 / 3628800.0);             // 1/10!
  cc = fma(cc, -z, 1.0 / 40320.0);               // 1/8!
  cc = fma(cc, -z, 1.0 / 720.0);                 // 1/6!
  cc = fma(cc, -z, 1.0 / 24.0);                  // 1/4!
  cc = fma(cc, -z, 0.5);                         // 1/2!
  cc = fma(cc, -z, 1.0);
  double sn = sc * x;
  for (int k = 0; k < halvings; ++k) {  // sin 2x = 2 s c, cos 2x = c^2 - s^2
    const double s2 = 2.0 * sn * cc, c2 = fma(cc, cc, -(sn * sn));
    sn = s2;
    cc = c2;
    x *= 2.0;
  }
  sinc = (halvings == 0) ? sc : sn / x;
  c = cc;
}

// q <- exp(theta / 2) (x) q, renormalised (world-frame rotation increment)
__device__ __forceinline__ void quat_apply_rotvec64(double (&q)[4], double tx, double ty, double tz) {
  const double a2 = fma(tz, tz, fma(ty, ty, tx * tx));
  const double a = (a2 > 0.0) ? sqrt_from_rsqrt64(a2, rsqrt64(a2)) : 0.0;
  double sinc, c;
  sinc_cos64(0.5 * a, sinc, c);
  const double k = 0.5 * sinc;  // sin(a/2) / a
  const double dx = k * tx, dy = k * ty, dz = k * tz;
  const double x = q[0], y = q[1], z = q[2], w = q[3];
  const double nw = fma(-dz, z, fma(-dy, y, fma(-dx, x, c * w)));
  const double nx = fma(-dz, y, fma(dy, z, fma(w, dx, c * x)));
  const double ny = fma(-dx, z, fma(dz, x, fma(w, dy, c * y)));
  const double nz = fma(-dy, x, fma(dx, y, fma(w, dz, c * z)));
  const double inv = rsqrt64(fma(nw, nw, fma(nz, nz, fma(ny, ny, nx * nx))));
  q[0] = nx * inv;
  q[1] = ny * inv;
  q[2] = nz * inv;
  q[3] = nw * inv;
}

// One IK row (gen:113-118): l = p + R b - a, L = |l|, u = l / L, J row = [u, (R b) x u]
__device__ __forceinline__ void ik_row64(const double* g, const Rot64& r, const double (&p)[3], double& L, double (&j)[6]) {
  const double rbx = fma(r.r02, g[5], fma(r.r01, g[4], r.r00 * g[3]));
  const double rby = fma(r.r12, g[5], fma(r.r11, g[4], r.r10 * g[3]));
  const double rbz = fma(r.r22, g[5], fma(r.r21, g[4], r.r20 * g[3]));
  const double lx = p[0] + rbx - g[0], ly = p[1] + rby - g[1], lz = p[2] + rbz - g[2];
  const double l2 = fma(lz, lz, fma(ly, ly, lx * lx));
  const double inv = rsqrt64(l2);
  L = sqrt_from_rsqrt64(l2, inv);
  const double ux = lx * inv, uy = ly * inv, uz = lz * inv;
  j[0] = ux;
  j[1] = uy;
  j[2] = uz;
  j[3] = fma(rby, uz, -(rbz * uy));
  j[4] = fma(rbz, ux, -(rbx * uz));
  j[5] = fma(rbx, uy, -(rby * ux));
}

// m x = g for the SPD 6x6 system whose lower triangle is given (g -> x), Cholesky; factorization and the two substitutions
// apart (the role-split kernel factors the tension distribution's matrix before the forces arrive), same operations in the
// same order as one piece
__device__ __forceinline__ void chol_factor64(double (&m)[6][6], double (&invd)[6]) {
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    double d = m[j][j];
#pragma unroll
    for (int k = 0; k < j; ++k) d = fma(-m[j][k], m[j][k], d);
    invd[j] = rsqrt64(d);
#pragma unroll
    for (int i = j + 1; i < 6; ++i) {
      double s = m[i][j];
#pragma unroll
      for (int k = 0; k < j; ++k) s = fma(-m[i][k], m[j][k], s);
      m[i][j] = s * invd[j];
    }
  }
}
__device__ __forceinline__ void chol_apply64(const double (&m)[6][6], const double (&invd)[6], double (&g)[6]) {
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    double s = g[i];
#pragma unroll
    for (int k = 0; k < i; ++k) s = fma(-m[i][k], g[k], s);
    g[i] = s * invd[i];
  }
#pragma unroll
  for (int i = 5; i >= 0; --i) {
    double s = g[i];
#pragma unroll
    for (int k = i + 1; k < 6; ++k) s = fma(-m[k][i], g[k], s);
    g[i] = s * invd[i];
  }
}
__device__ __forceinline__ void chol_solve64(double (&m)[6][6], double (&g)[6]) {
  double invd[6];
  chol_factor64(m, invd);
  chol_apply64(m, invd, g);
}

// Per-cable scalars of a lane live in LDS columns (one column per lane), so the loops over the cables are real loops
// with a run-time index: the structure matrix is never held as a whole, every stage rebuilds the rows it needs and
// accumulates J^T J / J^T v on the fly.  (Fully unrolled, with both structure matrices in registers, the double
// version of this kernel needed 512 registers and 2 KiB of scratch per lane.)
// Pid::update (Pid.cpp:122-191) with Pid::derive (Pid.cpp:193-217) on a Pid's own rows, in double, for the HOLD instantiations: a
// Pid is called whenever its branch of JointForceCalculator::update runs, so its window holds samples taken at any steps;
// a full window's derivative is the least-squares fit on the real stamps (gen_fit: orthogonal polynomials, any gap
// pattern); a window whose samples are one step apart (the packed word keeps the length of the newest run of consecutive
// steps) goes through the fixed end-point filter of the other instantiations, weights looked up by the sample's age.
// R: the Pid's first row of this robot, rows `st` doubles apart.  No biquad cascades, a command clamp (cdpr_create refuses
// the rest).  A lane's step is one latency chain, so the shape matters more than the count: HoldRows64 is loaded - fourteen
// independent loads, no branch between them - BEFORE the cable's IK, whose arithmetic then hides the round trip;
// hold_finish64 is straight-line up to the stores (the first call since a reset by selects); only the non-uniform window
// branches, to reload with the stamps.
struct HoldPid64 {
  double kf, kp, ki, kd, imax, imin, cmax, cmin;
  int nbuf, degree;
  bool clamp;             // cmdMax > cmdMin
  int pcas, dcas;         // stages of the two cascades
};

// Pid::CascadeFilter::update (Pid.cpp:38-44): `stages` identical biquads in series (direct form I, Filter.h:152-165), states in the
// Pid's own rows.  A runtime loop, a round trip per stage: handles with cascades only (g.any_cas), and nothing of it in the
// registers of the others.
__device__ __forceinline__ double hold_cascade64(double* F, size_t st, int stages, int max_stages, const double* c, double x, bool writes) {
#pragma clang loop unroll(disable)
  for (int s = 0; s < max_stages; ++s) {
    double* const B = F + (size_t)(4 * s) * st;
    const double x1 = B[0], x2 = B[st], y1 = B[2 * st], y2 = B[3 * st];
    const double y0 = c[0] * x + c[1] * x1 + c[2] * x2 - c[3] * y1 - c[4] * y2;
    const bool on = s < stages;
    if (on && writes) {
      B[0] = x;
      B[st] = x1;
      B[2 * st] = y0;
      B[3 * st] = y1;
    }
    x = on ? y0 : x;
  }
  return x;
}
template <int HW = kHoldWin>
struct HoldRows64T {
  unsigned long long word;
  double ierr, held;
  double y[HW];
};
using HoldRows64 = HoldRows64T<>;

template <int HW = kHoldWin>
__device__ __forceinline__ HoldRows64T<HW> hold_load64(const double* R, const double* LP, size_t st) {
  HoldRows64T<HW> h;
  h.word = (unsigned long long)__double_as_longlong(R[0]);
  h.ierr = R[st];
  h.held = LP[0];
#pragma unroll
  for (int j = 0; j < HW; ++j) h.y[j] = R[(size_t)(2 + j) * st];  // (rows beyond nbuf exist and stay zero)
  return h;
}

// The controller tables of a HOLD kernel, in LDS, filled by the wave that reads them: w[pid][age] the uniform-grid derivative weights
// (F64Args::hold_w), rot[pid][head][slot] the same by ring slot for every ring head, par[pid][..] the Pid's gains and limits
// (kf kp ki kd imax imin cmax cmin w[0] imax/ki imin/ki) behind a per-lane choice of the Pid - pid 0 the POSITION Pid (alt_*), 1 the VELOCITY Pid.
constexpr int kHoldPar = 12;
template <int HW = kHoldWin>
__device__ __forceinline__ void hold_tables_fill(const F64Args& a, uint32_t lane, double (*w)[HW], double (*rot)[HW][HW], double (*par)[kHoldPar]) {
  if (lane < 2 * HW) w[lane / HW][lane % HW] = a.hold_w[lane / HW][lane % HW];
  if (lane == 0) {
    par[1][0] = a.kf, par[1][1] = a.kp, par[1][2] = a.ki, par[1][3] = a.kd, par[1][4] = a.imax, par[1][5] = a.imin, par[1][6] = a.cmax, par[1][7] = a.cmin;
    par[0][0] = a.alt_kf, par[0][1] = a.alt_kp, par[0][2] = a.alt_ki, par[0][3] = a.alt_kd, par[0][4] = a.alt_imax, par[0][5] = a.alt_imin, par[0][6] = a.alt_cmax,
    par[0][7] = a.alt_cmin;
    par[1][8] = a.hold_w[1][0], par[0][8] = a.hold_w[0][0];
    par[1][9] = a.imax / a.ki, par[1][10] = a.imin / a.ki, par[0][9] = a.alt_imax / a.alt_ki, par[0][10] = a.alt_imin / a.alt_ki;
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  for (int e = (int)lane; e < 2 * HW * HW; e += 64) {
    const int pd = e / (HW * HW), hd = (e / HW) % HW, j = e % HW;
    const int nb = pd ? a.nbuf : a.alt_nbuf;
    int age = hd - j;
    age = age < 0 ? age + nb : age;
    rot[pd][hd][j] = (hd < nb && j < nb && j != hd) ? w[pd][min(max(age, 0), HW - 1)] : 0.0;
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// A Pid call that needs nothing but the straight line: the Pid has been called since its reset and its window, with this call's
// sample in it, is either a uniform one (the newest run of consecutive steps covers it) or still filling (derive() returns 0) -
// the conditions of hold_finish64's first-call and fit branches, negated.
__device__ __forceinline__ bool hold_steady64(unsigned long long word, int now, int nbuf) {
  const int last = (int)(uint32_t)word, count_old = (int)(word >> kHwCount) & 63, run_old = (int)(word >> kHwRun) & 255;
  const bool was = ((word >> kHwWas) & 1ull) != 0ull;
  const int count = min(count_old + 1, nbuf);
  const int run = (now == last + 1) ? min(run_old + 1, 255) : 1;
  return was && !(run < nbuf && count >= nbuf);
}
// ... and that call: Pid::update (Pid.cpp:122-191) + JointForceCalculator's mLastPosition (JFC.cpp:68,75,87) without a branch - the
// operations of hold_finish64 in its order (the integral's clamp by selects: i_term / ki of a clamped term is imax / ki or imin / ki,
// par[9], par[10]; a held cable stores the mLastPosition it read), every store unconditional: a lane beyond the batch shadows the
// last robot and writes that robot's values again.  Same bits as hold_finish64 (tests/test_gpu_fp64.py compares the two kernels).
// In two stages so that a pass of several cables can finish with the windows (eleven of a cable's fourteen registers pairs) before the
// gains arrive: hold_fast_fir64 - sample, ring position, the fixed filter's sum; hold_fast64 - the rest.
struct HoldFir64 {
  double error, dts, acc;
  int head, count, run;
};
__device__ __forceinline__ HoldFir64 hold_fast_fir64(const HoldRows64& h, const double* wrot, double w0, int nbuf, double desired, double actual, int now, double dt) {
  HoldFir64 f;
  const int last = (int)(uint32_t)h.word, head_old = (int)(h.word >> kHwHead) & 63, count_old = (int)(h.word >> kHwCount) & 63, run_old = (int)(h.word >> kHwRun) & 255;
  f.dts = (double)(now - last) * dt;
  f.error = desired - actual;
  f.head = (count_old == 0) ? 0 : ((head_old + 1 >= nbuf) ? 0 : head_old + 1);
  f.count = min(count_old + 1, nbuf);
  f.run = (now == last + 1) ? min(run_old + 1, 255) : 1;
  const double* const wr = wrot + f.head * kHoldWin;
  double acc = w0 * f.error;
#pragma unroll
  for (int j = 0; j < kHoldWin; ++j) acc = fma(wr[j], h.y[j], acc);
  f.acc = acc;
  return f;
}
__device__ __forceinline__ double hold_fast64(double* R, double* LP, size_t st, double ierr_old, double held, const HoldFir64& f, const double* pp, int nbuf, bool hold, double q, double desired,
                                              int now, double dt, double& p_out, double& i_out, double& d_out) {
  const double kf = pp[0], kp = pp[1], ki = pp[2], kd = pp[3], imax = pp[4], imin = pp[5], cmax = pp[6], cmin = pp[7], ie_max = pp[9], ie_min = pp[10];
  const double error = f.error, dts = f.dts;
  const double p_term = kp * error;
  double ie = fma(dts, error, ierr_old);
  double i_term = ki * ie;
  const double i_raw = i_term;
  const bool hi = i_term > imax, lo = !hi && i_term < imin;  // Pid.cpp:143-152
  i_term = hi ? imax : (lo ? imin : i_term);
  ie = hi ? ie_max : (lo ? ie_min : ie);
  const double derived = (f.run >= nbuf) ? f.acc / dt : 0.0;
  const double d_term = kd * derived;
  const double cmd = fma(kf, desired, p_term) + i_term + d_term;  // Pid.cpp:170
  double out = fmax(fmin(cmd, cmax), cmin);                       // Pid.cpp:175-177
  const bool sat = out != cmd;                                    // Pid.cpp:181-184
  ie = sat ? ierr_old : ie;
  out = sat ? fma(dts * error, ki, out) : out;
  const unsigned long long word =
      (unsigned long long)(uint32_t)now | (unsigned long long)f.head << kHwHead | (unsigned long long)f.count << kHwCount | (unsigned long long)f.run << kHwRun | 1ull << kHwWas;
  R[0] = __longlong_as_double((long long)word);
  R[st] = ie;
  R[(size_t)(2 + f.head) * st] = error;
  R[(size_t)(2 + kHoldWin + f.head) * st] = (double)now;
  LP[0] = hold ? held : q;
  p_out = p_term, i_out = i_raw, d_out = d_term;
  return out;
}

// FULL: the handle has a biquad cascade or a Pid without the command clamp (the HOLD = 2 instantiations); otherwise none of that
// is compiled in (its uniform branches cost the others 1.3 us per step).
template <bool FULL, int HW = kHoldWin>
__device__ __forceinline__ double hold_finish64(double* R, size_t st, const HoldRows64T<HW>& h, const double* wrot, double w0, double desired, double actual, int now, double dt,
                                                const HoldPid64& g, const F64Args& a, bool velocity_pid, bool& ran, double& p_out, double& i_out, double& d_out,
                                                bool live = true) {
  // (the coefficients stay where they are - the argument block - behind a per-lane choice of the Pid)
  const double* const pc = velocity_pid ? a.pcoef[1] : a.pcoef[0];
  const double* const dc = velocity_pid ? a.dcoef[1] : a.dcoef[0];
  const int last = (int)(uint32_t)h.word, head_old = (int)(h.word >> kHwHead) & 63, count_old = (int)(h.word >> kHwCount) & 63, run_old = (int)(h.word >> kHwRun) & 255;
  const bool was = ((h.word >> kHwWas) & 1ull) != 0ull;  // Pid.cpp:123-126: the first call since reset returns 0 (and pushes no sample)
  const double dts = (double)(now - last) * dt;
  const double error = desired - actual;
  double perr = error;
  if (FULL && a.any_cas) perr = hold_cascade64(R + (size_t)hold_cas_row(HW) * st, st, g.pcas, a.max_cas, pc, error, was && live);  // Pid.cpp:131
  const double p_term = g.kp * perr;
  double ie = fma(dts, error, h.ierr);
  double i_term = g.ki * ie;
  const double i_raw = i_term;
  if (i_term > g.imax) {  // Pid.cpp:143-152
    i_term = g.imax;
    ie = i_term / g.ki;
  } else if (i_term < g.imin) {
    i_term = g.imin;
    ie = i_term / g.ki;
  }
  // Pid::derive: push the sample (dt > 0 always: a Pid is called at most once per world step)
  const int head = (count_old == 0) ? 0 : ((head_old + 1 >= g.nbuf) ? 0 : head_old + 1);
  const int count = min(count_old + 1, g.nbuf);
  const int run = (now == last + 1) ? min(run_old + 1, 255) : 1;
  // the uniform window's weights BY SLOT for this ring head (hold_tables_fill: the weight of the slot's age, 0 for the slot the new
  // sample goes to and for slots beyond nbuf - round 6: one LDS read and one fma per slot where the age, its wrap, its clamp, the
  // look-up and two selects were 13 instructions)
  const double* const wr = wrot + head * HW;
  double acc = w0 * error;
#pragma unroll
  for (int j = 0; j < HW; ++j) acc = fma(wr[j], h.y[j], acc);
  double derived = (run >= g.nbuf) ? acc / dt : 0.0;  // mDbufferMissing != 0: derive() returns 0 (Pid.cpp:200-203)
  if (was && run < g.nbuf && count >= g.nbuf) {     // a full window with a gap in it: the fit on the real stamps
    double y[HW];
    int t[HW];
    int t_old = now;
    double stamp[HW];
#pragma unroll
    for (int j = 0; j < HW; ++j) {
      stamp[j] = R[(size_t)(2 + HW + j) * st];
      asm volatile("" : "+v"(stamp[j]));  // (all eleven in flight at once: otherwise each is sunk into its own `j != head` branch, a round trip each)
    }
#pragma unroll
    for (int j = 0; j < HW; ++j) {
      y[j] = (j == head) ? error : h.y[j];
      t[j] = (j == head) ? now : (int)stamp[j];
      t_old = (j < g.nbuf) ? min(t_old, t[j]) : t_old;
    }
    derived = gen_fit<HW, double>(y, t, g.nbuf, g.degree, now, t_old) / dt;
  }
  if (FULL && a.any_cas) derived = hold_cascade64(R + (size_t)(hold_cas_row(HW) + 4 * kHoldMaxCas) * st, st, g.dcas, a.max_cas, dc, derived, was && live);  // Pid.cpp:157
  const double d_term = g.kd * derived;
  const double cmd = fma(g.kf, desired, p_term) + i_term + d_term;  // Pid.cpp:170
  double stale = 0.0;                                                // mCmd as the last call left it: what the Pid returns without a clamp
  if (FULL && a.any_noclamp) stale = R[(size_t)hold_cmd_row(HW) * st];
  double out = (!FULL || g.clamp) ? fmax(fmin(cmd, g.cmax), g.cmin) : stale;  // Pid.cpp:175-177
  if (out != cmd) {                                                  // Pid.cpp:181-184
    ie = h.ierr;
    out = fma(dts * error, g.ki, out);
  }
  const unsigned long long word = was ? ((unsigned long long)(uint32_t)now | (unsigned long long)head << kHwHead | (unsigned long long)count << kHwCount |
                                         (unsigned long long)run << kHwRun | 1ull << kHwWas)
                                      : ((h.word & ~0xffffffffull) | (unsigned long long)(uint32_t)now | 1ull << kHwWas);
  if (live) R[0] = __longlong_as_double((long long)word);
  if (FULL && a.any_noclamp && live) R[(size_t)hold_cmd_row(HW) * st] = was ? out : 0.0;  // (the first call since a reset: mCmd = 0, Pid.cpp:125)
  if (was && live) {
    R[st] = ie;
    R[(size_t)(2 + head) * st] = error;
    R[(size_t)(2 + HW + head) * st] = (double)now;
  }
  ran = was, p_out = p_term, i_out = i_raw, d_out = d_term;
  return was ? out : 0.0;
}

// RING_LDS: the derivative rings of a lane's cables are loaded once, in one batch, into LDS columns (40 KiB per wave at
// n = 8) and written back at the end: the FIR then costs no memory round trip per cable and step (small batches: the step
// is one latency chain).  Otherwise they stay in HBM / L2 (large batches: four waves per CU need the LDS).
// JCACHE (small batches: one workgroup per CU): the structure-matrix rows are kept in private LDS columns where they are
// needed again - the rows at the true pose (IK stage) for the world step, the rows at the estimate (the Newton stage's
// closing evaluation) for both passes of the tension distribution - instead of being recomputed: 48 row evaluations per
// step instead of 72 at n = 8 (a row is ~60 fp64 instructions, a reload 6 LDS reads).  Same values, same bits.
// HOLD: the position-hold branch live (velocityEpsilon >= 0): both Pids of every cable in the rows behind the state (f64_hold_row),
// selected per cable and step as JointForceCalculator::update does (JFC.cpp:67-89); uniform-mode handles, rings in memory.
// TSTOP: the joint stop itself (cdpr_config_t.travel_stop > 0, [EXT] Gazebo/ODE -> reduced; apply_travel_stop of the fp32 kernels):
// between the velocity and the pose half of the world step a joint at or beyond a limit that
// still moves outward takes the impulse that stops it, cables in index order, travel_stop sweeps; the rows of the structure
// matrix at t_k wait in private LDS columns.
// W: prior errors kept per cable (the derivative ring).  kWin = 10 serves windows to 11 samples (shorter ones by zero weights);
// W = kWinLong = 31 (round 6: Pid.h:135 allows any mDbufferLength, the engine takes 32) serves 12 .. 32 samples on the plain
// instantiation - same code, W + 1 rows per cable instead of 11, a weight table of W rows.
// HW: samples a HOLD record's window holds (kHoldWin; kHoldWinLong later in round 6: the hold branch / cascades / cmd_limit 0 with derivative windows
// of 12 .. 32 samples - the same code over records of 32 samples and their 32 stamps)
template <int N, bool RING_LDS = false, bool JCACHE = false, bool PR = false, int HOLD = 0, bool TSTOP = false, int W = kWin, int HW = kHoldWin>  // HOLD: 0 | 1 | 2 (+ cascades, cmd_limit 0)
__global__ __launch_bounds__(64) void cdpr_step_kernel_f64(const F64Args a) {
  static_assert(HW == kHoldWin || (HOLD != 0 && W == kWin), "long HOLD records: a HOLD instantiation (the plain ring rows stay at their short length, unused)");
  static_assert(W == kWin || (!RING_LDS && !JCACHE && !HOLD), "long derivative windows: rings in memory, one Pid record per cable (with or without PR and TSTOP)");
  static_assert(!HOLD || (!RING_LDS && !JCACHE), "the hold branch: the plain instantiation (uniform modes, or PR: the mode per lane)");
  static_assert(!TSTOP || (!RING_LDS && !JCACHE), "the joint stop / the lumped legs: rings in memory (with or without PR and HOLD: the world step does not care who set the forces)");
  __shared__ double c_js[TSTOP ? N : 1][TSTOP ? 6 : 1][64];
  // cables per pass of the loops over the cables.  One lane's step is a dependent chain (rotate the anchor, length,
  // reciprocal square root, row, accumulate): with a single wave per SIMD a dependent fp64 instruction waits ~8 cycles for
  // its predecessor, and only several cables in flight fill those slots.  The role-split kernel
  // unrolls fully (14.3 us at one robot x 8 against 18.9 rolled); this kernel's LDS variants take four cables per pass
  // (fully unrolled: 512 registers and 668 B of scratch per lane); the large-batch variant, whose registers also carry the global ring addresses, gains
  // nothing beyond two (65 536 x 8: 35.3 us at 2, 36.5 at 4, 54.8 at 8 - `profiles/r04final_fp64_timing.txt`)
  constexpr int kCableUnroll = RING_LDS ? (N < 4 ? N : 4) : (N < 2 ? N : 2);
  __shared__ double c_len[N][64], c_q[N][64], c_qd[N][64], c_f[N][64], c_des[N][64], c_ierr[N][64];
  __shared__ double c_win[RING_LDS ? N : 1][RING_LDS ? W : 1][64];
  __shared__ double c_jt[JCACHE ? N : 1][JCACHE ? 6 : 1][64];  // rows at the true pose
  __shared__ double c_je[JCACHE ? N : 1][JCACHE ? 6 : 1][64];  // rows at the FK estimate
  const uint32_t lane = threadIdx.x;
  const uint32_t r = blockIdx.x * 64u + lane;
  // the geometry table (7 doubles per cable) in LDS: read through the argument pointer every row evaluation is 2 - 4 vector
  // loads of a uniform address (the compiler does not make them scalar loads: stores lie between them) and each group of
  // cables waits for them again - six evaluations per step
  // (the large-batch variant only: 65 536 x 8 27.2 -> 24.4 us; with a workgroup per CU or less the loads hide and the copy is
  //  a round trip more: one robot 14.4 -> 14.7 us)
  constexpr bool kGeomLds = !RING_LDS;
  __shared__ double c_geom_lds[kGeomLds ? N * 7 : 1];
  __shared__ double c_hold_w[HOLD ? 2 : 1][HW], c_hold_rot[HOLD ? 2 : 1][HOLD ? HW : 1][HW], c_hold_par[HOLD ? 2 : 1][kHoldPar];
  if constexpr (HOLD != 0) hold_tables_fill<HW>(a, lane, c_hold_w, c_hold_rot, c_hold_par);
  if (kGeomLds) {
    for (int g = (int)lane; g < N * 7; g += 64) c_geom_lds[g] = a.geom[g];  // (84 doubles at twelve cables)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
  const double* const c_geom = kGeomLds ? c_geom_lds : a.geom;
  if (r >= a.batch) return;  // no barrier below: lanes are independent, LDS columns are private
  const size_t st = a.stride;
  double* const S = a.state + r;
  double p[3] = {S[0 * st], S[1 * st], S[2 * st]};
  double q4[4] = {S[3 * st], S[4 * st], S[5 * st], S[6 * st]};
  double v[3] = {S[7 * st], S[8 * st], S[9 * st]}, om[3] = {S[10 * st], S[11 * st], S[12 * st]};
  double fkp[3] = {S[13 * st], S[14 * st], S[15 * st]};
  double fkq[4] = {S[16 * st], S[17 * st], S[18 * st], S[19 * st]};
  // the integrals and the commands (and, RING_LDS, the derivative rings) in one batch of loads
#pragma unroll
  for (int i = 0; i < N; ++i) {
    if (!HOLD) c_ierr[i][lane] = S[(size_t)(20 + (W + 1) * i + W) * st];
    c_des[i][lane] = (double)a.cmd[(size_t)r * N + i];
    if (RING_LDS) {
#pragma unroll
      for (int k = 0; k < W; ++k) c_win[RING_LDS ? i : 0][RING_LDS ? k : 0][lane] = S[(size_t)(20 + (W + 1) * i + k) * st];
    }
  }
  // uniform handles: mode and call count are launch arguments; PR: this robot's own (per lane)
  uint32_t meta = 0u;
  if (PR) meta = a.meta[r];
  const bool actual_is_vel = PR ? ((meta & kMetaModeMask) == kMetaVelocity) : ((a.flags & kFlagActualIsVelocity) != 0u);
  int calls = PR ? (int)(meta >> kMetaCallShift) : a.pid_calls;
  // the Pid this lane runs (PR: the velocity Pid's gains in the primary fields, the position Pid's in alt_*)
  const bool alt = PR && !actual_is_vel;
  const double kf = alt ? a.alt_kf : a.kf, kp = alt ? a.alt_kp : a.kp, ki = alt ? a.alt_ki : a.ki, kd = alt ? a.alt_kd : a.kd;
  const double imax = alt ? a.alt_imax : a.imax, imin = alt ? a.alt_imin : a.imin, cmax = alt ? a.alt_cmax : a.cmax, cmin = alt ? a.alt_cmin : a.cmin;
  const bool clamp_cmd = (alt ? a.alt_clamp_cmd : a.clamp_cmd) != 0;

  for (int step = 0; step < a.nsteps; ++step) {
    const bool first_world = (step == 0) && (a.flags & kFlagFirstWorldStep);
    const bool force_mode = PR ? ((meta & kMetaModeMask) == kMetaForce) : ((a.flags & kFlagForceMode) != 0u);  // UpdateMode::Force (JFC.cpp:67-70): no Pid
    const bool run_pid = !first_world && !force_mode && calls != 0;  // Pid.cpp:123-126: the first call since reset returns 0
    const bool full = calls >= a.nbuf;
    const int ring_slot = (a.ring_slot + step) % W;
    const double* wt = a.wtab + ring_slot * (W + 2);
    double dbg_p = 0.0, dbg_i = 0.0, dbg_d = 0.0, dbg_des = 0.0;
    bool dbg_ran = false;  // (HOLD: cable 0's Pid really ran this step)
    // ---- IK on the state at t_k and the per-cable force (PLG.cpp:222-228 -> JFC.cpp:59-96 -> Pid.cpp:122-191)
    {
      const Rot64 R = quat_to_rot64(q4[0], q4[1], q4[2], q4[3]);
#pragma clang loop unroll_count(kCableUnroll)
      for (int i = 0; i < N; ++i) {
        // (HOLD: the rows of the Pid this cable calls in this step - known from the command alone - before the IK)
        double* const LP = S + (size_t)(f64_state_rows(N) + (HOLD ? i : 0) * hold_cable_rows(HW)) * st;  // mLastPosition
        // (PR: this robot's own mode - the meta byte's kMetaForce / kMetaPosition / kMetaVelocity are 0 / 1 / 2 as hold_mode's)
        const int hmode = PR ? (int)(meta & kMetaModeMask) : a.hold_mode;
        const bool vel_branch = HOLD && hmode == 2 && fabs(c_des[i][lane]) > a.hold_eps;  // JFC.cpp:72
        double* const HR = S + (size_t)(f64_hold_row(N, HOLD ? i : 0, 0, HW) + (vel_branch ? hold_pid_rows(HW) : 0)) * st;
        HoldRows64T<HW> hrows;
        if constexpr (HOLD) hrows = hold_load64<HW>(HR, LP, st);
        double L, j[6];
        ik_row64(c_geom + i * 7, R, p, L, j);
        const double q = c_geom[i * 7 + 6] - L;
        const double qd = -fma(j[5], om[2], fma(j[4], om[1], fma(j[3], om[0], fma(j[2], v[2], fma(j[1], v[1], j[0] * v[0])))));
        c_len[i][lane] = L;
        c_q[i][lane] = q;
        c_qd[i][lane] = qd;
        if (JCACHE) {
#pragma unroll
          for (int c = 0; c < 6; ++c) c_jt[JCACHE ? i : 0][JCACHE ? c : 0][lane] = j[c];
        }
        double force = (force_mode && !first_world) ? c_des[i][lane] : 0.0;
        if constexpr (HOLD) {  // JFC.cpp:59-96 with both Pids alive
          if (!first_world) {
            const double target = c_des[i][lane];
            const bool pos_branch = hmode == 1 || (hmode == 2 && !vel_branch);
            const bool hold = hmode == 2 && !vel_branch;
            if (!hold) LP[0] = q;  // JFC.cpp:68,75,87
            if (vel_branch || pos_branch) {
              HoldPid64 g;
              const double* const pp = c_hold_par[HOLD && vel_branch ? 1 : 0];  // (LDS reads behind one per-lane address: sixteen selects before)
              g.kf = pp[0], g.kp = pp[1], g.ki = pp[2], g.kd = pp[3], g.imax = pp[4], g.imin = pp[5], g.cmax = pp[6], g.cmin = pp[7];
              g.nbuf = vel_branch ? a.nbuf : a.alt_nbuf, g.degree = vel_branch ? a.degree : a.alt_degree;
              g.clamp = (vel_branch ? a.clamp_cmd : a.alt_clamp_cmd) != 0;
              g.pcas = vel_branch ? a.pcas[1] : a.pcas[0], g.dcas = vel_branch ? a.dcas[1] : a.dcas[0];  // (selects: a dynamic index copies the arrays to scratch)
              const double desired = vel_branch ? target : (hold ? hrows.held : target);
              bool ran = false;
              double tp = 0.0, ti = 0.0, td = 0.0;
              force = hold_finish64<HOLD == 2, HW>(HR, st, hrows, &c_hold_rot[HOLD && vel_branch ? 1 : 0][0][0], pp[8], desired, vel_branch ? qd : q, (int)(a.step0 + step), a.dt, g, a, vel_branch, ran, tp, ti, td);
              if (i == 0 && ran) {
                dbg_p = tp, dbg_i = ti, dbg_d = td;
                dbg_ran = true;
                dbg_des = desired;
              }
            }
          }
        } else if (run_pid) {
          const double desired = c_des[i][lane];
          const double error = desired - (actual_is_vel ? qd : q);
          double acc = wt[W] * error;
#pragma unroll
          for (int k = 0; k < W; ++k) acc = fma(wt[k], RING_LDS ? c_win[RING_LDS ? i : 0][RING_LDS ? k : 0][lane] : S[(size_t)(20 + (W + 1) * i + k) * st], acc);
          const double p_term = kp * error;
          const double prev_ierr = c_ierr[i][lane];
          double ie = fma(a.dt, error, prev_ierr);
          double i_term = ki * ie;
          const double i_raw = i_term;
          if (i_term > imax) {  // Pid.cpp:143-152
            i_term = imax;
            ie = i_term / ki;
          } else if (i_term < imin) {
            i_term = imin;
            ie = i_term / ki;
          }
          const double derived = full ? acc * a.inv_dt : 0.0;
          const double d_term = kd * derived;
          const double cmd = fma(kf, desired, p_term) + i_term + d_term;
          double out = clamp_cmd ? fmax(fmin(cmd, cmax), cmin) : cmd;  // Pid.cpp:175-177
          if (out != cmd) {                                              // Pid.cpp:181-184
            ie = prev_ierr;
            out = fma(a.dt * error, ki, out);
          }
          c_ierr[i][lane] = ie;
          force = out;
          if (RING_LDS) c_win[RING_LDS ? i : 0][RING_LDS ? ring_slot : 0][lane] = error;  // (the next steps of this launch read it there)
          S[(size_t)(20 + (W + 1) * i + ring_slot) * st] = error;  // after the FIR has read the slot's old content; the slot, not the ring at the end
          if (i == 0) {
            dbg_p = p_term;
            dbg_i = i_raw;
            dbg_d = d_term;
          }
        }
        c_f[i][lane] = force;
      }
      // (PR: the count saturates where the meta byte does; a robot in Force mode calls no Pid)
      if (!first_world) calls = PR ? (force_mode ? calls : min(calls + 1, (int)kMetaCallMax)) : calls + 1;
    }
    // ---- Newton-Raphson forward kinematics ([NEW] SURVEY 8(a) row 14)
    double fk_res = 0.0;
    int fk_it = 0, td_flag = 0;
    if (a.fk) {
      bool active = true;
      for (int it = 0; it < a.fk_iters; ++it) {
        const Rot64 R = quat_to_rot64(fkq[0], fkq[1], fkq[2], fkq[3]);
        double m[6][6], g[6], rmax = 0.0;
#pragma unroll
        for (int x = 0; x < 6; ++x) {
          g[x] = 0.0;
#pragma unroll
          for (int y = 0; y <= x; ++y) m[x][y] = (x == y) ? a.fk_lambda : 0.0;
        }
#pragma clang loop unroll_count(kCableUnroll)
        for (int i = 0; i < N; ++i) {
          double L, j[6];
          ik_row64(c_geom + i * 7, R, fkp, L, j);
          const double res = c_len[i][lane] - L;
          rmax = fmax(rmax, fabs(res));
#pragma unroll
          for (int x = 0; x < 6; ++x) {
            g[x] = fma(j[x], res, g[x]);
#pragma unroll
            for (int y = 0; y <= x; ++y) m[x][y] = fma(j[x], j[y], m[x][y]);
          }
        }
        active = active && !(rmax < a.fk_tol);
        chol_solve64(m, g);
        if (active) {
          fkp[0] += g[0];
          fkp[1] += g[1];
          fkp[2] += g[2];
          quat_apply_rotvec64(fkq, g[3], g[4], g[5]);
          ++fk_it;
        }
      }
      const Rot64 R = quat_to_rot64(fkq[0], fkq[1], fkq[2], fkq[3]);
#pragma clang loop unroll_count(kCableUnroll)
      for (int i = 0; i < N; ++i) {
        double L, j[6];
        ik_row64(c_geom + i * 7, R, fkp, L, j);
        fk_res = fmax(fk_res, fabs(c_len[i][lane] - L));
        if (JCACHE) {
#pragma unroll
          for (int c = 0; c < 6; ++c) c_je[JCACHE ? i : 0][JCACHE ? c : 0][lane] = j[c];
        }
      }
    }
    // ---- tension distribution ([NEW] SURVEY 8(a) row 15): T = Tm 1 + J (J^T J)^-1 J^T (f - Tm 1) with J at the FK
    //      estimate (the true pose without FK), bounds; then the SetForce limits; c_f becomes the applied force
    uint32_t lim = 0u;
    {
      double g[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
      const Rot64 Rt = a.fk ? quat_to_rot64(fkq[0], fkq[1], fkq[2], fkq[3]) : quat_to_rot64(q4[0], q4[1], q4[2], q4[3]);
      const double pt[3] = {a.fk ? fkp[0] : p[0], a.fk ? fkp[1] : p[1], a.fk ? fkp[2] : p[2]};
      if (a.td) {
        double m[6][6];
#pragma unroll
        for (int x = 0; x < 6; ++x) {
#pragma unroll
          for (int y = 0; y <= x; ++y) m[x][y] = 0.0;
        }
#pragma clang loop unroll_count(kCableUnroll)
        for (int i = 0; i < N; ++i) {
          double L, j[6];
          if (JCACHE) {
#pragma unroll
            for (int c = 0; c < 6; ++c) j[c] = a.fk ? c_je[JCACHE ? i : 0][JCACHE ? c : 0][lane] : c_jt[JCACHE ? i : 0][JCACHE ? c : 0][lane];
          } else {
            ik_row64(c_geom + i * 7, Rt, pt, L, j);
          }
          const double df = c_f[i][lane] - a.td_mid;
#pragma unroll
          for (int x = 0; x < 6; ++x) {
            g[x] = fma(j[x], df, g[x]);
#pragma unroll
            for (int y = 0; y <= x; ++y) m[x][y] = fma(j[x], j[y], m[x][y]);
          }
        }
        chol_solve64(m, g);
      }
#pragma clang loop unroll_count(kCableUnroll)
      for (int i = 0; i < N; ++i) {
        double applied = c_f[i][lane];
        if (a.td) {
          double L, j[6];
          if (JCACHE) {
#pragma unroll
            for (int c = 0; c < 6; ++c) j[c] = a.fk ? c_je[JCACHE ? i : 0][JCACHE ? c : 0][lane] : c_jt[JCACHE ? i : 0][JCACHE ? c : 0][lane];
          } else {
            ik_row64(c_geom + i * 7, Rt, pt, L, j);
          }
          double t = a.td_mid;
#pragma unroll
          for (int c = 0; c < 6; ++c) t = fma(g[c], j[c], t);
          const double tc = fmax(fmin(t, a.td_max), a.td_min);
          td_flag |= (tc != t) ? 1 : 0;
          applied = tc;
        }
        const double qd = c_qd[i][lane], q = c_q[i][lane];
        if (a.vel_limit > 0.0)  // Joint::SetForce velocity truncation [EXT]
          applied = ((qd > a.vel_limit && applied > 0.0) || (qd < -a.vel_limit && applied < 0.0)) ? 0.0 : applied;
        if (a.effort >= 0.0) applied = fmax(fmin(applied, a.effort), -a.effort);  // Joint::SetForce clamp (cube.sdf:438)
        if (a.travel_on && (q < a.travel_lo || q > a.travel_hi)) lim |= 1u << i;
        c_f[i][lane] = applied;
      }
    }
    if (a.dbg) {  // `pid` topic, cable 0 only (PLG.cpp:223-227; Pid.cpp:139-142,158-168)
      double* d = a.dbg + (size_t)r * 9;
      if (HOLD ? dbg_ran : run_pid) {
        d[0] = dbg_p;
        d[1] = dbg_i;
        d[2] = dbg_d;
        d[3] = HOLD ? dbg_des : c_des[0][lane];
      }
      d[4] = c_f[0][lane];
    }
    // ---- observables of step t_k (PLG.cpp:236-242, 248-280)
    if ((a.publish_mask >> step) & 1ull) {
      double* const O = a.obs + (size_t)step * a.obs_step_stride + r;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        O[(size_t)c * st] = p[c];
        O[(size_t)(7 + c) * st] = v[c];
        O[(size_t)(10 + c) * st] = om[c];
      }
#pragma unroll
      for (int c = 0; c < 4; ++c) O[(size_t)(3 + c) * st] = q4[c];
      O[13 * st] = fk_res;
      O[14 * st] = (double)fk_it;
      O[15 * st] = (double)((uint32_t)td_flag | (lim << 1));
#pragma clang loop unroll_count(kCableUnroll)
      for (int i = 0; i < N; ++i) {
        O[(size_t)(16 + i) * st] = c_q[i][lane];
        O[(size_t)(16 + N + i) * st] = c_qd[i][lane];
        O[(size_t)(16 + 2 * N + i) * st] = c_f[i][lane];
      }
    }
    // ---- world step to t_{k+1}: wrench = -J^T (applied - d qdot) + m g, semi-implicit Euler
    {
      double w[6] = {a.fgx, a.fgy, a.fgz, 0.0, 0.0, 0.0};
      const Rot64 R = quat_to_rot64(q4[0], q4[1], q4[2], q4[3]);
#pragma clang loop unroll_count(kCableUnroll)
      for (int i = 0; i < N; ++i) {
        double L, j[6];
        if (JCACHE) {
#pragma unroll
          for (int c = 0; c < 6; ++c) j[c] = c_jt[JCACHE ? i : 0][JCACHE ? c : 0][lane];
        } else {
          ik_row64(c_geom + i * 7, R, p, L, j);
        }
        if (TSTOP) {
#pragma unroll
          for (int c = 0; c < 6; ++c) c_js[TSTOP ? i : 0][TSTOP ? c : 0][lane] = j[c];
        }
        double t = fma(-a.damping, c_qd[i][lane], c_f[i][lane]);
        if (a.unilateral) t = fmax(t, 0.0);
#pragma unroll
        for (int c = 0; c < 6; ++c) w[c] = fma(-j[c], t, w[c]);
      }
      bool lumped_done = false;
      if constexpr (TSTOP) {
        if (a.ph_lumped) {  // (wave-uniform) the lumped legs: a 6 x 6 mass matrix instead of the platform's own (see f64_lumped_twist)
          lumped_done = true;
          double M[6][6];
#pragma unroll
          for (int x = 0; x < 6; ++x)
#pragma unroll
            for (int y = 0; y < 6; ++y) M[x][y] = 0.0;
          // world inertia R Ib R^T (+ the share of every leg that turns with the platform), gyroscopic torque om x (Iw om)
          const double rr[3][3] = {{R.r00, R.r01, R.r02}, {R.r10, R.r11, R.r12}, {R.r20, R.r21, R.r22}};
          const double ibm[3][3] = {{a.ib[0], a.ib[3], a.ib[4]}, {a.ib[3], a.ib[1], a.ib[5]}, {a.ib[4], a.ib[5], a.ib[2]}};
          double iw[3][3];
#pragma unroll
          for (int x = 0; x < 3; ++x)
#pragma unroll
            for (int y = 0; y < 3; ++y) {
              double acc = 0.0;
#pragma unroll
              for (int c = 0; c < 3; ++c)
#pragma unroll
                for (int d = 0; d < 3; ++d) acc = fma(rr[x][c] * ibm[c][d], rr[y][d], acc);
              iw[x][y] = acc + (x == y ? a.ph_iadd_total : 0.0);
            }
#pragma unroll
          for (int x = 0; x < 3; ++x) {
            M[x][x] = a.ph_mass;
#pragma unroll
            for (int y = 0; y < 3; ++y) M[3 + x][3 + y] = iw[x][y];
          }
          const double iom[3] = {fma(iw[0][2], om[2], fma(iw[0][1], om[1], iw[0][0] * om[0])), fma(iw[1][2], om[2], fma(iw[1][1], om[1], iw[1][0] * om[0])),
                                 fma(iw[2][2], om[2], fma(iw[2][1], om[1], iw[2][0] * om[0]))};
          w[3] -= fma(om[1], iom[2], -(om[2] * iom[1]));
          w[4] -= fma(om[2], iom[0], -(om[0] * iom[2]));
          w[5] -= fma(om[0], iom[1], -(om[1] * iom[0]));
#pragma clang loop unroll(disable)
          for (int i = 0; i < N; ++i) {
            double u[6];
#pragma unroll
            for (int c = 0; c < 6; ++c) u[c] = c_js[TSTOP ? i : 0][TSTOP ? c : 0][lane];  // the structure-matrix row [u, rb x u] at t_k
            const double* g = c_geom + i * 7;
            const double rb[3] = {fma(R.r02, g[5], fma(R.r01, g[4], R.r00 * g[3])), fma(R.r12, g[5], fma(R.r11, g[4], R.r10 * g[3])),
                                  fma(R.r22, g[5], fma(R.r21, g[4], R.r20 * g[3]))};
            const double L = g[6] - c_q[i][lane], il = 1.0 / L;
            const double vp[3] = {v[0] + fma(om[1], rb[2], -(om[2] * rb[1])), v[1] + fma(om[2], rb[0], -(om[0] * rb[2])), v[2] + fma(om[0], rb[1], -(om[1] * rb[0]))};
            const double along = fma(u[2], vp[2], fma(u[1], vp[1], u[0] * vp[0]));
            const double vt[3] = {fma(-u[0], along, vp[0]), fma(-u[1], along, vp[1]), fma(-u[2], along, vp[2])};
            const double ou[3] = {fma(om[1], u[2], -(om[2] * u[1])), fma(om[2], u[0], -(om[0] * u[2])), fma(om[0], u[1], -(om[1] * u[0]))};
            const double uvp[3] = {fma(u[1], vp[2], -(u[2] * vp[1])), fma(u[2], vp[0], -(u[0] * vp[2])), fma(u[0], vp[1], -(u[1] * vp[0]))};
            const double cl = -(a.ph_c * il);
            // damper force at the anchor + the weight of the point masses there; the spherical joint's torque
            const double fa[3] = {fma(cl, fma(2.0 * il, vt[0], -ou[0]), a.ph_mpt * a.gx), fma(cl, fma(2.0 * il, vt[1], -ou[1]), a.ph_mpt * a.gy),
                                  fma(cl, fma(2.0 * il, vt[2], -ou[2]), a.ph_mpt * a.gz)};
            w[0] += fa[0], w[1] += fa[1], w[2] += fa[2];
            w[3] += fma(rb[1], fa[2], -(rb[2] * fa[1])) + a.ph_c * fma(uvp[0], il, -om[0]);
            w[4] += fma(rb[2], fa[0], -(rb[0] * fa[2])) + a.ph_c * fma(uvp[1], il, -om[1]);
            w[5] += fma(rb[0], fa[1], -(rb[1] * fa[0])) + a.ph_c * fma(uvp[2], il, -om[2]);
            // apparent mass of the leg at the anchor: alpha I + beta u u^T through G = [I, -[rb]x]
            const double mu = a.ph_jleg * il * il, alpha = mu + a.ph_mpt, beta = a.ph_max - mu;
            const double rb2 = fma(rb[2], rb[2], fma(rb[1], rb[1], rb[0] * rb[0]));
            const double X[3][3] = {{0.0, -rb[2], rb[1]}, {rb[2], 0.0, -rb[0]}, {-rb[1], rb[0], 0.0}};  // [rb]x
#pragma unroll
            for (int x = 0; x < 3; ++x) {
              M[x][x] += alpha;
#pragma unroll
              for (int y = 0; y < 3; ++y) {
                M[3 + x][y] = fma(alpha, X[x][y], M[3 + x][y]);
                M[3 + x][3 + y] = fma(alpha, ((x == y) ? rb2 : 0.0) - rb[x] * rb[y], M[3 + x][3 + y]);
              }
            }
#pragma unroll
            for (int x = 0; x < 6; ++x)
#pragma unroll
              for (int y = 0; y <= x; ++y) M[x][y] = fma(beta * u[x], u[y], M[x][y]);
          }
          double acc6[6] = {w[0], w[1], w[2], w[3], w[4], w[5]};
          chol_solve64(M, acc6);  // (reads the lower triangle)
          v[0] = fma(a.dt, acc6[0], v[0]), v[1] = fma(a.dt, acc6[1], v[1]), v[2] = fma(a.dt, acc6[2], v[2]);
          om[0] = fma(a.dt, acc6[3], om[0]), om[1] = fma(a.dt, acc6[4], om[1]), om[2] = fma(a.dt, acc6[5], om[2]);
        }
      }
      if (!lumped_done) {
      v[0] = fma(a.dt * w[0], a.inv_mass, v[0]);
      v[1] = fma(a.dt * w[1], a.inv_mass, v[1]);
      v[2] = fma(a.dt * w[2], a.inv_mass, v[2]);
      double tb[3] = {fma(R.r20, w[5], fma(R.r10, w[4], R.r00 * w[3])), fma(R.r21, w[5], fma(R.r11, w[4], R.r01 * w[3])), fma(R.r22, w[5], fma(R.r12, w[4], R.r02 * w[3]))};
      const double ob[3] = {fma(R.r20, om[2], fma(R.r10, om[1], R.r00 * om[0])), fma(R.r21, om[2], fma(R.r11, om[1], R.r01 * om[0])),
                            fma(R.r22, om[2], fma(R.r12, om[1], R.r02 * om[0]))};
      const double io[3] = {fma(a.ib[4], ob[2], fma(a.ib[3], ob[1], a.ib[0] * ob[0])), fma(a.ib[5], ob[2], fma(a.ib[1], ob[1], a.ib[3] * ob[0])),
                            fma(a.ib[2], ob[2], fma(a.ib[5], ob[1], a.ib[4] * ob[0]))};
      tb[0] -= fma(ob[1], io[2], -(ob[2] * io[1]));
      tb[1] -= fma(ob[2], io[0], -(ob[0] * io[2]));
      tb[2] -= fma(ob[0], io[1], -(ob[1] * io[0]));
      const double ab[3] = {fma(a.ibinv[4], tb[2], fma(a.ibinv[3], tb[1], a.ibinv[0] * tb[0])), fma(a.ibinv[5], tb[2], fma(a.ibinv[1], tb[1], a.ibinv[3] * tb[0])),
                            fma(a.ibinv[2], tb[2], fma(a.ibinv[5], tb[1], a.ibinv[4] * tb[0]))};
      om[0] = fma(a.dt, fma(R.r02, ab[2], fma(R.r01, ab[1], R.r00 * ab[0])), om[0]);
      om[1] = fma(a.dt, fma(R.r12, ab[2], fma(R.r11, ab[1], R.r10 * ab[0])), om[1]);
      om[2] = fma(a.dt, fma(R.r22, ab[2], fma(R.r21, ab[1], R.r20 * ab[0])), om[2]);
      }  // (!lumped_done)
      if constexpr (TSTOP) {
        for (int sweep = 0; sweep < a.travel_stop; ++sweep) {
#pragma clang loop unroll(disable)
          for (int i = 0; i < N; ++i) {
            double j[6];
#pragma unroll
            for (int c = 0; c < 6; ++c) j[c] = c_js[TSTOP ? i : 0][TSTOP ? c : 0][lane];
            const double qi = c_q[i][lane];
            const double qdn = -fma(j[5], om[2], fma(j[4], om[1], fma(j[3], om[0], fma(j[2], v[2], fma(j[1], v[1], j[0] * v[0])))));
            const bool hit = (qi >= a.travel_hi && qdn > 0.0) || (qi <= a.travel_lo && qdn < 0.0);
            // Iw^-1 (rb x u) = R Ib^-1 R^T (rb x u)
            const double sb[3] = {fma(R.r20, j[5], fma(R.r10, j[4], R.r00 * j[3])), fma(R.r21, j[5], fma(R.r11, j[4], R.r01 * j[3])),
                                  fma(R.r22, j[5], fma(R.r12, j[4], R.r02 * j[3]))};
            const double cb[3] = {fma(a.ibinv[4], sb[2], fma(a.ibinv[3], sb[1], a.ibinv[0] * sb[0])), fma(a.ibinv[5], sb[2], fma(a.ibinv[1], sb[1], a.ibinv[3] * sb[0])),
                                  fma(a.ibinv[2], sb[2], fma(a.ibinv[5], sb[1], a.ibinv[4] * sb[0]))};
            const double aw[3] = {fma(R.r02, cb[2], fma(R.r01, cb[1], R.r00 * cb[0])), fma(R.r12, cb[2], fma(R.r11, cb[1], R.r10 * cb[0])),
                                  fma(R.r22, cb[2], fma(R.r21, cb[1], R.r20 * cb[0]))};
            const double d = fma(j[5], aw[2], fma(j[4], aw[1], fma(j[3], aw[0], fma(j[2], j[2], fma(j[1], j[1], j[0] * j[0])) * a.inv_mass)));
            const double lam = hit ? qdn / d : 0.0;
            const double lm = lam * a.inv_mass;
            v[0] = fma(lm, j[0], v[0]);
            v[1] = fma(lm, j[1], v[1]);
            v[2] = fma(lm, j[2], v[2]);
            om[0] = fma(lam, aw[0], om[0]);
            om[1] = fma(lam, aw[1], om[1]);
            om[2] = fma(lam, aw[2], om[2]);
          }
        }
      }
      p[0] = fma(a.dt, v[0], p[0]);
      p[1] = fma(a.dt, v[1], p[1]);
      p[2] = fma(a.dt, v[2], p[2]);
      const double h = a.half_dt, x = q4[0], y = q4[1], z = q4[2], ww = q4[3];
      const double nx = fma(h, fma(-om[2], y, fma(om[1], z, ww * om[0])), x);
      const double ny = fma(h, fma(-om[0], z, fma(om[2], x, ww * om[1])), y);
      const double nz = fma(h, fma(-om[1], x, fma(om[0], y, ww * om[2])), z);
      const double nw = fma(-h, fma(om[2], z, fma(om[1], y, om[0] * x)), ww);
      const double inv = rsqrt64(fma(nw, nw, fma(nz, nz, fma(ny, ny, nx * nx))));
      q4[0] = nx * inv;
      q4[1] = ny * inv;
      q4[2] = nz * inv;
      q4[3] = nw * inv;
    }
  }
  // ---- store
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    S[(size_t)c * st] = p[c];
    S[(size_t)(7 + c) * st] = v[c];
    S[(size_t)(10 + c) * st] = om[c];
    S[(size_t)(13 + c) * st] = fkp[c];
  }
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    S[(size_t)(3 + c) * st] = q4[c];
    S[(size_t)(16 + c) * st] = fkq[c];
  }
#pragma clang loop unroll_count(kCableUnroll)
  for (int i = 0; i < N; ++i) {
    if (!HOLD) S[(size_t)(20 + (W + 1) * i + W) * st] = c_ierr[i][lane];  // (HOLD: the integrals live in the Pids' own rows)
  }
  if (PR) a.meta[r] = (uint8_t)((meta & kMetaModeMask) | ((uint32_t)calls << kMetaCallShift));
}

// Per-robot command arrival on a precision = 64 handle (cdpr_set_*_command_masked; PLG.cpp:206-219 per model): what
// cdpr_latch_fast_kernel does on the fp32 state - the Joy's row becomes the robot's active target, entering the mode from
// another one resets its Pid (JFC.cpp:101-103,113-115: integral 0, call count 0), setForce resets nothing.
struct LatchF64Args {
  const uint8_t* mask;   // uint8[B], or nullptr = every robot
  uint8_t* meta;
  const float* pending;  // float[B][n]
  float* target;         // float[B][n]
  double* state;
  size_t stride;
  uint32_t batch, n;
  uint32_t new_mode;     // kMetaForce / kMetaPosition / kMetaVelocity
  uint32_t hold;         // HOLD handles: both Pids of every cable live in their own rows (f64_hold_row); entering a mode clears THAT Pid's
  uint32_t hold_win;     // ... samples such a record's window holds (kHoldWin | kHoldWinLong)
  uint32_t win;          // prior errors kept per cable (kWin, or kWinLong on handles with windows of 12 .. 32 samples): the integral is row 20 + (win + 1) i + win
};
static __global__ __launch_bounds__(256) void cdpr_latch_f64_kernel(const LatchF64Args a) {
  const uint32_t r = blockIdx.x * 256u + threadIdx.x;
  if (r >= a.batch) return;
  if (a.mask && !a.mask[r]) return;
  for (uint32_t i = 0; i < a.n; ++i) a.target[(size_t)r * a.n + i] = a.pending[(size_t)r * a.n + i];
  uint32_t m = a.meta[r];
  if (a.new_mode == kMetaForce) {
    m = (m & ~kMetaModeMask) | kMetaForce;
  } else if ((m & kMetaModeMask) != a.new_mode) {
    if (a.hold) {  // Pid::reset of the Pid of the mode entered (JFC.cpp:101-103,113-115), every cable: word, integral, window, stamps
      const int pid = (a.new_mode == kMetaVelocity) ? 1 : 0;
      for (uint32_t i = 0; i < a.n; ++i)
        for (int row = 0; row < hold_pid_rows((int)a.hold_win); ++row) a.state[(size_t)(f64_hold_row((int)a.n, (int)i, pid, (int)a.hold_win) + row) * a.stride + r] = 0.0;
    } else {
      for (uint32_t i = 0; i < a.n; ++i) a.state[(size_t)(20 + (a.win + 1u) * i + a.win) * a.stride + r] = 0.0;
    }
    m = a.new_mode;  // call count 0
  }
  a.meta[r] = (uint8_t)m;
}

// cdpr_split_kernel_f64 - one step per launch with TWO waves per 64 robots, split by role like cdpr_split_kernel: FK + TD
// handles, up to one workgroup per CU (117 KiB of LDS).  The step in double is one dependent chain of ~6 500 vector
// instructions, 60 % of them the Newton stage: here the estimator wave (measured lengths, Newton-Raphson FK, the
// tension distribution's matrix and factor, then - forces in - the tensions) and the controller wave (IK, PID, forces out,
// - tensions in - SetForce limits, observables, world step) each have a SIMD to themselves.  The same statements in the
// same order as cdpr_step_kernel_f64<N, true, true> over the same LDS columns: same bits (tested).
// LEAN (batches beyond one workgroup per CU): nothing cached in LDS (26 KiB: four workgroups per CU, both waves of a
// workgroup on one SIMD) - the rings are read from HBM / L2 inside the PID loop and the structure-matrix rows are rebuilt
// where they are needed again, as in the one-wave kernel's large-batch variant; what the split buys there is the other
// wave's arithmetic under this wave's memory round trips.
// HOLD: the position-hold branch live, as in cdpr_step_kernel_f64<.., HOLD> (the controller wave is the lighter of the two:
// the selected Pid's round trip and its arithmetic run under the estimator wave's Newton stage).
template <int N, bool LEAN = false, int HOLD = 0>  // HOLD: 0 | 1 | 2 (+ cascades, cmd_limit 0)
__global__ __launch_bounds__(128, LEAN ? 2 : 1) void cdpr_split_kernel_f64(const F64Args a) {
  __shared__ double c_len[N][64], c_q[N][64], c_qd[N][64], c_f[N][64], c_des[N][64], c_ierr[N][64];
  __shared__ double c_win[LEAN ? 1 : N][LEAN ? 1 : kWin][64];
  __shared__ double c_jt[LEAN ? 1 : N][LEAN ? 1 : 6][64];  // rows at the true pose (controller wave: IK stage -> world step)
  __shared__ double c_je[LEAN ? 1 : N][LEAN ? 1 : 6][64];  // rows at the FK estimate (estimator wave: closing evaluation -> tension distribution)
  __shared__ double x_est[3][64];    // estimator -> controller: residual, iterations, infeasible flag
  __shared__ double c_park[HOLD ? 13 : 1][64];  // HOLD: the controller wave's platform state between its IK stage and the world step
#ifndef CDPR_F64_HOLD_KU
#define CDPR_F64_HOLD_KU N
#endif
  constexpr int kU = LEAN ? (N < 4 ? N : 4) : (HOLD ? (N < CDPR_F64_HOLD_KU ? N : CDPR_F64_HOLD_KU) : N);  // cables per pass of the loops over the cables (see cdpr_step_kernel_f64); LEAN at
                                                  // 65 536 x 8: 31.8 us with 2, 29.4 with 4, 44.9 with 8 (356 B of scratch)
  const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
  const uint32_t r = blockIdx.x * 64u + lane;
  __shared__ double c_geom_lds[LEAN ? N * 7 : 1];  // LEAN: the geometry table in LDS (see cdpr_step_kernel_f64); each wave writes all of it and reads what it wrote
  if (LEAN) {
    if (lane < N * 7) c_geom_lds[lane] = a.geom[lane];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
  const double* const c_geom = LEAN ? c_geom_lds : a.geom;
  __shared__ double c_hold_w[HOLD ? 2 : 1][kHoldWin], c_hold_rot[HOLD ? 2 : 1][HOLD ? kHoldWin : 1][kHoldWin], c_hold_par[HOLD ? 2 : 1][kHoldPar];
  const bool live = r < a.batch;  // (no early return: both waves meet at two barriers; tail lanes shadow the last robot)
  const size_t st = a.stride;
  double* const S = a.state + (live ? r : a.batch - 1u);
  const bool first_world = (a.flags & kFlagFirstWorldStep) != 0u;
  const bool force_mode = (a.flags & kFlagForceMode) != 0u;

  if (wave == 0) {
    // ------------------------------------------------------------------------------------------------ estimator wave
    // (measured and not kept: s_setprio 3 for this wave in the LEAN build, as cdpr_split_kernel has it: 24.1 against 23.65 us)
    CDPR_F64_STAMP(0);
    CDPR_F64_CLOCK(6);
    const double p[3] = {S[0 * st], S[1 * st], S[2 * st]};
    const double q4[4] = {S[3 * st], S[4 * st], S[5 * st], S[6 * st]};
    double fkp[3] = {S[13 * st], S[14 * st], S[15 * st]};
    double fkq[4] = {S[16 * st], S[17 * st], S[18 * st], S[19 * st]};
    CDPR_F64_LANDED();
    CDPR_F64_STAMP(1);
    {
      const Rot64 R = quat_to_rot64(q4[0], q4[1], q4[2], q4[3]);
#pragma clang loop unroll_count(kU)
      for (int i = 0; i < N; ++i) {
        double L, j[6];
        ik_row64(c_geom + i * 7, R, p, L, j);
        c_len[i][lane] = L;
      }
    }
    CDPR_F64_STAMP(2);
    double fk_res = 0.0;
    int fk_it = 0, td_flag = 0;
    {
      bool active = true;
      for (int it = 0; it < a.fk_iters; ++it) {
        const Rot64 R = quat_to_rot64(fkq[0], fkq[1], fkq[2], fkq[3]);
        double m[6][6], g[6], rmax = 0.0;
#pragma unroll
        for (int x = 0; x < 6; ++x) {
          g[x] = 0.0;
#pragma unroll
          for (int y = 0; y <= x; ++y) m[x][y] = (x == y) ? a.fk_lambda : 0.0;
        }
#pragma clang loop unroll_count(kU)
        for (int i = 0; i < N; ++i) {
          double L, j[6];
          ik_row64(c_geom + i * 7, R, fkp, L, j);
          const double res = c_len[i][lane] - L;
          rmax = fmax(rmax, fabs(res));
#pragma unroll
          for (int x = 0; x < 6; ++x) {
            g[x] = fma(j[x], res, g[x]);
#pragma unroll
            for (int y = 0; y <= x; ++y) m[x][y] = fma(j[x], j[y], m[x][y]);
          }
        }
        active = active && !(rmax < a.fk_tol);
        chol_solve64(m, g);
        if (active) {
          fkp[0] += g[0];
          fkp[1] += g[1];
          fkp[2] += g[2];
          quat_apply_rotvec64(fkq, g[3], g[4], g[5]);
          ++fk_it;
        }
      }
      if (!LEAN) {  // the closing evaluation: the residual of the estimate, its rows kept for the tension distribution
        const Rot64 R = quat_to_rot64(fkq[0], fkq[1], fkq[2], fkq[3]);
#pragma clang loop unroll_count(kU)
        for (int i = 0; i < N; ++i) {
          double L, j[6];
          ik_row64(c_geom + i * 7, R, fkp, L, j);
          fk_res = fmax(fk_res, fabs(c_len[i][lane] - L));
#pragma unroll
          for (int c = 0; c < 6; ++c) c_je[LEAN ? 0 : i][LEAN ? 0 : c][lane] = j[c];
        }
      }
    }
    if (live) {
#pragma unroll
      for (int c = 0; c < 3; ++c) S[(size_t)(13 + c) * st] = fkp[c];
#pragma unroll
      for (int c = 0; c < 4; ++c) S[(size_t)(16 + c) * st] = fkq[c];
    }
    // the tension distribution's matrix and its factor need no forces
    double m[6][6], invd[6];
#pragma unroll
    for (int x = 0; x < 6; ++x) {
#pragma unroll
      for (int y = 0; y <= x; ++y) m[x][y] = 0.0;
    }
    const Rot64 Rt = quat_to_rot64(fkq[0], fkq[1], fkq[2], fkq[3]);
    double jr[LEAN ? N : 1][6];  // LEAN: the rows at the estimate, built once in registers for the three passes below (fully unrolled)
#pragma unroll
    for (int i = 0; i < N; ++i) {
      double L, j[6];
      if (LEAN) {  // ... and the closing evaluation is this pass too (round 6: the rows were evaluated twice - 480 instructions of a 7 500-instruction step)
        ik_row64(c_geom + i * 7, Rt, fkp, L, j);
        fk_res = fmax(fk_res, fabs(c_len[i][lane] - L));
#pragma unroll
        for (int c = 0; c < 6; ++c) jr[LEAN ? i : 0][c] = j[c];
      } else {
#pragma unroll
        for (int c = 0; c < 6; ++c) j[c] = c_je[LEAN ? 0 : i][LEAN ? 0 : c][lane];
      }
#pragma unroll
      for (int x = 0; x < 6; ++x) {
#pragma unroll
        for (int y = 0; y <= x; ++y) m[x][y] = fma(j[x], j[y], m[x][y]);
      }
    }
    chol_factor64(m, invd);
    // (pinned here: the compiler sinks the whole block - rows, matrix, factor: ~700 instructions - behind the barrier, next to its
    //  first use; with the hold branch live the controller wave is the later one and this wave would do that work after the wait;
    //  without it this wave is the later one, the order makes no difference and the sunk form needs fewer registers)
    if constexpr (HOLD != 0) {
#pragma unroll
      for (int x = 0; x < 6; ++x) {
        asm volatile("" : "+v"(invd[x]));
#pragma unroll
        for (int y = 0; y <= x; ++y) asm volatile("" : "+v"(m[x][y]));
      }
      if (LEAN) {
#pragma unroll
        for (int i = 0; i < N; ++i)
#pragma unroll
          for (int c = 0; c < 6; ++c) asm volatile("" : "+v"(jr[LEAN ? i : 0][c]));
      }
    }
    CDPR_F64_STAMP(3);
    __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0)
    __builtin_amdgcn_s_barrier();        // #1: the controller wave's forces are in c_f
    CDPR_F64_STAMP(4);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    double g[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int i = 0; i < N; ++i) {
      double j[6];
#pragma unroll
      for (int c = 0; c < 6; ++c) j[c] = LEAN ? jr[LEAN ? i : 0][c] : c_je[LEAN ? 0 : i][LEAN ? 0 : c][lane];
      const double df = c_f[i][lane] - a.td_mid;
#pragma unroll
      for (int x = 0; x < 6; ++x) g[x] = fma(j[x], df, g[x]);
    }
    chol_apply64(m, invd, g);
#pragma unroll
    for (int i = 0; i < N; ++i) {
      double j[6];
#pragma unroll
      for (int c = 0; c < 6; ++c) j[c] = LEAN ? jr[LEAN ? i : 0][c] : c_je[LEAN ? 0 : i][LEAN ? 0 : c][lane];
      double t = a.td_mid;
#pragma unroll
      for (int c = 0; c < 6; ++c) t = fma(g[c], j[c], t);
      const double tc = fmax(fmin(t, a.td_max), a.td_min);
      td_flag |= (tc != t) ? 1 : 0;
      c_f[i][lane] = tc;
    }
    x_est[0][lane] = fk_res;
    x_est[1][lane] = (double)fk_it;
    x_est[2][lane] = (double)td_flag;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_s_barrier();  // #2: tensions and estimator results are out
    CDPR_F64_STAMP(5);
    CDPR_F64_CLOCK(7);
    return;
  }
  // -------------------------------------------------------------------------------------------------- controller wave
#ifndef CDPR_F64_CTL_PRIO
#define CDPR_F64_CTL_PRIO 3
#endif
  // LEAN: four workgroups share a CU, every SIMD hosts the estimator wave of one and the controller wave of another, and the issue
  // arbiter favours the OLDER wave: the fourth workgroup of a CU has the junior wave on both of its SIMDs - its controller wave, short
  // of slots under the first workgroup's Newton stage, handed its forces over 6 us after the others' and the launch waited for that
  // quarter of the grid (`profiles/r06_fp64_timeline.txt`: end of a workgroup by quarter of the grid 20.6 / 22.4 / 21.8 / 26.0 us).
  // The controller wave is short and sleeps at the barriers most of the launch: at raised priority it is out of the way early, every
  // estimator wave loses the same share, and no workgroup trails.
  if (LEAN && CDPR_F64_CTL_PRIO) __builtin_amdgcn_s_setprio(CDPR_F64_CTL_PRIO);
#ifndef CDPR_F64_CTL_SLEEP
#define CDPR_F64_CTL_SLEEP 10  // (x 64 cycles; 65 536 x 8 plain, same box: 22.4 us at 10, 22.4 - 23.0 at 25, 23.0 at 0 and at 50; no difference with the hold branch)
#endif
  if (LEAN && CDPR_F64_CTL_SLEEP) __builtin_amdgcn_s_sleep(CDPR_F64_CTL_SLEEP);  // the estimator wave's fourteen rows ahead of this wave's fifty: its chain is the launch's
  CDPR_F64_STAMP(8);
  double p[3] = {S[0 * st], S[1 * st], S[2 * st]};
  double q4[4] = {S[3 * st], S[4 * st], S[5 * st], S[6 * st]};
  double v[3] = {S[7 * st], S[8 * st], S[9 * st]}, om[3] = {S[10 * st], S[11 * st], S[12 * st]};
  const uint32_t rc = live ? r : a.batch - 1u;
  if constexpr (HOLD != 0) {
    float joy[N];
#pragma unroll
    for (int i = 0; i < N; ++i) joy[i] = a.cmd[(size_t)rc * N + i];
    // the controller tables (this wave's: it reads what it wrote) while the platform rows and the Joy are on their way
    hold_tables_fill(a, lane, c_hold_w, c_hold_rot, c_hold_par);
#pragma unroll
    for (int i = 0; i < N; ++i) c_des[i][lane] = (double)joy[i];
  } else {
#pragma clang loop unroll_count(kU)
    for (int i = 0; i < N; ++i) {
      c_ierr[i][lane] = S[(size_t)(20 + 11 * i + 10) * st];
      c_des[i][lane] = (double)a.cmd[(size_t)rc * N + i];
      if (!LEAN) {
#pragma unroll
        for (int k = 0; k < kWin; ++k) c_win[LEAN ? 0 : i][LEAN ? 0 : k][lane] = S[(size_t)(20 + 11 * i + k) * st];
      }
    }
  }
  CDPR_F64_LANDED();
  CDPR_F64_STAMP(9);
  // observables of step t_k that are known before the forces (PLG.cpp:236-242, 248-280) go out as they arise: pose and twist here, joint
  // position and velocity from the IK stage - not in the tail behind the estimator wave, where every store is on the launch's critical path
  const bool publish = (a.publish_mask & 1ull) && live;
  double* const O = a.obs + r;
  const bool actual_is_vel = (a.flags & kFlagActualIsVelocity) != 0u;
  const int calls = a.pid_calls;
  const bool run_pid = !first_world && !force_mode && calls != 0;  // Pid.cpp:123-126: the first call since reset returns 0
  const bool full = calls >= a.nbuf;
  const int ring_slot = a.ring_slot % kWin;
  const double* wt = a.wtab + ring_slot * (kWin + 2);
  double dbg_p = 0.0, dbg_i = 0.0, dbg_d = 0.0, dbg_des = 0.0;
  bool dbg_ran = false;  // (HOLD: cable 0's Pid really ran this step)
  // ---- IK on the state at t_k and the per-cable force (PLG.cpp:222-228 -> JFC.cpp:59-96 -> Pid.cpp:122-191)
  {
    const Rot64 R = quat_to_rot64(q4[0], q4[1], q4[2], q4[3]);
    auto hold_rows_of = [&](int i, bool& vb) -> double* {
      vb = HOLD && a.hold_mode == 2 && fabs(c_des[i][lane]) > a.hold_eps;  // JFC.cpp:72
      return S + (size_t)(f64_hold_row(N, HOLD ? i : 0, 0) + (vb ? kHoldPidRows : 0)) * st;
    };
    auto last_position_of = [&](int i) -> double* { return S + (size_t)(f64_state_rows(N) + (HOLD ? i : 0) * kHoldCableRows) * st; };  // mLastPosition
    // (the stores below come behind the first requests: a store may alias a later load as far as the compiler can tell)
    if constexpr (HOLD != 0) {
      // ---- the hold branch live (JFC.cpp:59-96 with both Pids alive), FOUR CABLES PER PASS (round 6): their rows - the fourteen of the
      // Pid each calls in this step, known from its command alone - are requested together, and where every lane's four cables are
      // steady (the Pid was called one step ago or its window is still filling: no fit, no first call) they run through hold_fast64,
      // straight-line code the scheduler interleaves.  The per-cable form before it was ~340 instructions a cable in ~25 basic blocks,
      // each one a dependent chain: 840 issue slots a cable, 14 us of an 18 us step at one robot (`profiles/r06_fp64_timeline.txt`).
      // A pass shorter than four cables repeats the last one: same rows in (all requests precede all stores), same values out.
#ifndef CDPR_F64_HOLD_Q
#define CDPR_F64_HOLD_Q 4
#endif
#ifndef CDPR_F64_HOLD_Q_LEAN
#define CDPR_F64_HOLD_Q_LEAN 2  // (65 536 x 8, same box: 29.0 us at 2, 30.0 at 3, 30.6 at 4 - the longer pass takes the slots of the Newton stage next door)
#endif
      constexpr int kQ = LEAN ? CDPR_F64_HOLD_Q_LEAN : CDPR_F64_HOLD_Q;
      constexpr int Q = N < kQ ? N : kQ;
      int ci[Q];
      bool vb[Q];
      double *HRq[Q], *LPq[Q];
      HoldRows64 hq[Q];
      auto request_pass = [&](int i0) {
#pragma unroll
        for (int k = 0; k < Q; ++k) {
          ci[k] = min(i0 + k, N - 1);
          HRq[k] = hold_rows_of(ci[k], vb[k]);
          LPq[k] = last_position_of(ci[k]);
          hq[k] = hold_load64(HRq[k], LPq[k], st);
        }
      };
      request_pass(0);  // ... the first pass's ahead of the IK stage
      if (publish) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          O[(size_t)c * st] = p[c];
          O[(size_t)(7 + c) * st] = v[c];
          O[(size_t)(10 + c) * st] = om[c];
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) O[(size_t)(3 + c) * st] = q4[c];
      }
      // IK of every cable, straight-line: joint position and velocity into their LDS columns (and out as observables); the platform
      // state then waits in LDS for the world step - its 26 registers are the passes' (two cables at a time: the first pass's rows are
      // arriving in 28 registers a cable, and a longer straight line spilled them)
#ifndef CDPR_F64_HOLD_IK_KU
#define CDPR_F64_HOLD_IK_KU 4
#endif
#pragma clang loop unroll_count(CDPR_F64_HOLD_IK_KU)
      for (int i = 0; i < N; ++i) {
        double L, j[6];
        ik_row64(c_geom + i * 7, R, p, L, j);
        const double q = c_geom[i * 7 + 6] - L;
        const double qd = -fma(j[5], om[2], fma(j[4], om[1], fma(j[3], om[0], fma(j[2], v[2], fma(j[1], v[1], j[0] * v[0])))));
        c_q[i][lane] = q;
        c_qd[i][lane] = qd;
        if (publish) {
          O[(size_t)(16 + i) * st] = q;
          O[(size_t)(16 + N + i) * st] = qd;
        }
        if (!LEAN) {
#pragma unroll
          for (int c = 0; c < 6; ++c) c_jt[LEAN ? 0 : i][LEAN ? 0 : c][lane] = j[c];
        }
      }
#pragma unroll
      for (int c = 0; c < 3; ++c) c_park[c][lane] = p[c], c_park[7 + c][lane] = v[c], c_park[10 + c][lane] = om[c];
#pragma unroll
      for (int c = 0; c < 4; ++c) c_park[3 + c][lane] = q4[c];
#pragma clang loop unroll(disable)
      for (int i0 = 0;;) {  // (the next pass's requests at the END of a pass: asked for at its head behind `i0 > 0`, the rows would stay alive across the rare path)
        double qq[Q], qdq[Q];
        bool unsteady = false;
#pragma unroll
        for (int k = 0; k < Q; ++k) {
          qq[k] = c_q[ci[k]][lane];
          qdq[k] = c_qd[ci[k]][lane];
          unsteady = unsteady || !hold_steady64(hq[k].word, a.step0, vb[k] ? a.nbuf : a.alt_nbuf);
        }
        const bool fast = HOLD == 1 && !first_world && a.hold_mode != 0 && __builtin_amdgcn_ballot_w64(unsteady) == 0ull;
        if (__builtin_expect(fast, 1)) {  // (the other branch is the cold one: what has to be spilled is spilled there)
          double fq[Q], des[Q], ierr_old[Q], held[Q];
          HoldFir64 fir[Q];
#pragma unroll
          for (int k = 0; k < Q; ++k) {
            const bool hold = a.hold_mode == 2 && !vb[k];
            const double target = c_des[ci[k]][lane];
            des[k] = vb[k] ? target : (hold ? hq[k].held : target);
            ierr_old[k] = hq[k].ierr, held[k] = hq[k].held;
            fir[k] = hold_fast_fir64(hq[k], &c_hold_rot[HOLD && vb[k] ? 1 : 0][0][0], c_hold_par[HOLD && vb[k] ? 1 : 0][8], vb[k] ? a.nbuf : a.alt_nbuf, des[k], vb[k] ? qdq[k] : qq[k],
                                     a.step0, a.dt);
          }
          __builtin_amdgcn_sched_barrier(0);  // (the windows are done with before the gains are fetched: see hold_fast_fir64)
#pragma unroll
          for (int k = 0; k < Q; ++k) {
            const bool hold = a.hold_mode == 2 && !vb[k];
            const double desired = des[k];
            double tp, ti, td;
            fq[k] = hold_fast64(HRq[k], LPq[k], st, ierr_old[k], held[k], fir[k], c_hold_par[HOLD && vb[k] ? 1 : 0], vb[k] ? a.nbuf : a.alt_nbuf, hold, qq[k], desired, a.step0, a.dt, tp, ti,
                                td);
            if (k == 0) {  // cable 0 is the first cable of the first pass only
              const bool c0 = i0 == 0;
              dbg_p = c0 ? tp : dbg_p, dbg_i = c0 ? ti : dbg_i, dbg_d = c0 ? td : dbg_d, dbg_des = c0 ? desired : dbg_des;
              dbg_ran = dbg_ran || c0;
            }
          }
#pragma unroll
          for (int k = 0; k < Q; ++k) c_f[ci[k]][lane] = fq[k];
        } else {
          // any other call (first call since a reset, a window with a gap: the fit, Force mode, the first world step, cascades): a cable
          // at a time, rolled - the form before round 6; it requests its rows again and keeps nothing of the pass alive
#pragma clang loop unroll(disable)
          for (int i = i0; i < (i0 + Q < N ? i0 + Q : N); ++i) {
            bool vel_branch;
            double* const HR = hold_rows_of(i, vel_branch);
            double* const LP = last_position_of(i);
            const double q = c_q[i][lane], qd = c_qd[i][lane];
            double force = (force_mode && !first_world) ? c_des[i][lane] : 0.0;
            if (!first_world) {
              const double target = c_des[i][lane];
              const bool pos_branch = a.hold_mode == 1 || (a.hold_mode == 2 && !vel_branch);
              const bool hold = a.hold_mode == 2 && !vel_branch;
              if (!hold && live) LP[0] = q;  // JFC.cpp:68,75,87
              if (vel_branch || pos_branch) {
                const HoldRows64 hs = hold_load64(HR, LP, st);
                HoldPid64 g;
                const double* const pp = c_hold_par[HOLD && vel_branch ? 1 : 0];
                g.kf = pp[0], g.kp = pp[1], g.ki = pp[2], g.kd = pp[3], g.imax = pp[4], g.imin = pp[5], g.cmax = pp[6], g.cmin = pp[7];
                g.nbuf = vel_branch ? a.nbuf : a.alt_nbuf, g.degree = vel_branch ? a.degree : a.alt_degree;
                g.clamp = (vel_branch ? a.clamp_cmd : a.alt_clamp_cmd) != 0;
                g.pcas = vel_branch ? a.pcas[1] : a.pcas[0], g.dcas = vel_branch ? a.dcas[1] : a.dcas[0];  // (selects: a dynamic index copies the arrays to scratch)
                const double desired = vel_branch ? target : (hold ? hs.held : target);
                bool ran = false;
                double tp = 0.0, ti = 0.0, td = 0.0;
                force = hold_finish64<HOLD == 2>(HR, st, hs, &c_hold_rot[HOLD && vel_branch ? 1 : 0][0][0], pp[8], desired, vel_branch ? qd : q, a.step0, a.dt, g, a, vel_branch, ran, tp, ti, td, live);
                if (i == 0 && ran) {
                  dbg_p = tp, dbg_i = ti, dbg_d = td;
                  dbg_ran = true;
                  dbg_des = desired;
                }
              }
            }
            c_f[i][lane] = force;
          }
        }
        i0 += Q;
        if (i0 >= N) break;
        request_pass(i0);
      }
    } else {
    // The rings of a cable are requested while the cables BEFORE it are computed (LEAN, round 6): a cable's store of its ring slot may
    // alias the next cable's loads as far as the compiler can tell, so it never moved a load across one - eight round trips in a row
    constexpr bool kRingAhead = LEAN && !HOLD;
#ifndef CDPR_F64_RING_KU
#define CDPR_F64_RING_KU 2
#endif
    constexpr int kUc = kRingAhead ? (N < CDPR_F64_RING_KU ? N : CDPR_F64_RING_KU) : kU;  // (the requests are placed by hand: the pass no longer has to be long for them)
    double rn0[kWin], rn1[kWin];  // kRingAhead: the rings of cable i (on entry of its pass) and of cable i + 1
    if (kRingAhead && run_pid) {
#pragma unroll
      for (int k = 0; k < kWin; ++k) {
        rn0[k] = S[(size_t)(20 + 11 * 0 + k) * st];
        rn1[k] = S[(size_t)(20 + 11 * (N > 1 ? 1 : 0) + k) * st];
      }
    }
    if (publish) {
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        O[(size_t)c * st] = p[c];
        O[(size_t)(7 + c) * st] = v[c];
        O[(size_t)(10 + c) * st] = om[c];
      }
#pragma unroll
      for (int c = 0; c < 4; ++c) O[(size_t)(3 + c) * st] = q4[c];
    }
#pragma clang loop unroll_count(kUc)
    for (int i = 0; i < N; ++i) {
      double ring[kWin];
      if (kRingAhead && run_pid) {
#pragma unroll
        for (int k = 0; k < kWin; ++k) {
          ring[k] = rn0[k];
          rn0[k] = rn1[k];
        }
        if (i + 2 < N) {
#pragma unroll
          for (int k = 0; k < kWin; ++k) rn1[k] = S[(size_t)(20 + 11 * (i + 2) + k) * st];
        }
      }
      double L, j[6];
      ik_row64(c_geom + i * 7, R, p, L, j);
      const double q = c_geom[i * 7 + 6] - L;
      const double qd = -fma(j[5], om[2], fma(j[4], om[1], fma(j[3], om[0], fma(j[2], v[2], fma(j[1], v[1], j[0] * v[0])))));
      c_q[i][lane] = q;
      c_qd[i][lane] = qd;
      if (publish) {
        O[(size_t)(16 + i) * st] = q;
        O[(size_t)(16 + N + i) * st] = qd;
      }
      if (!LEAN) {
#pragma unroll
        for (int c = 0; c < 6; ++c) c_jt[LEAN ? 0 : i][LEAN ? 0 : c][lane] = j[c];
      }
      double force = (force_mode && !first_world) ? c_des[i][lane] : 0.0;
      if (run_pid) {
        const double desired = c_des[i][lane];
        const double error = desired - (actual_is_vel ? qd : q);
        double acc = wt[kWin] * error;
#pragma unroll
        for (int k = 0; k < kWin; ++k) acc = fma(wt[k], LEAN ? ring[k] : c_win[LEAN ? 0 : i][LEAN ? 0 : k][lane], acc);
        const double p_term = a.kp * error;
        const double prev_ierr = c_ierr[i][lane];
        double ie = fma(a.dt, error, prev_ierr);
        double i_term = a.ki * ie;
        const double i_raw = i_term;
        if (i_term > a.imax) {  // Pid.cpp:143-152
          i_term = a.imax;
          ie = i_term / a.ki;
        } else if (i_term < a.imin) {
          i_term = a.imin;
          ie = i_term / a.ki;
        }
        const double derived = full ? acc * a.inv_dt : 0.0;
        const double d_term = a.kd * derived;
        const double cmd = fma(a.kf, desired, p_term) + i_term + d_term;
        double out = a.clamp_cmd ? fmax(fmin(cmd, a.cmax), a.cmin) : cmd;  // Pid.cpp:175-177
        if (out != cmd) {                                                    // Pid.cpp:181-184
          ie = prev_ierr;
          out = fma(a.dt * error, a.ki, out);
        }
        force = out;
        // after the FIR has read the slot's old content; the one slot that changed, not the ring (the LDS build wrote all ten rows of
        // every cable back at the end of the launch until round 6: 1.4 us of stores behind the critical path at one robot); and the
        // integral from here as well, not from the tail
        if (live) {
          a.state[(size_t)(20 + 11 * i + ring_slot) * st + r] = error;
          a.state[(size_t)(20 + 11 * i + 10) * st + r] = ie;
        }
        if (i == 0) {
          dbg_p = p_term;
          dbg_i = i_raw;
          dbg_d = d_term;
        }
      }
      c_f[i][lane] = force;
    }
    }  // (!HOLD)
  }
  // Experiment (CDPR_F64_EARLY_ROWS = 1; LEAN without HOLD): the rows of the world step (the structure matrix at t_k again: nothing of the IK
  // stage is kept, the registers were the rings') rebuilt HERE, where this wave would wait for the Newton stage anyway, not behind the
  // second barrier where all 480 instructions are the launch's critical path; they cross the barriers in registers
#ifndef CDPR_F64_EARLY_ROWS
#define CDPR_F64_EARLY_ROWS 0  // (measured, 65 536 x 8, same box, three runs each: 22.08 / 22.46 / 22.41 us with, 22.72 / 22.30 / 22.49 without - the slots come
#endif                         //  out of the Newton stage next door; not kept)
  constexpr bool kEarlyRows = LEAN && !HOLD && CDPR_F64_EARLY_ROWS;
  double jw[kEarlyRows ? N : 1][6];
  if (kEarlyRows) {
    const Rot64 R = quat_to_rot64(q4[0], q4[1], q4[2], q4[3]);
#pragma unroll
    for (int i = 0; i < N; ++i) {
      double L, j[6];
      ik_row64(c_geom + i * 7, R, p, L, j);
#pragma unroll
      for (int c = 0; c < 6; ++c) {
        jw[kEarlyRows ? i : 0][c] = j[c];
        asm volatile("" : "+v"(jw[kEarlyRows ? i : 0][c]));  // (here, not sunk behind the barriers next to their use)
      }
    }
  }
  CDPR_F64_STAMP(10);
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): the forces are in LDS
  __builtin_amdgcn_s_barrier();        // #1
  __builtin_amdgcn_s_barrier();        // #2: the estimator wave has finished the tension distribution
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  CDPR_F64_STAMP(11);
  if constexpr (HOLD != 0) {  // (the platform state waited in LDS)
#pragma unroll
    for (int c = 0; c < 3; ++c) p[c] = c_park[c][lane], v[c] = c_park[7 + c][lane], om[c] = c_park[10 + c][lane];
#pragma unroll
    for (int c = 0; c < 4; ++c) q4[c] = c_park[3 + c][lane];
  }
  const double fk_res = x_est[0][lane], fk_it = x_est[1][lane];
  const uint32_t td_flag = (uint32_t)x_est[2][lane];
  uint32_t lim = 0u;
#pragma clang loop unroll_count(kU)
  for (int i = 0; i < N; ++i) {
    double applied = c_f[i][lane];
    const double qd = c_qd[i][lane], q = c_q[i][lane];
    if (a.vel_limit > 0.0)  // Joint::SetForce velocity truncation [EXT]
      applied = ((qd > a.vel_limit && applied > 0.0) || (qd < -a.vel_limit && applied < 0.0)) ? 0.0 : applied;
    if (a.effort >= 0.0) applied = fmax(fmin(applied, a.effort), -a.effort);  // Joint::SetForce clamp (cube.sdf:438)
    if (a.travel_on && (q < a.travel_lo || q > a.travel_hi)) lim |= 1u << i;
    c_f[i][lane] = applied;
  }
  if (a.dbg && live) {  // `pid` topic, cable 0 only (PLG.cpp:223-227; Pid.cpp:139-142,158-168)
    double* d = a.dbg + (size_t)r * 9;
    if (HOLD ? dbg_ran : run_pid) {
      d[0] = dbg_p;
      d[1] = dbg_i;
      d[2] = dbg_d;
      d[3] = HOLD ? dbg_des : c_des[0][lane];
    }
    d[4] = c_f[0][lane];
  }
  // ---- observables of step t_k (PLG.cpp:236-242, 248-280)
  if (publish) {  // (the rest of the image left with the IK stage)
    O[13 * st] = fk_res;
    O[14 * st] = fk_it;
    O[15 * st] = (double)(td_flag | (lim << 1));
#pragma clang loop unroll_count(kU)
    for (int i = 0; i < N; ++i) O[(size_t)(16 + 2 * N + i) * st] = c_f[i][lane];
  }
  // ---- world step to t_{k+1}: wrench = -J^T (applied - d qdot) + m g, semi-implicit Euler
  {
    double w[6] = {a.fgx, a.fgy, a.fgz, 0.0, 0.0, 0.0};
    const Rot64 R = quat_to_rot64(q4[0], q4[1], q4[2], q4[3]);
#pragma clang loop unroll_count(kEarlyRows ? N : kU)
    for (int i = 0; i < N; ++i) {
      double L, j[6];
      if (kEarlyRows) {
#pragma unroll
        for (int c = 0; c < 6; ++c) j[c] = jw[kEarlyRows ? i : 0][c];
      } else if (LEAN) {
        ik_row64(c_geom + i * 7, R, p, L, j);
      } else {
#pragma unroll
        for (int c = 0; c < 6; ++c) j[c] = c_jt[LEAN ? 0 : i][LEAN ? 0 : c][lane];
      }
      double t = fma(-a.damping, c_qd[i][lane], c_f[i][lane]);
      if (a.unilateral) t = fmax(t, 0.0);
#pragma unroll
      for (int c = 0; c < 6; ++c) w[c] = fma(-j[c], t, w[c]);
    }
    v[0] = fma(a.dt * w[0], a.inv_mass, v[0]);
    v[1] = fma(a.dt * w[1], a.inv_mass, v[1]);
    v[2] = fma(a.dt * w[2], a.inv_mass, v[2]);
    double tb[3] = {fma(R.r20, w[5], fma(R.r10, w[4], R.r00 * w[3])), fma(R.r21, w[5], fma(R.r11, w[4], R.r01 * w[3])), fma(R.r22, w[5], fma(R.r12, w[4], R.r02 * w[3]))};
    const double ob[3] = {fma(R.r20, om[2], fma(R.r10, om[1], R.r00 * om[0])), fma(R.r21, om[2], fma(R.r11, om[1], R.r01 * om[0])),
                            fma(R.r22, om[2], fma(R.r12, om[1], R.r02 * om[0]))};
    const double io[3] = {fma(a.ib[4], ob[2], fma(a.ib[3], ob[1], a.ib[0] * ob[0])), fma(a.ib[5], ob[2], fma(a.ib[1], ob[1], a.ib[3] * ob[0])),
                            fma(a.ib[2], ob[2], fma(a.ib[5], ob[1], a.ib[4] * ob[0]))};
    tb[0] -= fma(ob[1], io[2], -(ob[2] * io[1]));
    tb[1] -= fma(ob[2], io[0], -(ob[0] * io[2]));
    tb[2] -= fma(ob[0], io[1], -(ob[1] * io[0]));
    const double ab[3] = {fma(a.ibinv[4], tb[2], fma(a.ibinv[3], tb[1], a.ibinv[0] * tb[0])), fma(a.ibinv[5], tb[2], fma(a.ibinv[1], tb[1], a.ibinv[3] * tb[0])),
                            fma(a.ibinv[2], tb[2], fma(a.ibinv[5], tb[1], a.ibinv[4] * tb[0]))};
    om[0] = fma(a.dt, fma(R.r02, ab[2], fma(R.r01, ab[1], R.r00 * ab[0])), om[0]);
    om[1] = fma(a.dt, fma(R.r12, ab[2], fma(R.r11, ab[1], R.r10 * ab[0])), om[1]);
    om[2] = fma(a.dt, fma(R.r22, ab[2], fma(R.r21, ab[1], R.r20 * ab[0])), om[2]);
    p[0] = fma(a.dt, v[0], p[0]);
    p[1] = fma(a.dt, v[1], p[1]);
    p[2] = fma(a.dt, v[2], p[2]);
    const double h = a.half_dt, x = q4[0], y = q4[1], z = q4[2], ww = q4[3];
    const double nx = fma(h, fma(-om[2], y, fma(om[1], z, ww * om[0])), x);
    const double ny = fma(h, fma(-om[0], z, fma(om[2], x, ww * om[1])), y);
    const double nz = fma(h, fma(-om[1], x, fma(om[0], y, ww * om[2])), z);
    const double nw = fma(-h, fma(om[2], z, fma(om[1], y, om[0] * x)), ww);
    const double inv = rsqrt64(fma(nw, nw, fma(nz, nz, fma(ny, ny, nx * nx))));
    q4[0] = nx * inv;
    q4[1] = ny * inv;
    q4[2] = nz * inv;
    q4[3] = nw * inv;
  }
  // ---- store
  CDPR_F64_STAMP(12);
  if (live) {
    double* const W = a.state + r;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      W[(size_t)c * st] = p[c];
      W[(size_t)(7 + c) * st] = v[c];
      W[(size_t)(10 + c) * st] = om[c];
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) W[(size_t)(3 + c) * st] = q4[c];
  }
#ifdef CDPR_STAMPS
  CDPR_F64_STAMP(13);
  __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): on gfx9 the stores count too - every store of this wave acknowledged
  CDPR_F64_STAMP(14);
#endif
}

// Read-out of double rows into robot-major arrays (double or float), one thread per (robot, column)
// ---- cdpr_rollout_velocity on a precision = 64 handle (round 6), by composition: every (robot, sampled sequence) becomes a column
// of a scratch state (expand), every step of the horizon is ONE launch of the handle's own step kernel over those columns with the
// step's commands gathered into a Joy batch (cmd), and a third kernel adds |p(t_k+1) - p_ref|^2 to the trajectory's cost (cost).
// No rollout kernel of its own: the rollout in double is there for checking the fp32 one, not for throughput.
struct Roll64Args {
  const double* src;  // the handle's state rows
  double* dst;        // the trajectories' state rows
  uint32_t src_stride, dst_stride, rows, batch, samples;
  uint32_t zero_from;  // rows >= this are cleared in the copies (a Joy on jointVelocities in Position mode resets the velocity Pid,
                       // JFC.cpp:113-115); == rows: none
  // HOLD handles: that Pid has a record of its own per cable (f64_hold_row) - hold_zero_pid = 1 + the Pid whose records are cleared, 0: none
  uint32_t hold_base, hold_cable_rows, hold_pid_rows, hold_zero_pid;
  // per-robot handles: the robot's mode / call-count byte decides (a robot that is not in Velocity mode enters it: its - velocity - Pid
  // is reset, its count 0); every trajectory gets its own byte
  const uint8_t* meta_src;
  uint8_t* meta_dst;
};
static __global__ __launch_bounds__(256) void cdpr_roll64_expand_kernel(const Roll64Args a) {
  const uint32_t t = blockIdx.x * 256u + threadIdx.x;
  if (t >= a.batch * a.samples) return;
  const uint32_t b = t / a.samples;
  bool resets = true;  // (uniform handles: zero_from / hold_zero_pid already say whether the Joy resets the Pid)
  if (a.meta_src) {
    const uint32_t m = a.meta_src[b];
    resets = (m & kMetaModeMask) != kMetaVelocity;
    a.meta_dst[t] = (uint8_t)(resets ? kMetaVelocity : m);
  }
  for (uint32_t r = 0; r < a.rows; ++r) {
    bool zero = resets && r >= a.zero_from;
    if (resets && a.hold_zero_pid && r >= a.hold_base) {
      const uint32_t in_cable = (r - a.hold_base) % a.hold_cable_rows, first = 1u + (a.hold_zero_pid - 1u) * a.hold_pid_rows;
      zero = zero || (in_cable >= first && in_cable < first + a.hold_pid_rows);
    }
    a.dst[(size_t)r * a.dst_stride + t] = zero ? 0.0 : a.src[(size_t)r * a.src_stride + b];
  }
}
struct Roll64CmdArgs {
  const float* commands;  // float[B][H][S][n]
  float* out;             // float[B * S][n]: the Joy batch of step k
  uint32_t batch, samples, horizon, n, k;
};
static __global__ __launch_bounds__(256) void cdpr_roll64_cmd_kernel(const Roll64CmdArgs a) {
  const uint32_t e = blockIdx.x * 256u + threadIdx.x;
  if (e >= a.batch * a.samples * a.n) return;
  const uint32_t t = e / a.n, i = e - t * a.n, b = t / a.samples, s = t - b * a.samples;
  a.out[e] = a.commands[(((size_t)b * a.horizon + a.k) * a.samples + s) * a.n + i];
}
struct Roll64CostArgs {
  const double* state;  // the trajectories' state rows (rows 0..2: position)
  const float* ref;     // float[B][3]
  double* acc;          // double[B * S]
  float* out;           // float[B][S], written with the last step (nullptr before)
  uint32_t stride, batch, samples;
};
static __global__ __launch_bounds__(256) void cdpr_roll64_cost_kernel(const Roll64CostArgs a) {
  const uint32_t t = blockIdx.x * 256u + threadIdx.x;
  if (t >= a.batch * a.samples) return;
  const uint32_t b = t / a.samples;
  const double ex = a.state[t] - (double)a.ref[3 * b], ey = a.state[(size_t)a.stride + t] - (double)a.ref[3 * b + 1],
               ez = a.state[(size_t)2 * a.stride + t] - (double)a.ref[3 * b + 2];
  const double c = a.acc[t] + fma(ez, ez, fma(ey, ey, ex * ex));
  a.acc[t] = c;
  if (a.out) a.out[t] = (float)c;
}

// Several row ranges of a double row buffer -> robot-major arrays in ONE launch (cdpr_get_observables_f64: position, velocity,
// effort, pose, twist): segment s = rows [first_row[s], + width[s]) -> out + off[s] elements, [batch][width[s]].
struct Unpack64MultiArgs {
  const double* rows;
  void* out;
  uint32_t stride, batch, nseg, total_width;
  uint32_t first_row[5], width[5], cum[5], off[5];  // cum: widths of the segments before s; off: element offset of segment s in `out`
  int as_float;
};
static __global__ __launch_bounds__(256) void cdpr_unpack64_multi_kernel(const Unpack64MultiArgs a) {
  const uint32_t t = blockIdx.x * 256u + threadIdx.x;
  if (t >= a.batch * a.total_width) return;
  const uint32_t r = t / a.total_width, j = t - r * a.total_width;
  uint32_t s = 0;
#pragma unroll
  for (uint32_t k = 1; k < 5; ++k) s = (k < a.nseg && j >= a.cum[k]) ? k : s;
  const uint32_t jj = j - a.cum[s];
  const double v = a.rows[(size_t)(a.first_row[s] + jj) * a.stride + r];
  const size_t o = (size_t)a.off[s] + (size_t)r * a.width[s] + jj;
  if (a.as_float)
    static_cast<float*>(a.out)[o] = (float)v;
  else
    static_cast<double*>(a.out)[o] = v;
}

struct Unpack64Args {
  const double* rows;
  void* out;
  uint32_t stride, batch, width, first_row;
  int as_float;
};
static __global__ __launch_bounds__(256) void cdpr_unpack64_kernel(const Unpack64Args a) {
  const uint32_t t = blockIdx.x * 256u + threadIdx.x;
  if (t >= a.batch * a.width) return;
  const uint32_t r = t / a.width, j = t - r * a.width;
  const double v = a.rows[(size_t)(a.first_row + j) * a.stride + r];
  if (a.as_float)
    static_cast<float*>(a.out)[t] = (float)v;
  else
    static_cast<double*>(a.out)[t] = v;
}

}  // namespace cdpr
