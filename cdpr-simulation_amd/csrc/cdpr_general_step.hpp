// cdpr_general_step.hpp — the general controller path in ONE launch per step (gfx950, fp32 + an fp64 fit), lane per robot.
//
// What the register-resident fast path of cdpr_step_kernel.hpp cannot represent:
//   * the position-hold branch of JointForceCalculator::update (JFC.cpp:78-82, velocityEpsilon >= 0): both Pids of a
//     cable stay alive in Velocity mode and are sampled at non-uniform times;
//   * biquad cascades on the P and D inputs (Pid::CascadeFilter, Pid.cpp:27-44; Filter.h:130-165);
//   * derivative windows up to 32 samples / degree 4 (Pid::derive + fitPolynomial, Pid.cpp:193-247);
//   * cmdLimit == 0 (no command clamp: mCmd keeps its old value, Pid.cpp:175-186);
//   * per-robot modes combined with any of these, with the lumped-leg physics or with two different windows.
// Rounds 1-3 ran this as two launches (one thread per (robot, cable) for the controller, forces through HBM, then the
// platform kernel).  Here one lane owns one robot for the whole step, as on the fast path: forces never leave the
// registers, and the same kernel serves one step, several steps per launch, the trajectory record and the MPC rollout.
//
// Records (HBM, one dword ROW per field, one column per robot: a wave touches 256 contiguous bytes per row):
//   row i (i < n)                           mLastPosition of cable i (JFC.h:45)
//   block(pid, i) = n + (pid n + i) R       pid 0 = position Pid, 1 = velocity Pid; R = 2 nb + 4 + 8 ncas rows:
//     + j          (j < nb)   mDbufferY ring slot j (the error samples)
//     + nb + j                mDbufferX ring slot j as a world-step index (int32 bit pattern)
//     + 2 nb                  mIerr
//     + 2 nb + 1              meta: bit 0 mWasLastTime | samples in the window (6 bits) | ring head (6) | `run` (6) =
//                             consecutive one-step gaps ending at the newest sample, saturating
//     + 2 nb + 2              mLastTime as a world-step index
//     + 2 nb + 3              mCmd (only read or written when some Pid has no command clamp)
//     + 2 nb + 4 ...          P-input cascade x1 x2 y1 y2 per stage, then the D-input cascade
//   A Pid::reset of the whole batch is one memset over the Pid's rows; all-zero rows ARE the reset state.
//   (mDerr is not stored: Pid.cpp:154-157 reads the old value only when dt <= 0, and a Pid is updated at most once per
//    world step with strictly increasing stamps.)
//
// Per step and cable only the ACTIVE Pid's rows are touched (position Pid in Position mode and in the hold branch,
// velocity Pid otherwise): nb + 5 rows read (values, integral, meta, stamp of the last call, hold position), 5 written
// (one value, one stamp, integral, meta, last call) = 84 B per cable at the shipped 11-sample window.  The stamps of the
// window are only read when they are needed:
//   * a window whose nb samples were taken at consecutive world steps (meta.run >= nb - 1) is the uniform grid of the
//     fast path: the derivative is the closed-form FIR, weights looked up by ring head in LDS;
//   * anything else (the nb - 1 steps after a switch between the two Pids in the hold branch) is a least-squares fit on
//     the real stamps.  These are rare and scattered over lanes and cables, so they are COMPACTED: every lane queues its
//     (cable, Pid) items in LDS, then the wave works the queue with one item per lane - orthogonal polynomials on the
//     sample stamps (Forsythe recurrence, fp64): no normal equations, well conditioned for any gap pattern.
// The record rows travel global -> LDS by LDS-DMA as soon as the commands (which select the Pid) are known, and stay in
// flight under the IK and the Newton stage (stage order as in cdpr_onestep_kernel: IK -> early observables -> Newton ->
// controller -> tension distribution -> world step).
//
// Reference paths: Pid.cpp, JFC.cpp = JointForceCalculator.cpp, Filter.h (relative to src/cdpr_gazebo/).
#pragma once
#include "cdpr_step_kernel.hpp"

namespace cdpr {

constexpr int kGenMaxBuf = 32;   // CDPR_MAX_D_BUFFER
constexpr int kGenMaxDeg = 4;    // CDPR_MAX_D_DEGREE
constexpr int kGenMaxCas = 4;    // CDPR_MAX_CASCADE

struct GenLayout {
  int n, nb, ncas;  // cables, longest window of the two Pids, deepest cascade
  __host__ __device__ int rows_per_block() const { return 2 * nb + 4 + 8 * ncas; }
  __host__ __device__ int block(int pid, int cable) const { return n + (pid * n + cable) * rows_per_block(); }
  __host__ __device__ int pid_rows() const { return n * rows_per_block(); }      // one Pid of every cable: contiguous
  __host__ __device__ int total_rows() const { return n + 2 * pid_rows(); }
  __host__ __device__ int r_stamp() const { return nb; }
  __host__ __device__ int r_ierr() const { return 2 * nb; }
  __host__ __device__ int r_meta() const { return 2 * nb + 1; }
  __host__ __device__ int r_last() const { return 2 * nb + 2; }
  __host__ __device__ int r_cmd() const { return 2 * nb + 3; }
  __host__ __device__ int r_pfilt() const { return 2 * nb + 4; }
  __host__ __device__ int r_dfilt() const { return 2 * nb + 4 + 4 * ncas; }
};

constexpr uint32_t kGmWasLast = 1u, kGmCountShift = 1u, kGmHeadShift = 7u, kGmRunShift = 13u, kGmField = 63u;

struct GenPid {
  float kf, kp, ki, kd, imax, imin, cmax, cmin;
  int nbuf, degree, pcas, dcas, clamp;
  float pa0, pa1, pa2, pb1, pb2;  // BiQuad::SetFc(relCutoff, 1.0, quality), Filter.h:130-140
  float da0, da1, da2, db1, db2;
};

struct GenCtl {
  float* rec;            // record rows: rec[row * rstride + column]
  uint32_t rstride;      // columns per row (robots, or trajectories of a rollout, rounded up to 64)
  uint32_t rec_bytes;    // rows x rstride x 4 (< 4 GiB: the buffer is addressed with 32-bit offsets)
  const float* vel_cmd;  // latched jointVelocities float[B][n], or nullptr (target 0)
  const float* pos_cmd;  // latched jointPositions, or nullptr
  const float* frc_cmd;  // latched force command (setForce), or nullptr
  const uint8_t* mode_arr;  // per-robot mode (0 Force, 1 Position, 2 Velocity), or nullptr: `mode` for every robot
  const float* wtab;     // FIR weights by ring head: [pid][head][slot], rows of kNbp(NBMAX) floats
  int mode;
  int now_step;          // world step of the launch's first step
  int any_noclamp;       // some Pid has cmdMax <= cmdMin: the mCmd rows are live
  float eps, dt;
  GenLayout lay;
  GenPid pid[2];         // [0] position Pid, [1] velocity Pid
  // ROLLOUT: every trajectory works on a private copy of its robot's records (column = trajectory index in `rec`)
  const float* src_rec;
  uint32_t src_rstride;
};

__host__ __device__ constexpr int gen_nbp(int nbmax) { return nbmax <= 11 ? 12 : 32; }  // padded weight-row length (float4 reads)

// The record buffer is addressed as ONE buffer resource (4 scalar registers): row = a 32-bit scalar offset, the lane's
// column (and, where the row is the lane's own, its ring slot) = a 32-bit vector offset.  A cable's 16 rows then cost
// one address register and one scalar per row; as 64-bit pointers the compiler hoists every row address of a launch
// out of the step loop and spills them (measured: 1.4 KiB of scratch per lane).
struct GenBuf {
  __amdgpu_buffer_rsrc_t rsrc;
  uint32_t rs4;  // bytes per row
  CDPR_DEV float load(int row, uint32_t voff) const { return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, voff, (uint32_t)row * rs4, 0)); }
  CDPR_DEV int loadi(int row, uint32_t voff) const { return (int)__builtin_amdgcn_raw_buffer_load_b32(rsrc, voff, (uint32_t)row * rs4, 0); }
  CDPR_DEV void store(int row, uint32_t voff, float v) const { __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rsrc, voff, (uint32_t)row * rs4, 0); }
  CDPR_DEV void storei(int row, uint32_t voff, int v) const { __builtin_amdgcn_raw_buffer_store_b32((unsigned)v, rsrc, voff, (uint32_t)row * rs4, 0); }
  // global -> LDS without a register destination: lane l's dword lands at dst_row + 4 l
  CDPR_DEV void to_lds(int row, uint32_t voff, float* dst_row) const {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)dst_row, 4, voff, (uint32_t)row * rs4, 0, 0);
  }
};

// Derivative at the newest stamp of the least-squares polynomial of degree `degree` through the nb samples (t_j, y_j):
// what Pid::derive + fitPolynomial compute (Pid.cpp:193-247), posed on orthogonal polynomials over the sample stamps
// (Forsythe's three-term recurrence) instead of normal equations: p_{k+1}(x) = (x - alpha_k) p_k(x) - beta_k p_{k-1}(x),
// alpha_k = sum x p_k^2 / sum p_k^2, beta_k = sum p_k^2 / sum p_{k-1}^2, fit = sum c_k p_k with c_k = sum y p_k / sum p_k^2.
// Abscissae x_j = (t_j - t_new) / h, h = mean spacing: integers before the division, so exact for any stamps.
// Returns d/dt in 1 / world steps (the caller divides by the step length).  Slots j >= nb are ignored.
template <int NBMAX>
CDPR_DEV double gen_fit(const float (&y)[NBMAX], const int (&t)[NBMAX], int nb, int degree, int t_new, int t_old) {
  double h = (double)(t_new - t_old) / (double)(nb - 1);
  if (!(h > 0.0)) h = 1.0;
  const double inv_h = 1.0 / h;
  double pp[NBMAX], pc[NBMAX];  // p_{k-1}, p_k on the sample points
  double sc = 0.0;              // sum p_k^2
#pragma unroll
  for (int j = 0; j < NBMAX; ++j) {
    pp[j] = 0.0;
    pc[j] = (j < nb) ? 1.0 : 0.0;
    sc += pc[j];
  }
  double sp = 1.0, beta = 0.0;
  double vp = 0.0, vc = 1.0;  // p_{k-1}(0), p_k(0): the newest stamp is x = 0
  double dp = 0.0, dc = 0.0;  // their derivatives at 0
  double deriv = 0.0;
#pragma unroll
  for (int k = 0; k < kGenMaxDeg; ++k) {
    if (k < degree) {
      double sx = 0.0;
#pragma unroll
      for (int j = 0; j < NBMAX; ++j) {
        const double x = (double)(t[j] - t_new) * inv_h;
        sx = fma(x * pc[j], pc[j], sx);
      }
      const double alpha = sx / sc;
      double sn = 0.0, bn = 0.0;
#pragma unroll
      for (int j = 0; j < NBMAX; ++j) {
        const double x = (double)(t[j] - t_new) * inv_h;
        const double pn = (j < nb) ? fma(x - alpha, pc[j], -(beta * pp[j])) : 0.0;
        pp[j] = pc[j];
        pc[j] = pn;
        sn = fma(pn, pn, sn);
        bn = fma((double)y[j], pn, bn);
      }
      const double vn = fma(-alpha, vc, -(beta * vp));
      const double dn = vc + fma(-alpha, dc, -(beta * dp));
      vp = vc;
      vc = vn;
      dp = dc;
      dc = dn;
      deriv = fma(bn / sn, dn, deriv);
      beta = sn / sc;
      sp = sc;
      sc = sn;
    }
  }
  (void)sp;
  return deriv * inv_h;
}

// Pid::CascadeFilter::update (Pid.cpp:38-44) over BiQuad::process (Filter.h:152-165), states in the record rows of this lane.
// `stages` is wave-uniform (the deeper cascade of the two Pids), `mine` this lane's own depth (0: the value passes through,
// nothing is stored): no divergent control flow around the loads and stores.
CDPR_DEV float gen_cascade(const GenBuf& B, int row0, uint32_t voff, int stages, int mine, bool store, float a0, float a1, float a2, float b1,
                           float b2, float x) {
  float out = x;
  for (int c = 0; c < stages; ++c) {
    const int rw = row0 + 4 * c;
    const bool on = c < mine;
    const float x1 = B.load(rw, voff), x2 = B.load(rw + 1, voff), y1 = B.load(rw + 2, voff), y2 = B.load(rw + 3, voff);
    const float y0 = a0 * out + a1 * x1 + a2 * x2 - b1 * y1 - b2 * y2;
    if (on && store) {
      B.store(rw + 1, voff, x1);
      B.store(rw, voff, out);
      B.store(rw + 3, voff, y1);
      B.store(rw + 2, voff, y0);
    }
    out = on ? y0 : out;
  }
  return out;
}

// The Pid a lane runs for one cable this step: position or velocity Pid, field by field (v_cndmask); everything by value -
// a reference into the kernel arguments makes the compiler keep a private copy of them in scratch memory.
struct GenSel {
  float kf, kp, ki, kd, imax, imin, cmax, cmin;
  int nbuf, pcas, dcas;
  float pa0, pa1, pa2, pb1, pb2, da0, da1, da2, db1, db2;
};
CDPR_DEV GenSel gen_select(bool vel, const GenPid a, const GenPid b) {
  GenSel c;
  c.kf = vel ? b.kf : a.kf, c.kp = vel ? b.kp : a.kp, c.ki = vel ? b.ki : a.ki, c.kd = vel ? b.kd : a.kd;
  c.imax = vel ? b.imax : a.imax, c.imin = vel ? b.imin : a.imin, c.cmax = vel ? b.cmax : a.cmax, c.cmin = vel ? b.cmin : a.cmin;
  c.nbuf = vel ? b.nbuf : a.nbuf, c.pcas = vel ? b.pcas : a.pcas, c.dcas = vel ? b.dcas : a.dcas;
  c.pa0 = vel ? b.pa0 : a.pa0, c.pa1 = vel ? b.pa1 : a.pa1, c.pa2 = vel ? b.pa2 : a.pa2, c.pb1 = vel ? b.pb1 : a.pb1, c.pb2 = vel ? b.pb2 : a.pb2;
  c.da0 = vel ? b.da0 : a.da0, c.da1 = vel ? b.da1 : a.da1, c.da2 = vel ? b.da2 : a.da2, c.db1 = vel ? b.db1 : a.db1, c.db2 = vel ? b.db2 : a.db2;
  return c;
}

// Pid.cpp:154-186 from the derivative on: D term (through the D-input cascade), command, clamp, anti-windup; returns mCmd.
// `on`: this lane really finishes this cable's Pid::update now (the stores follow it).  row0 = first row of the cable's
// position-Pid block, bo = the lane's byte offset (selects the Pid and the column).
CDPR_DEV float gen_finish(const GenBuf& RB, const GenSel c, int row_ierr, int row_cmd, int row_dfilt, int dcas_max, bool noclamp, bool on, uint32_t bo,
                          float derived, float err, float dt, float pre, float ie, float prev, float old, float& d_term) {
  float derr = derived;
  if (dcas_max) derr = gen_cascade(RB, row_dfilt, bo, dcas_max, c.dcas, on, c.da0, c.da1, c.da2, c.db1, c.db2, derr);
  d_term = c.kd * derr;
  const float cmd = pre + d_term;
  float out = (c.cmax > c.cmin) ? fmaxf(fminf(cmd, c.cmax), c.cmin) : old;  // Pid.cpp:175-177: without a clamp mCmd keeps its value
  const bool wind = out != cmd;                                              // Pid.cpp:181-184
  ie = wind ? prev : ie;
  out = wind ? fmaf(dt * err, c.ki, out) : out;
  if (on) {
    RB.store(row_ierr, bo, ie);
    if (noclamp) RB.store(row_cmd, bo, out);
  }
  return out;
}

// NBMAX bounds the window at compile time: 11 (the shipped length and everything below it: record rows staged through
// LDS) or 32 (the maximum; rows read straight from HBM).
template <int N, bool FK, bool TD, bool ROLLOUT, int NBMAX>
__global__ __launch_bounds__(64, 1) void cdpr_gen_step_kernel(const StepArgs a, const GenCtl g) {
  constexpr int NP = cable_pairs(N);
  constexpr int G = joint_groups(N);
  constexpr bool STAGE = NBMAX <= 11;
  constexpr int NBP = gen_nbp(NBMAX);
  constexpr int kStageRows = NBMAX + 5;  // values | ierr | meta | last | cmd | hold position
  constexpr int kSg = STAGE ? 64 : N * 64;  // floats between consecutive staged rows of one cable
  __shared__ __attribute__((aligned(16))) float lds[NP * kGeomFloatsPerPair];
  __shared__ __attribute__((aligned(16))) float wrot[2][NBMAX][NBP];
  __shared__ float stage[STAGE ? N : 1][STAGE ? kStageRows : 1][64];
  __shared__ float park[STAGE ? 1 : 7][STAGE ? 1 : N][64];  // NBMAX = 32 (rows not staged): where a queued cable parks its Pid terms
  __shared__ uint32_t q_item[64 * N];
  __shared__ float q_err[64 * N];
  __shared__ float q_res[N][64];
  __shared__ uint32_t q_count;

  const uint32_t lane = threadIdx.x;
  const uint32_t r = blockIdx.x * 64u + lane;  // robot, or trajectory index in a rollout
  const uint32_t units = ROLLOUT ? a.batch * a.roll_samples : a.batch;
  const uint32_t ru = (r < units) ? r : (units - 1u);  // tail lanes shadow the last unit, stores are masked
  const uint32_t rr = ROLLOUT ? ru / a.roll_samples : ru;
  const uint32_t sample = ROLLOUT ? ru - rr * a.roll_samples : 0u;
  const bool live = r < units;
  const size_t st = a.stride;
  GenLayout L;
  L.n = g.lay.n, L.nb = g.lay.nb, L.ncas = g.lay.ncas;
  const uint32_t rs = g.rstride;
  GenBuf RB;
  RB.rsrc = __builtin_amdgcn_make_buffer_rsrc(g.rec, 0, (int)g.rec_bytes, 0x00020000);
  RB.rs4 = rs * 4u;
  const uint32_t col = ru;  // record column: the robot, or this trajectory's private copy

  const float gval = (lane < NP * kGeomFloatsPerPair) ? a.geom[lane] : 0.f;
  const uint32_t off = rr * 16u, woff = r * 16u;
  const float4 p0 = load_slot(a.state, st, 0, off), p1 = load_slot(a.state, st, 1, off), p2 = load_slot(a.state, st, 2, off),
               p3 = load_slot(a.state, st, 3, off);
  float4 p4 = make_float4(0.f, 0.f, 0.f, 1.f);
  if (FK) p4 = load_slot(a.state, st, 4, off);
  int mode = g.mode_arr ? (int)g.mode_arr[rr] : g.mode;
  if (lane < NP * kGeomFloatsPerPair) lds[lane] = gval;
  for (uint32_t k = lane; k < 2u * NBMAX * NBP; k += 64u) (&wrot[0][0][0])[k] = g.wtab[k];
  if (lane == 0) q_count = 0u;

  if (ROLLOUT) {
    // private copy of the robot's records; a Joy on jointVelocities reaching a robot that is not in Velocity mode resets
    // its velocity Pid (JFC.cpp:113-115): those rows start from zero
    const bool reset_vel = (mode != 2);
    const int v0 = L.block(1, 0), v1 = v0 + L.pid_rows();
    for (int row = 0; row < L.total_rows(); ++row) {
      const float v = (reset_vel && row >= v0 && row < v1) ? 0.f : g.src_rec[(size_t)row * g.src_rstride + rr];
      if (live) g.rec[(size_t)row * rs + col] = v;
    }
    mode = 2;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

  Platform s;
  s.px = p0.x; s.py = p0.y; s.pz = p0.z; s.qx = p0.w;
  s.qy = p1.x; s.qz = p1.y; s.qw = p1.z; s.vx = p1.w;
  s.vy = p2.x; s.vz = p2.y; s.wx = p2.z; s.wy = p2.w;
  s.wz = p3.x;
  float fkx = p3.y, fky = p3.z, fkz = p3.w, fkqx = p4.x, fkqy = p4.y, fkqz = p4.z, fkqw = p4.w;
  float cost = 0.f, refx = 0.f, refy = 0.f, refz = 0.f;
  if (ROLLOUT) {
    refx = a.roll_ref[(size_t)rr * 3 + 0];
    refy = a.roll_ref[(size_t)rr * 3 + 1];
    refz = a.roll_ref[(size_t)rr * 3 + 2];
  }
  // the latched command of the lane's mode (constant over the launch unless this is a rollout)
  const float* cmd_src = (mode == 2) ? g.vel_cmd : (mode == 1) ? g.pos_cmd : g.frc_cmd;
  float target[N];
#pragma unroll
  for (int i = 0; i < N; ++i) target[i] = 0.f;
  if (!ROLLOUT) {
    if (g.mode_arr) {  // per-robot modes: the buffer differs from lane to lane
#pragma unroll
      for (int i = 0; i < N; ++i) target[i] = cmd_src ? cmd_src[(size_t)rr * N + i] : 0.f;
    } else if (cmd_src) {
      const float* cp = cmd_src + (size_t)rr * N;
      if (N % 4 == 0) {
#pragma unroll
        for (int q4 = 0; q4 < N / 4; ++q4) {
          const float4 v = reinterpret_cast<const float4*>(cp)[q4];
          target[4 * q4] = v.x, target[4 * q4 + 1] = v.y, target[4 * q4 + 2] = v.z, target[4 * q4 + 3] = v.w;
        }
      } else {
#pragma unroll
        for (int i = 0; i < N; ++i) target[i] = cp[i];
      }
    }
  }

  for (int step = 0; step < a.nsteps; ++step) {
    const int now = g.now_step + step;
    const bool first_world = (step == 0) && (a.flags & kFlagFirstWorldStep);
    if (ROLLOUT) {  // this step's Joy for this trajectory
      const float* cp = a.roll_cmd + (((size_t)rr * a.nsteps + step) * a.roll_samples + sample) * N;
#pragma unroll
      for (int i = 0; i < N; ++i) target[i] = cp[i];
    }
    if (step > 0) {
      // the records this step reads were written by the step before (own global stores, and the fit pass reads other
      // lanes' columns): make them visible past the vector L1
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    // ---- which Pid serves each cable this step (JFC.cpp:67-89), and its rows on their way to LDS
    int sel[N];           // 0 position Pid, 1 velocity Pid (Force mode: 0, unused)
    uint32_t boff[N];     // byte offset of this lane's column in the selected Pid's rows, relative to the position Pid's
    const uint32_t cbytes = col * 4u;
#pragma unroll
    for (int i = 0; i < N; ++i) {
      sel[i] = (mode == 2 && fabsf(target[i]) > g.eps) ? 1 : 0;
      // (opaque: in a launch of several steps the selection is the same in every step, and the compiler would hoist the
      //  selected gains of all cables - some hundred registers - out of the step loop)
      asm volatile("" : "+v"(sel[i]));
      boff[i] = ((uint32_t)(sel[i] * L.pid_rows()) * rs + col) * 4u;
    }
    const bool run_ctl = !first_world;
    if (STAGE && run_ctl) {
      float keep = (s.px + s.qy) + (s.vy + s.wz) + fkqw;  // every ordinary load issued so far is consumed before the DMA is queued
#pragma unroll
      for (int i = 0; i < N; ++i) keep += target[i];
      asm volatile("" ::"v"(keep));
#pragma unroll
      for (int i = 0; i < N; ++i) {
        const int b0 = L.block(0, i);
#pragma unroll
        for (int j = 0; j < NBMAX; ++j)
          if (j < L.nb) RB.to_lds(b0 + j, boff[i], &stage[STAGE ? i : 0][STAGE ? j : 0][0]);
        RB.to_lds(b0 + L.r_ierr(), boff[i], &stage[STAGE ? i : 0][STAGE ? NBMAX : 0][0]);
        RB.to_lds(b0 + L.r_meta(), boff[i], &stage[STAGE ? i : 0][STAGE ? NBMAX + 1 : 0][0]);
        RB.to_lds(b0 + L.r_last(), boff[i], &stage[STAGE ? i : 0][STAGE ? NBMAX + 2 : 0][0]);
        if (g.any_noclamp) RB.to_lds(b0 + L.r_cmd(), boff[i], &stage[STAGE ? i : 0][STAGE ? NBMAX + 3 : 0][0]);
        RB.to_lds(i, cbytes, &stage[STAGE ? i : 0][STAGE ? NBMAX + 4 : 0][0]);
      }
    }

    // ---- IK on the state at t_k: only joint positions, rates and the measured lengths live on; the structure matrix is
    //      rebuilt where it is needed again (after the controller: 56 registers that would otherwise ride through it)
    v2f len[NP], q[NP], qd[NP];
    {
      v2f l0[NP], jac[NP][6];
      ik_pairs<N, true>(lds, s.px, s.py, s.pz, s.qx, s.qy, s.qz, s.qw, len, jac, l0);
#pragma unroll
      for (int k = 0; k < NP; ++k) {
        q[k] = l0[k] - len[k];
        qd[k] = -fma2(s.wz, jac[k][5], fma2(s.wy, jac[k][4], fma2(s.wx, jac[k][3],
                      fma2(s.vz, jac[k][2], fma2(s.vy, jac[k][1], splat(s.vx) * jac[k][0])))));
      }
    }
    const bool publish = !ROLLOUT && ((a.publish_mask >> step) & 1ull) != 0ull;
    float4* const obs = a.obs + (size_t)step * a.obs_step_stride;
    if (publish && live) {  // the part of the observables that is final already (PLG.cpp:248-280)
      store_slot(obs, st, 0, woff, make_float4(s.px, s.py, s.pz, s.qx));
      store_slot(obs, st, 1, woff, make_float4(s.qy, s.qz, s.qw, s.vx));
      store_slot(obs, st, 2, woff, make_float4(s.vy, s.vz, s.wx, s.wy));
#pragma unroll
      for (int gI = 0; gI < G; ++gI) {
        const int k0 = 2 * gI, k1 = (2 * gI + 1 < NP) ? 2 * gI + 1 : 2 * gI;
        const bool has = (2 * gI + 1 < NP);
        store_slot(obs, st, 4 + gI, woff, make_float4(q[k0].x, q[k0].y, has ? q[k1].x : 0.f, has ? q[k1].y : 0.f));
        store_slot(obs, st, 4 + G + gI, woff, make_float4(qd[k0].x, qd[k0].y, has ? qd[k1].x : 0.f, has ? qd[k1].y : 0.f));
      }
    }

    // ---- Newton-Raphson forward kinematics ([NEW] SURVEY 8(a) row 14)
    float fk_res = 0.f;
    int fk_it = 0, td_flag = 0;
    if (FK) {
      v2f elen[NP], unused[NP], jest[NP][6];
      bool active = true;
      for (int it = 0; it < a.fk_iters; ++it) {
        ik_pairs<N, false>(lds, fkx, fky, fkz, fkqx, fkqy, fkqz, fkqw, elen, jest, unused);
        v2f res[NP];
        v2f rm = splat(0.f);
#pragma unroll
        for (int k = 0; k < NP; ++k) {
          res[k] = len[k] - elen[k];
          rm = max2(rm, abs2(res[k]));
        }
        active = active && !(fmaxf(rm.x, rm.y) < a.fk_tol);
        float gg[6];
        jt_times<NP>(jest, res, gg);
        normal_solve<NP>(jest, a.fk_lambda, gg);
        if (active) {
          fkx += gg[0];
          fky += gg[1];
          fkz += gg[2];
          quat_apply_rotvec(fkqx, fkqy, fkqz, fkqw, gg[3], gg[4], gg[5]);
          ++fk_it;
        }
      }
      ik_pairs<N, false>(lds, fkx, fky, fkz, fkqx, fkqy, fkqz, fkqw, elen, jest, unused);
      v2f rm = splat(0.f);
#pragma unroll
      for (int k = 0; k < NP; ++k) rm = max2(rm, abs2(len[k] - elen[k]));
      fk_res = fmaxf(rm.x, rm.y);
    }

    // ---- per-cable force (PLG.cpp:222-228 -> JFC.cpp:59-96 -> Pid.cpp:122-191).  Only q, qd and the force live across the
    //      cables: a cable whose derivative is known at once (uniform window: the FIR; window not full: 0) runs its whole
    //      Pid::update here; one that needs the fit parks six numbers in its own staged rows and finishes after the pass.
    //      Written WITHOUT divergent control flow around anything heavy (per-lane cases are selects and predicated
    //      stores): the register allocator splits long live ranges around high-pressure regions, and a split made under a
    //      partial exec mask does not carry the lanes that were masked off.
    float force[N];
#pragma unroll
    for (int i = 0; i < N; ++i) force[i] = 0.f;
    float dbg_p = 0.f, dbg_i = 0.f, dbg_d = 0.f, dbg_des = 0.f;
    bool dbg_pi = false, dbg_dw = false;
    if (run_ctl) {
      if (STAGE) {
        __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): the DMA has landed
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      }
      const int pcas_max = max(g.pid[0].pcas, g.pid[1].pcas), dcas_max = max(g.pid[0].dcas, g.pid[1].dcas);
      const bool noclamp = g.any_noclamp != 0;
      uint32_t need = 0u;  // cables whose derivative comes from the fit queue
#pragma unroll
      for (int i = 0; i < N; ++i) {
        __builtin_amdgcn_sched_barrier(0);  // one cable at a time: hoisting every cable's staged rows costs 128 registers
        const float qi = (i & 1) ? q[i / 2].y : q[i / 2].x;
        const float qdi = (i & 1) ? qd[i / 2].y : qd[i / 2].x;
        const int b0 = L.block(0, i);
        const uint32_t bo = boff[i];
        float* const sg = STAGE ? &stage[STAGE ? i : 0][0][lane] : &park[0][STAGE ? 0 : i][lane];  // this cable's staged rows, kSg floats apart
        const bool is_force = (mode == 0);  // JFC.cpp:67-70
        const bool sv = sel[i] != 0;
        const bool hold = (mode == 2) && !sv;
        const float held = STAGE ? sg[(NBMAX + 4) * kSg] : RB.load(i, cbytes);
        const float desired = hold ? held : target[i];  // JFC.cpp:81: mLastPosition in the hold branch
        if (live && !hold) RB.store(i, cbytes, qi);      // JFC.cpp:68,75,87: mLastPosition = joint position
        const float actual = (mode == 2 && sv) ? qdi : qi;
        const GenSel c = gen_select(sv, g.pid[0], g.pid[1]);
        const float kf = c.kf, kp = c.kp, ki = c.ki, imax = c.imax, imin = c.imin;
        const int nbuf = c.nbuf;
        const uint32_t meta = __float_as_uint(STAGE ? sg[(NBMAX + 1) * kSg] : RB.load(b0 + L.r_meta(), bo));
        const int last = __float_as_int(STAGE ? sg[(NBMAX + 2) * kSg] : RB.load(b0 + L.r_last(), bo));
        const bool first = !is_force && !(meta & kGmWasLast);  // Pid.cpp:123-126: the first call since reset returns 0
        const bool runs = !is_force && !first;
        const float prev_ierr = STAGE ? sg[NBMAX * kSg] : RB.load(b0 + L.r_ierr(), bo);
        const float old_cmd = g.any_noclamp ? (STAGE ? sg[(NBMAX + 3) * kSg] : RB.load(b0 + L.r_cmd(), bo)) : 0.f;
        const float error = desired - actual;
        const float dt = (float)(now - last) * g.dt;
        float perr = error;
        if (pcas_max) perr = gen_cascade(RB, b0 + L.r_pfilt(), bo, pcas_max, c.pcas, runs && live, c.pa0, c.pa1, c.pa2, c.pb1, c.pb2, error);
        const float p_term = kp * perr;
        float ie = fmaf(dt, error, prev_ierr);
        float i_term = ki * ie;
        if (i == 0) {  // `pid` topic (Pid.cpp:139-142,158-159): what the Pid call of cable 0 writes, when there is one
          dbg_p = runs ? p_term : dbg_p;
          dbg_i = runs ? i_term : dbg_i;
          dbg_des = runs ? desired : dbg_des;
          dbg_pi = runs;
        }
        const bool over = i_term > imax, under = i_term < imin;  // Pid.cpp:143-152
        i_term = over ? imax : (under ? imin : i_term);
        ie = (over || under) ? i_term / ki : ie;
        const float pre = (kf * desired + p_term) + i_term;  // Pid.cpp:170: fTerm + pTerm + iTerm (+ dTerm in finish)
        // Pid::derive (Pid.cpp:193-217): push the sample (dt > 0 always: a Pid is called at most once per world step)
        const int count = (int)((meta >> kGmCountShift) & kGmField), head = (int)((meta >> kGmHeadShift) & kGmField);
        const int run = (int)((meta >> kGmRunShift) & kGmField);
        const int nhead = (count == 0) ? 0 : ((head + 1 == nbuf) ? 0 : head + 1);
        const int ncount = min(count + 1, nbuf);
        const int nrun = (count > 0 && now - last == 1) ? min(run + 1, (int)kGmField) : 0;
        if (live && !is_force) {
          const uint32_t nmeta = first ? (meta | kGmWasLast)
                                       : (kGmWasLast | ((uint32_t)ncount << kGmCountShift) | ((uint32_t)nhead << kGmHeadShift) | ((uint32_t)nrun << kGmRunShift));
          RB.storei(b0 + L.r_meta(), bo, (int)nmeta);
          RB.storei(b0 + L.r_last(), bo, now);
          if (g.any_noclamp && first) RB.store(b0 + L.r_cmd(), bo, 0.f);
          if (runs) {
            const uint32_t ho = bo + (uint32_t)nhead * rs * 4u;  // the ring slot is the lane's own
            RB.store(b0, ho, error);
            RB.storei(b0 + L.r_stamp(), ho, now);
          }
        }
        // full window that is not a uniform grid (the nbuf - 1 steps after a switch between the two Pids): queued for the fit
        const bool queued = runs && ncount >= nbuf && nrun < nbuf - 1;
        // nbuf samples one world step apart: the closed-form end-point LS derivative, weights by ring head (zero for slots >= nbuf)
        const float* wr = &wrot[sv ? 1 : 0][nhead][0];
        float acc = 0.f;
#pragma unroll
        for (int j = 0; j < NBMAX; ++j) {
          const float yj = (j == nhead) ? error : (STAGE ? sg[j * kSg] : ((j < L.nb) ? RB.load(b0 + j, bo) : 0.f));
          acc = fmaf(wr[j], yj, acc);
        }
        const float derived = (ncount >= nbuf) ? acc / g.dt : 0.f;  // mDbufferMissing != 0: derive() returns 0 (Pid.cpp:200-203)
        float d_term;
        const float out = gen_finish(RB, c, b0 + L.r_ierr(), b0 + L.r_cmd(), b0 + L.r_dfilt(), dcas_max, noclamp, runs && !queued && live, bo, derived, error, dt,
                                     pre, ie, prev_ierr, old_cmd, d_term);
        if (i == 0) {
          dbg_d = (runs && !queued) ? d_term : dbg_d;
          dbg_dw = runs && !queued;
        }
        force[i] = is_force ? target[i] : ((runs && !queued) ? out : 0.f);
        need |= queued ? (1u << i) : 0u;
        if (queued) {  // park what the rest of Pid::update needs in the cable's own staged rows (the values there are done with)
          sg[0 * kSg] = error, sg[1 * kSg] = dt, sg[2 * kSg] = pre, sg[3 * kSg] = ie, sg[4 * kSg] = prev_ierr, sg[5 * kSg] = old_cmd;
          sg[6 * kSg] = __int_as_float(nhead);
        }
      }
      __builtin_amdgcn_sched_barrier(0);

      // the windows that are not a uniform grid, compacted over the wave: one (robot, cable) per lane and pass
      if (__builtin_amdgcn_ballot_w64(need != 0u) != 0ull) {  // (wave-uniform)
        const uint32_t cnt = (uint32_t)__builtin_popcount(need);
        uint32_t slot = 0u;
        if (cnt) slot = __hip_atomic_fetch_add(&q_count, cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#pragma unroll
        for (int i = 0; i < N; ++i) {
          if (need & (1u << i)) {
            const float* sg = STAGE ? &stage[STAGE ? i : 0][0][lane] : &park[0][STAGE ? 0 : i][lane];
            q_item[slot] = lane | ((uint32_t)i << 6) | ((uint32_t)sel[i] << 9) | ((uint32_t)__float_as_int(sg[6 * kSg]) << 10);
            q_err[slot] = sg[0];
            ++slot;
          }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const uint32_t total = q_count;
        for (uint32_t first = 0; first < total; first += 64u) {  // (wave-uniform trip count)
          // every lane runs the fit - lanes past the end of the queue on a copy of item 0 - and only the result is predicated
          const uint32_t idx = first + lane;
          const bool mine = idx < total;
          const uint32_t it = q_item[mine ? idx : 0u];
          const float e_new = q_err[mine ? idx : 0u];
          const uint32_t ol = it & 63u, ci = (it >> 6) & 7u, sp = (it >> 9) & 1u, hd = (it >> 10) & 63u;
          const uint32_t ro = blockIdx.x * 64u + ol;
          const uint32_t ocol = (ro < units) ? ro : (units - 1u);
          const uint32_t oo = ((uint32_t)(L.block(0, (int)ci) + (int)sp * L.pid_rows()) * rs + ocol) * 4u;  // the item's block, its owner's column
          const int nbuf = sp ? g.pid[1].nbuf : g.pid[0].nbuf, degree = sp ? g.pid[1].degree : g.pid[0].degree;
          float y[NBMAX];
          int t[NBMAX];
#pragma unroll
          for (int j = 0; j < NBMAX; ++j) {
            y[j] = RB.load(j, oo);
            t[j] = RB.loadi(L.r_stamp() + j, oo);
          }
          const uint32_t old = (hd + 1u == (uint32_t)nbuf) ? 0u : hd + 1u;  // the oldest sample sits right after the head
          int t_old = now;
#pragma unroll
          for (int j = 0; j < NBMAX; ++j) {
            y[j] = ((uint32_t)j == hd) ? e_new : ((j < nbuf) ? y[j] : 0.f);  // the sample just pushed (its store may still be in flight)
            t[j] = ((uint32_t)j == hd || j >= nbuf) ? now : t[j];
            t_old = ((uint32_t)j == old) ? t[j] : t_old;
          }
          const float res = (float)(gen_fit<NBMAX>(y, t, nbuf, degree, now, t_old) / (double)g.dt);
          if (mine) q_res[ci][ol] = res;
        }
        if (lane == 0) q_count = 0u;  // for the next step
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int i = 0; i < N; ++i) {
          const bool queued = (need & (1u << i)) != 0u;
          const float* sg = STAGE ? &stage[STAGE ? i : 0][0][lane] : &park[0][STAGE ? 0 : i][lane];
          const int b0 = L.block(0, i);
          float d_term;
          const float out = gen_finish(RB, gen_select(sel[i] != 0, g.pid[0], g.pid[1]), b0 + L.r_ierr(), b0 + L.r_cmd(), b0 + L.r_dfilt(), dcas_max, noclamp,
                                       queued && live, boff[i], q_res[i][lane], sg[0 * kSg], sg[1 * kSg], sg[2 * kSg], sg[3 * kSg], sg[4 * kSg], sg[5 * kSg], d_term);
          if (i == 0) {
            dbg_d = queued ? d_term : dbg_d;
            dbg_dw = dbg_dw || queued;
          }
          force[i] = queued ? out : force[i];
        }
      }
    }
    v2f f[NP];
#pragma unroll
    for (int k = 0; k < NP; ++k) f[k] = (v2f){force[2 * k], (2 * k + 1 < N) ? force[2 * k + 1] : 0.f};

    // ---- optional tension distribution ([NEW] SURVEY 8(a) row 15), SetForce limits
    v2f applied[NP];
    if (TD) {
      v2f df[NP];
#pragma unroll
      for (int k = 0; k < NP; ++k) df[k] = f[k] - splat(a.td_mid);
      // structure matrix at the FK estimate (at the true pose without an estimator), rebuilt: same inputs, same bits.  The
      // geometry offset is opaque so that the compiler cannot keep the earlier evaluation alive instead.
      uint32_t again = 0;
      asm volatile("" : "+v"(again));
      v2f jtd[NP][6], ltd[NP], l0td[NP];
      if (FK)
        ik_pairs<N, false>(lds + again, fkx, fky, fkz, fkqx, fkqy, fkqz, fkqw, ltd, jtd, l0td);
      else
        ik_pairs<N, false>(lds + again, s.px, s.py, s.pz, s.qx, s.qy, s.qz, s.qw, ltd, jtd, l0td);
      float gg[6];
      jt_times<NP>(jtd, df, gg);
      normal_solve<NP, false>(jtd, 0.f, gg);
#pragma unroll
      for (int k = 0; k < NP; ++k) {
        v2f t = splat(a.td_mid);
#pragma unroll
        for (int c = 0; c < 6; ++c) t = fma2(gg[c], jtd[k][c], t);
        const v2f tc = max2(min2(t, splat(a.td_max)), splat(a.td_min));
        td_flag |= (tc.x != t.x) ? 1 : 0;
        if (2 * k + 1 < N) td_flag |= (tc.y != t.y) ? 1 : 0;
        applied[k] = tc;
      }
    } else {
#pragma unroll
      for (int k = 0; k < NP; ++k) applied[k] = f[k];
    }
    if (a.vel_limit > 0.f) {  // Joint::SetForce velocity truncation [EXT]
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

    if (!ROLLOUT && a.dbg && live) {  // `pid` topic, cable 0 only: stale entries stay (Pid.cpp:139-142,158-168)
      float* d = a.dbg + (size_t)r * 9;
      if (dbg_pi) {
        d[0] = dbg_p;
        d[1] = dbg_i;
        d[3] = dbg_des;
      }
      if (dbg_dw) d[2] = dbg_d;
      d[4] = applied[0].x;
    }
    if (publish && live) {  // the rest of the observables
      store_slot(obs, st, 3, woff, make_float4(s.wz, fk_res, (float)fk_it, pack_flags(td_flag, travel_mask<N>(a, q))));
#pragma unroll
      for (int gI = 0; gI < G; ++gI) {
        const int k0 = 2 * gI, k1 = (2 * gI + 1 < NP) ? 2 * gI + 1 : 2 * gI;
        const bool has = (2 * gI + 1 < NP);
        store_slot(obs, st, 4 + 2 * G + gI, woff,
                   make_float4(applied[k0].x, applied[k0].y, has ? applied[k1].x : 0.f, has ? applied[k1].y : 0.f));
      }
    }

    // ---- world step to t_{k+1}: wrench = -J^T (applied - d qdot) + m g; the optional physics terms by run-time flags
    {
      v2f tens[NP];
#pragma unroll
      for (int k = 0; k < NP; ++k) {
        tens[k] = fma2(-a.damping, qd[k], applied[k]);
        if (a.unilateral) tens[k] = max2(tens[k], splat(0.f));
      }
      uint32_t again = 0;  // the true structure matrix, rebuilt (see above)
      asm volatile("" : "+v"(again));
      v2f jac[NP][6], l0w[NP];
      ik_pairs<N, false>(lds + again, s.px, s.py, s.pz, s.qx, s.qy, s.qz, s.qw, len, jac, l0w);
      float w[6];
      jt_times<NP>(jac, tens, w);
      w[0] = a.fgx - w[0];
      w[1] = a.fgy - w[1];
      w[2] = a.fgz - w[2];
      w[3] = -w[3];
      w[4] = -w[4];
      w[5] = -w[5];
      if (a.ph_lumped)
        integrate_lumped_velocity<N>(a, lds, s, jac, len, w);
      else
        integrate_velocity(a, s, w);
      if (a.travel_stop) apply_travel_stop<N>(a, s, q, jac);
      integrate_pose(a, s);
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
  if (live) {
    CDPR_STORE_STATE(a.state, st, 0, woff, make_float4(s.px, s.py, s.pz, s.qx));
    CDPR_STORE_STATE(a.state, st, 1, woff, make_float4(s.qy, s.qz, s.qw, s.vx));
    CDPR_STORE_STATE(a.state, st, 2, woff, make_float4(s.vy, s.vz, s.wx, s.wy));
    CDPR_STORE_STATE(a.state, st, 3, woff, make_float4(s.wz, fkx, fky, fkz));
    if (FK) CDPR_STORE_STATE(a.state, st, 4, woff, make_float4(fkqx, fkqy, fkqz, fkqw));
  }
}

// Per-robot command arrival on the general path (cdpr_set_*_command_masked; PLG.cpp:206-219 per model): one thread per
// robot copies the Joy's row into the latched buffer of its kind, and entering the mode from another one resets that
// mode's Pid (JFC.cpp:101-103,113-115) = zero the robot's column of the Pid's rows.  setForce resets nothing.
struct GenLatchArgs {
  const uint8_t* mask;   // uint8[B], or nullptr = every robot
  uint8_t* mode;         // per-robot mode
  const float* pending;  // float[B][n]
  float* latched;        // float[B][n]
  float* rec;
  uint32_t rstride, batch, n;
  int first_row, rows;   // the Pid's rows (rows = 0: nothing to reset)
  int new_mode;
};

static __global__ __launch_bounds__(256) void cdpr_gen_latch_kernel(const GenLatchArgs a) {
  const uint32_t r = blockIdx.x * 256u + threadIdx.x;
  if (r >= a.batch) return;
  if (a.mask && !a.mask[r]) return;
  for (uint32_t i = 0; i < a.n; ++i) a.latched[(size_t)r * a.n + i] = a.pending[(size_t)r * a.n + i];
  if ((int)a.mode[r] != a.new_mode) {
    for (int row = 0; row < a.rows; ++row) a.rec[(size_t)(a.first_row + row) * a.rstride + r] = 0.f;
    a.mode[r] = (uint8_t)a.new_mode;
  }
}

}  // namespace cdpr
