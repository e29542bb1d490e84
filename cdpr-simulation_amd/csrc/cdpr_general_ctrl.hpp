// cdpr_general_ctrl.hpp — general controller path (gfx950): everything the register-resident
// fast path of cdpr_step_kernel.hpp cannot represent.
//
//   * position-hold branch of JointForceCalculator::update (JFC.cpp:78-82, velocityEpsilon >= 0):
//     both Pids of a cable stay alive in Velocity mode and are sampled at non-uniform times;
//   * biquad cascades on the P and D inputs (Pid::CascadeFilter, Pid.cpp:27-44; Filter.h:130-165);
//   * derivative windows up to 32 samples / degree 4, fitted on the real sample stamps
//     (Pid::derive + fitPolynomial, Pid.cpp:193-247) instead of the uniform-grid FIR;
//   * cmdLimit == 0 (no command clamp: mCmd keeps its old value, Pid.cpp:175-186).
//
// One thread owns one (robot, cable): it evaluates the cable's IK row on the state at t_k, runs the
// mode machine and the Pid, and writes the raw force.  The platform kernel (cdpr_step_kernel with
// EXT = true) then reads the forces and does FK / TD / observables / world step.  Records live in
// HBM as struct-of-arrays over T = B*n threads, one contiguous block per Pid so a Pid::reset of the
// whole batch is one memset.  The polynomial fit is done in fp64 on centred, scaled stamps.
#pragma once
#include "cdpr_step_kernel.hpp"

namespace cdpr {

constexpr int kGenMaxBuf = 32;   // CDPR_MAX_D_BUFFER
constexpr int kGenMaxDeg = 4;    // CDPR_MAX_D_DEGREE
constexpr int kGenMaxCas = 4;    // CDPR_MAX_CASCADE

// fields of one Pid record (each a row of T floats).  The window and filter rows are sized by the CONFIGURED window
// length and cascade count (GenLayout), not by the maxima: the shipped 11-sample window without cascades is 29 rows per
// Pid (236 B per cable for both Pids and the hold position) instead of the 103 (828 B) a 32-sample / 4-cascade record takes.
enum GenField : int {
  kGfWasLast = 0,  // Pid::mWasLastTime
  kGfLastStep,     // Pid::mLastTime as a world-step index
  kGfIerr,
  kGfDerr,
  kGfCmd,
  kGfCount,        // samples pushed since reset (mDbufferLength - mDbufferMissing, saturating)
  kGfHead,         // ring index of the newest sample
  kGfWinVal,       // nb rows: mDbufferY; then nb rows mDbufferX as world-step indices; then ncas x (x1 x2 y1 y2) for the P
                   // input's cascade and the same for the D input's
};

// Row offsets inside one Pid block for a handle whose longest window is nb samples and deepest cascade ncas stages.
struct GenLayout {
  int nb, ncas;
  __host__ __device__ int win_val() const { return kGfWinVal; }
  __host__ __device__ int win_stamp() const { return kGfWinVal + nb; }
  __host__ __device__ int p_filt() const { return kGfWinVal + 2 * nb; }
  __host__ __device__ int d_filt() const { return kGfWinVal + 2 * nb + 4 * ncas; }
  __host__ __device__ int rows() const { return kGfWinVal + 2 * nb + 8 * ncas; }
};

struct GenPid {
  float kf, kp, ki, kd, imax, imin, cmax, cmin;
  int nbuf, degree, pcas, dcas;
  float pa0, pa1, pa2, pb1, pb2;  // BiQuad::SetFc(relCutoff, 1.0, quality), Filter.h:130-140
  float da0, da1, da2, db1, db2;
};

struct GenArgs {
  const float4* state;  // platform slots of the step kernel
  size_t stride;
  uint32_t batch, n;
  const float* cable;   // plain per-cable geometry: ax ay az bx by bz l0, 7 rows of n
  const float* vel_cmd; // latched jointVelocities, float[B][n] (or nullptr)
  const float* pos_cmd; // latched jointPositions, float[B][n] (or nullptr -> target 0)
  const float* frc_cmd; // latched force command (setForce, JFC.h:92-95), float[B][n] (or nullptr -> 0)
  float* rec;           // [last_pos row][pos Pid block][vel Pid block], rows of `tstride` floats
  size_t tstride;
  float* force;         // out: raw force per cable, float[B][n]
  float* dbg;           // `pid` topic (cable 0), float[B][9], or nullptr
  int mode;             // 1 = Position, 2 = Velocity (JFC.h:35-37): the whole batch, when mode_arr is null
  const uint8_t* mode_arr;  // per-robot mode (handles created with per_robot_commands), uint8[B], or nullptr
  int first_world;      // t = 0: stepTime <= 0 -> force 0, nothing else (JFC.cpp:61-66)
  int now_step;
  float eps, dt;
  GenLayout lay;        // row offsets of a Pid block
  GenPid pid[2];        // [0] position Pid, [1] velocity Pid
};

__host__ __device__ inline size_t gen_record_rows(const GenLayout& l) { return 1 + 2 * (size_t)l.rows(); }

struct GenRec {
  float* base;  // this thread's column of its Pid block
  size_t ts;
  __device__ __forceinline__ float& f(int row) { return base[(size_t)row * ts]; }
  // integer-valued fields (world-step stamps, ring head, sample count) are kept as int32 bit patterns in the
  // float rows: a float holds integers exactly only up to 2^24 (4.6 h of sim time at 1 ms), an int32 to 2^31;
  // an all-zero row (Pid::reset = memset) reads as 0 either way
  __device__ __forceinline__ int geti(int row) { return __float_as_int(base[(size_t)row * ts]); }
  __device__ __forceinline__ void seti(int row, int v) { base[(size_t)row * ts] = __int_as_float(v); }
};

__device__ __forceinline__ float gen_cascade(GenRec& r, int first_row, int cascade, float a0, float a1, float a2, float b1,
                                             float b2, float x) {
  // Pid::CascadeFilter::update (Pid.cpp:38-44) over BiQuad::process (Filter.h:152-165)
  float out = x;
  for (int c = 0; c < cascade; ++c) {
    float& x1 = r.f(first_row + 4 * c + 0);
    float& x2 = r.f(first_row + 4 * c + 1);
    float& y1 = r.f(first_row + 4 * c + 2);
    float& y2 = r.f(first_row + 4 * c + 3);
    const float y0 = a0 * out + a1 * x1 + a2 * x2 - b1 * y1 - b2 * y2;
    x2 = x1;
    x1 = out;
    y2 = y1;
    y1 = y0;
    out = y0;
  }
  return out;
}

// Pid::derive + fitPolynomial (Pid.cpp:193-247) on the real stamps, fp64, centred and scaled time.  NBMAX bounds the
// window at compile time (11: the shipped length and everything below it; 32: the maximum): the window's 2 x nbuf rows
// are loaded in ONE unrolled, predicated batch before anything is computed (with a run-time trip count every iteration
// waited for its own two loads: 2 x 11 dependent round trips, most of the kernel's 13-20 us), and the power sums run over
// registers.
template <int NBMAX>
__device__ __forceinline__ float gen_derive(const GenPid& p, GenRec& r, const GenLayout& lay, float value, int now_step, float dt) {
  const int nbuf = p.nbuf;
  const int wv = lay.win_val(), ws = lay.win_stamp();
  int head = r.geti(kGfHead);
  int count = r.geti(kGfCount);
  float y[NBMAX];
  int t[NBMAX];
#pragma unroll
  for (int j = 0; j < NBMAX; ++j) {
    y[j] = (j < nbuf) ? r.f(wv + j) : 0.f;
    t[j] = (j < nbuf) ? r.geti(ws + j) : 0;
  }
  head = (count == 0) ? 0 : ((head + 1 == nbuf) ? 0 : head + 1);
  r.f(wv + head) = value;
  r.seti(ws + head, now_step);
  r.seti(kGfHead, head);
  if (count < nbuf) ++count;
  r.seti(kGfCount, count);
  if (count < nbuf) return 0.f;  // mDbufferMissing != 0 (Pid.cpp:200-203)
#pragma unroll
  for (int j = 0; j < NBMAX; ++j) {  // the sample just pushed, in registers too
    y[j] = (j == head) ? value : y[j];
    t[j] = (j == head) ? now_step : t[j];
  }

  // oldest sample sits right after the head in the ring
  const int oldest = (head + 1 == nbuf) ? 0 : head + 1;
  int t_old_i = 0;
  double mean = 0.0;
#pragma unroll
  for (int j = 0; j < NBMAX; ++j) {
    if (j < nbuf) mean += (double)t[j];
    t_old_i = (j == oldest) ? t[j] : t_old_i;
  }
  const double t_new = (double)now_step, t_old = (double)t_old_i;
  mean /= (double)nbuf;
  double h = (t_new - t_old) / (double)(nbuf - 1);
  if (!(h > 0.0)) h = 1.0;
  const double inv_h = 1.0 / h;
  const int m = p.degree + 1;
  double sx[2 * kGenMaxDeg + 1], sb[kGenMaxDeg + 1];
#pragma unroll
  for (int i = 0; i < 2 * kGenMaxDeg + 1; ++i) sx[i] = 0.0;
#pragma unroll
  for (int i = 0; i < kGenMaxDeg + 1; ++i) sb[i] = 0.0;
#pragma unroll
  for (int j = 0; j < NBMAX; ++j) {
    if (j < nbuf) {
      const double x = ((double)t[j] - mean) * inv_h;
      const double yy = (double)y[j];
      double pw = 1.0;
#pragma unroll
      for (int i = 0; i < 2 * kGenMaxDeg + 1; ++i) {  // powers beyond the fit's degree are summed too and never used
        sx[i] += pw;
        if (i < kGenMaxDeg + 1) sb[i] += pw * yy;
        pw *= x;
      }
    }
  }
  // normal equations of the degree-(m-1) fit in the leading m x m block, identity rows below it: every index in the
  // elimination is a compile-time constant (the matrix stays in registers; a run-time pivot row sends it to scratch memory)
  double a[kGenMaxDeg + 1][kGenMaxDeg + 2];
#pragma unroll
  for (int i = 0; i < kGenMaxDeg + 1; ++i) {
#pragma unroll
    for (int j = 0; j < kGenMaxDeg + 1; ++j) a[i][j] = (i < m && j < m) ? sx[i + j] : ((i == j) ? 1.0 : 0.0);
    a[i][kGenMaxDeg + 1] = (i < m) ? sb[i] : 0.0;
  }
  // Gauss-Jordan without pivoting: the block is symmetric positive definite (centred, scaled abscissae: well conditioned)
#pragma unroll
  for (int col = 0; col < kGenMaxDeg + 1; ++col) {
    const double inv = 1.0 / a[col][col];
#pragma unroll
    for (int rr = 0; rr < kGenMaxDeg + 1; ++rr) {
      if (rr == col) continue;
      const double fct = a[rr][col] * inv;
#pragma unroll
      for (int k = col; k < kGenMaxDeg + 2; ++k) a[rr][k] -= fct * a[col][k];
    }
  }
  // derivative of the fitted polynomial at the newest stamp (Pid.cpp:205-212), back to seconds
  const double xn = (t_new - mean) * inv_h;
  double deriv = 0.0, pw = 1.0;
#pragma unroll
  for (int i = 1; i < kGenMaxDeg + 1; ++i) {
    if (i < m) deriv += (double)i * (a[i][kGenMaxDeg + 1] / a[i][i]) * pw;
    pw *= xn;
  }
  return (float)(deriv / (h * (double)dt));
}

struct GenTerms {
  float p, i, d, desired;
  bool pi_written, d_written, desired_written;
};

// Pid::update (Pid.cpp:122-191)
template <int NBMAX>
__device__ __forceinline__ float gen_pid_update(const GenPid& p, GenRec& r, const GenLayout& lay, float desired, float actual, int now_step,
                                                float dt_step, GenTerms& t) {
  t.pi_written = t.d_written = t.desired_written = false;
  float cmd_out;
  if (r.f(kGfWasLast) == 0.f) {
    r.f(kGfWasLast) = 1.f;
    r.f(kGfCmd) = 0.f;
    cmd_out = 0.f;
  } else {
    const float f_term = p.kf * desired;
    const float error = desired - actual;
    const float dt = (float)(now_step - r.geti(kGfLastStep)) * dt_step;
    const float perr = gen_cascade(r, lay.p_filt(), p.pcas, p.pa0, p.pa1, p.pa2, p.pb1, p.pb2, error);
    const float p_term = p.kp * perr;
    const float prev_ierr = r.f(kGfIerr);
    float ierr = fmaf(dt, error, prev_ierr);
    float i_term = p.ki * ierr;
    t.p = p_term;
    t.i = i_term;
    t.pi_written = true;
    if (i_term > p.imax) {
      i_term = p.imax;
      ierr = i_term / p.ki;
    } else if (i_term < p.imin) {
      i_term = p.imin;
      ierr = i_term / p.ki;
    }
    float derr = r.f(kGfDerr);
    if (dt > 0.f) {
      const float derived = gen_derive<NBMAX>(p, r, lay, error, now_step, dt_step);
      derr = gen_cascade(r, lay.d_filt(), p.dcas, p.da0, p.da1, p.da2, p.db1, p.db2, derived);
      r.f(kGfDerr) = derr;
      t.desired = desired;
      t.desired_written = true;
    }
    const float d_term = p.kd * derr;
    t.d = d_term;
    t.d_written = true;
    const float cmd = f_term + p_term + i_term + d_term;
    float out = r.f(kGfCmd);
    if (p.cmax > p.cmin) out = fmaxf(fminf(cmd, p.cmax), p.cmin);
    if (out != cmd) {
      ierr = prev_ierr;
      out = fmaf(dt * error, p.ki, out);
    }
    r.f(kGfIerr) = ierr;
    r.f(kGfCmd) = out;
    cmd_out = out;
  }
  r.seti(kGfLastStep, now_step);
  return cmd_out;
}

template <int NBMAX>
__global__ __launch_bounds__(256) void cdpr_general_ctrl_kernel(const GenArgs a) {
  const uint32_t t = blockIdx.x * 256u + threadIdx.x;
  const uint32_t total = a.batch * a.n;
  if (t >= total) return;
  const uint32_t r = t / a.n, i = t - r * a.n;
  if (a.first_world) {  // JFC.cpp:61-66: nothing but the time stamp moves at t = 0
    a.force[t] = 0.f;
    if (a.dbg && i == 0) a.dbg[(size_t)r * 9 + 4] = 0.f;
    return;
  }
  // IK row of this cable on the state at t_k
  const float4 p0 = a.state[0 * a.stride + r], p1 = a.state[1 * a.stride + r], p2 = a.state[2 * a.stride + r],
               p3 = a.state[3 * a.stride + r];
  const Rot rot = quat_to_rot(p0.w, p1.x, p1.y, p1.z);
  const float ax = a.cable[0 * a.n + i], ay = a.cable[1 * a.n + i], az = a.cable[2 * a.n + i];
  const float bx = a.cable[3 * a.n + i], by = a.cable[4 * a.n + i], bz = a.cable[5 * a.n + i];
  const float l0 = a.cable[6 * a.n + i];
  const float rbx = fmaf(rot.r02, bz, fmaf(rot.r01, by, rot.r00 * bx));
  const float rby = fmaf(rot.r12, bz, fmaf(rot.r11, by, rot.r10 * bx));
  const float rbz = fmaf(rot.r22, bz, fmaf(rot.r21, by, rot.r20 * bx));
  const float lx = (rbx - ax) + p0.x, ly = (rby - ay) + p0.y, lz = (rbz - az) + p0.z;
  const float l2 = fmaf(lz, lz, fmaf(ly, ly, lx * lx));
  const float inv = __frsqrt_rn(l2);
  const float len = l2 * inv;
  const float ux = lx * inv, uy = ly * inv, uz = lz * inv;
  const float j3 = fmaf(rby, uz, -(rbz * uy)), j4 = fmaf(rbz, ux, -(rbx * uz)), j5 = fmaf(rbx, uy, -(rby * ux));
  const float q = l0 - len;
  const float qd = -fmaf(p3.x, j5, fmaf(p2.w, j4, fmaf(p2.z, j3, fmaf(p2.y, uz, fmaf(p2.x, uy, p1.w * ux)))));

  float* last_pos = a.rec + t;
  GenRec pos{a.rec + a.tstride * 1 + t, a.tstride};
  GenRec vel{a.rec + a.tstride * (1 + (size_t)a.lay.rows()) + t, a.tstride};
  GenTerms terms;
  terms.pi_written = terms.d_written = terms.desired_written = false;
  float force = 0.f;
  const int mode = a.mode_arr ? (int)a.mode_arr[r] : a.mode;
  if (mode == 0) {  // Force (JFC.cpp:67-70)
    *last_pos = q;
    force = a.frc_cmd ? a.frc_cmd[t] : 0.f;
  } else if (mode == 2) {  // Velocity (JFC.cpp:71-83)
    const float vt = a.vel_cmd ? a.vel_cmd[t] : 0.f;
    if (fabsf(vt) > a.eps) {
      *last_pos = q;
      force = gen_pid_update<NBMAX>(a.pid[1], vel, a.lay, vt, qd, a.now_step, a.dt, terms);
    } else {
      force = gen_pid_update<NBMAX>(a.pid[0], pos, a.lay, *last_pos, q, a.now_step, a.dt, terms);
    }
  } else {  // Position (JFC.cpp:84-89)
    const float target = a.pos_cmd ? a.pos_cmd[t] : 0.f;
    *last_pos = q;
    force = gen_pid_update<NBMAX>(a.pid[0], pos, a.lay, target, q, a.now_step, a.dt, terms);
  }
  a.force[t] = force;
  if (a.dbg && i == 0) {  // `pid` topic: stale entries stay (Pid.cpp:139-142,158-168)
    float* d = a.dbg + (size_t)r * 9;
    if (terms.pi_written) {
      d[0] = terms.p;
      d[1] = terms.i;
    }
    if (terms.d_written) d[2] = terms.d;
    if (terms.desired_written) d[3] = terms.desired;
  }
}

// Per-robot command arrival (cdpr_set_*_command_masked): what B independent plugin instances do when only some of
// them receive a Joy before the next update() (PLG.cpp:206-219 per model).  One thread per (robot, cable):
// robots in the mask latch their new target; entering the mode from the other one resets that mode's Pid
// (JFC.cpp:101-103,113-115) = zero this thread's column of the Pid's record block.  The mode itself is written by
// cdpr_set_mode_masked_kernel afterwards (the threads of one robot may sit in different waves).
struct LatchArgs {
  const uint8_t* mask;   // uint8[B], or nullptr = every robot
  const uint8_t* mode;   // current per-robot mode
  const float* pending;  // float[B][n]
  float* latched;        // float[B][n]
  float* pid_block;      // first row of the Pid block this command's mode owns (position or velocity)
  size_t tstride;
  uint32_t batch, n;
  int new_mode;
  int rows;              // rows of one Pid block (GenLayout::rows)
};

__global__ __launch_bounds__(256) void cdpr_latch_masked_kernel(const LatchArgs a) {
  const uint32_t t = blockIdx.x * 256u + threadIdx.x;
  if (t >= a.batch * a.n) return;
  const uint32_t r = t / a.n;
  if (a.mask && !a.mask[r]) return;
  a.latched[t] = a.pending[t];
  if (a.pid_block && (int)a.mode[r] != a.new_mode) {  // (setForce resets no Pid: pid_block is null)
    for (int row = 0; row < a.rows; ++row) a.pid_block[(size_t)row * a.tstride + t] = 0.f;
  }
}

// The same arrival on the REGISTER-RESIDENT path of a per-robot handle (every robot keeps one Pid record, the active
// mode's: cdpr_step_kernel.hpp, StepArgs::meta).  One thread per robot: the Joy's row becomes the robot's active target,
// and entering the mode from the other one resets its Pid (JFC.cpp:101-103,113-115): integral 0 and call count 0, so the
// next Pid::update is the "first" and returns 0 (Pid.cpp:123-126).  The derivative ring needs no clearing: derive()
// returns 0 until nbuf new samples are in, and by then they have overwritten every slot the weights reach.
struct LatchFastArgs {
  const uint8_t* mask;   // uint8[B], or nullptr = every robot
  uint8_t* meta;         // StepArgs::meta
  const float* pending;  // float[B][n]
  float* target;         // float[B][n]: the robots' active target rows (StepArgs::cmd of the PR kernels)
  float4* hot;           // first integral row of the state: slot P + 5 * pairs
  size_t stride;
  uint32_t batch, n, hot_rows;
  uint32_t new_mode;     // kMetaPosition / kMetaVelocity
};

__global__ __launch_bounds__(256) void cdpr_latch_fast_kernel(const LatchFastArgs a) {
  const uint32_t r = blockIdx.x * 256u + threadIdx.x;
  if (r >= a.batch) return;
  if (a.mask && !a.mask[r]) return;
  for (uint32_t i = 0; i < a.n; ++i) a.target[(size_t)r * a.n + i] = a.pending[(size_t)r * a.n + i];
  uint32_t m = a.meta[r];
  if (a.new_mode == kMetaForce) {  // setForce (JFC.h:92-95) resets nothing; leaving Force mode resets the Pid entered
    m = (m & ~kMetaModeMask) | kMetaForce;
  } else if ((m & kMetaModeMask) != a.new_mode) {
    for (uint32_t g = 0; g < a.hot_rows; ++g) a.hot[(size_t)g * a.stride + r] = make_float4(0.f, 0.f, 0.f, 0.f);
    m = a.new_mode;  // call count 0
  }
  a.meta[r] = (uint8_t)m;
}

__global__ __launch_bounds__(256) void cdpr_set_mode_masked_kernel(const uint8_t* mask, uint8_t* mode, uint32_t batch, int new_mode) {
  const uint32_t r = blockIdx.x * 256u + threadIdx.x;
  if (r >= batch) return;
  if (!mask || mask[r]) mode[r] = (uint8_t)new_mode;
}

}  // namespace cdpr
