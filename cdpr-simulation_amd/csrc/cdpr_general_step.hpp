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
// (One-step launches of FK + TD handles up to two workgroups per CU run gen_controller in the controller wave of a
//  role-split workgroup instead: cdpr_general_split.hpp.)
//
// Records (HBM, one buffer, one column per robot), in two regions:
//   A: float4 SLOT rows (16 B per robot and row: a wave moves 1 KiB per row, by LDS-DMA on the way in)
//        slot g (g < ceil(n/4))                mLastPosition of cables 4g .. 4g+3 (JFC.h:45)
//        blockA(pid, i) = lp + (pid n + i)(nv + 1), pid 0 = position Pid, 1 = velocity Pid, nv = ceil(nb / 4):
//          + s (s < nv)     mDbufferY ring slots 4s .. 4s+3 (the error samples)
//          + nv             "H": meta | mLastTime (world-step index) | mIerr | mCmd
//                           meta: bit 0 mWasLastTime | samples in the window (6 bits) | ring head (6) | `run` (6) =
//                           consecutive one-step gaps ending at the newest sample, saturating
//                         The ring slot of a sample is its stamp mod nbuf (head = mLastTime mod nbuf): the first call of a
//                         Pid after a gap turns its ring, values and stamps, so that this holds again.
//   B: dword rows (4 B per robot and row), only touched where needed
//        blockB(pid, i) = (pid n + i)(nb + 8 ncas): mDbufferX ring slot j as a world-step index (int32), then the P-input
//        cascade's x1 x2 y1 y2 per stage, then the D-input cascade's
//   A Pid::reset of the whole batch is two memsets (the Pid's slots, the Pid's rows); all-zero IS the reset state.
//   (mDerr is not stored: Pid.cpp:154-157 reads the old value only when dt <= 0, and a Pid is updated at most once per
//    world step with strictly increasing stamps.)
//
// Per step and cable only the ACTIVE Pid is touched (position Pid in Position mode and in the hold branch, velocity Pid
// otherwise): nv + 1 slots read (64 B at the shipped 11-sample window), the slot of the new sample and H written (32 B)
// plus one stamp (4 B).  The stamps of the window are only READ when they are needed:
//   * a window whose nb samples were taken at consecutive world steps (meta.run >= nb - 1) is the uniform grid of the
//     fast path: the derivative is the closed-form FIR, weights looked up by ring head in LDS (0 at the head slot: the
//     newest sample enters from its register, weight in the Pid table).  When that holds for every cable of every robot
//     of a wave - all steps but the dozen after a mode change or a Pid switch - the wave takes a branch with ONE scalar
//     ring head, one weight row and static LDS addresses (gen_controller, "steady state");
//   * anything else (the nb - 1 steps after a switch between the two Pids in the hold branch) is a least-squares fit on
//     the real stamps.  These are rare and scattered over lanes and cables, so they are COMPACTED: every lane queues its
//     (cable, Pid) items in LDS, then the wave works the queue with one item per lane - orthogonal polynomials on the
//     sample stamps (Forsythe recurrence, fp64): no normal equations, well conditioned for any gap pattern.
// The slots travel global -> LDS by LDS-DMA (16 B per lane and instruction; a first version with dword rows spent 6 us per
// launch in 128 dword-wide DMA instructions per wave) as soon as the commands (which select the Pid) are known, and stay
// in flight under the IK and the Newton stage (stage order as in cdpr_onestep_kernel: IK -> early observables -> Newton
// -> controller -> tension distribution -> world step).
//
// Reference paths: Pid.cpp, JFC.cpp = JointForceCalculator.cpp, Filter.h (relative to src/cdpr_gazebo/).
#pragma once
#include <cstring>

#include "cdpr_step_kernel.hpp"

namespace cdpr {

constexpr int kGenMaxBuf = 32;   // CDPR_MAX_D_BUFFER
constexpr int kGenMaxDeg = 4;    // CDPR_MAX_D_DEGREE
constexpr int kGenMaxCas = 4;    // CDPR_MAX_CASCADE
constexpr int kGenQueueWords = 4 + 4 * 64;  // a wave's fit-queue counter, then four rows of 64 words (gen_controller, tier 1)

struct GenLayout {
  int n, nb, ncas;  // cables, longest window of the two Pids, deepest cascade
  __host__ __device__ int nv() const { return (nb + 3) / 4; }                    // value slots of one Pid of one cable
  __host__ __device__ int lp() const { return (n + 3) / 4; }                     // hold-position slots
  __host__ __device__ int spb() const { return nv() + 1; }                       // slots per (Pid, cable)
  __host__ __device__ int block_a(int pid, int cable) const { return lp() + (pid * n + cable) * spb(); }
  __host__ __device__ int pid_slots() const { return n * spb(); }                // one Pid of every cable: contiguous
  __host__ __device__ int slots() const { return lp() + 2 * pid_slots(); }       // region A
  __host__ __device__ int rpb() const { return nb + 8 * ncas; }                  // dword rows per (Pid, cable)
  __host__ __device__ int block_b(int pid, int cable) const { return (pid * n + cable) * rpb(); }
  __host__ __device__ int pid_rows() const { return n * rpb(); }
  __host__ __device__ int rows() const { return 2 * pid_rows(); }                // region B
  __host__ __device__ int r_pfilt() const { return nb; }
  __host__ __device__ int r_dfilt() const { return nb + 4 * ncas; }
  // region C, "hot rows" (dword rows, round 5): row 0 a world step + 1 (0: none), row 1 a Pid mask, rows 2 .. 2 + n - 1 mIerr
  // of the cables' Pids of that mask - what a robot in the deep steady state rewrites instead of its H slots (gen_hot_*)
  __host__ __device__ int hot_rows() const { return 2 + n; }
  __host__ __device__ size_t bytes(size_t rstride) const { return ((size_t)slots() * 16 + (size_t)(rows() + hot_rows()) * 4) * rstride; }
};

constexpr uint32_t kGmWasLast = 1u, kGmCountShift = 1u, kGmHeadShift = 7u, kGmRunShift = 13u, kGmField = 63u;
// Two-run windows (round 5; handles whose configuration admits the consecutive-call branches: mCmd is not state there and the
// H slot's fourth word is free): bit 25 says that the samples behind the newest run - ages run + 1 ... - were taken on
// consecutive steps ending at the world step in H.w ("last2"), kGmRun2Shift holds that run's length - 1.  A window with ONE
// gap in it (what a switch between the two Pids of a hold-branch cable leaves) then needs no stamp in memory at all.
constexpr uint32_t kGmRun2Shift = 19u, kGmTwo = 1u << 25;

struct GenPid {
  float kf, kp, ki, kd, imax, imin, cmax, cmin;
  int nbuf, degree, pcas, dcas, clamp;
  float pa0, pa1, pa2, pb1, pb2;  // BiQuad::SetFc(relCutoff, 1.0, quality), Filter.h:130-140
  float da0, da1, da2, db1, db2;
};

struct GenCtl {
  float* rec;            // the record buffer: region A (slots), then region B (rows)
  uint32_t rstride;      // columns (robots, or trajectories of a rollout, rounded up to 64)
  uint32_t rec_bytes;    // GenLayout::bytes(rstride) (< 4 GiB: the buffer is addressed with 32-bit offsets)
  const float* vel_cmd;  // latched jointVelocities float[B][n], or nullptr (target 0)
  const float* pos_cmd;  // latched jointPositions, or nullptr
  const float* frc_cmd;  // latched force command (setForce), or nullptr
  const uint8_t* mode_arr;  // per-robot mode (0 Force, 1 Position, 2 Velocity), or nullptr: `mode` for every robot
  const float* wtab;     // FIR weights by ring head: [pid][head][slot], rows of gen_nbp(NBMAX) floats
  int mode;
  int now_step;          // world step of the launch's first step
  float eps, dt;
  GenLayout lay;
  const float* ptab;     // the two Pids' parameters, [pid][kGenPidFloats] (gen_pid_table): copied to LDS, read per lane by
                         // the selected Pid - as kernel arguments the 2 x 23 scalars do not fit the scalar register file
                         // next to the step's own constants (measured: 2 400 v_readlane / v_writelane per robot-step)
  int pcas_max, dcas_max;  // deepest P-input / D-input cascade of the two Pids
  int nbuf0, nbuf1;        // window lengths of the position / velocity Pid (the ring slot of a sample is its stamp mod nbuf)
  int hot;                 // hot rows in use (the engine: simple_ok handles; CDPR_GEN_HOT=0 turns them off)
  int simple_ok;           // the steady-state branch may be taken: both Pids have the same window length and degree (one ring
                           // head and one weight row serve every cable), no cascades, a command clamp and iMin <= iMax
  // ROLLOUT: every trajectory works on a private copy of its robot's records (column = trajectory index in `rec`)
  const float* src_rec;
  uint32_t src_rstride;
};

// GenPid as the kernel reads it from LDS: six float4 per Pid
//   [kf kp ki kd] [imax imin cmax cmin] [nbuf pcas dcas (int bit patterns) 1/ki] [pa0 pa1 pa2 pb1] [pb2 da0 da1 da2] [db1 db2 degree w_new]
//   (w_new: the FIR weight of the newest sample, set by the engine next to the weight table)
constexpr int kGenPidFloats = 24;
inline void gen_pid_table(const GenPid& p, float* t) {
  auto bits = [](int v) { float f; memcpy(&f, &v, sizeof f); return f; };
  const float row[kGenPidFloats] = {p.kf, p.kp, p.ki, p.kd, p.imax, p.imin, p.cmax, p.cmin, bits(p.nbuf), bits(p.pcas), bits(p.dcas), p.ki != 0.f ? 1.f / p.ki : 0.f,
                                    p.pa0, p.pa1, p.pa2, p.pb1, p.pb2, p.da0, p.da1, p.da2, p.db1, p.db2, bits(p.degree), 0.f};
  for (int i = 0; i < kGenPidFloats; ++i) t[i] = row[i];
}

__host__ __device__ constexpr int gen_nbp(int nbmax) { return nbmax <= 11 ? 12 : 32; }  // padded weight-row length (float4 reads)
__host__ __device__ constexpr int gen_nv(int nbmax) { return (nbmax + 3) / 4; }

// The record buffer is addressed as ONE buffer resource (4 scalar registers): slot or row = a 32-bit scalar offset, the
// lane's column (and, where the row is the lane's own, its ring slot) = a 32-bit vector offset.  A cable's rows then cost
// one address register and one scalar per row; as 64-bit pointers the compiler hoists every row address of a launch out
// of the step loop and spills them (measured: 1.4 KiB of scratch per lane).  Predicated stores need no branch: the
// buffer's range check (vector offset >= num_records) drops the store of a lane whose offset is all ones.
struct GenBuf {
  __amdgpu_buffer_rsrc_t rsrc;
  uint32_t rs16;   // bytes per slot row (region A)
  uint32_t rs4;    // bytes per dword row (region B)
  uint32_t base_b; // byte offset of region B
  uint32_t base_c; // byte offset of region C (hot rows)
  // region A: float4 slots; voff = the lane's column * 16 (+ the selected Pid's offset)
  CDPR_DEV float4 load4(int slot, uint32_t voff) const {
    typedef float f4 __attribute__((ext_vector_type(4)));
    const f4 v = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, (uint32_t)slot * rs16, 0));
    return make_float4(v.x, v.y, v.z, v.w);
  }
  // (The row goes into the VECTOR offset, soffset stays the immediate 0: gfx950 needs a wait state between a 128-bit
  //  buffer store whose soffset is an SGPR and a VALU write of its data registers - lanes 12-15 of every 16 otherwise
  //  store the new value now and then (scripts/micro/store_hazard.hip, profiles/r05_store_hazard.txt) - and LLVM's hazard
  //  recognizer inserts none in that case (the ISA manual exempts it); with an immediate soffset it inserts the two wait
  //  states the part needs.  scripts/exec_lint.py fails the build's test if the pattern appears anywhere.)
  CDPR_DEV void store4_if(bool on, int slot, uint32_t voff, const float4& v) const {
    const u32x4 d = {__builtin_bit_cast(unsigned, v.x), __builtin_bit_cast(unsigned, v.y), __builtin_bit_cast(unsigned, v.z), __builtin_bit_cast(unsigned, v.w)};
    __builtin_amdgcn_raw_buffer_store_b128(d, rsrc, on ? voff + (uint32_t)slot * rs16 : 0xFFFFFFFFu, 0, 0);
  }
  // global -> LDS without a register destination: lane l's 16 bytes land at dst_row + 16 l
  CDPR_DEV void slot_to_lds(int slot, uint32_t voff, float4* dst_row) const {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)dst_row, 16, voff, (uint32_t)slot * rs16, 0, 0);
  }
  // region B: dword rows; voff = the lane's column * 4 (+ the selected Pid's offset, + the lane's own ring slot)
  CDPR_DEV float load(int row, uint32_t voff) const { return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, voff, base_b + (uint32_t)row * rs4, 0)); }
  CDPR_DEV int loadi(int row, uint32_t voff) const { return (int)__builtin_amdgcn_raw_buffer_load_b32(rsrc, voff, base_b + (uint32_t)row * rs4, 0); }
  CDPR_DEV void store_if(bool on, int row, uint32_t voff, float v) const {
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rsrc, on ? voff : 0xFFFFFFFFu, base_b + (uint32_t)row * rs4, 0);
  }
  CDPR_DEV void storei_if(bool on, int row, uint32_t voff, int v) const {
    __builtin_amdgcn_raw_buffer_store_b32((unsigned)v, rsrc, on ? voff : 0xFFFFFFFFu, base_b + (uint32_t)row * rs4, 0);
  }
  // region C: hot rows; voff = the lane's column * 4
  CDPR_DEV void rowc_to_lds(int row, uint32_t voff, void* dst_row) const {  // lane l's dword lands at dst_row + 4 l
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)dst_row, 4, voff, base_c + (uint32_t)row * rs4, 0, 0);
  }
  CDPR_DEV uint32_t loadc(int row, uint32_t voff) const { return __builtin_amdgcn_raw_buffer_load_b32(rsrc, voff, base_c + (uint32_t)row * rs4, 0); }
  CDPR_DEV void storec_if(bool on, int row, uint32_t voff, uint32_t v) const {
    __builtin_amdgcn_raw_buffer_store_b32(v, rsrc, on ? voff : 0xFFFFFFFFu, base_c + (uint32_t)row * rs4, 0);
  }
};
CDPR_DEV GenBuf gen_buffer(float* rec, uint32_t rstride, uint32_t rec_bytes, const GenLayout& L) {
  GenBuf b;
  b.rsrc = __builtin_amdgcn_make_buffer_rsrc(rec, 0, (int)rec_bytes, 0x00020000);
  b.rs16 = rstride * 16u;
  b.rs4 = rstride * 4u;
  b.base_b = (uint32_t)L.slots() * b.rs16;
  b.base_c = b.base_b + (uint32_t)L.rows() * b.rs4;
  return b;
}

// Derivative at the newest stamp of the least-squares polynomial of degree `degree` through the nb samples (t_j, y_j):
// what Pid::derive + fitPolynomial compute (Pid.cpp:193-247), posed on orthogonal polynomials over the sample stamps
// (Forsythe's three-term recurrence) instead of normal equations: p_{k+1}(x) = (x - alpha_k) p_k(x) - beta_k p_{k-1}(x),
// alpha_k = sum x p_k^2 / sum p_k^2, beta_k = sum p_k^2 / sum p_{k-1}^2, fit = sum c_k p_k with c_k = sum y p_k / sum p_k^2.
// Abscissae x_j = (t_j - t_new) / h, h = mean spacing: integers before the division, so exact for any stamps.
// Returns d/dt in 1 / world steps (the caller divides by the step length).  Slots j >= nb are ignored.
// STREAMING form (round 5): the polynomials' values on the sample points are not kept in arrays (2 x NBMAX doubles: 44
// registers at the shipped window) but re-evaluated from the recurrence coefficients found so far - one pass over the
// samples per degree, which yields sum p^2, sum y p and the next sum x p^2 together; one division per degree (1 / sum p^2
// serves c_k, then alpha and beta of the next degree).  ~40 registers instead of ~110: the fit runs inside the kernels'
// inlined controller branch; ~260 fp64 instructions at 11 samples, degree 2 (the array form: ~500, three divisions per degree).
template <int NBMAX, typename Y = float>
CDPR_DEV double gen_fit(const Y (&y)[NBMAX], const int (&t)[NBMAX], int nb, int degree, int t_new, int t_old) {
  double h = (double)(t_new - t_old) / (double)(nb - 1);
  if (!(h > 0.0)) h = 1.0;
  const double inv_h = 1.0 / h;
  int dx[NBMAX];  // stamps relative to the newest one (exact)
#pragma unroll
  for (int j = 0; j < NBMAX; ++j) dx[j] = t[j] - t_new;
  double sx = 0.0;  // sum x p_0^2
#pragma unroll
  for (int j = 0; j < NBMAX; ++j) sx += (j < nb) ? (double)dx[j] * inv_h : 0.0;
  double r = 1.0 / (double)nb;  // 1 / sum p_k^2
  double alpha[kGenMaxDeg], beta[kGenMaxDeg];
  double bk = 0.0;            // beta_k
  double vp = 0.0, vc = 1.0;  // p_{k-1}(0), p_k(0): the newest stamp is x = 0
  double dp = 0.0, dc = 0.0;  // their derivatives at 0
  double deriv = 0.0;
#pragma unroll
  for (int k = 0; k < kGenMaxDeg; ++k) {
    if (k < degree) {
      alpha[k] = sx * r;
      beta[k] = bk;
      double sn = 0.0, bn = 0.0, sxn = 0.0;
#pragma unroll
      for (int j = 0; j < NBMAX; ++j) {
        const double x = (double)dx[j] * inv_h;
        double qm = 0.0, qc = 1.0;  // p_{i-1}(x), p_i(x)
#pragma unroll
        for (int i = 0; i <= k; ++i) {
          const double qn = fma(x - alpha[i], qc, -(beta[i] * qm));
          qm = qc;
          qc = qn;
        }
        const double pn = (j < nb) ? qc : 0.0;
        sn = fma(pn, pn, sn);
        bn = fma((double)y[j], pn, bn);
        sxn = fma(x * pn, pn, sxn);
      }
      const double vn = fma(-alpha[k], vc, -(bk * vp));
      const double dn = vc + fma(-alpha[k], dc, -(bk * dp));
      vp = vc;
      vc = vn;
      dp = dc;
      dc = dn;
      const double rn = 1.0 / sn;
      deriv = fma(bn * rn, dn, deriv);
      bk = sn * r;
      r = rn;
      sx = sxn;
    }
  }
  return deriv * inv_h;
}

// gen_fit as a CALL, for the kernel that must fit two waves per SIMD (cdpr_gen_lean_kernel inlines tier 1 for waves without
// a gap call): inlined, the fit's registers come on top of everything the controller wave keeps for its epilogue and the
// allocator spills on the first branch's path (20 registers, `force` among them: stores in the Pid loop, loads and a wait
// in front of the hand-off).  As a function with 26 scalar arguments the pressure stays where the call is.
__device__ __attribute__((noinline)) float gen_fit11_call(float y0, float y1, float y2, float y3, float y4, float y5, float y6, float y7, float y8, float y9, float y10,
                                                          int t0, int t1, int t2, int t3, int t4, int t5, int t6, int t7, int t8, int t9, int t10, int nb, int degree,
                                                          int t_new, int t_old, float dt) {
  const float y[11] = {y0, y1, y2, y3, y4, y5, y6, y7, y8, y9, y10};
  const int t[11] = {t0, t1, t2, t3, t4, t5, t6, t7, t8, t9, t10};
  return (float)(gen_fit<11>(y, t, nb, degree, t_new, t_old) / (double)dt);
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
    B.store_if(on && store, rw + 1, voff, x1);
    B.store_if(on && store, rw, voff, out);
    B.store_if(on && store, rw + 3, voff, y1);
    B.store_if(on && store, rw + 2, voff, y0);
    out = on ? y0 : out;
  }
  return out;
}

// The Pid a lane runs for one cable this step: position or velocity Pid, read from the LDS table by the lane's selection;
// everything by value - a reference into the kernel arguments makes the compiler keep a private copy of them in scratch.
struct GenSel {
  float kf, kp, ki, kd, imax, imin, cmax, cmin, inv_ki, w_new;
  int nbuf, pcas, dcas;
  float pa0, pa1, pa2, pb1, pb2, da0, da1, da2, db1, db2;
};
CDPR_DEV GenSel gen_select(bool vel, const float4 (*ptab)[kGenPidFloats / 4], bool filters) {
  const float4* t = ptab[vel ? 1 : 0];
  const float4 r0 = t[0], r1 = t[1], r2 = t[2];
  GenSel c;
  c.kf = r0.x, c.kp = r0.y, c.ki = r0.z, c.kd = r0.w;
  c.imax = r1.x, c.imin = r1.y, c.cmax = r1.z, c.cmin = r1.w;
  c.nbuf = __float_as_int(r2.x), c.pcas = __float_as_int(r2.y), c.dcas = __float_as_int(r2.z), c.inv_ki = r2.w;
  c.w_new = t[5].w;
  c.pa0 = c.pa1 = c.pa2 = c.pb1 = c.pb2 = c.da0 = c.da1 = c.da2 = c.db1 = c.db2 = 0.f;
  if (filters) {  // (wave-uniform: some Pid has a cascade)
    const float4 r3 = t[3], r4 = t[4], r5 = t[5];
    c.pa0 = r3.x, c.pa1 = r3.y, c.pa2 = r3.z, c.pb1 = r3.w;
    c.pb2 = r4.x, c.da0 = r4.y, c.da1 = r4.z, c.da2 = r4.w;
    c.db1 = r5.x, c.db2 = r5.y;
  }
  return c;
}

// Pid.cpp:154-186 from the derivative on: D term (through the D-input cascade), command, clamp, anti-windup, and the H slot
// (meta | mLastTime | mIerr | mCmd) back to the records; returns mCmd.  `on`: this lane really finishes this cable's
// Pid::update now (the stores follow it).
CDPR_DEV float gen_finish(const GenBuf& RB, const GenSel c, int slot_h, int row_dfilt, int dcas_max, bool on, bool filt_on, uint32_t voff_a, uint32_t voff_b, float derived,
                          float err, float dt, float pre, float ie, float prev, float old, uint32_t nmeta, int now, float& d_term) {
  float derr = derived;
  if (dcas_max) derr = gen_cascade(RB, row_dfilt, voff_b, dcas_max, filt_on ? c.dcas : 0, filt_on, c.da0, c.da1, c.da2, c.db1, c.db2, derr);
  d_term = c.kd * derr;
  const float cmd = pre + d_term;
  float out = (c.cmax > c.cmin) ? fmaxf(fminf(cmd, c.cmax), c.cmin) : old;  // Pid.cpp:175-177: without a clamp mCmd keeps its value
  const bool wind = out != cmd;                                              // Pid.cpp:181-184
  ie = wind ? prev : ie;
  out = wind ? fmaf(dt * err, c.ki, out) : out;
  RB.store4_if(on, slot_h, voff_a, make_float4(__uint_as_float(nmeta), __int_as_float(now), ie, out));
  return out;
}

// The controller of one wave's 64 robots for one world step (PLG.cpp:222-228 -> JFC.cpp:59-96 -> Pid.cpp:122-191): per-cable
// force into force[], the `pid` topic's terms of cable 0 into dbg.  Shared by the one-wave stepping kernel and the
// controller wave of the role-split kernel.  LDS working set: `cab` = per cable NV + 1 DMA-staged slots of 64 float4 (the
// values, then H); seen as rows of 64 floats, a cable's rows 0-6 double as the parking place of a cable that waits for the
// fit and rows 7-9 as the fit queue (item, new sample, result) once the values are done with (rows 10-11: dump words of the
// lanes a predicated LDS store does not concern); `hold` = the staged
// hold-position slots.
// Only q, qd and the force live across the cables: a cable whose derivative is known at once (uniform window: the FIR;
// window not full: 0) runs its whole Pid::update in the first loop; one that needs the fit parks seven numbers and
// finishes after the pass.  Written WITHOUT divergent control flow around anything heavy (per-lane cases are selects and
// predicated stores): the register allocator splits long live ranges around high-pressure regions, and a split made under
// a partial exec mask does not carry the lanes that were masked off (seen: NaN in values that only crossed the fit pass).
struct GenCtlConst {
  int pcas_max, dcas_max;
  float dt, inv_dt;
  int nm0, nm1;  // this step's ring slot in the position / velocity Pid's window: now mod nbuf
  int nbuf0;     // the position Pid's window length
  bool simple_ok;  // the handle's configuration admits the steady-state branch (GenCtl::simple_ok)
  float* force_rows;  // nullptr, or LDS rows of 64 floats (cable i: row i) that take the per-cable forces of the consecutive-call
                      // branches instead of the caller's array (the lean role-split kernel: eight registers less across the controller)
#ifdef CDPR_STAMPS
  unsigned long long* stamps;  // this wave's eight stamps (diagnostic build)
#endif
};
#if defined(CDPR_STAMPS) && defined(CDPR_STAMPS_CTL)  // stamps 1..3 inside the controller instead of the phases before it
#define GEN_CTL_STAMP(i)                                                                      \
  do {                                                                                        \
    __builtin_amdgcn_sched_barrier(0);                                                        \
    if (kc.stamps && lane == 0) kc.stamps[i] = __builtin_amdgcn_s_memrealtime();              \
    __builtin_amdgcn_sched_barrier(0);                                                        \
  } while (0)
#define GEN_PHASE_STAMP(i) do { } while (0)
#elif defined(CDPR_STAMPS) && defined(CDPR_STAMPS_PRO)  // stamps 2, 3 inside the prologue (loads back | Pid selection done)
#define GEN_CTL_STAMP(i) do { } while (0)
#define GEN_PHASE_STAMP(i) do { if ((i) == 1) CDPR_STAMP(1); } while (0)
#else
#define GEN_CTL_STAMP(i) do { } while (0)
#define GEN_PHASE_STAMP(i) CDPR_STAMP(i)
#endif
// diagnostic build with -DCDPR_STAMPS_COLD (scripts/stamp_probe_cold.py): the rare paths' own timeline and queue length, in a
// second block of eight words per workgroup behind the phase stamps
#if defined(CDPR_STAMPS) && defined(CDPR_STAMPS_COLD)
#define GEN_COLD_STAMP(i, v)                                                                  \
  do {                                                                                        \
    __builtin_amdgcn_sched_barrier(0);                                                        \
    if (kc.stamps && lane == 0) kc.stamps[i] = (v);                                           \
    __builtin_amdgcn_sched_barrier(0);                                                        \
  } while (0)
#else
#define GEN_COLD_STAMP(i, v) do { } while (0)
#endif
struct GenDbg {
  float p, i, d, des;
  bool pi, dw;
};
// Hot rows (region C; round 5).  A robot whose cables all sit in the deep steady state - window full, `run` saturated, no
// second run - changes only mLastTime and mIerr of its H slots from step to step.  Such a robot stops writing (and reading)
// the eight H slots: it keeps one word "step s + 1 | Pid mask" and the integrals in dword rows (40 B read and 40 B written
// per step against 256).  While the word is valid the H slots of the mask's Pids are STALE in memory; every kernel restores
// them in LDS behind the DMA (gen_hot_restore), and a robot that leaves the state writes them back that very step.
struct GenHot {
  bool on;        // the handle uses hot rows
  bool has;       // this robot's word is valid: the H slots of the Pids of `mask` are stale in memory
  bool fresh;     // ... and those Pids were called one world step ago and are this step's Pids
  uint32_t mask;  // bit i: the Pid (0 position, 1 velocity) that served cable i at that step
  int step;       // that step
};
template <int N>
CDPR_DEV GenHot gen_hot_state(bool on, uint32_t step1, uint32_t mask, const int (&sel)[N], int mode, int now) {
  GenHot h;
  h.on = on, h.has = on && step1 != 0u, h.step = (int)step1 - 1, h.mask = mask;
  uint32_t sm = 0u;
#pragma unroll
  for (int i = 0; i < N; ++i) sm |= (uint32_t)sel[i] << i;
  h.fresh = h.has && h.step == now - 1 && mask == sm && mode != 0;
  return h;
}
// In front of the DMA of a step's record slots: the robot's word -> its hot state; the integrals on their way (ordinary
// loads: first read behind the wait for the DMA); `skip_h`: every lane of the wave fresh - no H slot is loaded.
// (step1, mask: the robot's word - rows 0 and 1 - loaded by the caller with its first loads: read here, they would be a round
//  trip to memory between the Joy that selects the Pids and the DMA of their slots: + 0.9 us per step at 16 384 robots.  The
//  integrals: where every lane of the wave is fresh, their dword rows ride the DMA into the H slots' staging rows, which
//  stay empty then (gen_stage_records); kept in registers from here to gen_hot_restore they were spilled, with a wait for
//  them in front of the spill)
template <int N>
CDPR_DEV GenHot gen_hot_begin(bool on, uint32_t step1, uint32_t mask, const int (&sel)[N], int mode, int now, bool& skip_h) {
  const GenHot hot = gen_hot_state<N>(on, step1, mask, sel, mode, now);
  skip_h = on && __builtin_amdgcn_ballot_w64(!hot.fresh) == 0ull;
  return hot;
}
// Behind the DMA of a step's record slots: the staged H slots of the robots with a valid word, restored from it (every
// lane fresh: the H slots were not loaded at all, gen_stage_records); a robot that is NOT fresh leaves the state - its
// eight H slots go back to memory as they stand and its word is cleared.
template <int N, int NBMAX>
CDPR_DEV void gen_hot_restore(const GenCtlConst kc, const GenBuf& RB, const GenLayout L, uint32_t lane, bool live, uint32_t col, const GenHot hot, const int (&sel)[N],
                              float4* cab, const float* ierr_rows = nullptr) {  // ierr_rows: the integrals came with the DMA (gen_stage_records)
  constexpr int NV = gen_nv(NBMAX);
  constexpr int kCab = (NV + 1) * 64;
  if (__builtin_amdgcn_ballot_w64(hot.has) == 0ull) return;  // (wave-uniform)
  const int nbuf = kc.nbuf0;
  const int hd = hot.step - (hot.step / nbuf) * nbuf;  // the ring slot of a sample is its stamp mod nbuf
  const uint32_t meta = kGmWasLast | ((uint32_t)nbuf << kGmCountShift) | ((uint32_t)hd << kGmHeadShift) | (kGmField << kGmRunShift);
  const uint32_t pid_a = (uint32_t)L.pid_slots() * RB.rs16;
  const bool all_fresh = __builtin_amdgcn_ballot_w64(!hot.fresh) == 0ull;
  float ierr[N];
  if (all_fresh) {  // (wave-uniform) the integrals came with the DMA: a dword row at the head of every cable's (empty) H staging row
#pragma unroll
    for (int i = 0; i < N; ++i) ierr[i] = reinterpret_cast<const float*>(cab + i * kCab + NV * 64)[lane];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");  // (every lane has read its words before a lane's H slot is written over them)
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int i = 0; i < N; ++i) cab[i * kCab + NV * 64 + lane] = make_float4(__uint_as_float(meta), __int_as_float(hot.step), ierr[i], 0.f);
    return;
  }
  // a wave with robots that are not fresh (rare): the integrals from memory, the staged H slots patched where they are this Pid's;
  // a robot that is not fresh leaves the state: H slots back to memory, word cleared
#pragma unroll
  for (int i = 0; i < N; ++i) ierr[i] = ierr_rows ? ierr_rows[i * 64 + lane] : __uint_as_float(RB.loadc(2 + i, col * 4u));
  const bool leaves = hot.has && !hot.fresh;
#pragma unroll
  for (int i = 0; i < N; ++i) {
    const float4 canon = make_float4(__uint_as_float(meta), __int_as_float(hot.step), ierr[i], 0.f);
    const bool mine = hot.has && ((hot.mask >> i) & 1u) == (uint32_t)sel[i];  // the staged H slot is this Pid's
    float4* const hs = cab + i * kCab + NV * 64 + lane;
    const float4 rd = *hs;  // (component by component: a select of two float4 objects goes through scratch memory)
    *hs = make_float4(mine ? canon.x : rd.x, mine ? canon.y : rd.y, mine ? canon.z : rd.z, mine ? canon.w : rd.w);
    RB.store4_if(live && leaves, L.block_a(0, i) + L.nv(), col * 16u + (((hot.mask >> i) & 1u) ? pid_a : 0u), canon);
  }
  RB.storec_if(live && leaves, 0, col * 4u, 0u);
}

// Pid::update of every cable of a wave whose calls are all CONSECUTIVE (each Pid called one world step ago, mWasLastTime
// set, no cascades, command clamp, iMin <= iMax, one window length for both Pids): ONE ring head for the whole wave (now
// mod nbuf, a scalar), one row of weights, static LDS addresses; the table weighs the head slot with 0 and the new sample
// enters the FIR from its register, so nothing is written to the staged window before it is read.  No stamp is stored:
// the stamps of a run of consecutive steps are implied by mLastTime and `run` (lazy stamps, see the general loop).
//   PASS 0  no cable waits for the fit (need == 0): everything is done here
//   PASS 1  the cables of `need` (a per-lane mask) wait for the fit: their new error goes to the queue's error row
//           (qrows + 64, at the item's index: qslot0 + its rank among the lane's items) and six more numbers to the rows of
//           `park`; the item's fit lane finishes the Pid::update (H, force) behind the fit (gen_controller, tier 1)
// GAPS (passes 1 and 2): some lane's Pid was NOT called one world step ago (the first call after a switch between the two Pids
// of a hold-branch cable): its integrator step is now - mLastTime, its run of consecutive calls starts again at 0 - and its
// ring has been turned already (gen_turn_rings) so that the wave's one ring head is this lane's head as well.
// Arithmetic and its order are gen_finish's and the general loop's: the same bits on the same inputs.
template <int N, int NBMAX, int PASS, int GWIDE = 4, bool GAPS = false>
CDPR_DEV void gen_consecutive(const GenCtlConst kc, const GenBuf& RB, const GenLayout L, uint32_t lane, bool live, uint32_t col, int mode, int now,
                              const float (&target)[N], const int (&sel)[N], const v2f (&q)[cable_pairs(N)], const v2f (&qd)[cable_pairs(N)],
                              const float4* cab, const float4 (&held4)[(N + 3) / 4], const float* wrot, const float4 (*ptab)[kGenPidFloats / 4],
                              uint32_t need, uint32_t qslot0, float* qrows, float (&force)[N], float (&newpos)[N], GenDbg& dbg, float* park = nullptr,
                              bool sat = false) {
  constexpr int NV = gen_nv(NBMAX);
  constexpr int NBP = gen_nbp(NBMAX);
  constexpr int kCab = (NV + 1) * 64;
  static_assert(PASS != 2, "the second pass is the fit lanes' (gen_controller, tier 1)");
  const uint32_t pid_a = (uint32_t)L.pid_slots() * RB.rs16;
  const int nhead = kc.nm0, nbuf = kc.nbuf0;
  const int q4 = nhead >> 2, qc = nhead & 3;
  const uint32_t meta_lo = kGmWasLast | ((uint32_t)nhead << kGmHeadShift);
  float4 w[NV];
  if constexpr (PASS != 2) {
    const float4* wr = reinterpret_cast<const float4*>(wrot + nhead * NBP);
#pragma unroll
    for (int s4 = 0; s4 < NV; ++s4) w[s4] = wr[s4];
  }
  // cables per group: 4 NV + 20 registers per cable; groups of 2 or 8 and no scheduling barrier between the groups were
  // measured too (+-0.05 us: the phase is bound by the instruction count, ~2.5 ns per instruction of any kind)
  constexpr int GW = (PASS == 2) ? 1 : ((NV > 3) ? 2 : GWIDE);
  constexpr int G = (N < GW) ? N : GW;
#pragma unroll
  for (int b = 0; b < N; b += G) {
    __builtin_amdgcn_sched_barrier(0);
    if (PASS == 0 && b == G) GEN_CTL_STAMP(2);
    if constexpr (PASS == 2) {  // (wave-uniform) no lane waits for this cable's fit
      if (__builtin_amdgcn_ballot_w64(((need >> b) & 1u) != 0u) == 0ull) continue;
    }
    float4 g0[G], g1[G], g2[G];  // kf kp ki kd | imax imin cmax cmin | nbuf . . 1/ki
    float wn[G];
    float4 v[G][NV], vs[G], hh[G];
#pragma unroll
    for (int j = 0; j < G; ++j) {
      const int i = b + j;
      if (i < N) {
        hh[j] = cab[i * kCab + NV * 64 + lane];
        const float4* pt = ptab[sel[i] ? 1 : 0];
        g0[j] = pt[0], g1[j] = pt[1], g2[j] = pt[2], wn[j] = pt[5].w;
        if constexpr (PASS != 2) {
          const float4* cs = cab + i * kCab + lane;
#pragma unroll
          for (int s4 = 0; s4 < NV; ++s4) v[j][s4] = cs[s4 * 64];
          vs[j] = cs[q4 * 64];  // the slot that takes the new sample
        }
      }
    }
#pragma unroll
    for (int j = 0; j < G; ++j) {
      const int i = b + j;
      if (i < N) {
        const float qi = (i & 1) ? q[i / 2].y : q[i / 2].x;
        const float qdi = (i & 1) ? qd[i / 2].y : qd[i / 2].x;
        const bool sv = sel[i] != 0;
        const bool hold = (mode == 2) && !sv;
        const float held = comp4(held4[i / 4], i % 4);
        const float desired = hold ? held : target[i];  // JFC.cpp:81
        if constexpr (PASS != 2) newpos[i] = hold ? held : qi;  // JFC.cpp:75,87
        const float error = desired - ((mode == 2 && sv) ? qdi : qi);
        const uint32_t va = col * 16u + (sv ? pid_a : 0u);
        const int sa = L.block_a(0, i);
        const uint32_t meta = __float_as_uint(hh[j].x);
        const int count = (int)((meta >> kGmCountShift) & kGmField), run = (int)((meta >> kGmRunShift) & kGmField);
        const int ncount = min(count + 1, nbuf);
        const int since = GAPS ? now - __float_as_int(hh[j].y) : 1;  // world steps since this Pid's last call
        const float dt = GAPS ? (float)since * kc.dt : kc.dt;          // (one step: the same bits)
        const int nrun = (count > 0 && since == 1) ? min(run + 1, (int)kGmField) : 0;
        const bool first = GAPS && !(meta & kGmWasLast);  // Pid.cpp:123-126: the first call since reset returns 0 and takes no sample
        // the run behind the newest one (two-run windows, see kGmTwo): kept while one of its samples is still in the window; a
        // call after a gap starts a new newest run and makes the old one - the whole window: gen_consecutive_test - the second
        const bool gapped = GAPS && since != 1 && count > 0;
        const uint32_t run2 = gapped ? (uint32_t)run : ((meta >> kGmRun2Shift) & kGmField);
        const int last2 = gapped ? __float_as_int(hh[j].y) : __float_as_int(hh[j].w);
        const bool two = (gapped || (meta & kGmTwo) != 0u) && ncount > nrun + 1;
        const uint32_t nmeta = first ? (meta | kGmWasLast)
                                     : (meta_lo | ((uint32_t)ncount << kGmCountShift) | ((uint32_t)nrun << kGmRunShift) | (two ? (kGmTwo | (run2 << kGmRun2Shift)) : 0u));
        const bool waits = (PASS != 0) && ((need >> i) & 1u) != 0u;  // this lane's cable i goes through the fit queue
        const uint32_t qslot = (PASS != 0) ? qslot0 + (uint32_t)__builtin_popcount(need & ((1u << i) - 1u)) : 0u;
        const float prev_ierr = hh[j].z;
        const float p_term = g0[j].y * error;
        float ie = fmaf(dt, error, prev_ierr);
        const float i_term = g0[j].z * ie;
        if (i == 0 && PASS != 2) dbg.p = first ? dbg.p : p_term, dbg.i = first ? dbg.i : i_term, dbg.des = first ? dbg.des : desired, dbg.pi = !first;
        const float i_cl = __builtin_amdgcn_fmed3f(i_term, g1[j].y, g1[j].x);  // Pid.cpp:143-152 (iMin <= iMax here)
        ie = (i_cl != i_term) ? i_cl * g2[j].w : ie;
        float derived;
        if constexpr (PASS != 2) {
          float acc = 0.f;
#pragma unroll
          for (int s4 = 0; s4 < NV; ++s4)
#pragma unroll
            for (int k = 0; k < 4; ++k)
              if (4 * s4 + k < NBMAX) acc = fmaf(comp4(w[s4], k), comp4(v[j][s4], k), acc);
          acc = fmaf(wn[j], error, acc);
          derived = (ncount >= nbuf) ? acc * kc.inv_dt : 0.f;  // mDbufferMissing != 0: derive() returns 0 (Pid.cpp:200-203)
        } else {
          derived = qrows[128 + (waits ? qslot : 0u)];
        }
        const float d_term = g0[j].w * derived;
        if (i == 0) {
          if constexpr (PASS == 0) dbg.d = d_term, dbg.dw = true;
          if constexpr (PASS == 1) dbg.d = (waits || first) ? dbg.d : d_term, dbg.dw = !waits && !first;
          if constexpr (PASS == 2) dbg.d = waits ? d_term : dbg.d, dbg.dw = dbg.dw || waits;
        }
        const float cmd = ((g0[j].x * desired + p_term) + i_cl) + d_term;  // Pid.cpp:170
        float out = __builtin_amdgcn_fmed3f(cmd, g1[j].w, g1[j].z);          // Pid.cpp:175-177 (cmdMin < cmdMax here)
        const bool wind = out != cmd;                                         // Pid.cpp:181-184
        ie = wind ? prev_ierr : ie;
        out = wind ? fmaf(dt * error, g0[j].z, out) : out;
        if constexpr (GAPS) {  // the first call: H = (meta | 1, now, mIerr as it was, the clamp of 0), no force (as the general loop leaves it)
          ie = first ? prev_ierr : ie;
          out = first ? __builtin_amdgcn_fmed3f(0.f, g1[j].w, g1[j].z) : out;
        }
        if (kc.force_rows) {  // (known where the kernel builds kc: one of the two forms is compiled)
          kc.force_rows[i * 64 + lane] = (PASS == 0) ? out : ((waits || first) ? 0.f : out);
        } else {
          if constexpr (PASS == 0) force[i] = out;
          if constexpr (PASS == 1) force[i] = (waits || first) ? 0.f : out;
        }
        if constexpr (PASS != 2) {
          float4 o = vs[j];
          o.x = (qc == 0) ? error : o.x, o.y = (qc == 1) ? error : o.y, o.z = (qc == 2) ? error : o.z, o.w = (qc == 3) ? error : o.w;
          RB.store4_if(live && !first, sa + q4, va, o);
        }
        if constexpr (PASS == 1) {
          // what the item's fit lane needs to finish this Pid::update (no branch: a lane without an item writes its dump word)
          qrows[waits ? 64u + qslot : 192u + lane] = error;
          float* const pk = waits ? park + qslot : qrows + 192u + lane;
          const uint32_t ps = waits ? 64u : 0u;
          pk[0 * ps] = dt, pk[1 * ps] = (g0[j].x * desired + p_term) + i_cl, pk[2 * ps] = (i_cl != i_term) ? i_cl * g2[j].w : fmaf(dt, error, prev_ierr);
          pk[3 * ps] = prev_ierr, pk[4 * ps] = __uint_as_float(nmeta), pk[5 * ps] = __int_as_float(last2);
        }
        // (PASS 0, `sat`: the robot keeps its steady state in the hot rows - mIerr there, no H slot written: GenHot)
        if constexpr (PASS == 0) RB.storec_if(live && sat, 2 + i, col * 4u, __float_as_uint(ie));
        const bool writes_h = (PASS == 0) ? (live && !sat) : ((PASS == 1) ? (live && !waits) : (live && waits));
        RB.store4_if(writes_h, sa + L.nv(), va, make_float4(__uint_as_float(nmeta), __int_as_float(now), ie, (two && !first) ? __int_as_float(last2) : out));
      }
    }
  }
  __builtin_amdgcn_sched_barrier(0);
}

// TIER1 = false: the second branch (tier 1) is left out - inlined into the one-wave kernels it doubles the time of their first
// branch (12.7 -> 27.1 us at 16 384 x 8, steady: scripts/one_wave_ab.py), so those go from the first branch to the general loop.
// STEADY_ONLY: only the first branch is compiled (the lean role-split kernel inlines this much and leaves the rest to
// gen_lean_cold_tail); returns false - having stored nothing - when the wave does not qualify for it.
// Which branch of gen_controller serves this lane's cables (see there):
//   `simple`  every cable calls a Pid that was called one world step ago (mWasLastTime set, not Force mode) and none of them
//             needs the fit (a uniform or a filling window): the first branch;
//   `fast`    not Force mode, and where a cable's Pid was not called one step ago the stamps of its window are all implied
//             (run + 1 >= count: no earlier gap in the window): the second branch (tier 1) serves it;
//   `gaps`    some cable of this lane was not called one step ago, or never since its reset.
template <int N, int NBMAX>
CDPR_DEV void gen_consecutive_test(const GenCtlConst kc, const float4* cab, uint32_t lane, int mode, int now, bool& simple, bool& fast, bool& gaps) {
  constexpr int NV = gen_nv(NBMAX);
  constexpr int kCab = (NV + 1) * 64;
  simple = false, fast = false, gaps = false;
  if (kc.simple_ok) {  // (scalar: no cascades, one window for both Pids, both with a command clamp and iMin <= iMax)
    // all-integer, no per-cable lane masks: sign bits collect "the window is full after this push and not a uniform grid" and
    // "a gap now and an earlier one in the window", any bit collects "not called one step ago", an AND collects mWasLastTime
    const int nbuf = kc.nbuf0;
    int neg = 0, old_gap = 0;
    uint32_t nz = 0u, was = 1u;
#pragma unroll
    for (int i = 0; i < N; ++i) {
      const float2 ml = *reinterpret_cast<const float2*>(&cab[i * kCab + NV * 64 + lane]);
      const uint32_t meta = __float_as_uint(ml.x);
      const int count = (int)((meta >> kGmCountShift) & kGmField), run = (int)((meta >> kGmRunShift) & kGmField);
      const int away = now - 1 - __float_as_int(ml.y);  // 0: called one step ago
      neg |= ~(count + 1 - nbuf) & (run + 2 - nbuf);
      old_gap |= (away != 0) ? ~(count - run - 2) : 0;  // sign bit: count >= run + 2 (a sample older than the run) at a gap
      nz |= (uint32_t)away;
      was &= meta;
    }
    const bool called = (was & 1u) != 0u;
    gaps = nz != 0u || !called;  // (a Pid's first call since reset is served by the variant for gaps)
    simple = called && mode != 0 && nz == 0u && (neg >= 0);
    fast = mode != 0 && (old_gap >= 0);
  }
}

// The rings of the lanes whose Pid was not called one world step ago, turned so that the ring slot of every sample is again
// its stamp mod nbuf relative to THIS step (the previous sample right before the slot that takes the new one): values through
// LDS (each lane gathers its own window, turned, and puts it back: a fit lane reads it there) and back to the records.  No
// stamp is written: the window's stamps - implied until now by mLastTime and `run` - stay implied as its second run (kGmTwo).
// Per cable, skipped where no lane of the wave has a gap (wave-uniform).  Pid.cpp:193-217 keeps insertion order; the
// layout here is an implementation choice that buys the one scalar ring head of the branches above.
template <int N, int NBMAX>
CDPR_DEV void gen_turn_rings(const GenCtlConst kc, const GenBuf& RB, const GenLayout L, uint32_t lane, bool live, uint32_t col, int now, const int (&sel)[N], float4* cab) {
  constexpr int NV = gen_nv(NBMAX);
  constexpr int kCab = (NV + 1) * 64;
  constexpr int kCabF = kCab * 4;
  float* const cabf = reinterpret_cast<float*>(cab);
  const uint32_t pid_a = (uint32_t)L.pid_slots() * RB.rs16;
  const int nhead = kc.nm0, nbuf = kc.nbuf0;
#pragma unroll
  for (int i = 0; i < N; ++i) {
    const float2 ml = *reinterpret_cast<const float2*>(&cab[i * kCab + NV * 64 + lane]);
    const uint32_t meta = __float_as_uint(ml.x);
    const int last = __float_as_int(ml.y);
    const int count = (int)((meta >> kGmCountShift) & kGmField), head = (int)((meta >> kGmHeadShift) & kGmField);
    const bool gap = count > 0 && (now - last) != 1;
    int shift = nhead - 1 - head;
    shift += (shift < 0) ? nbuf : 0;
    const bool rot = gap && shift != 0;
    if (__builtin_amdgcn_ballot_w64(rot) == 0ull) continue;  // (wave-uniform)
    shift = rot ? shift : 0;
    const bool sv = sel[i] != 0;
    const uint32_t va = col * 16u + (sv ? pid_a : 0u);
    const int sa = L.block_a(0, i);
    float* const win = cabf + i * kCabF + lane * 4;  // the lane's float4 inside a slot row; slot rows 256 floats apart
    float tv[NV * 4];
#pragma unroll
    for (int j = 0; j < NV * 4; ++j) {
      int src = j - shift;
      src += (src < 0) ? nbuf : 0;
      src = (j < nbuf) ? src : j;
      tv[j] = win[(src >> 2) * 256 + (src & 3)];
    }
#pragma unroll
    for (int s4 = 0; s4 < NV; ++s4) {
      const float4 v = make_float4(tv[4 * s4], tv[4 * s4 + 1], tv[4 * s4 + 2], tv[4 * s4 + 3]);
      cab[i * kCab + s4 * 64 + lane] = v;
      RB.store4_if(live && rot && s4 < L.nv(), sa + min(s4, L.nv() - 1), va, v);
    }
  }
}

// gen_controller's first branch: every cable of the wave on a uniform or a filling window (gen_consecutive<0>), then
// mLastPosition back.
template <int N, int NBMAX, int GWIDE, bool HOT = false>
CDPR_DEV void gen_steady(const GenCtlConst kc, const GenBuf& RB, const GenLayout L, uint32_t lane, bool live, uint32_t col, int mode, int now, const float (&target)[N],
                         const int (&sel)[N], const v2f (&q)[cable_pairs(N)], const v2f (&qd)[cable_pairs(N)], const float4* cab, const float4* hold_slots,
                         const float* wrot, const float4 (*ptab)[kGenPidFloats / 4], float (&force)[N], GenDbg& dbg, const GenHot hot = GenHot{false, false, false, 0u, 0}) {
  constexpr int LP = (N + 3) / 4;
  constexpr int NV = gen_nv(NBMAX);
  constexpr int kCab = (NV + 1) * 64;
  float4 held4[LP];
#pragma unroll
  for (int g4 = 0; g4 < LP; ++g4) held4[g4] = hold_slots[g4 * 64 + lane];
  float newpos[N];
  // HOT: a robot whose cables are all in the deep steady state AFTER this step (window full, run saturated, one run) keeps
  // mLastTime and mIerr in its hot rows from now on (or goes on doing so) instead of writing eight H slots
  bool sat = false;
  if (HOT && hot.on) {
    int bad = 0;
#pragma unroll
    for (int i = 0; i < N; ++i) {
      const uint32_t meta = __float_as_uint(reinterpret_cast<const float*>(cab)[(i * kCab + NV * 64 + lane) * 4]);
      const int count = (int)((meta >> kGmCountShift) & kGmField), run = (int)((meta >> kGmRunShift) & kGmField);
      // (round 6: a run of nbuf steps is enough - every sample of the full window then lies in the newest run, which is all any later decision
      //  asks of `run`; the restored word says `saturated` (kGmField).  Through round 5 the run had to BE saturated, 62 steps: with cables
      //  switching Pids 7 % of the robots were without a word at any time and nearly every wave held one - 21.5 -> 21.2 us)
#ifndef CDPR_HOT_RUN_MIN
#define CDPR_HOT_RUN_MIN kc.nbuf0
#endif
      bad |= (count - kc.nbuf0) | (run - (CDPR_HOT_RUN_MIN)) | ((meta & kGmTwo) ? -1 : 0);
    }
    sat = bad >= 0;
  }
  GEN_CTL_STAMP(1);
  gen_consecutive<N, NBMAX, 0, GWIDE>(kc, RB, L, lane, live, col, mode, now, target, sel, q, qd, cab, held4, wrot, ptab, 0u, 0u, nullptr, force, newpos, dbg, nullptr, sat);
  GEN_CTL_STAMP(3);
  if (hot.on) {  // (wave-uniform) the word: valid from this step on, or cleared where a robot with one does not qualify (H slots written above)
    uint32_t sm = 0u;
#pragma unroll
    for (int i = 0; i < N; ++i) sm |= (uint32_t)sel[i] << i;
    RB.storec_if(live && (sat || hot.has), 0, col * 4u, sat ? (uint32_t)(now + 1) : 0u);
    RB.storec_if(live && sat && !(hot.fresh && hot.mask == sm), 1, col * 4u, sm);
  }
#pragma unroll
  for (int g4 = 0; g4 < LP; ++g4)
    RB.store4_if(live, g4, col * 16u, make_float4(newpos[4 * g4], (4 * g4 + 1 < N) ? newpos[4 * g4 + 1] : 0.f, (4 * g4 + 2 < N) ? newpos[4 * g4 + 2] : 0.f,
                                                  (4 * g4 + 3 < N) ? newpos[4 * g4 + 3] : 0.f));
}

#ifndef CDPR_LEAN_GROUP
#define CDPR_LEAN_GROUP 2  // cables per group in the branch the lean role-split kernel inlines (register pressure: 32 registers per cable of a group)
#endif
template <int N, int NBMAX, bool STEADY_ONLY = false, bool TIER1 = true, bool HOT = false>
CDPR_DEV bool gen_controller(const GenCtlConst kc, const GenBuf& RB, const GenLayout L, uint32_t lane, bool live, uint32_t col, uint32_t first_unit,
                             uint32_t units, int mode, int now, const float (&target)[N], const int (&sel)[N], const v2f (&q)[cable_pairs(N)],
                             const v2f (&qd)[cable_pairs(N)], float4* cab, const float4* hold_slots, const float* wrot,
                             const float4 (*ptab)[kGenPidFloats / 4], uint32_t* q_count, float (&force)[N], GenDbg& dbg,
                             const GenHot hot = GenHot{false, false, false, 0u, 0}) {
  constexpr int NV = gen_nv(NBMAX);
  constexpr int NBP = gen_nbp(NBMAX);
  constexpr int kCab = (NV + 1) * 64;    // float4 elements between the slot sets of consecutive cables
  constexpr int kCabF = kCab * 4;        // the same in floats
  constexpr int LP = (N + 3) / 4;
  static_assert((NV + 1) * 4 >= 12, "parking, queue and dump rows");
  float* const cabf = reinterpret_cast<float*>(cab);
  const int pcas_max = kc.pcas_max, dcas_max = kc.dcas_max;
  const bool filters = (pcas_max | dcas_max) != 0;
  const uint32_t pid_a = (uint32_t)L.pid_slots() * RB.rs16, pid_b = (uint32_t)L.pid_rows() * RB.rs4;
  auto qitem = [&](uint32_t idx) -> uint32_t* { return reinterpret_cast<uint32_t*>(cabf + (idx >> 6) * kCabF + 7 * 64 + (idx & 63u)); };
  auto qerr = [&](uint32_t idx) -> float* { return cabf + (idx >> 6) * kCabF + 8 * 64 + (idx & 63u); };

  // ---- consecutive calls, decided for the whole wave: every cable of every robot calls a Pid that was called one world step
  //      ago, no cascades, a command clamp - what a handle does on all steps but the one of a mode change or of a switch
  //      between the two Pids of a hold-branch cable.  Then Pid::update is the fast path's arithmetic (Pid.cpp:128-186) plus the
  //      ring bookkeeping, without any of the per-lane case selection of the general loop below: ~100 instructions of all
  //      kinds per cable instead of ~300.  The derivative of a cable is then one of three things:
  //        S  a full window on a uniform grid: the closed-form FIR;
  //        F  a window that is still filling (count + 1 < nbuf): 0, as derive() returns it (Pid.cpp:200-203);
  //        Q  a full window with a gap in it (the nbuf - 1 steps after a switch between the two Pids): the least-squares fit
  //           on the real stamps, through the wave's fit queue;
  //      and a Pid that was NOT called one step ago (G: the step of the switch itself) has its ring turned first.
  //      Waves with S and F cables only take the first branch below (gen_consecutive<0>); waves with some Q or G cable the
  //      second one (tier 1).  STEADY_ONLY (the lean role-split kernel) compiles the first branch and nothing else.
  // (a single wave per SIMD hides nothing: every dependent LDS round trip costs ~50 ns, so the reads of a phase are issued
  //  for a group of cables together and the group then pays the latency once)
  bool simple, fast, gaps;
  gen_consecutive_test<N, NBMAX>(kc, cab, lane, mode, now, simple, fast, gaps);
#ifdef CDPR_EXPECT_STEADY  // build variant (scripts/build_variants.sh): the branch-layout hint; results must not depend on it
  if (__builtin_expect(__builtin_amdgcn_ballot_w64(!simple) == 0ull, 1)) {
#else
  if (__builtin_amdgcn_ballot_w64(!simple) == 0ull) {  // (wave-uniform)
#endif
    gen_steady<N, NBMAX, STEADY_ONLY ? CDPR_LEAN_GROUP : 4, HOT>(kc, RB, L, lane, live, col, mode, now, target, sel, q, qd, cab, hold_slots, wrot, ptab, force, dbg, hot);
    return true;
  }
  // (the other tiers write every H slot they touch: a robot that comes here with a valid word - restored in LDS by
  //  gen_hot_restore - has it cleared)
  if (hot.on) RB.storec_if(live && hot.has, 0, col * 4u, 0u);
  // ---- tier 1: every Pid called with mWasLastTime set; some windows carry a gap (the nbuf - 1 steps after a switch between
  //      the two Pids: the fit), some Pids are called for the first time after one (their rings are turned first,
  //      gen_turn_rings).  The queue is built FIRST, from the staged H slots alone, so that the stamps a fit needs - the only
  //      thing it reads from memory: the window values are the owner's staged slots in LDS - are in flight under the Pid
  //      arithmetic of the whole wave (gen_consecutive<1>); the waiting cables are finished after the fit by a second pass
  //      over them alone (gen_consecutive<2>).  One pass of the queue (<= 64 items per wave); a wave with more, or with a
  //      lane that does not qualify, takes the general loop below (the lean role-split kernel: in its cold tail).
  //      `q_count` is followed by four rows of 64 words: items, new errors, results, dump words.
#ifndef CDPR_LEAN_TIER1
#define CDPR_LEAN_TIER1 1  // the lean kernel inlines tier 1 for waves WITHOUT a gap call (fits behind consecutive calls); 0: the first branch only
#endif
  if constexpr (STEADY_ONLY && !CDPR_LEAN_TIER1) return false;  // (inlined with all of tier 1 the lean kernel spilled 332 registers, on the first branch's path too)
  if constexpr (TIER1)
  if (kc.simple_ok && __builtin_amdgcn_ballot_w64(!fast) == 0ull) {  // (wave-uniform)
    GEN_COLD_STAMP(0, __builtin_amdgcn_s_memrealtime());
    uint32_t* const qitems = q_count + 4;
    float* const qrows = reinterpret_cast<float*>(q_count + 4);
    const int nbuf = kc.nbuf0, nhead = kc.nm0;
    const bool any_gap = __builtin_amdgcn_ballot_w64(gaps) != 0ull;
#ifndef CDPR_LEAN_GAPS
#define CDPR_LEAN_GAPS 0  // 1: the gap calls inline as well - measured and not kept: 44 spilled registers (in that branch), the steady
                          // step 16.0 -> 16.8 us at 65 536 x 8, the switching workload the same 22.8 us.  2: inline unless a ring has
                          // to turn (no spill) - the same 23.0 us: the few waves that turn rings AND fit decide the refresh step
#endif
    if constexpr (STEADY_ONLY && CDPR_LEAN_GAPS == 0) {  // (the gap calls - one step in ten of a switching workload - go to the tail with their ring turns)
      if (any_gap) return false;
    }
    if constexpr (STEADY_ONLY && CDPR_LEAN_GAPS == 2) {  // gap calls inline unless some lane's ring has to turn (few waves of a refresh step)
      if (any_gap) {
        bool turns = false;
#pragma unroll
        for (int i = 0; i < N; ++i) {
          const float2 ml = *reinterpret_cast<const float2*>(&cab[i * kCab + NV * 64 + lane]);
          const uint32_t meta = __float_as_uint(ml.x);
          const int count = (int)((meta >> kGmCountShift) & kGmField), head = (int)((meta >> kGmHeadShift) & kGmField);
          int shift = nhead - 1 - head;
          shift += (shift < 0) ? nbuf : 0;
          turns = turns || (count > 0 && (now - __float_as_int(ml.y)) != 1 && shift != 0);
        }
        if (__builtin_amdgcn_ballot_w64(turns) != 0ull) return false;
      }
    }
    uint32_t need = 0u;
    uint32_t word[N];  // the items' upper bits: gap flag (bit 15) and the run of consecutive calls ending at the new sample
#pragma unroll
    for (int i = 0; i < N; ++i) {  // (every H read before the first queue word is written: the compiler must take them for aliases)
      const float2 ml = *reinterpret_cast<const float2*>(&cab[i * kCab + NV * 64 + lane]);
      const uint32_t meta = __float_as_uint(ml.x);
      const int count = (int)((meta >> kGmCountShift) & kGmField), run = (int)((meta >> kGmRunShift) & kGmField);
      const bool cons = (now - __float_as_int(ml.y)) == 1;
      need |= ((count + 1 >= nbuf) && ((cons ? run + 2 : 1) < nbuf)) ? (1u << i) : 0u;  // full after this push, run + 1 (or 0) < nbuf - 1
      // bit 15: the first call after a gap (the run behind the new sample ends at the owner's mLastTime); bit 14: the stamps behind
      // the newest run are in memory (a window the general loop wrote out), not implied by a second run (kGmTwo)
      word[i] = cons ? (((uint32_t)min(run + 1, (int)kGmField) << 16) | ((meta & kGmTwo) ? 0u : (1u << 14))) : (1u << 15);
    }
    // the items' places in the queue, lane by lane: an exclusive prefix sum of popcount(need) over the wave from four ballots
    // (no LDS atomic, no barrier: `total` is a scalar at once; `q_count` is the general loop's alone)
    uint32_t total = 0u, slot0 = 0u;
    {
      const uint32_t cnt = (uint32_t)__builtin_popcount(need);
      static_assert(N <= 15, "four bits of items per lane");
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const uint64_t m = __builtin_amdgcn_ballot_w64(((cnt >> b) & 1u) != 0u);
        slot0 += __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u)) << b;
        total += (uint32_t)__builtin_popcountll(m) << b;
      }
    }
    if (total != 0u && total <= 64u) {  // (wave-uniform) some cable waits for the fit
#pragma unroll
      for (int i = 0; i < N; ++i) {
        const uint32_t sl = slot0 + (uint32_t)__builtin_popcount(need & ((1u << i) - 1u));
        const bool has = ((need >> i) & 1u) != 0u;
        qitems[has ? sl : 192u + lane] = lane | ((uint32_t)i << 6) | ((uint32_t)sel[i] << 9) | word[i];
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    GEN_COLD_STAMP(4, (unsigned long long)total | ((any_gap ? 3ull : 2ull) << 32));
    if (total <= 64u) {  // (wave-uniform)
      float4 held4[LP];
#pragma unroll
      for (int g4 = 0; g4 < LP; ++g4) held4[g4] = hold_slots[g4 * 64 + lane];
      float newpos[N];
      // the hold-position slots are in registers now: their LDS rows park what an item's fit lane needs to finish the owner's
      // Pid::update (six rows of 64 by item), the force it returns (row 6) and the `pid` topic's D term (row 7)
      static_assert(LP >= 2, "eight rows of 64 floats in the hold-position slots");
      float* const park = const_cast<float*>(reinterpret_cast<const float*>(hold_slots));
      const bool mine = lane < total;
      int t[NBMAX];
      uint32_t ol = 0u, ci = 0u, sp = 0u;
      int irun = 0, t_second = 0;
      bool in_memory = false, owner_live = false;
      uint32_t ocol = 0u;
      // (the item and - for a window that was written out - its stamps: fetched in front of the first pass, so that the loads
      //  fly under it; STEADY_ONLY, the kernel at 256 registers: behind it - a dozen registers less across the pass)
      auto fetch_item = [&]() {
        const uint32_t it = qitems[mine ? lane : 0u];
        ol = it & 63u, ci = (it >> 6) & 7u, sp = (it >> 9) & 1u;
        in_memory = ((it >> 14) & 1u) != 0u;
        irun = (int)((it >> 16) & 63u);
        // where the run behind the newest one ends: the owner's mLastTime before this step (first call after a gap) or its H.w
        const float4 oh = cab[ci * kCab + NV * 64 + ol];
        t_second = ((it >> 15) & 1u) ? __float_as_int(oh.y) : __float_as_int(oh.w);
#pragma unroll
        for (int j = 0; j < NBMAX; ++j) t[j] = 0;
        const uint32_t ro = first_unit + ol;
        owner_live = ro < units;
        ocol = owner_live ? ro : (units - 1u);
        if (__builtin_amdgcn_ballot_w64(mine && in_memory) != 0ull) {  // (wave-uniform) a window whose stamps were written out: on their way
          const uint32_t ob = ocol * 4u + (uint32_t)L.block_b(0, (int)ci) * RB.rs4 + (sp ? pid_b : 0u);
#pragma unroll
          for (int j = 0; j < NBMAX; ++j) t[j] = RB.loadi(min(j, L.nb - 1), ob);
        }
      };
      if (!STEADY_ONLY && total != 0u) fetch_item();  // (wave-uniform)
      GEN_COLD_STAMP(1, __builtin_amdgcn_s_memrealtime());
      if ((!STEADY_ONLY || CDPR_LEAN_GAPS) && any_gap) {  // (wave-uniform)
        if constexpr (!(STEADY_ONLY && CDPR_LEAN_GAPS == 2)) gen_turn_rings<N, NBMAX>(kc, RB, L, lane, live, col, now, sel, cab);
        gen_consecutive<N, NBMAX, 1, STEADY_ONLY ? CDPR_LEAN_GROUP : 4, true>(kc, RB, L, lane, live, col, mode, now, target, sel, q, qd, cab, held4, wrot, ptab, need, slot0, qrows, force, newpos,
                                                                              dbg, park);
      } else {
        gen_consecutive<N, NBMAX, 1, STEADY_ONLY ? CDPR_LEAN_GROUP : 4, false>(kc, RB, L, lane, live, col, mode, now, target, sel, q, qd, cab, held4, wrot, ptab, need, slot0, qrows, force, newpos,
                                                                               dbg, park);
      }
      if (total != 0u) {  // (wave-uniform)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (STEADY_ONLY) fetch_item();
        GEN_COLD_STAMP(2, __builtin_amdgcn_s_memrealtime());
        const float e_new = qrows[64u + (mine ? lane : 0u)];
        float y[NBMAX];
#pragma unroll
        for (int s4 = 0; s4 < NV; ++s4) {
          const float4 v = cab[ci * kCab + s4 * 64 + ol];  // the owner's staged window (turned, where it had to be)
#pragma unroll
          for (int k = 0; k < 4; ++k)
            if (4 * s4 + k < NBMAX) y[4 * s4 + k] = comp4(v, k);
        }
        const uint32_t hd = (uint32_t)nhead;
        const uint32_t old = (hd + 1u == (uint32_t)nbuf) ? 0u : hd + 1u;  // the oldest sample sits right after the head
        const int degree = __float_as_int(ptab[sp][5].z);
        int t_old = now;
#pragma unroll
        for (int j = 0; j < NBMAX; ++j) {
          y[j] = ((uint32_t)j == hd) ? e_new : ((j < nbuf) ? y[j] : 0.f);
          int age = (int)hd - j;
          age += (age < 0) ? nbuf : 0;
          // the new sample and the run behind it: implied by `now`; the samples behind that run: the window's second run, implied
          // by the step it ends at - or, for a window the general loop wrote out, from memory
          t[j] = (j >= nbuf) ? now : ((age <= irun) ? now - age : (in_memory ? t[j] : t_second - (age - irun - 1)));
          t_old = ((uint32_t)j == old) ? t[j] : t_old;
        }
        float res;
        if constexpr (STEADY_ONLY && NBMAX == 11)
          res = gen_fit11_call(y[0], y[1], y[2], y[3], y[4], y[5], y[6], y[7], y[8], y[9], y[10], t[0], t[1], t[2], t[3], t[4], t[5], t[6], t[7], t[8], t[9], t[10], nbuf, degree,
                               now, t_old, kc.dt);
        else
          res = (float)(gen_fit<NBMAX>(y, t, nbuf, degree, now, t_old) / (double)kc.dt);
        GEN_COLD_STAMP(7, __builtin_amdgcn_s_memrealtime());  // (lands in FRONT of the fit: the call is pure and sinks to its use - the item's window is assembled here)
        {  // the rest of the owner's Pid::update (Pid.cpp:154-186) by the item's lane: D term, command, clamp, anti-windup, H
          const uint32_t ix = mine ? lane : 0u;
          const float dt_i = park[ix], pre = park[64u + ix], prev = park[192u + ix];
          float ie = park[128u + ix];
          const uint32_t nmeta = __float_as_uint(park[256u + ix]);
          const float last2 = park[320u + ix];
          const float4 ga = ptab[sp][0], gb = ptab[sp][1];
          const float d_term = ga.w * res;
          const float cmd = pre + d_term;
          float out = __builtin_amdgcn_fmed3f(cmd, gb.w, gb.z);
          const bool wind = out != cmd;
          ie = wind ? prev : ie;
          out = wind ? fmaf(dt_i * e_new, ga.z, out) : out;
          const uint32_t oh = ocol * 16u + (sp ? pid_a : 0u) + (uint32_t)(L.block_a(0, (int)ci) + L.nv()) * RB.rs16;
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");  // (the parked rows are read: rows 6 and 7 may be written)
          __builtin_amdgcn_wave_barrier();
          RB.store4_if(mine && owner_live, 0, oh, make_float4(__uint_as_float(nmeta), __int_as_float(now), ie, (nmeta & kGmTwo) ? last2 : out));
          park[mine ? 384u + lane : 0u] = mine ? out : dt_i;  // (a lane without an item rewrites what item 0's row 0 holds)
          park[mine ? 448u + lane : 0u] = mine ? d_term : dt_i;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int i = 0; i < N; ++i) {  // the owners take their forces
          const bool waits = ((need >> i) & 1u) != 0u;
          const uint32_t qslot = slot0 + (uint32_t)__builtin_popcount(need & ((1u << i) - 1u));
          const float f = park[384u + (waits ? qslot : 0u)];
          if (kc.force_rows)
            *(waits ? kc.force_rows + i * 64 + lane : qrows + 192u + lane) = f;  // (no branch: a lane that does not wait writes its dump word)
          else
            force[i] = waits ? f : force[i];
          if (i == 0) {
            const float dd = park[448u + (waits ? qslot : 0u)];
            dbg.d = waits ? dd : dbg.d;
            dbg.dw = dbg.dw || waits;
          }
        }
      }
#pragma unroll
      for (int g4 = 0; g4 < LP; ++g4)
        RB.store4_if(live, g4, col * 16u, make_float4(newpos[4 * g4], (4 * g4 + 1 < N) ? newpos[4 * g4 + 1] : 0.f, (4 * g4 + 2 < N) ? newpos[4 * g4 + 2] : 0.f,
                                                      (4 * g4 + 3 < N) ? newpos[4 * g4 + 3] : 0.f));
      GEN_COLD_STAMP(3, __builtin_amdgcn_s_memrealtime());
      return true;
    }
  }
  if constexpr (STEADY_ONLY) return false;
  GEN_COLD_STAMP(0, __builtin_amdgcn_s_memrealtime());

  float4 held4[LP];
#pragma unroll
  for (int g4 = 0; g4 < LP; ++g4) held4[g4] = hold_slots[g4 * 64 + lane];
  float newpos[N];
  uint32_t need = 0u;  // cables whose derivative comes from the fit queue
  bool any_rot = false;  // some ring of this wave turned this step (wave-uniform)
  GEN_COLD_STAMP(4, 0ull);
  GEN_COLD_STAMP(2, 0ull);
#pragma unroll
  for (int i = 0; i < N; ++i) {
    __builtin_amdgcn_sched_barrier(0);  // one cable at a time: hoisting every cable's staged slots costs 128 registers (letting the
                                        // scheduler interleave 2 or 4 cables changes nothing: 15.5 / 23.7 us at 16 384 / 65 536 x 8 either way)
    const float qi = (i & 1) ? q[i / 2].y : q[i / 2].x;
    const float qdi = (i & 1) ? qd[i / 2].y : qd[i / 2].x;
    const bool is_force = (mode == 0);  // JFC.cpp:67-70
    const bool sv = sel[i] != 0;
    const bool hold = (mode == 2) && !sv;
    const uint32_t va = col * 16u + (sv ? pid_a : 0u), vb = col * 4u + (sv ? pid_b : 0u);  // the lane's offsets into the selected Pid
    const int sa = L.block_a(0, i), rb = L.block_b(0, i);
    float4* const cs = cab + i * kCab + lane;  // this cable's staged slots of this lane, 64 float4 apart
    float* const park = cabf + i * kCabF + lane;  // its float rows, 64 floats apart
    const float held = comp4(held4[i / 4], i % 4);
    const float desired = hold ? held : target[i];  // JFC.cpp:81: mLastPosition in the hold branch
    newpos[i] = hold ? held : qi;                    // JFC.cpp:68,75,87: mLastPosition = joint position
    const float actual = (mode == 2 && sv) ? qdi : qi;
    const GenSel c = gen_select(sv, ptab, filters);
    const int nbuf = c.nbuf;
    const float4 h = cs[NV * 64];
    const uint32_t meta = __float_as_uint(h.x);
    const int last = __float_as_int(h.y);
    const float prev_ierr = h.z, old_cmd = h.w;
    const bool first = !is_force && !(meta & kGmWasLast);  // Pid.cpp:123-126: the first call since reset returns 0
    const bool runs = !is_force && !first;
    const float error = desired - actual;
    const float dt = (float)(now - last) * kc.dt;
    // A two-run window (kGmTwo: written by the consecutive-call branches, maybe in another launch) is written out before this
    // loop works on it: the stamps of its second run go to their rows, and from here on the window is what this loop knows -
    // a newest run implied by mLastTime and `run`, everything older in memory (the H this call writes carries no kGmTwo).
    if (__builtin_amdgcn_ballot_w64(runs && (meta & kGmTwo) != 0u) != 0ull) {  // (wave-uniform)
      any_rot = true;  // (the fit reads other lanes' rows: the fence in front of it)
      const int count0 = (int)((meta >> kGmCountShift) & kGmField), head0 = (int)((meta >> kGmHeadShift) & kGmField);
      const int run0 = (int)((meta >> kGmRunShift) & kGmField), run2 = (int)((meta >> kGmRun2Shift) & kGmField);
      const int last2 = __float_as_int(h.w);
#pragma clang loop unroll(disable)  // (a rare path and no array indexed by j: kept as a loop, 12 instructions)
      for (int j = 0; j < L.nb; ++j) {
        int a = head0 - j;  // of ring slot j, in calls before the newest sample of the window as it stands
        a += (a < 0) ? c.nbuf : 0;
        const bool on = live && runs && (meta & kGmTwo) != 0u && j < c.nbuf && a > run0 && a < count0 && a - run0 - 1 <= run2;
        RB.storei_if(on, rb + j, vb, last2 - (a - run0 - 1));
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    float perr = error;
    if (pcas_max) perr = gen_cascade(RB, rb + L.r_pfilt(), vb, pcas_max, c.pcas, runs && live, c.pa0, c.pa1, c.pa2, c.pb1, c.pb2, error);
    const float p_term = c.kp * perr;
    float ie = fmaf(dt, error, prev_ierr);
    float i_term = c.ki * ie;
    if (i == 0) {  // `pid` topic (Pid.cpp:139-142,158-159): what the Pid call of cable 0 writes, when there is one
      dbg.p = runs ? p_term : dbg.p;
      dbg.i = runs ? i_term : dbg.i;
      dbg.des = runs ? desired : dbg.des;
      dbg.pi = runs;
    }
    const bool over = i_term > c.imax, under = i_term < c.imin;  // Pid.cpp:143-152
    i_term = over ? c.imax : (under ? c.imin : i_term);
    ie = (over || under) ? i_term * c.inv_ki : ie;  // (mIerr = iTerm / mIgain, as a multiplication like the fast path)
    const float pre = (c.kf * desired + p_term) + i_term;  // Pid.cpp:170: fTerm + pTerm + iTerm (+ dTerm in gen_finish)
    // Pid::derive (Pid.cpp:193-217): push the sample (dt > 0 always: a Pid is called at most once per world step)
    const int count = (int)((meta >> kGmCountShift) & kGmField), head = (int)((meta >> kGmHeadShift) & kGmField);
    const int run = (int)((meta >> kGmRunShift) & kGmField);
    // the ring slot of a sample is its stamp mod nbuf: a Pid called on consecutive steps moves its head by one, and windows
    // of consecutive steps have the same head in every lane (what the steady-state branch above lives on).  The first call
    // after a gap turns the ring, values and stamps, so that the previous sample sits right before the new one.
    const int nhead = sv ? kc.nm1 : kc.nm0;
    int shift = nhead - 1 - head;
    shift += (shift < 0) ? nbuf : 0;
    // Lazy stamps: the steady-state branch above stores no stamp (mDbufferX): the newest run + 1 samples were taken on
    // consecutive steps ending at mLastTime, so their stamps are mLastTime - age.  The first call after a GAP (now - last
    // != 1: the run ends) writes the window's stamps out - the implied ones computed, the older ones as stored - turned
    // with the ring where the ring turns; from then on `run` starts again at 0 and every stamp of the window is in memory.
    const bool gap = runs && count > 0 && (now - last) != 1;
    const bool rot = gap && shift != 0;
    shift = rot ? shift : 0;
    if (__builtin_amdgcn_ballot_w64(gap) != 0ull) {  // (wave-uniform; lanes that do not turn move every sample onto itself)
      any_rot = true;
      float tv[NBMAX];
      int ts[NBMAX];
#pragma unroll
      for (int j = 0; j < NBMAX; ++j) {
        int src = j - shift;
        src += (src < 0) ? nbuf : 0;
        src = (j < nbuf) ? src : j;
        tv[j] = park[(src >> 2) * 256 + lane * 3 + (src & 3)];
        int age = head - src;  // of the sample in ring slot `src`, in calls before the newest one
        age += (age < 0) ? nbuf : 0;
        const int stored = RB.loadi(rb, vb + (uint32_t)min(src, L.nb - 1) * RB.rs4);
        ts[j] = (age <= run && age < count) ? last - age : stored;
      }
#pragma unroll
      for (int j = 0; j < NBMAX; ++j) park[(j >> 2) * 256 + lane * 3 + (j & 3)] = tv[j];
#pragma unroll
      for (int j = 0; j < NBMAX; ++j) RB.storei_if(live && gap && j < nbuf && j != nhead, rb + min(j, L.nb - 1), vb, ts[j]);
#pragma unroll
      for (int s4 = 0; s4 < NV; ++s4) RB.store4_if(live && rot && s4 != (nhead >> 2) && s4 < L.nv(), sa + min(s4, L.nv() - 1), va, cs[s4 * 64]);
    }
    const int ncount = min(count + 1, nbuf);
    const int nrun = (count > 0 && now - last == 1) ? min(run + 1, (int)kGmField) : 0;
    const uint32_t nmeta = first ? (meta | kGmWasLast)
                                 : (kGmWasLast | ((uint32_t)ncount << kGmCountShift) | ((uint32_t)nhead << kGmHeadShift) | ((uint32_t)nrun << kGmRunShift));
    // the new sample into its ring slot of the staged window, that slot back to the records with its stamp
    park[(nhead >> 2) * 256 + lane * 3 + (nhead & 3)] = error;  // (park + lane * 3 = the lane's float4 inside a slot row)
    const float4 vs = cs[(nhead >> 2) * 64];
    RB.store4_if(live && runs, sa, va + (uint32_t)(nhead >> 2) * RB.rs16, vs);
    RB.storei_if(live && runs, rb, vb + (uint32_t)nhead * RB.rs4, now);
    // full window that is not a uniform grid (the nbuf - 1 steps after a switch between the two Pids): queued for the fit
    const bool queued = runs && ncount >= nbuf && nrun < nbuf - 1;
    // nbuf samples one world step apart: the closed-form end-point LS derivative, weights by ring head (zero for slots >= nbuf)
    const float* wr = wrot + ((sv ? NBMAX : 0) + nhead) * NBP;
    float acc = 0.f;
#pragma unroll
    for (int s4 = 0; s4 < NV; ++s4) {
      const float4 v = cs[s4 * 64];
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (4 * s4 + k < NBMAX) acc = fmaf(wr[4 * s4 + k], comp4(v, k), acc);
    }
    acc = fmaf(c.w_new, error, acc);  // (the table weighs the head slot with 0: the newest sample enters here)
    const float derived = (ncount >= nbuf) ? acc * kc.inv_dt : 0.f;  // mDbufferMissing != 0: derive() returns 0 (Pid.cpp:200-203)
    float d_term;
    const bool done = (runs && !queued) || first;  // first call: H = (meta | 1, now, mIerr as it was, mCmd = 0); out is not used
    const float out = gen_finish(RB, c, sa + L.nv(), rb + L.r_dfilt(), dcas_max, done && live, runs && !queued && live, va, vb, derived, first ? 0.f : error, dt,
                                 first ? 0.f : pre, first ? prev_ierr : ie, prev_ierr, first ? 0.f : old_cmd, nmeta, now, d_term);
    if (i == 0) {
      dbg.d = (runs && !queued) ? d_term : dbg.d;
      dbg.dw = runs && !queued;
    }
    force[i] = is_force ? target[i] : ((runs && !queued) ? out : 0.f);
    need |= queued ? (1u << i) : 0u;
    {  // park what the rest of Pid::update needs in the cable's own staged rows (the values there are done with).  No branch:
       // a lane that is not queued writes the same seven numbers to its dump word (row 10 of the cable's block, unused)
      float* const pk = queued ? park : park + 10 * 64;
      const int ps = queued ? 64 : 0;
      pk[0 * ps] = error, pk[1 * ps] = dt, pk[2 * ps] = pre, pk[3 * ps] = ie, pk[4 * ps] = prev_ierr, pk[5 * ps] = old_cmd;
      pk[6 * ps] = __uint_as_float(nmeta);
    }
  }
  __builtin_amdgcn_sched_barrier(0);
  GEN_COLD_STAMP(1, __builtin_amdgcn_s_memrealtime());
  // mLastPosition of every cable back (whole slots: a held cable keeps what it had)
#pragma unroll
  for (int g4 = 0; g4 < LP; ++g4)
    RB.store4_if(live, g4, col * 16u, make_float4(newpos[4 * g4], (4 * g4 + 1 < N) ? newpos[4 * g4 + 1] : 0.f, (4 * g4 + 2 < N) ? newpos[4 * g4 + 2] : 0.f,
                                                    (4 * g4 + 3 < N) ? newpos[4 * g4 + 3] : 0.f));

  // the windows that are not a uniform grid, compacted over the wave: one (robot, cable) per lane and pass
  if (__builtin_amdgcn_ballot_w64(need != 0u) != 0ull) {  // (wave-uniform)
    if (any_rot) {  // the fit reads other lanes' rings from the records: the turned ones have to be there
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    // (no divergent branches from here to the end of the function: per-lane cases are address selects - a lane without an
    //  item writes to its dump word, row 11 of cable 0's block - see the note on exec restores in DESIGN.md section 4)
    const uint32_t cnt = (uint32_t)__builtin_popcount(need);
    uint32_t slot = __hip_atomic_fetch_add(q_count, cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#pragma unroll
    for (int i = 0; i < N; ++i) {
      const bool has = (need & (1u << i)) != 0u;
      const float* park = cabf + i * kCabF + lane;
      const uint32_t hd = (__float_as_uint(park[6 * 64]) >> kGmHeadShift) & kGmField;
      uint32_t* const qi = has ? qitem(slot) : reinterpret_cast<uint32_t*>(cabf + 11 * 64 + lane);
      float* const qe = has ? qerr(slot) : cabf + 11 * 64 + lane;
      *qi = lane | ((uint32_t)i << 6) | ((uint32_t)sel[i] << 9) | (hd << 10) | (((__float_as_uint(park[6 * 64]) >> kGmRunShift) & kGmField) << 16);
      *qe = park[0];
      slot += has ? 1u : 0u;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const uint32_t total = *q_count;
    GEN_COLD_STAMP(2, __builtin_amdgcn_s_memrealtime());
    GEN_COLD_STAMP(4, (unsigned long long)total | (any_rot ? (1ull << 32) : 0ull));
    for (uint32_t first = 0; first < total; first += 64u) {  // (wave-uniform trip count)
      // every lane runs the fit - lanes past the end of the queue on a copy of item 0 - and only the result is predicated
      const uint32_t idx = first + lane;
      const bool mine = idx < total;
      const uint32_t it = *qitem(mine ? idx : 0u);
      const float e_new = *qerr(mine ? idx : 0u);
      const uint32_t ol = it & 63u, ci = (it >> 6) & 7u, sp = (it >> 9) & 1u, hd = (it >> 10) & 63u;
      const int irun = (int)((it >> 16) & 63u);  // consecutive steps ending at the sample just pushed: those stamps are implied
      const uint32_t ro = first_unit + ol;
      const uint32_t ocol = (ro < units) ? ro : (units - 1u);
      // the item's Pid of its cable, its owner's column: everything per lane goes into the vector offsets
      const uint32_t oa = ocol * 16u + (uint32_t)(L.block_a(0, (int)ci) - L.block_a(0, 0)) * RB.rs16 + (sp ? pid_a : 0u);
      const uint32_t ob = ocol * 4u + (uint32_t)L.block_b(0, (int)ci) * RB.rs4 + (sp ? pid_b : 0u);
      const int nbuf = __float_as_int(ptab[sp][2].x), degree = __float_as_int(ptab[sp][5].z);
      float y[NBMAX];
      int t[NBMAX];
#pragma unroll
      for (int s4 = 0; s4 < NV; ++s4) {
        const float4 v = RB.load4(L.block_a(0, 0) + s4, oa);
#pragma unroll
        for (int k = 0; k < 4; ++k)
          if (4 * s4 + k < NBMAX) y[4 * s4 + k] = comp4(v, k);
      }
#pragma unroll
      for (int j = 0; j < NBMAX; ++j) t[j] = RB.loadi(min(j, L.nb - 1), ob);
      const uint32_t old = (hd + 1u == (uint32_t)nbuf) ? 0u : hd + 1u;  // the oldest sample sits right after the head
      int t_old = now;
#pragma unroll
      for (int j = 0; j < NBMAX; ++j) {
        y[j] = ((uint32_t)j == hd) ? e_new : ((j < nbuf) ? y[j] : 0.f);  // the sample just pushed (its store may still be in flight)
        int age = (int)hd - j;
        age += (age < 0) ? nbuf : 0;
        t[j] = (j >= nbuf) ? now : ((age <= irun) ? now - age : t[j]);  // (age 0: the sample just pushed)
        t_old = ((uint32_t)j == old) ? t[j] : t_old;
      }
      const float res = (float)(gen_fit<NBMAX>(y, t, nbuf, degree, now, t_old) / (double)kc.dt);
      *(mine ? cabf + ci * kCabF + 9 * 64 + ol : cabf + 11 * 64 + lane) = res;
    }
    *q_count = 0u;  // for the next step (every lane writes the same word)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int i = 0; i < N; ++i) {
      const bool queued = (need & (1u << i)) != 0u;
      const bool sv = sel[i] != 0;
      const float* park = cabf + i * kCabF + lane;
      const uint32_t va = col * 16u + (sv ? pid_a : 0u), vb = col * 4u + (sv ? pid_b : 0u);
      float d_term;
      const float out = gen_finish(RB, gen_select(sv, ptab, filters), L.block_a(0, i) + L.nv(), L.block_b(0, i) + L.r_dfilt(), dcas_max, queued && live, queued && live, va, vb,
                                   cabf[i * kCabF + 9 * 64 + lane], park[0 * 64], park[1 * 64], park[2 * 64], park[3 * 64], park[4 * 64], park[5 * 64],
                                   __float_as_uint(park[6 * 64]), now, d_term);
      if (i == 0) {
        dbg.d = queued ? d_term : dbg.d;
        dbg.dw = dbg.dw || queued;
      }
      force[i] = queued ? out : force[i];
    }
  }
  GEN_COLD_STAMP(3, __builtin_amdgcn_s_memrealtime());
  return true;
}

// Helpers of the lean role-split kernel's cold tail (cdpr_general_split.hpp): a function called from a kernel receives every
// argument in vector registers; wave-uniform ones are made scalars again, LDS objects travel as their LDS addresses.
CDPR_DEV uint32_t lds_address(const void* p) { return (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) void*)p; }
template <typename T>
CDPR_DEV T* lds_pointer(uint32_t addr) { return (T*)(__attribute__((address_space(3))) T*)(uintptr_t)addr; }
CDPR_DEV uint32_t uni(uint32_t v) { return __builtin_amdgcn_readfirstlane(v); }
CDPR_DEV int uni(int v) { return (int)__builtin_amdgcn_readfirstlane((uint32_t)v); }
CDPR_DEV float uni(float v) { return __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(v))); }

// Issue the LDS-DMA of one wave's record slots for this step: per cable the selected Pid's NV value slots and H, plus the
// hold-position slots.  `keep`: a value every ordinary load issued so far feeds (hipcc drains every outstanding VMEM
// operation at the first use of an ordinary load's result while an LDS-DMA is pending).
template <int N, int NBMAX>
CDPR_DEV void gen_stage_records(const GenBuf& RB, const GenLayout L, uint32_t col, const int (&sel)[N], float4* cab, float4* hold_slots, float keep, bool need_hold,
                                bool skip_h = false, float* ierr_rows = nullptr, bool lane_fresh = false) {  // skip_h (wave-uniform): every lane's H slots come from its hot rows (gen_hot_restore)
  // ierr_rows (wave-uniform, round 6): a wave that holds robots with a hot word but is not all fresh - the integrals' dword rows ride the
  // DMA too, into N rows of 64 dwords of their own (gen_hot_restore read them from memory behind the wait for this DMA: a second round
  // trip in every such wave, and with cables switching Pids nearly every wave holds a robot without a word)
  constexpr int NV = gen_nv(NBMAX);
  constexpr int kCab = (NV + 1) * 64;
  constexpr int LP = (N + 3) / 4;
  asm volatile("" ::"v"(keep));
  const uint32_t pid_a = (uint32_t)L.pid_slots() * RB.rs16;
  if (need_hold) {  // (wave-uniform) mLastPosition is only read by a cable in the hold branch (JFC.cpp:81)
#pragma unroll
    for (int g4 = 0; g4 < LP; ++g4) RB.slot_to_lds(g4, col * 16u, hold_slots + g4 * 64);
  }
#pragma unroll
  for (int i = 0; i < N; ++i) {
    const uint32_t va = col * 16u + (sel[i] ? pid_a : 0u);
    const int sa = L.block_a(0, i);
#pragma unroll
    for (int s4 = 0; s4 < NV; ++s4) RB.slot_to_lds(sa + min(s4, L.nv() - 1), va, cab + i * kCab + s4 * 64);  // (slots past nv: a copy, weight 0)
    if (!skip_h) {
      // (ierr_rows: a fresh lane of such a wave takes its H slots from its hot rows like the lanes of an all-fresh wave - its request is
      //  sent out of the descriptor's range and moves nothing; gen_hot_restore writes the slot in LDS)
      RB.slot_to_lds(sa + L.nv(), (ierr_rows && lane_fresh) ? 0xFFFFFFFFu : va, cab + i * kCab + NV * 64);
      if (ierr_rows) RB.rowc_to_lds(2 + i, col * 4u, ierr_rows + i * 64);
    } else {
      RB.rowc_to_lds(2 + i, col * 4u, cab + i * kCab + NV * 64);  // (the cable's integral, a dword per lane: gen_hot_restore)
    }
  }
}

// NBMAX bounds the window at compile time: 11 (the shipped length and everything below it: 32 KiB of staged slots per
// wave) or 32 (the maximum: 72 KiB).  SINGLE: one world step per launch (no step loop: what is only needed late is not
// live from the start).
template <int N, bool FK, bool TD, bool ROLLOUT, int NBMAX, bool SINGLE = false>
__global__ __launch_bounds__(64, 1) void cdpr_gen_step_kernel(const StepArgs a, const GenCtl g) {
  constexpr int NP = cable_pairs(N);
  constexpr int G = joint_groups(N);
  constexpr int NV = gen_nv(NBMAX);
  constexpr int NBP = gen_nbp(NBMAX);
  constexpr int LP = (N + 3) / 4;
  __shared__ __attribute__((aligned(16))) float lds[NP * kGeomFloatsPerPair];
  __shared__ __attribute__((aligned(16))) float wrot[2][NBMAX][NBP];
  __shared__ float4 stage[N][NV + 1][64];  // gen_controller's working set: the DMA-staged value slots and H of every cable
  __shared__ float4 hold_slots[LP][64];
  __shared__ uint32_t q_count[kGenQueueWords];
  __shared__ float4 ptab[2][kGenPidFloats / 4];

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
  GenBuf RB = gen_buffer(g.rec, rs, g.rec_bytes, L);
  const uint32_t col = ru;  // record column: the robot, or this trajectory's private copy

  CDPR_STAMP(0);
  // every load of the prologue is issued before anything waits: one round trip to memory (a table loaded and stored to
  // LDS in a loop costs a round trip per pass)
  constexpr uint32_t kW4 = 2u * NBMAX * NBP / 4u;  // the weight table in float4
  constexpr int kWPass = (int)((kW4 + 63u) / 64u);
  static_assert((2 * NBMAX * NBP) % 4 == 0, "weight rows are float4 multiples");
  const float gval = (lane < NP * kGeomFloatsPerPair) ? a.geom[lane] : 0.f;
  float4 wv[kWPass];
#pragma unroll
  for (int j = 0; j < kWPass; ++j) wv[j] = reinterpret_cast<const float4*>(g.wtab)[min(lane + 64u * j, kW4 - 1u)];
  const float pv = g.ptab[min(lane, 2u * kGenPidFloats - 1u)];
  const uint32_t off = rr * 16u, woff = r * 16u;
  const float4 p0 = load_slot(a.state, st, 0, off), p1 = load_slot(a.state, st, 1, off), p2 = load_slot(a.state, st, 2, off),
               p3 = load_slot(a.state, st, 3, off);
  float4 p4 = make_float4(0.f, 0.f, 0.f, 1.f);
  if (FK) p4 = load_slot(a.state, st, 4, off);
  int mode = g.mode_arr ? (int)g.mode_arr[rr] : g.mode;
  // the latched command of the lane's mode (constant over the launch unless this is a rollout); issued with the state
  // loads: one round trip to memory, not two
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

  if (lane < NP * kGeomFloatsPerPair) lds[lane] = gval;
#pragma unroll
  for (int j = 0; j < kWPass; ++j)
    if (lane + 64u * j < kW4) reinterpret_cast<float4*>(&wrot[0][0][0])[lane + 64u * j] = wv[j];
  if (lane == 0) q_count[0] = 0u;
  if (lane < 2 * kGenPidFloats) (&ptab[0][0].x)[lane] = pv;

  if (ROLLOUT) {
    // private copy of the robot's records; a Joy on jointVelocities reaching a robot that is not in Velocity mode resets
    // its velocity Pid (JFC.cpp:113-115): those slots and rows start from zero
    const bool reset_vel = (mode != 2);
    const float4* srca = reinterpret_cast<const float4*>(g.src_rec);
    float4* dsta = reinterpret_cast<float4*>(g.rec);
    const int va0 = L.block_a(1, 0), va1 = va0 + L.pid_slots();
    for (int sl = 0; sl < L.slots(); ++sl) {
      const float4 v = (reset_vel && sl >= va0 && sl < va1) ? make_float4(0.f, 0.f, 0.f, 0.f) : srca[(size_t)sl * g.src_rstride + rr];
      if (live) dsta[(size_t)sl * rs + col] = v;
    }
    const float* srcb = g.src_rec + (size_t)L.slots() * g.src_rstride * 4;
    float* dstb = g.rec + (size_t)L.slots() * rs * 4;
    const int vb0 = L.block_b(1, 0), vb1 = vb0 + L.pid_rows();
    for (int row = 0; row < L.rows(); ++row) {
      const float v = (reset_vel && row >= vb0 && row < vb1) ? 0.f : srcb[(size_t)row * g.src_rstride + rr];
      if (live) dstb[(size_t)row * rs + col] = v;
    }
    for (int row = L.rows(); row < L.rows() + L.hot_rows(); ++row) {  // the hot rows (GenHot) travel with the records they stand for
      if (live) dstb[(size_t)row * rs + col] = srcb[(size_t)row * g.src_rstride + rr];
    }
    mode = 2;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

#ifdef CDPR_STAMPS_PRO
  CDPR_STAMP(2);
#endif
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
  int nm0 = g.now_step % max(g.nbuf0, 1), nm1 = g.now_step % max(g.nbuf1, 1);  // this step's ring slot per Pid (scalar)
  for (int step = 0; step < (SINGLE ? 1 : a.nsteps); ++step) {
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
    // (opaque per step: in a launch of several steps every slot offset is loop-invariant, and hoisted out of the step loop
    //  they overflow the scalar register file into v_writelane / v_readlane pairs)
    if (!SINGLE) asm volatile("" : "+s"(RB.rs16), "+s"(RB.rs4));
    const bool hot_on = g.hot != 0 && g.simple_ok != 0;
    uint32_t hot_step1 = 0u, hot_mask = 0u;  // the robot's hot-row word (GenHot), on its way under the selection below
    if (hot_on) hot_step1 = RB.loadc(0, col * 4u), hot_mask = RB.loadc(1, col * 4u);
    // ---- which Pid serves each cable this step (JFC.cpp:67-89), and its slots on their way to LDS
    int sel[N];  // 0 position Pid, 1 velocity Pid (Force mode: 0, unused)
#pragma unroll
    for (int i = 0; i < N; ++i) {
      sel[i] = (mode == 2 && fabsf(target[i]) > g.eps) ? 1 : 0;
      // (opaque: in a launch of several steps the selection is the same in every step, and the compiler would hoist the
      //  selected gains of all cables - some hundred registers - out of the step loop)
      asm volatile("" : "+v"(sel[i]));  // (also keeps the selection a 0 / 1 register: as eight lane masks it crowds the scalar file)
    }
    const bool run_ctl = !first_world;
    GenHot hot{false, false, false, 0u, 0};  // hot rows (GenHot): this kernel restores and clears them, it does not start them
    bool hot_skip = false;
#ifdef CDPR_STAMPS_PRO
    { float k2 = 0.f;
#pragma unroll
      for (int i = 0; i < N; ++i) k2 += target[i] + (float)sel[i];
      asm volatile("" ::"v"(k2)); }
    CDPR_STAMP(3);
#endif
    if (run_ctl) {
      float keep = (s.px + s.qy) + (s.vy + s.wz) + fkqw;
#pragma unroll
      for (int i = 0; i < N; ++i) keep += target[i];
      bool holds = false;  // some cable of this lane is in the hold branch (JFC.cpp:78-82)
#pragma unroll
      for (int i = 0; i < N; ++i) holds = holds || (mode == 2 && sel[i] == 0);
      hot = gen_hot_begin<N>(hot_on, hot_step1, hot_mask, sel, mode, now, hot_skip);
      gen_stage_records<N, NBMAX>(RB, L, col, sel, &stage[0][0][0], &hold_slots[0][0], keep, __builtin_amdgcn_ballot_w64(holds) != 0ull, hot_skip);
    }
    GEN_PHASE_STAMP(1);

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

    GEN_PHASE_STAMP(2);
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

    GEN_PHASE_STAMP(3);
    // ---- per-cable force: the general controller (gen_controller above)
    float force[N];
#pragma unroll
    for (int i = 0; i < N; ++i) force[i] = 0.f;
    GenDbg dbg{0.f, 0.f, 0.f, 0.f, false, false};
    if (run_ctl) {
      __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): the DMA has landed
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      CDPR_STAMP(4);
      GenCtlConst cc;
      cc.pcas_max = g.pcas_max, cc.dcas_max = g.dcas_max, cc.dt = g.dt, cc.inv_dt = a.inv_dt;
      cc.nm0 = nm0, cc.nm1 = nm1, cc.nbuf0 = g.nbuf0, cc.simple_ok = g.simple_ok != 0;
      cc.force_rows = nullptr;
#ifdef CDPR_STAMPS
      cc.stamps = a.stamps ? a.stamps + (size_t)blockIdx.x * 8 : nullptr;
#endif
      gen_hot_restore<N, NBMAX>(cc, RB, L, lane, live, col, hot, sel, &stage[0][0][0]);
      gen_controller<N, NBMAX, false, false>(cc, RB, L, lane, live, col, blockIdx.x * 64u, units, mode, now, target, sel, q, qd, &stage[0][0][0], &hold_slots[0][0],
                               &wrot[0][0][0], ptab, q_count, force, dbg, hot);
    }
    CDPR_STAMP(5);
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
      if (dbg.pi) {
        d[0] = dbg.p;
        d[1] = dbg.i;
        d[3] = dbg.des;
      }
      if (dbg.dw) d[2] = dbg.d;
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

    CDPR_STAMP(6);
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
    nm0 = (nm0 + 1 == g.nbuf0) ? 0 : nm0 + 1;
    nm1 = (nm1 + 1 == g.nbuf1) ? 0 : nm1 + 1;
#ifdef CDPR_HYP_STEP_FENCE  // diagnosis: everything of this step (global and LDS) complete before the next one starts
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    __builtin_amdgcn_s_barrier();
#endif
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
  CDPR_STAMP(7);
}

// Per-robot command arrival on the general path (cdpr_set_*_command_masked; PLG.cpp:206-219 per model): one thread per
// robot copies the Joy's row into the latched buffer of its kind, and entering the mode from another one resets that
// mode's Pid (JFC.cpp:101-103,113-115) = zero the robot's column of the Pid's slots and rows.  setForce resets nothing.
struct GenLatchArgs {
  const uint8_t* mask;   // uint8[B], or nullptr = every robot
  uint8_t* mode;         // per-robot mode
  const float* pending;  // float[B][n]
  float* latched;        // float[B][n]
  float* rec;
  uint32_t rstride, batch, n;
  int reset_pid;         // 0 / 1: the Pid whose records this mode owns; -1: none (setForce)
  GenLayout lay;
  int new_mode;
};

// The hot rows (GenHot) written back: every robot with a valid word gets the H slots of its word's Pids as they stand and
// its word cleared.  Runs in front of anything that resets a Pid's records from outside the step kernels (a mode change),
// so that those see - and zero - plain records.
struct GenFlushArgs {
  float* rec;
  uint32_t rstride, batch;
  GenLayout lay;
  int nbuf;
};
static __global__ __launch_bounds__(256) void cdpr_gen_flush_hot_kernel(const GenFlushArgs a) {
  const uint32_t r = blockIdx.x * 256u + threadIdx.x;
  if (r >= a.batch) return;
  uint32_t* const c = reinterpret_cast<uint32_t*>(a.rec) + ((size_t)a.lay.slots() * 4 + (size_t)a.lay.rows()) * a.rstride + r;
  const uint32_t step1 = c[0];
  if (step1 == 0u) return;
  const uint32_t mask = c[a.rstride];
  const int step = (int)step1 - 1;
  const uint32_t meta = kGmWasLast | ((uint32_t)a.nbuf << kGmCountShift) | ((uint32_t)(step % a.nbuf) << kGmHeadShift) | (kGmField << kGmRunShift);
  float4* const slots = reinterpret_cast<float4*>(a.rec);
  for (int i = 0; i < a.lay.n; ++i) {
    const float ierr = __uint_as_float(c[(size_t)(2 + i) * a.rstride]);
    slots[(size_t)(a.lay.block_a((int)((mask >> i) & 1u), i) + a.lay.nv()) * a.rstride + r] = make_float4(__uint_as_float(meta), __int_as_float(step), ierr, 0.f);
  }
  c[0] = 0u;
}

static __global__ __launch_bounds__(256) void cdpr_gen_latch_kernel(const GenLatchArgs a) {
  const uint32_t r = blockIdx.x * 256u + threadIdx.x;
  if (r >= a.batch) return;
  if (a.mask && !a.mask[r]) return;
  for (uint32_t i = 0; i < a.n; ++i) a.latched[(size_t)r * a.n + i] = a.pending[(size_t)r * a.n + i];
  if ((int)a.mode[r] != a.new_mode) {
    if (a.reset_pid >= 0) {
      float4* sa = reinterpret_cast<float4*>(a.rec) + (size_t)a.lay.block_a(a.reset_pid, 0) * a.rstride + r;
      for (int sl = 0; sl < a.lay.pid_slots(); ++sl) sa[(size_t)sl * a.rstride] = make_float4(0.f, 0.f, 0.f, 0.f);
      float* rb = a.rec + (size_t)a.lay.slots() * a.rstride * 4 + (size_t)a.lay.block_b(a.reset_pid, 0) * a.rstride + r;
      for (int row = 0; row < a.lay.pid_rows(); ++row) rb[(size_t)row * a.rstride] = 0.f;
    }
    a.mode[r] = (uint8_t)a.new_mode;
  }
}

}  // namespace cdpr
