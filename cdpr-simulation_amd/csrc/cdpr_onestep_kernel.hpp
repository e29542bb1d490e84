// cdpr_onestep_kernel.hpp — the one-world-step-per-launch kernel (gfx950, fp32, lane-per-robot), second generation.
//
// Same arithmetic as cdpr_step_kernel<N, FK, TD, SINGLE = true> (the device functions are shared, the file is
// compiled with -ffp-contract=off, so the two are bit-identical — tested), re-staged around what the profile of the
// first generation showed (profiles/r02a_onestep_summary.json: one wave per SIMD, 47 % of wave cycles issuing,
// 27 % waiting on memory, 25 % dependent-issue stalls):
//
//   * the controller rows (20 ring rows + 2 integral rows at n = 8: 22 of the 29 rows a robot reads) are NOT needed
//     until the per-cable PID, but as ordinary loads they either stall the wave right away or sit in 88 VGPRs while
//     the Newton stage wants every register.  They go global -> LDS directly (global_load_lds_dwordx4: one 1 KiB
//     row per wave-instruction, lane-linear, no VGPR destination), are issued once the platform rows have arrived,
//     stay in flight under the whole Newton-Raphson stage, and are read back with ds_read_b128 for the PID;
//   * the stage order is IK -> early observables -> Newton FK -> [wait for the DMA] PID -> TD -> world step, so the
//     only memory wait on the critical path is the first round trip of the 5 platform rows;
//   * the observables that are final after the IK (pose, twist, joint position, joint velocity: 7 of 10 rows) are
//     stored before the Newton stage: the stores drain under compute instead of forming a tail;
//   * nothing of the true-state IK but the measured lengths L* lives through the Newton stage: the structure
//     matrix is rebuilt afterwards (one more IK evaluation, ~4 % more instructions) instead of holding 64 VGPRs.
//
// LDS per wave: 16 floats x cable pairs of geometry + (controller rows + command rows) x 1 KiB = 24.25 KiB at n = 8;
// 4 waves per CU (one per SIMD) use 97 KiB of the CU's 160 KiB.
#pragma once
#include "cdpr_step_kernel.hpp"

namespace cdpr {

// One controller / command row of the calling wave, global -> LDS without a register destination: lane l copies the
// 16 bytes at src to dst_row + 16 l (the LDS side of an LDS-DMA is always wave-base + lane * size).
CDPR_DEV void row_to_lds(const float4* src, float4* dst_row) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)dst_row, 16, 0, 0);
}

// Row stores go through store_slot (one buffer descriptor per row).  One descriptor per BUFFER with the row as the scalar
// offset operand was measured too: 100 fewer scalar instructions and 14 -> 6 spilled SGPRs, the same time within noise
// (profiles/r02f_onestep_v2_variants_ab.txt) - and the FK instantiations at n = 8 then failed the bit-identity and
// parity tests on MI355X (cause not found in the ISA), so that form is not in the tree.  Also measured, neither faster:
// weights staged in LDS instead of kernel arguments, cable constants re-read per Newton iteration.

// Pid::update for every cable pair (Pid.cpp:122-191; the same statements as in cdpr_step_kernel).  PR (per-robot handles):
// `calls` and `is_vel` are the lane's own, the coefficients and weights are selected per lane, and a robot whose Pid was
// just reset (calls == 0: "first call returns 0", Pid.cpp:123-126) keeps force 0 and its integral.
template <int NP, bool PR = false>
CDPR_DEV void pid_pairs(const StepArgs& a, int calls, bool is_vel, const v2f (&desired)[NP], const v2f (&actual)[NP], const v2f (&win)[NP][kWin],
                        v2f (&ierr)[NP], v2f (&f)[NP], v2f (&e_new)[NP], float& dbg_p, float& dbg_i, float& dbg_d, bool is_force = false) {
  const PidCoef c = pid_coef<PR>(a, is_vel);
  const bool run = (!PR || calls != 0) && !is_force;  // is_force (PR only): the robot is driven open loop (JFC.cpp:67-70)
  const bool full = calls >= c.nbuf;  // derive(): 0 until the window holds nbuf samples (Pid.cpp:200-203)
  // the host put the weights of this launch's ring position (StepArgs::ring_slot) into the arguments; on per-robot handles
  // both Pids fit the same window (same length and degree: cdpr_create sends anything else down the general path), so
  // the weights stay scalars whatever the lane's mode
  const float (&wt)[kWin + 2] = a.wrow;
  v2f error[NP], acc[NP];
#pragma unroll
  for (int k = 0; k < NP; ++k) {
    error[k] = desired[k] - actual[k];
    acc[k] = splat(wt[kWin]) * error[k];
  }
#pragma unroll
  for (int j = 0; j < kWin; ++j) {
#pragma unroll
    for (int k = 0; k < NP; ++k) acc[k] = fma2(wt[j], win[k][j], acc[k]);
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
}

// PERSIST (batches beyond one robot per hardware lane): the grid is one wave per SIMD and every wave walks over blocks of
// 64 robots, blockIdx.x, blockIdx.x + gridDim.x, ...  A launch whose waves each take ONE block is bulk-synchronous (load,
// ~8 us of arithmetic, store), and a second wave on the same SIMD runs in phase with the first and adds its whole issue
// time (profiles/r04_cliff_analysis.txt); here the NEXT block's rows are requested while the current block computes -
// its controller rows and Joy by LDS-DMA as soon as the PID stage has read the staged ones, its platform rows into
// registers right after - so from the second block on a wave never waits for memory.
// A row load the compiler does not track (inline asm): the persistent kernel requests the next block's platform rows a
// whole block ahead and waits for them with an explicit vmcnt(N) that leaves the N later operations (the next block's
// LDS-DMA, this block's last observable stores) in flight.  Through ordinary loads the compiler would drain EVERY
// outstanding memory operation at their first use (it does so whenever an LDS-DMA is pending), and the DMA - 24 KiB per
// wave that the next block needs 9 us later - would have to land within 2 us of its issue, all waves at once.
typedef float f32x4 __attribute__((ext_vector_type(4)));
CDPR_DEV f32x4 load_slot_untracked(const float4* base, size_t stride, int slot, uint32_t off) {
  const char* p = reinterpret_cast<const char*>(base + (size_t)slot * stride) + off;
  f32x4 v;
  // (into ACCUMULATION registers: the compiler takes an asm's result for valid at once and may copy it anywhere - a
  //  vector-register result it moved into a spare AGPR straight away, before the data was there; an AGPR result has no
  //  reason to move before its use.  Checked in the ISA: nothing reads these registers between the load and the wait.)
  asm volatile("global_load_dwordx4 %0, %1, off" : "=a"(v) : "v"(p) : "memory");
  return v;
}
// s_waitcnt immediate (gfx9): vmcnt = n (6 bits, split 3:0 | 15:14), expcnt and lgkmcnt untouched
constexpr int vmcnt_imm(int n) { return (n & 15) | ((n >> 4) << 14) | (7 << 4) | (15 << 8); }

template <int N, bool FK, bool TD, bool PERSIST = false>
__global__ __launch_bounds__(64, 1) void cdpr_onestep_kernel(const StepArgs a_in) {
  const StepArgs& a = a_in;  // (shadowed inside the block loop, see there)
  constexpr int NP = cable_pairs(N);
  constexpr int P = plat_slots(FK);
  constexpr int G = joint_groups(N);
  constexpr int NH = (NP + 1) / 2;           // integral ("hot") rows
  constexpr int kCtrl = 5 * NP + NH;         // controller rows of one robot
  constexpr bool kCmdLds = (N % 4) == 0;     // Joy rows are float4-addressable: stage them through LDS as well
  constexpr int kCmdRows = kCmdLds ? N / 4 : 0;
  __shared__ __attribute__((aligned(16))) float lds[NP * kGeomFloatsPerPair];
  __shared__ float4 stage[kCtrl + (kCmdRows ? kCmdRows : 1)][64];

  const uint32_t lane = threadIdx.x;
  const size_t st0 = a.stride;
  const uint32_t nblocks = (a.batch + 63u) / 64u;
  uint32_t blk = blockIdx.x;
  uint32_t r = blk * 64u + lane;
  uint32_t rr = (r < a.batch) ? r : (a.batch - 1u);  // tail lanes shadow the last robot, stores are masked
  bool live = r < a.batch;
  uint32_t off = rr * 16u, woff = r * 16u;

  // ---- loads that the first stages need: geometry (oldest), platform rows, and the Joy when it is not LDS-staged
  CDPR_STAMP(0);
  const float gval = (lane < NP * kGeomFloatsPerPair) ? a.geom[lane] : 0.f;
  float4 p0 = load_slot(a.state, st0, 0, off), p1 = load_slot(a.state, st0, 1, off), p2 = load_slot(a.state, st0, 2, off),
         p3 = load_slot(a.state, st0, 3, off);
  float4 p4 = make_float4(0.f, 0.f, 0.f, 1.f);
  if (FK) p4 = load_slot(a.state, st0, 4, off);
  if (lane < NP * kGeomFloatsPerPair) lds[lane] = gval;
  const bool force_mode = (a.flags & kFlagForceMode) != 0u;  // UpdateMode::Force (JFC.cpp:67-70): no Pid, no controller rows
  const bool first_world = (a.flags & kFlagFirstWorldStep) != 0u;
  const bool run_pid = !first_world && !force_mode && a.pid_calls != 0;  // wave-uniform (Pid.cpp:123-126: the first call returns 0)
  bool staged = false;  // PERSIST: this block's controller rows were requested during the previous block

  for (;;) {  // (one pass unless PERSIST)
  // (opaque per block: every row address of the loop is loop-invariant, and hoisted out of it they overflow the scalar
  //  register file into v_writelane / v_readlane pairs - measured: 762 of them)
  size_t st = st0;
  uint32_t goff = 0;  // (likewise the geometry in LDS: hoisted, its 7n values would ride through the whole loop in registers)
  if (PERSIST) asm volatile("" : "+s"(st), "+v"(goff));
  const float* const geo = lds + goff;
  // ... and the kernel arguments themselves: every scalar the loop body reads would stay in a register across the whole
  // loop, and everything computed from them once (packed broadcasts, negated constants) with it; read through an opaque
  // pointer to the kernarg segment they are scalar loads of THIS block, as in the one-block kernel
  const StepArgs a = block_args<PERSIST>(a_in);
  v2f desired[NP];
#pragma unroll
  for (int k = 0; k < NP; ++k) desired[k] = splat(0.f);
  const float* cp = a.cmd + (size_t)rr * N;  // never null: before the first Joy the latched buffer holds zeros
  if (!kCmdLds || force_mode) {
#pragma unroll
    for (int i = 0; i < N; ++i) {
      if (i & 1)
        desired[i / 2].y = cp[i];
      else
        desired[i / 2].x = cp[i];
    }
  }
  // single-wave workgroup: LDS operations of one wave execute in order (see cdpr_step_kernel)
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

  Platform s;
  s.px = p0.x; s.py = p0.y; s.pz = p0.z; s.qx = p0.w;
  s.qy = p1.x; s.qz = p1.y; s.qw = p1.z; s.vx = p1.w;
  s.vy = p2.x; s.vz = p2.y; s.wx = p2.z; s.wy = p2.w;
  s.wz = p3.x;
  float fkx = p3.y, fky = p3.z, fkz = p3.w, fkqx = p4.x, fkqy = p4.y, fkqz = p4.z, fkqw = p4.w;

  // ---- controller rows and the Joy: global -> LDS, in flight until the PID stage.  Issued AFTER the platform rows
  //      are in registers: hipcc drains every outstanding VMEM operation (vmcnt(0)) at the first use of an ordinary
  //      load's result while an LDS-DMA is pending, so the DMA must not be pending yet when the platform rows are used.
  {
    // every ordinary load issued so far is consumed here (data dependence: no use of one can sink below the DMA issue)
    float keep = (s.px + s.qy) + (s.vy + s.wz) + fkqw;
#pragma unroll
    for (int k = 0; k < NP; ++k) keep += desired[k].x + desired[k].y;
    asm volatile("" ::"v"(keep));
    if (run_pid && !staged) {
#pragma unroll
      for (int j = 0; j < kCtrl; ++j)
        row_to_lds(reinterpret_cast<const float4*>(reinterpret_cast<const char*>(a.state + (size_t)(P + j) * st) + off), &stage[j][0]);
      if (kCmdLds) {
#pragma unroll
        for (int g = 0; g < kCmdRows; ++g) row_to_lds(reinterpret_cast<const float4*>(cp) + g, &stage[kCtrl + g][0]);
      }
    }
  }

  // ---- PERSIST: the next block's platform rows, requested a whole block ahead (untracked: see load_slot_untracked)
  const uint32_t nblk = blk + gridDim.x;
  const bool more = PERSIST && nblk < nblocks;
  uint32_t nr = 0, nrr = 0;
  f32x4 n0 = {0.f, 0.f, 0.f, 0.f}, n1 = n0, n2 = n0, n3 = n0, n4 = {0.f, 0.f, 0.f, 1.f};
  if (PERSIST && more) {
    nr = nblk * 64u + lane;
    nrr = (nr < a.batch) ? nr : (a.batch - 1u);
    const uint32_t noff = nrr * 16u;
    n0 = load_slot_untracked(a.state, st, 0, noff), n1 = load_slot_untracked(a.state, st, 1, noff);
    n2 = load_slot_untracked(a.state, st, 2, noff), n3 = load_slot_untracked(a.state, st, 3, noff);
    if (FK) n4 = load_slot_untracked(a.state, st, 4, noff);
  }

  CDPR_STAMP(1);
  // ---- IK on the state at t_k: joint positions and rates (observables), measured lengths L* for the estimator
  v2f len[NP], q[NP], qd[NP], jac[NP][6];
  {
    v2f l0[NP];
    ik_pairs<N, true>(geo, s.px, s.py, s.pz, s.qx, s.qy, s.qz, s.qw, len, jac, l0);
#pragma unroll
    for (int k = 0; k < NP; ++k) {
      q[k] = l0[k] - len[k];
      qd[k] = -fma2(s.wz, jac[k][5], fma2(s.wy, jac[k][4], fma2(s.wx, jac[k][3],
                    fma2(s.vz, jac[k][2], fma2(s.vy, jac[k][1], splat(s.vx) * jac[k][0])))));
    }
  }
  const bool publish = (a.publish_mask & 1ull) != 0ull;
  if (publish && live) {  // the part of the observables that is final already (PLG.cpp:248-280)
    store_slot(a.obs, st, 0, woff, make_float4(s.px, s.py, s.pz, s.qx));
    store_slot(a.obs, st, 1, woff, make_float4(s.qy, s.qz, s.qw, s.vx));
    store_slot(a.obs, st, 2, woff, make_float4(s.vy, s.vz, s.wx, s.wy));
#pragma unroll
    for (int g = 0; g < G; ++g) {
      const int k0 = 2 * g, k1 = (2 * g + 1 < NP) ? 2 * g + 1 : 2 * g;
      const bool has = (2 * g + 1 < NP);
      store_slot(a.obs, st, 4 + g, woff, make_float4(q[k0].x, q[k0].y, has ? q[k1].x : 0.f, has ? q[k1].y : 0.f));
      store_slot(a.obs, st, 4 + G + g, woff, make_float4(qd[k0].x, qd[k0].y, has ? qd[k1].x : 0.f, has ? qd[k1].y : 0.f));
    }
  }

  CDPR_STAMP(2);
  // ---- Newton-Raphson forward kinematics ([NEW] SURVEY 8(a) row 14); only len lives in from the true state
  float fk_res = 0.f;
  int fk_it = 0;
  v2f jest[NP][6];
  if (FK) {
    v2f elen[NP], unused[NP];
    bool active = true;
    for (int it = 0; it < a.fk_iters; ++it) {
      ik_pairs<N, false>(geo, fkx, fky, fkz, fkqx, fkqy, fkqz, fkqw, elen, jest, unused);
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
    ik_pairs<N, false>(geo, fkx, fky, fkz, fkqx, fkqy, fkqz, fkqw, elen, jest, unused);
    v2f rm = splat(0.f);
#pragma unroll
    for (int k = 0; k < NP; ++k) rm = max2(rm, abs2(len[k] - elen[k]));
    fk_res = fmaxf(rm.x, rm.y);
  }

  CDPR_STAMP(3);
  // ---- per-cable force (PLG.cpp:222-228 -> JFC.cpp:59-96 -> Pid.cpp:122-191) on the controller rows from LDS
  v2f f[NP], e_new[NP], ierr[NP];
#pragma unroll
  for (int k = 0; k < NP; ++k) {
    f[k] = splat(0.f);
    e_new[k] = splat(0.f);
    ierr[k] = splat(0.f);
  }
  float dbg_p = 0.f, dbg_i = 0.f, dbg_d = 0.f;
  bool dbg_wrote = false;
  if (run_pid) {
    // the DMA has landed once vmcnt reaches 0 (its LDS writes are counted there).  hipcc tracks an LDS-DMA against the
    // ds_reads of the same LDS object and places this wait by itself; it is spelled out anyway.  No "memory" clobber:
    // that would stop the derivative weights below from being scalar loads.  0x0F70 = vmcnt(0), expcnt / lgkmcnt untouched.
    __builtin_amdgcn_s_waitcnt(0x0F70);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    CDPR_STAMP(4);
    if (kCmdLds) {
#pragma unroll
      for (int g = 0; g < kCmdRows; ++g) {
        const float4 v = stage[kCtrl + g][lane];
        desired[2 * g] = (v2f){v.x, v.y};
        desired[2 * g + 1] = (v2f){v.z, v.w};
      }
    }
    v2f win[NP][kWin];
#pragma unroll
    for (int k = 0; k < NP; ++k) {
#pragma unroll
      for (int m = 0; m < 5; ++m) {
        const float4 w = stage[5 * k + m][lane];
        win[k][2 * m] = (v2f){w.x, w.y};
        win[k][2 * m + 1] = (v2f){w.z, w.w};
      }
    }
#pragma unroll
    for (int g = 0; g < NH; ++g) {
      const float4 h = stage[5 * NP + g][lane];
      ierr[2 * g] = (v2f){h.x, h.y};
      if (2 * g + 1 < NP) ierr[2 * g + 1] = (v2f){h.z, h.w};
    }
    const bool actual_is_vel = (a.flags & kFlagActualIsVelocity) != 0u;
    v2f actual[NP];
#pragma unroll
    for (int k = 0; k < NP; ++k) actual[k] = actual_is_vel ? qd[k] : q[k];
    const int ring_slot = a.ring_slot;
    pid_pairs<NP>(a, a.pid_calls, actual_is_vel, desired, actual, win, ierr, f, e_new, dbg_p, dbg_i, dbg_d);
    dbg_wrote = true;
    if (live) {  // one ring row per cable pair (the one that takes the new error) + the integral rows
#pragma unroll
      for (int m = 0; m < 5; ++m) {
        if (m == (ring_slot >> 1)) {
#pragma unroll
          for (int k = 0; k < NP; ++k) CDPR_STORE_STATE(a.state, st, P + 5 * k + m, woff, ring_row(win[k], m, e_new[k], ring_slot));
        }
      }
#pragma unroll
      for (int g = 0; g < NH; ++g) {
        const int k1 = (2 * g + 1 < NP) ? 2 * g + 1 : 2 * g;
        CDPR_STORE_STATE(a.state, st, P + 5 * NP + g, woff, make_float4(ierr[2 * g].x, ierr[2 * g].y, ierr[k1].x, ierr[k1].y));
      }
    }
  }

  if (force_mode && !first_world) {
#pragma unroll
    for (int k = 0; k < NP; ++k) f[k] = desired[k];
  }

  // ---- PERSIST: the staged rows have been read: request the next block's (controller rows and Joy by LDS-DMA into the
  //      same staging buffer, platform rows into registers); they arrive under this block's remaining stages
  if (PERSIST && more) {
    const uint32_t noff = nrr * 16u;
    if (run_pid) {
#pragma unroll
      for (int j = 0; j < kCtrl; ++j)
        row_to_lds(reinterpret_cast<const float4*>(reinterpret_cast<const char*>(a.state + (size_t)(P + j) * st) + noff), &stage[j][0]);
      if (kCmdLds) {
        const float* ncp = a.cmd + (size_t)nrr * N;
#pragma unroll
        for (int g = 0; g < kCmdRows; ++g) row_to_lds(reinterpret_cast<const float4*>(ncp) + g, &stage[kCtrl + g][0]);
      }
    }
  }

  CDPR_STAMP(5);
  // ---- tension distribution ([NEW] SURVEY 8(a) row 15), SetForce limits
  v2f applied[NP];
  int td_flag = 0;
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

  if (a.dbg && live) {  // `pid` topic, cable 0 only (PLG.cpp:223-227; Pid.cpp:139-142,158-168)
    float* d = a.dbg + (size_t)r * 9;
    if (dbg_wrote) {
      d[0] = dbg_p;
      d[1] = dbg_i;
      d[2] = dbg_d;
      d[3] = desired[0].x;
    }
    d[4] = applied[0].x;
  }
  if (publish && live) {  // the rest of the observables
    store_slot(a.obs, st, 3, woff, make_float4(s.wz, fk_res, (float)fk_it, pack_flags(td_flag, travel_mask<N>(a, q))));
#pragma unroll
    for (int g = 0; g < G; ++g) {
      const int k0 = 2 * g, k1 = (2 * g + 1 < NP) ? 2 * g + 1 : 2 * g;
      const bool has = (2 * g + 1 < NP);
      store_slot(a.obs, st, 4 + 2 * G + g, woff,
                make_float4(applied[k0].x, applied[k0].y, has ? applied[k1].x : 0.f, has ? applied[k1].y : 0.f));
    }
  }

  CDPR_STAMP(6);
  // ---- world step to t_{k+1}: wrench = -J^T (applied - d qdot) + m g
  {
    v2f tens[NP];
#pragma unroll
    for (int k = 0; k < NP; ++k) {
      tens[k] = fma2(-a.damping, qd[k], applied[k]);
      if (a.unilateral) tens[k] = max2(tens[k], splat(0.f));
    }
    if (FK) {
      // the true structure matrix did not live through the Newton stage: rebuild it (same inputs, same instructions,
      // same bits).  The geometry pointer is made opaque so the compiler cannot keep the first evaluation alive instead.
      uint32_t again = 0;  // an opaque zero OFFSET (an opaque pointer would lose the LDS address space: flat loads)
      asm volatile("" : "+v"(again));
      v2f len2[NP], l02[NP];
      ik_pairs<N, false>(lds + again, s.px, s.py, s.pz, s.qx, s.qy, s.qz, s.qw, len2, jac, l02);
    }
    float w[6];
    jt_times<NP>(jac, tens, w);
    w[0] = a.fgx - w[0];
    w[1] = a.fgy - w[1];
    w[2] = a.fgz - w[2];
    w[3] = -w[3];
    w[4] = -w[4];
    w[5] = -w[5];
    integrate(a, s, w);
  }
  if (PERSIST && more) {
    // the next block's rows are waited for HERE, before this block's last stores are issued: the compiler drains every
    // outstanding memory operation at the first use of a load's result while an LDS-DMA is pending, and at the top of
    // the next block that would include these stores
    // (everything issued before the last kDmaOps operations is complete: the platform rows went out before the DMA)
    constexpr int kDmaOps = kCtrl + kCmdRows;
    static_assert(kDmaOps < 64, "vmcnt is a 6-bit counter");
    if (run_pid)
      __builtin_amdgcn_s_waitcnt(vmcnt_imm(kDmaOps));
    else
      __builtin_amdgcn_s_waitcnt(vmcnt_imm(0));
    asm volatile("" : "+a"(n0), "+a"(n1), "+a"(n2), "+a"(n3), "+a"(n4));
  }
  if (live) {
    CDPR_STORE_STATE(a.state, st, 0, woff, make_float4(s.px, s.py, s.pz, s.qx));
    CDPR_STORE_STATE(a.state, st, 1, woff, make_float4(s.qy, s.qz, s.qw, s.vx));
    CDPR_STORE_STATE(a.state, st, 2, woff, make_float4(s.vy, s.vz, s.wx, s.wy));
    CDPR_STORE_STATE(a.state, st, 3, woff, make_float4(s.wz, fkx, fky, fkz));
    if (FK) CDPR_STORE_STATE(a.state, st, 4, woff, make_float4(fkqx, fkqy, fkqz, fkqw));
  }
  CDPR_STAMP(7);
  if (!(PERSIST && more)) break;
  blk = nblk;
  r = nr;
  rr = nrr;
  live = r < a.batch;
  off = rr * 16u, woff = r * 16u;
  p0 = make_float4(n0.x, n0.y, n0.z, n0.w), p1 = make_float4(n1.x, n1.y, n1.z, n1.w), p2 = make_float4(n2.x, n2.y, n2.z, n2.w);
  p3 = make_float4(n3.x, n3.y, n3.z, n3.w), p4 = make_float4(n4.x, n4.y, n4.z, n4.w);
  staged = run_pid;
  }  // blocks
}

// =====================================================================================================================
// cdpr_split_kernel — the one-step launch for FK + TD handles with TWO waves per 64 robots (a 128-thread workgroup), each
// with its own role, so that every SIMD hosts two waves although 65 536 robots are only 1 024 wavefronts of lanes:
//
//   estimator wave (wave 0)   platform rows -> measured lengths -> Newton-Raphson FK (4 iterations + closing evaluation)
//                             -> [forces from the controller wave] tension distribution -> tensions back
//   controller wave (wave 1)  platform + controller rows -> IK (structure matrix, joint positions / rates) -> early
//                             observables -> per-cable PID -> forces out -> [tensions from the estimator wave] SetForce
//                             limits -> remaining observables -> world step -> state
//
// Why: one wave issues a vector instruction every 5 cycles at best (8.3 when it needs its predecessor's result), while
// the SIMD takes a plain instruction every 2.5 cycles and a packed one every 4.2 (scripts/micro/valu_issue.hip): the
// scarce thing is the issue rate of the wave that carries a robot's serial chain, so everything that is not on that
// chain belongs on another (younger, lower-priority) wave that uses the slots the first leaves.  The lane-pair mapping gets its second wave by giving every robot two lanes, which
// duplicates the serial 6x6 solves (+46 % instructions: measured slower from 65 536 robots on).  Splitting by ROLE
// duplicates only the platform-row loads and one length evaluation (~3 %), and it takes the PID, the observable stores
// and all controller-row traffic off the Newton stage's critical path.  The hand-offs are 8 forces one way and 8
// tensions + 6 estimator scalars the other way, through LDS, with two workgroup barriers.  Same arithmetic as the
// one-wave kernels (shared device functions, -ffp-contract=off): bit-identical, tested.
// Registers: __launch_bounds__(128, 2) = at most 256 per wave, VGPR + AGPR together.
#ifndef CDPR_CTL_LOAD_DELAY
#define CDPR_CTL_LOAD_DELAY 8  // s_sleep units (64 cycles) the controller wave waits before it issues its 22 controller-row loads: the
#endif                         // estimator waves' two rows (the start of the serial chain) then meet a memory system that is not yet
                               // flooded by 2 048 x 27 row requests.  Interleaved A/Bs on three leases, us/step against 0: 4: 9.90-10.07
                               // vs 10.20-10.38; 6 / 8: 10.20-10.34 / 10.12-10.36 vs 10.50-10.72; 6 / 8 / 12: 10.27 / 10.26 / 10.20 vs
                               // 10.38 (medians); 16: no gain
#ifndef CDPR_SPLIT_PRIO
#define CDPR_SPLIT_PRIO 1  // 1 = estimator wave at s_setprio 3 (measured 10.89 vs 11.05 us/step), 2 = controller wave at 3 (11.7), 0 = none
#endif
#ifdef CDPR_STAMPS
#define CDPR_SPLIT_STAMP(i)                                                                                          \
  do {                                                                                                               \
    __builtin_amdgcn_sched_barrier(0);                                                                               \
    if (a.stamps && lane == 0) a.stamps[(size_t)blockIdx.x * 8 + (i)] = __builtin_amdgcn_s_memrealtime();           \
    __builtin_amdgcn_sched_barrier(0);                                                                               \
  } while (0)
#else
#define CDPR_SPLIT_STAMP(i) do { } while (0)
#endif
#if defined(CDPR_STAMPS) && defined(CDPR_STAMPS_ITER)  // stamps 4..6 = end of Newton iterations 1..3 instead of the controller's
#define CDPR_CTL_STAMP(i) do { } while (0)
#else
#define CDPR_CTL_STAMP(i) CDPR_SPLIT_STAMP(i)
#endif

// The estimator wave of a role-split workgroup (cdpr_split_kernel): platform rows -> measured lengths -> Newton-Raphson FK -> [forces from the controller wave] tension distribution
// -> tensions and estimator results back.  x_force / x_tension: v2f rows XS elements apart, x_est: float rows ES apart.
// RAISED: this wave at raised issue priority (the fast path's kernel: its chain is the launch's; the general kernels, whose controller
// wave is the longer one, pass their own choice)
template <int N, int XS, int ES, bool RAISED = true>
CDPR_DEV void split_estimator_wave(const StepArgs& a, float* geo, float gval, uint32_t lane, bool live, size_t st, uint32_t off, uint32_t woff,
                                   const float4& p0, const float4& p1, const float4& p3, const v2f* x_force, v2f* x_tension, float* x_est) {
  constexpr int NP = cable_pairs(N);
#if defined(CDPR_STAMPS) && defined(CDPR_STAMPS_CLOCK)  // the shader clock over this wave: s_memtime ticks between entry and stamp 3, into slot 7
  const unsigned long long clk0 = __builtin_amdgcn_s_memtime();
#endif
#if CDPR_SPLIT_PRIO == 1 || CDPR_SPLIT_PRIO == 4
  if (RAISED) __builtin_amdgcn_s_setprio(3);  // the estimator is the critical path: it wins the SIMD's issue arbitration
#elif CDPR_SPLIT_PRIO == 3 || CDPR_SPLIT_PRIO == 5
  if (RAISED) __builtin_amdgcn_s_setprio(2);
#endif
  const float4 p4 = load_slot(a.state, st, 4, off);
  if (lane < NP * kGeomFloatsPerPair) geo[lane] = gval;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  float fkx = p3.y, fky = p3.z, fkz = p3.w, fkqx = p4.x, fkqy = p4.y, fkqz = p4.z, fkqw = p4.w;
  v2f len[NP];
  {
    v2f jac[NP][6], l0[NP];  // only the measured lengths L* are kept of the true-state evaluation
    ik_pairs<N, false>(geo, p0.x, p0.y, p0.z, p0.w, p1.x, p1.y, p1.z, len, jac, l0);
  }
  float fk_res = 0.f;
  int fk_it = 0;
  v2f jest[NP][6];
  {
    v2f elen[NP], unused[NP];
    bool active = true;
    for (int it = 0; it < a.fk_iters; ++it) {
      ik_pairs<N, false>(geo, fkx, fky, fkz, fkqx, fkqy, fkqz, fkqw, elen, jest, unused);
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
#ifdef CDPR_STAMPS_ITER
      if (it < 3) {
        asm volatile("" ::"v"(fkqw));
        __builtin_amdgcn_sched_barrier(0);
        if (a.stamps && lane == 0) a.stamps[(size_t)blockIdx.x * 8 + 4 + it] = __builtin_amdgcn_s_memrealtime();
        __builtin_amdgcn_sched_barrier(0);
      }
#endif
    }
    ik_pairs<N, false>(geo, fkx, fky, fkz, fkqx, fkqy, fkqz, fkqw, elen, jest, unused);
    v2f rm = splat(0.f);
#pragma unroll
    for (int k = 0; k < NP; ++k) rm = max2(rm, abs2(len[k] - elen[k]));
    fk_res = fmaxf(rm.x, rm.y);
  }
  if (live) store_slot_aux<CDPR_SPLIT_PLAT_AUX>(a.state, st, 4, woff, make_float4(fkqx, fkqy, fkqz, fkqw));
  CDPR_SPLIT_STAMP(1);
  // the tension distribution's matrix and its factor need no forces: done while the controller wave may still be busy
  v2f td_l[6][3];
  float td_invd[6];
  normal_matrix_pk<NP, false>(jest, 0.f, td_l);
  chol_factor_pk(td_l, td_invd);
  __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): nothing of this wave's LDS traffic is pending
  __builtin_amdgcn_s_barrier();        // #1: the controller wave's forces are in x_force
  CDPR_SPLIT_STAMP(2);
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  v2f f[NP], df[NP], t_out[NP];
#pragma unroll
  for (int k = 0; k < NP; ++k) {
    f[k] = x_force[k * XS + lane];
    df[k] = f[k] - splat(a.td_mid);
  }
  int td_flag = 0;
  {
    float g[6];
    jt_times<NP>(jest, df, g);
    chol_apply_pk(td_l, td_invd, g);
#pragma unroll
    for (int k = 0; k < NP; ++k) {
      v2f t = splat(a.td_mid);
#pragma unroll
      for (int c = 0; c < 6; ++c) t = fma2(g[c], jest[k][c], t);
      const v2f tc = max2(min2(t, splat(a.td_max)), splat(a.td_min));
      td_flag |= (tc.x != t.x) ? 1 : 0;
      if (2 * k + 1 < N) td_flag |= (tc.y != t.y) ? 1 : 0;
      t_out[k] = tc;
    }
  }
#pragma unroll
  for (int k = 0; k < NP; ++k) x_tension[k * XS + lane] = t_out[k];
  x_est[0 * ES + lane] = fkx;
  x_est[1 * ES + lane] = fky;
  x_est[2 * ES + lane] = fkz;
  x_est[3 * ES + lane] = fk_res;
  x_est[4 * ES + lane] = (float)fk_it;
  x_est[5 * ES + lane] = (float)td_flag;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_s_waitcnt(0xC07F);
  CDPR_SPLIT_STAMP(3);
#if defined(CDPR_STAMPS) && defined(CDPR_STAMPS_CLOCK)
  if (a.stamps && lane == 0) a.stamps[(size_t)blockIdx.x * 8 + 7] = __builtin_amdgcn_s_memtime() - clk0;
#endif
  __builtin_amdgcn_s_barrier();  // #2: tensions and estimator results are out
}

// PR = true: per-robot handles (StepArgs::meta, see cdpr_step_kernel.hpp): the controller wave takes mode and Pid call
// count per lane; the estimator wave is the same.
template <int N, bool PR = false>
__global__ __launch_bounds__(128, 2) void cdpr_split_kernel(const StepArgs a) {
  constexpr int NP = cable_pairs(N);
  constexpr int P = plat_slots(true);
  constexpr int G = joint_groups(N);
  constexpr int NH = (NP + 1) / 2;
  __shared__ __attribute__((aligned(16))) float lds[2][NP * kGeomFloatsPerPair];  // one geometry copy per wave: no barrier before first use
  __shared__ v2f x_force[NP][64];      // controller -> estimator: raw per-cable forces
  __shared__ v2f x_tension[NP][64];    // estimator -> controller: distributed tensions (before the SetForce limits)
  __shared__ float x_est[6][64];       // estimator -> controller: fk x y z, residual, iterations, infeasible flag

  // which of the two waves estimates: swapped from workgroup to workgroup (bits of the workgroup index chosen by the
  // host, StepArgs::split_swap) so that the two waves a SIMD hosts tend to be one of each role
  const uint32_t swap = __builtin_popcount(blockIdx.x & a.split_swap) & 1u;
  const uint32_t wave = (threadIdx.x >> 6) ^ swap, lane = threadIdx.x & 63u;
  const uint32_t r = blockIdx.x * 64u + lane;
  const uint32_t rr = (r < a.batch) ? r : (a.batch - 1u);  // tail lanes shadow the last robot, stores are masked
  const bool live = r < a.batch;
  const size_t st = a.stride;
  const uint32_t off = rr * 16u, woff = r * 16u;
  float* const geo = lds[wave];

  if (wave == 0) CDPR_SPLIT_STAMP(0);
#ifdef CDPR_STAMPS
  if (a.stamps && lane == 0)  // where this wave runs: HW_ID (wave, SIMD, CU, SH, SE) | XCC_ID << 16, per PHYSICAL wave of the workgroup
    reinterpret_cast<uint32_t*>(&a.stamps[(size_t)blockIdx.x * 8 + 7])[threadIdx.x >> 6] =
        (__builtin_amdgcn_s_getreg((16 - 1) << 11 | 4) & 0xffffu) | ((__builtin_amdgcn_s_getreg((4 - 1) << 11 | 20) & 0xfu) << 16);
#endif
  const float gval = (lane < NP * kGeomFloatsPerPair) ? a.geom[lane] : 0.f;
  const float4 p0 = load_slot(a.state, st, 0, off), p1 = load_slot(a.state, st, 1, off), p2 = load_slot(a.state, st, 2, off),
               p3 = load_slot(a.state, st, 3, off);
  if (wave == 0) {
    split_estimator_wave<N, 64, 64>(a, geo, gval, lane, live, st, off, woff, p0, p1, p3, &x_force[0][0], &x_tension[0][0], &x_est[0][0]);
    return;
  }
  // ---------------------------------------------------------------------------------------------------- controller wave
#if CDPR_SPLIT_PRIO == 2 || CDPR_SPLIT_PRIO == 5
  __builtin_amdgcn_s_setprio(3);  // (5: from its first instruction to the force hand-off)
#endif
#if CDPR_CTL_LOAD_DELAY > 0
  __builtin_amdgcn_s_sleep(CDPR_CTL_LOAD_DELAY);  // the estimator waves' two rows go first through the memory system
#endif
  constexpr int kCtrl = 5 * NP + NH;
  float4 wraw[NP][5], hraw[NH];
#pragma unroll
  for (int k = 0; k < NP; ++k) {
#pragma unroll
    for (int m = 0; m < 5; ++m) wraw[k][m] = load_slot(a.state, st, P + 5 * k + m, off);
  }
#pragma unroll
  for (int g = 0; g < NH; ++g) hraw[g] = load_slot(a.state, st, P + 5 * NP + g, off);
  (void)kCtrl;
  v2f desired[NP];
#pragma unroll
  for (int k = 0; k < NP; ++k) desired[k] = splat(0.f);
  const float* cp = a.cmd + (size_t)rr * N;  // never null: before the first Joy the latched buffer holds zeros
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
  if (lane < NP * kGeomFloatsPerPair) geo[lane] = gval;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

  Platform s;
  s.px = p0.x; s.py = p0.y; s.pz = p0.z; s.qx = p0.w;
  s.qy = p1.x; s.qz = p1.y; s.qw = p1.z; s.vx = p1.w;
  s.vy = p2.x; s.vz = p2.y; s.wx = p2.z; s.wy = p2.w;
  s.wz = p3.x;

  // ---- IK on the state at t_k; the structure matrix stays alive for the world step (this wave runs no Newton stage)
  v2f len[NP], q[NP], qd[NP], jac[NP][6];
  {
    v2f l0[NP];
    ik_pairs<N, true>(geo, s.px, s.py, s.pz, s.qx, s.qy, s.qz, s.qw, len, jac, l0);
#pragma unroll
    for (int k = 0; k < NP; ++k) {
      q[k] = l0[k] - len[k];
      qd[k] = -fma2(s.wz, jac[k][5], fma2(s.wy, jac[k][4], fma2(s.wx, jac[k][3],
                    fma2(s.vz, jac[k][2], fma2(s.vy, jac[k][1], splat(s.vx) * jac[k][0])))));
    }
  }
  const bool publish = (a.publish_mask & 1ull) != 0ull;
  if (publish && live) {
    store_slot(a.obs, st, 0, woff, make_float4(s.px, s.py, s.pz, s.qx));
    store_slot(a.obs, st, 1, woff, make_float4(s.qy, s.qz, s.qw, s.vx));
    store_slot(a.obs, st, 2, woff, make_float4(s.vy, s.vz, s.wx, s.wy));
#pragma unroll
    for (int g = 0; g < G; ++g) {
      const int k0 = 2 * g, k1 = (2 * g + 1 < NP) ? 2 * g + 1 : 2 * g;
      const bool has = (2 * g + 1 < NP);
      store_slot(a.obs, st, 4 + g, woff, make_float4(q[k0].x, q[k0].y, has ? q[k1].x : 0.f, has ? q[k1].y : 0.f));
      store_slot(a.obs, st, 4 + G + g, woff, make_float4(qd[k0].x, qd[k0].y, has ? qd[k1].x : 0.f, has ? qd[k1].y : 0.f));
    }
  }

  // ---- per-cable force (PLG.cpp:222-228 -> JFC.cpp:59-96 -> Pid.cpp:122-191)
  v2f f[NP], e_new[NP], ierr[NP];
#pragma unroll
  for (int k = 0; k < NP; ++k) {
    f[k] = splat(0.f);
    e_new[k] = splat(0.f);
    ierr[k] = (k & 1) ? (v2f){hraw[k / 2].z, hraw[k / 2].w} : (v2f){hraw[k / 2].x, hraw[k / 2].y};
  }
  float dbg_p = 0.f, dbg_i = 0.f, dbg_d = 0.f;
  bool dbg_wrote = false;
  const bool first_world = (a.flags & kFlagFirstWorldStep) != 0u;
  uint32_t meta = 0u;
  if (PR) meta = a.meta[rr];
  const int calls = PR ? (int)(meta >> kMetaCallShift) : a.pid_calls;
  if (PR && live && !first_world) a.meta[r] = (uint8_t)((meta & kMetaModeMask) | ((uint32_t)min(calls + 1, (int)kMetaCallMax) << kMetaCallShift));
  const bool force_mode = !PR && (a.flags & kFlagForceMode) != 0u;  // UpdateMode::Force on a uniform handle (JFC.cpp:67-70)
  if (force_mode && !first_world) {
#pragma unroll
    for (int k = 0; k < NP; ++k) f[k] = desired[k];
  } else if (!first_world && (PR || calls != 0)) {
    v2f win[NP][kWin];
#pragma unroll
    for (int k = 0; k < NP; ++k) {
#pragma unroll
      for (int m = 0; m < 5; ++m) {
        win[k][2 * m] = (v2f){wraw[k][m].x, wraw[k][m].y};
        win[k][2 * m + 1] = (v2f){wraw[k][m].z, wraw[k][m].w};
      }
    }
    const bool actual_is_vel = PR ? ((meta & kMetaModeMask) == kMetaVelocity) : ((a.flags & kFlagActualIsVelocity) != 0u);
    v2f actual[NP];
#pragma unroll
    for (int k = 0; k < NP; ++k) actual[k] = actual_is_vel ? qd[k] : q[k];
    const int ring_slot = a.ring_slot;
    const bool is_force = PR && (meta & kMetaModeMask) == kMetaForce;
#if CDPR_SPLIT_PRIO == 3 || CDPR_SPLIT_PRIO == 4
    __builtin_amdgcn_s_setprio(3);  // (experiment) the forces are what the estimator wave will wait for
#endif
    pid_pairs<NP, PR>(a, calls, actual_is_vel, desired, actual, win, ierr, f, e_new, dbg_p, dbg_i, dbg_d, is_force);
    dbg_wrote = (!PR || calls != 0) && !is_force;
    if (live) {
#pragma unroll
      for (int m = 0; m < 5; ++m) {
        if (m == (ring_slot >> 1)) {
#pragma unroll
          for (int k = 0; k < NP; ++k) CDPR_STORE_STATE(a.state, st, P + 5 * k + m, woff, ring_row(win[k], m, e_new[k], ring_slot));
        }
      }
#pragma unroll
      for (int g = 0; g < NH; ++g) {
        const int k1 = (2 * g + 1 < NP) ? 2 * g + 1 : 2 * g;
        CDPR_STORE_STATE(a.state, st, P + 5 * NP + g, woff, make_float4(ierr[2 * g].x, ierr[2 * g].y, ierr[k1].x, ierr[k1].y));
      }
    }
  }
#pragma unroll
  for (int k = 0; k < NP; ++k) x_force[k][lane] = f[k];
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): the forces are in LDS (vector memory operations stay in flight)
  CDPR_CTL_STAMP(4);
#if CDPR_SPLIT_PRIO >= 3
  __builtin_amdgcn_s_setprio(0);
#endif
  __builtin_amdgcn_s_barrier();        // #1
  __builtin_amdgcn_s_barrier();        // #2: the estimator wave has finished the tension distribution
  CDPR_CTL_STAMP(5);
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  v2f applied[NP];
#pragma unroll
  for (int k = 0; k < NP; ++k) applied[k] = x_tension[k][lane];
  const float fkx = x_est[0][lane], fky = x_est[1][lane], fkz = x_est[2][lane], fk_res = x_est[3][lane], fk_it = x_est[4][lane],
              td_flag = x_est[5][lane];
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
  if (a.dbg && live) {  // `pid` topic, cable 0 only (PLG.cpp:223-227; Pid.cpp:139-142,158-168)
    float* d = a.dbg + (size_t)r * 9;
    if (dbg_wrote) {
      d[0] = dbg_p;
      d[1] = dbg_i;
      d[2] = dbg_d;
      d[3] = desired[0].x;
    }
    d[4] = applied[0].x;
  }
  if (publish && live) {
    store_slot(a.obs, st, 3, woff, make_float4(s.wz, fk_res, fk_it, a.travel_on ? pack_flags((int)td_flag, travel_mask<N>(a, q)) : td_flag));
#pragma unroll
    for (int g = 0; g < G; ++g) {
      const int k0 = 2 * g, k1 = (2 * g + 1 < NP) ? 2 * g + 1 : 2 * g;
      const bool has = (2 * g + 1 < NP);
      store_slot(a.obs, st, 4 + 2 * G + g, woff,
                 make_float4(applied[k0].x, applied[k0].y, has ? applied[k1].x : 0.f, has ? applied[k1].y : 0.f));
    }
  }
  // ---- world step to t_{k+1}: wrench = -J^T (applied - d qdot) + m g
  {
    v2f tens[NP];
#pragma unroll
    for (int k = 0; k < NP; ++k) {
      tens[k] = fma2(-a.damping, qd[k], applied[k]);
      if (a.unilateral) tens[k] = max2(tens[k], splat(0.f));
    }
    float w[6];
    jt_times<NP>(jac, tens, w);
    w[0] = a.fgx - w[0];
    w[1] = a.fgy - w[1];
    w[2] = a.fgz - w[2];
    w[3] = -w[3];
    w[4] = -w[4];
    w[5] = -w[5];
    integrate(a, s, w);
  }
  if (live) {
    store_slot_aux<CDPR_SPLIT_PLAT_AUX>(a.state, st, 0, woff, make_float4(s.px, s.py, s.pz, s.qx));
    store_slot_aux<CDPR_SPLIT_PLAT_AUX>(a.state, st, 1, woff, make_float4(s.qy, s.qz, s.qw, s.vx));
    store_slot_aux<CDPR_SPLIT_PLAT_AUX>(a.state, st, 2, woff, make_float4(s.vy, s.vz, s.wx, s.wy));
    store_slot_aux<CDPR_SPLIT_PLAT_AUX>(a.state, st, 3, woff, make_float4(s.wz, fkx, fky, fkz));
  }
  CDPR_CTL_STAMP(6);
}

}  // namespace cdpr
