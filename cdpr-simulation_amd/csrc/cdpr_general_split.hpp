// cdpr_general_split.hpp — the general controller path's one-step launch with TWO waves per 64 robots, split by role as in
// cdpr_split_kernel (cdpr_onestep_kernel.hpp): FK + TD handles, batches up to one workgroup per two SIMDs.
//
//   estimator wave    split_estimator_wave, unchanged: platform rows -> measured lengths -> Newton-Raphson FK -> [forces in]
//                     tension distribution -> tensions and estimator results out
//   controller wave   tables, platform rows, Joy -> Pid selection -> record slots by LDS-DMA -> IK -> early observables ->
//                     gen_controller (cdpr_general_step.hpp) -> forces out -> [tensions in] SetForce limits -> observables ->
//                     world step (optional physics by run-time flags) -> state
//
// The one-wave kernel (cdpr_gen_step_kernel<..., SINGLE>) carries IK, Newton, controller, tension distribution and world
// step on ONE wave: ~3 900 vector instructions of a serial chain.  Here the Newton stage and the tension distribution
// (~2 200 of them) run beside the controller on another SIMD.  Compiled for ONE workgroup per pair of SIMDs
// (__launch_bounds__(128, 1): each wave may use the whole register file, as the one-wave kernel does), so it serves
// batches up to 2 workgroups per CU; the engine uses the one-wave kernel beyond.  Measured and not kept: the same kernel
// compiled for two waves per SIMD (__launch_bounds__(128, 2): 256 registers, 300 B of scratch per lane in the controller
// wave) runs 65 536 x 8 in 30 - 47 us against the one-wave kernel's 21.4, and 28.7 us with the scratch cut to 48 B (cables
// in groups of two, structure matrix rebuilt for the world step); this build at 65 536 (two rounds of 512 workgroups) in
// 22.5.  At 65 536 the launch moves 87 MB - 13.6 us at the 6.4 TB/s a copy reaches - so there is little to win there.
// Same device functions, same arithmetic order as the one-wave kernel: bit-identical (tested).
#pragma once
#include "cdpr_general_step.hpp"
#include "cdpr_onestep_kernel.hpp"

namespace cdpr {

template <int N, int NBMAX>
__global__ __launch_bounds__(128, 1) void cdpr_gen_split_kernel(const StepArgs a, const GenCtl g) {
  constexpr int NP = cable_pairs(N);
  constexpr int G = joint_groups(N);
  constexpr int NV = gen_nv(NBMAX);
  constexpr int NBP = gen_nbp(NBMAX);
  constexpr int LP = (N + 3) / 4;
  __shared__ __attribute__((aligned(16))) float lds[2][NP * kGeomFloatsPerPair];  // one geometry copy per wave
  __shared__ __attribute__((aligned(16))) float wrot[2][NBMAX][NBP];
  __shared__ float4 stage[N][NV + 1][64];
  __shared__ float4 hold_slots[LP][64];
  __shared__ uint32_t q_count[kGenQueueWords];
  __shared__ float4 ptab[2][kGenPidFloats / 4];
  // the hand-off buffers live in the staged record slots, which are done with when the controller returns (before barrier
  // #1; the estimator touches them after it): 5.5 KiB that decide whether four workgroups fit a CU's LDS
  static_assert(sizeof(stage) >= (2 * NP * 64) * sizeof(v2f) + 6 * 64 * sizeof(float), "hand-off buffers fit the staging area");
  v2f (*const x_force)[64] = reinterpret_cast<v2f(*)[64]>(&stage[0][0][0]);            // controller -> estimator: raw per-cable forces
  v2f (*const x_tension)[64] = x_force + NP;                                            // estimator -> controller: distributed tensions
  float (*const x_est)[64] = reinterpret_cast<float(*)[64]>(x_tension + NP);            // estimator -> controller: fk x y z, residual, iterations, flag

  const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
  const uint32_t r = blockIdx.x * 64u + lane;
  const uint32_t units = a.batch;
  const uint32_t rr = (r < units) ? r : (units - 1u);  // tail lanes shadow the last robot, stores are masked
  const bool live = r < units;
  const size_t st = a.stride;
  const uint32_t off = rr * 16u, woff = r * 16u;
  float* const geo = lds[wave];

  if (wave == 0) CDPR_SPLIT_STAMP(0);
  const float gval = (lane < NP * kGeomFloatsPerPair) ? a.geom[lane] : 0.f;
  if (wave == 0) {
    const float4 p0 = load_slot(a.state, st, 0, off), p1 = load_slot(a.state, st, 1, off), p3 = load_slot(a.state, st, 3, off);
    split_estimator_wave<N, 64, 64>(a, geo, gval, lane, live, st, off, woff, p0, p1, p3, &x_force[0][0], &x_tension[0][0], &x_est[0][0]);
    return;
  }
  // ---------------------------------------------------------------------------------------------------- controller wave
  GenLayout L;
  L.n = g.lay.n, L.nb = g.lay.nb, L.ncas = g.lay.ncas;
  GenBuf RB = gen_buffer(g.rec, g.rstride, g.rec_bytes, L);
  const uint32_t col = rr;
  // every load of the prologue is issued before anything waits (cdpr_gen_step_kernel)
  constexpr uint32_t kW4 = 2u * NBMAX * NBP / 4u;
  constexpr int kWPass = (int)((kW4 + 63u) / 64u);
  float4 wv[kWPass];
#pragma unroll
  for (int j = 0; j < kWPass; ++j) wv[j] = reinterpret_cast<const float4*>(g.wtab)[min(lane + 64u * j, kW4 - 1u)];
  const float pv = g.ptab[min(lane, 2u * kGenPidFloats - 1u)];
  const float4 p0 = load_slot(a.state, st, 0, off), p1 = load_slot(a.state, st, 1, off), p2 = load_slot(a.state, st, 2, off),
               p3 = load_slot(a.state, st, 3, off);
  const int mode = g.mode_arr ? (int)g.mode_arr[rr] : g.mode;
  const float* cmd_src = (mode == 2) ? g.vel_cmd : (mode == 1) ? g.pos_cmd : g.frc_cmd;
  float target[N];
#pragma unroll
  for (int i = 0; i < N; ++i) target[i] = 0.f;
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
  if (lane < NP * kGeomFloatsPerPair) geo[lane] = gval;
#pragma unroll
  for (int j = 0; j < kWPass; ++j)
    if (lane + 64u * j < kW4) reinterpret_cast<float4*>(&wrot[0][0][0])[lane + 64u * j] = wv[j];
  if (lane == 0) q_count[0] = 0u;
  if (lane < 2 * kGenPidFloats) (&ptab[0][0].x)[lane] = pv;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

  Platform s;
  s.px = p0.x; s.py = p0.y; s.pz = p0.z; s.qx = p0.w;
  s.qy = p1.x; s.qz = p1.y; s.qw = p1.z; s.vx = p1.w;
  s.vy = p2.x; s.vz = p2.y; s.wx = p2.z; s.wy = p2.w;
  s.wz = p3.x;

  const int now = g.now_step;
  const bool first_world = (a.flags & kFlagFirstWorldStep) != 0u;
  const bool run_ctl = !first_world;
  // ---- which Pid serves each cable this step (JFC.cpp:67-89), and its slots on their way to LDS
  int sel[N];
#pragma unroll
  for (int i = 0; i < N; ++i) {
    sel[i] = (mode == 2 && fabsf(target[i]) > g.eps) ? 1 : 0;
    asm volatile("" : "+v"(sel[i]));
  }
  // (no hot rows here - GenHot: the engine keeps them to the handles the lean kernel steps)
  if (run_ctl) {
    float keep = (s.px + s.qy) + (s.vy + s.wz);
#pragma unroll
    for (int i = 0; i < N; ++i) keep += target[i];
    bool holds = false;  // some cable of this lane is in the hold branch (JFC.cpp:78-82)
#pragma unroll
    for (int i = 0; i < N; ++i) holds = holds || (mode == 2 && sel[i] == 0);
    gen_stage_records<N, NBMAX>(RB, L, col, sel, &stage[0][0][0], &hold_slots[0][0], keep, __builtin_amdgcn_ballot_w64(holds) != 0ull);
  }

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
  float4* const obs = a.obs;
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

  // ---- per-cable force: the general controller
  float force[N];
#pragma unroll
  for (int i = 0; i < N; ++i) force[i] = 0.f;
  GenDbg dbg{0.f, 0.f, 0.f, 0.f, false, false};
  if (run_ctl) {
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): the DMA has landed
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    GenCtlConst cc;
    cc.pcas_max = g.pcas_max, cc.dcas_max = g.dcas_max, cc.dt = g.dt, cc.inv_dt = a.inv_dt;
    cc.nm0 = now % max(g.nbuf0, 1), cc.nm1 = now % max(g.nbuf1, 1), cc.nbuf0 = g.nbuf0, cc.simple_ok = g.simple_ok != 0;
  cc.force_rows = nullptr;
#ifdef CDPR_STAMPS
    cc.stamps = nullptr;
#endif
    gen_controller<N, NBMAX>(cc, RB, L, lane, live, col, blockIdx.x * 64u, units, mode, now, target, sel, q, qd, &stage[0][0][0], &hold_slots[0][0],
                             &wrot[0][0][0], ptab, q_count, force, dbg);
  }
#pragma unroll
  for (int k = 0; k < NP; ++k) x_force[k][lane] = (v2f){force[2 * k], (2 * k + 1 < N) ? force[2 * k + 1] : 0.f};
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): the forces are in LDS (vector memory operations stay in flight)
  CDPR_CTL_STAMP(4);
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
  if (a.dbg && live) {  // `pid` topic, cable 0 only: stale entries stay (Pid.cpp:139-142,158-168)
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
    store_slot(obs, st, 3, woff, make_float4(s.wz, fk_res, fk_it, pack_flags((int)td_flag, travel_mask<N>(a, q))));
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
    float w[6];
    jt_times<NP>(jac, tens, w);
    w[0] = a.fgx - w[0];
    w[1] = a.fgy - w[1];
    w[2] = a.fgz - w[2];
    w[3] = -w[3];
    w[4] = -w[4];
    w[5] = -w[5];
    if (a.ph_lumped)
      integrate_lumped_velocity<N>(a, geo, s, jac, len, w);
    else
      integrate_velocity(a, s, w);
    if (a.travel_stop) apply_travel_stop<N>(a, s, q, jac);
    integrate_pose(a, s);
  }
  if (live) {
    CDPR_STORE_STATE(a.state, st, 0, woff, make_float4(s.px, s.py, s.pz, s.qx));
    CDPR_STORE_STATE(a.state, st, 1, woff, make_float4(s.qy, s.qz, s.qw, s.vx));
    CDPR_STORE_STATE(a.state, st, 2, woff, make_float4(s.vy, s.vz, s.wx, s.wy));
    CDPR_STORE_STATE(a.state, st, 3, woff, make_float4(s.wz, fkx, fky, fkz));
  }
  CDPR_CTL_STAMP(6);
}

// ---------------------------------------------------------------------------------------------------------------------
// The LEAN role-split kernel: batches beyond two workgroups per CU (round 5).
//
// cdpr_gen_split_kernel needs the whole register file per wave (256 + 77) because one function carries every path of the
// general controller: the ring rotation, the fit queue and the fp64 fit live next to the consecutive-call branch that runs
// on all steps but those after a mode change or a Pid switch.  Compiled for two waves per SIMD it spills on the common path
// too (round 4: 48 - 300 B of scratch, 28.7 - 47 us at 65 536 x 8).  Here the controller wave INLINES only the branch for
// waves whose cables are all on a uniform or a filling window (gen_controller<..., STEADY_ONLY>, no scratch) and hands
// every other wave to gen_lean_cold_tail: a function with its own register allocation that FINISHES the controller wave's
// work - general controller, hand-offs with the estimator wave, observables, world step, state - and ends the program.
// It never returns, so nothing of the caller is saved around it, and it takes its few per-lane inputs in argument
// registers (the platform state), joint positions, rates and the structure matrix through the lane's private memory, the
// Joy targets and modes through LDS, and reads the launch's arguments from the kernel-argument segment itself.  (First version of this round: an ordinary
// call, gen_controller_cold.  Its 75 argument words went through the stack, the caller saved 46 registers around it and
// the callee's entry waits for vmcnt(0): 3.6 us on the way in and 1.4 - 2.7 us on the way out of a 5.6 us loop -
// profiles/r05_cold_probe_before.txt.)  Two waves per SIMD fit, so at 65 536 robots a SIMD hosts an estimator wave and a
// controller wave side by side as the fast path's cdpr_split_kernel does.  No optional physics here (those handles keep
// the other kernels).  Same device functions, same arithmetic order: bit-identical to the other general kernels (tested).
// Measured and not kept: the steady-only kernel followed by the one-wave kernel over the blocks it flagged (two launches
// per step: 18.1 us steady, 46.7 switching - every block of a switching batch holds a switching robot).
template <int N>
struct LeanShared {
  static constexpr int NBMAX = 11;
  static constexpr int NP = cable_pairs(N);
  static constexpr int NV = gen_nv(NBMAX);
  static constexpr int NBP = gen_nbp(NBMAX);
  static constexpr int LP = (N + 3) / 4;
  float lds[2][NP * kGeomFloatsPerPair] __attribute__((aligned(16)));  // one geometry copy per wave
  float wrot[2][NBMAX][NBP] __attribute__((aligned(16)));
  float4 stage[N][NV + 1][64];
  float4 hold_slots[LP][64];
  float4 ptab[2][kGenPidFloats / 4];
  uint32_t q_count[kGenQueueWords];
  float4 dump[64];  // where the lanes a predicated LDS store does not concern write (no divergent branch in the controller wave's prologue)
  float4 tgt[2][64];  // the Joy targets of the wave's robots, for the cold tail (31 argument registers carry the state, q and qd)
  int mode_row[64];   // and their modes (read from memory again, the tail's entry waits for every store in flight: vmcnt is one counter)
  // the hand-off buffers live in the staged record slots, which are done with when the controller returns
  static_assert(sizeof(float4) * N * (NV + 1) * 64 >= (2 * NP * 64) * sizeof(v2f) + 6 * 64 * sizeof(float), "hand-off buffers fit the staging area");
  CDPR_DEV v2f (*x_force())[64] { return reinterpret_cast<v2f(*)[64]>(&stage[0][0][0]); }
  CDPR_DEV v2f (*x_tension())[64] { return x_force() + NP; }
  CDPR_DEV float (*x_est())[64] { return reinterpret_cast<float(*)[64]>(x_tension() + NP); }
};

// What the controller wave does once the per-cable forces are known: forces to the estimator wave, observables, the
// distributed tensions back, SetForce limits, world step, state.  Shared by the kernel and by gen_lean_cold_tail.
// REBUILD: the structure matrix is not in registers (the cold tail): read back from jac_src between the two barriers, while
// the estimator wave runs the tension distribution.
template <int N, bool REBUILD>
CDPR_DEV void lean_controller_epilogue(const StepArgs& a, LeanShared<N>& sm, float* geo, uint32_t lane, uint32_t r, bool live, Platform s, const v2f (&q)[cable_pairs(N)],
                                       const v2f (&qd)[cable_pairs(N)], v2f (&jac)[cable_pairs(N)][6], const float (&force)[N], const GenDbg dbg,
                                       const float4* jac_src = nullptr) {
  constexpr int NP = cable_pairs(N);
  constexpr int G = joint_groups(N);
  const size_t st = a.stride;
  const uint32_t woff = r * 16u;
  v2f (*const x_force)[64] = sm.x_force();
  v2f (*const x_tension)[64] = sm.x_tension();
  float (*const x_est)[64] = sm.x_est();
#pragma unroll
  for (int k = 0; k < NP; ++k) x_force[k][lane] = (v2f){force[2 * k], (2 * k + 1 < N) ? force[2 * k + 1] : 0.f};
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): the forces are in LDS
  CDPR_CTL_STAMP(4);
  __builtin_amdgcn_s_barrier();        // #1
  const bool publish = (a.publish_mask & 1ull) != 0ull;
  float4* const obs = a.obs;
  // the observables that are final after the IK (PLG.cpp:248-280) go out HERE, while this wave waits for the tensions: in
  // front of the controller they sat between the DMA and the vmcnt(0) that waits for it, and the wave paid their
  // completion (2 - 3 us at 65 536 robots, profiles/r05_lean_timeline.txt) on the path the estimator wave waits for
  // (every store of this function is predicated through its offset, no divergent branch: with one, LLVM structurizes the
  //  kernel's "inlined branch or cold tail" decision and everything this function reads stays live across the tail's call)
  if (publish) {
    store_slot_if(live, obs, st, 0, woff, make_float4(s.px, s.py, s.pz, s.qx));
    store_slot_if(live, obs, st, 1, woff, make_float4(s.qy, s.qz, s.qw, s.vx));
    store_slot_if(live, obs, st, 2, woff, make_float4(s.vy, s.vz, s.wx, s.wy));
#pragma unroll
    for (int gI = 0; gI < G; ++gI) {
      const int k0 = 2 * gI, k1 = (2 * gI + 1 < NP) ? 2 * gI + 1 : 2 * gI;
      const bool has = (2 * gI + 1 < NP);
      store_slot_if(live, obs, st, 4 + gI, woff, make_float4(q[k0].x, q[k0].y, has ? q[k1].x : 0.f, has ? q[k1].y : 0.f));
      store_slot_if(live, obs, st, 4 + G + gI, woff, make_float4(qd[k0].x, qd[k0].y, has ? qd[k1].x : 0.f, has ? qd[k1].y : 0.f));
    }
  }
  if constexpr (REBUILD) {  // the cold tail: the structure matrix comes back from the kernel's spill buffer (see gen_lean_cold_tail)
#pragma unroll
    for (int k = 0; k < NP; ++k)
#pragma unroll
      for (int c = 0; c < 6; c += 2) {
        const float4 v = jac_src[(k * 6 + c) / 2];
        jac[k][c] = (v2f){v.x, v.y};
        jac[k][c + 1] = (v2f){v.z, v.w};
      }
  }
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
  if (a.dbg) {  // `pid` topic, cable 0 only: stale entries stay (Pid.cpp:139-142,158-168)
    const auto drs = __builtin_amdgcn_make_buffer_rsrc(a.dbg, 0, (int)(a.batch * 9u * sizeof(float)), 0x00020000);
    const uint32_t d0 = r * 36u;
    auto put = [&](bool on, uint32_t k, float v) { __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), drs, on ? d0 + 4u * k : 0xFFFFFFFFu, 0, 0); };
    put(live && dbg.pi, 0, dbg.p);
    put(live && dbg.pi, 1, dbg.i);
    put(live && dbg.pi, 3, dbg.des);
    put(live && dbg.dw, 2, dbg.d);
    put(live, 4, applied[0].x);
  }
  if (publish) {  // the rest of the observables
    store_slot_if(live, obs, st, 3, woff, make_float4(s.wz, fk_res, fk_it, pack_flags((int)td_flag, travel_mask<N>(a, q))));
#pragma unroll
    for (int gI = 0; gI < G; ++gI) {
      const int k0 = 2 * gI, k1 = (2 * gI + 1 < NP) ? 2 * gI + 1 : 2 * gI;
      const bool has = (2 * gI + 1 < NP);
      store_slot_if(live, obs, st, 4 + 2 * G + gI, woff,
                    make_float4(applied[k0].x, applied[k0].y, has ? applied[k1].x : 0.f, has ? applied[k1].y : 0.f));
    }
  }
  // ---- world step to t_{k+1}: wrench = -J^T (applied - d qdot) + m g (no optional physics here: those handles never get this kernel)
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
    integrate_velocity(a, s, w);
    integrate_pose(a, s);
  }
  store_slot_if(live, a.state, st, 0, woff, make_float4(s.px, s.py, s.pz, s.qx));
  store_slot_if(live, a.state, st, 1, woff, make_float4(s.qy, s.qz, s.qw, s.vx));
  store_slot_if(live, a.state, st, 2, woff, make_float4(s.vy, s.vz, s.wx, s.wy));
  store_slot_if(live, a.state, st, 3, woff, make_float4(s.wz, fkx, fky, fkz));
  CDPR_CTL_STAMP(6);
}

// Joy targets of one robot for the general controller (PLG.cpp:206-219 latched them), by the robot's mode.
template <int N>
CDPR_DEV void lean_load_targets(const GenCtl& g, uint32_t rr, int mode, float (&target)[N]) {
  const float* cmd_src = (mode == 2) ? g.vel_cmd : (mode == 1) ? g.pos_cmd : g.frc_cmd;
#pragma unroll
  for (int i = 0; i < N; ++i) target[i] = 0.f;
  if (g.mode_arr) {  // per-robot modes: the buffer differs from lane to lane; a lane without one reads the Pid table and drops it
    const float* cp = cmd_src ? cmd_src + (size_t)rr * N : g.ptab;
#pragma unroll
    for (int i = 0; i < N; ++i) {
      const float v = cp[i];
      target[i] = cmd_src ? v : 0.f;
    }
  } else if (const float* const uni_src = (g.mode == 2) ? g.vel_cmd : (g.mode == 1) ? g.pos_cmd : g.frc_cmd) {  // (from the scalar mode: a uniform branch)
    const float* cp = uni_src + (size_t)rr * N;
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

typedef float lean_f4 __attribute__((ext_vector_type(4)));

// The kernel's two arguments as they lie in the kernel-argument segment (cdpr_gen_lean_kernel(StepArgs, GenCtl)).
struct LeanKernArgs {
  StepArgs a;
  GenCtl g;
};

// The controller wave's work from the general controller on, for the waves the inlined branch does not serve.  Never returns.
// (noreturn only helps the caller if NO divergent branch precedes the call in the kernel: LLVM routes an unreachable that sits
//  behind any divergent branch into the kernel's common exit ("divergent unreachable"), the call then looks like one that
//  comes back and the kernel saves what the rest of it reads - 46 registers - around it.  Hence the controller wave's
//  prologue without a single divergent branch: predicated LDS stores go to dump words, loads to clamped addresses.)
#define CDPR_LEAN_TAIL_ATTR __device__ __attribute__((noinline, noreturn))
template <int N>
CDPR_LEAN_TAIL_ATTR void gen_lean_cold_tail(lean_f4 s0, lean_f4 s1, lean_f4 s2, float s_wz, const float4* spill, uint32_t sm_and_ka_hi, uint32_t ka_lo) {
  constexpr int NBMAX = 11;
  constexpr int NP = cable_pairs(N);
  // (the kernel hands its kernel-argument segment over: __builtin_amdgcn_kernarg_segment_ptr() is null inside a function.  The
  //  cast to a generic pointer is undone by LLVM's address-space inference: the fields are read with scalar loads from the
  //  constant address space where they are used)
  const __attribute__((address_space(4))) void* const kap = (const __attribute__((address_space(4))) void*)(((uint64_t)(uni(sm_and_ka_hi) >> 16) << 32) | uni(ka_lo));
  const uint32_t sm_addr = sm_and_ka_hi & 0xFFFFu;  // (31 argument registers: the LDS address and the upper 16 bits of a 48-bit address share one)
  const LeanKernArgs& ka = *(const LeanKernArgs*)(const void*)kap;
  // (copies: read through references the fields are loaded again at every use - eight scalar loads of velocityEpsilon, each
  //  with its own wait, in front of the controller)
  const StepArgs a = ka.a;
  const GenCtl g = ka.g;
  LeanShared<N>& sm = *lds_pointer<LeanShared<N>>(uni(sm_addr));
  const uint32_t lane = threadIdx.x & 63u;
#if defined(CDPR_STAMPS) && defined(CDPR_STAMPS_COLD)
  if (a.stamps && lane == 0) a.stamps[((size_t)gridDim.x + blockIdx.x) * 8 + 5] = __builtin_amdgcn_s_memrealtime();
#endif
  const uint32_t r = blockIdx.x * 64u + lane;
  const uint32_t units = a.batch;
  const uint32_t rr = (r < units) ? r : (units - 1u);
  const bool live = r < units;
  float* const geo = sm.lds[1];
  GenLayout L;
  L.n = g.lay.n, L.nb = g.lay.nb, L.ncas = g.lay.ncas;
  const GenBuf RB = gen_buffer(g.rec, g.rstride, g.rec_bytes, L);
  const int mode = sm.mode_row[lane];
  float target[N];
  {
    const float4 ta = sm.tgt[0][lane], tb = sm.tgt[N > 4 ? 1 : 0][lane];
    const float tv[8] = {ta.x, ta.y, ta.z, ta.w, tb.x, tb.y, tb.z, tb.w};
#pragma unroll
    for (int i = 0; i < N; ++i) target[i] = tv[i];
  }
  Platform s;
  s.px = s0.x; s.py = s0.y; s.pz = s0.z; s.qx = s0.w;
  s.qy = s1.x; s.qz = s1.y; s.qw = s1.z; s.vx = s1.w;
  s.vy = s2.x; s.vz = s2.y; s.wx = s2.z; s.wy = s2.w;
  s.wz = s_wz;
  v2f q[NP], qd[NP];  // pairs 2k, 2k+1 of q in spill[k / 2], of qd behind them, then the structure matrix (LeanSpill)
#pragma unroll
  for (int k = 0; k < NP; k += 2) {
    const float4 a4 = spill[k / 2], b4 = spill[(NP + 1) / 2 + k / 2];
    q[k] = (v2f){a4.x, a4.y}, qd[k] = (v2f){b4.x, b4.y};
    if (k + 1 < NP) q[k + 1] = (v2f){a4.z, a4.w}, qd[k + 1] = (v2f){b4.z, b4.w};
  }
  const int now = g.now_step;
  int sel[N];
#pragma unroll
  for (int i = 0; i < N; ++i) sel[i] = (mode == 2 && fabsf(target[i]) > g.eps) ? 1 : 0;
  float force[N];
#pragma unroll
  for (int i = 0; i < N; ++i) force[i] = 0.f;
  GenDbg dbg{0.f, 0.f, 0.f, 0.f, false, false};
  GenCtlConst cc;
  cc.pcas_max = g.pcas_max, cc.dcas_max = g.dcas_max, cc.dt = g.dt, cc.inv_dt = a.inv_dt;
  cc.nm0 = now % max(g.nbuf0, 1), cc.nm1 = now % max(g.nbuf1, 1), cc.nbuf0 = g.nbuf0, cc.simple_ok = g.simple_ok != 0;
  cc.force_rows = nullptr;
#ifdef CDPR_STAMPS
#ifdef CDPR_STAMPS_COLD
  cc.stamps = a.stamps ? a.stamps + ((size_t)gridDim.x + blockIdx.x) * 8 : nullptr;
#else
  cc.stamps = nullptr;
#endif
#endif
#if defined(CDPR_STAMPS) && defined(CDPR_STAMPS_COLD)
  {
    float keep = 0.f;
#pragma unroll
    for (int i = 0; i < N; ++i) keep += target[i];
    asm volatile("" ::"v"(keep));
    if (a.stamps && lane == 0) a.stamps[((size_t)gridDim.x + blockIdx.x) * 8 + 6] = __builtin_amdgcn_s_memrealtime();
  }
#endif
  gen_controller<N, NBMAX, false>(cc, RB, L, lane, live, rr, blockIdx.x * 64u, units, mode, now, target, sel, q, qd, &sm.stage[0][0][0], &sm.hold_slots[0][0],
                                  &sm.wrot[0][0][0], sm.ptab, sm.q_count, force, dbg);
  v2f jac[NP][6];
  lean_controller_epilogue<N, true>(a, sm, geo, lane, r, live, s, q, qd, jac, force, dbg, spill + 2 * ((NP + 1) / 2));
  // (as an instruction the compiler does not know: behind __builtin_amdgcn_endpgm it restores every callee-saved register it
  //  saved on entry first - a hundred loads nobody reads.  The function's own epilogue behind it is never executed.)
  asm volatile("s_endpgm");
  __builtin_unreachable();
}

template <int N>
__global__ __launch_bounds__(128, 2) void cdpr_gen_lean_kernel(const StepArgs a, const GenCtl g) {
  constexpr int NBMAX = 11;
  constexpr int NP = cable_pairs(N);
  constexpr int NBP = gen_nbp(NBMAX);
  __shared__ LeanShared<N> sm;

  // (wave index as a scalar the compiler can see is uniform: otherwise `if (wave == 0)` is structurized as a divergent branch,
  //  the cold tail's call block flows into the common exit of both roles and everything the epilogue reads is saved around the call)
  const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63u;
  const uint32_t r = blockIdx.x * 64u + lane;
  const uint32_t units = a.batch;
  const uint32_t rr = (r < units) ? r : (units - 1u);
  const bool live = r < units;
  const size_t st = a.stride;
  const uint32_t off = rr * 16u, woff = r * 16u;
  float* const geo = sm.lds[wave];

  if (wave == 0) CDPR_SPLIT_STAMP(0);
  const float gval = a.geom[min(lane, (uint32_t)(NP * kGeomFloatsPerPair - 1))];
  if (wave == 0) {
    const float4 p0 = load_slot(a.state, st, 0, off), p1 = load_slot(a.state, st, 1, off), p3 = load_slot(a.state, st, 3, off);
    // The general controller makes the CONTROLLER wave the longer one (its forces leave 5 us after the Newton stage has finished): it takes
    // the raised issue priority the fast path's kernel gives the estimator wave, the estimator wave runs at the default.  65 536 x 8, same
    // box, us per step steady / cables switching Pids: estimator raised (round 5) 14.8 / 22.3, controller raised 14.5 / 21.6, neither
    // 15.0 / 22.2, both 15.2 / 22.3 (`profiles/r06_general_ab.txt`).
#ifndef CDPR_GEN_EST_RAISED
#define CDPR_GEN_EST_RAISED 0
#endif
#ifndef CDPR_GEN_CTL_PRIO
#define CDPR_GEN_CTL_PRIO 3
#endif
    split_estimator_wave<N, 64, 64, CDPR_GEN_EST_RAISED != 0>(a, geo, gval, lane, live, st, off, woff, p0, p1, p3, &sm.x_force()[0][0], &sm.x_tension()[0][0], &sm.x_est()[0][0]);
    return;
  }
  // ---------------------------------------------------------------------------------------------------- controller wave
  if (CDPR_GEN_CTL_PRIO) __builtin_amdgcn_s_setprio(CDPR_GEN_CTL_PRIO);
  GenLayout L;
  L.n = g.lay.n, L.nb = g.lay.nb, L.ncas = g.lay.ncas;
  GenBuf RB = gen_buffer(g.rec, g.rstride, g.rec_bytes, L);
  const uint32_t col = rr;
  constexpr uint32_t kW4 = 2u * NBMAX * NBP / 4u;
  constexpr int kWPass = (int)((kW4 + 63u) / 64u);
  float4 wv[kWPass];
#pragma unroll
  for (int j = 0; j < kWPass; ++j) wv[j] = reinterpret_cast<const float4*>(g.wtab)[min(lane + 64u * j, kW4 - 1u)];
  const float pv = g.ptab[min(lane, 2u * kGenPidFloats - 1u)];
  const float4 p0 = load_slot(a.state, st, 0, off), p1 = load_slot(a.state, st, 1, off), p2 = load_slot(a.state, st, 2, off),
               p3 = load_slot(a.state, st, 3, off);
  const bool hot_on = g.hot != 0 && g.simple_ok != 0;
  uint32_t hot_step1 = 0u, hot_mask = 0u;  // the robot's hot-row word (GenHot)
  if (hot_on) hot_step1 = RB.loadc(0, col * 4u), hot_mask = RB.loadc(1, col * 4u);
  const int mode = g.mode_arr ? (int)g.mode_arr[rr] : g.mode;
  float target[N];
  lean_load_targets<N>(g, rr, mode, target);
  sm.mode_row[lane] = mode;
  // (the Joy targets go to sm.tgt behind gen_hot_restore, round 6: until then those 2 KiB take the integrals' dword rows of a wave that is not all fresh)
  float* const dumpf = &sm.dump[0].x;
  *((lane < NP * kGeomFloatsPerPair) ? geo + lane : dumpf + lane) = gval;
#pragma unroll
  for (int j = 0; j < kWPass; ++j) *((lane + 64u * j < kW4) ? reinterpret_cast<float4*>(&sm.wrot[0][0][0]) + lane + 64u * j : &sm.dump[lane]) = wv[j];
  sm.q_count[0] = 0u;
  *((lane < 2 * kGenPidFloats) ? &sm.ptab[0][0].x + lane : dumpf + lane) = pv;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

  Platform s;
  s.px = p0.x; s.py = p0.y; s.pz = p0.z; s.qx = p0.w;
  s.qy = p1.x; s.qz = p1.y; s.qw = p1.z; s.vx = p1.w;
  s.vy = p2.x; s.vz = p2.y; s.wx = p2.z; s.wy = p2.w;
  s.wz = p3.x;

  const int now = g.now_step;
  int sel[N];
#pragma unroll
  for (int i = 0; i < N; ++i) {
    sel[i] = (mode == 2 && fabsf(target[i]) > g.eps) ? 1 : 0;
    asm volatile("" : "+v"(sel[i]));
  }
  GenHot hot{false, false, false, 0u, 0};  // hot rows (GenHot): this kernel starts and keeps them
  bool hot_skip = false;
  float* hot_rows = nullptr;  // the integrals' dword rows of a wave that is not all fresh, by DMA (in sm.tgt until gen_hot_restore has read them)
  {
    float keep = (s.px + s.qy) + (s.vy + s.wz);
#pragma unroll
    for (int i = 0; i < N; ++i) keep += target[i];
    bool holds = false;  // some cable of this lane is in the hold branch (JFC.cpp:78-82)
#pragma unroll
    for (int i = 0; i < N; ++i) holds = holds || (mode == 2 && sel[i] == 0);
    hot = gen_hot_begin<N>(hot_on, hot_step1, hot_mask, sel, mode, now, hot_skip);
    // 0: as in round 5 (a wave that is not all fresh requests every lane's H slots and reads the integrals from memory behind the wait for
    // the DMA); 1: the integrals' dword rows ride the DMA (no gain by itself: 22.1 against 21.9 us); 2: ... and the fresh lanes of such a
    // wave request no H slot (cables switching Pids, same box: 21.45 against 21.8 us by HIP events, 19.5 against 20.3 us and 92.1 against
    // 96.1 MB per launch by rocprofv3: `profiles/r06_general_ab.txt`, `r06d_general65k_sw_*`)
#ifndef CDPR_HOT_ROWS_BY_DMA
#define CDPR_HOT_ROWS_BY_DMA 2
#endif
    static_assert(sizeof(sm.tgt) >= sizeof(float) * 64 * N, "the integrals' dword rows fit the target rows");
    hot_rows = (CDPR_HOT_ROWS_BY_DMA && hot.on && !hot_skip && __builtin_amdgcn_ballot_w64(hot.has) != 0ull) ? &sm.tgt[0][0].x : nullptr;  // (wave-uniform)
    gen_stage_records<N, NBMAX>(RB, L, col, sel, &sm.stage[0][0][0], &sm.hold_slots[0][0], keep, __builtin_amdgcn_ballot_w64(holds) != 0ull, hot_skip, hot_rows, CDPR_HOT_ROWS_BY_DMA == 2 && hot.fresh);
  }

  // ---- IK on the state at t_k; the structure matrix stays alive for the world step (the inlined controller branch leaves room)
  v2f q[NP], qd[NP], jac[NP][6];
  {
    v2f len[NP], l0[NP];
    ik_pairs<N, true>(geo, s.px, s.py, s.pz, s.qx, s.qy, s.qz, s.qw, len, jac, l0);
#pragma unroll
    for (int k = 0; k < NP; ++k) {
      q[k] = l0[k] - len[k];
      qd[k] = -fma2(s.wz, jac[k][5], fma2(s.wy, jac[k][4], fma2(s.wx, jac[k][3],
                    fma2(s.vz, jac[k][2], fma2(s.vy, jac[k][1], splat(s.vx) * jac[k][0])))));
    }
  }

  // ---- per-cable force: the consecutive-call branch of the general controller inline, everything else in the cold tail
  float force[N];
#pragma unroll
  for (int i = 0; i < N; ++i) force[i] = 0.f;
  GenDbg dbg{0.f, 0.f, 0.f, 0.f, false, false};
  __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): the DMA has landed
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  CDPR_CTL_STAMP(7);
  GenCtlConst cc;
  cc.pcas_max = g.pcas_max, cc.dcas_max = g.dcas_max, cc.dt = g.dt, cc.inv_dt = a.inv_dt;
  cc.nm0 = now % max(g.nbuf0, 1), cc.nm1 = now % max(g.nbuf1, 1), cc.nbuf0 = g.nbuf0, cc.simple_ok = g.simple_ok != 0;
  cc.force_rows = nullptr;
#ifdef CDPR_STAMPS
#ifdef CDPR_STAMPS_COLD
  cc.stamps = a.stamps ? a.stamps + ((size_t)gridDim.x + blockIdx.x) * 8 : nullptr;  // (tier 1 inline: the same stamps as in the tail)
#else
  cc.stamps = nullptr;
#endif
#endif
  gen_hot_restore<N, NBMAX>(cc, RB, L, lane, live, col, hot, sel, &sm.stage[0][0][0], hot_rows);
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");  // (every lane has read its integrals before a lane's targets go over them)
  __builtin_amdgcn_wave_barrier();
  sm.tgt[0][lane] = make_float4(target[0], (1 < N) ? target[1 < N ? 1 : 0] : 0.f, (2 < N) ? target[2 < N ? 2 : 0] : 0.f, (3 < N) ? target[3 < N ? 3 : 0] : 0.f);
  if (N > 4) sm.tgt[1][lane] = make_float4(target[4 < N ? 4 : 0], (5 < N) ? target[5 < N ? 5 : 0] : 0.f, (6 < N) ? target[6 < N ? 6 : 0] : 0.f, (7 < N) ? target[7 < N ? 7 : 0] : 0.f);
  // tier 0 and - for waves without a gap call - tier 1 of the general controller inline (gen_controller<STEADY_ONLY>); what it
  // does not serve goes to the tail.
  // Two `if`s in sequence on the same wave-uniform decision, the second through a scalar the compiler cannot see through.  As ONE
  // if / else LLVM's structurizer puts the call block FIRST and lets it flow into the block in front of the epilogue:
  // everything the epilogue reads is then live across the call, the call block saves it (33 stores, 60 v_writelane) and the
  // tail's entry - s_waitcnt vmcnt(0) by the calling convention - waits for those stores.  Behind the epilogue nothing is
  // live any more but the tail's own arguments.
  cc.force_rows = &sm.tgt[0][0].x;  // (the Joy targets parked there are the tail's: it is entered only where nothing was written)
  const bool served = gen_controller<N, NBMAX, true, true, true>(cc, RB, L, lane, live, col, blockIdx.x * 64u, units, mode, now, target, sel, q, qd, &sm.stage[0][0][0],
                                                                 &sm.hold_slots[0][0], &sm.wrot[0][0][0], sm.ptab, sm.q_count, force, dbg, hot);
  uint32_t cold = served ? 0u : 1u;
  if (cold == 0u) {
#pragma unroll
    for (int i = 0; i < N; ++i) force[i] = (&sm.tgt[0][0].x)[i * 64 + lane];
    lean_controller_epilogue<N, false>(a, sm, geo, lane, r, live, s, q, qd, jac, force, dbg);
  }
  cold = __builtin_amdgcn_readfirstlane(cold);
  asm volatile("" : "+s"(cold));
  if (cold != 0u) {  // the other paths finish this wave's work in a function of their own and end the program there
    // (this wave is late already and decides when the launch ends: it wins the SIMD's issue arbitration from here on)
    __builtin_amdgcn_s_setprio(3);
    // (the tail's tiers write every H slot they touch: a robot that goes there with a valid word - its H slots restored in LDS
    //  above - has it cleared here; the tail itself runs with the hot rows off)
    if (hot.on) RB.storec_if(live && hot.has, 0, col * 4u, 0u);
    const uint64_t kaddr = (uint64_t)(const __attribute__((address_space(4))) void*)__builtin_amdgcn_kernarg_segment_ptr();
    // q, qd and the structure matrix travel through the lane's private memory (a stack object of this kernel that only this
    // block touches): the tail reads q and qd on entry and the matrix between the two barriers (rebuilding it there:
    // ~750 instructions; measured the same within the noise, 27.0 against 27.8 us per step with cables switching Pids)
    float4 spill[2 * ((NP + 1) / 2) + 3 * NP];
#pragma unroll
    for (int k = 0; k < NP; k += 2) {
      const v2f qz = (k + 1 < NP) ? q[k + 1 < NP ? k + 1 : k] : (v2f){0.f, 0.f}, qdz = (k + 1 < NP) ? qd[k + 1 < NP ? k + 1 : k] : (v2f){0.f, 0.f};
      spill[k / 2] = make_float4(q[k].x, q[k].y, qz.x, qz.y);
      spill[(NP + 1) / 2 + k / 2] = make_float4(qd[k].x, qd[k].y, qdz.x, qdz.y);
    }
#pragma unroll
    for (int k = 0; k < NP; ++k)
#pragma unroll
      for (int c = 0; c < 6; c += 2) spill[2 * ((NP + 1) / 2) + (k * 6 + c) / 2] = make_float4(jac[k][c].x, jac[k][c].y, jac[k][c + 1].x, jac[k][c + 1].y);
    gen_lean_cold_tail<N>((lean_f4){s.px, s.py, s.pz, s.qx}, (lean_f4){s.qy, s.qz, s.qw, s.vx}, (lean_f4){s.vy, s.vz, s.wx, s.wy}, s.wz, spill,
                          lds_address(&sm) | ((uint32_t)(kaddr >> 32) << 16), (uint32_t)kaddr);
  }
}

}  // namespace cdpr
