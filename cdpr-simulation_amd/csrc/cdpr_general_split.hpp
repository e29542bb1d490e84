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
  __shared__ uint32_t q_count;
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
  if (lane == 0) q_count = 0u;
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
#ifdef CDPR_STAMPS
    cc.stamps = nullptr;
#endif
    gen_controller<N, NBMAX>(cc, RB, L, lane, live, col, blockIdx.x * 64u, units, mode, now, target, sel, q, qd, &stage[0][0][0], &hold_slots[0][0],
                             &wrot[0][0][0], ptab, &q_count, force, dbg);
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
// general controller: the ring rotation, the fit queue and the fp64 fit live next to the steady-state branch that runs on
// all steps but the dozen after a mode change or a Pid switch.  Compiled for two waves per SIMD it spills on the steady
// path too (round 4: 48 - 300 B of scratch, 28.7 - 47 us at 65 536 x 8; this round 224 B: 20.4 us against the one-wave
// kernel's 21.8).  Here the controller wave INLINES only the steady-state branch (gen_controller<..., STEADY_ONLY>: 229
// registers with the structure matrix held for the world step, no scratch, no spilled scalar) and CALLS the rest
// (gen_controller_cold, a function with its own register allocation: 248 registers, 12 B of stack) on the steps that need
// it; the caller's live registers go to the stack around the call and nowhere else (80 scratch operations, all in the
// call's block).  Two waves per SIMD fit, so at 65 536 robots a SIMD hosts an estimator wave and a controller wave side
// by side as the fast path's cdpr_split_kernel does.  No optional physics here (those handles keep the other kernels).
// 65 536 x 8, hold branch live, HIP events (scripts/gen_lean_scan.py, profiles/r05_gen_lean_scan.txt): steady 16.5 us
// (one-wave kernel 22.4), cables switching Pids 32.0 (34.1); 16 384: 8.85 / 19.5 (role-split kernel above: 8.87 / 18.3).
// Measured and not kept: the steady-only kernel followed by the one-wave kernel over the blocks it flagged (two launches
// per step: 18.1 us steady, 46.7 switching - every block of a switching batch holds a switching robot).
// Same device functions, same arithmetic order: bit-identical to the other general kernels (tested).
template <int N>
__global__ __launch_bounds__(128, 2) void cdpr_gen_lean_kernel(const StepArgs a, const GenCtl g) {
  constexpr int NBMAX = 11;
  constexpr int NP = cable_pairs(N);
  constexpr int G = joint_groups(N);
  constexpr int NV = gen_nv(NBMAX);
  constexpr int NBP = gen_nbp(NBMAX);
  constexpr int LP = (N + 3) / 4;
  __shared__ __attribute__((aligned(16))) float lds[2][NP * kGeomFloatsPerPair];
  __shared__ __attribute__((aligned(16))) float wrot[2][NBMAX][NBP];
  __shared__ float4 stage[N][NV + 1][64];
  __shared__ float4 hold_slots[LP][64];
  __shared__ uint32_t q_count;
  __shared__ float4 ptab[2][kGenPidFloats / 4];
  static_assert(sizeof(stage) >= (2 * NP * 64) * sizeof(v2f) + 6 * 64 * sizeof(float), "hand-off buffers fit the staging area");
  v2f (*const x_force)[64] = reinterpret_cast<v2f(*)[64]>(&stage[0][0][0]);
  v2f (*const x_tension)[64] = x_force + NP;
  float (*const x_est)[64] = reinterpret_cast<float(*)[64]>(x_tension + NP);

  const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
  const uint32_t r = blockIdx.x * 64u + lane;
  const uint32_t units = a.batch;
  const uint32_t rr = (r < units) ? r : (units - 1u);
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
  if (g.mode_arr) {
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
  q_count = 0u;
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
  int sel[N];
#pragma unroll
  for (int i = 0; i < N; ++i) {
    sel[i] = (mode == 2 && fabsf(target[i]) > g.eps) ? 1 : 0;
    asm volatile("" : "+v"(sel[i]));
  }
  {
    float keep = (s.px + s.qy) + (s.vy + s.wz);
#pragma unroll
    for (int i = 0; i < N; ++i) keep += target[i];
    bool holds = false;  // some cable of this lane is in the hold branch (JFC.cpp:78-82)
#pragma unroll
    for (int i = 0; i < N; ++i) holds = holds || (mode == 2 && sel[i] == 0);
    gen_stage_records<N, NBMAX>(RB, L, col, sel, &stage[0][0][0], &hold_slots[0][0], keep, __builtin_amdgcn_ballot_w64(holds) != 0ull);
  }

  // ---- IK on the state at t_k; the structure matrix stays alive for the world step (the steady-state controller leaves room)
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
  const bool publish = (a.publish_mask & 1ull) != 0ull;
  float4* const obs = a.obs;

  // ---- per-cable force: the steady-state branch of the general controller inline, everything else by call
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
#ifdef CDPR_STAMPS
  cc.stamps = nullptr;
#endif
  const bool steady = gen_controller<N, NBMAX, true>(cc, RB, L, lane, live, col, blockIdx.x * 64u, units, mode, now, target, sel, q, qd, &stage[0][0][0],
                                                     &hold_slots[0][0], &wrot[0][0][0], ptab, &q_count, force, dbg);
  if (!steady) {  // (wave-uniform) the rare paths as a call: their registers are not this kernel's (gen_controller_cold)
    GenColdIn<N> in;
#pragma unroll
    for (int i = 0; i < N; ++i) in.target[i] = target[i], in.sel[i] = sel[i];
#pragma unroll
    for (int k = 0; k < NP; ++k) in.q[k] = q[k], in.qd[k] = qd[k];
    const GenColdOut<N> out = gen_controller_cold<N, NBMAX>(cc, (uint64_t)(uintptr_t)g.rec, g.rstride, g.rec_bytes, L, lane, live, col, blockIdx.x * 64u, units, mode, now, in,
                                                            lds_address(&stage[0][0][0]), lds_address(&hold_slots[0][0]), lds_address(&wrot[0][0][0]),
                                                            lds_address(&ptab[0][0]), lds_address(&q_count));
#pragma unroll
    for (int i = 0; i < N; ++i) force[i] = out.force[i];
    dbg = out.dbg;
  }
#pragma unroll
  for (int k = 0; k < NP; ++k) x_force[k][lane] = (v2f){force[2 * k], (2 * k + 1 < N) ? force[2 * k + 1] : 0.f};
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): the forces are in LDS
  CDPR_CTL_STAMP(4);
  __builtin_amdgcn_s_barrier();        // #1
  // the observables that are final after the IK (PLG.cpp:248-280) go out HERE, while this wave waits for the tensions: in
  // front of the controller they sat between the DMA and the vmcnt(0) that waits for it, and the wave paid their
  // completion (2 - 3 us at 65 536 robots, profiles/r05_lean_timeline.txt) on the path the estimator wave waits for
  if (publish && live) {
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
  if (live) {
    CDPR_STORE_STATE(a.state, st, 0, woff, make_float4(s.px, s.py, s.pz, s.qx));
    CDPR_STORE_STATE(a.state, st, 1, woff, make_float4(s.qy, s.qz, s.qw, s.vx));
    CDPR_STORE_STATE(a.state, st, 2, woff, make_float4(s.vy, s.vz, s.wx, s.wy));
    CDPR_STORE_STATE(a.state, st, 3, woff, make_float4(s.wz, fkx, fky, fkz));
  }
  CDPR_CTL_STAMP(6);
}

}  // namespace cdpr
