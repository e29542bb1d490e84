// cdpr_step_kernel_pair.hpp — "two lanes per robot" mapping of the fused CDPR step (gfx950).
//
// Why: with one lane per robot, 65 536 robots are 1 024 wavefronts = ONE wave per SIMD; the profile
// (profiles/r01_b_*) shows what that costs: every VALU instruction issues at 4 cycles instead of 2 and
// every dependent latency (the serial 6x6 Cholesky, v_rsq) is exposed — 44 % of the wave's cycles
// issuing, 32 % dependency stalls, nothing to switch to.  Here adjacent lanes (2r, 2r+1) share robot
// r: lane parity p owns cable pairs [p*NPL, (p+1)*NPL) — half the cables — so the per-cable work (IK
// rows, FIR windows, PID, Gram partial sums, wrench partial sums, controller rows in HBM) halves per
// lane and the machine runs TWO waves per SIMD.  Partial sums meet through one DPP quad_perm add
// (lane ^ 1); fp addition commutes, so both lanes hold bit-identical sums and run the small serial
// part (6x6 Cholesky, quaternion update, integration) redundantly on identical data.
//
// HBM layout is exactly the lane-per-robot one (cdpr_step_kernel.hpp): the two mappings are
// interchangeable on the same state.  Built for n = 4 and n = 8 (an even number of cable pairs).
#pragma once
#include <type_traits>

#include "cdpr_step_kernel.hpp"

namespace cdpr {

// value of the partner lane (lane ^ 1)
CDPR_DEV float partner(float x) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0xB1, 0xF, 0xF, true));
}
CDPR_DEV float pair_sum(float x) { return x + partner(x); }
CDPR_DEV float pair_max(float x) { return fmaxf(x, partner(x)); }

// Solve (J^T J + lambda I) x = g in place; J^T J summed over this lane's pairs and the partner's.
template <int NPL>
CDPR_DEV void normal_solve_shared(const v2f (&jac)[NPL][6], float lambda, float (&g)[6]) {
  v2f acc[21];
  gram_partial<NPL>(jac, acc);
  float m[6][6];
#pragma unroll
  for (int a = 0, e = 0; a < 6; ++a) {
#pragma unroll
    for (int b = 0; b <= a; ++b, ++e) m[a][b] = pair_sum(hsum(acc[e])) + ((a == b) ? lambda : 0.f);
  }
  chol_solve(m, g);
}

template <int NPL>
CDPR_DEV void jt_times_shared(const v2f (&jac)[NPL][6], const v2f (&v)[NPL], float (&g)[6]) {
  v2f acc[6];
  jt_partial<NPL>(jac, v, acc);
#pragma unroll
  for (int c = 0; c < 6; ++c) g[c] = pair_sum(hsum(acc[c]));
}

template <int N, bool FK, bool TD, bool SINGLE>
__global__ __launch_bounds__(64, 2) void cdpr_step_kernel_pair(const StepArgs a) {
  static_assert(N == 4 || N == 8, "the lane-pair mapping needs an even number of cable pairs");
  constexpr int NP = N / 2;    // cable pairs of the robot
  constexpr int NPL = NP / 2;  // cable pairs owned by one lane
  constexpr int NL = N / 2;    // cables owned by one lane
  constexpr int P = plat_slots(FK);
  constexpr int G = joint_groups(N);
  __shared__ __attribute__((aligned(16))) float lds[NP * kGeomFloatsPerPair];

  const uint32_t lane = threadIdx.x;
  const uint32_t par = lane & 1u;                       // which half of the cables
  const uint32_t r = blockIdx.x * 32u + (lane >> 1);    // robot
  const uint32_t rr = (r < a.batch) ? r : (a.batch - 1u);
  const bool live = r < a.batch;
  const size_t st = a.stride;
  const int c0 = (int)par * NL;                         // first cable of this lane

  CDPR_STAMP(0);
  const float gval = (lane < NP * kGeomFloatsPerPair) ? a.geom[lane] : 0.f;
  const uint32_t off = rr * 16u, woff = r * 16u;
  const float4 p0 = load_slot(a.state, st, 0, off), p1 = load_slot(a.state, st, 1, off),
               p2 = load_slot(a.state, st, 2, off), p3 = load_slot(a.state, st, 3, off);
  float4 p4 = make_float4(0.f, 0.f, 0.f, 1.f);
  if (FK) p4 = load_slot(a.state, st, 4, off);
  // controller rows of this lane's cable pairs: the row index depends on the lane parity, so these are per-lane
  // addresses (two 512-B runs per wave instruction).  Layout: see cdpr_step_kernel.hpp (ring rows + hot rows).
  const int k0 = (int)par * NPL;  // first cable pair of this lane
  const float4* crow = a.state + (size_t)(P + 5 * k0) * st + rr;
  float4 wraw[NPL][5];
#pragma unroll
  for (int k = 0; k < NPL; ++k) {
#pragma unroll
    for (int m = 0; m < 5; ++m) wraw[k][m] = crow[(size_t)(5 * k + m) * st];
  }
  // hot row: n = 8 -> row `par` holds exactly this lane's two pairs; n = 4 -> one row shared by both lanes
  const float4 hraw = (a.state + (size_t)(P + 5 * NP + (NPL == 2 ? par : 0u)) * st)[rr];
  v2f desired[NPL];
#pragma unroll
  for (int k = 0; k < NPL; ++k) desired[k] = splat(0.f);
  auto load_joy = [&](const float* cp) {  // this lane's half of one robot's Joy.axes
    if (NL == 4) {
      const float4 v = *reinterpret_cast<const float4*>(cp);
      desired[0] = (v2f){v.x, v.y};
      desired[NPL - 1] = (v2f){v.z, v.w};
    } else {
      const float2 v = *reinterpret_cast<const float2*>(cp);
      desired[0] = (v2f){v.x, v.y};
    }
  };
  if (!SINGLE && a.sched_refresh > 0) sched_wait(a, 0);
  load_joy(a.cmd + (size_t)rr * N + c0);

  if (lane < NP * kGeomFloatsPerPair) lds[lane] = gval;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  const float* mylds = lds + par * (NPL * kGeomFloatsPerPair);

  Platform s;
  s.px = p0.x; s.py = p0.y; s.pz = p0.z; s.qx = p0.w;
  s.qy = p1.x; s.qz = p1.y; s.qw = p1.z; s.vx = p1.w;
  s.vy = p2.x; s.vz = p2.y; s.wx = p2.z; s.wy = p2.w;
  s.wz = p3.x;
  float fkx = p3.y, fky = p3.z, fkz = p3.w, fkqx = p4.x, fkqy = p4.y, fkqz = p4.z, fkqw = p4.w;

  v2f win[NPL][kWin], ierr[NPL];
#pragma unroll
  for (int k = 0; k < NPL; ++k) {
#pragma unroll
    for (int m = 0; m < 5; ++m) {
      win[k][2 * m] = (v2f){wraw[k][m].x, wraw[k][m].y};      // register pairs as loaded: no moves
      win[k][2 * m + 1] = (v2f){wraw[k][m].z, wraw[k][m].w};
    }
  }
  if (NPL == 2) {
    ierr[0] = (v2f){hraw.x, hraw.y};
    ierr[NPL - 1] = (v2f){hraw.z, hraw.w};
  } else {
    ierr[0] = par ? (v2f){hraw.z, hraw.w} : (v2f){hraw.x, hraw.y};
  }
  const bool actual_is_vel = (a.flags & kFlagActualIsVelocity) != 0u;
  int calls = a.pid_calls;

  for (int step = 0; step < (SINGLE ? 1 : a.nsteps); ++step) {
    if (!SINGLE && a.sched_refresh > 0 && step > 0 && step % a.sched_refresh == 0) {
      // a launch over a command schedule (cdpr_update_scheduled): the next Joy batch at every refresh boundary
      const int j = step / a.sched_refresh;
      sched_wait(a, j);
      load_joy(a.cmd + (size_t)j * a.sched_stride + (size_t)rr * N + c0);
    }
    CDPR_STAMP(1);
    // ---- IK rows of this lane's cables on the state at t_k
    v2f len[NPL], jac[NPL][6], l0[NPL], q[NPL], qd[NPL];
    ik_rows<NPL, false, true>(mylds, s.px, s.py, s.pz, s.qx, s.qy, s.qz, s.qw, len, jac, l0);
#pragma unroll
    for (int k = 0; k < NPL; ++k) {
      q[k] = l0[k] - len[k];
      qd[k] = -fma2(s.wz, jac[k][5], fma2(s.wy, jac[k][4], fma2(s.wx, jac[k][3],
                    fma2(s.vz, jac[k][2], fma2(s.vy, jac[k][1], splat(s.vx) * jac[k][0])))));
    }

    CDPR_STAMP(2);
    // ---- per-cable force (PLG.cpp:222-228 -> JFC.cpp:59-96 -> Pid.cpp:122-191)
    v2f f[NPL], e_new[NPL];
#pragma unroll
    for (int k = 0; k < NPL; ++k) {
      f[k] = splat(0.f);
      e_new[k] = splat(0.f);
    }
    float dbg_p = 0.f, dbg_i = 0.f, dbg_d = 0.f;
    bool dbg_wrote = false;
    const bool first_world = (step == 0) && (a.flags & kFlagFirstWorldStep);
    int ring_slot = -1;
    if (!first_world && (a.flags & kFlagForceMode)) {  // UpdateMode::Force (JFC.cpp:67-70): no Pid
#pragma unroll
      for (int k = 0; k < NPL; ++k) f[k] = desired[k];
    } else if (!first_world) {
      if (calls != 0) {
        const bool full = calls >= a.nbuf;
        ring_slot = (a.ring_slot + step) % kWin;
        const float* wt = a.wtab + ring_slot * (kWin + 2);
        v2f error[NPL], acc[NPL];
#pragma unroll
        for (int k = 0; k < NPL; ++k) {
          error[k] = desired[k] - (actual_is_vel ? qd[k] : q[k]);
          acc[k] = splat(wt[kWin]) * error[k];
        }
#pragma unroll
        for (int j = 0; j < kWin; ++j) {
#pragma unroll
          for (int k = 0; k < NPL; ++k) acc[k] = fma2(wt[j], win[k][j], acc[k]);
        }
#pragma unroll
        for (int k = 0; k < NPL; ++k) {
          const v2f p_term = splat(a.kp) * error[k];
          const v2f prev_ierr = ierr[k];
          v2f ie = fma2(a.dt, error[k], prev_ierr);
          const v2f i_term = splat(a.ki) * ie;
          const v2f i_cl = max2(min2(i_term, splat(a.imax)), splat(a.imin));
          const v2f ie_cl = i_cl * splat(a.inv_ki);
          ie.x = (i_cl.x != i_term.x) ? ie_cl.x : ie.x;
          ie.y = (i_cl.y != i_term.y) ? ie_cl.y : ie.y;
          const v2f derived = full ? acc[k] * splat(a.inv_dt) : splat(0.f);
          const v2f d_term = splat(a.kd) * derived;
          const v2f cmd = fma2(a.kf, desired[k], p_term) + i_cl + d_term;
          v2f out = a.clamp_cmd ? max2(min2(cmd, splat(a.cmax)), splat(a.cmin)) : cmd;
          const v2f bumped = fma2(splat(a.dt) * error[k], splat(a.ki), out);
          ie.x = (out.x != cmd.x) ? prev_ierr.x : ie.x;
          ie.y = (out.y != cmd.y) ? prev_ierr.y : ie.y;
          out.x = (out.x != cmd.x) ? bumped.x : out.x;
          out.y = (out.y != cmd.y) ? bumped.y : out.y;
          ierr[k] = ie;
          f[k] = out;
          e_new[k] = error[k];
          if (k == 0) {
            dbg_p = p_term.x;
            dbg_i = i_term.x;
            dbg_d = d_term.x;
          }
        }
        dbg_wrote = true;
      }
      ++calls;
    }

    if (!SINGLE && ring_slot >= 0) ring_push<NPL>(win, e_new, ring_slot);
    {
      // hot row of n = 4 is shared: lane 0 collects the partner's pair (DPP outside any divergent branch)
      const float oi0 = partner(ierr[0].x), oi1 = partner(ierr[0].y);
      if (SINGLE && live) {
        float4* wrow = a.state + (size_t)(P + 5 * k0) * st + r;
        if (ring_slot >= 0) {
#pragma unroll
          for (int m = 0; m < 5; ++m) {
            if (m == (ring_slot >> 1)) {
#pragma unroll
              for (int k = 0; k < NPL; ++k) wrow[(size_t)(5 * k + m) * st] = ring_row(win[k], m, e_new[k], ring_slot);
            }
          }
        }
        if (NPL == 2) {
          (a.state + (size_t)(P + 5 * NP + par) * st)[r] = make_float4(ierr[0].x, ierr[0].y, ierr[NPL - 1].x, ierr[NPL - 1].y);
        } else if (par == 0u) {
          (a.state + (size_t)(P + 5 * NP) * st)[r] = make_float4(ierr[0].x, ierr[0].y, oi0, oi1);
        }
      }
    }

    CDPR_STAMP(3);
    // ---- Newton-Raphson FK: the two lanes iterate on identical estimates
    v2f applied[NPL];
    float fk_res = 0.f;
    int fk_it = 0, td_flag = 0;
    v2f jest[NPL][6];
    if (FK) {
      v2f elen[NPL], unused[NPL];
      bool active = true;
      for (int it = 0; it < a.fk_iters; ++it) {
        ik_rows<NPL, false, false>(mylds, fkx, fky, fkz, fkqx, fkqy, fkqz, fkqw, elen, jest, unused);
        v2f res[NPL];
        v2f rm = splat(0.f);
#pragma unroll
        for (int k = 0; k < NPL; ++k) {
          res[k] = len[k] - elen[k];
          rm = max2(rm, abs2(res[k]));
        }
        active = active && !(pair_max(fmaxf(rm.x, rm.y)) < a.fk_tol);
        float g[6];
        jt_times_shared<NPL>(jest, res, g);
        normal_solve_shared<NPL>(jest, a.fk_lambda, g);
        if (active) {
          fkx += g[0];
          fky += g[1];
          fkz += g[2];
          quat_apply_rotvec(fkqx, fkqy, fkqz, fkqw, g[3], g[4], g[5]);
          ++fk_it;
        }
      }
      ik_rows<NPL, false, false>(mylds, fkx, fky, fkz, fkqx, fkqy, fkqz, fkqw, elen, jest, unused);
      v2f rm = splat(0.f);
#pragma unroll
      for (int k = 0; k < NPL; ++k) rm = max2(rm, abs2(len[k] - elen[k]));
      fk_res = pair_max(fmaxf(rm.x, rm.y));
    }

    CDPR_STAMP(4);
    // ---- tension distribution
    if (TD) {
      v2f df[NPL];
#pragma unroll
      for (int k = 0; k < NPL; ++k) df[k] = f[k] - splat(a.td_mid);
      float g[6];
      if (FK) {
        jt_times_shared<NPL>(jest, df, g);
        normal_solve_shared<NPL>(jest, 0.f, g);
      } else {
        jt_times_shared<NPL>(jac, df, g);
        normal_solve_shared<NPL>(jac, 0.f, g);
      }
      float flagf = 0.f;
#pragma unroll
      for (int k = 0; k < NPL; ++k) {
        v2f t = splat(a.td_mid);
#pragma unroll
        for (int c = 0; c < 6; ++c) t = fma2(g[c], FK ? jest[k][c] : jac[k][c], t);
        const v2f tc = max2(min2(t, splat(a.td_max)), splat(a.td_min));
        flagf = ((tc.x != t.x) || (tc.y != t.y)) ? 1.f : flagf;
        applied[k] = tc;
      }
      td_flag = (pair_max(flagf) != 0.f) ? 1 : 0;
    } else {
#pragma unroll
      for (int k = 0; k < NPL; ++k) applied[k] = f[k];
    }
    if (a.vel_limit > 0.f) {  // Joint::SetForce velocity truncation [EXT]: no pushing a runaway joint further out
#pragma unroll
      for (int k = 0; k < NPL; ++k) {
        applied[k].x = (qd[k].x > a.vel_limit && applied[k].x > 0.f) || (qd[k].x < -a.vel_limit && applied[k].x < 0.f) ? 0.f : applied[k].x;
        applied[k].y = (qd[k].y > a.vel_limit && applied[k].y > 0.f) || (qd[k].y < -a.vel_limit && applied[k].y < 0.f) ? 0.f : applied[k].y;
      }
    }
    if (a.effort >= 0.f) {
#pragma unroll
      for (int k = 0; k < NPL; ++k) applied[k] = max2(min2(applied[k], splat(a.effort)), splat(-a.effort));
    }

    if (a.dbg && live && par == 0u) {  // `pid` topic, cable 0 only
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
    // ---- observables of step t_k: lane 0 writes the platform rows, each lane its own joint group
    float4* const obs = a.obs + (size_t)step * a.obs_step_stride;
    // travel-limit flags (cube.sdf:436-437): each lane tests its own cables, the two halves meet through the partner exchange
    uint32_t limit_mask = 0u;
    if (a.travel_on) {
      uint32_t mine = 0u;
#pragma unroll
      for (int k = 0; k < NPL; ++k) {
        mine |= (q[k].x < a.travel_lo || q[k].x > a.travel_hi) ? (1u << (2 * k)) : 0u;
        mine |= (q[k].y < a.travel_lo || q[k].y > a.travel_hi) ? (1u << (2 * k + 1)) : 0u;
      }
      mine <<= par * NL;
      limit_mask = mine | __float_as_uint(partner(__uint_as_float(mine)));
    }
    if (step_published(a, step) && live) {
      if (par == 0u) {
        store_slot(obs, st, 0, woff, make_float4(s.px, s.py, s.pz, s.qx));
        store_slot(obs, st, 1, woff, make_float4(s.qy, s.qz, s.qw, s.vx));
      } else {
        store_slot(obs, st, 2, woff, make_float4(s.vy, s.vz, s.wx, s.wy));
        store_slot(obs, st, 3, woff, make_float4(s.wz, fk_res, (float)fk_it, pack_flags(td_flag, limit_mask)));
      }
      if (NL == 4) {  // n = 8: group g = this lane's four cables
        float4* orow = obs + (size_t)(4 + par) * st + r;
        orow[0] = make_float4(q[0].x, q[0].y, q[NPL - 1].x, q[NPL - 1].y);
        orow[(size_t)G * st] = make_float4(qd[0].x, qd[0].y, qd[NPL - 1].x, qd[NPL - 1].y);
        orow[(size_t)2 * G * st] = make_float4(applied[0].x, applied[0].y, applied[NPL - 1].x, applied[NPL - 1].y);
      }
    }
    if (NL == 2) {  // n = 4: one group of four = both lanes' pairs; lane 0 collects the partner's pair
      const float oq0 = partner(q[0].x), oq1 = partner(q[0].y), ov0 = partner(qd[0].x), ov1 = partner(qd[0].y);
      const float oa0 = partner(applied[0].x), oa1 = partner(applied[0].y);
      if (step_published(a, step) && live && par == 0u) {
        store_slot(obs, st, 4, woff, make_float4(q[0].x, q[0].y, oq0, oq1));
        store_slot(obs, st, 4 + G, woff, make_float4(qd[0].x, qd[0].y, ov0, ov1));
        store_slot(obs, st, 4 + 2 * G, woff, make_float4(applied[0].x, applied[0].y, oa0, oa1));
      }
    }

    // ---- world step: wrench partial sums meet through the DPP add, both lanes integrate identically
    {
      v2f tens[NPL];
#pragma unroll
      for (int k = 0; k < NPL; ++k) {
        tens[k] = fma2(-a.damping, qd[k], applied[k]);
        if (a.unilateral) tens[k] = max2(tens[k], splat(0.f));  // [NEW] option: a cable cannot push
      }
      float w[6];
      jt_times_shared<NPL>(jac, tens, w);
      w[0] = a.fgx - w[0];
      w[1] = a.fgy - w[1];
      w[2] = a.fgz - w[2];
      w[3] = -w[3];
      w[4] = -w[4];
      w[5] = -w[5];
      integrate(a, s, w);
    }
  }

  CDPR_STAMP(6);
  // ---- store (platform rows split between the two lanes)
  const float oi_final0 = partner(ierr[0].x), oi_final1 = partner(ierr[0].y);
  if (live) {
    if (par == 0u) {
      store_slot(a.state, st, 0, woff, make_float4(s.px, s.py, s.pz, s.qx));
      store_slot(a.state, st, 1, woff, make_float4(s.qy, s.qz, s.qw, s.vx));
      if (FK) store_slot(a.state, st, 4, woff, make_float4(fkqx, fkqy, fkqz, fkqw));
    } else {
      store_slot(a.state, st, 2, woff, make_float4(s.vy, s.vz, s.wx, s.wy));
      store_slot(a.state, st, 3, woff, make_float4(s.wz, fkx, fky, fkz));
    }
    if (!SINGLE) {
      float4* wrow = a.state + (size_t)(P + 5 * k0) * st + r;
#pragma unroll
      for (int k = 0; k < NPL; ++k) {
#pragma unroll
        for (int m = 0; m < 5; ++m) wrow[(size_t)(5 * k + m) * st] = ring_row(win[k], m, splat(0.f), -1);
      }
      if (NPL == 2) {
        (a.state + (size_t)(P + 5 * NP + par) * st)[r] = make_float4(ierr[0].x, ierr[0].y, ierr[NPL - 1].x, ierr[NPL - 1].y);
      } else if (par == 0u) {
        (a.state + (size_t)(P + 5 * NP) * st)[r] = make_float4(ierr[0].x, ierr[0].y, oi_final0, oi_final1);
      }
    }
  }
  CDPR_STAMP(7);
}

// =====================================================================================================================
// cdpr_pair_stream_kernel — the several-steps launch of the lane-pair mapping in its STEADY STATE (round 6): what
// cdpr_update_scheduled / cdpr_update_fused / cdpr_update_record run on FK-less handles once every derivative window is
// full.  BASELINE config 2 (4 096 x 4 cables) is 128 waves on 1 024 SIMDs: a wave has its SIMD to itself, it issues one
// instruction every ~5 cycles whatever the instruction is, and the step is one wave's instruction COUNT - scalar
// bookkeeping included.  The general several-steps kernel above spends ~500 instructions per step on ~230 of arithmetic
// (profiles/r06_config2_region_budget.txt): the ring position is a run-time value (20 selects to push one error, the weight
// row fetched from memory on the PID's critical path every step), every optional feature is a uniform branch, the scalar
// registers spill into VGPR lanes, each observable row builds its own buffer descriptor.  This kernel serves only the
// launches where none of that is needed - the host checks (pair_stream_ok in cdpr_engine.hip): not world step 0, Pid mode,
// windows full for the whole launch, every step published, command and effort clamps on, no travel flags / velocity limit /
// unilateral cables / pid debug topic / mailbox - and runs:
//   * the step body TEN times, once per ring position, entered through a switch on the launch's first position (Duff's
//     device): the position is a compile-time constant in each copy, so pushing the new error is a register assignment and
//     the weight of every window register is a fixed scalar register (the 11 weights by age, from StepArgs::wrow);
//   * one buffer descriptor per observable image, advanced by two scalar adds per step, the rows as per-lane offsets
//     computed once; every lane stores ITS half of the joint rows (8-byte stores at n = 4) instead of collecting the
//     partner's half through DPP moves;
//   * the next Joy batch of a schedule fetched one refresh period ahead.
// Same device functions, same order of every floating-point operation as the general kernel: bit-identical results (tested:
// scheduled against launch-per-step, fused against single).  VEL: the Pid sees the joint velocity (velocity mode).
// ring register J holds, in the copy of the step for ring position S, the error of this many steps ago
constexpr int stream_ring_age(int S, int J) { return ((S - J) % kWin + kWin) % kWin == 0 ? kWin : ((S - J) % kWin + kWin) % kWin; }
template <int S, int... J>
CDPR_DEV v2f stream_fir(const float (&wage)[kWin + 1], const v2f (&win)[kWin], v2f acc, std::integer_sequence<int, J...>) {
  ((acc = fma2(wage[stream_ring_age(S, J)], win[J], acc)), ...);  // J ascending: the general kernels' order of summation
  return acc;
}
// max(min(v, hi), lo) per half as ONE v_med3_f32 (lo <= hi; equal to the two-instruction form for every non-NaN v, and the
// operands need no canonicalising v_max_f32 x, x first - min2 / max2 on opaque constants cost one per constant per step)
CDPR_DEV v2f clamp2_med3(v2f v, float lo, float hi) { return (v2f){__builtin_amdgcn_fmed3f(v.x, lo, hi), __builtin_amdgcn_fmed3f(v.y, lo, hi)}; }
// the bits of a float taken BY VALUE: __builtin_bit_cast applied directly to an element of an ext_vector_type (q[0].y) reads
// the vector's storage from its start - element 0 whatever the element named (clang 20 / ROCm 7.2; found in round 6 as joint
// rows with x in both halves)
CDPR_DEV unsigned fbits(float x) { return __builtin_bit_cast(unsigned, x); }
// a wave-uniform constant held in a VECTOR register: a packed operation broadcasts a VGPR half through op_sel, where a scalar
// operand needs an aligned SGPR pair with the value in both halves - and the kernel has more constants than SGPRs
CDPR_DEV float in_vgpr(float x) {
  asm volatile("" : "+v"(x));
  return x;
}

template <int N, bool VEL>
__global__ __launch_bounds__(64, 2) void cdpr_pair_stream_kernel(const StepArgs a) {
  static_assert(N == 4 || N == 8, "the lane-pair mapping needs an even number of cable pairs");
  constexpr int NP = N / 2, NPL = NP / 2, NL = N / 2;
  constexpr int P = plat_slots(false);
  constexpr int G = joint_groups(N);
  __shared__ __attribute__((aligned(16))) float lds[NP * kGeomFloatsPerPair];

  const uint32_t lane = threadIdx.x;
  const uint32_t par = lane & 1u;
  const uint32_t r = blockIdx.x * 32u + (lane >> 1);
  const uint32_t rr = (r < a.batch) ? r : (a.batch - 1u);
  const bool live = r < a.batch;
  const size_t st = a.stride;
  const int c0 = (int)par * NL;
  const float gval = (lane < NP * kGeomFloatsPerPair) ? a.geom[lane] : 0.f;
  const uint32_t off = rr * 16u, woff = r * 16u;
  const float4 p0 = load_slot(a.state, st, 0, off), p1 = load_slot(a.state, st, 1, off),
               p2 = load_slot(a.state, st, 2, off), p3 = load_slot(a.state, st, 3, off);
  const int k0 = (int)par * NPL;
  const float4* crow = a.state + (size_t)(P + 5 * k0) * st + rr;
  float4 wraw[NPL][5];
#pragma unroll
  for (int k = 0; k < NPL; ++k) {
#pragma unroll
    for (int m = 0; m < 5; ++m) wraw[k][m] = crow[(size_t)(5 * k + m) * st];
  }
  const float4 hraw = (a.state + (size_t)(P + 5 * NP + (NPL == 2 ? par : 0u)) * st)[rr];
  v2f desired[NPL], next_joy[NPL];
  auto load_joy = [&](const float* cp, v2f(&d)[NPL]) {
    if (NL == 4) {
      const float4 v = *reinterpret_cast<const float4*>(cp);
      d[0] = (v2f){v.x, v.y};
      d[NPL - 1] = (v2f){v.z, v.w};
    } else {
      const float2 v = *reinterpret_cast<const float2*>(cp);
      d[0] = (v2f){v.x, v.y};
    }
  };
  const float* const joy0 = a.cmd + (size_t)rr * N + c0;
  load_joy(joy0, desired);
  const int refresh = a.sched_refresh > 0 ? a.sched_refresh : 0x7fffffff;
  const int nbatches = a.sched_refresh > 0 ? (a.nsteps + a.sched_refresh - 1) / a.sched_refresh : 1;
  int batch_next = 1;  // the schedule batch `next_joy` holds (or will hold)
#pragma unroll
  for (int k = 0; k < NPL; ++k) next_joy[k] = desired[k];
  if (batch_next < nbatches) load_joy(joy0 + (size_t)batch_next * a.sched_stride, next_joy);

  if (lane < NP * kGeomFloatsPerPair) lds[lane] = gval;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  const float* mylds = lds + par * (NPL * kGeomFloatsPerPair);

  // the platform as thirteen scalars carried from step to step (a Platform struct captured by the step body stays a stack
  // object: LLVM then keeps it in LDS, 52 B per lane)
  float s_px = p0.x, s_py = p0.y, s_pz = p0.z, s_qx = p0.w, s_qy = p1.x, s_qz = p1.y, s_qw = p1.z, s_vx = p1.w, s_vy = p2.x, s_vz = p2.y,
        s_wx = p2.z, s_wy = p2.w, s_wz = p3.x;
  const float fkx = p3.y, fky = p3.z, fkz = p3.w;  // (no estimator on this handle: the row's other words pass through)

  v2f win[NPL][kWin], ierr[NPL];
#pragma unroll
  for (int k = 0; k < NPL; ++k) {
#pragma unroll
    for (int m = 0; m < 5; ++m) {
      win[k][2 * m] = (v2f){wraw[k][m].x, wraw[k][m].y};
      win[k][2 * m + 1] = (v2f){wraw[k][m].z, wraw[k][m].w};
    }
  }
  if (NPL == 2) {
    ierr[0] = (v2f){hraw.x, hraw.y};
    ierr[NPL - 1] = (v2f){hraw.z, hraw.w};
  } else {
    ierr[0] = par ? (v2f){hraw.z, hraw.w} : (v2f){hraw.x, hraw.y};
  }
  // the derivative weights BY AGE: wage[0] the new error's, wage[j] that of the error of j steps ago.  StepArgs::wrow is the
  // weight row of ring position 0 here (the host's choice for this kernel): position 0 holds age 10, position s age 10 - s
  const float wage[kWin + 1] = {in_vgpr(a.wrow[kWin]), in_vgpr(a.wrow[9]), in_vgpr(a.wrow[8]), in_vgpr(a.wrow[7]), in_vgpr(a.wrow[6]), in_vgpr(a.wrow[5]),
                               in_vgpr(a.wrow[4]), in_vgpr(a.wrow[3]), in_vgpr(a.wrow[2]), in_vgpr(a.wrow[1]), in_vgpr(a.wrow[0])};
  const float kp = in_vgpr(a.kp), ki = in_vgpr(a.ki), kd = in_vgpr(a.kd), kf = in_vgpr(a.kf), inv_ki = in_vgpr(a.inv_ki), imax = in_vgpr(a.imax),
              imin = in_vgpr(a.imin), cmax = in_vgpr(a.cmax), cmin = in_vgpr(a.cmin), inv_dt = in_vgpr(a.inv_dt), effort = in_vgpr(a.effort),
              neg_effort = in_vgpr(-a.effort), neg_damping = in_vgpr(-a.damping), dt_v = in_vgpr(a.dt);

  // observable image of one step: ONE descriptor (advanced per step), rows as per-lane byte offsets.  Lanes of robots past the
  // batch get an offset the descriptor's range check drops.
  const uint32_t image_bytes = (uint32_t)(obs_slots(N) * st * sizeof(float4));
  const uint32_t row_bytes = (uint32_t)(st * sizeof(float4));
  const uint32_t dead = 0xFFFFFFFFu;
  // platform rows 0, 1 are lane 0's, rows 2, 3 lane 1's: every lane issues all four stores, with a dropped offset for the
  // rows of the other lane (a select of the DATA by lane parity costs eight moves and an exec-mask region per step)
  const uint32_t o_plat = (live && par == 0u) ? woff : dead, o_plat1 = (live && par == 0u) ? row_bytes + woff : dead;
  const uint32_t o_plat2 = (live && par == 1u) ? 2u * row_bytes + woff : dead, o_plat3 = (live && par == 1u) ? 3u * row_bytes + woff : dead;
  const uint32_t o_joint = live ? (NL == 4 ? (4u + par) * row_bytes + woff       // n = 8: group `par` is this lane's four cables
                                           : 4u * row_bytes + woff + par * 8u)   // n = 4: this lane's half of the one group
                                : dead;
  const uint32_t o_joint_v = live ? o_joint + (uint32_t)G * row_bytes : dead, o_joint_e = live ? o_joint + 2u * (uint32_t)G * row_bytes : dead;
  const float4* obs_base = a.obs;
  int remaining = a.nsteps, since = 0;

  auto body = [&](auto slot_c) __attribute__((always_inline)) {
    constexpr int S = decltype(slot_c)::value;
    if (since == refresh) {  // the next Joy batch of the schedule (fetched a refresh period ago), and the fetch of the one after
      since = 0;
#pragma unroll
      for (int k = 0; k < NPL; ++k) desired[k] = next_joy[k];
      ++batch_next;
      if (batch_next < nbatches) load_joy(joy0 + (size_t)batch_next * a.sched_stride, next_joy);
    }
    ++since;
    Platform s;
    s.px = s_px; s.py = s_py; s.pz = s_pz; s.qx = s_qx; s.qy = s_qy; s.qz = s_qz; s.qw = s_qw;
    s.vx = s_vx; s.vy = s_vy; s.vz = s_vz; s.wx = s_wx; s.wy = s_wy; s.wz = s_wz;
    // ---- IK rows of this lane's cables on the state at t_k
    v2f len[NPL], jac[NPL][6], l0[NPL], q[NPL], qd[NPL];
    ik_rows<NPL, false, true>(mylds, s.px, s.py, s.pz, s.qx, s.qy, s.qz, s.qw, len, jac, l0);
#pragma unroll
    for (int k = 0; k < NPL; ++k) {
      q[k] = l0[k] - len[k];
      qd[k] = -fma2(s.wz, jac[k][5], fma2(s.wy, jac[k][4], fma2(s.wx, jac[k][3],
                    fma2(s.vz, jac[k][2], fma2(s.vy, jac[k][1], splat(s.vx) * jac[k][0])))));
    }
    // ---- per-cable force (PLG.cpp:222-228 -> JFC.cpp:59-96 -> Pid.cpp:122-191), windows full
    v2f f[NPL];
#pragma unroll
    for (int k = 0; k < NPL; ++k) {
      const v2f error = desired[k] - (VEL ? qd[k] : q[k]);
      const v2f acc = stream_fir<S>(wage, win[k], splat(wage[0]) * error, std::make_integer_sequence<int, kWin>{});
      const v2f p_term = splat(kp) * error;
      const v2f prev_ierr = ierr[k];
      v2f ie = fma2(dt_v, error, prev_ierr);
      const v2f i_term = splat(ki) * ie;
      const v2f i_cl = clamp2_med3(i_term, imin, imax);
      const v2f ie_cl = i_cl * splat(inv_ki);
      ie.x = (i_cl.x != i_term.x) ? ie_cl.x : ie.x;
      ie.y = (i_cl.y != i_term.y) ? ie_cl.y : ie.y;
      const v2f derived = acc * splat(inv_dt);
      const v2f d_term = splat(kd) * derived;
      const v2f cmd = fma2(kf, desired[k], p_term) + i_cl + d_term;
      v2f out = clamp2_med3(cmd, cmin, cmax);
      const v2f bumped = fma2(splat(dt_v) * error, splat(ki), out);
      ie.x = (out.x != cmd.x) ? prev_ierr.x : ie.x;
      ie.y = (out.y != cmd.y) ? prev_ierr.y : ie.y;
      out.x = (out.x != cmd.x) ? bumped.x : out.x;
      out.y = (out.y != cmd.y) ? bumped.y : out.y;
      ierr[k] = ie;
      f[k] = out;
      win[k][S] = error;  // the ring position is this copy's constant: a register assignment
    }
    v2f applied[NPL];
#pragma unroll
    for (int k = 0; k < NPL; ++k) applied[k] = clamp2_med3(f[k], neg_effort, effort);  // Joint::SetForce clamp (cube.sdf:438)

    // ---- observables of step t_k (every step of such a launch is published)
    {
      const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float4*>(obs_base), 0, (int)image_bytes, 0x00020000);
      auto row = [](float x, float y, float z, float w) { return (u32x4){fbits(x), fbits(y), fbits(z), fbits(w)}; };
      __builtin_amdgcn_raw_buffer_store_b128(row(s.px, s.py, s.pz, s.qx), rsrc, o_plat, 0, CDPR_STORE_AUX);
      __builtin_amdgcn_raw_buffer_store_b128(row(s.qy, s.qz, s.qw, s.vx), rsrc, o_plat1, 0, CDPR_STORE_AUX);
      __builtin_amdgcn_raw_buffer_store_b128(row(s.vy, s.vz, s.wx, s.wy), rsrc, o_plat2, 0, CDPR_STORE_AUX);
      __builtin_amdgcn_raw_buffer_store_b128(row(s.wz, 0.f, 0.f, 0.f), rsrc, o_plat3, 0, CDPR_STORE_AUX);
      if (NL == 4) {
        const u32x4 dq = {fbits(q[0].x), fbits(q[0].y), fbits(q[NPL - 1].x), fbits(q[NPL - 1].y)};
        const u32x4 dv = {fbits(qd[0].x), fbits(qd[0].y), fbits(qd[NPL - 1].x), fbits(qd[NPL - 1].y)};
        const u32x4 de = {fbits(applied[0].x), fbits(applied[0].y), fbits(applied[NPL - 1].x), fbits(applied[NPL - 1].y)};
        __builtin_amdgcn_raw_buffer_store_b128(dq, rsrc, o_joint, 0, CDPR_STORE_AUX);
        __builtin_amdgcn_raw_buffer_store_b128(dv, rsrc, o_joint_v, 0, CDPR_STORE_AUX);
        __builtin_amdgcn_raw_buffer_store_b128(de, rsrc, o_joint_e, 0, CDPR_STORE_AUX);
      } else {
        typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
        const u32x2 dq = {fbits(q[0].x), fbits(q[0].y)};
        const u32x2 dv = {fbits(qd[0].x), fbits(qd[0].y)};
        const u32x2 de = {fbits(applied[0].x), fbits(applied[0].y)};
        __builtin_amdgcn_raw_buffer_store_b64(dq, rsrc, o_joint, 0, CDPR_STORE_AUX);
        __builtin_amdgcn_raw_buffer_store_b64(dv, rsrc, o_joint_v, 0, CDPR_STORE_AUX);
        __builtin_amdgcn_raw_buffer_store_b64(de, rsrc, o_joint_e, 0, CDPR_STORE_AUX);
      }
      obs_base += a.obs_step_stride;
    }

    // ---- world step: wrench partial sums meet through the DPP add, both lanes integrate identically
    v2f tens[NPL];
#pragma unroll
    for (int k = 0; k < NPL; ++k) tens[k] = fma2(neg_damping, qd[k], applied[k]);
    float w[6];
    jt_times_shared<NPL>(jac, tens, w);
    w[0] = a.fgx - w[0];
    w[1] = a.fgy - w[1];
    w[2] = a.fgz - w[2];
    w[3] = -w[3];
    w[4] = -w[4];
    w[5] = -w[5];
    integrate(a, s, w);
    s_px = s.px; s_py = s.py; s_pz = s.pz; s_qx = s.qx; s_qy = s.qy; s_qz = s.qz; s_qw = s.qw;
    s_vx = s.vx; s_vy = s.vy; s_vz = s.vz; s_wx = s.wx; s_wy = s.wy; s_wz = s.wz;
    --remaining;
  };

  // Duff's device over the ring position: enter at the launch's first position, then whole turns of the ring
  int entry = a.ring_slot;
  while (remaining > 0) {
    switch (entry) {
#define CDPR_STREAM_CASE(S)                              \
  case S:                                                \
    body(std::integral_constant<int, S>{});              \
    if (remaining == 0) break;                           \
    [[fallthrough]];
      CDPR_STREAM_CASE(0)
      CDPR_STREAM_CASE(1)
      CDPR_STREAM_CASE(2)
      CDPR_STREAM_CASE(3)
      CDPR_STREAM_CASE(4)
      CDPR_STREAM_CASE(5)
      CDPR_STREAM_CASE(6)
      CDPR_STREAM_CASE(7)
      CDPR_STREAM_CASE(8)
      default:
        body(std::integral_constant<int, 9>{});
#undef CDPR_STREAM_CASE
    }
    entry = 0;
  }

  // ---- store (platform rows split between the two lanes; the whole window; the integrals)
  const float oi_final0 = partner(ierr[0].x), oi_final1 = partner(ierr[0].y);
  if (live) {
    if (par == 0u) {
      store_slot(a.state, st, 0, woff, make_float4(s_px, s_py, s_pz, s_qx));
      store_slot(a.state, st, 1, woff, make_float4(s_qy, s_qz, s_qw, s_vx));
    } else {
      store_slot(a.state, st, 2, woff, make_float4(s_vy, s_vz, s_wx, s_wy));
      store_slot(a.state, st, 3, woff, make_float4(s_wz, fkx, fky, fkz));
    }
    float4* wrow = a.state + (size_t)(P + 5 * k0) * st + r;
#pragma unroll
    for (int k = 0; k < NPL; ++k) {
#pragma unroll
      for (int m = 0; m < 5; ++m) wrow[(size_t)(5 * k + m) * st] = ring_row(win[k], m, splat(0.f), -1);
    }
    if (NPL == 2) {
      (a.state + (size_t)(P + 5 * NP + par) * st)[r] = make_float4(ierr[0].x, ierr[0].y, ierr[NPL - 1].x, ierr[NPL - 1].y);
    } else if (par == 0u) {
      (a.state + (size_t)(P + 5 * NP) * st)[r] = make_float4(ierr[0].x, ierr[0].y, oi_final0, oi_final1);
    }
  }
}

}  // namespace cdpr
