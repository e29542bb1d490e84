// cdpr_step_kernel_cable.hpp — the step kernel with ONE LANE PER CABLE (gfx950, fp32): the mapping BASELINE.json's
// north-star sketches ("J staged ..., wavefront shuffles for the per-robot 6-DoF reductions"), built to be measured against
// the lane-per-robot and lane-pair mappings (DESIGN.md section 4 has the table).
//
// A robot owns a group of GS adjacent lanes (GS = 8 for 5..8 cables, 4 for 1..4), so a wavefront carries 64 / GS robots
// and a batch becomes GS times as many wavefronts as under the lane-per-robot mapping.  Lane `sub` of a group owns
// cable `sub`: its geometry, its IK row (one row of the structure matrix J: 6 registers, never staged anywhere), its Pid
// record (ring of 10 errors + integral: 11 registers), its force.  What couples the cables of a robot goes through DPP
// cross-lane adds inside the group (xor 1, xor 2 by quad_perm, then row_half_mirror for the 8-lane group: 3 instructions
// per reduced value, every lane ends up with the sum):
//     J^T J  (21 entries)  and  J^T r  (6)   per Newton iteration,   J^T (f - Tm), J^T T  once per step.
// The 6x6 Cholesky factorization and the two substitutions, and the platform's world step, run REDUNDANTLY in every lane
// of the group (all lanes hold the same platform state and the same reduced sums, bit for bit, so they stay in step
// without any exchange).  That is the price of this mapping: per robot, the serial 6x6 part is executed GS times
// (in one instruction, but the lanes could have carried GS robots), and nothing per-cable can be packed into
// v_pk_fma_f32 (a lane has one cable, not a pair).
//
// HBM layout: exactly the one of cdpr_step_kernel.hpp (the kernels are interchangeable on the same state): a lane
// reads the float4 ring rows of its cable PAIR and keeps its half of every row, and writes back ONE dword per step (the
// ring slot of its own cable) plus one dword of the integral row; the platform rows are read by every lane of the group
// (same address: one request) and written by lanes 0..4, one row each; observables likewise, joint rows by dword.
//
// Arithmetic: same formulas as the other mappings, but the sums over cables are tree reductions here, so results agree
// with the other mappings to fp32 rounding, not bit for bit (the parity tests run at the usual tolerances).
#pragma once
#include "cdpr_step_kernel.hpp"

namespace cdpr {

template <int GS>
CDPR_DEV float group_sum(float x) {
  x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0xB1, 0xF, 0xF, true));   // lane ^ 1
  x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x4E, 0xF, 0xF, true));   // lane ^ 2
  if (GS == 8) x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x141, 0xF, 0xF, true));  // 7 - lane: the other quad
  return x;
}
template <int GS>
CDPR_DEV float group_max(float x) {
  x = fmaxf(x, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0xB1, 0xF, 0xF, true)));
  x = fmaxf(x, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x4E, 0xF, 0xF, true)));
  if (GS == 8) x = fmaxf(x, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x141, 0xF, 0xF, true)));
  return x;
}

struct CableGeom {
  float ax, ay, az, bx, by, bz, l0, mask;
};

// One row of the structure matrix (gen:113-118): l = p + R b - a, L = |l|, u = l / L, J row = [u, (R b) x u]; a masked
// lane (cable index >= N) returns zeros.
CDPR_DEV void ik_row(const CableGeom& g, float px, float py, float pz, const Rot& r, float& len, float (&j)[6]) {
  const float rbx = fmaf(r.r02, g.bz, fmaf(r.r01, g.by, r.r00 * g.bx));
  const float rby = fmaf(r.r12, g.bz, fmaf(r.r11, g.by, r.r10 * g.bx));
  const float rbz = fmaf(r.r22, g.bz, fmaf(r.r21, g.by, r.r20 * g.bx));
  const float lx = (rbx - g.ax) + px, ly = (rby - g.ay) + py, lz = (rbz - g.az) + pz;
  const float l2 = fmaf(lz, lz, fmaf(ly, ly, lx * lx));
  const float inv = __frsqrt_rn(l2);
  len = l2 * inv * g.mask;
  const float ux = lx * inv, uy = ly * inv, uz = lz * inv;
  j[0] = ux * g.mask;
  j[1] = uy * g.mask;
  j[2] = uz * g.mask;
  j[3] = fmaf(rby, uz, -(rbz * uy)) * g.mask;
  j[4] = fmaf(rbz, ux, -(rbx * uz)) * g.mask;
  j[5] = fmaf(rbx, uy, -(rby * ux)) * g.mask;
}

// (J^T J + lambda I) x = J^T v over the group's cables, solved in every lane (v -> x in g)
template <int GS, bool LAMBDA>
CDPR_DEV void group_normal_solve(const float (&j)[6], float v, float lambda, float (&g)[6]) {
  float m[6][6];
#pragma unroll
  for (int a = 0; a < 6; ++a) {
#pragma unroll
    for (int b = 0; b <= a; ++b) m[a][b] = group_sum<GS>(j[a] * j[b]) + ((LAMBDA && a == b) ? lambda : 0.f);
  }
#pragma unroll
  for (int c = 0; c < 6; ++c) g[c] = group_sum<GS>(j[c] * v);
  chol_solve(m, g);
}

template <int N, bool FK, bool TD>
__global__ __launch_bounds__(64, 2) void cdpr_step_kernel_cable(const StepArgs a) {
  constexpr int GS = (N <= 4) ? 4 : 8;
  constexpr int RPW = 64 / GS;  // robots per wavefront
  constexpr int NP = cable_pairs(N);
  constexpr int P = plat_slots(FK);
  constexpr int G = joint_groups(N);

  const uint32_t lane = threadIdx.x;
  const uint32_t sub = lane % GS;  // cable of this lane
  const uint32_t r = blockIdx.x * RPW + lane / GS;
  const uint32_t rr = (r < a.batch) ? r : (a.batch - 1u);  // tail groups shadow the last robot, stores are masked
  const bool live = r < a.batch;
  const bool cable_live = sub < (uint32_t)N;
  const uint32_t cab = cable_live ? sub : 0u;  // masked lanes shadow cable 0 (mask 0: they contribute nothing)
  const uint32_t k = cab / 2u, h = cab & 1u;   // cable pair and half, as the HBM layout counts
  const size_t st = a.stride;

  // ---- loads: cable constants (pair-interleaved table, see geom_pairs in cdpr_engine.hip), platform rows, Pid record
  CableGeom geo;
  {
    const float* gp = a.geom + k * kGeomFloatsPerPair + h;
    geo.ax = gp[0]; geo.ay = gp[2]; geo.az = gp[4];
    geo.bx = gp[6]; geo.by = gp[8]; geo.bz = gp[10];
    geo.l0 = gp[12];
    geo.mask = cable_live ? 1.f : 0.f;
  }
  const float4 p0 = a.state[0 * st + rr], p1 = a.state[1 * st + rr], p2 = a.state[2 * st + rr], p3 = a.state[3 * st + rr];
  float4 p4 = make_float4(0.f, 0.f, 0.f, 1.f);
  if (FK) p4 = a.state[4 * st + rr];
  float win[kWin];
#pragma unroll
  for (int m = 0; m < 5; ++m) {
    const float4 w = a.state[(size_t)(P + 5 * k + m) * st + rr];
    win[2 * m] = h ? w.y : w.x;
    win[2 * m + 1] = h ? w.w : w.z;
  }
  float ierr;
  {
    const float4 hr = a.state[(size_t)(P + 5 * NP + k / 2u) * st + rr];
    const uint32_t c = (k & 1u) * 2u + h;
    ierr = c == 0u ? hr.x : (c == 1u ? hr.y : (c == 2u ? hr.z : hr.w));
  }
  const float desired = a.cmd[(size_t)rr * N + cab];  // never null: before the first Joy the latched buffer holds zeros

  Platform s;
  s.px = p0.x; s.py = p0.y; s.pz = p0.z; s.qx = p0.w;
  s.qy = p1.x; s.qz = p1.y; s.qw = p1.z; s.vx = p1.w;
  s.vy = p2.x; s.vz = p2.y; s.wx = p2.z; s.wy = p2.w;
  s.wz = p3.x;
  float fkx = p3.y, fky = p3.z, fkz = p3.w, fkqx = p4.x, fkqy = p4.y, fkqz = p4.z, fkqw = p4.w;
  const bool actual_is_vel = (a.flags & kFlagActualIsVelocity) != 0u;
  int calls = a.pid_calls;

  for (int step = 0; step < a.nsteps; ++step) {
    // ---- IK row of this cable on the state at t_k
    float len, jr[6];
    ik_row(geo, s.px, s.py, s.pz, quat_to_rot(s.qx, s.qy, s.qz, s.qw), len, jr);
    const float q = (geo.l0 - len) * geo.mask;
    const float qd = -fmaf(s.wz, jr[5], fmaf(s.wy, jr[4], fmaf(s.wx, jr[3], fmaf(s.vz, jr[2], fmaf(s.vy, jr[1], s.vx * jr[0])))));
    const bool publish = ((a.publish_mask >> step) & 1ull) != 0ull;
    float4* const obs = a.obs + (size_t)step * a.obs_step_stride;

    // ---- per-cable force (PLG.cpp:222-228 -> JFC.cpp:59-96 -> Pid.cpp:122-191)
    float f = 0.f;
    float dbg_p = 0.f, dbg_i = 0.f, dbg_d = 0.f;
    bool dbg_wrote = false;
    const bool first_world = (step == 0) && (a.flags & kFlagFirstWorldStep);
    int ring_slot = -1;
    float e_new = 0.f;
    if (!first_world && (a.flags & kFlagForceMode)) {
      f = desired;  // UpdateMode::Force (JFC.cpp:67-70): no Pid
    } else if (!first_world) {
      if (calls != 0) {  // not the first call since reset (Pid.cpp:123-126: that one returns 0)
        const bool full = calls >= a.nbuf;
        ring_slot = (a.ring_slot + step) % kWin;
        const float* wt = a.wtab + ring_slot * (kWin + 2);
        const float error = desired - (actual_is_vel ? qd : q);
        float acc = wt[kWin] * error;
#pragma unroll
        for (int j = 0; j < kWin; ++j) acc = fmaf(wt[j], win[j], acc);
        const float p_term = a.kp * error;
        const float prev_ierr = ierr;
        float ie = fmaf(a.dt, error, prev_ierr);
        const float i_term = a.ki * ie;
        const float i_cl = fmaxf(fminf(i_term, a.imax), a.imin);  // Pid.cpp:143-152
        ie = (i_cl != i_term) ? i_cl * a.inv_ki : ie;
        const float derived = full ? acc * a.inv_dt : 0.f;
        const float d_term = a.kd * derived;
        const float cmd = fmaf(a.kf, desired, p_term) + i_cl + d_term;
        float out = a.clamp_cmd ? fmaxf(fminf(cmd, a.cmax), a.cmin) : cmd;  // Pid.cpp:175-177
        const float bumped = fmaf(a.dt * error, a.ki, out);                   // Pid.cpp:181-184
        ie = (out != cmd) ? prev_ierr : ie;
        out = (out != cmd) ? bumped : out;
        ierr = ie;
        f = out;
        e_new = error;
        dbg_p = p_term;
        dbg_i = i_term;
        dbg_d = d_term;
        dbg_wrote = true;
#pragma unroll
        for (int j = 0; j < kWin; ++j) win[j] = (j == ring_slot) ? e_new : win[j];
      }
      ++calls;
    }
    f *= geo.mask;

    // ---- Newton-Raphson forward kinematics ([NEW] SURVEY 8(a) row 14): measured length of this cable = len
    float fk_res = 0.f;
    int fk_it = 0, td_flag = 0;
    float jest[6];
    if (FK) {
      float elen;
      bool active = true;
      for (int it = 0; it < a.fk_iters; ++it) {
        ik_row(geo, fkx, fky, fkz, quat_to_rot(fkqx, fkqy, fkqz, fkqw), elen, jest);
        const float res = len - elen;
        active = active && !(group_max<GS>(fabsf(res)) < a.fk_tol);
        float g[6];
        group_normal_solve<GS, true>(jest, res, a.fk_lambda, g);
        if (active) {
          fkx += g[0];
          fky += g[1];
          fkz += g[2];
          quat_apply_rotvec(fkqx, fkqy, fkqz, fkqw, g[3], g[4], g[5]);
          ++fk_it;
        }
      }
      ik_row(geo, fkx, fky, fkz, quat_to_rot(fkqx, fkqy, fkqz, fkqw), elen, jest);
      fk_res = group_max<GS>(fabsf(len - elen));
    }

    // ---- tension distribution ([NEW] SURVEY 8(a) row 15), SetForce limits
    float applied = f;
    if (TD) {
      const float df = (f - a.td_mid) * geo.mask;
      float g[6];
      if (FK)
        group_normal_solve<GS, false>(jest, df, 0.f, g);
      else
        group_normal_solve<GS, false>(jr, df, 0.f, g);
      float t = a.td_mid;
#pragma unroll
      for (int c = 0; c < 6; ++c) t = fmaf(g[c], FK ? jest[c] : jr[c], t);
      const float tc = fmaxf(fminf(t, a.td_max), a.td_min);
      td_flag = (group_max<GS>((cable_live && tc != t) ? 1.f : 0.f) != 0.f) ? 1 : 0;
      applied = tc * geo.mask;
    }
    if (a.vel_limit > 0.f)  // Joint::SetForce velocity truncation [EXT]
      applied = ((qd > a.vel_limit && applied > 0.f) || (qd < -a.vel_limit && applied < 0.f)) ? 0.f : applied;
    if (a.effort >= 0.f) applied = fmaxf(fminf(applied, a.effort), -a.effort);  // Joint::SetForce clamp (cube.sdf:438)

    if (a.dbg && live && sub == 0u) {  // `pid` topic, cable 0 only (PLG.cpp:223-227; Pid.cpp:139-142,158-168)
      float* d = a.dbg + (size_t)r * 9;
      if (dbg_wrote) {
        d[0] = dbg_p;
        d[1] = dbg_i;
        d[2] = dbg_d;
        d[3] = desired;
      }
      d[4] = applied;
    }

    // ---- observables of step t_k (PLG.cpp:236-242, 248-280): platform rows by lanes 0..3, joint rows by dword
    if (publish && live) {
      uint32_t lim = 0u;
      if (a.travel_on) {
        const float outside = (cable_live && (q < a.travel_lo || q > a.travel_hi)) ? (float)(1u << cab) : 0.f;
        lim = (uint32_t)group_sum<GS>(outside);  // distinct powers of two: the sum is the mask
      }
      if (sub == 0u) obs[0 * st + r] = make_float4(s.px, s.py, s.pz, s.qx);
      if (sub == 1u) obs[1 * st + r] = make_float4(s.qy, s.qz, s.qw, s.vx);
      if (sub == 2u) obs[2 * st + r] = make_float4(s.vy, s.vz, s.wx, s.wy);
      if (sub == 3u) obs[3 * st + r] = make_float4(s.wz, fk_res, (float)fk_it, pack_flags(td_flag, lim));
      if (cable_live) {
        float* jq = reinterpret_cast<float*>(obs + (size_t)(4 + cab / 4u) * st + r) + (cab & 3u);
        float* jqd = reinterpret_cast<float*>(obs + (size_t)(4 + G + cab / 4u) * st + r) + (cab & 3u);
        float* je = reinterpret_cast<float*>(obs + (size_t)(4 + 2 * G + cab / 4u) * st + r) + (cab & 3u);
        *jq = q;
        *jqd = qd;
        *je = applied;
      } else if (sub < (uint32_t)(4 * G)) {  // padding components of the last joint row read 0, as in the other mappings
        reinterpret_cast<float*>(obs + (size_t)(4 + sub / 4u) * st + r)[sub & 3u] = 0.f;
        reinterpret_cast<float*>(obs + (size_t)(4 + G + sub / 4u) * st + r)[sub & 3u] = 0.f;
        reinterpret_cast<float*>(obs + (size_t)(4 + 2 * G + sub / 4u) * st + r)[sub & 3u] = 0.f;
      }
    }

    // ---- controller record of this step: one dword of the ring row, one of the integral row
    if (live && cable_live && ring_slot >= 0) {
      float* wrow = reinterpret_cast<float*>(a.state + (size_t)(P + 5 * k + (uint32_t)(ring_slot >> 1)) * st + r);
      wrow[(uint32_t)(ring_slot & 1) * 2u + h] = e_new;
    }

    // ---- world step to t_{k+1}: wrench = -J^T (applied - d qdot) + m g, in every lane of the group
    {
      float tens = fmaf(-a.damping, qd, applied);
      if (a.unilateral) tens = fmaxf(tens, 0.f);
      float w[6];
#pragma unroll
      for (int c = 0; c < 6; ++c) w[c] = group_sum<GS>(jr[c] * tens);
      w[0] = a.fgx - w[0];
      w[1] = a.fgy - w[1];
      w[2] = a.fgz - w[2];
      w[3] = -w[3];
      w[4] = -w[4];
      w[5] = -w[5];
      integrate(a, s, w);
    }
  }

  // ---- state: platform rows by lanes 0..4, the integral by dword (ring slots went out step by step)
  if (live) {
    if (sub == 0u) a.state[0 * st + r] = make_float4(s.px, s.py, s.pz, s.qx);
    if (sub == 1u) a.state[1 * st + r] = make_float4(s.qy, s.qz, s.qw, s.vx);
    if (sub == 2u) a.state[2 * st + r] = make_float4(s.vy, s.vz, s.wx, s.wy);
    if (sub == 3u) a.state[3 * st + r] = make_float4(s.wz, fkx, fky, fkz);
    if (FK && sub == (GS == 8 ? 4u : 0u)) a.state[4 * st + r] = make_float4(fkqx, fkqy, fkqz, fkqw);
    if (cable_live) {
      float* hrow = reinterpret_cast<float*>(a.state + (size_t)(P + 5 * NP + k / 2u) * st + r);
      hrow[(k & 1u) * 2u + h] = ierr;
    }
  }
}

}  // namespace cdpr
