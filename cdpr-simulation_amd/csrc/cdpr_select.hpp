// cdpr_select.hpp — which kernel serves which handle and launch: the routing rules of CDPR_MAP_AUTO as PURE functions of the
// configuration (plus the CU count and the A/B environment overrides), with no HIP call in them.
//
// Through round 5 these rules lived inside cdpr_create and select_step_kernel (cdpr_engine.hip): what AUTO picks for a given
// (cables, stages, batch, controller features, precision) could only be observed by running it.  Here they are one unit that
// the engine calls (cdpr_create copies the plan into the handle; every launch asks planned_kernel for its kernel) and that
// the C-ABI exposes without a GPU (cdpr_plan_kernel, include/cdpr.h): tests/test_kernel_selection.py enumerates the matrix
// on the CPU and pins the kernel name for every cell (tests/golden/kernel_selection.json).
//
// Every threshold below is a measured crossover on MI355X; the measurement is named where the number stands.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <string>

#include "../../include/cdpr.h"

namespace cdpr {

constexpr int kSelWin = 10;  // = kWin (cdpr_step_kernel.hpp): prior errors kept per cable on the register-resident path

inline bool mat3_inverse_sym(const double in[6], double out[6]) {
  // in / out: xx yy zz xy xz yz
  const double m[9] = {in[0], in[3], in[4], in[3], in[1], in[5], in[4], in[5], in[2]};
  const double det = m[0] * (m[4] * m[8] - m[5] * m[7]) - m[1] * (m[3] * m[8] - m[5] * m[6]) + m[2] * (m[3] * m[7] - m[4] * m[6]);
  if (!(std::fabs(det) > 1e-300)) return false;
  out[0] = (m[4] * m[8] - m[5] * m[7]) / det;
  out[1] = (m[0] * m[8] - m[2] * m[6]) / det;
  out[2] = (m[0] * m[4] - m[1] * m[3]) / det;
  out[3] = (m[2] * m[7] - m[1] * m[8]) / det;
  out[4] = (m[1] * m[5] - m[2] * m[4]) / det;
  out[5] = (m[1] * m[6] - m[0] * m[7]) / det;
  return true;
}

// Returns "" when the configuration is acceptable, else the reason.
inline std::string validate_config(const cdpr_config_t& c) {
  char buf[256];
  if (c.abi_version != CDPR_ABI_VERSION) return "abi_version mismatch";
  if (c.n_cables < 1 || c.n_cables > CDPR_MAX_CABLES) {
    snprintf(buf, sizeof buf, "invalid joint count %u (PLG.cpp:167-168; engine takes 1..%u)", c.n_cables, CDPR_MAX_CABLES);
    return buf;
  }
  if (c.batch < 1 || c.batch > (1ull << 27)) return "batch out of range (1 .. 2^27 robots per handle)";
  if (!(c.dt > 0.0)) return "dt must be > 0";
  if (!(c.mass > 0.0)) return "mass must be > 0";
  if (c.passive_damping < 0.0 || c.leg_inertia < 0.0 || c.cable_axial_mass < 0.0 || c.anchor_point_mass < 0.0 || c.anchor_inertia < 0.0)
    return "lumped-leg terms (passive_damping, leg_inertia, cable_axial_mass, anchor_point_mass, anchor_inertia) must be >= 0";
  double inv[6];
  if (!mat3_inverse_sym(c.inertia, inv)) return "inertia is singular";
  for (uint32_t i = 0; i < c.n_cables; ++i)
    if (!(c.cable_ref_length[i] > 0.0)) return "cable_ref_length must be > 0";
  const cdpr_pid_params_t* pids[2] = {&c.velocity_pid, &c.position_pid};
  for (auto* p : pids) {
    if (p->d_buffer_length < 2 || p->d_buffer_length > CDPR_MAX_D_BUFFER) return "d_buffer_length out of range";
    if (p->d_degree < 1 || p->d_degree > CDPR_MAX_D_DEGREE || p->d_degree >= p->d_buffer_length) return "d_degree out of range";
    if (p->p_filter.cascade > CDPR_MAX_CASCADE || p->d_filter.cascade > CDPR_MAX_CASCADE) return "filter cascade out of range";
  }
  if (c.precision != 0 && c.precision != 32 && c.precision != 64) return "precision must be 32 (or 0) or 64";
  if (c.travel_lower > c.travel_upper) return "travel_lower must not exceed travel_upper";
  if (c.travel_stop && !(c.travel_lower < c.travel_upper)) return "travel_stop needs travel limits (travel_lower < travel_upper)";
  if (c.travel_stop > 64) return "travel_stop (sweeps of the joint stop) must be <= 64";
  if ((c.stages & (CDPR_STAGE_FK | CDPR_STAGE_TD)) && c.n_cables < 6) return "FK / tension distribution need >= 6 cables";
  if ((c.stages & CDPR_STAGE_FK) && (c.fk_max_iterations < 1 || c.fk_max_iterations > 64)) return "fk_max_iterations out of range";
  if ((c.stages & CDPR_STAGE_TD) && !(c.td_f_max > c.td_f_min)) return "td_f_max must exceed td_f_min";
  if (c.mapping > CDPR_MAP_LANE_PER_CABLE) return "unknown mapping";
  if (c.mapping == CDPR_MAP_LANE_PAIR && c.n_cables != 4 && c.n_cables != 8) return "the lane-pair mapping needs 4 or 8 cables";
  return "";
}

// What the register-resident fast path cannot represent (it keeps ONE Pid record per
// cable: the active mode's): the position-hold branch (JFC.cpp:78-82) keeps both PIDs
// alive in Velocity mode, filters add state, long windows do not fit the record.
inline std::string fast_path_obstacle(const cdpr_config_t& c) {
  if (!(c.velocity_epsilon < 0.0)) return "velocity_epsilon >= 0 (position-hold branch live)";
  const cdpr_pid_params_t* pids[2] = {&c.velocity_pid, &c.position_pid};
  for (auto* p : pids) {
    if (p->p_filter.cascade || p->d_filter.cascade) return "biquad cascades enabled";
    if (p->d_buffer_length > (uint32_t)kSelWin + 1) return "derivative window longer than 11 samples";
    if (!(std::fabs(p->cmd_limit) > 0.0)) return "cmd_limit == 0 (command clamp disabled, Pid.cpp:175)";
  }
  return "";
}

// What cdpr_create decides from the configuration alone.
struct KernelPlan {
  int rc = CDPR_OK;      // CDPR_OK, or why no handle can be made (error says it)
  std::string error;
  uint32_t n = 0, batch = 0;
  bool fk = false, td = false, per_robot = false;
  bool phys = false;       // lumped legs / joint stop: the PHYS instantiations
  bool general = false;    // fp32 general controller path (hold branch, cascades, long windows, cmd_limit 0, ...)
  bool fp64 = false, hold64 = false, tstop64 = false;
  bool long64 = false;     // precision = 64 with derivative windows of 12 .. 32 samples: the one-wave kernel with a ring of 31
  bool lane_pair = false, lane_cable = false;
  bool lowreg = false, persist = false, onestep_v2 = false, split = false;
  bool gen_split = false, gen_lean = false, gen_hot = false;
  bool hold_full = false;  // fp64 HOLD = 2 instantiations (a cascade, or a Pid without the command clamp)
  bool hold_long = false;  // ... over Pid records of 32 samples: the hold branch / cascades / cmd_limit 0 with derivative windows of 12 .. 32 samples
  bool pair_stream = true;
  uint32_t chunk = 0;
  int gen_nb = 0;          // general path: longest derivative window of the two Pids
};

using EnvFn = const char* (*)(const char*);
inline const char* no_env(const char*) { return nullptr; }
inline const char* process_env(const char* key) { return std::getenv(key); }

inline KernelPlan plan_kernels(const cdpr_config_t& c, int cus = 256, EnvFn env = process_env) {
  KernelPlan p;
  const cdpr_config_t* cfg = &c;
  std::string why = validate_config(c);
  if (!why.empty()) {
    p.rc = CDPR_ERR_INVALID;
    p.error = why;
    return p;
  }
  p.n = c.n_cables;
  p.batch = (uint32_t)c.batch;
  // the PHYS instantiations carry the lumped legs and the joint stop
  const bool phys_cfg = cfg->passive_damping != 0.0 || cfg->leg_inertia != 0.0 || cfg->cable_axial_mass != 0.0 || cfg->anchor_point_mass != 0.0 ||
                        cfg->anchor_inertia != 0.0 || cfg->travel_stop != 0;
  // per-robot commands run on the register-resident kernels too (PR instantiations); only what those cannot represent
  // (hold branch, cascades, long windows, cmdLimit 0), and per-robot modes combined with the lumped-leg physics or with
  // two Pids that fit different derivative windows, take the general controller path
  const bool pr_windows_differ = cfg->velocity_pid.d_buffer_length != cfg->position_pid.d_buffer_length || cfg->velocity_pid.d_degree != cfg->position_pid.d_degree;
  const bool general_cfg = !fast_path_obstacle(*cfg).empty() || (cfg->per_robot_commands != 0 && (phys_cfg || pr_windows_differ));
  // The per-robot kernels of the register-resident path do not clear a reset Pid's derivative ring (the latch zeroes only
  // the integral rows) and a velocity rollout keeps stale position-Pid errors in it: that is correct only while
  // full = calls >= nbuf hides every stale slot, i.e. nbuf <= kWin + 1 and one window shared by both Pids.  Both follow
  // from the routing above; checked here so that a change to the routing cannot silently break the kernels' invariant.
  if (!general_cfg && cfg->per_robot_commands != 0 &&
      (pr_windows_differ || cfg->velocity_pid.d_buffer_length > (uint32_t)kSelWin + 1 || cfg->position_pid.d_buffer_length > (uint32_t)kSelWin + 1)) {
    p.rc = CDPR_ERR_UNSUPPORTED;
    p.error = "internal: per-robot handle routed to the register-resident path with windows it cannot hold";
    return p;
  }
  // precision = 64 with the hold branch as the ONLY thing the register-resident path cannot represent: the HOLD instantiations of the
  // fp64 kernel (uniform-mode handles; round 5)
  // The HOLD instantiations are the fp64 kernels' whole Pid::update: besides the hold branch they carry the biquad cascades and
  // cmd_limit = 0 (the Pid then returns its stale mCmd member plus the anti-windup increment, Pid.cpp:175-184: a row of its own).
  // What stays out: derivative windows beyond 11 samples.
  const bool windows_fit = cfg->velocity_pid.d_buffer_length <= (uint32_t)kSelWin + 1 && cfg->position_pid.d_buffer_length <= (uint32_t)kSelWin + 1;
  const bool clean64 = cfg->precision == 64 && windows_fit;
  // ... and with the optional physics - the joint stop, the lumped legs (round 6) - on uniform-mode handles without the hold branch:
  // the TSTOP instantiations
  const bool lumped_cfg = cfg->passive_damping != 0.0 || cfg->leg_inertia != 0.0 || cfg->cable_axial_mass != 0.0 || cfg->anchor_point_mass != 0.0 || cfg->anchor_inertia != 0.0;
  (void)lumped_cfg;
  // ... and derivative windows of 12 .. 32 samples as the ONLY thing beyond the register-resident path (round 6): the plain one-wave
  // fp64 kernel with a ring of 31 errors per cable (uniform-mode handles, reduced physics)
  // (later in round 6: with per-robot modes - both Pids on one window, as on every register-resident per-robot handle - and with the
  //  optional physics as well; not with the hold branch / cascades / cmd_limit 0: the HOLD records hold eleven samples)
  bool long64 = false;
  if (cfg->precision == 64 && !windows_fit && !(cfg->per_robot_commands != 0 && pr_windows_differ)) {
    cdpr_config_t probe = *cfg;
    probe.velocity_pid.d_buffer_length = probe.position_pid.d_buffer_length = 11;
    probe.velocity_pid.d_degree = probe.position_pid.d_degree = std::min(cfg->velocity_pid.d_degree, 4u);
    long64 = fast_path_obstacle(probe).empty();  // (no hold branch, no cascades, a command clamp)
  }
  // the HOLD instantiations: whatever else the register-resident path cannot represent (per-robot modes whose Pids fit different windows
  // too: each Pid has its own rows and its own window); later in round 6 with windows of 12 .. 32 samples as well (hold_long)
  const bool hold64 = cfg->precision == 64 && !long64 && (!fast_path_obstacle(*cfg).empty() || (cfg->per_robot_commands != 0 && pr_windows_differ));
  const bool hold_long = hold64 && !windows_fit;
  (void)clean64;
  const bool tstop64 = cfg->precision == 64 && phys_cfg;  // (round 6: the lumped legs too, with or without the stop; and together with
                                                                                     //  per-robot modes, the hold branch, long windows)
  if (cfg->precision == 64 && (general_cfg || phys_cfg) && !hold64 && !tstop64 && !long64) {
    p.rc = CDPR_ERR_UNSUPPORTED;
    p.error = "precision = 64 covers the controller (modes, per-robot arrival, hold branch, cascades, cmd_limit 0) and the optional physics (joint stop, lumped legs) "
              "with windows to 11 samples; windows to 32 samples with per-robot modes and the optional physics, not with the hold branch / cascades / cmd_limit 0: " +
              (general_cfg ? (fast_path_obstacle(*cfg).empty() ? std::string("per-robot modes whose two Pids fit different windows, one of them beyond 11 samples") : fast_path_obstacle(*cfg))
                           : std::string("optional physics"));
    return p;
  }
  const bool general = general_cfg && cfg->precision != 64;  // (a precision = 64 handle that got here runs on the fp64 kernels' own instantiations)
  // Nine to twelve cables (round 6; cube.yaml:21-29 is a free-length list of anchor points): the first-generation lane-per-robot
  // kernels carry them - one step or several per launch, FK and TD, trajectory record, schedules, the MPC rollout, the one-shot
  // solvers - on uniform-mode fp32 handles with the reduced physics.  Everything else is instantiated to eight cables.
  if (cfg->n_cables > 8u) {
    const char* why9 = general_cfg ? "the general controller path (hold branch, cascades, long windows, cmd_limit 0)"
                       : cfg->per_robot_commands != 0 ? "per_robot_commands"
                       : phys_cfg ? "the lumped legs / the joint stop"
                       : (cfg->mapping != CDPR_MAP_AUTO && cfg->mapping != CDPR_MAP_LANE_PER_ROBOT) ? "a mapping other than one lane per robot" : nullptr;
    if (why9) {
      p.rc = CDPR_ERR_UNSUPPORTED;
      p.error = std::string("more than 8 cables: not together with ") + why9 + " (those kernels are instantiated to 8 cables)";
      return p;
    }
  }
  p.fk = (cfg->stages & CDPR_STAGE_FK) != 0;
  p.td = (cfg->stages & CDPR_STAGE_TD) != 0;
  p.general = general;  // (precision = 64 with the hold branch: the fp64 kernel's HOLD instantiations, not the fp32 general path)
  p.fp64 = cfg->precision == 64;
  p.hold64 = hold64;
  p.tstop64 = tstop64;
  p.long64 = long64;
  p.per_robot = cfg->per_robot_commands != 0;
  p.phys = phys_cfg;
  if (hold64) {
    const bool any_cas = cfg->velocity_pid.p_filter.cascade || cfg->velocity_pid.d_filter.cascade || cfg->position_pid.p_filter.cascade || cfg->position_pid.d_filter.cascade;
    const bool any_noclamp = !(std::fabs(cfg->velocity_pid.cmd_limit) > 0.0) || !(std::fabs(cfg->position_pid.cmd_limit) > 0.0);
    p.hold_full = any_cas || any_noclamp || hold_long;  // (records of 32 samples: instantiated at HOLD = 2 only)
    p.hold_long = hold_long;
  }
  {
    // Mapping: measured on MI355X (scripts/ab_bench.py), two lanes per robot win while the batch leaves SIMDs
    // under-filled (16 384 x 8 cables: 9.0 vs 10.2 us/step; 4 096 x 4: 3.0 vs 3.4) and lose from 65 536 robots on
    // (16.4 vs 15.7 us/step: the duplicated 6x6 solves cost more than the second wave per SIMD hides), so AUTO
    // takes the pair mapping up to 32 768 robots at n = 8 (32 768: 9.5 vs 10.5 us/step; 49 152: 13.6 vs 11.2) and up
    // to 65 536 at n = 4 (65 536: 4.9 vs 5.2; 131 072: 8.5 vs 8.1).  CDPR_MAPPING=1|2 overrides AUTO (for A/B runs).
    const bool can_pair = !general && !p.phys && !p.per_robot && (cfg->n_cables == 4 || cfg->n_cables == 8);
    uint32_t mapping = cfg->mapping;
    if (mapping == CDPR_MAP_AUTO) {
      const char* mv = env("CDPR_MAPPING");
      if (mv && (mv[0] == '1' || mv[0] == '2' || mv[0] == '3')) mapping = (uint32_t)(mv[0] - '0');
    }
    // FK + TD handles: the role-split kernel (cdpr_split_kernel: two waves per 64 robots with different roles) beats both
    // mappings up to one robot per hardware lane (profiles/r02o_split_kernel_batch_scan.txt, us/step pair or one-wave ->
    // split: 4 096: 7.6 -> 7.2; 16 384: 8.4 -> 7.7; 32 768: 9.7 -> 9.1; 49 152: 10.8 -> 9.7; 65 536: 11.9 -> 10.9), so AUTO
    // keeps those on the lane-per-robot mapping; above ~90 000 robots the low-register kernel takes over (below)
    const bool split_case = !general && !p.phys && (cfg->stages & CDPR_STAGE_FK) && (cfg->stages & CDPR_STAGE_TD) && cfg->n_cables >= 6;
    if (mapping == CDPR_MAP_AUTO)
      mapping = (can_pair && !split_case && cfg->batch <= (cfg->n_cables == 4 ? 65536u : 32768u)) ? CDPR_MAP_LANE_PAIR : CDPR_MAP_LANE_PER_ROBOT;
    if (p.fp64 || cfg->n_cables > 8u) mapping = CDPR_MAP_LANE_PER_ROBOT;  // one plain kernel (the CDPR_MAPPING override does not reach more than 8 cables)
    // a mapping the CONFIGURATION asks for by name is served or refused; the CDPR_MAPPING environment override (A/B runs over
    // whole test suites) keeps falling back to one lane per robot where the requested mapping does not exist
    const bool can_cable = !general && !p.phys && !p.per_robot;
    if (!p.fp64 && ((cfg->mapping == CDPR_MAP_LANE_PAIR && !can_pair) || (cfg->mapping == CDPR_MAP_LANE_PER_CABLE && !can_cable))) {
      p.rc = CDPR_ERR_UNSUPPORTED;
      p.error = std::string(cfg->mapping == CDPR_MAP_LANE_PAIR ? "CDPR_MAP_LANE_PAIR" : "CDPR_MAP_LANE_PER_CABLE") +
                " is not available with the general controller path, the optional physics terms or per_robot_commands" +
                (cfg->mapping == CDPR_MAP_LANE_PAIR ? " (and needs 4 or 8 cables)" : "") + "; use CDPR_MAP_AUTO or CDPR_MAP_LANE_PER_ROBOT";
      return p;
    }
    p.lane_pair = (mapping == CDPR_MAP_LANE_PAIR) && can_pair;
    // one lane per cable: any cable count; not with the optional physics, per-robot modes or the general path (those
    // handles silently keep the lane-per-robot mapping, as the lane-pair request does where it cannot be served)
    p.lane_cable = (mapping == CDPR_MAP_LANE_PER_CABLE) && can_cable;
    // more robots than hardware lanes (65 536): two co-resident waves per SIMD pay, if the kernel fits twice.
    // Measured (scripts/ab_bench.py with CDPR_LOWREG=0|1, us/step without -> with): 65 536: 13.4 -> 13.8; 98 304: 26.0 -> 21.6;
    // 131 072: 29.9 -> 27.0; 196 608: 41.2 -> 36.3; 524 288: 89.9 -> 79.0 (6.6e9 state-steps/s)
    p.lowreg = !general && !p.phys && !p.lane_pair && !p.lane_cable && (cfg->stages & CDPR_STAGE_FK) && cfg->n_cables >= 6 && cfg->batch > 90112u;  // crossover measured: profiles/r03j_cliff_scan.txt
    if (const char* lr = env("CDPR_LOWREG"))
      p.lowreg = (lr[0] == '1') && !general && !p.phys && !p.lane_pair && !p.lane_cable && (cfg->stages & CDPR_STAGE_FK) && cfg->n_cables >= 6;
    // Between one and a few robots per hardware lane a single launch is a bulk-synchronous load -> compute -> store in which
    // the co-resident waves of a SIMD start together: their memory phases coincide and their compute phases coincide, so two
    // waves per SIMD cost 2.1-2.7x one (rocprofv3 PMC at 65 536 / 98 304 / 131 072 / 196 608: the vector pipes are busy 45 % of
    // the launch at one AND at two waves per SIMD, profiles/r04_cliff_analysis.txt).  Round 4 measured the two obvious
    // mitigations and keeps neither as a default: the same step as back-to-back launches over blocks of <= 65 536 robots
    // (CDPR_CHUNK=N; bit-identical, tested) is within +-4 % of the single launch at every size and 20-30 % slower from 262 144
    // robots on, where a large launch de-phases by itself (waves start as slots free up) and streams at the copy rate; delaying
    // the second wave slot's workgroups by 2-6 us (s_sleep) changes nothing.  What would: a persistent kernel that prefetches
    // the next block's rows while it computes.
    if (const char* ck = env("CDPR_CHUNK")) {  // A/B: 0 = never, N = blocks of at most N robots whatever the batch
      const long v = std::atol(ck);
      p.chunk = (v > 0 && !general && !p.fp64 && !p.lane_pair && !p.lane_cable) ? (uint32_t)((v + 63) & ~63L) : 0u;
    }
    if (p.chunk && p.chunk <= 90112u && !env("CDPR_LOWREG")) p.lowreg = false;  // every block runs in the role-split kernel's range
    // the persistent one-wave kernel (cdpr_onestep_kernel<..., PERSIST>): uniform-mode handles on the lane-per-robot mapping
    const bool can_persist = !general && !p.phys && !p.lane_pair && !p.lane_cable && cfg->per_robot_commands == 0 && cfg->precision != 64 && !p.chunk;
    p.persist = false;
    if (const char* ps = env("CDPR_PERSIST")) p.persist = (ps[0] == '1') && can_persist;
    if (p.persist) p.lowreg = false;
  }
  // second-generation one-step kernel: wins wherever there is a Newton stage to hide the controller rows under, and
  // without one from ~32 768 robots on (65 536 x 8, no FK: 6.3 vs 7.0 us/step); small batches without FK are pure
  // latency and the extra LDS round trip loses (4 096 x 4: 4.18 vs 3.99 us by rocprofv3)
  p.onestep_v2 = p.fk || cfg->batch > 32768u;
  if (const char* os = env("CDPR_ONESTEP")) p.onestep_v2 = (os[0] != '1');
  if (cfg->n_cables > 8u) p.onestep_v2 = p.lowreg = p.persist = false;  // (more than 8 cables: the first-generation kernel)
  p.split = (p.onestep_v2 || p.per_robot) && !general && !p.phys && !p.lane_pair && !p.lane_cable && !p.lowreg && !p.persist && p.fk && p.td && cfg->n_cables >= 6;
  if (const char* sp = env("CDPR_SPLIT")) p.split = p.split && sp[0] != '0';
  {
    const char* ps = env("CDPR_PAIR_STREAM");
    p.pair_stream = !(ps && ps[0] == '0');
  }
  if (p.general) {
    p.gen_nb = (int)std::max(cfg->velocity_pid.d_buffer_length, cfg->position_pid.d_buffer_length);
    // role-split one-step kernel: compiled for one workgroup per pair of SIMDs (each wave may use the whole register
    // file), so it serves batches up to two workgroups of 64 robots per CU; CDPR_GEN_SPLIT=0|1 overrides (A/B)
    if (cus <= 0) cus = 256;
    const bool can = p.fk && p.td && p.n >= 6 && p.gen_nb <= 11;
    p.gen_split = can && cfg->batch <= (uint32_t)cus * 128u;
    const char* gs = env("CDPR_GEN_SPLIT");
    if (gs) p.gen_split = can && gs[0] == '1';
    // beyond that: the lean role-split kernel (not with the optional physics: it carries none); CDPR_GEN_LEAN=0|1 overrides
    // (A/B; 1 also below the role-split kernel's limit).  An explicit CDPR_GEN_SPLIT=0 means "the one-wave kernel" (the A/B
    // scripts' meaning since round 4): it keeps the lean kernel off too unless CDPR_GEN_LEAN=1 asks for it.
    p.gen_lean = can && !p.phys && !p.gen_split && !(gs && gs[0] == '0');
    if (const char* gl = env("CDPR_GEN_LEAN")) {
      p.gen_lean = can && !p.phys && gl[0] == '1';
      if (p.gen_lean) p.gen_split = false;
    }
    // hot rows: where the lean kernel steps the handle (beyond 32 768 robots: 176 B per robot-step less traffic at the same
    // time per step; below, a workgroup per CU or less, the extra loads and the restore cost 0.5 us of a 9 us step);
    // they take effect on handles whose configuration admits the consecutive-call branches (GenCtl::simple_ok)
    p.gen_hot = p.gen_lean;
    if (const char* gh = env("CDPR_GEN_HOT")) p.gen_hot = p.gen_lean && gh[0] != '0';  // (the role-split kernel carries no code for them)
  }
  return p;
}

// ---- one launch --------------------------------------------------------------------------------------------------------
enum class KernelId {
  None,
  // register-resident path, one lane per robot (k_step.hip, k_onestep.hip, k_pr.hip)
  StepSingle, StepMulti, Lowreg, OnestepPersist, Split, Onestep, PhysStep, Rollout, PhysRollout,
  PrSingle, PrLowreg, PrSplit, PrMulti, PrRollout,
  // the other mappings (k_pair.hip, k_cable.hip)
  PairSingle, PairMulti, PairStream, Cable,
  // general controller path (k_gen*.hip)
  GenOne, GenMulti, GenSplit, GenLean, GenRollout,
  // precision = 64 (k_f64.hip)
  F64, F64Pr, F64Tstop, F64Long, F64Hold, F64HoldPr, F64Split, F64SplitHold,
};

struct LaunchShape {
  int steps = 1;             // world steps of the launch
  bool first_world = false;  // the launch starts at world step 0
  bool scheduled = false;    // cdpr_update_scheduled's in-launch form (always the several-steps kernel)
  bool rollout = false;      // cdpr_rollout_velocity*
  bool steady = true;        // Pid mode, windows full for the whole launch, every step published, no debug topic, travel flags, velocity
                             // limit, unilateral cables or mailbox, effort and command clamps on (what pair_stream_ok checks per launch)
  // precision = 64: the per-call A/B overrides (-1: none)
  int f64_ring_lds = -1, f64_jcache = -1, f64_split = -1;
};

struct PlannedKernel {
  KernelId id = KernelId::None;
  uint32_t block = 64;        // threads per workgroup
  uint32_t robots_per_block = 64;
  bool f64_ring_lds = false, f64_jcache = false, f64_lean = false;
};

inline PlannedKernel planned_kernel(const KernelPlan& p, const LaunchShape& s) {
  PlannedKernel k;
  const int steps = s.scheduled ? std::max(s.steps, 2) : s.steps;  // a launch over a schedule runs on the several-steps kernel even for one step
  if (p.fp64) {
    const bool ring_lds = p.n <= 8u && (s.f64_ring_lds >= 0 ? s.f64_ring_lds != 0 : p.batch <= 32768u);  // (nine to twelve cables: the plain one-wave kernel, rings in memory)
    const bool jcache = ring_lds && (s.f64_jcache >= 0 ? s.f64_jcache != 0 : p.batch <= 16384u);
    const bool lean = s.f64_split >= 0 ? s.f64_split == 2 : p.batch > 16384u;
    const bool can_split = p.fk && p.td && p.n <= 8u && s.f64_split != 0 && !p.per_robot && !p.tstop64 && !p.long64 && !p.hold_long;
    k.f64_ring_lds = ring_lds, k.f64_jcache = jcache, k.f64_lean = lean;
    // up to one workgroup per CU the role-split kernel's one-step launches beat the one-wave kernel's several-steps ones (14.4
    // against 20.8 us per step at one robot x 8, same bits): the engine then runs a fused update as one-step launches
    if (can_split && (steps == 1 || !lean)) {
      k.id = p.hold64 ? KernelId::F64SplitHold : KernelId::F64Split;
      k.block = 128;
      return k;
    }
    k.id = p.long64 ? KernelId::F64Long : p.tstop64 ? KernelId::F64Tstop : p.hold64 ? (p.per_robot ? KernelId::F64HoldPr : KernelId::F64Hold) : p.per_robot ? KernelId::F64Pr : KernelId::F64;
    return k;
  }
  if (p.general) {
    if (s.rollout) { k.id = KernelId::GenRollout; return k; }
    // where a role-split one-step kernel serves the handle its launches beat the one-wave kernel's several-steps ones (16 384 x 8:
    // 9.2 against 13.7 us per step, same bits): fused updates and the trajectory record then run as one-step launches
    const int gsteps = (p.gen_split || p.gen_lean) ? 1 : steps;
    if (p.gen_split) { k.id = KernelId::GenSplit; k.block = 128; return k; }
    if (p.gen_lean && !s.first_world) { k.id = KernelId::GenLean; k.block = 128; return k; }  // (world step 0 runs no controller: the one-wave kernel's case)
    k.id = gsteps == 1 ? KernelId::GenOne : KernelId::GenMulti;
    return k;
  }
  if (s.rollout) {
    k.id = p.per_robot ? KernelId::PrRollout : p.phys ? KernelId::PhysRollout : KernelId::Rollout;
    return k;
  }
  if (p.per_robot) {
    k.id = steps == 1 ? (p.lowreg ? KernelId::PrLowreg : p.split ? KernelId::PrSplit : KernelId::PrSingle) : KernelId::PrMulti;
  } else if (p.lane_cable) {
    k.id = KernelId::Cable;
    k.robots_per_block = p.n <= 4 ? 16u : 8u;
  } else if (p.phys) {
    k.id = KernelId::PhysStep;
  } else if (p.lane_pair) {
    const bool stream = p.pair_stream && !p.fk && !p.td && (p.n == 4 || p.n == 8) && steps > 1 && !s.first_world && s.steady;
    k.id = stream ? KernelId::PairStream : steps == 1 ? KernelId::PairSingle : KernelId::PairMulti;
    k.robots_per_block = 32;
  } else if (steps == 1) {
    k.id = p.persist ? KernelId::OnestepPersist : p.lowreg ? KernelId::Lowreg : p.split ? KernelId::Split : p.onestep_v2 ? KernelId::Onestep : KernelId::StepSingle;
  } else {
    k.id = KernelId::StepMulti;
  }
  // the role-split kernel runs two waves (estimator, controller) per 64 robots
  if (k.id == KernelId::Split || k.id == KernelId::PrSplit) k.block = 128;
  return k;
}

// The kernel's name as rocprofv3 prints it (template arguments included), for the table test and for reports.
inline std::string planned_kernel_name(const KernelPlan& p, const PlannedKernel& k) {
  char b[160];
  const unsigned n = p.n;
  // FK / TD exist from six cables on: below that the stage flags fall back to the plain instantiation (CDPR_PICK_STAGES)
  const char* fk = (p.fk && n >= 6) ? "true" : "false";
  const char* td = (p.td && n >= 6) ? "true" : "false";
  const int nbmax = p.gen_nb > 11 ? 32 : 11;
  switch (k.id) {
    case KernelId::None: return "";
    case KernelId::StepSingle: snprintf(b, sizeof b, "cdpr_step_kernel<%u, %s, %s, SINGLE>", n, fk, td); break;
    case KernelId::StepMulti: snprintf(b, sizeof b, "cdpr_step_kernel<%u, %s, %s>", n, fk, td); break;
    case KernelId::Lowreg: snprintf(b, sizeof b, "cdpr_step_kernel<%u, true, %s, SINGLE, LOWREG>", n, td); break;
    case KernelId::OnestepPersist: snprintf(b, sizeof b, "cdpr_onestep_kernel<%u, %s, %s, PERSIST>", n, fk, td); break;
    case KernelId::Split: snprintf(b, sizeof b, "cdpr_split_kernel<%u, false>", n); break;
    case KernelId::Onestep: snprintf(b, sizeof b, "cdpr_onestep_kernel<%u, %s, %s>", n, fk, td); break;
    case KernelId::PhysStep: snprintf(b, sizeof b, "cdpr_step_kernel<%u, %s, %s, PHYS>", n, fk, td); break;
    case KernelId::Rollout: snprintf(b, sizeof b, "cdpr_step_kernel<%u, %s, %s, ROLLOUT>", n, fk, td); break;
    case KernelId::PhysRollout: snprintf(b, sizeof b, "cdpr_step_kernel<%u, %s, %s, ROLLOUT, PHYS>", n, fk, td); break;
    case KernelId::PrSingle: snprintf(b, sizeof b, "cdpr_step_kernel<%u, %s, %s, SINGLE, PR>", n, fk, td); break;
    case KernelId::PrLowreg: snprintf(b, sizeof b, "cdpr_step_kernel<%u, %s, %s, SINGLE, LOWREG, PR>", n, fk, td); break;
    case KernelId::PrSplit: snprintf(b, sizeof b, "cdpr_split_kernel<%u, true>", n); break;
    case KernelId::PrMulti: snprintf(b, sizeof b, "cdpr_step_kernel<%u, %s, %s, PR>", n, fk, td); break;
    case KernelId::PrRollout: snprintf(b, sizeof b, "cdpr_step_kernel<%u, %s, %s, ROLLOUT, PR>", n, fk, td); break;
    case KernelId::PairSingle: snprintf(b, sizeof b, "cdpr_step_kernel_pair<%u, %s, %s, true>", n, fk, td); break;
    case KernelId::PairMulti: snprintf(b, sizeof b, "cdpr_step_kernel_pair<%u, %s, %s, false>", n, fk, td); break;
    case KernelId::PairStream: snprintf(b, sizeof b, "cdpr_pair_stream_kernel<%u>", n); break;
    case KernelId::Cable: snprintf(b, sizeof b, "cdpr_step_kernel_cable<%u, %s, %s>", n, fk, td); break;
    case KernelId::GenOne: snprintf(b, sizeof b, "cdpr_gen_step_kernel<%u, %s, %s, false, %d, SINGLE>", n, fk, td, nbmax); break;
    case KernelId::GenMulti: snprintf(b, sizeof b, "cdpr_gen_step_kernel<%u, %s, %s, false, %d>", n, fk, td, nbmax); break;
    case KernelId::GenRollout: snprintf(b, sizeof b, "cdpr_gen_step_kernel<%u, %s, %s, ROLLOUT, %d>", n, fk, td, nbmax); break;
    case KernelId::GenSplit: snprintf(b, sizeof b, "cdpr_gen_split_kernel<%u>", n); break;
    case KernelId::GenLean: snprintf(b, sizeof b, "cdpr_gen_lean_kernel<%u>%s", n, p.gen_hot ? " + hot rows" : ""); break;
    case KernelId::F64: snprintf(b, sizeof b, "cdpr_step_kernel_f64<%u%s%s>", n, k.f64_ring_lds ? ", RING_LDS" : "", k.f64_jcache ? ", JCACHE" : ""); break;
    case KernelId::F64Pr: snprintf(b, sizeof b, "cdpr_step_kernel_f64<%u, PR%s>", n, k.f64_ring_lds ? ", RING_LDS" : ""); break;
    case KernelId::F64Tstop:
      if (p.hold64)
        snprintf(b, sizeof b, "cdpr_step_kernel_f64<%u, %sHOLD = %d, TSTOP%s>", n, p.per_robot ? "PR, " : "", p.hold_full ? 2 : 1, p.hold_long ? ", HW = 32" : "");
      else
        snprintf(b, sizeof b, "cdpr_step_kernel_f64<%u, %sTSTOP>", n, p.per_robot ? "PR, " : "");
      break;
    case KernelId::F64Long: snprintf(b, sizeof b, "cdpr_step_kernel_f64<%u, %s%sW = 31>", n, p.per_robot ? "PR, " : "", p.tstop64 ? "TSTOP, " : ""); break;
    case KernelId::F64Hold: snprintf(b, sizeof b, "cdpr_step_kernel_f64<%u, HOLD = %d%s>", n, p.hold_full ? 2 : 1, p.hold_long ? ", HW = 32" : ""); break;
    case KernelId::F64HoldPr: snprintf(b, sizeof b, "cdpr_step_kernel_f64<%u, PR, HOLD = %d%s>", n, p.hold_full ? 2 : 1, p.hold_long ? ", HW = 32" : ""); break;
    case KernelId::F64Split: snprintf(b, sizeof b, "cdpr_split_kernel_f64<%u%s>", n, k.f64_lean ? ", LEAN" : ""); break;
    case KernelId::F64SplitHold: snprintf(b, sizeof b, "cdpr_split_kernel_f64<%u%s, HOLD = %d>", n, k.f64_lean ? ", LEAN" : "", p.hold_full ? 2 : 1); break;
  }
  return b;
}

}  // namespace cdpr
