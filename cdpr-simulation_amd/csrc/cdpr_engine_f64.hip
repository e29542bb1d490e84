// cdpr_engine_f64.hip - host side of cdpr_config_t.precision = 64: the launch chain of the fp64 step kernels (cdpr_step_kernel_f64.hpp),
// their read-out and the rollout by composition.  Reference paths as in cdpr_engine.hip.
#include "cdpr_engine_internal.hpp"

namespace cdpr_host {

// rows of an fp64 handle's state: platform, FK estimate, one Pid's rows per cable - and, hold branch live, both Pids' records
size_t state64_rows(const cdpr_engine* h) { return (size_t)f64_state_rows((int)h->n, h->win64) + (h->hold64 ? (size_t)f64_hold_rows((int)h->n, h->hold_win) : 0); }

// fp64 handles: home state (platform at home, FK seed at home, controller rows zero), observables before the first publish
int upload_home64(cdpr_engine* h) {
  const size_t st = h->stride;
  std::vector<double> s(state64_rows(h) * st, 0.0), o((size_t)f64_obs_rows((int)h->n) * st, 0.0);
  for (uint32_t r = 0; r < h->stride; ++r)
    for (int c = 0; c < 7; ++c) {
      s[(size_t)c * st + r] = h->cfg.home_pose[c];
      s[(size_t)(13 + c) * st + r] = h->cfg.home_pose[c];
      o[(size_t)c * st + r] = h->cfg.home_pose[c];
    }
  HIP_TRY(h, hipMemcpyAsync(h->d_state64, s.data(), s.size() * sizeof(double), hipMemcpyHostToDevice, h->stream));
  HIP_TRY(h, hipMemcpyAsync(h->d_obs64, o.data(), o.size() * sizeof(double), hipMemcpyHostToDevice, h->stream));
  if (h->d_dbg64) HIP_TRY(h, hipMemsetAsync(h->d_dbg64, 0, (size_t)h->batch * CDPR_PID_DEBUG_AXES * sizeof(double), h->stream));
  if (h->d_mode) HIP_TRY(h, hipMemsetAsync(h->d_mode, kModePosition, h->batch, h->stream));  // PLG.cpp:153-157 (call count 0)
  if (h->d_target) HIP_TRY(h, hipMemsetAsync(h->d_target, 0, (size_t)h->stride * h->n * sizeof(float), h->stream));
  for (int i = 0; i < 2; ++i) {
    HIP_TRY(h, hipMemsetAsync(h->d_vel[i], 0, (size_t)h->stride * h->n * sizeof(float), h->stream));
    HIP_TRY(h, hipMemsetAsync(h->d_pos[i], 0, (size_t)h->stride * h->n * sizeof(float), h->stream));
    HIP_TRY(h, hipMemsetAsync(h->d_frc[i], 0, (size_t)h->stride * h->n * sizeof(float), h->stream));
  }
  HIP_TRY(h, wait_stream(h));
  return CDPR_OK;
}

void fill_pid64(const cdpr_pid_params_t& p, double dt, F64Args& k) {
  k.kf = p.forward_gain; k.kp = p.p_gain; k.ki = p.i_gain; k.kd = p.d_gain;
  k.imax = std::fabs(p.i_limit); k.imin = -std::fabs(p.i_limit);  // Pid.cpp:70-73 (abs -> fabs)
  k.cmax = std::fabs(p.cmd_limit); k.cmin = -std::fabs(p.cmd_limit);
  k.inv_dt = 1.0 / dt;
  k.nbuf = (int)p.d_buffer_length;
  k.clamp_cmd = k.cmax > k.cmin;
}

// The arguments of a launch that depend on the handle's mode, not on the step: the Pid(s), the command buffer, the weight table; per-robot
// handles: both Pids and the meta row; HOLD handles: both Pids, the hold branch's parameters, the cascades' coefficients
static void f64_mode_args(const cdpr_engine* h, bool vel, bool frc, F64Args& a) {
  const bool pr = h->per_robot;
  fill_pid64(vel ? h->cfg.velocity_pid : h->cfg.position_pid, h->cfg.dt, a);
  a.cmd = pr ? h->d_target
             : frc ? (h->ext_frc[0] ? h->ext_frc[0] : h->d_frc[0])
                   : vel ? (h->ext_vel[0] ? h->ext_vel[0] : h->d_vel[0]) : (h->ext_pos[0] ? h->ext_pos[0] : h->d_pos[0]);
  a.wtab = h->d_wtab64 + (vel ? 0 : h->win64 * (h->win64 + 2));
  if (pr) {  // mode, Pid call count and so the Pid per lane: the velocity Pid in the primary fields, the position Pid in alt_*
    F64Args p = h->base64;
    fill_pid64(h->cfg.position_pid, h->cfg.dt, p);
    a.alt_kf = p.kf, a.alt_kp = p.kp, a.alt_ki = p.ki, a.alt_kd = p.kd;
    a.alt_imax = p.imax, a.alt_imin = p.imin, a.alt_cmax = p.cmax, a.alt_cmin = p.cmin, a.alt_clamp_cmd = p.clamp_cmd;
    a.meta = h->d_mode;
  }
  if (h->hold64) {  // both Pids alive: the velocity Pid in the primary fields, the position Pid in alt_*
    F64Args v = h->base64, p = h->base64;
    fill_pid64(h->cfg.velocity_pid, h->cfg.dt, v);
    fill_pid64(h->cfg.position_pid, h->cfg.dt, p);
    a.kf = v.kf, a.kp = v.kp, a.ki = v.ki, a.kd = v.kd, a.imax = v.imax, a.imin = v.imin, a.cmax = v.cmax, a.cmin = v.cmin, a.nbuf = v.nbuf, a.clamp_cmd = v.clamp_cmd;
    a.alt_kf = p.kf, a.alt_kp = p.kp, a.alt_ki = p.ki, a.alt_kd = p.kd, a.alt_imax = p.imax, a.alt_imin = p.imin, a.alt_cmax = p.cmax, a.alt_cmin = p.cmin;
    a.alt_clamp_cmd = p.clamp_cmd, a.alt_nbuf = p.nbuf;
    a.degree = (int)h->cfg.velocity_pid.d_degree, a.alt_degree = (int)h->cfg.position_pid.d_degree;
    a.hold_eps = h->cfg.velocity_epsilon;
    a.hold_mode = frc ? 0 : (vel ? 2 : 1);
    const cdpr_pid_params_t* pids[2] = {&h->cfg.position_pid, &h->cfg.velocity_pid};
    a.any_cas = a.max_cas = 0;
    a.any_noclamp = (!v.clamp_cmd || !p.clamp_cmd) ? 1 : 0;
    for (int t = 0; t < 2; ++t) {  // BiQuad::SetFc(fc, fs = 1.0, q), Filter.h:130-140, in double
      const cdpr_filter_params_t* fl[2] = {&pids[t]->p_filter, &pids[t]->d_filter};
      double* co[2] = {a.pcoef[t], a.dcoef[t]};
      for (int f = 0; f < 2; ++f) {
        for (int c = 0; c < 5; ++c) co[f][c] = 0.0;
        if (!fl[f]->cascade) continue;
        const double k = std::tan(M_PI * fl[f]->rel_cutoff / 1.0);
        const double den = k * k + k / fl[f]->quality + 1.0;
        co[f][0] = k * k / den, co[f][1] = 2.0 * co[f][0], co[f][2] = co[f][0];
        co[f][3] = 2.0 * (k * k - 1.0) / den, co[f][4] = (k * k - k / fl[f]->quality + 1.0) / den;
      }
      a.pcas[t] = (int)std::min<uint32_t>(pids[t]->p_filter.cascade, (uint32_t)kHoldMaxCas);
      a.dcas[t] = (int)std::min<uint32_t>(pids[t]->d_filter.cascade, (uint32_t)kHoldMaxCas);
      a.max_cas = std::max(a.max_cas, std::max(a.pcas[t], a.dcas[t]));
    }
    a.any_cas = a.max_cas > 0 ? 1 : 0;
    for (int t = 0; t < 2; ++t) {  // uniform-grid weights by age of the sample (derivative_weights: oldest first)
      double w[CDPR_MAX_D_BUFFER];
      const uint32_t nb = pids[t]->d_buffer_length;
      if (derivative_weights(nb, pids[t]->d_degree, w) == CDPR_OK)
        for (uint32_t age = 0; age < nb && age < (uint32_t)h->hold_win; ++age) a.hold_w[t][age] = w[nb - 1 - age];
    }
  }
}

// the fp64 kernel a planned id stands for on this handle
static F64Kernel f64_kernel_for(const cdpr_engine* h, const PlannedKernel& q) {
  const uint32_t n = h->n;
  const bool pr = h->per_robot, hold_full = h->plan.hold_full;
  switch (q.id) {
    case KernelId::F64Split: return pick_f64_split_kernel(n, q.f64_lean);
    case KernelId::F64SplitHold: return pick_f64_split_hold_kernel(n, q.f64_lean, hold_full);
    case KernelId::F64Hold: return h->plan.hold_long ? pick_f64_hold_long_kernel(n, false, false) : pick_f64_hold_kernel(n, hold_full);
    case KernelId::F64HoldPr: return h->plan.hold_long ? pick_f64_hold_long_kernel(n, true, false) : pick_f64_hold_pr_kernel(n, hold_full);
    case KernelId::F64Tstop: return h->plan.hold_long ? pick_f64_hold_long_kernel(n, pr, true) : pick_f64_tstop_kernel(n, pr, h->hold64 ? (hold_full ? 2 : 1) : 0);
    case KernelId::F64Long: return pick_f64_long_kernel(n, pr, h->tstop64);
    case KernelId::F64Pr: return pick_f64_pr_kernel(n, q.f64_ring_lds);
    default: return pick_f64_kernel(n, q.f64_ring_lds, q.f64_jcache);
  }
}

// precision = 64: the same host logic (commands are latched by run_steps before this is reached), the fp64 kernel
int run_steps_f64(cdpr_engine* h, int nsteps, int per_launch, bool reset_pid, double* record) {
  const uint32_t n = h->n;
  if (reset_pid && !h->hold64) {  // Pid::reset (Pid.cpp:100-115): zero every controller row (hold branch live: the latch reset that Pid's own rows)
    h->pid_calls = 0;
    HIP_TRY(h, hipMemsetAsync(h->d_state64 + (size_t)20 * h->stride, 0, (size_t)(h->win64 + 1) * n * h->stride * sizeof(double), h->stream));
  }
  F64Args a = h->base64;
  a.stamps = h->base.stamps;
  const bool pr = h->per_robot;
  const bool vel = pr || h->mode == kModeVelocity, frc = !pr && h->mode == kModeForce;
  f64_mode_args(h, vel, frc, a);
  const size_t image64 = (size_t)f64_obs_rows((int)n) * h->stride;  // doubles per observable image
  a.obs_step_stride = record ? image64 : 0;
  // the rings in LDS (64 KiB per wave at n = 8: two waves per CU) while the batch leaves CUs to spare
  const int ring_env = [] { const char* v = std::getenv("CDPR_F64_RING_LDS"); return v ? atoi(v) : -1; }();  // (read per call: A/B in one process)
  // ... and the structure-matrix rows too (112 KiB: one wave per CU) up to one workgroup per CU
  const int jc_env = [] { const char* v = std::getenv("CDPR_F64_JCACHE"); return v ? atoi(v) : -1; }();  // (read per call: A/B in one process)
  a.travel_stop = h->tstop64 ? (int)h->cfg.travel_stop : 0;
  // one step per launch on FK + TD handles up to one workgroup per CU: estimator wave + controller wave (cdpr_split_kernel_f64)
  const int sp_env = [] { const char* v = std::getenv("CDPR_F64_SPLIT"); return v ? atoi(v) : -1; }();  // (read per call: A/B in one process)
  // (CDPR_F64_SPLIT = 0 never, 1 the LDS-cached build, 2 the lean build whatever the batch)
  // The routing itself: planned_kernel (cdpr_select.hpp) for a one-step and for a several-steps launch of this handle
  LaunchShape s1 = launch_shape(h, 1), sk = launch_shape(h, 2);
  s1.f64_ring_lds = sk.f64_ring_lds = ring_env, s1.f64_jcache = sk.f64_jcache = jc_env, s1.f64_split = sk.f64_split = sp_env;
  const PlannedKernel pk1 = planned_kernel(h->plan, s1), pkk = planned_kernel(h->plan, sk);
  auto f64_kernel_of = [&](const PlannedKernel& q) -> F64Kernel { return f64_kernel_for(h, q); };
  const bool split1 = pk1.id == KernelId::F64Split || pk1.id == KernelId::F64SplitHold;
  F64Kernel split_kern = split1 ? f64_kernel_of(pk1) : nullptr;  // (per-robot handles: the one-wave kernel)
  // up to one workgroup per CU the role-split kernel's one-step launches beat the one-wave kernel's multi-step ones
  // (14.4 against 20.8 us per step at one robot x 8, same bits): a fused update then runs as one-step launches
  const bool fused_as_single = pkk.id == KernelId::F64Split || pkk.id == KernelId::F64SplitHold;
  if (fused_as_single) per_launch = 1;
  PlannedKernel one_wave = pkk;  // the one-wave kernel of this handle (what a several-steps launch runs, or would run)
  if (fused_as_single) { LaunchShape so = sk; so.f64_split = 0; one_wave = planned_kernel(h->plan, so); }
  F64Kernel kern = f64_kernel_of(one_wave);
  int done = 0;
  while (done < nsteps) {
    const int k = std::min(per_launch, nsteps - done);
    a.nsteps = k;
    a.flags = pr ? 0u : (vel ? kFlagActualIsVelocity : (frc ? kFlagForceMode : 0u));
    const bool first_world = (h->step == 0);
    if (first_world) a.flags |= kFlagFirstWorldStep;
    if (record) a.obs = record + (size_t)done * image64;
    a.pid_calls = sat_pid_calls(h->pid_calls);
    a.ring_slot = ring_slot_of(h->step, h->win64);
    a.step0 = (int)h->step;
    a.publish_mask = 0;
    for (int j = 0; j < k; ++j) {  // PLG.cpp:236-242: strict '>' against the last published stamp
      const double now = sim_time(h->step + (uint64_t)j, h->cfg.dt);
      if ((now - h->prev_publish) > h->cfg.publish_period) {
        h->prev_publish = now;
        a.publish_mask |= (1ull << j);
      }
    }
    if (k == 1 && split_kern) {
      hipLaunchKernelGGL(split_kern, dim3((h->batch + 63u) / 64u), dim3(128), 0, h->stream, a);
      h->last_kernel = pk1;
    } else {
      hipLaunchKernelGGL(kern, dim3((h->batch + 63u) / 64u), dim3(64), 0, h->stream, a);
      h->last_kernel = one_wave;
    }
    HIP_TRY(h, hipGetLastError());
    ++h->launches;
    h->step += (uint64_t)k;
    if (!frc) h->pid_calls = sat_pid_calls(h->pid_calls + k - (first_world ? 1 : 0));  // (no Pid call in Force mode)
    done += k;
  }
  if (record && h->cfg.publish_period == 0.0 && h->step > 1)  // keep cdpr_get_* consistent: latest image into the engine's own
    HIP_TRY(h, hipMemcpyAsync(h->d_obs64, record + (size_t)(nsteps - 1) * image64, image64 * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
  return CDPR_OK;
}

// rows [first_row, first_row + width) of a double row buffer -> robot-major host array (double, or float when as_float)
int fetch_rows64(cdpr_engine* h, const double* rows, uint32_t first_row, uint32_t width, void* host_out, bool as_float) {
  if (!host_out) return CDPR_OK;
  const size_t count = (size_t)h->batch * width, bytes = count * (as_float ? sizeof(float) : sizeof(double));
  if (h->unpack64_cap < bytes) {
    HIP_TRY(h, wait_stream(h));
    if (h->d_unpack64) (void)hipFree(h->d_unpack64);
    h->d_unpack64 = nullptr;
    h->unpack64_cap = 0;
    HIP_TRY(h, hipMalloc(&h->d_unpack64, bytes));
    h->unpack64_cap = bytes;
  }
  Unpack64Args u{};
  u.rows = rows;
  u.out = h->d_unpack64;
  u.stride = h->stride;
  u.batch = h->batch;
  u.width = width;
  u.first_row = first_row;
  u.as_float = as_float ? 1 : 0;
  hipLaunchKernelGGL(cdpr_unpack64_kernel, dim3((uint32_t)((count + 255) / 256)), dim3(256), 0, h->stream, u);
  HIP_TRY(h, hipGetLastError());
  HIP_TRY(h, hipMemcpyAsync(host_out, h->d_unpack64, bytes, hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(h, wait_stream(h));
  return CDPR_OK;
}

// the five observable arrays of an fp64 handle (any may be null), double or rounded to float
// The five observable arrays of a precision = 64 handle in ONE device round trip (round 6; five unpack launches, five copies and
// five waits before: 88 us per world step for one robot with the host in the loop): one gather launch writes them - small
// batches straight into a mapped pinned host image, larger ones into device scratch followed by one copy each - one wait.
int fetch_observables64(cdpr_engine* h, void* position, void* velocity, void* effort, void* pose7, void* twist6, bool as_float) {
  const uint32_t n = h->n;
  void* dst[5] = {position, velocity, effort, pose7, twist6};
  const uint32_t first[5] = {16u, 16u + n, 16u + 2u * n, 0u, 7u}, width[5] = {n, n, n, 7u, 6u};
  Unpack64MultiArgs u{};
  u.rows = h->d_obs64;
  u.stride = h->stride;
  u.batch = h->batch;
  u.as_float = as_float ? 1 : 0;
  void* want[5];
  uint32_t cum = 0;
  size_t off = 0;
  for (int i = 0; i < 5; ++i) {
    if (!dst[i]) continue;
    const uint32_t k = u.nseg++;
    u.first_row[k] = first[i], u.width[k] = width[i], u.cum[k] = cum, u.off[k] = (uint32_t)off;
    want[k] = dst[i];
    cum += width[i];
    off += (size_t)h->batch * width[i];
  }
  if (u.nseg == 0) return CDPR_OK;
  u.total_width = cum;
  const size_t esz = as_float ? sizeof(float) : sizeof(double), bytes = off * esz;
  if (off >= (1ull << 32)) {  // element offsets are 32-bit
    int rc = CDPR_OK;
    for (int i = 0; i < 5 && rc == CDPR_OK; ++i) rc = fetch_rows64(h, h->d_obs64, first[i], width[i], dst[i], as_float);
    return rc;
  }
  const bool pinned = bytes <= (2u << 20);
  if (pinned) {
    if (!h->h_pub64) HIP_TRY(h, hipHostMalloc(&h->h_pub64, 2u << 20, hipHostMallocMapped | hipHostMallocCoherent));
    HIP_TRY(h, hipHostGetDevicePointer(&u.out, h->h_pub64, 0));
  } else {
    if (h->unpack64_cap < bytes) {
      HIP_TRY(h, wait_stream(h));
      if (h->d_unpack64) (void)hipFree(h->d_unpack64);
      h->d_unpack64 = nullptr;
      h->unpack64_cap = 0;
      HIP_TRY(h, hipMalloc(&h->d_unpack64, bytes));
      h->unpack64_cap = bytes;
    }
    u.out = h->d_unpack64;
  }
  hipLaunchKernelGGL(cdpr_unpack64_multi_kernel, dim3((uint32_t)(((size_t)h->batch * cum + 255) / 256)), dim3(256), 0, h->stream, u);
  HIP_TRY(h, hipGetLastError());
  if (!pinned)
    for (uint32_t k = 0; k < u.nseg; ++k)
      HIP_TRY(h, hipMemcpyAsync(want[k], static_cast<const char*>(h->d_unpack64) + (size_t)u.off[k] * esz, (size_t)h->batch * u.width[k] * esz, hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(h, wait_stream(h));
  if (pinned)
    for (uint32_t k = 0; k < u.nseg; ++k) memcpy(want[k], static_cast<const char*>(h->h_pub64) + (size_t)u.off[k] * esz, (size_t)h->batch * u.width[k] * esz);
  return CDPR_OK;
}

int set_platform_state64(cdpr_engine* h, const double* pose7, const double* twist6) {
  const size_t st = h->stride;
  std::vector<double> s((size_t)20 * st);
  HIP_TRY(h, hipMemcpyAsync(s.data(), h->d_state64, s.size() * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(h, wait_stream(h));
  for (uint32_t r = 0; r < h->batch; ++r) {
    if (pose7)
      for (int c = 0; c < 7; ++c) s[(size_t)c * st + r] = s[(size_t)(13 + c) * st + r] = pose7[(size_t)r * 7 + c];  // the FK seed follows the spawn pose
    if (twist6)
      for (int c = 0; c < 6; ++c) s[(size_t)(7 + c) * st + r] = twist6[(size_t)r * 6 + c];
  }
  HIP_TRY(h, hipMemcpyAsync(h->d_state64, s.data(), s.size() * sizeof(double), hipMemcpyHostToDevice, h->stream));
  HIP_TRY(h, wait_stream(h));
  return CDPR_OK;
}

// image of a precision = 64 handle -> robot-major arrays, as double or rounded to float
template <typename T>
void decode_image64(const cdpr_engine* h, const double* o, T* position, T* velocity, T* effort, T* pose7, T* twist6) {
  const size_t st = h->stride;
  const uint32_t n = h->n;
  T* dst[3] = {position, velocity, effort};
  for (int f = 0; f < 3; ++f) {
    if (!dst[f]) continue;
    for (uint32_t r = 0; r < h->batch; ++r)
      for (uint32_t i = 0; i < n; ++i) dst[f][(size_t)r * n + i] = (T)o[(size_t)(16 + f * n + i) * st + r];
  }
  for (uint32_t r = 0; r < h->batch; ++r) {
    if (pose7)
      for (int c = 0; c < 7; ++c) pose7[(size_t)r * 7 + c] = (T)o[(size_t)c * st + r];
    if (twist6)
      for (int c = 0; c < 6; ++c) twist6[(size_t)r * 6 + c] = (T)o[(size_t)(7 + c) * st + r];
  }
}
void decode_image64_to_float(const cdpr_engine* h, const double* image, float* position, float* velocity, float* effort, float* pose7, float* twist6) {
  decode_image64(h, image, position, velocity, effort, pose7, twist6);
}

}  // namespace cdpr_host

int cdpr_decode_observables_f64(cdpr_handle_t h, const void* image, double* position, double* velocity, double* effort, double* pose7, double* twist6) {
  if (!h || !image) return CDPR_ERR_INVALID;
  if (!h->fp64) {
    h->err = "cdpr_decode_observables_f64: the handle was not created with precision = 64";
    return CDPR_ERR_UNSUPPORTED;
  }
  decode_image64(h, static_cast<const double*>(image), position, velocity, effort, pose7, twist6);
  return CDPR_OK;
}
namespace cdpr_host {

static int need_fp64(cdpr_engine* h, const char* what) {
  if (h->fp64) return CDPR_OK;
  h->err = std::string(what) + ": the handle was not created with precision = 64";
  return CDPR_ERR_UNSUPPORTED;
}
}  // namespace cdpr_host

int cdpr_get_observables_f64(cdpr_handle_t h, double* position, double* velocity, double* effort, double* pose7, double* twist6) {
  if (!h) return CDPR_ERR_INVALID;
  if (int rc = need_fp64(h, "cdpr_get_observables_f64")) return rc;
  if (set_device(h) != CDPR_OK) return CDPR_ERR_DEVICE;
  return checked(h, fetch_observables64(h, position, velocity, effort, pose7, twist6, false));
}

int cdpr_get_raw_state_f64(cdpr_handle_t h, double* pose7, double* twist6) {
  if (!h) return CDPR_ERR_INVALID;
  if (int rc = need_fp64(h, "cdpr_get_raw_state_f64")) return rc;
  if (set_device(h) != CDPR_OK) return CDPR_ERR_DEVICE;
  int rc = fetch_rows64(h, h->d_state64, 0, 7, pose7, false);
  return rc != CDPR_OK ? rc : checked(h, fetch_rows64(h, h->d_state64, 7, 6, twist6, false));
}

int cdpr_set_platform_state_f64(cdpr_handle_t h, const double* pose7, const double* twist6) {
  if (!h) return CDPR_ERR_INVALID;
  if (int rc = need_fp64(h, "cdpr_set_platform_state_f64")) return rc;
  if (set_device(h) != CDPR_OK) return CDPR_ERR_DEVICE;
  return set_platform_state64(h, pose7, twist6);
}
namespace cdpr_host {

// fp64 handles: one observable row of doubles as int32 per robot (iteration count, flags)
int fetch_int_row64(cdpr_engine* h, uint32_t row, int32_t* out) {
  if (!out) return CDPR_OK;
  std::vector<double> d(h->batch);
  int rc = fetch_rows64(h, h->d_obs64, row, 1, d.data(), false);
  if (rc == CDPR_OK)
    for (uint32_t b = 0; b < h->batch; ++b) out[b] = (int32_t)d[b];
  return rc;
}

// Queue one rollout on the handle's stream: trajectories = batch * samples, reference positions and costs in
// DEVICE buffers.  Nothing is allocated, copied or synchronised here.
// The rollout of a precision = 64 handle (uniform modes; with the hold branch / cascades / cmd_limit 0 since the end of round 6): see Roll64Args
// (cdpr_step_kernel_f64.hpp).
int rollout_enqueue_f64(cdpr_engine* h, int samples, int horizon, const float* d_commands, const float* d_ref, float* d_cost) {
  const uint32_t n = h->n;
  const uint64_t traj = (uint64_t)h->batch * (uint64_t)samples;
  const size_t cols = (size_t)((traj + 63u) & ~(uint64_t)63u);
  const uint32_t rows = (uint32_t)state64_rows(h);  // (hold branch live: both Pids' records of every cable travel with a trajectory)
  if (h->roll64_cols < cols) {
    HIP_TRY(h, wait_stream(h));
    for (void** p64 : {(void**)&h->d_roll64, (void**)&h->d_roll64_acc, (void**)&h->d_roll64_cmd, (void**)&h->d_roll64_meta}) {
      if (*p64) (void)hipFree(*p64);
      *p64 = nullptr;
    }
    h->roll64_cols = 0;
    HIP_TRY(h, hipMalloc(&h->d_roll64, (size_t)rows * cols * sizeof(double)));
    HIP_TRY(h, hipMalloc(&h->d_roll64_acc, cols * sizeof(double)));
    HIP_TRY(h, hipMalloc(&h->d_roll64_cmd, cols * n * sizeof(float)));
    if (h->per_robot) HIP_TRY(h, hipMalloc(&h->d_roll64_meta, cols));
    h->roll64_cols = cols;
  }
  const bool pr = h->per_robot;
  const bool reset = pr || h->mode != kModeVelocity;  // JFC.cpp:113-115: the copies start from a reset velocity Pid, the handle's rows stay (per-robot handles: per robot, from its meta byte)
  const uint32_t blocks = (uint32_t)((traj + 255u) / 256u);
  Roll64Args e{};
  e.src = h->d_state64, e.dst = h->d_roll64, e.src_stride = h->stride, e.dst_stride = (uint32_t)h->roll64_cols, e.rows = rows, e.batch = h->batch,
  e.samples = (uint32_t)samples, e.zero_from = (reset && !h->hold64) ? 20u : rows;
  if (h->hold64 && reset) {  // ... there the velocity Pid's own record of every cable is what the Joy resets
    e.hold_base = (uint32_t)f64_state_rows((int)n), e.hold_cable_rows = (uint32_t)hold_cable_rows(h->hold_win), e.hold_pid_rows = (uint32_t)hold_pid_rows(h->hold_win);
    e.hold_zero_pid = 2u;  // 1 + the Pid's index
  }
  if (pr) e.meta_src = h->d_mode, e.meta_dst = h->d_roll64_meta;
  hipLaunchKernelGGL(cdpr_roll64_expand_kernel, dim3(blocks), dim3(256), 0, h->stream, e);
  HIP_TRY(h, hipGetLastError());
  HIP_TRY(h, hipMemsetAsync(h->d_roll64_acc, 0, cols * sizeof(double), h->stream));
  F64Args a = h->base64;
  f64_mode_args(h, true, false, a);  // Velocity mode (hold branch live: both Pids, the branch's parameters)
  a.state = h->d_roll64;
  a.obs = h->d_obs64;  // (nothing is published: publish_mask = 0)
  a.dbg = nullptr;
  a.cmd = h->d_roll64_cmd;  // (per-robot handles: the trajectory's ACTIVE target row)
  if (pr) a.meta = h->d_roll64_meta;
  a.wtab = h->d_wtab64;
  a.batch = (uint32_t)traj;
  a.stride = (uint32_t)h->roll64_cols;
  a.nsteps = 1;
  a.publish_mask = 0;
  a.obs_step_stride = 0;
  LaunchShape shape = launch_shape(h, 2);  // the handle's one-wave kernel, rings in memory
  shape.f64_split = 0, shape.f64_ring_lds = 0, shape.f64_jcache = 0;
  F64Kernel kern = f64_kernel_for(h, planned_kernel(h->plan, shape));
  a.travel_stop = h->tstop64 ? (int)h->cfg.travel_stop : 0;
  int calls = reset ? 0 : h->pid_calls;  // (uniform handles; per-robot handles count in the meta bytes)
  for (int k = 0; k < horizon; ++k) {
    Roll64CmdArgs c{};
    c.commands = d_commands, c.out = h->d_roll64_cmd, c.batch = h->batch, c.samples = (uint32_t)samples, c.horizon = (uint32_t)horizon, c.n = n, c.k = (uint32_t)k;
    hipLaunchKernelGGL(cdpr_roll64_cmd_kernel, dim3((uint32_t)((traj * n + 255u) / 256u)), dim3(256), 0, h->stream, c);
    const bool first_world = (h->step + (uint64_t)k) == 0;
    a.flags = (pr ? 0u : kFlagActualIsVelocity) | (first_world ? kFlagFirstWorldStep : 0u);
    a.pid_calls = sat_pid_calls(calls);
    a.ring_slot = ring_slot_of(h->step + (uint64_t)k, h->win64);
    a.step0 = (int)(h->step + (uint64_t)k);
    hipLaunchKernelGGL(kern, dim3((uint32_t)((traj + 63u) / 64u)), dim3(64), 0, h->stream, a);
    calls = sat_pid_calls(calls + (first_world ? 0 : 1));
    Roll64CostArgs q{};
    q.state = h->d_roll64, q.ref = d_ref, q.acc = h->d_roll64_acc, q.out = (k == horizon - 1) ? d_cost : nullptr, q.stride = (uint32_t)h->roll64_cols,
    q.batch = h->batch, q.samples = (uint32_t)samples;
    hipLaunchKernelGGL(cdpr_roll64_cost_kernel, dim3(blocks), dim3(256), 0, h->stream, q);
    HIP_TRY(h, hipGetLastError());
    h->launches += 1;
  }
  return CDPR_OK;
}

}  // namespace cdpr_host
