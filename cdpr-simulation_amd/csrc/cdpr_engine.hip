// cdpr_engine.hip — host side of libcdpr_hip.so: the C-ABI of include/cdpr.h over the
// gfx950 step kernels.  No CPU compute path exists here: every entry point that
// touches robot state needs a GPU and fails with CDPR_ERR_DEVICE without one.
//
// Reference paths (relative to src/cdpr_gazebo/ of balazs-bamer/cdpr-simulation):
//   PLG.cpp = src/CdprGazeboPlugin.cpp, JFC.cpp = src/JointForceCalculator.cpp, Pid.cpp = src/Pid.cpp
#include "cdpr_engine_internal.hpp"

namespace {
thread_local std::string g_create_error;
}  // namespace


namespace cdpr_host {

// ---------------------------------------------------------------------------------
// Least-squares end-point derivative weights on a uniform grid: the closed form of
// Pid::derive + fitPolynomial (Pid.cpp:193-247) when samples are one step apart.
// ---------------------------------------------------------------------------------
int derivative_weights(uint32_t n, uint32_t degree, double* w) {
  if (n < 2 || n > CDPR_MAX_D_BUFFER || degree < 1 || degree > CDPR_MAX_D_DEGREE || degree >= n) return CDPR_ERR_INVALID;
  const int m = (int)degree + 1;
  long double a[5][10];  // [XtX | I], centred abscissae x_j = j - (n-1)/2
  for (int i = 0; i < m; ++i)
    for (int j = 0; j < 2 * m; ++j) a[i][j] = 0.0L;
  std::vector<long double> x(n);
  for (uint32_t j = 0; j < n; ++j) x[j] = (long double)j - (long double)(n - 1) / 2.0L;
  for (int i = 0; i < m; ++i) {
    for (int k = 0; k < m; ++k) {
      long double s = 0;
      for (uint32_t j = 0; j < n; ++j) s += powl(x[j], i + k);
      a[i][k] = s;
    }
    a[i][m + i] = 1.0L;
  }
  for (int col = 0; col < m; ++col) {  // Gauss-Jordan, partial pivoting
    int piv = col;
    for (int r = col + 1; r < m; ++r)
      if (fabsl(a[r][col]) > fabsl(a[piv][col])) piv = r;
    if (piv != col)
      for (int k = 0; k < 2 * m; ++k) std::swap(a[piv][k], a[col][k]);
    long double d = a[col][col];
    for (int k = 0; k < 2 * m; ++k) a[col][k] /= d;
    for (int r = 0; r < m; ++r) {
      if (r == col) continue;
      long double f = a[r][col];
      for (int k = 0; k < 2 * m; ++k) a[r][k] -= f * a[col][k];
    }
  }
  // derivative at the newest sample x_e: sum_k k x_e^(k-1) c_k, c = (XtX)^-1 Xt y
  const long double xe = x[n - 1];
  for (uint32_t j = 0; j < n; ++j) {
    long double wj = 0;
    for (int k = 1; k < m; ++k) {
      long double ck = 0;  // row k of (XtX)^-1 Xt, column j
      for (int i = 0; i < m; ++i) ck += a[k][m + i] * powl(x[j], i);
      wj += (long double)k * powl(xe, k - 1) * ck;
    }
    w[j] = (double)wj;
  }
  return CDPR_OK;
}

// (mat3_inverse_sym, validate_config, fast_path_obstacle: cdpr_select.hpp - pure functions, shared with cdpr_plan_kernel)

void fill_pid(const cdpr_pid_params_t& p, double dt, StepArgs& k, float* wtab_host) {
  k.kf = (float)p.forward_gain;
  k.kp = (float)p.p_gain;
  k.ki = (float)p.i_gain;
  k.kd = (float)p.d_gain;
  k.inv_ki = (p.i_gain != 0.0) ? (float)(1.0 / p.i_gain) : 0.f;
  k.imax = (float)std::fabs(p.i_limit);  // Pid.cpp:70-73 (abs -> fabs)
  k.imin = -(float)std::fabs(p.i_limit);
  k.cmax = (float)std::fabs(p.cmd_limit);
  k.cmin = -(float)std::fabs(p.cmd_limit);
  k.inv_dt = (float)(1.0 / dt);
  k.nbuf = (int)p.d_buffer_length;
  k.clamp_cmd = k.cmax > k.cmin;
  // end-point LS derivative weights, oldest..newest, zero padded at the old end to kWin + 1 entries
  double w[CDPR_MAX_D_BUFFER];
  float wpad[kWin + 1];
  for (int j = 0; j <= kWin; ++j) wpad[j] = 0.f;
  if (derivative_weights(p.d_buffer_length, p.d_degree, w) == CDPR_OK && p.d_buffer_length <= (uint32_t)kWin + 1)
    for (uint32_t j = 0; j < p.d_buffer_length; ++j) wpad[kWin + 1 - p.d_buffer_length + j] = (float)w[j];
  // ring: when the new error goes to slot ws, slot (ws - j) mod 10 holds the error of j steps ago (j = 1..10;
  // j = 10 is slot ws itself, the sample about to be overwritten), whose weight is wpad[10 - j]
  for (int ws = 0; ws < kWin; ++ws) {
    for (int s = 0; s < kWin; ++s) {
      int j = ((ws - s) % kWin + kWin) % kWin;
      if (j == 0) j = kWin;
      wtab_host[ws * (kWin + 2) + s] = wpad[kWin - j];
    }
    wtab_host[ws * (kWin + 2) + kWin] = wpad[kWin];
    wtab_host[ws * (kWin + 2) + kWin + 1] = 0.f;
  }
}

void copy_pid(const StepArgs& src, StepArgs& dst) {
  dst.kf = src.kf; dst.kp = src.kp; dst.ki = src.ki; dst.kd = src.kd; dst.inv_ki = src.inv_ki;
  dst.imax = src.imax; dst.imin = src.imin; dst.cmax = src.cmax; dst.cmin = src.cmin; dst.inv_dt = src.inv_dt;
  dst.wtab = src.wtab;
  dst.nbuf = src.nbuf; dst.clamp_cmd = src.clamp_cmd;
}

// per-robot kernels: the position Pid rides along as StepArgs::alt (the primary fields hold the velocity Pid)
void copy_pid_alt(const StepArgs& src, PidSet& dst) {
  dst.kf = src.kf; dst.kp = src.kp; dst.ki = src.ki; dst.kd = src.kd; dst.inv_ki = src.inv_ki;
  dst.imax = src.imax; dst.imin = src.imin; dst.cmax = src.cmax; dst.cmin = src.cmin;
  dst.clamp_cmd = src.clamp_cmd;
}

void fill_consts(const cdpr_config_t& c, StepArgs& k) {
  k.dt = (float)c.dt;
  k.half_dt = (float)(0.5 * c.dt);
  k.dt_inv_mass = (float)(c.dt / c.mass);
  k.fgx = (float)(c.mass * c.gravity[0]);
  k.fgy = (float)(c.mass * c.gravity[1]);
  k.fgz = (float)(c.mass * c.gravity[2]);
  double inv[6];
  mat3_inverse_sym(c.inertia, inv);
  for (int i = 0; i < 6; ++i) {
    k.ib[i] = (float)c.inertia[i];
    k.ibinv[i] = (float)inv[i];
  }
  k.damping = (float)c.joint_damping;
  k.effort = (float)c.effort_limit;
  k.vel_limit = (float)c.velocity_limit;
  k.unilateral = c.unilateral_cables ? 1 : 0;
  k.travel_lo = (float)c.travel_lower;
  k.travel_hi = (float)c.travel_upper;
  k.travel_on = (c.travel_lower != 0.0 || c.travel_upper != 0.0) ? 1 : 0;
  k.travel_stop = k.travel_on ? (int)c.travel_stop : 0;
  k.inv_mass = (float)(1.0 / c.mass);
  k.ph_lumped = (c.passive_damping != 0.0 || c.leg_inertia != 0.0 || c.cable_axial_mass != 0.0 || c.anchor_point_mass != 0.0 || c.anchor_inertia != 0.0) ? 1 : 0;
  k.ph_c = (float)c.passive_damping;
  k.ph_jleg = (float)c.leg_inertia;
  k.ph_max = (float)c.cable_axial_mass;
  k.ph_mpt = (float)c.anchor_point_mass;
  k.ph_iadd_total = (float)(c.anchor_inertia * (double)c.n_cables);
  k.ph_mass = (float)c.mass;
  k.gx = (float)c.gravity[0];
  k.gy = (float)c.gravity[1];
  k.gz = (float)c.gravity[2];
  k.split_swap = 0x9;  // measured on MI355X (65 536 x 8): masks 0 .. 0xf8 give 10.8-11.3 us/step, 0x9 the best; bits 8-9 (the
                       // workgroups that share a CU) make it 12.8: the dispatcher already alternates the SIMD pairs there
  if (const char* sw = std::getenv("CDPR_SPLIT_SWAP")) k.split_swap = (uint32_t)strtoul(sw, nullptr, 0);
  k.fk_lambda = (float)c.fk_lambda;
  k.fk_tol = (float)c.fk_tolerance;
  k.fk_iters = (int)c.fk_max_iterations;
  k.td_min = (float)c.td_f_min;
  k.td_max = (float)c.td_f_max;
  k.td_mid = (float)(0.5 * (c.td_f_min + c.td_f_max));
}

void fill_gen_pid(const cdpr_pid_params_t& p, GenPid& g) {
  g.kf = (float)p.forward_gain; g.kp = (float)p.p_gain; g.ki = (float)p.i_gain; g.kd = (float)p.d_gain;
  g.imax = (float)std::fabs(p.i_limit); g.imin = -(float)std::fabs(p.i_limit);
  g.cmax = (float)std::fabs(p.cmd_limit); g.cmin = -(float)std::fabs(p.cmd_limit);
  g.nbuf = (int)p.d_buffer_length; g.degree = (int)p.d_degree;
  g.pcas = (int)p.p_filter.cascade; g.dcas = (int)p.d_filter.cascade;
  g.clamp = g.cmax > g.cmin ? 1 : 0;
  auto biquad = [](const cdpr_filter_params_t& f, float& a0, float& a1, float& a2, float& b1, float& b2) {
    // BiQuad::SetFc(fc, fs = 1.0, q), Filter.h:130-140
    const double k = std::tan(M_PI * f.rel_cutoff / 1.0);
    const double den = k * k + k / f.quality + 1.0;
    a0 = (float)(k * k / den); a1 = (float)(2.0 * (k * k / den)); a2 = a0;
    b1 = (float)(2.0 * (k * k - 1.0) / den); b2 = (float)((k * k - k / f.quality + 1.0) / den);
  };
  g.pa0 = g.pa1 = g.pa2 = g.pb1 = g.pb2 = g.da0 = g.da1 = g.da2 = g.db1 = g.db2 = 0.f;
  if (g.pcas) biquad(p.p_filter, g.pa0, g.pa1, g.pa2, g.pb1, g.pb2);
  if (g.dcas) biquad(p.d_filter, g.da0, g.da1, g.da2, g.db1, g.db2);
}

// Cable geometry as the kernel wants it in LDS: per cable pair
// [ax0 ax1 ay0 ay1 | az0 az1 bx0 bx1 | by0 by1 bz0 bz1 | l00 l01 mask0 mask1].
std::vector<float> geom_pairs(const cdpr_config_t& c) {
  const int np = cable_pairs((int)c.n_cables);
  std::vector<float> g((size_t)np * kGeomFloatsPerPair, 0.f);
  for (int k = 0; k < np; ++k) {
    for (int h = 0; h < 2; ++h) {
      const uint32_t i = 2 * k + h;
      const bool real = i < c.n_cables;
      // the padding cable of an odd count sits far away (finite length) and is masked to zero
      const double a[3] = {real ? c.frame_anchor[i][0] : 7.0, real ? c.frame_anchor[i][1] : 11.0, real ? c.frame_anchor[i][2] : 13.0};
      const double b[3] = {real ? c.platform_anchor[i][0] : 0.0, real ? c.platform_anchor[i][1] : 0.0, real ? c.platform_anchor[i][2] : 0.0};
      float* p = &g[(size_t)k * kGeomFloatsPerPair];
      p[0 + h] = (float)a[0];
      p[2 + h] = (float)a[1];
      p[4 + h] = (float)a[2];
      p[6 + h] = (float)b[0];
      p[8 + h] = (float)b[1];
      p[10 + h] = (float)b[2];
      p[12 + h] = real ? (float)c.cable_ref_length[i] : 0.f;
      p[14 + h] = real ? 1.f : 0.f;
    }
  }
  return g;
}

double sim_time(uint64_t step, double dt) {
  // gazebo::common::Time keeps integer sec + nsec; Double() = sec + nsec * 1e-9 [EXT]
  const int64_t dt_ns = (int64_t)std::llround(dt * 1e9);
  const int64_t now_ns = (int64_t)step * dt_ns;
  return (double)(now_ns / 1000000000LL) + (double)(now_ns % 1000000000LL) * 1e-9;
}

int set_device(cdpr_engine* h) {
  HIP_TRY(h, hipSetDevice(h->device));
  return CDPR_OK;
}

// Host image of the state a fresh Load leaves: platform at home, zero twist, FK seed at
// home, every controller record zero (count 0 => the first Pid call returns 0).
std::vector<float4> home_state(const cdpr_engine* h) {
  std::vector<float4> s((size_t)h->n_state * h->stride, make_float4(0.f, 0.f, 0.f, 0.f));
  const double* hp = h->cfg.home_pose;
  for (uint32_t r = 0; r < h->stride; ++r) {
    s[0 * (size_t)h->stride + r] = make_float4((float)hp[0], (float)hp[1], (float)hp[2], (float)hp[3]);
    s[1 * (size_t)h->stride + r] = make_float4((float)hp[4], (float)hp[5], (float)hp[6], 0.f);
    s[3 * (size_t)h->stride + r] = make_float4(0.f, (float)hp[0], (float)hp[1], (float)hp[2]);
    if (h->fk) s[4 * (size_t)h->stride + r] = make_float4((float)hp[3], (float)hp[4], (float)hp[5], (float)hp[6]);
  }
  return s;
}

int upload_home(cdpr_engine* h) {
  if (h->fp64) return upload_home64(h);
  std::vector<float4> s = home_state(h);
  HIP_TRY(h, hipMemcpyAsync(h->d_state, s.data(), s.size() * sizeof(float4), hipMemcpyHostToDevice, h->stream));
  // observables before the first publish: the home pose, zeros elsewhere
  std::vector<float4> o((size_t)h->n_obs * h->stride, make_float4(0.f, 0.f, 0.f, 0.f));
  for (uint32_t r = 0; r < h->stride; ++r) {
    o[0 * (size_t)h->stride + r] = s[0 * (size_t)h->stride + r];
    o[1 * (size_t)h->stride + r] = s[1 * (size_t)h->stride + r];
  }
  HIP_TRY(h, hipMemcpyAsync(h->d_obs, o.data(), o.size() * sizeof(float4), hipMemcpyHostToDevice, h->stream));
  if (h->d_dbg) HIP_TRY(h, hipMemsetAsync(h->d_dbg, 0, (size_t)h->batch * CDPR_PID_DEBUG_AXES * sizeof(float), h->stream));
  for (int i = 0; i < 2; ++i) {  // latched and pending Joy buffers: target 0 after Load / reset
    HIP_TRY(h, hipMemsetAsync(h->d_vel[i], 0, (size_t)h->stride * h->n * sizeof(float), h->stream));
    HIP_TRY(h, hipMemsetAsync(h->d_pos[i], 0, (size_t)h->stride * h->n * sizeof(float), h->stream));
    HIP_TRY(h, hipMemsetAsync(h->d_frc[i], 0, (size_t)h->stride * h->n * sizeof(float), h->stream));
  }
  if (h->d_rec) HIP_TRY(h, hipMemsetAsync(h->d_rec, 0, h->glay.bytes(h->stride), h->stream));
  if (h->d_mode) HIP_TRY(h, hipMemsetAsync(h->d_mode, kModePosition, h->batch, h->stream));  // PLG.cpp:153-157 (call count 0)
  if (h->d_target) HIP_TRY(h, hipMemsetAsync(h->d_target, 0, (size_t)h->stride * h->n * sizeof(float), h->stream));  // target 0 after Load
  HIP_TRY(h, wait_stream(h));
  return CDPR_OK;
}

void free_all(cdpr_engine* h) {
  if (!h) return;
  (void)hipSetDevice(h->device);
  if (h->d_state) (void)hipFree(h->d_state);
  if (h->d_obs) (void)hipFree(h->d_obs);
  if (h->d_dbg) (void)hipFree(h->d_dbg);
  if (h->d_geom) (void)hipFree(h->d_geom);
  if (h->d_wtab) (void)hipFree(h->d_wtab);
  if (h->d_rec) (void)hipFree(h->d_rec);
  if (h->d_gwtab) (void)hipFree(h->d_gwtab);
  if (h->d_gptab) (void)hipFree(h->d_gptab);
  if (h->d_roll_rec) (void)hipFree(h->d_roll_rec);
  for (void* p64 : {(void*)h->d_roll64, (void*)h->d_roll64_acc, (void*)h->d_roll64_cmd, (void*)h->d_roll64_meta})
    if (p64) (void)hipFree(p64);
  if (h->d_mode) (void)hipFree(h->d_mode);
  if (h->d_target) (void)hipFree(h->d_target);
  for (int i = 0; i < 3; ++i)
    if (h->d_mask[i]) (void)hipFree(h->d_mask[i]);
  if (h->d_unpack) (void)hipFree(h->d_unpack);
  for (void* p64 : {(void*)h->d_state64, (void*)h->d_obs64, (void*)h->d_geom64, (void*)h->d_wtab64, (void*)h->d_dbg64, h->d_unpack64})
    if (p64) (void)hipFree(p64);
  if (h->h_fault) (void)hipHostFree(h->h_fault);
  if (h->h_pub) (void)hipHostFree(h->h_pub);
  if (h->h_pub64) (void)hipHostFree(h->h_pub64);
  if (h->h_pub_done) (void)hipHostFree(h->h_pub_done);
  if (h->d_pub_arrivals) (void)hipFree(h->d_pub_arrivals);
  if (h->d_roll_ref) (void)hipFree(h->d_roll_ref);
  if (h->d_roll_cost) (void)hipFree(h->d_roll_cost);
  for (int i = 0; i < 2; ++i) {
    if (h->d_vel[i]) (void)hipFree(h->d_vel[i]);
    if (h->d_pos[i]) (void)hipFree(h->d_pos[i]);
    if (h->d_frc[i]) (void)hipFree(h->d_frc[i]);
  }
  for (auto& g : h->graphs) {
    (void)hipGraphExecDestroy(g.exec);
    (void)hipGraphDestroy(g.graph);
  }
  if (h->copy_stream) (void)hipStreamSynchronize(h->copy_stream);
  for (int k = 0; k < 3; ++k) {
    for (int i = 0; i < 2; ++i) {
      if (h->h_stage[k][i]) (void)hipHostFree(h->h_stage[k][i]);
      if (h->stage_ev[k][i]) (void)hipEventDestroy(h->stage_ev[k][i]);
    }
    if (h->free_ev[k]) (void)hipEventDestroy(h->free_ev[k]);
  }
  if (h->copy_stream) (void)hipStreamDestroy(h->copy_stream);
  if (h->ev0) (void)hipEventDestroy(h->ev0);
  if (h->ev1) (void)hipEventDestroy(h->ev1);
  if (h->stream) (void)hipStreamDestroy(h->stream);
  delete h;
}

void engine_reset_host(cdpr_engine* h) {
  h->vel_pending = h->pos_pending = h->frc_pending = false;
  h->vel_masked = h->pos_masked = h->frc_masked = false;
  for (int k = 0; k < 3; ++k) h->sched_rows[k] = nullptr, h->sched_mask[k] = nullptr;
  if (h->h_fault) *h->h_fault = 0u;
  h->ext_vel[0] = h->ext_vel[1] = h->ext_pos[0] = h->ext_pos[1] = h->ext_frc[0] = h->ext_frc[1] = nullptr;
  h->have_vel = h->have_pos = h->have_frc = false;
  h->mode = kModePosition;  // PLG.cpp:153-157: Position mode, target 0 after operator= -> reset()
  h->step = 0;
  h->pid_calls = 0;
  h->prev_publish = 0.0;  // PLG.cpp:59
}

// Everything the copy stream still has in flight lands before the compute stream (or the host) touches a pending buffer
// in any other way than latching it (device-side staging, masked merges, resets).
static int drain_copy_stream(cdpr_engine* h) {
  if (h->copy_stream) HIP_TRY(h, hipStreamSynchronize(h->copy_stream));
  h->ready_wait[0] = h->ready_wait[1] = h->ready_wait[2] = nullptr;
  return CDPR_OK;
}

int stage_command(cdpr_engine* h, float* dst, const float* src, size_t count, bool from_device) {
  const size_t n = h->n, B = h->batch;
  if (!src) {
    h->err = "null command buffer";
    return CDPR_ERR_INVALID;
  }
  if (count != n * B && count != n) return CDPR_IGNORED;  // PLG.cpp:68-73,77-82: silently dropped
  if (set_device(h) != CDPR_OK) return CDPR_ERR_DEVICE;
  const size_t bytes = n * B * sizeof(float);
  if (from_device) {
    if (int rc = drain_copy_stream(h)) return rc;
    if (count == n * B) {
      HIP_TRY(h, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, h->stream));
    } else {
      std::vector<float> one(n), all(n * B);
      HIP_TRY(h, hipMemcpy(one.data(), src, n * sizeof(float), hipMemcpyDeviceToHost));
      for (size_t b = 0; b < B; ++b) memcpy(&all[b * n], one.data(), n * sizeof(float));
      HIP_TRY(h, hipMemcpyAsync(dst, all.data(), bytes, hipMemcpyHostToDevice, h->stream));
      HIP_TRY(h, wait_stream(h));
    }
    return CDPR_OK;
  }
  // host source: rows -> pinned staging -> pending device buffer on the copy stream, no wait for the launches in flight
  const int kind = (dst == h->d_vel[1]) ? 0 : (dst == h->d_pos[1]) ? 1 : 2;
  if (!h->copy_stream) HIP_TRY(h, hipStreamCreateWithFlags(&h->copy_stream, hipStreamNonBlocking));
  const int idx = (h->stage_idx[kind] ^= 1);
  if (!h->h_stage[kind][idx]) {
    HIP_TRY(h, hipHostMalloc((void**)&h->h_stage[kind][idx], bytes, hipHostMallocDefault));
    HIP_TRY(h, hipEventCreateWithFlags(&h->stage_ev[kind][idx], hipEventDisableTiming));
  }
  if (!h->free_ev[kind]) HIP_TRY(h, hipEventCreateWithFlags(&h->free_ev[kind], hipEventDisableTiming));
  if (h->stage_ev_set[kind][idx]) HIP_TRY(h, hipEventSynchronize(h->stage_ev[kind][idx]));  // two commands back: long done
  float* stage = h->h_stage[kind][idx];
  if (count == n * B) {
    memcpy(stage, src, bytes);  // the caller may reuse its buffer on return
  } else {
    for (size_t b = 0; b < B; ++b) memcpy(stage + b * n, src, n * sizeof(float));
  }
  if (h->free_ev_set[kind]) HIP_TRY(h, hipStreamWaitEvent(h->copy_stream, h->free_ev[kind], 0));
  HIP_TRY(h, hipMemcpyAsync(dst, stage, bytes, hipMemcpyHostToDevice, h->copy_stream));
  HIP_TRY(h, hipEventRecord(h->stage_ev[kind][idx], h->copy_stream));
  h->stage_ev_set[kind][idx] = true;
  h->ready_wait[kind] = h->stage_ev[kind][idx];
  return CDPR_OK;
}

// The Pid call counter only matters through calls != 0 and calls >= nbuf (<= 11 on the fast path): it saturates.  The
// ring position does not come from it but from the world step (StepArgs::ring_slot).
// ... of a ring of w errors: the first sample of a handle that runs without a Pid reset since Load is taken at world step 2

// Weights of the ring position a launch starts at, copied into its arguments (see StepArgs::wrow).
void set_weight_row(const cdpr_engine* h, StepArgs& a) {
  const int slot = a.ring_slot;
  // (per-robot handles: both Pids share one window, so the velocity Pid's table serves every lane)
  memcpy(a.wrow, &h->wtab_host[(h->per_robot || h->mode == kModeVelocity) ? 0 : 1][slot * (kWin + 2)], sizeof a.wrow);
}

// The kernel a launch uses: the routing is planned_kernel's (cdpr_select.hpp: a pure function of the handle's plan and the
// launch's shape, the same one cdpr_plan_kernel answers from without a GPU); here its answer becomes a function pointer.
LaunchShape launch_shape(const cdpr_engine* h, int k, bool steady) {
  LaunchShape s;
  s.steps = k;
  s.first_world = h->step == 0;
  s.scheduled = h->sched_refresh != 0;
  s.steady = steady;
  return s;
}
StepKernel step_kernel_of(const cdpr_engine* h, const PlannedKernel& pk) {
  const uint32_t n = h->n;
  switch (pk.id) {
    case KernelId::StepSingle: return pick_step_kernel(true, n, h->fk, h->td);
    case KernelId::StepMulti: return pick_step_kernel(false, n, h->fk, h->td);
    case KernelId::Lowreg: return pick_lowreg_kernel(n, h->td);
    case KernelId::OnestepPersist: return pick_onestep_persist_kernel(n, h->fk, h->td);
    case KernelId::Split: return pick_split_kernel(n);
    case KernelId::Onestep: return pick_onestep_kernel(n, h->fk, h->td);
    case KernelId::PhysStep: return pick_phys_kernel(n, h->fk, h->td, kPhysStep);
    case KernelId::Rollout: return pick_rollout_kernel(n, h->fk, h->td);
    case KernelId::PhysRollout: return pick_phys_kernel(n, h->fk, h->td, kPhysRollout);
    case KernelId::PrSingle: return pick_pr_kernel(true, false, false, n, h->fk, h->td);
    case KernelId::PrLowreg: return pick_pr_kernel(true, false, true, n, h->fk, h->td);
    case KernelId::PrSplit: return pick_pr_split_kernel(n);
    case KernelId::PrMulti: return pick_pr_kernel(false, false, false, n, h->fk, h->td);
    case KernelId::PrRollout: return pick_pr_kernel(false, true, false, n, h->fk, h->td);
    case KernelId::PairSingle: return pick_pair_kernel(true, n, h->fk, h->td);
    case KernelId::PairMulti: return pick_pair_kernel(false, n, h->fk, h->td);
    case KernelId::PairStream: return pick_pair_stream_kernel(n, h->mode == kModeVelocity);
    case KernelId::Cable: return pick_cable_kernel(n, h->fk, h->td);
    default: return nullptr;  // (general path and precision = 64: their own launch functions)
  }
}
StepKernel select_step_kernel(const cdpr_engine* h, int k, bool steady) { return step_kernel_of(h, planned_kernel(h->plan, launch_shape(h, k, steady))); }
// May a launch of k > 1 world steps with the arguments `a` (flags, pid_calls, pointers set) run on cdpr_pair_stream_kernel?
// That kernel has no branch for anything but the steady state of a plain handle (cdpr_step_kernel_pair.hpp): every
// condition below is one the general several-steps kernel tests per step instead (LaunchShape::steady).  CDPR_PAIR_STREAM=0: never (A/B).
bool pair_stream_steady(const cdpr_engine* h, const StepArgs& a) {
  return h->step != 0 && h->mode != kModeForce && a.pid_calls >= a.nbuf && h->cfg.publish_period == 0.0 && !a.dbg && !a.travel_on &&
         !(a.vel_limit > 0.f) && !a.unilateral && a.effort >= 0.f && a.clamp_cmd && !h->sched_ready;
}
uint32_t step_block_threads(const cdpr_engine* h, int k) { return planned_kernel(h->plan, launch_shape(h, k)).block; }

// The controller half of a general-path launch: records, latched commands, gains.
GenCtl general_ctl(const cdpr_engine* h) {
  GenCtl g{};
  g.rec = h->d_rec;
  g.rstride = h->stride;
  g.rec_bytes = (uint32_t)h->glay.bytes(h->stride);
  g.vel_cmd = h->have_vel ? (h->ext_vel[0] ? h->ext_vel[0] : h->d_vel[0]) : nullptr;
  g.pos_cmd = h->have_pos ? (h->ext_pos[0] ? h->ext_pos[0] : h->d_pos[0]) : nullptr;
  g.frc_cmd = h->have_frc ? (h->ext_frc[0] ? h->ext_frc[0] : h->d_frc[0]) : nullptr;
  g.mode_arr = h->per_robot ? h->d_mode : nullptr;
  g.wtab = h->d_gwtab;
  g.mode = h->mode;
  g.eps = (float)h->cfg.velocity_epsilon;
  g.dt = (float)h->cfg.dt;
  g.lay = h->glay;
  g.ptab = h->d_gptab;
  g.pcas_max = std::max(h->gpid[0].pcas, h->gpid[1].pcas);
  g.dcas_max = std::max(h->gpid[0].dcas, h->gpid[1].dcas);
  g.nbuf0 = h->gpid[0].nbuf, g.nbuf1 = h->gpid[1].nbuf;
  g.hot = h->gen_hot ? 1 : 0;
  {
    const GenPid &p0 = h->gpid[0], &p1 = h->gpid[1];
    const bool same_window = p0.nbuf == p1.nbuf && p0.degree == p1.degree;
    const bool clamps = p0.cmax > p0.cmin && p1.cmax > p1.cmin && p0.imax >= p0.imin && p1.imax >= p1.imin;
    g.simple_ok = (same_window && clamps && g.pcas_max == 0 && g.dcas_max == 0) ? 1 : 0;
  }
  return g;
}

// General controller path: ONE launch per `per_launch` world steps (cdpr_general_step.hpp); with `record`, the observable
// image of step j of the call goes to record + j * n_obs * stride.
int run_steps_general(cdpr_engine* h, int nsteps, int per_launch, float4* record) {
  if (h->step + (uint64_t)nsteps >= (1ull << 31)) {  // world-step stamps are int32 in the controller records
    h->err = "general controller path: world-step counter would pass 2^31";
    return CDPR_ERR_UNSUPPORTED;
  }
  StepArgs a = h->base;
  a.state = h->d_state;
  a.obs = h->d_obs;
  a.cmd = nullptr;
  a.dbg = h->dbg ? h->d_dbg : nullptr;
  a.geom = h->d_geom;
  a.batch = h->batch;
  a.stride = h->stride;
  a.pid_calls = 0;
  copy_pid(h->pid_pos, a);  // unused
  const size_t image = (size_t)h->n_obs * h->stride;
  a.obs_step_stride = record ? image : 0;
  GenCtl g = general_ctl(h);
  // where the role-split one-step kernel serves the handle, its launches beat the one-wave kernel's multi-step ones
  // (16 384 x 8: 9.2 against 13.7 us per step, 32 768: 10.1 against 21.6; same bits): fused updates and the trajectory
  // record then run as one-step launches
  if (h->gen_split || h->gen_lean) per_launch = 1;  // (likewise where the lean kernel + one-wave kernel pair steps the handle)
  int done = 0;
  while (done < nsteps) {
    const int k = std::min(per_launch, nsteps - done);
    // one step per launch on FK + TD handles up to two workgroups per CU: the role-split form (cdpr_general_split.hpp), beyond: the
    // lean role-split kernel (two waves per SIMD; the rare controller paths by call) - planned_kernel, cdpr_select.hpp
    const PlannedKernel pk = planned_kernel(h->plan, launch_shape(h, k));
    const bool gsplit = pk.id == KernelId::GenSplit, glean = pk.id == KernelId::GenLean;
    GenKernel kern = gsplit ? pick_gen_split11(h->n) : glean ? pick_gen_lean11(h->n) : pick_gen_kernel(h->n, h->fk, h->td, false, h->glay.nb > 11, pk.id == KernelId::GenOne);
    h->last_kernel = pk;
    a.nsteps = k;
    a.flags = (h->step == 0) ? kFlagFirstWorldStep : 0u;
    g.now_step = (int)h->step;
    if (record) a.obs = record + (size_t)done * image;
    a.publish_mask = 0;
    for (int j = 0; j < k; ++j) {  // PLG.cpp:236-242: strict '>' against the last published stamp
      const double now = sim_time(h->step + (uint64_t)j, h->cfg.dt);
      if ((now - h->prev_publish) > h->cfg.publish_period) {
        h->prev_publish = now;
        a.publish_mask |= (1ull << j);
      }
    }
    hipLaunchKernelGGL(kern, dim3((h->batch + 63u) / 64u), dim3((gsplit || glean) ? 128 : 64), 0, h->stream, a, g);
    HIP_TRY(h, hipGetLastError());
    ++h->launches;
    h->step += (uint64_t)k;
    done += k;
  }
  if (record && h->cfg.publish_period == 0.0 && h->step > 1)  // keep cdpr_get_* consistent: latest image into the engine's own
    HIP_TRY(h, hipMemcpyAsync(h->d_obs, record + (size_t)(nsteps - 1) * image, image * sizeof(float4), hipMemcpyDeviceToDevice, h->stream));
  return CDPR_OK;
}

// cdpr_create pays the cold costs of the handle's one-step kernel (the runtime loads a kernel's code object and sets up
// its argument buffers on the FIRST launch: measured ~1 ms), not the first cdpr_update: one launch of exactly that kernel
// over one workgroup of home-state robots in a scratch buffer, which is then freed.  The handle's own state is not touched.
int warm_first_launch(cdpr_engine* h) {
  if (h->general || h->fp64) return CDPR_OK;
  if (const char* w = std::getenv("CDPR_NO_WARM_LAUNCH"))
    if (w[0] == '1') return CDPR_OK;
  const uint32_t rows = 64;
  std::vector<float4> s((size_t)h->n_state * rows, make_float4(0.f, 0.f, 0.f, 0.f));
  const double* hp = h->cfg.home_pose;
  for (uint32_t r = 0; r < rows; ++r) {
    s[0 * (size_t)rows + r] = make_float4((float)hp[0], (float)hp[1], (float)hp[2], (float)hp[3]);
    s[1 * (size_t)rows + r] = make_float4((float)hp[4], (float)hp[5], (float)hp[6], 0.f);
    s[3 * (size_t)rows + r] = make_float4(0.f, (float)hp[0], (float)hp[1], (float)hp[2]);
    if (h->fk) s[4 * (size_t)rows + r] = make_float4((float)hp[3], (float)hp[4], (float)hp[5], (float)hp[6]);
  }
  DevBuf st, ob, cm, mt;
  HIP_TRY(h, mt.alloc(rows));
  HIP_TRY(h, hipMemsetAsync(mt.p, (int)(kMetaPosition | (20u << kMetaCallShift)), rows, h->stream));
  HIP_TRY(h, st.alloc(s.size() * sizeof(float4)));
  HIP_TRY(h, ob.alloc((size_t)h->n_obs * rows * sizeof(float4)));
  HIP_TRY(h, cm.alloc((size_t)rows * h->n * sizeof(float)));
  HIP_TRY(h, hipMemcpyAsync(st.p, s.data(), s.size() * sizeof(float4), hipMemcpyHostToDevice, h->stream));
  HIP_TRY(h, hipMemsetAsync(cm.p, 0, (size_t)rows * h->n * sizeof(float), h->stream));
  StepArgs a = h->base;
  a.state = st.as<float4>();
  a.obs = ob.as<float4>();
  a.cmd = cm.as<float>();
  a.dbg = nullptr;
  a.geom = h->d_geom;
  a.batch = h->lane_cable ? (h->n <= 4 ? 16u : 8u) : h->lane_pair ? 32u : 64u;
  a.stride = rows;
  a.nsteps = 1;
  a.obs_step_stride = 0;
  a.flags = 0u;
  a.publish_mask = 1ull;
  copy_pid(h->per_robot ? h->pid_vel : h->pid_pos, a);
  copy_pid_alt(h->pid_pos, a.alt);
  a.meta = mt.as<uint8_t>();
  a.pid_calls = kCallSat;  // steady state: every branch of the step is taken, as in the launches that follow
  a.ring_slot = 0;
  set_weight_row(h, a);
  hipLaunchKernelGGL(select_step_kernel(h, 1), dim3(1), dim3(step_block_threads(h, 1)), 0, h->stream, a);
  HIP_TRY(h, hipGetLastError());
  HIP_TRY(h, wait_stream(h));
  return CDPR_OK;
}

// One step-kernel launch over the whole batch, or (h->chunk) the same launch cut into contiguous blocks of robots issued
// back to back on the handle's stream: every pointer that is indexed by robot moves to the block's first robot, the row
// stride stays.  Robots are independent, so the results are bit-identical to the single launch (tested).
void launch_step(cdpr_engine* h, StepKernel kern, uint32_t robots_per_block, dim3 block, const StepArgs& a, uint32_t max_grid = 0) {
  const uint32_t B = a.batch;
  const uint32_t chunk = (h->chunk && B > h->chunk) ? h->chunk : B;
  const uint32_t pieces = (B + chunk - 1u) / chunk;
  const uint32_t each = (((B + pieces - 1u) / pieces) + 63u) & ~63u;  // balanced, whole wavefronts
  for (uint32_t first = 0; first < B; first += each) {
    StepArgs c = a;
    c.batch = std::min(each, B - first);
    c.state = a.state + first;
    c.obs = a.obs + first;
    if (a.cmd) c.cmd = a.cmd + (size_t)first * h->n;
    if (a.dbg) c.dbg = a.dbg + (size_t)first * CDPR_PID_DEBUG_AXES;
    if (a.meta) c.meta = a.meta + first;
    uint32_t grid = (c.batch + robots_per_block - 1u) / robots_per_block;
    if (max_grid && grid > max_grid) grid = max_grid;  // persistent kernel: the waves walk over the blocks
    hipLaunchKernelGGL(kern, dim3(grid), block, 0, h->stream, c);
    ++h->launches;
  }
}

int run_steps(cdpr_engine* h, int nsteps, int per_launch, float4* record = nullptr) {
  if (!h) return CDPR_ERR_INVALID;
  if (nsteps < 0 || per_launch < 1 || (per_launch > 64 && !h->sched_refresh)) {
    h->err = "nsteps must be >= 0 and steps_per_launch in 1..64";
    return CDPR_ERR_INVALID;
  }
  if (nsteps == 0) return CDPR_OK;
  if (set_device(h) != CDPR_OK) return CDPR_ERR_DEVICE;

  // --- PLG.cpp:206-219: latch pending commands, velocity first, then position
  const bool latch_kind[3] = {h->vel_pending, h->pos_pending, h->frc_pending};
  for (int k = 0; k < 3; ++k) {
    if (latch_kind[k] && h->ready_wait[k]) {  // a host Joy batch is (or was) on its way on the copy stream
      HIP_TRY(h, hipStreamWaitEvent(h->stream, h->ready_wait[k], 0));
      h->ready_wait[k] = nullptr;
    }
  }
  // After the latch below, what is then the PENDING buffer of a kind whose buffers were swapped (or, on per-robot handles,
  // read by the latch kernel) was last touched by the work queued on the compute stream so far: the next host batch of
  // that kind, which travels on the copy stream, may overwrite it only once that work is through.  The event is recorded
  // on EVERY such latch once a copy stream exists, however the latched command itself arrived (host, _device or masked):
  // a stale event from an earlier host batch would let the copy run into launches that still read the buffer.  Bound
  // commands swap nothing and handles that never saw a host batch have no copy stream: no event, no device time.
  bool touched[3] = {false, false, false};
  auto mark_free = [&]() -> int {
    for (int k = 0; k < 3; ++k) {
      if (!touched[k] || !h->copy_stream) continue;
      if (!h->free_ev[k]) HIP_TRY(h, hipEventCreateWithFlags(&h->free_ev[k], hipEventDisableTiming));
      HIP_TRY(h, hipEventRecord(h->free_ev[k], h->stream));
      h->free_ev_set[k] = true;
    }
    return CDPR_OK;
  };
  bool reset_pid = false;
  auto flush_hot = [&]() {  // the general path's hot rows back into the H slots (cdpr_gen_flush_hot_kernel) before a Pid's records are reset
    if (!h->d_rec || !h->gen_hot) return;
    GenFlushArgs fa{};
    fa.rec = h->d_rec, fa.rstride = h->stride, fa.batch = h->batch, fa.lay = h->glay, fa.nbuf = std::max(h->gpid[0].nbuf, 1);
    hipLaunchKernelGGL(cdpr_gen_flush_hot_kernel, dim3((h->batch + 255u) / 256u), dim3(256), 0, h->stream, fa);
  };
  auto reset_block = [&](int which) -> hipError_t {  // Pid::reset of one Pid of every cable (general path): its slots, its rows
    const GenLayout& L = h->glay;
    flush_hot();
    hipError_t e1 = hipMemsetAsync(h->d_rec + (size_t)L.block_a(which, 0) * h->stride * 4, 0, (size_t)L.pid_slots() * h->stride * 16, h->stream);
    if (e1 != hipSuccess || L.pid_rows() == 0) return e1;
    return hipMemsetAsync(h->d_rec + ((size_t)L.slots() * 4 + (size_t)L.block_b(which, 0)) * h->stride, 0, (size_t)L.pid_rows() * h->stride * 4, h->stream);
  };
  auto reset_block64 = [&](int which) -> hipError_t {  // precision = 64 with the hold branch: Pid::reset of one Pid of every cable (its rows behind the state)
    for (uint32_t i = 0; i < h->n; ++i) {
      hipError_t e1 = hipMemsetAsync(h->d_state64 + (size_t)f64_hold_row((int)h->n, (int)i, which, h->hold_win) * h->stride, 0, (size_t)hold_pid_rows(h->hold_win) * h->stride * sizeof(double),
                                     h->stream);
      if (e1 != hipSuccess) return e1;
    }
    return hipSuccess;
  };
  if (h->per_robot) {
    // every robot has its own mode: commands (masked or not) are latched on the device, robot by robot
    auto latch = [&](float* pending, float* latched, const uint8_t* mask, int which, int new_mode) -> int {
      if (h->fp64) {  // precision = 64: the same on the double state (integral rows 20 + 11 i + 10)
        LatchF64Args lf{};
        lf.mask = mask;
        lf.meta = h->d_mode;
        lf.pending = pending;
        lf.target = h->d_target;
        lf.state = h->d_state64;
        lf.stride = h->stride;
        lf.batch = h->batch;
        lf.n = h->n;
        lf.new_mode = (new_mode == kModeVelocity) ? kMetaVelocity : (new_mode == kModePosition) ? kMetaPosition : kMetaForce;
        lf.hold = h->hold64 ? 1u : 0u;
        lf.win = (uint32_t)h->win64;
        lf.hold_win = (uint32_t)h->hold_win;
        hipLaunchKernelGGL(cdpr_latch_f64_kernel, dim3((h->batch + 255u) / 256u), dim3(256), 0, h->stream, lf);
        HIP_TRY(h, hipGetLastError());
        return CDPR_OK;
      }
      if (!h->general) {  // register-resident path: one active target row and one Pid record per robot
        LatchFastArgs lf{};
        lf.mask = mask;
        lf.meta = h->d_mode;
        lf.pending = pending;
        lf.target = h->d_target;
        lf.hot = h->d_state + (size_t)(plat_slots(h->fk) + 5 * cable_pairs((int)h->n)) * h->stride;
        lf.stride = h->stride;
        lf.batch = h->batch;
        lf.n = h->n;
        lf.hot_rows = (uint32_t)((cable_pairs((int)h->n) + 1) / 2);
        lf.new_mode = (new_mode == kModeVelocity) ? kMetaVelocity : (new_mode == kModePosition) ? kMetaPosition : kMetaForce;
        hipLaunchKernelGGL(cdpr_latch_fast_kernel, dim3((h->batch + 255u) / 256u), dim3(256), 0, h->stream, lf);
        HIP_TRY(h, hipGetLastError());
        return CDPR_OK;
      }
      flush_hot();
      GenLatchArgs la{};
      la.mask = mask;
      la.mode = h->d_mode;
      la.pending = pending;
      la.latched = latched;
      la.rec = h->d_rec;
      la.rstride = h->stride;
      la.batch = h->batch;
      la.n = h->n;
      la.reset_pid = which;  // (-1: setForce resets no Pid)
      la.lay = h->glay;
      la.new_mode = new_mode;
      hipLaunchKernelGGL(cdpr_gen_latch_kernel, dim3((h->batch + 255u) / 256u), dim3(256), 0, h->stream, la);
      HIP_TRY(h, hipGetLastError());
      return CDPR_OK;
    };
    // (a batch of a device-resident schedule, cdpr_update_scheduled_kind, is latched from the caller's buffers in place)
    auto rows_of = [&](int kind, float* pending) -> float* { return h->sched_rows[kind] ? const_cast<float*>(h->sched_rows[kind]) : pending; };
    auto mask_of = [&](int kind, bool masked) -> const uint8_t* { return h->sched_rows[kind] ? h->sched_mask[kind] : (masked ? h->d_mask[kind] : nullptr); };
    if (h->vel_pending) {
      if (int rc = latch(rows_of(0, h->d_vel[1]), h->d_vel[0], mask_of(0, h->vel_masked), 1, kModeVelocity)) return rc;
      h->vel_pending = h->vel_masked = false;
      h->have_vel = true;
      touched[0] = !h->sched_rows[0];  // the latch kernel reads the pending buffer
    }
    if (h->pos_pending) {
      if (int rc = latch(rows_of(1, h->d_pos[1]), h->d_pos[0], mask_of(1, h->pos_masked), 0, kModePosition)) return rc;
      h->pos_pending = h->pos_masked = false;
      h->have_pos = true;
      touched[1] = !h->sched_rows[1];
    }
    if (h->frc_pending) {  // [NEW] ordering: after the two Joy topics (the reference has no force callback)
      if (int rc = latch(rows_of(2, h->d_frc[1]), h->d_frc[0], mask_of(2, h->frc_masked), -1, kModeForce)) return rc;
      h->frc_pending = h->frc_masked = false;
      h->have_frc = true;
      touched[2] = !h->sched_rows[2];
    }
    h->sched_rows[0] = h->sched_rows[1] = h->sched_rows[2] = nullptr;
    if (int rc = mark_free()) return rc;
    touched[0] = touched[1] = touched[2] = false;  // recorded; nothing below latches on a per-robot handle
    if (h->general) return run_steps_general(h, nsteps, per_launch, record);
    if (h->fp64) return run_steps_f64(h, nsteps, per_launch, false, reinterpret_cast<double*>(record));
  }
  if (h->vel_pending) {
    if (h->ext_vel[1]) {  // bound caller buffer: latched by pointer, nothing copied
      h->ext_vel[0] = h->ext_vel[1];
      h->ext_vel[1] = nullptr;
    } else {
      std::swap(h->d_vel[0], h->d_vel[1]);
      h->ext_vel[0] = nullptr;
      touched[0] = true;  // what is pending now was the latched buffer of the launches queued so far
    }
    h->vel_pending = false;
    h->have_vel = true;
    reset_pid = (h->mode != kModeVelocity);  // JFC.cpp:113-115
    if (h->general && reset_pid) HIP_TRY(h, reset_block(1));
    if (h->hold64 && reset_pid) HIP_TRY(h, reset_block64(1));
    h->mode = kModeVelocity;
  }
  if (h->pos_pending) {
    if (h->ext_pos[1]) {
      h->ext_pos[0] = h->ext_pos[1];
      h->ext_pos[1] = nullptr;
    } else {
      std::swap(h->d_pos[0], h->d_pos[1]);
      h->ext_pos[0] = nullptr;
      touched[1] = true;
    }
    h->pos_pending = false;
    h->have_pos = true;
    reset_pid = (h->mode != kModePosition);  // JFC.cpp:101-103 (fast path: the single record now belongs to the position Pid)
    if (h->general && reset_pid) HIP_TRY(h, reset_block(0));
    if (h->hold64 && reset_pid) HIP_TRY(h, reset_block64(0));
    h->mode = kModePosition;
  }
  if (h->frc_pending) {  // setForce (JFC.h:92-95): mode Force, no Pid is reset.  [NEW] ordering: after the two Joy topics
    if (h->ext_frc[1]) {
      h->ext_frc[0] = h->ext_frc[1];
      h->ext_frc[1] = nullptr;
    } else {
      std::swap(h->d_frc[0], h->d_frc[1]);
      h->ext_frc[0] = nullptr;
      touched[2] = true;
    }
    h->frc_pending = false;
    h->have_frc = true;
    h->mode = kModeForce;
  }
  if (int rc = mark_free()) return rc;
  if (h->general) return run_steps_general(h, nsteps, per_launch, record);
  if (h->fp64) return run_steps_f64(h, nsteps, per_launch, reset_pid, reinterpret_cast<double*>(record));

  if (reset_pid) {  // Pid::reset (Pid.cpp:100-115): zero every controller record; rare, so done outside the step kernel
    h->pid_calls = 0;
    const int P = plat_slots(h->fk);
    HIP_TRY(h, hipMemsetAsync(h->d_state + (size_t)P * h->stride, 0, (size_t)ctrl_slots((int)h->n) * h->stride * sizeof(float4), h->stream));
  }
  StepArgs a = h->base;
  a.state = h->d_state;
  a.obs = h->d_obs;
  a.dbg = h->dbg ? h->d_dbg : nullptr;
  a.geom = h->d_geom;
  a.batch = h->batch;
  a.stride = h->stride;
  // trajectory record: the observable image of step j of the call goes to record + j * n_obs * stride
  const size_t image = (size_t)h->n_obs * h->stride;
  a.obs_step_stride = record ? image : 0;
  if (h->per_robot) {
    copy_pid(h->pid_vel, a);
    copy_pid_alt(h->pid_pos, a.alt);
    a.cmd = h->d_target;
    a.meta = h->d_mode;
  } else if (h->mode == kModeVelocity) {
    copy_pid(h->pid_vel, a);
    a.cmd = h->ext_vel[0] ? h->ext_vel[0] : h->d_vel[0];
  } else if (h->mode == kModeForce) {
    copy_pid(h->pid_pos, a);  // unused: no Pid runs in Force mode
    a.cmd = h->ext_frc[0] ? h->ext_frc[0] : h->d_frc[0];
  } else {
    copy_pid(h->pid_pos, a);
    a.cmd = h->ext_pos[0] ? h->ext_pos[0] : h->d_pos[0];  // all zeros until the first jointPositions message: target 0 (PLG.cpp:153-157)
  }
  const uint32_t robots_per_block = h->lane_cable ? (h->n <= 4 ? 16u : 8u) : h->lane_pair ? 32u : 64u;
  const dim3 grid((h->batch + robots_per_block - 1u) / robots_per_block);

  constexpr int kGraphChunk = 10;  // launches per captured graph = one ring period: a chain ends on the ring slot it started from.
                                   // (Ten, not more: a caller that refreshes its Joy batch every 10 steps - bench.py, the reference's
                                   // 100 Hz publishers against the 1 kHz world - hands over 10 steps per call.)
  int done = 0;
  while (done < nsteps) {
    const int k = std::min(per_launch, nsteps - done);
    a.nsteps = k;
    a.flags = h->per_robot ? 0u : (h->mode == kModeVelocity ? kFlagActualIsVelocity : h->mode == kModeForce ? kFlagForceMode : 0u);
    const bool first_world = (h->step == 0);
    if (first_world) a.flags |= kFlagFirstWorldStep;
    // the kernel uses calls != 0 and calls >= nbuf (<= 11): the count saturates, and the ring position follows the world
    // step, so steady-state launch sequences repeat with period 10
    a.pid_calls = sat_pid_calls(h->pid_calls);
    a.ring_slot = ring_slot_of(h->step);
    set_weight_row(h, a);
    // a launch over a command schedule always runs on the several-steps kernel, even for one step: only that one reads the
    // schedule, its mailbox and kFlagPublishAll
    const int kk = h->sched_refresh ? std::max(k, 2) : k;
    const PlannedKernel pk = planned_kernel(h->plan, launch_shape(h, kk, pair_stream_steady(h, a)));
    StepKernel kern = step_kernel_of(h, pk);
    const dim3 block(pk.block);
    const bool stream = pk.id == KernelId::PairStream;  // steady state of a plain lane-pair handle: the branch-free several-steps kernel
    auto weights_for = [&](StepArgs& x) {                // ... which reads the weights by AGE from the row of ring position 0
      const int slot = x.ring_slot;
      if (stream) x.ring_slot = 0;
      set_weight_row(h, x);
      x.ring_slot = slot;
    };
    if (stream) weights_for(a);
    h->last_kernel = pk;

    // Steady state (every step published, derivative window full, not t = 0): the next launches are
    // byte-identical, so replay them from a captured hipGraph instead of paying a host launch each.
    // (measured on MI355X: 3.57 -> 3.41 us/step at 4 096 x 4 cables; at 65 536 x 8 cables the 15 us kernels already
    // hide the host launch and the replay's fixed cost makes it 2 % slower, so only small batches use it)
    // (the ring position advances with every step, so a captured chain is only valid from the position it was captured at:
    //  part of the cache key; the call count must be saturated (per-robot handles keep theirs on the device))
    if (record) a.obs = record + (size_t)done * image;
    const bool steady = !record && !h->sched_refresh && h->use_graphs && (size_t)h->batch * h->n <= 131072u && !first_world && (h->per_robot || a.pid_calls == kCallSat || h->mode == kModeForce) &&
                        h->cfg.publish_period == 0.0 && (nsteps - done) >= kGraphChunk * k;
    if (steady) {
      a.publish_mask = (k >= 64) ? ~0ull : ((1ull << k) - 1ull);
      cdpr_engine::GraphEntry* ge = nullptr;
      for (auto& g : h->graphs)
        if (g.kern == (void*)kern && g.cmd == a.cmd && g.steps_per_launch == k && g.flags == a.flags && g.start_slot == a.ring_slot) ge = &g;
      if (!ge) {
        cdpr_engine::GraphEntry g{(void*)kern, a.cmd, k, kGraphChunk, a.ring_slot, a.flags, nullptr, nullptr};
        // any failure inside the capture: end it, drop the partial graph, stop using graphs on this handle and
        // fall through to the eager launches below (the stream must never be left capturing)
        bool captured = hipStreamBeginCapture(h->stream, hipStreamCaptureModeThreadLocal) == hipSuccess;
        if (captured) {
          bool launched = true;
          for (int j = 0; j < kGraphChunk; ++j) {
            StepArgs aj = a;  // each node carries its own ring position
            aj.ring_slot = (a.ring_slot + j * k) % kWin;
            weights_for(aj);
            hipLaunchKernelGGL(kern, grid, block, 0, h->stream, aj);
            launched = launched && (hipGetLastError() == hipSuccess);
          }
          captured = (hipStreamEndCapture(h->stream, &g.graph) == hipSuccess) && launched && g.graph;
          if (captured && hipGraphInstantiate(&g.exec, g.graph, nullptr, nullptr, 0) != hipSuccess) captured = false;
          if (!captured && g.graph) (void)hipGraphDestroy(g.graph);
        }
        if (!captured) {
          (void)hipGetLastError();
          h->use_graphs = false;
          continue;  // same `done`: this chunk is launched eagerly on the next pass
        }
        if (h->graphs.size() >= 32) {  // small cache (two Joy buffers x a few ring positions x step counts): drop the oldest
          (void)hipGraphExecDestroy(h->graphs.front().exec);
          (void)hipGraphDestroy(h->graphs.front().graph);
          h->graphs.erase(h->graphs.begin());
        }
        h->graphs.push_back(g);
        ge = &h->graphs.back();
      }
      HIP_TRY(h, hipGraphLaunch(ge->exec, h->stream));
      const int steps = kGraphChunk * k;
      h->launches += kGraphChunk;
      h->step += (uint64_t)steps;
      if (h->mode != kModeForce) h->pid_calls = sat_pid_calls(h->pid_calls + steps);  // (no Pid call in Force mode)
      h->prev_publish = sim_time(h->step - 1, h->cfg.dt);
      done += steps;
      continue;
    }

    a.publish_mask = 0;
    if (h->sched_refresh) {  // a whole schedule in one launch (publish_period == 0: every step but world step 0 is published)
      a.flags |= kFlagPublishAll;
      a.sched_refresh = h->sched_refresh;
      a.sched_stride = (size_t)h->batch * h->n;
      a.sched_ready = h->sched_ready;
      a.fault = h->d_fault;
      h->prev_publish = sim_time(h->step + (uint64_t)k - 1, h->cfg.dt);
    } else {
      for (int j = 0; j < k; ++j) {  // PLG.cpp:236-242: strict '>' against the last published stamp
        const double now = sim_time(h->step + (uint64_t)j, h->cfg.dt);
        if ((now - h->prev_publish) > h->cfg.publish_period) {
          h->prev_publish = now;
          a.publish_mask |= (1ull << j);
        }
      }
    }
    launch_step(h, kern, robots_per_block, block, a, (h->persist && k == 1 && !h->per_robot) ? h->persist_grid : 0u);
    HIP_TRY(h, hipGetLastError());
    h->step += (uint64_t)k;
    if (h->mode != kModeForce) h->pid_calls = sat_pid_calls(h->pid_calls + k - (first_world ? 1 : 0));
    done += k;
  }
  if (record && h->cfg.publish_period == 0.0 && h->step > 1)  // keep cdpr_get_* consistent: latest image into the engine's own
    HIP_TRY(h, hipMemcpyAsync(h->d_obs, record + (size_t)(nsteps - 1) * image, image * sizeof(float4), hipMemcpyDeviceToDevice, h->stream));
  return CDPR_OK;
}

int fetch_slots(cdpr_engine* h, const float4* dsrc, int nslots, std::vector<float4>& host) {
  host.resize((size_t)nslots * h->stride);
  HIP_TRY(h, hipMemcpyAsync(host.data(), dsrc, host.size() * sizeof(float4), hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(h, wait_stream(h));
  return CDPR_OK;
}

// One field group of every robot out of a slot-row buffer into a caller's robot-major host array: a device-side gather
// into the read-out scratch, then one contiguous copy.  fields = (slot, component) per output column.
int fetch_fields(cdpr_engine* h, const float4* rows, const std::vector<std::pair<int, int>>& fields, void* host_out, uint32_t as_int = 0) {
  if (!host_out) return CDPR_OK;
  const size_t count = (size_t)h->batch * fields.size();
  if (h->unpack_cap < count) {
    HIP_TRY(h, wait_stream(h));
    if (h->d_unpack) (void)hipFree(h->d_unpack);
    h->d_unpack = nullptr;
    h->unpack_cap = 0;
    HIP_TRY(h, hipMalloc(&h->d_unpack, count * sizeof(float)));
    h->unpack_cap = count;
  }
  UnpackArgs u{};
  u.rows = rows;
  u.out = h->d_unpack;
  u.stride = h->stride;
  u.batch = h->batch;
  u.width = (uint32_t)fields.size();
  u.as_int = as_int;
  for (size_t j = 0; j < fields.size(); ++j) {
    u.slot[j] = (uint8_t)fields[j].first;
    u.comp[j] = (uint8_t)fields[j].second;
  }
  hipLaunchKernelGGL(cdpr_unpack_kernel, dim3((uint32_t)((count + 255) / 256)), dim3(256), 0, h->stream, u);
  HIP_TRY(h, hipGetLastError());
  HIP_TRY(h, hipMemcpyAsync(host_out, h->d_unpack, count * sizeof(float), hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(h, wait_stream(h));  // the scratch is reused by the next call
  return CDPR_OK;
}

// pose7 = slot 0 xyzw, slot 1 xyz; twist6 = slot 1 w, slot 2 xyzw, slot 3 x
const std::vector<std::pair<int, int>> kPoseFields = {{0, 0}, {0, 1}, {0, 2}, {0, 3}, {1, 0}, {1, 1}, {1, 2}};
const std::vector<std::pair<int, int>> kTwistFields = {{1, 3}, {2, 0}, {2, 1}, {2, 2}, {2, 3}, {3, 0}};

int fetch_platform(cdpr_engine* h, const float4* rows, float* pose7, float* twist6) {
  int rc = fetch_fields(h, rows, kPoseFields, pose7);
  if (rc != CDPR_OK) return rc;
  return fetch_fields(h, rows, kTwistFields, twist6);
}

inline float comp(const float4& v, int k) { return k == 0 ? v.x : (k == 1 ? v.y : (k == 2 ? v.z : v.w)); }

// Wait for the handle's stream with the host in the loop in mind (update -> synchronize -> read observables, every
// step): poll hipStreamQuery for the first kSpinUs microseconds (the blocking wait's wake-up costs 15-25 us, which is two
// step kernels; measured by scripts/short_run_probe.py), then hand over to the blocking hipStreamSynchronize so that long
// waits do not burn a core.  CDPR_SYNC_SPIN_US overrides (0 = always block).
hipError_t wait_stream(cdpr_engine* h) {
  static const long spin_us = [] {
    const char* v = std::getenv("CDPR_SYNC_SPIN_US");
    return v ? std::atol(v) : 2000L;
  }();
  if (spin_us > 0) {
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
      const hipError_t q = hipStreamQuery(h->stream);
      if (q == hipSuccess) return hipSuccess;
      if (q != hipErrorNotReady) return q;
      if (std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count() >= spin_us) break;
    }
  }
  return hipStreamSynchronize(h->stream);
}

// A schedule mailbox that never delivered (cdpr_update_scheduled with d_ready): the waiting kernel gave up after its poll
// budget, raised the handle's status word and went on with whatever the schedule held - the trajectory is not what the
// caller asked for, and every call that hands results out says so until cdpr_reset.
int check_fault(cdpr_engine* h) {
  if (h->h_fault && *(volatile uint32_t*)h->h_fault != 0u) {
    h->err = "a command schedule's mailbox timed out (d_ready never became non-zero): the steps since are not the scheduled trajectory; cdpr_reset clears this";
    return CDPR_ERR_DEVICE;
  }
  return CDPR_OK;
}

// every path that hands results out ends here (include/cdpr.h: after a mailbox timeout cdpr_synchronize, the getters and
// cdpr_device_download return CDPR_ERR_DEVICE until cdpr_reset - the fp64 read-outs and the FK / TD / limit / debug getters too)
int checked(cdpr_engine* h, int rc) { return rc != CDPR_OK ? rc : check_fault(h); }

}  // namespace cdpr_host

// =================================================================================
// C-ABI
// =================================================================================
extern "C" {

uint32_t cdpr_abi_version(void) { return CDPR_ABI_VERSION; }
size_t cdpr_config_size(void) { return sizeof(cdpr_config_t); }

int cdpr_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int cdpr_device_pci_bus_id(int device, char* out, size_t len) {
  if (!out || len < 13) return CDPR_ERR_INVALID;
  out[0] = 0;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || device < 0 || device >= n) return CDPR_ERR_INVALID;
  if (hipDeviceGetPCIBusId(out, (int)len, device) != hipSuccess) return CDPR_ERR_DEVICE;
  return CDPR_OK;
}

size_t cdpr_bytes_per_state_step(const cdpr_config_t* cfg) {
  // read command n; read+write platform 13 and controller 12 per cable;
  // write observables 13 + 3n.  4 * (39 + 28 n).
  if (!cfg) return 0;
  return 4u * (39u + 28u * (size_t)cfg->n_cables);
}

int cdpr_derivative_weights(uint32_t n, uint32_t degree, double* w) {
  if (!w) return CDPR_ERR_INVALID;
  return derivative_weights(n, degree, w);
}

const char* cdpr_last_error(cdpr_handle_t h) { return h ? h->err.c_str() : g_create_error.c_str(); }

int cdpr_create(const cdpr_config_t* cfg, int device, cdpr_handle_t* out) {
  if (out) *out = nullptr;
  if (!cfg || !out) {
    g_create_error = "null argument";
    return CDPR_ERR_INVALID;
  }
  // the routing of this configuration (which kernel family, which mapping): cdpr_select.hpp, a pure function of the
  // configuration shared with cdpr_plan_kernel; 256 CUs assumed until the device is known (re-planned below)
  KernelPlan plan = plan_kernels(*cfg);
  if (plan.rc != CDPR_OK) {
    g_create_error = plan.error;
    return plan.rc;
  }
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev <= 0) {
    g_create_error = "no HIP device visible (this engine has no CPU path)";
    return CDPR_ERR_DEVICE;
  }
  if (device < 0 || device >= ndev) {
    g_create_error = "device index out of range";
    return CDPR_ERR_INVALID;
  }
  cdpr_engine* h = new cdpr_engine();
  h->cfg = *cfg;
  h->device = device;
  h->n = cfg->n_cables;
  h->batch = (uint32_t)cfg->batch;
  h->stride = (h->batch + 63u) & ~63u;
  // A robot's rows lie `stride` x 16 B apart.  When that is a multiple of 2 MiB (131 072 robots) every row of a wavefront
  // maps to the same HBM channels and the launch loses 17 % (131 072 x 8: 29.8 -> 24.7 us per step, 262 144: 46-54 -> 43.3;
  // profiles/r04_stride_padding.txt: this was most of "the 131 072 anomaly").  Any pad of 64 .. 1 088 columns cures it
  // alike; smaller batches are not padded (16 384: a pad costs 3-4 %).  CDPR_STRIDE_PAD=<columns> forces a pad (A/B).
  if (h->stride % 131072u == 0u) h->stride += 128u;
  if (const char* sp = std::getenv("CDPR_STRIDE_PAD")) h->stride = ((h->batch + 63u) & ~63u) + ((uint32_t)(std::max(0L, std::atol(sp)) + 63L) & ~63u);
  {
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || cus <= 0) cus = 256;
    plan = plan_kernels(*cfg, cus);  // (the general path's role-split limit follows the device's CU count)
    h->cus = cus;
  }
  h->plan = plan;
  const bool general = plan.general;
  h->fk = plan.fk;
  h->td = plan.td;
  h->dbg = (cfg->stages & CDPR_STAGE_PID_DEBUG) != 0;
  h->general = plan.general;  // (precision = 64 with the hold branch: the fp64 kernel's HOLD instantiations, not the fp32 general path)
  h->fp64 = plan.fp64;
  h->hold64 = plan.hold64;
  h->tstop64 = plan.tstop64;
  h->win64 = plan.long64 ? kWinLong : kWin;
  h->hold_win = plan.hold_long ? kHoldWinLong : kHoldWin;
  h->per_robot = plan.per_robot;
  h->phys = plan.phys;
  h->lane_pair = plan.lane_pair;
  h->lane_cable = plan.lane_cable;
  h->lowreg = plan.lowreg;
  h->chunk = plan.chunk;
  h->persist = plan.persist;
  if (h->persist) {
    h->persist_grid = (uint32_t)h->cus * 4u;
    if (const char* pg = std::getenv("CDPR_PERSIST_GRID")) h->persist_grid = (uint32_t)std::max(1L, std::atol(pg));
  }
  h->onestep_v2 = plan.onestep_v2;
  h->split = plan.split;
  h->gen_split = plan.gen_split;
  h->gen_lean = plan.gen_lean;
  h->gen_hot = plan.gen_hot;
  h->pair_stream = plan.pair_stream;
  h->n_state = general ? plat_slots(h->fk) : state_slots((int)h->n, h->fk);
  h->n_obs = obs_slots((int)h->n);
  memset(&h->base, 0, sizeof h->base);
  fill_consts(*cfg, h->base);
  auto& wtab_host = h->wtab_host;
  fill_pid(cfg->velocity_pid, cfg->dt, h->pid_vel, wtab_host[0]);
  fill_pid(cfg->position_pid, cfg->dt, h->pid_pos, wtab_host[1]);
  engine_reset_host(h);
  {
    const char* ng = std::getenv("CDPR_NO_GRAPH");
    h->use_graphs = !(ng && ng[0] == '1');
  }

  auto fail = [&](const char* what, hipError_t code) {
    g_create_error = std::string(what) + ": " + hipGetErrorString(code);
    free_all(h);
    return code == hipErrorOutOfMemory ? CDPR_ERR_NOMEM : CDPR_ERR_DEVICE;
  };
  if ((e = hipSetDevice(device)) != hipSuccess) return fail("hipSetDevice", e);
  if ((e = hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking)) != hipSuccess) return fail("hipStreamCreate", e);
  if ((e = hipEventCreate(&h->ev0)) != hipSuccess) return fail("hipEventCreate", e);
  if ((e = hipEventCreate(&h->ev1)) != hipSuccess) return fail("hipEventCreate", e);
  const size_t slot_bytes = (size_t)h->stride * sizeof(float4);
  if (h->fp64) {
    const size_t row = (size_t)h->stride * sizeof(double);
    if ((e = hipMalloc(&h->d_state64, row * state64_rows(h))) != hipSuccess) return fail("hipMalloc(state64)", e);
    if ((e = hipMalloc(&h->d_obs64, row * f64_obs_rows((int)h->n))) != hipSuccess) return fail("hipMalloc(obs64)", e);
    const int W64 = h->win64;  // ring length of this handle's fp64 kernels (10, or 31 with windows of 12 .. 32 samples)
    std::vector<double> g((size_t)h->n * 7), wt((size_t)2 * W64 * (W64 + 2), 0.0);
    for (uint32_t i = 0; i < h->n; ++i) {
      for (int k = 0; k < 3; ++k) {
        g[(size_t)i * 7 + k] = cfg->frame_anchor[i][k];
        g[(size_t)i * 7 + 3 + k] = cfg->platform_anchor[i][k];
      }
      g[(size_t)i * 7 + 6] = cfg->cable_ref_length[i];
    }
    const cdpr_pid_params_t* pids[2] = {&cfg->velocity_pid, &cfg->position_pid};
    for (int t = 0; t < 2; ++t) {  // the rotated weight tables of fill_pid, in double
      double w[CDPR_MAX_D_BUFFER];
      std::vector<double> wpad((size_t)W64 + 1, 0.0);
      if (derivative_weights(pids[t]->d_buffer_length, pids[t]->d_degree, w) == CDPR_OK && pids[t]->d_buffer_length <= (uint32_t)W64 + 1)
        for (uint32_t j = 0; j < pids[t]->d_buffer_length; ++j) wpad[(size_t)W64 + 1 - pids[t]->d_buffer_length + j] = w[j];
      double* tab = &wt[(size_t)t * W64 * (W64 + 2)];
      for (int ws = 0; ws < W64; ++ws) {
        for (int sl = 0; sl < W64; ++sl) {
          int j = ((ws - sl) % W64 + W64) % W64;
          if (j == 0) j = W64;
          tab[ws * (W64 + 2) + sl] = wpad[(size_t)W64 - j];
        }
        tab[ws * (W64 + 2) + W64] = wpad[(size_t)W64];
      }
    }
    if ((e = hipMalloc(&h->d_geom64, g.size() * sizeof(double))) != hipSuccess) return fail("hipMalloc(geom64)", e);
    if ((e = hipMemcpy(h->d_geom64, g.data(), g.size() * sizeof(double), hipMemcpyHostToDevice)) != hipSuccess) return fail("hipMemcpy(geom64)", e);
    if ((e = hipMalloc(&h->d_wtab64, wt.size() * sizeof(double))) != hipSuccess) return fail("hipMalloc(wtab64)", e);
    if ((e = hipMemcpy(h->d_wtab64, wt.data(), wt.size() * sizeof(double), hipMemcpyHostToDevice)) != hipSuccess) return fail("hipMemcpy(wtab64)", e);
    if (h->dbg && (e = hipMalloc(&h->d_dbg64, (size_t)h->batch * CDPR_PID_DEBUG_AXES * sizeof(double))) != hipSuccess) return fail("hipMalloc(dbg64)", e);
    F64Args& b = h->base64;
    memset(&b, 0, sizeof b);
    b.state = h->d_state64; b.obs = h->d_obs64; b.dbg = h->d_dbg64; b.geom = h->d_geom64;
    b.batch = h->batch; b.stride = h->stride;
    b.fk = h->fk ? 1 : 0; b.td = h->td ? 1 : 0;
    b.dt = cfg->dt; b.half_dt = 0.5 * cfg->dt; b.inv_mass = 1.0 / cfg->mass;
    b.fgx = cfg->mass * cfg->gravity[0]; b.fgy = cfg->mass * cfg->gravity[1]; b.fgz = cfg->mass * cfg->gravity[2];
    double inv6[6];
    mat3_inverse_sym(cfg->inertia, inv6);
    for (int i = 0; i < 6; ++i) { b.ib[i] = cfg->inertia[i]; b.ibinv[i] = inv6[i]; }
    b.damping = cfg->joint_damping; b.effort = cfg->effort_limit; b.vel_limit = cfg->velocity_limit;
    b.unilateral = cfg->unilateral_cables ? 1 : 0;
    b.travel_on = (cfg->travel_lower != 0.0 || cfg->travel_upper != 0.0) ? 1 : 0;
    b.travel_lo = cfg->travel_lower; b.travel_hi = cfg->travel_upper;
    b.ph_lumped = (cfg->passive_damping != 0.0 || cfg->leg_inertia != 0.0 || cfg->cable_axial_mass != 0.0 || cfg->anchor_point_mass != 0.0 || cfg->anchor_inertia != 0.0) ? 1 : 0;
    b.ph_c = cfg->passive_damping; b.ph_jleg = cfg->leg_inertia; b.ph_max = cfg->cable_axial_mass; b.ph_mpt = cfg->anchor_point_mass;
    b.ph_iadd_total = cfg->anchor_inertia * (double)cfg->n_cables; b.ph_mass = cfg->mass;
    b.gx = cfg->gravity[0]; b.gy = cfg->gravity[1]; b.gz = cfg->gravity[2];
    b.fk_lambda = cfg->fk_lambda; b.fk_tol = cfg->fk_tolerance; b.fk_iters = (int)cfg->fk_max_iterations;
    b.td_min = cfg->td_f_min; b.td_max = cfg->td_f_max; b.td_mid = 0.5 * (cfg->td_f_min + cfg->td_f_max);
  } else {
    if ((e = hipMalloc(&h->d_state, slot_bytes * h->n_state)) != hipSuccess) return fail("hipMalloc(state)", e);
    if ((e = hipMalloc(&h->d_obs, slot_bytes * h->n_obs)) != hipSuccess) return fail("hipMalloc(obs)", e);
  }
  const size_t cmd_bytes = (size_t)h->stride * h->n * sizeof(float);
  for (int i = 0; i < 2; ++i) {
    if ((e = hipMalloc(&h->d_vel[i], cmd_bytes)) != hipSuccess) return fail("hipMalloc(cmd)", e);
    if ((e = hipMalloc(&h->d_pos[i], cmd_bytes)) != hipSuccess) return fail("hipMalloc(cmd)", e);
    if ((e = hipMalloc(&h->d_frc[i], cmd_bytes)) != hipSuccess) return fail("hipMalloc(cmd)", e);
    (void)hipMemset(h->d_vel[i], 0, cmd_bytes);
    (void)hipMemset(h->d_pos[i], 0, cmd_bytes);
    (void)hipMemset(h->d_frc[i], 0, cmd_bytes);
  }
  if ((e = hipMalloc(&h->d_wtab, sizeof wtab_host)) != hipSuccess) return fail("hipMalloc(wtab)", e);
  if ((e = hipMemcpy(h->d_wtab, wtab_host, sizeof wtab_host, hipMemcpyHostToDevice)) != hipSuccess) return fail("hipMemcpy(wtab)", e);
  h->pid_vel.wtab = h->d_wtab;
  h->pid_pos.wtab = h->d_wtab + kWin * (kWin + 2);
  {
    std::vector<float> g = geom_pairs(*cfg);
    if ((e = hipMalloc(&h->d_geom, g.size() * sizeof(float))) != hipSuccess) return fail("hipMalloc(geom)", e);
    if ((e = hipMemcpy(h->d_geom, g.data(), g.size() * sizeof(float), hipMemcpyHostToDevice)) != hipSuccess) return fail("hipMemcpy(geom)", e);
  }
  if (h->general) {
    h->glay.n = (int)h->n;
    h->glay.nb = (int)std::max(cfg->velocity_pid.d_buffer_length, cfg->position_pid.d_buffer_length);
    h->glay.ncas = (int)std::max(std::max(cfg->velocity_pid.p_filter.cascade, cfg->velocity_pid.d_filter.cascade),
                                 std::max(cfg->position_pid.p_filter.cascade, cfg->position_pid.d_filter.cascade));
    const size_t rec_bytes = h->glay.bytes(h->stride);
    if (rec_bytes >= (1ull << 32)) {  // the record buffer is addressed with 32-bit offsets (one buffer resource)
      g_create_error = "general controller path: the controller records of this batch pass 4 GiB; split the batch over several handles";
      free_all(h);
      return CDPR_ERR_UNSUPPORTED;
    }
    if ((e = hipMalloc(&h->d_rec, rec_bytes)) != hipSuccess) return fail("hipMalloc(rec)", e);
    fill_gen_pid(cfg->position_pid, h->gpid[0]);
    fill_gen_pid(cfg->velocity_pid, h->gpid[1]);
    // FIR weights by ring head: when the newest sample sits in slot `head`, slot j holds the sample of age (head - j) mod
    // nbuf, whose end-point LS weight (oldest first) is w[nbuf - 1 - age]; slots >= nbuf weigh nothing.  The head slot itself
    // weighs nothing in the table: the newest sample is still in a register when the FIR runs, its weight rides in the Pid
    // table (last float)
    const int nbmax = h->glay.nb > 11 ? kGenMaxBuf : 11, nbp = gen_nbp(nbmax);
    std::vector<float> wt((size_t)2 * nbmax * nbp, 0.f);
    const cdpr_pid_params_t* pp[2] = {&cfg->position_pid, &cfg->velocity_pid};
    float w_new[2] = {0.f, 0.f};
    for (int p = 0; p < 2; ++p) {
      double w[CDPR_MAX_D_BUFFER];
      const int nb = (int)pp[p]->d_buffer_length;
      if (derivative_weights((uint32_t)nb, pp[p]->d_degree, w) != CDPR_OK) continue;
      for (int head = 0; head < nb; ++head)
        for (int j = 0; j < nb; ++j) wt[((size_t)p * nbmax + head) * nbp + j] = (j == head) ? 0.f : (float)w[nb - 1 - (((head - j) % nb + nb) % nb)];
      w_new[p] = (float)w[nb - 1];
    }
    float pt[2 * kGenPidFloats];
    gen_pid_table(h->gpid[0], pt);
    gen_pid_table(h->gpid[1], pt + kGenPidFloats);
    pt[kGenPidFloats - 1] = w_new[0], pt[2 * kGenPidFloats - 1] = w_new[1];
    if ((e = hipMalloc(&h->d_gptab, sizeof pt)) != hipSuccess) return fail("hipMalloc(gptab)", e);
    if ((e = hipMemcpy(h->d_gptab, pt, sizeof pt, hipMemcpyHostToDevice)) != hipSuccess) return fail("hipMemcpy(gptab)", e);
    if ((e = hipMalloc(&h->d_gwtab, wt.size() * sizeof(float))) != hipSuccess) return fail("hipMalloc(gwtab)", e);
    if ((e = hipMemcpy(h->d_gwtab, wt.data(), wt.size() * sizeof(float), hipMemcpyHostToDevice)) != hipSuccess) return fail("hipMemcpy(gwtab)", e);
  }
  if (h->per_robot) {
    if (!h->general && (e = hipMalloc(&h->d_target, cmd_bytes)) != hipSuccess) return fail("hipMalloc(target)", e);
    if ((e = hipMalloc(&h->d_mode, h->batch)) != hipSuccess) return fail("hipMalloc(mode)", e);
    for (int i = 0; i < 3; ++i)
      if ((e = hipMalloc(&h->d_mask[i], h->batch)) != hipSuccess) return fail("hipMalloc(mask)", e);
  }
  if (h->dbg && !h->fp64)
    if ((e = hipMalloc(&h->d_dbg, (size_t)h->batch * CDPR_PID_DEBUG_AXES * sizeof(float))) != hipSuccess)
      return fail("hipMalloc(dbg)", e);
  if (upload_home(h) != CDPR_OK || warm_first_launch(h) != CDPR_OK) {
    g_create_error = h->err;
    free_all(h);
    return CDPR_ERR_DEVICE;
  }
  *out = h;
  return CDPR_OK;
}

void cdpr_destroy(cdpr_handle_t h) {
  if (!h) return;
  (void)hipSetDevice(h->device);
  if (h->stream) (void)hipStreamSynchronize(h->stream);
  free_all(h);
}

int cdpr_reset(cdpr_handle_t h) {
  if (!h) return CDPR_ERR_INVALID;
  if (set_device(h) != CDPR_OK) return CDPR_ERR_DEVICE;
  if (int rc = drain_copy_stream(h)) return rc;
  HIP_TRY(h, wait_stream(h));
  h->free_ev_set[0] = h->free_ev_set[1] = h->free_ev_set[2] = false;
  engine_reset_host(h);
  return upload_home(h);
}

int cdpr_set_platform_state(cdpr_handle_t h, const float* pose7, const float* twist6) {
  if (!h) return CDPR_ERR_INVALID;
  if (set_device(h) != CDPR_OK) return CDPR_ERR_DEVICE;
  if (h->fp64) {
    std::vector<double> p, t;
    if (pose7) p.assign(pose7, pose7 + (size_t)h->batch * 7);
    if (twist6) t.assign(twist6, twist6 + (size_t)h->batch * 6);
    return set_platform_state64(h, pose7 ? p.data() : nullptr, twist6 ? t.data() : nullptr);
  }
  const int P = plat_slots(h->fk);
  std::vector<float4> s;
  int rc = fetch_slots(h, h->d_state, P, s);
  if (rc != CDPR_OK) return rc;
  const size_t st = h->stride;
  for (uint32_t r = 0; r < h->batch; ++r) {
    float4 &a = s[0 * st + r], &b = s[1 * st + r], &c = s[2 * st + r], &d = s[3 * st + r];
    if (pose7) {
      const float* p = pose7 + (size_t)r * 7;
      a = make_float4(p[0], p[1], p[2], p[3]);
      b.x = p[4]; b.y = p[5]; b.z = p[6];
      d.y = p[0]; d.z = p[1]; d.w = p[2];  // the FK seed follows the spawn pose
      if (h->fk) s[4 * st + r] = make_float4(p[3], p[4], p[5], p[6]);
    }
    if (twist6) {
      const float* t = twist6 + (size_t)r * 6;
      b.w = t[0]; c.x = t[1]; c.y = t[2]; c.z = t[3]; c.w = t[4]; d.x = t[5];
    }
  }
  HIP_TRY(h, hipMemcpyAsync(h->d_state, s.data(), s.size() * sizeof(float4), hipMemcpyHostToDevice, h->stream));
  HIP_TRY(h, wait_stream(h));
  return CDPR_OK;
}

// A pending command replaces the one before it (PLG.cpp:69,78: the callback overwrites the stored message); on a
// per-robot handle an UNMASKED command after a masked one would have to merge with it, which the callbacks of
// independent plugins never need: the later call wins for the robots it addresses, the earlier one keeps the rest.
static int stage_masked(cdpr_engine* h, int which, const float* axes, size_t count, const uint8_t* robot_mask) {
  if (!h->per_robot) {
    h->err = "masked commands need a handle created with per_robot_commands = 1";
    return CDPR_ERR_UNSUPPORTED;
  }
  if (!robot_mask) {
    h->err = "null robot mask";
    return CDPR_ERR_INVALID;
  }
  const size_t n = h->n, B = h->batch;
  if (count != n * B && count != n) return CDPR_IGNORED;  // PLG.cpp:68-73,77-82
  if (set_device(h) != CDPR_OK) return CDPR_ERR_DEVICE;
  if (int rc = drain_copy_stream(h)) return rc;
  float* pending = which == 0 ? h->d_vel[1] : which == 1 ? h->d_pos[1] : h->d_frc[1];
  bool& is_pending = which == 0 ? h->vel_pending : which == 1 ? h->pos_pending : h->frc_pending;
  bool& masked = which == 0 ? h->vel_masked : which == 1 ? h->pos_masked : h->frc_masked;
  // merge with a command of the same kind that is already pending: rows and mask bits of the robots addressed now
  std::vector<float> rows(n * B);
  std::vector<uint8_t> mask(B, 0);
  if (is_pending) {
    HIP_TRY(h, hipMemcpyAsync(rows.data(), pending, n * B * sizeof(float), hipMemcpyDeviceToHost, h->stream));
    if (masked)
      HIP_TRY(h, hipMemcpyAsync(mask.data(), h->d_mask[which], B, hipMemcpyDeviceToHost, h->stream));
    else
      std::fill(mask.begin(), mask.end(), (uint8_t)1);
    HIP_TRY(h, wait_stream(h));
  }
  for (size_t b = 0; b < B; ++b) {
    if (!robot_mask[b]) continue;
    mask[b] = 1;
    memcpy(&rows[b * n], count == n ? axes : axes + b * n, n * sizeof(float));
  }
  HIP_TRY(h, hipMemcpyAsync(pending, rows.data(), n * B * sizeof(float), hipMemcpyHostToDevice, h->stream));
  HIP_TRY(h, hipMemcpyAsync(h->d_mask[which], mask.data(), B, hipMemcpyHostToDevice, h->stream));
  HIP_TRY(h, wait_stream(h));
  is_pending = true;
  masked = true;
  return CDPR_OK;
}

int cdpr_set_velocity_command(cdpr_handle_t h, const float* axes, size_t count) {
  if (!h) return CDPR_ERR_INVALID;
  int rc = stage_command(h, h->d_vel[1], axes, count, false);
  if (rc == CDPR_OK) {
    h->vel_pending = true;
    h->vel_masked = false;
    h->ext_vel[1] = nullptr;
  }
  return rc;
}

int cdpr_set_position_command(cdpr_handle_t h, const float* axes, size_t count) {
  if (!h) return CDPR_ERR_INVALID;
  int rc = stage_command(h, h->d_pos[1], axes, count, false);
  if (rc == CDPR_OK) {
    h->pos_pending = true;
    h->pos_masked = false;
    h->ext_pos[1] = nullptr;
  }
  return rc;
}

int cdpr_set_force_command(cdpr_handle_t h, const float* axes, size_t count) {
  if (!h) return CDPR_ERR_INVALID;
  int rc = stage_command(h, h->d_frc[1], axes, count, false);
  if (rc == CDPR_OK) {
    h->frc_pending = true;
    h->frc_masked = false;
    h->ext_frc[1] = nullptr;
  }
  return rc;
}

int cdpr_set_force_command_device(cdpr_handle_t h, const float* d_axes, size_t count) {
  if (!h) return CDPR_ERR_INVALID;
  int rc = stage_command(h, h->d_frc[1], d_axes, count, true);
  if (rc == CDPR_OK) {
    h->frc_pending = true;
    h->frc_masked = false;
    h->ext_frc[1] = nullptr;
  }
  return rc;
}

int cdpr_set_force_command_masked(cdpr_handle_t h, const float* axes, size_t count, const uint8_t* robot_mask) {
  if (!h) return CDPR_ERR_INVALID;
  if (!axes) {
    h->err = "null command buffer";
    return CDPR_ERR_INVALID;
  }
  return stage_masked(h, 2, axes, count, robot_mask);
}

int cdpr_set_velocity_command_masked(cdpr_handle_t h, const float* axes, size_t count, const uint8_t* robot_mask) {
  if (!h) return CDPR_ERR_INVALID;
  if (!axes) {
    h->err = "null command buffer";
    return CDPR_ERR_INVALID;
  }
  return stage_masked(h, 0, axes, count, robot_mask);
}

int cdpr_set_position_command_masked(cdpr_handle_t h, const float* axes, size_t count, const uint8_t* robot_mask) {
  if (!h) return CDPR_ERR_INVALID;
  if (!axes) {
    h->err = "null command buffer";
    return CDPR_ERR_INVALID;
  }
  return stage_masked(h, 1, axes, count, robot_mask);
}

int cdpr_set_velocity_command_device(cdpr_handle_t h, const float* d_axes, size_t count) {
  if (!h) return CDPR_ERR_INVALID;
  int rc = stage_command(h, h->d_vel[1], d_axes, count, true);
  if (rc == CDPR_OK) {
    h->vel_pending = true;
    h->vel_masked = false;
    h->ext_vel[1] = nullptr;
  }
  return rc;
}

int cdpr_set_position_command_device(cdpr_handle_t h, const float* d_axes, size_t count) {
  if (!h) return CDPR_ERR_INVALID;
  int rc = stage_command(h, h->d_pos[1], d_axes, count, true);
  if (rc == CDPR_OK) {
    h->pos_pending = true;
    h->pos_masked = false;
    h->ext_pos[1] = nullptr;
  }
  return rc;
}

// Zero-copy form: the caller's device buffer float[B][n] IS the latched Joy batch from the next update on (no copy, no
// synchronisation); it must stay valid and unchanged until another command of the same kind has been latched.
static int bind_command(cdpr_engine* h, int which, const float* d_axes, size_t count) {
  if (!d_axes) {
    h->err = "null command buffer";
    return CDPR_ERR_INVALID;
  }
  if (count != (size_t)h->n * h->batch) return CDPR_IGNORED;  // one Joy per robot; no broadcast without a copy
  if (h->per_robot) {
    h->err = "cdpr_bind_*_command_device: not available on a per_robot_commands handle (commands are latched robot by robot)";
    return CDPR_ERR_UNSUPPORTED;
  }
  if (which == 0) {
    h->ext_vel[1] = d_axes;
    h->vel_pending = true;
  } else if (which == 1) {
    h->ext_pos[1] = d_axes;
    h->pos_pending = true;
  } else {
    h->ext_frc[1] = d_axes;
    h->frc_pending = true;
  }
  return CDPR_OK;
}

int cdpr_bind_force_command_device(cdpr_handle_t h, const float* d_axes, size_t count) {
  return h ? bind_command(h, 2, d_axes, count) : CDPR_ERR_INVALID;
}

int cdpr_bind_velocity_command_device(cdpr_handle_t h, const float* d_axes, size_t count) {
  return h ? bind_command(h, 0, d_axes, count) : CDPR_ERR_INVALID;
}

int cdpr_bind_position_command_device(cdpr_handle_t h, const float* d_axes, size_t count) {
  return h ? bind_command(h, 1, d_axes, count) : CDPR_ERR_INVALID;
}

int cdpr_update(cdpr_handle_t h, int nsteps) { return run_steps(h, nsteps, 1); }

int cdpr_update_fused(cdpr_handle_t h, int nsteps, int steps_per_launch) { return run_steps(h, nsteps, steps_per_launch); }

// One observable image: fp32 handles n_obs float4 slot rows, precision = 64 handles f64_obs_rows(n) rows of doubles
static size_t image_bytes(const cdpr_engine* h) {
  return h->fp64 ? (size_t)f64_obs_rows((int)h->n) * h->stride * sizeof(double) : (size_t)h->n_obs * h->stride * sizeof(float4);
}

int cdpr_observable_image_bytes(cdpr_handle_t h, size_t* bytes) {
  if (!h || !bytes) return CDPR_ERR_INVALID;
  *bytes = image_bytes(h);
  return CDPR_OK;
}

int cdpr_update_record(cdpr_handle_t h, int nsteps, int steps_per_launch, void* d_record, size_t record_bytes) {
  if (!h) return CDPR_ERR_INVALID;
  if (h->cfg.publish_period != 0.0) {
    h->err = "cdpr_update_record needs publish_period == 0 (every step published)";
    return CDPR_ERR_UNSUPPORTED;
  }
  const size_t image = image_bytes(h);
  if (!d_record || nsteps < 0 || record_bytes < image * (size_t)nsteps) {
    h->err = "cdpr_update_record: record buffer missing or smaller than nsteps observable images";
    return CDPR_ERR_INVALID;
  }
  return run_steps(h, nsteps, steps_per_launch, static_cast<float4*>(d_record));
}

// A command schedule resident in HBM, any kind of command, any handle.  Semantics: for j = 0 ..: the callback of `kind`
// with batch j (for the robots of mask j), then refresh_steps x update().  Two forms serve it:
//   * in the launch: uniform-mode handles on the register-resident path with one or two lanes per robot run the whole
//     schedule in ONE launch of the several-steps kernel (the lanes read batch j at step j * refresh_steps themselves;
//     state, windows and integrals stay on chip);
//   * as a chain: every other handle (general controller path, per-robot modes with or without masks, precision = 64,
//     one lane per cable) latches batch j from the caller's buffers in place and queues its steps, batch after batch,
//     without a host round trip; a mailbox is honoured by a one-thread wait kernel in front of the batch's latch.
// A command of another kind pending at the call is latched with batch 0 in update()'s usual order (velocity, position,
// force): batch 0 then goes the chain's way, so that a mode change between batch 0 and batch 1 is what the call sequence
// would do (the schedule's rows must never be read as targets of another mode).
static int scheduled_update(cdpr_engine* h, uint32_t kind, int nsteps, int refresh_steps, const float* d_commands, const uint32_t* d_ready,
                            const uint8_t* d_masks, void* d_record, size_t record_bytes) {
  if (kind > 2u) {
    h->err = "cdpr_update_scheduled_kind: kind must be CDPR_COMMAND_VELOCITY, _POSITION or _FORCE";
    return CDPR_ERR_INVALID;
  }
  if (nsteps < 0 || refresh_steps < 1 || !d_commands) {
    h->err = "cdpr_update_scheduled: nsteps >= 0, refresh_steps >= 1 and the command schedule are required";
    return CDPR_ERR_INVALID;
  }
  if (d_masks && !h->per_robot) {
    h->err = "cdpr_update_scheduled_kind: robot masks need a handle created with per_robot_commands = 1";
    return CDPR_ERR_UNSUPPORTED;
  }
  if (h->cfg.publish_period != 0.0 && d_record) {
    h->err = "cdpr_update_scheduled: a trajectory record needs publish_period == 0 (every step published)";
    return CDPR_ERR_UNSUPPORTED;
  }
  const size_t image = image_bytes(h) / sizeof(float4);  // in float4 units (both image kinds are multiples of 16 B: stride is a multiple of 64)
  if (d_record && record_bytes < image_bytes(h) * (size_t)nsteps) {
    h->err = "cdpr_update_scheduled: record buffer smaller than nsteps observable images";
    return CDPR_ERR_INVALID;
  }
  if (nsteps == 0) return CDPR_OK;
  if (set_device(h) != CDPR_OK) return CDPR_ERR_DEVICE;
  if (d_ready && !h->h_fault) {  // the status word a mailbox that never delivers is reported through
    HIP_TRY(h, hipHostMalloc((void**)&h->h_fault, sizeof(uint32_t), hipHostMallocMapped | hipHostMallocCoherent));
    *h->h_fault = 0u;
    HIP_TRY(h, hipHostGetDevicePointer((void**)&h->d_fault, h->h_fault, 0));
  }
  float4* const record = static_cast<float4*>(d_record);
  const size_t batch_floats = (size_t)h->batch * h->n;
  bool* pending[3] = {&h->vel_pending, &h->pos_pending, &h->frc_pending};
  bool* masked[3] = {&h->vel_masked, &h->pos_masked, &h->frc_masked};
  const float** ext[3] = {h->ext_vel, h->ext_pos, h->ext_frc};
  const int nbatches = (nsteps + refresh_steps - 1) / refresh_steps;
  // batch j becomes the pending command of `kind`, in place
  auto stage_batch = [&](int j) {
    const float* rows = d_commands + (size_t)j * batch_floats;
    if (h->per_robot) {
      h->sched_rows[kind] = rows;
      h->sched_mask[kind] = d_masks ? d_masks + (size_t)j * h->batch : nullptr;
    } else {
      ext[kind][1] = rows;
    }
    *pending[kind] = true;
    *masked[kind] = false;
  };
  // Per-robot handles: a command of the SAME kind still pending at the call (cdpr_set_*_command[_masked] without an update since)
  // would have to reach its robots before batch 0 reaches batch 0's - B independent plugins each keep the last message that
  // reached THEM.  Staging batch 0 would overwrite it silently: refused, the caller latches it with cdpr_update first
  // or leaves it out.  (Uniform handles: the later message of a kind replaces the earlier one, as in the plugin, PLG.cpp:67-83.)
  if (h->per_robot && *pending[kind]) {
    h->err = "cdpr_update_scheduled_kind: a command of the same kind is still pending on this per-robot handle; step it in with cdpr_update first";
    return CDPR_ERR_INVALID;
  }
  // whatever way this function is left, nothing of the caller's schedule stays staged in the handle: an error return must not
  // leave `pending` set on pointers the caller may free, nor the schedule fields set for the next plain cdpr_update
  struct Unstage {
    cdpr_engine* h; uint32_t kind; bool* pend; const float** ext; bool ok = false;
    ~Unstage() {
      h->sched_rows[kind] = nullptr;
      h->sched_mask[kind] = nullptr;
      h->sched_refresh = 0;
      h->sched_ready = nullptr;
      if (!ok) {  // an error in between: drop the batch that was staged but not latched
        if (*pend && (h->per_robot || ext[1])) *pend = false;
        ext[1] = nullptr;
      }
    }
  } unstage{h, kind, pending[kind], ext[kind]};
  const bool in_launch = !(h->general || h->fp64 || h->per_robot || h->lane_cable) && h->cfg.publish_period == 0.0;
  bool others = false;
  for (uint32_t k = 0; k < 3; ++k) others = others || (k != kind && *pending[k]);
  int j = 0, done = 0;
  // ---- the chain: every batch of a handle the launch form does not serve; batch 0 where another command is pending
  const int chain_spl = ((size_t)h->batch * h->n <= 131072u) ? std::min(refresh_steps, 64) : 1;  // small batches: the hold in one launch
  while (j < nbatches && (!in_launch || (j == 0 && others))) {
    if (d_ready) {
      hipLaunchKernelGGL(cdpr_mailbox_wait_kernel, dim3(1), dim3(1), 0, h->stream, d_ready + j, h->d_fault);
      HIP_TRY(h, hipGetLastError());
    }
    stage_batch(j);
    const int k = std::min(refresh_steps, nsteps - done);
    if (int rc = run_steps(h, k, chain_spl, record ? record + (size_t)done * image : nullptr)) return rc;
    done += k;
    ++j;
  }
  while (j < nbatches) {
    // ---- the rest in one launch: batch j is latched as any command of its kind is (entering the mode resets its Pid,
    //      JFC.cpp:101-103,113-115), the later ones are read by the kernel.  Lane-pair handles whose derivative windows are not
    //      full yet (world step 0, a mode just entered) take the batches that fill them in a launch of their own: from there on
    //      the launch qualifies for the steady-state kernel (pair_stream_ok), which has no branch for a filling window.
    stage_batch(j);
    h->sched_refresh = refresh_steps;
    h->sched_ready = d_ready ? d_ready + j : nullptr;
    int rest = nsteps - done;
    if (h->lane_pair && h->pair_stream && !h->fk && !h->td && kind != CDPR_COMMAND_FORCE) {
      const int new_mode = (kind == CDPR_COMMAND_VELOCITY) ? kModeVelocity : kModePosition;
      const int nbuf = (kind == CDPR_COMMAND_VELOCITY) ? h->pid_vel.nbuf : h->pid_pos.nbuf;
      const int calls = (h->mode == new_mode) ? h->pid_calls : 0;  // (entering the mode resets the Pid)
      const int fill = (calls >= nbuf ? 0 : nbuf - calls) + (h->step == 0 ? 1 : 0);
      const int pre = ((fill + refresh_steps - 1) / refresh_steps) * refresh_steps;  // whole batches
      if (pre > 0 && pre < rest) rest = pre;
    }
    const int rc = run_steps(h, rest, rest, record ? record + (size_t)done * image : nullptr);
    h->sched_refresh = 0;
    h->sched_ready = nullptr;
    if (rc != CDPR_OK) return rc;
    done += rest;
    j += (rest + refresh_steps - 1) / refresh_steps;
    if (j >= nbatches) ext[kind][0] = d_commands + (size_t)(nbatches - 1) * batch_floats;  // the batch that stays latched
  }
  unstage.ok = true;
  return CDPR_OK;
}

int cdpr_update_scheduled(cdpr_handle_t h, int nsteps, int refresh_steps, const float* d_commands, const uint32_t* d_ready, void* d_record,
                          size_t record_bytes) {
  if (!h) return CDPR_ERR_INVALID;
  return scheduled_update(h, CDPR_COMMAND_VELOCITY, nsteps, refresh_steps, d_commands, d_ready, nullptr, d_record, record_bytes);
}

int cdpr_update_scheduled_kind(cdpr_handle_t h, uint32_t kind, int nsteps, int refresh_steps, const float* d_commands, const uint32_t* d_ready,
                               const uint8_t* d_robot_masks, void* d_record, size_t record_bytes) {
  if (!h) return CDPR_ERR_INVALID;
  return scheduled_update(h, kind, nsteps, refresh_steps, d_commands, d_ready, d_robot_masks, d_record, record_bytes);
}

int cdpr_decode_observables(cdpr_handle_t h, const void* image, float* position, float* velocity, float* effort,
                            float* pose7, float* twist6) {
  if (!h || !image) return CDPR_ERR_INVALID;
  if (h->fp64) {  // (the float getters of a precision = 64 handle round)
    decode_image64_to_float(h, static_cast<const double*>(image), position, velocity, effort, pose7, twist6);
    return CDPR_OK;
  }
  const float4* o = static_cast<const float4*>(image);
  const size_t st = h->stride;
  const int G = joint_groups((int)h->n);
  float* dst[3] = {position, velocity, effort};
  for (int f = 0; f < 3; ++f) {
    if (!dst[f]) continue;
    for (uint32_t r = 0; r < h->batch; ++r)
      for (uint32_t i = 0; i < h->n; ++i) dst[f][(size_t)r * h->n + i] = comp(o[(size_t)(4 + f * G + i / 4) * st + r], i % 4);
  }
  for (uint32_t r = 0; r < h->batch; ++r) {
    const float4 a = o[0 * st + r], b = o[1 * st + r], c = o[2 * st + r], d = o[3 * st + r];
    if (pose7) {
      float* p = pose7 + (size_t)r * 7;
      p[0] = a.x; p[1] = a.y; p[2] = a.z; p[3] = a.w; p[4] = b.x; p[5] = b.y; p[6] = b.z;
    }
    if (twist6) {
      float* t = twist6 + (size_t)r * 6;
      t[0] = b.w; t[1] = c.x; t[2] = c.y; t[3] = c.z; t[4] = c.w; t[5] = d.x;
    }
  }
  return CDPR_OK;
}

int cdpr_synchronize(cdpr_handle_t h) {
  if (!h) return CDPR_ERR_INVALID;
  if (set_device(h) != CDPR_OK) return CDPR_ERR_DEVICE;
  HIP_TRY(h, wait_stream(h));
  return check_fault(h);
}

static int copy_name(const std::string& text, char* name, size_t len) {
  if (!name || len == 0) return CDPR_ERR_INVALID;
  const size_t k = std::min(text.size(), len - 1);
  memcpy(name, text.data(), k);
  name[k] = 0;
  return CDPR_OK;
}

int cdpr_plan_kernel(const cdpr_config_t* cfg, int steps_per_launch, uint32_t flags, char* name, size_t len) {
  if (!cfg || !name || len == 0 || steps_per_launch < 1) return CDPR_ERR_INVALID;
  const KernelPlan plan = plan_kernels(*cfg);  // (no HIP call: works on a box without a GPU)
  if (plan.rc != CDPR_OK) {
    copy_name(plan.error, name, len);
    return plan.rc;
  }
  LaunchShape s;
  s.steps = steps_per_launch;
  s.first_world = (flags & CDPR_PLAN_FIRST_WORLD_STEP) != 0u;
  s.scheduled = (flags & CDPR_PLAN_SCHEDULED) != 0u;
  s.rollout = (flags & CDPR_PLAN_ROLLOUT) != 0u;
  s.steady = (flags & CDPR_PLAN_NOT_STEADY) == 0u;
  return copy_name(planned_kernel_name(plan, planned_kernel(plan, s)), name, len);
}

int cdpr_kernel_name(cdpr_handle_t h, char* name, size_t len) {
  if (!h) return CDPR_ERR_INVALID;
  return copy_name(planned_kernel_name(h->plan, h->last_kernel), name, len);
}

uint32_t cdpr_mapping(cdpr_handle_t h) {
  return !h ? CDPR_MAP_AUTO : (h->lane_cable ? CDPR_MAP_LANE_PER_CABLE : h->lane_pair ? CDPR_MAP_LANE_PAIR : CDPR_MAP_LANE_PER_ROBOT);
}

uint64_t cdpr_step_count(cdpr_handle_t h) { return h ? h->step : 0; }

int cdpr_get_joint_states(cdpr_handle_t h, float* position, float* velocity, float* effort) {
  if (!h) return CDPR_ERR_INVALID;
  if (set_device(h) != CDPR_OK) return CDPR_ERR_DEVICE;
  if (h->fp64) return checked(h, fetch_observables64(h, position, velocity, effort, nullptr, nullptr, true));
  const int G = joint_groups((int)h->n);
  float* dst[3] = {position, velocity, effort};
  for (int f = 0; f < 3; ++f) {
    std::vector<std::pair<int, int>> fields;
    for (uint32_t i = 0; i < h->n; ++i) fields.push_back({4 + f * G + (int)(i / 4), (int)(i % 4)});
    int rc = fetch_fields(h, h->d_obs, fields, dst[f]);
    if (rc != CDPR_OK) return rc;
  }
  return check_fault(h);
}

// JointState + PlatformState of the last published step in ONE device round trip (PLG.cpp:248-280 publishes both every
// step): the gather kernel writes the five arrays into a pinned host image and then a completion word the host spins
// on.  (cdpr_get_joint_states + cdpr_get_platform_state are five gathers, five copies and five waits.)
int cdpr_get_observables(cdpr_handle_t h, float* position, float* velocity, float* effort, float* pose7, float* twist6) {
  if (!h) return CDPR_ERR_INVALID;
  if (set_device(h) != CDPR_OK) return CDPR_ERR_DEVICE;
  if (h->fp64) return checked(h, fetch_observables64(h, position, velocity, effort, pose7, twist6, true));
  const uint32_t n = h->n, width = 3u * n + 13u;
  const size_t count = (size_t)h->batch * width;
  // lazy set-up, every allocation guarded on its own pointer (a failure half way leaves nothing to leak or to skip next
  // time).  Coherent host memory: the host spins on the completion word while the kernel is still running.
  if (!h->h_pub && count * sizeof(float) <= (2u << 20))  // the pinned image serves the two small tiers only
    HIP_TRY(h, hipHostMalloc((void**)&h->h_pub, count * sizeof(float), hipHostMallocMapped | hipHostMallocCoherent));
  if (!h->h_pub_done) {
    HIP_TRY(h, hipHostMalloc((void**)&h->h_pub_done, sizeof(uint64_t), hipHostMallocMapped | hipHostMallocCoherent));
    *h->h_pub_done = 0;
  }
  if (!h->d_pub_arrivals) {
    HIP_TRY(h, hipMalloc(&h->d_pub_arrivals, sizeof(uint32_t)));
    hipError_t me = hipMemsetAsync(h->d_pub_arrivals, 0, sizeof(uint32_t), h->stream);
    if (me != hipSuccess) {  // never launch the publish kernel on an uninitialised arrival counter
      (void)hipFree(h->d_pub_arrivals);
      h->d_pub_arrivals = nullptr;
      HIP_TRY(h, me);
    }
  }
  // small images go straight to host memory from the gather kernel (a per-step caller of a few robots: ~5 us); large
  // ones through device scratch and the copy engine (kernel stores over PCIe reach ~7 GB/s, the copy engine ~30)
  const bool direct = count * sizeof(float) <= (256u << 10);
  if (!direct && h->unpack_cap < count) {
    HIP_TRY(h, wait_stream(h));
    if (h->d_unpack) (void)hipFree(h->d_unpack);
    h->d_unpack = nullptr;
    h->unpack_cap = 0;
    HIP_TRY(h, hipMalloc(&h->d_unpack, count * sizeof(float)));
    h->unpack_cap = count;
  }
  PublishArgs u{};
  u.rows = h->d_obs;
  if (direct) {
    HIP_TRY(h, hipHostGetDevicePointer((void**)&u.out, h->h_pub, 0));
    HIP_TRY(h, hipHostGetDevicePointer((void**)&u.done, h->h_pub_done, 0));
  } else {
    u.out = h->d_unpack;
    u.done = nullptr;
  }
  u.arrivals = h->d_pub_arrivals;
  u.epoch = ++h->pub_epoch;
  u.stride = h->stride;
  u.batch = h->batch;
  u.n = n;
  u.width = width;
  const int G = joint_groups((int)n);
  uint32_t j = 0;
  for (int f = 0; f < 3; ++f)
    for (uint32_t i = 0; i < n; ++i, ++j) {
      u.slot[j] = (uint8_t)(4 + f * G + (int)(i / 4));
      u.comp[j] = (uint8_t)(i % 4);
    }
  for (const auto& pc : kPoseFields) u.slot[j] = (uint8_t)pc.first, u.comp[j] = (uint8_t)pc.second, ++j;
  for (const auto& pc : kTwistFields) u.slot[j] = (uint8_t)pc.first, u.comp[j] = (uint8_t)pc.second, ++j;
  hipLaunchKernelGGL(cdpr_publish_kernel, dim3((uint32_t)((count + 255) / 256)), dim3(256), 0, h->stream, u);
  HIP_TRY(h, hipGetLastError());
  if (direct) {
    // wait on the completion word: plain host memory, no runtime call; past the spin budget fall back to the stream wait
    // (which also surfaces a device fault instead of spinning on a word that will never come)
    volatile uint64_t* done = h->h_pub_done;
    const auto t0 = std::chrono::steady_clock::now();
    uint32_t spins = 0;
    while (*done != u.epoch) {
      if ((++spins & 0x3FFu) == 0 && std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - t0).count() >= 20) {
        HIP_TRY(h, hipStreamSynchronize(h->stream));
        if (*done != u.epoch) {
          h->err = "cdpr_get_observables: the publish kernel finished without its completion word";
          return CDPR_ERR_DEVICE;
        }
        break;
      }
    }
  } else if (count * sizeof(float) <= (2u << 20)) {  // one copy into the pinned image, split up on the host below
    HIP_TRY(h, hipMemcpyAsync(h->h_pub, h->d_unpack, count * sizeof(float), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, wait_stream(h));
  } else {  // straight into the caller's arrays (reading a large pinned image back on the host costs more than the DMA)
    const size_t bnl = (size_t)h->batch * n;
    float* dst[5] = {position, velocity, effort, pose7, twist6};
    const size_t off[5] = {0, bnl, 2 * bnl, 3 * bnl, 3 * bnl + (size_t)h->batch * 7};
    const size_t len[5] = {bnl, bnl, bnl, (size_t)h->batch * 7, (size_t)h->batch * 6};
    for (int k = 0; k < 5; ++k)
      if (dst[k]) HIP_TRY(h, hipMemcpyAsync(dst[k], h->d_unpack + off[k], len[k] * sizeof(float), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, wait_stream(h));
    return CDPR_OK;
  }
  std::atomic_thread_fence(std::memory_order_acquire);
  const size_t bn = (size_t)h->batch * n;
  const float* src = h->h_pub;
  if (position) std::memcpy(position, src, bn * sizeof(float));
  if (velocity) std::memcpy(velocity, src + bn, bn * sizeof(float));
  if (effort) std::memcpy(effort, src + 2 * bn, bn * sizeof(float));
  if (pose7) std::memcpy(pose7, src + 3 * bn, (size_t)h->batch * 7 * sizeof(float));
  if (twist6) std::memcpy(twist6, src + 3 * bn + (size_t)h->batch * 7, (size_t)h->batch * 6 * sizeof(float));
  return check_fault(h);
}

int cdpr_get_platform_state(cdpr_handle_t h, float* pose7, float* twist6) {
  if (!h) return CDPR_ERR_INVALID;
  if (set_device(h) != CDPR_OK) return CDPR_ERR_DEVICE;
  if (h->fp64) return checked(h, fetch_observables64(h, nullptr, nullptr, nullptr, pose7, twist6, true));
  const int rc = fetch_platform(h, h->d_obs, pose7, twist6);
  return rc != CDPR_OK ? rc : check_fault(h);
}

int cdpr_get_raw_state(cdpr_handle_t h, float* pose7, float* twist6) {
  if (!h) return CDPR_ERR_INVALID;
  if (set_device(h) != CDPR_OK) return CDPR_ERR_DEVICE;
  if (h->fp64) {
    int rc = fetch_rows64(h, h->d_state64, 0, 7, pose7, true);
    return rc != CDPR_OK ? rc : checked(h, fetch_rows64(h, h->d_state64, 7, 6, twist6, true));
  }
  const int rc = fetch_platform(h, h->d_state, pose7, twist6);
  return rc != CDPR_OK ? rc : check_fault(h);
}

int cdpr_get_pid_debug(cdpr_handle_t h, float* axes9) {
  if (!h || !axes9) return CDPR_ERR_INVALID;
  if (!h->dbg) {
    h->err = "CDPR_STAGE_PID_DEBUG not enabled";
    return CDPR_ERR_UNSUPPORTED;
  }
  if (set_device(h) != CDPR_OK) return CDPR_ERR_DEVICE;
  if (h->fp64) {
    std::vector<double> d((size_t)h->batch * CDPR_PID_DEBUG_AXES);
    HIP_TRY(h, hipMemcpyAsync(d.data(), h->d_dbg64, d.size() * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, wait_stream(h));
    for (size_t i = 0; i < d.size(); ++i) axes9[i] = (float)d[i];
    return check_fault(h);
  }
  HIP_TRY(h, hipMemcpyAsync(axes9, h->d_dbg, (size_t)h->batch * CDPR_PID_DEBUG_AXES * sizeof(float), hipMemcpyDeviceToHost,
                            h->stream));
  HIP_TRY(h, wait_stream(h));
  return check_fault(h);
}

int cdpr_get_fk_state(cdpr_handle_t h, float* pose7, float* residual, int32_t* iterations) {
  if (!h) return CDPR_ERR_INVALID;
  if (!h->fk) {
    h->err = "CDPR_STAGE_FK not enabled";
    return CDPR_ERR_UNSUPPORTED;
  }
  if (set_device(h) != CDPR_OK) return CDPR_ERR_DEVICE;
  if (h->fp64) {
    int rc = fetch_rows64(h, h->d_state64, 13, 7, pose7, true);
    if (rc == CDPR_OK) rc = fetch_rows64(h, h->d_obs64, 13, 1, residual, true);
    return rc != CDPR_OK ? rc : checked(h, fetch_int_row64(h, 14, iterations));
  }
  // estimate: state slot 3 yzw + slot 4 xyzw; residual / iteration count: observable slot 3 y, z
  int rc = fetch_fields(h, h->d_state, {{3, 1}, {3, 2}, {3, 3}, {4, 0}, {4, 1}, {4, 2}, {4, 3}}, pose7);
  if (rc != CDPR_OK) return rc;
  rc = fetch_fields(h, h->d_obs, {{3, 1}}, residual);
  if (rc != CDPR_OK) return rc;
  return checked(h, fetch_fields(h, h->d_obs, {{3, 2}}, iterations, 1u));
}

int cdpr_get_td_state(cdpr_handle_t h, float* tension, int32_t* infeasible) {
  if (!h) return CDPR_ERR_INVALID;
  if (!h->td) {
    h->err = "CDPR_STAGE_TD not enabled";
    return CDPR_ERR_UNSUPPORTED;
  }
  int rc = cdpr_get_joint_states(h, nullptr, nullptr, tension);  // applied force == distributed tension
  if (rc != CDPR_OK) return rc;
  rc = h->fp64 ? fetch_int_row64(h, 15, infeasible) : fetch_fields(h, h->d_obs, {{3, 3}}, infeasible, 1u);
  if (rc == CDPR_OK && infeasible)
    for (uint32_t b = 0; b < h->batch; ++b) infeasible[b] &= 1;  // the travel-limit mask shares the component (pack_flags)
  return checked(h, rc);
}

int cdpr_get_limit_state(cdpr_handle_t h, uint32_t* cable_mask) {
  if (!h || !cable_mask) return CDPR_ERR_INVALID;
  if (set_device(h) != CDPR_OK) return CDPR_ERR_DEVICE;
  int rc = h->fp64 ? fetch_int_row64(h, 15, reinterpret_cast<int32_t*>(cable_mask)) : fetch_fields(h, h->d_obs, {{3, 3}}, cable_mask, 1u);
  if (rc == CDPR_OK)
    for (uint32_t b = 0; b < h->batch; ++b) cable_mask[b] >>= 1;  // bit 0 is the tension-distribution flag
  return checked(h, rc);
}

#ifdef CDPR_STAMPS
// diagnostic builds only: point the step kernel at a stamp buffer (uint64[blocks][8]); nullptr disables
int cdpr_debug_set_stamps(cdpr_handle_t h, unsigned long long* d_stamps) {
  if (!h) return CDPR_ERR_INVALID;
  h->base.stamps = d_stamps;
  return CDPR_OK;
}
#endif

int cdpr_device_malloc(cdpr_handle_t h, size_t bytes, void** out) {
  if (!h || !out) return CDPR_ERR_INVALID;
  if (set_device(h) != CDPR_OK) return CDPR_ERR_DEVICE;
  HIP_TRY(h, hipMalloc(out, bytes));
  return CDPR_OK;
}

int cdpr_device_free(cdpr_handle_t h, void* ptr) {
  if (!h) return CDPR_ERR_INVALID;
  if (set_device(h) != CDPR_OK) return CDPR_ERR_DEVICE;
  HIP_TRY(h, hipFree(ptr));
  return CDPR_OK;
}

int cdpr_device_upload(cdpr_handle_t h, void* dst, const void* src, size_t bytes) {
  if (!h || !dst || !src) return CDPR_ERR_INVALID;
  if (set_device(h) != CDPR_OK) return CDPR_ERR_DEVICE;
  HIP_TRY(h, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, h->stream));
  HIP_TRY(h, wait_stream(h));
  return CDPR_OK;
}

int cdpr_device_download(cdpr_handle_t h, void* dst, const void* src, size_t bytes) {
  if (!h || !dst || !src) return CDPR_ERR_INVALID;
  if (set_device(h) != CDPR_OK) return CDPR_ERR_DEVICE;
  HIP_TRY(h, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(h, wait_stream(h));
  return check_fault(h);  // (a trajectory record of a schedule whose mailbox timed out is not the scheduled one)
}

int cdpr_profile_begin(cdpr_handle_t h) {
  if (!h) return CDPR_ERR_INVALID;
  if (set_device(h) != CDPR_OK) return CDPR_ERR_DEVICE;
  h->launches_mark = h->launches;
  HIP_TRY(h, hipEventRecord(h->ev0, h->stream));
  return CDPR_OK;
}

int cdpr_profile_end(cdpr_handle_t h, float* elapsed_ms, uint64_t* kernel_launches) {
  if (!h) return CDPR_ERR_INVALID;
  if (set_device(h) != CDPR_OK) return CDPR_ERR_DEVICE;
  HIP_TRY(h, hipEventRecord(h->ev1, h->stream));
  // ONE wait for everything queued on the stream, the closing event included.  (hipEventSynchronize followed by the
  // caller's hipStreamSynchronize costs two wake-ups: measured 28 us on a 20-launch timed region, scripts/short_run_probe.py.)
  HIP_TRY(h, wait_stream(h));
  float ms = 0.f;
  HIP_TRY(h, hipEventElapsedTime(&ms, h->ev0, h->ev1));
  if (elapsed_ms) *elapsed_ms = ms;
  if (kernel_launches) *kernel_launches = h->launches - h->launches_mark;
  return CDPR_OK;
}

#undef UP
#undef DOWN

}  // extern "C"
