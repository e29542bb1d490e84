// cdpr_engine_rollout.hip - cdpr_rollout_velocity*: the MPC fan-out (BASELINE config 5) on register-resident, general-path and
// precision = 64 handles.
#include "cdpr_engine_internal.hpp"

namespace cdpr_host {

static int rollout_enqueue(cdpr_engine* h, int samples, int horizon, const float* d_commands, const float* d_ref, float* d_cost) {
  if (h->fp64) return rollout_enqueue_f64(h, samples, horizon, d_commands, d_ref, d_cost);
  if (h->general) {
    // every trajectory steps a private copy of its robot's controller records (both Pids of every cable: the hold branch
    // switches between them from step to step): one column per trajectory in a persistent, grow-only scratch
    const uint64_t traj = (uint64_t)h->batch * (uint64_t)samples;
    const size_t cols = (size_t)((traj + 63u) & ~(uint64_t)63u);
    const size_t bytes = h->glay.bytes(cols);
    if (bytes >= (1ull << 32)) {
      h->err = "rollout on the general controller path: the trajectories' controller records pass 4 GiB; use fewer samples per call";
      return CDPR_ERR_UNSUPPORTED;
    }
    if (h->roll_rec_cols < cols) {
      HIP_TRY(h, wait_stream(h));
      if (h->d_roll_rec) (void)hipFree(h->d_roll_rec);
      h->d_roll_rec = nullptr;
      h->roll_rec_cols = 0;
      HIP_TRY(h, hipMalloc(&h->d_roll_rec, bytes));
      h->roll_rec_cols = cols;
    }
    if (h->step + (uint64_t)horizon >= (1ull << 31)) {
      h->err = "general controller path: world-step counter would pass 2^31";
      return CDPR_ERR_UNSUPPORTED;
    }
    StepArgs a = h->base;
    a.state = h->d_state;
    a.obs = h->d_obs;
    a.geom = h->d_geom;
    a.batch = h->batch;
    a.stride = h->stride;
    a.nsteps = horizon;
    a.publish_mask = 0;
    copy_pid(h->pid_vel, a);  // unused
    a.flags = (h->step == 0) ? kFlagFirstWorldStep : 0u;
    a.roll_cmd = d_commands;
    a.roll_ref = d_ref;
    a.roll_cost = d_cost;
    a.roll_samples = (uint32_t)samples;
    GenCtl g = general_ctl(h);
    g.src_rec = h->d_rec;
    g.src_rstride = h->stride;
    g.rec = h->d_roll_rec;
    g.rstride = (uint32_t)h->roll_rec_cols;
    g.rec_bytes = (uint32_t)h->glay.bytes(h->roll_rec_cols);
    g.now_step = (int)h->step;
    GenKernel kern = pick_gen_kernel(h->n, h->fk, h->td, true, h->glay.nb > 11, false);
    hipLaunchKernelGGL(kern, dim3((uint32_t)((traj + 63u) / 64u)), dim3(64), 0, h->stream, a, g);
    HIP_TRY(h, hipGetLastError());
    ++h->launches;
    return CDPR_OK;
  }
  StepArgs a = h->base;
  a.state = h->d_state;
  a.obs = h->d_obs;
  a.cmd = nullptr;
  a.dbg = nullptr;
  a.geom = h->d_geom;
  a.batch = h->batch;
  a.stride = h->stride;
  a.nsteps = horizon;
  a.publish_mask = 0;
  copy_pid(h->pid_vel, a);
  a.flags = kFlagActualIsVelocity;
  if (h->step == 0) a.flags |= kFlagFirstWorldStep;
  // a Joy on jointVelocities while in Position mode resets the velocity Pid (JFC.cpp:113-115); the handle's own
  // records stay untouched, the rollout starts from zeroed copies (per-robot handles: decided per lane from meta)
  if (!h->per_robot && h->mode != kModeVelocity) a.flags |= kFlagRolloutResetPid;
  a.pid_calls = (h->mode == kModeVelocity) ? sat_pid_calls(h->pid_calls) : 0;
  a.ring_slot = ring_slot_of(h->step);
  if (h->per_robot) {
    copy_pid_alt(h->pid_pos, a.alt);
    a.meta = h->d_mode;
  }
  a.roll_cmd = d_commands;
  a.roll_ref = d_ref;
  a.roll_cost = d_cost;
  a.roll_samples = (uint32_t)samples;
  const uint64_t traj = (uint64_t)h->batch * (uint64_t)samples;
  LaunchShape rs = launch_shape(h, horizon);
  rs.rollout = true;
  StepKernel kern = step_kernel_of(h, planned_kernel(h->plan, rs));
  hipLaunchKernelGGL(kern, dim3((uint32_t)((traj + 63u) / 64u)), dim3(64), 0, h->stream, a);
  HIP_TRY(h, hipGetLastError());
  ++h->launches;
  return CDPR_OK;
}

static int rollout_check(cdpr_engine* h, int samples, int horizon, const void* d_commands) {
  if (samples < 1 || horizon < 1 || !d_commands) {
    h->err = "rollout: samples, horizon >= 1 and the command buffer are required";
    return CDPR_ERR_INVALID;
  }
  if ((uint64_t)h->batch * (uint64_t)samples > (1ull << 30)) {
    h->err = "rollout: too many trajectories";
    return CDPR_ERR_INVALID;
  }
  return set_device(h);
}
}  // namespace cdpr_host

int cdpr_rollout_velocity_device(cdpr_handle_t h, int samples, int horizon, const float* d_commands, const float* d_ref_position,
                                 float* d_cost) {
  if (!h) return CDPR_ERR_INVALID;
  int rc = rollout_check(h, samples, horizon, d_commands);
  if (rc != CDPR_OK) return rc;
  if (!d_ref_position || !d_cost) {
    h->err = "cdpr_rollout_velocity_device: d_ref_position and d_cost are required";
    return CDPR_ERR_INVALID;
  }
  return rollout_enqueue(h, samples, horizon, d_commands, d_ref_position, d_cost);
}

int cdpr_rollout_velocity_launch(cdpr_handle_t h, int samples, int horizon, const float* d_commands, const float* ref_position) {
  if (!h) return CDPR_ERR_INVALID;
  int rc = rollout_check(h, samples, horizon, d_commands);
  if (rc != CDPR_OK) return rc;
  if (!ref_position) {
    h->err = "cdpr_rollout_velocity_launch: ref_position is required";
    return CDPR_ERR_INVALID;
  }
  const uint64_t traj = (uint64_t)h->batch * (uint64_t)samples;
  if (!h->d_roll_ref) HIP_TRY(h, hipMalloc(&h->d_roll_ref, (size_t)h->batch * 3 * sizeof(float)));
  if (h->roll_cost_cap < traj) {  // grow-only; the stream may still be reading the old buffer
    HIP_TRY(h, wait_stream(h));
    if (h->d_roll_cost) (void)hipFree(h->d_roll_cost);
    h->d_roll_cost = nullptr;
    h->roll_cost_cap = 0;
    HIP_TRY(h, hipMalloc(&h->d_roll_cost, (size_t)traj * sizeof(float)));
    h->roll_cost_cap = traj;
  }
  // the caller may reuse ref_position on return: a pageable source is staged before hipMemcpyAsync returns
  HIP_TRY(h, hipMemcpyAsync(h->d_roll_ref, ref_position, (size_t)h->batch * 3 * sizeof(float), hipMemcpyHostToDevice, h->stream));
  rc = rollout_enqueue(h, samples, horizon, d_commands, h->d_roll_ref, h->d_roll_cost);
  if (rc == CDPR_OK) h->roll_pending = traj;
  return rc;
}

int cdpr_rollout_velocity_fetch(cdpr_handle_t h, float* cost) {
  if (!h) return CDPR_ERR_INVALID;
  if (!cost || h->roll_pending == 0) {
    h->err = "cdpr_rollout_velocity_fetch: no rollout pending (or null cost buffer)";
    return CDPR_ERR_INVALID;
  }
  if (set_device(h) != CDPR_OK) return CDPR_ERR_DEVICE;
  HIP_TRY(h, hipMemcpyAsync(cost, h->d_roll_cost, (size_t)h->roll_pending * sizeof(float), hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(h, wait_stream(h));
  h->roll_pending = 0;
  return CDPR_OK;
}

int cdpr_rollout_velocity(cdpr_handle_t h, int samples, int horizon, const float* d_commands, const float* ref_position,
                          float* cost) {
  if (!h) return CDPR_ERR_INVALID;
  if (!cost) {
    h->err = "cdpr_rollout_velocity: cost is required";
    return CDPR_ERR_INVALID;
  }
  int rc = cdpr_rollout_velocity_launch(h, samples, horizon, d_commands, ref_position);
  if (rc != CDPR_OK) return rc;
  return cdpr_rollout_velocity_fetch(h, cost);
}

