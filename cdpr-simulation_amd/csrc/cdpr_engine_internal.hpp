// cdpr_engine_internal.hpp - what the translation units of the host side share: the handle (struct cdpr_engine), the error macro
// and the helpers one unit defines and another calls.  Units: cdpr_engine.hip (create / destroy, commands, the fp32 launch chains,
// read-out), cdpr_engine_f64.hip (precision = 64), cdpr_engine_rollout.hip (cdpr_rollout_velocity*), cdpr_engine_solvers.hip
// (cdpr_solve_ik / fk / td).  Not installed, not part of the C-ABI (include/cdpr.h is).
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <string>
#include <vector>

#include "../../include/cdpr.h"
#include "cdpr_kernels.hpp"
#include "cdpr_select.hpp"
#include "cdpr_latch.hpp"
#include "cdpr_solvers.hpp"

using namespace cdpr;

namespace cdpr_host {

enum Mode { kModeForce = 0, kModePosition = 1, kModeVelocity = 2 };  // JFC.h:35-37

#define HIP_TRY(h, expr)                                                                         \
  do {                                                                                           \
    hipError_t e_ = (expr);                                                                      \
    if (e_ != hipSuccess) {                                                                      \
      (h)->err = std::string(#expr) + ": " + hipGetErrorString(e_);                              \
      return CDPR_ERR_DEVICE;                                                                    \
    }                                                                                            \
  } while (0)

}  // namespace cdpr_host
using namespace cdpr_host;

struct cdpr_engine {
  cdpr_config_t cfg{};
  int device = 0;
  hipStream_t stream = nullptr;
  uint32_t n = 0, batch = 0, stride = 0;
  bool fk = false, td = false, dbg = false;
  int n_state = 0, n_obs = 0;
  float4* d_state = nullptr;
  float4* d_obs = nullptr;
  float* d_dbg = nullptr;
  float* d_geom = nullptr;  // pair-interleaved cable geometry, staged in LDS by the kernel
  int pid_calls = 0;        // Pid::update calls since the last Pid reset (uniform over the batch)
  bool lane_pair = false;   // two lanes per robot (cdpr_step_kernel_pair.hpp) instead of one
  bool lane_cable = false;  // one lane per cable, 8 (or 4) lanes per robot (cdpr_step_kernel_cable.hpp)
  bool phys = false;        // lumped-leg physics terms enabled: the PHYS instantiations of the first-generation kernels
  bool lowreg = false;      // one-step launches use the <= 256-register build (two waves per SIMD; large batches)
  bool gen_split = false;   // general path: one-step launches use the role-split kernel (FK + TD, n >= 6, windows <= 11, <= 2 workgroups per CU)
  bool gen_hot = false;     // general path: robots in the deep steady state keep mLastTime / mIerr in hot rows instead of their H slots (GenHot; CDPR_GEN_HOT=0: off)
  bool gen_lean = false;    // general path, larger batches: one-step launches use the lean role-split kernel (two waves per SIMD, the rare
                            // controller paths by call: cdpr_general_split.hpp)
  bool persist = false;     // one-step launches use the persistent one-wave kernel: one wave per SIMD walks over blocks of 64
                            // robots, the next block's rows in flight under the current block's arithmetic (large batches)
  uint32_t persist_grid = 0;  // waves of such a launch: SIMDs of the device
  bool split = false;       // FK + TD one-step launches use cdpr_split_kernel (estimator wave + controller wave per 64 robots)
  int sched_refresh = 0;            // cdpr_update_scheduled in progress: Joy batches per launch (StepArgs::sched_*)
  const uint32_t* sched_ready = nullptr;
  // cdpr_update_scheduled_kind on a per-robot handle: batch j of the schedule is latched straight from the caller's device
  // buffers (rows d_commands + j * B * n, mask d_robot_masks + j * B or nullptr = every robot), nothing staged
  const float* sched_rows[3] = {nullptr, nullptr, nullptr};
  const uint8_t* sched_mask[3] = {nullptr, nullptr, nullptr};
  uint32_t* h_fault = nullptr;      // pinned, device-mapped status word: a schedule mailbox that never delivered (kernels OR bits into it)
  uint32_t* d_fault = nullptr;      // its device address
  uint32_t chunk = 0;       // > 0: a step over the batch is issued as back-to-back launches over contiguous blocks of at most
                            // this many robots (batches between one and ~5 robots per hardware lane: see cdpr_create)
  bool onestep_v2 = true;   // one-step launches use cdpr_onestep_kernel (controller rows through LDS); CDPR_ONESTEP=1: first generation
  // general controller path (hold branch, cascades, long windows): see cdpr_general_step.hpp
  bool general = false;
  float* d_rec = nullptr;    // record rows: [mLastPosition per cable][position Pid rows][velocity Pid rows], one column per robot
  float* d_gwtab = nullptr;  // FIR weights by ring head, [pid][head][slot]
  float* d_gptab = nullptr;  // the two Pids' parameters as the kernel stages them in LDS (gen_pid_table)
  GenPid gpid[2]{};
  GenLayout glay{};          // rows of a Pid block: sized by the configured window length and cascade count
  double* d_roll64 = nullptr;    // MPC rollout on a precision = 64 handle: the trajectories' state rows, their cost accumulators, the step's Joy batch
  double* d_roll64_acc = nullptr;
  float* d_roll64_cmd = nullptr;
  uint8_t* d_roll64_meta = nullptr;  // ... per-robot handles: every trajectory's mode / Pid call count byte
  size_t roll64_cols = 0;
  float* d_roll_rec = nullptr;   // MPC rollout on the general path: every trajectory's private copy of the records
  size_t roll_rec_cols = 0;      // columns d_roll_rec can hold
  // hipGraph cache: chains of identical steady-state launches (see run_steps)
  struct GraphEntry {
    void* kern;
    const float* cmd;
    int steps_per_launch, launches, start_slot;
    uint32_t flags;
    hipGraph_t graph;
    hipGraphExec_t exec;
  };
  std::vector<GraphEntry> graphs;
  PlannedKernel last_kernel;  // what the last step launch ran on (cdpr_kernel_name)
  int hold_win = kHoldWin;  // precision = 64, HOLD handles: samples a Pid record's window holds (kHoldWinLong with derivative windows of 12 .. 32 samples)
  int win64 = kWin;         // precision = 64: prior errors kept per cable (kWinLong on handles with windows of 12 .. 32 samples)
  KernelPlan plan;          // the routing cdpr_create took for this configuration (cdpr_select.hpp)
  int cus = 256;
  bool use_graphs = true;
  bool pair_stream = true;  // cdpr_pair_stream_kernel serves the steady several-steps launches of lane-pair handles (CDPR_PAIR_STREAM=0: never; A/B and tests)
  float* d_vel[2] = {nullptr, nullptr};  // [0] latched, [1] pending
  float* d_pos[2] = {nullptr, nullptr};
  float* d_frc[2] = {nullptr, nullptr};  // force commands (cdpr_set_force_command; JFC.h:92-95)
  // zero-copy commands (cdpr_bind_*_command_device): a caller-owned device buffer takes the place of d_*[0] / d_*[1]
  const float* ext_vel[2] = {nullptr, nullptr};
  const float* ext_pos[2] = {nullptr, nullptr};
  const float* ext_frc[2] = {nullptr, nullptr};
  // per-robot command arrival (cfg.per_robot_commands): every robot has its own mode; general controller path only
  bool per_robot = false;
  uint8_t* d_mode = nullptr;        // uint8[B]: 1 = Position, 2 = Velocity; on the register-resident path also the robot's
                                    // Pid call count in bits 2-7 (StepArgs::meta)
  float* d_target = nullptr;        // per-robot handles on the register-resident path: float[B][n], every robot's ACTIVE target row
  uint8_t* d_mask[3] = {nullptr, nullptr, nullptr};  // pending masks of the velocity / position / force command, uint8[B]
  bool vel_masked = false, pos_masked = false, frc_masked = false;  // the pending command came with a mask
  // Host-side Joy batches travel on their own stream (cdpr_set_*_command with a host pointer): the caller's rows go into
  // one of two pinned staging buffers per kind and from there to the PENDING device buffer while earlier launches still
  // run; the call returns without waiting.  kind 0 = velocity, 1 = position, 2 = force.
  hipStream_t copy_stream = nullptr;
  float* h_stage[3][2] = {{nullptr, nullptr}, {nullptr, nullptr}, {nullptr, nullptr}};
  hipEvent_t stage_ev[3][2] = {{nullptr, nullptr}, {nullptr, nullptr}, {nullptr, nullptr}};  // the copy out of that staging buffer has completed
  bool stage_ev_set[3][2] = {{false, false}, {false, false}, {false, false}};
  int stage_idx[3] = {0, 0, 0};
  hipEvent_t ready_wait[3] = {nullptr, nullptr, nullptr};  // event the compute stream has to pass before it touches the pending buffer
  hipEvent_t free_ev[3] = {nullptr, nullptr, nullptr};     // every launch that read what is now the pending buffer has completed
  bool free_ev_set[3] = {false, false, false};
  bool vel_pending = false, pos_pending = false, frc_pending = false;
  bool have_vel = false, have_pos = false, have_frc = false;  // a command of that kind has been latched since Load
  int mode = kModePosition;
  uint64_t step = 0;
  double prev_publish = 0.0;
  StepArgs base{};               // world/body/FK/TD constants, pointers; Pid fields filled per launch
  StepArgs pid_vel{}, pid_pos{};  // only the Pid fields of these are used
  float* d_wtab = nullptr;        // [velocity | position] rotated derivative-weight tables, kWin * (kWin + 2) floats each
  float wtab_host[2][kWin * (kWin + 2)]{};  // the same tables on the host: one-step launches take their row by value
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  uint64_t launches = 0, launches_mark = 0;
  // cdpr_get_observables: pinned, device-mapped host image of one published step + completion word
  float* h_pub = nullptr;        // host pointer (hipHostMalloc)
  uint64_t* h_pub_done = nullptr;
  uint32_t* d_pub_arrivals = nullptr;
  uint64_t pub_epoch = 0;
  // MPC rollout scratch, persistent and grow-only (no hipMalloc / hipFree inside a rollout)
  float* d_roll_ref = nullptr;   // float[B][3]
  float* d_roll_cost = nullptr;  // float[B][samples]
  size_t roll_cost_cap = 0;      // trajectories d_roll_cost can hold
  uint64_t roll_pending = 0;     // trajectories of the launched, not yet fetched rollout
  // cdpr_config_t.precision = 64: the step in double (cdpr_step_kernel_f64.hpp); its own state, observables, tables
  bool fp64 = false;
  bool tstop64 = false;  // ... with the joint stop modelled (travel_stop > 0; TSTOP kernels)
  bool hold64 = false;   // ... with the position-hold branch live (velocity_epsilon >= 0): both Pids of every cable in rows behind the state (HOLD kernels)
  double* d_state64 = nullptr;
  double* d_obs64 = nullptr;
  double* d_geom64 = nullptr;    // [n][7]
  double* d_wtab64 = nullptr;    // [velocity | position] x [10][12]
  double* d_dbg64 = nullptr;
  void* d_unpack64 = nullptr;    // read-out scratch of the fp64 getters (bytes)
  void* h_pub64 = nullptr;       // mapped pinned image the fp64 getters of small batches are gathered into (2 MiB)
  size_t unpack64_cap = 0;
  F64Args base64{};
  float* d_unpack = nullptr;     // read-out scratch (cdpr_get_*): robot-major copy of the requested fields, grow-only
  size_t unpack_cap = 0;
  std::string err;
};

namespace cdpr_host {

// Scratch device buffer holding caller data for the one-shot solvers.
struct DevBuf {
  void* p = nullptr;
  ~DevBuf() { if (p) (void)hipFree(p); }
  hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 4); }
  template <typename T> T* as() { return static_cast<T*>(p); }
};

constexpr int kCallSat = 64;  // Pid call counts saturate here on the host (the kernels ask "0?", ">= nbuf?")
inline int sat_pid_calls(int calls) { return calls < kCallSat ? calls : kCallSat; }
// ring slot the error of world step `step` is written to (fp32 kernels: a ring of kWin; fp64: of w = win64)
inline int ring_slot_of(uint64_t step) { return (int)((step + 8u) % (uint64_t)kWin); }
inline int ring_slot_of(uint64_t step, int w) { return (int)((step + (uint64_t)(w - 2)) % (uint64_t)w); }

// cdpr_engine.hip
hipError_t wait_stream(cdpr_engine* h);  // poll, then block
int set_device(cdpr_engine* h);
int check_fault(cdpr_engine* h);
int checked(cdpr_engine* h, int rc);
int derivative_weights(uint32_t n, uint32_t degree, double* w);
double sim_time(uint64_t step, double dt);
LaunchShape launch_shape(const cdpr_engine* h, int k, bool steady = false);
StepKernel step_kernel_of(const cdpr_engine* h, const PlannedKernel& pk);
StepKernel select_step_kernel(const cdpr_engine* h, int k, bool steady = false);
uint32_t step_block_threads(const cdpr_engine* h, int k);
void set_weight_row(const cdpr_engine* h, StepArgs& a);
void copy_pid(const StepArgs& src, StepArgs& dst);
void copy_pid_alt(const StepArgs& src, PidSet& dst);
GenCtl general_ctl(const cdpr_engine* h);
int fetch_slots(cdpr_engine* h, const float4* dsrc, int nslots, std::vector<float4>& host);
// cdpr_engine_f64.hip
size_t state64_rows(const cdpr_engine* h);
int upload_home64(cdpr_engine* h);
int run_steps_f64(cdpr_engine* h, int nsteps, int per_launch, bool reset_pid, double* record = nullptr);
int fetch_rows64(cdpr_engine* h, const double* rows, uint32_t first_row, uint32_t width, void* host_out, bool as_float);
int fetch_observables64(cdpr_engine* h, void* position, void* velocity, void* effort, void* pose7, void* twist6, bool as_float);
int set_platform_state64(cdpr_engine* h, const double* pose7, const double* twist6);
int fetch_int_row64(cdpr_engine* h, uint32_t row, int32_t* out);
void decode_image64_to_float(const cdpr_engine* h, const double* image, float* position, float* velocity, float* effort, float* pose7, float* twist6);
int rollout_enqueue_f64(cdpr_engine* h, int samples, int horizon, const float* d_commands, const float* d_ref, float* d_cost);

}  // namespace cdpr_host
