/*
 * cdpr.h — C-ABI of the MI355X-native batched CDPR step engine (libcdpr_hip.so).
 *
 * This is the drop-in boundary for the per-step hot path of the reference Gazebo
 * plugin `cdpr_gazebo` (balazs-bamer/cdpr-simulation).  The reference has no C ABI
 * of its own (it is a Gazebo ModelPlugin registered by GZ_REGISTER_MODEL_PLUGIN,
 * CdprGazeboPlugin.h:105); every entry point below names the reference interface
 * it replaces so a maintainer can bind it from the plugin shell (see
 * INTEGRATION.md for the C++ stub).  Paths are relative to
 * src/cdpr_gazebo/ in the reference tree:
 *   PLG.h/.cpp = include/cdpr_gazebo/CdprGazeboPlugin.h, src/CdprGazeboPlugin.cpp
 *   JFC.h/.cpp = .../JointForceCalculator.h/.cpp      Pid.h/.cpp, Filter.h likewise
 *
 * Conventions
 *   - plain C, no C++ exceptions cross the boundary; every call returns an int:
 *       0  = CDPR_OK, >0 = accepted-but-ignored (mirrors the reference's silent
 *       drops), <0 = error (text via cdpr_last_error()).
 *   - the caller owns every host buffer passed in or out; the library owns all
 *     device state behind the opaque handle and never retains a caller pointer
 *     after the call returns (PLG.cpp:69,78: messages are copied on receipt).
 *   - batched payloads are robot-major ("leading B dimension"): a Joy.axes of
 *     one robot is float[n]; the batch is float[B][n].
 *   - a handle is bound to one GPU and is not thread-safe (the reference runs
 *     update() and both callbacks on the one physics thread, PLG.cpp:203-204).
 *   - quaternions are x,y,z,w on the wire (PLG.cpp:266-269).
 *   - sign: positive joint position / velocity / force = cable shortening /
 *     tension (prismatic axis = -u, gen_cdpr.py:181, cube.sdf:434).
 */
#ifndef CDPR_H_
#define CDPR_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CDPR_ABI_VERSION 6u   /* 6 = 5 + cdpr_plan_kernel, cdpr_kernel_name, CDPR_MAX_CABLES 8 -> 12 (cdpr_config_t's anchor arrays grow); 5 = 4 + cdpr_update_scheduled_kind, cdpr_device_pci_bus_id, cdpr_decode_observables_f64 */
#define CDPR_MAX_CABLES 12u         /* PLG.h:20 fixes 4 (kCableCount); cube.yaml:21-29 is a free-length `points` list: the engine takes 1..12
                                      (9..12: uniform-mode fp32 handles on the lane-per-robot kernels, FK and TD included; see cdpr_create) */
#define CDPR_MAX_D_BUFFER 32u       /* Pid: mDbufferLength                      */
#define CDPR_MAX_D_DEGREE 4u        /* Pid: mDpolynomialDegree                  */
#define CDPR_MAX_CASCADE 4u         /* Pid::CascadeFilter: mCascade             */
#define CDPR_PID_DEBUG_AXES 9u      /* PLG.cpp:194  pidMsg.axes.resize(9)       */

/* return codes */
#define CDPR_OK 0
#define CDPR_IGNORED 1              /* wrong-length command, dropped (PLG.cpp:68-73,77-82) */
#define CDPR_ERR_INVALID (-1)       /* bad argument / bad configuration         */
#define CDPR_ERR_DEVICE (-2)        /* HIP runtime failure                      */
#define CDPR_ERR_UNSUPPORTED (-3)   /* valid request the engine cannot serve    */
#define CDPR_ERR_NOMEM (-4)

/* cdpr_config_t.stages: optional stages added to the always-on IK -> PID -> dynamics loop */
#define CDPR_STAGE_FK 0x1u          /* Newton-Raphson forward kinematics (n >= 6) */
#define CDPR_STAGE_TD 0x2u          /* closed-form tension distribution  (n >= 6) */
#define CDPR_STAGE_PID_DEBUG 0x4u   /* record the `pid` debug topic for cable 0 (PLG.cpp:223-227,233-235) */

/* cdpr_config_t.mapping: how robots are laid onto wavefronts */
#define CDPR_MAP_AUTO 0u
#define CDPR_MAP_LANE_PER_ROBOT 1u  /* one lane owns one robot                                               */
#define CDPR_MAP_LANE_PAIR 2u       /* two adjacent lanes share a robot, half the cables each (n = 4 or 8):   */
                                    /* twice the wavefronts, partial sums meet through a DPP add             */
#define CDPR_MAP_LANE_PER_CABLE 3u  /* one lane owns one cable, 8 (n <= 4: 4) adjacent lanes a robot: the structure matrix */
                                    /* row by row in registers, J^T J / J^T r by DPP reductions inside the group, the 6x6  */
                                    /* solve redundantly in every lane (BASELINE.json's north-star mapping; wins only     */
                                    /* while the batch leaves most SIMDs empty: DESIGN.md section 4)                     */

/* Pid::FilterParameters (Pid.h:64-68) */
typedef struct cdpr_filter_params {
  double rel_cutoff;                /* relCutoff: fc relative to fs = 1 (Pid.cpp:34) */
  double quality;                   /* Q                                               */
  uint32_t cascade;                 /* number of identical biquads in series; 0 = bypass */
  uint32_t reserved_;
} cdpr_filter_params_t;

/* Pid::PidParameters (Pid.h:70-81) */
typedef struct cdpr_pid_params {
  double forward_gain;
  double p_gain;
  double i_gain;
  double d_gain;
  uint32_t d_degree;                /* polynomial degree of the derivative fit */
  uint32_t d_buffer_length;         /* samples in the derivative window        */
  double i_limit;
  double cmd_limit;
  cdpr_filter_params_t p_filter;
  cdpr_filter_params_t d_filter;
} cdpr_pid_params_t;

/*
 * Everything CdprGazeboPlugin::Load reads from the ROS parameter server
 * (PLG.h:32-54, PLG.cpp:57,102-138) plus the model constants Gazebo takes from
 * cube.sdf / cube.yaml (anchors, mass, inertia, damping, effort limit) and the
 * world constants that are Gazebo defaults (dt, gravity).
 */
typedef struct cdpr_config {
  uint32_t abi_version;             /* CDPR_ABI_VERSION */
  uint32_t n_cables;                /* PLG.h:20 cWireCount */
  uint64_t batch;                   /* B independent robots on this handle */
  double dt;                        /* world step, Gazebo default 1e-3 s */

  double frame_anchor[CDPR_MAX_CABLES][3];     /* a_i, frame coords    (cube.yaml:21-29 `frame`)    */
  double platform_anchor[CDPR_MAX_CABLES][3];  /* b_i, platform coords (cube.yaml:21-29 `platform`) */
  double cable_ref_length[CDPR_MAX_CABLES];    /* L0_i = cable length at joint position 0           */
  double home_pose[7];              /* x y z qx qy qz qw (cube.sdf:310); state after create/reset   */

  double mass;                      /* cube.sdf:340 */
  double inertia[6];                /* ixx iyy izz ixy ixz iyz, body frame (cube.sdf:332-339) */
  double gravity[3];                /* frame coords; Gazebo default (0,0,-9.8) */
  double joint_damping;             /* cube.sdf:442 actuated-joint damping */
  double effort_limit;              /* cube.sdf:438; Joint::SetForce clamp; < 0 disables */
  double velocity_limit;            /* cube.sdf:439 (10); Joint::SetForce drops a force that pushes a joint already past
                                       +-limit further out [EXT Gazebo]; <= 0 disables (the contract's reduced model,
                                       and what a zero-initialised struct gets) */
  uint32_t unilateral_cables;       /* 1: a cable cannot push, axial force max(T, 0) ([NEW]; the reference has no slack model) */
  uint32_t precision;               /* 0 or 32: fp32 kernels (what every throughput figure is quoted on).  64: the step in the
                                       reference's own precision (Pid.h, Gazebo/ODE compute in double): one plain fp64 kernel for
                                       handles on the register-resident path (IK, Pid, FK, TD, limits, observables, world step; any steps
                                       per launch; trajectory records, command schedules and per_robot_commands since ABI 5), meant for
                                       one robot / small batches; read it out with the *_f64 getters.  velocity_epsilon >= 0 (the position-hold
                                       branch, JFC.cpp:72-82: both Pids of every cable alive, derivative windows on real stamps) is served in
                                       double too (round 5), with the rest of Pid::update - biquad cascades, cmd_limit = 0 - and with
                                       per_robot_commands (two different derivative windows included).  Round 6: the joint stop
                                       (travel_stop > 0), the lumped legs, derivative windows to 32 samples and cdpr_rollout_velocity* in
                                       double as well; every combination of the controller's and the physics' options is served, nine to
                                       twelve cables where the fp32 kernels serve them */

  cdpr_pid_params_t velocity_pid;   /* PLG.cpp:102-120 */
  cdpr_pid_params_t position_pid;   /* PLG.cpp:123-134 (forward gain and filters are forced to 0 by the facade) */
  double velocity_epsilon;          /* PLG.cpp:137-138, JFC.cpp:72 */
  double publish_period;            /* PLG.cpp:57,237; observables are refreshed when now - prev > period */

  uint32_t stages;                  /* CDPR_STAGE_* */
  uint32_t mapping;                 /* CDPR_MAP_*   */
  uint32_t fk_max_iterations;       /* Newton-Raphson iteration cap */
  uint32_t per_robot_commands;      /* 1: every robot has its own JointForceCalculator mode and Pid call history, as B
                                       independent plugin instances have (PLG.cpp:206-219 runs per model): a Joy may reach
                                       some robots only (cdpr_set_*_command_masked).  Runs on the same register-resident kernels as a uniform handle
                                       (mode and Pid call count per lane), unless the configuration needs the general controller path.
                                       0: a Joy batch always addresses every robot (mode uniform over the batch) */
  double fk_lambda;                 /* Levenberg damping added to the diagonal of J^T J */
  double fk_tolerance;              /* stop when max_i |L*_i - L_i| < tol; 0 = always run the cap */
  double td_f_min;                  /* cube.yaml:9 `min: 5`   */
  double td_f_max;                  /* cube.yaml:9 `effort`   */

  /* What the massless-cable reduction drops of the 22-link SDF model, as lumped first-order terms ([EXT] Gazebo/ODE
   * integrates these bodies and joints; closed forms: DESIGN.md section 1).  All 0 = the contract's reduced model. */
  double passive_damping;           /* damping of EVERY passive revolute joint of a leg: the frame-side universal pair
                                       rev_X / rev_Y and the platform-side spherical triple rev_Xpf / rev_Ypf / rev_Zpf
                                       (cube.sdf:396,425,471,500,515: 0.01) */
  double leg_inertia;               /* inertia of the links that turn with a cable about its frame anchor: virt_X, virt_Y,
                                       the cable link, virt_Ypf (cube.sdf:359-366,...: 4 x 0.001 kg m^2) */
  double cable_axial_mass;          /* mass that slides along the cable axis with the prismatic joint: the cable link
                                       (cube.sdf:368: 0.001 kg) */
  double anchor_point_mass;         /* mass carried at each platform anchor: virt_Xpf, virt_Ypf (cube.sdf:456,485: 2 x 0.001 kg) */
  double anchor_inertia;            /* inertia each leg adds to the platform: virt_Xpf (cube.sdf:448-455: 0.001 kg m^2) */

  /* Travel limits of the prismatic joints (cube.sdf:436-437: <lower>-0.5196</lower> <upper>0.5196</upper>; [EXT]
   * Gazebo/ODE enforces them as joint stops).  lower == upper == 0 (a zero-initialised struct): no limits.  With limits
   * set, every step raises a per-cable flag where the joint position q_i = L0_i - L_i lies outside [lower, upper]
   * (cdpr_get_limit_state).  travel_stop > 0 additionally models the stop itself, inelastically: after the velocity
   * update of the world step, a joint at or beyond a limit that is still moving outward (q_i >= upper and qdot_i > 0, or
   * q_i <= lower and qdot_i < 0) takes the impulse that brings its rate to zero, lambda = qdot_i / (J_i M^-1 J_i^T),
   * twist += M^-1 J_i^T lambda, with M the platform's own mass and inertia; cables in index order, travel_stop sweeps
   * over the cables (projected Gauss-Seidel on the velocity constraints: what ODE's quickstep iterates 50 times [EXT];
   * with several joints on their stops one sweep lets them creep, 4 sweeps hold them to within one step's travel;
   * DESIGN.md section 1). */
  double travel_lower;
  double travel_upper;
  uint32_t travel_stop;             /* 0: flag only; k > 0: k sweeps of the stop (<= 64) */
  uint32_t reserved3_;
} cdpr_config_t;

typedef struct cdpr_engine *cdpr_handle_t;

/* Library-level queries (no GPU needed). */
uint32_t cdpr_abi_version(void);
size_t cdpr_config_size(void);                       /* sizeof(cdpr_config_t) as compiled */
int cdpr_device_count(void);                         /* visible GPUs, 0 if none */
/* PCI address "dddd:bb:dd.f" of HIP ordinal `device` (len >= 13): a multi-process host that also runs another GPU runtime
 * (bench.py: torch.distributed over RCCL) checks with it that both runtimes mean the same GPU by ordinal r. */
int cdpr_device_pci_bus_id(int device, char *out, size_t len);
/* Algorithmic HBM bytes one robot moves per state-step for this configuration
 * (state round trip + command read + observables written); see DESIGN.md. */
size_t cdpr_bytes_per_state_step(const cdpr_config_t *cfg);
/* Least-squares end-point derivative weights for an N-sample, degree-d window
 * on a uniform grid (the closed form of Pid::derive + fitPolynomial,
 * Pid.cpp:193-247): derivative = (1/dt) * sum_j w[j] * e[j], oldest first. */
int cdpr_derivative_weights(uint32_t n, uint32_t degree, double *w);

/* Replaces CdprGazeboPlugin::Load (PLG.cpp:49-65): validates the configuration
 * (invalid cable count -> error, PLG.cpp:167-168), allocates device state for
 * `batch` robots on GPU `device`, puts every robot at home_pose with zero twist
 * and every JointForceCalculator in Position mode with target 0
 * (PLG.cpp:153-157, JFC.cpp:38-51). */
int cdpr_create(const cdpr_config_t *cfg, int device, cdpr_handle_t *out);
void cdpr_destroy(cdpr_handle_t h);
/* World reset: back to the state cdpr_create leaves (JFC.h:69-73 reset()). */
int cdpr_reset(cdpr_handle_t h);
const char *cdpr_last_error(cdpr_handle_t h);       /* h may be NULL: last create error */

/* Overwrite the platform state of every robot: pose7[B][7] (x y z qx qy qz qw),
 * twist6[B][6] (world-frame linear, angular).  Either may be NULL (unchanged).
 * Stands in for spawning the model at a pose (launch file `-x -y -z -R -P -Y`). */
int cdpr_set_platform_state(cdpr_handle_t h, const float *pose7, const float *twist6);

/* Replaces cableVelocityCommandCallback / cablePositionCommandCallback
 * (PLG.cpp:67-83).  `count` = number of floats in `axes`: n*B (one Joy per
 * robot) or n (one Joy broadcast to every robot).  Any other count returns
 * CDPR_IGNORED and changes nothing.  The command is latched at the next
 * cdpr_update and zero-order-held until replaced (PLG.cpp:206-219).
 * `axes` is host memory; it is copied before the call returns (the caller may
 * reuse it at once) and travels to the device on a copy stream of the handle's
 * own while earlier launches still run: the call does not wait for them. */
int cdpr_set_velocity_command(cdpr_handle_t h, const float *axes, size_t count);
int cdpr_set_position_command(cdpr_handle_t h, const float *axes, size_t count);
/* Same, from a device buffer already resident in HBM (float[B][n]); the batch is copied (device to device). */
int cdpr_set_velocity_command_device(cdpr_handle_t h, const float *d_axes, size_t count);
int cdpr_set_position_command_device(cdpr_handle_t h, const float *d_axes, size_t count);
/* Zero-copy form for command schedules that live in HBM: the caller's device buffer float[B][n] (count = n*B) IS the
 * latched Joy batch from the next cdpr_update on; nothing is copied or synchronised.  The buffer must stay valid and
 * unchanged until another command of the same kind has been latched.  (The one exception to "no caller pointer is
 * retained": that is its point.) */
int cdpr_bind_velocity_command_device(cdpr_handle_t h, const float *d_axes, size_t count);
int cdpr_bind_position_command_device(cdpr_handle_t h, const float *d_axes, size_t count);
/* Per-robot arrival: the Joy batch reaches only the robots with robot_mask[b] != 0 (uint8[B], host); the others keep
 * their target, their mode and their Pid state, exactly as plugin instances that received nothing before this
 * update() (PLG.cpp:206-219 is per model).  axes: float[B][n] (rows of unmasked robots are ignored) or float[n]
 * broadcast to the masked robots.  Needs cdpr_config_t.per_robot_commands = 1, else CDPR_ERR_UNSUPPORTED. */
int cdpr_set_velocity_command_masked(cdpr_handle_t h, const float *axes, size_t count, const uint8_t *robot_mask);
int cdpr_set_position_command_masked(cdpr_handle_t h, const float *axes, size_t count, const uint8_t *robot_mask);

/* Replaces JointForceCalculator::setForce (JFC.h:92-95; UpdateMode::Force, JFC.h:35-42, JFC.cpp:67-70): the joints are
 * driven open loop, force_i = axes[i], held until another command arrives: `mLastPosition = joint position; force =
 * mForce`, no Pid runs.  This is the mode a JointForceCalculator is constructed in (JFC.h:42); the shipped plugin never
 * calls setForce (no topic reaches it), a tension-distribution / MPC caller does.  Same count rules as the two Joy
 * callbacks (n*B or n, anything else CDPR_IGNORED).  Latched at the next cdpr_update AFTER a pending velocity and a
 * pending position command ([NEW] ordering: the reference has no force callback to order against; the last setter
 * wins).  setForce resets no Pid; leaving Force mode through a velocity / position command resets the Pid of the mode
 * entered (JFC.cpp:99-119), as from any other mode.  The optional stages still apply: tension distribution redistributes
 * the commanded forces, SetForce limits clamp them.  _masked needs per_robot_commands = 1. */
int cdpr_set_force_command(cdpr_handle_t h, const float *axes, size_t count);
int cdpr_set_force_command_device(cdpr_handle_t h, const float *d_axes, size_t count);
int cdpr_bind_force_command_device(cdpr_handle_t h, const float *d_axes, size_t count);
int cdpr_set_force_command_masked(cdpr_handle_t h, const float *axes, size_t count, const uint8_t *robot_mask);

/* Replaces nsteps x { CdprGazeboPlugin::update (PLG.cpp:202-246) followed by
 * the Gazebo/ODE world step }.  Asynchronous: returns once the work is queued
 * on the handle's stream. */
int cdpr_update(cdpr_handle_t h, int nsteps);
/* Same loop with `steps_per_launch` world steps fused into each kernel launch
 * (state stays on chip between them; observables still written every step). */
int cdpr_update_fused(cdpr_handle_t h, int nsteps, int steps_per_launch);
/* Trajectory record: like cdpr_update_fused, but the observables of EVERY step are kept instead of each step
 * overwriting the last (what a subscriber with a deep queue would have collected from jointStates / platformPose at
 * publishPeriod 0).  d_record is a caller-owned DEVICE buffer of nsteps observable images
 * (cdpr_observable_image_bytes each); image j belongs to world step first+j.  Step 0 of a fresh handle is not
 * published (PLG.cpp:237), its image is left untouched.  Decode a downloaded image with cdpr_decode_observables. */
int cdpr_observable_image_bytes(cdpr_handle_t h, size_t *bytes);
int cdpr_update_record(cdpr_handle_t h, int nsteps, int steps_per_launch, void *d_record, size_t record_bytes);
int cdpr_decode_observables(cdpr_handle_t h, const void *image, float *position, float *velocity, float *effort,
                            float *pose7, float *twist6);
/* precision = 64 handles: their images hold doubles (cdpr_observable_image_bytes says how many bytes); this decodes without
 * the rounding to float (cdpr_decode_observables works on them too and rounds).  CDPR_ERR_UNSUPPORTED on fp32 handles. */
int cdpr_decode_observables_f64(cdpr_handle_t h, const void *image, double *position, double *velocity, double *effort,
                                double *pose7, double *twist6);
/* A whole command schedule resident in HBM, queued with one call: what
 *   for j: cableVelocityCommandCallback(batch j); refresh_steps x update()
 * does (the reference's 100 Hz / 10 Hz publishers against the 1 kHz world: a Joy every 10 or 100 world steps, PLG.cpp:206-211
 * + sinevelocitytest.cpp:34-48, squarevelocitytest.cpp:20-34, squarepositiontest.cpp:21-35), for callers whose batch is too
 * small for a launch per step to pay (a 4 096 x 4-cable step is 1.4 us of work behind 3-4 us of launch).
 * d_commands: DEVICE buffer float[ceil(nsteps / refresh_steps)][B][n], batch j is latched at world step first + j *
 * refresh_steps.  d_ready: optional DEVICE-visible mailbox uint32[batches]: batch j is taken only once d_ready[j] != 0 (a
 * host or a producer kernel that fills the schedule while the work runs.  A HOST producer should keep the words in pinned host memory
 * mapped to the device and post them by plain stores: a copy enqueued on another stream can be placed on the hardware queue of the
 * launch that is waiting for it and then never completes);
 * NULL = the whole schedule is there.  The wait is bounded (~2^23 polls, a few seconds): a mailbox that never delivers
 * does not hang the GPU - the handle's status word is raised instead and cdpr_synchronize and the getters return
 * CDPR_ERR_DEVICE until cdpr_reset.  d_record: as cdpr_update_record (every step's observable image kept; needs
 * publish_period == 0), or NULL (each published step overwrites the last, as the topic does).
 * Bit-identical to the call sequence above (tested).  Afterwards the last batch stays latched (on uniform-mode handles it
 * is read in place and must stay valid like a bound buffer).  Asynchronous.
 *
 * cdpr_update_scheduled_kind: the same for any command kind - jointVelocities, jointPositions (squarepositiontest) or
 * setForce (JFC.h:92-95; a tension-distribution / MPC caller) - and, on per_robot_commands handles, with one robot mask per
 * batch (d_robot_masks: DEVICE buffer uint8[batches][B], batch j reaches the robots with mask[j][b] != 0; NULL = every
 * robot), i.e. for j: cdpr_set_<kind>_command_masked(batch j, mask j); refresh_steps x update().
 * Every handle type is served.  Uniform-mode handles on the register-resident path with one or two lanes per robot run the
 * schedule in ONE launch (the lanes read batch j themselves; state, windows and integrals stay on chip); the others
 * (general controller path, per-robot modes, precision = 64, one lane per cable) latch batch j from the caller's buffers in
 * place and queue its steps, batch after batch, without a host round trip.  A command of ANOTHER kind that is pending at
 * the call is latched together with batch 0 in update()'s order (velocity, position, force: PLG.cpp:206-219), exactly as
 * the call sequence would. */
#define CDPR_COMMAND_VELOCITY 0u    /* jointVelocities, cableVelocityCommandCallback (PLG.cpp:67-74) */
#define CDPR_COMMAND_POSITION 1u    /* jointPositions, cablePositionCommandCallback (PLG.cpp:76-83)  */
#define CDPR_COMMAND_FORCE 2u       /* JointForceCalculator::setForce (JFC.h:92-95)                   */
int cdpr_update_scheduled(cdpr_handle_t h, int nsteps, int refresh_steps, const float *d_commands, const uint32_t *d_ready,
                          void *d_record, size_t record_bytes);
int cdpr_update_scheduled_kind(cdpr_handle_t h, uint32_t kind, int nsteps, int refresh_steps, const float *d_commands,
                               const uint32_t *d_ready, const uint8_t *d_robot_masks, void *d_record, size_t record_bytes);
/* Waits until everything queued on the handle has completed.  Every wait of the library polls the stream for up to 2 ms
 * before it blocks (a blocked host thread wakes up 15-25 us late, two step kernels; environment CDPR_SYNC_SPIN_US
 * overrides, 0 = always block). */
int cdpr_synchronize(cdpr_handle_t h);
uint32_t cdpr_mapping(cdpr_handle_t h);              /* CDPR_MAP_* actually in use (what CDPR_MAP_AUTO resolved to) */
/* Which kernel serves a launch - the routing CDPR_MAP_AUTO takes, answered WITHOUT a GPU from the configuration alone (a
 * 256-CU part is assumed; the CDPR_* A/B environment overrides are honoured as cdpr_create honours them).  No counterpart
 * in the reference (one robot, one code path: PLG.cpp:202-246); what it replaces is "run it and read the profiler".
 *   steps_per_launch  world steps of the launch (1 = cdpr_update's launches)
 *   flags             CDPR_PLAN_FIRST_WORLD_STEP: the launch starts at world step 0; CDPR_PLAN_SCHEDULED: cdpr_update_scheduled's
 *                     in-launch form; CDPR_PLAN_ROLLOUT: cdpr_rollout_velocity*; CDPR_PLAN_NOT_STEADY: a derivative window still
 *                     filling, Force mode, publish decimation, pid debug topic, travel flags, velocity limit, unilateral cables
 *                     or a mailbox (launches of several steps on lane-pair handles then keep the general kernel)
 * name receives the kernel's name with its template arguments as rocprofv3 prints the family (NUL-terminated, truncated to
 * len).  Returns CDPR_OK, or the code cdpr_create would return for this configuration (name = the reason).
 * cdpr_kernel_name: the same for the LAST step launch a handle made (what really ran). */
#define CDPR_PLAN_FIRST_WORLD_STEP 1u
#define CDPR_PLAN_SCHEDULED 2u
#define CDPR_PLAN_ROLLOUT 4u
#define CDPR_PLAN_NOT_STEADY 8u
int cdpr_plan_kernel(const cdpr_config_t *cfg, int steps_per_launch, uint32_t flags, char *name, size_t len);
int cdpr_kernel_name(cdpr_handle_t h, char *name, size_t len);
uint64_t cdpr_step_count(cdpr_handle_t h);           /* world steps since create/reset; sim time = count * dt */

/* Replaces publishJointStates (PLG.cpp:248-256): sensor_msgs/JointState
 * position / velocity / effort, float[B][n] each, as of the last published
 * step.  Any pointer may be NULL.  Synchronises the stream. */
int cdpr_get_joint_states(cdpr_handle_t h, float *position, float *velocity, float *effort);
/* Replaces publishPlatformState (PLG.cpp:258-280): cdpr_gazebo/PlatformState
 * pose7[B][7] (x y z qx qy qz qw) and twist6[B][6] (linear, angular), relative
 * to the frame link. */
int cdpr_get_platform_state(cdpr_handle_t h, float *pose7, float *twist6);
/* Both messages of one published step in one device round trip: the five arrays of cdpr_get_joint_states and
 * cdpr_get_platform_state (any of them may be NULL), as of the last published step.  What a per-step caller uses
 * (PLG.cpp:236-242 publishes jointStates and platformPose on every step): one gather launch into a pinned host image and
 * a completion word the host spins on, instead of five gather / copy / wait rounds. */
int cdpr_get_observables(cdpr_handle_t h, float *position, float *velocity, float *effort, float *pose7, float *twist6);
/* Handles created with cdpr_config_t.precision = 64: the same read-outs in double (the float getters work too and round).
 * Any pointer may be NULL.  cdpr_get_raw_state_f64 = the current state (not decimated by publish_period);
 * cdpr_set_platform_state_f64 = cdpr_set_platform_state without the rounding to float.  CDPR_ERR_UNSUPPORTED on fp32 handles. */
int cdpr_get_observables_f64(cdpr_handle_t h, double *position, double *velocity, double *effort, double *pose7, double *twist6);
int cdpr_get_raw_state_f64(cdpr_handle_t h, double *pose7, double *twist6);
int cdpr_set_platform_state_f64(cdpr_handle_t h, const double *pose7, const double *twist6);
/* Replaces the `pid` debug topic (PLG.cpp:193-194,223-227,233-235;
 * Pid.cpp:139-142,158-168): axes9[B][9] = P term, I term before clamp, D term,
 * desired, applied force of cable 0, then four unused zeros.  Needs
 * CDPR_STAGE_PID_DEBUG. */
int cdpr_get_pid_debug(cdpr_handle_t h, float *axes9);
/* Forward-kinematics estimator (CDPR_STAGE_FK): pose7[B][7] = the estimate after the last step; residual[B] =
 * max_i |L*_i - L_i(estimate)| and iterations[B] travel with the observables: as of the last PUBLISHED step. */
int cdpr_get_fk_state(cdpr_handle_t h, float *pose7, float *residual, int32_t *iterations);
/* Tension distribution of the last PUBLISHED step (CDPR_STAGE_TD): tension[B][n] = the force applied to the joints
 * (after the bounds and the SetForce limits: the `effort` observable), infeasible[B] = 1 where a bound was active. */
int cdpr_get_td_state(cdpr_handle_t h, float *tension, int32_t *infeasible);
/* Travel limits (cdpr_config_t.travel_lower / travel_upper; cube.sdf:436-437): cable_mask[B], bit i set where joint i's
 * position was outside [lower, upper] at the last PUBLISHED step (travels with the observables).  All zero when the
 * configuration sets no limits. */
int cdpr_get_limit_state(cdpr_handle_t h, uint32_t *cable_mask);
/* Current platform state (not decimated by publish_period), for checkpoints
 * and tests: pose7[B][7], twist6[B][6]. */
int cdpr_get_raw_state(cdpr_handle_t h, float *pose7, float *twist6);

/* MPC fan-out (BASELINE config 5): from every robot's CURRENT state run `samples` hypothetical trajectories of
 * `horizon` world steps, each driven by its own jointVelocities sequence, and return one cost per trajectory,
 * cost[B][samples] = sum over the horizon of |p(t_{k+1}) - ref_position|^2.  d_commands is a DEVICE buffer
 * float[B][horizon][samples][n] (per robot and step, a batch of `samples` Joy.axes); ref_position[B][3] and cost
 * are host buffers.  The handle's state is not modified; trajectories never leave the chip.  Every stage the
 * handle was created with (FK, TD) runs in every step.  Synchronous. */
int cdpr_rollout_velocity(cdpr_handle_t h, int samples, int horizon, const float *d_commands,
                          const float *ref_position, float *cost);
/* The same rollout split for hosts that drive several GPUs from one thread (one handle per GPU): _launch stages
 * ref_position and queues the kernel on the handle's stream and returns at once (no allocation in steady state: the
 * scratch buffers are persistent); _fetch copies cost[B][samples] of the pending rollout back and synchronises.
 * Launch on every handle first, then fetch from each: all GPUs run concurrently. */
int cdpr_rollout_velocity_launch(cdpr_handle_t h, int samples, int horizon, const float *d_commands,
                                 const float *ref_position);
int cdpr_rollout_velocity_fetch(cdpr_handle_t h, float *cost);
/* Fully device-resident form: reference positions float[B][3] and costs float[B][samples] in caller-owned DEVICE
 * buffers; asynchronous on the handle's stream, nothing copied (an MPC loop that keeps its sampler on the GPU). */
int cdpr_rollout_velocity_device(cdpr_handle_t h, int samples, int horizon, const float *d_commands,
                                 const float *d_ref_position, float *d_cost);

/* Device-buffer helpers for hosts that have no GPU runtime of their own (ctypes):
 * allocate / free / fill a caller-owned device buffer on the handle's GPU, e.g. to keep
 * a schedule of Joy batches resident in HBM for cdpr_set_*_command_device. */
int cdpr_device_malloc(cdpr_handle_t h, size_t bytes, void **out);
int cdpr_device_free(cdpr_handle_t h, void *ptr);
int cdpr_device_upload(cdpr_handle_t h, void *dst, const void *src, size_t bytes);
int cdpr_device_download(cdpr_handle_t h, void *dst, const void *src, size_t bytes);

/* Timing of the step kernel on the handle's own stream with HIP events:
 * begin records an event, end records another, synchronises THE STREAM (on return
 * everything queued on the handle has completed, as after cdpr_synchronize), and
 * returns the elapsed milliseconds and the number of step-kernel launches in between. */
int cdpr_profile_begin(cdpr_handle_t h);
int cdpr_profile_end(cdpr_handle_t h, float *elapsed_ms, uint64_t *kernel_launches);

/* One-shot batched solvers on caller data (host buffers, float32), for
 * callers that want the kinematics without the step loop.
 * IK  (Joint::Position / GetVelocity restated, JFC.cpp:68,75-76): pose7[B][7],
 *     twist6[B][6] -> q[B][n], qdot[B][n], jac[B][n][6]. Outputs may be NULL. */
int cdpr_solve_ik(cdpr_handle_t h, const float *pose7, const float *twist6, float *q, float *qdot, float *jac);
/* FK: lengths[B][n], seed7[B][7] -> pose7[B][7], residual[B], iterations[B]. */
int cdpr_solve_fk(cdpr_handle_t h, const float *lengths, const float *seed7, float *pose7, float *residual,
                  int32_t *iterations);
/* TD: pose7[B][7], wrench6[B][6] (wrench the cables must apply to the
 *     platform) -> tension[B][n], infeasible[B]. */
int cdpr_solve_td(cdpr_handle_t h, const float *pose7, const float *wrench6, float *tension, int32_t *infeasible);

#ifdef __cplusplus
}
#endif
#endif /* CDPR_H_ */
