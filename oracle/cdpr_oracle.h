/*
 * cdpr_oracle.h — CPU fp64 restatement of the cdpr_gazebo per-step path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product (cdpr-simulation_amd/, the
 * C-ABI library, the facade) may include, link or call this; only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, as the checker.
 *
 * Parity status (see DESIGN.md "Oracle"):
 *   - BiQuad filter: PINNED against the reference's own Filter.h compiled here
 *     (oracle/_ref/ref_filter, recipe in oracle/Makefile).
 *   - Geometry / IK at the home pose: PINNED against cube.sdf numbers and the
 *     reference's transformations.py (tests/golden/geometry.json).
 *   - Pid / JointForceCalculator / update() ordering: restated line by line from
 *     Pid.cpp, JointForceCalculator.cpp, CdprGazeboPlugin.cpp.  These files need
 *     Gazebo, ROS and Eigen headers that this image lacks, so they are
 *     unbuildable here and the reference ships no tests or golden vectors:
 *     PARITY UNPINNED (spot values recorded in SURVEY.md Appendix A are checked
 *     as known-answer tests but are not a reproducible pin).
 *   - Platform dynamics: analytic restatement of what Gazebo/ODE (not vendored)
 *     does to the platform link: PARITY UNPINNED, self-consistency KATs only.
 *   - Newton-Raphson FK and tension distribution: not in the reference at all
 *     (named by the north star): PARITY UNPINNED, round-trip / residual tests.
 *
 * Reference paths below are relative to /root/reference/src/cdpr_gazebo/.
 */
#ifndef CDPR_ORACLE_H_
#define CDPR_ORACLE_H_

#include "../include/cdpr.h"

#ifdef __cplusplus
extern "C" {
#endif

/* how Pid::derive's polynomial fit is evaluated */
#define ORC_DERIV_FAITHFUL 0 /* normal equations in absolute sim time + pow(), Pid.cpp:219-247 (drifts for t >~ 2 s) */
#define ORC_DERIV_EXACT 1    /* same least-squares problem in centred, scaled time: the exact answer        */
#define ORC_DERIV_FIR 2      /* BASELINE.md section 3's `fir` mode: a window whose samples are one step apart takes the fixed
                                end-point filter (11 taps at the shipped N = 11, d = 2), any other window
                                the EXACT fit; the CPU counterpart of the product's cdpr_derivative_weights */

/* ---- unit-level pieces (each is one reference function restated) ---- */

typedef struct orc_biquad {
  double a0, a1, a2, b1, b2;
  double x1, x2, y1, y2;
} orc_biquad;
void orc_biquad_set_fc(orc_biquad *f, double fc, double fs, double q); /* Filter.h:130-140 */
void orc_biquad_set_value(orc_biquad *f, double v);                    /* Filter.h:144-147 */
double orc_biquad_process(orc_biquad *f, double x);                    /* Filter.h:152-165 */

typedef struct orc_pid {
  cdpr_pid_params_t prm;
  double i_max, i_min, cmd_max, cmd_min;
  int deriv_mode;
  int was_last_time;
  double last_time, perr, ierr, derr, cmd;
  orc_biquad pf[CDPR_MAX_CASCADE], df[CDPR_MAX_CASCADE];
  unsigned missing;
  double bx[CDPR_MAX_D_BUFFER], by[CDPR_MAX_D_BUFFER];
  double fir[CDPR_MAX_D_BUFFER]; /* ORC_DERIV_FIR: weights, oldest sample first, unit spacing */
  /* what the `pid` debug topic would carry if this PID served cable 0 */
  double dbg_p, dbg_i, dbg_d, dbg_desired;
  int dbg_pi_written, dbg_d_written, dbg_desired_written;
} orc_pid;
void orc_pid_init(orc_pid *p, const cdpr_pid_params_t *prm, int deriv_mode); /* Pid.cpp:63-77  */
void orc_pid_reset(orc_pid *p);                                              /* Pid.cpp:100-115 */
double orc_pid_update(orc_pid *p, double desired, double actual, double now);/* Pid.cpp:122-191 */
double orc_pid_derive(orc_pid *p, double value, double now);                 /* Pid.cpp:193-217 */

/* Dense solve by column-pivoted Householder QR: Eigen 3.3's colPivHouseholderQr().solve (Pid.cpp:246) restated from its
 * published algorithm - pivoting on down-dated column norms, the nonzeroPivots() truncation rule, reflectors as
 * makeHouseholder builds them (cdpr_oracle.c has the steps).  Same algorithm, not the same bits: Eigen is not in this
 * image and its vectorised inner products sum in another order.  a is n x n row-major, overwritten. */
void orc_colpiv_qr_solve(int n, double *a, double *b, double *x);
/* End-point derivative weights of the least-squares polynomial through nbuf equally spaced samples (oldest first). */
void orc_fir_weights(unsigned nbuf, unsigned degree, double *w);

/* IK (Joint::Position / GetVelocity restated; geometry statement gen_cdpr.py:113-118).
 * pose = x y z qx qy qz qw, twist = v(3) w(3) world frame.
 * q[n], qdot[n], len[n], jac[n][6] (row i = [u_i, (R b_i) x u_i]); any output may be NULL. */
void orc_ik(const cdpr_config_t *cfg, const double pose[7], const double twist[6], double *q, double *qdot,
            double *len, double *jac);
/* Newton-Raphson FK ([NEW]). Returns iterations performed. */
int orc_fk(const cdpr_config_t *cfg, const double *lengths, const double seed[7], double pose_out[7],
           double *residual);
/* Closed-form tension distribution ([NEW]) for the wrench
 * A*f the raw forces f[n] would apply, A = -J^T at `pose`. Returns 1 if a bound was active. */
int orc_td_forces(const cdpr_config_t *cfg, const double pose[7], const double *f, double *tension);
/* Same for an explicit desired wrench w_d[6]. */
int orc_td_wrench(const cdpr_config_t *cfg, const double pose[7], const double wrench[6], double *tension);

/* ---- batched simulator (the CPU baseline; OpenMP over robots) ---- */
typedef struct orc_sim orc_sim;
orc_sim *orc_create(const cdpr_config_t *cfg, int deriv_mode);
void orc_destroy(orc_sim *s);
void orc_reset(orc_sim *s);
int orc_set_platform_state(orc_sim *s, const double *pose7, const double *twist6);
int orc_set_velocity_command(orc_sim *s, const float *axes, size_t count); /* PLG.cpp:67-74 */
int orc_set_position_command(orc_sim *s, const float *axes, size_t count); /* PLG.cpp:76-83 */
int orc_set_force_command(orc_sim *s, const float *axes, size_t count);    /* JFC.h:92-95 setForce on every joint */
int orc_set_force_command_masked(orc_sim *s, const float *axes, size_t count, const unsigned char *mask);
/* the same callbacks reaching only the robots with mask[b] != 0: B independent plugin instances, some of which got no message */
int orc_set_velocity_command_masked(orc_sim *s, const float *axes, size_t count, const unsigned char *mask);
int orc_set_position_command_masked(orc_sim *s, const float *axes, size_t count, const unsigned char *mask);
int orc_update(orc_sim *s, int nsteps, int nthreads);                      /* PLG.cpp:202-246 + world step */
int orc_rollout_velocity(const orc_sim *s, int samples, int horizon, const float *commands, const double *ref,
                         double *cost, int nthreads);
uint64_t orc_step_count(const orc_sim *s);
void orc_get_joint_states(const orc_sim *s, double *position, double *velocity, double *effort);
void orc_get_platform_state(const orc_sim *s, double *pose7, double *twist6);
void orc_get_raw_state(const orc_sim *s, double *pose7, double *twist6);
void orc_get_pid_debug(const orc_sim *s, double *axes9);
void orc_get_fk_state(const orc_sim *s, double *pose7, double *residual, int32_t *iterations);
void orc_get_td_state(const orc_sim *s, double *tension, int32_t *infeasible);
/* travel limits (cube.sdf:436-437): bit i of cable_mask[b] = joint i outside [travel_lower, travel_upper] at the last published step */
void orc_get_limit_state(const orc_sim *s, uint32_t *cable_mask);
int orc_max_threads(void);

#ifdef __cplusplus
}
#endif
#endif
