/* c_abi_demo.c — the C-ABI of include/cdpr.h from plain C99: what a maintainer's plugin shell does, for the
 * shipped 4-cable robot (cube.yaml:21-29, cube.sdf:310-342,436-442) with the launch-file gains (cdpr_gazebo.launch:20-45)
 * under the sinevelocitytest stimulus (sinevelocitytest.cpp:6-10,34-49).
 *
 *   gcc -std=c99 -O2 -Iinclude examples/c_abi_demo.c -Lcdpr-simulation_amd -lcdpr_hip -lm -Wl,-rpath,$PWD/cdpr-simulation_amd -o c_abi_demo
 *   ./c_abi_demo [robots]
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "cdpr.h"

static void shipped_pid(cdpr_pid_params_t *p, double kp, double ki, double kd) {
  memset(p, 0, sizeof *p);
  p->p_gain = kp, p->i_gain = ki, p->d_gain = kd;
  p->d_degree = 2, p->d_buffer_length = 11; /* differentialFilterWindow / Degree */
  p->i_limit = 100.0, p->cmd_limit = 100.0;
  p->p_filter.rel_cutoff = p->d_filter.rel_cutoff = 0.1;
  p->p_filter.quality = p->d_filter.quality = 0.707; /* cascade 0: bypassed */
}

int main(int argc, char **argv) {
  const unsigned long robots = argc > 1 ? strtoul(argv[1], NULL, 10) : 1ul;
  static const double frame[4][3] = {{-0.3, -0.3, 0.6}, {-0.3, 0.3, 0.6}, {0.3, 0.3, 0.6}, {0.3, -0.3, 0.6}};
  static const double platform[4][3] = {{-0.03, -0.03, 0.0}, {-0.03, 0.03, 0.0}, {0.03, 0.03, 0.0}, {0.03, -0.03, 0.0}};
  cdpr_config_t cfg;
  cdpr_handle_t h = NULL;
  float *pos, *vel, *eff, *pose, *twist, axes[4];
  int i, k, rc;

  if (cdpr_abi_version() != CDPR_ABI_VERSION || cdpr_config_size() != sizeof cfg) {
    fprintf(stderr, "header and library disagree (ABI %u vs %u)\n", (unsigned)cdpr_abi_version(), (unsigned)CDPR_ABI_VERSION);
    return 2;
  }
  memset(&cfg, 0, sizeof cfg);
  cfg.abi_version = CDPR_ABI_VERSION;
  cfg.n_cables = 4;
  cfg.batch = robots;
  cfg.dt = 1e-3;
  for (i = 0; i < 4; ++i) {
    double l2 = 0.0;
    for (k = 0; k < 3; ++k) {
      const double home[3] = {0.0, 0.0, 0.3};
      cfg.frame_anchor[i][k] = frame[i][k];
      cfg.platform_anchor[i][k] = platform[i][k];
      l2 += (home[k] + platform[i][k] - frame[i][k]) * (home[k] + platform[i][k] - frame[i][k]);
    }
    cfg.cable_ref_length[i] = sqrt(l2); /* joint position 0 at the spawn pose: 0.485592... (gen_cdpr.py:113-118) */
  }
  cfg.home_pose[2] = 0.3, cfg.home_pose[6] = 1.0;
  cfg.mass = 1.0;
  cfg.inertia[0] = cfg.inertia[1] = cfg.inertia[2] = 1.0;
  cfg.gravity[2] = -9.8;
  cfg.joint_damping = 1.0, cfg.effort_limit = 100.0;
  shipped_pid(&cfg.velocity_pid, 200.0, 20.0, 1.0);
  shipped_pid(&cfg.position_pid, 200.0, 70.0, 80.0);
  cfg.velocity_epsilon = -0.001; /* hold branch dead, as shipped */
  cfg.publish_period = 0.0;      /* every step */
  cfg.stages = 0;                /* IK + PID + dynamics (FK / TD need six cables) */
  cfg.mapping = CDPR_MAP_AUTO;
  cfg.fk_max_iterations = 4, cfg.fk_lambda = 1e-9;
  cfg.td_f_min = 5.0, cfg.td_f_max = 100.0;

  rc = cdpr_create(&cfg, 0, &h);
  if (rc != CDPR_OK) {
    fprintf(stderr, "cdpr_create: %d (%s)\n", rc, cdpr_last_error(NULL)); /* no GPU: an error, never a CPU fallback */
    return 1;
  }
  pos = malloc(sizeof(float) * robots * 4), vel = malloc(sizeof(float) * robots * 4), eff = malloc(sizeof(float) * robots * 4);
  pose = malloc(sizeof(float) * robots * 7), twist = malloc(sizeof(float) * robots * 6);
  for (k = 0; k < 300; ++k) { /* 3 s: a 100 Hz Joy, ten 1 ms world steps per sample */
    const float v = (float)(0.05 * sin(2.0 * 3.14159265358979323846 * 0.1 * (k * 0.01)));
    for (i = 0; i < 4; ++i) axes[i] = v;
    if (cdpr_set_velocity_command(h, axes, 4) != CDPR_OK) return 1; /* 4 floats: broadcast to every robot */
    if (cdpr_update(h, 10) != CDPR_OK) return 1;
    if (k % 50 == 49) {
      if (cdpr_get_observables(h, pos, vel, eff, pose, twist) != CDPR_OK) return 1;
      printf("t = %5.2f s  platform z = %.6f m  vz = %+.6f m/s  cable0: q = %+.6f m  F = %+.4f N\n", (double)cdpr_step_count(h) * 1e-3,
             pose[2], twist[2], pos[0], eff[0]);
    }
  }
  if (cdpr_set_velocity_command(h, axes, 3) != CDPR_IGNORED) return 1; /* wrong length: dropped like PLG.cpp:68-73 */
  free(pos), free(vel), free(eff), free(pose), free(twist);
  cdpr_destroy(h);
  return 0;
}
