/*
 * cdpr_oracle.c — CPU fp64 restatement of the cdpr_gazebo per-step path.
 * TEST INFRASTRUCTURE ONLY (see cdpr_oracle.h for the rules and the parity status;
 * parts of this file are "parity unpinned" and say so there).
 *
 * Reference paths are relative to /root/reference/src/cdpr_gazebo/:
 *   PLG.cpp = src/CdprGazeboPlugin.cpp, JFC.cpp = src/JointForceCalculator.cpp,
 *   Pid.cpp = src/Pid.cpp, Filter.h = include/cdpr_gazebo/Filter.h,
 *   gen = sdf/gen_cdpr.py.
 */
#include "cdpr_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

/* ------------------------------------------------------------------ */
/* BiQuad low-pass — Filter.h:102-172                                  */
/* ------------------------------------------------------------------ */
void orc_biquad_set_fc(orc_biquad *f, double fc, double fs, double q) {
  /* Filter.h:130-140 */
  double k = tan(M_PI * fc / fs);
  double den = k * k + k / q + 1.0;
  f->a0 = k * k / den;
  f->a1 = 2 * f->a0;
  f->a2 = f->a0;
  f->b1 = 2 * (k * k - 1.0) / den;
  f->b2 = (k * k - k / q + 1.0) / den;
}

void orc_biquad_set_value(orc_biquad *f, double v) {
  /* Filter.h:144-147 (y0 is not state that survives a call) */
  f->x1 = f->x2 = f->y1 = f->y2 = v;
}

double orc_biquad_process(orc_biquad *f, double x) {
  /* Filter.h:152-165, direct form I */
  double y0 = f->a0 * x + f->a1 * f->x1 + f->a2 * f->x2 - f->b1 * f->y1 - f->b2 * f->y2;
  f->x2 = f->x1;
  f->x1 = x;
  f->y2 = f->y1;
  f->y1 = y0;
  return y0;
}

/* Pid::CascadeFilter::update — Pid.cpp:38-44 */
static double cascade_update(orc_biquad *f, unsigned cascade, double x) {
  double out = x;
  for (unsigned i = 0; i < cascade; ++i) out = orc_biquad_process(&f[i], out);
  return out;
}

/* ------------------------------------------------------------------ */
/* Column-pivoted Householder QR solve — Eigen 3.3 ColPivHouseholderQR  */
/* (third-party dependency of the reference: `find_package(Eigen3)`,    */
/* CMakeLists.txt; Ubuntu 18.04 / ROS melodic ship Eigen 3.3.4; not     */
/* vendored, absent from this image; call site Pid.cpp:246              */
/* `A.colPivHouseholderQr().solve(b)`).  Restated step by step from the */
/* published algorithm of Eigen 3.3's                                   */
/* ColPivHouseholderQR::computeInPlace / _solve_impl and                */
/* MatrixBase::makeHouseholder / applyHouseholderOnTheLeft:             */
/*  (1) column norms ||A(:,j)|| computed once ("direct") and kept as    */
/*      "updated" norms; threshold_helper = (eps * max norm)^2 / rows;  */
/*  (2) step k: pivot = column of the largest UPDATED norm among k..;   */
/*      the first k at which that squared norm falls below              */
/*      threshold_helper * (rows - k) fixes nonzero_pivots = k (the     */
/*      decomposition itself continues: "bug 941");                     */
/*  (3) swap columns and both norm tables; Householder vector of        */
/*      A(k.., k): tail == 0 (<= DBL_MIN) -> tau = 0, beta = A(k,k);    */
/*      else beta = -sign(A(k,k)) * ||A(k..,k)||, essential = tail /    */
/*      (A(k,k) - beta), tau = (beta - A(k,k)) / beta; A(k,k) = beta;   */
/*  (4) trailing columns: tmp = essential^T * bottom + row k; row k -=  */
/*      tau * tmp; bottom -= (tau * essential) * tmp;                   */
/*  (5) norm down-date of every trailing column (LAPACK xGEQP3, LAWN    */
/*      176): temp = (1 + |A(k,j)| / upd_j)(1 - |A(k,j)| / upd_j)       */
/*      clamped at 0, temp2 = temp * (upd_j / direct_j)^2; if temp2 <=  */
/*      sqrt(eps) the norm of A(k+1.., j) is recomputed and becomes     */
/*      both direct_j and upd_j, else upd_j *= sqrt(temp);              */
/*  solve: c = b with the first nonzero_pivots reflectors applied,      */
/*      back substitution on the leading nonzero_pivots x nonzero_pivots */
/*      triangle, x(perm(i)) = c(i) for i < nonzero_pivots, 0 beyond.   */
/* Not bit-compatible with a build of Eigen: inner products are summed  */
/* here in index order, Eigen's vectorised reductions pair them         */
/* differently (last-bit differences, amplified only where the          */
/* reference's fit is already ill-conditioned).  PARITY UNPINNED.       */
/* ------------------------------------------------------------------ */
void orc_colpiv_qr_solve(int n, double *a, double *b, double *x) {
  const double eps = 2.220446049250313e-16;   /* NumTraits<double>::epsilon() */
  const double tiny = 2.2250738585072014e-308; /* (std::numeric_limits<double>::min)() */
  int perm[16];
  double c[16], hcoef[16], tmp[16], upd[16], direct[16];
  if (n > 16) n = 16;
  const int rows = n, cols = n, size = n;
  for (int j = 0; j < cols; ++j) perm[j] = j;
  /* (1) */
  double maxnorm = 0.0;
  for (int j = 0; j < cols; ++j) {
    double s = 0.0;
    for (int i = 0; i < rows; ++i) s += a[i * n + j] * a[i * n + j];
    direct[j] = upd[j] = sqrt(s);
    if (upd[j] > maxnorm) maxnorm = upd[j];
  }
  const double threshold_helper = (maxnorm * eps) * (maxnorm * eps) / (double)rows;
  const double norm_downdate_threshold = sqrt(eps);
  int nonzero_pivots = size;
  for (int k = 0; k < size; ++k) {
    /* (2) first maximum wins, as DenseBase::maxCoeff(&index) */
    int piv = k;
    for (int j = k + 1; j < cols; ++j)
      if (upd[j] > upd[piv]) piv = j;
    double biggest_sq = upd[piv] * upd[piv];
    if (nonzero_pivots == size && biggest_sq < threshold_helper * (double)(rows - k)) nonzero_pivots = k;
    /* (3) */
    if (piv != k) {
      for (int i = 0; i < rows; ++i) {
        double t = a[i * n + k];
        a[i * n + k] = a[i * n + piv];
        a[i * n + piv] = t;
      }
      double t = upd[k]; upd[k] = upd[piv]; upd[piv] = t;
      t = direct[k]; direct[k] = direct[piv]; direct[piv] = t;
      int ti = perm[k]; perm[k] = perm[piv]; perm[piv] = ti;
    }
    double c0 = a[k * n + k], tail = 0.0, beta, tau;
    for (int i = k + 1; i < rows; ++i) tail += a[i * n + k] * a[i * n + k];
    if (tail <= tiny) {
      tau = 0.0;
      beta = c0;
      for (int i = k + 1; i < rows; ++i) a[i * n + k] = 0.0;
    } else {
      beta = sqrt(c0 * c0 + tail);
      if (c0 >= 0.0) beta = -beta;
      for (int i = k + 1; i < rows; ++i) a[i * n + k] /= (c0 - beta);
      tau = (beta - c0) / beta;
    }
    hcoef[k] = tau;
    a[k * n + k] = beta;
    /* (4) on the block rows k.., columns k+1..: one row left -> scale it by 1 - tau */
    if (rows - k == 1) {
      for (int j = k + 1; j < cols; ++j) a[k * n + j] *= 1.0 - tau;
    } else if (tau != 0.0) {
      for (int j = k + 1; j < cols; ++j) {
        double s = 0.0;
        for (int i = k + 1; i < rows; ++i) s += a[i * n + k] * a[i * n + j];
        tmp[j] = s + a[k * n + j];
      }
      for (int j = k + 1; j < cols; ++j) a[k * n + j] -= tau * tmp[j];
      for (int j = k + 1; j < cols; ++j)
        for (int i = k + 1; i < rows; ++i) a[i * n + j] -= (tau * a[i * n + k]) * tmp[j];
    }
    /* (5) */
    for (int j = k + 1; j < cols; ++j) {
      if (upd[j] != 0.0) {
        double temp = fabs(a[k * n + j]) / upd[j];
        temp = (1.0 + temp) * (1.0 - temp);
        temp = temp < 0.0 ? 0.0 : temp;
        double r = upd[j] / direct[j];
        double temp2 = temp * (r * r);
        if (temp2 <= norm_downdate_threshold) {
          double s = 0.0;
          for (int i = k + 1; i < rows; ++i) s += a[i * n + j] * a[i * n + j];
          direct[j] = upd[j] = sqrt(s);
        } else {
          upd[j] *= sqrt(temp);
        }
      }
    }
  }
  /* _solve_impl */
  for (int i = 0; i < n; ++i) x[i] = 0.0;
  if (nonzero_pivots == 0) return;
  for (int i = 0; i < rows; ++i) c[i] = b[i];
  for (int k = 0; k < nonzero_pivots; ++k) { /* c = H_{r-1} ... H_0 c, each by applyHouseholderOnTheLeft */
    if (rows - k == 1) {
      c[k] *= 1.0 - hcoef[k];
    } else if (hcoef[k] != 0.0) {
      double s = 0.0;
      for (int i = k + 1; i < rows; ++i) s += a[i * n + k] * c[i];
      s += c[k];
      c[k] -= hcoef[k] * s;
      for (int i = k + 1; i < rows; ++i) c[i] -= (hcoef[k] * a[i * n + k]) * s;
    }
  }
  for (int i = nonzero_pivots - 1; i >= 0; --i) { /* triangularView<Upper>().solveInPlace */
    double s = c[i];
    for (int j = i + 1; j < nonzero_pivots; ++j) s -= a[i * n + j] * c[j];
    c[i] = s / a[i * n + i];
  }
  for (int i = 0; i < nonzero_pivots; ++i) x[perm[i]] = c[i];
}

/* End-point derivative weights of the degree-`degree` least-squares polynomial through `nbuf` equally spaced samples
 * (oldest first; unit spacing): what Pid::derive (Pid.cpp:193-217) computes when every sample of the window is one
 * step apart, as a fixed filter.  Discrete orthogonal polynomials by the three-term recurrence
 * on the centred abscissae in long double; w_j = sum_k p_k(x_j) p_k'(x_last) / <p_k, p_k>.  The oracle's own
 * derivation (the product's is cdpr_derivative_weights in the HIP library; tests compare the two). */
void orc_fir_weights(unsigned nbuf, unsigned degree, double *w) {
  long double xs[CDPR_MAX_D_BUFFER], pm[CDPR_MAX_D_BUFFER], pc[CDPR_MAX_D_BUFFER], pn[CDPR_MAX_D_BUFFER];
  long double dm[CDPR_MAX_D_BUFFER], dc[CDPR_MAX_D_BUFFER], dn[CDPR_MAX_D_BUFFER], acc[CDPR_MAX_D_BUFFER];
  if (nbuf > CDPR_MAX_D_BUFFER) nbuf = CDPR_MAX_D_BUFFER;
  if (degree + 1u > nbuf) degree = nbuf ? nbuf - 1u : 0u;
  for (unsigned j = 0; j < nbuf; ++j) {
    xs[j] = (long double)j - 0.5L * (long double)(nbuf - 1u);
    pm[j] = 0.0L; dm[j] = 0.0L; /* p_{-1} */
    pc[j] = 1.0L; dc[j] = 0.0L; /* p_0 */
    acc[j] = 0.0L;
  }
  long double norm_prev = 1.0L;
  for (unsigned k = 0; k <= degree; ++k) {
    long double nk = 0.0L, xk = 0.0L;
    for (unsigned j = 0; j < nbuf; ++j) { nk += pc[j] * pc[j]; xk += xs[j] * pc[j] * pc[j]; }
    long double dlast = dc[nbuf - 1u];
    for (unsigned j = 0; j < nbuf; ++j) acc[j] += pc[j] * dlast / nk;
    /* p_{k+1} = (x - alpha) p_k - beta p_{k-1} */
    long double alpha = xk / nk, betak = (k == 0) ? 0.0L : nk / norm_prev;
    for (unsigned j = 0; j < nbuf; ++j) {
      pn[j] = (xs[j] - alpha) * pc[j] - betak * pm[j];
      dn[j] = pc[j] + (xs[j] - alpha) * dc[j] - betak * dm[j];
    }
    for (unsigned j = 0; j < nbuf; ++j) { pm[j] = pc[j]; dm[j] = dc[j]; pc[j] = pn[j]; dc[j] = dn[j]; }
    norm_prev = nk;
  }
  for (unsigned j = 0; j < nbuf; ++j) w[j] = (double)acc[j];
}

/* ------------------------------------------------------------------ */
/* Pid — Pid.cpp                                                       */
/* ------------------------------------------------------------------ */
void orc_pid_reset(orc_pid *p) {
  /* Pid.cpp:100-115 */
  p->was_last_time = 0;
  p->perr = p->ierr = p->derr = p->cmd = 0.0;
  for (unsigned i = 0; i < p->prm.p_filter.cascade; ++i) orc_biquad_set_value(&p->pf[i], 0.0); /* Pid.h:48-52 */
  for (unsigned i = 0; i < p->prm.d_filter.cascade; ++i) orc_biquad_set_value(&p->df[i], 0.0);
  for (unsigned i = 0; i < p->prm.d_buffer_length; ++i) p->bx[i] = p->by[i] = 0.0;
  p->missing = p->prm.d_buffer_length;
}

void orc_pid_init(orc_pid *p, const cdpr_pid_params_t *prm, int deriv_mode) {
  /* Pid.cpp:63-77.  The reference writes abs(double) unqualified (Pid.cpp:70-73),
   * which may bind to int abs(int); it is exact for the shipped 100.0 — fabs here. */
  memset(p, 0, sizeof(*p));
  p->prm = *prm;
  if (p->prm.d_buffer_length > CDPR_MAX_D_BUFFER) p->prm.d_buffer_length = CDPR_MAX_D_BUFFER;
  if (p->prm.d_degree > CDPR_MAX_D_DEGREE) p->prm.d_degree = CDPR_MAX_D_DEGREE;
  if (p->prm.p_filter.cascade > CDPR_MAX_CASCADE) p->prm.p_filter.cascade = CDPR_MAX_CASCADE;
  if (p->prm.d_filter.cascade > CDPR_MAX_CASCADE) p->prm.d_filter.cascade = CDPR_MAX_CASCADE;
  p->i_max = fabs(prm->i_limit);
  p->i_min = -fabs(prm->i_limit);
  p->cmd_max = fabs(prm->cmd_limit);
  p->cmd_min = -fabs(prm->cmd_limit);
  p->deriv_mode = deriv_mode;
  if (deriv_mode == ORC_DERIV_FIR) orc_fir_weights(p->prm.d_buffer_length, p->prm.d_degree, p->fir);
  /* CascadeFilter ctor, Pid.cpp:27-36: SetValue(0), SetFc(relCutoff, 1.0, quality) */
  for (unsigned i = 0; i < p->prm.p_filter.cascade; ++i) {
    orc_biquad_set_value(&p->pf[i], 0.0);
    orc_biquad_set_fc(&p->pf[i], prm->p_filter.rel_cutoff, 1.0, prm->p_filter.quality);
  }
  for (unsigned i = 0; i < p->prm.d_filter.cascade; ++i) {
    orc_biquad_set_value(&p->df[i], 0.0);
    orc_biquad_set_fc(&p->df[i], prm->d_filter.rel_cutoff, 1.0, prm->d_filter.quality);
  }
  orc_pid_reset(p);
}

/* Pid::fitPolynomial — Pid.cpp:219-247.  coef[0..d] of the fitted polynomial in
 * the abscissa x[] handed in (absolute time for FAITHFUL, scaled time for EXACT). */
static void fit_polynomial(const double *x, const double *yv, unsigned nbuf, unsigned degree, double *coef) {
  double fx[2 * CDPR_MAX_D_DEGREE + 1];
  double fa[(CDPR_MAX_D_DEGREE + 1) * (CDPR_MAX_D_DEGREE + 1)];
  double fb[CDPR_MAX_D_DEGREE + 1];
  unsigned dp1 = degree + 1u, d2p1 = 2u * degree + 1u;
  for (unsigned i = 0; i < d2p1; ++i) { /* Pid.cpp:224-229 */
    fx[i] = 0.0;
    for (unsigned j = 0; j < nbuf; ++j) fx[i] += pow(x[j], (double)i);
  }
  for (unsigned i = 0; i < dp1; ++i) /* Pid.cpp:231-235 */
    for (unsigned j = 0; j < dp1; ++j) fa[i * dp1 + j] = fx[i + j];
  for (unsigned i = 0; i < dp1; ++i) { /* Pid.cpp:238-244 */
    double tmp = 0.0;
    for (unsigned j = 0; j < nbuf; ++j) tmp += pow(x[j], (double)i) * yv[j];
    fb[i] = tmp;
  }
  orc_colpiv_qr_solve((int)dp1, fa, fb, coef); /* Pid.cpp:246 */
}

double orc_pid_derive(orc_pid *p, double value, double now) {
  /* Pid.cpp:193-217 */
  unsigned nbuf = p->prm.d_buffer_length, degree = p->prm.d_degree;
  for (unsigned i = 1; i < nbuf; ++i) {
    p->bx[i - 1] = p->bx[i];
    p->by[i - 1] = p->by[i];
  }
  p->bx[nbuf - 1] = now;
  p->by[nbuf - 1] = value;
  p->missing -= (p->missing > 0u ? 1u : 0u);

  double derived = 0;
  if (p->missing == 0u) {
    double coef[CDPR_MAX_D_DEGREE + 2];
    int uniform = 0;
    if (p->deriv_mode == ORC_DERIV_FIR && nbuf > 1) { /* every sample one and the same step apart? (stamps are k*dt) */
      double h0 = p->bx[nbuf - 1] - p->bx[nbuf - 2];
      uniform = h0 > 0.0;
      for (unsigned j = 1; uniform && j + 1 < nbuf; ++j) uniform = fabs((p->bx[j] - p->bx[j - 1]) - h0) <= 1e-9 * h0;
      if (uniform) {
        double acc = 0.0;
        for (unsigned j = 0; j < nbuf; ++j) acc += p->fir[j] * p->by[j];
        derived = acc / h0;
      }
    }
    if (uniform) {
      /* done: the fixed filter */
    } else if (p->deriv_mode == ORC_DERIV_FAITHFUL) {
      fit_polynomial(p->bx, p->by, nbuf, degree, coef);
      for (unsigned i = 1; i <= degree; ++i) coef[i - 1] = i * coef[i]; /* Pid.cpp:205-208 */
      coef[degree] = 0.0;
      for (unsigned i = degree; i > 0; --i) derived = now * (derived + coef[i]); /* Pid.cpp:209-212 */
      derived += coef[0];
    } else {
      /* the same least-squares problem, posed in centred time scaled to the mean
       * sample spacing so the normal equations stay well conditioned at any t */
      double xs[CDPR_MAX_D_BUFFER];
      double mean = 0;
      for (unsigned j = 0; j < nbuf; ++j) mean += p->bx[j];
      mean /= (double)nbuf;
      double h = (nbuf > 1) ? (p->bx[nbuf - 1] - p->bx[0]) / (double)(nbuf - 1) : 1.0;
      if (!(h > 0.0)) h = 1.0;
      for (unsigned j = 0; j < nbuf; ++j) xs[j] = (p->bx[j] - mean) / h;
      fit_polynomial(xs, p->by, nbuf, degree, coef);
      double xn = (now - mean) / h;
      for (unsigned i = 1; i <= degree; ++i) coef[i - 1] = i * coef[i];
      coef[degree] = 0.0;
      for (unsigned i = degree; i > 0; --i) derived = xn * (derived + coef[i]);
      derived += coef[0];
      derived /= h;
    }
  }
  return derived;
}

static double clampd(double v, double lo, double hi) {
  /* ignition::math::clamp = max(min(v, hi), lo) */
  double m = v < hi ? v : hi;
  return m > lo ? m : lo;
}

double orc_pid_update(orc_pid *p, double desired, double actual, double now) {
  /* Pid.cpp:122-191 */
  p->dbg_pi_written = p->dbg_d_written = p->dbg_desired_written = 0;
  if (!p->was_last_time) {
    p->was_last_time = 1;
    p->cmd = 0.0;
  } else {
    double f_term = p->prm.forward_gain * desired;
    double error = desired - actual;
    double dt = now - p->last_time;
    p->last_time = now;

    p->perr = cascade_update(p->pf, p->prm.p_filter.cascade, error);
    double p_term = p->prm.p_gain * p->perr;

    double prev_ierr = p->ierr;
    p->ierr += dt * error;
    double i_term = p->prm.i_gain * p->ierr;
    p->dbg_p = p_term; /* Pid.cpp:139-142 */
    p->dbg_i = i_term;
    p->dbg_pi_written = 1;
    if (i_term > p->i_max) {
      i_term = p->i_max;
      p->ierr = i_term / p->prm.i_gain;
    } else if (i_term < p->i_min) {
      i_term = p->i_min;
      p->ierr = i_term / p->prm.i_gain;
    }

    if (dt > 0.0) {
      double derived = orc_pid_derive(p, error, now);
      p->derr = cascade_update(p->df, p->prm.d_filter.cascade, derived);
      p->dbg_desired = desired; /* Pid.cpp:159 */
      p->dbg_desired_written = 1;
    }
    double d_term = p->prm.d_gain * p->derr;
    p->dbg_d = d_term; /* Pid.cpp:166-168 */
    p->dbg_d_written = 1;

    double cmd = f_term + p_term + i_term + d_term;
    if (p->cmd_max > p->cmd_min) p->cmd = clampd(cmd, p->cmd_min, p->cmd_max); /* Pid.cpp:175-177 */
    if (p->cmd != cmd) { /* Pid.cpp:181-184: anti-windup, may leave cmd one increment past the clamp */
      p->ierr = prev_ierr;
      p->cmd += dt * error * p->prm.i_gain;
    }
  }
  p->last_time = now;
  return p->cmd;
}

/* ------------------------------------------------------------------ */
/* JointForceCalculator — JFC.cpp                                      */
/* ------------------------------------------------------------------ */
enum { MODE_FORCE = 0, MODE_POSITION = 1, MODE_VELOCITY = 2 }; /* JFC.h:35-37 */

typedef struct orc_jfc {
  orc_pid pos, vel;
  int mode;
  double eps, last_pos, force, pos_target, vel_target;
  int64_t last_update_ns; /* gazebo::common::Time is integer sec + nsec [EXT] */
  int last_pid;           /* which PID ran in the last update: 0 none, 1 pos, 2 vel */
} orc_jfc;

static void jfc_set_position_target(orc_jfc *j, double target) {
  /* JFC.cpp:99-107 */
  j->pos_target = target;
  if (j->mode != MODE_POSITION) orc_pid_reset(&j->pos);
  j->mode = MODE_POSITION;
}

static void jfc_set_velocity_target(orc_jfc *j, double target) {
  /* JFC.cpp:111-119 */
  j->vel_target = target;
  if (j->mode != MODE_VELOCITY) orc_pid_reset(&j->vel);
  j->mode = MODE_VELOCITY;
}

static void jfc_set_force(orc_jfc *j, double force) {
  /* JFC.h:92-95 */
  j->force = force;
  j->mode = MODE_FORCE;
}

static void jfc_init(orc_jfc *j, const cdpr_config_t *cfg, int deriv_mode, double joint_position) {
  /* PLG.cpp:153-157: construct, setPositionTarget(joint->Position()), then copy-assign
   * into the slot; operator= (JFC.cpp:38-51) copies the mode and calls reset()
   * (JFC.h:69-73), which zeroes force and both targets and resets both PIDs.
   * mLastPosition is not copied and keeps its default 0 (JFC.h:45). */
  orc_pid_init(&j->pos, &cfg->position_pid, deriv_mode);
  orc_pid_init(&j->vel, &cfg->velocity_pid, deriv_mode);
  j->mode = MODE_FORCE; /* JFC.h:42 default */
  j->eps = cfg->velocity_epsilon;
  j->last_pos = 0.0;
  j->force = j->pos_target = j->vel_target = 0.0;
  jfc_set_position_target(j, joint_position);
  j->force = j->pos_target = j->vel_target = 0.0; /* reset() */
  orc_pid_reset(&j->vel);
  orc_pid_reset(&j->pos);
  j->last_update_ns = 0; /* World::SimTime() at Load */
  j->last_pid = 0;
}

static double jfc_update(orc_jfc *j, int64_t now_ns, double now, double q, double qd) {
  /* JFC.cpp:59-96 */
  int64_t step_ns = now_ns - j->last_update_ns;
  j->last_update_ns = now_ns;
  double force = 0.0;
  j->last_pid = 0;
  if (step_ns > 0) {
    if (j->mode == MODE_FORCE) {
      j->last_pos = q;
      force = j->force;
    } else if (j->mode == MODE_VELOCITY) {
      /* JFC.cpp:72 writes abs() unqualified (int-abs hazard); fabs here, the two agree
       * for the shipped eps = -0.001 where the test is always true */
      if (fabs(j->vel_target) > j->eps) {
        j->last_pos = q;
        force = orc_pid_update(&j->vel, j->vel_target, qd, now);
        j->last_pid = 2;
      } else {
        force = orc_pid_update(&j->pos, j->last_pos, q, now);
        j->last_pid = 1;
      }
    } else if (j->mode == MODE_POSITION) {
      j->last_pos = q;
      force = orc_pid_update(&j->pos, j->pos_target, q, now);
      j->last_pid = 1;
    }
  }
  return force;
}

/* ------------------------------------------------------------------ */
/* small linear algebra                                                */
/* ------------------------------------------------------------------ */
static void quat_to_rot(const double q[4], double r[9]) {
  double x = q[0], y = q[1], z = q[2], w = q[3];
  r[0] = 1 - 2 * (y * y + z * z);
  r[1] = 2 * (x * y - z * w);
  r[2] = 2 * (x * z + y * w);
  r[3] = 2 * (x * y + z * w);
  r[4] = 1 - 2 * (x * x + z * z);
  r[5] = 2 * (y * z - x * w);
  r[6] = 2 * (x * z - y * w);
  r[7] = 2 * (y * z + x * w);
  r[8] = 1 - 2 * (x * x + y * y);
}

static void cross3(const double a[3], const double b[3], double c[3]) {
  c[0] = a[1] * b[2] - a[2] * b[1];
  c[1] = a[2] * b[0] - a[0] * b[2];
  c[2] = a[0] * b[1] - a[1] * b[0];
}

/* in-place Cholesky solve of the SPD 6x6 system m x = rhs */
static void chol6_solve(double m[6][6], double rhs[6]) {
  for (int j = 0; j < 6; ++j) {
    double d = m[j][j];
    for (int k = 0; k < j; ++k) d -= m[j][k] * m[j][k];
    d = sqrt(d);
    m[j][j] = d;
    for (int i = j + 1; i < 6; ++i) {
      double s = m[i][j];
      for (int k = 0; k < j; ++k) s -= m[i][k] * m[j][k];
      m[i][j] = s / d;
    }
  }
  for (int i = 0; i < 6; ++i) {
    double s = rhs[i];
    for (int k = 0; k < i; ++k) s -= m[i][k] * rhs[k];
    rhs[i] = s / m[i][i];
  }
  for (int i = 5; i >= 0; --i) {
    double s = rhs[i];
    for (int k = i + 1; k < 6; ++k) s -= m[k][i] * rhs[k];
    rhs[i] = s / m[i][i];
  }
}

/* ------------------------------------------------------------------ */
/* IK — Joint::Position / GetVelocity restated;     */
/* geometry statement gen:113-118: pp = pf_t + pf_R b, u = (pp-fp)/|..| */
/* ------------------------------------------------------------------ */
void orc_ik(const cdpr_config_t *cfg, const double pose[7], const double twist[6], double *q, double *qdot,
            double *len, double *jac) {
  double r[9];
  quat_to_rot(pose + 3, r);
  for (unsigned i = 0; i < cfg->n_cables; ++i) {
    const double *b = cfg->platform_anchor[i], *a = cfg->frame_anchor[i];
    double rb[3] = {r[0] * b[0] + r[1] * b[1] + r[2] * b[2], r[3] * b[0] + r[4] * b[1] + r[5] * b[2],
                    r[6] * b[0] + r[7] * b[1] + r[8] * b[2]};
    double l[3] = {pose[0] + rb[0] - a[0], pose[1] + rb[1] - a[1], pose[2] + rb[2] - a[2]};
    double L = sqrt(l[0] * l[0] + l[1] * l[1] + l[2] * l[2]);
    double u[3] = {l[0] / L, l[1] / L, l[2] / L};
    double rbxu[3];
    cross3(rb, u, rbxu);
    double row[6] = {u[0], u[1], u[2], rbxu[0], rbxu[1], rbxu[2]};
    if (len) len[i] = L;
    if (q) q[i] = cfg->cable_ref_length[i] - L; /* prismatic axis = -u (gen:181) */
    if (qdot && twist) {
      double s = 0;
      for (int k = 0; k < 6; ++k) s += row[k] * twist[k];
      qdot[i] = -s;
    }
    if (jac)
      for (int k = 0; k < 6; ++k) jac[i * 6 + k] = row[k];
  }
}

/* world-frame rotation increment: q <- exp(theta/2) (x) q */
static void quat_apply_rotvec(double q[4], const double th[3]) {
  double a2 = th[0] * th[0] + th[1] * th[1] + th[2] * th[2];
  double a = sqrt(a2);
  double k = (a < 1e-8) ? 0.5 - a2 / 48.0 : sin(0.5 * a) / a;
  double c = cos(0.5 * a);
  double d[4] = {k * th[0], k * th[1], k * th[2], c};
  double x = q[0], y = q[1], z = q[2], w = q[3];
  double n[4];
  n[3] = d[3] * w - d[0] * x - d[1] * y - d[2] * z;
  n[0] = d[3] * x + w * d[0] + d[1] * z - d[2] * y;
  n[1] = d[3] * y + w * d[1] + d[2] * x - d[0] * z;
  n[2] = d[3] * z + w * d[2] + d[0] * y - d[1] * x;
  double nn = sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2] + n[3] * n[3]);
  for (int i = 0; i < 4; ++i) q[i] = n[i] / nn;
}

/* J^T J (+ lambda I) and J^T r */
static void normal_eq(unsigned n, const double *jac, const double *r, double lambda, double m[6][6], double g[6]) {
  for (int a = 0; a < 6; ++a) {
    for (int b = 0; b < 6; ++b) {
      double s = 0;
      for (unsigned i = 0; i < n; ++i) s += jac[i * 6 + a] * jac[i * 6 + b];
      m[a][b] = s + (a == b ? lambda : 0.0);
    }
    double s = 0;
    for (unsigned i = 0; i < n; ++i) s += jac[i * 6 + a] * r[i];
    g[a] = s;
  }
}

/* Newton-Raphson FK — [NEW], SURVEY 8(a) row 14 */
int orc_fk(const cdpr_config_t *cfg, const double *lengths, const double seed[7], double pose_out[7],
           double *residual) {
  unsigned n = cfg->n_cables;
  double pose[7], len[CDPR_MAX_CABLES], jac[CDPR_MAX_CABLES * 6], r[CDPR_MAX_CABLES];
  memcpy(pose, seed, sizeof(pose));
  int it = 0;
  for (; it < (int)cfg->fk_max_iterations; ++it) {
    orc_ik(cfg, pose, NULL, NULL, NULL, len, jac);
    double rmax = 0;
    for (unsigned i = 0; i < n; ++i) {
      r[i] = lengths[i] - len[i];
      if (fabs(r[i]) > rmax) rmax = fabs(r[i]);
    }
    if (rmax < cfg->fk_tolerance) break;
    double m[6][6], g[6];
    normal_eq(n, jac, r, cfg->fk_lambda, m, g);
    chol6_solve(m, g);
    pose[0] += g[0];
    pose[1] += g[1];
    pose[2] += g[2];
    quat_apply_rotvec(pose + 3, g + 3);
  }
  orc_ik(cfg, pose, NULL, NULL, NULL, len, NULL);
  double rmax = 0;
  for (unsigned i = 0; i < n; ++i)
    if (fabs(lengths[i] - len[i]) > rmax) rmax = fabs(lengths[i] - len[i]);
  if (residual) *residual = rmax;
  memcpy(pose_out, pose, sizeof(pose));
  return it;
}

/* Tension distribution — [NEW], SURVEY 8(a) row 15.
 * A = -J^T, T = Tm 1 + A^+ (w_d - A Tm 1) with A^+ = A^T (A A^T)^-1; A A^T = J^T J. */
static int td_core(const cdpr_config_t *cfg, const double *jac, const double wd[6], double *tension) {
  unsigned n = cfg->n_cables;
  double tm = 0.5 * (cfg->td_f_min + cfg->td_f_max);
  double m[6][6], g[6], ones[CDPR_MAX_CABLES];
  for (unsigned i = 0; i < n; ++i) ones[i] = tm;
  normal_eq(n, jac, ones, 0.0, m, g); /* g = J^T (Tm 1) = -(A Tm 1) */
  double rhs[6];
  for (int a = 0; a < 6; ++a) rhs[a] = wd[a] + g[a]; /* w_d - A Tm 1 */
  chol6_solve(m, rhs);
  int flag = 0;
  for (unsigned i = 0; i < n; ++i) {
    double s = 0;
    for (int a = 0; a < 6; ++a) s += jac[i * 6 + a] * rhs[a];
    double t = tm - s; /* A^T = -J */
    if (t < cfg->td_f_min) {
      t = cfg->td_f_min;
      flag = 1;
    } else if (t > cfg->td_f_max) {
      t = cfg->td_f_max;
      flag = 1;
    }
    tension[i] = t;
  }
  return flag;
}

int orc_td_wrench(const cdpr_config_t *cfg, const double pose[7], const double wrench[6], double *tension) {
  double jac[CDPR_MAX_CABLES * 6];
  orc_ik(cfg, pose, NULL, NULL, NULL, NULL, jac);
  return td_core(cfg, jac, wrench, tension);
}

int orc_td_forces(const cdpr_config_t *cfg, const double pose[7], const double *f, double *tension) {
  double jac[CDPR_MAX_CABLES * 6], wd[6];
  orc_ik(cfg, pose, NULL, NULL, NULL, NULL, jac);
  for (int a = 0; a < 6; ++a) {
    double s = 0;
    for (unsigned i = 0; i < cfg->n_cables; ++i) s += jac[i * 6 + a] * f[i];
    wd[a] = -s; /* A f */
  }
  return td_core(cfg, jac, wd, tension);
}

/* ------------------------------------------------------------------ */
/* batched simulator                                                   */
/* ------------------------------------------------------------------ */
typedef struct orc_robot {
  double pose[7], twist[6];
  orc_jfc jfc[CDPR_MAX_CABLES];
  double fk_pose[7], fk_res;
  int32_t fk_iters, td_flag;
  double td_tension[CDPR_MAX_CABLES];
  /* published observables */
  double pub_q[CDPR_MAX_CABLES], pub_qd[CDPR_MAX_CABLES], pub_eff[CDPR_MAX_CABLES];
  double pub_pose[7], pub_twist[6];
  double pub_fk_res;            /* estimator residual / iteration count and the infeasibility flag travel with the */
  int32_t pub_fk_iters, pub_td_flag; /* observables (they share a row with them on the device): as of the last PUBLISHED step */
  uint32_t limit_mask, pub_limit_mask; /* bit i: joint i outside its travel range (cube.sdf:436-437) at this / the last published step */
  double dbg[CDPR_PID_DEBUG_AXES];
} orc_robot;

struct orc_sim {
  cdpr_config_t cfg;
  int deriv_mode;
  uint64_t step;
  double prev_publish;
  orc_robot *rob;
  float *vel_cmd, *pos_cmd; /* latched Joy.axes, float32 on the wire */
  float *frc_cmd;           /* force command (JointForceCalculator::setForce, JFC.h:92-95): no topic of the plugin reaches it */
  unsigned char *vel_received, *pos_received; /* per robot: a Joy arrived since the last update() (PLG.cpp:206,213) */
  unsigned char *frc_received;
  double ib[9], ib_inv[9]; /* body inertia and its inverse */
};

static void mat3_inv(const double m[9], double inv[9]) {
  double c0 = m[4] * m[8] - m[5] * m[7], c1 = m[5] * m[6] - m[3] * m[8], c2 = m[3] * m[7] - m[4] * m[6];
  double det = m[0] * c0 + m[1] * c1 + m[2] * c2;
  inv[0] = c0 / det;
  inv[1] = (m[2] * m[7] - m[1] * m[8]) / det;
  inv[2] = (m[1] * m[5] - m[2] * m[4]) / det;
  inv[3] = c1 / det;
  inv[4] = (m[0] * m[8] - m[2] * m[6]) / det;
  inv[5] = (m[2] * m[3] - m[0] * m[5]) / det;
  inv[6] = c2 / det;
  inv[7] = (m[1] * m[6] - m[0] * m[7]) / det;
  inv[8] = (m[0] * m[4] - m[1] * m[3]) / det;
}

static void robot_reset(const orc_sim *s, orc_robot *r) {
  memset(r, 0, sizeof(*r));
  memcpy(r->pose, s->cfg.home_pose, sizeof(r->pose));
  memcpy(r->fk_pose, s->cfg.home_pose, sizeof(r->fk_pose));
  memcpy(r->pub_pose, s->cfg.home_pose, sizeof(r->pub_pose));
  double q[CDPR_MAX_CABLES];
  orc_ik(&s->cfg, r->pose, NULL, q, NULL, NULL, NULL);
  for (unsigned i = 0; i < s->cfg.n_cables; ++i) jfc_init(&r->jfc[i], &s->cfg, s->deriv_mode, q[i]);
}

orc_sim *orc_create(const cdpr_config_t *cfg, int deriv_mode) {
  if (!cfg || cfg->n_cables < 1 || cfg->n_cables > CDPR_MAX_CABLES || cfg->batch < 1) return NULL;
  orc_sim *s = (orc_sim *)calloc(1, sizeof(*s));
  if (!s) return NULL;
  s->cfg = *cfg;
  s->deriv_mode = deriv_mode;
  s->rob = (orc_robot *)malloc(sizeof(orc_robot) * cfg->batch);
  s->vel_cmd = (float *)calloc(cfg->batch * cfg->n_cables, sizeof(float));
  s->pos_cmd = (float *)calloc(cfg->batch * cfg->n_cables, sizeof(float));
  s->frc_cmd = (float *)calloc(cfg->batch * cfg->n_cables, sizeof(float));
  s->vel_received = (unsigned char *)calloc(cfg->batch, 1);
  s->pos_received = (unsigned char *)calloc(cfg->batch, 1);
  s->frc_received = (unsigned char *)calloc(cfg->batch, 1);
  if (!s->rob || !s->vel_cmd || !s->pos_cmd || !s->frc_cmd || !s->vel_received || !s->pos_received || !s->frc_received) {
    orc_destroy(s);
    return NULL;
  }
  const double *in = cfg->inertia;
  double ib[9] = {in[0], in[3], in[4], in[3], in[1], in[5], in[4], in[5], in[2]};
  memcpy(s->ib, ib, sizeof(ib));
  mat3_inv(ib, s->ib_inv);
  orc_reset(s);
  return s;
}

void orc_destroy(orc_sim *s) {
  if (!s) return;
  free(s->rob);
  free(s->vel_cmd);
  free(s->pos_cmd);
  free(s->frc_cmd);
  free(s->vel_received);
  free(s->pos_received);
  free(s->frc_received);
  free(s);
}

void orc_reset(orc_sim *s) {
  s->step = 0;
  s->prev_publish = 0.0; /* PLG.cpp:59 */
  memset(s->vel_received, 0, s->cfg.batch);
  memset(s->pos_received, 0, s->cfg.batch);
  memset(s->frc_received, 0, s->cfg.batch);
  for (uint64_t b = 0; b < s->cfg.batch; ++b) robot_reset(s, &s->rob[b]);
}

int orc_set_platform_state(orc_sim *s, const double *pose7, const double *twist6) {
  for (uint64_t b = 0; b < s->cfg.batch; ++b) {
    if (pose7) {
      memcpy(s->rob[b].pose, pose7 + 7 * b, 7 * sizeof(double));
      memcpy(s->rob[b].fk_pose, pose7 + 7 * b, 7 * sizeof(double));
    }
    if (twist6) memcpy(s->rob[b].twist, twist6 + 6 * b, 6 * sizeof(double));
  }
  return CDPR_OK;
}

static int latch(orc_sim *s, float *dst, const float *axes, size_t count, unsigned char *flag, const unsigned char *mask) {
  /* PLG.cpp:67-83: accept iff axes.size() == cWireCount, else silently ignore.  mask (per-robot arrival, one plugin
   * instance per robot): only the robots with mask[b] != 0 receive the message; NULL = every robot */
  size_t n = s->cfg.n_cables, B = s->cfg.batch;
  int per_robot = (count == n * B && B != 1);
  if (!per_robot && count != n) return CDPR_IGNORED;
  for (size_t b = 0; b < B; ++b) {
    if (mask && !mask[b]) continue;
    memcpy(dst + b * n, per_robot ? axes + b * n : axes, sizeof(float) * n);
    flag[b] = 1;
  }
  return CDPR_OK;
}

int orc_set_velocity_command(orc_sim *s, const float *axes, size_t count) {
  return latch(s, s->vel_cmd, axes, count, s->vel_received, NULL);
}
int orc_set_position_command(orc_sim *s, const float *axes, size_t count) {
  return latch(s, s->pos_cmd, axes, count, s->pos_received, NULL);
}
/* JointForceCalculator::setForce for every joint (JFC.h:92-95); same length rule as the Joy callbacks */
int orc_set_force_command(orc_sim *s, const float *axes, size_t count) {
  return latch(s, s->frc_cmd, axes, count, s->frc_received, NULL);
}
int orc_set_force_command_masked(orc_sim *s, const float *axes, size_t count, const unsigned char *mask) {
  return latch(s, s->frc_cmd, axes, count, s->frc_received, mask);
}
int orc_set_velocity_command_masked(orc_sim *s, const float *axes, size_t count, const unsigned char *mask) {
  return latch(s, s->vel_cmd, axes, count, s->vel_received, mask);
}
int orc_set_position_command_masked(orc_sim *s, const float *axes, size_t count, const unsigned char *mask) {
  return latch(s, s->pos_cmd, axes, count, s->pos_received, mask);
}

/* One world iteration for one robot at step index k:
 * CdprGazeboPlugin::update (PLG.cpp:202-246) on the state at t_k, then the
 * world step to t_{k+1} (Gazebo/ODE restated). */
static void robot_step(const orc_sim *s, orc_robot *r, uint64_t k, int publish) {
  const cdpr_config_t *cfg = &s->cfg;
  unsigned n = cfg->n_cables;
  /* gazebo::common::Time: integer sec/nsec; Double() = sec + nsec*1e-9 [EXT] */
  int64_t dt_ns = (int64_t)llround(cfg->dt * 1e9);
  int64_t now_ns = (int64_t)k * dt_ns;
  double now = (double)(now_ns / 1000000000LL) + (double)(now_ns % 1000000000LL) * 1e-9;

  double q[CDPR_MAX_CABLES], qd[CDPR_MAX_CABLES], jac[CDPR_MAX_CABLES * 6], f[CDPR_MAX_CABLES];
  orc_ik(cfg, r->pose, r->twist, q, qd, NULL, jac);

  /* PLG.cpp:222-228: per-cable force from the state at t_k */
  for (unsigned i = 0; i < n; ++i) f[i] = jfc_update(&r->jfc[i], now_ns, now, q[i], qd[i]);

  /* travel limits of the prismatic joints (cube.sdf:436-437; [EXT] Gazebo/ODE joint stops): which joints are outside */
  const int travel_on = cfg->travel_lower != 0.0 || cfg->travel_upper != 0.0;
  r->limit_mask = 0;
  if (travel_on)
    for (unsigned i = 0; i < n; ++i)
      if (q[i] < cfg->travel_lower || q[i] > cfg->travel_upper) r->limit_mask |= 1u << i;

  /* `pid` debug topic: the global pidMsg keeps stale entries when a branch does
   * not write them (Pid.cpp:139-142,158-168) */
  {
    const orc_jfc *j0 = &r->jfc[0];
    const orc_pid *p = j0->last_pid == 2 ? &j0->vel : (j0->last_pid == 1 ? &j0->pos : NULL);
    if (p) {
      if (p->dbg_pi_written) {
        r->dbg[0] = (float)p->dbg_p;
        r->dbg[1] = (float)p->dbg_i;
      }
      if (p->dbg_d_written) r->dbg[2] = (float)p->dbg_d;
      if (p->dbg_desired_written) r->dbg[3] = (float)p->dbg_desired;
    }
  }

  const double *jtd = jac;
  double jac_est[CDPR_MAX_CABLES * 6];
  if (cfg->stages & CDPR_STAGE_FK) {
    double lstar[CDPR_MAX_CABLES];
    for (unsigned i = 0; i < n; ++i) lstar[i] = cfg->cable_ref_length[i] - q[i]; /* encoder lengths */
    double est[7];
    r->fk_iters = orc_fk(cfg, lstar, r->fk_pose, est, &r->fk_res);
    memcpy(r->fk_pose, est, sizeof(est));
    if (cfg->stages & CDPR_STAGE_TD) {
      orc_ik(cfg, est, NULL, NULL, NULL, NULL, jac_est);
      jtd = jac_est;
    }
  }
  double applied[CDPR_MAX_CABLES] = {0};
  if (cfg->stages & CDPR_STAGE_TD) {
    double wd[6];
    for (int a = 0; a < 6; ++a) {
      double sum = 0;
      for (unsigned i = 0; i < n; ++i) sum += jtd[i * 6 + a] * f[i];
      wd[a] = -sum;
    }
    r->td_flag = td_core(cfg, jtd, wd, r->td_tension);
    for (unsigned i = 0; i < n; ++i) applied[i] = r->td_tension[i];
  } else {
    for (unsigned i = 0; i < n; ++i) applied[i] = f[i];
  }
  /* Joint::SetForce -> CheckAndTruncateForce [EXT]: a force that drives a joint already beyond its velocity limit
   * further out is dropped, then the effort limit clamps (cube.sdf:438-439) */
  if (cfg->velocity_limit > 0.0)
    for (unsigned i = 0; i < n; ++i) {
      if (qd[i] > cfg->velocity_limit)
        applied[i] = applied[i] > 0 ? 0.0 : applied[i];
      else if (qd[i] < -cfg->velocity_limit)
        applied[i] = applied[i] < 0 ? 0.0 : applied[i];
    }
  if (cfg->effort_limit >= 0.0)
    for (unsigned i = 0; i < n; ++i) applied[i] = clampd(applied[i], -cfg->effort_limit, cfg->effort_limit);
  r->dbg[4] = (float)applied[0]; /* PLG.cpp:226 */

  if (publish) { /* PLG.cpp:248-280; the frame link is static at the origin in the reduced model */
    for (unsigned i = 0; i < n; ++i) {
      r->pub_q[i] = q[i];
      r->pub_qd[i] = qd[i];
      r->pub_eff[i] = applied[i];
    }
    memcpy(r->pub_pose, r->pose, sizeof(r->pub_pose));
    memcpy(r->pub_twist, r->twist, sizeof(r->pub_twist));
    r->pub_fk_res = r->fk_res;
    r->pub_fk_iters = r->fk_iters;
    r->pub_td_flag = r->td_flag;
    r->pub_limit_mask = r->limit_mask;
  }

  /* world step: explicit joint damping, wrench = -J^T T + m g, semi-implicit Euler */
  double w[6] = {cfg->mass * cfg->gravity[0], cfg->mass * cfg->gravity[1], cfg->mass * cfg->gravity[2], 0, 0, 0};
  for (unsigned i = 0; i < n; ++i) {
    double t = applied[i] - cfg->joint_damping * qd[i];
    if (cfg->unilateral_cables && t < 0.0) t = 0.0; /* [NEW] option: a cable cannot push */
    for (int a = 0; a < 6; ++a) w[a] -= jac[i * 6 + a] * t;
  }
  double rm[9];
  quat_to_rot(r->pose + 3, rm);
  double *v = r->twist, *om = r->twist + 3;
  const int lumped = cfg->passive_damping != 0.0 || cfg->leg_inertia != 0.0 || cfg->cable_axial_mass != 0.0 ||
                     cfg->anchor_point_mass != 0.0 || cfg->anchor_inertia != 0.0;
  if (!lumped) {
    for (int a = 0; a < 3; ++a) v[a] += cfg->dt * w[a] / cfg->mass;
    double tb[3], ob[3], iob[3], gyro[3], ab[3], aw[3];
    for (int a = 0; a < 3; ++a) {
      tb[a] = rm[0 + a] * w[3] + rm[3 + a] * w[4] + rm[6 + a] * w[5]; /* R^T tau */
      ob[a] = rm[0 + a] * om[0] + rm[3 + a] * om[1] + rm[6 + a] * om[2];
    }
    for (int a = 0; a < 3; ++a) iob[a] = s->ib[3 * a] * ob[0] + s->ib[3 * a + 1] * ob[1] + s->ib[3 * a + 2] * ob[2];
    cross3(ob, iob, gyro);
    for (int a = 0; a < 3; ++a) tb[a] -= gyro[a];
    for (int a = 0; a < 3; ++a) ab[a] = s->ib_inv[3 * a] * tb[0] + s->ib_inv[3 * a + 1] * tb[1] + s->ib_inv[3 * a + 2] * tb[2];
    for (int a = 0; a < 3; ++a) aw[a] = rm[3 * a] * ab[0] + rm[3 * a + 1] * ab[1] + rm[3 * a + 2] * ab[2];
    for (int a = 0; a < 3; ++a) om[a] += cfg->dt * aw[a];
  } else {
    /* Lumped legs ([EXT] -> reduced, HISTORY.md section 1).  Leg i turns about its frame anchor with angular velocity
     * (u x vP)/L, vP = v + omega x rb the velocity of its platform anchor.  Massless-leg torque balance gives the force
     * the passive joint dampers put on the platform at the anchor, Fd = -(c/L) (2 vt/L - omega x u), vt the part of vP
     * across the cable, plus the spherical joint's torque c ((u x vP)/L - omega).  The links' inertia appears at the
     * anchor as the apparent mass A = alpha I + beta u u^T, alpha = J_leg/L^2 + m_pt, beta = m_ax - J_leg/L^2, so the
     * platform's 6x6 mass matrix becomes M0 + sum_i G_i^T A_i G_i with G_i = [I, -[rb_i]x]; velocity-product terms of
     * the legs are neglected (first order). */
    double M[6][6] = {{0}};
    double iw[9]; /* world inertia R Ib R^T, plus the per-leg share that turns with the platform */
    for (int a = 0; a < 3; ++a)
      for (int b = 0; b < 3; ++b) {
        double acc = 0;
        for (int c = 0; c < 3; ++c)
          for (int d = 0; d < 3; ++d) acc += rm[3 * a + c] * s->ib[3 * c + d] * rm[3 * b + d];
        iw[3 * a + b] = acc + (a == b ? (double)n * cfg->anchor_inertia : 0.0);
      }
    for (int a = 0; a < 3; ++a) {
      M[a][a] = cfg->mass;
      for (int b = 0; b < 3; ++b) M[3 + a][3 + b] = iw[3 * a + b];
    }
    double iom[3] = {iw[0] * om[0] + iw[1] * om[1] + iw[2] * om[2], iw[3] * om[0] + iw[4] * om[1] + iw[5] * om[2],
                     iw[6] * om[0] + iw[7] * om[1] + iw[8] * om[2]};
    double gyro[3];
    cross3(om, iom, gyro);
    for (int a = 0; a < 3; ++a) w[3 + a] -= gyro[a];
    for (unsigned i = 0; i < n; ++i) {
      const double *u = jac + i * 6;
      double rb[3], b3[3] = {cfg->platform_anchor[i][0], cfg->platform_anchor[i][1], cfg->platform_anchor[i][2]};
      for (int a = 0; a < 3; ++a) rb[a] = rm[3 * a] * b3[0] + rm[3 * a + 1] * b3[1] + rm[3 * a + 2] * b3[2];
      const double L = cfg->cable_ref_length[i] - q[i];
      double orb[3], vp[3], vt[3], ou[3], uvp[3], fd[3], td_[3], rbf[3];
      cross3(om, rb, orb);
      for (int a = 0; a < 3; ++a) vp[a] = v[a] + orb[a];
      const double along = u[0] * vp[0] + u[1] * vp[1] + u[2] * vp[2];
      for (int a = 0; a < 3; ++a) vt[a] = vp[a] - u[a] * along;
      cross3(om, u, ou);
      cross3(u, vp, uvp);
      const double c = cfg->passive_damping;
      for (int a = 0; a < 3; ++a) fd[a] = -(c / L) * (2.0 * vt[a] / L - ou[a]);
      cross3(rb, fd, rbf);
      for (int a = 0; a < 3; ++a) td_[a] = rbf[a] + c * (uvp[a] / L - om[a]);
      /* weight of the point masses at the anchor */
      double fg[3] = {cfg->anchor_point_mass * cfg->gravity[0], cfg->anchor_point_mass * cfg->gravity[1], cfg->anchor_point_mass * cfg->gravity[2]};
      double rbg[3];
      cross3(rb, fg, rbg);
      for (int a = 0; a < 3; ++a) {
        w[a] += fd[a] + fg[a];
        w[3 + a] += td_[a] + rbg[a];
      }
      const double mu = cfg->leg_inertia / (L * L);
      const double alpha = mu + cfg->anchor_point_mass, beta = cfg->cable_axial_mass - mu;
      const double rb2 = rb[0] * rb[0] + rb[1] * rb[1] + rb[2] * rb[2];
      const double X[9] = {0, -rb[2], rb[1], rb[2], 0, -rb[0], -rb[1], rb[0], 0}; /* [rb]x */
      for (int a = 0; a < 3; ++a) {
        M[a][a] += alpha;
        for (int b = 0; b < 3; ++b) {
          M[a][3 + b] -= alpha * X[3 * a + b];
          M[3 + a][b] += alpha * X[3 * a + b];
          M[3 + a][3 + b] += alpha * ((a == b ? rb2 : 0.0) - rb[a] * rb[b]);
        }
      }
      for (int a = 0; a < 6; ++a)
        for (int b = 0; b < 6; ++b) M[a][b] += beta * u[a] * u[b]; /* u[0..5] is the structure-matrix row [u, rb x u] = G^T u */
    }
    /* solve M acc = w (SPD): Cholesky */
    double acc6[6];
    for (int j = 0; j < 6; ++j) {
      for (int k = 0; k < j; ++k) M[j][j] -= M[j][k] * M[j][k];
      M[j][j] = sqrt(M[j][j]);
      for (int i2 = j + 1; i2 < 6; ++i2) {
        for (int k = 0; k < j; ++k) M[i2][j] -= M[i2][k] * M[j][k];
        M[i2][j] /= M[j][j];
      }
    }
    for (int i2 = 0; i2 < 6; ++i2) {
      double sum = w[i2];
      for (int k = 0; k < i2; ++k) sum -= M[i2][k] * acc6[k];
      acc6[i2] = sum / M[i2][i2];
    }
    for (int i2 = 5; i2 >= 0; --i2) {
      double sum = acc6[i2];
      for (int k = i2 + 1; k < 6; ++k) sum -= M[k][i2] * acc6[k];
      acc6[i2] = sum / M[i2][i2];
    }
    for (int a = 0; a < 3; ++a) {
      v[a] += cfg->dt * acc6[a];
      om[a] += cfg->dt * acc6[3 + a];
    }
  }
  /* [EXT] -> reduced: the joint stop, inelastic.  A joint at or beyond a travel limit that still moves outward after the
   * velocity update takes the impulse that brings its rate to zero: qdot_i = -J_i twist; lambda = qdot_i / (J_i M^-1 J_i^T);
   * twist += M^-1 J_i^T lambda (then qdot_i = 0), M = the platform's own mass and inertia (massless-cable reduction); cables in
   * index order (a later cable sees the twist the earlier ones left), cfg->travel_stop sweeps over the cables: projected
   * Gauss-Seidel on the velocity constraints, what ODE's quickstep iterates 50 times [EXT]. */
  for (unsigned sweep = 0; travel_on && sweep < cfg->travel_stop; ++sweep) {
    for (unsigned i = 0; i < n; ++i) {
      const double *ji = jac + i * 6;
      const double qdn = -(ji[0] * v[0] + ji[1] * v[1] + ji[2] * v[2] + ji[3] * om[0] + ji[4] * om[1] + ji[5] * om[2]);
      const int out_hi = q[i] >= cfg->travel_upper && qdn > 0.0, out_lo = q[i] <= cfg->travel_lower && qdn < 0.0;
      if (!out_hi && !out_lo) continue;
      double tb[3], ab[3], aw[3];
      for (int a = 0; a < 3; ++a) tb[a] = rm[0 + a] * ji[3] + rm[3 + a] * ji[4] + rm[6 + a] * ji[5]; /* R^T (rb x u) */
      for (int a = 0; a < 3; ++a) ab[a] = s->ib_inv[3 * a] * tb[0] + s->ib_inv[3 * a + 1] * tb[1] + s->ib_inv[3 * a + 2] * tb[2];
      for (int a = 0; a < 3; ++a) aw[a] = rm[3 * a] * ab[0] + rm[3 * a + 1] * ab[1] + rm[3 * a + 2] * ab[2]; /* Iw^-1 (rb x u) */
      const double d = (ji[0] * ji[0] + ji[1] * ji[1] + ji[2] * ji[2]) / cfg->mass + ji[3] * aw[0] + ji[4] * aw[1] + ji[5] * aw[2];
      const double lam = qdn / d;
      for (int a = 0; a < 3; ++a) {
        v[a] += lam * ji[a] / cfg->mass;
        om[a] += lam * aw[a];
      }
    }
  }
  for (int a = 0; a < 3; ++a) r->pose[a] += cfg->dt * v[a];
  {
    double *qq = r->pose + 3;
    double x = qq[0], y = qq[1], z = qq[2], ww = qq[3];
    double h = 0.5 * cfg->dt;
    double nq[4];
    nq[0] = x + h * (ww * om[0] + om[1] * z - om[2] * y);
    nq[1] = y + h * (ww * om[1] + om[2] * x - om[0] * z);
    nq[2] = z + h * (ww * om[2] + om[0] * y - om[1] * x);
    nq[3] = ww - h * (om[0] * x + om[1] * y + om[2] * z);
    double nn = sqrt(nq[0] * nq[0] + nq[1] * nq[1] + nq[2] * nq[2] + nq[3] * nq[3]);
    for (int a = 0; a < 4; ++a) qq[a] = nq[a] / nn;
  }
}

int orc_update(orc_sim *s, int nsteps, int nthreads) {
  if (nsteps <= 0) return CDPR_OK;
  const unsigned n = s->cfg.n_cables;
  /* publish schedule (PLG.cpp:236-242): strict '>' against the last published stamp */
  unsigned char *pub = (unsigned char *)malloc((size_t)nsteps);
  if (!pub) return CDPR_ERR_NOMEM;
  int64_t dt_ns = (int64_t)llround(s->cfg.dt * 1e9);
  for (int k = 0; k < nsteps; ++k) {
    int64_t now_ns = (int64_t)(s->step + (uint64_t)k) * dt_ns;
    double now = (double)(now_ns / 1000000000LL) + (double)(now_ns % 1000000000LL) * 1e-9;
    if ((now - s->prev_publish) > s->cfg.publish_period) {
      s->prev_publish = now;
      pub[k] = 1;
    } else {
      pub[k] = 0;
    }
  }
  const uint64_t step0 = s->step;
  const int64_t B = (int64_t)s->cfg.batch;
#ifdef _OPENMP
  if (nthreads <= 0) nthreads = omp_get_max_threads();
#pragma omp parallel for schedule(static) num_threads(nthreads)
#endif
  for (int64_t b = 0; b < B; ++b) {
    orc_robot *r = &s->rob[b];
    /* PLG.cpp:206-219: velocity command first, then position command */
    if (s->vel_received[b])
      for (unsigned i = 0; i < n; ++i) jfc_set_velocity_target(&r->jfc[i], (double)s->vel_cmd[b * n + i]);
    if (s->pos_received[b])
      for (unsigned i = 0; i < n; ++i) jfc_set_position_target(&r->jfc[i], (double)s->pos_cmd[b * n + i]);
    /* [NEW] ordering: a force command (no callback of the reference reaches setForce) is applied after the two Joy topics */
    if (s->frc_received[b])
      for (unsigned i = 0; i < n; ++i) jfc_set_force(&r->jfc[i], (double)s->frc_cmd[b * n + i]);
    s->vel_received[b] = s->pos_received[b] = s->frc_received[b] = 0;
    for (int k = 0; k < nsteps; ++k) robot_step(s, r, step0 + (uint64_t)k, pub[k]);
  }
  (void)nthreads;
  s->step += (uint64_t)nsteps;
  free(pub);
  return CDPR_OK;
}

/* MPC fan-out (BASELINE config 5, [NEW]): from each robot's current state, `samples` hypothetical trajectories of
 * `horizon` steps, each with its own jointVelocities sequence commands[B][H][S][n]; cost = sum |p(t_{k+1}) - ref|^2.
 * The simulator's own state is left untouched. */
int orc_rollout_velocity(const orc_sim *s, int samples, int horizon, const float *commands, const double *ref,
                         double *cost, int nthreads) {
  const unsigned n = s->cfg.n_cables;
  const int64_t B = (int64_t)s->cfg.batch;
#ifdef _OPENMP
  if (nthreads <= 0) nthreads = omp_get_max_threads();
#pragma omp parallel for schedule(static) num_threads(nthreads)
#endif
  for (int64_t t = 0; t < B * samples; ++t) {
    const int64_t b = t / samples, sm = t % samples;
    orc_robot *r = (orc_robot *)malloc(sizeof(orc_robot));
    memcpy(r, &s->rob[b], sizeof(orc_robot));
    double c = 0.0;
    for (int k = 0; k < horizon; ++k) {
      const float *cmd = commands + ((((size_t)b * horizon + k) * samples + sm) * n);
      for (unsigned i = 0; i < n; ++i) jfc_set_velocity_target(&r->jfc[i], (double)cmd[i]); /* PLG.cpp:206-211 */
      robot_step(s, r, s->step + (uint64_t)k, 0);
      const double ex = r->pose[0] - ref[3 * b], ey = r->pose[1] - ref[3 * b + 1], ez = r->pose[2] - ref[3 * b + 2];
      c += ex * ex + ey * ey + ez * ez;
    }
    cost[t] = c;
    free(r);
  }
  (void)nthreads;
  return CDPR_OK;
}

uint64_t orc_step_count(const orc_sim *s) { return s->step; }

void orc_get_joint_states(const orc_sim *s, double *position, double *velocity, double *effort) {
  unsigned n = s->cfg.n_cables;
  for (uint64_t b = 0; b < s->cfg.batch; ++b)
    for (unsigned i = 0; i < n; ++i) {
      if (position) position[b * n + i] = s->rob[b].pub_q[i];
      if (velocity) velocity[b * n + i] = s->rob[b].pub_qd[i];
      if (effort) effort[b * n + i] = s->rob[b].pub_eff[i];
    }
}

void orc_get_platform_state(const orc_sim *s, double *pose7, double *twist6) {
  for (uint64_t b = 0; b < s->cfg.batch; ++b) {
    if (pose7) memcpy(pose7 + 7 * b, s->rob[b].pub_pose, 7 * sizeof(double));
    if (twist6) memcpy(twist6 + 6 * b, s->rob[b].pub_twist, 6 * sizeof(double));
  }
}

void orc_get_raw_state(const orc_sim *s, double *pose7, double *twist6) {
  for (uint64_t b = 0; b < s->cfg.batch; ++b) {
    if (pose7) memcpy(pose7 + 7 * b, s->rob[b].pose, 7 * sizeof(double));
    if (twist6) memcpy(twist6 + 6 * b, s->rob[b].twist, 6 * sizeof(double));
  }
}

void orc_get_pid_debug(const orc_sim *s, double *axes9) {
  for (uint64_t b = 0; b < s->cfg.batch; ++b)
    memcpy(axes9 + CDPR_PID_DEBUG_AXES * b, s->rob[b].dbg, CDPR_PID_DEBUG_AXES * sizeof(double));
}

void orc_get_limit_state(const orc_sim *s, uint32_t *cable_mask) {
  for (uint64_t b = 0; b < s->cfg.batch; ++b) cable_mask[b] = s->rob[b].pub_limit_mask;
}

void orc_get_fk_state(const orc_sim *s, double *pose7, double *residual, int32_t *iterations) {
  for (uint64_t b = 0; b < s->cfg.batch; ++b) {
    if (pose7) memcpy(pose7 + 7 * b, s->rob[b].fk_pose, 7 * sizeof(double));
    if (residual) residual[b] = s->rob[b].pub_fk_res;
    if (iterations) iterations[b] = s->rob[b].pub_fk_iters;
  }
}

void orc_get_td_state(const orc_sim *s, double *tension, int32_t *infeasible) {
  unsigned n = s->cfg.n_cables;
  for (uint64_t b = 0; b < s->cfg.batch; ++b) {
    if (tension) memcpy(tension + n * b, s->rob[b].pub_eff, n * sizeof(double)); /* the applied force of the published step */
    if (infeasible) infeasible[b] = s->rob[b].pub_td_flag;
  }
}

int orc_max_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}
