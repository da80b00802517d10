/*
 * amenv_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * A CPU restatement, in plain C and fp64, of the one hot path of
 * LahiruCooray/rl-aerial-manipulator that libamenv.so replaces on the GPU:
 * WaypointQuadEnv.step()/reset() and Quadcopter.update().  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this file's
 * library; the product path never does.
 *
 * Citations are <file>:<line> relative to the reference root,
 *   v2 = initial-implementation-v2.
 *
 * PINNING.  The oracle is checked (tests/test_oracle_golden.py) against golden vectors
 * produced by running the unmodified reference in the build container
 * (tools/gen_golden.py -> tests/golden/*.npz):
 *   - everything except the ODE solve is an exact restatement and agrees to fp64 rounding;
 *   - the ODE solve is classical RK4 (what BASELINE.json's north_star specifies) standing in
 *     for the reference's scipy odeint/LSODA (quadcopter.py:113): teacher-forced, per step,
 *     it agrees with the reference to < 1e-6 abs on all 13 states (measured 5e-8), which is
 *     the size of LSODA's own default-tolerance error (rtol = atol = 1.49e-8).
 * The reset RNG is NOT the reference's (global MT19937 with a data-dependent draw count,
 * rl_env_scaledObs.py:44-72, cannot be reproduced per env on a GPU); it is the counter-based
 * Philox4x32-10 spec of DESIGN.md, pinned to the reference only in distribution
 * (tests/golden/reset_samples.npz).
 *
 * State layout = include/amenv.h: fstate[field][env] (double here), istate[field][env].
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../include/amenv.h"

#ifdef _OPENMP
#include <omp.h>
#endif

#define PI_D 3.14159265358979323846

/* ------------------------------------------------------------------------------------
 * Reference constants: v2/simul_files/model/params.py:10-36, v2/rl_env_scaledObs.py:30,47,56,59,126
 * ---------------------------------------------------------------------------------- */
static void inv4(const double* a, double* out) { /* Gauss-Jordan with partial pivoting */
  double m[4][8];
  for (int i = 0; i < 4; i++)
    for (int j = 0; j < 4; j++) { m[i][j] = a[i * 4 + j]; m[i][4 + j] = (i == j); }
  for (int c = 0; c < 4; c++) {
    int p = c;
    for (int r = c + 1; r < 4; r++) if (fabs(m[r][c]) > fabs(m[p][c])) p = r;
    if (p != c) for (int j = 0; j < 8; j++) { double t = m[c][j]; m[c][j] = m[p][j]; m[p][j] = t; }
    double d = m[c][c];
    for (int j = 0; j < 8; j++) m[c][j] /= d;
    for (int r = 0; r < 4; r++) if (r != c) { double f = m[r][c]; for (int j = 0; j < 8; j++) m[r][j] -= f * m[c][j]; }
  }
  for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) out[i * 4 + j] = m[i][4 + j];
}

static void inv3(const double* a, double* o) {
  double det = a[0] * (a[4] * a[8] - a[5] * a[7]) - a[1] * (a[3] * a[8] - a[5] * a[6]) + a[2] * (a[3] * a[7] - a[4] * a[6]);
  o[0] = (a[4] * a[8] - a[5] * a[7]) / det; o[1] = (a[2] * a[7] - a[1] * a[8]) / det; o[2] = (a[1] * a[5] - a[2] * a[4]) / det;
  o[3] = (a[5] * a[6] - a[3] * a[8]) / det; o[4] = (a[0] * a[8] - a[2] * a[6]) / det; o[5] = (a[2] * a[3] - a[0] * a[5]) / det;
  o[6] = (a[3] * a[7] - a[4] * a[6]) / det; o[7] = (a[1] * a[6] - a[0] * a[7]) / det; o[8] = (a[0] * a[4] - a[1] * a[3]) / det;
}

int orc_reference_quad(amenv_config* cfg, int32_t num_envs) {
  memset(cfg, 0, sizeof(*cfg));
  cfg->struct_size = (uint32_t)sizeof(*cfg);
  cfg->abi_version = AMENV_ABI_VERSION;
  cfg->num_envs = num_envs;
  cfg->dtype = AMENV_F64;
  cfg->flags = AMENV_FLAG_AUTO_RESET;
  amenv_vehicle* v = &cfg->vehicle;
  v->n_rotors = 4;
  v->mass = 0.18;                                  /* params.py:10 */
  v->g = 9.81;                                     /* params.py:11 */
  const double I[9] = {0.00025, 0, 2.55e-6, 0, 0.000232, 0, 2.55e-6, 0, 0.0003738}; /* params.py:12-14 */
  memcpy(v->inertia, I, sizeof(I));
  inv3(I, v->inv_inertia);                         /* params.py:16 */
  const double L = 0.086;                          /* params.py:17 */
  const double r = 1.5e-9 / 6.11e-8;               /* params.py:23-25  km/kf */
  const double A[16] = {1, 1, 1, 1, 0, L, 0, -L, -L, 0, L, 0, r, -r, r, -r}; /* params.py:31-34 */
  double invA[16];
  inv4(A, invA);                                   /* params.py:36 */
  for (int i = 0; i < 4; i++)
    for (int j = 0; j < 4; j++) { v->alloc[i * 4 + j] = invA[i * 4 + j]; v->mix[i * 4 + j] = A[i * 4 + j]; }
  const double maxF = 2.0 * v->mass * v->g;        /* params.py:20 */
  for (int i = 0; i < 4; i++) { v->t_min[i] = 0.0 / 4; v->t_max[i] = maxF / 4; } /* quadcopter.py:110 */
  v->moment_scale = 0.1;                           /* rl_env_scaledObs.py:126 */
  amenv_task* t = &cfg->task;
  t->variant = AMENV_TASK_V2_SCALED20;
  t->num_waypoints = 1;                            /* rl_env_scaledObs.py:47 */
  t->max_episode_steps = 2000;                     /* :56 */
  t->counter_limit = 500;                          /* :59 */
  t->rk4_substeps = 1;
  t->dt = 1.0 / 200.0;                             /* :30 */
  t->traj_sin[0] = sin(2.0 * 1.0 * PI_D);          /* utils2/utils.py:39 with t = 1 */
  t->traj_cos[0] = cos(2.0 * PI_D * 1.0);          /* utils2/utils.py:83-86 */
  return 0;
}

/* Fill the k/K trig table for a K-waypoint task (utils2/utils.py:38-39, 83). */
void orc_set_num_waypoints(amenv_config* cfg, int32_t K) {
  cfg->task.num_waypoints = K;
  for (int k = 1; k <= K; k++) {
    double t = (double)k / (double)K;
    cfg->task.traj_sin[k - 1] = sin(2.0 * t * PI_D);
    cfg->task.traj_cos[k - 1] = cos(t * 2.0 * PI_D);
  }
}

static int n_float_fields(const amenv_config* cfg) { return AMENV_F_WP0 + 3 * cfg->task.num_waypoints + 2 * cfg->vehicle.n_joints; }
int orc_n_float_fields(const amenv_config* cfg) { return n_float_fields(cfg); }

/* ------------------------------------------------------------------------------------
 * Dynamics: v2/simul_files/model/quadcopter.py
 * ---------------------------------------------------------------------------------- */

/* state_dot, quadcopter.py:66-103.  F, M are the post-mixer wrench. */
static void state_dot(const amenv_vehicle* v, const double* s, double F, const double* M, double* d) {
  const double vx = s[3], vy = s[4], vz = s[5];
  const double qw = s[6], qx = s[7], qy = s[8], qz = s[9];
  const double p = s[10], q = s[11], r = s[12];
  /* quaternion.py:46-77 builds R of the NORMALISED quaternion through (axis, angle); the
   * closed form of its third row (= wRb[:,2], the only part used at quadcopter.py:73) is
   * the usual quadratic form in q/|q| (agrees to 2e-15). */
  const double n2 = qw * qw + qx * qx + qy * qy + qz * qz;
  const double nrm = sqrt(n2);
  const double w_ = qw / nrm, x_ = qx / nrm, y_ = qy / nrm, z_ = qz / nrm;
  const double r02 = 2.0 * (x_ * z_ - w_ * y_);
  const double r12 = 2.0 * (y_ * z_ + w_ * x_);
  const double r22 = 1.0 - 2.0 * (x_ * x_ + y_ * y_);
  /* accel = 1/m * (wRb.[0,0,F] - [0,0,m g])      quadcopter.py:73-74 */
  const double im = 1.0 / v->mass;
  d[0] = vx; d[1] = vy; d[2] = vz;                 /* :89-91 */
  d[3] = im * (r02 * F);
  d[4] = im * (r12 * F);
  d[5] = im * (r22 * F - v->mass * v->g);
  /* qdot = -1/2 Omega(p,q,r) quat + K_quat*(1-|q|^2) quat, K_quat = 2   :77-82 */
  const double qe = 1.0 - n2;
  d[6] = -0.5 * (0 * qw - p * qx - q * qy - r * qz) + 2.0 * qe * qw;
  d[7] = -0.5 * (p * qw + 0 * qx - r * qy + q * qz) + 2.0 * qe * qx;
  d[8] = -0.5 * (q * qw + r * qx + 0 * qy - p * qz) + 2.0 * qe * qy;
  d[9] = -0.5 * (r * qw - q * qx + p * qy + 0 * qz) + 2.0 * qe * qz;
  /* pqrdot = invI.(M - omega x (I.omega))         :86-87 */
  const double* I = v->inertia; const double* J = v->inv_inertia;
  const double Iw0 = I[0] * p + I[1] * q + I[2] * r;
  const double Iw1 = I[3] * p + I[4] * q + I[5] * r;
  const double Iw2 = I[6] * p + I[7] * q + I[8] * r;
  const double c0 = q * Iw2 - r * Iw1, c1 = r * Iw0 - p * Iw2, c2 = p * Iw1 - q * Iw0;
  const double t0 = M[0] - c0, t1 = M[1] - c1, t2 = M[2] - c2;
  d[10] = J[0] * t0 + J[1] * t1 + J[2] * t2;
  d[11] = J[3] * t0 + J[4] * t1 + J[5] * t2;
  d[12] = J[6] * t0 + J[7] * t1 + J[8] * t2;
}

/* Action scaling, rl_env_scaledObs.py:125-126.  With a float32 action and NumPy >= 2 the
 * products are evaluated in float32, left to right, and only then widened. */
static void scale_action(const amenv_vehicle* v, const float* a, double* u) {
  volatile float f = a[0] * (float)v->mass; /* volatile: forbid excess precision / contraction */
  f = f * (float)v->g;
  u[0] = (double)f;
  for (int i = 0; i < 3; i++) { volatile float m = a[1 + i] * (float)v->moment_scale; u[1 + i] = (double)m; }
}

/* Quadcopter.update, quadcopter.py:105-114: mixer + clamp + re-mix, integrate dt, renormalise q. */
void orc_dynamics_step(const amenv_config* cfg, double* s, const float* action, double* wrench_out) {
  const amenv_vehicle* v = &cfg->vehicle;
  const int n = v->n_rotors;
  double u[4], T[AMENV_MAX_ROTORS];
  scale_action(v, action, u);
  for (int r = 0; r < n; r++) {                                   /* :109 */
    double t = 0; for (int j = 0; j < 4; j++) t += v->alloc[r * 4 + j] * u[j];
    t = fmax(fmin(t, v->t_max[r]), v->t_min[r]);                  /* :110 */
    T[r] = t;
  }
  double F = 0, M[3] = {0, 0, 0};
  for (int r = 0; r < n; r++) F += v->mix[0 * n + r] * T[r];      /* :111 (row 0 of A is all ones) */
  for (int i = 0; i < 3; i++) for (int r = 0; r < n; r++) M[i] += v->mix[(1 + i) * n + r] * T[r]; /* :112 */
  if (wrench_out) { wrench_out[0] = u[0]; wrench_out[1] = u[1]; wrench_out[2] = u[2]; wrench_out[3] = u[3];
                    wrench_out[4] = F; wrench_out[5] = M[0]; wrench_out[6] = M[1]; wrench_out[7] = M[2]; }
  /* :113 -- odeint over [0,dt] restated as classical RK4 (north_star), wrench held constant */
  const int ns = cfg->task.rk4_substeps > 0 ? cfg->task.rk4_substeps : 1;
  const double h = cfg->task.dt / ns;
  for (int it = 0; it < ns; it++) {
    double k1[13], k2[13], k3[13], k4[13], y[13];
    state_dot(v, s, F, M, k1);
    for (int i = 0; i < 13; i++) y[i] = s[i] + 0.5 * h * k1[i];
    state_dot(v, y, F, M, k2);
    for (int i = 0; i < 13; i++) y[i] = s[i] + 0.5 * h * k2[i];
    state_dot(v, y, F, M, k3);
    for (int i = 0; i < 13; i++) y[i] = s[i] + h * k3[i];
    state_dot(v, y, F, M, k4);
    for (int i = 0; i < 13; i++) s[i] += h / 6.0 * (k1[i] + 2.0 * k2[i] + 2.0 * k3[i] + k4[i]);
  }
  const double nq = sqrt(s[6] * s[6] + s[7] * s[7] + s[8] * s[8] + s[9] * s[9]); /* :114 */
  s[6] /= nq; s[7] /= nq; s[8] /= nq; s[9] /= nq;
}


/* ------------------------------------------------------------------------------------
 * Hexacopter + 3-joint arm (BASELINE config 3).  NO reference dynamics exist for it (the reference simulates it in
 * Gazebo): this is the fp64 statement of the model specified in DESIGN.md "arm"; parity unpinned, validated by
 * invariants (tests/test_arm_cpu.py: momentum conservation, rigid limits, locked-arm = composite rigid body).
 *   state  s[19] = [p(3) world, v(3) world, q(4), w(3) body, th(3), thd(3)]
 *   joints : acceleration-limited position servos  thdd = clamp(kp (cmd - th) - kd thd, +-amax)
 *   base   : exact rigid multibody reaction (Newton-Euler summed over base + links about the body origin O)
 * ---------------------------------------------------------------------------------- */
static void cross3(const double* a, const double* b, double* c) { c[0] = a[1] * b[2] - a[2] * b[1]; c[1] = a[2] * b[0] - a[0] * b[2]; c[2] = a[0] * b[1] - a[1] * b[0]; }
static void matvec3(const double* M, const double* x, double* y) { for (int i = 0; i < 3; i++) y[i] = M[3 * i] * x[0] + M[3 * i + 1] * x[1] + M[3 * i + 2] * x[2]; }
static void matmul3(const double* A, const double* B, double* C) { for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { double t = 0; for (int k = 0; k < 3; k++) t += A[3 * i + k] * B[3 * k + j]; C[3 * i + j] = t; } }
static void rodrigues(const double* a, double th, double* R) { /* rotation by th about unit axis a */
  const double c = cos(th), s_ = sin(th), t = 1 - c;
  R[0] = c + a[0] * a[0] * t;        R[1] = a[0] * a[1] * t - a[2] * s_; R[2] = a[0] * a[2] * t + a[1] * s_;
  R[3] = a[1] * a[0] * t + a[2] * s_; R[4] = c + a[1] * a[1] * t;        R[5] = a[1] * a[2] * t - a[0] * s_;
  R[6] = a[2] * a[0] * t - a[1] * s_; R[7] = a[2] * a[1] * t + a[0] * s_; R[8] = c + a[2] * a[2] * t;
}

void orc_arm_rhs(const amenv_config* cfg, const double* s, double F, const double* M, const double* th_cmd, double* d) {
  const amenv_vehicle* v = &cfg->vehicle;
  const int nj = v->n_joints;
  const double* om = &s[10]; const double* th = &s[13]; const double* thd = &s[16];
  /* rotation of the normalised quaternion; body->world is its transpose (same convention as state_dot above) */
  const double n2 = s[6] * s[6] + s[7] * s[7] + s[8] * s[8] + s[9] * s[9], nr = sqrt(n2);
  const double qw = s[6] / nr, qx = s[7] / nr, qy = s[8] / nr, qz = s[9] / nr;
  const double Rq[9] = {1 - 2 * (qy * qy + qz * qz), 2 * (qx * qy - qw * qz), 2 * (qx * qz + qw * qy),
                        2 * (qx * qy + qw * qz), 1 - 2 * (qx * qx + qz * qz), 2 * (qy * qz - qw * qx),
                        2 * (qx * qz - qw * qy), 2 * (qy * qz + qw * qx), 1 - 2 * (qx * qx + qy * qy)};
  const double gb[3] = {-v->g * Rq[2], -v->g * Rq[5], -v->g * Rq[8]};     /* gravity in body components: Rq (0,0,-g) */
  double thdd[AMENV_MAX_JOINTS];
  for (int k = 0; k < nj; k++) {
    double a = v->joint_kp * (th_cmd[k] - th[k]) - v->joint_kd * thd[k];
    thdd[k] = a > v->joint_acc_max ? v->joint_acc_max : (a < -v->joint_acc_max ? -v->joint_acc_max : a);
  }
  double m_links = 0; for (int k = 0; k < nj; k++) m_links += v->link_mass[k];
  const double m0 = v->mass - m_links, mtot = v->mass;
  /* accumulators: S = sum m r, I_O, bias force fb, bias moment nb */
  double S[3] = {0, 0, 0}, IO[9], fb[3] = {0, 0, 0}, nb[3] = {0, 0, 0};
  memcpy(IO, v->inertia, sizeof(IO));                                      /* base: r = 0, J = I0 */
  { double Jw[3], c[3]; matvec3(v->inertia, om, Jw); cross3(om, Jw, c); for (int i = 0; i < 3; i++) nb[i] += c[i]; }   /* Omega x J Omega */
  (void)m0;
  /* chain kinematics relative to the body frame */
  double R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, p[3] = {0, 0, 0}, w[3] = {0, 0, 0}, al[3] = {0, 0, 0}, pd[3] = {0, 0, 0}, pdd[3] = {0, 0, 0};
  for (int k = 0; k < nj; k++) {
    const double* o = &v->joint_origin[3 * k]; const double* ax = &v->joint_axis[3 * k];
    double Ro[3], t1[3], t2[3];
    matvec3(R, o, Ro);                                                     /* joint origin offset in body components */
    /* origin of frame k: p += R_{k-1} o ; its velocity / acceleration use the PARENT's w, alpha */
    cross3(w, Ro, t1); for (int i = 0; i < 3; i++) pd[i] += t1[i];
    cross3(al, Ro, t1); double wxRo[3]; cross3(w, Ro, wxRo); cross3(w, wxRo, t2);
    for (int i = 0; i < 3; i++) { pdd[i] += t1[i] + t2[i]; p[i] += Ro[i]; }
    double z[3]; matvec3(R, ax, z);                                        /* joint axis in body components */
    double wxz[3]; cross3(w, z, wxz);
    for (int i = 0; i < 3; i++) { al[i] += z[i] * thdd[k] + wxz[i] * thd[k]; }   /* alpha_k = alpha_{k-1} + z thdd + (w_{k-1} x z) thd */
    for (int i = 0; i < 3; i++) w[i] += z[i] * thd[k];                     /* w_k */
    double Rj[9], Rn[9]; rodrigues(ax, th[k], Rj); matmul3(R, Rj, Rn); memcpy(R, Rn, sizeof(R));
    /* link k centre of mass */
    double Rc[3], r[3], u[3], a_[3];
    matvec3(R, &v->link_com[3 * k], Rc);
    cross3(w, Rc, t1); cross3(al, Rc, t2); double wxt1[3]; cross3(w, t1, wxt1);
    for (int i = 0; i < 3; i++) { r[i] = p[i] + Rc[i]; u[i] = pd[i] + t1[i]; a_[i] = pdd[i] + t2[i] + wxt1[i]; }
    /* inertia in body components: J = R I R^T */
    double RI[9], Rt[9], J[9];
    matmul3(R, &v->link_inertia[9 * k], RI);
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) Rt[3 * i + j] = R[3 * j + i];
    matmul3(RI, Rt, J);
    const double m = v->link_mass[k];
    /* bias acceleration of the CoM: w x (w x r) + 2 w x u + a_rel (with the base's angular velocity om) */
    double oxr[3], oxoxr[3], oxu[3], ab[3];
    cross3(om, r, oxr); cross3(om, oxr, oxoxr); cross3(om, u, oxu);
    for (int i = 0; i < 3; i++) ab[i] = oxoxr[i] + 2 * oxu[i] + a_[i];
    double rxab[3]; cross3(r, ab, rxab);
    /* angular part: J (alpha + om x w) + Omega x (J Omega), Omega = om + w */
    double oxw[3], aa[3], Jaa[3], Om[3], JOm[3], OxJO[3];
    cross3(om, w, oxw); for (int i = 0; i < 3; i++) { aa[i] = al[i] + oxw[i]; Om[i] = om[i] + w[i]; }
    matvec3(J, aa, Jaa); matvec3(J, Om, JOm); cross3(Om, JOm, OxJO);
    const double r2 = r[0] * r[0] + r[1] * r[1] + r[2] * r[2];
    for (int i = 0; i < 3; i++) {
      S[i] += m * r[i]; fb[i] += m * ab[i]; nb[i] += m * rxab[i] + Jaa[i] + OxJO[i];
      for (int j = 0; j < 3; j++) IO[3 * i + j] += J[3 * i + j] + m * ((i == j ? r2 : 0.0) - r[i] * r[j]);
    }
  }
  /* external wrench about O in body components: rotor thrust + moments, gravity at every CoM */
  double f[3], n[3], Sxg[3];
  cross3(S, gb, Sxg);
  for (int i = 0; i < 3; i++) { f[i] = mtot * gb[i] - fb[i]; n[i] = M[i] + Sxg[i] - nb[i]; }
  f[2] += F;
  /* [mtot 1, -[S]x; [S]x, I_O] [A; wd] = [f; n]  ->  I_c wd = n - S x f / mtot,  I_c = I_O - (|S|^2 1 - S S^T)/mtot */
  double Ic[9], Sxf[3], rhs[3];
  const double S2 = S[0] * S[0] + S[1] * S[1] + S[2] * S[2];
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) Ic[3 * i + j] = IO[3 * i + j] - ((i == j ? S2 : 0.0) - S[i] * S[j]) / mtot;
  cross3(S, f, Sxf);
  for (int i = 0; i < 3; i++) rhs[i] = n[i] - Sxf[i] / mtot;
  double Ici[9], wd[3], A[3], Sxwd[3];
  inv3(Ic, Ici); matvec3(Ici, rhs, wd);
  cross3(S, wd, Sxwd);
  for (int i = 0; i < 3; i++) A[i] = (f[i] + Sxwd[i]) / mtot;
  /* derivatives */
  d[0] = s[3]; d[1] = s[4]; d[2] = s[5];
  for (int i = 0; i < 3; i++) d[3 + i] = Rq[0 + i] * A[0] + Rq[3 + i] * A[1] + Rq[6 + i] * A[2];   /* v' = Rq^T A */
  const double pq = om[0], qq = om[1], rq = om[2], qe = 1.0 - n2;
  d[6] = -0.5 * (-pq * s[7] - qq * s[8] - rq * s[9]) + 2.0 * qe * s[6];
  d[7] = -0.5 * (pq * s[6] - rq * s[8] + qq * s[9]) + 2.0 * qe * s[7];
  d[8] = -0.5 * (qq * s[6] + rq * s[7] - pq * s[9]) + 2.0 * qe * s[8];
  d[9] = -0.5 * (rq * s[6] - qq * s[7] + pq * s[8]) + 2.0 * qe * s[9];
  d[10] = wd[0]; d[11] = wd[1]; d[12] = wd[2];
  for (int k = 0; k < 3; k++) { d[13 + k] = k < nj ? thd[k] : 0.0; d[16 + k] = k < nj ? thdd[k] : 0.0; }
}

/* Forward kinematics of the arm: world position of the tool point.  s = [p(3), v(3), q(4) unit, w(3), th(3), thd(3)].
 * Chain of revolute joints (manipulator.sdf:99,159,233: origins; :103,163,237: axes) from the body origin O, then
 * tool_offset in the last link's frame (:371,450), then world = p + Rq^T (body components), Rq as in orc_arm_rhs. */
void orc_ee_position(const amenv_config* cfg, const double* s, double* ee) {
  const amenv_vehicle* v = &cfg->vehicle;
  double R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, p[3] = {0, 0, 0};
  for (int k = 0; k < v->n_joints; k++) {
    double Ro[3], Rj[9], Rn[9];
    matvec3(R, &v->joint_origin[3 * k], Ro);
    for (int i = 0; i < 3; i++) p[i] += Ro[i];
    rodrigues(&v->joint_axis[3 * k], s[13 + k], Rj); matmul3(R, Rj, Rn); memcpy(R, Rn, sizeof(R));
  }
  double Rt[3];
  matvec3(R, v->tool_offset, Rt);
  for (int i = 0; i < 3; i++) p[i] += v->n_joints ? Rt[i] : 0.0;
  const double n2 = s[6] * s[6] + s[7] * s[7] + s[8] * s[8] + s[9] * s[9], nr = sqrt(n2);
  const double qw = s[6] / nr, qx = s[7] / nr, qy = s[8] / nr, qz = s[9] / nr;
  const double Rq[9] = {1 - 2 * (qy * qy + qz * qz), 2 * (qx * qy - qw * qz), 2 * (qx * qz + qw * qy),
                        2 * (qx * qy + qw * qz), 1 - 2 * (qx * qx + qz * qz), 2 * (qy * qz - qw * qx),
                        2 * (qx * qz - qw * qy), 2 * (qy * qz + qw * qx), 1 - 2 * (qx * qx + qy * qy)};
  for (int i = 0; i < 3; i++) ee[i] = s[i] + Rq[0 + i] * p[0] + Rq[3 + i] * p[1] + Rq[6 + i] * p[2];   /* p + Rq^T ee_body */
}

/* One control step of the arm vehicle: mixer as for the rigid body, joint commands from actions 4..6, RK4 on 19 states. */
void orc_arm_dynamics_step(const amenv_config* cfg, double* s, const float* action, double* wrench_out) {
  const amenv_vehicle* v = &cfg->vehicle;
  const int n = v->n_rotors;
  double u[4], T[AMENV_MAX_ROTORS];
  scale_action(v, action, u);
  double F = 0, M[3] = {0, 0, 0};
  for (int r = 0; r < n; r++) {
    double t = 0; for (int j = 0; j < 4; j++) t += v->alloc[r * 4 + j] * u[j];
    t = fmax(fmin(t, v->t_max[r]), v->t_min[r]); T[r] = t; F += t;
  }
  for (int i = 0; i < 3; i++) for (int r = 0; r < n; r++) M[i] += v->mix[(1 + i) * n + r] * T[r];
  double cmd[3];
  for (int k = 0; k < 3; k++) {   /* action -1..1 -> joint range; formed in fp32 like the other action scalings */
    if (k >= v->n_joints) { cmd[k] = 0.0; continue; }   /* an arm with fewer joints has fewer action entries */
    const float lo = (float)v->joint_limit[2 * k], hi = (float)v->joint_limit[2 * k + 1];
    volatile float half = 0.5f * (hi - lo), mid = 0.5f * (hi + lo);
    volatile float c = fmaf(action[4 + k], half, mid);
    cmd[k] = (double)c;
  }
  if (wrench_out) { wrench_out[0] = F; wrench_out[1] = M[0]; wrench_out[2] = M[1]; wrench_out[3] = M[2]; wrench_out[4] = cmd[0]; wrench_out[5] = cmd[1]; wrench_out[6] = cmd[2]; }
  const int ns = cfg->task.rk4_substeps > 0 ? cfg->task.rk4_substeps : 1;
  const double h = cfg->task.dt / ns;
  for (int it = 0; it < ns; it++) {
    double k1[19], k2[19], k3[19], k4[19], y[19];
    orc_arm_rhs(cfg, s, F, M, cmd, k1);
    for (int i = 0; i < 19; i++) y[i] = s[i] + 0.5 * h * k1[i];
    orc_arm_rhs(cfg, y, F, M, cmd, k2);
    for (int i = 0; i < 19; i++) y[i] = s[i] + 0.5 * h * k2[i];
    orc_arm_rhs(cfg, y, F, M, cmd, k3);
    for (int i = 0; i < 19; i++) y[i] = s[i] + h * k3[i];
    orc_arm_rhs(cfg, y, F, M, cmd, k4);
    for (int i = 0; i < 19; i++) s[i] += h / 6.0 * (k1[i] + 2.0 * k2[i] + 2.0 * k3[i] + k4[i]);
  }
  const double nq = sqrt(s[6] * s[6] + s[7] * s[7] + s[8] * s[8] + s[9] * s[9]);
  s[6] /= nq; s[7] /= nq; s[8] /= nq; s[9] /= nq;
}

/* ------------------------------------------------------------------------------------
 * Environment: v2/rl_env_scaledObs.py
 * ---------------------------------------------------------------------------------- */
typedef struct {
  double s[19];   /* 13 rigid-body states (+ 3 joint angles + 3 joint rates with an arm) */
  double wp[AMENV_MAX_WAYPOINTS][3];
  double final_yaw, last_distance, ep_return;
  int step, counter, wp_index, fwr, counter_activated, episode;
  int k_env; /* v1: waypoints of THIS episode (v1/rl_env_scaledObs.py:38); 0 = cfg->task.num_waypoints */
} env_t;

static void load_env(const amenv_config* cfg, int n, const double* f, const int32_t* is, int i, env_t* e) {
  for (int k = 0; k < 13; k++) e->s[k] = f[(size_t)k * n + i];
  { /* joint fields follow the waypoints: th[0..nj) then thd[0..nj) (enum amenv_float_field); absent joints stay 0 */
    const int nj = cfg->vehicle.n_joints, j0 = AMENV_F_WP0 + 3 * cfg->task.num_waypoints;
    for (int k = 0; k < 3; k++) {
      e->s[13 + k] = k < nj ? f[(size_t)(j0 + k) * n + i] : 0.0;
      e->s[16 + k] = k < nj ? f[(size_t)(j0 + nj + k) * n + i] : 0.0;
    }
  }
  e->final_yaw = f[(size_t)AMENV_F_FINAL_YAW * n + i];
  e->last_distance = f[(size_t)AMENV_F_LAST_DISTANCE * n + i];
  e->ep_return = f[(size_t)AMENV_F_EP_RETURN * n + i];
  for (int k = 0; k < cfg->task.num_waypoints; k++)
    for (int c = 0; c < 3; c++) e->wp[k][c] = f[(size_t)(AMENV_F_WP0 + 3 * k + c) * n + i];
  e->step = is[(size_t)AMENV_I_STEP * n + i];
  e->counter = is[(size_t)AMENV_I_COUNTER * n + i];
  int fl = is[(size_t)AMENV_I_FLAGS * n + i];
  e->wp_index = fl & 15; e->k_env = (fl >> 4) & 15; e->fwr = (fl & AMENV_FLAGBIT_FWR) != 0; e->counter_activated = (fl & AMENV_FLAGBIT_COUNTER_ACTIVE) != 0;
  e->episode = is[(size_t)AMENV_I_EPISODE * n + i];
}

static void store_env(const amenv_config* cfg, int n, double* f, int32_t* is, int i, const env_t* e) {
  for (int k = 0; k < 13; k++) f[(size_t)k * n + i] = e->s[k];
  {
    const int nj = cfg->vehicle.n_joints, j0 = AMENV_F_WP0 + 3 * cfg->task.num_waypoints;
    for (int k = 0; k < nj; k++) { f[(size_t)(j0 + k) * n + i] = e->s[13 + k]; f[(size_t)(j0 + nj + k) * n + i] = e->s[16 + k]; }
  }
  f[(size_t)AMENV_F_FINAL_YAW * n + i] = e->final_yaw;
  f[(size_t)AMENV_F_LAST_DISTANCE * n + i] = e->last_distance;
  f[(size_t)AMENV_F_EP_RETURN * n + i] = e->ep_return;
  for (int k = 0; k < cfg->task.num_waypoints; k++)
    for (int c = 0; c < 3; c++) f[(size_t)(AMENV_F_WP0 + 3 * k + c) * n + i] = e->wp[k][c];
  is[(size_t)AMENV_I_STEP * n + i] = e->step;
  is[(size_t)AMENV_I_COUNTER * n + i] = e->counter;
  is[(size_t)AMENV_I_FLAGS * n + i] = (e->wp_index & 15) | ((e->k_env & 15) << 4) | (e->fwr ? AMENV_FLAGBIT_FWR : 0) | (e->counter_activated ? AMENV_FLAGBIT_COUNTER_ACTIVE : 0);
  is[(size_t)AMENV_I_EPISODE * n + i] = e->episode;
}

static int is_v1(const amenv_config* cfg) { return cfg->task.variant == AMENV_TASK_V1_SCALED17 || cfg->task.variant == AMENV_TASK_V1_RAW17; }
static int obs_dim(const amenv_config* cfg) { return is_v1(cfg) ? 17 : 20 + 2 * cfg->vehicle.n_joints + (cfg->vehicle.n_joints ? 3 : 0); }
/* the point the waypoint task measures from: base position, or the arm's tool point (AMENV_EE_TASK_TOOL; this build's extension) */
static void task_point(const amenv_config* cfg, const env_t* e, double* pt) {
  if (cfg->vehicle.n_joints && cfg->task.ee_task == AMENV_EE_TASK_TOOL) orc_ee_position(cfg, e->s, pt);
  else { pt[0] = e->s[0]; pt[1] = e->s[1]; pt[2] = e->s[2]; }
}
static int act_dim(const amenv_config* cfg) { return 4 + cfg->vehicle.n_joints; }
int orc_act_dim(const amenv_config* cfg) { return act_dim(cfg); }
int orc_obs_dim(const amenv_config* cfg) { return obs_dim(cfg); }
static int env_K(const amenv_config* cfg, const env_t* e) { return e->k_env ? e->k_env : cfg->task.num_waypoints; }

/* v1 _get_observation: v1/rl_env_scaledObs.py:65-83 (scaled) and v1/rl_env.py:65-83 (raw: pos, vel, omega, rel_pos unscaled) */
static void observe_v1(const amenv_config* cfg, const env_t* e, float* obs) {
  const int K = env_K(cfg, e);
  const int idx = e->wp_index < K ? e->wp_index : K - 1;
  const double* cw = e->wp[idx];
  const int raw = cfg->task.variant == AMENV_TASK_V1_RAW17;
  const double sp = raw ? 1.0 : 10.0, sv = raw ? 1.0 : 5.0, sr = raw ? 1.0 : 2.0;
  for (int c = 0; c < 3; c++) obs[c] = (float)(e->s[c] / sp);              /* :75 */
  for (int c = 0; c < 3; c++) obs[3 + c] = (float)(e->s[3 + c] / sv);      /* :76 */
  for (int c = 0; c < 4; c++) obs[6 + c] = (float)(e->s[6 + c]);           /* :77 */
  for (int c = 0; c < 3; c++) obs[10 + c] = (float)(e->s[10 + c] / sv);    /* :78 */
  for (int c = 0; c < 3; c++) obs[13 + c] = (float)((cw[c] - e->s[c]) / sr); /* :71,79 */
  obs[16] = idx >= K - 1 ? 1.0f : 0.0f;  /* :80 np.allclose(current_waypoint, waypoint_list[-1]) */
}

/* _get_observation, rl_env_scaledObs.py:98-121 */
static void observe_v2(const amenv_config* cfg, const env_t* e, float* obs) {
  const int K = cfg->task.num_waypoints;
  const int idx = e->wp_index < K ? e->wp_index : K - 1; /* current_waypoint stays at the last one (:151-152) */
  const double* cw = e->wp[idx];
  for (int c = 0; c < 3; c++) obs[c] = (float)(e->s[c] / 10.0);            /* :112 */
  for (int c = 0; c < 3; c++) obs[3 + c] = (float)(e->s[3 + c] / 5.0);     /* :113 */
  for (int c = 0; c < 4; c++) obs[6 + c] = (float)(e->s[6 + c]);           /* :114 */
  for (int c = 0; c < 3; c++) obs[10 + c] = (float)(e->s[10 + c] / 5.0);   /* :115 */
  double tp[3]; task_point(cfg, e, tp);
  for (int c = 0; c < 3; c++) obs[13 + c] = (float)((cw[c] - tp[c]) / 2.0); /* :104,116 */
  for (int c = 0; c < 3; c++) {                                            /* :105-108,117 */
    double rel = 0.0;
    if (!(e->wp_index >= K - 1)) rel = e->wp[e->wp_index + 1][c] - cw[c];
    obs[16 + c] = (float)(rel / 2.0);
  }
  obs[19] = (float)(e->final_yaw / PI_D);                                  /* :118 */
  for (int k = 0; k < cfg->vehicle.n_joints; k++) {                        /* arm (this build's extension): joint angle / pi, rate / 5 */
    obs[20 + k] = (float)(e->s[13 + k] / PI_D);
    obs[20 + cfg->vehicle.n_joints + k] = (float)(e->s[16 + k] / 5.0);
  }
  if (cfg->vehicle.n_joints) {                                             /* (tool point - base position) / 0.5, world axes */
    double ee[3]; orc_ee_position(cfg, e->s, ee);
    for (int c = 0; c < 3; c++) obs[20 + 2 * cfg->vehicle.n_joints + c] = (float)((ee[c] - e->s[c]) / 0.5);
  }
}

static void observe(const amenv_config* cfg, const env_t* e, float* obs) {
  if (is_v1(cfg)) observe_v1(cfg, e, obs); else observe_v2(cfg, e, obs);
}

/* quaternion_to_rpy, utils2/utils.py:4-9: scipy Rotation.from_quat([x,y,z,w]).as_euler('xyz'),
 * closed form (agrees to 1e-14 away from gimbal lock). */
static void quat_to_rpy(const double* q, double* rpy) {
  const double w = q[0], x = q[1], y = q[2], z = q[3];
  rpy[0] = atan2(2.0 * (w * x + y * z), 1.0 - 2.0 * (x * x + y * y));
  double sp = 2.0 * (w * y - z * x);
  sp = sp > 1.0 ? 1.0 : (sp < -1.0 ? -1.0 : sp);
  rpy[1] = asin(sp);
  rpy[2] = atan2(2.0 * (w * z + x * y), 1.0 - 2.0 * (y * y + z * z));
}

static double norm3(const double* a) { return sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]); }

/* One WaypointQuadEnv.step, rl_env_scaledObs.py:123-196 (+ _calculate_reward :198-231).
 * Returns info bits; *reward_out the f64 reward.  No auto-reset here. */
static uint32_t env_step_v2(const amenv_config* cfg, env_t* e, const float* action, double* reward_out) {
  const int K = cfg->task.num_waypoints;
  if (cfg->vehicle.n_joints) orc_arm_dynamics_step(cfg, e->s, action, NULL);
  else orc_dynamics_step(cfg, e->s, action, NULL);                         /* :125-131 */

  uint32_t bits = 0;
  if (cfg->flags & AMENV_FLAG_NAN_GUARD) {   /* deviation from the reference, documented in DESIGN.md */
    int bad = 0; for (int k = 0; k < 13 + 2 * cfg->vehicle.n_joints; k++) bad |= !isfinite(e->s[k]);
    if (bad) {
      bits = AMENV_INFO_TERMINATED | AMENV_INFO_NONFINITE;
      if (e->step >= cfg->task.max_episode_steps) bits |= AMENV_INFO_TRUNCATED;
      e->step += 1;
      *reward_out = -100.0;
      return bits;
    }
  }
  const double* pos = &e->s[0]; const double* vel = &e->s[3]; const double* om = &e->s[10];
  const int idx0 = e->wp_index < K ? e->wp_index : K - 1;
  const double* cw = e->wp[idx0];

  /* ---- _calculate_reward (:198-231): post-update state, pre-update waypoint/flags */
  double tp[3]; task_point(cfg, e, tp);                                    /* = pos for rigid vehicles (the reference) */
  double dvec[3] = {tp[0] - cw[0], tp[1] - cw[1], tp[2] - cw[2]};
  const double distance = norm3(dvec);                                     /* :204 */
  double distance_reward = -distance * 10;                                 /* :207 */
  const double vn = norm3(vel), wn = norm3(om);
  double speed_penalty = -0.1 * (vn * vn);                                 /* :208 */
  if (wn > 0.1) speed_penalty -= 0.01 * (wn * wn);                         /* :209-210 */
  double time_penalty = -0.1;                                              /* :211 */
  double progress_reward;
  if (e->last_distance >= 0.0) {                                           /* :214 (None encoded < 0) */
    const double progress = e->last_distance - distance;                   /* :215 */
    progress_reward = 20 * progress;                                       /* :216 */
    if (progress_reward > 0) progress_reward += 2;                         /* :217-218 */
  } else progress_reward = 0.0;                                            /* :220 */
  e->last_distance = distance;                                             /* :222 */
  if (e->fwr) {                                                            /* :224-228 */
    progress_reward = 0.0; time_penalty = 0.0;
    if (distance < 0.1) distance_reward = 1.0;
  }
  double reward = distance_reward + speed_penalty + time_penalty + progress_reward; /* :231 */

  /* ---- step body (:138-196) */
  double rpy[3];
  quat_to_rpy(&e->s[6], rpy);                                              /* :141 */
  const double roll = rpy[0], pitch = rpy[1], yaw = rpy[2];
  const int truncated = e->step >= cfg->task.max_episode_steps;            /* :144 */
  e->step += 1;                                                            /* :145 */
  if (truncated) bits |= AMENV_INFO_TRUNCATED;
  /* distance_to_waypoint (:143) is the same quantity as `distance` above */
  if (distance < 0.1) {                                                    /* :147 */
    if (!e->fwr) { e->wp_index += 1; reward += 100.0; }                    /* :148-150 */
    if (e->wp_index < K) {                                                 /* :151-152: next waypoint, fall through */
    } else {
      const int stopped = (vn < 0.1) && (wn < 0.1);
      if (!e->fwr) {                                                       /* :156-164 */
        e->counter_activated = 1; e->fwr = 1;
        const double stopping_bonus = vn < 1 ? 150.0 * (1 - vn * vn) : 0.0; /* :162 */
        const double dy = fabs(yaw - e->final_yaw);
        const double yaw_bonus = dy < 2 * PI_D ? 100.0 * (1 - dy / (2 * PI_D)) : 0.0; /* :163 */
        *reward_out = reward + 200.0 + stopping_bonus + yaw_bonus;         /* :164 */
        return bits | AMENV_INFO_SUCCESS | (stopped ? AMENV_INFO_STOPPED : 0);
      } else {
        const double dy = fabs(yaw - e->final_yaw);
        const double yaw_bonus = dy < 2 * PI_D ? 30.0 * (1 - dy / (2 * PI_D)) : 0.0;          /* :169,175 */
        const double roll_bonus = fabs(roll) < 0.2 ? 10.0 * (1 - fabs(roll) / 0.2) : -.1 * fabs(roll);   /* :170,176 */
        const double pitch_bonus = fabs(pitch) < 0.2 ? 10.0 * (1 - fabs(pitch) / 0.2) : -.1 * fabs(pitch); /* :171,177 */
        *reward_out = reward + yaw_bonus + roll_bonus + pitch_bonus;
        bits |= AMENV_INFO_SUCCESS | (stopped ? AMENV_INFO_STOPPED : 0);
        if (e->counter <= cfg->task.counter_limit) { e->counter += 1; return bits; } /* :166-173 */
        return bits | AMENV_INFO_TERMINATED;                               /* :174-179 */
      }
    }
  }
  if (e->counter_activated) e->counter += 1;                               /* :181-182 */
  if (pos[2] < 0.1) {                                                      /* :188-192 */
    reward -= 100;
    if (vel[2] < 0) reward += vel[2] * 100.0;
    *reward_out = reward;
    return bits | AMENV_INFO_TERMINATED | AMENV_INFO_CRASHED;
  }
  if (norm3(pos) > 10) {                                                   /* :193-195 */
    reward -= 100.0;
    *reward_out = reward;
    return bits | AMENV_INFO_TERMINATED | AMENV_INFO_OOB;
  }
  *reward_out = reward;                                                    /* :196 */
  return bits;
}

/* One v1 WaypointQuadEnv.step: v1/rl_env_scaledObs.py:85-140 (+ _calculate_reward :142-168); v1/rl_env.py is identical
 * except for the observation scaling. */
static uint32_t env_step_v1(const amenv_config* cfg, env_t* e, const float* action, double* reward_out) {
  const int K = env_K(cfg, e);
  orc_dynamics_step(cfg, e->s, action, NULL);                              /* :87-90 */
  uint32_t bits = 0;
  if (cfg->flags & AMENV_FLAG_NAN_GUARD) {
    int bad = 0; for (int k = 0; k < 13; k++) bad |= !isfinite(e->s[k]);
    if (bad) {
      bits = AMENV_INFO_TERMINATED | AMENV_INFO_NONFINITE;
      if (e->step >= cfg->task.max_episode_steps) bits |= AMENV_INFO_TRUNCATED;
      e->step += 1; *reward_out = -100.0; return bits;
    }
  }
  const double* pos = &e->s[0]; const double* vel = &e->s[3]; const double* om = &e->s[10];
  const int idx0 = e->wp_index < K ? e->wp_index : K - 1;
  const double* cw = e->wp[idx0];
  /* _calculate_reward (:142-168) */
  double dvec[3] = {pos[0] - cw[0], pos[1] - cw[1], pos[2] - cw[2]};
  const double distance = norm3(dvec);                                     /* :148 */
  const double distance_reward = -distance * 2;                            /* :151 */
  const double vn = norm3(vel), wn = norm3(om);
  double speed_penalty = -0.1 * (vn * vn);                                 /* :152 */
  if (wn > 0.1) speed_penalty -= 0.01 * (wn * wn);                         /* :153-154 */
  const double time_penalty = -0.1;                                        /* :155 */
  double progress_reward;
  if (e->last_distance >= 0.0) {                                           /* :158-164 */
    progress_reward = 20 * (e->last_distance - distance);
    if (progress_reward > 0) progress_reward += 2;
  } else progress_reward = 0.0;
  e->last_distance = distance;                                             /* :166 */
  double reward = distance_reward + speed_penalty + time_penalty + progress_reward; /* :168 */
  /* approach shaping (:102-110) */
  const double dir[3] = {cw[0] - pos[0], cw[1] - pos[1], cw[2] - pos[2]};
  const double dn = norm3(dir);
  const double vtw = vel[0] * (dir[0] / dn) + vel[1] * (dir[1] / dn) + vel[2] * (dir[2] / dn);   /* :103-104 */
  if (distance < 0.5 && vtw > 0.1) reward += 10.0;                         /* :107-108 */
  else if (distance < 0.5 && vtw < 0.1) reward -= 10.0;                    /* :109-110 */
  if (distance < 0.1) {                                                    /* :111 */
    reward += 100.0; e->wp_index += 1;                                     /* :112-113 */
    if (!(e->wp_index < K)) {                                              /* :117-124: all waypoints reached */
      const double rot_b = wn < 0.1 ? 100.0 : -20.0 * wn;                  /* :122 */
      const double stop_b = vn < 0.1 ? 100.0 : -10.0 * vn;                 /* :123 */
      *reward_out = reward + 400.0 + stop_b + rot_b;                       /* :124: returns BEFORE current_step += 1, truncated = False */
      return AMENV_INFO_TERMINATED | AMENV_INFO_SUCCESS | (vn < 0.1 ? AMENV_INFO_STOPPED : 0);
    }
  }
  const int truncated = e->step >= cfg->task.max_episode_steps;            /* :126 */
  e->step += 1;                                                            /* :127 */
  if (truncated) bits |= AMENV_INFO_TRUNCATED;
  if (pos[2] < 0.1) {                                                      /* :131-136 */
    reward -= 100;
    if (vel[2] < 0) reward += vel[2] * 100.0;
    *reward_out = reward;
    return bits | AMENV_INFO_TERMINATED | AMENV_INFO_CRASHED;
  }
  if (norm3(pos) > 10) {                                                   /* :137-139 */
    reward -= 100.0; *reward_out = reward;
    return bits | AMENV_INFO_TERMINATED | AMENV_INFO_OOB;
  }
  *reward_out = reward;                                                    /* :140 */
  return bits;
}

/* ------------------------------------------------------------------------------------
 * reset: rl_env_scaledObs.py:40-79 + utils2/utils.py:12-95, with the DESIGN.md Philox spec
 * in place of global np.random.  All draws are formed in fp32 (the product dtype) and widened.
 * ---------------------------------------------------------------------------------- */
static void philox4x32_10(const uint32_t key[2], const uint32_t ctr[4], uint32_t out[4]) {
  uint32_t k0 = key[0], k1 = key[1], c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
  for (int r = 0; r < 10; r++) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

void orc_philox(uint64_t seed, uint64_t gid, uint32_t episode, uint32_t block, uint32_t* out) {
  const uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
  const uint32_t ctr[4] = {(uint32_t)gid, (uint32_t)(gid >> 32), episode, block};
  philox4x32_10(key, ctr, out);
}

static float u01(uint32_t r) { return (float)(r >> 8) * 5.9604644775390625e-08f; } /* [0,1), 24 bits, exact */

static uint32_t env_step(const amenv_config* cfg, env_t* e, const float* action, double* reward_out) {
  return is_v1(cfg) ? env_step_v1(cfg, e, action, reward_out) : env_step_v2(cfg, e, action, reward_out);
}


/* v1 reset: v1/rl_env_scaledObs.py:32-63.  Draw table: start = (2u0-1, 2u1-1, 1+u3); K = 1 + (r2 >> 31);
 * waypoint k = (2u(4+3k)-1, 2u(5+3k)-1, 1+2u(6+3k)), k = 0..K-1 (capped by cfg num_waypoints, the storage bound). */
static void env_reset_v1(const amenv_config* cfg, int64_t gid, env_t* e) {
  uint32_t r[12];
  for (uint32_t b = 0; b < 3; b++) orc_philox(cfg->seed, (uint64_t)gid, (uint32_t)e->episode, b, &r[4 * b]);
  int K = 1 + (int)(r[2] >> 31);                                   /* :38 randint(1,3) */
  if (K > cfg->task.num_waypoints) K = cfg->task.num_waypoints;
  memset(e->s, 0, sizeof(e->s));
  e->s[0] = fmaf(2.0f, u01(r[0]), -1.0f); e->s[1] = fmaf(2.0f, u01(r[1]), -1.0f);  /* :36 */
  e->s[2] = fmaf(1.0f, u01(r[3]), 1.0f);                           /* :37 */
  e->s[6] = 1.0;                                                   /* :40 attitude (0,0,0) */
  for (int k = 0; k < AMENV_MAX_WAYPOINTS; k++) for (int c = 0; c < 3; c++) e->wp[k][c] = 0.0;
  for (int k = 0; k < K; k++) {                                    /* :53-63 */
    e->wp[k][0] = fmaf(2.0f, u01(r[4 + 3 * k]), -1.0f);
    e->wp[k][1] = fmaf(2.0f, u01(r[5 + 3 * k]), -1.0f);
    e->wp[k][2] = fmaf(2.0f, u01(r[6 + 3 * k]), 1.0f);
  }
  e->k_env = K;
  e->final_yaw = 0.0; e->last_distance = -1.0; e->ep_return = 0.0;
  e->step = 0; e->counter = 0; e->wp_index = 0; e->fwr = 0; e->counter_activated = 0;
  e->episode += 1;
}

static void env_reset(const amenv_config* cfg, int64_t gid, env_t* e) {
  if (is_v1(cfg)) { env_reset_v1(cfg, gid, e); return; }
  e->k_env = 0;
  const int K = cfg->task.num_waypoints;
  uint32_t r[12];
  for (uint32_t b = 0; b < 3; b++) orc_philox(cfg->seed, (uint64_t)gid, (uint32_t)e->episode, b, &r[4 * b]);
  const float PIF = 3.14159274101257324f; /* (float)pi */
  float start[3], end[3];
  start[0] = fmaf(2.0f, u01(r[0]), -1.0f);                /* :44 uniform(-1,1,3) */
  start[1] = fmaf(2.0f, u01(r[1]), -1.0f);
  start[2] = fmaf(1.0f, u01(r[3]), 1.0f);                 /* :45 z = uniform(1,2) */
  const float sel0 = u01(r[4]), sel1 = u01(r[5]);         /* :63,65 */
  end[0] = fmaf(2.0f, u01(r[6]), -1.0f);                  /* utils.py:15-16 / 37-38 */
  end[1] = fmaf(2.0f, u01(r[7]), -1.0f);
  end[2] = fmaf(2.5f, u01(r[9]), 0.5f);
  const uint32_t axis = r[10] % 3u;                       /* utils.py:39 randint(0,3): 0->z 1->y 2->x */
  const float fyaw = fmaf(2.0f * PIF, u01(r[11]), -PIF);  /* :72,94-96 */
  const int kind = sel0 < 0.3f ? 0 : (sel1 < 0.6f ? 1 : 2);
  for (int k = 1; k <= K; k++) {
    float w[3];
    if (kind == 2) {                                      /* helical, utils.py:61-95 */
      w[0] = fmaf(0.8f, (float)cfg->task.traj_cos[k - 1], start[0]);
      w[1] = fmaf(0.8f, (float)cfg->task.traj_sin[k - 1], start[1]);
      w[2] = fmaf((float)k, 0.4f, start[2]);
      w[2] = fmaxf(w[2], 0.2f);
    } else {
      const float t = (float)k / (float)K;                /* utils.py:19-22 / 41-55 */
      for (int c = 0; c < 3; c++) w[c] = fmaf(t, end[c] - start[c], start[c]);
      if (kind == 1) {
        const int c = axis == 0 ? 2 : (axis == 1 ? 1 : 0);
        w[c] = w[c] + (float)cfg->task.traj_sin[k - 1];
        w[2] = fmaxf(w[2], 0.2f);
      }
    }
    for (int c = 0; c < 3; c++) e->wp[k - 1][c] = (double)w[c];
  }
  memset(e->s, 0, sizeof(e->s));
  e->s[0] = start[0]; e->s[1] = start[1]; e->s[2] = start[2];
  e->s[6] = 1.0;                                           /* quadcopter.py:28-38 with attitude (0,0,0) (:52) */
  e->final_yaw = (double)fyaw;
  e->last_distance = -1.0;                                 /* :76 None */
  e->ep_return = 0.0;
  e->step = 0; e->counter = 0; e->wp_index = 0; e->fwr = 0; e->counter_activated = 0; /* :55-59,74-77 */
  e->episode += 1;
}

/* ------------------------------------------------------------------------------------
 * Batched entry points (SoA blobs, same layout as libamenv's get/set_state)
 * ---------------------------------------------------------------------------------- */
int orc_reset(const amenv_config* cfg, double* fstate, int32_t* istate, const uint8_t* mask, float* obs_out) {
  const int n = cfg->num_envs;
  const int od = obs_dim(cfg);
  for (int i = 0; i < n; i++) {
    env_t e; load_env(cfg, n, fstate, istate, i, &e);
    if (!mask || mask[i]) { env_reset(cfg, cfg->env_id_offset + i, &e); store_env(cfg, n, fstate, istate, i, &e); }
    if (obs_out) observe(cfg, &e, obs_out + (size_t)i * od);
  }
  return 0;
}

int orc_observe(const amenv_config* cfg, const double* fstate, const int32_t* istate, float* obs_out) {
  const int n = cfg->num_envs;
  for (int i = 0; i < n; i++) { env_t e; load_env(cfg, n, fstate, istate, i, &e); observe(cfg, &e, obs_out + (size_t)i * obs_dim(cfg)); }
  return 0;
}

/* step + (optional) SB3 DummyVecEnv/Monitor semantics: on done, terminal_obs = obs,
 * ep_return/ep_len reported, env reset, obs replaced by the reset observation. */
int orc_step(const amenv_config* cfg, double* fstate, int32_t* istate, const float* actions, float* obs, double* reward,
             uint8_t* done, uint32_t* info_bits, float* terminal_obs, float* ep_return, int32_t* ep_len, int nthreads) {
  const int n = cfg->num_envs;
  const int od = obs_dim(cfg), ad = act_dim(cfg);
#ifdef _OPENMP
  if (nthreads > 0) omp_set_num_threads(nthreads);
#pragma omp parallel for schedule(static) if (nthreads != 1)
#endif
  for (int i = 0; i < n; i++) {
    env_t e; load_env(cfg, n, fstate, istate, i, &e);
    double r; uint32_t bits = env_step(cfg, &e, actions + (size_t)i * ad, &r);
    e.ep_return += r;
    const int d = (bits & (AMENV_INFO_TERMINATED | AMENV_INFO_TRUNCATED)) != 0;
    float* o = obs + (size_t)i * od;
    observe(cfg, &e, o);
    if (d) {
      if (terminal_obs) memcpy(terminal_obs + (size_t)i * od, o, sizeof(float) * od);
      if (ep_return) ep_return[i] = (float)e.ep_return;
      if (ep_len) ep_len[i] = e.step;
      if (cfg->flags & AMENV_FLAG_AUTO_RESET) { env_reset(cfg, cfg->env_id_offset + i, &e); observe(cfg, &e, o); bits |= AMENV_INFO_WAS_RESET; }
    }
    store_env(cfg, n, fstate, istate, i, &e);
    reward[i] = r; done[i] = (uint8_t)d; info_bits[i] = bits;
  }
  return 0;
}

/* T open-loop steps for every env (actions [T,N,4]); outputs only the last step.  Used as the
 * CPU baseline engine in bench.py (per-env state stays in cache across the T steps). */
int orc_rollout(const amenv_config* cfg, double* fstate, int32_t* istate, int T, const float* actions, double* reward_sum, int nthreads) {
  const int n = cfg->num_envs;
#ifdef _OPENMP
  if (nthreads > 0) omp_set_num_threads(nthreads);
#pragma omp parallel for schedule(static) if (nthreads != 1)
#endif
  for (int i = 0; i < n; i++) {
    env_t e; load_env(cfg, n, fstate, istate, i, &e);
    double acc = 0; float o[32];
    for (int t = 0; t < T; t++) {
      double r; uint32_t bits = env_step(cfg, &e, actions + ((size_t)t * n + i) * act_dim(cfg), &r);
      e.ep_return += r; acc += r;
      observe(cfg, &e, o);
      if ((bits & (AMENV_INFO_TERMINATED | AMENV_INFO_TRUNCATED)) && (cfg->flags & AMENV_FLAG_AUTO_RESET)) env_reset(cfg, cfg->env_id_offset + i, &e);
    }
    store_env(cfg, n, fstate, istate, i, &e);
    if (reward_sum) reward_sum[i] = acc + o[0] * 0.0;
  }
  return 0;
}

/* tool-point positions of every env (forward kinematics), out [N,3] */
int orc_ee_positions(const amenv_config* cfg, const double* fstate, double* out) {
  const int n = cfg->num_envs;
  int32_t* zi = (int32_t*)calloc((size_t)AMENV_I_NFIELDS * n, sizeof(int32_t));
  if (!zi) return -1;
  for (int i = 0; i < n; i++) { env_t e; load_env(cfg, n, fstate, zi, i, &e); orc_ee_position(cfg, e.s, out + (size_t)3 * i); }
  free(zi);
  return 0;
}

int orc_max_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}
