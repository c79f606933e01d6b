/* mjstep_oracle.c -- TEST INFRASTRUCTURE ONLY (parity oracle + CPU baseline).
 *
 * A plain-C, fp64, single-environment restatement of the arithmetic behind the
 * reference's hot path: dm_control.mujoco.Physics.step()
 * (dm_control/mujoco/engine.py:147-176) -> mujoco.mj_step / mj_step1 / mj_step2
 * / mj_forward (engine.py:156,158,160,162,176,343).
 *
 * The arithmetic itself lives in the third-party dependency `mujoco==3.11.0`
 * (requirements.txt:9, setup.py:194), whose source is NOT under /root/reference
 * and whose wheel cannot be installed here.  This file therefore restates
 * MuJoCo's published algorithm (computation chapter of the MuJoCo docs; stage
 * list in SURVEY.md Appendix A) stage by stage, and is anchored on what the
 * reference itself pins at the boundary: the step1/step2 split documented in
 * engine.py:147-162,335-341, the default constants in dm_control/mjcf/schema.xml,
 * the quaternion conventions in dm_control/utils/transformations.py, and the
 * known-answer tests of SURVEY.md 8(c) (tests/test_oracle_kat.py), including the
 * one numeric golden the reference holds: dm_control/mujoco/README.md:46-49.
 *
 * PARITY UNPINNED at trajectory level: the reference holds no golden qpos
 * vectors and MuJoCo itself cannot be run here.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * this library.  The product path (dm_control_amd/csrc) never links it.
 *
 * Conventions: quaternions wxyz; free joint qvel = [v_world(3), omega_local(3)];
 * spatial vectors are [angular(3), linear(3)] expressed in a frame located at
 * the subtree COM of the kinematic-tree root and aligned with the world.
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../include/dmc_model_layout.h"

#define MINVAL DMC_MINVAL
#define MAXVAL DMC_MAXVAL
#define mjMAX(a, b) ((a) > (b) ? (a) : (b))
#define mjMIN(a, b) ((a) < (b) ? (a) : (b))

/* ------------------------------------------------------------------------- */
/* model / data                                                               */
/* ------------------------------------------------------------------------- */
typedef struct {
#define X(n) int n;
  DMC_MODEL_HEADER_INTS(X)
#undef X
#define X(n) double n;
  DMC_MODEL_HEADER_REALS(X)
#undef X
#define X(n, c) int* n;
  DMC_MODEL_INT_FIELDS(X)
#undef X
#define X(n, c) double* n;
  DMC_MODEL_REAL_FIELDS(X)
#undef X
  int* ibuf;
  double* rbuf;
  int nconmax, njmax;
} Model;

enum { CT_LIMIT = 0, CT_FRICTIONLESS = 1, CT_PYRAMIDAL = 2, CT_ELLIPTIC = 3, CT_FRICTION_DOF = 4, CT_LIMIT_TENDON = 5, CT_EQUALITY = 6 };

typedef struct {
  double dist, pos[3], frame[9], includemargin, friction[5], solref[2], solimp[5], mu;
  double H[36]; /* elliptic cone Hessian of the middle zone (dim x dim) */
  int dim, geom1, geom2, efc_address, exclude;
} Contact;

typedef struct {
  /* state */
  double time;
  double *qpos, *qvel, *act, *act_dot, *qacc_warmstart, *ctrl, *qfrc_applied, *xfrc_applied;
  double *mocap_pos, *mocap_quat; /* (nmocap, 3 / 4): poses of the mocap bodies, inputs like ctrl */
  /* position-dependent */
  double *xpos, *xquat, *xmat, *xipos, *ximat, *xanchor, *xaxis;
  double *geom_xpos, *geom_xmat, *site_xpos, *site_xmat;
  double *subtree_com, *cinert, *cdof, *crb, *qM, *qL; /* qL: dense Cholesky of M */
  /* velocity-dependent */
  double *cvel, *cdof_dot, *qfrc_bias, *qfrc_passive, *subtree_linvel;
  double *actuator_length, *actuator_velocity;
  double *ten_length, *ten_J; /* tendon lengths and their Jacobians (ntendon x nv) */
  /* acceleration */
  double *actuator_force, *qfrc_actuator, *qfrc_smooth, *qacc_smooth, *qacc;
  double *qfrc_constraint, *cacc, *cfrc_int, *cfrc_ext;
  double *sensordata;
  /* contacts & constraints */
  int ncon, nefc;
  Contact* contact;
  int *efc_type, *efc_id, *efc_state;
  double *efc_J, *efc_pos, *efc_margin, *efc_diagApprox, *efc_R, *efc_D, *efc_aref;
  double *efc_vel, *efc_force, *efc_KBIP;
  /* solver stats */
  int solver_iter;
  int nisland;                 /* constraint islands of the last mj_fwdConstraint (0: solved as one system) */
  int *dof_island, *efc_island; /* island of each dof / constraint row, -1: none (mjData.dof_island / efc_island) */
  void* island_scratch;        /* a second Data the per-island sub-problems are assembled in (solve_islands) */
  int warning[DMC_NWARNING];
  /* scratch */
  double *w_Jaref, *w_Jv, *w_Ma, *w_Mv, *w_grad, *w_Mgrad, *w_search, *w_quad, *w_H, *w_tmp;
  double* mem;
} Data;

/* ------------------------------------------------------------------------- */
/* small math                                                                 */
/* ------------------------------------------------------------------------- */
static double dot3(const double* a, const double* b) { return a[0]*b[0] + a[1]*b[1] + a[2]*b[2]; }
static void cross3(double* r, const double* a, const double* b) {
  double t0 = a[1]*b[2] - a[2]*b[1], t1 = a[2]*b[0] - a[0]*b[2], t2 = a[0]*b[1] - a[1]*b[0];
  r[0] = t0; r[1] = t1; r[2] = t2;
}
static double normalize3(double* v) {
  double n = sqrt(dot3(v, v));
  if (n < MINVAL) { v[0] = 1; v[1] = 0; v[2] = 0; }
  else { double s = 1 / n; v[0] *= s; v[1] *= s; v[2] *= s; }
  return n;
}
static double normalize4(double* q) {
  double n = sqrt(q[0]*q[0] + q[1]*q[1] + q[2]*q[2] + q[3]*q[3]);
  if (n < MINVAL) { q[0] = 1; q[1] = q[2] = q[3] = 0; }
  else if (fabs(n - 1) > MINVAL) { double s = 1 / n; q[0] *= s; q[1] *= s; q[2] *= s; q[3] *= s; }
  return n;
}
static void mul_quat(double* r, const double* a, const double* b) {
  double t[4] = {a[0]*b[0] - a[1]*b[1] - a[2]*b[2] - a[3]*b[3],
                 a[0]*b[1] + a[1]*b[0] + a[2]*b[3] - a[3]*b[2],
                 a[0]*b[2] - a[1]*b[3] + a[2]*b[0] + a[3]*b[1],
                 a[0]*b[3] + a[1]*b[2] - a[2]*b[1] + a[3]*b[0]};
  memcpy(r, t, sizeof t);
}
static void quat2mat(double* m, const double* q) {
  double q00 = q[0]*q[0], q01 = q[0]*q[1], q02 = q[0]*q[2], q03 = q[0]*q[3];
  double q11 = q[1]*q[1], q12 = q[1]*q[2], q13 = q[1]*q[3];
  double q22 = q[2]*q[2], q23 = q[2]*q[3], q33 = q[3]*q[3];
  m[0] = q00 + q11 - q22 - q33; m[4] = q00 - q11 + q22 - q33; m[8] = q00 - q11 - q22 + q33;
  m[1] = 2*(q12 - q03); m[2] = 2*(q13 + q02);
  m[3] = 2*(q12 + q03); m[5] = 2*(q23 - q01);
  m[6] = 2*(q13 - q02); m[7] = 2*(q23 + q01);
}
static void rot_vec_quat(double* r, const double* v, const double* q) {
  double m[9]; quat2mat(m, q);
  double t0 = m[0]*v[0] + m[1]*v[1] + m[2]*v[2];
  double t1 = m[3]*v[0] + m[4]*v[1] + m[5]*v[2];
  double t2 = m[6]*v[0] + m[7]*v[1] + m[8]*v[2];
  r[0] = t0; r[1] = t1; r[2] = t2;
}
static void mul_mat_vec3(double* r, const double* m, const double* v) {
  double t0 = m[0]*v[0] + m[1]*v[1] + m[2]*v[2];
  double t1 = m[3]*v[0] + m[4]*v[1] + m[5]*v[2];
  double t2 = m[6]*v[0] + m[7]*v[1] + m[8]*v[2];
  r[0] = t0; r[1] = t1; r[2] = t2;
}
static void mul_matT_vec3(double* r, const double* m, const double* v) {
  double t0 = m[0]*v[0] + m[3]*v[1] + m[6]*v[2];
  double t1 = m[1]*v[0] + m[4]*v[1] + m[7]*v[2];
  double t2 = m[2]*v[0] + m[5]*v[1] + m[8]*v[2];
  r[0] = t0; r[1] = t1; r[2] = t2;
}
static void axisangle2quat(double* q, const double* axis, double angle) {
  if (angle == 0) { q[0] = 1; q[1] = q[2] = q[3] = 0; return; }
  double s = sin(angle * 0.5);
  q[0] = cos(angle * 0.5); q[1] = axis[0]*s; q[2] = axis[1]*s; q[3] = axis[2]*s;
}
static void quat_integrate(double* quat, const double* vel, double scale) {
  double tmp[3] = {vel[0], vel[1], vel[2]}, qrot[4];
  double angle = scale * normalize3(tmp);
  axisangle2quat(qrot, tmp, angle);
  normalize4(quat);
  mul_quat(quat, quat, qrot);
}
/* spatial algebra */
static void inert_com(double* res, const double* inert, const double* mat, const double* dif, double mass) {
  double tmp[9];
  for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) tmp[3*r + c] = mat[3*r + c] * inert[c];
  /* res = tmp * mat' */
  res[0] = tmp[0]*mat[0] + tmp[1]*mat[1] + tmp[2]*mat[2];
  res[1] = tmp[3]*mat[3] + tmp[4]*mat[4] + tmp[5]*mat[5];
  res[2] = tmp[6]*mat[6] + tmp[7]*mat[7] + tmp[8]*mat[8];
  res[3] = tmp[0]*mat[3] + tmp[1]*mat[4] + tmp[2]*mat[5];
  res[4] = tmp[0]*mat[6] + tmp[1]*mat[7] + tmp[2]*mat[8];
  res[5] = tmp[3]*mat[6] + tmp[4]*mat[7] + tmp[5]*mat[8];
  res[0] += mass * (dif[1]*dif[1] + dif[2]*dif[2]);
  res[1] += mass * (dif[0]*dif[0] + dif[2]*dif[2]);
  res[2] += mass * (dif[0]*dif[0] + dif[1]*dif[1]);
  res[3] -= mass * dif[0]*dif[1];
  res[4] -= mass * dif[0]*dif[2];
  res[5] -= mass * dif[1]*dif[2];
  res[6] = mass*dif[0]; res[7] = mass*dif[1]; res[8] = mass*dif[2];
  res[9] = mass;
}
static void mul_inert_vec(double* res, const double* i, const double* v) {
  res[0] = i[0]*v[0] + i[3]*v[1] + i[4]*v[2] - i[8]*v[4] + i[7]*v[5];
  res[1] = i[3]*v[0] + i[1]*v[1] + i[5]*v[2] + i[8]*v[3] - i[6]*v[5];
  res[2] = i[4]*v[0] + i[5]*v[1] + i[2]*v[2] - i[7]*v[3] + i[6]*v[4];
  res[3] = i[8]*v[1] - i[7]*v[2] + i[9]*v[3];
  res[4] = i[6]*v[2] - i[8]*v[0] + i[9]*v[4];
  res[5] = i[7]*v[0] - i[6]*v[1] + i[9]*v[5];
}
static void cross_motion(double* res, const double* vel, const double* v) {
  res[0] = -vel[2]*v[1] + vel[1]*v[2];
  res[1] =  vel[2]*v[0] - vel[0]*v[2];
  res[2] = -vel[1]*v[0] + vel[0]*v[1];
  res[3] = -vel[2]*v[4] + vel[1]*v[5];
  res[4] =  vel[2]*v[3] - vel[0]*v[5];
  res[5] = -vel[1]*v[3] + vel[0]*v[4];
  res[3] += -vel[5]*v[1] + vel[4]*v[2];
  res[4] +=  vel[5]*v[0] - vel[3]*v[2];
  res[5] += -vel[4]*v[0] + vel[3]*v[1];
}
static void cross_force(double* res, const double* vel, const double* f) {
  res[0] = -vel[2]*f[1] + vel[1]*f[2];
  res[1] =  vel[2]*f[0] - vel[0]*f[2];
  res[2] = -vel[1]*f[0] + vel[0]*f[1];
  res[3] = -vel[2]*f[4] + vel[1]*f[5];
  res[4] =  vel[2]*f[3] - vel[0]*f[5];
  res[5] = -vel[1]*f[3] + vel[0]*f[4];
  res[0] += -vel[5]*f[4] + vel[4]*f[5];
  res[1] +=  vel[5]*f[3] - vel[3]*f[5];
  res[2] += -vel[4]*f[3] + vel[3]*f[4];
}
static void dof_com(double* res, const double* axis, const double* offset) {
  if (offset) { res[0] = axis[0]; res[1] = axis[1]; res[2] = axis[2]; cross3(res + 3, axis, offset); }
  else { res[0] = res[1] = res[2] = 0; res[3] = axis[0]; res[4] = axis[1]; res[5] = axis[2]; }
}
static double dot_n(const double* a, const double* b, int n) {
  double s = 0; for (int i = 0; i < n; i++) s += a[i]*b[i]; return s;
}
/* dense Cholesky A = L L' (lower, row-major n x n); the DIAGONAL of L holds
 * 1/L[j][j] (MuJoCo keeps the inverse diagonal too: qLDiagInv) and the strict
 * lower part L[i][j] = t * (1/L[j][j]); returns rank deficiency */
static int chol_factor(double* L, const double* A, int n) {
  int bad = 0;
  for (int j = 0; j < n; j++) {
    double s = A[j*n + j];
    for (int k = 0; k < j; k++) s -= L[j*n + k]*L[j*n + k];
    if (s < MINVAL) { s = MINVAL; bad++; }
    double inv = 1 / sqrt(s);
    L[j*n + j] = inv;
    for (int i = j + 1; i < n; i++) {
      double t = A[i*n + j];
      for (int k = 0; k < j; k++) t -= L[i*n + k]*L[j*n + k];
      L[i*n + j] = t * inv;
    }
    for (int i = 0; i < j; i++) L[i*n + j] = 0;
  }
  return bad;
}
static void chol_solve(double* x, const double* L, const double* b, int n) {
  for (int i = 0; i < n; i++) {
    double s = b[i];
    for (int k = 0; k < i; k++) s -= L[i*n + k]*x[k];
    x[i] = s * L[i*n + i];
  }
  for (int i = n - 1; i >= 0; i--) {   /* k descending: the order a column-oriented parallel solve produces */
    double s = x[i];
    for (int k = n - 1; k > i; k--) s -= L[k*n + i]*x[k];
    x[i] = s * L[i*n + i];
  }
}

/* ------------------------------------------------------------------------- */
/* construction                                                               */
/* ------------------------------------------------------------------------- */
static const char* g_err = "";
const char* ora_last_error(void) { return g_err; }

Model* ora_model_create(const int32_t* ints, int nints, const double* reals, int nreals) {
  if (nints < 2 || ints[0] != (int32_t)DMC_MODEL_MAGIC || ints[1] != DMC_MODEL_VERSION) {
    g_err = "bad model blob magic/version"; return NULL;
  }
  Model* m = (Model*)calloc(1, sizeof(Model));
  m->ibuf = (int*)malloc(sizeof(int) * (size_t)nints);
  m->rbuf = (double*)malloc(sizeof(double) * (size_t)nreals);
  memcpy(m->ibuf, ints, sizeof(int) * (size_t)nints);
  memcpy(m->rbuf, reals, sizeof(double) * (size_t)nreals);
  int ip = 2, rp = 0;
#define X(n) m->n = m->ibuf[ip++];
  DMC_MODEL_HEADER_INTS(X)
#undef X
#define X(n) m->n = m->rbuf[rp++];
  DMC_MODEL_HEADER_REALS(X)
#undef X
  int nq = m->nq, nv = m->nv, nu = m->nu, nbody = m->nbody, njnt = m->njnt, ngeom = m->ngeom;
  int nsite = m->nsite, nsensor = m->nsensor, npair = m->npair, nkey = m->nkey, ntendon = m->ntendon, nwrap = m->nwrap, neq = m->neq, na = m->na, nmocap = m->nmocap;
  (void)na; (void)nmocap; (void)nq; (void)nv; (void)nu; (void)nbody; (void)njnt; (void)ngeom; (void)nsite;
  (void)nsensor; (void)npair; (void)nkey; (void)ntendon; (void)nwrap; (void)neq;
#define X(n, c) m->n = m->ibuf + ip; ip += (c);
  DMC_MODEL_INT_FIELDS(X)
#undef X
#define X(n, c) m->n = m->rbuf + rp; rp += (c);
  DMC_MODEL_REAL_FIELDS(X)
#undef X
  if (ip != nints || rp != nreals) {
    g_err = "model blob size mismatch"; free(m->ibuf); free(m->rbuf); free(m); return NULL;
  }
  m->nconmax = 4 * npair + 4;
  m->njmax = m->neq + m->nv + 2 * njnt + 2 * m->ntendon + 10 * m->nconmax;
  return m;
}
void ora_model_free(Model* m) { if (m) { free(m->ibuf); free(m->rbuf); free(m); } }
int ora_model_opt_int(Model* m, const char* name, int set, int value) {
#define OI(f) if (!strcmp(name, #f)) { if (set) m->opt_##f = value; return m->opt_##f; }
  OI(integrator) OI(cone) OI(solver) OI(iterations) OI(ls_iterations) OI(disableflags) OI(enableflags)
#undef OI
  return -1;
}
double ora_model_opt_real(Model* m, const char* name, int set, double value) {
#define OR(f) if (!strcmp(name, #f)) { if (set) m->opt_##f = value; return m->opt_##f; }
  OR(timestep) OR(gravity_x) OR(gravity_y) OR(gravity_z) OR(impratio) OR(tolerance) OR(ls_tolerance)
#undef OR
  return NAN;
}
int* ora_model_int_field(Model* m, const char* name, int* count) {
  int nq = m->nq, nv = m->nv, nu = m->nu, nbody = m->nbody, njnt = m->njnt, ngeom = m->ngeom;
  int nsite = m->nsite, nsensor = m->nsensor, npair = m->npair, nkey = m->nkey, ntendon = m->ntendon, nwrap = m->nwrap, neq = m->neq, na = m->na, nmocap = m->nmocap;
  (void)na; (void)nmocap; (void)nq; (void)nv; (void)nu; (void)nbody; (void)njnt; (void)ngeom; (void)nsite;
  (void)nsensor; (void)npair; (void)nkey; (void)ntendon; (void)nwrap; (void)neq;
#define X(n, c) if (!strcmp(name, #n)) { *count = (c); return m->n; }
  DMC_MODEL_INT_FIELDS(X)
#undef X
  return NULL;
}
double* ora_model_real_field(Model* m, const char* name, int* count) {
  int nq = m->nq, nv = m->nv, nu = m->nu, nbody = m->nbody, njnt = m->njnt, ngeom = m->ngeom;
  int nsite = m->nsite, nsensor = m->nsensor, npair = m->npair, nkey = m->nkey, ntendon = m->ntendon, nwrap = m->nwrap, neq = m->neq, na = m->na, nmocap = m->nmocap;
  (void)na; (void)nmocap; (void)nq; (void)nv; (void)nu; (void)nbody; (void)njnt; (void)ngeom; (void)nsite;
  (void)nsensor; (void)npair; (void)nkey; (void)ntendon; (void)nwrap; (void)neq;
#define X(n, c) if (!strcmp(name, #n)) { *count = (c); return m->n; }
  DMC_MODEL_REAL_FIELDS(X)
#undef X
  return NULL;
}

#define DATA_REAL_FIELDS(X) \
  X(qpos, m->nq) X(qvel, m->nv) X(act, m->na) X(act_dot, m->na) X(qacc_warmstart, m->nv) X(ctrl, m->nu) \
  X(qfrc_applied, m->nv) X(xfrc_applied, 6*m->nbody) X(mocap_pos, 3*m->nmocap) X(mocap_quat, 4*m->nmocap) \
  X(xpos, 3*m->nbody) X(xquat, 4*m->nbody) X(xmat, 9*m->nbody) X(xipos, 3*m->nbody) \
  X(ximat, 9*m->nbody) X(xanchor, 3*m->njnt) X(xaxis, 3*m->njnt) \
  X(geom_xpos, 3*m->ngeom) X(geom_xmat, 9*m->ngeom) X(site_xpos, 3*m->nsite) X(site_xmat, 9*m->nsite) \
  X(subtree_com, 3*m->nbody) X(cinert, 10*m->nbody) X(cdof, 6*m->nv) X(crb, 10*m->nbody) \
  X(qM, m->nv*m->nv) X(qL, m->nv*m->nv) \
  X(cvel, 6*m->nbody) X(cdof_dot, 6*m->nv) X(qfrc_bias, m->nv) X(qfrc_passive, m->nv) \
  X(subtree_linvel, 3*m->nbody) X(actuator_length, m->nu) X(actuator_velocity, m->nu) \
  X(ten_length, m->ntendon) X(ten_J, m->ntendon*m->nv) \
  X(actuator_force, m->nu) X(qfrc_actuator, m->nv) X(qfrc_smooth, m->nv) X(qacc_smooth, m->nv) \
  X(qacc, m->nv) X(qfrc_constraint, m->nv) X(cacc, 6*m->nbody) X(cfrc_int, 6*m->nbody) \
  X(cfrc_ext, 6*m->nbody) X(sensordata, m->nsensordata) \
  X(efc_J, m->njmax*m->nv) X(efc_pos, m->njmax) X(efc_margin, m->njmax) \
  X(efc_diagApprox, m->njmax) X(efc_R, m->njmax) X(efc_D, m->njmax) X(efc_aref, m->njmax) \
  X(efc_vel, m->njmax) X(efc_force, m->njmax) X(efc_KBIP, 4*m->njmax) \
  X(w_Jaref, m->njmax) X(w_Jv, m->njmax) X(w_Ma, m->nv) X(w_Mv, m->nv) X(w_grad, m->nv) \
  X(w_Mgrad, m->nv) X(w_search, m->nv) X(w_quad, 3*m->njmax) X(w_H, 2*m->nv*m->nv) \
  X(w_tmp, 16*(m->nv + m->nbody + 8))

void ora_reset(const Model* m, Data* d, int key);

Data* ora_data_create(const Model* m) {
  Data* d = (Data*)calloc(1, sizeof(Data));
  size_t total = 0;
#define X(n, c) total += (size_t)(c) + 1;
  DATA_REAL_FIELDS(X)
#undef X
  d->mem = (double*)calloc(total, sizeof(double));
  double* p = d->mem;
#define X(n, c) d->n = p; p += (size_t)(c) + 1;
  DATA_REAL_FIELDS(X)
#undef X
  d->contact = (Contact*)calloc((size_t)m->nconmax + 1, sizeof(Contact));
  d->efc_type = (int*)calloc((size_t)m->njmax + 1, sizeof(int));
  d->efc_id = (int*)calloc((size_t)m->njmax + 1, sizeof(int));
  d->efc_state = (int*)calloc((size_t)m->njmax + 1, sizeof(int));
  d->dof_island = (int*)calloc((size_t)m->nv + 1, sizeof(int));
  d->efc_island = (int*)calloc((size_t)m->njmax + 1, sizeof(int));
  ora_reset(m, d, -1);
  return d;
}
void ora_data_free(Data* d) {
  if (!d) return;
  if (d->island_scratch) ora_data_free((Data*)d->island_scratch);
  free(d->mem); free(d->contact); free(d->efc_type); free(d->efc_id); free(d->efc_state); free(d->dof_island); free(d->efc_island); free(d);
}
void ora_data_copy(const Model* m, Data* dst, const Data* src) {
  size_t total = 0;
#define X(n, c) total += (size_t)(c) + 1;
  DATA_REAL_FIELDS(X)
#undef X
  memcpy(dst->mem, src->mem, total * sizeof(double));
  memcpy(dst->contact, src->contact, ((size_t)m->nconmax + 1) * sizeof(Contact));
  memcpy(dst->efc_type, src->efc_type, ((size_t)m->njmax + 1) * sizeof(int));
  memcpy(dst->efc_id, src->efc_id, ((size_t)m->njmax + 1) * sizeof(int));
  memcpy(dst->efc_state, src->efc_state, ((size_t)m->njmax + 1) * sizeof(int));
  dst->time = src->time; dst->ncon = src->ncon; dst->nefc = src->nefc;
  dst->solver_iter = src->solver_iter;
  memcpy(dst->warning, src->warning, sizeof src->warning);
}
double* ora_data_field(const Model* m, Data* d, const char* name, int* count) {
#define X(n, c) if (!strcmp(name, #n)) { *count = (c); return d->n; }
  DATA_REAL_FIELDS(X)
#undef X
  if (!strcmp(name, "time")) { *count = 1; return &d->time; }
  return NULL;
}
int ora_data_int(const Data* d, const char* name) {
  if (!strcmp(name, "ncon")) return d->ncon;
  if (!strcmp(name, "nefc")) return d->nefc;
  if (!strcmp(name, "solver_iter")) return d->solver_iter;
  if (!strcmp(name, "nisland")) return d->nisland;
  return -1;
}
int* ora_data_warning(Data* d) { return d->warning; }
int* ora_data_efc_int(Data* d, const char* name) {
  if (!strcmp(name, "efc_type")) return d->efc_type;
  if (!strcmp(name, "efc_id")) return d->efc_id;
  if (!strcmp(name, "efc_state")) return d->efc_state;
  if (!strcmp(name, "efc_island")) return d->efc_island;
  if (!strcmp(name, "dof_island")) return d->dof_island;
  return NULL;
}
/* contact i -> out[0..29]: dist, pos3, frame9, includemargin, friction5, solref2,
 * solimp5, dim, geom1, geom2, efc_address */
void ora_data_contact(const Data* d, int i, double* out) {
  const Contact* c = d->contact + i;
  int k = 0;
  out[k++] = c->dist;
  for (int j = 0; j < 3; j++) out[k++] = c->pos[j];
  for (int j = 0; j < 9; j++) out[k++] = c->frame[j];
  out[k++] = c->includemargin;
  for (int j = 0; j < 5; j++) out[k++] = c->friction[j];
  for (int j = 0; j < 2; j++) out[k++] = c->solref[j];
  for (int j = 0; j < 5; j++) out[k++] = c->solimp[j];
  out[k++] = c->dim; out[k++] = c->geom1; out[k++] = c->geom2; out[k++] = c->efc_address;
}

/* mj_resetData / mj_resetDataKeyframe (engine.py:318,323) */
void ora_reset(const Model* m, Data* d, int key) {
  size_t total = 0;
#define X(n, c) total += (size_t)(c) + 1;
  DATA_REAL_FIELDS(X)
#undef X
  memset(d->mem, 0, total * sizeof(double));
  d->time = 0; d->ncon = 0; d->nefc = 0; d->solver_iter = 0;
  memset(d->warning, 0, sizeof d->warning);
  memcpy(d->qpos, m->qpos0, sizeof(double) * (size_t)m->nq);
  for (int i = 0; i < m->nbody; i++) if (m->body_mocapid[i] >= 0) {      /* mj_resetData: mocap poses from the model */
    memcpy(d->mocap_pos + 3*m->body_mocapid[i], m->body_pos + 3*i, 3 * sizeof(double));
    memcpy(d->mocap_quat + 4*m->body_mocapid[i], m->body_quat + 4*i, 4 * sizeof(double));
  }
  for (int i = 0; i < m->nbody; i++) { d->xquat[4*i] = 1; d->xmat[9*i] = d->xmat[9*i+4] = d->xmat[9*i+8] = 1;
    d->ximat[9*i] = d->ximat[9*i+4] = d->ximat[9*i+8] = 1; }
  if (key >= 0 && key < m->nkey) {
    memcpy(d->qpos, m->key_qpos + (size_t)key*m->nq, sizeof(double) * (size_t)m->nq);
    memcpy(d->qvel, m->key_qvel + (size_t)key*m->nv, sizeof(double) * (size_t)m->nv);
    memcpy(d->ctrl, m->key_ctrl + (size_t)key*m->nu, sizeof(double) * (size_t)m->nu);
    d->time = m->key_time[key];
    memcpy(d->act, m->key_act + (size_t)key*m->na, sizeof(double) * (size_t)m->na);
    memcpy(d->mocap_pos, m->key_mpos + (size_t)key*3*m->nmocap, sizeof(double) * 3 * (size_t)m->nmocap);
    memcpy(d->mocap_quat, m->key_mquat + (size_t)key*4*m->nmocap, sizeof(double) * 4 * (size_t)m->nmocap);
  }
}

/* ------------------------------------------------------------------------- */
/* position stage                                                             */
/* ------------------------------------------------------------------------- */
static void kinematics(const Model* m, Data* d) {
  d->xpos[0] = d->xpos[1] = d->xpos[2] = 0;
  d->xquat[0] = 1; d->xquat[1] = d->xquat[2] = d->xquat[3] = 0;
  d->xipos[0] = d->xipos[1] = d->xipos[2] = 0;
  for (int k = 0; k < 9; k++) d->xmat[k] = d->ximat[k] = (k % 4 == 0);
  for (int i = 1; i < m->nbody; i++) {
    double xpos[3], xquat[4];
    int jntadr = m->body_jntadr[i], jntnum = m->body_jntnum[i];
    if (jntnum == 1 && m->jnt_type[jntadr] == DMC_JNT_FREE) {
      int qa = m->jnt_qposadr[jntadr];
      memcpy(xpos, d->qpos + qa, 3 * sizeof(double));
      memcpy(xquat, d->qpos + qa + 3, 4 * sizeof(double));
      normalize4(xquat);
      memcpy(d->xanchor + 3*jntadr, xpos, 3 * sizeof(double));
      memcpy(d->xaxis + 3*jntadr, m->jnt_axis + 3*jntadr, 3 * sizeof(double));
    } else {
      int pid = m->body_parentid[i];
      if (m->body_mocapid[i] >= 0) {      /* a mocap body: its pose is data, not model (mj_kinematics) */
        memcpy(xpos, d->mocap_pos + 3*m->body_mocapid[i], 3 * sizeof(double));
        memcpy(xquat, d->mocap_quat + 4*m->body_mocapid[i], 4 * sizeof(double));
        normalize4(xquat);
      } else {
        mul_mat_vec3(xpos, d->xmat + 9*pid, m->body_pos + 3*i);
        xpos[0] += d->xpos[3*pid]; xpos[1] += d->xpos[3*pid+1]; xpos[2] += d->xpos[3*pid+2];
        mul_quat(xquat, d->xquat + 4*pid, m->body_quat + 4*i);
      }
      for (int j = jntadr; j < jntadr + jntnum; j++) {
        int qa = m->jnt_qposadr[j];
        double* anchor = d->xanchor + 3*j; double* axis = d->xaxis + 3*j;
        rot_vec_quat(axis, m->jnt_axis + 3*j, xquat);
        rot_vec_quat(anchor, m->jnt_pos + 3*j, xquat);
        anchor[0] += xpos[0]; anchor[1] += xpos[1]; anchor[2] += xpos[2];
        int t = m->jnt_type[j];
        if (t == DMC_JNT_SLIDE) {
          double q = d->qpos[qa] - m->qpos0[qa];
          xpos[0] += axis[0]*q; xpos[1] += axis[1]*q; xpos[2] += axis[2]*q;
        } else if (t == DMC_JNT_BALL || t == DMC_JNT_HINGE) {
          double qloc[4], vec[3];
          if (t == DMC_JNT_BALL) { memcpy(qloc, d->qpos + qa, 4 * sizeof(double)); normalize4(qloc); }
          else axisangle2quat(qloc, m->jnt_axis + 3*j, d->qpos[qa] - m->qpos0[qa]);
          mul_quat(xquat, xquat, qloc);
          rot_vec_quat(vec, m->jnt_pos + 3*j, xquat);
          xpos[0] = anchor[0] - vec[0]; xpos[1] = anchor[1] - vec[1]; xpos[2] = anchor[2] - vec[2];
        }
      }
    }
    normalize4(xquat);
    memcpy(d->xpos + 3*i, xpos, sizeof xpos);
    memcpy(d->xquat + 4*i, xquat, sizeof xquat);
    quat2mat(d->xmat + 9*i, xquat);
    double v[3], q[4];
    mul_mat_vec3(v, d->xmat + 9*i, m->body_ipos + 3*i);
    d->xipos[3*i] = xpos[0] + v[0]; d->xipos[3*i+1] = xpos[1] + v[1]; d->xipos[3*i+2] = xpos[2] + v[2];
    mul_quat(q, xquat, m->body_iquat + 4*i);
    quat2mat(d->ximat + 9*i, q);
  }
  for (int g = 0; g < m->ngeom; g++) {
    int b = m->geom_bodyid[g]; double v[3], q[4];
    mul_mat_vec3(v, d->xmat + 9*b, m->geom_pos + 3*g);
    for (int k = 0; k < 3; k++) d->geom_xpos[3*g + k] = d->xpos[3*b + k] + v[k];
    mul_quat(q, d->xquat + 4*b, m->geom_quat + 4*g);
    quat2mat(d->geom_xmat + 9*g, q);
  }
  for (int s = 0; s < m->nsite; s++) {
    int b = m->site_bodyid[s]; double v[3], q[4];
    mul_mat_vec3(v, d->xmat + 9*b, m->site_pos + 3*s);
    for (int k = 0; k < 3; k++) d->site_xpos[3*s + k] = d->xpos[3*b + k] + v[k];
    mul_quat(q, d->xquat + 4*b, m->site_quat + 4*s);
    quat2mat(d->site_xmat + 9*s, q);
  }
}

static void com_pos(const Model* m, Data* d) {
  int nbody = m->nbody;
  memset(d->subtree_com, 0, sizeof(double) * 3 * (size_t)nbody);
  for (int i = nbody - 1; i >= 0; i--) {
    double* sc = d->subtree_com + 3*i;
    for (int k = 0; k < 3; k++) sc[k] += d->xipos[3*i + k] * m->body_mass[i];
    if (i) { double* pc = d->subtree_com + 3*m->body_parentid[i]; for (int k = 0; k < 3; k++) pc[k] += sc[k]; }
    if (m->body_subtreemass[i] < MINVAL) memcpy(sc, d->xipos + 3*i, 3 * sizeof(double));
    else { double s = 1.0 / mjMAX(MINVAL, m->body_subtreemass[i]); sc[0] *= s; sc[1] *= s; sc[2] *= s; }
  }
  memset(d->cinert, 0, 10 * sizeof(double));
  for (int i = 1; i < nbody; i++) {
    double off[3]; const double* rc = d->subtree_com + 3*m->body_rootid[i];
    for (int k = 0; k < 3; k++) off[k] = d->xipos[3*i + k] - rc[k];
    inert_com(d->cinert + 10*i, m->body_inertia + 3*i, d->ximat + 9*i, off, m->body_mass[i]);
  }
  for (int j = 0; j < m->njnt; j++) {
    int da = 6 * m->jnt_dofadr[j], bi = m->jnt_bodyid[j], skip = 0;
    double off[3], axis[3]; const double* rc = d->subtree_com + 3*m->body_rootid[bi];
    for (int k = 0; k < 3; k++) off[k] = rc[k] - d->xanchor[3*j + k];
    switch (m->jnt_type[j]) {
      case DMC_JNT_FREE:
        for (int i = 0; i < 3; i++) { axis[0] = axis[1] = axis[2] = 0; axis[i] = 1; dof_com(d->cdof + da + 6*i, axis, NULL); }
        skip = 3; /* fallthrough */
      case DMC_JNT_BALL:
        for (int i = 0; i < 3; i++) {
          axis[0] = d->xmat[9*bi + i]; axis[1] = d->xmat[9*bi + 3 + i]; axis[2] = d->xmat[9*bi + 6 + i];
          dof_com(d->cdof + da + 6*(i + skip), axis, off);
        }
        break;
      case DMC_JNT_SLIDE: dof_com(d->cdof + da, d->xaxis + 3*j, NULL); break;
      case DMC_JNT_HINGE: dof_com(d->cdof + da, d->xaxis + 3*j, off); break;
    }
  }
}

static void crb(const Model* m, Data* d) {
  int nbody = m->nbody, nv = m->nv;
  memcpy(d->crb, d->cinert, sizeof(double) * 10 * (size_t)nbody);
  for (int i = nbody - 1; i > 0; i--) if (m->body_parentid[i] > 0)
    for (int k = 0; k < 10; k++) d->crb[10*m->body_parentid[i] + k] += d->crb[10*i + k];
  memset(d->qM, 0, sizeof(double) * (size_t)nv * (size_t)nv);
  for (int i = 0; i < nv; i++) {
    double buf[6];
    mul_inert_vec(buf, d->crb + 10*m->dof_bodyid[i], d->cdof + 6*i);
    for (int j = i; j >= 0; j = m->dof_parentid[j]) {
      double v = dot_n(d->cdof + 6*j, buf, 6);
      if (j == i) v += m->dof_armature[i];
      d->qM[i*nv + j] = v; d->qM[j*nv + i] = v;
    }
  }
  chol_factor(d->qL, d->qM, nv);
}

/* ---- collision ----------------------------------------------------------- */
static void make_frame(double* f) {
  normalize3(f);
  if (sqrt(dot3(f + 3, f + 3)) < 0.5) {
    f[3] = f[4] = f[5] = 0;
    if (f[1] < 0.5 && f[1] > -0.5) f[4] = 1; else f[5] = 1;
  }
  double t = dot3(f, f + 3);
  f[3] -= t*f[0]; f[4] -= t*f[1]; f[5] -= t*f[2];
  normalize3(f + 3);
  cross3(f + 6, f, f + 3);
}
static int raw_plane_sphere(Contact* c, double margin, const double* ppos, const double* nrm,
                            const double* spos, double radius) {
  double dif[3] = {spos[0] - ppos[0], spos[1] - ppos[1], spos[2] - ppos[2]};
  double dist = dot3(dif, nrm) - radius;
  if (dist > margin) return 0;
  c->dist = dist;
  for (int k = 0; k < 3; k++) { c->pos[k] = spos[k] - nrm[k]*(radius + dist*0.5); c->frame[k] = nrm[k]; c->frame[3 + k] = 0; }
  return 1;
}
static int raw_sphere_sphere(Contact* c, double margin, const double* p1, double r1, const double* p2, double r2) {
  double dif[3] = {p2[0] - p1[0], p2[1] - p1[1], p2[2] - p1[2]};
  double cdist = sqrt(dot3(dif, dif));
  double dist = cdist - r1 - r2;
  if (dist > margin) return 0;
  double n[3];
  if (cdist < MINVAL) { n[0] = 1; n[1] = n[2] = 0; } else { n[0] = dif[0]/cdist; n[1] = dif[1]/cdist; n[2] = dif[2]/cdist; }
  c->dist = dist;
  for (int k = 0; k < 3; k++) { c->pos[k] = p1[k] + n[k]*(r1 + dist*0.5); c->frame[k] = n[k]; c->frame[3 + k] = 0; }
  return 1;
}
static int collide_plane_capsule(Contact* c, double margin, const double* p1, const double* m1,
                                 const double* p2, const double* m2, const double* s2) {
  double nrm[3] = {m1[2], m1[5], m1[8]}, axis[3] = {m2[2], m2[5], m2[8]};
  double seg[3] = {axis[0]*s2[1], axis[1]*s2[1], axis[2]*s2[1]}, pos[3], t1[3], t2[3];
  double dp = dot3(nrm, axis);
  for (int k = 0; k < 3; k++) t1[k] = axis[k] - nrm[k]*dp;
  normalize3(t1);
  cross3(t2, nrm, t1);
  int n = 0;
  for (int k = 0; k < 3; k++) pos[k] = p2[k] + seg[k];
  n += raw_plane_sphere(c + n, margin, p1, nrm, pos, s2[0]);
  for (int k = 0; k < 3; k++) pos[k] = p2[k] - seg[k];
  n += raw_plane_sphere(c + n, margin, p1, nrm, pos, s2[0]);
  for (int i = 0; i < n; i++) for (int k = 0; k < 3; k++) { c[i].frame[3 + k] = t1[k]; c[i].frame[6 + k] = t2[k]; }
  return n;
}
static int collide_plane_box(Contact* c, double margin, const double* p1, const double* m1,
                             const double* p2, const double* m2, const double* s2) {
  double nrm[3] = {m1[2], m1[5], m1[8]};
  double dif[3] = {p2[0] - p1[0], p2[1] - p1[1], p2[2] - p1[2]};
  double dist = dot3(dif, nrm);
  int cnt = 0;
  for (int i = 0; i < 8; i++) {
    double vec[3] = {(i & 1 ? s2[0] : -s2[0]), (i & 2 ? s2[1] : -s2[1]), (i & 4 ? s2[2] : -s2[2])}, corner[3];
    mul_mat_vec3(corner, m2, vec);
    double ldist = dot3(nrm, corner);
    if (dist + ldist > margin || ldist > 0) continue;
    c[cnt].dist = dist + ldist;
    for (int k = 0; k < 3; k++) { c[cnt].frame[k] = nrm[k]; c[cnt].frame[3 + k] = 0;
      c[cnt].pos[k] = corner[k] + p2[k] - nrm[k]*c[cnt].dist*0.5; }
    if (++cnt >= 4) return 4;
  }
  return cnt;
}
static int collide_sphere_capsule(Contact* c, double margin, const double* p1, const double* s1,
                                  const double* p2, const double* m2, const double* s2) {
  double axis[3] = {m2[2], m2[5], m2[8]}, vec[3] = {p1[0] - p2[0], p1[1] - p2[1], p1[2] - p2[2]};
  double x = dot3(axis, vec);
  x = mjMAX(-s2[1], mjMIN(s2[1], x));
  double q[3] = {p2[0] + axis[0]*x, p2[1] + axis[1]*x, p2[2] + axis[2]*x};
  return raw_sphere_sphere(c, margin, p1, s1[0], q, s2[0]);
}
/* ---- sphere / capsule against a cylinder ------------------------------------------------------------------------------
 * MuJoCo: sphere-cylinder is analytic (mjc_SphereCylinder: side / cap / rim cases, nearest face when the centre is
 * inside), capsule-cylinder goes through its convex collider (one contact: the pair of closest points, contact point
 * half way between the surfaces).  Both are restated as exact geometry: the closest point of the SOLID cylinder to a point
 * is closed form; for the capsule the point runs over the axis segment, along which that distance is convex, so the
 * minimiser is found by bisection on its slope (the middle of the stretch where the slope vanishes).  A capsule whose AXIS touches or enters the
 * cylinder (penetration deeper than its radius) has no unique closest pair: returns -1 and the caller raises
 * DMC_WARN_COLLISION, as for the cylinder pairs that are not restated (cylinder-cylinder, box-cylinder, ellipsoid). */
static double point_cylinder(const double* q, const double* p, const double* a, double R, double H, double* closest) {
  double v[3] = {q[0] - p[0], q[1] - p[1], q[2] - p[2]};
  double x = dot3(v, a), perp[3] = {v[0] - x*a[0], v[1] - x*a[1], v[2] - x*a[2]};
  double d = sqrt(dot3(perp, perp)), xc = mjMAX(-H, mjMIN(H, x)), sc = d > R ? R/d : 1.0;
  for (int k = 0; k < 3; k++) closest[k] = p[k] + xc*a[k] + sc*perp[k];
  double dif[3] = {q[0] - closest[0], q[1] - closest[1], q[2] - closest[2]};
  return sqrt(dot3(dif, dif));
}
static int sphere_cylinder_core(Contact* c, double margin, const double* ps, double rs,
                                const double* p2, const double* m2, const double* s2) {
  double a[3] = {m2[2], m2[5], m2[8]}, R = s2[0], H = s2[1], closest[3];
  double g = point_cylinder(ps, p2, a, R, H, closest), n[3], dist;
  if (g >= MINVAL) {      /* centre outside: side, cap or rim, whichever holds the closest point */
    dist = g - rs;
    for (int k = 0; k < 3; k++) n[k] = (closest[k] - ps[k]) / g;
  } else {                /* centre inside the solid: out through the nearest face */
    double v[3] = {ps[0] - p2[0], ps[1] - p2[1], ps[2] - p2[2]};
    double x = dot3(v, a), perp[3] = {v[0] - x*a[0], v[1] - x*a[1], v[2] - x*a[2]}, d = sqrt(dot3(perp, perp));
    if (H - fabs(x) < R - d) { dist = -(H - fabs(x)) - rs; for (int k = 0; k < 3; k++) n[k] = x >= 0 ? -a[k] : a[k]; }
    else {
      dist = -(R - d) - rs;
      if (d < MINVAL) { n[0] = 1; n[1] = n[2] = 0; } else for (int k = 0; k < 3; k++) n[k] = -perp[k] / d;
    }
  }
  if (dist > margin) return 0;
  c->dist = dist;
  for (int k = 0; k < 3; k++) { c->pos[k] = ps[k] + n[k]*(rs + dist*0.5); c->frame[k] = n[k]; c->frame[3 + k] = 0; }
  return 1;
}
static int collide_sphere_cylinder(Contact* c, double margin, const double* p1, const double* s1,
                                   const double* p2, const double* m2, const double* s2) {
  return sphere_cylinder_core(c, margin, p1, s1[0], p2, m2, s2);
}
/* slope of the point-to-cylinder distance along the unit direction u at q = p1 + t u (the cosine between u and the
 * direction away from the closest point); nondecreasing in t because the distance to a convex set is convex */
static double segment_slope(double t, const double* p1, const double* u, const double* p2, const double* a, double R, double H) {
  double q[3] = {p1[0] + t*u[0], p1[1] + t*u[1], p1[2] + t*u[2]}, closest[3];
  double g = point_cylinder(q, p2, a, R, H, closest);
  if (g < MINVAL) return 0;
  return ((q[0] - closest[0])*u[0] + (q[1] - closest[1])*u[1] + (q[2] - closest[2])*u[2]) / g;
}
/* first t of [-h, h] at which that slope exceeds thr (h if it never does): bisection on the monotone slope */
static double slope_crossing(double thr, double h, const double* p1, const double* u, const double* p2, const double* a, double R, double H) {
  if (segment_slope(-h, p1, u, p2, a, R, H) > thr) return -h;
  if (!(segment_slope(h, p1, u, p2, a, R, H) > thr)) return h;
  double lo = -h, hi = h;
  for (int it = 0; it < 60; it++) {
    const double t = 0.5*(lo + hi);
    if (segment_slope(t, p1, u, p2, a, R, H) > thr) hi = t; else lo = t;
  }
  return 0.5*(lo + hi);
}
static int collide_capsule_cylinder(Contact* c, double margin, const double* p1, const double* m1, const double* s1,
                                    const double* p2, const double* m2, const double* s2) {
  double u[3] = {m1[2], m1[5], m1[8]}, a[3] = {m2[2], m2[5], m2[8]}, closest[3], q[3];
  /* the minimiser of the distance over the axis segment: where its slope changes sign.  When the slope is (numerically)
   * zero over a stretch -- the capsule lies along the cap or alongside the cylinder -- the middle of that stretch */
  const double tol = 1e-7;
  const double ta = slope_crossing(-tol, s1[1], p1, u, p2, a, s2[0], s2[1]);
  const double tb = slope_crossing(tol, s1[1], p1, u, p2, a, s2[0], s2[1]);
  const double t = 0.5*(ta + tb);
  for (int k = 0; k < 3; k++) q[k] = p1[k] + t*u[k];
  if (point_cylinder(q, p2, a, s2[0], s2[1], closest) < 1e-9*(s2[0] + s2[1])) return -1;      /* the axis reaches the cylinder */
  return sphere_cylinder_core(c, margin, q, s1[0], p2, m2, s2);
}
static int collide_capsule_capsule(Contact* c, double margin, const double* p1, const double* m1, const double* s1,
                                   const double* p2, const double* m2, const double* s2) {
  double a1[3] = {m1[2], m1[5], m1[8]}, a2[3] = {m2[2], m2[5], m2[8]};
  double dif[3] = {p1[0] - p2[0], p1[1] - p2[1], p1[2] - p2[2]};
  double ma = dot3(a1, a1), mb = -dot3(a1, a2), mc = dot3(a2, a2);
  double u = -dot3(a1, dif), v = dot3(a2, dif), det = ma*mc - mb*mb;
  double v1[3], v2[3];
  if (fabs(det) >= MINVAL) {
    double x1 = (mc*u - mb*v) / det, x2 = (ma*v - mb*u) / det;
    if (x1 > s1[1]) { x1 = s1[1]; x2 = (v - mb*s1[1]) / mc; }
    else if (x1 < -s1[1]) { x1 = -s1[1]; x2 = (v + mb*s1[1]) / mc; }
    if (x2 > s2[1]) { x2 = s2[1]; x1 = (u - mb*s2[1]) / ma; x1 = mjMAX(-s1[1], mjMIN(s1[1], x1)); }
    else if (x2 < -s2[1]) { x2 = -s2[1]; x1 = (u + mb*s2[1]) / ma; x1 = mjMAX(-s1[1], mjMIN(s1[1], x1)); }
    for (int k = 0; k < 3; k++) { v1[k] = p1[k] + a1[k]*x1; v2[k] = p2[k] + a2[k]*x2; }
    return raw_sphere_sphere(c, margin, v1, s1[0], v2, s2[0]);
  }
  /* parallel axes: up to two contacts from the segment end points */
  int n = 0;
  for (int s = 1; s >= -1 && n < 2; s -= 2) {
    double x1 = s * s1[1], x2 = (v - mb*x1) / mc;
    if (x2 >= -s2[1] && x2 <= s2[1]) {
      for (int k = 0; k < 3; k++) { v1[k] = p1[k] + a1[k]*x1; v2[k] = p2[k] + a2[k]*x2; }
      n += raw_sphere_sphere(c + n, margin, v1, s1[0], v2, s2[0]);
    }
  }
  for (int s = 1; s >= -1 && n < 2; s -= 2) {
    double x2 = s * s2[1], x1 = (u - mb*x2) / ma;
    if (x1 > -s1[1] && x1 < s1[1]) {
      for (int k = 0; k < 3; k++) { v1[k] = p1[k] + a1[k]*x1; v2[k] = p2[k] + a2[k]*x2; }
      n += raw_sphere_sphere(c + n, margin, v1, s1[0], v2, s2[0]);
    }
  }
  return n;
}


/* ---- ellipsoid pairs ------------------------------------------------------
 * MuJoCo routes every pair that involves an ellipsoid (other than plane-ellipsoid)
 * through its general convex collider (mjc_Convex: MPR / GJK+EPA, tolerance 1e-6,
 * one contact per pair).  Source not available here; this restates the quantity that
 * collider converges to.  For convex A, B with support functions h_A, h_B the signed
 * distance is  max_{|n|=1} F(n),  F(n) = -h_A(n) - h_B(-n)  (n from A to B): positive =
 * separation, negative = penetration depth.  Spheres and ellipsoids are smooth and
 * centrally symmetric, h_X(n) = n.c_X + |S_X R_X^T n|, so F is maximised by a Newton
 * iteration on the unit sphere; a capsule is a sphere swept along its axis segment and
 * the distance is minimised over the sweep parameter.  The contact reports
 * dist = F(n*), frame normal n*, position = midpoint of the two witness points
 * (PARITY_ASSUMPTIONS.md row 29). */
typedef struct { const double* c; const double* R; double s[3]; } Quadric;
/* support offset g = R S w/|w|, w = S R^T n; returns |w| and the unit w */
static double quadric_support(const Quadric* q, const double* n, double* g, double* wh) {
  double w[3];
  for (int k = 0; k < 3; k++) w[k] = q->s[k]*(q->R[k]*n[0] + q->R[3 + k]*n[1] + q->R[6 + k]*n[2]);
  double wn = sqrt(dot3(w, w));
  if (wn < MINVAL) { g[0] = g[1] = g[2] = 0; wh[0] = wh[1] = wh[2] = 0; return 0; }
  double v[3];
  for (int k = 0; k < 3; k++) { wh[k] = w[k]/wn; v[k] = q->s[k]*wh[k]; }
  mul_mat_vec3(g, q->R, v);
  return wn;
}
/* t_i^T H t_j for the support-function Hessian H = R S (I - wh wh^T) S R^T / |w| */
static void quadric_curv(const Quadric* q, const double* wh, double wn, const double* t1, const double* t2, double* K) {
  if (wn < MINVAL) return;
  double y1[3], y2[3];
  for (int k = 0; k < 3; k++) {
    y1[k] = q->s[k]*(q->R[k]*t1[0] + q->R[3 + k]*t1[1] + q->R[6 + k]*t1[2]);
    y2[k] = q->s[k]*(q->R[k]*t2[0] + q->R[3 + k]*t2[1] + q->R[6 + k]*t2[2]);
  }
  double a1 = dot3(y1, wh), a2 = dot3(y2, wh);
  K[0] += (dot3(y1, y1) - a1*a1)/wn; K[1] += (dot3(y1, y2) - a1*a2)/wn; K[2] += (dot3(y2, y2) - a2*a2)/wn;
}
static double quadric_gap_value(const Quadric* A, const Quadric* B, const double* n) {
  double g[3], wh[3], d[3] = {B->c[0] - A->c[0], B->c[1] - A->c[1], B->c[2] - A->c[2]};
  double hA = quadric_support(A, n, g, wh), hB = quadric_support(B, n, g, wh);
  return dot3(n, d) - hA - hB;
}
#define CCD_MAXIT 30
#define CCD_TOL 1e-10
/* maximise F over the unit sphere starting from n (in/out); gA, gB = support offsets at the optimum */
static double quadric_gap(const Quadric* A, const Quadric* B, double* n, double* gA, double* gB) {
  double d[3] = {B->c[0] - A->c[0], B->c[1] - A->c[1], B->c[2] - A->c[2]};
  double F = 0;
  for (int it = 0; ; it++) {
    double wA[3], wB[3];
    double hA = quadric_support(A, n, gA, wA), hB = quadric_support(B, n, gB, wB);
    F = dot3(n, d) - hA - hB;
    if (it >= CCD_MAXIT) break;
    double f[9] = {n[0], n[1], n[2], 0, 0, 0, 0, 0, 0};
    make_frame(f);
    const double *t1 = f + 3, *t2 = f + 6;
    double grad[3] = {d[0] - gA[0] - gB[0], d[1] - gA[1] - gB[1], d[2] - gA[2] - gB[2]};
    double g1 = dot3(t1, grad), g2 = dot3(t2, grad);
    double K[3] = {F, 0, F};
    quadric_curv(A, wA, hA, t1, t2, K); quadric_curv(B, wB, hB, t1, t2, K);
    double tr = K[0] + K[2], det = K[0]*K[2] - K[1]*K[1];
    double floor_ = 1e-3*(hA + hB) + MINVAL;
    double lmin = 0.5*(tr - sqrt(mjMAX(0.0, tr*tr - 4*det)));
    if (lmin < floor_) { double sh = floor_ - lmin; K[0] += sh; K[2] += sh; det = K[0]*K[2] - K[1]*K[1]; }
    double d1 = (K[2]*g1 - K[1]*g2)/det, d2 = (K[0]*g2 - K[1]*g1)/det;
    if (d1*d1 + d2*d2 < CCD_TOL*CCD_TOL) break;
    double nn[3];
    for (int ls = 0; ; ls++) {
      for (int k = 0; k < 3; k++) nn[k] = n[k] + t1[k]*d1 + t2[k]*d2;
      normalize3(nn);
      if (ls >= 8 || quadric_gap_value(A, B, nn) >= F) break;
      d1 *= 0.5; d2 *= 0.5;
    }
    n[0] = nn[0]; n[1] = nn[1]; n[2] = nn[2];
  }
  return F;
}
static void quadric_init_dir(const Quadric* A, const Quadric* B, double* n) {
  for (int k = 0; k < 3; k++) n[k] = B->c[k] - A->c[k];
  if (dot3(n, n) < MINVAL*MINVAL) { n[0] = 1; n[1] = n[2] = 0; }
  normalize3(n);
}
static int quadric_contact(Contact* c, double margin, const Quadric* A, const Quadric* B, double* n) {
  double gA[3], gB[3];
  double dist = quadric_gap(A, B, n, gA, gB);
  if (dist > margin) return 0;
  c->dist = dist;
  for (int k = 0; k < 3; k++) {
    c->pos[k] = 0.5*((A->c[k] + gA[k]) + (B->c[k] - gB[k]));
    c->frame[k] = n[k]; c->frame[3 + k] = 0;
  }
  return 1;
}
static Quadric make_quadric(int type, const double* pos, const double* mat, const double* size) {
  Quadric q; q.c = pos; q.R = mat;
  if (type == DMC_GEOM_ELLIPSOID) { q.s[0] = size[0]; q.s[1] = size[1]; q.s[2] = size[2]; }
  else q.s[0] = q.s[1] = q.s[2] = size[0];   /* sphere, or the swept sphere of a capsule */
  return q;
}
static int collide_plane_ellipsoid(Contact* c, double margin, const double* p1, const double* m1,
                                   const double* p2, const double* m2, const double* s2) {
  /* plane vs smooth convex: the support point of the ellipsoid along -normal */
  double nrm[3] = {m1[2], m1[5], m1[8]}, g[3], wh[3];
  Quadric q = make_quadric(DMC_GEOM_ELLIPSOID, p2, m2, s2);
  quadric_support(&q, nrm, g, wh);
  double pt[3] = {p2[0] - g[0], p2[1] - g[1], p2[2] - g[2]};
  double dif[3] = {pt[0] - p1[0], pt[1] - p1[1], pt[2] - p1[2]};
  double dist = dot3(dif, nrm);
  if (dist > margin) return 0;
  c->dist = dist;
  for (int k = 0; k < 3; k++) { c->pos[k] = pt[k] - nrm[k]*dist*0.5; c->frame[k] = nrm[k]; c->frame[3 + k] = 0; }
  return 1;
}
#define DMC_PI 3.14159265358979323846
/* ---- box pairs -----------------------------------------------------------
 * MuJoCo has dedicated routines (mjc_SphereBox, mjc_CapsuleBox, mjc_BoxBox); their source is not
 * available here.  These are restatements of the same geometric problems (PARITY_ASSUMPTIONS.md row 33):
 *   sphere-box   closest point of the box to the centre (inside: the nearest face), one contact
 *   capsule-box  the axis point nearest to the box, plus the far end cap when it is in range too
 *   box-box      separating-axis test over the 15 axes; a face axis gives the incident face clipped
 *                against the reference face (at most 4 points kept, spread around the deepest one),
 *                an edge axis gives the closest points of the two edges */
static int sphere_box_core(Contact* c, double margin, const double* ps, double r,
                           const double* pb, const double* mb, const double* sb) {
  double dif[3] = {ps[0] - pb[0], ps[1] - pb[1], ps[2] - pb[2]}, cl[3], q[3], nb[3] = {0, 0, 0};
  mul_matT_vec3(cl, mb, dif);
  int outside = 0;
  for (int k = 0; k < 3; k++) { q[k] = mjMAX(-sb[k], mjMIN(sb[k], cl[k])); if (q[k] != cl[k]) outside = 1; }
  double dist;
  if (outside) {
    double d[3] = {cl[0] - q[0], cl[1] - q[1], cl[2] - q[2]};
    const double dn = sqrt(dot3(d, d));
    dist = dn - r;
    if (dist > margin) return 0;
    for (int k = 0; k < 3; k++) nb[k] = d[k]/dn;
  } else {
    int best = 0; double depth = sb[0] - fabs(cl[0]);
    for (int k = 1; k < 3; k++) { const double dk = sb[k] - fabs(cl[k]); if (dk < depth) { depth = dk; best = k; } }
    nb[best] = cl[best] >= 0 ? 1 : -1;
    q[best] = nb[best]*sb[best];
    dist = -depth - r;
  }
  double nw[3], qw[3];
  mul_mat_vec3(nw, mb, nb); mul_mat_vec3(qw, mb, q);
  c->dist = dist;
  /* frame normal: from the sphere (geom 1) to the box (geom 2) */
  for (int k = 0; k < 3; k++) { c->pos[k] = pb[k] + qw[k] + nw[k]*dist*0.5; c->frame[k] = -nw[k]; c->frame[3 + k] = 0; }
  return 1;
}
/* squared distance from the point p0 + t u (box frame) to the box, and its derivative in t */
static double seg_box_dd(const double* p0, const double* u, const double* sb, double t, double* deriv) {
  double f = 0, g = 0;
  for (int k = 0; k < 3; k++) {
    const double x = p0[k] + t*u[k], e = x - mjMAX(-sb[k], mjMIN(sb[k], x));
    f += e*e; g += 2*e*u[k];
  }
  *deriv = g;
  return f;
}
static int collide_capsule_box(Contact* c, double margin, const double* p1, const double* m1, const double* s1,
                               const double* p2, const double* m2, const double* s2) {
  double axw[3] = {m1[2], m1[5], m1[8]}, dif[3] = {p1[0] - p2[0], p1[1] - p2[1], p1[2] - p2[2]}, p0[3], u[3];
  mul_matT_vec3(p0, m2, dif); mul_matT_vec3(u, m2, axw);
  const double h = s1[1];
  /* convex in t: bisection on the derivative */
  double lo = -h, hi = h, g, t;
  seg_box_dd(p0, u, s2, lo, &g);
  if (g >= 0) t = lo;
  else {
    seg_box_dd(p0, u, s2, hi, &g);
    if (g <= 0) t = hi;
    else {
      for (int it = 0; it < 60; it++) { t = 0.5*(lo + hi); seg_box_dd(p0, u, s2, t, &g); if (g > 0) hi = t; else lo = t; }
      t = 0.5*(lo + hi);
    }
  }
  int n = 0;
  double ps[3];
  for (int k = 0; k < 3; k++) ps[k] = p1[k] + axw[k]*t;
  n += sphere_box_core(c + n, margin, ps, s1[0], p2, m2, s2);
  /* second contact: the end cap farther from t, when it is in range as well (capsule lying on a face) */
  const double t2 = t <= 0 ? h : -h;
  if (fabs(t2 - t) > 1e-3*h) {
    for (int k = 0; k < 3; k++) ps[k] = p1[k] + axw[k]*t2;
    n += sphere_box_core(c + n, margin, ps, s1[0], p2, m2, s2);
  }
  return n;
}
static int collide_box_box(Contact* c, double margin, const double* pA, const double* RA, const double* sA,
                           const double* pB, const double* RB, const double* sB) {
  double d[3] = {pB[0] - pA[0], pB[1] - pA[1], pB[2] - pA[2]};
  double colA[3][3], colB[3][3];
  for (int i = 0; i < 3; i++) for (int k = 0; k < 3; k++) { colA[i][k] = RA[3*k + i]; colB[i][k] = RB[3*k + i]; }
  /* face axes */
  double best = -1e300; int code = -1; double bestn[3] = {0, 0, 0};
  for (int i = 0; i < 6; i++) {
    const double* L = i < 3 ? colA[i] : colB[i - 3];
    double ra = 0, rb = 0;
    for (int k = 0; k < 3; k++) { ra += sA[k]*fabs(dot3(L, colA[k])); rb += sB[k]*fabs(dot3(L, colB[k])); }
    const double proj = dot3(L, d), sep = fabs(proj) - ra - rb;
    if (sep > margin) return 0;
    if (sep > best) { best = sep; code = i; for (int k = 0; k < 3; k++) bestn[k] = proj >= 0 ? L[k] : -L[k]; }
  }
  /* edge axes: preferred only when clearly less penetrating than the best face (factor 1.05 on the depth) */
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) {
    double L[3];
    cross3(L, colA[i], colB[j]);
    const double ln = sqrt(dot3(L, L));
    if (ln < 1e-6) continue;
    for (int k = 0; k < 3; k++) L[k] /= ln;
    double ra = 0, rb = 0;
    for (int k = 0; k < 3; k++) { ra += sA[k]*fabs(dot3(L, colA[k])); rb += sB[k]*fabs(dot3(L, colB[k])); }
    const double proj = dot3(L, d), sep = fabs(proj) - ra - rb;
    if (sep > margin) return 0;
    if (sep > 0 ? sep > best : sep*1.05 > best) {
      if (!(sep > 0) && !(best < 0)) continue;
      best = sep; code = 6 + 3*i + j; for (int k = 0; k < 3; k++) bestn[k] = proj >= 0 ? L[k] : -L[k];
    }
  }
  if (code >= 6) {
    /* edge-edge: the edges of A and B that face each other along n, closest points of their lines */
    const int i = (code - 6)/3, j = (code - 6) % 3;
    double pa[3] = {pA[0], pA[1], pA[2]}, pb[3] = {pB[0], pB[1], pB[2]};
    for (int k = 0; k < 3; k++) if (k != i) { const double sg = dot3(bestn, colA[k]) > 0 ? 1 : -1; for (int a = 0; a < 3; a++) pa[a] += sg*sA[k]*colA[k][a]; }
    for (int k = 0; k < 3; k++) if (k != j) { const double sg = dot3(bestn, colB[k]) > 0 ? -1 : 1; for (int a = 0; a < 3; a++) pb[a] += sg*sB[k]*colB[k][a]; }
    const double *ua = colA[i], *ub = colB[j];
    double w[3] = {pb[0] - pa[0], pb[1] - pa[1], pb[2] - pa[2]};
    const double uaub = dot3(ua, ub), q1 = dot3(ua, w), q2 = -dot3(ub, w), den = 1 - uaub*uaub;
    double alpha = 0, beta = 0;
    if (den > 1e-12) { alpha = (q1 + uaub*q2)/den; beta = (uaub*q1 + q2)/den; }
    alpha = mjMAX(-sA[i], mjMIN(sA[i], alpha)); beta = mjMAX(-sB[j], mjMIN(sB[j], beta));
    c->dist = best;
    for (int k = 0; k < 3; k++) {
      c->pos[k] = 0.5*((pa[k] + alpha*ua[k]) + (pb[k] + beta*ub[k]));
      c->frame[k] = bestn[k]; c->frame[3 + k] = 0;
    }
    return 1;
  }
  /* face contact: reference box = owner of the axis, incident face = the face of the other box most
   * opposed to n; Sutherland-Hodgman clipping of the incident face against the reference face sides */
  const int refA = code < 3;
  const double *pR = refA ? pA : pB, *sR = refA ? sA : sB, *pI = refA ? pB : pA, *sI = refA ? sB : sA;
  double (*cR)[3] = refA ? colA : colB, (*cI)[3] = refA ? colB : colA;
  const int ax = refA ? code : code - 3;
  double nref[3];                              /* outward normal of the reference face, towards the other box */
  for (int k = 0; k < 3; k++) nref[k] = refA ? bestn[k] : -bestn[k];
  int inc = 0; double incdot = 1e300;
  for (int k = 0; k < 3; k++) { const double dk = dot3(nref, cI[k]); if (-fabs(dk) < incdot) { incdot = -fabs(dk); inc = k; } }
  const double incsign = dot3(nref, cI[inc]) > 0 ? -1 : 1;
  const int i1 = (inc + 1) % 3, i2 = (inc + 2) % 3, r1 = (ax + 1) % 3, r2 = (ax + 2) % 3;
  /* incident face vertices in the 2-D frame (r1, r2) of the reference face, plus their height above it */
  double poly[16][3], tmp[16][3];
  int np_ = 4;
  for (int v = 0; v < 4; v++) {
    const double a = (v == 0 || v == 3) ? 1 : -1, b = v < 2 ? 1 : -1;
    double pt[3];
    for (int k = 0; k < 3; k++) pt[k] = pI[k] + incsign*sI[inc]*cI[inc][k] + a*sI[i1]*cI[i1][k] + b*sI[i2]*cI[i2][k] - pR[k];
    poly[v][0] = dot3(pt, cR[r1]); poly[v][1] = dot3(pt, cR[r2]); poly[v][2] = dot3(pt, nref) - sR[ax];
  }
  for (int side = 0; side < 4 && np_ > 0; side++) {
    const int coord = side >> 1; const double sg = (side & 1) ? -1 : 1, lim = coord ? sR[r2] : sR[r1];
    int nn = 0;
    for (int v = 0; v < np_; v++) {
      const double* P = poly[v]; const double* Q = poly[(v + 1) % np_];
      const double dp = lim - sg*P[coord], dq = lim - sg*Q[coord];
      if (dp >= 0) { tmp[nn][0] = P[0]; tmp[nn][1] = P[1]; tmp[nn][2] = P[2]; nn++; }
      if ((dp >= 0) != (dq >= 0)) { const double f = dp/(dp - dq); for (int k = 0; k < 3; k++) tmp[nn][k] = P[k] + f*(Q[k] - P[k]); nn++; }
    }
    np_ = nn;
    for (int v = 0; v < np_; v++) for (int k = 0; k < 3; k++) poly[v][k] = tmp[v][k];
  }
  /* keep the points at or below the margin */
  int nk = 0;
  for (int v = 0; v < np_; v++) if (poly[v][2] <= margin) { for (int k = 0; k < 3; k++) poly[nk][k] = poly[v][k]; nk++; }
  if (!nk) return 0;
  /* at most 4: the deepest point, then the candidates nearest to +90, 180, 270 degrees around the centroid */
  int pick[4], npick = 0;
  if (nk <= 4) { for (int v = 0; v < nk; v++) pick[npick++] = v; }
  else {
    double cx = 0, cy = 0; int deep = 0;
    for (int v = 0; v < nk; v++) { cx += poly[v][0]; cy += poly[v][1]; if (poly[v][2] < poly[deep][2]) deep = v; }
    cx /= nk; cy /= nk;
    const double a0 = atan2(poly[deep][1] - cy, poly[deep][0] - cx);
    int used[16] = {0};
    pick[npick++] = deep; used[deep] = 1;
    for (int q = 1; q < 4; q++) {
      const double target = a0 + q*(DMC_PI/2);
      int bv = -1; double bd = 1e300;
      for (int v = 0; v < nk; v++) if (!used[v]) {
        double da = fabs(fmod(atan2(poly[v][1] - cy, poly[v][0] - cx) - target + 5*DMC_PI, 2*DMC_PI) - DMC_PI);
        if (da < bd) { bd = da; bv = v; }
      }
      pick[npick++] = bv; used[bv] = 1;
    }
  }
  for (int q = 0; q < npick; q++) {
    const double* P = poly[pick[q]];
    c[q].dist = P[2];
    for (int k = 0; k < 3; k++) {
      /* midpoint between the incident point and its projection on the reference face */
      c[q].pos[k] = pR[k] + P[0]*cR[r1][k] + P[1]*cR[r2][k] + (sR[ax] + 0.5*P[2])*nref[k];
      c[q].frame[k] = bestn[k]; c[q].frame[3 + k] = 0;
    }
  }
  return npick;
}
/* plane (geom 1) vs cylinder (mjc_PlaneCylinder): the deepest rim point of the cap facing the plane,
 * the matching rim point of the other cap, and two more points of the near disc at +-120 degrees */
static int collide_plane_cylinder(Contact* c, double margin, const double* p1, const double* m1,
                                  const double* p2, const double* m2, const double* s2) {
  double nrm[3] = {m1[2], m1[5], m1[8]}, axis[3] = {m2[2], m2[5], m2[8]};
  double prjaxis = dot3(nrm, axis);
  if (prjaxis > 0) { axis[0] = -axis[0]; axis[1] = -axis[1]; axis[2] = -axis[2]; prjaxis = -prjaxis; }
  double dif[3] = {p2[0] - p1[0], p2[1] - p1[1], p2[2] - p1[2]};
  const double dist0 = dot3(dif, nrm);
  /* direction on the disc that points most towards the plane */
  double vec[3] = {axis[0]*prjaxis - nrm[0], axis[1]*prjaxis - nrm[1], axis[2]*prjaxis - nrm[2]};
  const double len2 = dot3(vec, vec);
  if (len2 >= MINVAL*MINVAL) { const double sc = s2[0]/sqrt(len2); vec[0] *= sc; vec[1] *= sc; vec[2] *= sc; }
  else { vec[0] = m2[0]*s2[0]; vec[1] = m2[3]*s2[0]; vec[2] = m2[6]*s2[0]; }   /* disc parallel to the plane */
  const double prjvec = dot3(vec, nrm);
  axis[0] *= s2[1]; axis[1] *= s2[1]; axis[2] *= s2[1]; prjaxis *= s2[1];
  int n = 0;
  if (dist0 + prjaxis + prjvec > margin) return 0;
  c[n].dist = dist0 + prjaxis + prjvec;
  for (int k = 0; k < 3; k++) { c[n].pos[k] = p2[k] + vec[k] + axis[k] - nrm[k]*c[n].dist*0.5; c[n].frame[k] = nrm[k]; c[n].frame[3 + k] = 0; }
  n++;
  if (dist0 - prjaxis + prjvec <= margin) {
    c[n].dist = dist0 - prjaxis + prjvec;
    for (int k = 0; k < 3; k++) { c[n].pos[k] = p2[k] + vec[k] - axis[k] - nrm[k]*c[n].dist*0.5; c[n].frame[k] = nrm[k]; c[n].frame[3 + k] = 0; }
    n++;
  }
  const double prjvec1 = -prjvec*0.5;
  if (dist0 + prjaxis + prjvec1 <= margin) {
    double vec1[3];
    cross3(vec1, vec, axis);
    normalize3(vec1);
    const double sc = s2[0]*sqrt(3.0)/2;
    for (int sg = 1; sg >= -1; sg -= 2) {
      c[n].dist = dist0 + prjaxis + prjvec1;
      for (int k = 0; k < 3; k++) { c[n].pos[k] = p2[k] + sg*sc*vec1[k] + axis[k] - vec[k]*0.5 - nrm[k]*c[n].dist*0.5; c[n].frame[k] = nrm[k]; c[n].frame[3 + k] = 0; }
      n++;
    }
  }
  return n;
}
/* capsule (geom 1) vs ellipsoid (geom 2): minimise the swept-sphere gap over the axis parameter */
static int collide_capsule_ellipsoid(Contact* c, double margin, const double* p1, const double* m1, const double* s1,
                                     const double* p2, const double* m2, const double* s2) {
  double u[3] = {m1[2], m1[5], m1[8]}, cc[3], n[3], gA[3], gB[3];
  const double h = s1[1];
  Quadric A = make_quadric(DMC_GEOM_SPHERE, cc, m1, s1), B = make_quadric(DMC_GEOM_ELLIPSOID, p2, m2, s2);
  /* psi(t) = n*(t).u is decreasing in t; the minimising t is its root clamped to [-h, h] */
  double tlo = -h, thi = h, plo, phi, t;
  for (int k = 0; k < 3; k++) cc[k] = p1[k] + u[k]*tlo;
  quadric_init_dir(&A, &B, n);
  quadric_gap(&A, &B, n, gA, gB); plo = dot3(n, u);
  if (plo <= 0) t = tlo;
  else {
    for (int k = 0; k < 3; k++) cc[k] = p1[k] + u[k]*thi;
    quadric_gap(&A, &B, n, gA, gB); phi = dot3(n, u);
    if (phi >= 0) t = thi;
    else {
      t = 0;
      int side = 0;   /* Illinois variant of regula falsi */
      for (int it = 0; it < 40; it++) {
        t = (tlo*phi - thi*plo)/(phi - plo);
        for (int k = 0; k < 3; k++) cc[k] = p1[k] + u[k]*t;
        quadric_gap(&A, &B, n, gA, gB);
        double pt = dot3(n, u);
        if (fabs(pt) < CCD_TOL || thi - tlo < CCD_TOL*h) break;
        if (pt > 0) { tlo = t; plo = pt; if (side == 1) phi *= 0.5; side = 1; }
        else { thi = t; phi = pt; if (side == -1) plo *= 0.5; side = -1; }
      }
    }
  }
  for (int k = 0; k < 3; k++) cc[k] = p1[k] + u[k]*t;
  return quadric_contact(c, margin, &A, &B, n);
}

static void collision(const Model* m, Data* d) {
  d->ncon = 0;
  if (m->opt_disableflags & (DMC_DSBL_CONTACT | DMC_DSBL_CONSTRAINT)) return;
  for (int p = 0; p < m->npair; p++) {
    int g1 = m->pair_geom1[p], g2 = m->pair_geom2[p];
    int t1 = m->geom_type[g1], t2 = m->geom_type[g2];
    double margin = mjMAX(m->geom_margin[g1], m->geom_margin[g2]);
    double gap = mjMAX(m->geom_gap[g1], m->geom_gap[g2]);
    const double *p1 = d->geom_xpos + 3*g1, *p2 = d->geom_xpos + 3*g2;
    const double *m1 = d->geom_xmat + 9*g1, *m2 = d->geom_xmat + 9*g2;
    const double *s1 = m->geom_size + 3*g1, *s2 = m->geom_size + 3*g2;
    /* bounding-sphere / plane-distance rejection (pure pruning: the narrow
     * phase below only returns contacts with dist <= margin) */
    if (t1 == DMC_GEOM_PLANE) {
      double dif[3] = {p2[0] - p1[0], p2[1] - p1[1], p2[2] - p1[2]}, nrm[3] = {m1[2], m1[5], m1[8]};
      if (dot3(dif, nrm) > m->geom_rbound[g2] + margin) continue;
    } else {
      double dif[3] = {p2[0] - p1[0], p2[1] - p1[1], p2[2] - p1[2]};
      double bound = m->geom_rbound[g1] + m->geom_rbound[g2] + margin;
      if (dot3(dif, dif) > bound*bound) continue;
    }
    if (d->ncon + 4 > m->nconmax) { d->warning[DMC_WARN_CONTACTFULL]++; break; }
    Contact* c = d->contact + d->ncon;
    int n = 0;
    /* cylinders: against a plane, a sphere or a capsule the narrow phase is restated; every other pair is tested as
     * the cylinder's enclosing capsule (same radius / half-length) and a hit only raises DMC_WARN_COLLISION */
    const int cyl_pair = t2 == DMC_GEOM_CYLINDER && (t1 == DMC_GEOM_SPHERE || t1 == DMC_GEOM_CAPSULE);      /* restated exactly */
    const int guard = (t1 == DMC_GEOM_CYLINDER || t2 == DMC_GEOM_CYLINDER) && t1 != DMC_GEOM_PLANE && !cyl_pair;
    const int plane_cyl = t1 == DMC_GEOM_PLANE && t2 == DMC_GEOM_CYLINDER;
    if (cyl_pair) {
      n = t1 == DMC_GEOM_SPHERE ? collide_sphere_cylinder(c, margin, p1, s1, p2, m2, s2)
                                : collide_capsule_cylinder(c, margin, p1, m1, s1, p2, m2, s2);
      if (n < 0) { d->warning[DMC_WARN_COLLISION]++; continue; }
      t1 = t2 = -1;      /* (handled: none of the branches below) */
    }
    if (t1 == DMC_GEOM_CYLINDER) t1 = DMC_GEOM_CAPSULE;
    if (t2 == DMC_GEOM_CYLINDER) t2 = DMC_GEOM_CAPSULE;
    if (cyl_pair) {}
    else if (plane_cyl) n = collide_plane_cylinder(c, margin, p1, m1, p2, m2, s2);
    else if (t1 == DMC_GEOM_PLANE && t2 == DMC_GEOM_SPHERE) { double nrm[3] = {m1[2], m1[5], m1[8]}; n = raw_plane_sphere(c, margin, p1, nrm, p2, s2[0]); }
    else if (t1 == DMC_GEOM_PLANE && t2 == DMC_GEOM_CAPSULE) n = collide_plane_capsule(c, margin, p1, m1, p2, m2, s2);
    else if (t1 == DMC_GEOM_PLANE && t2 == DMC_GEOM_BOX) n = collide_plane_box(c, margin, p1, m1, p2, m2, s2);
    else if (t1 == DMC_GEOM_SPHERE && t2 == DMC_GEOM_SPHERE) n = raw_sphere_sphere(c, margin, p1, s1[0], p2, s2[0]);
    else if (t1 == DMC_GEOM_SPHERE && t2 == DMC_GEOM_CAPSULE) n = collide_sphere_capsule(c, margin, p1, s1, p2, m2, s2);
    else if (t1 == DMC_GEOM_CAPSULE && t2 == DMC_GEOM_CAPSULE) n = collide_capsule_capsule(c, margin, p1, m1, s1, p2, m2, s2);
    else if (t1 == DMC_GEOM_SPHERE && t2 == DMC_GEOM_BOX) n = sphere_box_core(c, margin, p1, s1[0], p2, m2, s2);
    else if (t1 == DMC_GEOM_CAPSULE && t2 == DMC_GEOM_BOX) n = collide_capsule_box(c, margin, p1, m1, s1, p2, m2, s2);
    else if (t1 == DMC_GEOM_BOX && t2 == DMC_GEOM_BOX) n = collide_box_box(c, margin, p1, m1, s1, p2, m2, s2);
    else if (t1 == DMC_GEOM_PLANE && t2 == DMC_GEOM_ELLIPSOID) n = collide_plane_ellipsoid(c, margin, p1, m1, p2, m2, s2);
    else if (t1 == DMC_GEOM_CAPSULE && t2 == DMC_GEOM_ELLIPSOID) n = collide_capsule_ellipsoid(c, margin, p1, m1, s1, p2, m2, s2);
    else if ((t1 == DMC_GEOM_SPHERE || t1 == DMC_GEOM_ELLIPSOID) && t2 == DMC_GEOM_ELLIPSOID) {
      Quadric A = make_quadric(t1, p1, m1, s1), B = make_quadric(t2, p2, m2, s2);
      double nn[3]; quadric_init_dir(&A, &B, nn);
      n = quadric_contact(c, margin, &A, &B, nn);
    }
    else { d->warning[DMC_WARN_COLLISION]++; continue; } /* pair type not restated and within bounding range */
    if (guard) { if (n > 0) d->warning[DMC_WARN_COLLISION]++; continue; }
    /* contact parameters (max / priority / solmix) */
    for (int i = 0; i < n; i++) {
      Contact* ci = c + i;
      ci->geom1 = g1; ci->geom2 = g2; ci->includemargin = margin - gap; ci->exclude = (ci->dist >= margin - gap); ci->efc_address = -1;
      int pr1 = m->geom_priority[g1], pr2 = m->geom_priority[g2];
      double fr[3];
      if (pr1 == pr2) {
        ci->dim = mjMAX(m->geom_condim[g1], m->geom_condim[g2]);
        for (int k = 0; k < 3; k++) fr[k] = mjMAX(m->geom_friction[3*g1 + k], m->geom_friction[3*g2 + k]);
      } else {
        int gp = pr1 > pr2 ? g1 : g2;
        ci->dim = m->geom_condim[gp];
        for (int k = 0; k < 3; k++) fr[k] = m->geom_friction[3*gp + k];
      }
      double mix;
      if (pr1 != pr2) mix = pr1 > pr2 ? 1 : 0;
      else {
        double sm1 = m->geom_solmix[g1], sm2 = m->geom_solmix[g2];
        if (sm1 >= MINVAL && sm2 >= MINVAL) mix = sm1 / (sm1 + sm2);
        else if (sm1 < MINVAL && sm2 < MINVAL) mix = 0.5;
        else mix = sm1 < MINVAL ? 0.0 : 1.0;
      }
      const double *r1 = m->geom_solref + 2*g1, *r2 = m->geom_solref + 2*g2;
      if (r1[0] > 0 && r2[0] > 0) for (int k = 0; k < 2; k++) ci->solref[k] = mix*r1[k] + (1 - mix)*r2[k];
      else for (int k = 0; k < 2; k++) ci->solref[k] = mjMIN(r1[k], r2[k]);
      for (int k = 0; k < 5; k++) ci->solimp[k] = mix*m->geom_solimp[5*g1 + k] + (1 - mix)*m->geom_solimp[5*g2 + k];
      ci->friction[0] = ci->friction[1] = mjMAX(DMC_MINMU, fr[0]);
      ci->friction[2] = mjMAX(DMC_MINMU, fr[1]);
      ci->friction[3] = ci->friction[4] = mjMAX(DMC_MINMU, fr[2]);
      if (ci->frame[3] == 0 && ci->frame[4] == 0 && ci->frame[5] == 0) make_frame(ci->frame);
    }
    d->ncon += n;
  }
}

/* ---- constraints --------------------------------------------------------- */
/* translational (jp) and rotational (jr) Jacobian column `dof` of a world point
 * attached to `body`; zero if dof is not an ancestor of body */
static void jac_col(const Model* m, const Data* d, int body, const double* point, int dof, double* jp, double* jr) {
  jp[0] = jp[1] = jp[2] = 0; jr[0] = jr[1] = jr[2] = 0;
  while (body > 0 && m->body_dofnum[body] == 0) body = m->body_parentid[body];
  if (body <= 0) return;
  int i = m->body_dofadr[body] + m->body_dofnum[body] - 1, found = 0;
  for (; i >= 0; i = m->dof_parentid[i]) if (i == dof) { found = 1; break; }
  if (!found) return;
  const double* rc = d->subtree_com + 3*m->body_rootid[body];
  double off[3] = {point[0] - rc[0], point[1] - rc[1], point[2] - rc[2]}, tmp[3];
  const double* cd = d->cdof + 6*dof;
  jr[0] = cd[0]; jr[1] = cd[1]; jr[2] = cd[2];
  cross3(tmp, cd, off);
  jp[0] = cd[3] + tmp[0]; jp[1] = cd[4] + tmp[1]; jp[2] = cd[5] + tmp[2];
}
static void get_impedance(const double* solimp_in, double pos, double margin, double* imp) {
  double s[5];
  s[0] = mjMAX(DMC_MINIMP, mjMIN(DMC_MAXIMP, solimp_in[0]));
  s[1] = mjMAX(DMC_MINIMP, mjMIN(DMC_MAXIMP, solimp_in[1]));
  s[2] = mjMAX(0, solimp_in[2]);
  s[3] = mjMAX(DMC_MINIMP, mjMIN(DMC_MAXIMP, solimp_in[3]));
  s[4] = mjMAX(1, solimp_in[4]);
  if (s[0] == s[1] || s[2] <= MINVAL) { *imp = 0.5*(s[0] + s[1]); return; }
  double x = (pos - margin) / s[2];
  if (x < 0) x = -x;
  if (x >= 1) { *imp = s[1]; return; }
  if (x == 0) { *imp = s[0]; return; }
  double y;
  if (s[4] == 1) y = x;
  else if (x <= s[3]) y = (1 / pow(s[3], s[4] - 1)) * pow(x, s[4]);
  else y = 1 - (1 / pow(1 - s[3], s[4] - 1)) * pow(1 - x, s[4]);
  *imp = s[0] + y*(s[1] - s[0]);
}

/* mj_tendon for the supported kinds: fixed (sum coef*q) and spatial site-to-site paths
 * (sum of segment lengths; J = unit segment direction times the difference of the point Jacobians) */
static void tendon_kinematics(const Model* m, Data* d) {
  int nv = m->nv;
  for (int t = 0; t < m->ntendon; t++) {
    double* J = d->ten_J + (size_t)t*nv;
    memset(J, 0, sizeof(double) * (size_t)nv);
    int w0 = m->tendon_adr[t], wn = m->tendon_num[t];
    double len = 0;
    if (wn && m->wrap_type[w0] == DMC_WRAP_JOINT) {
      for (int w = w0; w < w0 + wn; w++) {
        int j = m->wrap_objid[w];
        len += m->wrap_prm[w] * d->qpos[m->jnt_qposadr[j]];
        J[m->jnt_dofadr[j]] += m->wrap_prm[w];
      }
    } else {
      for (int w = w0; w + 1 < w0 + wn; w++) {
        int s0 = m->wrap_objid[w], s1 = m->wrap_objid[w + 1];
        const double *p0 = d->site_xpos + 3*s0, *p1 = d->site_xpos + 3*s1;
        double dif[3] = {p1[0] - p0[0], p1[1] - p0[1], p1[2] - p0[2]};
        double n = sqrt(dot3(dif, dif));
        len += n;
        if (n < MINVAL) continue;
        for (int a = 0; a < 3; a++) dif[a] /= n;
        for (int k = 0; k < nv; k++) {
          double jp0[3], jp1[3], jr[3];
          jac_col(m, d, m->site_bodyid[s0], p0, k, jp0, jr);
          jac_col(m, d, m->site_bodyid[s1], p1, k, jp1, jr);
          J[k] += dif[0]*(jp1[0] - jp0[0]) + dif[1]*(jp1[1] - jp0[1]) + dif[2]*(jp1[2] - jp0[2]);
        }
      }
    }
    d->ten_length[t] = len;
  }
}

static void make_constraint(const Model* m, Data* d) {
  int nv = m->nv;
  d->nefc = 0;
  for (int i = 0; i < d->ncon; i++) d->contact[i].efc_address = -1;
  if (m->opt_disableflags & DMC_DSBL_CONSTRAINT) return;
  /* equality constraints (first, as in MuJoCo; mj_instantiateEquality), always active, two-sided:
   *   tendon   (L - L0) - polycoef[0], J = tendon Jacobian
   *   joint    (q1 - q1_0) - poly(q2 - q2_0), J = e_dof1 - poly'(q2 - q2_0) e_dof2
   *   connect  pos1 - pos2 of the two anchor points (3 rows), J = point Jacobian difference
   *   weld     3 rows as connect (anchor of body2, its image in body1), 3 rows torquescale * vec(q2^-1 q1 relquat),
   *            J_rot column = torquescale * 0.5 * vec(q2^-1 (0, w1 - w2) q1 relquat) */
  if (!(m->opt_disableflags & DMC_DSBL_EQUALITY)) for (int k = 0; k < m->neq; k++) {
    if (!m->eq_active0[k]) continue;
    const int type = m->eq_type[k];
    const int nrow = type == DMC_EQ_CONNECT ? 3 : type == DMC_EQ_WELD ? 6 : 1;
    if (d->nefc + nrow > m->njmax) { d->warning[DMC_WARN_CNSTRFULL]++; return; }
    const int r0 = d->nefc;
    d->nefc += nrow;
    const double* data = m->eq_data + 11*k;
    for (int a = 0; a < nrow; a++) { d->efc_margin[r0 + a] = 0; d->efc_type[r0 + a] = CT_EQUALITY; d->efc_id[r0 + a] = k; }
    if (type == DMC_EQ_TENDON) {
      const int t = m->eq_obj1id[k];
      memcpy(d->efc_J + (size_t)r0*nv, d->ten_J + (size_t)t*nv, sizeof(double) * (size_t)nv);
      d->efc_pos[r0] = d->ten_length[t] - m->tendon_length0[t] - data[0];
    } else if (type == DMC_EQ_JOINT) {
      const int j1 = m->eq_obj1id[k], j2 = m->eq_obj2id[k];
      memset(d->efc_J + (size_t)r0*nv, 0, sizeof(double) * (size_t)nv);
      double pos = d->qpos[m->jnt_qposadr[j1]] - m->qpos0[m->jnt_qposadr[j1]], deriv = 0;
      if (j2 >= 0) {
        const double dif = d->qpos[m->jnt_qposadr[j2]] - m->qpos0[m->jnt_qposadr[j2]];
        double pw = 1, poly = 0;
        for (int p = 0; p < 5; p++) { poly += data[p]*pw; if (p < 4) deriv += (p + 1)*data[p + 1]*pw; pw *= dif; }
        pos -= poly;
        d->efc_J[(size_t)r0*nv + m->jnt_dofadr[j2]] = -deriv;
      } else pos -= data[0];
      d->efc_J[(size_t)r0*nv + m->jnt_dofadr[j1]] += 1;
      d->efc_pos[r0] = pos;
    } else {
      const int b1 = m->eq_obj1id[k], b2 = m->eq_obj2id[k];
      /* connect: data[0:3] on body1, data[3:6] on body2; weld: data[3:6] on body1, data[0:3] on body2 */
      const double *l1 = type == DMC_EQ_CONNECT ? data : data + 3, *l2 = type == DMC_EQ_CONNECT ? data + 3 : data;
      double p1[3], p2[3], tmp[3];
      mul_mat_vec3(tmp, d->xmat + 9*b1, l1); for (int a = 0; a < 3; a++) p1[a] = d->xpos[3*b1 + a] + tmp[a];
      mul_mat_vec3(tmp, d->xmat + 9*b2, l2); for (int a = 0; a < 3; a++) p2[a] = d->xpos[3*b2 + a] + tmp[a];
      for (int a = 0; a < 3; a++) d->efc_pos[r0 + a] = p1[a] - p2[a];
      double quat[4] = {1, 0, 0, 0}, q2inv[4] = {1, 0, 0, 0};
      if (type == DMC_EQ_WELD) {
        double err[4];
        mul_quat(quat, d->xquat + 4*b1, data + 6);                       /* q1 * relquat */
        q2inv[0] = d->xquat[4*b2]; for (int a = 1; a < 4; a++) q2inv[a] = -d->xquat[4*b2 + a];
        mul_quat(err, q2inv, quat);
        for (int a = 0; a < 3; a++) d->efc_pos[r0 + 3 + a] = data[10]*err[1 + a];
      }
      for (int dof = 0; dof < nv; dof++) {
        double jp1[3], jr1[3], jp2[3], jr2[3];
        jac_col(m, d, b1, p1, dof, jp1, jr1);
        jac_col(m, d, b2, p2, dof, jp2, jr2);
        for (int a = 0; a < 3; a++) d->efc_J[(size_t)(r0 + a)*nv + dof] = jp1[a] - jp2[a];
        if (type == DMC_EQ_WELD) {
          const double w[4] = {0, jr1[0] - jr2[0], jr1[1] - jr2[1], jr1[2] - jr2[2]};
          double t1[4], t2[4];
          mul_quat(t1, q2inv, w);
          mul_quat(t2, t1, quat);
          for (int a = 0; a < 3; a++) d->efc_J[(size_t)(r0 + 3 + a)*nv + dof] = data[10]*0.5*t2[1 + a];
        }
      }
    }
  }
  /* dof friction loss (rows come first, as in MuJoCo: equality, friction, limit, contact) */
  if (!(m->opt_disableflags & DMC_DSBL_FRICTIONLOSS)) for (int i = 0; i < nv; i++) {
    if (m->dof_frictionloss[i] <= 0) continue;
    if (d->nefc >= m->njmax) { d->warning[DMC_WARN_CNSTRFULL]++; return; }
    int r = d->nefc++;
    memset(d->efc_J + (size_t)r*nv, 0, sizeof(double) * (size_t)nv);
    d->efc_J[(size_t)r*nv + i] = 1;
    d->efc_pos[r] = 0; d->efc_margin[r] = 0; d->efc_type[r] = CT_FRICTION_DOF; d->efc_id[r] = i;
  }
  /* joint limits */
  if (!(m->opt_disableflags & DMC_DSBL_LIMIT)) for (int j = 0; j < m->njnt; j++) {
    if (!m->jnt_limited[j]) continue;
    int t = m->jnt_type[j];
    if (t == DMC_JNT_BALL) {
      /* mj_instantiateLimit, ball: the rotation angle (mju_quat2Vel: axis = unit vec of the quaternion's vector part, angle
       * = 2 atan2(|vec|, w) taken in (-pi, pi]; a negative angle flips the axis) against max(range); one row, J = -axis on
       * the joint's three dofs */
      const double* q = d->qpos + m->jnt_qposadr[j];
      double qn = sqrt(q[0]*q[0] + q[1]*q[1] + q[2]*q[2] + q[3]*q[3]);
      double w = q[0]/qn, ax[3] = {q[1]/qn, q[2]/qn, q[3]/qn};
      double sn = sqrt(ax[0]*ax[0] + ax[1]*ax[1] + ax[2]*ax[2]);
      double angle = 2*atan2(sn, w);
      if (angle > DMC_PI) angle -= 2*DMC_PI;
      if (sn < MINVAL) { ax[0] = ax[1] = ax[2] = 0; angle = 0; } else for (int k = 0; k < 3; k++) ax[k] /= sn;
      if (angle < 0) { angle = -angle; for (int k = 0; k < 3; k++) ax[k] = -ax[k]; }
      double margin = m->jnt_margin[j];
      double dist = mjMAX(m->jnt_range[2*j], m->jnt_range[2*j + 1]) - angle;
      if (dist < margin) {
        if (d->nefc >= m->njmax) { d->warning[DMC_WARN_CNSTRFULL]++; return; }
        int r = d->nefc++;
        memset(d->efc_J + (size_t)r*nv, 0, sizeof(double) * (size_t)nv);
        for (int k = 0; k < 3; k++) d->efc_J[(size_t)r*nv + m->jnt_dofadr[j] + k] = -ax[k];
        d->efc_pos[r] = dist; d->efc_margin[r] = margin; d->efc_type[r] = CT_LIMIT; d->efc_id[r] = j;
      }
      continue;
    }
    if (t != DMC_JNT_SLIDE && t != DMC_JNT_HINGE) continue; /* (free joints have no limits) */
    double value = d->qpos[m->jnt_qposadr[j]], margin = m->jnt_margin[j];
    for (int side = -1; side <= 1; side += 2) {
      double dist = side * (m->jnt_range[2*j + (side + 1)/2] - value);
      if (dist < margin) {
        if (d->nefc >= m->njmax) { d->warning[DMC_WARN_CNSTRFULL]++; return; }
        int r = d->nefc++;
        memset(d->efc_J + (size_t)r*nv, 0, sizeof(double) * (size_t)nv);
        d->efc_J[(size_t)r*nv + m->jnt_dofadr[j]] = -(double)side;
        d->efc_pos[r] = dist; d->efc_margin[r] = margin; d->efc_type[r] = CT_LIMIT; d->efc_id[r] = j;
      }
    }
  }
  /* tendon limits (after the joint limits, as in MuJoCo) */
  if (!(m->opt_disableflags & DMC_DSBL_LIMIT)) for (int t = 0; t < m->ntendon; t++) {
    if (!m->tendon_limited[t]) continue;
    double value = d->ten_length[t], margin = m->tendon_margin[t];
    for (int side = -1; side <= 1; side += 2) {
      double dist = side * (m->tendon_range[2*t + (side + 1)/2] - value);
      if (dist < margin) {
        if (d->nefc >= m->njmax) { d->warning[DMC_WARN_CNSTRFULL]++; return; }
        int r = d->nefc++;
        for (int k = 0; k < nv; k++) d->efc_J[(size_t)r*nv + k] = -(double)side * d->ten_J[(size_t)t*nv + k];
        d->efc_pos[r] = dist; d->efc_margin[r] = margin; d->efc_type[r] = CT_LIMIT_TENDON; d->efc_id[r] = t;
      }
    }
  }
  /* contacts */
  if (!(m->opt_disableflags & DMC_DSBL_CONTACT)) for (int ci = 0; ci < d->ncon; ci++) {
    Contact* c = d->contact + ci;
    if (c->exclude) continue;
    int dim = c->dim, b1 = m->geom_bodyid[c->geom1], b2 = m->geom_bodyid[c->geom2];
    int elliptic = m->opt_cone == DMC_CONE_ELLIPTIC && dim > 1;
    int nrow = dim == 1 ? 1 : (elliptic ? dim : 2*(dim - 1));
    if (d->nefc + nrow > m->njmax) { d->warning[DMC_WARN_CNSTRFULL]++; return; }
    /* contact-frame Jacobian difference: rows 0..2 translational, 3..5 rotational */
    double* jac = d->w_tmp; /* 6*nv */
    for (int k = 0; k < nv; k++) {
      double jp1[3], jr1[3], jp2[3], jr2[3], dp[3], dr[3];
      jac_col(m, d, b1, c->pos, k, jp1, jr1);
      jac_col(m, d, b2, c->pos, k, jp2, jr2);
      for (int a = 0; a < 3; a++) { dp[a] = jp2[a] - jp1[a]; dr[a] = jr2[a] - jr1[a]; }
      for (int a = 0; a < 3; a++) {
        jac[a*nv + k] = dot3(c->frame + 3*a, dp);
        jac[(3 + a)*nv + k] = dot3(c->frame + 3*a, dr);
      }
    }
    c->efc_address = d->nefc;
    if (dim == 1) {
      int r = d->nefc++;
      memcpy(d->efc_J + (size_t)r*nv, jac, sizeof(double) * (size_t)nv);
      d->efc_pos[r] = c->dist; d->efc_margin[r] = c->includemargin; d->efc_type[r] = CT_FRICTIONLESS; d->efc_id[r] = ci;
    } else if (elliptic) {
      /* one row per contact-frame axis: normal carries (dist, margin), friction rows (0, 0) */
      for (int k = 0; k < dim; k++) {
        int r = d->nefc++;
        memcpy(d->efc_J + (size_t)r*nv, jac + (size_t)k*nv, sizeof(double) * (size_t)nv);
        d->efc_pos[r] = k == 0 ? c->dist : 0; d->efc_margin[r] = k == 0 ? c->includemargin : 0;
        d->efc_type[r] = CT_ELLIPTIC; d->efc_id[r] = ci;
      }
    } else {
      for (int k = 1; k < dim; k++) for (int s = 0; s < 2; s++) {
        int r = d->nefc++;
        double f = s == 0 ? c->friction[k - 1] : -c->friction[k - 1];
        for (int a = 0; a < nv; a++) d->efc_J[(size_t)r*nv + a] = jac[a] + f*jac[k*nv + a];
        d->efc_pos[r] = c->dist; d->efc_margin[r] = c->includemargin; d->efc_type[r] = CT_PYRAMIDAL; d->efc_id[r] = ci;
      }
    }
  }
  /* diagApprox, impedance, R, D, KBIP, aref */
  int nefc = d->nefc;
  for (int i = 0; i < nefc; i++) {
    const double *solref, *solimp; double dA;
    if (d->efc_type[i] == CT_EQUALITY) {
      int k = d->efc_id[i];
      solref = m->eq_solref + 2*k; solimp = m->eq_solimp + 5*k;
      const int et = m->eq_type[k];
      if (et == DMC_EQ_TENDON) dA = m->tendon_invweight0[m->eq_obj1id[k]];
      else if (et == DMC_EQ_JOINT) {
        dA = m->dof_invweight0[m->jnt_dofadr[m->eq_obj1id[k]]];
        if (m->eq_obj2id[k] >= 0) dA += m->dof_invweight0[m->jnt_dofadr[m->eq_obj2id[k]]];
      } else {
        /* rows 0..2 translational, 3..5 (weld) rotational: sums of the two bodies' inverse weights */
        int first = i; while (first > 0 && d->efc_type[first - 1] == CT_EQUALITY && d->efc_id[first - 1] == k) first--;
        const int col = (i - first) < 3 ? 0 : 1;
        dA = m->body_invweight0[2*m->eq_obj1id[k] + col] + m->body_invweight0[2*m->eq_obj2id[k] + col];
      }
    } else if (d->efc_type[i] == CT_FRICTION_DOF) {
      int k = d->efc_id[i];
      solref = m->dof_solref + 2*k; solimp = m->dof_solimp + 5*k;
      dA = m->dof_invweight0[k];
    } else if (d->efc_type[i] == CT_LIMIT) {
      int j = d->efc_id[i];
      solref = m->jnt_solref + 2*j; solimp = m->jnt_solimp + 5*j;
      dA = m->dof_invweight0[m->jnt_dofadr[j]];
    } else if (d->efc_type[i] == CT_LIMIT_TENDON) {
      int t = d->efc_id[i];
      solref = m->tendon_solref_lim + 2*t; solimp = m->tendon_solimp_lim + 5*t;
      dA = m->tendon_invweight0[t];
    } else {
      const Contact* c = d->contact + d->efc_id[i];
      int b1 = m->geom_bodyid[c->geom1], b2 = m->geom_bodyid[c->geom2];
      double tran = m->body_invweight0[2*b1] + m->body_invweight0[2*b2];
      double rot = m->body_invweight0[2*b1 + 1] + m->body_invweight0[2*b2 + 1];
      solref = c->solref; solimp = c->solimp;
      if (d->efc_type[i] == CT_FRICTIONLESS) dA = tran;
      else if (d->efc_type[i] == CT_ELLIPTIC) dA = (i - c->efc_address) < 3 ? tran : rot;
      else { int j = i - c->efc_address; double fri = c->friction[j/2]; dA = tran + fri*fri*(j < 4 ? tran : rot); }
    }
    d->efc_diagApprox[i] = dA;
    double ref[2] = {solref[0], solref[1]};
    if (!(m->opt_disableflags & DMC_DSBL_REFSAFE) && ref[0] > 0) ref[0] = mjMAX(ref[0], 2*m->opt_timestep);
    double imp; get_impedance(solimp, d->efc_pos[i], d->efc_margin[i], &imp);
    d->efc_R[i] = mjMAX(MINVAL, (1 - imp)*dA/imp);
    double dmax = mjMAX(DMC_MINIMP, mjMIN(DMC_MAXIMP, solimp[1])), K, B;
    if (ref[0] > 0) { K = 1 / mjMAX(MINVAL, dmax*dmax*ref[0]*ref[0]*ref[1]*ref[1]); B = 2 / mjMAX(MINVAL, dmax*ref[0]); }
    else { K = -ref[0] / mjMAX(MINVAL, dmax*dmax); B = -ref[1] / mjMAX(MINVAL, dmax); }
    if (d->efc_type[i] == CT_FRICTION_DOF) K = 0; /* friction rows have no position term */
    d->efc_KBIP[4*i] = K; d->efc_KBIP[4*i + 1] = B; d->efc_KBIP[4*i + 2] = imp; d->efc_KBIP[4*i + 3] = 0;
  }
  /* frictional contacts: the friction rows' regularisation is tied to the normal's.
   * R(first friction) = R(normal)/impratio; regularised cone mu = friction[0]*sqrt(R1/R0);
   * elliptic: R_j mu_j^2 is the same for every friction row; pyramidal: all edges
   * share Rpy = 2 mu^2 R(first edge). */
  for (int i = 0; i < nefc; i++) if (d->efc_type[i] == CT_PYRAMIDAL || d->efc_type[i] == CT_ELLIPTIC) {
    Contact* c = d->contact + d->efc_id[i];
    double R1 = d->efc_R[i] / mjMAX(MINVAL, m->opt_impratio);
    c->mu = c->friction[0] * sqrt(R1 / d->efc_R[i]);
    if (d->efc_type[i] == CT_ELLIPTIC) {
      d->efc_R[i + 1] = R1;
      for (int j = 2; j < c->dim; j++)
        d->efc_R[i + j] = R1 * c->friction[0]*c->friction[0] / (c->friction[j - 1]*c->friction[j - 1]);
      i += c->dim - 1;
    } else {
      int n = 2*(c->dim - 1);
      double Rpy = 2 * c->mu * c->mu * d->efc_R[i];
      for (int j = 0; j < n; j++) d->efc_R[i + j] = Rpy;
      i += n - 1;
    }
  }
  for (int i = 0; i < nefc; i++) {
    d->efc_D[i] = 1 / d->efc_R[i];
    d->efc_vel[i] = dot_n(d->efc_J + (size_t)i*nv, d->qvel, nv);
    d->efc_aref[i] = -d->efc_KBIP[4*i + 1]*d->efc_vel[i]
                     - d->efc_KBIP[4*i]*d->efc_KBIP[4*i + 2]*(d->efc_pos[i] - d->efc_margin[i]);
  }
}

/* fixed tendon t: sum_k prm_k * v[index of wrapped joint k] (length from qpos, velocity from qvel) */
static double tendon_dot(const Model* m, int t, const double* v, int use_dof) {
  double s = 0;
  for (int w = m->tendon_adr[t]; w < m->tendon_adr[t] + m->tendon_num[t]; w++) {
    int j = m->wrap_objid[w];
    s += m->wrap_prm[w] * v[use_dof ? m->jnt_dofadr[j] : m->jnt_qposadr[j]];
  }
  return s;
}
static void transmission(const Model* m, Data* d) {
  for (int i = 0; i < m->nu; i++) {
    int j = m->actuator_trnid[2*i];
    if (m->actuator_trntype[i] == DMC_TRN_TENDON) d->actuator_length[i] = m->actuator_gear[6*i] * tendon_dot(m, j, d->qpos, 0);
    else d->actuator_length[i] = m->actuator_gear[6*i] * d->qpos[m->jnt_qposadr[j]];
  }
}

/* ------------------------------------------------------------------------- */
/* velocity stage                                                             */
/* ------------------------------------------------------------------------- */
static void com_vel(const Model* m, Data* d) {
  memset(d->cvel, 0, 6 * sizeof(double));
  for (int i = 1; i < m->nbody; i++) {
    double cvel[6], tmp[6];
    memcpy(cvel, d->cvel + 6*m->body_parentid[i], sizeof cvel);
    int bda = m->body_dofadr[i];
    int dofs = 0;
    for (int j = m->body_jntadr[i]; j < m->body_jntadr[i] + m->body_jntnum[i]; j++) {
      int t = m->jnt_type[j];
      if (t == DMC_JNT_FREE) {
        memset(d->cdof_dot + 6*bda, 0, 18 * sizeof(double));
        for (int k = 0; k < 3; k++) for (int a = 0; a < 6; a++) cvel[a] += d->cdof[6*(bda + k) + a] * d->qvel[bda + k];
        dofs += 3;
      }
      if (t == DMC_JNT_FREE || t == DMC_JNT_BALL) {
        for (int k = 0; k < 3; k++) cross_motion(d->cdof_dot + 6*(bda + dofs + k), cvel, d->cdof + 6*(bda + dofs + k));
        for (int a = 0; a < 6; a++) tmp[a] = 0;
        for (int k = 0; k < 3; k++) for (int a = 0; a < 6; a++) tmp[a] += d->cdof[6*(bda + dofs + k) + a] * d->qvel[bda + dofs + k];
        for (int a = 0; a < 6; a++) cvel[a] += tmp[a];
        dofs += 3;
      } else {
        cross_motion(d->cdof_dot + 6*(bda + dofs), cvel, d->cdof + 6*(bda + dofs));
        for (int a = 0; a < 6; a++) cvel[a] += d->cdof[6*(bda + dofs) + a] * d->qvel[bda + dofs];
        dofs += 1;
      }
    }
    memcpy(d->cvel + 6*i, cvel, sizeof cvel);
  }
}
#define DMC_PI 3.14159265358979323846
static void object_velocity(const Model* m, const Data* d, int body, const double* pos, const double* mat, int local, double* res);
static void passive(const Model* m, Data* d) {
  int nv = m->nv;
  memset(d->qfrc_passive, 0, sizeof(double) * (size_t)nv);
  if (!(m->opt_disableflags & DMC_DSBL_SPRING)) for (int j = 0; j < m->njnt; j++) {
    double k = m->jnt_stiffness[j];
    if (k == 0) continue;
    int t = m->jnt_type[j], qa = m->jnt_qposadr[j], da = m->jnt_dofadr[j];
    if (t == DMC_JNT_SLIDE || t == DMC_JNT_HINGE) d->qfrc_passive[da] -= k * (d->qpos[qa] - m->qpos_spring[qa]);
    /* free/ball springs: not present in the supported models */
  }
  if (!(m->opt_disableflags & DMC_DSBL_DAMPER)) for (int i = 0; i < nv; i++)
    d->qfrc_passive[i] -= m->dof_damping[i] * d->qvel[i];
  /* fixed-tendon spring / damper */
  for (int t = 0; t < m->ntendon; t++) {
    double f = 0;
    if (m->tendon_stiffness[t] != 0 && !(m->opt_disableflags & DMC_DSBL_SPRING))
      f -= m->tendon_stiffness[t] * (tendon_dot(m, t, d->qpos, 0) - m->tendon_lengthspring[t]);
    if (m->tendon_damping[t] != 0 && !(m->opt_disableflags & DMC_DSBL_DAMPER))
      f -= m->tendon_damping[t] * tendon_dot(m, t, d->qvel, 1);
    if (f != 0) for (int w = m->tendon_adr[t]; w < m->tendon_adr[t] + m->tendon_num[t]; w++)
      d->qfrc_passive[m->jnt_dofadr[m->wrap_objid[w]]] += m->wrap_prm[w] * f;
  }
  /* fluid forces, inertia-box model: every body with mass is replaced by the box with the
   * same inertia; viscous (Stokes, sphere of the mean box size) and quadratic drag terms act
   * on its local inertial-frame velocity; the wrench is applied at the body's COM */
  if (m->opt_density > 0 || m->opt_viscosity > 0) for (int i = 1; i < m->nbody; i++) {
    double mass = m->body_mass[i];
    if (mass < MINVAL) continue;
    const double* I = m->body_inertia + 3*i;
    double box[3] = {sqrt(mjMAX(MINVAL, I[1] + I[2] - I[0]) / mass * 6.0),
                     sqrt(mjMAX(MINVAL, I[0] + I[2] - I[1]) / mass * 6.0),
                     sqrt(mjMAX(MINVAL, I[0] + I[1] - I[2]) / mass * 6.0)};
    double lvel[6], lfrc[6] = {0, 0, 0, 0, 0, 0}, bf[3], bt[3];
    object_velocity(m, d, i, d->xipos + 3*i, d->ximat + 9*i, 1, lvel);
    if (m->opt_viscosity > 0) {
      double diam = (box[0] + box[1] + box[2]) / 3.0;
      for (int k = 0; k < 3; k++) {
        lfrc[k] = -DMC_PI*diam*diam*diam*m->opt_viscosity * lvel[k];
        lfrc[3 + k] = -3.0*DMC_PI*diam*m->opt_viscosity * lvel[3 + k];
      }
    }
    if (m->opt_density > 0) {
      double rho = m->opt_density;
      lfrc[3] -= 0.5*rho*box[1]*box[2]*fabs(lvel[3])*lvel[3];
      lfrc[4] -= 0.5*rho*box[0]*box[2]*fabs(lvel[4])*lvel[4];
      lfrc[5] -= 0.5*rho*box[0]*box[1]*fabs(lvel[5])*lvel[5];
      lfrc[0] -= rho*box[0]*(box[1]*box[1]*box[1]*box[1] + box[2]*box[2]*box[2]*box[2])*fabs(lvel[0])*lvel[0]/64.0;
      lfrc[1] -= rho*box[1]*(box[0]*box[0]*box[0]*box[0] + box[2]*box[2]*box[2]*box[2])*fabs(lvel[1])*lvel[1]/64.0;
      lfrc[2] -= rho*box[2]*(box[0]*box[0]*box[0]*box[0] + box[1]*box[1]*box[1]*box[1])*fabs(lvel[2])*lvel[2]/64.0;
    }
    mul_mat_vec3(bt, d->ximat + 9*i, lfrc); mul_mat_vec3(bf, d->ximat + 9*i, lfrc + 3);
    for (int k = 0; k < nv; k++) {
      double jp[3], jr[3];
      jac_col(m, d, i, d->xipos + 3*i, k, jp, jr);
      d->qfrc_passive[k] += dot3(jp, bf) + dot3(jr, bt);
    }
  }
}
/* mj_rne with flg_acc = 0 */
static void rne(const Model* m, Data* d) {
  int nbody = m->nbody, nv = m->nv;
  double* loc_cacc = d->w_tmp;                 /* 6*nbody */
  double* loc_cfrc = d->w_tmp + 6*nbody;       /* 6*nbody */
  memset(loc_cacc, 0, 6 * sizeof(double));
  if (!(m->opt_disableflags & DMC_DSBL_GRAVITY)) {
    loc_cacc[3] = -m->opt_gravity_x; loc_cacc[4] = -m->opt_gravity_y; loc_cacc[5] = -m->opt_gravity_z;
  }
  memset(loc_cfrc, 0, 6 * sizeof(double));
  for (int i = 1; i < nbody; i++) {
    int bda = m->body_dofadr[i];
    double tmp[6], tmp1[6];
    for (int a = 0; a < 6; a++) tmp[a] = 0;
    for (int k = 0; k < m->body_dofnum[i]; k++) for (int a = 0; a < 6; a++) tmp[a] += d->cdof_dot[6*(bda + k) + a] * d->qvel[bda + k];
    for (int a = 0; a < 6; a++) loc_cacc[6*i + a] = loc_cacc[6*m->body_parentid[i] + a] + tmp[a];
    mul_inert_vec(loc_cfrc + 6*i, d->cinert + 10*i, loc_cacc + 6*i);
    mul_inert_vec(tmp, d->cinert + 10*i, d->cvel + 6*i);
    cross_force(tmp1, d->cvel + 6*i, tmp);
    for (int a = 0; a < 6; a++) loc_cfrc[6*i + a] += tmp1[a];
  }
  for (int i = nbody - 1; i > 0; i--) if (m->body_parentid[i])
    for (int a = 0; a < 6; a++) loc_cfrc[6*m->body_parentid[i] + a] += loc_cfrc[6*i + a];
  for (int i = 0; i < nv; i++) d->qfrc_bias[i] = dot_n(d->cdof + 6*i, loc_cfrc + 6*m->dof_bodyid[i], 6);
}
static void subtree_vel(const Model* m, Data* d) {
  int nbody = m->nbody;
  for (int i = 0; i < nbody; i++) {
    const double* rc = d->subtree_com + 3*m->body_rootid[i];
    double dif[3] = {d->xipos[3*i] - rc[0], d->xipos[3*i + 1] - rc[1], d->xipos[3*i + 2] - rc[2]}, tmp[3];
    cross3(tmp, dif, d->cvel + 6*i);
    for (int k = 0; k < 3; k++) d->subtree_linvel[3*i + k] = m->body_mass[i] * (d->cvel[6*i + 3 + k] - tmp[k]);
  }
  for (int i = nbody - 1; i >= 0; i--) {
    if (i) for (int k = 0; k < 3; k++) d->subtree_linvel[3*m->body_parentid[i] + k] += d->subtree_linvel[3*i + k];
    double s = 1 / mjMAX(MINVAL, m->body_subtreemass[i]);
    for (int k = 0; k < 3; k++) d->subtree_linvel[3*i + k] *= s;
  }
}

/* ------------------------------------------------------------------------- */
/* sensors                                                                    */
/* ------------------------------------------------------------------------- */
/* 6D velocity [rot; lin] of a frame at `pos` attached to `body`; local != 0 rotates into `mat` */
static void object_velocity(const Model* m, const Data* d, int body, const double* pos, const double* mat, int local, double* res) {
  const double* rc = d->subtree_com + 3*m->body_rootid[body];
  double dif[3] = {pos[0] - rc[0], pos[1] - rc[1], pos[2] - rc[2]}, tmp[3];
  const double* cv = d->cvel + 6*body;
  cross3(tmp, dif, cv);
  double lin[3] = {cv[3] - tmp[0], cv[4] - tmp[1], cv[5] - tmp[2]};
  if (local) { mul_matT_vec3(res, mat, cv); mul_matT_vec3(res + 3, mat, lin); }
  else { memcpy(res, cv, 3 * sizeof(double)); memcpy(res + 3, lin, 3 * sizeof(double)); }
}
void ora_object_velocity(const Model* m, const Data* d, int objtype, int id, int local, double* res) {
  if (objtype == DMC_OBJ_SITE) object_velocity(m, d, m->site_bodyid[id], d->site_xpos + 3*id, d->site_xmat + 9*id, local, res);
  else object_velocity(m, d, id, d->xpos + 3*id, d->xmat + 9*id, local, res);
}
static double ray_geom(const double* pos, const double* mat, const double* size, const double* pnt, const double* vec, int type);
/* pose of a sensor's reference frame (mj_sensorPos: get_xpos_xmat / get_xquat of the reference object) */
static void sensor_ref_pose(const Model* m, const Data* d, int rt, int rid, const double** p, const double** R, double* q) {
  if (rt == DMC_OBJ_SITE) { *p = d->site_xpos + 3*rid; *R = d->site_xmat + 9*rid; mul_quat(q, d->xquat + 4*m->site_bodyid[rid], m->site_quat + 4*rid); }
  else if (rt == DMC_OBJ_GEOM) { *p = d->geom_xpos + 3*rid; *R = d->geom_xmat + 9*rid; mul_quat(q, d->xquat + 4*m->geom_bodyid[rid], m->geom_quat + 4*rid); }
  else if (rt == DMC_OBJ_BODY) { *p = d->xipos + 3*rid; *R = d->ximat + 9*rid; mul_quat(q, d->xquat + 4*rid, m->body_iquat + 4*rid); }
  else { *p = d->xpos + 3*rid; *R = d->xmat + 9*rid; memcpy(q, d->xquat + 4*rid, 4 * sizeof(double)); }
}
static void sensor_stage(const Model* m, Data* d, int stage) {
  if (m->opt_disableflags & DMC_DSBL_SENSOR) return;
  int need_subtree = 0;
  for (int i = 0; i < m->nsensor; i++) if (m->sensor_needstage[i] == stage && m->sensor_type[i] == DMC_SENS_SUBTREELINVEL) need_subtree = 1;
  if (need_subtree) subtree_vel(m, d);
  for (int i = 0; i < m->nsensor; i++) {
    if (m->sensor_needstage[i] != stage) continue;
    double* out = d->sensordata + m->sensor_adr[i];
    int id = m->sensor_objid[i];
    double v6[6];
    switch (m->sensor_type[i]) {
      case DMC_SENS_JOINTPOS: out[0] = d->qpos[m->jnt_qposadr[id]]; break;
      case DMC_SENS_JOINTVEL: out[0] = d->qvel[m->jnt_dofadr[id]]; break;
      case DMC_SENS_ACTUATORFRC: out[0] = d->actuator_force[id]; break;
      case DMC_SENS_SUBTREECOM: memcpy(out, d->subtree_com + 3*id, 3 * sizeof(double)); break;
      case DMC_SENS_FRAMEXAXIS: case DMC_SENS_FRAMEYAXIS: case DMC_SENS_FRAMEZAXIS: {
        /* column of the object's world rotation matrix */
        const double* R = m->sensor_objtype[i] == DMC_OBJ_SITE ? d->site_xmat + 9*id
                        : m->sensor_objtype[i] == DMC_OBJ_GEOM ? d->geom_xmat + 9*id
                        : m->sensor_objtype[i] == DMC_OBJ_BODY ? d->ximat + 9*id : d->xmat + 9*id;
        int c = m->sensor_type[i] - DMC_SENS_FRAMEXAXIS;
        out[0] = R[c]; out[1] = R[3 + c]; out[2] = R[6 + c];
        if (m->sensor_refid[i] >= 0) {      /* the axis in the reference frame: R_ref' axis */
          const double *pr, *Rr; double qr[4], ax[3] = {out[0], out[1], out[2]};
          sensor_ref_pose(m, d, m->sensor_reftype[i], m->sensor_refid[i], &pr, &Rr, qr);
          for (int k = 0; k < 3; k++) out[k] = Rr[k]*ax[0] + Rr[3 + k]*ax[1] + Rr[6 + k]*ax[2];
        }
        break; }
      case DMC_SENS_FRAMEPOS: { /* position of the object's frame origin: world frame, or R_ref' (p - p_ref) */
        const double* p = m->sensor_objtype[i] == DMC_OBJ_SITE ? d->site_xpos + 3*id
                        : m->sensor_objtype[i] == DMC_OBJ_GEOM ? d->geom_xpos + 3*id
                        : m->sensor_objtype[i] == DMC_OBJ_BODY ? d->xipos + 3*id : d->xpos + 3*id;
        memcpy(out, p, 3 * sizeof(double));
        if (m->sensor_refid[i] >= 0) {
          const double *pr, *Rr; double qr[4], df[3];
          sensor_ref_pose(m, d, m->sensor_reftype[i], m->sensor_refid[i], &pr, &Rr, qr);
          for (int k = 0; k < 3; k++) df[k] = p[k] - pr[k];
          for (int k = 0; k < 3; k++) out[k] = Rr[k]*df[0] + Rr[3 + k]*df[1] + Rr[6 + k]*df[2];
        }
        break; }
      case DMC_SENS_SUBTREELINVEL: memcpy(out, d->subtree_linvel + 3*id, 3 * sizeof(double)); break;
      case DMC_SENS_RANGEFINDER: { /* mj_ray along the site's z axis: nearest visible geom not on the site's body, -1 if none */
        const double* R = d->site_xmat + 9*id;
        const double vec[3] = {R[2], R[5], R[8]};
        double best = -1;
        for (int g = 0; g < m->ngeom; g++) {
          if (m->geom_bodyid[g] == m->site_bodyid[id] || m->geom_invisible[g]) continue;
          const double x = ray_geom(d->geom_xpos + 3*g, d->geom_xmat + 9*g, m->geom_size + 3*g, d->site_xpos + 3*id, vec, m->geom_type[g]);
          if (x >= 0 && (best < 0 || x < best)) best = x;
        }
        out[0] = best; break; }
      case DMC_SENS_FRAMEQUAT: { /* orientation of the object's frame: body quaternion times the local one */
        const int ot = m->sensor_objtype[i];
        if (ot == DMC_OBJ_SITE) mul_quat(out, d->xquat + 4*m->site_bodyid[id], m->site_quat + 4*id);
        else if (ot == DMC_OBJ_GEOM) mul_quat(out, d->xquat + 4*m->geom_bodyid[id], m->geom_quat + 4*id);
        else if (ot == DMC_OBJ_BODY) mul_quat(out, d->xquat + 4*id, m->body_iquat + 4*id);
        else memcpy(out, d->xquat + 4*id, 4 * sizeof(double));
        if (m->sensor_refid[i] >= 0) {      /* conj(q_ref) q */
          const double *pr, *Rr; double qr[4], qo[4] = {out[0], out[1], out[2], out[3]};
          sensor_ref_pose(m, d, m->sensor_reftype[i], m->sensor_refid[i], &pr, &Rr, qr);
          qr[1] = -qr[1]; qr[2] = -qr[2]; qr[3] = -qr[3];
          mul_quat(out, qr, qo);
        }
        break; }
      case DMC_SENS_FRAMELINVEL: case DMC_SENS_FRAMEANGVEL: { /* mj_objectVelocity, world orientation */
        const int ot = m->sensor_objtype[i];
        const int body = ot == DMC_OBJ_SITE ? m->site_bodyid[id] : ot == DMC_OBJ_GEOM ? m->geom_bodyid[id] : id;
        const double* p = ot == DMC_OBJ_SITE ? d->site_xpos + 3*id : ot == DMC_OBJ_GEOM ? d->geom_xpos + 3*id
                        : ot == DMC_OBJ_BODY ? d->xipos + 3*id : d->xpos + 3*id;
        object_velocity(m, d, body, p, NULL, 0, v6);
        if (m->sensor_refid[i] >= 0) {
          /* relative to a moving reference frame (mj_sensorVel): the time derivative of the pose the position-stage
           * sensor reports, R_ref' (v - v_ref + r x w_ref) with r = p - p_ref, and R_ref' (w - w_ref) */
          const int rt = m->sensor_reftype[i], rid = m->sensor_refid[i];
          const int rbody = rt == DMC_OBJ_SITE ? m->site_bodyid[rid] : rt == DMC_OBJ_GEOM ? m->geom_bodyid[rid] : rid;
          const double *pr, *Rr; double qr[4], vr[6], rel[6], r[3], cr[3];
          sensor_ref_pose(m, d, rt, rid, &pr, &Rr, qr);
          object_velocity(m, d, rbody, pr, NULL, 0, vr);
          for (int k = 0; k < 6; k++) rel[k] = v6[k] - vr[k];
          for (int k = 0; k < 3; k++) r[k] = p[k] - pr[k];
          cross3(cr, r, vr);
          for (int k = 0; k < 3; k++) rel[3 + k] += cr[k];
          for (int k = 0; k < 3; k++) {
            v6[k] = Rr[k]*rel[0] + Rr[3 + k]*rel[1] + Rr[6 + k]*rel[2];
            v6[3 + k] = Rr[k]*rel[3] + Rr[3 + k]*rel[4] + Rr[6 + k]*rel[5];
          }
        }
        memcpy(out, m->sensor_type[i] == DMC_SENS_FRAMELINVEL ? v6 + 3 : v6, 3 * sizeof(double)); break; }
      case DMC_SENS_VELOCIMETER:
        object_velocity(m, d, m->site_bodyid[id], d->site_xpos + 3*id, d->site_xmat + 9*id, 1, v6);
        memcpy(out, v6 + 3, 3 * sizeof(double)); break;
      case DMC_SENS_GYRO:
        object_velocity(m, d, m->site_bodyid[id], d->site_xpos + 3*id, d->site_xmat + 9*id, 1, v6);
        memcpy(out, v6, 3 * sizeof(double)); break;
      default: break; /* acceleration-stage site sensors: see sensor_acc() */
    }
  }
}
/* mj_rnePostConstraint: cacc, cfrc_int, cfrc_ext including contact forces */
static void contact_force_local(const Model* m, const Data* d, int id, double* f6);
static void rne_post_constraint(const Model* m, Data* d) {
  int nbody = m->nbody;
  memset(d->cfrc_ext, 0, sizeof(double) * 6 * (size_t)nbody);
  for (int i = 1; i < nbody; i++) {
    const double* xf = d->xfrc_applied + 6*i;
    int nz = 0; for (int a = 0; a < 6; a++) if (xf[a] != 0) nz = 1;
    if (!nz) continue;
    /* xfrc_applied = [force, torque] at the body COM (xipos) */
    const double* rc = d->subtree_com + 3*m->body_rootid[i];
    double dif[3] = {d->xipos[3*i] - rc[0], d->xipos[3*i + 1] - rc[1], d->xipos[3*i + 2] - rc[2]}, t[3];
    cross3(t, dif, xf);
    for (int k = 0; k < 3; k++) { d->cfrc_ext[6*i + k] += xf[3 + k] + t[k]; d->cfrc_ext[6*i + 3 + k] += xf[k]; }
  }
  for (int ci = 0; ci < d->ncon; ci++) {
    const Contact* c = d->contact + ci;
    if (c->efc_address < 0) continue;
    double lf[6], gf[3], gt[3];
    contact_force_local(m, d, ci, lf);
    mul_matT_vec3(gf, c->frame, lf); mul_matT_vec3(gt, c->frame, lf + 3);
    int b[2] = {m->geom_bodyid[c->geom1], m->geom_bodyid[c->geom2]};
    for (int s = 0; s < 2; s++) {
      if (!b[s]) continue;
      double sign = s == 0 ? -1 : 1;
      const double* rc = d->subtree_com + 3*m->body_rootid[b[s]];
      double dif[3] = {c->pos[0] - rc[0], c->pos[1] - rc[1], c->pos[2] - rc[2]}, t[3];
      cross3(t, dif, gf);
      for (int k = 0; k < 3; k++) { d->cfrc_ext[6*b[s] + k] += sign*(gt[k] + t[k]); d->cfrc_ext[6*b[s] + 3 + k] += sign*gf[k]; }
    }
  }
  memset(d->cacc, 0, 6 * sizeof(double));
  if (!(m->opt_disableflags & DMC_DSBL_GRAVITY)) { d->cacc[3] = -m->opt_gravity_x; d->cacc[4] = -m->opt_gravity_y; d->cacc[5] = -m->opt_gravity_z; }
  memset(d->cfrc_int, 0, 6 * sizeof(double));
  for (int i = 1; i < nbody; i++) {
    int bda = m->body_dofadr[i];
    double csum[6], tmp[6], tmp1[6];
    for (int a = 0; a < 6; a++) csum[a] = d->cacc[6*m->body_parentid[i] + a];
    for (int k = 0; k < m->body_dofnum[i]; k++) for (int a = 0; a < 6; a++)
      csum[a] += d->cdof_dot[6*(bda + k) + a]*d->qvel[bda + k] + d->cdof[6*(bda + k) + a]*d->qacc[bda + k];
    memcpy(d->cacc + 6*i, csum, sizeof csum);
    mul_inert_vec(tmp, d->cinert + 10*i, d->cacc + 6*i);
    mul_inert_vec(tmp1, d->cinert + 10*i, d->cvel + 6*i);
    double tmp2[6]; cross_force(tmp2, d->cvel + 6*i, tmp1);
    for (int a = 0; a < 6; a++) d->cfrc_int[6*i + a] = tmp[a] + tmp2[a] - d->cfrc_ext[6*i + a];
  }
  for (int i = nbody - 1; i > 0; i--) for (int a = 0; a < 6; a++) d->cfrc_int[6*m->body_parentid[i] + a] += d->cfrc_int[6*i + a];
}
/* ray (pnt, vec) vs site volume; returns distance or -1 (sphere, capsule, box) */
static double ray_geom(const double* pos, const double* mat, const double* size, const double* pnt, const double* vec, int type) {
  double dif[3] = {pnt[0] - pos[0], pnt[1] - pos[1], pnt[2] - pos[2]}, lp[3], lv[3];
  mul_matT_vec3(lp, mat, dif); mul_matT_vec3(lv, mat, vec);
  double best = -1;
  if (type == DMC_GEOM_SPHERE || type == DMC_GEOM_CAPSULE) {
    /* sphere(s): solve |lp + x lv - c|^2 = r^2 */
    double r = size[0];
    for (int part = 0; part < (type == DMC_GEOM_CAPSULE ? 3 : 1); part++) {
      double a, b, c;
      if (type == DMC_GEOM_CAPSULE && part == 0) { /* cylinder side */
        a = lv[0]*lv[0] + lv[1]*lv[1]; b = lp[0]*lv[0] + lp[1]*lv[1]; c = lp[0]*lp[0] + lp[1]*lp[1] - r*r;
      } else {
        double cz = type == DMC_GEOM_CAPSULE ? (part == 1 ? size[1] : -size[1]) : 0;
        double q[3] = {lp[0], lp[1], lp[2] - cz};
        a = dot3(lv, lv); b = dot3(q, lv); c = dot3(q, q) - r*r;
      }
      if (a < MINVAL) continue;
      double det = b*b - a*c;
      if (det < 0) continue;
      double sq = sqrt(det), xs[2] = {(-b - sq)/a, (-b + sq)/a};
      for (int k = 0; k < 2; k++) {
        double x = xs[k];
        if (x < 0) continue;
        double z = lp[2] + x*lv[2];
        if (type == DMC_GEOM_CAPSULE) {
          if (part == 0 && fabs(z) > size[1]) continue;
          if (part == 1 && z < size[1]) continue;
          if (part == 2 && z > -size[1]) continue;
        }
        if (best < 0 || x < best) best = x;
      }
    }
    /* point inside volume counts as hit at 0 */
    return best;
  }
  if (type == DMC_GEOM_ELLIPSOID) {
    /* (lp + x lv)' diag(1/size^2) (lp + x lv) = 1: the unit-sphere test in scaled coordinates */
    double q[3] = {lp[0]/size[0], lp[1]/size[1], lp[2]/size[2]}, w[3] = {lv[0]/size[0], lv[1]/size[1], lv[2]/size[2]};
    double a = dot3(w, w), b = dot3(q, w), c = dot3(q, q) - 1;
    if (a < MINVAL) return -1;
    double det = b*b - a*c;
    if (det < 0) return -1;
    double sq = sqrt(det), x0 = (-b - sq)/a, x1 = (-b + sq)/a;
    return x0 >= 0 ? x0 : (x1 >= 0 ? x1 : -1);
  }
  if (type == DMC_GEOM_PLANE) {
    /* front side only; a plane with positive half-sizes is finite for rays */
    if (lv[2] > -MINVAL) return -1;
    const double x = -lp[2]/lv[2];
    if (x < 0) return -1;
    const double px = lp[0] + x*lv[0], py = lp[1] + x*lv[1];
    if ((size[0] <= 0 || fabs(px) <= size[0]) && (size[1] <= 0 || fabs(py) <= size[1])) return x;
    return -1;
  }
  if (type == DMC_GEOM_CYLINDER) {
    /* side wall, then the two caps */
    const double a = lv[0]*lv[0] + lv[1]*lv[1], b = lp[0]*lv[0] + lp[1]*lv[1], c = lp[0]*lp[0] + lp[1]*lp[1] - size[0]*size[0];
    if (a >= MINVAL) {
      const double det = b*b - a*c;
      if (det >= 0) {
        const double sq = sqrt(det), xs[2] = {(-b - sq)/a, (-b + sq)/a};
        for (int k = 0; k < 2; k++) if (xs[k] >= 0 && fabs(lp[2] + xs[k]*lv[2]) <= size[1]) if (best < 0 || xs[k] < best) best = xs[k];
      }
    }
    if (fabs(lv[2]) >= MINVAL) for (int s = -1; s <= 1; s += 2) {
      const double x = (s*size[1] - lp[2]) / lv[2];
      if (x < 0) continue;
      const double px = lp[0] + x*lv[0], py = lp[1] + x*lv[1];
      if (px*px + py*py <= size[0]*size[0]) if (best < 0 || x < best) best = x;
    }
    return best;
  }
  if (type == DMC_GEOM_BOX) {
    /* nearest face crossing with x >= 0 (from inside: the exit face) */
    for (int ax = 0; ax < 3; ax++) {
      if (fabs(lv[ax]) < MINVAL) continue;
      for (int s = -1; s <= 1; s += 2) {
        double x = (s*size[ax] - lp[ax]) / lv[ax];
        if (x < 0) continue;
        int a1 = (ax + 1) % 3, a2 = (ax + 2) % 3;
        if (fabs(lp[a1] + x*lv[a1]) <= size[a1] && fabs(lp[a2] + x*lv[a2]) <= size[a2])
          if (best < 0 || x < best) best = x;
      }
    }
    return best;
  }
  return -1;
}
static void sensor_acc(const Model* m, Data* d) {
  if (m->opt_disableflags & DMC_DSBL_SENSOR) return;
  int need_rne = 0;
  for (int i = 0; i < m->nsensor; i++) if (m->sensor_needstage[i] == DMC_STAGE_ACC) {
    int t = m->sensor_type[i];
    if (t == DMC_SENS_ACCELEROMETER || t == DMC_SENS_FORCE || t == DMC_SENS_TORQUE) need_rne = 1;
  }
  if (need_rne) rne_post_constraint(m, d);
  for (int i = 0; i < m->nsensor; i++) {
    if (m->sensor_needstage[i] != DMC_STAGE_ACC) continue;
    double* out = d->sensordata + m->sensor_adr[i];
    int id = m->sensor_objid[i], t = m->sensor_type[i];
    if (t == DMC_SENS_ACTUATORFRC) { out[0] = d->actuator_force[id]; continue; }
    int body = m->site_bodyid[id];
    const double* spos = d->site_xpos + 3*id; const double* smat = d->site_xmat + 9*id;
    const double* rc = d->subtree_com + 3*m->body_rootid[body];
    if (t == DMC_SENS_TOUCH) {
      out[0] = 0;
      for (int ci = 0; ci < d->ncon; ci++) {
        const Contact* c = d->contact + ci;
        if (c->efc_address < 0) continue;
        int b1 = m->geom_bodyid[c->geom1], b2 = m->geom_bodyid[c->geom2];
        if (b1 != body && b2 != body) continue;
        double lf[6]; contact_force_local(m, d, ci, lf);
        if (lf[0] <= 0) continue;
        double ray[3] = {c->frame[0]*lf[0], c->frame[1]*lf[0], c->frame[2]*lf[0]};
        normalize3(ray);
        if (b2 == body) { ray[0] = -ray[0]; ray[1] = -ray[1]; ray[2] = -ray[2]; }
        if (ray_geom(spos, smat, m->site_size + 3*id, c->pos, ray, m->site_type[id]) >= 0) out[0] += lf[0];
      }
    } else if (t == DMC_SENS_ACCELEROMETER) {
      /* linear acceleration of the site frame, incl. Coriolis term, in site frame */
      double dif[3] = {spos[0] - rc[0], spos[1] - rc[1], spos[2] - rc[2]}, tmp[3], lin[3], vel[6], cor[3];
      const double* ca = d->cacc + 6*body;
      cross3(tmp, dif, ca);
      for (int k = 0; k < 3; k++) lin[k] = ca[3 + k] - tmp[k];
      object_velocity(m, d, body, spos, smat, 0, vel);
      cross3(cor, vel, vel + 3);
      for (int k = 0; k < 3; k++) lin[k] += cor[k];
      mul_matT_vec3(out, smat, lin);
    } else if (t == DMC_SENS_FORCE) {
      mul_matT_vec3(out, smat, d->cfrc_int + 6*body + 3);
    } else if (t == DMC_SENS_TORQUE) {
      double dif[3] = {spos[0] - rc[0], spos[1] - rc[1], spos[2] - rc[2]}, tmp[3], tq[3];
      cross3(tmp, dif, d->cfrc_int + 6*body + 3);
      for (int k = 0; k < 3; k++) tq[k] = d->cfrc_int[6*body + k] - tmp[k];
      mul_matT_vec3(out, smat, tq);
    }
  }
}

/* ------------------------------------------------------------------------- */
/* actuation / smooth acceleration                                            */
/* ------------------------------------------------------------------------- */
static void fwd_actuation(const Model* m, Data* d) {
  int nv = m->nv, nu = m->nu;
  memset(d->qfrc_actuator, 0, sizeof(double) * (size_t)nv);
  memset(d->actuator_force, 0, sizeof(double) * (size_t)nu);
  memset(d->act_dot, 0, sizeof(double) * (size_t)m->na);
  if (m->opt_disableflags & DMC_DSBL_ACTUATION) return;
  for (int i = 0; i < nu; i++) if (isnan(d->ctrl[i]) || fabs(d->ctrl[i]) > MAXVAL) {
    d->warning[DMC_WARN_BADCTRL]++;
    memset(d->ctrl, 0, sizeof(double) * (size_t)nu);
    break;
  }
  int kact = 0;
  for (int i = 0; i < nu; i++) {
    double ctrl = d->ctrl[i];
    if (m->actuator_ctrllimited[i] && !(m->opt_disableflags & DMC_DSBL_CLAMPCTRL))
      ctrl = mjMAX(m->actuator_ctrlrange[2*i], mjMIN(m->actuator_ctrlrange[2*i + 1], ctrl));
    const double *gp = m->actuator_gainprm + 10*i, *bp = m->actuator_biasprm + 10*i;
    double gain = gp[0], bias = 0;
    if (m->actuator_gaintype[i] == DMC_GAIN_AFFINE) gain = gp[0] + gp[1]*d->actuator_length[i] + gp[2]*d->actuator_velocity[i];
    if (m->actuator_biastype[i] == DMC_BIAS_AFFINE) bias = bp[0] + bp[1]*d->actuator_length[i] + bp[2]*d->actuator_velocity[i];
    /* actuators with dynamics: activation state act[k] (k-th stateful actuator) drives the gain;
     * integrator: act_dot = ctrl; filter / filterexact: act_dot = (ctrl - act) / tau */
    double input = ctrl;
    if (m->actuator_dyntype[i] != DMC_DYN_NONE) {
      const double act = d->act[kact];
      if (m->actuator_dyntype[i] == DMC_DYN_INTEGRATOR) d->act_dot[kact] = ctrl;
      else d->act_dot[kact] = (ctrl - act) / mjMAX(MINVAL, m->actuator_dynprm[10*i]);
      input = act;
      kact++;
    }
    double force = gain*input + bias;
    if (m->actuator_forcelimited[i]) force = mjMAX(m->actuator_forcerange[2*i], mjMIN(m->actuator_forcerange[2*i + 1], force));
    d->actuator_force[i] = force;
    if (m->actuator_trntype[i] == DMC_TRN_TENDON) {
      int t = m->actuator_trnid[2*i];
      for (int w = m->tendon_adr[t]; w < m->tendon_adr[t] + m->tendon_num[t]; w++)
        d->qfrc_actuator[m->jnt_dofadr[m->wrap_objid[w]]] += m->actuator_gear[6*i] * m->wrap_prm[w] * force;
    } else d->qfrc_actuator[m->jnt_dofadr[m->actuator_trnid[2*i]]] += m->actuator_gear[6*i] * force;
  }
}
static void fwd_acceleration(const Model* m, Data* d) {
  int nv = m->nv;
  for (int i = 0; i < nv; i++) d->qfrc_smooth[i] = d->qfrc_passive[i] - d->qfrc_bias[i] + d->qfrc_applied[i] + d->qfrc_actuator[i];
  /* xfrc_applied: Cartesian [force, torque] at body COM projected through the body Jacobian */
  for (int b = 1; b < m->nbody; b++) {
    const double* xf = d->xfrc_applied + 6*b;
    int nz = 0; for (int a = 0; a < 6; a++) if (xf[a] != 0) nz = 1;
    if (!nz) continue;
    for (int k = 0; k < nv; k++) { double jp[3], jr[3]; jac_col(m, d, b, d->xipos + 3*b, k, jp, jr); d->qfrc_smooth[k] += dot3(jp, xf) + dot3(jr, xf + 3); }
  }
  chol_solve(d->qacc_smooth, d->qL, d->qfrc_smooth, nv);
}

/* ------------------------------------------------------------------------- */
/* constraint solver: Newton on the primal          */
/* ------------------------------------------------------------------------- */
enum { ST_SATISFIED = 0, ST_QUADRATIC = 1, ST_CONE = 2, ST_LINEARNEG = 3, ST_LINEARPOS = 4 };
static double constraint_update(const Model* m, Data* d, const double* jar, int flg_state_only) {
  (void)m; (void)flg_state_only;
  double cost = 0;
  for (int i = 0; i < d->nefc; i++) {
    if (d->efc_type[i] == CT_ELLIPTIC) {
      /* map the residual to the regular dual cone: U = diag(mu, friction) * jar */
      Contact* c = d->contact + d->efc_id[i];
      int dim = c->dim; double mu = c->mu, U[6], T = 0;
      U[0] = jar[i]*mu;
      for (int j = 1; j < dim; j++) { U[j] = jar[i + j]*c->friction[j - 1]; T += U[j]*U[j]; }
      T = sqrt(T);
      double N = U[0];
      if (N >= mu*T || (T <= 0 && N >= 0)) {            /* top zone: inside the dual cone */
        for (int j = 0; j < dim; j++) { d->efc_state[i + j] = ST_SATISFIED; d->efc_force[i + j] = 0; }
      } else if (mu*N + T <= 0 || (T <= 0 && N < 0)) {  /* bottom zone: plain quadratic */
        for (int j = 0; j < dim; j++) {
          d->efc_state[i + j] = ST_QUADRATIC; d->efc_force[i + j] = -d->efc_D[i + j]*jar[i + j];
          cost += 0.5*d->efc_D[i + j]*jar[i + j]*jar[i + j];
        }
      } else {                                          /* middle zone: distance to the cone surface */
        double Dm = d->efc_D[i] / mjMAX(MINVAL, mu*mu*(1 + mu*mu)), NT = N - mu*T;
        cost += 0.5*Dm*NT*NT;
        d->efc_force[i] = -Dm*NT*mu;
        for (int j = 1; j < dim; j++) d->efc_force[i + j] = -d->efc_force[i]/T * U[j]*c->friction[j - 1];
        for (int j = 0; j < dim; j++) d->efc_state[i + j] = ST_CONE;
        /* Hessian of 0.5 Dm (N - mu T)^2 w.r.t. jar */
        double* H = c->H;
        H[0] = 1;
        for (int j = 1; j < dim; j++) H[j] = -mu*U[j]/T;
        double s3 = mu*N/(T*T*T), dg = mu*mu - mu*N/T;
        for (int k = 1; k < dim; k++) for (int j = k; j < dim; j++) H[k*dim + j] = s3*U[j]*U[k] + (j == k ? dg : 0);
        for (int k = 0; k < dim; k++) for (int j = k; j < dim; j++) {
          double sc = Dm * (j == 0 ? mu : c->friction[j - 1]) * (k == 0 ? mu : c->friction[k - 1]);
          H[k*dim + j] *= sc; H[j*dim + k] = H[k*dim + j];
        }
      }
      i += dim - 1;
      continue;
    }
    if (d->efc_type[i] == CT_EQUALITY) {   /* two-sided: always quadratic */
      d->efc_state[i] = ST_QUADRATIC; d->efc_force[i] = -d->efc_D[i]*jar[i]; cost += 0.5*d->efc_D[i]*jar[i]*jar[i];
      continue;
    }
    if (d->efc_type[i] == CT_FRICTION_DOF) {
      /* Huber cost: quadratic inside |jar| < R*floss, linear (force saturated at +-floss) outside */
      double f = m->dof_frictionloss[d->efc_id[i]], rf = d->efc_R[i]*f;
      if (jar[i] <= -rf) { d->efc_state[i] = ST_LINEARNEG; d->efc_force[i] = f; cost += f*(-0.5*rf - jar[i]); }
      else if (jar[i] >= rf) { d->efc_state[i] = ST_LINEARPOS; d->efc_force[i] = -f; cost += f*(-0.5*rf + jar[i]); }
      else { d->efc_state[i] = ST_QUADRATIC; d->efc_force[i] = -d->efc_D[i]*jar[i]; cost += 0.5*d->efc_D[i]*jar[i]*jar[i]; }
      continue;
    }
    if (jar[i] < 0) { d->efc_state[i] = ST_QUADRATIC; d->efc_force[i] = -d->efc_D[i]*jar[i]; cost += 0.5*d->efc_D[i]*jar[i]*jar[i]; }
    else { d->efc_state[i] = ST_SATISFIED; d->efc_force[i] = 0; }
  }
  return cost;
}
typedef struct { double alpha, cost, deriv[2]; } LSPoint;
typedef struct { double quadGauss[3]; int nefc; const double *jar, *jv, *quad; int evals; const Data* d; const Model* m; } LSCtx;
static void ls_eval(const LSCtx* c, LSPoint* p) {
  double a = p->alpha, qt[3] = {c->quadGauss[0], c->quadGauss[1], c->quadGauss[2]};
  double ccost = 0, cd0 = 0, cd1 = 0; /* non-quadratic part: elliptic contacts in the middle zone */
  for (int i = 0; i < c->nefc; i++) {
    if (c->d->efc_type[i] == CT_ELLIPTIC) {
      const Contact* con = c->d->contact + c->d->efc_id[i];
      int dim = con->dim; double mu = con->mu;
      double U0 = c->jar[i]*mu, V0 = c->jv[i]*mu, UU = 0, UV = 0, VV = 0;
      for (int j = 1; j < dim; j++) {
        double u = c->jar[i + j]*con->friction[j - 1], v = c->jv[i + j]*con->friction[j - 1];
        UU += u*u; UV += u*v; VV += v*v;
      }
      double N = U0 + a*V0, Tsqr = UU + a*(2*UV + a*VV);
      int bottom = 0;
      if (Tsqr <= 0) bottom = N < 0;
      else {
        double T = sqrt(Tsqr);
        if (N >= mu*T) {}
        else if (mu*N + T <= 0) bottom = 1;
        else {
          double Dm = c->d->efc_D[i] / mjMAX(MINVAL, mu*mu*(1 + mu*mu));
          double N1 = V0, T1 = (UV + a*VV)/T, T2 = VV/T - (UV + a*VV)*T1/(T*T);
          double NT = N - mu*T, NT1 = N1 - mu*T1;
          ccost += 0.5*Dm*NT*NT; cd0 += Dm*NT*NT1; cd1 += Dm*(NT1*NT1 - NT*mu*T2);
        }
      }
      if (bottom) for (int j = 0; j < dim; j++) { qt[0] += c->quad[3*(i + j)]; qt[1] += c->quad[3*(i + j) + 1]; qt[2] += c->quad[3*(i + j) + 2]; }
      i += dim - 1;
      continue;
    }
    if (c->d->efc_type[i] == CT_EQUALITY) { qt[0] += c->quad[3*i]; qt[1] += c->quad[3*i + 1]; qt[2] += c->quad[3*i + 2]; continue; }
    if (c->d->efc_type[i] == CT_FRICTION_DOF) {
      double f = c->m->dof_frictionloss[c->d->efc_id[i]], rf = c->d->efc_R[i]*f, x = c->jar[i] + a*c->jv[i];
      if (x <= -rf) { qt[0] += f*(-0.5*rf - c->jar[i]); qt[1] += -f*c->jv[i]; }
      else if (x >= rf) { qt[0] += f*(-0.5*rf + c->jar[i]); qt[1] += f*c->jv[i]; }
      else { qt[0] += c->quad[3*i]; qt[1] += c->quad[3*i + 1]; qt[2] += c->quad[3*i + 2]; }
      continue;
    }
    if (c->jar[i] + a*c->jv[i] < 0) { qt[0] += c->quad[3*i]; qt[1] += c->quad[3*i + 1]; qt[2] += c->quad[3*i + 2]; }
  }
  p->cost = a*a*qt[2] + a*qt[1] + qt[0] + ccost;
  p->deriv[0] = 2*a*qt[2] + qt[1] + cd0;
  p->deriv[1] = 2*qt[2] + cd1;
  if (p->deriv[1] <= 0) p->deriv[1] = MINVAL;
  ((LSCtx*)c)->evals++;
}
static int ls_update_bracket(const LSCtx* c, LSPoint* p, const LSPoint cand[3], LSPoint* pnext) {
  int flag = 0;
  for (int i = 0; i < 3; i++) {
    if (p->deriv[0] < 0 && cand[i].deriv[0] < 0 && p->deriv[0] < cand[i].deriv[0]) { *p = cand[i]; flag = 1; }
    else if (p->deriv[0] > 0 && cand[i].deriv[0] > 0 && p->deriv[0] > cand[i].deriv[0]) { *p = cand[i]; flag = 2; }
  }
  if (flag) { pnext->alpha = p->alpha - p->deriv[0]/p->deriv[1]; ls_eval(c, pnext); }
  return flag;
}
static double primal_search(const Model* m, Data* d, double gauss, double scale) {
  int nv = m->nv, nefc = d->nefc;
  double *search = d->w_search, *Mv = d->w_Mv, *jv = d->w_Jv, *jar = d->w_Jaref, *quad = d->w_quad;
  for (int i = 0; i < nv; i++) Mv[i] = dot_n(d->qM + (size_t)i*nv, search, nv);
  for (int i = 0; i < nefc; i++) jv[i] = dot_n(d->efc_J + (size_t)i*nv, search, nv);
  LSCtx c; c.d = d; c.m = m; c.nefc = nefc; c.jar = jar; c.jv = jv; c.quad = quad; c.evals = 0;
  c.quadGauss[0] = gauss;
  c.quadGauss[1] = dot_n(search, d->w_Ma, nv) - dot_n(d->qfrc_smooth, search, nv);
  c.quadGauss[2] = 0.5 * dot_n(search, Mv, nv);
  for (int i = 0; i < nefc; i++) {
    double dj0 = d->efc_D[i]*jar[i];
    quad[3*i] = 0.5*jar[i]*dj0; quad[3*i + 1] = jv[i]*dj0; quad[3*i + 2] = 0.5*d->efc_D[i]*jv[i]*jv[i];
  }
  double snorm = sqrt(dot_n(search, search, nv));
  if (snorm < MINVAL) return 0;
  double gtol = m->opt_tolerance * m->opt_ls_tolerance * snorm / scale;
  int lsmax = m->opt_ls_iterations;
  LSPoint p0, p1, p2, pmid, p1next, p2next;
  p0.alpha = 0; ls_eval(&c, &p0);
  p1.alpha = p0.alpha - p0.deriv[0]/p0.deriv[1]; ls_eval(&c, &p1);
  if (p0.cost < p1.cost) p1 = p0;
  if (fabs(p1.deriv[0]) < gtol) return p1.alpha;
  int dir = p1.deriv[0] < 0 ? 1 : -1, p2update = 0;
  p2 = p1;
  while (p1.deriv[0]*dir <= -gtol && c.evals < lsmax) {
    p2 = p1; p2update = 1;
    p1.alpha -= p1.deriv[0]/p1.deriv[1]; ls_eval(&c, &p1);
    if (fabs(p1.deriv[0]) < gtol) return p1.alpha;
  }
  if (c.evals >= lsmax) return p1.alpha;
  if (!p2update) return p1.alpha;
  p2next = p1;
  p1next.alpha = p1.alpha - p1.deriv[0]/p1.deriv[1]; ls_eval(&c, &p1next);
  while (c.evals < lsmax) {
    pmid.alpha = 0.5*(p1.alpha + p2.alpha); ls_eval(&c, &pmid);
    LSPoint cand[3] = {p1next, p2next, pmid};
    int best = -1;
    for (int i = 0; i < 3; i++) if (fabs(cand[i].deriv[0]) < gtol && (best == -1 || cand[i].cost < cand[best].cost)) best = i;
    if (best >= 0) return cand[best].alpha;
    int b1 = ls_update_bracket(&c, &p1, cand, &p1next);
    int b2 = ls_update_bracket(&c, &p2, cand, &p2next);
    if (!b1 && !b2) return pmid.cost < p0.cost ? pmid.alpha : 0;
  }
  if (p1.cost <= p2.cost && p1.cost < p0.cost) return p1.alpha;
  if (p2.cost <= p1.cost && p2.cost < p0.cost) return p2.alpha;
  return 0;
}
static void newton_gradient(const Model* m, Data* d) {
  int nv = m->nv, nefc = d->nefc;
  double *H = d->w_H, *L = d->w_H + (size_t)nv*nv;
  for (int i = 0; i < nv; i++) d->qfrc_constraint[i] = 0;
  for (int r = 0; r < nefc; r++) if (d->efc_force[r] != 0) for (int i = 0; i < nv; i++) d->qfrc_constraint[i] += d->efc_J[(size_t)r*nv + i]*d->efc_force[r];
  for (int i = 0; i < nv; i++) d->w_grad[i] = d->w_Ma[i] - d->qfrc_smooth[i] - d->qfrc_constraint[i];
  memcpy(H, d->qM, sizeof(double) * (size_t)nv * (size_t)nv);
  for (int r = 0; r < nefc; r++) if (d->efc_state[r] == ST_QUADRATIC) {
    const double* J = d->efc_J + (size_t)r*nv; double D = d->efc_D[r];
    for (int i = 0; i < nv; i++) { if (J[i] == 0) continue; double s = D*J[i]; for (int j = 0; j <= i; j++) H[i*nv + j] += s*J[j]; }
  }
  for (int r = 0; r < nefc; r++) if (d->efc_state[r] == ST_CONE) {
    /* H += J_c^T Hcone J_c for a contact in the middle zone */
    const Contact* c = d->contact + d->efc_id[r]; int dim = c->dim;
    for (int a = 0; a < dim; a++) for (int b = 0; b < dim; b++) {
      double h = c->H[a*dim + b]; if (h == 0) continue;
      const double *Ja = d->efc_J + (size_t)(r + a)*nv, *Jb = d->efc_J + (size_t)(r + b)*nv;
      for (int i = 0; i < nv; i++) { if (Ja[i] == 0) continue; double s = h*Ja[i]; for (int j = 0; j <= i; j++) H[i*nv + j] += s*Jb[j]; }
    }
    r += dim - 1;
  }
  for (int i = 0; i < nv; i++) for (int j = i + 1; j < nv; j++) H[i*nv + j] = H[j*nv + i];
  chol_factor(L, H, nv);
  chol_solve(d->w_Mgrad, L, d->w_grad, nv);
}
/* CG (mjSOL_CG, engine_solver.c mj_solPrimal with flg_Newton = 0): same cost, line search and stopping tests as
 * Newton; the gradient is preconditioned with M^-1 (mj_solveM on the factor mj_factorM left in qL) instead of H^-1. */
static void cg_gradient(const Model* m, Data* d) {
  int nv = m->nv, nefc = d->nefc;
  for (int i = 0; i < nv; i++) d->qfrc_constraint[i] = 0;
  for (int r = 0; r < nefc; r++) if (d->efc_force[r] != 0) for (int i = 0; i < nv; i++) d->qfrc_constraint[i] += d->efc_J[(size_t)r*nv + i]*d->efc_force[r];
  for (int i = 0; i < nv; i++) d->w_grad[i] = d->w_Ma[i] - d->qfrc_smooth[i] - d->qfrc_constraint[i];
  chol_solve(d->w_Mgrad, d->qL, d->w_grad, nv);
}
static double total_cost(const Model* m, Data* d, double constraint_cost, double* gauss_out) {
  int nv = m->nv; double g = 0;
  for (int i = 0; i < nv; i++) g += (d->w_Ma[i] - d->qfrc_smooth[i]) * (d->qacc[i] - d->qacc_smooth[i]);
  g *= 0.5; *gauss_out = g;
  return constraint_cost + g;
}

/* ---- noslip post-solver (mj_solNoSlip, engine_solver.c; restated from the documented algorithm) ----
 * Gauss-Seidel sweeps in force space over the friction dimensions only (dof friction loss, the edge
 * pairs of pyramidal contacts, the tangential rows of elliptic contacts) with the regulariser R removed,
 * so that the residual slip velocity goes to zero; normal forces and limit / equality forces stay.
 *   A = J M^-1 J^T (no R),  res = b + A f = J qacc - aref
 *   dof friction:  f -= res/A_ii, clamped to [-floss, floss]
 *   pyramid edge pair (f0, f1): f0 + f1 is kept; y = (f0 - f1)/2 minimises the 2x2 quadratic, |y| <= mid
 *   elliptic: min 1/2 x'A x + x'bc over the tangential forces s.t. sum (x_k/mu_k)^2 <= f_n^2 (QCQP,
 *   Newton on the multiplier), rescaled onto the cone when the constraint is active
 * a block update that increases the cost by more than 1e-10 is undone.  PARITY_ASSUMPTIONS.md row 30. */
static int qcqp(double* res, const double* Ain, const double* bin, const double* dd, double r, int n) {
  double A[25], b[5], Lc[25], v[5], pv[5], la = 0;
  for (int i = 0; i < n; i++) { b[i] = bin[i]*dd[i]; for (int j = 0; j < n; j++) A[i*n + j] = Ain[i*n + j]*dd[i]*dd[j]; }
  for (int iter = 0; iter < 20; iter++) {
    /* Cholesky of A + la I; not positive definite -> give up with zero forces */
    for (int i = 0; i < n; i++) for (int j = 0; j <= i; j++) {
      double t = A[i*n + j] + (i == j ? la : 0);
      for (int k = 0; k < j; k++) t -= Lc[i*n + k]*Lc[j*n + k];
      if (i == j) { if (t < 1e-10) { for (int k = 0; k < n; k++) res[k] = 0; return 0; } Lc[i*n + i] = sqrt(t); }
      else Lc[i*n + j] = t/Lc[j*n + j];
    }
    for (int i = 0; i < n; i++) { double t = -b[i]; for (int k = 0; k < i; k++) t -= Lc[i*n + k]*v[k]; v[i] = t/Lc[i*n + i]; }
    for (int i = n - 1; i >= 0; i--) { double t = v[i]; for (int k = i + 1; k < n; k++) t -= Lc[k*n + i]*v[k]; v[i] = t/Lc[i*n + i]; }
    double val = -r*r;
    for (int i = 0; i < n; i++) val += v[i]*v[i];
    if (val < 1e-10) break;
    /* deriv = -2 v' (A + la)^-1 v */
    for (int i = 0; i < n; i++) { double t = v[i]; for (int k = 0; k < i; k++) t -= Lc[i*n + k]*pv[k]; pv[i] = t/Lc[i*n + i]; }
    double deriv = 0;
    for (int i = 0; i < n; i++) deriv += pv[i]*pv[i];
    deriv *= -2;
    double delta = -val/deriv;
    if (delta < 1e-10) break;
    la += delta;
  }
  for (int i = 0; i < n; i++) res[i] = v[i]*dd[i];
  return la != 0;
}
/* cost change of a block update; an update that increases the cost is undone */
static double noslip_cost_change(const double* Ac, double* force, const double* oldforce, const double* res, int n) {
  double delta[6], change = 0;
  for (int i = 0; i < n; i++) delta[i] = force[i] - oldforce[i];
  for (int i = 0; i < n; i++) { double t = 0; for (int j = 0; j < n; j++) t += Ac[i*n + j]*delta[j]; change += 0.5*delta[i]*t + delta[i]*res[i]; }
  if (change > 1e-10) { for (int i = 0; i < n; i++) force[i] = oldforce[i]; change = 0; }
  return change;
}
static void noslip(const Model* m, Data* d) {
  const int nv = m->nv, nefc = d->nefc;
  int* rows = (int*)malloc(sizeof(int) * (size_t)(nefc + 1));
  int nf = 0;
  for (int i = 0; i < nefc; i++) {
    const int t = d->efc_type[i];
    if (t == CT_FRICTION_DOF || t == CT_PYRAMIDAL) rows[nf++] = i;
    else if (t == CT_ELLIPTIC && i != d->contact[d->efc_id[i]].efc_address) rows[nf++] = i;
  }
  if (!nf) { free(rows); return; }
  double* W = (double*)malloc(sizeof(double) * (size_t)nf * (size_t)nv);   /* M^-1 J_F^T */
  double* A = (double*)malloc(sizeof(double) * (size_t)nf * (size_t)nf);
  double* res = (double*)malloc(sizeof(double) * (size_t)nf);
  for (int a = 0; a < nf; a++) { chol_solve(W + (size_t)a*nv, d->qL, d->efc_J + (size_t)rows[a]*nv, nv); }
  /* symmetric by construction: entry (a, b), a >= b, is J_a . (M^-1 J_b^T), mirrored */
  for (int a = 0; a < nf; a++) for (int b = 0; b <= a; b++) A[a*nf + b] = A[b*nf + a] = dot_n(d->efc_J + (size_t)rows[a]*nv, W + (size_t)b*nv, nv);
  for (int a = 0; a < nf; a++) res[a] = dot_n(d->efc_J + (size_t)rows[a]*nv, d->qacc, nv) - d->efc_aref[rows[a]];
  double* force = d->efc_force;
  const double scale = 1 / (m->stat_meaninertia * mjMAX(1, nv));
  int iter = 0;
  while (iter < m->opt_noslip_iterations) {
    double improvement = 0;
    if (iter == 0) for (int i = 0; i < nefc; i++) improvement += 0.5*force[i]*force[i]*d->efc_R[i];
    for (int a = 0; a < nf; ) {
      const int i = rows[a], t = d->efc_type[i];
      int n;   /* block size */
      double Ac[25], old[5], bres[5], fnew[5];
      if (t == CT_FRICTION_DOF) n = 1;
      else if (t == CT_PYRAMIDAL) n = 2;
      else n = d->contact[d->efc_id[i]].dim - 1;
      for (int p = 0; p < n; p++) { old[p] = force[rows[a + p]]; bres[p] = res[a + p]; for (int q = 0; q < n; q++) Ac[p*n + q] = A[(a + p)*nf + a + q]; }
      if (t == CT_FRICTION_DOF) {
        const double fl = m->dof_frictionloss[d->efc_id[i]];
        fnew[0] = old[0] - bres[0]/Ac[0];
        if (fnew[0] < -fl) fnew[0] = -fl; else if (fnew[0] > fl) fnew[0] = fl;
      } else if (t == CT_PYRAMIDAL) {
        const double bc0 = bres[0] - Ac[0]*old[0] - Ac[1]*old[1], bc1 = bres[1] - Ac[2]*old[0] - Ac[3]*old[1];
        const double mid = 0.5*(old[0] + old[1]);
        const double K1 = Ac[0] + Ac[3] - Ac[1] - Ac[2], K0 = mid*(Ac[0] - Ac[3]) + bc0 - bc1;
        if (K1 < MINVAL) fnew[0] = fnew[1] = mid;
        else {
          const double y = -K0/K1;
          if (y < -mid) { fnew[0] = 0; fnew[1] = 2*mid; }
          else if (y > mid) { fnew[0] = 2*mid; fnew[1] = 0; }
          else { fnew[0] = mid + y; fnew[1] = mid - y; }
        }
      } else {
        const Contact* c = d->contact + d->efc_id[i];
        const double fn = force[c->efc_address];
        double bc[5];
        for (int p = 0; p < n; p++) { bc[p] = bres[p]; for (int q = 0; q < n; q++) bc[p] -= Ac[p*n + q]*old[q]; }
        if (fn < MINVAL) for (int p = 0; p < n; p++) fnew[p] = 0;
        else {
          const int active = qcqp(fnew, Ac, bc, c->friction, fn, n);
          if (active) {
            double ss = 0;
            for (int p = 0; p < n; p++) ss += (fnew[p]/c->friction[p])*(fnew[p]/c->friction[p]);
            ss = sqrt(fn*fn / mjMAX(MINVAL, ss));
            for (int p = 0; p < n; p++) fnew[p] *= ss;
          }
        }
      }
      improvement -= noslip_cost_change(Ac, fnew, old, bres, n);
      for (int p = 0; p < n; p++) {
        const double delta = fnew[p] - old[p];
        force[rows[a + p]] = fnew[p];
        if (delta != 0) for (int b = 0; b < nf; b++) res[b] += A[b*nf + a + p]*delta;
      }
      a += n;
    }
    improvement *= scale;
    iter++;
    if (improvement < m->opt_noslip_tolerance) break;
  }
  /* accelerations from the updated forces */
  for (int i = 0; i < nv; i++) d->qfrc_constraint[i] = 0;
  for (int r = 0; r < nefc; r++) if (force[r] != 0) for (int i = 0; i < nv; i++) d->qfrc_constraint[i] += d->efc_J[(size_t)r*nv + i]*force[r];
  double* tmp = (double*)malloc(sizeof(double) * (size_t)nv);
  for (int i = 0; i < nv; i++) tmp[i] = d->qfrc_smooth[i] + d->qfrc_constraint[i];
  chol_solve(d->qacc, d->qL, tmp, nv);
  free(tmp); free(res); free(A); free(W); free(rows);
}
/* ---- PGS (mj_solPGS; MuJoCo engine_solver.c, restated from the published algorithm; option solver="PGS",
 * dm_control/mjcf/schema.xml:69-72; the reference's own hot-path asset mujoco/testing/assets/humanoid.xml:9 uses it).
 * Dual problem: min over forces of 1/2 f'AR f + f'b, AR = J M^-1 J' + diag(R), b = J qacc_smooth - aref, with
 *   equality rows free, dof-friction rows |f| <= frictionloss, limit / frictionless / pyramid-edge rows f >= 0,
 *   elliptic contacts inside their friction cone.
 * Gauss-Seidel over the rows: a scalar row takes its unconstrained minimum and is clamped; an elliptic contact does
 * a normal (or ray) update followed by a QCQP over the friction dimensions with the normal fixed.  A block update that
 * increases the cost by more than 1e-10 is undone.  Warm start (warmstart() in engine_forward.c): forces of
 * qacc_warmstart through the primal map, kept only if their dual cost is below that of zero forces. */
static void pgs_solve(const Model* m, Data* d) {
  const int nv = m->nv, nefc = d->nefc;
  double* W = (double*)malloc(sizeof(double) * (size_t)nefc * (size_t)nv);   /* M^-1 J' */
  double* AR = (double*)malloc(sizeof(double) * (size_t)nefc * (size_t)nefc);
  double* b = (double*)malloc(sizeof(double) * (size_t)nefc);
  double* res = (double*)malloc(sizeof(double) * (size_t)nefc);
  double* force = d->efc_force;
  for (int a = 0; a < nefc; a++) chol_solve(W + (size_t)a*nv, d->qL, d->efc_J + (size_t)a*nv, nv);
  for (int a = 0; a < nefc; a++) for (int c = 0; c <= a; c++) AR[a*nefc + c] = AR[c*nefc + a] = dot_n(d->efc_J + (size_t)a*nv, W + (size_t)c*nv, nv);
  for (int a = 0; a < nefc; a++) AR[a*nefc + a] += d->efc_R[a];
  for (int a = 0; a < nefc; a++) b[a] = dot_n(d->efc_J + (size_t)a*nv, d->qacc_smooth, nv) - d->efc_aref[a];
  /* warm start */
  if (!(m->opt_disableflags & DMC_DSBL_WARMSTART)) {
    double* jar = d->w_Jaref;
    for (int i = 0; i < nefc; i++) jar[i] = dot_n(d->efc_J + (size_t)i*nv, d->qacc_warmstart, nv) - d->efc_aref[i];
    constraint_update(m, d, jar, 0);
    double cost = 0;
    for (int i = 0; i < nefc; i++) { cost += force[i]*b[i]; cost += 0.5*force[i]*dot_n(AR + (size_t)i*nefc, force, nefc); }
    if (cost > 0) for (int i = 0; i < nefc; i++) force[i] = 0;
  } else for (int i = 0; i < nefc; i++) force[i] = 0;
  const double scale = 1 / (m->stat_meaninertia * mjMAX(1, nv));
  int iter = 0;
  while (iter < m->opt_iterations) {
    double improvement = 0;
    for (int i = 0; i < nefc; ) {
      const int t = d->efc_type[i];
      const int dim = t == CT_ELLIPTIC ? d->contact[d->efc_id[i]].dim : 1;
      double Athis[36], old[6], rs[6];
      for (int p = 0; p < dim; p++) { rs[p] = b[i + p] + dot_n(AR + (size_t)(i + p)*nefc, force, nefc); old[p] = force[i + p]; }
      if (dim == 1) {
        Athis[0] = AR[i*nefc + i];
        force[i] -= rs[0]/Athis[0];
        if (t == CT_FRICTION_DOF) {
          const double fl = m->dof_frictionloss[d->efc_id[i]];
          if (force[i] < -fl) force[i] = -fl; else if (force[i] > fl) force[i] = fl;
        } else if (t != CT_EQUALITY) { if (force[i] < 0) force[i] = 0; }
      } else {
        const Contact* c = d->contact + d->efc_id[i];
        for (int p = 0; p < dim; p++) for (int q = 0; q < dim; q++) Athis[p*dim + q] = AR[(i + p)*nefc + i + q];
        if (force[i] < MINVAL) {                       /* normal force too small: normal update, friction cleared */
          force[i] -= rs[0]/Athis[0];
          if (force[i] < 0) force[i] = 0;
          for (int p = 1; p < dim; p++) force[i + p] = 0;
        } else {                                       /* ray update along the current force */
          double v[6], v1[6], denom = 0;
          for (int p = 0; p < dim; p++) v[p] = force[i + p];
          for (int p = 0; p < dim; p++) { v1[p] = dot_n(Athis + p*dim, v, dim); }
          denom = dot_n(v, v1, dim);
          if (denom >= MINVAL) {
            double x = -dot_n(v, rs, dim) / denom;
            if (force[i] + x*v[0] < 0) x = -force[i]/v[0];
            for (int p = 0; p < dim; p++) force[i + p] += x*v[p];
          }
        }
        /* friction update with the normal fixed */
        double Ac[25], bc[5], v[5];
        const int n = dim - 1;
        for (int p = 0; p < n; p++) {
          for (int q = 0; q < n; q++) Ac[p*n + q] = Athis[(p + 1)*dim + q + 1];
          bc[p] = rs[p + 1];
          for (int q = 0; q < n; q++) bc[p] -= Ac[p*n + q]*old[1 + q];
          bc[p] += Athis[(p + 1)*dim]*(force[i] - old[0]);
        }
        if (force[i] < MINVAL) for (int p = 0; p < n; p++) force[i + 1 + p] = 0;
        else {
          const int active = qcqp(v, Ac, bc, c->friction, force[i], n);
          if (active) {
            double ss = 0;
            for (int p = 0; p < n; p++) ss += v[p]*v[p]/(c->friction[p]*c->friction[p]);
            ss = sqrt(force[i]*force[i] / mjMAX(MINVAL, ss));
            for (int p = 0; p < n; p++) v[p] *= ss;
          }
          for (int p = 0; p < n; p++) force[i + 1 + p] = v[p];
        }
      }
      improvement -= noslip_cost_change(Athis, force + i, old, rs, dim);
      i += dim;
    }
    improvement *= scale;
    iter++;
    if (improvement < m->opt_tolerance) break;
  }
  d->solver_iter = iter;
  /* dualFinish: qfrc_constraint = J' f, qacc = qacc_smooth + M^-1 qfrc_constraint */
  for (int i = 0; i < nv; i++) d->qfrc_constraint[i] = 0;
  for (int r = 0; r < nefc; r++) if (force[r] != 0) for (int i = 0; i < nv; i++) d->qfrc_constraint[i] += d->efc_J[(size_t)r*nv + i]*force[r];
  chol_solve(d->qacc, d->qL, d->qfrc_constraint, nv);
  for (int i = 0; i < nv; i++) d->qacc[i] += d->qacc_smooth[i];
  free(res); free(b); free(AR); free(W);
}
static int solve_primal(const Model* m, Data* d, double scale);
/* mj_island: the kinematic trees (maximal sets of bodies joined by joints; the static bodies belong to none) are the
 * vertices, every constraint row is an edge set over the trees whose dofs it moves, the islands are the connected
 * components that own at least one row.  Trees without a constraint are in no island: their qacc is qacc_smooth.
 * The trees a row touches are read off its Jacobian (a row moves a tree iff it has a non-zero entry on one of its dofs);
 * the rows of one contact always travel together.  Islands are numbered by their lowest dof.  Fills d->dof_island and
 * d->efc_island, returns the island count. */
static int find_islands(const Model* m, Data* d) {
  const int nv = m->nv, nefc = d->nefc, nbody = m->nbody;
  int* root = (int*)malloc(sizeof(int) * (size_t)(nbody + nv + 2));      /* tree of a body = its ancestor below the world */
  int* parent = root + nbody + 1;                                        /* union-find over dofs */
  root[0] = 0;
  /* (a body welded to the world -- body_weldid 0 -- is static and belongs to no tree: what hangs off it by a joint starts a tree of its own) */
  for (int b = 1; b < nbody; b++) root[b] = m->body_weldid[m->body_parentid[b]] == 0 ? b : root[m->body_parentid[b]];
  for (int i = 0; i < nv; i++) parent[i] = i;
#define FIND(x, r) { r = (x); while (parent[r] != r) r = parent[r]; for (int _y = (x); parent[_y] != _y; ) { int _n = parent[_y]; parent[_y] = r; _y = _n; } }
#define UNITE(a, b) { int _ra, _rb; FIND(a, _ra); FIND(b, _rb); if (_ra != _rb) { if (_ra < _rb) parent[_rb] = _ra; else parent[_ra] = _rb; } }
  /* the dofs of one tree are one vertex */
  for (int i = 1; i < nv; i++) for (int j = i - 1; j >= 0; j--) if (root[m->dof_bodyid[i]] == root[m->dof_bodyid[j]]) { UNITE(i, j); break; }
  for (int r = 0; r < nefc; r++) {
    /* rows of one contact (frictionless 1, pyramidal 2 (dim - 1), elliptic dim) form one edge set */
    int r1 = r + 1;
    const int t = d->efc_type[r];
    if (t == CT_PYRAMIDAL || t == CT_ELLIPTIC) while (r1 < nefc && d->efc_type[r1] == t && d->efc_id[r1] == d->efc_id[r]) r1++;
    int first = -1;
    for (int q = r; q < r1; q++) for (int i = 0; i < nv; i++) if (d->efc_J[(size_t)q*nv + i] != 0) { if (first < 0) first = i; else UNITE(first, i); }
    for (int q = r; q < r1; q++) d->efc_island[q] = first;      /* a dof of the edge set for now (-1: a row that moves nothing) */
    r = r1 - 1;
  }
  /* number the components that own a row by their lowest dof */
  int ni = 0;
  for (int i = 0; i < nv; i++) d->dof_island[i] = -1;
  for (int i = 0; i < nv; i++) {
    int ri; FIND(i, ri);
    if (ri != i) { d->dof_island[i] = d->dof_island[ri]; continue; }      /* (the root of a component is its lowest dof) */
    int owns = 0;
    for (int r = 0; r < nefc && !owns; r++) if (d->efc_island[r] >= 0) { int rr; FIND(d->efc_island[r], rr); owns = rr == i; }
    if (owns) d->dof_island[i] = ni++;
  }
  for (int r = 0; r < nefc; r++) if (d->efc_island[r] >= 0) d->efc_island[r] = d->dof_island[d->efc_island[r]];
#undef FIND
#undef UNITE
  free(root);
  return ni;
}
/* The per-island solves of mj_fwdConstraint: island k's dofs and rows are gathered into a dense sub-problem (M and J
 * restricted to them: the cross terms are exact zeros), solved by the same mj_solPrimal from the same starting point, and
 * scattered back.  Dofs in no island take qacc_smooth. */
static void solve_islands(const Model* m, Data* d, int nisland) {
  const int nv = m->nv, nefc = d->nefc;
  if (!d->island_scratch) d->island_scratch = ora_data_create(m);
  Data* s = (Data*)d->island_scratch;
  int* dofs = (int*)malloc(sizeof(int) * (size_t)(nv + nefc + nv + 2));
  int *rows = dofs + nv + 1, *local = rows + nefc;
  double* floss = (double*)malloc(sizeof(double) * (size_t)(nv + 1));
  Contact* own_contacts = s->contact;
  s->contact = d->contact;      /* contact rows keep their contact ids */
  const double scale = 1 / (m->stat_meaninertia * mjMAX(1, nv));
  d->solver_iter = 0;
  for (int i = 0; i < nv; i++) if (d->dof_island[i] < 0) d->qacc[i] = d->qacc_smooth[i];
  /* (k == nisland: the rows that move no dof -- an all-zero Jacobian row, e.g. a contact between two static bodies --
   * are in no island; their force is a function of their own constant residual, evaluated as a system without dofs) */
  for (int k = 0; k <= nisland; k++) {
    int nd = 0, nr = 0;
    const int sel = k < nisland ? k : -1;
    for (int i = 0; i < nv; i++) { local[i] = -1; if (k < nisland && d->dof_island[i] == k) { local[i] = nd; dofs[nd++] = i; } }
    for (int r = 0; r < nefc; r++) if (d->efc_island[r] == sel) rows[nr++] = r;
    if (!nr) continue;
    Model sm = *m;
    sm.nv = nd;
    sm.dof_frictionloss = floss;
    for (int a = 0; a < nd; a++) {
      floss[a] = m->dof_frictionloss[dofs[a]];
      s->qacc[a] = d->qacc[dofs[a]]; s->qacc_smooth[a] = d->qacc_smooth[dofs[a]]; s->qfrc_smooth[a] = d->qfrc_smooth[dofs[a]];
      for (int b = 0; b < nd; b++) { s->qM[(size_t)a*nd + b] = d->qM[(size_t)dofs[a]*nv + dofs[b]]; s->qL[(size_t)a*nd + b] = 0; }
    }
    if (m->opt_solver == DMC_SOL_CG) chol_factor(s->qL, s->qM, nd);      /* the CG preconditioner: the island's block of M */
    for (int q = 0; q < nr; q++) {
      const int r = rows[q];
      for (int a = 0; a < nd; a++) s->efc_J[(size_t)q*nd + a] = d->efc_J[(size_t)r*nv + dofs[a]];
      s->efc_D[q] = d->efc_D[r]; s->efc_R[q] = d->efc_R[r]; s->efc_aref[q] = d->efc_aref[r];
      s->efc_type[q] = d->efc_type[r];
      s->efc_id[q] = d->efc_type[r] == CT_FRICTION_DOF ? local[d->efc_id[r]] : d->efc_id[r];
    }
    s->nefc = nr;
    const int it = solve_primal(&sm, s, scale);
    if (k < nisland && it > d->solver_iter) d->solver_iter = it;      /* the island that took longest */
    for (int a = 0; a < nd; a++) d->qacc[dofs[a]] = s->qacc[a];
    for (int q = 0; q < nr; q++) { d->efc_force[rows[q]] = s->efc_force[q]; d->efc_state[rows[q]] = s->efc_state[q]; }
  }
  s->contact = own_contacts;
  free(dofs); free(floss);
}
static void fwd_constraint(const Model* m, Data* d) {
  int nv = m->nv, nefc = d->nefc;
  d->solver_iter = 0;
  if (!nefc) {
    memcpy(d->qacc, d->qacc_smooth, sizeof(double) * (size_t)nv);
    memcpy(d->qacc_warmstart, d->qacc_smooth, sizeof(double) * (size_t)nv);
    memset(d->qfrc_constraint, 0, sizeof(double) * (size_t)nv);
    return;
  }
  if (m->opt_solver == DMC_SOL_PGS) {
    pgs_solve(m, d);
    memcpy(d->qacc_warmstart, d->qacc, sizeof(double) * (size_t)nv);
    if (m->opt_noslip_iterations > 0) noslip(m, d);
    return;
  }
  double *jar = d->w_Jaref, *Ma = d->w_Ma;
  /* warmstart: keep qacc_warmstart only if its cost beats qacc_smooth */
  if (!(m->opt_disableflags & DMC_DSBL_WARMSTART)) {
    memcpy(d->qacc, d->qacc_warmstart, sizeof(double) * (size_t)nv);
    for (int i = 0; i < nefc; i++) jar[i] = dot_n(d->efc_J + (size_t)i*nv, d->qacc, nv) - d->efc_aref[i];
    for (int i = 0; i < nv; i++) Ma[i] = dot_n(d->qM + (size_t)i*nv, d->qacc, nv);
    double gauss, cw = total_cost(m, d, constraint_update(m, d, jar, 1), &gauss);
    for (int i = 0; i < nefc; i++) jar[i] = dot_n(d->efc_J + (size_t)i*nv, d->qacc_smooth, nv) - d->efc_aref[i];
    double cs = constraint_update(m, d, jar, 1);
    if (cw > cs) memcpy(d->qacc, d->qacc_smooth, sizeof(double) * (size_t)nv);
  } else memcpy(d->qacc, d->qacc_smooth, sizeof(double) * (size_t)nv);
  /* constraint islands (mj_island; mjDSBL_ISLAND is a DISABLE flag, mjcf/schema.xml:102): the solver runs once per
   * island -- own line search, own iteration count, own stopping test -- exactly when mj_fwdConstraint does: flag on,
   * at least one island, no noslip pass, a primal solver (CG / Newton).  PARITY_ASSUMPTIONS rows 8, 37-39. */
  d->nisland = 0;
  if (!(m->opt_disableflags & DMC_DSBL_ISLAND) && m->opt_noslip_iterations == 0) {
    const int ni = find_islands(m, d);
    /* one island that holds every dof IS the joint problem: solved in place (same rows, same dofs, same arithmetic) */
    int whole = ni == 1;
    for (int i = 0; whole && i < nv; i++) if (d->dof_island[i] != 0) whole = 0;
    for (int r = 0; whole && r < nefc; r++) if (d->efc_island[r] != 0) whole = 0;
    d->nisland = ni;
    if (ni > 0 && !whole) { solve_islands(m, d, ni); goto solved; }
  }
  d->solver_iter = solve_primal(m, d, 1 / (m->stat_meaninertia * mjMAX(1, nv)));
solved:
  /* final forces at the solution */
  for (int i = 0; i < nv; i++) d->qfrc_constraint[i] = 0;
  for (int r = 0; r < nefc; r++) if (d->efc_force[r] != 0) for (int i = 0; i < nv; i++) d->qfrc_constraint[i] += d->efc_J[(size_t)r*nv + i]*d->efc_force[r];
  memcpy(d->qacc_warmstart, d->qacc, sizeof(double) * (size_t)nv);
  /* the warm start keeps the main solver's solution; noslip then edits qacc / efc_force */
  if (m->opt_noslip_iterations > 0) noslip(m, d);
}
/* mj_solPrimal (Newton / CG) from d->qacc on the system held by (m, d): m->nv dofs, d->nefc rows.  `scale` is the
 * tolerance scaling 1 / (meaninertia * max(1, nv)) of the WHOLE model, also when (m, d) is an island's sub-problem. */
static int solve_primal(const Model* m, Data* d, double scale) {
  int nv = m->nv, nefc = d->nefc;
  double *jar = d->w_Jaref, *Ma = d->w_Ma;
  for (int i = 0; i < nv; i++) Ma[i] = dot_n(d->qM + (size_t)i*nv, d->qacc, nv);
  for (int i = 0; i < nefc; i++) jar[i] = dot_n(d->efc_J + (size_t)i*nv, d->qacc, nv) - d->efc_aref[i];
  double gauss, cost = total_cost(m, d, constraint_update(m, d, jar, 0), &gauss);
  const int cg = m->opt_solver == DMC_SOL_CG;
  double *gradold = d->w_H, *Mgradold = d->w_H + nv;      /* CG only: w_H is free (no Hessian) */
  if (cg) cg_gradient(m, d); else newton_gradient(m, d);
  for (int i = 0; i < nv; i++) d->w_search[i] = -d->w_Mgrad[i];
  int iter = 0;
  while (iter < m->opt_iterations) {
    double alpha = primal_search(m, d, gauss, scale);
    if (alpha == 0) break;
    for (int i = 0; i < nv; i++) { d->qacc[i] += alpha*d->w_search[i]; Ma[i] += alpha*d->w_Mv[i]; }
    for (int i = 0; i < nefc; i++) jar[i] += alpha*d->w_Jv[i];
    double oldcost = cost;
    if (cg) { memcpy(gradold, d->w_grad, sizeof(double) * (size_t)nv); memcpy(Mgradold, d->w_Mgrad, sizeof(double) * (size_t)nv); }
    cost = total_cost(m, d, constraint_update(m, d, jar, 0), &gauss);
    if (cg) {
      cg_gradient(m, d);
      /* Polak-Ribiere, restarted when negative */
      double num = 0, den = 0;
      for (int i = 0; i < nv; i++) { num += d->w_grad[i]*(d->w_Mgrad[i] - Mgradold[i]); den += gradold[i]*Mgradold[i]; }
      double beta = num / mjMAX(MINVAL, den);
      if (beta < 0) beta = 0;
      for (int i = 0; i < nv; i++) d->w_search[i] = -d->w_Mgrad[i] + beta*d->w_search[i];
    } else {
      newton_gradient(m, d);
      for (int i = 0; i < nv; i++) d->w_search[i] = -d->w_Mgrad[i];
    }
    double improvement = scale*(oldcost - cost);
    double gradient = scale*sqrt(dot_n(d->w_grad, d->w_grad, nv));
    iter++;
    if (improvement < m->opt_tolerance || gradient < m->opt_tolerance) break;
  }
  return iter;
}
/* mj_contactForce in the contact frame: [normal, tangent1, tangent2, torsion, roll1, roll2] */
static void contact_force_local(const Model* m, const Data* d, int id, double* f6) {
  (void)m;
  const Contact* c = d->contact + id;
  memset(f6, 0, 6 * sizeof(double));
  if (c->efc_address < 0) return;
  const double* f = d->efc_force + c->efc_address;
  if (c->dim == 1) { f6[0] = f[0]; return; }
  if (d->efc_type[c->efc_address] == CT_ELLIPTIC) { for (int k = 0; k < c->dim; k++) f6[k] = f[k]; return; }
  for (int k = 0; k < 2*(c->dim - 1); k++) f6[0] += f[k];
  for (int k = 1; k < c->dim; k++) f6[k] = (f[2*(k - 1)] - f[2*(k - 1) + 1]) * c->friction[k - 1];
}
void ora_contact_force(const Model* m, const Data* d, int id, double* f6) { contact_force_local(m, d, id, f6); }

/* ------------------------------------------------------------------------- */
/* pipeline                                                                   */
/* ------------------------------------------------------------------------- */
static void fwd_position(const Model* m, Data* d) {
  kinematics(m, d); com_pos(m, d); tendon_kinematics(m, d); crb(m, d); collision(m, d); make_constraint(m, d); transmission(m, d);
}
static void fwd_velocity(const Model* m, Data* d) {
  for (int i = 0; i < m->nu; i++) {
    int j = m->actuator_trnid[2*i];
    d->actuator_velocity[i] = m->actuator_gear[6*i] * (m->actuator_trntype[i] == DMC_TRN_TENDON ? tendon_dot(m, j, d->qvel, 1) : d->qvel[m->jnt_dofadr[j]]);
  }
  com_vel(m, d); passive(m, d); rne(m, d);
}
static int bad_vec(const double* v, int n) { for (int i = 0; i < n; i++) if (isnan(v[i]) || v[i] > MAXVAL || v[i] < -MAXVAL) return 1; return 0; }
static void check_pos(const Model* m, Data* d) {
  if (bad_vec(d->qpos, m->nq)) { int w[DMC_NWARNING]; memcpy(w, d->warning, sizeof w); w[DMC_WARN_BADQPOS]++;
    if (!(m->opt_disableflags & DMC_DSBL_AUTORESET)) ora_reset(m, d, -1);
    memcpy(d->warning, w, sizeof w); }
}
static void check_vel(const Model* m, Data* d) {
  if (bad_vec(d->qvel, m->nv)) { int w[DMC_NWARNING]; memcpy(w, d->warning, sizeof w); w[DMC_WARN_BADQVEL]++;
    if (!(m->opt_disableflags & DMC_DSBL_AUTORESET)) ora_reset(m, d, -1);
    memcpy(d->warning, w, sizeof w); }
}
static void forward_skip(const Model* m, Data* d, int skipsensor) {
  fwd_position(m, d); if (!skipsensor) sensor_stage(m, d, DMC_STAGE_POS);
  fwd_velocity(m, d); if (!skipsensor) sensor_stage(m, d, DMC_STAGE_VEL);
  fwd_actuation(m, d); fwd_acceleration(m, d); fwd_constraint(m, d);
  if (!skipsensor) sensor_acc(m, d);
}
void ora_forward(const Model* m, Data* d) { forward_skip(m, d, 0); }
static void check_acc(const Model* m, Data* d) {
  if (bad_vec(d->qacc, m->nv)) { int w[DMC_NWARNING]; memcpy(w, d->warning, sizeof w); w[DMC_WARN_BADQACC]++;
    if (!(m->opt_disableflags & DMC_DSBL_AUTORESET)) { ora_reset(m, d, -1); memcpy(d->warning, w, sizeof w); ora_forward(m, d); }
    memcpy(d->warning, w, sizeof w); }
}
static void integrate_pos(const Model* m, double* qpos, const double* qvel, double dt) {
  for (int j = 0; j < m->njnt; j++) {
    int qa = m->jnt_qposadr[j], da = m->jnt_dofadr[j];
    switch (m->jnt_type[j]) {
      case DMC_JNT_FREE:
        for (int k = 0; k < 3; k++) qpos[qa + k] += dt*qvel[da + k];
        quat_integrate(qpos + qa + 3, qvel + da + 3, dt); break;
      case DMC_JNT_BALL: quat_integrate(qpos + qa, qvel + da, dt); break;
      default: qpos[qa] += dt*qvel[da];
    }
  }
}
static void advance(const Model* m, Data* d, const double* qacc, const double* qvel_for_pos) {
  double dt = m->opt_timestep;
  /* activations first (mj_advance): explicit Euler, or the exact exponential for filterexact */
  for (int i = 0, k = 0; i < m->nu; i++) if (m->actuator_dyntype[i] != DMC_DYN_NONE) {
    if (m->actuator_dyntype[i] == DMC_DYN_FILTEREXACT) {
      const double tau = mjMAX(MINVAL, m->actuator_dynprm[10*i]);
      d->act[k] += d->act_dot[k] * tau * (1 - exp(-dt/tau));
    } else d->act[k] += dt*d->act_dot[k];
    /* mj_nextActivation: the advanced activation is clamped to actrange */
    if (m->actuator_actlimited[i]) d->act[k] = mjMAX(m->actuator_actrange[2*i], mjMIN(m->actuator_actrange[2*i+1], d->act[k]));
    k++;
  }
  for (int i = 0; i < m->nv; i++) d->qvel[i] += dt*qacc[i];
  integrate_pos(m, d->qpos, qvel_for_pos ? qvel_for_pos : d->qvel, dt);
  d->time += dt;
}
static void euler(const Model* m, Data* d) {
  int nv = m->nv, damped = 0;
  if (!(m->opt_disableflags & (DMC_DSBL_EULERDAMP | DMC_DSBL_DAMPER))) for (int i = 0; i < nv; i++) if (m->dof_damping[i] > 0) damped = 1;
  if (damped) {
    double *H = d->w_H, *L = d->w_H + (size_t)nv*nv, *qfrc = d->w_grad, *qacc = d->w_Mgrad;
    memcpy(H, d->qM, sizeof(double) * (size_t)nv * (size_t)nv);
    for (int i = 0; i < nv; i++) { H[i*nv + i] += m->opt_timestep*m->dof_damping[i]; qfrc[i] = d->qfrc_smooth[i] + d->qfrc_constraint[i]; }
    chol_factor(L, H, nv);
    chol_solve(qacc, L, qfrc, nv);
    advance(m, d, qacc, NULL);
  } else advance(m, d, d->qacc, NULL);
}
/* mj_implicit for mjINT_IMPLICITFAST: (M - h dF/dv) qacc = qfrc_smooth + qfrc_constraint, where dF/dv keeps the
 * velocity derivatives of the passive forces (mjd_passive_vel: -dof_damping on the diagonal, -b J_t' J_t per damped
 * tendon) and of the actuator forces (mjd_actuator_vel: moment' (biasprm[2] + gainprm[2] * ctrl-or-act) moment for
 * affine bias / gain, skipped for an actuator whose force sits on its forcerange) and drops the Coriolis term (that is
 * what distinguishes it from mjINT_IMPLICIT), so the matrix stays symmetric positive definite and is Cholesky
 * factored.  Unlike mj_Euler the solve always happens (no "any damping" shortcut, mjDSBL_EULERDAMP does not apply). */
static void implicitfast(const Model* m, Data* d) {
  int nv = m->nv; double h = m->opt_timestep;
  double *H = d->w_H, *L = d->w_H + (size_t)nv*nv, *qfrc = d->w_grad, *qacc = d->w_Mgrad;
  double* mom = (double*)calloc((size_t)nv + 1, sizeof(double));
  memcpy(H, d->qM, sizeof(double) * (size_t)nv * (size_t)nv);
  for (int i = 0; i < nv; i++) qfrc[i] = d->qfrc_smooth[i] + d->qfrc_constraint[i];
  if (!(m->opt_disableflags & DMC_DSBL_DAMPER)) {
    for (int i = 0; i < nv; i++) H[i*nv + i] += h*m->dof_damping[i];
    for (int t = 0; t < m->ntendon; t++) if (m->tendon_damping[t] > 0) {
      memset(mom, 0, sizeof(double) * (size_t)nv);
      for (int w = m->tendon_adr[t]; w < m->tendon_adr[t] + m->tendon_num[t]; w++) mom[m->jnt_dofadr[m->wrap_objid[w]]] += m->wrap_prm[w];
      for (int i = 0; i < nv; i++) for (int j = 0; j < nv; j++) H[i*nv + j] += h*m->tendon_damping[t]*mom[i]*mom[j];
    }
  }
  if (!(m->opt_disableflags & DMC_DSBL_ACTUATION)) for (int i = 0, kact = 0; i < m->nu; i++) {
    const int stateful = m->actuator_dyntype[i] != DMC_DYN_NONE;
    const int k = kact; kact += stateful;
    if (m->actuator_forcelimited[i] && (d->actuator_force[i] <= m->actuator_forcerange[2*i] || d->actuator_force[i] >= m->actuator_forcerange[2*i + 1])) continue;
    double bias_vel = 0, gain_vel = 0;
    if (m->actuator_biastype[i] == DMC_BIAS_AFFINE) bias_vel = m->actuator_biasprm[10*i + 2];
    if (m->actuator_gaintype[i] == DMC_GAIN_AFFINE) gain_vel = m->actuator_gainprm[10*i + 2];
    if (gain_vel != 0) bias_vel += gain_vel * (stateful ? d->act[k] : d->ctrl[i]);
    if (bias_vel == 0) continue;
    memset(mom, 0, sizeof(double) * (size_t)nv);
    if (m->actuator_trntype[i] == DMC_TRN_TENDON) {
      int t = m->actuator_trnid[2*i];
      for (int w = m->tendon_adr[t]; w < m->tendon_adr[t] + m->tendon_num[t]; w++) mom[m->jnt_dofadr[m->wrap_objid[w]]] += m->actuator_gear[6*i] * m->wrap_prm[w];
    } else mom[m->jnt_dofadr[m->actuator_trnid[2*i]]] = m->actuator_gear[6*i];
    for (int a = 0; a < nv; a++) for (int b = 0; b < nv; b++) H[a*nv + b] -= h*bias_vel*mom[a]*mom[b];
  }
  free(mom);
  chol_factor(L, H, nv);
  chol_solve(qacc, L, qfrc, nv);
  advance(m, d, qacc, NULL);
}
static void integrate(const Model* m, Data* d) {      /* mj_step / mj_step2: Euler, or mj_implicit for implicitfast */
  if (m->opt_integrator == DMC_INT_IMPLICITFAST) implicitfast(m, d); else euler(m, d);
}
static void rk4(const Model* m, Data* d) {
  static const double A[9] = {0.5, 0, 0, 0, 0.5, 0, 0, 0, 1}, B[4] = {1.0/6, 1.0/3, 1.0/3, 1.0/6}, T[3] = {0.5, 0.5, 1};
  int nq = m->nq, nv = m->nv; double h = m->opt_timestep, time = d->time;
  double* buf = (double*)malloc(sizeof(double) * (size_t)(4*(nq + nv) + 4*2*nv + 2*nv));
  double *X[4], *F[4], *dX = buf + 4*(nq + nv) + 8*nv;
  for (int i = 0; i < 4; i++) { X[i] = buf + i*(nq + nv); F[i] = buf + 4*(nq + nv) + i*2*nv; }
  memcpy(X[0], d->qpos, sizeof(double)*(size_t)nq); memcpy(X[0] + nq, d->qvel, sizeof(double)*(size_t)nv);
  memcpy(F[0], d->qvel, sizeof(double)*(size_t)nv); memcpy(F[0] + nv, d->qacc, sizeof(double)*(size_t)nv);
  for (int i = 1; i < 4; i++) {
    for (int k = 0; k < 2*nv; k++) { dX[k] = 0; for (int j = 0; j < 3; j++) dX[k] += A[(i - 1)*3 + j] * (j < i ? F[j][k] : 0); }
    memcpy(X[i], X[0], sizeof(double)*(size_t)nq);
    integrate_pos(m, X[i], dX, h);
    for (int k = 0; k < nv; k++) X[i][nq + k] = X[0][nq + k] + h*dX[nv + k];
    memcpy(d->qpos, X[i], sizeof(double)*(size_t)nq); memcpy(d->qvel, X[i] + nq, sizeof(double)*(size_t)nv);
    d->time = time + h*T[i - 1];
    forward_skip(m, d, 1);
    memcpy(F[i], d->qvel, sizeof(double)*(size_t)nv); memcpy(F[i] + nv, d->qacc, sizeof(double)*(size_t)nv);
  }
  for (int k = 0; k < 2*nv; k++) { dX[k] = 0; for (int j = 0; j < 4; j++) dX[k] += B[j]*F[j][k]; }
  memcpy(d->qpos, X[0], sizeof(double)*(size_t)nq); memcpy(d->qvel, X[0] + nq, sizeof(double)*(size_t)nv);
  d->time = time;
  advance(m, d, dX + nv, dX);
  free(buf);
}
void ora_step1(const Model* m, Data* d) {
  check_pos(m, d); check_vel(m, d);
  fwd_position(m, d); sensor_stage(m, d, DMC_STAGE_POS);
  fwd_velocity(m, d); sensor_stage(m, d, DMC_STAGE_VEL);
}
void ora_step2(const Model* m, Data* d) {
  fwd_actuation(m, d); fwd_acceleration(m, d); fwd_constraint(m, d); sensor_acc(m, d);
  check_acc(m, d);
  integrate(m, d); /* mj_step2: mj_implicit for the implicit integrators, mj_Euler otherwise (RK4 never gets here: engine.py:149-154) */
}
void ora_step(const Model* m, Data* d, int nstep) {
  for (int s = 0; s < nstep; s++) {
    check_pos(m, d); check_vel(m, d);
    ora_forward(m, d);
    check_acc(m, d);
    if (m->opt_integrator == DMC_INT_RK4) rk4(m, d); else integrate(m, d);
  }
}
/* Physics.step(nstep) with legacy_step=True (engine.py:147-162) */
void ora_physics_step_legacy(const Model* m, Data* d, int nstep) {
  if (m->opt_integrator != DMC_INT_RK4) { ora_step2(m, d); if (nstep > 1) ora_step(m, d, nstep - 1); }
  else ora_step(m, d, nstep);
  ora_step1(m, d);
}
/* batch convenience for the CPU baseline: B independent datas, nstep legacy steps each */
void ora_physics_step_legacy_many(const Model* m, Data** ds, int B, int nstep) {
  for (int e = 0; e < B; e++) ora_physics_step_legacy(m, ds[e], nstep);
}

/* CPU-baseline rollout: T legacy Physics.step() calls for B independent datas,
 * actions laid out (T, B, nu).  Single thread; callers shard envs over threads. */
void ora_rollout_legacy(const Model* m, Data** ds, int B, int T, int nsub, const double* actions) {
  for (int t = 0; t < T; t++) for (int e = 0; e < B; e++) {
    memcpy(ds[e]->ctrl, actions + ((size_t)t*B + e)*m->nu, sizeof(double) * (size_t)m->nu);
    ora_physics_step_legacy(m, ds[e], nsub);
  }
}
