/*
 * vnl_oracle.c -- CPU ORACLE (test infrastructure, NOT a product path).
 *
 * A plain-C restatement of the computation behind the reference's
 * RodentTracking.reset / RodentTracking.step (reference envs/rodent.py:119-239):
 * five MJX physics substeps driven through Brax's PipelineEnv, then the
 * observation / reference-trajectory / reward / termination glue.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this file.  The product (vnl-brax-imitation_amd/csrc) never does.
 *
 * PARITY STATUS: "parity unpinned" for dynamics.  The physics lives in
 * third-party packages (mujoco-mjx / brax, unpinned in the reference's
 * requirements.txt:3-17) that are neither vendored under /root/reference nor
 * installed here, and the reference ships no tests.  This file restates the
 * published MJX algorithms (mjx/_src/{smooth,collision_primitive,constraint,
 * solver,passive,forward}.py, 3.1.x series) LITERALLY and DENSELY -- dense qM,
 * dense Cholesky, dense efc_J, exactly the `jacobian=dense` route the reference
 * selects at envs/rodent.py:63 -- so that the HIP product, which uses
 * tree-sparse / matrix-free algorithms, is checked against an independently
 * structured implementation.  What IS pinned by reference data: forward
 * kinematics, subtree COM and the egocentric transform against the shipped
 * clip (tests/golden/groom_clip.npz; see tests/test_oracle.py and tests/test_model_golden.py).
 *
 * Precision: `real` is double unless -DORC_F32 (then float, as JAX's default).
 *
 * Each function cites what it follows.  [UPSTREAM] = MJX/Brax/MuJoCo public
 * algorithm; file:line = reference repo.
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#ifdef ORC_F32
typedef float real;
#define RSQRT sqrtf
#define RSIN sinf
#define RCOS cosf
#define RACOS acosf
#define REXP expf
#define RFABS fabsf
#define RPOW powf
#else
typedef double real;
#define RSQRT sqrt
#define RSIN sin
#define RCOS cos
#define RACOS acos
#define REXP exp
#define RFABS fabs
#define RPOW pow
#endif

#define MJ_MINVAL ((real)1e-15)
#define MJ_MINIMP ((real)0.0001)
#define MJ_MAXIMP ((real)0.9999)

/*
 * Named switches for the items of the MJX restatement that are recalled rather than verified
 * ("least-certain items" 1-5 and 8).  Defaults = the reading the product
 * implements; tests/test_oracle_switches.py shows which outputs move, and by how much, under each
 * alternative, so that a later run against real MJX can settle them one at a time.
 */
typedef struct {
  int quat_writeback;       /* B.1 (item 1): kinematics writes the normalised free-joint quaternion back into qpos */
  int capsule_frame_axis;   /* item 2: plane-capsule tangent = capsule axis projected on the plane (else make_frame(n)) */
  int ls_mid_first;         /* item 3: line search applies the two midpoint replacement tests before the Newton ones */
  int ls_tie_lo;            /* item 3: lo.cost == hi.cost picks lo (default: lo.cost < hi.cost ? lo : hi, i.e. hi) */
  int inactive_pos_zero;    /* item 4: rows masked out by make_constraint report pos = 0 (else they keep their pos) */
  int contact_rows_by_type; /* item 5: contact rows grouped by collision function (sphere, capsule, ellipsoid) */
  int reset_warmstart_zero; /* item 8: qacc_warmstart after pipeline_init is zero (else the init solve's qacc) */
} orc_options;
static orc_options g_opt = {1, 1, 0, 0, 1, 0, 0};
int orc_set_option(const char *name, int value) {
#define OPT(f)                 \
  if (strcmp(name, #f) == 0) { \
    g_opt.f = value;           \
    return 0;                  \
  }
  OPT(quat_writeback) OPT(capsule_frame_axis) OPT(ls_mid_first) OPT(ls_tie_lo) OPT(inactive_pos_zero)
  OPT(contact_rows_by_type) OPT(reset_warmstart_zero)
#undef OPT
  return -1;
}
int orc_get_option(const char *name) {
#define OPT(f) \
  if (strcmp(name, #f) == 0) return g_opt.f;
  OPT(quat_writeback) OPT(capsule_frame_axis) OPT(ls_mid_first) OPT(ls_tie_lo) OPT(inactive_pos_zero)
  OPT(contact_rows_by_type) OPT(reset_warmstart_zero)
#undef OPT
  return -1;
}

/* Solver trace: the DISCRETE decisions of one solver.solve call, so that a test can tell "same decisions, same
 * numbers" from "a decision flipped" (the 6-iteration CG is not converged, so a flipped decision moves the result
 * by far more than rounding).  Same layout as the product's debug trace (csrc/vnl_body.h, VNL_TRACE_*):
 * [0] warm start used, [1] iterations, then per iteration 64 ints: [0] float32 bits of the accepted step length,
 * [1] line-search iterations, [2] the four replacement decisions of every line-search iteration (4 bits each),
 * [3] pick (0 none, 1 lo, 2 hi), [4..24) rows active at every trial step length (p0, first Newton point, then
 * lo_next / hi_next / mid per line-search iteration). */
#define ORC_TRACE_ITERS 8
#define ORC_TRACE_REC 64 /* ints per iteration record */
#define ORC_TRACE_ROWS (8 + ORC_TRACE_REC * ORC_TRACE_ITERS) /* then 16 ints: bit r set = constraint row r present (not masked out) */
#define ORC_TRACE_INTS (ORC_TRACE_ROWS + 16)

#define JNT_FREE 0
#define JNT_HINGE 3
#define GEOM_SPHERE 2
#define GEOM_CAPSULE 3
#define GEOM_ELLIPSOID 4

/* ------------------------------------------------------------------------- */
/* model blob                                                                */
/* ------------------------------------------------------------------------- */
typedef struct {
  char name[24];
  uint32_t dtype; /* 1 = f64, 2 = i32 */
  uint32_t count;
  uint64_t offset;
} blob_entry;

typedef struct orc_model {
  int nq, nv, nu, nbody, njnt, ncg, ncon, nlimit, nefc;
  int iterations, ls_iterations, eulerdamp;
  int solver_newton; /* opt.solver == NEWTON (reference configs/env_config.yaml:16-21, the ant): Mgrad = H^-1 grad, search = -Mgrad */
  real timestep, tolerance, ls_tolerance, impratio, meaninertia;
  real gravity[3];
  /* bodies */
  int *body_parentid, *body_rootid, *body_jntadr, *body_jntnum, *body_dofadr, *body_dofnum;
  real *body_pos, *body_quat, *body_ipos, *body_iquat, *body_inertia, *body_mass, *body_invweight0;
  /* joints */
  int *jnt_type, *jnt_bodyid, *jnt_qposadr, *jnt_dofadr, *jnt_limited;
  real *jnt_pos, *jnt_axis, *jnt_range, *jnt_stiffness, *jnt_margin, *jnt_solref, *jnt_solimp;
  real *qpos0, *qpos_spring;
  /* dofs */
  int *dof_bodyid, *dof_jntid, *dof_parentid;
  real *dof_armature, *dof_damping, *dof_invweight0;
  /* actuators */
  int *act_dof, *act_ctrllimited;
  real *act_gain, *act_gear, *act_tau, *act_ctrlrange;
  /* collidable geoms (all vs one plane) */
  int *cg_type, *cg_bodyid, *cg_ncon, *cg_conadr;
  real *cg_pos, *cg_quat, *cg_size, *cg_friction, *cg_solref, *cg_solimp, *cg_margin;
  real plane_pos[3], plane_normal[3];
  /* derived */
  int *limit_jnt; /* nlimit: joint id of each limit row */
} orc_model;

static const blob_entry *blob_find(const uint8_t *blob, size_t n, const char *name) {
  uint32_t ns;
  if (n < 16 || memcmp(blob, "VNLMDL01", 8) != 0) return NULL;
  memcpy(&ns, blob + 8, 4);
  const blob_entry *e = (const blob_entry *)(blob + 16);
  for (uint32_t i = 0; i < ns; i++)
    if (strncmp(e[i].name, name, 24) == 0) return &e[i];
  return NULL;
}

static real *blob_real(const uint8_t *blob, size_t n, const char *name, int expect) {
  const blob_entry *e = blob_find(blob, n, name);
  if (!e || e->dtype != 1 || (expect >= 0 && (int)e->count != expect)) {
    fprintf(stderr, "orc: blob section %s missing/mismatched (want %d)\n", name, expect);
    return NULL;
  }
  real *out = (real *)malloc(sizeof(real) * (e->count ? e->count : 1));
  const double *src = (const double *)(blob + e->offset);
  for (uint32_t i = 0; i < e->count; i++) out[i] = (real)src[i];
  return out;
}

static int *blob_int(const uint8_t *blob, size_t n, const char *name, int expect) {
  const blob_entry *e = blob_find(blob, n, name);
  if (!e || e->dtype != 2 || (expect >= 0 && (int)e->count != expect)) {
    fprintf(stderr, "orc: blob section %s missing/mismatched (want %d)\n", name, expect);
    return NULL;
  }
  int *out = (int *)malloc(sizeof(int) * (e->count ? e->count : 1));
  memcpy(out, blob + e->offset, sizeof(int) * e->count);
  return out;
}

static double blob_scalar(const uint8_t *blob, size_t n, const char *name) {
  const blob_entry *e = blob_find(blob, n, name);
  if (!e || e->dtype != 1 || e->count != 1) {
    fprintf(stderr, "orc: blob scalar %s missing\n", name);
    return NAN;
  }
  double v;
  memcpy(&v, blob + e->offset, 8);
  return v;
}

#define LOADR(field, cnt)                                  \
  if (!(m->field = blob_real(b, n, #field, (cnt)))) goto fail;
#define LOADI(field, cnt)                                 \
  if (!(m->field = blob_int(b, n, #field, (cnt)))) goto fail;

int orc_model_create(const void *blob, size_t n, orc_model **out) {
  const uint8_t *b = (const uint8_t *)blob;
  orc_model *m = (orc_model *)calloc(1, sizeof(orc_model));
  m->nq = (int)blob_scalar(b, n, "nq");
  m->nv = (int)blob_scalar(b, n, "nv");
  m->nu = (int)blob_scalar(b, n, "nu");
  m->nbody = (int)blob_scalar(b, n, "nbody");
  m->njnt = (int)blob_scalar(b, n, "njnt");
  m->ncg = (int)blob_scalar(b, n, "ncg");
  m->ncon = (int)blob_scalar(b, n, "ncon");
  m->nlimit = (int)blob_scalar(b, n, "nlimit");
  m->nefc = (int)blob_scalar(b, n, "nefc");
  m->iterations = (int)blob_scalar(b, n, "iterations");
  m->ls_iterations = (int)blob_scalar(b, n, "ls_iterations");
  m->eulerdamp = (int)blob_scalar(b, n, "eulerdamp");
  m->solver_newton = (int)blob_scalar(b, n, "solver_newton");
  m->timestep = (real)blob_scalar(b, n, "timestep");
  m->tolerance = (real)blob_scalar(b, n, "tolerance");
  m->ls_tolerance = (real)blob_scalar(b, n, "ls_tolerance");
  m->impratio = (real)blob_scalar(b, n, "impratio");
  m->meaninertia = (real)blob_scalar(b, n, "meaninertia");
  int nb = m->nbody, nj = m->njnt, nv = m->nv, nq = m->nq, nu = m->nu, ng = m->ncg;
  LOADI(body_parentid, nb) LOADI(body_rootid, nb) LOADI(body_jntadr, nb) LOADI(body_jntnum, nb)
  LOADI(body_dofadr, nb) LOADI(body_dofnum, nb)
  LOADR(body_pos, 3 * nb) LOADR(body_quat, 4 * nb) LOADR(body_ipos, 3 * nb) LOADR(body_iquat, 4 * nb)
  LOADR(body_inertia, 3 * nb) LOADR(body_mass, nb) LOADR(body_invweight0, 2 * nb)
  LOADI(jnt_type, nj) LOADI(jnt_bodyid, nj) LOADI(jnt_qposadr, nj) LOADI(jnt_dofadr, nj) LOADI(jnt_limited, nj)
  LOADR(jnt_pos, 3 * nj) LOADR(jnt_axis, 3 * nj) LOADR(jnt_range, 2 * nj) LOADR(jnt_stiffness, nj)
  LOADR(jnt_margin, nj) LOADR(jnt_solref, 2 * nj) LOADR(jnt_solimp, 5 * nj)
  LOADR(qpos0, nq) LOADR(qpos_spring, nq)
  LOADI(dof_bodyid, nv) LOADI(dof_jntid, nv) LOADI(dof_parentid, nv)
  LOADR(dof_armature, nv) LOADR(dof_damping, nv) LOADR(dof_invweight0, nv)
  LOADI(act_dof, nu) LOADI(act_ctrllimited, nu)
  LOADR(act_gain, nu) LOADR(act_gear, nu) LOADR(act_tau, nu) LOADR(act_ctrlrange, 2 * nu)
  LOADI(cg_type, ng) LOADI(cg_bodyid, ng) LOADI(cg_ncon, ng) LOADI(cg_conadr, ng)
  LOADR(cg_pos, 3 * ng) LOADR(cg_quat, 4 * ng) LOADR(cg_size, 3 * ng) LOADR(cg_friction, 3 * ng)
  LOADR(cg_solref, 2 * ng) LOADR(cg_solimp, 5 * ng) LOADR(cg_margin, ng)
  {
    real *g = blob_real(b, n, "gravity", 3), *pp = blob_real(b, n, "plane_pos", 3),
         *pn = blob_real(b, n, "plane_normal", 3);
    if (!g || !pp || !pn) goto fail;
    for (int i = 0; i < 3; i++) m->gravity[i] = g[i], m->plane_pos[i] = pp[i], m->plane_normal[i] = pn[i];
    free(g), free(pp), free(pn);
  }
  m->limit_jnt = (int *)malloc(sizeof(int) * (m->nlimit + 1));
  {
    int k = 0;
    for (int j = 0; j < nj; j++)
      if (m->jnt_limited[j] && m->jnt_type[j] == JNT_HINGE) m->limit_jnt[k++] = j;
    if (k != m->nlimit || m->nefc != m->nlimit + 4 * m->ncon) goto fail;
  }
  *out = m;
  return 0;
fail:
  free(m);
  return -1;
}

void orc_model_destroy(orc_model *m) { free(m); /* arrays leak by design: test-only, process-lifetime */ }

/* ------------------------------------------------------------------------- */
/* data                                                                      */
/* ------------------------------------------------------------------------- */
typedef struct orc_data {
  /* state */
  real *qpos, *qvel, *act, *ctrl, *qacc_warmstart;
  /* position-dependent */
  real *xpos, *xquat, *xmat, *xipos, *ximat, *xanchor, *xaxis;
  real *subtree_com, *cinert, *cdof, *crb, *qM, *qLD;
  real *con_dist, *con_pos, *con_frame;
  real *efc_J, *efc_pos, *efc_D, *efc_aref, *efc_force;
  /* velocity-dependent */
  real *cvel, *cdof_dot, *qfrc_passive, *qfrc_bias;
  /* actuation / acceleration */
  real *act_dot, *qfrc_actuator, *qfrc_smooth, *qacc_smooth, *qacc, *qfrc_constraint;
  int solver_niter;
  int trace[ORC_TRACE_INTS]; /* discrete decisions of the last solve (layout above) */
  int *row_live;             /* nefc: row not masked out by make_constraint */
  const int *follow;         /* NULL, or a trace whose decisions the next solve takes instead of its own */
  real follow_report[12];    /* legitimacy of the followed decisions (see slv section) */
  /* scratch */
  real *w0, *w1, *w2, *w3, *w4, *w5, *wefc0, *wefc1, *quad;
  real *arena;
  size_t arena_used, arena_cap;
} orc_data;

static real *ralloc(size_t n) { return (real *)calloc(n ? n : 1, sizeof(real)); }

/* bump allocation out of one block so that orc_data_destroy frees everything */
static real *dalloc(orc_data *d, size_t n) {
  if (n == 0) n = 1;
  if (d->arena_used + n > d->arena_cap) {
    fprintf(stderr, "orc: arena overflow\n");
    abort();
  }
  real *p = d->arena + d->arena_used;
  d->arena_used += n;
  return p;
}

orc_data *orc_data_create(const orc_model *m) {
  orc_data *d = (orc_data *)calloc(1, sizeof(orc_data));
  int nb = m->nbody, nv = m->nv, nj = m->njnt, ne = m->nefc, nc = m->ncon;
  d->arena_cap = (size_t)2 * nv * nv + (size_t)ne * nv + 64 * (size_t)nb + 32 * (size_t)nv + 16 * (size_t)nj +
                 16 * (size_t)nc + 12 * (size_t)ne + 8 * (size_t)m->nu + m->nq + 256;
  d->arena = (real *)calloc(d->arena_cap, sizeof(real));
#define ralloc(n) dalloc(d, (n))
  d->qpos = ralloc(m->nq), d->qvel = ralloc(nv), d->act = ralloc(m->nu), d->ctrl = ralloc(m->nu);
  d->qacc_warmstart = ralloc(nv);
  d->xpos = ralloc(3 * nb), d->xquat = ralloc(4 * nb), d->xmat = ralloc(9 * nb), d->xipos = ralloc(3 * nb);
  d->ximat = ralloc(9 * nb), d->xanchor = ralloc(3 * nj), d->xaxis = ralloc(3 * nj);
  d->subtree_com = ralloc(3 * nb), d->cinert = ralloc(10 * nb), d->cdof = ralloc(6 * nv), d->crb = ralloc(10 * nb);
  d->qM = ralloc((size_t)nv * nv), d->qLD = ralloc((size_t)nv * nv);
  d->con_dist = ralloc(nc), d->con_pos = ralloc(3 * nc), d->con_frame = ralloc(9 * nc);
  d->efc_J = ralloc((size_t)ne * nv), d->efc_pos = ralloc(ne), d->efc_D = ralloc(ne), d->efc_aref = ralloc(ne);
  d->efc_force = ralloc(ne);
  d->cvel = ralloc(6 * nb), d->cdof_dot = ralloc(6 * nv), d->qfrc_passive = ralloc(nv), d->qfrc_bias = ralloc(nv);
  d->act_dot = ralloc(m->nu), d->qfrc_actuator = ralloc(nv), d->qfrc_smooth = ralloc(nv);
  d->qacc_smooth = ralloc(nv), d->qacc = ralloc(nv), d->qfrc_constraint = ralloc(nv);
  d->w0 = ralloc(nv), d->w1 = ralloc(nv), d->w2 = ralloc(nv), d->w3 = ralloc(nv), d->w4 = ralloc(nv);
  d->w5 = ralloc(6 * nb + 6 * nv);
  d->wefc0 = ralloc(ne), d->wefc1 = ralloc(ne), d->quad = ralloc(3 * (size_t)ne);
#undef ralloc
  d->row_live = (int *)calloc(ne ? ne : 1, sizeof(int));
  return d;
}

void orc_data_destroy(orc_data *d) {
  free(d->row_live);
  free(d->arena);
  free(d);
}

/* named access for tests (bisection) */
int orc_data_field(const orc_model *m, orc_data *d, const char *name, real **ptr, int *count) {
  int nb = m->nbody, nv = m->nv, nj = m->njnt, ne = m->nefc, nc = m->ncon;
#define F(nm, cnt)                  \
  if (strcmp(name, #nm) == 0) {     \
    *ptr = d->nm, *count = (cnt);   \
    return 0;                       \
  }
  F(qpos, m->nq) F(qvel, nv) F(act, m->nu) F(ctrl, m->nu) F(qacc_warmstart, nv)
  F(xpos, 3 * nb) F(xquat, 4 * nb) F(xmat, 9 * nb) F(xipos, 3 * nb) F(ximat, 9 * nb)
  F(xanchor, 3 * nj) F(xaxis, 3 * nj) F(subtree_com, 3 * nb) F(cinert, 10 * nb) F(cdof, 6 * nv)
  F(crb, 10 * nb) F(qM, nv * nv) F(qLD, nv * nv) F(con_dist, nc) F(con_pos, 3 * nc) F(con_frame, 9 * nc)
  F(efc_J, ne * nv) F(efc_pos, ne) F(efc_D, ne) F(efc_aref, ne) F(efc_force, ne)
  F(cvel, 6 * nb) F(cdof_dot, 6 * nv) F(qfrc_passive, nv) F(qfrc_bias, nv)
  F(act_dot, m->nu) F(qfrc_actuator, nv) F(qfrc_smooth, nv) F(qacc_smooth, nv) F(qacc, nv)
  F(qfrc_constraint, nv)
#undef F
  return -1;
}
int orc_real_size(void) { return (int)sizeof(real); }
int orc_solver_niter(const orc_data *d) { return d->solver_niter; }
const int *orc_solver_trace(const orc_data *d) { return d->trace; }
int orc_trace_ints(void) { return ORC_TRACE_INTS; }

/* ------------------------------------------------------------------------- */
/* math [UPSTREAM mjx/_src/math.py]                                          */
/* ------------------------------------------------------------------------- */
static inline real dot3(const real *a, const real *b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
static inline void cross3(real *o, const real *a, const real *b) {
  real x = a[1] * b[2] - a[2] * b[1], y = a[2] * b[0] - a[0] * b[2], z = a[0] * b[1] - a[1] * b[0];
  o[0] = x, o[1] = y, o[2] = z;
}
static inline real norm3(const real *a) { return RSQRT(dot3(a, a)); }
static inline void quat_mul(real *o, const real *u, const real *v) {
  real w = u[0] * v[0] - u[1] * v[1] - u[2] * v[2] - u[3] * v[3];
  real x = u[0] * v[1] + u[1] * v[0] + u[2] * v[3] - u[3] * v[2];
  real y = u[0] * v[2] - u[1] * v[3] + u[2] * v[0] + u[3] * v[1];
  real z = u[0] * v[3] + u[1] * v[2] - u[2] * v[1] + u[3] * v[0];
  o[0] = w, o[1] = x, o[2] = y, o[3] = z;
}
/* math.rotate: r = 2(u.v)u + (s^2 - u.u)v + 2s(u x v) */
static inline void rotate(real *o, const real *vec, const real *q) {
  real s = q[0];
  const real *u = q + 1;
  real uv = dot3(u, vec), uu = dot3(u, u), c[3];
  cross3(c, u, vec);
  for (int i = 0; i < 3; i++) o[i] = 2 * (uv * u[i]) + (s * s - uu) * vec[i] + 2 * s * c[i];
}
static inline void quat_to_mat(real *m, const real *q) {
  real q00 = q[0] * q[0], q01 = q[0] * q[1], q02 = q[0] * q[2], q03 = q[0] * q[3];
  real q11 = q[1] * q[1], q12 = q[1] * q[2], q13 = q[1] * q[3];
  real q22 = q[2] * q[2], q23 = q[2] * q[3], q33 = q[3] * q[3];
  m[0] = q00 + q11 - q22 - q33, m[1] = 2 * (q12 - q03), m[2] = 2 * (q13 + q02);
  m[3] = 2 * (q12 + q03), m[4] = q00 - q11 + q22 - q33, m[5] = 2 * (q23 - q01);
  m[6] = 2 * (q13 - q02), m[7] = 2 * (q23 + q01), m[8] = q00 - q11 - q22 + q33;
}
static inline void axis_angle_to_quat(real *q, const real *axis, real angle) {
  real s = RSIN(angle * (real)0.5), c = RCOS(angle * (real)0.5);
  q[0] = c, q[1] = axis[0] * s, q[2] = axis[1] * s, q[3] = axis[2] * s;
}
static inline void normalize4(real *q) {
  real n = RSQRT(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  if (n > 0) q[0] /= n, q[1] /= n, q[2] /= n, q[3] /= n;
}
/* inert_mul: cinert (10) x motion (6) -> force (6) */
static inline void inert_mul(real *o, const real *i, const real *v) {
  /* inr tri order [0]=xx [1]=yy [2]=zz [3]=xy [4]=xz [5]=yz ; pos = i[6:9] (mass*offset), mass=i[9] */
  real ang[3], c[3];
  ang[0] = i[0] * v[0] + i[3] * v[1] + i[4] * v[2];
  ang[1] = i[3] * v[0] + i[1] * v[1] + i[5] * v[2];
  ang[2] = i[4] * v[0] + i[5] * v[1] + i[2] * v[2];
  cross3(c, i + 6, v + 3);
  o[0] = ang[0] + c[0], o[1] = ang[1] + c[1], o[2] = ang[2] + c[2];
  cross3(c, i + 6, v);
  o[3] = i[9] * v[3] - c[0], o[4] = i[9] * v[4] - c[1], o[5] = i[9] * v[5] - c[2];
}
static inline void motion_cross(real *o, const real *u, const real *v) {
  real a[3], b[3], c[3];
  cross3(a, u, v);
  cross3(b, u + 3, v);
  cross3(c, u, v + 3);
  o[0] = a[0], o[1] = a[1], o[2] = a[2];
  o[3] = b[0] + c[0], o[4] = b[1] + c[1], o[5] = b[2] + c[2];
}
static inline void motion_cross_force(real *o, const real *v, const real *f) {
  real a[3], b[3], c[3];
  cross3(a, v, f);
  cross3(b, v + 3, f + 3);
  cross3(c, v, f + 3);
  o[0] = a[0] + b[0], o[1] = a[1] + b[1], o[2] = a[2] + b[2];
  o[3] = c[0], o[4] = c[1], o[5] = c[2];
}
/* math.make_frame */
static void make_frame(real *frame, const real *a_in) {
  real a[3] = {a_in[0], a_in[1], a_in[2]}, n = norm3(a);
  if (n > 0) a[0] /= n, a[1] /= n, a[2] /= n;
  real b[3] = {0, 0, 0};
  if (-0.5 < a[1] && a[1] < 0.5) b[1] = 1; else b[2] = 1;
  real ab = dot3(a, b);
  for (int i = 0; i < 3; i++) b[i] -= a[i] * ab;
  n = norm3(b);
  for (int i = 0; i < 3; i++) b[i] /= n;
  memcpy(frame, a, sizeof(a)), memcpy(frame + 3, b, sizeof(b));
  cross3(frame + 6, a, b);
}

/* ------------------------------------------------------------------------- */
/* smooth.kinematics [UPSTREAM mjx/_src/smooth.py kinematics]                */
/* ------------------------------------------------------------------------- */
void orc_kinematics(const orc_model *m, orc_data *d) {
  d->xquat[0] = 1;
  quat_to_mat(d->xmat, d->xquat);
  for (int b = 1; b < m->nbody; b++) {
    int p = m->body_parentid[b];
    real pos[3], quat[4], tmp[3];
    rotate(tmp, m->body_pos + 3 * b, d->xquat + 4 * p);
    for (int i = 0; i < 3; i++) pos[i] = d->xpos[3 * p + i] + tmp[i];
    quat_mul(quat, d->xquat + 4 * p, m->body_quat + 4 * b);
    for (int k = 0; k < m->body_jntnum[b]; k++) {
      int j = m->body_jntadr[b] + k, qa = m->jnt_qposadr[j];
      real *anchor = d->xanchor + 3 * j, *axis = d->xaxis + 3 * j;
      if (m->jnt_type[j] == JNT_FREE) {
        for (int i = 0; i < 3; i++) anchor[i] = d->qpos[qa + i], pos[i] = d->qpos[qa + i];
        axis[0] = 0, axis[1] = 0, axis[2] = 1;
        for (int i = 0; i < 4; i++) quat[i] = d->qpos[qa + 3 + i];
        normalize4(quat);
        if (g_opt.quat_writeback)
          for (int i = 0; i < 4; i++) d->qpos[qa + 3 + i] = quat[i]; /* normalised quat written back */
      } else {
        real qloc[4], q2[4];
        rotate(tmp, m->jnt_pos + 3 * j, quat);
        for (int i = 0; i < 3; i++) anchor[i] = tmp[i] + pos[i];
        rotate(axis, m->jnt_axis + 3 * j, quat);
        axis_angle_to_quat(qloc, m->jnt_axis + 3 * j, d->qpos[qa] - m->qpos0[qa]);
        quat_mul(q2, quat, qloc);
        memcpy(quat, q2, sizeof(q2));
        rotate(tmp, m->jnt_pos + 3 * j, quat);
        for (int i = 0; i < 3; i++) pos[i] = anchor[i] - tmp[i];
      }
    }
    memcpy(d->xpos + 3 * b, pos, sizeof(pos));
    memcpy(d->xquat + 4 * b, quat, sizeof(quat));
    quat_to_mat(d->xmat + 9 * b, quat);
  }
  /* xipos / ximat = local_to_global(xpos, xquat, body_ipos, body_iquat) */
  for (int b = 0; b < m->nbody; b++) {
    real tmp[3], q[4];
    rotate(tmp, m->body_ipos + 3 * b, d->xquat + 4 * b);
    for (int i = 0; i < 3; i++) d->xipos[3 * b + i] = d->xpos[3 * b + i] + tmp[i];
    quat_mul(q, d->xquat + 4 * b, m->body_iquat + 4 * b);
    quat_to_mat(d->ximat + 9 * b, q);
  }
}

/* smooth.com_pos [UPSTREAM] */
void orc_com_pos(const orc_model *m, orc_data *d) {
  int nb = m->nbody;
  real *mp = d->w5; /* 3*nb */
  real *ms = d->w5 + 3 * nb;
  for (int b = 0; b < nb; b++) {
    for (int i = 0; i < 3; i++) mp[3 * b + i] = d->xipos[3 * b + i] * m->body_mass[b];
    ms[b] = m->body_mass[b];
  }
  for (int b = nb - 1; b > 0; b--) {
    int p = m->body_parentid[b];
    for (int i = 0; i < 3; i++) mp[3 * p + i] += mp[3 * b + i];
    ms[p] += ms[b];
  }
  for (int b = 0; b < nb; b++)
    for (int i = 0; i < 3; i++)
      d->subtree_com[3 * b + i] = ms[b] < MJ_MINVAL ? d->xipos[3 * b + i] : mp[3 * b + i] / ms[b];
  /* cinert: inertia about subtree_com[root] in world axes */
  for (int b = 0; b < nb; b++) {
    const real *R = d->ximat + 9 * b, *I = m->body_inertia + 3 * b;
    const real *rc = d->subtree_com + 3 * m->body_rootid[b];
    real off[3], mass = m->body_mass[b], A[9];
    for (int i = 0; i < 3; i++) off[i] = d->xipos[3 * b + i] - rc[i];
    /* (ximat * inert) @ ximat.T  + h h^T mass, h = cross(off, -eye) */
    for (int r = 0; r < 3; r++)
      for (int c = 0; c < 3; c++) {
        real s = 0;
        for (int k = 0; k < 3; k++) s += R[3 * r + k] * I[k] * R[3 * c + k];
        A[3 * r + c] = s;
      }
    real oo = dot3(off, off);
    for (int r = 0; r < 3; r++)
      for (int c = 0; c < 3; c++) A[3 * r + c] += mass * ((r == c ? oo : 0) - off[r] * off[c]);
    real *ci = d->cinert + 10 * b;
    ci[0] = A[0], ci[1] = A[4], ci[2] = A[8], ci[3] = A[1], ci[4] = A[2], ci[5] = A[5];
    ci[6] = off[0] * mass, ci[7] = off[1] * mass, ci[8] = off[2] * mass, ci[9] = mass;
  }
  /* cdof */
  for (int j = 0; j < m->njnt; j++) {
    int b = m->jnt_bodyid[j], da = m->jnt_dofadr[j];
    const real *rc = d->subtree_com + 3 * m->body_rootid[b];
    real off[3];
    for (int i = 0; i < 3; i++) off[i] = rc[i] - d->xanchor[3 * j + i];
    if (m->jnt_type[j] == JNT_FREE) {
      for (int k = 0; k < 3; k++) {
        real *c = d->cdof + 6 * (da + k);
        memset(c, 0, 6 * sizeof(real));
        c[3 + k] = 1;
      }
      for (int k = 0; k < 3; k++) { /* rows of xmat.T = columns of xmat */
        real *c = d->cdof + 6 * (da + 3 + k), a[3];
        for (int i = 0; i < 3; i++) a[i] = d->xmat[9 * b + 3 * i + k];
        c[0] = a[0], c[1] = a[1], c[2] = a[2];
        cross3(c + 3, a, off);
      }
    } else {
      real *c = d->cdof + 6 * da;
      const real *a = d->xaxis + 3 * j;
      c[0] = a[0], c[1] = a[1], c[2] = a[2];
      cross3(c + 3, a, off);
    }
  }
}

/* smooth.crb + support.make_m (dense) [UPSTREAM] */
void orc_crb(const orc_model *m, orc_data *d) {
  int nb = m->nbody, nv = m->nv;
  memcpy(d->crb, d->cinert, sizeof(real) * 10 * nb);
  for (int b = nb - 1; b > 0; b--) {
    int p = m->body_parentid[b];
    for (int i = 0; i < 10; i++) d->crb[10 * p + i] += d->crb[10 * b + i];
  }
  memset(d->crb, 0, sizeof(real) * 10);
  memset(d->qM, 0, sizeof(real) * nv * nv);
  for (int i = 0; i < nv; i++) {
    real f[6];
    inert_mul(f, d->crb + 10 * m->dof_bodyid[i], d->cdof + 6 * i);
    for (int j = i; j >= 0; j = m->dof_parentid[j]) {
      real s = 0;
      for (int k = 0; k < 6; k++) s += f[k] * d->cdof[6 * j + k];
      if (i == j) s += m->dof_armature[i];
      d->qM[i * nv + j] = s;
      d->qM[j * nv + i] = s;
    }
  }
}

/* dense Cholesky (lower), as jax.scipy.linalg.cho_factor in smooth.factor_m */
static int chol_factor(real *L, const real *A, int n) {
  memcpy(L, A, sizeof(real) * n * n);
  for (int j = 0; j < n; j++) {
    real s = L[j * n + j];
    for (int k = 0; k < j; k++) s -= L[j * n + k] * L[j * n + k];
    if (!(s > 0)) s = MJ_MINVAL;
    real ljj = RSQRT(s);
    L[j * n + j] = ljj;
    for (int i = j + 1; i < n; i++) {
      real t = L[i * n + j];
      for (int k = 0; k < j; k++) t -= L[i * n + k] * L[j * n + k];
      L[i * n + j] = t / ljj;
    }
  }
  return 0;
}
static void chol_solve(const real *L, real *x, int n) {
  for (int i = 0; i < n; i++) {
    real s = x[i];
    for (int k = 0; k < i; k++) s -= L[i * n + k] * x[k];
    x[i] = s / L[i * n + i];
  }
  for (int i = n - 1; i >= 0; i--) {
    real s = x[i];
    for (int k = i + 1; k < n; k++) s -= L[k * n + i] * x[k];
    x[i] = s / L[i * n + i];
  }
}
void orc_factor_m(const orc_model *m, orc_data *d) { chol_factor(d->qLD, d->qM, m->nv); }
static void solve_m(const orc_model *m, const orc_data *d, real *out, const real *in) {
  if (out != in) memcpy(out, in, sizeof(real) * m->nv);
  chol_solve(d->qLD, out, m->nv);
}
static void mul_m(const orc_model *m, const orc_data *d, real *out, const real *v) {
  int nv = m->nv;
  for (int i = 0; i < nv; i++) {
    real s = 0;
    for (int j = 0; j < nv; j++) s += d->qM[i * nv + j] * v[j];
    out[i] = s;
  }
}

/* collision_primitive.plane_sphere / plane_capsule / plane_ellipsoid [UPSTREAM] */
void orc_collision(const orc_model *m, orc_data *d) {
  const real *n = m->plane_normal, *pp = m->plane_pos;
  for (int g = 0; g < m->ncg; g++) {
    int b = m->cg_bodyid[g], c0 = m->cg_conadr[g];
    real gpos[3], gq[4], gmat[9], tmp[3];
    rotate(tmp, m->cg_pos + 3 * g, d->xquat + 4 * b);
    for (int i = 0; i < 3; i++) gpos[i] = d->xpos[3 * b + i] + tmp[i];
    quat_mul(gq, d->xquat + 4 * b, m->cg_quat + 4 * g);
    quat_to_mat(gmat, gq);
    const real *size = m->cg_size + 3 * g;
    if (m->cg_type[g] == GEOM_SPHERE) {
      real rel[3];
      for (int i = 0; i < 3; i++) rel[i] = gpos[i] - pp[i];
      real dist = dot3(rel, n) - size[0];
      d->con_dist[c0] = dist;
      for (int i = 0; i < 3; i++) d->con_pos[3 * c0 + i] = gpos[i] - n[i] * (size[0] + (real)0.5 * dist);
      make_frame(d->con_frame + 9 * c0, n);
    } else if (m->cg_type[g] == GEOM_CAPSULE) {
      real axis[3] = {gmat[2], gmat[5], gmat[8]};
      real na = dot3(n, axis), bvec[3];
      for (int i = 0; i < 3; i++) bvec[i] = axis[i] - n[i] * na;
      real bn = norm3(bvec);
      if (bn > 0)
        for (int i = 0; i < 3; i++) bvec[i] /= bn;
      if (bn < 0.5) {
        bvec[0] = 0, bvec[1] = 0, bvec[2] = 0;
        if (-0.5 < n[1] && n[1] < 0.5) bvec[1] = 1; else bvec[2] = 1;
      }
      real frame[9];
      memcpy(frame, n, 3 * sizeof(real)), memcpy(frame + 3, bvec, 3 * sizeof(real));
      cross3(frame + 6, n, bvec);
      if (!g_opt.capsule_frame_axis) make_frame(frame, n);
      for (int s = 0; s < 2; s++) {
        real sgn = s == 0 ? 1 : -1, c[3], rel[3];
        for (int i = 0; i < 3; i++) c[i] = gpos[i] + sgn * axis[i] * size[1], rel[i] = c[i] - pp[i];
        real dist = dot3(rel, n) - size[0];
        d->con_dist[c0 + s] = dist;
        for (int i = 0; i < 3; i++) d->con_pos[3 * (c0 + s) + i] = c[i] - n[i] * (size[0] + (real)0.5 * dist);
        memcpy(d->con_frame + 9 * (c0 + s), frame, sizeof(frame));
      }
    } else { /* ellipsoid */
      real ln[3], sup[3], wp[3], pos[3], rel[3];
      for (int i = 0; i < 3; i++) /* mat.T @ n */
        ln[i] = (gmat[0 + i] * n[0] + gmat[3 + i] * n[1] + gmat[6 + i] * n[2]) * size[i];
      real nn = norm3(ln);
      for (int i = 0; i < 3; i++) sup[i] = -(nn > 0 ? ln[i] / nn : ln[i]) * size[i];
      for (int i = 0; i < 3; i++) wp[i] = gmat[3 * i] * sup[0] + gmat[3 * i + 1] * sup[1] + gmat[3 * i + 2] * sup[2];
      for (int i = 0; i < 3; i++) pos[i] = gpos[i] + wp[i], rel[i] = pos[i] - pp[i];
      real dist = dot3(n, rel);
      d->con_dist[c0] = dist;
      for (int i = 0; i < 3; i++) d->con_pos[3 * c0 + i] = pos[i] - n[i] * dist * (real)0.5;
      make_frame(d->con_frame + 9 * c0, n);
    }
  }
}

/* support.jac: translational Jacobian of a world point attached to `body` */
static void jacp_point(const orc_model *m, const orc_data *d, real *jacp /* nv x 3 */, const real *point, int body) {
  int nv = m->nv;
  memset(jacp, 0, sizeof(real) * 3 * nv);
  if (body == 0) return;
  const real *rc = d->subtree_com + 3 * m->body_rootid[body];
  real off[3] = {point[0] - rc[0], point[1] - rc[1], point[2] - rc[2]};
  int bb = body;
  while (bb > 0 && m->body_dofnum[bb] == 0) bb = m->body_parentid[bb];
  if (bb == 0) return;
  for (int dd = m->body_dofadr[bb] + m->body_dofnum[bb] - 1; dd >= 0; dd = m->dof_parentid[dd]) {
    const real *c = d->cdof + 6 * dd;
    real x[3];
    cross3(x, c, off);
    for (int i = 0; i < 3; i++) jacp[3 * dd + i] = c[3 + i] + x[i];
  }
}

/* constraint.make_constraint [UPSTREAM mjx/_src/constraint.py]: limit rows then pyramidal contact rows */
void orc_make_constraint(const orc_model *m, orc_data *d) {
  int nv = m->nv, ne = m->nefc;
  memset(d->efc_J, 0, sizeof(real) * ne * nv);
  real *invweight = d->wefc0;
  real *solref = (real *)malloc(sizeof(real) * 2 * ne), *solimp = (real *)malloc(sizeof(real) * 5 * ne);
  int r = 0;
  for (int k = 0; k < m->nlimit; k++, r++) {
    int j = m->limit_jnt[k], qa = m->jnt_qposadr[j], da = m->jnt_dofadr[j];
    real q = d->qpos[qa], dmin = q - m->jnt_range[2 * j], dmax = m->jnt_range[2 * j + 1] - q;
    real pos = (dmin < dmax ? dmin : dmax) - m->jnt_margin[j];
    int active = pos < 0;
    if (d->follow && r < 512) {
      int f = (d->follow[ORC_TRACE_ROWS + (r >> 5)] >> (r & 31)) & 1;
      if (f != active) {
        d->follow_report[6] += 1;
        if (RFABS(pos) > d->follow_report[7]) d->follow_report[7] = RFABS(pos);
        active = f;
      }
    }
    d->efc_J[r * nv + da] = active ? (dmin < dmax ? (real)1 : (real)-1) : 0;
    d->row_live[r] = active;
    d->efc_pos[r] = (active || !g_opt.inactive_pos_zero) ? pos : 0;
    invweight[r] = m->dof_invweight0[da];
    memcpy(solref + 2 * r, m->jnt_solref + 2 * j, 2 * sizeof(real));
    memcpy(solimp + 5 * r, m->jnt_solimp + 5 * j, 5 * sizeof(real));
  }
  real *jac = (real *)malloc(sizeof(real) * 3 * nv);
  int *gorder = (int *)malloc(sizeof(int) * (m->ncg + 1));
  {
    int k = 0;
    if (g_opt.contact_rows_by_type) {
      for (int ty = 0; ty < 8; ty++)
        for (int g = 0; g < m->ncg; g++)
          if (m->cg_type[g] == ty) gorder[k++] = g;
    } else {
      for (int g = 0; g < m->ncg; g++) gorder[k++] = g;
    }
  }
  for (int gi = 0; gi < m->ncg; gi++) {
    int g = gorder[gi];
    int b = m->cg_bodyid[g];
    real mu[2] = {m->cg_friction[3 * g], m->cg_friction[3 * g]}; /* contact friction[:2] = (slide, slide) */
    real t = m->body_invweight0[0] + m->body_invweight0[2 * b];
    for (int s = 0; s < m->cg_ncon[g]; s++) {
      int c = m->cg_conadr[g] + s;
      real dist = d->con_dist[c] - m->cg_margin[g];
      int active = dist < 0;
      if (d->follow && r < 512) {
        int f = (d->follow[ORC_TRACE_ROWS + (r >> 5)] >> (r & 31)) & 1;
        if (f != active) {
          d->follow_report[6] += 1;
          if (RFABS(dist) > d->follow_report[7]) d->follow_report[7] = RFABS(dist);
          active = f;
        }
      }
      jacp_point(m, d, jac, d->con_pos + 3 * c, b); /* body1 = world -> zero */
      const real *fr = d->con_frame + 9 * c;
      for (int t2 = 0; t2 < 2; t2++)
        for (int sg = 0; sg < 2; sg++, r++) {
          real f = sg == 0 ? mu[t2] : -mu[t2];
          if (active)
            for (int dd = 0; dd < nv; dd++) {
              real jn = dot3(fr, jac + 3 * dd), jt = dot3(fr + 3 * (1 + t2), jac + 3 * dd);
              d->efc_J[r * nv + dd] = jn + jt * f;
            }
          d->efc_pos[r] = (active || !g_opt.inactive_pos_zero) ? dist : 0;
          d->row_live[r] = active;
          invweight[r] = (t + f * f * t) * 2 * f * f / m->impratio;
          memcpy(solref + 2 * r, m->cg_solref + 2 * g, 2 * sizeof(real));
          memcpy(solimp + 5 * r, m->cg_solimp + 5 * g, 5 * sizeof(real));
        }
    }
  }
  free(jac), free(gorder);
  for (r = 0; r < ne; r++) {
    real timeconst = solref[2 * r], dampratio = solref[2 * r + 1];
    real dmin = solimp[5 * r], dmax = solimp[5 * r + 1], width = solimp[5 * r + 2], mid = solimp[5 * r + 3],
         power = solimp[5 * r + 4];
    if (timeconst < 2 * m->timestep) timeconst = 2 * m->timestep; /* refsafe */
    dmin = dmin < MJ_MINIMP ? MJ_MINIMP : (dmin > MJ_MAXIMP ? MJ_MAXIMP : dmin);
    dmax = dmax < MJ_MINIMP ? MJ_MINIMP : (dmax > MJ_MAXIMP ? MJ_MAXIMP : dmax);
    if (width < MJ_MINVAL) width = MJ_MINVAL;
    mid = mid < MJ_MINIMP ? MJ_MINIMP : (mid > MJ_MAXIMP ? MJ_MAXIMP : mid);
    if (power < 1) power = 1;
    real k = 1 / (dmax * dmax * timeconst * timeconst * dampratio * dampratio);
    real bb = 2 / (dmax * timeconst);
    if (solref[2 * r] <= 0) k = -solref[2 * r] / (dmax * dmax);
    if (solref[2 * r + 1] <= 0) bb = -solref[2 * r + 1] / dmax;
    real imp_x = RFABS(d->efc_pos[r]) / width;
    real imp_a = ((real)1 / RPOW(mid, power - 1)) * RPOW(imp_x, power);
    real imp_b = 1 - ((real)1 / RPOW(1 - mid, power - 1)) * RPOW(1 - imp_x, power);
    real imp_y = imp_x < mid ? imp_a : imp_b;
    real imp = dmin + imp_y * (dmax - dmin);
    imp = imp < dmin ? dmin : (imp > dmax ? dmax : imp);
    if (imp_x > 1) imp = dmax;
    real R = invweight[r] * (1 - imp) / imp;
    if (R < MJ_MINVAL) R = MJ_MINVAL;
    real jv = 0;
    for (int dd = 0; dd < nv; dd++) jv += d->efc_J[r * nv + dd] * d->qvel[dd];
    d->efc_aref[r] = -bb * jv - k * imp * d->efc_pos[r];
    d->efc_D[r] = 1 / R;
  }
  free(solref), free(solimp);
}

/* smooth.com_vel [UPSTREAM] */
void orc_com_vel(const orc_model *m, orc_data *d) {
  memset(d->cvel, 0, sizeof(real) * 6);
  for (int b = 1; b < m->nbody; b++) {
    real cvel[6];
    memcpy(cvel, d->cvel + 6 * m->body_parentid[b], sizeof(cvel));
    for (int k = 0; k < m->body_jntnum[b]; k++) {
      int j = m->body_jntadr[b] + k, da = m->jnt_dofadr[j];
      if (m->jnt_type[j] == JNT_FREE) {
        for (int t = 0; t < 3; t++)
          for (int i = 0; i < 6; i++) cvel[i] += d->cdof[6 * (da + t) + i] * d->qvel[da + t];
        for (int t = 0; t < 3; t++) memset(d->cdof_dot + 6 * (da + t), 0, 6 * sizeof(real));
        for (int t = 3; t < 6; t++) motion_cross(d->cdof_dot + 6 * (da + t), cvel, d->cdof + 6 * (da + t));
        for (int t = 3; t < 6; t++)
          for (int i = 0; i < 6; i++) cvel[i] += d->cdof[6 * (da + t) + i] * d->qvel[da + t];
      } else {
        motion_cross(d->cdof_dot + 6 * da, cvel, d->cdof + 6 * da);
        for (int i = 0; i < 6; i++) cvel[i] += d->cdof[6 * da + i] * d->qvel[da];
      }
    }
    memcpy(d->cvel + 6 * b, cvel, sizeof(cvel));
  }
}

/* passive._spring_damper [UPSTREAM] */
void orc_passive(const orc_model *m, orc_data *d) {
  memset(d->qfrc_passive, 0, sizeof(real) * m->nv);
  for (int j = 0; j < m->njnt; j++) {
    if (m->jnt_type[j] != JNT_HINGE) continue;
    int qa = m->jnt_qposadr[j], da = m->jnt_dofadr[j];
    d->qfrc_passive[da] = -m->jnt_stiffness[j] * (d->qpos[qa] - m->qpos_spring[qa]);
  }
  for (int i = 0; i < m->nv; i++) d->qfrc_passive[i] -= m->dof_damping[i] * d->qvel[i];
}

/* smooth.rne [UPSTREAM] */
void orc_rne(const orc_model *m, orc_data *d) {
  int nb = m->nbody, nv = m->nv;
  real *cacc = d->w5, *cfrc = (real *)malloc(sizeof(real) * 6 * nb);
  for (int i = 0; i < 3; i++) cacc[i] = 0, cacc[3 + i] = -m->gravity[i];
  for (int b = 1; b < nb; b++) {
    real *a = cacc + 6 * b;
    memcpy(a, cacc + 6 * m->body_parentid[b], 6 * sizeof(real));
    for (int k = 0; k < m->body_dofnum[b]; k++) {
      int dd = m->body_dofadr[b] + k;
      for (int i = 0; i < 6; i++) a[i] += d->cdof_dot[6 * dd + i] * d->qvel[dd];
    }
  }
  for (int b = 0; b < nb; b++) {
    real f1[6], iv[6], f2[6];
    inert_mul(f1, d->cinert + 10 * b, cacc + 6 * b);
    inert_mul(iv, d->cinert + 10 * b, d->cvel + 6 * b);
    motion_cross_force(f2, d->cvel + 6 * b, iv);
    for (int i = 0; i < 6; i++) cfrc[6 * b + i] = f1[i] + f2[i];
  }
  for (int b = nb - 1; b > 0; b--) {
    int p = m->body_parentid[b];
    for (int i = 0; i < 6; i++) cfrc[6 * p + i] += cfrc[6 * b + i];
  }
  for (int dd = 0; dd < nv; dd++) {
    real s = 0;
    for (int i = 0; i < 6; i++) s += d->cdof[6 * dd + i] * cfrc[6 * m->dof_bodyid[dd] + i];
    d->qfrc_bias[dd] = s;
  }
  free(cfrc);
}

/* forward.fwd_actuation [UPSTREAM]: filter dynamics, fixed gain, no bias, joint transmission */
void orc_actuation(const orc_model *m, orc_data *d) {
  memset(d->qfrc_actuator, 0, sizeof(real) * m->nv);
  for (int i = 0; i < m->nu; i++) {
    real ctrl = d->ctrl[i];
    if (m->act_ctrllimited[i]) {
      real lo = m->act_ctrlrange[2 * i], hi = m->act_ctrlrange[2 * i + 1];
      ctrl = ctrl < lo ? lo : (ctrl > hi ? hi : ctrl);
    }
    real ctrl_act = ctrl;
    if (m->act_tau[i] >= 0) {
      real tau = m->act_tau[i] < MJ_MINVAL ? MJ_MINVAL : m->act_tau[i];
      d->act_dot[i] = (ctrl - d->act[i]) / tau;
      ctrl_act = d->act[i];
    } else {
      d->act_dot[i] = 0;
    }
    real force = m->act_gain[i] * ctrl_act;
    d->qfrc_actuator[m->act_dof[i]] += m->act_gear[i] * force;
  }
}

/* forward.fwd_acceleration [UPSTREAM] */
void orc_acceleration(const orc_model *m, orc_data *d) {
  for (int i = 0; i < m->nv; i++) d->qfrc_smooth[i] = d->qfrc_passive[i] - d->qfrc_bias[i] + d->qfrc_actuator[i];
  solve_m(m, d, d->qacc_smooth, d->qfrc_smooth);
}

/* ------------------------------------------------------------------------- */
/* solver.solve: primal CG with exact line search [UPSTREAM mjx/_src/solver.py] */
/* ------------------------------------------------------------------------- */
typedef struct {
  real *qacc, *Ma, *Jaref, *grad, *Mgrad, *search, *qfrc_constraint, *efc_force;
  real gauss, cost, prev_cost;
  real cost_scale, prev_cost_scale; /* sum of the magnitudes of the terms `cost` is made of: its rounding error is eps x this */
} slv_ctx;

static void slv_update_constraint(const orc_model *m, const orc_data *d, slv_ctx *c) {
  int nv = m->nv, ne = m->nefc;
  real cost = 0;
  for (int r = 0; r < ne; r++) {
    int active = c->Jaref[r] < 0;
    c->efc_force[r] = active ? d->efc_D[r] * -c->Jaref[r] : 0;
    if (active) cost += d->efc_D[r] * c->Jaref[r] * c->Jaref[r];
  }
  for (int i = 0; i < nv; i++) {
    real s = 0;
    for (int r = 0; r < ne; r++) s += d->efc_J[r * nv + i] * c->efc_force[r];
    c->qfrc_constraint[i] = s;
  }
  real g = 0, gs = 0;
  for (int i = 0; i < nv; i++) {
    g += (c->Ma[i] - d->qfrc_smooth[i]) * (c->qacc[i] - d->qacc_smooth[i]);
    gs += (RFABS(c->Ma[i]) + RFABS(d->qfrc_smooth[i])) * (RFABS(c->qacc[i]) + RFABS(d->qacc_smooth[i]));
  }
  c->gauss = (real)0.5 * g;
  c->prev_cost = c->cost, c->prev_cost_scale = c->cost_scale;
  c->cost = (real)0.5 * cost + c->gauss;
  c->cost_scale = (real)0.5 * cost + (real)0.5 * gs;
}

/* solver._update_gradient [UPSTREAM mjx/_src/solver.py]: CG preconditions with M^-1 (the factor of qM); NEWTON with the
 * Hessian of the cost at the current active set, H = qM + J' diag(efc_D * active) J, dense Cholesky (cho_factor /
 * cho_solve), as MJX does on its dense route. */
static void slv_update_gradient(const orc_model *m, const orc_data *d, slv_ctx *c) {
  int nv = m->nv, ne = m->nefc;
  for (int i = 0; i < nv; i++) c->grad[i] = c->Ma[i] - d->qfrc_smooth[i] - c->qfrc_constraint[i];
  if (!m->solver_newton) {
    solve_m(m, d, c->Mgrad, c->grad);
    return;
  }
  real *H = (real *)malloc(sizeof(real) * 2 * nv * nv), *Lh = H + nv * nv;
  memcpy(H, d->qM, sizeof(real) * nv * nv);
  for (int r = 0; r < ne; r++) {
    if (!(c->Jaref[r] < 0) || d->efc_D[r] == 0) continue;
    const real *J = d->efc_J + (size_t)r * nv;
    for (int i = 0; i < nv; i++) {
      if (J[i] == 0) continue;
      real w = d->efc_D[r] * J[i];
      for (int j = 0; j < nv; j++) H[i * nv + j] += w * J[j];
    }
  }
  chol_factor(Lh, H, nv);
  memcpy(c->Mgrad, c->grad, sizeof(real) * nv);
  chol_solve(Lh, c->Mgrad, nv);
  free(H);
}

static void slv_init(const orc_model *m, const orc_data *d, slv_ctx *c, const real *qacc) {
  int nv = m->nv, ne = m->nefc;
  memcpy(c->qacc, qacc, sizeof(real) * nv);
  for (int r = 0; r < ne; r++) {
    real s = 0;
    for (int i = 0; i < nv; i++) s += d->efc_J[r * nv + i] * qacc[i];
    c->Jaref[r] = s - d->efc_aref[r];
  }
  mul_m(m, d, c->Ma, qacc);
  c->cost = INFINITY, c->prev_cost = 0, c->cost_scale = 0, c->prev_cost_scale = 0;
  slv_update_constraint(m, d, c);
}

typedef struct {
  real alpha, cost, deriv_0, deriv_1;
} ls_point;

/* Decision FOLLOWING (test instrument).  When d->follow points at a trace recorded by another implementation of
 * this solver (the product's debug trace, same layout as d->trace), every discrete decision -- warm start, number of
 * CG iterations, initial bracket order, number of line-search iterations, the four bracket replacements of each,
 * the final pick -- is taken from that trace instead of from the oracle's own comparisons, so that the oracle
 * evaluates the SAME branch of the algorithm and its float64 numbers can be compared with the other side's float32
 * numbers at rounding level.  Whether the followed decisions were legitimate is judged separately and reported in
 * d->follow_report:
 * All three tie measures are in units of 1e-6 x the sum of the magnitudes of the terms the compared quantity is a sum
 * of (~16 float32 roundings of it: the cost has a Gauss term 0.5 (Ma - f).(a - a_smooth) that cancels heavily), so <= 1
 * means "a float32 evaluation cannot tell the two sides apart":
 *   [0] worst line search: cost(alpha followed) - cost(alpha natural), where "natural" is this oracle's own float64
 *       line search from the same state;
 *   [1] worst CG-exit disagreement: distance of the natural exit test (improvement or gradient) from its threshold,
 *       0 if the exit iteration agreed;
 *   [2] warm-start disagreement: |cost_warm - cost_smooth|, 0 if the choice agreed;
 *   [3] number of trial step lengths at which the set of active rows had a different size;
 *   [4] the largest, over those, of min_r |Jaref_r + alpha jv_r| / (|(J qacc)_r| + |aref_r| + |alpha jv_r|): how far
 *       from its switching point the nearest row was, relative to the terms it is made of (rounding-sized: a row
 *       sat on its kink);
 *   [5] number of followed decisions that differ from the natural ones;
 *   [6] constraint rows whose presence was followed against this oracle's own test (limit violated / geom in contact);
 *   [7] the largest |violation depth| among those (rounding-sized: the joint / geom sat on its threshold);
 *   [8] the largest relative distance between a trial step length of this oracle and the followed side's. */
#define ORC_FOLLOW_REPORT 12

static ls_point ls_eval(const orc_model *m, const orc_data *d, const slv_ctx *c, const real *jv, const real *quad,
                        const real *quad_gauss, real alpha, int *nactive, real *kink) {
  real q0 = quad_gauss[0], q1 = quad_gauss[1], q2 = quad_gauss[2];
  int na = 0;
  real near = 1;
  for (int r = 0; r < m->nefc; r++) {
    real x = c->Jaref[r] + alpha * jv[r];
    if (x < 0) q0 += quad[3 * r], q1 += quad[3 * r + 1], q2 += quad[3 * r + 2];
    if ((nactive || kink) && d->row_live[r]) { /* rows make_constraint did not mask out */
      if (x < 0) na++;
      /* x = (J qacc)_r - aref_r + alpha jv_r: distance from the switching point relative to the terms it is made of */
      real den = RFABS(c->Jaref[r] + d->efc_aref[r]) + RFABS(d->efc_aref[r]) + RFABS(alpha * jv[r]);
      if (den > 0 && RFABS(x) / den < near) near = RFABS(x) / den;
    }
  }
  if (nactive) *nactive = na;
  if (kink) *kink = near;
  ls_point p;
  p.alpha = alpha;
  p.cost = alpha * alpha * q2 + alpha * q1 + q0;
  p.deriv_0 = 2 * alpha * q2 + q1;
  p.deriv_1 = 2 * q2 + (q2 == 0 ? MJ_MINVAL : 0);
  return p;
}

/* one exact line search from the current solver state; tr: trace record of this iteration to fill (or NULL);
 * fol: record to follow (or NULL).  Returns the accepted step length (0: no improvement). */
static real ls_run(const orc_model *m, orc_data *d, const slv_ctx *c, const real *jv, const real *quad, const real *qg,
                   real gtol, int *tr, const int *fol, real *report) {
  int nev = 0, cnt = 0;
  real kink = 1;
  real cur_alpha = 0;
#define EVAL(alpha_) (cur_alpha = (alpha_), ls_eval(m, d, c, jv, quad, qg, cur_alpha, &cnt, &kink))
  /* following: the number of active rows is compared AT THE OTHER SIDE'S trial step length (its float32 bits are in
   * the record), so that a mismatch can only come from a row sitting on its switching point; how far the two sides'
   * trial step lengths are apart is reported separately ([8] of the report, relative to the larger of them) */
#define NOTE()                                                                              \
  do {                                                                                      \
    if (tr && nev < 20) {                                                                   \
      float a32_ = (float)cur_alpha;                                                        \
      tr[4 + nev] = cnt;                                                                    \
      memcpy(&tr[24 + nev], &a32_, 4);                                                      \
    }                                                                                       \
    if (fol && report && nev < 20) {                                                        \
      float fa_;                                                                            \
      memcpy(&fa_, &fol[24 + nev], 4);                                                      \
      int cnt_f_ = 0;                                                                       \
      real kink_f_ = 1;                                                                     \
      (void)ls_eval(m, d, c, jv, quad, qg, (real)fa_, &cnt_f_, &kink_f_);                   \
      if (fol[4 + nev] != cnt_f_) {                                                         \
        report[3] += 1;                                                                     \
        if (kink_f_ > report[4]) report[4] = kink_f_;                                       \
      }                                                                                     \
      real am_ = RFABS(cur_alpha) > RFABS((real)fa_) ? RFABS(cur_alpha) : RFABS((real)fa_); \
      real da_ = am_ > 0 ? RFABS(cur_alpha - (real)fa_) / am_ : 0;                          \
      if (da_ > report[8]) report[8] = da_;                                                 \
    }                                                                                       \
    nev++;                                                                                  \
  } while (0)
  ls_point p0 = EVAL(0);
  NOTE();
  ls_point lo = EVAL(p0.alpha - p0.deriv_0 / p0.deriv_1), hi;
  NOTE();
  int first_lo = lo.deriv_0 < p0.deriv_0; /* the first Newton point becomes `lo`, p0 becomes `hi` */
  if (fol) {
    int f = (fol[2] >> 24) & 1;
    if (report && f != first_lo) report[5] += 1;
    first_lo = f;
  }
  if (tr) tr[2] |= first_lo << 24;
  if (first_lo) {
    hi = p0;
  } else {
    hi = lo, lo = p0;
  }
  int swap = 1, ls_iter = 0;
  for (;;) {
    int done = ls_iter >= m->ls_iterations;
    done |= !swap;
    done |= (lo.deriv_0 < 0) && (lo.deriv_0 > -gtol);
    done |= (hi.deriv_0 > 0) && (hi.deriv_0 < gtol);
    if (fol) {
      int fdone = ls_iter >= fol[1];
      if (report && fdone != done) report[5] += 1;
      done = fdone;
    }
    if (done) break;
    ls_point lo_next = EVAL(lo.alpha - lo.deriv_0 / lo.deriv_1);
    NOTE();
    ls_point hi_next = EVAL(hi.alpha - hi.deriv_0 / hi.deriv_1);
    NOTE();
    ls_point mid = EVAL((real)0.5 * (lo.alpha + hi.alpha));
    NOTE();
    int swap_lo_next = 0, swap_lo_mid = 0, swap_hi_next = 0, swap_hi_mid = 0;
    if (fol) {
      int bits = (fol[2] >> (4 * ls_iter)) & 15;
      int n1 = (lo.deriv_0 > 0) || (lo.deriv_0 < lo_next.deriv_0);
      swap_lo_next = bits & 1;
      if (swap_lo_next) lo = lo_next;
      int n2 = (mid.deriv_0 < 0) && (lo.deriv_0 < mid.deriv_0);
      swap_lo_mid = (bits >> 1) & 1;
      if (swap_lo_mid) lo = mid;
      int n3 = (hi.deriv_0 < 0) || (hi.deriv_0 > hi_next.deriv_0);
      swap_hi_next = (bits >> 2) & 1;
      if (swap_hi_next) hi = hi_next;
      int n4 = (mid.deriv_0 > 0) && (hi.deriv_0 > mid.deriv_0);
      swap_hi_mid = (bits >> 3) & 1;
      if (swap_hi_mid) hi = mid;
      if (report) report[5] += (n1 != swap_lo_next) + (n2 != swap_lo_mid) + (n3 != swap_hi_next) + (n4 != swap_hi_mid);
    } else {
      if (g_opt.ls_mid_first) { /* alternative order of the replacement tests (Appendix B item 3) */
        swap_lo_mid = (mid.deriv_0 < 0) && (lo.deriv_0 < mid.deriv_0);
        if (swap_lo_mid) lo = mid;
        swap_hi_mid = (mid.deriv_0 > 0) && (hi.deriv_0 > mid.deriv_0);
        if (swap_hi_mid) hi = mid;
      }
      swap_lo_next = (lo.deriv_0 > 0) || (lo.deriv_0 < lo_next.deriv_0);
      if (swap_lo_next) lo = lo_next;
      if (!g_opt.ls_mid_first) {
        swap_lo_mid = (mid.deriv_0 < 0) && (lo.deriv_0 < mid.deriv_0);
        if (swap_lo_mid) lo = mid;
      }
      swap_hi_next = (hi.deriv_0 < 0) || (hi.deriv_0 > hi_next.deriv_0);
      if (swap_hi_next) hi = hi_next;
      if (!g_opt.ls_mid_first) {
        swap_hi_mid = (mid.deriv_0 > 0) && (hi.deriv_0 > mid.deriv_0);
        if (swap_hi_mid) hi = mid;
      }
    }
    swap = swap_lo_next | swap_lo_mid | swap_hi_next | swap_hi_mid;
    if (tr && ls_iter < 6) tr[2] |= (swap_lo_next | swap_lo_mid << 1 | swap_hi_next << 2 | swap_hi_mid << 3) << (4 * ls_iter);
    ls_iter++;
  }
#undef EVAL
#undef NOTE
  int improved = (lo.cost < p0.cost) || (hi.cost < p0.cost);
  int pick_lo = g_opt.ls_tie_lo ? lo.cost <= hi.cost : lo.cost < hi.cost;
  int pick = improved ? (pick_lo ? 1 : 2) : 0;
  if (fol) {
    if (report && fol[3] != pick) report[5] += 1;
    pick = fol[3];
  }
  real alpha = pick == 0 ? 0 : (pick == 1 ? lo.alpha : hi.alpha);
  if (tr) {
    float a32 = (float)alpha;
    memcpy(&tr[0], &a32, 4);
    tr[1] = ls_iter, tr[3] = pick;
  }
  return alpha;
}

static void slv_linesearch(const orc_model *m, orc_data *d, slv_ctx *c, int *tr /* 32 ints of this iteration */,
                           const int *fol, real *report) {
  int nv = m->nv, ne = m->nefc;
  real *mv = d->w0, *jv = d->wefc1, *quad = d->quad;
  real snorm = 0;
  for (int i = 0; i < nv; i++) snorm += c->search[i] * c->search[i];
  real smag = RSQRT(snorm) * m->meaninertia * (nv > 1 ? nv : 1);
  real gtol = m->tolerance * m->ls_tolerance * smag;
  mul_m(m, d, mv, c->search);
  for (int r = 0; r < ne; r++) {
    real s = 0;
    for (int i = 0; i < nv; i++) s += d->efc_J[r * nv + i] * c->search[i];
    jv[r] = s;
  }
  real qg[3] = {c->gauss, 0, 0};
  for (int i = 0; i < nv; i++) {
    qg[1] += c->search[i] * c->Ma[i] - c->search[i] * d->qfrc_smooth[i];
    qg[2] += c->search[i] * mv[i];
  }
  qg[2] *= (real)0.5;
  for (int r = 0; r < ne; r++) {
    quad[3 * r] = (real)0.5 * c->Jaref[r] * c->Jaref[r] * d->efc_D[r];
    quad[3 * r + 1] = jv[r] * c->Jaref[r] * d->efc_D[r];
    quad[3 * r + 2] = (real)0.5 * jv[r] * jv[r] * d->efc_D[r];
  }
  real alpha = ls_run(m, d, c, jv, quad, qg, gtol, tr, fol, report);
  if (fol && report) { /* how much worse than this oracle's own line search is the followed step? */
    real a_nat = ls_run(m, d, c, jv, quad, qg, gtol, NULL, NULL, NULL);
    ls_point pf = ls_eval(m, d, c, jv, quad, qg, alpha, NULL, NULL), pn = ls_eval(m, d, c, jv, quad, qg, a_nat, NULL, NULL);
    real am = RFABS(alpha) > RFABS(a_nat) ? RFABS(alpha) : RFABS(a_nat);
    real scale = c->cost_scale + am * am * RFABS(qg[2]); /* magnitudes of the terms of the polynomial at |alpha| = am */
    for (int i = 0; i < nv; i++) scale += am * RFABS(c->search[i]) * (RFABS(c->Ma[i]) + RFABS(d->qfrc_smooth[i]));
    for (int r = 0; r < ne; r++) scale += RFABS(am * quad[3 * r + 1]) + am * am * RFABS(quad[3 * r + 2]);
    real excess = (pf.cost - pn.cost) / ((real)1e-6 * scale + MJ_MINVAL);
    if (excess > report[0]) report[0] = excess;
  }
  if (alpha != 0) {
    for (int i = 0; i < nv; i++) c->qacc[i] += c->search[i] * alpha, c->Ma[i] += mv[i] * alpha;
    for (int r = 0; r < ne; r++) c->Jaref[r] += jv[r] * alpha;
  }
}

void orc_solve(const orc_model *m, orc_data *d) {
  int nv = m->nv, ne = m->nefc;
  const int *fol = d->follow;
  real *report = fol ? d->follow_report : NULL;
  if (report) { /* [6], [7] were written by make_constraint of this forward pass */
    memset(report, 0, sizeof(real) * 6);
    memset(report + 8, 0, sizeof(real) * 4);
  }
  slv_ctx c;
  c.qacc = ralloc(nv), c.Ma = ralloc(nv), c.Jaref = ralloc(ne), c.grad = ralloc(nv), c.Mgrad = ralloc(nv);
  c.search = ralloc(nv), c.qfrc_constraint = ralloc(nv), c.efc_force = ralloc(ne);
  /* warmstart: pick the cheaper of qacc_warmstart and qacc_smooth */
  slv_init(m, d, &c, d->qacc_warmstart);
  real cost_warm = c.cost, scale_warm = c.cost_scale;
  slv_init(m, d, &c, d->qacc_smooth);
  real cost_smooth = c.cost, scale_smooth = c.cost_scale;
  int use_warm = cost_warm < cost_smooth;
  if (fol && fol[0] != use_warm) {
    report[2] = RFABS(cost_warm - cost_smooth) / ((real)1e-6 * (scale_warm + scale_smooth) + MJ_MINVAL);
    report[5] += 1;
    use_warm = fol[0];
  }
  const real *start = use_warm ? d->qacc_warmstart : d->qacc_smooth;
  memset(d->trace, 0, sizeof(d->trace));
  d->trace[0] = use_warm;
  for (int r = 0; r < ne && r < 512; r++)
    if (d->row_live[r]) d->trace[ORC_TRACE_ROWS + (r >> 5)] |= 1 << (r & 31);
  slv_init(m, d, &c, start);
  slv_update_gradient(m, d, &c);
  for (int i = 0; i < nv; i++) c.search[i] = -c.Mgrad[i];
  real scale = m->meaninertia * (nv > 1 ? nv : 1);
  int niter = 0;
  real *prev_grad = d->w1, *prev_Mgrad = d->w2;
  for (;;) {
    real improvement = (c.prev_cost - c.cost) / scale;
    real gn = 0;
    for (int i = 0; i < nv; i++) gn += c.grad[i] * c.grad[i];
    real gradient = RSQRT(gn) / scale;
    int done = niter >= m->iterations;
    done |= improvement < m->tolerance;
    done |= gradient < m->tolerance;
    if (fol) {
      int fdone = niter >= fol[1];
      if (fdone != done) {
        /* distance of the two exit tests from their thresholds, in units of the float32 rounding of their operands */
        real r_imp = RFABS(c.prev_cost - c.cost - m->tolerance * scale) / ((real)1e-6 * (c.prev_cost_scale + c.cost_scale) + MJ_MINVAL);
        real gs = 0;
        for (int i = 0; i < nv; i++) {
          real t = RFABS(c.Ma[i]) + RFABS(d->qfrc_smooth[i]) + RFABS(c.qfrc_constraint[i]);
          gs += t * t;
        }
        real r_grd = RFABS(gradient - m->tolerance) / ((real)1e-6 * RSQRT(gs) / scale + MJ_MINVAL);
        real r = r_imp < r_grd ? r_imp : r_grd;
        if (!isfinite((double)c.prev_cost)) r = r_grd; /* first pass: prev_cost = inf, only the gradient test can fire */
        if (r > report[1]) report[1] = r;
        report[5] += 1;
      }
      done = fdone;
    }
    if (done && !(m->iterations == 1 && niter == 0)) break;
    slv_linesearch(m, d, &c, niter < ORC_TRACE_ITERS ? d->trace + 8 + ORC_TRACE_REC * niter : NULL,
                   (fol && niter < ORC_TRACE_ITERS) ? fol + 8 + ORC_TRACE_REC * niter : NULL, report);
    memcpy(prev_grad, c.grad, sizeof(real) * nv), memcpy(prev_Mgrad, c.Mgrad, sizeof(real) * nv);
    slv_update_constraint(m, d, &c);
    slv_update_gradient(m, d, &c);
    real num = 0, den = 0;
    for (int i = 0; i < nv; i++) num += c.grad[i] * (c.Mgrad[i] - prev_Mgrad[i]), den += prev_grad[i] * prev_Mgrad[i];
    real beta = num / (den > MJ_MINVAL ? den : MJ_MINVAL);
    if (beta < 0) beta = 0;
    if (m->solver_newton) beta = 0; /* solver.solve: search = -Mgrad */
    for (int i = 0; i < nv; i++) c.search[i] = -c.Mgrad[i] + beta * c.search[i];
    niter++;
  }
  d->solver_niter = niter;
  d->trace[1] = niter;
  memcpy(d->qacc, c.qacc, sizeof(real) * nv), memcpy(d->qacc_warmstart, c.qacc, sizeof(real) * nv);
  memcpy(d->qfrc_constraint, c.qfrc_constraint, sizeof(real) * nv);
  memcpy(d->efc_force, c.efc_force, sizeof(real) * ne);
  free(c.qacc), free(c.Ma), free(c.Jaref), free(c.grad), free(c.Mgrad), free(c.search), free(c.qfrc_constraint),
      free(c.efc_force);
}

/* forward.forward [UPSTREAM] */
void orc_forward(const orc_model *m, orc_data *d) {
  d->follow_report[6] = 0, d->follow_report[7] = 0;
  orc_kinematics(m, d);
  orc_com_pos(m, d);
  orc_crb(m, d);
  orc_factor_m(m, d);
  orc_collision(m, d);
  orc_make_constraint(m, d);
  orc_com_vel(m, d);
  orc_passive(m, d);
  orc_rne(m, d);
  orc_actuation(m, d);
  orc_acceleration(m, d);
  orc_solve(m, d);
}

/* forward.euler + _advance [UPSTREAM] */
void orc_euler(const orc_model *m, orc_data *d) {
  int nv = m->nv;
  real *qacc = d->w3;
  memcpy(qacc, d->qacc, sizeof(real) * nv);
  if (m->eulerdamp) {
    real *Mh = (real *)malloc(sizeof(real) * nv * nv), *Lh = (real *)malloc(sizeof(real) * nv * nv);
    memcpy(Mh, d->qM, sizeof(real) * nv * nv);
    for (int i = 0; i < nv; i++) Mh[i * nv + i] += m->timestep * m->dof_damping[i];
    chol_factor(Lh, Mh, nv);
    for (int i = 0; i < nv; i++) qacc[i] = d->qfrc_smooth[i] + d->qfrc_constraint[i];
    chol_solve(Lh, qacc, nv);
    free(Mh), free(Lh);
  }
  for (int i = 0; i < m->nu; i++)
    if (m->act_tau[i] >= 0) d->act[i] += d->act_dot[i] * m->timestep;
  for (int i = 0; i < nv; i++) d->qvel[i] += qacc[i] * m->timestep;
  for (int j = 0; j < m->njnt; j++) {
    int qa = m->jnt_qposadr[j], da = m->jnt_dofadr[j];
    if (m->jnt_type[j] == JNT_FREE) {
      for (int i = 0; i < 3; i++) d->qpos[qa + i] += m->timestep * d->qvel[da + i];
      real v[3] = {d->qvel[da + 3], d->qvel[da + 4], d->qvel[da + 5]}, n = norm3(v), qr[4], q2[4];
      if (n > 0) v[0] /= n, v[1] /= n, v[2] /= n;
      axis_angle_to_quat(qr, v, m->timestep * n);
      quat_mul(q2, d->qpos + qa + 3, qr);
      normalize4(q2);
      memcpy(d->qpos + qa + 3, q2, sizeof(q2));
    } else {
      d->qpos[qa] += m->timestep * d->qvel[da];
    }
  }
}

void orc_step(const orc_model *m, orc_data *d) {
  orc_forward(m, d);
  orc_euler(m, d);
}

/* ------------------------------------------------------------------------- */
/* env glue: reference envs/rodent.py                                        */
/* ------------------------------------------------------------------------- */
typedef struct orc_envspec {
  int T;               /* frames in the clip arrays */
  int ref_len;         /* ref_traj_length (rodent.py:34) */
  int sub_clip_length; /* rodent.py:33 */
  int n_frames;        /* physics substeps per control step (rodent.py:97) */
  int nb;              /* width of filtered clip.body_positions (rodent.py:113-115) */
  int nee, napp, njc;  /* end effectors, appendages, joint columns */
  int body_idxs[64];   /* model body ids -> xpos gather (rodent.py:80-85) */
  int end_eff_idx[8];  /* model body ids (rodent.py:65-70) */
  int app_body[8];     /* model body ids for xpos[_app_idx] (rodent.py:307) */
  int app_ref_col[8];  /* _app_idx applied to the nb-wide clip axis, clamped (quirk C.4) */
  int com_ref_col;     /* _com_idx applied to the nb-wide clip axis, clamped (quirk C.4) */
  int joint_cols[128]; /* _joint_idxs applied to the (nq-7)-wide joints axis, clamped (quirk C.5) */
  double healthy_z_lo, healthy_z_hi, termination_threshold, body_error_multiplier;
  int flags;             /* ENV_* below: RodentTracking (0) or HumanoidTracking-style glue */
  double done_threshold; /* done when the UNSCALED rtrunk is below it (rodent.py:213: 0; humanoid.py:199: 0.5) */
  double reward_weights[6]; /* with ENV_WEIGHTS: rcom, rvel, rtrunk, rquat, ract, rapp (ant.py:182-188) */
} orc_envspec;
#define ENV_REWARD_OLD_STATE 1 /* humanoid.py:195: _calculate_reward(state, action) uses the state BEFORE the step */
#define ENV_TERM_MEAN 2        /* humanoid.py:256-260: means of |.| instead of the matrix-1 / L1 norms */
#define ENV_NO_RAPP 4
#define ENV_OBS_QPOS_QVEL 8    /* humanoid.py:354-368 */
#define ENV_WEIGHTS 16         /* ant.py:182-188: total = 0.05 rcom + 0.01 rvel + 0.20 rtrunk + 0.01 rquat + 0.001 ract */
#define ENV_RACT_ACTION 32     /* ant.py:277: ract = 0.01 * -0.015 * sum(action^2) / len(action) */
#define ENV_METRICS_UNSCALED 64 /* ant.py:197,216-225: metrics / termination_error hold the unweighted terms */
#define ENV_TRAJ_OLD_FRAME 128 /* ant.py:178: _get_obs(data, action, state.info) -- the un-incremented cur_frame */

typedef struct orc_clip {
  const float *position, *quaternion, *joints, *body_positions, *velocity, *angular_velocity, *joints_velocity;
  const float *center_of_mass; /* (T,3) or NULL: reference of rcom (humanoid.py:273) else body_positions[com_ref_col] */
} orc_clip;

static inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* rodent.py:241-264 _calculate_termination (matrix 1-norm on bodies: quirk C.3) */
static real env_termination(const orc_model *m, const orc_envspec *e, const orc_clip *c, const real *qpos,
                            const real *xpos, int cur_frame) {
  int f = clampi(cur_frame, 0, e->T - 1), nj = m->nq - 7;
  real ej = 0;
  for (int i = 0; i < nj; i++) ej += RFABS((real)c->joints[f * nj + i] - qpos[7 + i]);
  real col[3] = {0, 0, 0};
  for (int k = 0; k < e->nb; k++)
    for (int i = 0; i < 3; i++)
      col[i] += RFABS((real)c->body_positions[(f * e->nb + k) * 3 + i] - xpos[3 * e->body_idxs[k] + i]);
  real eb = col[0] > col[1] ? col[0] : col[1];
  eb = eb > col[2] ? eb : col[2];
  if (e->flags & ENV_TERM_MEAN) eb = (col[0] + col[1] + col[2]) / (real)(3 * e->nb), ej = ej / (real)nj;
  real err = (real)0.5 * (real)e->body_error_multiplier * eb + (real)0.5 * ej;
  return 1 - err / (real)e->termination_threshold;
}

/* rodent.py:318-344 _get_obs */
static void env_obs(const orc_model *m, const orc_envspec *e, const orc_data *d, real *obs) {
  int k = 0;
  for (int i = 0; i < m->nq; i++) obs[k++] = d->qpos[i];
  for (int i = 0; i < m->nv; i++) obs[k++] = d->qvel[i];
  if (e->flags & ENV_OBS_QPOS_QVEL) return;
  for (int i = 0; i < m->nv; i++) obs[k++] = d->qfrc_actuator[i];
  for (int j = 0; j < e->nee; j++)
    for (int i = 0; i < 3; i++) obs[k++] = d->xpos[3 * e->end_eff_idx[j] + i];
}

/* rodent.py:346-448 _get_traj and helpers; local frame = v @ xmat[1] */
static void env_traj(const orc_model *m, const orc_envspec *e, const orc_clip *c, const orc_data *d, int cur_frame,
                     real *traj) {
  int L = e->ref_len, s = clampi(cur_frame + 1, 0, e->T - L), nj = m->nq - 7, k = 0;
  const real *R = d->xmat + 9; /* xmat[1] */
  for (int t = 0; t < L; t++) /* get_reference_appendages_pos */
    for (int a = 0; a < e->napp; a++)
      for (int i = 0; i < 3; i++) traj[k++] = (real)c->body_positions[((s + t) * e->nb + e->app_ref_col[a]) * 3 + i];
  for (int t = 0; t < L; t++) /* rel bodies, local */
    for (int b = 0; b < e->nb; b++) {
      real v[3];
      for (int i = 0; i < 3; i++)
        v[i] = (real)c->body_positions[((s + t) * e->nb + b) * 3 + i] - d->xpos[3 * e->body_idxs[b] + i];
      for (int i = 0; i < 3; i++) traj[k++] = v[0] * R[i] + v[1] * R[3 + i] + v[2] * R[6 + i];
    }
  for (int t = 0; t < L; t++) /* rel bodies, global */
    for (int b = 0; b < e->nb; b++)
      for (int i = 0; i < 3; i++)
        traj[k++] = (real)c->body_positions[((s + t) * e->nb + b) * 3 + i] - d->xpos[3 * e->body_idxs[b] + i];
  for (int t = 0; t < L; t++) { /* root, local */
    real v[3];
    for (int i = 0; i < 3; i++) v[i] = (real)c->position[(s + t) * 3 + i] - d->qpos[i];
    for (int i = 0; i < 3; i++) traj[k++] = v[0] * R[i] + v[1] * R[3 + i] + v[2] * R[6 + i];
  }
  for (int t = 0; t < L; t++) /* joints */
    for (int j = 0; j < e->njc; j++) {
      int col = e->joint_cols[j];
      traj[k++] = (real)c->joints[(s + t) * nj + col] - d->qpos[7 + col];
    }
}

static int data_has_nan(const orc_model *m, const orc_data *d) {
  int bad = 0;
  for (int i = 0; i < m->nq; i++) bad |= isnan(d->qpos[i]);
  for (int i = 0; i < m->nv; i++) bad |= isnan(d->qvel[i]) | isnan(d->qacc_warmstart[i]) | isnan(d->qfrc_actuator[i]);
  for (int i = 0; i < m->nu; i++) bad |= isnan(d->act[i]);
  for (int i = 0; i < 3 * m->nbody; i++) bad |= isnan(d->xpos[i]) | isnan(d->subtree_com[i]);
  return bad;
}

/*
 * Batched state, AoS row-major (B, n).  Arrays are `real`: float32 in the ORC_F32 build (as
 * JAX arrays are), float64 in the default build so that step-to-step state carries no
 * float32 quantisation (the dynamics amplify a 1e-7 perturbation ~1e4x per control step).
 * metrics: rcom rvel rtrunk rquat ract rapp termination_error (rodent.py:158-166)
 */
typedef struct orc_state {
  real *qpos, *qvel, *act, *qacc_warmstart; /* carried physics state */
  real *xpos, *xmat1, *com1, *qfrc_actuator; /* derived, from the last forward */
  real *obs, *traj, *reward, *done, *metrics;
  int32_t *cur_frame, *sub_clip_frame;
  real *termination_error;
} orc_state;

static void load_state(const orc_model *m, orc_data *d, const orc_state *s, int i) {
  for (int k = 0; k < m->nq; k++) d->qpos[k] = s->qpos[(size_t)i * m->nq + k];
  for (int k = 0; k < m->nv; k++) d->qvel[k] = s->qvel[(size_t)i * m->nv + k];
  for (int k = 0; k < m->nv; k++) d->qacc_warmstart[k] = s->qacc_warmstart[(size_t)i * m->nv + k];
  for (int k = 0; k < m->nu; k++) d->act[k] = s->act[(size_t)i * m->nu + k];
}
static void store_state(const orc_model *m, const orc_data *d, orc_state *s, int i) {
  for (int k = 0; k < m->nq; k++) s->qpos[(size_t)i * m->nq + k] = (real)d->qpos[k];
  for (int k = 0; k < m->nv; k++) s->qvel[(size_t)i * m->nv + k] = (real)d->qvel[k];
  for (int k = 0; k < m->nv; k++) s->qacc_warmstart[(size_t)i * m->nv + k] = (real)d->qacc_warmstart[k];
  for (int k = 0; k < m->nu; k++) s->act[(size_t)i * m->nu + k] = (real)d->act[k];
  for (int k = 0; k < 3 * m->nbody; k++) s->xpos[(size_t)i * 3 * m->nbody + k] = (real)d->xpos[k];
  for (int k = 0; k < 9; k++) s->xmat1[(size_t)i * 9 + k] = (real)d->xmat[9 + k];
  for (int k = 0; k < 3; k++) s->com1[(size_t)i * 3 + k] = (real)d->subtree_com[3 + k];
  for (int k = 0; k < m->nv; k++) s->qfrc_actuator[(size_t)i * m->nv + k] = (real)d->qfrc_actuator[k];
}

static int obs_size(const orc_model *m, const orc_envspec *e) {
  return (e->flags & ENV_OBS_QPOS_QVEL) ? m->nq + m->nv : m->nq + 2 * m->nv + 3 * e->nee;
}
static int traj_size(const orc_envspec *e) { return e->ref_len * (3 * e->napp + 6 * e->nb + 3 + e->njc); }

/* rodent.py:119-176 reset: explicit start_frame and (already scaled) noise replace the JAX PRNG */
/* follow: NULL, or the other side's solver traces [B][n_frames][ORC_TRACE_INTS] (the reset's forward pass is entry 0 of an
 * env) whose decisions the oracle takes instead of its own; report: [B][ORC_FOLLOW_REPORT] */
int orc_env_reset_follow(const orc_model *m, const orc_envspec *e, const orc_clip *c, int B, const int32_t *start_frame,
                         const real *noise, orc_state *s, const int32_t *follow, real *report) {
  int nj = m->nq - 7, no = obs_size(m, e), nt = traj_size(e);
  real *obs = ralloc(no), *traj = ralloc(nt);
  orc_data *d = orc_data_create(m);
  for (int i = 0; i < B; i++) {
    int f = clampi(start_frame[i], 0, e->T - 1);
    for (int k = 0; k < 3; k++) d->qpos[k] = (real)c->position[f * 3 + k];
    for (int k = 0; k < 4; k++) d->qpos[3 + k] = (real)c->quaternion[f * 4 + k];
    for (int k = 0; k < nj; k++) d->qpos[7 + k] = (real)c->joints[f * nj + k];
    for (int k = 0; k < m->nq; k++) d->qpos[k] += (real)noise[(size_t)i * m->nq + k];
    for (int k = 0; k < 3; k++) d->qvel[k] = (real)c->velocity[f * 3 + k];
    for (int k = 0; k < 3; k++) d->qvel[3 + k] = (real)c->angular_velocity[f * 3 + k];
    for (int k = 0; k < nj; k++) d->qvel[6 + k] = (real)c->joints_velocity[f * nj + k];
    memset(d->act, 0, sizeof(real) * m->nu), memset(d->ctrl, 0, sizeof(real) * m->nu);
    memset(d->qacc_warmstart, 0, sizeof(real) * m->nv);
    d->follow = follow ? (const int *)follow + (size_t)i * (e->n_frames > 0 ? e->n_frames : 1) * ORC_TRACE_INTS : NULL;
    orc_forward(m, d); /* brax pipeline_init = mjx.forward */
    d->follow = NULL;
    if (report && follow) memcpy(report + (size_t)i * ORC_FOLLOW_REPORT, d->follow_report, sizeof(d->follow_report));
    if (g_opt.reset_warmstart_zero) memset(d->qacc_warmstart, 0, sizeof(real) * m->nv);
    store_state(m, d, s, i);
    env_traj(m, e, c, d, start_frame[i], traj);
    env_obs(m, e, d, obs);
    for (int k = 0; k < no; k++) s->obs[(size_t)i * no + k] = (real)obs[k];
    for (int k = 0; k < nt; k++) s->traj[(size_t)i * nt + k] = (real)traj[k];
    s->reward[i] = 0, s->done[i] = 0;
    for (int k = 0; k < 7; k++) s->metrics[(size_t)i * 7 + k] = 0;
    s->cur_frame[i] = start_frame[i], s->sub_clip_frame[i] = 0;
    s->termination_error[i] = (real)env_termination(m, e, c, d->qpos, d->xpos, start_frame[i]);
  }
  orc_data_destroy(d);
  free(obs), free(traj);
  return 0;
}

int orc_env_reset(const orc_model *m, const orc_envspec *e, const orc_clip *c, int B, const int32_t *start_frame,
                  const real *noise, orc_state *s) {
  return orc_env_reset_follow(m, e, c, B, start_frame, noise, s, NULL, NULL);
}

/* rodent.py:183-239 / humanoid.py:190-239: everything of `step` after pipeline_step.  `d` holds the NEW pipeline state
 * (qpos, qvel, act, qacc_warmstart and, from its last forward, xpos, xmat[1], subtree_com[1], qfrc_actuator); old_* the
 * state before the step.  Writes row i of the outputs and advances the frame counters. */
typedef struct {
  const real *qpos, *qvel, *com1, *qfrc_actuator, *xpos;
} reward_state;

static void env_glue(const orc_model *m, const orc_envspec *e, const orc_clip *c, const orc_data *d, const real *old_qpos,
                     const real *old_xpos, const reward_state *old /* state before the step, for ENV_REWARD_OLD_STATE */,
                     const real *action /* this env's action, for ENV_RACT_ACTION (else unused) */, real *obs, real *traj,
                     orc_state *s, int i) {
  int nj = m->nq - 7, no = obs_size(m, e), nt = traj_size(e);
  int old_frame = s->cur_frame[i], new_frame = old_frame + 1, new_sub = s->sub_clip_frame[i] + 1;
  env_obs(m, e, d, obs);
  env_traj(m, e, c, d, (e->flags & ENV_TRAJ_OLD_FRAME) ? old_frame : new_frame, traj);
  /* _calculate_reward: rodent.py:195 passes the NEW data (quirk C.1: against the clip row at the OLD cur_frame);
   * humanoid.py:195 passes `state`, i.e. every term comes from the state BEFORE the step */
  reward_state cur = {d->qpos, d->qvel, d->subtree_com + 3, d->qfrc_actuator, d->xpos};
  const reward_state *r = (e->flags & ENV_REWARD_OLD_STATE) ? old : &cur;
  int fo = clampi(old_frame, 0, e->T - 1);
  real dv[3], acc;
  for (int k = 0; k < 3; k++) {
    real ref = c->center_of_mass ? (real)c->center_of_mass[fo * 3 + k] : (real)c->body_positions[(fo * e->nb + e->com_ref_col) * 3 + k];
    dv[k] = r->com1[k] - ref;
  }
  real rcom = REXP(-100 * norm3(dv));
  acc = 0;
  for (int k = 0; k < 3; k++) {
    real a = r->qvel[k] - (real)c->velocity[fo * 3 + k], b = r->qvel[3 + k] - (real)c->angular_velocity[fo * 3 + k];
    acc += a * a + b * b;
  }
  for (int k = 0; k < nj; k++) {
    real a = r->qvel[6 + k] - (real)c->joints_velocity[fo * nj + k];
    acc += a * a;
  }
  real rvel = REXP((real)-0.1 * RSQRT(acc));
  /* rtrunk from the OLD pipeline state and OLD frame (quirk C.2) */
  real rtrunk = env_termination(m, e, c, old_qpos, old_xpos, old_frame);
  real qc[4], qr[4], nc = 0, nr = 0, dq = 0;
  for (int k = 0; k < 4; k++) qc[k] = r->qpos[3 + k], qr[k] = (real)c->quaternion[fo * 4 + k];
  for (int k = 0; k < 4; k++) nc += qc[k] * qc[k], nr += qr[k] * qr[k];
  nc = RSQRT(nc), nr = RSQRT(nr);
  for (int k = 0; k < 4; k++) dq += (qc[k] / nc) * (qr[k] / nr);
  real dist = 2 * dq * dq - 1;
  if (dist > 1) dist = 1;
  real rquat = REXP(-2 * RFABS((real)0.5 * RACOS(dist)));
  acc = 0;
  real ract;
  if (e->flags & ENV_RACT_ACTION) {
    for (int k = 0; k < m->nu; k++) acc += action[k] * action[k];
    ract = (real)0.01 * (real)-0.015 * acc / m->nu;
  } else {
    for (int k = 0; k < m->nv; k++) acc += r->qfrc_actuator[k] * r->qfrc_actuator[k];
    ract = (real)-0.015 * (acc / m->nv);
  }
  real rapp = 0;
  if (!(e->flags & ENV_NO_RAPP)) {
    acc = 0;
    for (int a = 0; a < e->napp; a++)
      for (int k = 0; k < 3; k++) {
        real x = r->xpos[3 * e->app_body[a] + k] - (real)c->body_positions[(fo * e->nb + e->app_ref_col[a]) * 3 + k];
        acc += x * x;
      }
    rapp = REXP(-400 * RSQRT(acc));
  }
  real healthy = 1;
  if (r->qpos[2] < (real)e->healthy_z_lo) healthy = 0;
  if (r->qpos[2] > (real)e->healthy_z_hi) healthy = 0;
  real done = rtrunk < (real)e->done_threshold ? 1 : 0; /* on the unscaled value */
  static const double builtin_w[6] = {0.01, 0.01, 0.01, 0.01, 0.0001, 0.01}; /* rodent.py:203-209 */
  const double *w = (e->flags & ENV_WEIGHTS) ? e->reward_weights : builtin_w;
  real wcom = rcom * (real)w[0], wvel = rvel * (real)w[1], wtrunk = rtrunk * (real)w[2], wquat = rquat * (real)w[3];
  real wact = ract * (real)w[4], wapp = rapp * (real)w[5];
  real total = wcom + wvel + wtrunk + wquat + wact + wapp;
  if (!(e->flags & ENV_METRICS_UNSCALED)) /* rodent.py:228-236 logs the weighted terms, ant.py:216-225 the raw ones */
    rcom = wcom, rvel = wvel, rtrunk = wtrunk, rquat = wquat, ract = wact, rapp = wapp;
  if (1 - healthy > done) done = 1 - healthy;
  real sub_ok = new_sub < e->sub_clip_length ? 1 : 0;
  if (1 - sub_ok > done) done = 1 - sub_ok;
  if (isnan(total)) total = 0; /* nan_to_num */
  if (isinf(total)) total = total > 0 ? (real)3.4028235e38 : (real)-3.4028235e38;
  for (int k = 0; k < no; k++)
    if (isnan(obs[k])) obs[k] = 0;
  if (data_has_nan(m, d)) done = 1;
  for (int k = 0; k < no; k++) s->obs[(size_t)i * no + k] = (real)obs[k];
  for (int k = 0; k < nt; k++) s->traj[(size_t)i * nt + k] = (real)traj[k];
  s->reward[i] = (real)total, s->done[i] = (real)done;
  real *mt = s->metrics + (size_t)i * 7;
  mt[0] = (real)rcom, mt[1] = (real)rvel, mt[2] = (real)rtrunk, mt[3] = (real)rquat, mt[4] = (real)ract;
  mt[5] = (real)rapp, mt[6] = (real)rtrunk;
  s->cur_frame[i] = new_frame, s->sub_clip_frame[i] = new_sub;
  s->termination_error[i] = (real)rtrunk;
}

/* rodent.py:178-239 step.  trace: NULL or [B][n_frames][ORC_TRACE_INTS], the solver decisions of every substep */
/* follow / report: NULL, or [B][n_frames][ORC_TRACE_INTS] decisions to follow and [B][n_frames][8] legitimacy report */
int orc_env_step_follow(const orc_model *m, const orc_envspec *e, const orc_clip *c, int B, const real *action,
                        orc_state *s, int32_t *trace, const int32_t *follow, real *report) {
  int no = obs_size(m, e), nt = traj_size(e), nb3 = 3 * m->nbody;
#pragma omp parallel
  {
    real *obs = ralloc(no), *traj = ralloc(nt), *old_qpos = ralloc(m->nq), *old_xpos = ralloc(nb3);
    real *old_qvel = ralloc(m->nv), *old_qfrc = ralloc(m->nv), old_com[3];
    orc_data *d = orc_data_create(m);
#pragma omp for schedule(static)
    for (int i = 0; i < B; i++) {
      load_state(m, d, s, i);
      for (int k = 0; k < m->nq; k++) old_qpos[k] = d->qpos[k];
      for (int k = 0; k < nb3; k++) old_xpos[k] = (real)s->xpos[(size_t)i * nb3 + k];
      for (int k = 0; k < m->nv; k++) old_qvel[k] = d->qvel[k], old_qfrc[k] = s->qfrc_actuator[(size_t)i * m->nv + k];
      for (int k = 0; k < 3; k++) old_com[k] = s->com1[(size_t)i * 3 + k];
      reward_state old = {old_qpos, old_qvel, old_com, old_qfrc, old_xpos};
      for (int k = 0; k < m->nu; k++) d->ctrl[k] = (real)action[(size_t)i * m->nu + k];
      for (int f = 0; f < e->n_frames; f++) { /* brax pipeline_step */
        d->follow = follow ? (const int *)follow + ((size_t)i * e->n_frames + f) * ORC_TRACE_INTS : NULL;
        orc_step(m, d);
        d->follow = NULL;
        if (trace) memcpy(trace + ((size_t)i * e->n_frames + f) * ORC_TRACE_INTS, d->trace, sizeof(d->trace));
        if (report && follow)
          memcpy(report + ((size_t)i * e->n_frames + f) * ORC_FOLLOW_REPORT, d->follow_report, sizeof(d->follow_report));
      }
      store_state(m, d, s, i);
      env_glue(m, e, c, d, old_qpos, old_xpos, &old, action + (size_t)i * m->nu, obs, traj, s, i);
    }
    orc_data_destroy(d);
    free(obs), free(traj), free(old_qpos), free(old_xpos), free(old_qvel), free(old_qfrc);
  }
  return 0;
}
int orc_env_step_trace(const orc_model *m, const orc_envspec *e, const orc_clip *c, int B, const real *action,
                       orc_state *s, int32_t *trace) {
  return orc_env_step_follow(m, e, c, B, action, s, trace, NULL, NULL);
}
int orc_env_step(const orc_model *m, const orc_envspec *e, const orc_clip *c, int B, const real *action, orc_state *s) {
  return orc_env_step_follow(m, e, c, B, action, s, NULL, NULL, NULL);
}

/* The glue alone, on a pipeline state supplied by the caller (the product's own post-step state): `s` holds the
 * NEW qpos / qvel / act / qacc_warmstart / xpos / xmat1 / com1 / qfrc_actuator and the OLD frame counters; old_qpos
 * [B][nq], old_xpos [B][3 nbody] the state before the step.  Fills obs / traj / reward / done / metrics /
 * termination_error and advances the counters, exactly as orc_env_step does after its substeps. */
int orc_env_glue(const orc_model *m, const orc_envspec *e, const orc_clip *c, int B, const real *old_qpos,
                 const real *old_xpos, const real *old_qvel, const real *old_com1, const real *old_qfrc /* ENV_REWARD_OLD_STATE
                 only: [B][nv], [B][3], [B][nv] before the step, else NULL */, const real *action /* [B][nu], ENV_RACT_ACTION
                 only, else NULL */, orc_state *s) {
  int no = obs_size(m, e), nt = traj_size(e), nb3 = 3 * m->nbody;
  real *obs = ralloc(no), *traj = ralloc(nt);
  orc_data *d = orc_data_create(m);
  for (int i = 0; i < B; i++) {
    load_state(m, d, s, i);
    for (int k = 0; k < nb3; k++) d->xpos[k] = s->xpos[(size_t)i * nb3 + k];
    for (int k = 0; k < 9; k++) d->xmat[9 + k] = s->xmat1[(size_t)i * 9 + k];
    for (int k = 0; k < 3; k++) d->subtree_com[3 + k] = s->com1[(size_t)i * 3 + k];
    for (int k = 0; k < m->nv; k++) d->qfrc_actuator[k] = s->qfrc_actuator[(size_t)i * m->nv + k];
    reward_state old = {old_qpos + (size_t)i * m->nq, old_qvel ? old_qvel + (size_t)i * m->nv : NULL,
                        old_com1 ? old_com1 + (size_t)i * 3 : NULL, old_qfrc ? old_qfrc + (size_t)i * m->nv : NULL,
                        old_xpos + (size_t)i * nb3};
    if ((e->flags & ENV_REWARD_OLD_STATE) && !(old_qvel && old_com1 && old_qfrc)) return -1;
    if ((e->flags & ENV_RACT_ACTION) && !action) return -1;
    env_glue(m, e, c, d, old_qpos + (size_t)i * m->nq, old_xpos + (size_t)i * nb3, &old,
             action ? action + (size_t)i * m->nu : NULL, obs, traj, s, i);
  }
  orc_data_destroy(d);
  free(obs), free(traj);
  return 0;
}

int orc_envspec_size(void) { return (int)sizeof(orc_envspec); }
