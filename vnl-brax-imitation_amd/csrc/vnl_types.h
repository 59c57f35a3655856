// Device-side tables of the env kernels (DevModel, DevEnv, WsLayout: uploaded once as one KernelConsts block and read
// through the constant address space; DevState: the caller's buffers, by value).
#pragma once
#include <stdint.h>

// Arithmetic type of the kernels.  The product is float32 (JAX default in the
// reference).  Tests build a float64 host simulation (-DVNL_REAL=double) to show the
// algorithms agree with the dense oracle far below float32 rounding.
#ifndef VNL_REAL
#define VNL_REAL float
#endif
typedef VNL_REAL vreal;

#define VNL_JNT_FREE 0
#define VNL_JNT_HINGE 3
#define VNL_GEOM_SPHERE 2
#define VNL_GEOM_CAPSULE 3
#define VNL_GEOM_ELLIPSOID 4

/* Debug trace of the DISCRETE decisions of one solver call (written only when vnl_env_debug is on; the test-side CPU
 * checker records the same layout): [0] warm start used, [1] iterations, then per iteration 64 ints:
 * [0] float bits of the accepted step length, [1] line-search iterations, [2] the four bracket-replacement decisions
 * of every line-search iteration (4 bits each), [3] pick (0 none, 1 lo, 2 hi), [4..24) rows active at every trial
 * step length (p0, first Newton point, then lo_next / hi_next / mid per line-search iteration); bit 24 of [2]: the
 * first Newton point became `lo`; [24..44) the float bits of those 20 trial step lengths. */
#define VNL_TRACE_ITERS 8
#define VNL_TRACE_REC 64 /* ints per iteration record */
#define VNL_TRACE_ROWS (8 + VNL_TRACE_REC * VNL_TRACE_ITERS) /* then 16 ints: bit r set = constraint row r present (D != 0) */
#define VNL_TRACE_INTS (VNL_TRACE_ROWS + 16)

#define VNL_LIVE_MAX 128 /* constraint rows kept in the compact list of existing rows (line search: one or two per lane) */
#define VNL_BLK_W 13 /* entries per block of EnvWave::blk_apply: with 13 the rodent's rows (113 blocks) AND columns (127) fit two trips of 64 lanes */
#define VNL_FAC_LINES 6 /* pivots per factorisation step (scratch lines in the dead CG vectors) */

struct DevModel {
  int nq, nv, nu, nbody, njnt, ncg, ncon, nlimit, nefc, nM;
  int iterations, ls_iterations, eulerdamp, root_free, max_depth, jump_rounds;
  int solver_newton; /* opt.solver == NEWTON (reference configs/env_config.yaml:16-21): small models only (dense H in LDS) */
  int fac_steps; /* number of steps of the factorisation schedule */
  int fac_nleaf; /* low byte: leaf dofs of the tree if <= VNL_FAC_LINES (factor_rows can then carry and solve a right-hand
                    side), else 0; bits 8..: depth of the deepest of the rows 64 .. (second lane set) */
  int nbody_out; /* bodies of the model as given (rows of xpos / xquat): nbody counts the DYNAMIC bodies, welded ones folded into their parents */
  int path_runs; /* most runs of consecutive dofs / ancestor bodies any body's path has (<= 4): EnvWave::path_sum walks that many */
  int blk_cfg;   /* EnvWave::blk_apply: trips of the row form | trips of the column form << 4 | combine steps (row) << 8 | (column) << 12; 0 = no table */
  int dbg_stage, dbg_count; /* timing knob, see EnvWave::forward */
  vreal dt, tolerance, ls_tolerance, scale /* meaninertia * max(1,nv) */;
  vreal gx, gy, gz;
  vreal pnx, pny, pnz, ppx, ppy, ppz; /* plane normal / point */
  vreal t1x, t1y, t1z;                /* make_frame(n)[1] (frame[2] = n x t1) */
  vreal total_mass_inv;
  vreal root_px, root_py, root_pz; /* reference point when the root is not a free joint */
  // bodies
  const int *body_parent, *body_jntadr, *body_jntnum, *body_dofadr, *body_dofnum, *body_nsub, *body_lastdof;
  const int *body_pathseg;           /* [nbody][8]: 4 runs of consecutive dofs on the path root -> body, then 4 runs of
                                        consecutive ancestor bodies (incl. the body): begin | end << 8 (0 = none) */
  const unsigned char* jump; /* [jump_rounds][nbody]: 2^r-th ancestor body, 0 = none */
  const vreal *body_pos, *body_quat, *body_ipos, *body_inertia6, *body_mass;
  // welded (jointless) bodies are folded into their parents for the dynamics (vnl_lib.hip: fuse_welded_bodies); every body of
  // the model as given still gets its pose: out_dyn[ob] = the dynamic body it rides on, (out_pos, out_quat)[ob] = its fixed
  // transform in that body's frame (identity for a dynamic body); body_out[b] = the given model's index of dynamic body b
  const int *out_dyn, *body_out;
  const vreal *out_pos, *out_quat;
  // joints
  const int *jnt_type, *jnt_qposadr, *jnt_dofadr, *jnt_body;
  const vreal *jnt_pos, *jnt_axis, *jnt_qpos0, *jnt_stiffness, *jnt_springref;
  // limit rows (one per limited hinge)
  const int *lim_qadr, *lim_dof;
  const vreal *lim_lo, *lim_hi, *lim_margin, *lim_invweight, *lim_solref, *lim_solimp;
  // dofs; tree-sparse qM layout (MuJoCo dof_Madr order: self, parent, grandparent, ...)
  const int *dof_body, *dof_Madr, *dof_depth, *dof_limrow;
  const int *M_anc, *M_row;          /* per entry: column dof / row dof */
  const int *dof_ftime, *dof_fslot;  /* factorisation schedule: step in which row a is the pivot; scratch line | one leaf under a << 8 | mask of all leaves under a << 16 */
  const int* fac_guest;              /* [64] row 64.. that lane l inverts after its own row (EnvWave::invert_aba; -1: none); null if
                                        the model has no such rows or they cannot be placed (host depth <= 12, guest depth <= 24) */
  const unsigned* blk_tab;           /* [trips_row x 64 | trips_col x 64] block descriptors of blk_apply (vnl_lib.hip builds them); null if blk_cfg == 0 */
  const unsigned char* fac_match;    /* [nv][fac_steps]: bit k set = row a absorbs the pivot published in scratch line k in that step */
  const int *dof_ndesc;              /* descendants of dof a are dofs a+1 .. a+ndesc[a] (DFS numbering) */
  const unsigned char* lvl_tab;      /* [nv] dofs sorted by depth, then [max_depth+2] level starts */
  const vreal *dof_armature, *dof_damping;
  // actuators
  const int *act_dof, *act_limited;
  const int* dof_act; /* [nv] the actuators aimed at each dof: (index + 1) per byte, ascending, 0 = none; null if some dof has more than 4 */
  const vreal *act_gain, *act_tau, *act_lo, *act_hi, *act_gear;
  // collidable geoms (all against the one plane); contact c belongs to geom con_geom[c]
  const int *cg_type, *cg_body, *cg_conadr, *cg_ncon, *con_geom;
  const vreal *cg_pos, *cg_quat, *cg_size, *cg_mu, *cg_solref, *cg_solimp, *cg_margin, *cg_invweight;
};

struct DevEnv {
  int T, C, ref_len, sub_clip_length, n_frames, nb, nee, napp, njc, com_ref_col;
  int obs_size, traj_size;
  int flags; /* VNL_ENV_* of include/vnl.h: which tracking env's glue (rodent / humanoid style) */
  vreal healthy_lo, healthy_hi, inv_term_threshold, body_err_mult, done_threshold;
  vreal w_reward[6]; /* weights of rcom, rvel, rtrunk, rquat, ract, rapp in the total (built-in or envspec.reward_weights) */
  const int *body_idxs, *end_eff_idx, *app_body, *app_ref_col, *joint_cols;
  const float *position, *quaternion, *joints, *body_positions, *velocity, *angular_velocity, *joints_velocity;
  const float* center_of_mass; /* (C,T,3) or null: reference for rcom (else body_positions[com_ref_col]) */
  vreal* fac2; /* library-owned scratch [num_envs][nM + nv]: the second inverse factor of a substep (EnvWave::invert_aba) */
};

// caller-owned buffers, row-major [env][feature] (see include/vnl.h vnl_state)
struct DevState {
  vreal *qpos, *qvel, *act, *warm, *xpos, *xquat, *com1, *qfrc_actuator;
  vreal *obs, *reward, *done, *metrics, *traj, *term_err;
  int *cur_frame, *sub_clip_frame, *clip_id;
};

// per-env LDS sections (offsets in vreal elements)
struct WsLayout {
  int qpos, qvel, act, ctrl, actdot, com;
  int cdof, LD, dinv;
  // One liveness-aliased pool:
  //   kinematics : two 7*nbody pose buffers            bias   : cinert 10nb | cacc/cfrc 6nb | cvel 6nb
  //   M build    : crb (in place of cinert)            solver : efc_D | Jaref | jv (3 nefc) | contact wrenches 6 ncon
  //   euler      : cinert/crb again
  int P;
  int efc_D, Jaref, jv; /* = P, P + nefc, P + 2 nefc */
  int smooth, qacc_smooth, qacc, Ma, grad, Mgrad, search, mv, qfrc_c, tmp, tmp2;
  int prof; /* diagnostic build (-DVNL_PROFILE) only: per-stage cycle sums kept by lane 0 */
  int con_r, con_t1;    /* 3 per contact / 3 per collidable geom */
  int tab_anc, tab_madr, tab_body, tab_jump, tab_lvl; /* 8/16-bit index tables staged in LDS */
  int act_list;         /* ncon bytes: contacts with D != 0, then their count (int) */
  int pair_room;        /* elements from LD to the part of the pool that stays live across the factorisation (cvel): (kept for the layout's stability; the articulated-body factorisation needs 12 nv elements of the pool below cvel) */
  int newt_M, newt_H, newt_J; /* Newton solver only: dense qM (nv^2), Hessian / its Cholesky factor (nv^2), dense efc_J (nefc x nv) */
  int total;
};

/* The integers the per-env LDS layout and the loop bounds of the kernels depend on. */
struct VnlDims {
  int nq, nv, nu, nbody, njnt, ncg, ncon, nlimit, nefc, nM;
  int iterations, ls_iterations, eulerdamp, root_free, max_depth, jump_rounds, fac_steps, fac_nleaf, solver_newton, blk_cfg, path_runs, nbody_out;
};

/* The LDS layout as a function of the dims: evaluated by the host at env creation and, for a model the kernels are
 * SPECIALISED for (VnlSpecRodent below), at compile time -- every offset then is an immediate of the LDS instructions. */
#ifndef VNL_NPROF_SLOTS
#ifdef VNL_PROFILE
#define VNL_NPROF_SLOTS (2 * (40 + 1))
#else
#define VNL_NPROF_SLOTS 0
#endif
#endif
constexpr int vnl_imax(int a, int b) { return a > b ? a : b; }
constexpr int vnl_words(long bytes) { return (int)((bytes + (long)sizeof(vreal) - 1) / (long)sizeof(vreal)); }
constexpr WsLayout vnl_make_layout(const VnlDims& d) {
  WsLayout L{};
  int o = 0;
  auto sec = [&o](int n) {
    int at = o;
    o += n;
    return at;
  };
  L.qpos = sec(d.nq), L.qvel = sec(d.nv), L.act = sec(d.nu), L.ctrl = sec(d.nu);
  L.actdot = sec(d.nu), L.com = sec(4);
  L.cdof = sec(6 * d.nv);
  L.LD = sec(vnl_imax(d.nM, 6 * (d.nv + 1) + 6 * (d.nbody + 1))); /* also holds the dof / body prefix sums of bias_forces */
  L.dinv = sec(d.nv);
  /* solve phase: efc_D | Jaref | jv, then the larger of the contact-wrench prefix sums and the dof prefix sums of jac_mul */
  int pool = vnl_imax(vnl_imax(14 * d.nbody, 22 * d.nbody), 3 * d.nefc + vnl_imax(6 * (d.ncon + 1), 6 * (d.nv + 1)));
  if (d.eulerdamp) pool = vnl_imax(pool, d.nM + d.nv); /* euler() brings the second factor of the substep back into the pool */
  L.P = sec(pool);
  L.efc_D = L.P, L.Jaref = L.P + d.nefc, L.jv = L.P + 2 * d.nefc;
  L.pair_room = (L.P + 16 * d.nbody) - L.LD; /* bias_forces keeps cvel at pool + 16 nbody until make_constraint has read it */
  L.smooth = sec(d.nv), L.qacc_smooth = sec(d.nv), L.qacc = sec(d.nv);
  L.Ma = sec(d.nv), L.grad = sec(d.nv), L.Mgrad = sec(d.nv), L.search = sec(d.nv);
  L.mv = sec(d.nv), L.qfrc_c = sec(d.nv), L.tmp = sec(d.nv), L.tmp2 = sec(d.nv);
  L.con_r = sec(3 * d.ncon), L.con_t1 = sec(3 * d.ncg);
  L.tab_anc = sec(vnl_words(d.nM)), L.tab_madr = sec(vnl_words(6 * (long)d.nv)); /* row start | row end | descendants, 16 bit each */
  L.tab_body = sec(vnl_words(3 * (long)d.nbody + 2 * (long)d.ncon));
  L.tab_jump = sec(vnl_words((long)(d.jump_rounds > 0 ? d.jump_rounds : 1) * d.nbody));
  L.tab_lvl = sec(vnl_words((long)d.nv + d.max_depth + 2));
  L.act_list = sec(vnl_words(4 * (long)((d.ncon + 3) / 4) + 8 + 2 * VNL_LIVE_MAX)); /* active contacts | their count | existing rows: count, list */
  L.newt_M = L.newt_H = L.newt_J = 0;
  if (d.solver_newton) {
    L.newt_M = sec(d.nv * d.nv), L.newt_H = sec(d.nv * d.nv);
    L.newt_J = sec(d.nefc * d.nv);
  }
  if (VNL_NPROF_SLOTS) {
    o = (o + 1) & ~1;
    L.prof = sec(VNL_NPROF_SLOTS);
  }
  L.total = (o + 3) & ~3;
  return L;
}

/* Compile-time model of a kernel specialisation.  `fixed == false`: every dimension and LDS offset is read from the constant
 * block at run time (any model the library accepts).  `fixed == true`: they are the constants below -- loop bounds fold,
 * LDS offsets become instruction immediates, the scalar loads and address arithmetic of the generic kernel disappear; the
 * library selects such a kernel only for an env whose dims and layout EQUAL the constants (vnl_env_create checks). */
struct VnlSpecGeneric {
  static constexpr bool fixed = false;
  static constexpr VnlDims D{};
  static constexpr WsLayout L{};
};
/* the reference's rodent (assets/rodent.xml as envs/rodent.py:39-63 compiles it), CG 6 / 6 */
struct VnlSpecRodent {
  static constexpr bool fixed = true;
  // (53 dynamic bodies: 13 of the model's 66 are welded to their parents and folded into them, vnl_lib.hip: fuse_welded_bodies)
  static constexpr VnlDims D{74, 73, 30, 53, 68, 32, 59, 67, 303, 1119, 6, 6, 1, 1, 35, 5, 36, 6 | (13 << 8), 0,
                             2 | (2 << 4) | (2 << 8) | (3 << 12), 2, 66};
  static constexpr WsLayout L = vnl_make_layout(D);
};

/* Everything the env kernels read that does not change between launches, in one device buffer.  The kernels read
 * it through the CONSTANT address space (scalar loads at the point of use) instead of taking ~350 SGPRs' worth of
 * by-value kernel arguments, which the compiler kept alive across the whole kernel by spilling them to VGPR lanes. */
struct KernelConsts {
  DevModel m;
  DevEnv ev;
  WsLayout L;
};

