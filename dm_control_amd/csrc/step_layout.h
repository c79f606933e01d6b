// step_layout.h -- LDS layout of the fused step kernel.
//
// One workgroup stages ONE copy of the model constant tables (identical for all
// environments of a batch) in LDS, followed by one private scratch region per
// environment instance handled by the workgroup.  Offsets are computed on the
// host once per (model, precision, caps) and passed to the kernel by value.
//
//   [ int tables | real tables | env0 real scratch | env0 int scratch | env1 ... ]
//
// Table / scratch lists are X-macros: X(name, count_expression).  Count
// expressions are evaluated on the host against `StepDims`.
#pragma once
#include <stdint.h>

// StepDims::jglobal of a model with nv dofs
#ifndef DMC_JGLOBAL_NV1
#define DMC_JGLOBAL_NV1 16      // (experiment knob: dofs from which level 1 applies)
#endif
#define DMC_JGLOBAL_LEVEL(nv) ((nv) > 48 ? 3 : ((nv) > 32 ? 2 : ((nv) > DMC_JGLOBAL_NV1 ? 1 : 0)))

struct StepDims {
  int nq, nv, nu, nbody, njnt, ngeom, nsite, nsensor, nsensordata, npair;
  int nlevel;    // number of tree depths >= 1
  int nchild;    // nbody - 1 (size of the child list)
  int nM;        // number of (dof i, ancestor dof j) pairs incl. diagonal
  int nconmax;   // contact cap per environment
  int njmax;     // constraint-row cap per environment
  int rk4;       // 1: RK4 integrator (extra stage buffers)
  int ntri;      // nv (nv + 1) / 2: lower-triangle entries of an nv x nv matrix
  int elliptic;  // 1: frictional contacts use elliptic cones (one row per contact-frame axis)
  int nfric;     // dofs with frictionloss > 0 (one Huber-cost row each)
  int ncyl;      // candidate pairs involving a cylinder other than plane-cylinder (a guard test may raise DMC_WARN_COLLISION)
  int ncylx;     // of those, sphere-cylinder / capsule-cylinder pairs: narrow phase restated (closest point of the solid cylinder)
  int ntendon, nwrap;  // fixed tendons (actuator transmissions, springs / dampers)
  int fluid;     // 1: option density / viscosity > 0 (inertia-box fluid forces in mj_passive)
  int nstv;      // number of subtreelinvel sensors (each is one masked reduction over the bodies)
  int nlimten;   // tendons with a length limit (fixed or site-to-site spatial)
  int nlimball;  // limited ball joints (one dense limit row each, emitted after the tendon limits)
  int neq;       // active equality constraints (tendon, joint: 1 row; connect: 3; weld: 6)
  int neqrow;    // their rows
  int nprm;      // distinct contact-parameter tuples (margin, gap, friction, solref, solimp) over the pairs
  int nslip;     // cap on the friction rows the noslip post-solver handles (0: model has noslip_iterations = 0)
  int na;        // activation states (actuators with integrator / filter dynamics)
  int nbox;      // candidate pairs sphere-box / capsule-box / box-box
  int nrf;       // rangefinder sensors (ray casts against every geom)
  int nell;      // candidate pairs involving an ellipsoid (iterative support-function narrow phase)
  int kmax;      // max over the candidate pairs of the dofs a contact Jacobian row can touch (chain symmetric difference)
  int njdense;   // cap on the constraint rows stored as dense Jacobian rows (equalities, tendon limits)
  int njcon;     // cap on the contact rows (stored compressed: kmax entries per row)
  int msparse;   // 1: qM holds the nM tree-sparse entries (nv > 16); 0: dense nv x nv (small models: no index arithmetic)
  int kwords;    // ints per contact holding its dof list as bytes: (kmax + 3) / 4
  int maxrow;    // most constraint rows a single contact can have (bound of the per-contact row loops)
  int coldlds;   // 1: the cold tables are small enough to be staged in LDS with the others
  int pgs;       // 1: option solver="PGS" (projected Gauss-Seidel on the dual; AR = J M^-1 J' + R lives in StepOpts::ns_A, nslip = njmax)
  int cg;        // 1: option solver="CG" (conjugate gradient on the same primal problem, preconditioned with M^-1)
  int jfull;     // 1 (nv <= 16): EVERY constraint row is stored as a dense row of nv entries in efc_Jd (row classes and
                 //   compression pay off for long chains; on a 9-dof model their index arithmetic cost 10 % of the step)
  int sitegl;    // 1 (nsite > 32): the real site tables (pos, quat, size) stay in global memory (StepOpts::g_mr) -- composed
                 //   scenes carry render-only sites (soccer: 120 hoarding boards of 136 sites) that only an output
                 //   pass over ALL sites ever reads; sensors touch a handful, one per lane
  int dfs;       // 1: bodies are numbered depth first (a subtree is the contiguous range [b, body_subend[b])): subtree sums in one pass
  int ntree;     // kinematic trees with at least one dof (M^-1 is block diagonal over them: noslip blocks of different trees are independent)
  int treemax;   // > 0: the dofs of every kinematic tree are one contiguous range and the largest tree has treemax <= nv / 2
                 //   dofs: M (always) and H = M + J'DJ (unless a constraint row moves two trees) are block diagonal over the
                 //   trees, and their factorisations / substitutions run the trees side by side (StepCore::split_*)
  int treeuni;   // treemax models: 1 when every tree has exactly treemax dofs (tree t = dofs [t treemax, (t + 1) treemax): known at compile time)
  int ntreetri;  // treemax models: entries of the lower triangles of the trees' diagonal blocks (the only entries of H a split solve has)
  int island;    // 1: the model can have more than one constraint island (two or more kinematic trees, no noslip pass, a
                 //   primal solver): scratch for the island partition (StepCore::find_islands)
  int nmocap;    // mocap bodies: static children of the world posed by mjData.mocap_pos / mocap_quat (StepOpts::mocap_*)
  int jglobal;   // what lives in the environment's global scratch / in global memory instead of LDS (DMC_JGLOBAL_LEVEL):
                 //   1 (nv > 16): the compressed contact rows (efc_Jc) and, for noslip models, the kept factor of M;
                 //   2 (nv > 32): also the sparse M, the contact frames and the cold real model tables;
                 //   3 (nv > 48): also xquat, geom_xmat, cinert and cdof_dot (written once per stage, read in a few places):
                 //   the 4.8 KB that let a FIFTH 62-dof environment share a CU's LDS.
                 // The residency per CU of these models is bounded by LDS and their step is long enough that a few L2
                 // round trips do not show: level 1 took humanoid 6 -> 8 and the 62-dof models 2 -> 3 environments per
                 // CU, level 2 the 62-dof models to 4; on the 27 / 30-dof models level 2 gains no residency and costs 1-2 %.
};

// Constraint Jacobian storage.  Rows come in MuJoCo's order (equality, dof friction, joint limit,
// tendon limit, contact) and in three classes:
//   dense      equality and tendon-limit rows: nv entries each, in efc_Jd;
//   simple     dof-friction and joint-limit rows: ONE nonzero (+-1 at a dof) -- never stored;
//   contact    a contact's rows touch only the dofs in the symmetric difference of its two bodies'
//              chains: per contact a 64-bit dof mask (con_mlo / con_mhi) and the same dofs as a byte list
//              (con_dofs, ascending), per row kmax entries in efc_Jc (entry k belongs to the k-th dof).
// imisc[IM_ROW_S0 / TL0 / C0] hold the first simple, first tendon-limit and first contact row.
//
// ---- model tables (ints) -----------------------------------------------------
#define STEP_MODEL_INT_TABLES(X)                                               \
  X(body_parentid, d.nbody) X(body_rootid, d.nbody) X(body_jntadr, d.nbody)    \
  X(body_jntnum, d.nbody) X(body_dofadr, d.nbody) X(body_dofnum, d.nbody)      \
  X(body_lastdof, d.nbody)     /* last dof on the path root->body, or -1 */    \
  X(body_subend, d.dfs ? d.nbody : 0)   /* 1 + last body of the subtree below a body (depth-first numbering) */ \
  X(body_mocapid, d.nmocap ? d.nbody : 0)   /* row of mocap_pos / mocap_quat, -1: not a mocap body */ \
  X(body_anc_lo, d.nstv ? d.nbody : 0) X(body_anc_hi, d.nstv ? d.nbody : 0) /* ancestor-or-self bodies */ \
  X(stv_sensor, d.nstv)        /* the subtreelinvel sensors */                 \
  X(level_adr, d.nlevel + 1) X(level_body, d.nchild)                           \
  X(child_adr, d.nbody + 1) X(child_list, d.nchild)   /* children, descending */ \
  X(jnt_type, d.njnt) X(jnt_qposadr, d.njnt) X(jnt_dofadr, d.njnt)             \
  X(jnt_bodyid, d.njnt) X(jnt_limited, d.njnt)                                 \
  X(dof_bodyid, d.nv) X(dof_jntid, d.nv) X(dof_parentid, d.nv)                 \
  X(dof_anc_lo, d.nv) X(dof_anc_hi, d.nv)  /* bitmask of ancestor dofs (incl. self) */ \
  X(dof_madr, d.nv + 1)        /* first entry of row i of the sparse M (entries: i, parent(i), ...) */ \
  X(dof_subend, d.nv)          /* 1 + last dof of the subtree below dof i */   \
  X(dof_tree0, d.treemax ? d.nv : 0) X(dof_tree1, d.treemax ? d.nv : 0)  /* first dof / 1 + last dof of the kinematic tree of dof i */ \
  X(tree_tri, d.ntreetri)      /* the in-tree entries (i >= j, same tree) as i | j << 16, tree after tree */ \
  X(tree_trim, d.ntreetri)     /* their place in the sparse M (index into qM's nM entries), -1: not an ancestor pair, M(i, j) = 0 */ \
  X(geom_type, d.ngeom) X(geom_bodyid, d.ngeom)                                \
  X(geom_invisible, d.nrf ? d.ngeom : 0)  /* rays skip geoms with alpha 0 */   \
  X(site_bodyid, d.nsite) X(site_type, d.nsite)                                \
  X(act_dof, d.nu) X(act_qpos, d.nu) X(act_flags, d.nu)                        \
  X(act_adr, d.na ? d.nu : 0)  /* activation index of a stateful actuator, -1 otherwise */ \
  X(sensor_type, d.nsensor) X(sensor_objid, d.nsensor) X(sensor_adr, d.nsensor) \
  X(sensor_stage, d.nsensor) X(sensor_objtype, d.nsensor)                      \
  X(fric_dof, d.nfric)                                                         \
  X(tendon_adr, d.ntendon) X(tendon_num, d.ntendon) X(wrap_dof, d.nwrap) X(wrap_qpos, d.nwrap) \
  X(wrap_site, d.nwrap)        /* site id of a spatial-tendon wrap, -1 for joint wraps */ \
  X(limten, d.nlimten)         /* the limited tendons */                       \
  X(limball, d.nlimball)       /* the limited ball joints */                   \
  X(eq_type, d.neq) X(eq_obj1, d.neq) X(eq_obj2, d.neq)   /* mjtEq, tendon / joint / body ids (-1: none) */ \
  X(eq_rowadr, d.neq)          /* first constraint row of each equality (equality rows come first) */

// ---- cold model tables (ints): stay in global memory (L2-resident), read coalesced ----
// once per step: the candidate pair list and the (i, j) list of the sparse mass matrix
#define STEP_MODEL_COLD_TABLES(X)                                              \
  X(mpair, d.nM)               /* i | j << 16 : entry p of the sparse M is M(i, j), j an ancestor dof of i */ \
  X(pair_geom, d.npair)        /* geom1 | geom2 << 16 */                       \
  X(pair_info, d.npair)        /* condim | contact-parameter tuple << 8 */

// ---- model tables (reals) ----------------------------------------------------
// Hot tables are staged in LDS by every workgroup.  The cold ones -- each read once per step, one element per lane, in
// a loop over bodies / joints / geoms / actuators -- are staged with them for the small models, and left in global
// memory (L2-resident, StepOpts::g_mr) for the large ones (StepDims::jglobal == 2), whose residency LDS bounds.
#define STEP_MODEL_HOT_REAL_TABLES(X)                                          \
  X(qpos0, d.nq) X(qpos_spring, d.nq)                                          \
  X(body_mass, d.nbody)                                                        \
  X(body_subtreemass, d.nbody) X(body_invsubtreemass, d.nbody)                 \
  X(body_invweight0, 2 * d.nbody)                                              \
  X(jnt_pos, 3 * d.njnt) X(jnt_axis, 3 * d.njnt)                               \
  X(dof_armature, d.nv) X(dof_damping, d.nv) X(dof_invweight0, d.nv)           \
  X(dof_frictionloss, d.nfric ? d.nv : 0) X(dof_solref, d.nfric ? 2 * d.nv : 0) \
  X(dof_solimp, d.nfric ? 5 * d.nv : 0)                                        \
  X(geom_size, 3 * d.ngeom)                                                    \
  X(geom_rbound, d.ngeom)                                                      \
  X(prm_margin, d.nprm) X(prm_gap, d.nprm) X(prm_friction, 3 * d.nprm)         \
  X(prm_solref, 2 * d.nprm) X(prm_solimp, 5 * d.nprm)  /* distinct contact-parameter tuples */ \
  X(wrap_prm, d.nwrap)                                                         \
  X(act_dynprm, d.na ? d.nu : 0)  /* time constant of filter dynamics */          \
  X(tendon_stiffness, d.ntendon) X(tendon_damping, d.ntendon) X(tendon_lengthspring, d.ntendon) \
  X(tendon_range, d.nlimten ? 2 * d.ntendon : 0) X(tendon_margin, d.nlimten ? d.ntendon : 0) \
  X(tendon_solref_lim, d.nlimten ? 2 * d.ntendon : 0) X(tendon_solimp_lim, d.nlimten ? 5 * d.ntendon : 0) \
  X(tendon_invweight0, (d.nlimten || d.neq) ? d.ntendon : 0)                   \
  X(eq_solref, 2 * d.neq) X(eq_solimp, 5 * d.neq)                             \
  X(eq_data, 13 * d.neq)  /* mjModel.eq_data (11) + reference coordinates: tendon length0, or the joints' qpos0 */
#define STEP_MODEL_COLD_REAL_TABLES(X)                                         \
  X(body_pos, 3 * d.nbody) X(body_quat, 4 * d.nbody) X(body_ipos, 3 * d.nbody) \
  X(body_iquat, 4 * d.nbody) X(body_inertia, 3 * d.nbody)                      \
  X(jnt_stiffness, d.njnt)                                                     \
  X(jnt_range, 2 * d.njnt) X(jnt_margin, d.njnt) X(jnt_solref, 2 * d.njnt)     \
  X(jnt_solimp, 5 * d.njnt)                                                    \
  X(geom_pos, 3 * d.ngeom) X(geom_quat, 4 * d.ngeom)                           \
  X(act_gear, d.nu) X(act_ctrlrange, 2 * d.nu) X(act_forcerange, 2 * d.nu)     \
  X(act_gainprm, 3 * d.nu) X(act_biasprm, 3 * d.nu)                             \
  X(act_actrange, d.na ? 2 * d.nu : 0)   /* clamp of the advanced activation (ACTF_ACTLIMITED) */
// site tables: hot (LDS) for ordinary models, behind the cold tables and left in global memory when StepDims::sitegl
#define STEP_MODEL_SITE_REAL_TABLES(X)                                         \
  X(site_pos, 3 * d.nsite) X(site_quat, 4 * d.nsite) X(site_size, 3 * d.nsite)
#define STEP_MODEL_REAL_TABLES(X) STEP_MODEL_HOT_REAL_TABLES(X) STEP_MODEL_COLD_REAL_TABLES(X) STEP_MODEL_SITE_REAL_TABLES(X)

// ---- per-environment scratch (reals) -------------------------------------------
// Persistent arrays (live across the whole substep) ...
#define STEP_SCRATCH_REAL(X)                                                   \
  X(qpos, d.nq) X(qvel, d.nv) X(ctrl, d.nu) X(qacc_warmstart, d.nv)            \
  X(act, d.na) X(act_dot, d.na)                                                \
  X(qfrc_applied, d.nv)                                                        \
  X(xpos, 3 * d.nbody) X(xquat, d.jglobal >= 3 ? 0 : 4 * d.nbody) X(xmat, 9 * d.nbody)              \
  X(xipos, 3 * d.nbody)                                                        \
  X(geom_xpos, 3 * d.ngeom) X(geom_xmat, d.jglobal >= 3 ? 0 : 9 * d.ngeom)                          \
  X(subtree_com, 3 * d.nbody)                                                  \
  X(cinert, d.jglobal >= 3 ? 0 : 10 * d.nbody) X(cdof, 6 * d.nv) X(cdof_dot, d.jglobal >= 3 ? 0 : 6 * d.nv)              \
  X(cvel, 6 * d.nbody)                                                         \
  X(qM, d.jglobal >= 2 ? 0 : (d.msparse ? d.nM : d.nv * d.nv))  /* sparse: entry p is M(i, j) of the (i, j) list (mpair); global scratch if jglobal */ \
  X(qLH, d.ntri)        /* Cholesky of M, later of H / M+hB: lower triangle packed by columns */ \
  X(qLM, (d.nslip && !d.jglobal) ? d.ntri : 0)   /* noslip models: the factor of M kept beside that of H (noslip needs M^-1 after the solve); global scratch if jglobal */ \
  X(qfrc_bias, d.nv) X(qfrc_passive, d.nv) X(qfrc_actuator, d.nv)              \
  X(qfrc_smooth, d.nv) X(qacc_smooth, d.nv) X(qacc, d.nv)                      \
  X(qfrc_constraint, d.nv) X(actuator_force, d.nu)                             \
  X(sensordata, d.nsensordata)                                                 \
  X(con_dist, d.nconmax) X(con_pos, 3 * d.nconmax) X(con_frame, d.jglobal >= 2 ? 0 : 9 * d.nconmax) \
  X(efc_Jd, d.njdense * d.nv) X(efc_Jc, (d.jglobal || d.jfull) ? 0 : d.njcon * d.kmax)                      \
  X(efc_D, d.njmax)     /* holds efc_margin until the row parameters are made */ \
  X(efc_aref, d.njmax)  /* holds efc_pos until the row parameters are made */    \
  X(efc_force, d.njmax)                                                        \
  X(rk_q0, d.rk4 * d.nq) X(rk_v0, d.rk4 * d.nv) X(rk_dq, d.rk4 * d.nv)         \
  X(rk_dv, d.rk4 * d.nv)                                                       \
  X(misc, 16)
// ... followed by ONE region shared by three overlays whose lifetimes do not
// intersect: position-stage temporaries, velocity-stage temporaries, solver.
#define STEP_SCRATCH_OVL_POS(X)                                                \
  X(ximat, 9 * d.nbody) X(xanchor, 3 * d.njnt) X(xaxis, 3 * d.njnt)            \
  X(crb, 10 * d.nbody) X(mbuf, 6 * d.nv) X(subtree_usum, 3 * d.nbody)
#define STEP_SCRATCH_OVL_VEL(X)                                                \
  X(cacc, 6 * d.nbody) X(cfrc, 6 * d.nbody) X(cfrc_ext, 6 * d.nbody)
#define STEP_SCRATCH_OVL_SOL(X)                                                \
  X(sv_Ma, d.nv) X(sv_Mv, d.nv) X(sv_grad, d.nv) X(sv_Mgrad, d.nv)             \
  X(sv_search, d.nv) X(efc_jar, d.njmax) X(efc_jv, d.njmax)                    \
  X(sv_gold, d.cg * d.nv) X(sv_Mgold, d.cg * d.nv)  /* CG: gradient and M^-1 gradient of the previous iterate */                    \
  /* M v through the tree (mul_M, sparse-M models): per-body spatial force of the acceleration field of v */ \
  X(sv_bf, d.msparse * 6 * d.nbody)                                            \
  /* elliptic cones: per-row coefficients of the middle-zone Hessian (newton_gradient) */ \
  X(efc_ca, d.elliptic * d.njmax) X(efc_cb, d.elliptic * d.njmax)              \
  X(efc_cg, d.elliptic * d.njmax)                                              \
  /* noslip: the running residual (A = J_F M^-1 J_F^T itself lives in global memory, StepOpts::ns_A) */ \
  X(ns_res, d.nslip)
#define STEP_SCRATCH_ALL_REAL(X) \
  STEP_SCRATCH_REAL(X) STEP_SCRATCH_OVL_POS(X) STEP_SCRATCH_OVL_VEL(X) STEP_SCRATCH_OVL_SOL(X)

#ifdef DMC_PROFILE
#define DMC_PROF_SLOTS 34
#else
#define DMC_PROF_SLOTS 0
#endif
// ---- per-environment scratch (ints) --------------------------------------------
#define STEP_SCRATCH_INT(X)                                                    \
  X(con_geom, d.nconmax)  /* geom1 | geom2 << 16 */                            \
  X(con_info, d.nconmax)  /* condim | contact-parameter tuple << 8 */          \
  X(con_efc, d.nconmax)                                                        \
  X(con_mlo, d.nconmax) X(con_mhi, d.nv > 32 ? d.nconmax : 0)  /* dof mask of the contact's Jacobian rows */ \
  X(con_dofs, d.jfull ? 0 : d.nconmax * d.kwords)  /* the mask's dofs in ascending order, one byte each */ \
  X(efc_tid, d.njmax)   /* (id << 3) | type */                                 \
  X(efc_active, d.njmax) /* active set the current factor of H was built for */ \
  X(isl_comp, d.island ? 2 * d.nv : 0)    /* islands: 64-bit mask (over root dofs = trees) of the component each tree is in */ \
  X(efc_tree, d.island ? 2 * d.njmax : 0) /* islands: 64-bit mask of the trees each constraint row moves */ \
  X(ns_row, d.nslip)     /* noslip: constraint row of each friction dimension */ \
  X(ns_blk, d.nslip ? 48 : 0)  /* noslip: per block (<= 16 blocks) start | size << 8 | level << 16 | coupled << 24, tree mask lo, hi */ \
  X(prof, DMC_PROF_SLOTS)   /* profiling builds: cycle counters per phase + the last time stamp */ \
  X(imisc, 16)

// indices into the `misc` / `imisc` scratch
enum { MISC_TIME = 0, MISC_EPOCH = 1 /* the stash epoch the launch started in (Entry::epoch), bit-cast: the tags it writes carry this one */ };
enum { IM_NCON = 0, IM_NEFC = 1, IM_ITER = 2, IM_WARN = 3 /* ..11: DMC_NWARNING counters */,
       IM_ROW_S0 = 12, IM_ROW_TL0 = 13, IM_ROW_C0 = 14 /* first simple / tendon-limit / contact row */,
       IM_ENV = 15 /* the environment's index in the batch (per-env model deltas) */ };

// act_flags bits
enum { ACTF_CTRLLIMITED = 1, ACTF_FORCELIMITED = 2, ACTF_GAIN_AFFINE = 4, ACTF_BIAS_AFFINE = 8,
       ACTF_TENDON = 16 /* act_dof holds a fixed-tendon id */,
       ACTF_DYN_INTEGRATOR = 32, ACTF_DYN_FILTER = 64, ACTF_DYN_FILTEREXACT = 128, ACTF_DYN_ANY = 32 | 64 | 128,
       ACTF_ACTLIMITED = 256 /* mj_nextActivation clamps the advanced activation to act_actrange */ };
// EFC_LIMIT rows carry id = (dof << 1) | upper_side
enum { EFC_LIMIT = 0, EFC_FRICTIONLESS = 1, EFC_PYRAMIDAL = 2, EFC_ELLIPTIC = 3, EFC_FRICTION = 4, EFC_TENDON_LIMIT = 5, EFC_EQUALITY = 6 };
enum { EFC_ST_SATISFIED = 0, EFC_ST_QUADRATIC = 1, EFC_ST_CONE = 2, EFC_ST_LINEARNEG = 3, EFC_ST_LINEARPOS = 4 };   /* efc_active values */
#define EFC_TID(type, id) (((id) << 3) | (type))
#define EFC_TYPE(tid) ((tid) & 7)
#define EFC_ID(tid) ((tid) >> 3)

struct StepLayout {
  StepDims d;
  // offsets in elements (ints for int tables, reals for real tables/scratch)
#define X(name, cnt) int mi_##name;
  STEP_MODEL_INT_TABLES(X)
#undef X
#define X(name, cnt) int mr_##name;
  STEP_MODEL_REAL_TABLES(X)
#undef X
#define X(name, cnt) int mc_##name;
  STEP_MODEL_COLD_TABLES(X)
#undef X
#define X(name, cnt) int s_##name;
  STEP_SCRATCH_ALL_REAL(X)
#undef X
#define X(name, cnt) int si_##name;
  STEP_SCRATCH_INT(X)
#undef X
  int n_mi, n_mr, n_mc;    // table sizes (elements)
  int n_mr_lds;            // reals staged in LDS: n_mr, or only the hot tables (d.jglobal)
  int n_sr, n_si;          // per-env scratch sizes (elements)
  int n_keep;              // reals before the overlay region: what survives a stage (the per-env stash in HBM)
  int n_gs, gs_Jc, gs_LM, gs_M, gs_cf;  // per-env global scratch (reals; 0 unless d.jglobal): size, offsets of efc_Jc, the factor of M, sparse M, con_frame
  int gs_xquat, gs_geom_xmat, gs_cinert, gs_cdof_dot;      // level 3: the kinematic arrays that left LDS
};

// scalar options broadcast to every wave
template <typename T>
struct StepOpts {
  T timestep, gravity[3], impratio, tolerance, ls_tolerance, meaninertia, density, viscosity, noslip_tolerance;
  int integrator, cone, iterations, ls_iterations, disableflags, noslip_iterations;
  int any_damping;   // some dof_damping > 0 (Euler implicit-damping path)
  int islands;       // per-island solves (mj_island semantics): 1 on, 0 off (one joint solve: the same minimiser), -1 = by
                     // precision (fp64 on: tracks the CPU reference; fp32 off: throughput)
  double timestep_d; // fp64 copy for the time accumulator
  // per-environment model deltas: world-fixed geoms whose pose / size differ between environments (soccer pitch
  // randomisation, per-env targets).  eg_data: (16 eg_n, eg_B) SoA in batch precision, rows of geom slot k:
  // pos(3) mat(9) size(3) rbound(1); eg_slot: (ngeom) slot of a geom or -1.  Both in global memory.
  const void* eg_data; const int* eg_slot; int eg_n, eg_B;
  // noslip: A = J_F M^-1 J_F^T of every environment, (B, nslip, nslip) full symmetric storage in global memory
  // (L2-resident for the environments in flight; 8 .. 36 KB per environment is LDS the solver needs elsewhere)
  void* ns_A;
  // mjData.xfrc_applied: Cartesian [force(3), torque(3)] per body at its COM, (6 nbody, B) SoA in global memory; null
  // until the caller touches the field (almost every batch): read only where it enters (mj_fwdAcceleration, cfrc_ext)
  const void* xfrc; int xfrc_B;
  // mjData.mocap_pos / mocap_quat: (3 nmocap, B) / (4 nmocap, B) SoA in batch precision, global memory (models with
  // mocap bodies only): the pose mj_kinematics gives a mocap body
  const void* mocap_pos; const void* mocap_quat; int mocap_B;
  // per-environment global scratch, (B, n_gs) reals (StepLayout::gs_*): arrays of the large models that LDS has no room for
  void* gscr;
  // the real model tables in global memory (batch precision): where the large models read the cold ones from
  const void* g_mr;
};

static inline void step_layout_build(StepLayout* L, const StepDims& d) {
  L->d = d;
  int o = 0;
#define X(name, cnt) L->mi_##name = o; o += (cnt);
  STEP_MODEL_INT_TABLES(X)
#undef X
  L->n_mi = (o + 3) & ~3;
  o = 0;
#define X(name, cnt) L->mr_##name = o; o += (cnt);
  STEP_MODEL_HOT_REAL_TABLES(X)
  if (!d.sitegl) { STEP_MODEL_SITE_REAL_TABLES(X) }
  o = (o + 3) & ~3;
  L->n_mr_lds = o;
  STEP_MODEL_COLD_REAL_TABLES(X)
  o = (o + 3) & ~3;
  if (d.jglobal < 2) L->n_mr_lds = o;
  if (d.sitegl) { STEP_MODEL_SITE_REAL_TABLES(X) }
#undef X
  L->n_mr = (o + 3) & ~3;
  o = 0;
#define X(name, cnt) L->mc_##name = o; o += (cnt);
  STEP_MODEL_COLD_TABLES(X)
#undef X
  L->n_mc = (o + 3) & ~3;
  o = 0;
#define X(name, cnt) L->s_##name = o; o += (cnt);
  STEP_SCRATCH_REAL(X)
#undef X
  if (!d.nslip || d.jglobal) L->s_qLM = L->s_qLH;   // no noslip: M's factor is not needed after H's took the buffer -- one buffer
  L->n_gs = L->gs_Jc = L->gs_LM = L->gs_M = L->gs_cf = 0;
  L->gs_xquat = L->gs_geom_xmat = L->gs_cinert = L->gs_cdof_dot = 0;
  if (d.jglobal) {
    int g = (d.njcon * d.kmax + 31) & ~31;     // 128-byte granules: an environment's arrays never share a cache line
    L->gs_LM = g;
    if (d.nslip) g += (d.ntri + 31) & ~31;
    L->gs_M = g;
    if (d.jglobal >= 2) g += (d.nM + 31) & ~31;
    L->gs_cf = g;
    if (d.jglobal >= 2) g += (9 * d.nconmax + 31) & ~31;
    if (d.jglobal >= 3) {
      L->gs_xquat = g; g += (4 * d.nbody + 31) & ~31;
      L->gs_geom_xmat = g; g += (9 * d.ngeom + 31) & ~31;
      L->gs_cinert = g; g += (10 * d.nbody + 31) & ~31;
      L->gs_cdof_dot = g; g += (6 * d.nv + 31) & ~31;
    }
    L->n_gs = g;
  }
  o = (o + 3) & ~3;
  L->n_keep = o;
  {
    const int base = o;
    int top = base;
    o = base;
#define X(name, cnt) L->s_##name = o; o += (cnt);
    STEP_SCRATCH_OVL_POS(X)
    if (o > top) top = o;
    o = base;
    STEP_SCRATCH_OVL_VEL(X)
    if (o > top) top = o;
    o = base;
    STEP_SCRATCH_OVL_SOL(X)
    if (o > top) top = o;
#undef X
    o = top;
  }
  L->n_sr = (o + 3) & ~3;
  o = 0;
#define X(name, cnt) L->si_##name = o; o += (cnt);
  STEP_SCRATCH_INT(X)
#undef X
  L->n_si = (o + 3) & ~3;
}

// name -> (offset, count) of a per-env scratch array; kind: 0 real, 1 int.  Used
// by the debug dump (tests compare kernel intermediates with the oracle).
#include <string.h>
static inline int step_layout_find(const StepLayout* L, const char* name, int* off, int* cnt, int* kind) {
  const StepDims& d = L->d;
  // the debug dump appends the environment's global scratch after its n_sr LDS reals
  if (d.jglobal && !strcmp(name, "efc_Jc")) { *off = L->n_sr + L->gs_Jc; *cnt = d.njcon * d.kmax; *kind = 0; return 1; }
  if (d.jglobal >= 2 && !strcmp(name, "qM")) { *off = L->n_sr + L->gs_M; *cnt = d.nM; *kind = 0; return 1; }
  if (d.jglobal >= 2 && !strcmp(name, "con_frame")) { *off = L->n_sr + L->gs_cf; *cnt = 9 * d.nconmax; *kind = 0; return 1; }
  if (d.jglobal >= 3 && !strcmp(name, "xquat")) { *off = L->n_sr + L->gs_xquat; *cnt = 4 * d.nbody; *kind = 0; return 1; }
  if (d.jglobal >= 3 && !strcmp(name, "geom_xmat")) { *off = L->n_sr + L->gs_geom_xmat; *cnt = 9 * d.ngeom; *kind = 0; return 1; }
  if (d.jglobal >= 3 && !strcmp(name, "cinert")) { *off = L->n_sr + L->gs_cinert; *cnt = 10 * d.nbody; *kind = 0; return 1; }
  if (d.jglobal >= 3 && !strcmp(name, "cdof_dot")) { *off = L->n_sr + L->gs_cdof_dot; *cnt = 6 * d.nv; *kind = 0; return 1; }
#define X(n, c) if (!strcmp(name, #n)) { *off = L->s_##n; *cnt = (c); *kind = 0; return 1; }
  STEP_SCRATCH_ALL_REAL(X)
#undef X
#define X(n, c) if (!strcmp(name, #n)) { *off = L->si_##n; *cnt = (c); *kind = 1; return 1; }
  STEP_SCRATCH_INT(X)
#undef X
  return 0;
}
