// step_tables.h -- host-side: model blob -> kernel constant tables + LDS layout.
//
// Pure C++ (no HIP calls) so that tests/emu can reuse it.  Everything here runs
// once per (model, caps); the per-step path never touches it.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <array>
#include <string>
#include <vector>

#include "../../include/dmc_model_layout.h"
#include "step_layout.h"

namespace dmc {

struct HostModel {
#define X(n) int n;
  DMC_MODEL_HEADER_INTS(X)
#undef X
#define X(n) double n;
  DMC_MODEL_HEADER_REALS(X)
#undef X
#define X(n, c) std::vector<int> n;
  DMC_MODEL_INT_FIELDS(X)
#undef X
#define X(n, c) std::vector<double> n;
  DMC_MODEL_REAL_FIELDS(X)
#undef X
  std::vector<int32_t> blob_i;
  std::vector<double> blob_r;
};

inline bool host_model_parse(HostModel* m, const int32_t* ints, int nints, const double* reals, int nreals, std::string* err) {
  if (nints < 2 || ints[0] != (int32_t)DMC_MODEL_MAGIC || ints[1] != DMC_MODEL_VERSION) { *err = "bad model blob magic/version"; return false; }
  long ip = 2, rp = 0;
  auto need_i = [&](long n) { return ip + n <= nints; };
  auto need_r = [&](long n) { return rp + n <= nreals; };
#define X(n) if (!need_i(1)) { *err = "model blob truncated"; return false; } m->n = ints[ip++];
  DMC_MODEL_HEADER_INTS(X)
#undef X
#define X(n) if (!need_r(1)) { *err = "model blob truncated"; return false; } m->n = reals[rp++];
  DMC_MODEL_HEADER_REALS(X)
#undef X
  const int nq = m->nq, nv = m->nv, nu = m->nu, nbody = m->nbody, njnt = m->njnt, ngeom = m->ngeom;
  const int nsite = m->nsite, nsensor = m->nsensor, npair = m->npair, nkey = m->nkey, ntendon = m->ntendon, nwrap = m->nwrap, neq = m->neq, na = m->na, nmocap = m->nmocap;
  (void)na; (void)nmocap; (void)ntendon; (void)nwrap; (void)neq; (void)nq; (void)nv; (void)nu; (void)nbody; (void)njnt; (void)ngeom; (void)nsite; (void)nsensor; (void)npair; (void)nkey;
#define X(n, c) { long cnt = (c); if (cnt < 0 || !need_i(cnt)) { *err = "model blob truncated"; return false; } m->n.assign(ints + ip, ints + ip + cnt); ip += cnt; }
  DMC_MODEL_INT_FIELDS(X)
#undef X
#define X(n, c) { long cnt = (c); if (cnt < 0 || !need_r(cnt)) { *err = "model blob truncated"; return false; } m->n.assign(reals + rp, reals + rp + cnt); rp += cnt; }
  DMC_MODEL_REAL_FIELDS(X)
#undef X
  if (ip != nints || rp != nreals) { *err = "model blob size mismatch"; return false; }
  m->blob_i.assign(ints, ints + nints);
  m->blob_r.assign(reals, reals + nreals);
  return true;
}

struct StepTables {
  StepLayout L;
  std::vector<int> mi;       // int tables, laid out per L.mi_*
  std::vector<int> mc;       // cold int tables (global memory), per L.mc_*
  std::vector<double> mr;    // real tables (fp64 master copy), per L.mr_*
  StepOpts<double> opts;
  int max_contacts, max_rows;  // upper bounds if no cap applied
  int has_unsupported_pairs = 0;  // pair types outside the collision kernel (legal only with contacts disabled)
};

// returns false + err for models the kernel does not support
// jlevel: StepDims::jglobal (what leaves LDS for the per-env global scratch), -1 = DMC_JGLOBAL_LEVEL(nv)
inline bool step_tables_build(StepTables* t, const HostModel& m, int nconmax, int njmax, std::string* err, int njcon = 0, int jlevel = -1) {
  StepDims d;
  std::memset(&d, 0, sizeof d);
  d.nq = m.nq; d.nv = m.nv; d.nu = m.nu; d.nbody = m.nbody; d.njnt = m.njnt; d.ngeom = m.ngeom;
  d.nsite = m.nsite; d.nsensor = m.nsensor; d.nsensordata = m.nsensordata; d.npair = m.npair;
  for (int i = 0; i < m.nu; i++) {
    if (m.actuator_dyntype[i] < DMC_DYN_NONE || m.actuator_dyntype[i] > DMC_DYN_FILTEREXACT) { *err = "actuator dyntype not implemented (none / integrator / filter / filterexact)"; return false; }
    if (m.actuator_dyntype[i] != DMC_DYN_NONE) d.na++;
  }
  if (d.na != m.na) { *err = "model na does not match the number of actuators with dynamics"; return false; }
  if (d.na && m.opt_integrator != DMC_INT_EULER && m.opt_integrator != DMC_INT_IMPLICITFAST) { *err = "actuator dynamics are only implemented with the Euler and implicitfast integrators"; return false; }
  if (m.nv > 64) { *err = "kernel supports nv <= 64"; return false; }
  if (m.ngeom > 65535) { *err = "kernel supports ngeom <= 65535"; return false; }
  const bool elliptic = m.opt_cone == DMC_CONE_ELLIPTIC;
  if (m.opt_integrator != DMC_INT_EULER && m.opt_integrator != DMC_INT_RK4 && m.opt_integrator != DMC_INT_IMPLICITFAST) { *err = "only the Euler, RK4 and implicitfast integrators are implemented in the HIP path"; return false; }
  if (m.opt_integrator == DMC_INT_IMPLICITFAST) {
    // the integration matrix M - h dF/dv is built with the DIAGONAL velocity derivatives only (StepCore::euler_state)
    if (m.opt_density > 0 || m.opt_viscosity > 0) { *err = "implicitfast with fluid forces is not implemented"; return false; }
    for (int t = 0; t < m.ntendon; t++) if (m.tendon_damping[t] > 0) { *err = "implicitfast with damped tendons is not implemented"; return false; }
    for (int i = 0; i < m.nu; i++) {
      const bool vel = (m.actuator_biastype[i] == DMC_BIAS_AFFINE && m.actuator_biasprm[10*i + 2] != 0) ||
                       (m.actuator_gaintype[i] == DMC_GAIN_AFFINE && m.actuator_gainprm[10*i + 2] != 0);
      if (vel && m.actuator_trntype[i] != DMC_TRN_JOINT) { *err = "implicitfast: velocity-dependent actuator on a tendon transmission is not implemented"; return false; }
    }
  }
  d.rk4 = m.opt_integrator == DMC_INT_RK4 ? 1 : 0;
  std::vector<int> fric_dof;
  for (int i = 0; i < m.nv; i++) {
    if (m.dof_frictionloss[i] < 0) { *err = "negative dof frictionloss"; return false; }
    if (m.dof_frictionloss[i] > 0) fric_dof.push_back(i);
  }
  d.nfric = (int)fric_dof.size();
  d.ntendon = m.ntendon; d.nwrap = m.nwrap;
  std::vector<int> stv;
  for (int i = 0; i < m.nsensor; i++) if (m.sensor_type[i] == DMC_SENS_SUBTREELINVEL) stv.push_back(i);
  d.nstv = (int)stv.size();
  d.nmocap = m.nmocap;
  for (int i = 0; i < m.nsensor; i++) if (m.sensor_type[i] == DMC_SENS_RANGEFINDER) d.nrf++;
  if (d.nstv && m.nbody > 64) { *err = "subtreelinvel sensors need nbody <= 64"; return false; }
  d.fluid = (m.opt_density > 0 || m.opt_viscosity > 0) ? 1 : 0;
  for (int w = 0; w < m.nwrap; w++) {
    const int j = m.wrap_objid[w];
    if (m.wrap_type[w] == DMC_WRAP_SITE) { if (j < 0 || j >= m.nsite) { *err = "spatial tendon refers to a missing site"; return false; } continue; }
    if (m.wrap_type[w] != DMC_WRAP_JOINT || j < 0 || j >= m.njnt || (m.jnt_type[j] != DMC_JNT_HINGE && m.jnt_type[j] != DMC_JNT_SLIDE)) { *err = "fixed tendons may only wrap hinge/slide joints"; return false; }
  }
  std::vector<int> limten;
  for (int t = 0; t < m.ntendon; t++) {
    const bool spatial = m.tendon_num[t] > 0 && m.wrap_type[m.tendon_adr[t]] == DMC_WRAP_SITE;
    if (spatial && (m.tendon_stiffness[t] != 0 || m.tendon_damping[t] != 0)) { *err = "spatial tendon springs / dampers are not implemented"; return false; }
    if (m.tendon_limited[t]) limten.push_back(t);
  }
  d.nlimten = (int)limten.size();
  std::vector<int> eq_src, eq_rowadr;
  for (int k = 0; k < m.neq; k++) {
    const int et = m.eq_type[k], o1 = m.eq_obj1id[k], o2 = m.eq_obj2id[k];
    if (et == DMC_EQ_TENDON) {
      if (o1 < 0 || o1 >= m.ntendon || (m.tendon_num[o1] > 0 && m.wrap_type[m.tendon_adr[o1]] != DMC_WRAP_JOINT)) { *err = "tendon equality constraints need a fixed tendon"; return false; }
    } else if (et == DMC_EQ_JOINT) {
      if (o1 < 0 || o1 >= m.njnt || o2 >= m.njnt) { *err = "joint equality refers to a missing joint"; return false; }
    } else if (et == DMC_EQ_CONNECT || et == DMC_EQ_WELD) {
      if (o1 < 0 || o1 >= m.nbody || o2 < 0 || o2 >= m.nbody) { *err = "connect / weld equality refers to a missing body"; return false; }
    } else { *err = "equality constraint type not implemented (connect, weld, joint, tendon)"; return false; }
    if (!m.eq_active0[k]) continue;
    eq_src.push_back(k); eq_rowadr.push_back(d.neqrow);
    d.neqrow += et == DMC_EQ_CONNECT ? 3 : (et == DMC_EQ_WELD ? 6 : 1);
  }
  d.neq = (int)eq_src.size();
  std::vector<int> limball;
  for (int j = 0; j < m.njnt; j++) {
    if (m.jnt_type[j] == DMC_JNT_BALL && m.jnt_limited[j]) limball.push_back(j);
    if ((m.jnt_type[j] == DMC_JNT_BALL || m.jnt_type[j] == DMC_JNT_FREE) && m.jnt_stiffness[j] != 0) { *err = "free/ball joint springs are not implemented"; return false; }
  }
  d.nlimball = (int)limball.size();
  // tree levels
  std::vector<int> depth(m.nbody, 0);
  int nlevel = 0;
  for (int b = 1; b < m.nbody; b++) { depth[b] = depth[m.body_parentid[b]] + 1; nlevel = std::max(nlevel, depth[b]); }
  d.nlevel = nlevel; d.nchild = std::max(0, m.nbody - 1);
  std::vector<int> subend(m.nbody, 0);
  {      // depth-first numbering: every subtree is a contiguous range of body ids
    std::vector<int> cnt(m.nbody, 1);
    for (int b = 0; b < m.nbody; b++) subend[b] = b + 1;
    for (int b = m.nbody - 1; b >= 1; b--) { const int p = m.body_parentid[b]; cnt[p] += cnt[b]; subend[p] = std::max(subend[p], subend[b]); }
    d.dfs = 1;
    for (int b = 0; b < m.nbody; b++) if (subend[b] - b != cnt[b]) d.dfs = 0;
  }
  { std::vector<int> seen(m.nbody, 0); d.ntree = 0; for (int i = 0; i < m.nv; i++) { const int r = m.body_rootid[m.dof_bodyid[i]]; if (!seen[r]) { seen[r] = 1; d.ntree++; } } }
  {      // trees as contiguous dof ranges (bodies are numbered so that a tree's dofs follow each other; checked, not assumed)
    d.treemax = 0;
    int big = 0, run = 0; bool contiguous = true;
    std::vector<int> closed(m.nbody, 0);
    for (int i = 0; i < m.nv; i++) {
      const int r = m.body_rootid[m.dof_bodyid[i]];
      if (i > 0 && r != m.body_rootid[m.dof_bodyid[i - 1]]) { closed[m.body_rootid[m.dof_bodyid[i - 1]]] = 1; run = 0; }
      if (closed[r]) contiguous = false;
      run++; big = std::max(big, run);
    }
    if (d.ntree >= 2 && contiguous && m.nv > 16 && m.nv <= 64 && 2 * big <= m.nv) d.treemax = big;      // (nv <= 16: dense M, register Hessians)
    d.treeuni = d.treemax && d.ntree * d.treemax == m.nv;      // (the largest has treemax, the sum is nv: all equal)
    d.ntreetri = 0;
    if (d.treemax) for (int i = 0; i < m.nv; i++) for (int j = 0; j <= i; j++) if (m.body_rootid[m.dof_bodyid[i]] == m.body_rootid[m.dof_bodyid[j]]) d.ntreetri++;
  }
  // M sparsity
  std::vector<int> mp_i, mp_j;
  for (int i = 0; i < m.nv; i++) for (int j = i; j >= 0; j = m.dof_parentid[j]) { mp_i.push_back(i); mp_j.push_back(j); }
  d.nM = (int)mp_i.size();
  d.ntri = m.nv * (m.nv + 1) / 2;
  // contact / row caps
  int maxc = 0, maxr = 0, nlim = 0;
  for (int j = 0; j < m.njnt; j++) if (m.jnt_limited[j]) nlim++;
  std::vector<int> pdim(m.npair);
  for (int p = 0; p < m.npair; p++) {
    const int g1 = m.pair_geom1[p], g2 = m.pair_geom2[p];
    int t1 = m.geom_type[g1], t2 = m.geom_type[g2];
    // cylinders: against a plane, a sphere or a capsule a real narrow phase; any other pair is guard-tested as the enclosing
    // capsule (DMC_WARN_COLLISION) and never yields a contact
    const bool plane_cyl = t1 == DMC_GEOM_PLANE && t2 == DMC_GEOM_CYLINDER;   // analytic, up to 4 contacts
    const bool cyl = (t1 == DMC_GEOM_CYLINDER || t2 == DMC_GEOM_CYLINDER) && !plane_cyl;
    if (t1 == DMC_GEOM_CYLINDER) t1 = DMC_GEOM_CAPSULE;
    if (t2 == DMC_GEOM_CYLINDER) t2 = DMC_GEOM_CAPSULE;
    if (cyl) d.ncyl++;
    // a sphere or a capsule against a cylinder has a real narrow phase (one contact); the other cylinder pairs stay guarded
    const bool cylx = cyl && m.geom_type[g2] == DMC_GEOM_CYLINDER && (m.geom_type[g1] == DMC_GEOM_SPHERE || m.geom_type[g1] == DMC_GEOM_CAPSULE);
    if (cylx) d.ncylx++;
    const bool ell = t2 == DMC_GEOM_ELLIPSOID && (t1 == DMC_GEOM_PLANE || t1 == DMC_GEOM_SPHERE || t1 == DMC_GEOM_CAPSULE || t1 == DMC_GEOM_ELLIPSOID) && !cyl;
    if (ell) d.nell++;
    // (a cylinder against a box is guard-tested as its enclosing capsule like every other cylinder pair)
    const bool boxp = t2 == DMC_GEOM_BOX && (t1 == DMC_GEOM_SPHERE || t1 == DMC_GEOM_CAPSULE || t1 == DMC_GEOM_BOX);
    int nc = 1;
    if (boxp) { d.nbox++; if (t1 == DMC_GEOM_CAPSULE) nc = 2; else if (t1 == DMC_GEOM_BOX) nc = 4; }
    if (t1 == DMC_GEOM_PLANE && t2 == DMC_GEOM_CAPSULE) nc = 2;
    else if (t1 == DMC_GEOM_PLANE && t2 == DMC_GEOM_BOX) nc = 4;
    if (plane_cyl) nc = 4;
    else if (t1 == DMC_GEOM_CAPSULE && t2 == DMC_GEOM_CAPSULE) nc = 1;   // 2 only for exactly parallel axes
    const bool known = (t1 == DMC_GEOM_PLANE && (t2 == DMC_GEOM_SPHERE || t2 == DMC_GEOM_CAPSULE || t2 == DMC_GEOM_BOX)) ||
                       (t1 == DMC_GEOM_SPHERE && (t2 == DMC_GEOM_SPHERE || t2 == DMC_GEOM_CAPSULE)) ||
                       (t1 == DMC_GEOM_CAPSULE && t2 == DMC_GEOM_CAPSULE) || ell || boxp;
    if (!known) {
      // tolerated only while contacts are disabled (e.g. suite cartpole): the pair then never collides
      if (m.opt_disableflags & (DMC_DSBL_CONTACT | DMC_DSBL_CONSTRAINT)) { t->has_unsupported_pairs = 1; nc = 0; }
      else { *err = "geom pair type not implemented in the HIP collision kernel (mesh / hfield, cylinder or ellipsoid against a box)"; return false; }
    }
    int dim;
    const int pr1 = m.geom_priority[g1], pr2 = m.geom_priority[g2];
    if (pr1 == pr2) dim = std::max(m.geom_condim[g1], m.geom_condim[g2]);
    else dim = m.geom_condim[pr1 > pr2 ? g1 : g2];
    pdim[p] = dim;
    if (cyl && !cylx) nc = 0;
    maxc += nc; maxr += nc * (dim == 1 ? 1 : (elliptic ? dim : 2*(dim - 1)));
  }
  // contact parameters (mixing rules: max / priority / solmix).  Most pairs of a
  // model share one parameter tuple (all geoms at their defaults), so the tuples are stored once and the
  // pairs index them: humanoid_CMU has 1118 candidate pairs but 2 distinct tuples
  std::vector<std::array<double, 12> > prm;
  std::vector<int> pair_prm(m.npair);
  for (int p = 0; p < m.npair; p++) {
    const int g1 = m.pair_geom1[p], g2 = m.pair_geom2[p];
    const int pr1 = m.geom_priority[g1], pr2 = m.geom_priority[g2];
    std::array<double, 12> v;
    v[0] = std::max(m.geom_margin[g1], m.geom_margin[g2]);
    v[1] = std::max(m.geom_gap[g1], m.geom_gap[g2]);
    double fr[3];
    if (pr1 == pr2) for (int k = 0; k < 3; k++) fr[k] = std::max(m.geom_friction[3*g1 + k], m.geom_friction[3*g2 + k]);
    else { const int gp = pr1 > pr2 ? g1 : g2; for (int k = 0; k < 3; k++) fr[k] = m.geom_friction[3*gp + k]; }
    for (int k = 0; k < 3; k++) v[2 + k] = std::max((double)DMC_MINMU, fr[k]);
    double mix;
    if (pr1 != pr2) mix = pr1 > pr2 ? 1 : 0;
    else {
      const double sm1 = m.geom_solmix[g1], sm2 = m.geom_solmix[g2];
      if (sm1 >= DMC_MINVAL && sm2 >= DMC_MINVAL) mix = sm1 / (sm1 + sm2);
      else if (sm1 < DMC_MINVAL && sm2 < DMC_MINVAL) mix = 0.5;
      else mix = sm1 < DMC_MINVAL ? 0.0 : 1.0;
    }
    const double *r1 = &m.geom_solref[2*g1], *r2 = &m.geom_solref[2*g2];
    for (int k = 0; k < 2; k++) v[5 + k] = (r1[0] > 0 && r2[0] > 0) ? mix*r1[k] + (1 - mix)*r2[k] : std::min(r1[k], r2[k]);
    for (int k = 0; k < 5; k++) v[7 + k] = mix*m.geom_solimp[5*g1 + k] + (1 - mix)*m.geom_solimp[5*g2 + k];
    int q = 0;
    while (q < (int)prm.size() && std::memcmp(prm[q].data(), v.data(), sizeof(double)*12)) q++;
    if (q == (int)prm.size()) prm.push_back(v);
    pair_prm[p] = q;
  }
  d.nprm = (int)prm.size();
  t->max_contacts = maxc; t->max_rows = maxr + nlim + d.nlimten + d.nfric + d.neqrow;
  int maxrow_per_contact = 1;
  for (int p = 0; p < m.npair; p++) maxrow_per_contact = std::max(maxrow_per_contact, pdim[p] == 1 ? 1 : (elliptic ? pdim[p] : 2*(pdim[p] - 1)));
  if (nconmax <= 0) nconmax = std::min(maxc, 16);
  nconmax = std::max(1, std::min(nconmax, std::max(1, maxc)));
  if (njmax <= 0) njmax = d.neqrow + d.nfric + nlim + d.nlimten + nconmax * maxrow_per_contact;
  njmax = std::max(1, std::min(njmax, std::max(1, maxr + nlim + d.nlimten + d.nfric + d.neqrow)));
  d.nconmax = nconmax; d.njmax = njmax;
  d.elliptic = (elliptic && maxrow_per_contact > 1) ? 1 : 0;
  if (m.opt_noslip_iterations > 0) {
    // friction dimensions: every dof friction row, all rows of a pyramidal contact, all but the normal of an elliptic one
    int maxfr = 0;
    for (int p = 0; p < m.npair; p++) if (pdim[p] > 1) maxfr = std::max(maxfr, elliptic ? pdim[p] - 1 : 2*(pdim[p] - 1));
    d.nslip = std::min(njmax, d.nfric + nconmax * maxfr);
  }
  // Jacobian storage classes (step_layout.h): dense rows for equalities / tendon limits, none for the
  // one-nonzero friction / joint-limit rows, kmax entries per contact row
  d.cg = m.opt_solver == DMC_SOL_CG ? 1 : 0;
  { // kinematic trees proper: dofs with the same root dof (what hangs off a static body by a joint is a tree of its own)
    int nroot = 0;
    for (int i = 0; i < m.nv; i++) if (m.dof_parentid[i] < 0) nroot++;
    d.island = (nroot > 1 && m.opt_noslip_iterations == 0 && m.opt_solver != DMC_SOL_PGS) ? 1 : 0; }
  d.pgs = m.opt_solver == DMC_SOL_PGS ? 1 : 0;
  if (d.pgs) d.nslip = njmax;      // the dual solver keeps a residual entry and a row of AR for EVERY constraint row
  d.jfull = m.nv <= 16 ? 1 : 0;
  d.njdense = d.jfull ? njmax : std::min(njmax, d.neqrow + 2 * d.nlimten + d.nlimball);
  // contact rows with a stored Jacobian: by default every contact slot may use its maximum number of rows; a
  // smaller pool (njcon > 0) trades LDS for a DMC_WARN_CNSTRFULL when the live contacts need more rows than that
  d.njcon = std::min(njmax, nconmax * maxrow_per_contact);
  if (njcon > 0) d.njcon = std::max(maxrow_per_contact, std::min(d.njcon, njcon));
  {
    std::vector<uint64_t> anc(m.nv, 0);
    for (int i = 0; i < m.nv; i++) for (int j = i; j >= 0; j = m.dof_parentid[j]) anc[i] |= (uint64_t)1 << j;
    std::vector<int> lastd(m.nbody, -1);
    for (int b = 1; b < m.nbody; b++) {
      lastd[b] = lastd[m.body_parentid[b]];
      if (m.body_dofnum[b]) lastd[b] = m.body_dofadr[b] + m.body_dofnum[b] - 1;
    }
    int kmax = 1;
    for (int p = 0; p < m.npair; p++) {
      const int b1 = m.geom_bodyid[m.pair_geom1[p]], b2 = m.geom_bodyid[m.pair_geom2[p]];
      const uint64_t mk = (lastd[b1] >= 0 ? anc[lastd[b1]] : 0) ^ (lastd[b2] >= 0 ? anc[lastd[b2]] : 0);
      kmax = std::max(kmax, __builtin_popcountll(mk));
    }
    d.kmax = kmax;
    d.kwords = (kmax + 3) / 4;
  }
  d.msparse = m.nv > 16 ? 1 : 0;
  d.jglobal = (jlevel >= 0 && jlevel <= DMC_JGLOBAL_LEVEL(m.nv)) ? jlevel : DMC_JGLOBAL_LEVEL(m.nv);      // (an override only ever keeps MORE in LDS)
  d.sitegl = m.nsite > 32 ? 1 : 0;
  d.maxrow = maxrow_per_contact;
  d.coldlds = (d.nM + 2 * m.npair) <= 256 ? 1 : 0;
  step_layout_build(&t->L, d);
  const StepLayout& L = t->L;
  t->mi.assign(L.n_mi, 0);
  t->mc.assign(std::max(L.n_mc, 4), 0);
  t->mr.assign(L.n_mr, 0.0);
  int* mi = t->mi.data(); double* mr = t->mr.data();
  auto cpi = [&](int off, const std::vector<int>& v) { std::copy(v.begin(), v.end(), mi + off); };
  auto cpr = [&](int off, const std::vector<double>& v) { std::copy(v.begin(), v.end(), mr + off); };
  cpi(L.mi_body_parentid, m.body_parentid); cpi(L.mi_body_rootid, m.body_rootid);
  cpi(L.mi_body_jntadr, m.body_jntadr); cpi(L.mi_body_jntnum, m.body_jntnum);
  cpi(L.mi_body_dofadr, m.body_dofadr); cpi(L.mi_body_dofnum, m.body_dofnum);
  {
    std::vector<int> last(m.nbody, -1);
    for (int b = 1; b < m.nbody; b++) {
      last[b] = last[m.body_parentid[b]];
      if (m.body_dofnum[b]) last[b] = m.body_dofadr[b] + m.body_dofnum[b] - 1;
    }
    cpi(L.mi_body_lastdof, last);
  }
  if (d.dfs) cpi(L.mi_body_subend, subend);
  {
    int* la = mi + L.mi_level_adr; int* lb = mi + L.mi_level_body;
    int k = 0;
    for (int lev = 1; lev <= nlevel; lev++) { la[lev - 1] = k; for (int b = 1; b < m.nbody; b++) if (depth[b] == lev) lb[k++] = b; }
    la[nlevel] = k;
    int* ca = mi + L.mi_child_adr; int* cl = mi + L.mi_child_list;
    k = 0;
    for (int b = 0; b < m.nbody; b++) { ca[b] = k; for (int c = m.nbody - 1; c > b; c--) if (m.body_parentid[c] == b && c != 0) cl[k++] = c; }
    ca[m.nbody] = k;
  }
  cpi(L.mi_jnt_type, m.jnt_type); cpi(L.mi_jnt_qposadr, m.jnt_qposadr); cpi(L.mi_jnt_dofadr, m.jnt_dofadr);
  cpi(L.mi_jnt_bodyid, m.jnt_bodyid); cpi(L.mi_jnt_limited, m.jnt_limited);
  cpi(L.mi_dof_bodyid, m.dof_bodyid); cpi(L.mi_dof_jntid, m.dof_jntid); cpi(L.mi_dof_parentid, m.dof_parentid);
  for (int i = 0; i < m.nv; i++) {
    uint64_t mask = 0;
    for (int j = i; j >= 0; j = m.dof_parentid[j]) mask |= (uint64_t)1 << j;
    mi[L.mi_dof_anc_lo + i] = (int)(uint32_t)(mask & 0xffffffffu);
    mi[L.mi_dof_anc_hi + i] = (int)(uint32_t)(mask >> 32);
  }
  {
    int* mc = t->mc.data();
    for (int p = 0; p < d.nM; p++) mc[L.mc_mpair + p] = mp_i[p] | (mp_j[p] << 16);
    for (int p = 0; p < m.npair; p++) {
      mc[L.mc_pair_geom + p] = m.pair_geom1[p] | (m.pair_geom2[p] << 16);
      mc[L.mc_pair_info + p] = pdim[p] | (pair_prm[p] << 8);
    }
    // sparse M: row i = entries (i, i), (i, parent(i)), ... in mpair order
    int k = 0;
    for (int i = 0; i < m.nv; i++) { mi[L.mi_dof_madr + i] = k; for (int j = i; j >= 0; j = m.dof_parentid[j]) k++; }
    mi[L.mi_dof_madr + m.nv] = k;
    for (int i = 0; i < m.nv; i++) {
      int end = i + 1;
      for (int dd = i + 1; dd < m.nv; dd++) for (int j = dd; j >= 0; j = m.dof_parentid[j]) if (j == i) { end = dd + 1; break; }
      mi[L.mi_dof_subend + i] = end;
    }
    if (d.treemax) {
      for (int i = 0, t0 = 0; i < m.nv; i++) {
        if (i > 0 && m.body_rootid[m.dof_bodyid[i]] != m.body_rootid[m.dof_bodyid[i - 1]]) t0 = i;
        mi[L.mi_dof_tree0 + i] = t0;
      }
      for (int i = m.nv - 1, t1 = m.nv; i >= 0; i--) {
        if (i < m.nv - 1 && m.body_rootid[m.dof_bodyid[i]] != m.body_rootid[m.dof_bodyid[i + 1]]) t1 = i + 1;
        mi[L.mi_dof_tree1 + i] = t1;
      }
      int k = 0;      // in-tree entries, column by column inside each tree (the order of the packed triangle)
      for (int j = 0; j < m.nv; j++) for (int i = j; i < m.nv; i++) {
        if (m.body_rootid[m.dof_bodyid[i]] != m.body_rootid[m.dof_bodyid[j]]) continue;
        int idx = -1, step = 0;
        for (int a = i; a >= 0; a = m.dof_parentid[a], step++) if (a == j) { idx = mi[L.mi_dof_madr + i] + step; break; }
        mi[L.mi_tree_tri + k] = i | (j << 16);
        mi[L.mi_tree_trim + k] = idx;
        k++;
      }
    }
  }
  cpi(L.mi_geom_type, m.geom_type); cpi(L.mi_geom_bodyid, m.geom_bodyid);
  if (d.nrf) cpi(L.mi_geom_invisible, m.geom_invisible);
  cpi(L.mi_site_bodyid, m.site_bodyid); cpi(L.mi_site_type, m.site_type);
  int nact = 0;
  for (int i = 0; i < m.nu; i++) {
    const int j = m.actuator_trnid[2*i];
    const bool tendon = m.actuator_trntype[i] == DMC_TRN_TENDON;
    if (tendon) { if (j < 0 || j >= m.ntendon) { *err = "actuator refers to a missing tendon"; return false; } }
    else if (m.actuator_trntype[i] != DMC_TRN_JOINT || j < 0 || (m.jnt_type[j] != DMC_JNT_HINGE && m.jnt_type[j] != DMC_JNT_SLIDE)) { *err = "only hinge/slide joint and fixed-tendon transmissions are implemented"; return false; }
    if (tendon) { mi[L.mi_act_dof + i] = j; mi[L.mi_act_qpos + i] = 0; }
    else { mi[L.mi_act_dof + i] = m.jnt_dofadr[j]; mi[L.mi_act_qpos + i] = m.jnt_qposadr[j]; }
    int fl = tendon ? ACTF_TENDON : 0;
    if (m.actuator_ctrllimited[i]) fl |= ACTF_CTRLLIMITED;
    if (m.actuator_forcelimited[i]) fl |= ACTF_FORCELIMITED;
    if (m.actuator_gaintype[i] == DMC_GAIN_AFFINE) fl |= ACTF_GAIN_AFFINE;
    if (m.actuator_biastype[i] == DMC_BIAS_AFFINE) fl |= ACTF_BIAS_AFFINE;
    if (m.actuator_dyntype[i] == DMC_DYN_INTEGRATOR) fl |= ACTF_DYN_INTEGRATOR;
    else if (m.actuator_dyntype[i] == DMC_DYN_FILTER) fl |= ACTF_DYN_FILTER;
    else if (m.actuator_dyntype[i] == DMC_DYN_FILTEREXACT) fl |= ACTF_DYN_FILTEREXACT;
    if (m.actuator_actlimited[i]) {
      if (!(fl & ACTF_DYN_ANY)) { *err = "actlimited on an actuator without activation dynamics"; return false; }
      fl |= ACTF_ACTLIMITED;
    }
    if (d.na) {
      mi[L.mi_act_adr + i] = (fl & ACTF_DYN_ANY) ? nact++ : -1; mr[L.mr_act_dynprm + i] = m.actuator_dynprm[10*i];
      mr[L.mr_act_actrange + 2*i] = m.actuator_actrange[2*i]; mr[L.mr_act_actrange + 2*i + 1] = m.actuator_actrange[2*i + 1];
    }
    mi[L.mi_act_flags + i] = fl;
    mr[L.mr_act_gear + i] = m.actuator_gear[6*i];
    for (int k = 0; k < 2; k++) { mr[L.mr_act_ctrlrange + 2*i + k] = m.actuator_ctrlrange[2*i + k]; mr[L.mr_act_forcerange + 2*i + k] = m.actuator_forcerange[2*i + k]; }
    for (int k = 0; k < 3; k++) { mr[L.mr_act_gainprm + 3*i + k] = m.actuator_gainprm[10*i + k]; mr[L.mr_act_biasprm + 3*i + k] = m.actuator_biasprm[10*i + k]; }
  }
  cpi(L.mi_sensor_type, m.sensor_type); cpi(L.mi_sensor_objid, m.sensor_objid);
  cpi(L.mi_sensor_adr, m.sensor_adr); cpi(L.mi_sensor_stage, m.sensor_needstage);
  // object type in the low byte; a reference frame (frame{pos,quat,?axis}) as reftype << 8 | (refid + 1) << 16
  for (int i = 0; i < m.nsensor; i++)
    mi[L.mi_sensor_objtype + i] = m.sensor_objtype[i] | (m.sensor_refid[i] >= 0 ? (m.sensor_reftype[i] << 8) | ((m.sensor_refid[i] + 1) << 16) : 0);
  cpr(L.mr_qpos0, m.qpos0); cpr(L.mr_qpos_spring, m.qpos_spring);
  cpr(L.mr_body_pos, m.body_pos); cpr(L.mr_body_quat, m.body_quat); cpr(L.mr_body_ipos, m.body_ipos);
  cpr(L.mr_body_iquat, m.body_iquat); cpr(L.mr_body_mass, m.body_mass); cpr(L.mr_body_inertia, m.body_inertia);
  cpr(L.mr_body_subtreemass, m.body_subtreemass); cpr(L.mr_body_invweight0, m.body_invweight0);
  for (int b = 0; b < m.nbody; b++) mr[L.mr_body_invsubtreemass + b] = 1.0 / std::max((double)DMC_MINVAL, m.body_subtreemass[b]);
  cpr(L.mr_jnt_pos, m.jnt_pos); cpr(L.mr_jnt_axis, m.jnt_axis); cpr(L.mr_jnt_stiffness, m.jnt_stiffness);
  cpr(L.mr_jnt_range, m.jnt_range); cpr(L.mr_jnt_margin, m.jnt_margin); cpr(L.mr_jnt_solref, m.jnt_solref);
  cpr(L.mr_jnt_solimp, m.jnt_solimp);
  cpr(L.mr_dof_armature, m.dof_armature); cpr(L.mr_dof_damping, m.dof_damping); cpr(L.mr_dof_invweight0, m.dof_invweight0);
  cpr(L.mr_geom_size, m.geom_size); cpr(L.mr_geom_pos, m.geom_pos); cpr(L.mr_geom_quat, m.geom_quat);
  cpr(L.mr_geom_rbound, m.geom_rbound);
  for (int q = 0; q < d.nprm; q++) {
    const double* v = prm[q].data();
    mr[L.mr_prm_margin + q] = v[0]; mr[L.mr_prm_gap + q] = v[1];
    for (int k = 0; k < 3; k++) mr[L.mr_prm_friction + 3*q + k] = v[2 + k];
    for (int k = 0; k < 2; k++) mr[L.mr_prm_solref + 2*q + k] = v[5 + k];
    for (int k = 0; k < 5; k++) mr[L.mr_prm_solimp + 5*q + k] = v[7 + k];
  }
  cpi(L.mi_fric_dof, fric_dof);
  cpi(L.mi_stv_sensor, stv);
  if (d.nmocap) for (int b = 0; b < m.nbody; b++) mi[L.mi_body_mocapid + b] = m.body_mocapid[b];
  if (d.nstv) for (int b = 0; b < m.nbody; b++) {
    uint64_t mask = 0;
    for (int a = b; ; a = m.body_parentid[a]) { mask |= 1ull << a; if (a == 0) break; }
    mi[L.mi_body_anc_lo + b] = (int)(uint32_t)(mask & 0xffffffffu); mi[L.mi_body_anc_hi + b] = (int)(uint32_t)(mask >> 32);
  }
  cpi(L.mi_tendon_adr, m.tendon_adr); cpi(L.mi_tendon_num, m.tendon_num); cpr(L.mr_wrap_prm, m.wrap_prm);
  cpr(L.mr_tendon_stiffness, m.tendon_stiffness); cpr(L.mr_tendon_damping, m.tendon_damping); cpr(L.mr_tendon_lengthspring, m.tendon_lengthspring);
  for (int w = 0; w < m.nwrap; w++) {
    const bool site = m.wrap_type[w] == DMC_WRAP_SITE;
    mi[L.mi_wrap_dof + w] = site ? 0 : m.jnt_dofadr[m.wrap_objid[w]]; mi[L.mi_wrap_qpos + w] = site ? 0 : m.jnt_qposadr[m.wrap_objid[w]];
    mi[L.mi_wrap_site + w] = site ? m.wrap_objid[w] : -1;
  }
  cpi(L.mi_limten, limten); cpi(L.mi_limball, limball);
  if (d.nlimten) {
    cpr(L.mr_tendon_range, m.tendon_range); cpr(L.mr_tendon_margin, m.tendon_margin); cpr(L.mr_tendon_solref_lim, m.tendon_solref_lim);
    cpr(L.mr_tendon_solimp_lim, m.tendon_solimp_lim);
  }
  if (d.nlimten || d.neq) cpr(L.mr_tendon_invweight0, m.tendon_invweight0);
  cpi(L.mi_eq_rowadr, eq_rowadr);
  for (int k = 0; k < d.neq; k++) {
    const int src = eq_src[k], et = m.eq_type[src];
    mi[L.mi_eq_type + k] = et; mi[L.mi_eq_obj1 + k] = m.eq_obj1id[src]; mi[L.mi_eq_obj2 + k] = m.eq_obj2id[src];
    for (int a = 0; a < 2; a++) mr[L.mr_eq_solref + 2*k + a] = m.eq_solref[2*src + a];
    for (int a = 0; a < 5; a++) mr[L.mr_eq_solimp + 5*k + a] = m.eq_solimp[5*src + a];
    for (int a = 0; a < 11; a++) mr[L.mr_eq_data + 13*k + a] = m.eq_data[11*src + a];
    if (et == DMC_EQ_TENDON) mr[L.mr_eq_data + 13*k + 11] = m.tendon_length0[m.eq_obj1id[src]];
    if (et == DMC_EQ_JOINT) {
      mr[L.mr_eq_data + 13*k + 11] = m.qpos0[m.jnt_qposadr[m.eq_obj1id[src]]];
      if (m.eq_obj2id[src] >= 0) mr[L.mr_eq_data + 13*k + 12] = m.qpos0[m.jnt_qposadr[m.eq_obj2id[src]]];
    }
  }
  if (d.nfric) { cpr(L.mr_dof_frictionloss, m.dof_frictionloss); cpr(L.mr_dof_solref, m.dof_solref); cpr(L.mr_dof_solimp, m.dof_solimp); }
  cpr(L.mr_site_pos, m.site_pos); cpr(L.mr_site_quat, m.site_quat); cpr(L.mr_site_size, m.site_size);
  // sensors the kernel can compute
  for (int i = 0; i < m.nsensor; i++) {
    const int st = m.sensor_type[i];
    const bool ok = st == DMC_SENS_JOINTPOS || st == DMC_SENS_JOINTVEL || st == DMC_SENS_ACTUATORFRC ||
                    st == DMC_SENS_SUBTREECOM || st == DMC_SENS_SUBTREELINVEL || st == DMC_SENS_VELOCIMETER ||
                    st == DMC_SENS_GYRO || st == DMC_SENS_ACCELEROMETER || st == DMC_SENS_FORCE ||
                    st == DMC_SENS_TORQUE || st == DMC_SENS_TOUCH || st == DMC_SENS_FRAMEPOS ||
                    st == DMC_SENS_FRAMEXAXIS || st == DMC_SENS_FRAMEYAXIS || st == DMC_SENS_FRAMEZAXIS ||
                    st == DMC_SENS_FRAMEQUAT || st == DMC_SENS_FRAMELINVEL || st == DMC_SENS_FRAMEANGVEL ||
                    st == DMC_SENS_RANGEFINDER;
    if (!ok) { *err = "sensor type not implemented"; return false; }
    if (m.sensor_refid[i] >= 0 && (m.sensor_refid[i] > 32766 || !(st == DMC_SENS_FRAMEPOS || st == DMC_SENS_FRAMEQUAT || (st >= DMC_SENS_FRAMEXAXIS && st <= DMC_SENS_FRAMEZAXIS) || st == DMC_SENS_FRAMELINVEL || st == DMC_SENS_FRAMEANGVEL))) { *err = "sensor reference frames are implemented for the frame* sensors"; return false; }
    if (st == DMC_SENS_TOUCH) { const int tt = m.site_type[m.sensor_objid[i]]; if (tt != DMC_GEOM_SPHERE && tt != DMC_GEOM_CAPSULE && tt != DMC_GEOM_BOX && tt != DMC_GEOM_ELLIPSOID) { *err = "touch sensor sites must be sphere, capsule, ellipsoid or box"; return false; } }
  }
  StepOpts<double>& o = t->opts;
  o.timestep = m.opt_timestep; o.timestep_d = m.opt_timestep; o.gravity[0] = m.opt_gravity_x; o.gravity[1] = m.opt_gravity_y; o.gravity[2] = m.opt_gravity_z;
  o.impratio = m.opt_impratio; o.tolerance = m.opt_tolerance; o.ls_tolerance = m.opt_ls_tolerance;
  o.meaninertia = m.stat_meaninertia; o.density = m.opt_density; o.viscosity = m.opt_viscosity;
  o.integrator = m.opt_integrator; o.cone = m.opt_cone; o.iterations = m.opt_iterations;
  o.ls_iterations = m.opt_ls_iterations; o.disableflags = m.opt_disableflags;
  o.noslip_iterations = m.opt_noslip_iterations; o.noslip_tolerance = m.opt_noslip_tolerance;
  o.any_damping = 0; o.islands = -1;
  o.eg_data = nullptr; o.eg_slot = nullptr; o.eg_n = 0; o.eg_B = 0; o.ns_A = nullptr; o.xfrc = nullptr; o.xfrc_B = 0; o.gscr = nullptr; o.g_mr = nullptr;
  o.mocap_pos = nullptr; o.mocap_quat = nullptr; o.mocap_B = 0;
  for (int i = 0; i < m.nv; i++) if (m.dof_damping[i] > 0) o.any_damping = 1;
  return true;
}

template <typename T>
inline StepOpts<T> step_opts_cast(const StepOpts<double>& s) {
  StepOpts<T> o;
  o.timestep = (T)s.timestep; for (int k = 0; k < 3; k++) o.gravity[k] = (T)s.gravity[k];
  o.impratio = (T)s.impratio; o.tolerance = (T)s.tolerance; o.ls_tolerance = (T)s.ls_tolerance;
  o.meaninertia = (T)s.meaninertia; o.density = (T)s.density; o.viscosity = (T)s.viscosity;
  o.integrator = s.integrator; o.cone = s.cone;
  o.iterations = s.iterations; o.ls_iterations = s.ls_iterations; o.disableflags = s.disableflags;
  o.noslip_iterations = s.noslip_iterations; o.noslip_tolerance = (T)s.noslip_tolerance;
  o.any_damping = s.any_damping; o.timestep_d = s.timestep_d; o.islands = s.islands;
  o.eg_data = s.eg_data; o.eg_slot = s.eg_slot; o.eg_n = s.eg_n; o.eg_B = s.eg_B; o.ns_A = s.ns_A; o.xfrc = s.xfrc; o.xfrc_B = s.xfrc_B; o.mocap_pos = s.mocap_pos; o.mocap_quat = s.mocap_quat; o.mocap_B = s.mocap_B; o.gscr = s.gscr; o.g_mr = s.g_mr;
  return o;
}

}  // namespace dmc
