// C-ABI (include/vnl.h) + kernels of the rodent rollout for gfx950.
// Build: hipcc --offload-arch=gfx950 -O3 -shared -fPIC (see csrc/build.py).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <map>
#include <string>
#include <vector>

#include "../../include/vnl.h"
#include "vnl_body.h"

// ----------------------------------------------------------------------------- errors
static thread_local char g_err[512] = "";
static int fail(int code, const char* fmt, const char* a = "", const char* b = "") {
  snprintf(g_err, sizeof(g_err), fmt, a, b);
  return code;
}
#define HIPCHK(call)                                                              \
  do {                                                                            \
    hipError_t e_ = (call);                                                       \
    if (e_ != hipSuccess) return fail(VNL_ERR_HIP, "%s: %s", #call, hipGetErrorString(e_)); \
  } while (0)

extern "C" const char* vnl_last_error(void) { return g_err; }
void vnl_set_error_(const char* msg) { snprintf(g_err, sizeof(g_err), "%s", msg); }  // used by vnl_policy.hip
extern "C" int vnl_version(void) { return 1; }

// ----------------------------------------------------------------------------- blob
struct BlobEntry {
  char name[24];
  uint32_t dtype, count;
  uint64_t offset;
};

struct vnl_model {
  std::map<std::string, std::vector<double>> f;
  std::map<std::string, std::vector<int>> i;
  double scalar(const char* k) const {
    auto it = f.find(k);
    return (it == f.end() || it->second.size() != 1) ? NAN : it->second[0];
  }
  bool has_f(const char* k, size_t n) const {
    auto it = f.find(k);
    return it != f.end() && it->second.size() == n;
  }
  bool has_i(const char* k, size_t n) const {
    auto it = i.find(k);
    return it != i.end() && it->second.size() == n;
  }
};

extern "C" int vnl_model_create(const void* blob, size_t nbytes, vnl_model** out) {
  if (!blob || !out) return fail(VNL_ERR_ARG, "vnl_model_create: null argument");
  const uint8_t* b = (const uint8_t*)blob;
  if (nbytes < 16 || memcmp(b, "VNLMDL01", 8) != 0) return fail(VNL_ERR_BLOB, "bad blob magic");
  uint32_t ns;
  memcpy(&ns, b + 8, 4);
  if (16 + (size_t)ns * sizeof(BlobEntry) > nbytes) return fail(VNL_ERR_BLOB, "truncated blob directory");
  vnl_model* m = new vnl_model();
  const BlobEntry* e = (const BlobEntry*)(b + 16);
  for (uint32_t k = 0; k < ns; k++) {
    size_t esz = e[k].dtype == 1 ? 8 : 4;
    if (e[k].offset + esz * e[k].count > nbytes) {
      delete m;
      return fail(VNL_ERR_BLOB, "truncated blob section %s", e[k].name);
    }
    std::string name(e[k].name, strnlen(e[k].name, 24));
    if (e[k].dtype == 1) {
      std::vector<double> v(e[k].count);
      memcpy(v.data(), b + e[k].offset, 8 * (size_t)e[k].count);
      m->f[name] = v;
    } else if (e[k].dtype == 2) {
      std::vector<int> v(e[k].count);
      memcpy(v.data(), b + e[k].offset, 4 * (size_t)e[k].count);
      m->i[name] = v;
    } else {
      delete m;
      return fail(VNL_ERR_BLOB, "unknown dtype in section %s", e[k].name);
    }
  }
  static const char* need[] = {"nq", "nv", "nu", "nbody", "njnt", "ncg", "ncon", "nlimit", "nefc", "timestep",
                               "tolerance", "ls_tolerance", "impratio", "meaninertia", "iterations",
                               "ls_iterations", "eulerdamp", "solver_newton"};
  for (const char* k : need)
    if (m->scalar(k) != m->scalar(k)) {
      delete m;
      return fail(VNL_ERR_BLOB, "blob lacks scalar %s", k);
    }
  *out = m;
  return VNL_OK;
}
extern "C" void vnl_model_destroy(vnl_model* m) { delete m; }

// ----------------------------------------------------------------------------- env
struct vnl_env {
  int device = 0, B = 0;
  DevModel dm{};
  DevEnv de{};
  WsLayout L{};
  KernelConsts* kc = nullptr;  // device copy of {dm, de, L}
  vreal* dump = nullptr;  // [B][L.total] image of the per-env LDS, written only when debug is on
  int* trace = nullptr;   // [B][n_frames][VNL_TRACE_INTS] solver decisions, written only when debug is on
  int debug = 0;          // 0 off, 1 image at the end of reset / step, 2 image after the last forward pass of a step
  int spec = 0;           // 1: the kernels specialised for the rodent's dims and layout (VnlSpecRodent) run this env
  size_t lds_bytes = 0;
  int blocks_per_cu = 0;
  std::vector<void*> allocs;
  std::map<std::string, std::pair<int, int>> sections;  // name -> (offset, count)
};

// makes `device` current for the scope and restores the caller's device afterwards
struct DeviceGuard {
  int prev = -1;
  bool ok = false, switched = false;
  explicit DeviceGuard(int device) {
    if (hipGetDevice(&prev) != hipSuccess) return;
    if (prev != device) {
      if (hipSetDevice(device) != hipSuccess) return;
      switched = true;
    }
    ok = true;
  }
  ~DeviceGuard() {
    if (switched) (void)hipSetDevice(prev);
  }
};

template <class T, class S>
static int upload(vnl_env* env, const std::vector<S>& src, const T** dst) {
  std::vector<T> tmp(src.begin(), src.end());
  if (tmp.empty()) tmp.push_back(T(0));
  void* p = nullptr;
  HIPCHK(hipMalloc(&p, tmp.size() * sizeof(T)));
  env->allocs.push_back(p);
  HIPCHK(hipMemcpy(p, tmp.data(), tmp.size() * sizeof(T), hipMemcpyHostToDevice));
  *dst = (const T*)p;
  return VNL_OK;
}
template <class T>
static int upload_raw(vnl_env* env, const T* src, size_t n, const T** dst) {
  std::vector<T> tmp(src, src + n);
  return upload<T, T>(env, tmp, dst);
}

// ---- welded bodies -------------------------------------------------------------------------------------------------
// A body without joints moves rigidly with its parent: for the dynamics it IS part of the parent.  The rodent has 13 such
// bodies among its 66, which is what pushed every per-body loop of the step kernel over the 64 lanes of a wave.  The model
// handed to the kernels therefore has them folded into their parents -- mass, centre of mass and inertia merged exactly
// (parallel-axis theorem in the parent's frame), children and collision geoms re-attached with composed fixed transforms --
// while every body of the model AS GIVEN keeps its row of xpos / xquat (pose of its dynamic body composed with its fixed
// transform).  The oracle works on the model as given: the parity tests check this transformation with everything else.
struct FuseMap {
  int nb_out = 0;
  std::vector<int> out_dyn, body_out;
  std::vector<double> out_pos, out_quat;
};
static void qmul_d(const double* a, const double* b, double* o) {
  o[0] = a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3];
  o[1] = a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2];
  o[2] = a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1];
  o[3] = a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0];
}
static void qmat_d(const double* q, double* R) {
  const double w = q[0], x = q[1], y = q[2], z = q[3];
  R[0] = 1 - 2 * (y * y + z * z), R[1] = 2 * (x * y - w * z), R[2] = 2 * (x * z + w * y);
  R[3] = 2 * (x * y + w * z), R[4] = 1 - 2 * (x * x + z * z), R[5] = 2 * (y * z - w * x);
  R[6] = 2 * (x * z - w * y), R[7] = 2 * (y * z + w * x), R[8] = 1 - 2 * (x * x + y * y);
}
static void mulv_d(const double* R, const double* v, double* o) {
  for (int r = 0; r < 3; r++) o[r] = R[3 * r] * v[0] + R[3 * r + 1] * v[1] + R[3 * r + 2] * v[2];
}
static bool fuse_welded_bodies(const vnl_model& in, vnl_model* out, FuseMap* fm) {
  const int nb = (int)in.scalar("nbody");
  const auto& bp = in.i.at("body_parentid");
  const auto& jn = in.i.at("body_jntnum");
  std::vector<char> keep(nb, 1);
  int nd = 0;
  for (int b = 0; b < nb; b++) {
    keep[b] = (b < 2 || jn[b] > 0) ? 1 : 0;
    // (a jointless body hanging off the world -- directly or through other jointless bodies -- is NOT folded: such a model is
    // not a single tree rooted at body 1 and is rejected below, as it always was)
    if (!keep[b]) {
      int a = bp[b];
      while (a > 1 && !keep[a]) a = bp[a];
      if (a == 0) keep[b] = 1;
    }
    nd += keep[b];
  }
  fm->nb_out = nb;
  fm->out_dyn.assign(nb, 0), fm->out_pos.assign(3 * (size_t)nb, 0.0), fm->out_quat.assign(4 * (size_t)nb, 0.0);
  fm->body_out.clear();
  const auto& bpos = in.f.at("body_pos");
  const auto& bquat = in.f.at("body_quat");
  for (int b = 0, k = 0; b < nb; b++) {  // fixed transform of every body in the frame of the dynamic body it rides on
    double* fp = &fm->out_pos[3 * (size_t)b];
    double* fq = &fm->out_quat[4 * (size_t)b];
    if (keep[b]) {
      fm->out_dyn[b] = k++, fm->body_out.push_back(b);
      fq[0] = 1.0;
    } else {
      const int p = bp[b];
      fm->out_dyn[b] = fm->out_dyn[p];
      double R[9], t[3];
      qmat_d(&fm->out_quat[4 * (size_t)p], R), mulv_d(R, &bpos[3 * (size_t)b], t);
      for (int c = 0; c < 3; c++) fp[c] = fm->out_pos[3 * (size_t)p + c] + t[c];
      qmul_d(&fm->out_quat[4 * (size_t)p], &bquat[4 * (size_t)b], fq);
    }
  }
  if (nd == nb) return false;
  *out = in;
  out->f["nbody"] = {(double)nd};
  auto pick_i = [&](const char* k) {
    std::vector<int> v;
    for (int b = 0; b < nb; b++)
      if (keep[b]) v.push_back(in.i.at(k)[b]);
    out->i[k] = v;
  };
  auto pick_f = [&](const char* k, int w) {
    std::vector<double> v;
    for (int b = 0; b < nb; b++)
      if (keep[b]) v.insert(v.end(), in.f.at(k).begin() + (size_t)w * b, in.f.at(k).begin() + (size_t)w * (b + 1));
    out->f[k] = v;
  };
  pick_i("body_jntadr"), pick_i("body_jntnum"), pick_i("body_rootid"), pick_i("body_dofadr"), pick_i("body_dofnum");
  pick_f("body_invweight0", 2);
  std::vector<int> par;
  std::vector<double> pos, quat;
  for (int b = 0; b < nb; b++) {
    if (!keep[b]) continue;
    const int p = b > 0 ? bp[b] : 0;
    par.push_back(fm->out_dyn[p]);
    double R[9], t[3], q[4];
    qmat_d(&fm->out_quat[4 * (size_t)p], R), mulv_d(R, &bpos[3 * (size_t)b], t);
    for (int c = 0; c < 3; c++) pos.push_back(fm->out_pos[3 * (size_t)p + c] + t[c]);
    qmul_d(&fm->out_quat[4 * (size_t)p], &bquat[4 * (size_t)b], q);
    quat.insert(quat.end(), q, q + 4);
  }
  out->i["body_parentid"] = par, out->f["body_pos"] = pos, out->f["body_quat"] = quat;
  // mass / centre of mass / inertia of every dynamic body with what is welded to it: moments about the body's origin, in its axes
  std::vector<double> M(nd, 0.0), mc(3 * (size_t)nd, 0.0), J(9 * (size_t)nd, 0.0);
  const auto& mass = in.f.at("body_mass");
  const auto& ipos = in.f.at("body_ipos");
  const auto& ifull = in.f.at("body_inertia_full");
  for (int b = 0; b < nb; b++) {
    const int dyn = fm->out_dyn[b];
    double R[9], c[3], t[3];
    qmat_d(&fm->out_quat[4 * (size_t)b], R), mulv_d(R, &ipos[3 * (size_t)b], t);
    for (int k = 0; k < 3; k++) c[k] = fm->out_pos[3 * (size_t)b + k] + t[k];
    const double m = mass[b], cc = c[0] * c[0] + c[1] * c[1] + c[2] * c[2];
    M[dyn] += m;
    for (int k = 0; k < 3; k++) mc[3 * (size_t)dyn + k] += m * c[k];
    const double* I = &ifull[9 * (size_t)b];
    for (int r = 0; r < 3; r++)
      for (int q = 0; q < 3; q++) {
        double v = 0;  // (R I R')(r, q)
        for (int a = 0; a < 3; a++)
          for (int e = 0; e < 3; e++) v += R[3 * r + a] * I[3 * a + e] * R[3 * q + e];
        J[9 * (size_t)dyn + 3 * r + q] += v + m * ((r == q ? cc : 0.0) - c[r] * c[q]);
      }
  }
  std::vector<double> nip(3 * (size_t)nd, 0.0), nI(9 * (size_t)nd, 0.0);
  for (int dyn = 0; dyn < nd; dyn++) {
    double c[3] = {0, 0, 0};
    if (M[dyn] > 0)
      for (int k = 0; k < 3; k++) c[k] = mc[3 * (size_t)dyn + k] / M[dyn];
    else
      for (int k = 0; k < 3; k++) c[k] = ipos[3 * (size_t)fm->body_out[dyn] + k];
    const double cc = c[0] * c[0] + c[1] * c[1] + c[2] * c[2];
    for (int k = 0; k < 3; k++) nip[3 * (size_t)dyn + k] = c[k];
    for (int r = 0; r < 3; r++)
      for (int q = 0; q < 3; q++) nI[9 * (size_t)dyn + 3 * r + q] = J[9 * (size_t)dyn + 3 * r + q] - M[dyn] * ((r == q ? cc : 0.0) - c[r] * c[q]);
  }
  out->f["body_mass"] = M, out->f["body_ipos"] = nip, out->f["body_inertia_full"] = nI;
  for (const char* k : {"jnt_bodyid", "dof_bodyid"}) {
    std::vector<int> v = in.i.at(k);
    for (int& x : v) x = fm->out_dyn[x];
    out->i[k] = v;
  }
  {  // collision geoms: onto the dynamic body, frames composed; the contact's inverse weight stays that of the body as given
    const auto& gb = in.i.at("cg_bodyid");
    const auto& gp = in.f.at("cg_pos");
    const auto& gq = in.f.at("cg_quat");
    const auto& iw0 = in.f.at("body_invweight0");
    std::vector<int> nb_(gb.size());
    std::vector<double> np = gp, nq = gq, giw(gb.size());
    for (size_t g = 0; g < gb.size(); g++) {
      const int ob = gb[g];
      nb_[g] = fm->out_dyn[ob], giw[g] = iw0[2 * (size_t)ob];
      double R[9], t[3];
      qmat_d(&fm->out_quat[4 * (size_t)ob], R), mulv_d(R, &gp[3 * g], t);
      for (int c = 0; c < 3; c++) np[3 * g + c] = fm->out_pos[3 * (size_t)ob + c] + t[c];
      qmul_d(&fm->out_quat[4 * (size_t)ob], &gq[4 * g], &nq[4 * g]);
    }
    out->i["cg_bodyid"] = nb_, out->f["cg_pos"] = np, out->f["cg_quat"] = nq, out->f["cg_body_invweight0"] = giw;
  }
  return true;
}

static int build_dev_model(vnl_env* env, const vnl_model* hm) {
  DevModel& d = env->dm;
  vnl_model fused;
  FuseMap fmap;
  for (const char* k : {"body_parentid", "body_jntnum"})
    if (hm->i.find(k) == hm->i.end()) return fail(VNL_ERR_BLOB, "blob section %s missing", k);
  for (const char* k : {"body_pos", "body_quat", "body_ipos", "body_inertia_full", "body_mass", "body_invweight0", "cg_pos", "cg_quat"})
    if (hm->f.find(k) == hm->f.end()) return fail(VNL_ERR_BLOB, "blob section %s missing", k);
  for (const char* k : {"jnt_bodyid", "dof_bodyid", "cg_bodyid", "body_jntadr", "body_rootid", "body_dofadr", "body_dofnum"})
    if (hm->i.find(k) == hm->i.end()) return fail(VNL_ERR_BLOB, "blob section %s missing", k);
  if (fuse_welded_bodies(*hm, &fused, &fmap)) hm = &fused;
  d.nbody_out = fmap.nb_out;
  auto S = [&](const char* k) { return hm->scalar(k); };
  d.nq = (int)S("nq"), d.nv = (int)S("nv"), d.nu = (int)S("nu"), d.nbody = (int)S("nbody"), d.njnt = (int)S("njnt");
  d.ncg = (int)S("ncg"), d.ncon = (int)S("ncon"), d.nlimit = (int)S("nlimit"), d.nefc = (int)S("nefc");
  d.iterations = (int)S("iterations"), d.ls_iterations = (int)S("ls_iterations"), d.eulerdamp = (int)S("eulerdamp");
  d.dt = (vreal)S("timestep"), d.tolerance = (vreal)S("tolerance"), d.ls_tolerance = (vreal)S("ls_tolerance");
  d.solver_newton = S("solver_newton") != 0 ? 1 : 0;
  // Newton (reference configs/env_config.yaml:16-21, the ant): the Hessian M + J' D J is formed and factorised DENSE in LDS
  if (d.solver_newton && (d.nv > 48 || d.nefc > 160))
    return fail(VNL_ERR_UNSUPPORTED, "the Newton solver is implemented for small models (nv <= 48, nefc <= 160): dense Hessian in LDS");
  d.dbg_stage = 0, d.dbg_count = 0;
#ifdef VNL_STAGE_KNOBS  // diagnostic library only: the product library reads no environment variables
  if (const char* dbg = getenv("VNL_DBG_REPEAT")) sscanf(dbg, "%d:%d", &d.dbg_stage, &d.dbg_count);
#endif
  d.scale = (vreal)(S("meaninertia") * (d.nv > 1 ? d.nv : 1));
  const int nb = d.nbody, nj = d.njnt, nv = d.nv, nu = d.nu, ng = d.ncg;
  std::vector<int> nsub_host, limrow_host;  // kept for the per-dof contact ranges further down
  struct Need {
    const char* k;
    size_t n;
    bool isint;
  } needs[] = {{"body_parentid", (size_t)nb, true}, {"body_jntadr", (size_t)nb, true}, {"body_jntnum", (size_t)nb, true},
               {"body_rootid", (size_t)nb, true}, {"body_dofadr", (size_t)nb, true}, {"body_dofnum", (size_t)nb, true}, {"body_pos", 3u * nb, false}, {"body_quat", 4u * nb, false},
               {"body_ipos", 3u * nb, false}, {"body_inertia_full", 9u * nb, false}, {"body_mass", (size_t)nb, false},
               {"body_invweight0", 2u * nb, false}, {"jnt_type", (size_t)nj, true}, {"jnt_qposadr", (size_t)nj, true},
               {"jnt_dofadr", (size_t)nj, true}, {"jnt_bodyid", (size_t)nj, true}, {"jnt_limited", (size_t)nj, true}, {"jnt_pos", 3u * nj, false},
               {"jnt_axis", 3u * nj, false}, {"jnt_range", 2u * nj, false}, {"jnt_stiffness", (size_t)nj, false},
               {"jnt_margin", (size_t)nj, false}, {"jnt_solref", 2u * nj, false}, {"jnt_solimp", 5u * nj, false},
               {"qpos0", (size_t)d.nq, false}, {"qpos_spring", (size_t)d.nq, false}, {"dof_bodyid", (size_t)nv, true},
               {"dof_parentid", (size_t)nv, true}, {"dof_armature", (size_t)nv, false},
               {"dof_damping", (size_t)nv, false}, {"dof_invweight0", (size_t)nv, false}, {"act_dof", (size_t)nu, true},
               {"act_ctrllimited", (size_t)nu, true}, {"act_gain", (size_t)nu, false}, {"act_gear", (size_t)nu, false},
               {"act_tau", (size_t)nu, false}, {"act_ctrlrange", 2u * nu, false}, {"cg_type", (size_t)ng, true},
               {"cg_bodyid", (size_t)ng, true}, {"cg_ncon", (size_t)ng, true}, {"cg_conadr", (size_t)ng, true},
               {"cg_pos", 3u * ng, false}, {"cg_quat", 4u * ng, false}, {"cg_size", 3u * ng, false},
               {"cg_friction", 3u * ng, false}, {"cg_solref", 2u * ng, false}, {"cg_solimp", 5u * ng, false},
               {"cg_margin", (size_t)ng, false}, {"gravity", 3, false}, {"plane_pos", 3, false},
               {"plane_normal", 3, false}};
  for (auto& n : needs)
    if (!(n.isint ? hm->has_i(n.k, n.n) : hm->has_f(n.k, n.n)))
      return fail(VNL_ERR_BLOB, "blob section %s missing or of unexpected size", n.k);
  auto F = [&](const char* k) -> const std::vector<double>& { return hm->f.at(k); };
  auto I = [&](const char* k) -> const std::vector<int>& { return hm->i.at(k); };

  // single kinematic tree rooted at body 1 (the spatial reference point is its origin)
  for (int b = 1; b < nb; b++)
    if (I("body_rootid")[b] != 1) return fail(VNL_ERR_UNSUPPORTED, "model must be a single tree rooted at body 1");
  const auto& jt = I("jnt_type");
  for (int j = 0; j < nj; j++)
    if (jt[j] != VNL_JNT_FREE && jt[j] != VNL_JNT_HINGE)
      return fail(VNL_ERR_UNSUPPORTED, "only free and hinge joints are implemented");
  d.root_free = (I("body_jntnum")[1] > 0 && jt[I("body_jntadr")[1]] == VNL_JNT_FREE) ? 1 : 0;
  d.root_px = (vreal)F("body_pos")[3], d.root_py = (vreal)F("body_pos")[4], d.root_pz = (vreal)F("body_pos")[5];
  d.gx = (vreal)F("gravity")[0], d.gy = (vreal)F("gravity")[1], d.gz = (vreal)F("gravity")[2];
  const auto& pn = F("plane_normal");
  const auto& pp = F("plane_pos");
  d.pnx = (vreal)pn[0], d.pny = (vreal)pn[1], d.pnz = (vreal)pn[2];
  d.ppx = (vreal)pp[0], d.ppy = (vreal)pp[1], d.ppz = (vreal)pp[2];
  {  // math.make_frame(n)[1]
    double a[3] = {pn[0], pn[1], pn[2]}, bb[3] = {0, 0, 0};
    if (-0.5 < a[1] && a[1] < 0.5) bb[1] = 1; else bb[2] = 1;
    double ab = a[0] * bb[0] + a[1] * bb[1] + a[2] * bb[2];
    for (int k = 0; k < 3; k++) bb[k] -= a[k] * ab;
    double nn = sqrt(bb[0] * bb[0] + bb[1] * bb[1] + bb[2] * bb[2]);
    d.t1x = (vreal)(bb[0] / nn), d.t1y = (vreal)(bb[1] / nn), d.t1z = (vreal)(bb[2] / nn);
  }
  double tm = 0;
  for (double x : F("body_mass")) tm += x;
  d.total_mass_inv = (vreal)(1.0 / tm);

  int rc;
#define UPF(name, vec) if ((rc = upload<vreal, double>(env, vec, &d.name)) != VNL_OK) return rc;
#define UPI(name, vec) if ((rc = upload<int, int>(env, vec, &d.name)) != VNL_OK) return rc;
  UPI(body_parent, I("body_parentid")) UPI(body_jntadr, I("body_jntadr")) UPI(body_jntnum, I("body_jntnum"))
  UPI(out_dyn, fmap.out_dyn) UPI(body_out, fmap.body_out) UPF(out_pos, fmap.out_pos) UPF(out_quat, fmap.out_quat)
  {
    std::vector<int> da(nb, 0), dn(nb, 0);
    for (int b = 0; b < nb; b++) {
      dn[b] = I("body_dofnum")[b];
      da[b] = dn[b] > 0 ? I("body_dofadr")[b] : 0;
    }
    UPI(body_dofadr, da) UPI(body_dofnum, dn)
    // subtree sizes (bodies are numbered depth-first: subtree(b) = b .. b + nsub[b]) and the
    // pointer-jumping tables jump[r][b] = 2^r-th ancestor of b (0 when it would be the world body)
    const auto& bp = I("body_parentid");
    std::vector<int> nsub(nb, 0);
    for (int b = nb - 1; b > 0; b--) nsub[bp[b]] += nsub[b] + 1;
    for (int b = 1; b < nb; b++)
      for (int c = b + 1; c <= b + nsub[b]; c++) {
        bool in = false;
        for (int a = c; a > 0; a = bp[a]) in |= (a == b);
        if (!in) return fail(VNL_ERR_UNSUPPORTED, "body numbering is not depth-first");
      }
    UPI(body_nsub, nsub)
    nsub_host = nsub;
    std::vector<int> lastdof(nb, nv);  // last dof on the path from the root to (and including) body b; nv = none
    for (int b = 1; b < nb; b++) lastdof[b] = dn[b] > 0 ? da[b] + dn[b] - 1 : lastdof[bp[b]];
    UPI(body_lastdof, lastdof)
    // The dofs on the path root -> body b as (at most 4) runs of consecutive dof indices, begin | end << 8 each:
    // jac_mul sums a body twist as differences of ONE prefix-sum array over the dofs instead of walking the path.
    {
      const auto& dpar = I("dof_parentid");
      std::vector<int> seg(8 * (size_t)nb, 0);  // per body: 4 runs of dofs, then 4 runs of ancestor bodies (incl. itself)
      int most = 1;
      for (int b = 1; b < nb; b++) {
        {
          std::vector<int> anc;
          for (int a = b; a > 0; a = bp[a]) anc.push_back(a);
          std::sort(anc.begin(), anc.end());
          int nseg = 0;
          for (size_t i = 0; i < anc.size();) {
            size_t j = i;
            while (j + 1 < anc.size() && anc[j + 1] == anc[j] + 1) j++;
            if (nseg == 4) return fail(VNL_ERR_UNSUPPORTED, "a body has more than 4 runs of consecutive ancestor bodies");
            seg[8 * b + 4 + nseg++] = anc[i] | ((anc[j] + 1) << 8);
            i = j + 1;
          }
          most = nseg > most ? nseg : most;
        }
        std::vector<int> path;
        for (int dd = lastdof[b] < nv ? lastdof[b] : -1; dd >= 0; dd = dpar[dd]) path.push_back(dd);
        std::sort(path.begin(), path.end());
        int nseg = 0;
        for (size_t i = 0; i < path.size();) {
          size_t j = i;
          while (j + 1 < path.size() && path[j + 1] == path[j] + 1) j++;
          if (nseg == 4) return fail(VNL_ERR_UNSUPPORTED, "a body path has more than 4 runs of consecutive dofs");
          seg[8 * b + nseg++] = path[i] | ((path[j] + 1) << 8);
          i = j + 1;
        }
        most = nseg > most ? nseg : most;
      }
      d.path_runs = most;
      UPI(body_pathseg, seg)
    }
    int bdepth = 0;
    for (int b = 1; b < nb; b++) {
      int dd = 0;
      for (int a = b; a > 0; a = bp[a]) dd++;
      bdepth = dd > bdepth ? dd : bdepth;
    }
    int rounds = 0;
    while ((1 << rounds) < bdepth) rounds++;
    d.jump_rounds = rounds;
    std::vector<unsigned char> jump((size_t)(rounds > 0 ? rounds : 1) * nb, 0);
    for (int b = 0; b < nb; b++) jump[b] = (unsigned char)bp[b];
    for (int r = 1; r < rounds; r++)
      for (int b = 0; b < nb; b++) jump[(size_t)r * nb + b] = jump[(size_t)(r - 1) * nb + jump[(size_t)(r - 1) * nb + b]];
    {
      const unsigned char* dp = nullptr;
      if ((rc = upload<unsigned char, unsigned char>(env, jump, &dp)) != VNL_OK) return rc;
      d.jump = dp;
    }
  }
  UPF(body_pos, F("body_pos")) UPF(body_quat, F("body_quat")) UPF(body_ipos, F("body_ipos")) UPF(body_mass, F("body_mass"))
  {
    std::vector<double> i6(6 * (size_t)nb);
    const auto& f9 = F("body_inertia_full");
    for (int b = 0; b < nb; b++) {
      const double* s = &f9[9 * (size_t)b];
      double* o = &i6[6 * (size_t)b];
      o[0] = s[0], o[1] = s[4], o[2] = s[8], o[3] = s[1], o[4] = s[2], o[5] = s[5];
    }
    UPF(body_inertia6, i6)
  }
  UPI(jnt_type, jt) UPI(jnt_qposadr, I("jnt_qposadr")) UPI(jnt_dofadr, I("jnt_dofadr")) UPI(jnt_body, I("jnt_bodyid"))
  UPF(jnt_pos, F("jnt_pos")) UPF(jnt_axis, F("jnt_axis")) UPF(jnt_stiffness, F("jnt_stiffness"))
  {
    std::vector<double> q0(nj), qs(nj);
    for (int j = 0; j < nj; j++) q0[j] = F("qpos0")[I("jnt_qposadr")[j]], qs[j] = F("qpos_spring")[I("jnt_qposadr")[j]];
    UPF(jnt_qpos0, q0) UPF(jnt_springref, qs)
  }
  {  // limit rows
    std::vector<int> qa, dof;
    std::vector<double> lo, hi, mg, iw, sr, si;
    for (int j = 0; j < nj; j++) {
      if (!I("jnt_limited")[j] || jt[j] != VNL_JNT_HINGE) continue;
      qa.push_back(I("jnt_qposadr")[j]), dof.push_back(I("jnt_dofadr")[j]);
      lo.push_back(F("jnt_range")[2 * j]), hi.push_back(F("jnt_range")[2 * j + 1]), mg.push_back(F("jnt_margin")[j]);
      iw.push_back(F("dof_invweight0")[I("jnt_dofadr")[j]]);
      for (int k = 0; k < 2; k++) sr.push_back(F("jnt_solref")[2 * j + k]);
      for (int k = 0; k < 5; k++) si.push_back(F("jnt_solimp")[5 * j + k]);
    }
    if ((int)qa.size() != d.nlimit) return fail(VNL_ERR_BLOB, "nlimit does not match jnt_limited");
    UPI(lim_qadr, qa) UPI(lim_dof, dof) UPF(lim_lo, lo) UPF(lim_hi, hi) UPF(lim_margin, mg) UPF(lim_invweight, iw)
    UPF(lim_solref, sr) UPF(lim_solimp, si)
  }
  {  // tree-sparse layout of qM (MuJoCo dof_Madr order: self, parent, grandparent, ...)
    const auto& par = I("dof_parentid");
    std::vector<int> madr(nv), depth(nv), anc;
    for (int i = 0; i < nv; i++) {
      madr[i] = (int)anc.size();
      int dep = 0;
      for (int j = i; j >= 0; j = par[j]) anc.push_back(j), dep++;
      depth[i] = dep - 1;
    }
    d.nM = (int)anc.size();
    std::vector<int> row(anc.size());
    for (int i = 0; i < nv; i++)
      for (int a = 0; a <= depth[i]; a++) row[madr[i] + a] = i;
    // descendants of a dof are the next ndesc dofs (DFS numbering); checked here because col_apply relies on it
    std::vector<int> ndesc(nv, 0);
    for (int i = 0; i < nv; i++)
      for (int j = par[i]; j >= 0; j = par[j]) ndesc[j]++;
    for (int a = 0; a < nv; a++)
      for (int i = a + 1; i <= a + ndesc[a]; i++) {
        bool is_desc = false;
        for (int j = par[i]; j >= 0; j = par[j]) is_desc |= (j == a);
        if (!is_desc) return fail(VNL_ERR_UNSUPPORTED, "dof numbering is not depth-first");
      }
    int maxd = 0;
    for (int i = 0; i < nv; i++) maxd = depth[i] > maxd ? depth[i] : maxd;
    d.max_depth = maxd;
    if (nv > 255 || nb > 255 || maxd > 62) return fail(VNL_ERR_UNSUPPORTED, "model too large for 8-bit index tables");
    std::vector<unsigned char> lvl_tab;  // dofs sorted by depth, then the level start offsets
    std::vector<int> lvl_start(maxd + 2, 0);
    for (int lev = 0; lev <= maxd; lev++) {
      lvl_start[lev] = (int)lvl_tab.size();
      int cnt = 0;
      for (int i = 0; i < nv; i++)
        if (depth[i] == lev) lvl_tab.push_back((unsigned char)i), cnt++;
      if (cnt * lev > 2 * nv) return fail(VNL_ERR_UNSUPPORTED, "tree level too wide for the inversion staging buffer");
    }
    lvl_start[maxd + 1] = (int)lvl_tab.size();
    for (int v : lvl_start) lvl_tab.push_back((unsigned char)v);
    {
      const unsigned char* dp = nullptr;
      if ((rc = upload<unsigned char, unsigned char>(env, lvl_tab, &dp)) != VNL_OK) return rc;
      d.lvl_tab = dp;
    }
    std::vector<int> limrow(nv, -1);
    {
      int r = 0;
      for (int j = 0; j < nj; j++)
        if (I("jnt_limited")[j] && jt[j] == VNL_JNT_HINGE) limrow[I("jnt_dofadr")[j]] = r++;
    }
    UPI(dof_body, I("dof_bodyid")) UPI(dof_Madr, madr) UPI(dof_depth, depth) UPI(M_anc, anc) UPI(M_row, row)
    UPI(dof_ndesc, ndesc)
    limrow_host = limrow;
    // Factorisation schedule: row j can be the pivot once all its descendants have been; rows whose
    // subtrees are disjoint go in the same step (at most VNL_FAC_LINES of them, one scratch line each).
    std::vector<int> ftime(nv, 0), fslot(nv, 0), count;
    for (int j = nv - 1; j >= 0; j--) {  // children have larger indices: their times are final here
      int t = 0;
      for (int i = j + 1; i <= j + ndesc[j]; i++)
        if (par[i] == j) t = ftime[i] + 1 > t ? ftime[i] + 1 : t;
      while ((int)count.size() <= t) count.push_back(0);
      while (count[t] >= VNL_FAC_LINES) {
        t++;
        if ((int)count.size() <= t) count.push_back(0);
      }
      ftime[j] = t, fslot[j] = count[t]++;
    }
    // a parent must come strictly after each child even when the child was pushed to a later step
    for (int j = nv - 1; j >= 0; j--)
      if (par[j] >= 0 && ftime[par[j]] <= ftime[j]) return fail(VNL_ERR_UNSUPPORTED, "factorisation schedule is not causal");
    d.fac_steps = (int)count.size();
    {  // leaves of the dof tree, for the forward substitution of factor_rows<.., true>
      std::vector<int> leaf_id(nv, -1);
      int nleaf = 0;
      for (int j = 0; j < nv; j++)
        if (ndesc[j] == 0) leaf_id[j] = nleaf++;
      d.fac_nleaf = nleaf <= VNL_FAC_LINES ? nleaf : 0;
      int deep1 = 0;  // deepest row of the second lane set (rows 64 ..): bits 8.. of fac_nleaf
      for (int j = 64; j < nv; j++) deep1 = depth[j] > deep1 ? depth[j] : deep1;
      if (d.fac_nleaf)
        for (int a = 0; a < nv; a++) {
          int mask = 0;
          for (int j = a; j <= a + ndesc[a]; j++)
            if (leaf_id[j] >= 0) mask |= 1 << leaf_id[j];
          fslot[a] |= (leaf_id[a + ndesc[a]] << 8) | (mask << 16);  // the last descendant is a leaf
        }
      d.fac_nleaf |= deep1 << 8;
    }
    UPI(dof_ftime, ftime) UPI(dof_fslot, fslot)
    {  // guests of invert_aba: rows 64.. placed with lanes whose own row is shallow (deepest guests first), so that own + guest <= 36 entries
      d.fac_guest = nullptr;
      if (nv > 64 && nv <= 128) {
        std::vector<int> guest(64, -1), order;
        for (int j = 64; j < nv; j++) order.push_back(j);
        std::sort(order.begin(), order.end(), [&](int x, int y) { return depth[x] > depth[y]; });
        bool ok = true;
        for (int j : order) {
          int host = -1;
          for (int l = 0; l < 64 && host < 0; l++)
            if (guest[l] < 0 && depth[l] <= 12) host = l;
          if (host < 0 || depth[j] > 24) ok = false;
          else guest[host] = j;
        }
        if (ok) UPI(fac_guest, guest)
      }
    }
    {  // fac_match[a][t]: the scratch lines whose pivot of step t lies strictly below row a (factor_aba absorbs them)
      const int nst = d.fac_steps;
      std::vector<unsigned char> match((size_t)nv * (nst > 0 ? nst : 1), 0);
      for (int j = 0; j < nv; j++)
        for (int a = par[j]; a >= 0; a = par[a]) match[(size_t)a * nst + ftime[j]] |= (unsigned char)(1u << (fslot[j] & 0xff));
      const unsigned char* dp = nullptr;
      if ((rc = upload<unsigned char, unsigned char>(env, match, &dp)) != VNL_OK) return rc;
      d.fac_match = dp;
    }
    {  // EnvWave::blk_apply: the sparse products with the factor / its inverse cut into blocks of <= 8 entries of ONE row (or
       // column), dealt out so that every lane has about the same number -- the rows of a tree are as long as the dofs are deep
       // (0 .. 35 for the rodent) and its columns as long as the subtrees are large (0 .. 72): a lane per row / column
       // makes the wave wait for the longest.  The blocks of a row sit in adjacent lanes of one 16-lane DPP row (their partial
       // sums are combined by shifts), `trips` x 64 descriptors per form:
       //   bits 0-15 payload (row form: index of the block's first entry; column form: first descendant | depth of the column << 7)
       //   16-19 entries in the block | 20-26 row / column | 27 first block of its row | 28-31 lane + 1, 2, 4, 8 continues the row
      d.blk_tab = nullptr, d.blk_cfg = 0;
      auto pack = [&](bool colform, std::vector<unsigned>& out, int& trips, int& steps) -> bool {
        struct Item { int r, nblk; };
        std::vector<Item> items;
        int total = 0, longest = 1;
        for (int r = 0; r < nv; r++) {
          const int len = colform ? ndesc[r] : depth[r], nb_ = len > 0 ? (len + VNL_BLK_W - 1) / VNL_BLK_W : 1;
          if (nb_ > 16) return false;
          items.push_back({r, nb_}), total += nb_, longest = nb_ > longest ? nb_ : longest;
        }
        std::stable_sort(items.begin(), items.end(), [](const Item& a, const Item& b) { return a.nblk > b.nblk; });
        steps = longest > 8 ? 4 : (longest > 4 ? 3 : (longest > 2 ? 2 : (longest > 1 ? 1 : 0)));
        for (trips = (total + 63) / 64; trips <= 8; trips++) {
          std::vector<int> fill(trips * 4, 0);
          std::vector<unsigned> tab((size_t)trips * 64, 0u);
          bool ok = true;
          for (const Item& it : items) {
            int bin = -1;
            for (int b = 0; b < trips * 4 && bin < 0; b++)
              if (fill[b] + it.nblk <= 16) bin = b;
            if (bin < 0) { ok = false; break; }
            const int len = colform ? ndesc[it.r] : depth[it.r];
            for (int k = 0; k < it.nblk; k++) {
              const int slot = bin * 16 + fill[bin] + k, left = it.nblk - 1 - k;  // blocks of the row after this one
              const int n = len - VNL_BLK_W * k > VNL_BLK_W ? VNL_BLK_W : (len - VNL_BLK_W * k > 0 ? len - VNL_BLK_W * k : 0);
              unsigned payload;
              if (colform) payload = (unsigned)(n > 0 ? it.r + 1 + VNL_BLK_W * k : it.r) | ((unsigned)depth[it.r] << 7);
              else payload = (unsigned)(madr[it.r] + (n > 0 ? 1 + VNL_BLK_W * k : 0));
              unsigned mask = 0;
              for (int b = 0; b < 4; b++)
                if ((1 << b) <= left) mask |= 1u << b;
              tab[slot] = payload | ((unsigned)n << 16) | ((unsigned)it.r << 20) | ((k == 0 ? 1u : 0u) << 27) | (mask << 28);
            }
            fill[bin] += it.nblk;
          }
          if (ok) {
            out = tab;
            return true;
          }
        }
        return false;
      };
      std::vector<unsigned> rowtab, coltab;
      int tr = 0, tc = 0, sr = 0, sc = 0;
      if (nv <= 127 && maxd <= 63 && d.nM < 65536 && pack(false, rowtab, tr, sr) && pack(true, coltab, tc, sc)) {
        rowtab.insert(rowtab.end(), coltab.begin(), coltab.end());
        const unsigned* dp = nullptr;
        if ((rc = upload<unsigned, unsigned>(env, rowtab, &dp)) != VNL_OK) return rc;
        d.blk_tab = dp;
        d.blk_cfg = tr | (tc << 4) | (sr << 8) | (sc << 12);
      }
    }
  }
  UPF(dof_armature, F("dof_armature")) UPF(dof_damping, F("dof_damping"))
  UPI(act_dof, I("act_dof")) UPI(act_limited, I("act_ctrllimited")) UPF(act_gain, F("act_gain"))
  {  // per dof: the actuators that drive it (smooth_forces gathers instead of walking all actuators per dof)
    std::vector<int> packed(nv, 0), cnt(nv, 0);
    bool ok = nu < 255;
    for (int a = 0; a < nu && ok; a++) {
      const int dd = I("act_dof")[a];
      if (dd < 0 || dd >= nv) continue;
      if (cnt[dd] == 4) ok = false;
      else packed[dd] |= (a + 1) << (8 * cnt[dd]++);
    }
    d.dof_act = nullptr;
    if (ok) UPI(dof_act, packed)
  }
  UPF(act_tau, F("act_tau")) UPF(act_gear, F("act_gear"))
  {
    std::vector<double> lo(nu), hi(nu);
    for (int k = 0; k < nu; k++) lo[k] = F("act_ctrlrange")[2 * k], hi[k] = F("act_ctrlrange")[2 * k + 1];
    UPF(act_lo, lo) UPF(act_hi, hi)
  }
  UPI(cg_type, I("cg_type")) UPI(cg_body, I("cg_bodyid")) UPI(cg_conadr, I("cg_conadr")) UPI(cg_ncon, I("cg_ncon"))
  {
    std::vector<int> cgeom(d.ncon, 0);
    for (int g = 0; g < ng; g++)
      for (int q = 0; q < I("cg_ncon")[g]; q++) cgeom[I("cg_conadr")[g] + q] = g;
    // The contacts under a dof (subtree of its body = contiguous body range) form ONE range [c0, c1) of the
    // contacts SORTED BY BODY: con_geom[k] carries the geom of contact k in its low byte and, in the next byte, the
    // k-th contact in body order; the range is packed next to the dof's limit row as
    // (limrow + 1) | c0 << 10 | c1 << 18 | valid << 26 (constraint_force sums wrenches by prefix differences).
    const auto& cgb = I("cg_bodyid");
    const bool ok_ = d.ncon <= 64 && d.nlimit < 1023 && ng < 256;
    if (!ok_) return fail(VNL_ERR_UNSUPPORTED, "more than 64 contacts / 255 collision geoms / 1022 limit rows (constraint_force scans the contact wrenches in one wave)");
    std::vector<int> order(d.ncon);
    for (int c = 0; c < d.ncon; c++) order[c] = c;
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return cgb[cgeom[a]] < cgb[cgeom[b]]; });
    if (ok_)
      for (int k = 0; k < d.ncon; k++) cgeom[k] |= order[k] << 8;
    UPI(con_geom, cgeom)
    std::vector<int> packed(nv);
    for (int dd = 0; dd < nv; dd++) {
      int b0 = I("dof_bodyid")[dd], b1 = b0 + nsub_host[b0], c0 = 0, c1 = 0;
      while (c0 < d.ncon && cgb[cgeom[order[c0]] & 0xff] < b0) c0++;
      c1 = c0;
      while (c1 < d.ncon && cgb[cgeom[order[c1]] & 0xff] <= b1) c1++;
      packed[dd] = (limrow_host[dd] + 1) | (c0 << 10) | (c1 << 18) | ((ok_ ? 1 : 0) << 26);
    }
    UPI(dof_limrow, packed)
  }
  UPF(cg_pos, F("cg_pos")) UPF(cg_quat, F("cg_quat")) UPF(cg_size, F("cg_size")) UPF(cg_solref, F("cg_solref"))
  UPF(cg_solimp, F("cg_solimp")) UPF(cg_margin, F("cg_margin"))
  {
    std::vector<double> mu(ng), iw(ng);
    double impratio = S("impratio");
    for (int g = 0; g < ng; g++) {
      mu[g] = F("cg_friction")[3 * g];
      double t = F("body_invweight0")[0] + (hm->f.count("cg_body_invweight0") ? F("cg_body_invweight0")[g]
                                                                                : F("body_invweight0")[2 * (size_t)I("cg_bodyid")[g]]);
      iw[g] = (t + mu[g] * mu[g] * t) * 2 * mu[g] * mu[g] / impratio;  // constraint._instantiate_contact
    }
    UPF(cg_mu, mu) UPF(cg_invweight, iw)
  }
#undef UPF
#undef UPI
  return VNL_OK;
}

static VnlDims dims_of(const DevModel& d) {
  return VnlDims{d.nq, d.nv, d.nu, d.nbody, d.njnt, d.ncg, d.ncon, d.nlimit, d.nefc, d.nM, d.iterations, d.ls_iterations, d.eulerdamp,
                 d.root_free, d.max_depth, d.jump_rounds, d.fac_steps, d.fac_nleaf, d.solver_newton, d.blk_cfg, d.path_runs, d.nbody_out};
}

static void layout(vnl_env* env) {
  const DevModel& d = env->dm;
  const WsLayout L = vnl_make_layout(dims_of(d));  // (csrc/vnl_types.h: the same function gives a specialised kernel its constants)
  env->L = L;
  auto sec = [&](const char* name, int at, int n) { env->sections[name] = {at, n}; };
  sec("qpos", L.qpos, d.nq), sec("qvel", L.qvel, d.nv), sec("act", L.act, d.nu), sec("ctrl", L.ctrl, d.nu);
  sec("act_dot", L.actdot, d.nu), sec("subtree_com1", L.com, 4), sec("cdof", L.cdof, 6 * d.nv);
  sec("qLD", L.LD, L.dinv - L.LD), sec("qLDiagInv", L.dinv, d.nv), sec("pool", L.P, L.smooth - L.P);
  sec("efc_D", L.efc_D, d.nefc), sec("Jaref", L.Jaref, d.nefc), sec("jv", L.jv, d.nefc);
  sec("qfrc_smooth", L.smooth, d.nv), sec("qacc_smooth", L.qacc_smooth, d.nv), sec("qacc", L.qacc, d.nv);
  sec("Ma", L.Ma, d.nv), sec("grad", L.grad, d.nv), sec("Mgrad", L.Mgrad, d.nv), sec("search", L.search, d.nv);
  sec("mv", L.mv, d.nv), sec("qfrc_constraint", L.qfrc_c, d.nv), sec("tmp", L.tmp, d.nv), sec("tmp2", L.tmp2, d.nv);
  sec("con_r", L.con_r, 3 * d.ncon), sec("con_t1", L.con_t1, 3 * d.ncg);
  sec("tab_anc", L.tab_anc, L.tab_madr - L.tab_anc), sec("tab_madr", L.tab_madr, L.tab_body - L.tab_madr);
  sec("tab_body", L.tab_body, L.tab_jump - L.tab_body), sec("tab_jump", L.tab_jump, L.tab_lvl - L.tab_jump);
  sec("tab_lvl", L.tab_lvl, L.act_list - L.tab_lvl);
  sec("act_list", L.act_list, vnl_words(4 * (long)((d.ncon + 3) / 4) + 8 + 2 * VNL_LIVE_MAX));
  if (d.solver_newton) {
    sec("newton_qM", L.newt_M, d.nv * d.nv), sec("newton_H", L.newt_H, d.nv * d.nv), sec("newton_efc_J", L.newt_J, d.nefc * d.nv);
  }
#ifdef VNL_PROFILE
  sec("prof", L.prof, 2 * (VNL_NPROF + 1));
#endif
}

static bool same_layout(const WsLayout& a, const WsLayout& b) { return memcmp(&a, &b, sizeof(WsLayout)) == 0; }
static bool same_dims(const VnlDims& a, const VnlDims& b) { return memcmp(&a, &b, sizeof(VnlDims)) == 0; }

// (VNL_KERNEL_ATTR: empty in the product; csrc/build.py --spill sets a VGPR cap to force register spills to scratch,
// the regression build for the "results must not depend on spilling" test.  VNL_SPEC_ATTR: the specialised instantiations
// are held to the two waves per SIMD of the generic kernel -- with every bound a constant the compiler unrolls further and
// would take a 257th register, i.e. half the occupancy)
#ifndef VNL_KERNEL_ATTR
#define VNL_KERNEL_ATTR
#endif
#define VNL_ENV_KERNEL __launch_bounds__(64) VNL_KERNEL_ATTR
template <class SP>
__global__ void VNL_ENV_KERNEL vnl_step_kernel(const KernelConsts* kc, DevState st, const vreal* action, vreal* dump, vreal* dump_mid,
                                               int* trace);

extern "C" void vnl_env_destroy(vnl_env* env) {
  if (!env) return;
  for (void* p : env->allocs) (void)hipFree(p);
  delete env;
}

extern "C" int vnl_env_create(const vnl_model* hm, const vnl_envspec* es, int32_t num_envs, int32_t device,
                              vnl_env** out) {
  if (!hm || !es || !out || num_envs <= 0) return fail(VNL_ERR_ARG, "vnl_env_create: bad argument");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    return fail(VNL_ERR_NO_DEVICE, "no HIP device: the rollout has no CPU fallback");
  if (device < 0 || device >= ndev) return fail(VNL_ERR_ARG, "device ordinal out of range");
  DeviceGuard guard(device);  // the caller's current device is restored on every return path
  if (!guard.ok) return fail(VNL_ERR_HIP, "hipSetDevice failed");
  vnl_env* env = new vnl_env();
  env->device = device, env->B = num_envs;
  int rc = build_dev_model(env, hm);
  if (rc != VNL_OK) {
    vnl_env_destroy(env);
    return rc;
  }
  const DevModel& d = env->dm;
  DevEnv& e = env->de;
  e.T = es->clip_frames, e.C = es->num_clips, e.ref_len = es->ref_traj_length, e.sub_clip_length = es->sub_clip_length;
  e.n_frames = es->n_frames, e.nb = es->num_track_bodies, e.nee = es->num_end_eff, e.napp = es->num_appendages;
  e.njc = es->num_joint_cols, e.com_ref_col = es->com_ref_col;
  e.healthy_lo = es->healthy_z_lo, e.healthy_hi = es->healthy_z_hi;
  e.inv_term_threshold = vreal(1) / es->termination_threshold, e.body_err_mult = es->body_error_multiplier;
  e.flags = es->flags, e.done_threshold = es->done_threshold, e.center_of_mass = nullptr;
  {
    const double builtin[6] = {0.01, 0.01, 0.01, 0.01, 0.0001, 0.01};  // rodent.py:203-209
    for (int k = 0; k < 6; k++) e.w_reward[k] = (es->flags & VNL_ENV_WEIGHTS) ? (vreal)es->reward_weights[k] : (vreal)builtin[k];
  }
  e.obs_size = (e.flags & VNL_ENV_OBS_QPOS_QVEL) ? d.nq + d.nv : d.nq + 2 * d.nv + 3 * e.nee;
  e.traj_size = e.ref_len * (3 * e.napp + 6 * e.nb + 3 + e.njc);
  bool ok = e.T >= e.ref_len && e.C >= 1 && e.nb >= 1 && e.com_ref_col >= 0 && e.com_ref_col < e.nb && d.nq >= 7 &&
            d.root_free;
  for (int k = 0; ok && k < e.nb; k++) ok = es->body_idxs[k] >= 0 && es->body_idxs[k] < d.nbody_out;
  for (int k = 0; ok && k < e.nee; k++) ok = es->end_eff_idx[k] >= 0 && es->end_eff_idx[k] < d.nbody_out;
  for (int k = 0; ok && k < e.napp; k++)
    ok = es->app_body[k] >= 0 && es->app_body[k] < d.nbody_out && es->app_ref_col[k] >= 0 && es->app_ref_col[k] < e.nb;
  for (int k = 0; ok && k < e.njc; k++) ok = es->joint_cols[k] >= 0 && es->joint_cols[k] < d.nq - 7;
  if (!ok) {
    vnl_env_destroy(env);
    return fail(VNL_ERR_ARG, "vnl_envspec: index out of range or unsupported shape");
  }
  size_t CT = (size_t)e.C * e.T, nj = d.nq - 7;
#define UP(call)              \
  if ((rc = (call)) != VNL_OK) { \
    vnl_env_destroy(env);     \
    return rc;                \
  }
  UP(upload_raw<int>(env, es->body_idxs, e.nb, &e.body_idxs))
  UP(upload_raw<int>(env, es->end_eff_idx, e.nee, &e.end_eff_idx))
  UP(upload_raw<int>(env, es->app_body, e.napp, &e.app_body))
  UP(upload_raw<int>(env, es->app_ref_col, e.napp, &e.app_ref_col))
  UP(upload_raw<int>(env, es->joint_cols, e.njc, &e.joint_cols))
  UP(upload_raw<float>(env, es->position, CT * 3, &e.position))
  UP(upload_raw<float>(env, es->quaternion, CT * 4, &e.quaternion))
  UP(upload_raw<float>(env, es->joints, CT * nj, &e.joints))
  UP(upload_raw<float>(env, es->body_positions, CT * e.nb * 3, &e.body_positions))
  UP(upload_raw<float>(env, es->velocity, CT * 3, &e.velocity))
  UP(upload_raw<float>(env, es->angular_velocity, CT * 3, &e.angular_velocity))
  UP(upload_raw<float>(env, es->joints_velocity, CT * nj, &e.joints_velocity))
  if (es->center_of_mass) UP(upload_raw<float>(env, es->center_of_mass, CT * 3, &e.center_of_mass))
#undef UP
  {
    void* p = nullptr;
    HIPCHK(hipMalloc(&p, (size_t)num_envs * (d.nM + d.nv) * sizeof(vreal)));
    env->allocs.push_back(p);
    e.fac2 = (vreal*)p;
  }
  layout(env);
  // a model whose every dimension and LDS offset equals the compile-time constants of a specialised kernel runs that kernel
  env->spec = (same_dims(dims_of(env->dm), VnlSpecRodent::D) && same_layout(env->L, VnlSpecRodent::L)) ? 1 : 0;
#ifdef VNL_NO_SPEC
  env->spec = 0;
#endif
  env->lds_bytes = (size_t)env->L.total * sizeof(vreal);
#ifdef VNL_STAGE_KNOBS
  if (const char* padk = getenv("VNL_DBG_LDS_BYTES")) {  // occupancy experiments only: force a larger LDS request
    size_t want = (size_t)atol(padk);
    if (want > env->lds_bytes) env->lds_bytes = want;
  }
#endif
  if (6 * d.nbody > 512) {
    vnl_env_destroy(env);
    return fail(VNL_ERR_UNSUPPORTED, "more than 85 dynamic bodies (the per-body scans of bias_forces keep 8 elements per lane)");
  }
  if (d.nu > d.nv) {
    vnl_env_destroy(env);
    return fail(VNL_ERR_UNSUPPORTED, "more actuators than dofs (the actuator forces are staged in two dof vectors)");
  }
  if (d.nefc > 512) {
    vnl_env_destroy(env);
    return fail(VNL_ERR_UNSUPPORTED, "more than 512 constraint rows (line-search rows are register-resident)");
  }
  if (env->lds_bytes > 64 * 1024) {
    vnl_env_destroy(env);
    return fail(VNL_ERR_UNSUPPORTED, "model too large: per-env working set exceeds 64 KB of LDS");
  }
  {
    int nb_ = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb_, vnl_step_kernel<VnlSpecGeneric>, 64, env->lds_bytes) == hipSuccess)
      env->blocks_per_cu = nb_;
  }
  {
    KernelConsts host{env->dm, env->de, env->L};
    void* p = nullptr;
    HIPCHK(hipMalloc(&p, sizeof(KernelConsts)));
    env->allocs.push_back(p);
    HIPCHK(hipMemcpy(p, &host, sizeof(KernelConsts), hipMemcpyHostToDevice));
    env->kc = (KernelConsts*)p;
  }
  *out = env;
  return VNL_OK;
}

extern "C" int vnl_env_dims(const vnl_env* env, vnl_dims* o) {
  if (!env || !o) return fail(VNL_ERR_ARG, "vnl_env_dims: null argument");
  const DevModel& d = env->dm;
  o->nq = d.nq, o->nv = d.nv, o->nu = d.nu, o->nbody = d.nbody_out, o->njnt = d.njnt, o->ngeom_collide = d.ncg;
  o->ncon = d.ncon, o->nefc = d.nefc, o->obs_size = env->de.obs_size, o->traj_size = env->de.traj_size;
  o->workspace_floats_per_env = env->L.total;
  o->workgroups_per_cu = env->blocks_per_cu;
  o->nbody_dynamic = d.nbody, o->kernel_specialised = env->spec;
  return VNL_OK;
}

extern "C" int vnl_env_scratch(const vnl_env* env, const char* name, float** dev_ptr, int32_t* count) {
  if (!env || !name || !dev_ptr || !count) return fail(VNL_ERR_ARG, "vnl_env_scratch: null argument");
  if (!env->dump) return fail(VNL_ERR_ARG, "scratch dump is off: call vnl_env_debug(env, 1) before reset/step");
  if (strcmp(name, "solver_trace") == 0) {  // int32 [num_envs][n_frames][VNL_TRACE_INTS], contiguous (not inside the image)
    *dev_ptr = (float*)env->trace;
    *count = (env->de.n_frames > 0 ? env->de.n_frames : 1) * VNL_TRACE_INTS;
    return VNL_OK;
  }
  auto it = env->sections.find(name);
  if (it == env->sections.end()) return fail(VNL_ERR_ARG, "unknown scratch section %s", name);
  *dev_ptr = (float*)(env->dump + it->second.first);
  *count = it->second.second;
  return VNL_OK;
}

extern "C" int vnl_env_debug(vnl_env* env, int32_t enable, int32_t* row_stride) {
  if (!env) return fail(VNL_ERR_ARG, "vnl_env_debug: null argument");
  if (enable < 0 || enable > 2) return fail(VNL_ERR_ARG, "vnl_env_debug: mode must be 0, 1 or 2");
  if (enable && !env->dump) {
    void* p = nullptr;
    HIPCHK(hipMalloc(&p, (size_t)env->L.total * env->B * sizeof(vreal)));
    env->allocs.push_back(p);
    env->dump = (vreal*)p;
    const size_t nt = (size_t)env->B * (env->de.n_frames > 0 ? env->de.n_frames : 1) * VNL_TRACE_INTS * sizeof(int);
    HIPCHK(hipMalloc(&p, nt));
    env->allocs.push_back(p);
    HIPCHK(hipMemset(p, 0, nt));
    env->trace = (int*)p;
  }
  env->debug = enable;
  if (row_stride) *row_stride = env->L.total;
  return VNL_OK;
}

// ----------------------------------------------------------------------------- kernels
// One env per 64-lane workgroup; the env's whole working set lives in dynamic LDS (~25 KB ->
// 6 workgroups per CU, 1536 envs in flight on 256 CUs).
template <class SP>
__global__ void VNL_ENV_KERNEL vnl_step_kernel(const KernelConsts* kc, DevState st, const vreal* action,
                                                      vreal* dump, vreal* dump_mid, int* trace) {
  VNL_LDS_DECL(lds);
  const VNL_CAS KernelConsts* k = VNL_TO_CAS(KernelConsts, kc);
  EnvWaveT<SP> w{k->m, k->ev, st, k->L, lds, blockIdx.x, threadIdx.x, k, nullptr};
  w.step(action, dump_mid, trace);
  if (dump) w.dump(dump);
}

template <class SP>
__global__ void VNL_ENV_KERNEL vnl_reset_kernel(const KernelConsts* kc, DevState st, const int* start_frame,
                                                       const vreal* noise, vreal* dump, int* trace) {
  VNL_LDS_DECL(lds);
  const VNL_CAS KernelConsts* k = VNL_TO_CAS(KernelConsts, kc);
  EnvWaveT<SP> w{k->m, k->ev, st, k->L, lds, blockIdx.x, threadIdx.x, k, nullptr};
  w.reset(start_frame, noise, trace);
  if (dump) w.dump(dump);
}

__global__ void __launch_bounds__(64) vnl_fk_kernel(const KernelConsts* kc, DevState st, const vreal* qpos) {
  VNL_LDS_DECL(lds);
  const VNL_CAS KernelConsts* k = VNL_TO_CAS(KernelConsts, kc);
  EnvWave w{k->m, k->ev, st, k->L, lds, blockIdx.x, threadIdx.x, k, nullptr};
  w.fk(qpos);
}

static int to_dev_state(const vnl_state* s, DevState* d) {
  const void* ptrs[] = {s->qpos, s->qvel, s->act, s->qacc_warmstart, s->xpos, s->xquat, s->subtree_com1,
                        s->qfrc_actuator, s->obs, s->reward, s->done, s->metrics, s->traj, s->termination_error,
                        s->cur_frame, s->sub_clip_frame, s->clip_id};
  for (const void* p : ptrs)
    if (!p) return fail(VNL_ERR_ARG, "vnl_state: null buffer");
  // (vreal is float in the product; the casts only matter for the float64 test build)
  d->qpos = (vreal*)s->qpos, d->qvel = (vreal*)s->qvel, d->act = (vreal*)s->act, d->warm = (vreal*)s->qacc_warmstart;
  d->xpos = (vreal*)s->xpos, d->xquat = (vreal*)s->xquat, d->com1 = (vreal*)s->subtree_com1;
  d->qfrc_actuator = (vreal*)s->qfrc_actuator, d->obs = (vreal*)s->obs, d->reward = (vreal*)s->reward;
  d->done = (vreal*)s->done, d->metrics = (vreal*)s->metrics, d->traj = (vreal*)s->traj;
  d->term_err = (vreal*)s->termination_error, d->cur_frame = s->cur_frame, d->sub_clip_frame = s->sub_clip_frame;
  d->clip_id = s->clip_id;
  return VNL_OK;
}

extern "C" int vnl_env_reset(vnl_env* env, const int32_t* start_frame, const float* noise, const vnl_state* state,
                             void* stream) {
  if (!env || !start_frame || !noise || !state) return fail(VNL_ERR_ARG, "vnl_env_reset: null argument");
  DevState ds;
  int rc = to_dev_state(state, &ds);
  if (rc != VNL_OK) return rc;
  DeviceGuard guard(env->device);  // the launch goes to the env's GPU whatever the caller's current device is
  if (!guard.ok) return fail(VNL_ERR_HIP, "hipSetDevice failed");
  if (env->spec)
    hipLaunchKernelGGL((vnl_reset_kernel<VnlSpecRodent>), dim3(env->B), dim3(64), env->lds_bytes, (hipStream_t)stream,
                       (const KernelConsts*)env->kc, ds, (const int*)start_frame, (const vreal*)noise,
                       env->debug ? env->dump : nullptr, env->debug ? env->trace : nullptr);
  else
    hipLaunchKernelGGL((vnl_reset_kernel<VnlSpecGeneric>), dim3(env->B), dim3(64), env->lds_bytes, (hipStream_t)stream,
                       (const KernelConsts*)env->kc, ds, (const int*)start_frame, (const vreal*)noise,
                       env->debug ? env->dump : nullptr, env->debug ? env->trace : nullptr);
  HIPCHK(hipGetLastError());
  return VNL_OK;
}

extern "C" int vnl_env_fk(vnl_env* env, const float* qpos, const vnl_state* state, void* stream) {
  if (!env || !qpos || !state) return fail(VNL_ERR_ARG, "vnl_env_fk: null argument");
  DevState ds;
  int rc = to_dev_state(state, &ds);
  if (rc != VNL_OK) return rc;
  DeviceGuard guard(env->device);
  if (!guard.ok) return fail(VNL_ERR_HIP, "hipSetDevice failed");
  hipLaunchKernelGGL(vnl_fk_kernel, dim3(env->B), dim3(64), env->lds_bytes, (hipStream_t)stream, (const KernelConsts*)env->kc, ds,
                     (const vreal*)qpos);
  HIPCHK(hipGetLastError());
  return VNL_OK;
}

extern "C" int vnl_env_step(vnl_env* env, const float* action, const vnl_state* state, void* stream) {
  if (!env || !action || !state) return fail(VNL_ERR_ARG, "vnl_env_step: null argument");
  DevState ds;
  int rc = to_dev_state(state, &ds);
  if (rc != VNL_OK) return rc;
  DeviceGuard guard(env->device);
  if (!guard.ok) return fail(VNL_ERR_HIP, "hipSetDevice failed");
  if (env->spec)
    hipLaunchKernelGGL((vnl_step_kernel<VnlSpecRodent>), dim3(env->B), dim3(64), env->lds_bytes, (hipStream_t)stream,
                       (const KernelConsts*)env->kc, ds, (const vreal*)action, env->debug == 1 ? env->dump : nullptr,
                       env->debug == 2 ? env->dump : nullptr, env->debug ? env->trace : nullptr);
  else
    hipLaunchKernelGGL((vnl_step_kernel<VnlSpecGeneric>), dim3(env->B), dim3(64), env->lds_bytes, (hipStream_t)stream,
                       (const KernelConsts*)env->kc, ds, (const vreal*)action, env->debug == 1 ? env->dump : nullptr,
                       env->debug == 2 ? env->dump : nullptr, env->debug ? env->trace : nullptr);
  HIPCHK(hipGetLastError());
  return VNL_OK;
}

#ifdef VNL_PROFILE
// diagnostic build only: read and clear the per-stage cycle sums
extern "C" int vnl_prof_read(unsigned long long* out) {
  unsigned long long z[VNL_NPROF] = {0};
  HIPCHK(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_vnl_prof), sizeof(z)));
  HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(g_vnl_prof), z, sizeof(z)));
  return VNL_OK;
}
#endif

// ---- rollout post-processing (include/vnl.h: vnl_rollout_post) ----------------------------------
// One workgroup per env: the episode / auto-reset decision is taken once, then every op moves its
// row (coalesced 32-bit words).  HBM-bound: ~5 KB per env.
#ifndef VNL_POST_THREADS
#define VNL_POST_THREADS 256
#endif
__global__ void __launch_bounds__(VNL_POST_THREADS) vnl_post_kernel(vnl_post_desc d) {
  const unsigned e = blockIdx.x, tid = threadIdx.x;
  // every thread evaluates the (cheap) decision itself from the OLD values; thread 0 stores the new ones
  // only after all threads have read them
  float st = d.steps ? ((d.prev_done && d.prev_done[e] != 0.f) ? 0.f : d.steps[e]) + (float)d.action_repeat : 0.f;
  float done = d.done[e];
  bool over = d.steps && st >= (float)d.episode_length;
  float trunc = over ? 1.f - done : 0.f;
  done = over ? 1.f : done;
  __syncthreads();
  if (tid == 0) {
    if (d.steps) d.steps[e] = st;
    if (d.prev_done) d.prev_done[e] = done;
    d.done[e] = done;
    if (d.truncation) d.truncation[e] = trunc;
    if (d.log_reward) d.log_reward[e] = d.reward[e];
    if (d.log_discount) d.log_discount[e] = 1.f - done;
    if (d.log_truncation) d.log_truncation[e] = trunc;
  }
  const bool reset = done != 0.f;
  for (int k = 0; k < d.num_ops; k++) {
    const vnl_post_op& op = d.ops[k];
    const bool from_first = op.first && reset;
    const unsigned* src = (const unsigned*)(from_first ? op.first : op.src) + (size_t)e * op.width;
    unsigned* dst = op.dst ? (unsigned*)op.dst + (size_t)e * op.width : nullptr;
    unsigned* log = op.log ? (unsigned*)op.log + (size_t)e * op.width : nullptr;
    const bool store = dst && (from_first || (const float*)op.dst != op.src);
    for (int i = (int)tid; i < op.width; i += VNL_POST_THREADS) {
      unsigned v = src[i];
      if (store) dst[i] = v;
      if (log) log[i] = v;
    }
  }
}

extern "C" int vnl_rollout_post(const vnl_post_desc* desc, int32_t num_envs, void* stream) {
  if (!desc || num_envs <= 0 || !desc->done) return fail(VNL_ERR_ARG, "vnl_rollout_post: null argument");
  if (desc->num_ops < 0 || desc->num_ops > VNL_POST_MAX_OPS) return fail(VNL_ERR_ARG, "vnl_rollout_post: too many ops");
  for (int k = 0; k < desc->num_ops; k++)
    if (!desc->ops[k].src || desc->ops[k].width <= 0) return fail(VNL_ERR_ARG, "vnl_rollout_post: bad op");
  if ((desc->log_reward && !desc->reward)) return fail(VNL_ERR_ARG, "vnl_rollout_post: log_reward without reward");
  hipLaunchKernelGGL(vnl_post_kernel, dim3(num_envs), dim3(VNL_POST_THREADS), 0, (hipStream_t)stream, *desc);
  HIPCHK(hipGetLastError());
  return VNL_OK;
}

// ---- PPO loss head (include/vnl.h: vnl_ppo_head) -------------------------------------------------
// Three launches on the caller's stream: (1) GAE, one column per thread, + advantage / reward statistics;
// (2) the per-sample head, 32 lanes per sample (lane = action component: coalesced rows, shuffle sums),
// per-block partial sums; (3) the partials summed in a fixed order -> metrics (deterministic).
#ifndef VNL_HEAD_THREADS
#define VNL_HEAD_THREADS 256
#define VNL_HEAD_GROUP 32
#define VNL_GROUP_SUM(v) \
  (v += __shfl_xor(v, 16, 32), v += __shfl_xor(v, 8, 32), v += __shfl_xor(v, 4, 32), v += __shfl_xor(v, 2, 32), v += __shfl_xor(v, 1, 32))
#endif
#define VNL_HEAD_MAXBLK 256
#ifndef VNL_FINISH_THREADS
#define VNL_FINISH_THREADS 64
#define VNL_FINISH_SUM(v) \
  (v += __shfl_xor(v, 32), v += __shfl_xor(v, 16), v += __shfl_xor(v, 8), v += __shfl_xor(v, 4), v += __shfl_xor(v, 2), v += __shfl_xor(v, 1))
#endif
__device__ __forceinline__ float vnl_block_sum(float v, float* red) {
  const unsigned tid = threadIdx.x;
  __syncthreads();
  red[tid] = v;
  __syncthreads();
  for (unsigned w = VNL_HEAD_THREADS / 2; w > 0; w >>= 1) {
    if (tid < w) red[tid] += red[tid + w];
    __syncthreads();
  }
  return red[0];
}
__device__ __forceinline__ float vnl_softplus(float x) { return x > 20.f ? x : log1pf(expf(x)); }  // torch F.softplus

// stats: [0] adv mean, [1] adv std (biased), [2] reward variance (biased), then VNL_HEAD_MAXBLK x 4 partial sums
__global__ void __launch_bounds__(VNL_HEAD_THREADS) vnl_ppo_gae_kernel(vnl_ppo_head_args a, float* stats) {
#if VNL_HEAD_THREADS > 1
  __shared__ float red[VNL_HEAD_THREADS];
#else
  float red[1];
#endif
  const int tid = (int)threadIdx.x, T = a.T, B = a.B, N = T * B;
  // compute_gae: reverse scan, intention_losses.py:63-87
  float sa = 0.f, sr = 0.f;
  for (int b = tid; b < B; b += VNL_HEAD_THREADS) {
    float acc = 0.f, v_next = a.bootstrap[b], vs_next = a.bootstrap[b];
    // eight time steps per trip: their 32 loads are in flight together, then the (sequential) recurrence runs on
    // registers -- one load latency per trip instead of one per time step (14 us -> a few for T = 20)
    for (int t1 = T; t1 > 0; t1 -= 8) {
      const int t0 = t1 > 8 ? t1 - 8 : 0;
      float tr[8], dc[8], rw[8], bl[8];
#pragma unroll
      for (int k = 0; k < 8; k++) {
        const int t = t1 - 1 - k, n = (t >= t0 ? t : t0) * B + b;
        tr[k] = a.truncation[n], dc[k] = a.discount[n], rw[k] = a.reward[n], bl[k] = a.baseline[n];
      }
#pragma unroll
      for (int k = 0; k < 8; k++) {
        const int t = t1 - 1 - k;
        if (t >= t0) {
          const int n = t * B + b;
          const float trunc = tr[k], mask = 1.f - trunc;
          const float term = (1.f - dc[k]) * (1.f - trunc);
          const float r = rw[k] * a.reward_scaling, v = bl[k];
          const float delta = (r + a.discounting * (1.f - term) * v_next - v) * mask;
          acc = delta + a.discounting * (1.f - term) * mask * a.gae_lambda * acc;
          const float vs = acc + v;
          const float adv = (r + a.discounting * (1.f - term) * vs_next - v) * mask;
          a.vs[n] = vs, a.advantages[n] = adv;
          sa += adv, sr += r;
          v_next = v, vs_next = vs;
        }
      }
    }
  }
  const float adv_mean = vnl_block_sum(sa, red) / (float)N, r_mean = vnl_block_sum(sr, red) / (float)N;
  float qa = 0.f, qr = 0.f;
  __syncthreads();
  const float inv_n = 1.f / (float)N;
  for (int n = tid; n < N; n += VNL_HEAD_THREADS) {
    const float da = a.advantages[n] - adv_mean, dr = a.reward[n] * a.reward_scaling - r_mean;
    qa += da * da, qr += dr * dr;
    // d v_loss / d baseline (v_loss = 0.25 mean((vs - baseline)^2), vs under stop_gradient: intention_losses.py:137-139) needs
    // nothing of the policy network: it is final here, so the value MLP's backward pass can start while the intention
    // network's forward is still running (csrc/vnl_ppo.hip)
    a.g_baseline[n] = -0.5f * (a.vs[n] - a.baseline[n]) * inv_n;
  }
  const float va = vnl_block_sum(qa, red) / (float)N, vr = vnl_block_sum(qr, red) / (float)N;
  if (tid == 0) stats[0] = adv_mean, stats[1] = sqrtf(va), stats[2] = vr;
}

__global__ void __launch_bounds__(VNL_HEAD_THREADS) vnl_ppo_head_kernel(vnl_ppo_head_args a, float* stats) {
#if VNL_HEAD_THREADS > 1
  __shared__ float red[VNL_HEAD_THREADS];
#else
  float red[1];
#endif
#if defined(__HIP_DEVICE_COMPILE__)
  __builtin_amdgcn_s_setprio(3);  // (in a PPO step this launch sits on the intention network's chain, beside the value MLP's GEMMs)
#endif
  const int tid = (int)threadIdx.x, A = a.act, N = a.T * a.B;
  const int lane = tid % VNL_HEAD_GROUP, group = (int)(blockIdx.x * (VNL_HEAD_THREADS / VNL_HEAD_GROUP)) + tid / VNL_HEAD_GROUP;
  const int ngroups = (int)gridDim.x * (VNL_HEAD_THREADS / VNL_HEAD_GROUP);
  const float adv_mean = stats[0], adv_std = stats[1];
  const float inv_n = 1.f / (float)N, half_log_2pi = 0.91893853320467274178f, ln2 = 0.69314718055994530942f;
  const float lo = 1.f - a.clipping_epsilon, hi = 1.f + a.clipping_epsilon;
  float s_pl = 0.f, s_vl = 0.f, s_ent = 0.f;  // counted once per sample (lane 0 of the group)
  for (int n = group; n < N; n += ngroups) {
    const float* lg = a.logits + (size_t)n * 2 * A;
    const float* ra = a.raw_action + (size_t)n * A;
    const float* ee = a.eps_entropy + (size_t)n * A;
    // tanh-Normal log-prob of the stored raw action and entropy estimate at loc + scale * eps, summed over actions
    float tlp = 0.f, ent = 0.f;
    for (int k = lane; k < A; k += VNL_HEAD_GROUP) {
      const float loc = lg[k], scale = (vnl_softplus(lg[A + k]) + a.min_std) * a.var_scale;
      const float x = ra[k], z = (x - loc) / scale, lsc = logf(scale);
      tlp += (-0.5f * z * z - half_log_2pi - lsc) - 2.f * (ln2 - x - vnl_softplus(-2.f * x));
      const float xe = loc + scale * ee[k];
      ent += (0.5f + half_log_2pi + lsc) + 2.f * (ln2 - xe - vnl_softplus(-2.f * xe));
    }
    VNL_GROUP_SUM(tlp);
    VNL_GROUP_SUM(ent);
    float adv = a.advantages[n];
    if (a.normalize_advantage) adv = (adv - adv_mean) / (adv_std + 1e-8f);
    const float rho = expf(tlp - a.behaviour_log_prob[n]);
    const float rc = fminf(fmaxf(rho, lo), hi);
    const float s1 = rho * adv, s2 = rc * adv;
    // d min(s1, s2) / d rho: adv where the unclipped branch is active (ties: both branches agree), else 0
    const float dmin = (s1 < s2 || (s1 == s2 && rho >= lo && rho <= hi)) ? adv : ((s1 == s2) ? 0.5f * adv : 0.f);
    const float g_tlp = -inv_n * dmin * rho;      // d policy_loss / d target_log_prob
    const float g_ent = -a.entropy_cost * inv_n;  // d entropy_loss / d entropy_n
    const float verr = a.vs[n] - a.baseline[n];
    if (lane == 0) s_pl += -fminf(s1, s2), s_vl += verr * verr, s_ent += ent;  // (d v_loss / d baseline: vnl_ppo_gae_kernel)
    float* gl = a.g_logits + (size_t)n * 2 * A;
    for (int k = lane; k < A; k += VNL_HEAD_GROUP) {
      const float sraw = lg[A + k], scale = (vnl_softplus(sraw) + a.min_std) * a.var_scale;
      const float loc = lg[k], z = (ra[k] - loc) / scale, xe = loc + scale * ee[k], th = tanhf(xe);
      const float dscale_ds = a.var_scale / (1.f + expf(-sraw));  // softplus' = sigmoid
      gl[k] = g_tlp * (z / scale) + g_ent * (-2.f * th);
      gl[A + k] = (g_tlp * ((z * z - 1.f) / scale) + g_ent * (1.f / scale - 2.f * th * ee[k])) * dscale_ds;
    }
  }
  // latent KL (kl_divergence: a mean over samples x latent), intention_losses.py:21-23
  float s_kl = 0.f;
  const int NL = N * a.latent;
  const float kscale = a.kl_weight / (float)NL;
  for (int i = (int)(blockIdx.x * VNL_HEAD_THREADS) + tid; i < NL; i += (int)gridDim.x * VNL_HEAD_THREADS) {
    const float mu = a.lat_mean[i], lv = a.lat_logvar[i], ev = expf(lv);
    s_kl += 1.f + lv - mu * mu - ev;
    a.g_lat_mean[i] = kscale * mu;
    a.g_lat_logvar[i] = -0.5f * kscale * (1.f - ev);
  }
  const float p0 = vnl_block_sum(s_pl, red), p1 = vnl_block_sum(s_vl, red), p2 = vnl_block_sum(s_ent, red),
              p3 = vnl_block_sum(s_kl, red);
  if (tid == 0) {
    float* part = stats + 4 + 4 * blockIdx.x;
    part[0] = p0, part[1] = p1, part[2] = p2, part[3] = p3;
  }
}

__global__ void vnl_ppo_finish_kernel(vnl_ppo_head_args a, const float* stats, int nblk) {
  // one 64-lane wave: lane l sums partials l, l+64, ... (fixed order), then a shuffle tree
#if defined(__HIP_DEVICE_COMPILE__)
  __builtin_amdgcn_s_setprio(3);
#endif
  const int lane = (int)threadIdx.x;
  float s[4] = {0.f, 0.f, 0.f, 0.f};
  for (int b = lane; b < nblk; b += VNL_FINISH_THREADS)
    for (int k = 0; k < 4; k++) s[k] += stats[4 + 4 * b + k];
  for (int k = 0; k < 4; k++) VNL_FINISH_SUM(s[k]);
  if (lane != 0) return;
  const int N = a.T * a.B;
  const float inv_n = 1.f / (float)N, kscale = a.kl_weight / (float)(N * a.latent);
  const float pl = s[0] * inv_n, vl = 0.25f * s[1] * inv_n, el = -a.entropy_cost * s[2] * inv_n, kl = -0.5f * kscale * s[3];
  a.metrics[0] = pl + vl + el + kl, a.metrics[1] = pl, a.metrics[2] = vl, a.metrics[3] = el, a.metrics[4] = kl;
  a.metrics[5] = 1.f - vl / stats[2], a.metrics[6] = stats[0], a.metrics[7] = stats[1];
}

int vnl_ppo_head_phase_(const vnl_ppo_head_args* a, float* workspace, void* stream, int phase);
extern "C" int vnl_ppo_head(const vnl_ppo_head_args* a, float* workspace, void* stream) {
  if (!a || !workspace || a->T <= 0 || a->B <= 0 || a->act <= 0 || a->latent <= 0)
    return fail(VNL_ERR_ARG, "vnl_ppo_head: bad sizes");
  const void* need[] = {a->logits, a->baseline, a->bootstrap, a->lat_mean, a->lat_logvar, a->raw_action,
                        a->behaviour_log_prob, a->reward, a->truncation, a->discount, a->eps_entropy, a->g_logits,
                        a->g_baseline, a->g_lat_mean, a->g_lat_logvar, a->vs, a->advantages, a->metrics};
  for (const void* p : need)
    if (!p) return fail(VNL_ERR_ARG, "vnl_ppo_head: null buffer");
  return vnl_ppo_head_phase_(a, workspace, stream, 3);
}

// phase bit 1: GAE + statistics (needs the value outputs only); bit 2: per-sample head + the metrics.  vnl_ppo.hip runs
// phase 1 on the value stream while the intention network's forward is still in flight on the other one.
int vnl_ppo_head_phase_(const vnl_ppo_head_args* a, float* workspace, void* stream, int phase) {
  const int groups_per_block = VNL_HEAD_THREADS / VNL_HEAD_GROUP, N = a->T * a->B;
  int nblk = (N + groups_per_block - 1) / groups_per_block;
  if (nblk > VNL_HEAD_MAXBLK) nblk = VNL_HEAD_MAXBLK;
  if (phase & 1) hipLaunchKernelGGL(vnl_ppo_gae_kernel, dim3(1), dim3(VNL_HEAD_THREADS), 0, (hipStream_t)stream, *a, workspace);
  if (phase & 2) {
    hipLaunchKernelGGL(vnl_ppo_head_kernel, dim3(nblk), dim3(VNL_HEAD_THREADS), 0, (hipStream_t)stream, *a, workspace);
    hipLaunchKernelGGL(vnl_ppo_finish_kernel, dim3(1), dim3(VNL_FINISH_THREADS), 0, (hipStream_t)stream, *a, (const float*)workspace, nblk);
  }
  HIPCHK(hipGetLastError());
  return VNL_OK;
}

// ---- minibatch gather (include/vnl.h: vnl_gather_rows): one workgroup per selected row j, all arrays, all t
__global__ void __launch_bounds__(VNL_POST_THREADS) vnl_gather_kernel(vnl_gather_desc d) {
  const int j = (int)blockIdx.x, t = (int)blockIdx.y, tid = (int)threadIdx.x;
  const long long src_row = d.idx[j];
  for (int k = 0; k < d.num_ops; k++) {
    const vnl_gather_op& op = d.ops[k];
    if (t >= op.T) continue;
    const int w = op.width;
    const unsigned* src = (const unsigned*)op.src + ((size_t)t * d.N + src_row) * w;
    unsigned* dst = (unsigned*)op.dst + ((size_t)t * d.M + j) * w;
    for (int c = tid; c < w; c += VNL_POST_THREADS) dst[c] = src[c];
  }
}

extern "C" int vnl_gather_rows(const vnl_gather_desc* d, void* stream) {
  if (!d || !d->idx || d->N <= 0 || d->M <= 0 || d->num_ops < 0 || d->num_ops > VNL_POST_MAX_OPS)
    return fail(VNL_ERR_ARG, "vnl_gather_rows: bad descriptor");
  for (int k = 0; k < d->num_ops; k++)
    if (!d->ops[k].dst || !d->ops[k].src || d->ops[k].T <= 0 || d->ops[k].width <= 0)
      return fail(VNL_ERR_ARG, "vnl_gather_rows: bad op");
  int tmax = 1;
  for (int k = 0; k < d->num_ops; k++) tmax = d->ops[k].T > tmax ? d->ops[k].T : tmax;
  hipLaunchKernelGGL(vnl_gather_kernel, dim3(d->M, tmax), dim3(VNL_POST_THREADS), 0, (hipStream_t)stream, *d);
  HIPCHK(hipGetLastError());
  return VNL_OK;
}

// ---- Adam (include/vnl.h: vnl_adam_step) ----------------------------------------------------------
#ifndef VNL_ADAM_THREADS
#define VNL_ADAM_THREADS 256
#endif
__global__ void __launch_bounds__(VNL_ADAM_THREADS) vnl_adam_kernel(float* p, const float* g, float* mu, float* nu,
                                                                    const long long* count, long long n, double lr_,
                                                                    double b1_, double b2_, double eps_) {
  // the two bias corrections: one thread of the block evaluates the double-precision powers (they were most of this
  // kernel's time when every thread did), the others pick them up from LDS
#if VNL_ADAM_THREADS > 1
  __shared__ float bc[2];
  if (threadIdx.x == 0) {
    const double t = (double)count[0];
    bc[0] = (float)(1.0 - pow(b1_, t)), bc[1] = (float)(1.0 - pow(b2_, t));
  }
  __syncthreads();
  const float bc1 = bc[0], bc2 = bc[1];
#else
  const double t = (double)count[0];
  const float bc1 = (float)(1.0 - pow(b1_, t)), bc2 = (float)(1.0 - pow(b2_, t));
#endif
  const float b1 = (float)b1_, b2 = (float)b2_, omb1 = (float)(1.0 - b1_), omb2 = (float)(1.0 - b2_);
  const float lr = (float)lr_, eps = (float)eps_;
  for (long long i = (long long)blockIdx.x * VNL_ADAM_THREADS + threadIdx.x; i < n;
       i += (long long)gridDim.x * VNL_ADAM_THREADS) {
    const float gi = g[i];
    const float m = b1 * mu[i] + omb1 * gi, v = b2 * nu[i] + omb2 * gi * gi;
    mu[i] = m, nu[i] = v;
    p[i] -= lr * ((m / bc1) / (sqrtf(v / bc2) + eps));
  }
}

extern "C" int vnl_adam_step(float* params, const float* grads, float* mu, float* nu, const int64_t* count, int64_t n,
                             double lr, double b1, double b2, double eps, void* stream) {
  if (!params || !grads || !mu || !nu || !count || n <= 0) return fail(VNL_ERR_ARG, "vnl_adam_step: null argument");
  long long blocks = (n + VNL_ADAM_THREADS - 1) / VNL_ADAM_THREADS;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(vnl_adam_kernel, dim3((unsigned)blocks), dim3(VNL_ADAM_THREADS), 0, (hipStream_t)stream, params,
                     grads, mu, nu, (const long long*)count, (long long)n, lr, b1, b2, eps);
  HIPCHK(hipGetLastError());
  return VNL_OK;
}

// the policy-forward entry points (vnl_policy_*) live in vnl_policy.hip
