// One environment per wavefront lane: the whole control step (n_frames physics
// substeps + observation / reference-trajectory / reward / termination glue) as
// straight per-lane code.  Control flow is wave-uniform (same model for every
// lane); table indices are scalars; all per-env storage is SoA [idx][env] so each
// wave access is one coalesced 256-B line.
//
// What each stage follows:
//   physics  : MJX forward/step [UPSTREAM mjx/_src/{smooth,collision_primitive,
//              constraint,solver,passive,forward}.py], called from reference
//              envs/rodent.py:148 (pipeline_init) and :181 (pipeline_step)
//   env glue : reference envs/rodent.py:178-470
//
// Algorithmic choices that differ from MJX's dense route (same mathematics):
//   * spatial quantities (cdof, cinert, cvel ...) are expressed about the root
//     body's origin O instead of subtree_com[root]; the choice of reference point
//     cancels in every scalar the step produces;
//   * qM is kept tree-sparse (MuJoCo's dof_Madr layout, 1119 entries for the
//     rodent) and factorised as L'DL (mj_factorM order) instead of dense Cholesky;
//   * efc_J is never materialised: J*v and J'*f are evaluated through the
//     kinematic tree (body twist forward pass / body wrench backward pass);
//   * M*v is evaluated matrix-free with the per-body inertias and shares the
//     forward pass of J*v.
#pragma once
#include <math.h>

#include "vnl_types.h"

#ifndef VNL_HD
#define VNL_HD __host__ __device__ __forceinline__
#endif
#ifndef VNL_WAVE_ANY
#define VNL_WAVE_ANY(x) (__ballot(x) != 0ull)
#endif

// Diagnostic build only (-DVNL_PROFILE, csrc/build.py --profile): per-stage s_memtime stamps summed
// into a __device__ array that no product code reads.  Never defined in the shipped library.
#ifdef VNL_PROFILE
#define VNL_NPROF 16
__device__ unsigned long long g_vnl_prof[VNL_NPROF];
#define VNL_PROF(i)                                             \
  do {                                                          \
    unsigned long long t_ = __builtin_amdgcn_s_memtime();       \
    prof_[i] += t_ - last_;                                     \
    last_ = __builtin_amdgcn_s_memtime();                       \
  } while (0)
#else
#define VNL_PROF(i)
#endif

#define VNL_MINVAL vreal(1e-15)
#define VNL_MINIMP vreal(0.0001)
#define VNL_MAXIMP vreal(0.9999)

struct V3 {
  vreal x, y, z;
};
VNL_HD V3 v3(vreal x, vreal y, vreal z) { return V3{x, y, z}; }
VNL_HD V3 operator+(V3 a, V3 b) { return V3{a.x + b.x, a.y + b.y, a.z + b.z}; }
VNL_HD V3 operator-(V3 a, V3 b) { return V3{a.x - b.x, a.y - b.y, a.z - b.z}; }
VNL_HD V3 operator*(V3 a, vreal s) { return V3{a.x * s, a.y * s, a.z * s}; }
VNL_HD vreal dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
VNL_HD V3 cross(V3 a, V3 b) { return V3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
struct Q4 {
  vreal w, x, y, z;
};
VNL_HD Q4 qmul(Q4 u, Q4 v) {
  return Q4{u.w * v.w - u.x * v.x - u.y * v.y - u.z * v.z, u.w * v.x + u.x * v.w + u.y * v.z - u.z * v.y,
            u.w * v.y - u.x * v.z + u.y * v.w + u.z * v.x, u.w * v.z + u.x * v.y - u.y * v.x + u.z * v.w};
}
// MJX math.rotate: 2(u.v)u + (s^2 - u.u)v + 2s(u x v)
VNL_HD V3 qrot(V3 v, Q4 q) {
  V3 u = v3(q.x, q.y, q.z);
  vreal uv = dot(u, v), uu = dot(u, u);
  V3 c = cross(u, v);
  return u * (vreal(2.) * uv) + v * (q.w * q.w - uu) + c * (vreal(2.) * q.w);
}
struct M3 {
  vreal a[9];
};
VNL_HD M3 qmat(Q4 q) {
  M3 m;
  vreal q00 = q.w * q.w, q01 = q.w * q.x, q02 = q.w * q.y, q03 = q.w * q.z;
  vreal q11 = q.x * q.x, q12 = q.x * q.y, q13 = q.x * q.z, q22 = q.y * q.y, q23 = q.y * q.z, q33 = q.z * q.z;
  m.a[0] = q00 + q11 - q22 - q33, m.a[1] = vreal(2.) * (q12 - q03), m.a[2] = vreal(2.) * (q13 + q02);
  m.a[3] = vreal(2.) * (q12 + q03), m.a[4] = q00 - q11 + q22 - q33, m.a[5] = vreal(2.) * (q23 - q01);
  m.a[6] = vreal(2.) * (q13 - q02), m.a[7] = vreal(2.) * (q23 + q01), m.a[8] = q00 - q11 - q22 + q33;
  return m;
}
VNL_HD V3 mmul(const M3& m, V3 v) {
  return V3{m.a[0] * v.x + m.a[1] * v.y + m.a[2] * v.z, m.a[3] * v.x + m.a[4] * v.y + m.a[5] * v.z,
            m.a[6] * v.x + m.a[7] * v.y + m.a[8] * v.z};
}
struct S6 {  // spatial vector (ang, lin)
  V3 a, l;
};
VNL_HD S6 operator+(S6 p, S6 q) { return S6{p.a + q.a, p.l + q.l}; }
VNL_HD S6 operator*(S6 p, vreal s) { return S6{p.a * s, p.l * s}; }
VNL_HD vreal dot(S6 p, S6 q) { return dot(p.a, q.a) + dot(p.l, q.l); }
VNL_HD S6 mcross(S6 u, S6 v) { return S6{cross(u.a, v.a), cross(u.l, v.a) + cross(u.a, v.l)}; }
VNL_HD S6 mcross_force(S6 v, S6 f) { return S6{cross(v.a, f.a) + cross(v.l, f.l), cross(v.a, f.l)}; }

struct EnvLane {
  const DevModel& m;
  const DevEnv& ev;
  const DevState& st;
  const WsLayout& L;
  vreal* ws;
  unsigned B, e;
#ifdef VNL_PROFILE
  mutable unsigned long long prof_[VNL_NPROF] = {0}, last_ = 0;
  VNL_HD void prof_begin() const { last_ = __builtin_amdgcn_s_memtime(); }
  VNL_HD void prof_end() const {
    if ((threadIdx.x & 63) == 0)
      for (int i = 0; i < VNL_NPROF; i++) atomicAdd(&g_vnl_prof[i], prof_[i]);
  }
#else
  VNL_HD void prof_begin() const {}
  VNL_HD void prof_end() const {}
#endif

  VNL_HD vreal& W(int o) const { return ws[(unsigned)o * B + e]; }
  VNL_HD static vreal& at(vreal* p, int k, unsigned B, unsigned e) { return p[(unsigned)k * B + e]; }
#define ST(field, k) st.field[(unsigned)(k)*B + e]

  VNL_HD V3 ld3(int o) const { return V3{W(o), W(o + 1), W(o + 2)}; }
  VNL_HD void st3(int o, V3 v) const { W(o) = v.x, W(o + 1) = v.y, W(o + 2) = v.z; }
  VNL_HD S6 ld6(int o) const { return S6{ld3(o), ld3(o + 3)}; }
  VNL_HD void st6(int o, S6 v) const { st3(o, v.a), st3(o + 3, v.l); }
  VNL_HD static V3 t3(const vreal* t, int i) { return V3{t[3 * i], t[3 * i + 1], t[3 * i + 2]}; }
  VNL_HD static Q4 t4(const vreal* t, int i) { return Q4{t[4 * i], t[4 * i + 1], t[4 * i + 2], t[4 * i + 3]}; }

  // cinert (10) x motion -> force   [MJX math.inert_mul]
  VNL_HD S6 inert_mul(int o, S6 v) const {
    vreal ixx = W(o), iyy = W(o + 1), izz = W(o + 2), ixy = W(o + 3), ixz = W(o + 4), iyz = W(o + 5);
    V3 mp = ld3(o + 6);
    vreal mass = W(o + 9);
    V3 ang = V3{ixx * v.a.x + ixy * v.a.y + ixz * v.a.z, ixy * v.a.x + iyy * v.a.y + iyz * v.a.z,
                ixz * v.a.x + iyz * v.a.y + izz * v.a.z} +
             cross(mp, v.l);
    V3 lin = v.l * mass - cross(mp, v.a);
    return S6{ang, lin};
  }

  // ------------------------------------------------------------------ kinematics
  // smooth.kinematics + com_pos (cdof, cinert) in one tree pass, reference point O.
  VNL_HD V3 ref_point() const {
    return m.root_free ? V3{ST(qpos, 0), ST(qpos, 1), ST(qpos, 2)} : V3{m.root_px, m.root_py, m.root_pz};
  }

  VNL_HD void kinematics() const {
    V3 O = ref_point();
    ST(xpos, 0) = vreal(0.), ST(xpos, 1) = vreal(0.), ST(xpos, 2) = vreal(0.);
    ST(xquat, 0) = vreal(1.), ST(xquat, 1) = vreal(0.), ST(xquat, 2) = vreal(0.), ST(xquat, 3) = vreal(0.);
    for (int k = 0; k < 10; k++) W(L.cinert + k) = vreal(0.);
    V3 csum = v3(vreal(0.), vreal(0.), vreal(0.));
    for (int b = 1; b < m.nbody; b++) {
      int p = m.body_parent[b];
      V3 ppos = V3{ST(xpos, 3 * p), ST(xpos, 3 * p + 1), ST(xpos, 3 * p + 2)};
      Q4 pq = Q4{ST(xquat, 4 * p), ST(xquat, 4 * p + 1), ST(xquat, 4 * p + 2), ST(xquat, 4 * p + 3)};
      V3 pos = ppos + qrot(t3(m.body_pos, b), pq);
      Q4 quat = qmul(pq, t4(m.body_quat, b));
      int jn = m.body_jntnum[b], ja = m.body_jntadr[b];
      for (int k = 0; k < jn; k++) {
        int j = ja + k, qa = m.jnt_qposadr[j], da = m.jnt_dofadr[j];
        if (m.jnt_type[j] == VNL_JNT_FREE) {
          pos = V3{ST(qpos, qa), ST(qpos, qa + 1), ST(qpos, qa + 2)};
          quat = Q4{ST(qpos, qa + 3), ST(qpos, qa + 4), ST(qpos, qa + 5), ST(qpos, qa + 6)};
          vreal n = sqrt(quat.w * quat.w + quat.x * quat.x + quat.y * quat.y + quat.z * quat.z);
          vreal inv = n > vreal(0.) ? vreal(1.) / n : vreal(1.);
          quat = Q4{quat.w * inv, quat.x * inv, quat.y * inv, quat.z * inv};
          ST(qpos, qa + 3) = quat.w, ST(qpos, qa + 4) = quat.x, ST(qpos, qa + 5) = quat.y, ST(qpos, qa + 6) = quat.z;
          M3 R = qmat(quat);
          V3 off = O - pos;
          for (int t = 0; t < 3; t++) {
            int o = L.cdof + 6 * (da + t);
            W(o) = vreal(0.), W(o + 1) = vreal(0.), W(o + 2) = vreal(0.);
            W(o + 3) = t == 0 ? vreal(1.) : vreal(0.), W(o + 4) = t == 1 ? vreal(1.) : vreal(0.), W(o + 5) = t == 2 ? vreal(1.) : vreal(0.);
          }
          for (int t = 0; t < 3; t++) {
            V3 ax = V3{R.a[t], R.a[3 + t], R.a[6 + t]};
            st6(L.cdof + 6 * (da + 3 + t), S6{ax, cross(ax, off)});
          }
        } else {
          V3 jp = t3(m.jnt_pos, j), jax = t3(m.jnt_axis, j);
          V3 anchor = qrot(jp, quat) + pos;
          V3 axis = qrot(jax, quat);
          vreal ang = ST(qpos, qa) - m.jnt_qpos0[j];
          vreal s = sin(vreal(0.5) * ang), c = cos(vreal(0.5) * ang);
          quat = qmul(quat, Q4{c, jax.x * s, jax.y * s, jax.z * s});
          pos = anchor - qrot(jp, quat);
          st6(L.cdof + 6 * da, S6{axis, cross(axis, O - anchor)});
        }
      }
      ST(xpos, 3 * b) = pos.x, ST(xpos, 3 * b + 1) = pos.y, ST(xpos, 3 * b + 2) = pos.z;
      ST(xquat, 4 * b) = quat.w, ST(xquat, 4 * b + 1) = quat.x, ST(xquat, 4 * b + 2) = quat.y,
                    ST(xquat, 4 * b + 3) = quat.z;
      // inertia about O in world axes: R I R' + m(|r|^2 1 - r r'), r = xipos - O
      M3 R = qmat(quat);
      vreal mass = m.body_mass[b];
      V3 xip = pos + mmul(R, t3(m.body_ipos, b));
      csum = csum + xip * mass;
      V3 r = xip - O;
      const vreal* I6 = m.body_inertia6 + 6 * b;  // xx yy zz xy xz yz (body axes, about ipos)
      vreal Ib[9] = {I6[0], I6[3], I6[4], I6[3], I6[1], I6[5], I6[4], I6[5], I6[2]};
      vreal T[9];
      for (int i = 0; i < 3; i++)
        for (int k = 0; k < 3; k++)
          T[3 * i + k] = R.a[3 * i] * Ib[k] + R.a[3 * i + 1] * Ib[3 + k] + R.a[3 * i + 2] * Ib[6 + k];
      vreal rr = dot(r, r);
      int o = L.cinert + 10 * b;
      W(o + 0) = T[0] * R.a[0] + T[1] * R.a[1] + T[2] * R.a[2] + mass * (rr - r.x * r.x);
      W(o + 1) = T[3] * R.a[3] + T[4] * R.a[4] + T[5] * R.a[5] + mass * (rr - r.y * r.y);
      W(o + 2) = T[6] * R.a[6] + T[7] * R.a[7] + T[8] * R.a[8] + mass * (rr - r.z * r.z);
      W(o + 3) = T[0] * R.a[3] + T[1] * R.a[4] + T[2] * R.a[5] - mass * r.x * r.y;
      W(o + 4) = T[0] * R.a[6] + T[1] * R.a[7] + T[2] * R.a[8] - mass * r.x * r.z;
      W(o + 5) = T[3] * R.a[6] + T[4] * R.a[7] + T[5] * R.a[8] - mass * r.y * r.z;
      W(o + 6) = r.x * mass, W(o + 7) = r.y * mass, W(o + 8) = r.z * mass, W(o + 9) = mass;
    }
    ST(com1, 0) = csum.x * m.total_mass_inv, ST(com1, 1) = csum.y * m.total_mass_inv,
             ST(com1, 2) = csum.z * m.total_mass_inv;
  }

  // ------------------------------------------------------------------ inertia
  // smooth.crb + make_m into the tree-sparse layout: row i = [M(i,i), M(i,anc1), ...]
  VNL_HD void crb_mass_matrix() const {
    int n10 = 10 * m.nbody;
    for (int k = 0; k < n10; k++) W(L.bodyA + k) = W(L.cinert + k);
    for (int b = m.nbody - 1; b > 1; b--) {
      int p = m.body_parent[b];
      if (p > 0)
        for (int k = 0; k < 10; k++) W(L.bodyA + 10 * p + k) += W(L.bodyA + 10 * b + k);
    }
    for (int i = 0; i < m.nv; i++) {
      S6 f = inert_mul(L.bodyA + 10 * m.dof_body[i], ld6(L.cdof + 6 * i));
      int adr = m.dof_Madr[i], dep = m.dof_depth[i];
      for (int a = 0; a <= dep; a++) {
        int j = m.M_anc[adr + a];
        vreal v = dot(f, ld6(L.cdof + 6 * j));
        if (a == 0) v += m.dof_armature[i];
        W(L.M + adr + a) = v;
      }
    }
  }

  // L'DL of (M + diag_scale*damping) in MuJoCo's mj_factorM order -> LD, dinv
  VNL_HD void factor(vreal diag_scale) const {
    for (int k = 0; k < m.nM; k++) W(L.LD + k) = W(L.M + k);
    if (diag_scale != vreal(0.))
      for (int i = 0; i < m.nv; i++) W(L.LD + m.dof_Madr[i]) += diag_scale * m.dof_damping[i];
    for (int k = m.nv - 1; k >= 0; k--) {
      int adr_k = m.dof_Madr[k], dk = m.dof_depth[k];
      vreal inv = vreal(1.) / W(L.LD + adr_k);
      W(L.dinv + k) = inv;
      for (int a = 1; a <= dk; a++) {
        int i = m.M_anc[adr_k + a];
        int adr_i = m.dof_Madr[i], len = dk - a + 1;
        vreal tmp = W(L.LD + adr_k + a) * inv;
        for (int c = 0; c < len; c++) W(L.LD + adr_i + c) -= tmp * W(L.LD + adr_k + a + c);
        W(L.LD + adr_k + a) = tmp;
      }
    }
  }

  // x <- (L'DL)^-1 x   [mj_solveLD]
  VNL_HD void solve_inplace(int x) const {
    for (int i = m.nv - 1; i >= 0; i--) {
      int adr = m.dof_Madr[i], dep = m.dof_depth[i];
      vreal xi = W(x + i);
      for (int a = 1; a <= dep; a++) W(x + m.M_anc[adr + a]) -= W(L.LD + adr + a) * xi;
    }
    for (int i = 0; i < m.nv; i++) {
      int adr = m.dof_Madr[i], dep = m.dof_depth[i];
      vreal s = W(x + i) * W(L.dinv + i);
      // forward substitution needs D^-1 applied to ancestors first: ancestors have lower index, already final
      for (int a = 1; a <= dep; a++) s -= W(L.LD + adr + a) * W(x + m.M_anc[adr + a]);
      W(x + i) = s;
    }
  }

  // ------------------------------------------------------------------ velocity
  // com_vel + rne in one forward / one backward pass; qfrc_bias -> L.bias
  VNL_HD void bias_forces() const {
    int cv = L.bodyB, ca = L.bodyB + 6 * m.nbody, cf = L.bodyC;
    st6(cv, S6{v3(0, 0, 0), v3(0, 0, 0)});
    st6(ca, S6{v3(0, 0, 0), v3(-m.gx, -m.gy, -m.gz)});
    for (int b = 1; b < m.nbody; b++) {
      int p = m.body_parent[b];
      S6 vel = ld6(cv + 6 * p), acc = ld6(ca + 6 * p);
      int jn = m.body_jntnum[b], ja = m.body_jntadr[b];
      for (int k = 0; k < jn; k++) {
        int j = ja + k, da = m.jnt_dofadr[j];
        if (m.jnt_type[j] == VNL_JNT_FREE) {
          for (int t = 0; t < 3; t++) vel = vel + ld6(L.cdof + 6 * (da + t)) * ST(qvel, da + t);
          S6 vel0 = vel;
          for (int t = 3; t < 6; t++) {
            S6 c = ld6(L.cdof + 6 * (da + t));
            vreal qd = ST(qvel, da + t);
            acc = acc + mcross(vel0, c) * qd;
            vel = vel + c * qd;
          }
        } else {
          S6 c = ld6(L.cdof + 6 * da);
          vreal qd = ST(qvel, da);
          acc = acc + mcross(vel, c) * qd;
          vel = vel + c * qd;
        }
      }
      st6(cv + 6 * b, vel), st6(ca + 6 * b, acc);
      S6 f = inert_mul(L.cinert + 10 * b, acc) + mcross_force(vel, inert_mul(L.cinert + 10 * b, vel));
      st6(cf + 6 * b, f);
    }
    for (int b = m.nbody - 1; b > 1; b--) {
      int p = m.body_parent[b];
      if (p > 0)
        for (int k = 0; k < 6; k++) W(cf + 6 * p + k) += W(cf + 6 * b + k);
    }
    for (int d = 0; d < m.nv; d++) W(L.bias + d) = dot(ld6(L.cdof + 6 * d), ld6(cf + 6 * m.dof_body[d]));
  }

  // passive + actuation + qfrc_smooth + qacc_smooth
  VNL_HD void smooth_forces() const {
    for (int d = 0; d < m.nv; d++) {
      W(L.smooth + d) = -m.dof_damping[d] * ST(qvel, d) - W(L.bias + d);
      ST(qfrc_actuator, d) = vreal(0.);
    }
    for (int j = 0; j < m.njnt; j++) {
      if (m.jnt_type[j] != VNL_JNT_HINGE) continue;
      vreal k = m.jnt_stiffness[j];
      if (k != vreal(0.)) W(L.smooth + m.jnt_dofadr[j]) -= k * (ST(qpos, m.jnt_qposadr[j]) - m.jnt_springref[j]);
    }
    for (int i = 0; i < m.nu; i++) {
      vreal ctrl = W(L.ctrl + i), a = ctrl;
      vreal tau = m.act_tau[i];
      if (tau >= vreal(0.)) {
        a = ST(act, i);
        W(L.actdot + i) = (ctrl - a) / fmax(tau, VNL_MINVAL);
      }
      ST(qfrc_actuator, m.act_dof[i]) += m.act_gear[i] * m.act_gain[i] * a;
    }
    for (int d = 0; d < m.nv; d++) {
      vreal s = W(L.smooth + d) + ST(qfrc_actuator, d);
      W(L.smooth + d) = s;
      W(L.qacc_smooth + d) = s;
    }
    solve_inplace(L.qacc_smooth);
  }

  // ------------------------------------------------------------------ constraints
  // solref/solimp -> (k, b, imp) [constraint.make_constraint]
  VNL_HD static void kbimp(const vreal* solref, const vreal* solimp, vreal dt, vreal pos, vreal& k, vreal& b,
                           vreal& imp) {
    vreal timeconst = fmax(solref[0], vreal(2.) * dt), dampratio = solref[1];
    vreal dmin = fmin(fmax(solimp[0], VNL_MINIMP), VNL_MAXIMP), dmax = fmin(fmax(solimp[1], VNL_MINIMP), VNL_MAXIMP);
    vreal width = fmax(solimp[2], VNL_MINVAL), mid = fmin(fmax(solimp[3], VNL_MINIMP), VNL_MAXIMP);
    vreal power = fmax(solimp[4], vreal(1.));
    k = vreal(1.) / (dmax * dmax * timeconst * timeconst * dampratio * dampratio);
    b = vreal(2.) / (dmax * timeconst);
    if (solref[0] <= vreal(0.)) k = -solref[0] / (dmax * dmax);
    if (solref[1] <= vreal(0.)) b = -solref[1] / dmax;
    vreal x = fabs(pos) / width, y;
    if (power == vreal(2.)) {
      y = x < mid ? x * x / mid : vreal(1.) - (vreal(1.) - x) * (vreal(1.) - x) / (vreal(1.) - mid);
    } else {
      vreal ia = (vreal(1.) / pow(mid, power - vreal(1.))) * pow(x, power);
      vreal ib = vreal(1.) - (vreal(1.) / pow(vreal(1.) - mid, power - vreal(1.))) * pow(vreal(1.) - x, power);
      y = x < mid ? ia : ib;
    }
    imp = dmin + y * (dmax - dmin);
    imp = fmin(fmax(imp, dmin), dmax);
    if (x > vreal(1.)) imp = dmax;
  }

  // collision (plane vs sphere / capsule / ellipsoid) + constraint rows.
  // Rows that MJX would mask out (pos >= 0) get D = 0: they add nothing to cost,
  // force or gradient.  Needs cvel (bodyB) from bias_forces for aref.
  VNL_HD void make_constraint() const {
    V3 O = ref_point();
    V3 n = v3(m.pnx, m.pny, m.pnz), pp = v3(m.ppx, m.ppy, m.ppz);
    for (int r = 0; r < m.nlimit; r++) {
      vreal q = ST(qpos, m.lim_qadr[r]);
      vreal dlo = q - m.lim_lo[r], dhi = m.lim_hi[r] - q;
      vreal pos = fmin(dlo, dhi) - m.lim_margin[r];
      vreal sign = dlo < dhi ? vreal(1.) : -vreal(1.);
      vreal k, b, imp;
      kbimp(m.lim_solref + 2 * r, m.lim_solimp + 5 * r, m.dt, pos, k, b, imp);
      vreal R = fmax(m.lim_invweight[r] * (vreal(1.) - imp) / imp, VNL_MINVAL);
      bool present = pos < vreal(0.);
      W(L.lim_sign + r) = sign;
      W(L.efc_D + r) = present ? vreal(1.) / R : vreal(0.);
      W(L.efc_aref + r) = -b * (sign * ST(qvel, m.lim_dof[r])) - k * imp * pos;
    }
    for (int g = 0; g < m.ncg; g++) {
      int bd = m.cg_body[g], c0 = m.cg_conadr[g], type = m.cg_type[g];
      V3 bpos = V3{ST(xpos, 3 * bd), ST(xpos, 3 * bd + 1), ST(xpos, 3 * bd + 2)};
      Q4 bq = Q4{ST(xquat, 4 * bd), ST(xquat, 4 * bd + 1), ST(xquat, 4 * bd + 2), ST(xquat, 4 * bd + 3)};
      V3 gpos = bpos + qrot(t3(m.cg_pos, g), bq);
      M3 R = qmat(qmul(bq, t4(m.cg_quat, g)));
      V3 size = t3(m.cg_size, g);
      vreal dist[2];
      V3 cpos[2], t1 = v3(m.t1x, m.t1y, m.t1z);
      int nc = 1;
      if (type == VNL_GEOM_CAPSULE) {
        nc = 2;
        V3 axis = V3{R.a[2], R.a[5], R.a[8]};
        V3 bv = axis - n * dot(n, axis);
        vreal bn = sqrt(dot(bv, bv));
        if (bn < vreal(0.5)) {
          bv = (-vreal(0.5) < n.y && n.y < vreal(0.5)) ? v3(vreal(0.), vreal(1.), vreal(0.)) : v3(vreal(0.), vreal(0.), vreal(1.));
        } else {
          bv = bv * (vreal(1.) / bn);
        }
        t1 = bv;
        for (int s = 0; s < 2; s++) {
          V3 c = gpos + axis * (s == 0 ? size.y : -size.y);
          vreal d = dot(c - pp, n) - size.x;
          dist[s] = d;
          cpos[s] = c - n * (size.x + vreal(0.5) * d);
        }
      } else if (type == VNL_GEOM_SPHERE) {
        vreal d = dot(gpos - pp, n) - size.x;
        dist[0] = d;
        cpos[0] = gpos - n * (size.x + vreal(0.5) * d);
      } else {
        V3 ln = V3{(R.a[0] * n.x + R.a[3] * n.y + R.a[6] * n.z) * size.x, (R.a[1] * n.x + R.a[4] * n.y + R.a[7] * n.z) * size.y,
                   (R.a[2] * n.x + R.a[5] * n.y + R.a[8] * n.z) * size.z};
        vreal nn = sqrt(dot(ln, ln));
        vreal inv = nn > vreal(0.) ? vreal(1.) / nn : vreal(1.);
        V3 sup = V3{-ln.x * inv * size.x, -ln.y * inv * size.y, -ln.z * inv * size.z};
        V3 pos = gpos + mmul(R, sup);
        vreal d = dot(n, pos - pp);
        dist[0] = d;
        cpos[0] = pos - n * (d * vreal(0.5));
      }
      V3 t2 = cross(n, t1);
      vreal mu = m.cg_mu[g], margin = m.cg_margin[g], invw = m.cg_invweight[g];
      S6 vel = ld6(L.bodyB + 6 * bd);
      for (int s = 0; s < nc; s++) {
        int c = c0 + s, r0 = m.nlimit + 4 * c;
        vreal d = dist[s] - margin;
        bool present = d < vreal(0.);
        W(L.con_dist + c) = dist[s];
        V3 rel = cpos[s] - O;
        st3(L.con_r + 3 * c, rel);
        st3(L.con_t1 + 3 * c, t1);
        vreal D = vreal(0.);
        vreal ar[4] = {vreal(0.), vreal(0.), vreal(0.), vreal(0.)};
        if (VNL_WAVE_ANY(present)) {
          vreal k, b, imp;
          kbimp(m.cg_solref + 2 * g, m.cg_solimp + 5 * g, m.dt, d, k, b, imp);
          vreal Rr = fmax(invw * (vreal(1.) - imp) / imp, VNL_MINVAL);
          D = present ? vreal(1.) / Rr : vreal(0.);
          V3 pv = vel.l + cross(vel.a, rel);
          vreal jn = dot(n, pv), j1 = dot(t1, pv) * mu, j2 = dot(t2, pv) * mu;
          vreal kp = k * imp * d;
          ar[0] = -b * (jn + j1) - kp, ar[1] = -b * (jn - j1) - kp;
          ar[2] = -b * (jn + j2) - kp, ar[3] = -b * (jn - j2) - kp;
        }
        for (int q = 0; q < 4; q++) W(L.efc_D + r0 + q) = D, W(L.efc_aref + r0 + q) = ar[q];
      }
    }
  }

  // body twists V[b] = sum_{d in path(b)} cdof[d] * vec[d]   -> bodyC[0 .. 6 nbody)
  VNL_HD void body_twists(int vec) const {
    int V = L.bodyC;
    st6(V, S6{v3(0, 0, 0), v3(0, 0, 0)});
    for (int b = 1; b < m.nbody; b++) {
      S6 v = ld6(V + 6 * m.body_parent[b]);
      int jn = m.body_jntnum[b], ja = m.body_jntadr[b];
      for (int k = 0; k < jn; k++) {
        int j = ja + k, da = m.jnt_dofadr[j];
        int nd = m.jnt_type[j] == VNL_JNT_FREE ? 6 : 1;
        for (int t = 0; t < nd; t++) v = v + ld6(L.cdof + 6 * (da + t)) * W(vec + da + t);
      }
      st6(V + 6 * b, v);
    }
  }

  // out[r] = (J vec)[r] - sub[r]*use_sub   using the twists left by body_twists(vec)
  VNL_HD void jac_mul(int vec, int out, bool subtract_aref) const {
    V3 n = v3(m.pnx, m.pny, m.pnz);
    for (int r = 0; r < m.nlimit; r++) {
      vreal v = W(L.lim_sign + r) * W(vec + m.lim_dof[r]);
      if (subtract_aref) v -= W(L.efc_aref + r);
      W(out + r) = v;
    }
    for (int g = 0; g < m.ncg; g++) {
      int bd = m.cg_body[g], c0 = m.cg_conadr[g], nc = m.cg_ncon[g];
      vreal mu = m.cg_mu[g];
      S6 vel = ld6(L.bodyC + 6 * bd);
      for (int s = 0; s < nc; s++) {
        int c = c0 + s, r0 = m.nlimit + 4 * c;
        vreal D = W(L.efc_D + r0);
        vreal o0 = vreal(0.), o1 = vreal(0.), o2 = vreal(0.), o3 = vreal(0.);
        if (VNL_WAVE_ANY(D != vreal(0.))) {
          V3 rel = ld3(L.con_r + 3 * c), t1 = ld3(L.con_t1 + 3 * c), t2 = cross(n, t1);
          V3 pv = vel.l + cross(vel.a, rel);
          vreal jn = dot(n, pv), j1 = dot(t1, pv) * mu, j2 = dot(t2, pv) * mu;
          o0 = jn + j1, o1 = jn - j1, o2 = jn + j2, o3 = jn - j2;
          if (subtract_aref)
            o0 -= W(L.efc_aref + r0), o1 -= W(L.efc_aref + r0 + 1), o2 -= W(L.efc_aref + r0 + 2),
                o3 -= W(L.efc_aref + r0 + 3);
        }
        W(out + r0) = o0, W(out + r0 + 1) = o1, W(out + r0 + 2) = o2, W(out + r0 + 3) = o3;
      }
    }
  }

  // out = M vec, matrix-free, using the twists left by body_twists(vec)
  VNL_HD void mass_mul(int vec, int out) const {
    int V = L.bodyC, F = L.bodyC + 6 * m.nbody;
    for (int b = 1; b < m.nbody; b++) st6(F + 6 * b, inert_mul(L.cinert + 10 * b, ld6(V + 6 * b)));
    for (int b = m.nbody - 1; b > 1; b--) {
      int p = m.body_parent[b];
      if (p > 0)
        for (int k = 0; k < 6; k++) W(F + 6 * p + k) += W(F + 6 * b + k);
    }
    for (int d = 0; d < m.nv; d++)
      W(out + d) = dot(ld6(L.cdof + 6 * d), ld6(F + 6 * m.dof_body[d])) + m.dof_armature[d] * W(vec + d);
  }

  // 0.5 * sum_r D Jaref^2 [Jaref<0]
  VNL_HD vreal constraint_cost(int jaref) const {
    vreal c = vreal(0.);
    for (int r = 0; r < m.nefc; r++) {
      vreal D = W(L.efc_D + r);
      if (!VNL_WAVE_ANY(D != vreal(0.))) continue;
      vreal x = W(jaref + r);
      c += x < vreal(0.) ? D * x * x : vreal(0.);
    }
    return vreal(0.5) * c;
  }

  // qfrc_constraint = J' f with f = -D Jaref [Jaref<0]; returns the constraint cost
  VNL_HD vreal constraint_force() const {
    V3 n = v3(m.pnx, m.pny, m.pnz);
    int Wb = L.bodyC + 6 * m.nbody;
    vreal cost = vreal(0.);
    for (int k = 6; k < 6 * m.nbody; k++) W(Wb + k) = vreal(0.);
    for (int d = 0; d < m.nv; d++) W(L.qfrc_c + d) = vreal(0.);
    for (int r = 0; r < m.nlimit; r++) {
      vreal D = W(L.efc_D + r);
      if (!VNL_WAVE_ANY(D != vreal(0.))) continue;
      vreal x = W(L.Jaref + r);
      vreal f = x < vreal(0.) ? -D * x : vreal(0.);
      cost += x < vreal(0.) ? D * x * x : vreal(0.);
      W(L.qfrc_c + m.lim_dof[r]) += W(L.lim_sign + r) * f;
    }
    for (int g = 0; g < m.ncg; g++) {
      int bd = m.cg_body[g], c0 = m.cg_conadr[g], nc = m.cg_ncon[g];
      vreal mu = m.cg_mu[g];
      for (int s = 0; s < nc; s++) {
        int c = c0 + s, r0 = m.nlimit + 4 * c;
        vreal D = W(L.efc_D + r0);
        if (!VNL_WAVE_ANY(D != vreal(0.))) continue;
        vreal f[4];
        for (int q = 0; q < 4; q++) {
          vreal x = W(L.Jaref + r0 + q);
          f[q] = x < vreal(0.) ? -D * x : vreal(0.);
          cost += x < vreal(0.) ? D * x * x : vreal(0.);
        }
        V3 rel = ld3(L.con_r + 3 * c), t1 = ld3(L.con_t1 + 3 * c), t2 = cross(n, t1);
        V3 Fw = n * (f[0] + f[1] + f[2] + f[3]) + t1 * (mu * (f[0] - f[1])) + t2 * (mu * (f[2] - f[3]));
        S6 w = S6{cross(rel, Fw), Fw};
        st6(Wb + 6 * bd, ld6(Wb + 6 * bd) + w);
      }
    }
    for (int b = m.nbody - 1; b > 1; b--) {
      int p = m.body_parent[b];
      if (p > 0)
        for (int k = 0; k < 6; k++) W(Wb + 6 * p + k) += W(Wb + 6 * b + k);
    }
    for (int d = 0; d < m.nv; d++) W(L.qfrc_c + d) += dot(ld6(L.cdof + 6 * d), ld6(Wb + 6 * m.dof_body[d]));
    return vreal(0.5) * cost;
  }

  VNL_HD vreal vdot(int a, int b) const {
    vreal s = vreal(0.);
    for (int d = 0; d < m.nv; d++) s += W(a + d) * W(b + d);
    return s;
  }

  // one pass over the rows for up to 3 step lengths: quad totals -> (cost, d0, d1)
  struct LsPoint {
    vreal alpha, cost, d0, d1;
  };
  template <int N>
  VNL_HD void ls_eval(const vreal* alpha, vreal qg0, vreal qg1, vreal qg2, LsPoint* out) const {
    vreal q0[N], q1[N], q2[N];
    for (int i = 0; i < N; i++) q0[i] = qg0, q1[i] = qg1, q2[i] = qg2;
    for (int r = 0; r < m.nefc; r++) {
      vreal D = W(L.efc_D + r);
      if (!VNL_WAVE_ANY(D != vreal(0.))) continue;
      vreal ja = W(L.Jaref + r), jv = W(L.jv + r);
      vreal a0 = vreal(0.5) * ja * ja * D, a1 = jv * ja * D, a2 = vreal(0.5) * jv * jv * D;
      for (int i = 0; i < N; i++) {
        bool act = ja + alpha[i] * jv < vreal(0.);
        q0[i] += act ? a0 : vreal(0.), q1[i] += act ? a1 : vreal(0.), q2[i] += act ? a2 : vreal(0.);
      }
    }
    for (int i = 0; i < N; i++) {
      vreal a = alpha[i];
      out[i].alpha = a;
      out[i].cost = a * a * q2[i] + a * q1[i] + q0[i];
      out[i].d0 = vreal(2.) * a * q2[i] + q1[i];
      out[i].d1 = vreal(2.) * q2[i] + (q2[i] == vreal(0.) ? VNL_MINVAL : vreal(0.));
    }
  }

  // solver.solve (CG) -- per-lane freezing reproduces vmap-of-while semantics
  VNL_HD void solve() const {
    const int nv = m.nv, ne = m.nefc;
    // --- warm start selection: cost at qacc_warmstart vs qacc_smooth
    for (int d = 0; d < nv; d++) W(L.qacc + d) = ST(warm, d);
    body_twists(L.qacc);
    jac_mul(L.qacc, L.jv, true);  // Jaref(warm) in jv
    mass_mul(L.qacc, L.mv);       // Ma(warm) in mv
    vreal gw = vreal(0.);
    for (int d = 0; d < nv; d++) gw += (W(L.mv + d) - W(L.smooth + d)) * (W(L.qacc + d) - W(L.qacc_smooth + d));
    vreal cost_w = constraint_cost(L.jv) + vreal(0.5) * gw;
    body_twists(L.qacc_smooth);
    jac_mul(L.qacc_smooth, L.Jaref, true);
    vreal cost_s = constraint_cost(L.Jaref);  // gauss term vanishes: M qacc_smooth = qfrc_smooth
    bool use_warm = cost_w < cost_s;
    for (int d = 0; d < nv; d++) {
      W(L.qacc + d) = use_warm ? W(L.qacc + d) : W(L.qacc_smooth + d);
      W(L.Ma + d) = use_warm ? W(L.mv + d) : W(L.smooth + d);
    }
    if (VNL_WAVE_ANY(use_warm))
      for (int r = 0; r < ne; r++) W(L.Jaref + r) = use_warm ? W(L.jv + r) : W(L.Jaref + r);
    vreal gauss = use_warm ? vreal(0.5) * gw : vreal(0.);
    vreal cost = constraint_force() + gauss;
    vreal prev_cost = INFINITY;
    for (int d = 0; d < nv; d++) {
      vreal g = W(L.Ma + d) - W(L.smooth + d) - W(L.qfrc_c + d);
      W(L.grad + d) = g, W(L.Mgrad + d) = g;
    }
    solve_inplace(L.Mgrad);
    for (int d = 0; d < nv; d++) W(L.search + d) = -W(L.Mgrad + d);
    VNL_PROF(6);

    bool done = false;
    for (int it = 0; it < m.iterations; it++) {
      vreal improvement = (prev_cost - cost) / m.scale;
      vreal gradient = sqrt(vdot(L.grad, L.grad)) / m.scale;
      done = done || (improvement < m.tolerance) || (gradient < m.tolerance);
      if (!VNL_WAVE_ANY(!done)) break;
      const bool run = !done;
      // ---- line search
      vreal smag = sqrt(vdot(L.search, L.search)) * m.scale;
      vreal gtol = m.tolerance * m.ls_tolerance * smag;
      body_twists(L.search);
      jac_mul(L.search, L.jv, false);
      mass_mul(L.search, L.mv);
      vreal qg1 = vreal(0.), qg2 = vreal(0.);
      for (int d = 0; d < nv; d++) {
        vreal s = W(L.search + d);
        qg1 += s * W(L.Ma + d) - s * W(L.smooth + d);
        qg2 += s * W(L.mv + d);
      }
      qg2 *= vreal(0.5);
      VNL_PROF(7);
      LsPoint p0, lo, hi;
      vreal a1[1] = {vreal(0.)};
      ls_eval<1>(a1, gauss, qg1, qg2, &p0);
      a1[0] = p0.alpha - p0.d0 / p0.d1;
      ls_eval<1>(a1, gauss, qg1, qg2, &lo);
      if (lo.d0 < p0.d0) {
        hi = p0;
      } else {
        hi = lo, lo = p0;
      }
      bool ls_done = !run, swap = true;
      for (int li = 0; li < m.ls_iterations; li++) {
        ls_done = ls_done || !swap || (lo.d0 < vreal(0.) && lo.d0 > -gtol) || (hi.d0 > vreal(0.) && hi.d0 < gtol);
        if (!VNL_WAVE_ANY(!ls_done)) break;
        vreal a3[3] = {lo.alpha - lo.d0 / lo.d1, hi.alpha - hi.d0 / hi.d1, vreal(0.5) * (lo.alpha + hi.alpha)};
        LsPoint p[3];
        ls_eval<3>(a3, gauss, qg1, qg2, p);
        LsPoint nlo = lo, nhi = hi;
        bool s1 = (nlo.d0 > vreal(0.)) || (nlo.d0 < p[0].d0);
        if (s1) nlo = p[0];
        bool s2 = (p[2].d0 < vreal(0.)) && (nlo.d0 < p[2].d0);
        if (s2) nlo = p[2];
        bool s3 = (nhi.d0 < vreal(0.)) || (nhi.d0 > p[1].d0);
        if (s3) nhi = p[1];
        bool s4 = (p[2].d0 > vreal(0.)) && (nhi.d0 > p[2].d0);
        if (s4) nhi = p[2];
        if (!ls_done) lo = nlo, hi = nhi, swap = s1 || s2 || s3 || s4;
      }
      bool improved = run && ((lo.cost < p0.cost) || (hi.cost < p0.cost));
      vreal alpha = improved ? (lo.cost < hi.cost ? lo.alpha : hi.alpha) : vreal(0.);
      for (int d = 0; d < nv; d++) {
        W(L.qacc + d) += alpha * W(L.search + d);
        W(L.Ma + d) += alpha * W(L.mv + d);
      }
      for (int r = 0; r < ne; r++) W(L.Jaref + r) += alpha * W(L.jv + r);
      VNL_PROF(8);
      // ---- constraint + gradient update
      vreal gp = vdot(L.grad, L.Mgrad);
      vreal g = vreal(0.);
      for (int d = 0; d < nv; d++) g += (W(L.Ma + d) - W(L.smooth + d)) * (W(L.qacc + d) - W(L.qacc_smooth + d));
      vreal ncost = constraint_force() + vreal(0.5) * g;
      if (run) prev_cost = cost, cost = ncost, gauss = vreal(0.5) * g;
      vreal d1 = vreal(0.);
      for (int d = 0; d < nv; d++) {
        vreal gn = W(L.Ma + d) - W(L.smooth + d) - W(L.qfrc_c + d);
        d1 += gn * W(L.Mgrad + d);
        W(L.grad + d) = gn, W(L.tmp + d) = gn;
      }
      VNL_PROF(9);
      solve_inplace(L.tmp);
      VNL_PROF(10);
      vreal d2 = vdot(L.grad, L.tmp);
      vreal beta = fmax(vreal(0.), (d2 - d1) / fmax(VNL_MINVAL, gp));
      for (int d = 0; d < nv; d++) {
        vreal mg = W(L.tmp + d);
        W(L.Mgrad + d) = mg;
        if (run) W(L.search + d) = -mg + beta * W(L.search + d);
      }
    }
    for (int d = 0; d < nv; d++) ST(warm, d) = W(L.qacc + d);
  }

  // forward.forward
  VNL_HD void forward() const {
    kinematics();
    VNL_PROF(0);
    crb_mass_matrix();
    VNL_PROF(1);
    factor(vreal(0.));
    VNL_PROF(2);
    bias_forces();
    VNL_PROF(3);
    smooth_forces();
    VNL_PROF(4);
    make_constraint();
    VNL_PROF(5);
    solve();
  }

  // forward.euler + _advance
  VNL_HD void euler() const {
    const int nv = m.nv;
    for (int d = 0; d < nv; d++) W(L.tmp + d) = m.eulerdamp ? W(L.smooth + d) + W(L.qfrc_c + d) : W(L.qacc + d);
    VNL_PROF(11);
    if (m.eulerdamp) {
      factor(m.dt);
      VNL_PROF(12);
      solve_inplace(L.tmp);
    }
    for (int i = 0; i < m.nu; i++)
      if (m.act_tau[i] >= vreal(0.)) ST(act, i) += W(L.actdot + i) * m.dt;
    for (int d = 0; d < nv; d++) ST(qvel, d) += W(L.tmp + d) * m.dt;
    for (int j = 0; j < m.njnt; j++) {
      int qa = m.jnt_qposadr[j], da = m.jnt_dofadr[j];
      if (m.jnt_type[j] == VNL_JNT_FREE) {
        for (int t = 0; t < 3; t++) ST(qpos, qa + t) += m.dt * ST(qvel, da + t);
        V3 w = V3{ST(qvel, da + 3), ST(qvel, da + 4), ST(qvel, da + 5)};
        vreal nrm = sqrt(dot(w, w));
        V3 ax = nrm > vreal(0.) ? w * (vreal(1.) / nrm) : w;
        vreal ang = m.dt * nrm, s = sin(vreal(0.5) * ang), c = cos(vreal(0.5) * ang);
        Q4 q = Q4{ST(qpos, qa + 3), ST(qpos, qa + 4), ST(qpos, qa + 5), ST(qpos, qa + 6)};
        Q4 r = qmul(q, Q4{c, ax.x * s, ax.y * s, ax.z * s});
        vreal n2 = sqrt(r.w * r.w + r.x * r.x + r.y * r.y + r.z * r.z);
        vreal inv = n2 > vreal(0.) ? vreal(1.) / n2 : vreal(1.);
        ST(qpos, qa + 3) = r.w * inv, ST(qpos, qa + 4) = r.x * inv, ST(qpos, qa + 5) = r.y * inv,
                 ST(qpos, qa + 6) = r.z * inv;
      } else {
        ST(qpos, qa) += m.dt * ST(qvel, da);
      }
    }
    VNL_PROF(13);
  }

  // ------------------------------------------------------------------ env glue
  VNL_HD static int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

  // rodent.py:241-264 (matrix 1-norm over the tracked bodies, L1 over joints)
  VNL_HD vreal termination(int clip, int frame) const {
    int f = clampi(frame, 0, ev.T - 1), nj = m.nq - 7;
    const float* cj = ev.joints + ((size_t)clip * ev.T + f) * nj;
    vreal ej = vreal(0.);
    for (int i = 0; i < nj; i++) ej += fabs(cj[i] - ST(qpos, 7 + i));
    const float* cb = ev.body_positions + ((size_t)clip * ev.T + f) * ev.nb * 3;
    vreal cx = vreal(0.), cy = vreal(0.), cz = vreal(0.);
    for (int k = 0; k < ev.nb; k++) {
      int bd = ev.body_idxs[k];
      cx += fabs(cb[3 * k] - ST(xpos, 3 * bd));
      cy += fabs(cb[3 * k + 1] - ST(xpos, 3 * bd + 1));
      cz += fabs(cb[3 * k + 2] - ST(xpos, 3 * bd + 2));
    }
    vreal eb = fmax(cx, fmax(cy, cz));
    vreal err = vreal(0.5) * ev.body_err_mult * eb + vreal(0.5) * ej;
    return vreal(1.) - err * ev.inv_term_threshold;
  }

  // rodent.py:318-344
  VNL_HD void write_obs() const {
    int k = 0;
    for (int i = 0; i < m.nq; i++) ST(obs, k++) = nan0(ST(qpos, i));
    for (int i = 0; i < m.nv; i++) ST(obs, k++) = nan0(ST(qvel, i));
    for (int i = 0; i < m.nv; i++) ST(obs, k++) = nan0(ST(qfrc_actuator, i));
    for (int j = 0; j < ev.nee; j++) {
      int bd = ev.end_eff_idx[j];
      for (int i = 0; i < 3; i++) ST(obs, k++) = nan0(ST(xpos, 3 * bd + i));
    }
  }
  // jp.nan_to_num
  VNL_HD static vreal nan0(vreal x) {
    if (x != x) return vreal(0.);
    if (x > vreal(3.4028235e38)) return vreal(3.4028235e38);
    if (x < -vreal(3.4028235e38)) return -vreal(3.4028235e38);
    return x;
  }

  // rodent.py:346-448; local frame = v @ xmat[1]
  VNL_HD void write_traj(int clip, int frame) const {
    int Lr = ev.ref_len, s = clampi(frame + 1, 0, ev.T - Lr), nj = m.nq - 7, k = 0;
    M3 R = qmat(Q4{ST(xquat, 4), ST(xquat, 5), ST(xquat, 6), ST(xquat, 7)});
    size_t fb = (size_t)clip * ev.T + s;
    for (int t = 0; t < Lr; t++) {
      const float* cb = ev.body_positions + (fb + t) * ev.nb * 3;
      for (int a = 0; a < ev.napp; a++) {
        int col = ev.app_ref_col[a];
        for (int i = 0; i < 3; i++) ST(traj, k++) = cb[3 * col + i];
      }
    }
    int kg = k + Lr * ev.nb * 3;  // global block follows the local block
    for (int t = 0; t < Lr; t++) {
      const float* cb = ev.body_positions + (fb + t) * ev.nb * 3;
      for (int b = 0; b < ev.nb; b++) {
        int bd = ev.body_idxs[b];
        V3 v = V3{cb[3 * b] - ST(xpos, 3 * bd), cb[3 * b + 1] - ST(xpos, 3 * bd + 1), cb[3 * b + 2] - ST(xpos, 3 * bd + 2)};
        ST(traj, k++) = v.x * R.a[0] + v.y * R.a[3] + v.z * R.a[6];
        ST(traj, k++) = v.x * R.a[1] + v.y * R.a[4] + v.z * R.a[7];
        ST(traj, k++) = v.x * R.a[2] + v.y * R.a[5] + v.z * R.a[8];
        ST(traj, kg++) = v.x, ST(traj, kg++) = v.y, ST(traj, kg++) = v.z;
      }
    }
    k = kg;
    for (int t = 0; t < Lr; t++) {
      const float* cp = ev.position + (fb + t) * 3;
      V3 v = V3{cp[0] - ST(qpos, 0), cp[1] - ST(qpos, 1), cp[2] - ST(qpos, 2)};
      ST(traj, k++) = v.x * R.a[0] + v.y * R.a[3] + v.z * R.a[6];
      ST(traj, k++) = v.x * R.a[1] + v.y * R.a[4] + v.z * R.a[7];
      ST(traj, k++) = v.x * R.a[2] + v.y * R.a[5] + v.z * R.a[8];
    }
    for (int t = 0; t < Lr; t++) {
      const float* cj = ev.joints + (fb + t) * nj;
      for (int j = 0; j < ev.njc; j++) {
        int col = ev.joint_cols[j];
        ST(traj, k++) = cj[col] - ST(qpos, 7 + col);
      }
    }
  }

  VNL_HD bool any_nan() const {
    bool bad = false;
    for (int i = 0; i < m.nq; i++) bad |= ST(qpos, i) != ST(qpos, i);
    for (int i = 0; i < m.nv; i++) {
      vreal a = ST(qvel, i), b = ST(warm, i), c = ST(qfrc_actuator, i);
      bad |= (a != a) | (b != b) | (c != c);
    }
    for (int i = 0; i < m.nu; i++) bad |= ST(act, i) != ST(act, i);
    for (int i = 0; i < 3 * m.nbody; i++) bad |= ST(xpos, i) != ST(xpos, i);
    for (int i = 0; i < 3; i++) bad |= ST(com1, i) != ST(com1, i);
    return bad;
  }

  // RodentTracking.reset, rodent.py:119-176 (start_frame / noise supplied by the caller)
  VNL_HD void reset(const int* start_frame, const vreal* noise) const {
    int clip = st.clip_id[e], sf = start_frame[e];
    int f = clampi(sf, 0, ev.T - 1), nj = m.nq - 7;
    size_t fb = (size_t)clip * ev.T + f;
    for (int k = 0; k < 3; k++) ST(qpos, k) = ev.position[fb * 3 + k] + noise[(unsigned)k * B + e];
    for (int k = 0; k < 4; k++) ST(qpos, 3 + k) = ev.quaternion[fb * 4 + k] + noise[(unsigned)(3 + k) * B + e];
    for (int k = 0; k < nj; k++) ST(qpos, 7 + k) = ev.joints[fb * nj + k] + noise[(unsigned)(7 + k) * B + e];
    for (int k = 0; k < 3; k++) ST(qvel, k) = ev.velocity[fb * 3 + k];
    for (int k = 0; k < 3; k++) ST(qvel, 3 + k) = ev.angular_velocity[fb * 3 + k];
    for (int k = 0; k < nj; k++) ST(qvel, 6 + k) = ev.joints_velocity[fb * nj + k];
    for (int i = 0; i < m.nu; i++) ST(act, i) = vreal(0.), W(L.ctrl + i) = vreal(0.);
    for (int d = 0; d < m.nv; d++) ST(warm, d) = vreal(0.);
    forward();
    write_traj(clip, sf);
    write_obs();
    st.reward[e] = vreal(0.), st.done[e] = vreal(0.);
    for (int k = 0; k < 7; k++) ST(metrics, k) = vreal(0.);
    st.cur_frame[e] = sf, st.sub_clip_frame[e] = 0;
    st.term_err[e] = termination(clip, sf);
  }

  // RodentTracking.step, rodent.py:178-239
  VNL_HD void step(const vreal* action) const {
    prof_begin();
    int clip = st.clip_id[e], old_frame = st.cur_frame[e];
    // rtrunk from the OLD pipeline state and OLD frame (rodent.py:250-262, 296)
    vreal rtrunk = termination(clip, old_frame);
    for (int i = 0; i < m.nu; i++) {
      vreal c = action[(unsigned)i * B + e];
      if (m.act_limited[i]) c = fmin(fmax(c, m.act_lo[i]), m.act_hi[i]);
      W(L.ctrl + i) = c;
    }
    for (int f = 0; f < ev.n_frames; f++) {
      forward();
      euler();
    }
    int new_frame = old_frame + 1, new_sub = st.sub_clip_frame[e] + 1;
    int fo = clampi(old_frame, 0, ev.T - 1), nj = m.nq - 7;
    size_t fb = (size_t)clip * ev.T + fo;
    // _calculate_reward: NEW data vs clip row at OLD frame (rodent.py:266-316)
    const float* cb = ev.body_positions + fb * ev.nb * 3;
    V3 dc = V3{ST(com1, 0) - cb[3 * ev.com_ref_col], ST(com1, 1) - cb[3 * ev.com_ref_col + 1],
               ST(com1, 2) - cb[3 * ev.com_ref_col + 2]};
    vreal rcom = exp(-vreal(100.) * sqrt(dot(dc, dc)));
    vreal acc = vreal(0.);
    for (int k = 0; k < 3; k++) {
      vreal a = ST(qvel, k) - ev.velocity[fb * 3 + k], b = ST(qvel, 3 + k) - ev.angular_velocity[fb * 3 + k];
      acc += a * a + b * b;
    }
    for (int k = 0; k < nj; k++) {
      vreal a = ST(qvel, 6 + k) - ev.joints_velocity[fb * nj + k];
      acc += a * a;
    }
    vreal rvel = exp(-vreal(0.1) * sqrt(acc));
    vreal nc = vreal(0.), nr = vreal(0.), dq = vreal(0.);
    for (int k = 0; k < 4; k++) {
      vreal a = ST(qpos, 3 + k), b = ev.quaternion[fb * 4 + k];
      nc += a * a, nr += b * b, dq += a * b;
    }
    dq = dq / (sqrt(nc) * sqrt(nr));
    vreal dist = fmin(vreal(1.), vreal(2.) * dq * dq - vreal(1.));
    vreal rquat = exp(-vreal(2.) * fabs(vreal(0.5) * acos(dist)));
    acc = vreal(0.);
    for (int d = 0; d < m.nv; d++) acc += ST(qfrc_actuator, d) * ST(qfrc_actuator, d);
    vreal ract = -vreal(0.015) * (acc / (vreal)m.nv);
    acc = vreal(0.);
    for (int a = 0; a < ev.napp; a++) {
      int bd = ev.app_body[a], col = ev.app_ref_col[a];
      for (int k = 0; k < 3; k++) {
        vreal x = ST(xpos, 3 * bd + k) - cb[3 * col + k];
        acc += x * x;
      }
    }
    vreal rapp = exp(-vreal(400.) * sqrt(acc));
    vreal z = ST(qpos, 2);
    vreal healthy = (z < ev.healthy_lo || z > ev.healthy_hi) ? vreal(0.) : vreal(1.);
    rcom *= vreal(0.01), rvel *= vreal(0.01), rapp *= vreal(0.01), rtrunk *= vreal(0.01), rquat *= vreal(0.01), ract *= vreal(0.0001);
    vreal total = rcom + rvel + rtrunk + rquat + ract + rapp;
    vreal done = rtrunk < vreal(0.) ? vreal(1.) : vreal(0.);
    done = fmax(vreal(1.) - healthy, done);
    done = fmax(new_sub < ev.sub_clip_length ? vreal(0.) : vreal(1.), done);
    if (any_nan()) done = vreal(1.);
    write_obs();
    write_traj(clip, new_frame);
    st.reward[e] = nan0(total), st.done[e] = done;
    ST(metrics, 0) = rcom, ST(metrics, 1) = rvel, ST(metrics, 2) = rtrunk, ST(metrics, 3) = rquat;
    ST(metrics, 4) = ract, ST(metrics, 5) = rapp, ST(metrics, 6) = rtrunk;
    st.cur_frame[e] = new_frame, st.sub_clip_frame[e] = new_sub;
    st.term_err[e] = rtrunk;
    VNL_PROF(14);
    prof_end();
  }
#undef ST
};
