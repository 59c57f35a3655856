// One environment per wavefront, all per-env state resident in LDS.
//
// A 64-lane workgroup owns one env for a whole control step (n_frames physics substeps +
// observation / reference-trajectory / reward / termination glue).  The code is fork-join
// over the wave:
//     VNL_FOR(i, n)  { ... }   parallel-for over the 64 lanes (rows, matrix entries, bodies,
//                              contacts, dofs), followed by VNL_SYNC()
//     VNL_SERIAL     { ... }   one lane walks the kinematic tree (parent -> child chains)
//     wave_sum(x)              cross-lane reduction; scalars derived from it are wave-uniform
// Every array that crosses a region lives in LDS (`s[...]`, 20.2 KB per env for the rodent: 8 envs per CU);
// HBM is touched only to load the carried state and to store the new state / obs / traj, with
// row-major [env][feature] buffers so those accesses are contiguous per wave.
//
// What each stage follows:
//   physics  : MJX forward/step [UPSTREAM mjx/_src/{smooth,collision_primitive,constraint,
//              solver,passive,forward}.py], called from reference envs/rodent.py:148
//              (pipeline_init) and :181 (pipeline_step)
//   env glue : reference envs/rodent.py:178-470
//
// Algorithmic choices that differ from MJX's dense route (same mathematics; checked against
// the dense float64 oracle by the float64 host build, tests/test_hostsim_parity.py):
//   * spatial quantities (cdof, cinert, cvel ...) are expressed about the root body's origin
//     O instead of subtree_com[root]; the reference point cancels in every scalar produced;
//   * qM is built tree-sparse (MuJoCo's dof_Madr layout, 1119 entries for the rodent) directly
//     into the factor buffer and factorised in place as L'DL (mj_factorM order);
//   * efc_J is never materialised: J*v and J'*f go through the kinematic tree (body twists
//     forward / body wrenches backward);
//   * inside the CG loop M*search is carried by the recurrence M s_new = -grad + beta * M s_old
//     (search = -M^-1 grad + beta search); the only explicit product, M*qacc_warmstart, uses
//     the factor (L' D L).
#pragma once
#include <math.h>

#include "vnl_types.h"

#ifndef VNL_HD
#define VNL_HD __device__ __forceinline__
#endif

// ---- fork-join primitives (the host simulation in tests/hostsim redefines them) ------------
#ifndef VNL_FORKJOIN_DEFINED
#define VNL_LANES 64
#define VNL_ROWS_PER_LANE 8 /* constraint rows a lane keeps in registers during a line search: nefc <= 512 */
#define VNL_ROWS_SMALL 5    /* specialisation for nefc <= 320 (the rodent has 303) */
#define VNL_CHAIN_WIDTH 8 /* entries of a sparse row / column fetched per trip of row_dot / col_apply */
#define VNL_ROWSETS_1 1 /* matrix rows a lane keeps in registers while factorising: nv <= 64 .. */
#define VNL_ROWSETS_2 2 /* .. or nv <= 128 */
#define VNL_FOR(i, n) for (int i = (int)lane; i < (n); i += VNL_LANES)
#define VNL_SERIAL if (lane == 0)
// Join of a fork-join region.  The workgroup IS one wave: its LDS operations execute in issue order, so data one lane wrote
// to LDS is what any lane's later ds_read returns -- the region boundary needs no wait at all (what __syncthreads() compiles
// to here is no s_barrier but a full `s_waitcnt vmcnt(0) lgkmcnt(0)`: an exposed LDS round trip at each of the several
// hundred joins of a substep), only that the compiler keeps the accesses in order: the wavefront-scope fence, no instruction.
#define VNL_SYNC() __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront")
// .. and the join behind GLOBAL memory that one lane wrote and another lane will read (xpos / xquat in the state buffers:
// kinematics -> make_constraint and the env glue; the second factor's scratch: invert_aba -> euler): the vector-memory
// counter is drained, so the stores have reached the cache the loads are served from
#define VNL_SYNC_GLOBAL()                                     \
  do {                                                        \
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");    \
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          \
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");    \
  } while (0)
#define VNL_LDS_DECL(name) extern __shared__ __align__(16) vreal name[]
// Cross-lane sum with DPP (VALU-rate) instead of ds_bpermute: xor-butterfly inside each row of 16 lanes (quad_perm,
// row_half_mirror, row_mirror: every lane then holds its row's total), the row totals combined by row_bcast:15 (rows 1, 3
// take rows 0, 2) and row_bcast:31 (row 3 takes rows 0+1), the grand total read from lane 63: 6 DPP adds + 1 v_readlane.
// Every lane returns the same value.
VNL_HD float vnl_wave_sum(float x) {
#define VNL_DPP_STEP(ctrl) x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), ctrl, 0xf, 0xf, true))
  VNL_DPP_STEP(0xB1);   // quad_perm [1,0,3,2]
  VNL_DPP_STEP(0x4E);   // quad_perm [2,3,0,1]
  VNL_DPP_STEP(0x141);  // row_half_mirror
  VNL_DPP_STEP(0x140);  // row_mirror
#undef VNL_DPP_STEP
  x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x142, 0xa, 0xf, false));  // row_bcast:15
  x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x143, 0xc, 0xf, false));  // row_bcast:31
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), 63));
}
// .. of values that are zero outside the first 16 lanes (the line search when at most 16 constraint rows exist, one per lane): the
// butterfly inside the first row of 16 is the whole sum -- 4 DPP adds + 1 v_readlane
VNL_HD float vnl_wave_sum16(float x) {
#define VNL_DPP_STEP(ctrl) x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), ctrl, 0xf, 0xf, true))
  VNL_DPP_STEP(0xB1);   // quad_perm [1,0,3,2]
  VNL_DPP_STEP(0x4E);   // quad_perm [2,3,0,1]
  VNL_DPP_STEP(0x141);  // row_half_mirror
  VNL_DPP_STEP(0x140);  // row_mirror
#undef VNL_DPP_STEP
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), 0));
}
VNL_HD bool vnl_wave_any(bool x) { return __ballot(x) != 0ull; }
// inclusive prefix sum over the 64 lanes: Hillis-Steele inside each row of 16 (row_shr 1, 2, 4, 8), then the row
// totals are carried over with row_bcast:15 (rows 1, 3) and row_bcast:31 (rows 2, 3)
VNL_HD float vnl_wave_scan(float x) {
#define VNL_DPP_SCAN(ctrl, rmask) \
  x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), ctrl, rmask, 0xf, false))
  VNL_DPP_SCAN(0x111, 0xf);  // row_shr:1
  VNL_DPP_SCAN(0x112, 0xf);  // row_shr:2
  VNL_DPP_SCAN(0x114, 0xf);  // row_shr:4
  VNL_DPP_SCAN(0x118, 0xf);  // row_shr:8
  VNL_DPP_SCAN(0x142, 0xa);  // row_bcast:15 -> rows 1 and 3
  VNL_DPP_SCAN(0x143, 0xc);  // row_bcast:31 -> rows 2 and 3
#undef VNL_DPP_SCAN
  return x;
}
#define VNL_SCAN_ADD(x, run) (x = vnl_wave_scan(x)) /* x <- sum over items <= this one (run: host simulation only) */
#define VNL_WAVE_ITEMS(n) VNL_LANES              /* trip count of a region in which every lane takes part in a scan */
#define VNL_PAD_ITEMS(n) (((n) + VNL_LANES - 1) / VNL_LANES * VNL_LANES) /* .. over several trips */
// scan carried across trips: x <- sum over all items up to this one; carry = total so far (the last lane's value)
#define VNL_SCAN_ADD_C(x, run, carry) \
  (x = vnl_wave_scan(x) + carry, carry = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), 63)))
// one value per lane (item j of a <= 64-item region lives in lane j) + uniform-index broadcast
#define VNL_PERLANE(T, name) T name
#define VNL_AT(name, j) name
#define VNL_GETF(name, a) __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, name), a))
#define VNL_GETI(name, a) __builtin_amdgcn_readlane(name, a)
// value that lane l holds for its q-th item (rows are dealt out as item = lane + 64 q)
#define VNL_ROWGETI(name, q, l) __builtin_amdgcn_readlane(name[q], l)
#define VNL_ROWGETF(expr, q, l) __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, expr), l))
// sum over the lanes of a SEGMENT of adjacent lanes inside one 16-lane DPP row, left in the segment's first lane (blk_apply):
// step k adds the value of lane + 2^k where the mask bit (cont >> k) & 1 says that lane still belongs to the segment
VNL_HD float vnl_seg_sum(float x, unsigned cont, int steps) {
#define VNL_DPP_SEG(ctrl, bit)                                                                                          \
  do {                                                                                                                  \
    const float t_ = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), ctrl, 0xf, 0xf, true)); \
    x += (cont & (bit)) ? t_ : 0.0f;                                                                                     \
  } while (0)
  if (steps > 0) VNL_DPP_SEG(0x101, 1u);  // row_shl:1
  if (steps > 1) VNL_DPP_SEG(0x102, 2u);  // row_shl:2
  if (steps > 2) VNL_DPP_SEG(0x104, 4u);  // row_shl:4
  if (steps > 3) VNL_DPP_SEG(0x108, 8u);  // row_shl:8
#undef VNL_DPP_SEG
  return x;
}
#define VNL_SEG_SUM(part, dsc, steps) (part = vnl_seg_sum(part, (dsc) >> 28, steps))
// orders the LDS accesses of the lanes of ONE wave (its LDS operations execute in issue order): no instruction
#define VNL_WAVE_FENCE() __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront")
// out[k] = (int)base[k * stride] for k < VNL_FAC_LINES: lane k loads, v_readlane broadcasts
#define VNL_LINE_HEADERS(out, base, stride)                                                       \
  do {                                                                                            \
    int hv_ = (int)(base)[((int)lane < VNL_FAC_LINES ? (int)lane : VNL_FAC_LINES - 1) * (stride)]; \
    _Pragma("unroll") for (int k_ = 0; k_ < VNL_FAC_LINES; k_++) out[k_] = __builtin_amdgcn_readlane(hv_, k_); \
  } while (0)
#define VNL_COUNT(pred) __popcll(__ballot(pred))                  /* lanes (items) for which pred holds */
/* stream compaction: rank = run + number of earlier items with pred; run += their total (run stays wave-uniform) */
#define VNL_RANK(pred, run, rank)                                                   \
  do {                                                                              \
    const unsigned long long b_ = __ballot(pred);                                   \
    rank = run + (int)__popcll(b_ & ((1ull << (threadIdx.x & 63)) - 1ull));         \
    run += (int)__popcll(b_);                                                       \
  } while (0)
#define VNL_UNIFORM_I(x) __builtin_amdgcn_readfirstlane(x)        /* a value known to be the same in every lane */
// 1/x: v_rcp_f32 (1 ulp) + one Newton step instead of the ~10-instruction IEEE division
VNL_HD float vnl_recip(float x) {
  float r = __builtin_amdgcn_rcpf(x);
  return r * (2.0f - x * r);
}
#endif

// Diagnostic build only (-DVNL_PROFILE, csrc/build.py --profile): per-stage s_memtime stamps summed
// into a __device__ array that no product code reads.  Never defined in the shipped library.
#ifdef VNL_PROFILE
#define VNL_NPROF 40
__device__ unsigned long long g_vnl_prof[VNL_NPROF];
// lane 0 keeps the running sums in LDS (section "prof", reserved in this build only) as floats (a substep is < 2^24
// clocks); slot VNL_NPROF = low word of the last stamp
#define VNL_PROF(i)                                                                             \
  do {                                                                                          \
    if (lane == 0) {                                                                            \
      unsigned t_ = (unsigned)__builtin_amdgcn_s_memtime();                                     \
      s[LO(prof) + (i)] += (float)(t_ - __builtin_bit_cast(unsigned, (float)s[LO(prof) + VNL_NPROF])); \
      s[LO(prof) + VNL_NPROF] = __builtin_bit_cast(float, (unsigned)__builtin_amdgcn_s_memtime()); \
    }                                                                                           \
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

// constant address space of the kernel-constant block (scalar loads, never clobbered); nothing special on the host
#if defined(__HIP_DEVICE_COMPILE__)
#define VNL_CAS __attribute__((address_space(4)))
#define VNL_TO_CAS(T, p) ((const VNL_CAS T*)(unsigned long long)(p))
#define VNL_LAUNDER(p) asm volatile("" : "+s"(p))
#else
#define VNL_CAS
#define VNL_TO_CAS(T, p) ((const T*)(p))
#define VNL_LAUNDER(p)
#endif
// SP: the compile-time model of this instantiation (csrc/vnl_types.h: VnlSpecGeneric reads every dimension and LDS offset
// from the constant block, VnlSpecRodent has them as constants)
#define MI(f) (SP::fixed ? SP::D.f : m.f)
#define LO(f) (SP::fixed ? SP::L.f : L.f)
template <class SP>
struct EnvWaveT {
  const VNL_CAS DevModel& m;
  const VNL_CAS DevEnv& ev;
  const DevState& st;
  const VNL_CAS WsLayout& L;
  vreal* s;  // LDS
  unsigned e, lane;
  const VNL_CAS KernelConsts* kc;
  int* trace;  // debug only (vnl_env_debug): VNL_TRACE_INTS ints for the solver call of the current forward pass, or null
  // The same view with the constant block's address made opaque to the optimiser: constants read by the stage that
  // follows are loaded there (scalar loads) instead of being kept alive -- spilled to VGPR lanes and fetched back
  // with v_readlane -- from the top of the kernel.
  VNL_HD EnvWaveT with_trace(int* t) const { return EnvWaveT{m, ev, st, L, s, e, lane, kc, t}; }
  VNL_HD EnvWaveT fresh() const {
    const VNL_CAS KernelConsts* k = kc;
    VNL_LAUNDER(k);
    return EnvWaveT{k->m, k->ev, st, k->L, s, e, lane, k, trace};
  }
#ifdef VNL_PROFILE
  VNL_HD void prof_begin() const {
    if (lane == 0) {
      for (int i = 0; i < VNL_NPROF; i++) s[LO(prof) + i] = 0.f;
      s[LO(prof) + VNL_NPROF] = __builtin_bit_cast(float, (unsigned)__builtin_amdgcn_s_memtime());
    }
  }
  VNL_HD void prof_end() const {
    if (lane == 0)
      for (int i = 0; i < VNL_NPROF; i++) atomicAdd(&g_vnl_prof[i], (unsigned long long)s[LO(prof) + i]);
  }
#else
  VNL_HD void prof_begin() const {}
  VNL_HD void prof_end() const {}
#endif

  VNL_HD V3 ld3(int o) const { return V3{s[o], s[o + 1], s[o + 2]}; }
  VNL_HD void st3(int o, V3 v) const { s[o] = v.x, s[o + 1] = v.y, s[o + 2] = v.z; }
  VNL_HD Q4 ld4(int o) const { return Q4{s[o], s[o + 1], s[o + 2], s[o + 3]}; }
  VNL_HD S6 ld6(int o) const { return S6{ld3(o), ld3(o + 3)}; }
  VNL_HD void st6(int o, S6 v) const { st3(o, v.a), st3(o + 3, v.l); }
  VNL_HD static V3 t3(const vreal* t, int i) { return V3{t[3 * i], t[3 * i + 1], t[3 * i + 2]}; }
  VNL_HD static Q4 t4(const vreal* t, int i) { return Q4{t[4 * i], t[4 * i + 1], t[4 * i + 2], t[4 * i + 3]}; }

  // cinert (10) x motion -> force   [MJX math.inert_mul]
  VNL_HD S6 inert_mul(int o, S6 v) const {
    vreal ixx = s[o], iyy = s[o + 1], izz = s[o + 2], ixy = s[o + 3], ixz = s[o + 4], iyz = s[o + 5];
    V3 mp = ld3(o + 6);
    vreal mass = s[o + 9];
    V3 ang = V3{ixx * v.a.x + ixy * v.a.y + ixz * v.a.z, ixy * v.a.x + iyy * v.a.y + iyz * v.a.z,
                ixz * v.a.x + iyz * v.a.y + izz * v.a.z} +
             cross(mp, v.l);
    V3 lin = v.l * mass - cross(mp, v.a);
    return S6{ang, lin};
  }

  // xpos / xquat / qfrc_actuator live in their (caller-owned, L2-resident) state buffers, not in LDS
  // (rows of xpos / xquat: the bodies of the model AS GIVEN -- welded bodies included; the dynamics' body b is row m.body_out[b])
  VNL_HD vreal* gxpos() const { return st.xpos + (size_t)e * 3 * MI(nbody_out); }
  VNL_HD vreal* gxquat() const { return st.xquat + (size_t)e * 4 * MI(nbody_out); }
  VNL_HD vreal* gqfrc_act() const { return st.qfrc_actuator + (size_t)e * MI(nv); }
  // library-owned global scratch of this env: the second inverse factor of a substep (invert_aba) and its reciprocal pivots (factor_aba), nM + nv elements
  VNL_HD vreal* fac2() const { return ev.fac2 + (size_t)e * (MI(nM) + MI(nv)); }
  VNL_HD V3 gpos3(int b) const {
    const vreal* x = gxpos() + 3 * b;
    return V3{x[0], x[1], x[2]};
  }
  VNL_HD Q4 gquat4(int b) const {
    const vreal* q = gxquat() + 4 * b;
    return Q4{q[0], q[1], q[2], q[3]};
  }
  VNL_HD V3 ref_point() const { return MI(root_free) ? ld3(LO(qpos)) : V3{m.root_px, m.root_py, m.root_pz}; }

  // ---- small index tables staged in LDS (dependent lookups in the sparse linear algebra and in the
  // tree walks would otherwise each pay a global / scalar-cache round trip)
  VNL_HD int anc_of(int k) const { return ((const unsigned char*)(s + LO(tab_anc)))[k]; }  // column dof of entry k
  VNL_HD int madr(int d) const { return ((const unsigned short*)(s + LO(tab_madr)))[d]; }            // first entry of row d
  VNL_HD int eadr(int d) const { return ((const unsigned short*)(s + LO(tab_madr)))[MI(nv) + d]; }     // madr(d) + depth(d)
  VNL_HD int depth(int d) const { return eadr(d) - madr(d); }
  VNL_HD int ndesc(int d) const { return ((const unsigned short*)(s + LO(tab_madr)))[2 * MI(nv) + d]; }       // descendants of dof d
  VNL_HD int parent_of(int b) const { return ((const unsigned char*)(s + LO(tab_body)))[b]; }
  VNL_HD int dofadr_of(int b) const { return ((const unsigned char*)(s + LO(tab_body)))[MI(nbody) + b]; }
  VNL_HD int dofnum_of(int b) const { return ((const unsigned char*)(s + LO(tab_body)))[2 * MI(nbody) + b]; }
  VNL_HD int con_body(int c) const { return ((const unsigned char*)(s + LO(tab_body)))[3 * MI(nbody) + c]; }
  // last dof on the path of contact c's body (== nv if the body hangs off the world without dofs)
  // compact list of the constraint rows that exist (after the active-contact list and the two counters)
  VNL_HD unsigned short* live_rows() const { return (unsigned short*)((int*)(s + LO(act_list)) + (MI(ncon) + 3) / 4 + 2); }
  VNL_HD int num_live_rows() const { return ((const int*)(s + LO(act_list)))[(MI(ncon) + 3) / 4 + 1]; }
  // body(r) for every constraint row that exists -- through the compact list when it holds them all (the rows make_constraint
  // masked out have D = 0 and never contribute to a cost, force or gradient, so their Jaref / jv need no upkeep)
  template <class F>
  VNL_HD void for_live_rows(F body) const {
    const int nl = num_live_rows();
    if (nl <= VNL_LIVE_MAX) {
      const unsigned short* lv = live_rows();
      VNL_FOR(l, nl) body((int)lv[l]);
    } else {
      VNL_FOR(r, MI(nefc)) body(r);
    }
  }
  VNL_HD int con_lastdof(int c) const { return ((const unsigned char*)(s + LO(tab_body)))[3 * MI(nbody) + MI(ncon) + c]; }
  // dofs sorted by depth: lvl_dof(q) for q in [lvl_start(l), lvl_start(l+1)) are the dofs of depth l
  VNL_HD int lvl_dof(int q) const { return ((const unsigned char*)(s + LO(tab_lvl)))[q]; }
  VNL_HD int lvl_start(int l) const { return ((const unsigned char*)(s + LO(tab_lvl)))[MI(nv) + l]; }
  VNL_HD int jump_of(int r, int b) const { return ((const unsigned char*)(s + LO(tab_jump)))[r * MI(nbody) + b]; }
  VNL_HD void load_tables() const {
    unsigned char* ta = (unsigned char*)(s + LO(tab_anc));
    VNL_FOR(k, MI(nM)) ta[k] = (unsigned char)m.M_anc[k];
    unsigned short* tm = (unsigned short*)(s + LO(tab_madr));
    VNL_FOR(d, MI(nv)) {
      tm[d] = (unsigned short)m.dof_Madr[d], tm[MI(nv) + d] = (unsigned short)(m.dof_Madr[d] + m.dof_depth[d]);
      tm[2 * MI(nv) + d] = (unsigned short)m.dof_ndesc[d];
    }
    unsigned char* tb = (unsigned char*)(s + LO(tab_body));
    VNL_FOR(b, MI(nbody)) {
      tb[b] = (unsigned char)m.body_parent[b];
      tb[MI(nbody) + b] = (unsigned char)m.body_dofadr[b];
      tb[2 * MI(nbody) + b] = (unsigned char)m.body_dofnum[b];
    }
    VNL_FOR(c, MI(ncon)) {
      int cb = m.cg_body[m.con_geom[c] & 0xff];
      tb[3 * MI(nbody) + c] = (unsigned char)cb;
      tb[3 * MI(nbody) + MI(ncon) + c] = (unsigned char)m.body_lastdof[cb];
    }
    unsigned char* tl = (unsigned char*)(s + LO(tab_lvl));
    VNL_FOR(q, MI(nv) + MI(max_depth) + 2) tl[q] = m.lvl_tab[q];
    unsigned char* tj = (unsigned char*)(s + LO(tab_jump));
    VNL_FOR(q, MI(jump_rounds) * MI(nbody)) tj[q] = m.jump[q];
    VNL_SYNC();
  }

  // ------------------------------------------------------------------ kinematics
  // smooth.kinematics in three fork-join phases:
  //  (1) per body, in parallel: its transform relative to the parent frame as a function of its own
  //      joint angles (MJX's anchor / off-centre rotation rule applied in the parent frame), plus the
  //      joint anchors / axes in that frame;
  //  (2) world poses = composition of the local transforms along each ancestor path, by pointer
  //      jumping (log2(depth) rounds) -- composition of rigid transforms is associative;
  //  (3) per dof, in parallel: cdof from the parent's world pose and the local anchor / axis.
  VNL_HD void kinematics() const {
    int A = LO(P), Bf = LO(P) + 7 * MI(nbody);  // 7 floats per body: pos(3) quat(4)
    // The model is one tree rooted at body 1 (vnl_lib.hip checks it), so a free joint can only sit there and every body
    // from 2 on carries hinges only.  Bodies 0 (world) and 1 are therefore dealt with apart: their poses are final after
    // phase 1 (no ancestor to compose with), and the loops over the OTHER bodies are hinge-only code over nbody - 2 items
    // -- 64 for the rodent: one trip per region instead of two (66 bodies) with both joint types' code in each.
    auto put_pose = [&](int base, int b, V3 pos, Q4 quat) {
      st3(base + 7 * b, pos);
      s[base + 7 * b + 3] = quat.w, s[base + 7 * b + 4] = quat.x, s[base + 7 * b + 5] = quat.y, s[base + 7 * b + 6] = quat.z;
    };
    auto hinge_chain = [&](int b, V3& pos, Q4& quat) {  // the body's hinges in turn: (axis, anchor) parked in cdof for phase 3
      const int jn = m.body_jntnum[b], ja = m.body_jntadr[b];
      for (int k = 0; k < jn; k++) {
        const int j = ja + k, qa = m.jnt_qposadr[j], da = m.jnt_dofadr[j];
        V3 jp = t3(m.jnt_pos, j), jax = t3(m.jnt_axis, j);
        V3 anchor = qrot(jp, quat) + pos;
        st6(LO(cdof) + 6 * da, S6{qrot(jax, quat), anchor});  // (axis, anchor) in the parent frame, for phase 3
        vreal ang = s[LO(qpos) + qa] - m.jnt_qpos0[j];
        vreal sn = sin(vreal(0.5) * ang), cs = cos(vreal(0.5) * ang);
        quat = qmul(quat, Q4{cs, jax.x * sn, jax.y * sn, jax.z * sn});
        pos = anchor - qrot(jp, quat);
      }
    };
    VNL_FOR(k, MI(nbody) - 2) {
      const int b = k + 2;
      V3 pos = t3(m.body_pos, b);
      Q4 quat = t4(m.body_quat, b);
      hinge_chain(b, pos, quat);
      put_pose(A, b, pos, quat);
    }
    VNL_SERIAL {  // the world body and the root body, into BOTH buffers of the composition rounds (which never touch them)
      put_pose(A, 0, v3(vreal(0.), vreal(0.), vreal(0.)), Q4{vreal(1.), vreal(0.), vreal(0.), vreal(0.)});
      put_pose(Bf, 0, v3(vreal(0.), vreal(0.), vreal(0.)), Q4{vreal(1.), vreal(0.), vreal(0.), vreal(0.)});
      if (MI(nbody) > 1) {
        V3 pos = t3(m.body_pos, 1);
        Q4 quat = t4(m.body_quat, 1);
        if (MI(root_free)) {
          const int qa = m.jnt_qposadr[m.body_jntadr[1]];
          pos = ld3(LO(qpos) + qa);
          quat = ld4(LO(qpos) + qa + 3);
          vreal n = sqrt(quat.w * quat.w + quat.x * quat.x + quat.y * quat.y + quat.z * quat.z);
          vreal inv = n > vreal(0.) ? vreal(1.) / n : vreal(1.);
          quat = Q4{quat.w * inv, quat.x * inv, quat.y * inv, quat.z * inv};
          s[LO(qpos) + qa + 3] = quat.w, s[LO(qpos) + qa + 4] = quat.x, s[LO(qpos) + qa + 5] = quat.y,
                          s[LO(qpos) + qa + 6] = quat.z;  // normalised quaternion written back
        } else {
          hinge_chain(1, pos, quat);
        }
        put_pose(A, 1, pos, quat);
        put_pose(Bf, 1, pos, quat);
      }
    }
    VNL_SYNC();
    int src = A, dst = Bf;
    for (int r = 0; r < MI(jump_rounds); r++) {
      VNL_FOR(k, MI(nbody) - 2) {
        const int b = k + 2;
        int j = jump_of(r, b);
        V3 pos = ld3(src + 7 * b);
        Q4 quat = ld4(src + 7 * b + 3);
        if (j > 0) {  // X_b <- X_j o X_b
          Q4 pq = ld4(src + 7 * j + 3);
          pos = ld3(src + 7 * j) + qrot(pos, pq);
          quat = qmul(pq, quat);
        }
        put_pose(dst, b, pos, quat);
      }
      VNL_SYNC();
      int t = src;
      src = dst, dst = t;
    }
    vreal* gx = gxpos();
    vreal* gq = gxquat();
    VNL_FOR(ob, MI(nbody_out)) {  // every body of the model as given: its dynamic body's pose, composed with its fixed transform
      const int b = m.out_dyn[ob];
      V3 p = ld3(src + 7 * b);
      Q4 q = ld4(src + 7 * b + 3);
      if (MI(nbody_out) != MI(nbody)) {  // (some bodies are welded: identity transforms for the others)
        p = p + qrot(t3(m.out_pos, ob), q);
        q = qmul(q, t4(m.out_quat, ob));
      }
      gx[3 * ob] = p.x, gx[3 * ob + 1] = p.y, gx[3 * ob + 2] = p.z;
      gq[4 * ob] = q.w, gq[4 * ob + 1] = q.x, gq[4 * ob + 2] = q.y, gq[4 * ob + 3] = q.z;
    }
    V3 O = ref_point();
    // cdof: the hinges (every joint but a free root's) ..
    const int j0 = MI(root_free) ? 1 : 0;
    VNL_FOR(k, MI(njnt) - j0) {
      const int j = k + j0, bd = m.jnt_body[j], da = m.jnt_dofadr[j];
      int p = parent_of(bd);
      Q4 pq = ld4(src + 7 * p + 3);
      S6 la = ld6(LO(cdof) + 6 * da);
      V3 axis = qrot(la.a, pq), anchor = ld3(src + 7 * p) + qrot(la.l, pq);
      st6(LO(cdof) + 6 * da, S6{axis, cross(axis, O - anchor)});
    }
    // .. and the free root's six
    if (MI(root_free)) {
      VNL_SERIAL {
        const int j = m.body_jntadr[1], bd = 1, da = m.jnt_dofadr[j];
        M3 R = qmat(ld4(src + 7 * bd + 3));
        V3 off = O - ld3(src + 7 * bd);
        for (int t = 0; t < 3; t++) {
          int o = LO(cdof) + 6 * (da + t);
          s[o] = vreal(0.), s[o + 1] = vreal(0.), s[o + 2] = vreal(0.);
          s[o + 3] = t == 0 ? vreal(1.) : vreal(0.), s[o + 4] = t == 1 ? vreal(1.) : vreal(0.),
                s[o + 5] = t == 2 ? vreal(1.) : vreal(0.);
        }
        for (int t = 0; t < 3; t++) {
          V3 ax = V3{R.a[t], R.a[3 + t], R.a[6 + t]};
          st6(LO(cdof) + 6 * (da + 3 + t), S6{ax, cross(ax, off)});
        }
      }
    }
    VNL_SYNC_GLOBAL();  // (xpos / xquat went to the state buffers: read across lanes from here on)
  }

  // com_pos: cinert about O in world axes, R I R' + m(|r|^2 1 - r r'), r = xipos - O -> pool[0..10 nbody); optionally com
  VNL_HD void body_inertias(bool with_com) const {
    V3 O = ref_point();
    V3 csum = v3(vreal(0.), vreal(0.), vreal(0.));
    VNL_FOR(b, MI(nbody)) {
      int o = LO(P) + 10 * b;
      if (b == 0) {
        for (int k = 0; k < 10; k++) s[o + k] = vreal(0.);
        continue;
      }
      const int ob = m.body_out[b];
      M3 R = qmat(gquat4(ob));
      vreal mass = m.body_mass[b];
      V3 xip = gpos3(ob) + mmul(R, t3(m.body_ipos, b));
      csum = csum + xip * mass;
      V3 r = xip - O;
      const vreal* I6 = m.body_inertia6 + 6 * b;  // xx yy zz xy xz yz (body axes, about ipos)
      vreal Ib[9] = {I6[0], I6[3], I6[4], I6[3], I6[1], I6[5], I6[4], I6[5], I6[2]};
      vreal T[9];
      for (int i = 0; i < 3; i++)
        for (int k = 0; k < 3; k++)
          T[3 * i + k] = R.a[3 * i] * Ib[k] + R.a[3 * i + 1] * Ib[3 + k] + R.a[3 * i + 2] * Ib[6 + k];
      vreal rr = dot(r, r);
      s[o + 0] = T[0] * R.a[0] + T[1] * R.a[1] + T[2] * R.a[2] + mass * (rr - r.x * r.x);
      s[o + 1] = T[3] * R.a[3] + T[4] * R.a[4] + T[5] * R.a[5] + mass * (rr - r.y * r.y);
      s[o + 2] = T[6] * R.a[6] + T[7] * R.a[7] + T[8] * R.a[8] + mass * (rr - r.z * r.z);
      s[o + 3] = T[0] * R.a[3] + T[1] * R.a[4] + T[2] * R.a[5] - mass * r.x * r.y;
      s[o + 4] = T[0] * R.a[6] + T[1] * R.a[7] + T[2] * R.a[8] - mass * r.x * r.z;
      s[o + 5] = T[3] * R.a[6] + T[4] * R.a[7] + T[5] * R.a[8] - mass * r.y * r.z;
      s[o + 6] = r.x * mass, s[o + 7] = r.y * mass, s[o + 8] = r.z * mass, s[o + 9] = mass;
    }
    if (with_com) {
      vreal cx = vnl_wave_sum(csum.x), cy = vnl_wave_sum(csum.y), cz = vnl_wave_sum(csum.z);
      VNL_SERIAL { st3(LO(com), v3(cx * m.total_mass_inv, cy * m.total_mass_inv, cz * m.total_mass_inv)); }
    }
    VNL_SYNC();
  }

  // Subtree sums (children into parents), one lane per component, bodies from the last to the first.  A child that
  // directly follows its parent in the numbering (a chain link: most links of a depth-first numbering) hands its
  // sum over in a REGISTER; only a child at a branching point adds into its parent through LDS.  (The plain form
  // `s[parent] += s[b]` made every one of the ~64 steps a dependent LDS read-modify-write.)
  VNL_HD void tree_accumulate(int base, int width) const {
    VNL_FOR(k, width) {
      vreal carry = vreal(0.);
      int b = MI(nbody) - 1;
      constexpr int G = 8;  // bodies per trip
      for (; b >= 1 && (b & (G - 1)) != G - 1; b--) {  // down to a group boundary
        const int o = base + width * b + k;
        const vreal v = s[o] + carry;
        const int p = parent_of(b);
        if (carry != vreal(0.)) s[o] = v;
        if (p == b - 1) {
          carry = v;
        } else {
          carry = vreal(0.);
          if (p > 0) s[base + width * p + k] += v;
        }
      }
      // G bodies per trip: their values and their parent bytes arrive in ONE LDS round trip; links inside the
      // group stay in registers, so only a branch child whose parent lies below the group pays another one.
      for (; b >= G - 1; b -= G) {
        const int b0 = b - (G - 1);
        const int o = base + width * b0 + k;
        unsigned pw[G / 4];
#pragma unroll
        for (int j = 0; j < G / 4; j++) pw[j] = (unsigned)VNL_UNIFORM_I((int)((const unsigned*)(s + LO(tab_body)))[(b0 >> 2) + j]);
        vreal x[G], x0[G];
#pragma unroll
        for (int j = 0; j < G; j++) x0[j] = x[j] = s[o + width * j];
#pragma unroll
        for (int j = G - 1; j >= 0; j--) {
          const int bb = b0 + j;
          if (bb >= 1) {
            const int p = (int)((pw[j >> 2] >> (8 * (j & 3))) & 255u);
            const vreal v = x[j] + carry;
            if (v != x0[j]) s[o + width * j] = v;
            if (p == bb - 1) {
              carry = v;
            } else {
              carry = vreal(0.);
              bool inside = false;
#pragma unroll
              for (int t = 0; t < G; t++)
                if (t < j - 1 && p == b0 + t) {
                  x[t] += v;
                  inside = true;
                }
              if (!inside && p > 0) s[base + width * p + k] += v;
            }
          }
        }
      }
    }
    VNL_SYNC();
  }

  // ------------------------------------------------------------------ inertia
  // smooth.crb + make_m straight into the factor buffer: LD <- qM + diag_scale * diag(damping).
  // Expects cinert in pool[0..10 nbody) (turned into crb in place).
  VNL_HD void mass_matrix(vreal diag_scale) const {
    tree_accumulate(LO(P), 10);
    VNL_PROF(5);
    // f_i = crb[body(i)] * cdof_i for every dof, parked in the six CG vectors Ma .. qfrc_c (contiguous,
    // dead whenever M is built); then one lane per run of consecutive matrix ENTRIES (not per row: rows
    // have 1 .. max_depth+1 entries): M(i, j) = f_i . cdof_j.
    const int F = LO(Ma);
    VNL_FOR(i, MI(nv)) st6(F + 6 * i, inert_mul(LO(P) + 10 * m.dof_body[i], ld6(LO(cdof) + 6 * i)));
    VNL_SYNC();
    const int per = (MI(nM) + VNL_LANES - 1) / VNL_LANES;
    VNL_FOR(l, VNL_LANES) {
      int e = l * per;
      const int e1 = e + per < MI(nM) ? e + per : MI(nM);
      if (e < e1) {
        int i = m.M_row[e], end = eadr(i);
        S6 f = ld6(F + 6 * i);
        for (; e < e1; e++) {
          if (e > end) {
            i++;
            end = eadr(i);
            f = ld6(F + 6 * i);
          }
          const int j = anc_of(e);
          s[LO(LD) + e] = dot(f, ld6(LO(cdof) + 6 * j));
        }
      }
    }
    VNL_SYNC();
    // armature and (Euler step) h * damping on the diagonals, one lane per dof: inside the entry loop these two table
    // reads were an L2 round trip in the middle of a serial chain
    VNL_FOR(i, MI(nv)) s[LO(LD) + madr(i)] += m.dof_armature[i] + diag_scale * m.dof_damping[i];
    VNL_SYNC();
    if (MI(solver_newton) && diag_scale == vreal(0.)) {  // dense symmetric copy of qM: the Newton solver's Hessian and M * search
      const int nv = MI(nv);
      VNL_FOR(k, nv * nv) s[LO(newt_M) + k] = vreal(0.);
      VNL_SYNC();
      VNL_FOR(e, MI(nM)) {
        const int i = m.M_row[e], j = anc_of(e);
        const vreal v = s[LO(LD) + e];
        s[LO(newt_M) + i * nv + j] = v, s[LO(newt_M) + j * nv + i] = v;
      }
      VNL_SYNC();
    }
    VNL_PROF(6);
  }

  // In-place L'DL in MuJoCo's mj_factorM order.  For a fixed k the rows that get updated are the
  // (distinct) ancestors of k and row k itself is only read, so all dk(dk+1)/2 multiply-adds of
  // iteration k run in one parallel region; pair p -> (a, a+c) comes from one universal triangular
  // table (ordered by a+c, so a prefix of it enumerates any depth).  The division of row k by its
  // pivot is deferred to one final pass (row k is never touched again after iteration k).
  VNL_HD void factor_lds() const {
    // (A column-per-lane variant that keeps the pivot row in registers and broadcasts it with
    // v_readlane was measured 2x slower: one LDS round trip in flight per step.  What matters is the
    // number of independent LDS accesses in flight, so each lane streams one ancestor row, 8-wide.)
    for (int k = MI(nv) - 1; k >= 0; k--) {
      int adr_k = madr(k), dk = eadr(k) - adr_k;
      vreal inv = vreal(1.) / s[LO(LD) + adr_k];
      VNL_SERIAL { s[LO(dinv) + k] = inv; }
      if (dk == 0) continue;
      VNL_FOR(a1, dk) {  // one lane per ancestor row: row(anc_a)[0..len) -= tmp * row_k[a .. a+len)
        int a = a1 + 1, len = dk - a + 1;
        vreal tmp = s[LO(LD) + adr_k + a] * inv;
        const vreal* src = s + LO(LD) + adr_k + a;
        vreal* dst = s + LO(LD) + madr(anc_of(adr_k + a));
        int c = 0;
        for (; c + 8 <= len; c += 8) {  // all 16 loads of a trip are issued before the first store
          vreal x0 = src[c], x1 = src[c + 1], x2 = src[c + 2], x3 = src[c + 3];
          vreal x4 = src[c + 4], x5 = src[c + 5], x6 = src[c + 6], x7 = src[c + 7];
          vreal y0 = dst[c], y1 = dst[c + 1], y2 = dst[c + 2], y3 = dst[c + 3];
          vreal y4 = dst[c + 4], y5 = dst[c + 5], y6 = dst[c + 6], y7 = dst[c + 7];
          dst[c] = y0 - tmp * x0, dst[c + 1] = y1 - tmp * x1, dst[c + 2] = y2 - tmp * x2, dst[c + 3] = y3 - tmp * x3;
          dst[c + 4] = y4 - tmp * x4, dst[c + 5] = y5 - tmp * x5, dst[c + 6] = y6 - tmp * x6, dst[c + 7] = y7 - tmp * x7;
        }
        for (; c + 4 <= len; c += 4) {
          vreal x0 = src[c], x1 = src[c + 1], x2 = src[c + 2], x3 = src[c + 3];
          vreal y0 = dst[c], y1 = dst[c + 1], y2 = dst[c + 2], y3 = dst[c + 3];
          dst[c] = y0 - tmp * x0, dst[c + 1] = y1 - tmp * x1, dst[c + 2] = y2 - tmp * x2, dst[c + 3] = y3 - tmp * x3;
        }
        for (; c < len; c++) dst[c] -= tmp * src[c];
      }
      VNL_SYNC();
    }
    VNL_SYNC();
    VNL_FOR(i, MI(nv)) {
      int adr = madr(i), dep = eadr(i) - adr;
      vreal di = s[LO(dinv) + i];
      for (int t = 1; t <= dep; t++) s[LO(LD) + adr + t] *= di;
    }
    VNL_SYNC();
  }

  // L -> L^-1 in place (same ancestor sparsity).  N(i,t) = -L(i,t) - sum_{u<t} L(i,u) N(anc_u, t-u):
  // the rows of one depth level only need already-converted ancestor rows and their own (still L)
  // row.  Per level: (A) stage the row base addresses of the ancestors, (B) all entries of the level
  // in parallel into a staging buffer, (C) write back.
  VNL_HD void invert_factor_lds() const {
    int stage = LO(Ma);              // Ma|grad free while factorising
    int* base = (int*)(s + LO(Mgrad));  // Mgrad|search likewise
    for (int lev = 1; lev <= MI(max_depth); lev++) {
      int q0 = lvl_start(lev), nrow = lvl_start(lev + 1) - q0, n = nrow * lev;
      VNL_FOR(q, n) {
        int i = lvl_dof(q0 + q / lev), t = q % lev + 1;
        base[q] = madr(anc_of(madr(i) + t));
      }
      VNL_SYNC();
      VNL_FOR(q, n) {
        int rr = q / lev, i = lvl_dof(q0 + rr), t = q - rr * lev + 1;
        int adr = madr(i);
        const int* bb = base + rr * lev - 1;  // bb[u] = madr(anc_u(i))
        vreal acc = -s[LO(LD) + adr + t];
        const vreal* row = s + LO(LD) + adr;
        const vreal* ld = s + LO(LD) + t;
        int u = 1;
        for (; u + 8 <= t; u += 8) {
          int b0 = bb[u], b1 = bb[u + 1], b2 = bb[u + 2], b3 = bb[u + 3];
          int b4 = bb[u + 4], b5 = bb[u + 5], b6 = bb[u + 6], b7 = bb[u + 7];
          vreal l0 = row[u], l1 = row[u + 1], l2 = row[u + 2], l3 = row[u + 3];
          vreal l4 = row[u + 4], l5 = row[u + 5], l6 = row[u + 6], l7 = row[u + 7];
          vreal n0 = ld[b0 - u], n1 = ld[b1 - u - 1], n2 = ld[b2 - u - 2], n3 = ld[b3 - u - 3];
          vreal n4 = ld[b4 - u - 4], n5 = ld[b5 - u - 5], n6 = ld[b6 - u - 6], n7 = ld[b7 - u - 7];
          acc -= (l0 * n0 + l1 * n1 + l2 * n2 + l3 * n3) + (l4 * n4 + l5 * n5 + l6 * n6 + l7 * n7);
        }
        for (; u + 4 <= t; u += 4) {  // loads first, so the four dependent (base -> N) chains overlap
          int b0 = bb[u], b1 = bb[u + 1], b2 = bb[u + 2], b3 = bb[u + 3];
          vreal l0 = row[u], l1 = row[u + 1], l2 = row[u + 2], l3 = row[u + 3];
          vreal n0 = ld[b0 - u], n1 = ld[b1 - u - 1], n2 = ld[b2 - u - 2], n3 = ld[b3 - u - 3];
          acc -= l0 * n0 + l1 * n1 + l2 * n2 + l3 * n3;
        }
        for (; u < t; u++) acc -= row[u] * ld[bb[u] - u];
        s[stage + q] = acc;
      }
      VNL_SYNC();
      VNL_FOR(q, n) {
        int rr = q / lev, i = lvl_dof(q0 + rr), t = q - rr * lev + 1;
        s[LO(LD) + madr(i) + t] = s[stage + q];
      }
      VNL_SYNC();
    }
  }

  // ---- factor / inversion with the matrix rows in registers --------------------------------------
  // Lane l keeps rows l, l+64, .. of the tree-sparse matrix in registers: entry c of a row is its
  // column at absolute depth c on the row's ancestor path (c < depth), the diagonal apart.  Static
  // register indices need static c, hence the MAXD template (max_depth < MAXD) and full unrolling.
  struct R4 {
    vreal x, y, z, w;
  };
  typedef vreal v2r __attribute__((vector_size(2 * sizeof(vreal))));  // (system 1, system 2) of a substep's two factorisations
  VNL_HD static R4 ld4a(const vreal* p) {  // p is 16-byte aligned in the float build
#if defined(__HIPCC__)
    if constexpr (sizeof(vreal) == 4) {
      typedef float f4 __attribute__((ext_vector_type(4)));
      f4 v = *(const f4*)p;
      return R4{v.x, v.y, v.z, v.w};
    }
#endif
    return R4{p[0], p[1], p[2], p[3]};
  }
  VNL_HD static void st4a(vreal* p, vreal x, vreal y, vreal z, vreal w) {
#if defined(__HIPCC__)
    if constexpr (sizeof(vreal) == 4) {
      typedef float f4 __attribute__((ext_vector_type(4)));
      f4 v = {(float)x, (float)y, (float)z, (float)w};
      *(f4*)p = v;
      return;
    }
#endif
    p[0] = x, p[1] = y, p[2] = z, p[3] = w;
  }

  // In-place L'DL (mj_factorM order, pivots nv-1 .. 0).  Step j: the lane that owns row j publishes
  // it (numerators + pivot) in an LDS scratch line; every proper ancestor row a of j then does
  // row_a[c] -= (row_j[a] / D_j) * row_j[c] on its registers, reading row_j as LDS broadcasts.  One
  // wave executes its LDS operations in order, so the scratch line needs no double buffering.
  // SOLVE: the matrix is only needed to solve ONE system (forward.euler's M + h diag(damping)): the right-hand
  // side rides along as an extra column (L' w = rhs is eliminated by the very same updates), the factor is
  // never stored, and x = L^-1 D^-1 w follows by a forward substitution over the depth: the rows of depth c
  // publish their x once per leaf below them, every deeper row reads "its" ancestor's x at a static offset.
  // Replaces factor + inversion + two sparse products for that system.
  // MAXD1: column bound of the SECOND lane set (rows 64 .. 127) when there are exactly two: those rows are often
  // shallow (rodent: depth <= 13), and registers for columns they do not have would be dead weight.
  template <int NSET, int MAXD, bool SOLVE = false, int MAXD1 = MAXD>
  VNL_HD void factor_rows(bool with_loop = true, int rhs = 0) const {
    static_assert(MAXD % 12 == 0 || MAXD == 16, "columns are processed in chunks of 12 (or 16)");
    constexpr int CH = MAXD % 12 == 0 ? 12 : 16;
    auto qd = [](int q) constexpr { return (NSET == 2 && q == 1) ? MAXD1 : MAXD; };
    vreal rr[NSET][MAXD], dg[NSET];
    int dep[NSET], last[NSET];
    const int sc = (LO(Ma) + 3) & ~3;  // Ma|grad|Mgrad|search are dead while factorising; VNL_FAC_LINES * (MAXD + 4) floats
#pragma unroll
    for (int q = 0; q < NSET; q++) {
      int a = (int)lane + q * VNL_LANES;
      bool ok = a < MI(nv);
      int adr = ok ? madr(a) : 0, d = ok ? eadr(a) - adr : 0;
      dep[q] = d, last[q] = ok ? a + ndesc(a) : -1;
      dg[q] = ok ? s[LO(LD) + adr] : vreal(1.);
#pragma unroll
      for (int c = 0; c < MAXD; c++)
        if (c < qd(q)) rr[q][c] = c < d ? s[LO(LD) + adr + d - c] : vreal(0.);
    }
    // Schedule (host, build_dev_model): row j is the pivot of step dof_ftime[j], after all of its
    // descendants; rows with disjoint subtrees share a step, each with its own scratch line
    // [row numerators (MAXD) | 1/pivot | j | pad], so the chain of dependent steps is the tree height.
    constexpr int LW = MAXD + 4;
    int ftime[NSET], fpack[NSET];  // fpack: scratch line | one leaf below << 8 | mask of all leaves below << 16
    vreal bb[NSET], myinv[NSET];   // SOLVE only
#pragma unroll
    for (int q = 0; q < NSET; q++) {
      int a = (int)lane + q * VNL_LANES;
      fpack[q] = a < MI(nv) ? m.dof_fslot[a] : 0;
      ftime[q] = a < MI(nv) ? m.dof_ftime[a] : -1;
      bb[q] = (SOLVE && a < MI(nv)) ? s[rhs + a] : vreal(0.), myinv[q] = vreal(0.);
    }
    VNL_SYNC();
    VNL_PROF(7);
    const int nsteps = with_loop ? MI(fac_steps) : 0;  // (false: diagnostic pricing of the load / store phases only)
    for (int step = 0; step < nsteps; step++) {
      int npub = 0;
#pragma unroll
      for (int q = 0; q < NSET; q++) {
        bool mine = ftime[q] == step;
        npub += VNL_COUNT(mine);
        if (mine) {
          int line = sc + (fpack[q] & 0xff) * LW, a = (int)lane + q * VNL_LANES;
#pragma unroll
          for (int c0 = 0; c0 < MAXD; c0 += CH) {
            if (c0 < qd(q) && c0 < dep[q]) {
#pragma unroll
              for (int c = c0; c < c0 + CH; c += 4)
                if (c < qd(q)) st4a(s + line + c, rr[q][c], rr[q][c + 1], rr[q][c + 2], rr[q][c + 3]);
            }
          }
          vreal inv = vnl_recip(dg[q]);
          s[line + MAXD] = inv;
          s[line + MAXD + 1] = vreal(a);  // exact: a < 2^24
          if constexpr (SOLVE) {
            s[line + MAXD + 2] = bb[q];
            myinv[q] = inv;
          } else {
            s[LO(dinv) + a] = inv;
          }
        }
      }
      VNL_WAVE_FENCE();
      // row_a[c] -= (row_j[a] / D_j) * row_j[c] for every published j that row a is an ancestor of.
      // Rows inside a chain have exactly one such j per step and take it in the first pass, all chains
      // at once (each lane reads ITS line); only rows above a branching point need further passes.
      unsigned match[NSET];
#pragma unroll
      for (int q = 0; q < NSET; q++) match[q] = 0u;
      int jk[VNL_FAC_LINES];  // pivot rows of this step: ONE wave-wide read of the line headers, then broadcasts
      VNL_LINE_HEADERS(jk, s + sc + MAXD + 1, LW);
#pragma unroll
      for (int k = 0; k < VNL_FAC_LINES; k++) {
        if (k < npub) {
          const int j = jk[k];
#pragma unroll
          for (int q = 0; q < NSET; q++) {
            int a = (int)lane + q * VNL_LANES;
            match[q] |= (a < j && j <= last[q]) ? (1u << k) : 0u;
          }
        }
      }
      for (int pass = 0; pass < VNL_FAC_LINES; pass++) {
        bool more = false;
#pragma unroll
        for (int q = 0; q < NSET; q++) more = more || match[q] != 0u;
        if (!vnl_wave_any(more)) break;
#pragma unroll
        for (int q = 0; q < NSET; q++) {
          if (match[q] != 0u) {
            const int k = __builtin_ctz(match[q]);
            match[q] &= match[q] - 1u;
            const vreal* line = s + sc + k * LW;
            // first column chunk fetched together with the pivot entry and 1/D: one LDS round trip, not two
            // (entries past the row's depth are don't-cares: never published, never stored)
            R4 x0[CH / 4];
#pragma unroll
            for (int c = 0; c < CH; c += 4) x0[c / 4] = ld4a(line + c);
            vreal traw = line[dep[q]];
            vreal t = traw * line[MAXD];
#pragma unroll
            for (int c = 0; c < CH; c += 4) {
              R4 x = x0[c / 4];
              rr[q][c] -= t * x.x, rr[q][c + 1] -= t * x.y, rr[q][c + 2] -= t * x.z, rr[q][c + 3] -= t * x.w;
            }
#pragma unroll
            for (int c0 = CH; c0 < MAXD; c0 += CH) {
              if (c0 < qd(q) && c0 < dep[q]) {
#pragma unroll
                for (int c = c0; c < c0 + CH; c += 4) {
                  if (c < qd(q)) {
                    R4 x = ld4a(line + c);
                    rr[q][c] -= t * x.x, rr[q][c + 1] -= t * x.y, rr[q][c + 2] -= t * x.z, rr[q][c + 3] -= t * x.w;
                  }
                }
              }
            }
            dg[q] -= t * traw;
            if constexpr (SOLVE) bb[q] -= t * line[MAXD + 2];
          }
        }
      }
      VNL_WAVE_FENCE();
    }
    if constexpr (SOLVE) {
      VNL_SYNC();
      const int xs = sc;  // [leaf][MAXD] published x values; the scratch lines are no longer needed
      vreal(&acc)[NSET] = bb;  // (same registers: w is consumed here)
#pragma unroll
      for (int q = 0; q < NSET; q++) acc[q] = bb[q] * myinv[q];  // D^-1 w
#pragma unroll
      for (int c = 0; c < MAXD; c++) {
        if (c <= MI(max_depth)) {
#pragma unroll
          for (int q = 0; q < NSET; q++) {
            int a = (int)lane + q * VNL_LANES;
            if (a < MI(nv) && dep[q] == c) {  // final: all ancestor terms are in
              s[rhs + a] = acc[q];
              for (int mk = fpack[q] >> 16; mk != 0; mk &= mk - 1) s[xs + __builtin_ctz(mk) * MAXD + c] = acc[q];
            }
          }
          VNL_WAVE_FENCE();
#pragma unroll
          for (int q = 0; q < NSET; q++)
            if (c < qd(q) && dep[q] > c) acc[q] -= rr[q][c] * myinv[q] * s[xs + ((fpack[q] >> 8) & 0xff) * MAXD + c];
        }
      }
      VNL_SYNC();
      return;
    }
    VNL_SYNC();
    VNL_PROF(8);
#pragma unroll
    for (int q = 0; q < NSET; q++) {
      int a = (int)lane + q * VNL_LANES;
      if (a < MI(nv)) {
        int adr = madr(a), d = dep[q];
        vreal di = s[LO(dinv) + a];
        s[LO(LD) + adr] = dg[q];
#pragma unroll
        for (int c = 0; c < MAXD; c++)
          if (c < qd(q) && c < d) s[LO(LD) + adr + d - c] = rr[q][c] * di;
      }
    }
    VNL_SYNC();
    VNL_PROF(9);
  }

  // L -> L^-1 in place from N L = I:  N(i,t) = -L(i,t) - sum_{0<u<t} N(i,u) L(anc_u(i), t-u).  Row i of
  // N only needs its own earlier entries (registers) and the ORIGINAL rows of its ancestors, so all
  // rows run at once without levels; the results are written back after one barrier.
  template <int NSET, int MAXD, int MAXD1 = MAXD>
  VNL_HD void invert_rows(int LDb) const {
    auto qd = [](int q) constexpr { return (NSET == 2 && q == 1) ? MAXD1 : MAXD; };
    // Sets are taken from the last to the first and written back one at a time: rows of a later
    // set are never ancestors of rows of an earlier one.
#pragma unroll
    for (int q = NSET - 1; q >= 0; q--) {
      vreal nn[MAXD];
      int a = (int)lane + q * VNL_LANES;
      bool ok = a < MI(nv);
      int adr = ok ? madr(a) : 0, d = ok ? eadr(a) - adr : 0;
      int own = LDb + adr;
      const vreal* pb[MAXD];  // row of the u-th ancestor (rows past the depth alias row 0: read, never used)
#pragma unroll
      for (int u = 1; u < MAXD; u++)
        if (u < qd(q)) pb[u] = s + LDb + (u < d ? madr(anc_of(adr + u)) : 0);
#pragma unroll
      for (int t = 1; t < MAXD; t++) {
        if (t < qd(q) && vnl_wave_any(t <= d)) {
          vreal x[MAXD];  // all operands of entry t are fetched before the (dependent) multiply-add chain
#pragma unroll
          for (int u = 1; u < t; u++) x[u] = pb[u][t - u];
          vreal acc = t <= d ? -s[own + t] : vreal(0.);
#pragma unroll
          for (int u = 1; u < t; u++) acc -= nn[u] * x[u];
          nn[t] = acc;
        }
      }
      VNL_SYNC();
#pragma unroll
      for (int t = 1; t < MAXD; t++)
        if (t < qd(q) && t <= d) s[own + t] = nn[t];
      VNL_SYNC();
    }
  }

  // The two factorisations WITHOUT the matrix: the articulated-body form of L'DL.  With every spatial quantity expressed
  // about one common origin (as they are here) the elimination of dof k from the joint-space matrix is a rank-one downdate
  // of the articulated inertia of k's subtree, IA <- IA - U_k U_k' / D_k with U_k = IA cdof_k, D_k = cdof_k . U_k + armature_k
  // (+ h damping_k for the second system), and row k of the unit-lower factor is a set of projections of ONE 6-vector:
  //     L(k, j) = cdof_j . U_k / D_k          for every ancestor j of k.
  // IA is never formed: applied to cdof_j it is  U_j = crb(body_j) cdof_j - sum_{k below j} L(k, j) U_k,  so each lane keeps
  // the 6-vector U_j of its dof (both systems side by side: 6 register pairs instead of 36 matrix-row pairs), a pivot
  // publishes (U_k, 1/D_k) -- 16 floats -- and each ancestor lane spends 6 + 6 packed multiply-adds on it instead of a
  // multiply-add per matrix column.  Same schedule (m.dof_ftime / dof_fslot / fac_match) and same output as the row
  // elimination this replaces (tests/test_hostsim_parity.py: the float64 host build against the dense oracle).  Out: L1 in
  // LO(LD) and 1/D1 in LO(dinv) (M * qacc_warmstart), V_k = U_k / D_k of both systems in the pool (vstore(): what invert_aba
  // builds both inverse factors from), 1/D2 in the env's global scratch.  Needs crb in the pool (mass_matrix's
  // tree_accumulate); the matrix entries themselves are not needed any more.
  template <int NSET>
  VNL_HD void factor_aba(vreal h) const {
    constexpr int LW = 16;  // [(U1, U2) x 6 | 1/D1, 1/D2 | end address of the pivot's row | pad]
    const int sc = (LO(Mgrad) + 3) & ~3;  // Mgrad .. tmp2 are dead while factorising
    const int VS = vstore();
    vreal* g2 = fac2();
    const int nsteps = MI(fac_steps);
    v2r U[NSET][6], diag[NSET];
    vreal S[NSET][6];
    int dep[NSET], ftime[NSET], myline[NSET], adrs[NSET];
    const unsigned char* mt[NSET];
    unsigned nxt[NSET];
#pragma unroll
    for (int q = 0; q < NSET; q++) {
      const int a = (int)lane + q * VNL_LANES;
      const bool ok = a < MI(nv);
      const int aa = ok ? a : 0;
      const S6 c = ld6(LO(cdof) + 6 * aa), f = inert_mul(LO(P) + 10 * m.dof_body[aa], c);
      S[q][0] = c.a.x, S[q][1] = c.a.y, S[q][2] = c.a.z, S[q][3] = c.l.x, S[q][4] = c.l.y, S[q][5] = c.l.z;
      U[q][0] = v2r{f.a.x, f.a.x}, U[q][1] = v2r{f.a.y, f.a.y}, U[q][2] = v2r{f.a.z, f.a.z};
      U[q][3] = v2r{f.l.x, f.l.x}, U[q][4] = v2r{f.l.y, f.l.y}, U[q][5] = v2r{f.l.z, f.l.z};
      const vreal arm = m.dof_armature[aa];
      diag[q] = v2r{arm, arm + h * m.dof_damping[aa]};
      adrs[q] = madr(aa), dep[q] = eadr(aa) - adrs[q];
      myline[q] = m.dof_fslot[aa] & 0xff;
      ftime[q] = ok ? m.dof_ftime[aa] : -1;
      mt[q] = m.fac_match + (size_t)aa * nsteps;
      nxt[q] = (ok && nsteps > 0) ? mt[q][0] : 0u;
    }
    VNL_SYNC();  // (crb is read: the image may now grow over it)
    VNL_PROF(7);
    for (int step = 0; step < nsteps; step++) {
      unsigned cur[NSET];
#pragma unroll
      for (int q = 0; q < NSET; q++) {
        cur[q] = nxt[q];
        nxt[q] = (ftime[q] >= 0 && step + 1 < nsteps) ? mt[q][step + 1] : 0u;  // prefetch
        if (ftime[q] == step) {  // this row is a pivot now: every dof below it has been absorbed
          const int a = (int)lane + q * VNL_LANES;
          v2r D = diag[q];
#pragma unroll
          for (int i = 0; i < 6; i++) D += v2r{S[q][i], S[q][i]} * U[q][i];
          const v2r iv = v2r{vnl_recip(D[0]), vnl_recip(D[1])};
          vreal* ln = s + sc + myline[q] * LW;
          st4a(ln, U[q][0][0], U[q][0][1], U[q][1][0], U[q][1][1]);
          st4a(ln + 4, U[q][2][0], U[q][2][1], U[q][3][0], U[q][3][1]);
          st4a(ln + 8, U[q][4][0], U[q][4][1], U[q][5][0], U[q][5][1]);
          st4a(ln + 12, iv[0], iv[1], (vreal)(adrs[q] + dep[q]), vreal(0.));
          // what stays: V = U / D of both systems (invert_aba), 1/D1 where every M^-1 product expects it, 1/D2 beside N2
          vreal* vs = s + VS + 12 * a;
          const v2r v0 = U[q][0] * iv, v1 = U[q][1] * iv, v2 = U[q][2] * iv, v3_ = U[q][3] * iv, v4 = U[q][4] * iv, v5 = U[q][5] * iv;
          st4a(vs, v0[0], v0[1], v1[0], v1[1]);
          st4a(vs + 4, v2[0], v2[1], v3_[0], v3_[1]);
          st4a(vs + 8, v4[0], v4[1], v5[0], v5[1]);
          s[LO(dinv) + a] = iv[0];
          g2[MI(nM) + a] = iv[1];
        }
      }
      VNL_WAVE_FENCE();
      for (int pass = 0; pass < VNL_FAC_LINES; pass++) {
        bool more = false;
#pragma unroll
        for (int q = 0; q < NSET; q++) more = more || cur[q] != 0u;
        if (!vnl_wave_any(more)) break;
#pragma unroll
        for (int q = 0; q < NSET; q++) {
          if (cur[q] != 0u) {
            const int k = __builtin_ctz(cur[q]);
            cur[q] &= cur[q] - 1u;
            const vreal* ln = s + sc + k * LW;
            const R4 u0 = ld4a(ln), u1 = ld4a(ln + 4), u2 = ld4a(ln + 8), hd = ld4a(ln + 12);
            const v2r Uk[6] = {v2r{u0.x, u0.y}, v2r{u0.z, u0.w}, v2r{u1.x, u1.y}, v2r{u1.z, u1.w}, v2r{u2.x, u2.y}, v2r{u2.z, u2.w}};
            v2r pa = v2r{S[q][0], S[q][0]} * Uk[0], pb = v2r{S[q][1], S[q][1]} * Uk[1];
            pa += v2r{S[q][2], S[q][2]} * Uk[2], pb += v2r{S[q][3], S[q][3]} * Uk[3];
            pa += v2r{S[q][4], S[q][4]} * Uk[4], pb += v2r{S[q][5], S[q][5]} * Uk[5];
            const v2r Lk = (pa + pb) * v2r{hd.x, hd.y};
#pragma unroll
            for (int i = 0; i < 6; i++) U[q][i] -= Lk * Uk[i];
            // L1(k, this dof) for M * qacc_warmstart: row k holds its ancestors from the parent (adr + 1) up to the root (end)
            s[LO(LD) + (int)hd.z - dep[q]] = Lk[0];
          }
        }
      }
      VNL_WAVE_FENCE();
    }
    VNL_SYNC();
    VNL_PROF(8);
    VNL_PROF(9);
  }

  VNL_HD int vstore() const { return (LO(P) + 3) & ~3; }  // [nv][(V1, V2) x 6]: over crb, below cvel (factor_pair_ok)
  // where bias_forces leaves cvel for make_constraint: behind cacc / cfrc (pool + 16 nbody) and, where the pool has the room,
  // behind the store of the V vectors too (with the welded bodies folded away the body arrays are shorter than that store)
  VNL_HD int cvel_at() const {
    const int a = LO(P) + 16 * MI(nbody), b = vstore() + 12 * MI(nv);
    return (b > a && b + 6 * MI(nbody) <= LO(smooth)) ? b : a;
  }

  // Both inverse factors N = L^-1 from the same 6-vectors: walking up from row k, with C = V_k at the start,
  //     N(k, j) = -cdof_j . C,     C += N(k, j) V_j          for the parent j, the grandparent, ..
  // (C is sum_i N(k, i) V_i over the path walked so far, so cdof_j . C = sum_i N(k, i) L(i, j)): 6 + 6 packed multiply-adds
  // per entry where the row recurrence on matrix entries needs one per entry ALREADY done -- the rodent's deep rows (35
  // ancestors) cost 420 instead of 612, and nothing depends on a row of registers per lane.  N1 goes to LO(LD) (over L1,
  // which M * qacc_warmstart has used by then), N2 to the env's global scratch for euler().  A lane takes row `lane`, then
  // (GUESTS) the row 64.. the host table m.fac_guest places with it -- both inside ONE loop, so that the nine short extra
  // rows of the rodent ride in the shadow of the deep ones.
  template <int NSET, bool GUESTS>
  VNL_HD void invert_aba(vreal* g2) const {
    const int VS = vstore();
    const unsigned char* an = (const unsigned char*)(s + LO(tab_anc));
    auto load_v = [&](int r, v2r* C) {
      const vreal* vs = s + VS + 12 * r;
      const R4 a = ld4a(vs), b = ld4a(vs + 4), c = ld4a(vs + 8);
      C[0] = v2r{a.x, a.y}, C[1] = v2r{a.z, a.w}, C[2] = v2r{b.x, b.y}, C[3] = v2r{b.z, b.w}, C[4] = v2r{c.x, c.y}, C[5] = v2r{c.z, c.w};
    };
#pragma unroll
    for (int q = 0; q < NSET; q++) {
      const int a = (int)lane + q * VNL_LANES;
      int row = a < MI(nv) ? a : -1, next = -1;
      if (GUESTS && q == 0 && m.fac_guest) next = m.fac_guest[lane];
      int adr = madr(row >= 0 ? row : 0), d = row >= 0 ? eadr(row) - adr : 0, tt = 0;
      v2r C[6];
      load_v(row >= 0 ? row : 0, C);
      int jn = an[adr + (d > 0 ? 1 : 0)];
      while (vnl_wave_any(row >= 0)) {
        if (row >= 0 && tt == d) {  // this row is done: on to the guest row, if any
          row = next, next = -1;
          if (row >= 0) {
            adr = madr(row), d = eadr(row) - adr, tt = 0;
            load_v(row, C);
            jn = an[adr + (d > 0 ? 1 : 0)];
          }
        }
        if (row >= 0 && tt < d) {
          const int j = jn;
          jn = an[adr + (tt + 2 <= d ? tt + 2 : d)];  // the next ancestor's index while this one's vectors arrive
          const S6 c = ld6(LO(cdof) + 6 * j);
          v2r V[6];
          load_v(j, V);
          v2r pa = v2r{c.a.x, c.a.x} * C[0], pb = v2r{c.a.y, c.a.y} * C[1];
          pa += v2r{c.a.z, c.a.z} * C[2], pb += v2r{c.l.x, c.l.x} * C[3];
          pa += v2r{c.l.y, c.l.y} * C[4], pb += v2r{c.l.z, c.l.z} * C[5];
          const v2r n = -(pa + pb);
#pragma unroll
          for (int i = 0; i < 6; i++) C[i] += n * V[i];
          tt++;
          s[LO(LD) + adr + tt] = n[0], g2[adr + tt] = n[1];
        }
      }
    }
    VNL_SYNC();
  }

  // models whose two factorisations go through factor_aba: the store of the V vectors (12 nv elements from the pool's start) ends
  // below cvel, which make_constraint still needs, and the six scratch lines fit into the dead vectors Mgrad .. tmp2
  VNL_HD bool factor_pair_ok() const {
    const int nv = MI(nv);
    if (!MI(eulerdamp) || !m.fac_match) return false;
    return nv <= VNL_ROWSETS_2 * VNL_LANES && vstore() + 12 * nv <= cvel_at() && VNL_FAC_LINES * 16 + 3 <= 6 * nv;
  }
  VNL_HD void factor_both(vreal h) const {
    if (MI(nv) <= VNL_ROWSETS_1 * VNL_LANES) factor_aba<VNL_ROWSETS_1>(h);
    else factor_aba<VNL_ROWSETS_2>(h);
  }

  VNL_HD void factor(bool with_loop = true) const {
    const int nv = MI(nv), md = MI(max_depth);
    // the scratch lines of factor_rows live in the four dead CG vectors (Ma .. search)
    const int room = 4 * nv - 3 - 4 * VNL_FAC_LINES;
    if (nv <= VNL_ROWSETS_1 * VNL_LANES && md < 16 && VNL_FAC_LINES * 16 <= room) factor_rows<VNL_ROWSETS_1, 16>(with_loop);
    else if (nv <= VNL_ROWSETS_1 * VNL_LANES && md < 36 && VNL_FAC_LINES * 36 <= room) factor_rows<VNL_ROWSETS_1, 36>(with_loop);
    else if (nv <= VNL_ROWSETS_2 * VNL_LANES && md < 36 && VNL_FAC_LINES * 36 <= room && (MI(fac_nleaf) >> 8) < 16)
      factor_rows<VNL_ROWSETS_2, 36, false, 16>(with_loop);  // (a second lane set with deeper rows takes the LDS route:
                                                             // every instantiation costs registers for the whole kernel)
    else factor_lds();
  }
  // solve (matrix in LD) x = s[rhs .. rhs+nv) in place without storing a factor; false if this model needs the
  // general route (factor + invert_factor + solve_inplace)
  VNL_HD bool factor_solve(int rhs) const {
    const int nv = MI(nv), md = MI(max_depth);
    const int room = 4 * nv - 3 - 4 * VNL_FAC_LINES;
    if ((MI(fac_nleaf) & 0xff) == 0) return false;
    if (nv <= VNL_ROWSETS_1 * VNL_LANES && md < 16 && VNL_FAC_LINES * 16 <= room) factor_rows<VNL_ROWSETS_1, 16, true>(true, rhs);
    else if (nv <= VNL_ROWSETS_1 * VNL_LANES && md < 36 && VNL_FAC_LINES * 36 <= room) factor_rows<VNL_ROWSETS_1, 36, true>(true, rhs);
    else if (nv <= VNL_ROWSETS_2 * VNL_LANES && md < 36 && VNL_FAC_LINES * 36 <= room && (MI(fac_nleaf) >> 8) < 16)
      factor_rows<VNL_ROWSETS_2, 36, true, 16>(true, rhs);
    else return false;
    return true;
  }
  // (LDb: where the factor sits -- LO(LD), or the copy of the second factor that euler() brings into the pool)
  VNL_HD void invert_factor(int LDb) const {
    const int nv = MI(nv), md = MI(max_depth);
    if (nv <= VNL_ROWSETS_1 * VNL_LANES && md < 16) invert_rows<VNL_ROWSETS_1, 16>(LDb);
    else if (nv <= VNL_ROWSETS_1 * VNL_LANES && md < 36) invert_rows<VNL_ROWSETS_1, 36>(LDb);
    else if (nv <= VNL_ROWSETS_2 * VNL_LANES && md < 36 && (MI(fac_nleaf) >> 8) < 16) invert_rows<VNL_ROWSETS_2, 36, 16>(LDb);
    else invert_factor_lds();  // (only ever reached with LDb == LO(LD): factor_pair_ok() excludes these models)
  }
  VNL_HD void invert_factor() const { invert_factor(LO(LD)); }

  // sum_{t=1..dep} LD[adr+t] * in[anc_of(adr+t)], four independent index->value chains per trip
  // (ST: element stride of the factor)
  template <int ST = 1>
  VNL_HD vreal row_dot(int adr, int dep, int in, int LDb) const {
    const unsigned char* an = (const unsigned char*)(s + LO(tab_anc)) + adr;
    const vreal* row = s + LDb + ST * adr;
    vreal acc = vreal(0.);
    int t = 1;
    constexpr int W = VNL_CHAIN_WIDTH;  // independent index->value chains per trip: two LDS round trips per trip
    for (; t + W - 1 <= dep; t += W) {
      int j[W];
      vreal l[W], x[W];
#pragma unroll
      for (int u = 0; u < W; u++) j[u] = an[t + u];
#pragma unroll
      for (int u = 0; u < W; u++) l[u] = row[ST * (t + u)];
#pragma unroll
      for (int u = 0; u < W; u++) x[u] = s[in + j[u]];
      vreal p0 = vreal(0.), p1 = vreal(0.);
#pragma unroll
      for (int u = 0; u < W; u += 2) p0 += l[u] * x[u], p1 += l[u + 1] * x[u + 1];
      acc += p0 + p1;
    }
    if (t <= dep) {  // the remainder in ONE predicated trip (4-wide and single-entry tails cost a trip each)
      int j[W];
      vreal l[W], x[W];
#pragma unroll
      for (int u = 0; u < W; u++) {
        const int tt = t + u <= dep ? t + u : dep;
        j[u] = an[tt], l[u] = row[ST * tt];
      }
#pragma unroll
      for (int u = 0; u < W; u++) x[u] = s[in + j[u]];
      vreal p0 = vreal(0.), p1 = vreal(0.);
#pragma unroll
      for (int u = 0; u < W; u += 2) {
        p0 += t + u <= dep ? l[u] * x[u] : vreal(0.);
        p1 += t + u + 1 <= dep ? l[u + 1] * x[u + 1] : vreal(0.);
      }
      acc += p0 + p1;
    }
    return acc;
  }

  // out[i] = in[i] + sum_t A(i, anc_t) in[anc_t]   (A = strictly-lower part held in LD: L or L^-1)
  VNL_HD void row_apply(int in, int out, bool scale_by_dinv, int LDb, int dinvb) const {
    VNL_FOR(i, MI(nv)) {
      int adr = madr(i), dep = eadr(i) - adr;
      vreal acc = s[in + i] + row_dot(adr, dep, in, LDb);
      s[out + i] = scale_by_dinv ? acc * s[dinvb + i] : acc;
    }
    VNL_SYNC();
  }
  // out[a] = (in[a] + sum_{i in desc(a)} A(i, a) in[i]) (* or / D); descendants are the next ndesc dofs
  template <int ST = 1>
  VNL_HD void col_apply(int in, int out, int dmode /*0 none, 1 multiply by dinv, 2 divide by dinv*/, int LDb, int dinvb) const {
    VNL_FOR(a, MI(nv)) {
      int da = eadr(a) - madr(a), nd = ndesc(a);
      vreal acc = s[in + a];
      const unsigned short* ea = (const unsigned short*)(s + LO(tab_madr)) + MI(nv);
      const vreal* ld = s + LDb - ST * da;
      int i = a + 1, iend = a + nd;
      constexpr int W = VNL_CHAIN_WIDTH;
      for (; i + W - 1 <= iend; i += W) {
        int e[W];
        vreal l[W], x[W];
#pragma unroll
        for (int u = 0; u < W; u++) e[u] = ea[i + u];
#pragma unroll
        for (int u = 0; u < W; u++) x[u] = s[in + i + u];
#pragma unroll
        for (int u = 0; u < W; u++) l[u] = ld[ST * e[u]];
        vreal p0 = vreal(0.), p1 = vreal(0.);
#pragma unroll
        for (int u = 0; u < W; u += 2) p0 += l[u] * x[u], p1 += l[u + 1] * x[u + 1];
        acc += p0 + p1;
      }
      if (i <= iend) {  // the remainder in one predicated trip
        int e[W];
        vreal l[W], x[W];
#pragma unroll
        for (int u = 0; u < W; u++) {
          const int ii = i + u <= iend ? i + u : iend;
          e[u] = ea[ii], x[u] = s[in + ii];
        }
#pragma unroll
        for (int u = 0; u < W; u++) l[u] = ld[ST * e[u]];
        vreal p0 = vreal(0.), p1 = vreal(0.);
#pragma unroll
        for (int u = 0; u < W; u += 2) {
          p0 += i + u <= iend ? l[u] * x[u] : vreal(0.);
          p1 += i + u + 1 <= iend ? l[u + 1] * x[u + 1] : vreal(0.);
        }
        acc += p0 + p1;
      }
      s[out + a] = dmode == 1 ? acc * s[dinvb + ST * a] : (dmode == 2 ? acc / s[dinvb + ST * a] : acc);
    }
    VNL_SYNC();
  }

  // The same two products, BALANCED: a lane per row (column) makes the wave wait for the deepest row (35 entries for the
  // rodent, the mean is 14) and the largest subtree (72, same mean).  The host cuts every row / column into blocks of <= 13
  // entries (VNL_BLK_W) and deals the blocks out over the lanes (m.blk_tab, csrc/vnl_lib.hip); the blocks of one row sit in adjacent
  // lanes of a 16-lane DPP row, their partial sums meet in the first of them by shifts (VNL_SEG_SUM) and that lane writes
  // the element: 2 trips of 13-entry blocks per product for the rodent instead of 5 (rows) and 10 (columns) of 8.
  //   COL == false: out[i] = in[i] + sum_t A(i, anc_t) in[anc_t]          COL == true: out[a] = in[a] + sum_{i in desc(a)} A(i, a) in[i]
  //   dmode: 0 none, 1 multiply by dinv, 2 divide by dinv
  template <int ST, bool COL>
  VNL_HD void blk_apply(int in, int out, int dmode, int LDb, int dinvb) const {
    const int cfg = MI(blk_cfg);
    const int trips = COL ? (cfg >> 4) & 15 : cfg & 15, steps = COL ? (cfg >> 12) & 15 : (cfg >> 8) & 15;
    const unsigned* tab = m.blk_tab + (COL ? (cfg & 15) * 64 : 0);
    constexpr int W = VNL_BLK_W, MAXT = 4;  // (the host builds no table that needs more trips)
#ifdef __HIP_DEVICE_COMPILE__
    unsigned pre[MAXT];  // every trip's descriptor requested before the first is used: one exposed global-memory round trip per product
#pragma unroll
    for (int t = 0; t < MAXT; t++) pre[t] = t < trips ? tab[t * 64 + (int)lane] : 0u;
#endif
#pragma unroll
    for (int trip = 0; trip < MAXT; trip++) {
      if (trip >= trips) break;
      VNL_PERLANE(vreal, part);
      VNL_PERLANE(unsigned, dsc);
      VNL_FOR(l, VNL_WAVE_ITEMS(64)) {
#ifdef __HIP_DEVICE_COMPILE__
        const unsigned d = pre[trip];
#else
        const unsigned d = tab[trip * 64 + l];
#endif
        const int n = (int)(d >> 16) & 15;
        // (entries past the block's own n are read -- they are inside the LDS image: the next row's entries, the next
        // table -- and dropped by the selects below, which costs less than clamping every index)
        vreal l_[W], x[W];
        if constexpr (!COL) {
          const int e0 = (int)(d & 0xffffu);
          const unsigned char* an = (const unsigned char*)(s + LO(tab_anc)) + e0;
          const vreal* row = s + LDb + ST * e0;
          int j[W];
#pragma unroll
          for (int u = 0; u < W; u++) j[u] = an[u], l_[u] = row[ST * u];
#pragma unroll
          for (int u = 0; u < W; u++) x[u] = s[in + j[u]];
        } else {
          const int i0 = (int)(d & 0x7fu), da = (int)(d >> 7) & 0x3f;
          const unsigned short* ea = (const unsigned short*)(s + LO(tab_madr)) + MI(nv) + i0;
          const vreal* ld = s + LDb - ST * da;
          int e[W];
#pragma unroll
          for (int u = 0; u < W; u++) e[u] = ea[u], x[u] = s[in + i0 + u];
#pragma unroll
          for (int u = 0; u < W; u++) l_[u] = ld[ST * e[u]];
        }
        vreal p0 = vreal(0.), p1 = vreal(0.);
#pragma unroll
        for (int u = 0; u < W; u++) {
          const vreal pr = u < n ? l_[u] * x[u] : vreal(0.);
          if (u & 1) p1 += pr;
          else p0 += pr;
        }
        VNL_AT(part, l) = p0 + p1;
        VNL_AT(dsc, l) = d;
      }
      VNL_SEG_SUM(part, dsc, steps);
      VNL_FOR(l, VNL_WAVE_ITEMS(64)) {
        const unsigned d = VNL_AT(dsc, l);
        if (d & (1u << 27)) {
          const int r = (int)(d >> 20) & 0x7f;
          const vreal acc = s[in + r] + VNL_AT(part, l);
          s[out + r] = dmode == 1 ? acc * s[dinvb + ST * r] : (dmode == 2 ? acc / s[dinvb + ST * r] : acc);
        }
      }
    }
    VNL_SYNC();
  }

  VNL_HD bool blk_on() const {
#ifdef VNL_NO_BLK  // regression build (csrc/build.py --noblk): the lane-per-row / lane-per-column products
    return false;
#else
    return MI(blk_cfg) != 0;
#endif
  }
  // x <- M^-1 x = L^-1 D^-1 L^-T x with the inverted factor: two dependency-free sparse products
  VNL_HD void solve_inplace(int x, int LDb, int dinvb) const {
    if (blk_on()) {
      blk_apply<1, true>(x, LO(tmp2), 1, LDb, dinvb);
      blk_apply<1, false>(LO(tmp2), x, 0, LDb, dinvb);
      return;
    }
    col_apply(x, LO(tmp2), 1, LDb, dinvb);
    row_apply(LO(tmp2), x, false, LDb, dinvb);
  }
  VNL_HD void solve_inplace(int x) const { solve_inplace(x, LO(LD), LO(dinv)); }

  // out = M v = L' D L v with the (not yet inverted) factor
  VNL_HD void mass_mul_factor(int vec, int out) const {
    if (blk_on()) {
      blk_apply<1, false>(vec, LO(tmp2), 2, LO(LD), LO(dinv));
      blk_apply<1, true>(LO(tmp2), out, 0, LO(LD), LO(dinv));
      return;
    }
    VNL_FOR(i, MI(nv)) {
      int adr = madr(i), dep = eadr(i) - adr;
      vreal acc = s[vec + i] + row_dot(adr, dep, vec, LO(LD));
      s[LO(tmp2) + i] = acc / s[LO(dinv) + i];
    }
    VNL_SYNC();
    col_apply(LO(tmp2), out, 0, LO(LD), LO(dinv));
  }

  VNL_HD void invert_both(vreal* g2) const {
    if (MI(nv) <= VNL_ROWSETS_1 * VNL_LANES) invert_aba<VNL_ROWSETS_1, false>(g2);
#if VNL_LANES == 64
    else if (m.fac_guest) invert_aba<VNL_ROWSETS_1, true>(g2);  // (the rows 64 .. ride with lanes whose own row is short)
#endif
    else invert_aba<VNL_ROWSETS_2, false>(g2);
  }

  // ------------------------------------------------------------------ velocity
  // com_vel + rne: -qfrc_bias - damping*qvel -> LO(smooth).  Body velocities / accelerations are tree prefixes
  // (pointer jumping) of per-body contributions; returns the LDS offset of cvel (kept for
  // make_constraint).  Needs cinert in T1.
  VNL_HD int bias_forces() const {
    int nb6 = 6 * MI(nbody);
    int X0 = LO(P) + 10 * MI(nbody);  // after cinert: cacc, then cfrc in place
    // cvel[b] = sum of cdof_d qvel_d over the dofs on b's path: differences of the dof prefix sums (parked in the
    // factor buffer, which is free until the mass matrix is built)
    dof_prefix(LO(qvel), LO(LD));
    const int cv = cvel_at();
    VNL_FOR(b, MI(nbody)) st6(cv + 6 * b, path_sum(LO(LD), m.body_pathseg + 8 * b));
    VNL_SYNC();
    VNL_PROF(2);
    int ca = X0;
    // own acceleration term of every body (sum over its dofs of cdof_dot * qvel), prefix-summed over the body index
    // in the same pass; cacc[b] = sum over b's ancestors = differences over its runs of consecutive ancestor bodies
    const int QB = LO(LD) + 6 * (MI(nv) + 1);
    S6 run = S6{v3(0, 0, 0), v3(0, 0, 0)}, carry = run;
    VNL_FOR(b, VNL_PAD_ITEMS(MI(nbody))) {
      S6 acc = S6{v3(0, 0, 0), v3(0, 0, 0)};
      if (b > 0 && b < MI(nbody)) {
        S6 vel = ld6(cv + 6 * parent_of(b));
        int jn = m.body_jntnum[b], ja = m.body_jntadr[b];
        for (int k = 0; k < jn; k++) {
          int j = ja + k, da = m.jnt_dofadr[j];
          if (m.jnt_type[j] == VNL_JNT_FREE) {
            for (int t = 0; t < 3; t++) vel = vel + ld6(LO(cdof) + 6 * (da + t)) * s[LO(qvel) + da + t];
            S6 vel0 = vel;
            for (int t = 3; t < 6; t++) {
              S6 c = ld6(LO(cdof) + 6 * (da + t));
              vreal qd = s[LO(qvel) + da + t];
              acc = acc + mcross(vel0, c) * qd;
              vel = vel + c * qd;
            }
          } else {
            S6 c = ld6(LO(cdof) + 6 * da);
            vreal qd = s[LO(qvel) + da];
            acc = acc + mcross(vel, c) * qd;
            vel = vel + c * qd;
          }
        }
      }
      VNL_SCAN_ADD_C(acc.a.x, run.a.x, carry.a.x), VNL_SCAN_ADD_C(acc.a.y, run.a.y, carry.a.y);
      VNL_SCAN_ADD_C(acc.a.z, run.a.z, carry.a.z), VNL_SCAN_ADD_C(acc.l.x, run.l.x, carry.l.x);
      VNL_SCAN_ADD_C(acc.l.y, run.l.y, carry.l.y), VNL_SCAN_ADD_C(acc.l.z, run.l.z, carry.l.z);
      if (b < MI(nbody)) st6(QB + 6 * (b + 1), acc);
      if (b == 0) st6(QB, S6{v3(0, 0, 0), v3(0, 0, 0)});
    }
    VNL_SYNC();
    VNL_FOR(b, MI(nbody)) st6(ca + 6 * b, path_sum(QB, m.body_pathseg + 8 * b + 4));
    VNL_SYNC();
    VNL_PROF(3);
    VNL_FOR(b, MI(nbody)) {  // cfrc overwrites cacc in place (each body only needs its own entries)
      if (b == 0) {
        st6(ca, S6{v3(0, 0, 0), v3(0, 0, 0)});
        continue;
      }
      S6 vel = ld6(cv + 6 * b), acc = ld6(ca + 6 * b);
      acc.l = acc.l + v3(-m.gx, -m.gy, -m.gz);  // cacc of the world body, inherited by every body
      st6(ca + 6 * b, inert_mul(LO(P) + 10 * b, acc) + mcross_force(vel, inert_mul(LO(P) + 10 * b, vel)));
    }
    VNL_SYNC();
    tree_accumulate(ca, 6);
    // qfrc_smooth starts as passive damping minus the bias force (springs / actuation added by smooth_forces)
    VNL_FOR(d, MI(nv))
      s[LO(smooth) + d] = -m.dof_damping[d] * s[LO(qvel) + d] - dot(ld6(LO(cdof) + 6 * d), ld6(ca + 6 * m.dof_body[d]));
    VNL_SYNC();
    VNL_PROF(4);
    return cv;
  }

  // passive + actuation + qfrc_smooth + qacc_smooth
  VNL_HD void smooth_forces() const {
    VNL_FOR(j, MI(njnt)) {
      if (m.jnt_type[j] == VNL_JNT_HINGE) {
        vreal k = m.jnt_stiffness[j];
        if (k != vreal(0.)) s[LO(smooth) + m.jnt_dofadr[j]] -= k * (s[LO(qpos) + m.jnt_qposadr[j]] - m.jnt_springref[j]);
      }
    }
    // Actuator forces in parallel (the model tables are L2 reads: the former one-lane loop paid 30 latencies in a
    // row).  Several actuators may drive one dof, so each dof's lane then walks the actuator list in order and
    // keeps the forces aimed at it: broadcast LDS reads that pipeline, where one lane adding into the dofs paid a
    // dependent read-modify-write per actuator.
    const int frc = LO(tmp2), adof = LO(tmp);  // both free here
    const bool listed = m.dof_act != nullptr;  // (host table: the <= 4 actuators of every dof, in actuator order)
    VNL_FOR(i, MI(nu)) {
      vreal ctrl = s[LO(ctrl) + i], a = ctrl;
      vreal tau = m.act_tau[i];
      if (tau >= vreal(0.)) {
        a = s[LO(act) + i];
        s[LO(actdot) + i] = (ctrl - a) / fmax(tau, VNL_MINVAL);
      }
      s[frc + i] = m.act_gear[i] * m.act_gain[i] * a;
      if (!listed) s[adof + i] = vreal(m.act_dof[i]);  // exact: dof < 2^24
    }
    VNL_SYNC();
    vreal* gf = gqfrc_act();
    VNL_FOR(d, MI(nv)) {
      vreal fa = vreal(0.);
      if (listed) {
        // the actuators aimed at this dof, packed by the host (index + 1 per byte, ascending: the same summation order as
        // the walk over all actuators below, which models with more than four actuators on one dof still take)
        for (unsigned pk = (unsigned)m.dof_act[d]; pk != 0u; pk >>= 8) fa += s[frc + (int)(pk & 0xffu) - 1];
      } else {
        const vreal me = vreal(d);
#pragma unroll 6
        for (int i = 0; i < MI(nu); i++) {
          const vreal f = s[frc + i];
          if (s[adof + i] == me) fa += f;
        }
      }
      gf[d] = fa;
      vreal v = s[LO(smooth) + d] + fa;
      s[LO(smooth) + d] = v;
      s[LO(qacc_smooth) + d] = v;
    }
    VNL_SYNC();
    solve_inplace(LO(qacc_smooth));
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

  // collision (plane vs sphere / capsule / ellipsoid) + constraint rows, one lane per limit row /
  // per geom.  Rows that MJX would mask out (pos >= 0) get D = 0: they add nothing to cost, force
  // or gradient.  Needs cvel (T2) from bias_forces for aref.
  VNL_HD void make_constraint(int cvel) const {
    V3 O = ref_point();
    V3 n = v3(m.pnx, m.pny, m.pnz), pp = v3(m.ppx, m.ppy, m.ppz);
    VNL_FOR(r, MI(nlimit)) {
      vreal q = s[LO(qpos) + m.lim_qadr[r]];
      vreal dlo = q - m.lim_lo[r], dhi = m.lim_hi[r] - q;
      vreal pos = fmin(dlo, dhi) - m.lim_margin[r];
      vreal sign = dlo < dhi ? vreal(1.) : vreal(-1.);
      vreal k, b, imp;
      kbimp(m.lim_solref + 2 * r, m.lim_solimp + 5 * r, m.dt, pos, k, b, imp);
      vreal R = fmax(m.lim_invweight[r] * (vreal(1.) - imp) / imp, VNL_MINVAL);
      // the sign of the limit Jacobian (+1 lower / -1 upper side) rides on efc_D: D = 0 <=> row absent
      s[LO(efc_D) + r] = pos < vreal(0.) ? sign / R : vreal(0.);
      s[LO(Jaref) + r] = b * (sign * s[LO(qvel) + m.lim_dof[r]]) + k * imp * pos;  // -aref
    }
    VNL_FOR(g, MI(ncg)) {
      int bd = m.cg_body[g], c0 = m.cg_conadr[g], type = m.cg_type[g];
      const int obd = m.body_out[bd];
      Q4 bq = gquat4(obd);
      V3 gpos = gpos3(obd) + qrot(t3(m.cg_pos, g), bq);
      M3 R = qmat(qmul(bq, t4(m.cg_quat, g)));
      V3 size = t3(m.cg_size, g);
      vreal dist0 = vreal(0.), dist1 = vreal(0.);
      V3 cpos0 = v3(0, 0, 0), cpos1 = v3(0, 0, 0), t1 = v3(m.t1x, m.t1y, m.t1z);
      int nc = 1;
      if (type == VNL_GEOM_CAPSULE) {
        nc = 2;
        V3 axis = V3{R.a[2], R.a[5], R.a[8]};
        V3 bv = axis - n * dot(n, axis);
        vreal bn = sqrt(dot(bv, bv));
        if (bn < vreal(0.5)) {
          bv = (vreal(-0.5) < n.y && n.y < vreal(0.5)) ? v3(vreal(0.), vreal(1.), vreal(0.)) : v3(vreal(0.), vreal(0.), vreal(1.));
        } else {
          bv = bv * (vreal(1.) / bn);
        }
        t1 = bv;
        V3 ca = gpos + axis * size.y, cb = gpos - axis * size.y;
        dist0 = dot(ca - pp, n) - size.x, dist1 = dot(cb - pp, n) - size.x;
        cpos0 = ca - n * (size.x + vreal(0.5) * dist0), cpos1 = cb - n * (size.x + vreal(0.5) * dist1);
      } else if (type == VNL_GEOM_SPHERE) {
        dist0 = dot(gpos - pp, n) - size.x;
        cpos0 = gpos - n * (size.x + vreal(0.5) * dist0);
      } else {
        V3 ln = V3{(R.a[0] * n.x + R.a[3] * n.y + R.a[6] * n.z) * size.x, (R.a[1] * n.x + R.a[4] * n.y + R.a[7] * n.z) * size.y,
                   (R.a[2] * n.x + R.a[5] * n.y + R.a[8] * n.z) * size.z};
        vreal nn = sqrt(dot(ln, ln));
        vreal inv = nn > vreal(0.) ? vreal(1.) / nn : vreal(1.);
        V3 sup = V3{-ln.x * inv * size.x, -ln.y * inv * size.y, -ln.z * inv * size.z};
        V3 pos = gpos + mmul(R, sup);
        dist0 = dot(n, pos - pp);
        cpos0 = pos - n * (dist0 * vreal(0.5));
      }
      V3 t2 = cross(n, t1);
      vreal mu = m.cg_mu[g], margin = m.cg_margin[g], invw = m.cg_invweight[g];
      S6 vel = ld6(cvel + 6 * bd);
      for (int q = 0; q < nc; q++) {
        int c = c0 + q, r0 = MI(nlimit) + 4 * c;
        vreal dist = q == 0 ? dist0 : dist1;
        V3 rel = (q == 0 ? cpos0 : cpos1) - O;
        vreal d = dist - margin;
        st3(LO(con_r) + 3 * c, rel);
        if (q == 0) st3(LO(con_t1) + 3 * g, t1);
        vreal k, b, imp;
        kbimp(m.cg_solref + 2 * g, m.cg_solimp + 5 * g, m.dt, d, k, b, imp);
        vreal Rr = fmax(invw * (vreal(1.) - imp) / imp, VNL_MINVAL);
        vreal D = d < vreal(0.) ? vreal(1.) / Rr : vreal(0.);
        V3 pv = vel.l + cross(vel.a, rel);
        vreal jn = dot(n, pv), j1 = dot(t1, pv) * mu, j2 = dot(t2, pv) * mu;
        vreal kp = k * imp * d;
        s[LO(efc_D) + r0] = D, s[LO(efc_D) + r0 + 1] = D, s[LO(efc_D) + r0 + 2] = D, s[LO(efc_D) + r0 + 3] = D;
        s[LO(Jaref) + r0] = b * (jn + j1) + kp, s[LO(Jaref) + r0 + 1] = b * (jn - j1) + kp;  // -aref
        s[LO(Jaref) + r0 + 2] = b * (jn + j2) + kp, s[LO(Jaref) + r0 + 3] = b * (jn - j2) + kp;
      }
    }
    VNL_SYNC();
    {  // list of contacts with D != 0 (typically a handful of the 59), in contact order
      unsigned char* act = (unsigned char*)(s + LO(act_list));
      int na = 0;
      VNL_FOR(c, VNL_PAD_ITEMS(MI(ncon))) {
        const bool on = c < MI(ncon) && s[LO(efc_D) + MI(nlimit) + 4 * (c < MI(ncon) ? c : 0)] != vreal(0.);
        int pos;
        VNL_RANK(on, na, pos);
        if (on) act[pos] = (unsigned char)c;
      }
      VNL_SERIAL { ((int*)(s + LO(act_list)))[(MI(ncon) + 3) / 4] = na; }
    }
    VNL_FOR(r, MI(nefc)) s[LO(jv) + r] = vreal(0.);  // rows of inactive contacts are never written again
    {  // the rows that exist (D != 0: violated limits, the four pyramid rows of every active contact), in row order:
       // typically a few dozen of the 303, so the line search keeps ONE row per lane (line_search<1, compact>)
      unsigned short* lv = live_rows();
      int nl = 0;
      VNL_FOR(r, VNL_PAD_ITEMS(MI(nefc))) {
        const bool on = r < MI(nefc) && s[LO(efc_D) + (r < MI(nefc) ? r : 0)] != vreal(0.);
        int pos;
        VNL_RANK(on, nl, pos);
        if (on && pos < VNL_LIVE_MAX) lv[pos] = (unsigned short)r;
      }
      VNL_SERIAL { ((int*)(s + LO(act_list)))[(MI(ncon) + 3) / 4 + 1] = nl; }
    }
    VNL_SYNC();
    if (trace) {  // debug trace: which rows exist (the pre-solver discrete decisions)
      VNL_SERIAL {
        for (int w = 0; w < 16; w++) {
          int bits = 0;
          for (int b = 0; b < 32; b++) {
            const int r = 32 * w + b;
            if (r < MI(nefc) && s[LO(efc_D) + r] != vreal(0.)) bits |= 1 << b;
          }
          trace[VNL_TRACE_ROWS + w] = bits;
        }
      }
    }
  }

  // Q[k] = sum_{d<k} cdof_d * vec_d, k = 0 .. nv (6 floats each): a wave scan carried across the trips
  VNL_HD void dof_prefix(int vec, int Q) const {
    S6 run = S6{v3(0, 0, 0), v3(0, 0, 0)}, carry = run;
    VNL_FOR(d, VNL_PAD_ITEMS(MI(nv))) {  // every lane takes part in the scans of every trip
      S6 x = S6{v3(0, 0, 0), v3(0, 0, 0)};
      if (d < MI(nv)) x = ld6(LO(cdof) + 6 * d) * s[vec + d];
      VNL_SCAN_ADD_C(x.a.x, run.a.x, carry.a.x), VNL_SCAN_ADD_C(x.a.y, run.a.y, carry.a.y);
      VNL_SCAN_ADD_C(x.a.z, run.a.z, carry.a.z), VNL_SCAN_ADD_C(x.l.x, run.l.x, carry.l.x);
      VNL_SCAN_ADD_C(x.l.y, run.l.y, carry.l.y), VNL_SCAN_ADD_C(x.l.z, run.l.z, carry.l.z);
      if (d < MI(nv)) st6(Q + 6 * (d + 1), x);
      if (d == 0) st6(Q, S6{v3(0, 0, 0), v3(0, 0, 0)});
    }
    VNL_SYNC();
  }
  // sum over the (at most 4) runs of consecutive dofs on a body's path: Q[end] - Q[begin] each
  VNL_HD S6 path_sum(int Q, const int* seg) const {
    S6 acc = S6{v3(0, 0, 0), v3(0, 0, 0)};
#pragma unroll
    for (int k = 0; k < 4; k++) {
      if (k >= MI(path_runs)) break;  // (no body of this model has more runs: two for the rodent)
      const int sg = seg[k], b = sg & 0xff, e = sg >> 8;  // (an unused run is 0 | 0 << 8: Q[0] - Q[0])
      S6 qe = ld6(Q + 6 * e), qb = ld6(Q + 6 * b);
      acc = S6{acc.a + (qe.a - qb.a), acc.l + (qe.l - qb.l)};
    }
    return acc;
  }

  // out[r] = (J vec)[r]  (accumulate: out[r] += ...).  Limit rows: one lane each.  Contact rows: the twist of
  // a contact's body is the sum of cdof_d * vec_d over the dofs d on the body's path; those dofs are a few runs of
  // consecutive indices (host table body_pathseg), so ONE prefix-sum array Q[k] = sum_{d<k} cdof_d vec_d (a wave
  // scan, carried across the trips) serves every contact: twist = sum over runs of Q[end] - Q[begin].
  VNL_HD void jac_mul(int vec, int out, bool accumulate) const {
    V3 n = v3(m.pnx, m.pny, m.pnz);
    for_live_rows([&](int r) {
      if (r < MI(nlimit)) {
        vreal v = copysign(vreal(1.), s[LO(efc_D) + r]) * s[vec + m.lim_dof[r]];
        s[out + r] = accumulate ? s[out + r] + v : v;
      }
    });
    const int Q = LO(P) + 3 * MI(nefc);  // the contact-wrench area of constraint_force: dead here
    dof_prefix(vec, Q);
    const unsigned char* act = (const unsigned char*)(s + LO(act_list));
    int na = ((const int*)(s + LO(act_list)))[(MI(ncon) + 3) / 4];
    VNL_FOR(j, na) {
      int c = act[j], g = m.con_geom[c] & 0xff, r0 = MI(nlimit) + 4 * c;
      const int* seg = m.body_pathseg + 8 * con_body(c);
      S6 vel = path_sum(Q, seg);
      vreal mu = m.cg_mu[g];
      V3 rel = ld3(LO(con_r) + 3 * c), t1 = ld3(LO(con_t1) + 3 * g), t2 = cross(n, t1);
      V3 pv = vel.l + cross(vel.a, rel);
      vreal jn = dot(n, pv), j1 = dot(t1, pv) * mu, j2 = dot(t2, pv) * mu;
      vreal o0 = jn + j1, o1 = jn - j1, o2 = jn + j2, o3 = jn - j2;
      if (accumulate) o0 += s[out + r0], o1 += s[out + r0 + 1], o2 += s[out + r0 + 2], o3 += s[out + r0 + 3];
      s[out + r0] = o0, s[out + r0 + 1] = o1, s[out + r0 + 2] = o2, s[out + r0 + 3] = o3;
    }
    VNL_SYNC();
  }

  // 0.5 * sum_r D Jaref^2 [Jaref<0]
  VNL_HD vreal constraint_cost(int jaref) const {
    vreal c = vreal(0.);
    for_live_rows([&](int r) {
      vreal x = s[jaref + r];
      c += x < vreal(0.) ? fabs(s[LO(efc_D) + r]) * x * x : vreal(0.);
    });
    return vreal(0.5) * vnl_wave_sum(c);
  }

  // qfrc_constraint = J' f with f = -D Jaref [Jaref<0]; returns the constraint cost.
  // Per-contact wrenches (about O) are formed in parallel; then each dof sums the wrenches of the
  // contacts in its body's subtree (bodies are numbered depth-first, so the subtree is a contiguous
  // body range) and projects onto its own cdof.  Only contacts with D != 0 are listed (active list
  // built by make_constraint).
  VNL_HD vreal constraint_force() const {
    V3 n = v3(m.pnx, m.pny, m.pnz);
    int Wc = LO(P) + 3 * MI(nefc);  // after efc_D | Jaref | jv
    vreal cost = vreal(0.);
    for_live_rows([&](int r) {
      if (r < MI(nlimit)) {
        vreal x = s[LO(Jaref) + r];
        cost += x < vreal(0.) ? fabs(s[LO(efc_D) + r]) * x * x : vreal(0.);
      }
    });
    {
      // Contacts are in body order and the subtree of a body is a contiguous body range, so the contacts under a
      // dof are ONE range [c0, c1) (host table): one lane per contact forms its wrench (zero if inactive), a wave
      // scan turns them into prefix sums P[k] = sum_{c<k}, and every dof takes P[c1] - P[c0] (exactly zero where
      // nothing is active) -- no loop over contacts per dof.
      S6 run = S6{v3(0, 0, 0), v3(0, 0, 0)};
      VNL_FOR(k, VNL_WAVE_ITEMS(MI(ncon))) {  // (ncon <= 64: one trip of ALL lanes; idle lanes carry zeros through the scan)
        S6 w = S6{v3(0, 0, 0), v3(0, 0, 0)};
        if (k < MI(ncon)) {
          const int c = (m.con_geom[k] >> 8) & 0xff;  // the k-th contact in body order
          int r0 = MI(nlimit) + 4 * c;
          vreal D = s[LO(efc_D) + r0];
          if (D != vreal(0.)) {
            int g = m.con_geom[c] & 0xff;
            vreal mu = m.cg_mu[g], f[4];
            for (int q = 0; q < 4; q++) {
              vreal x = s[LO(Jaref) + r0 + q];
              f[q] = x < vreal(0.) ? -D * x : vreal(0.);
              cost += x < vreal(0.) ? D * x * x : vreal(0.);
            }
            V3 rel = ld3(LO(con_r) + 3 * c), t1 = ld3(LO(con_t1) + 3 * g), t2 = cross(n, t1);
            V3 Fw = n * (f[0] + f[1] + f[2] + f[3]) + t1 * (mu * (f[0] - f[1])) + t2 * (mu * (f[2] - f[3]));
            w = S6{cross(rel, Fw), Fw};
          }
        }
        VNL_SCAN_ADD(w.a.x, run.a.x), VNL_SCAN_ADD(w.a.y, run.a.y), VNL_SCAN_ADD(w.a.z, run.a.z);
        VNL_SCAN_ADD(w.l.x, run.l.x), VNL_SCAN_ADD(w.l.y, run.l.y), VNL_SCAN_ADD(w.l.z, run.l.z);
        if (k < MI(ncon)) st6(Wc + 6 * (k + 1), w);
        if (k == 0) st6(Wc, S6{v3(0, 0, 0), v3(0, 0, 0)});
      }
      VNL_SYNC();
      VNL_FOR(d, MI(nv)) {
        const int pk = m.dof_limrow[d], r = (pk & 0x3ff) - 1, c0 = (pk >> 10) & 0xff, c1 = (pk >> 18) & 0xff;
        S6 p1 = ld6(Wc + 6 * c1), p0 = ld6(Wc + 6 * c0);
        S6 w = S6{p1.a - p0.a, p1.l - p0.l};
        vreal q = dot(ld6(LO(cdof) + 6 * d), w);
        if (r >= 0) {
          vreal x = s[LO(Jaref) + r];
          q += x < vreal(0.) ? -s[LO(efc_D) + r] * x : vreal(0.);  // sign(D) is the limit Jacobian entry
        }
        s[LO(qfrc_c) + d] = q;
      }
      VNL_SYNC();
      return vreal(0.5) * vnl_wave_sum(cost);
    }
    return vreal(0.);  // (build_dev_model rejects models whose contacts do not fit one wave)
  }

  VNL_HD vreal vdot(int a, int b) const {
    vreal p = vreal(0.);
    VNL_FOR(d, MI(nv)) p += s[a + d] * s[b + d];
    return vnl_wave_sum(p);
  }

  // Line search.  The rows a lane owns (r = lane + j * lanes) and their quadratic coefficients do not
  // change during one line search, so they are loaded into registers once per CG iteration and every
  // trial step length is then a pure register pass + 3 reductions.
  struct LsPoint {
    vreal alpha, cost, d0, d1;
  };
  template <int RPL>
  struct LsRows {
    vreal ja[RPL], jv[RPL], a0[RPL], a1[RPL], a2[RPL];
  };
  // debug trace only: rows of this lane that exist (D != 0) and are active at step length alpha, summed over the wave
  template <int RPL>
  VNL_HD int ls_count_active(const LsRows<RPL>& R, vreal alpha, bool compact) const {
    int n = 0;
#pragma unroll
    for (int j = 0; j < RPL; j++) {
      int r = (int)lane + j * VNL_LANES;
      bool live = r < MI(nefc) && s[LO(efc_D) + (r < MI(nefc) ? r : 0)] != vreal(0.);
      if (compact) live = r < num_live_rows();
      n += VNL_COUNT(live && R.ja[j] + alpha * R.jv[j] < vreal(0.));
    }
    return n;
  }
  template <int RPL>
  VNL_HD void ls_load(LsRows<RPL>& R, bool compact) const {
#pragma unroll
    for (int j = 0; j < RPL; j++) {
      int r = (int)lane + j * VNL_LANES;
      bool ok = r < MI(nefc);
      if (compact) {  // lane l takes the l-th existing row (and the (l + 64)-th, RPL == 2)
        ok = r < num_live_rows();
        r = ok ? (int)live_rows()[r] : 0;
      }
      vreal D = ok ? fabs(s[LO(efc_D) + r]) : vreal(0.);
      vreal ja = ok ? s[LO(Jaref) + r] : vreal(0.), jv = ok ? s[LO(jv) + r] : vreal(0.);
      R.ja[j] = ja, R.jv[j] = jv;
      R.a0[j] = vreal(0.5) * ja * ja * D, R.a1[j] = jv * ja * D, R.a2[j] = vreal(0.5) * jv * jv * D;
    }
  }
  template <int N, int RPL>
  VNL_HD void ls_eval(const LsRows<RPL>& R, const vreal* alpha, vreal qg0, vreal qg1, vreal qg2, LsPoint* out, bool few = false) const {
    vreal q0[N], q1[N], q2[N];
    for (int i = 0; i < N; i++) q0[i] = vreal(0.), q1[i] = vreal(0.), q2[i] = vreal(0.);
#pragma unroll
    for (int j = 0; j < RPL; j++) {
      for (int i = 0; i < N; i++) {
        bool act = R.ja[j] + alpha[i] * R.jv[j] < vreal(0.);
        q0[i] += act ? R.a0[j] : vreal(0.), q1[i] += act ? R.a1[j] : vreal(0.), q2[i] += act ? R.a2[j] : vreal(0.);
      }
    }
    // (few: every row sits in one of the first 16 lanes -- the usual case, a dozen live rows: the short reduction)
    if (few) {
      for (int i = 0; i < N; i++) q0[i] = vnl_wave_sum16(q0[i]), q1[i] = vnl_wave_sum16(q1[i]), q2[i] = vnl_wave_sum16(q2[i]);
    } else {
      for (int i = 0; i < N; i++) q0[i] = vnl_wave_sum(q0[i]), q1[i] = vnl_wave_sum(q1[i]), q2[i] = vnl_wave_sum(q2[i]);
    }
    for (int i = 0; i < N; i++) {
      vreal t0 = qg0 + q0[i], t1 = qg1 + q1[i], t2 = qg2 + q2[i];
      vreal a = alpha[i];
      out[i].alpha = a;
      out[i].cost = a * a * t2 + a * t1 + t0;
      out[i].d0 = vreal(2.) * a * t2 + t1;
      out[i].d1 = vreal(2.) * t2 + (t2 == vreal(0.) ? VNL_MINVAL : vreal(0.));
    }
  }

  // exact line search of solver._linesearch; returns the accepted step length (0 if no improvement)
  template <int RPL, bool COMPACT = false>
  VNL_HD vreal line_search(vreal gauss, vreal qg1, vreal qg2, vreal gtol, int* tr /* debug trace of this iteration or null */) const {
      LsPoint p0, lo, hi;
      LsRows<RPL> rows;
      ls_load(rows, COMPACT);
      const bool few = COMPACT && RPL == 1 && VNL_LANES >= 16 && num_live_rows() <= 16;
      vreal a1[1] = {vreal(0.)};
      ls_eval<1>(rows, a1, gauss, qg1, qg2, &p0, few);
      if (tr) {
        const int n0 = ls_count_active(rows, a1[0], COMPACT);
        VNL_SERIAL { tr[4] = n0, tr[24] = __builtin_bit_cast(int, (float)a1[0]); }
      }
      a1[0] = p0.alpha - p0.d0 / p0.d1;
      ls_eval<1>(rows, a1, gauss, qg1, qg2, &lo, few);
      if (tr) {
        const int n1 = ls_count_active(rows, a1[0], COMPACT);
        VNL_SERIAL { tr[5] = n1, tr[25] = __builtin_bit_cast(int, (float)a1[0]); }
      }
      if (tr) {
        const int first_lo = lo.d0 < p0.d0;
        VNL_SERIAL { tr[2] |= first_lo << 24; }
      }
      if (lo.d0 < p0.d0) {
        hi = p0;
      } else {
        hi = lo, lo = p0;
      }
      bool swap = true;
      for (int li = 0; li < MI(ls_iterations); li++) {
        if (!swap || (lo.d0 < vreal(0.) && lo.d0 > -gtol) || (hi.d0 > vreal(0.) && hi.d0 < gtol)) break;
        vreal a3[3] = {lo.alpha - lo.d0 / lo.d1, hi.alpha - hi.d0 / hi.d1, vreal(0.5) * (lo.alpha + hi.alpha)};
        LsPoint p[3];
        ls_eval<3>(rows, a3, gauss, qg1, qg2, p, few);
        bool s1 = (lo.d0 > vreal(0.)) || (lo.d0 < p[0].d0);
        if (s1) lo = p[0];
        bool s2 = (p[2].d0 < vreal(0.)) && (lo.d0 < p[2].d0);
        if (s2) lo = p[2];
        bool s3 = (hi.d0 < vreal(0.)) || (hi.d0 > p[1].d0);
        if (s3) hi = p[1];
        bool s4 = (p[2].d0 > vreal(0.)) && (hi.d0 > p[2].d0);
        if (s4) hi = p[2];
        swap = s1 || s2 || s3 || s4;
        if (tr) {
          const int c0 = ls_count_active(rows, a3[0], COMPACT), c1 = ls_count_active(rows, a3[1], COMPACT), c2 = ls_count_active(rows, a3[2], COMPACT);
          VNL_SERIAL {
            if (li < 6) {
              tr[6 + 3 * li] = c0, tr[7 + 3 * li] = c1, tr[8 + 3 * li] = c2;
              tr[26 + 3 * li] = __builtin_bit_cast(int, (float)a3[0]), tr[27 + 3 * li] = __builtin_bit_cast(int, (float)a3[1]);
              tr[28 + 3 * li] = __builtin_bit_cast(int, (float)a3[2]);
              tr[2] |= ((int)s1 | (int)s2 << 1 | (int)s3 << 2 | (int)s4 << 3) << (4 * li);
            }
            tr[1] = li + 1;
          }
        }
      }
      bool improved = (lo.cost < p0.cost) || (hi.cost < p0.cost);
      const vreal accepted = improved ? (lo.cost < hi.cost ? lo.alpha : hi.alpha) : vreal(0.);
      if (tr) {
        VNL_SERIAL {
          tr[0] = __builtin_bit_cast(int, (float)accepted);
          tr[3] = improved ? (lo.cost < hi.cost ? 1 : 2) : 0;
        }
      }
      return accepted;
  }

  // ---- Newton solver (solver.py _update_gradient, SolverType.NEWTON): Mgrad = H^-1 grad with the Hessian of the cost at the
  // current active set, H = qM + J' diag(efc_D * active) J -- formed and Cholesky-factorised dense in LDS (small models: the
  // reference selects it for the ant, nv 14, configs/env_config.yaml:16-21).  efc_J is materialised once per substep.
  VNL_HD void newton_jacobian() const {
    const int nv = MI(nv);
    V3 n = v3(m.pnx, m.pny, m.pnz);
    VNL_FOR(k, MI(nefc) * nv) s[LO(newt_J) + k] = vreal(0.);
    VNL_SYNC();
    VNL_FOR(r, MI(nlimit)) s[LO(newt_J) + r * nv + m.lim_dof[r]] = copysign(vreal(1.), s[LO(efc_D) + r]);
    VNL_FOR(q, MI(ncon) * nv) {
      const int c = q / nv, d = q - c * nv, g = m.con_geom[c] & 0xff, r0 = MI(nlimit) + 4 * c;
      if (s[LO(efc_D) + r0] == vreal(0.)) continue;
      const int* seg = m.body_pathseg + 8 * con_body(c);
      bool on_path = false;
#pragma unroll
      for (int k = 0; k < 4; k++) on_path = on_path || (d >= (seg[k] & 0xff) && d < (seg[k] >> 8));
      if (!on_path) continue;
      const S6 cd = ld6(LO(cdof) + 6 * d);
      const V3 rel = ld3(LO(con_r) + 3 * c), t1 = ld3(LO(con_t1) + 3 * g), t2 = cross(n, t1);
      const V3 pv = cd.l + cross(cd.a, rel);
      const vreal mu = m.cg_mu[g], jn = dot(n, pv), j1 = dot(t1, pv) * mu, j2 = dot(t2, pv) * mu;
      s[LO(newt_J) + r0 * nv + d] = jn + j1, s[LO(newt_J) + (r0 + 1) * nv + d] = jn - j1;
      s[LO(newt_J) + (r0 + 2) * nv + d] = jn + j2, s[LO(newt_J) + (r0 + 3) * nv + d] = jn - j2;
    }
    VNL_SYNC();
  }
  // x <- H^-1 x (x: a dof vector in LDS); Jaref as it stands decides the active set
  VNL_HD void newton_solve(int x) const {
    const int nv = MI(nv), H = LO(newt_H);
    VNL_FOR(k, nv * nv) {
      const int i = k / nv, j = k - i * nv;
      vreal h = s[LO(newt_M) + k];
      for (int r = 0; r < MI(nefc); r++) {
        const vreal D = s[LO(efc_D) + r];
        if (D != vreal(0.) && s[LO(Jaref) + r] < vreal(0.)) h += fabs(D) * s[LO(newt_J) + r * nv + i] * s[LO(newt_J) + r * nv + j];
      }
      s[H + k] = h;
    }
    VNL_SYNC();
    // dense Cholesky, lower, column by column (cho_factor): the diagonal by one lane, the column below it one row per lane
    for (int j = 0; j < nv; j++) {
      VNL_SERIAL {
        vreal d = s[H + j * nv + j];
        for (int k = 0; k < j; k++) d -= s[H + j * nv + k] * s[H + j * nv + k];
        if (!(d > vreal(0.))) d = VNL_MINVAL;
        s[H + j * nv + j] = sqrt(d);
      }
      VNL_SYNC();
      VNL_FOR(i, nv) {
        if (i > j) {
          vreal t = s[H + i * nv + j];
          for (int k = 0; k < j; k++) t -= s[H + i * nv + k] * s[H + j * nv + k];
          s[H + i * nv + j] = t / s[H + j * nv + j];
        }
      }
      VNL_SYNC();
    }
    VNL_SERIAL {  // cho_solve: forward and back substitution
      for (int i = 0; i < nv; i++) {
        vreal v = s[x + i];
        for (int k = 0; k < i; k++) v -= s[H + i * nv + k] * s[x + k];
        s[x + i] = v / s[H + i * nv + i];
      }
      for (int i = nv - 1; i >= 0; i--) {
        vreal v = s[x + i];
        for (int k = i + 1; k < nv; k++) v -= s[H + k * nv + i] * s[x + k];
        s[x + i] = v / s[H + i * nv + i];
      }
    }
    VNL_SYNC();
  }
  // out = qM v (dense copy): with search = -H^-1 grad the recurrence M s' = -grad + beta M s of the CG route does not hold
  VNL_HD void newton_mass_mul(int v, int out) const {
    const int nv = MI(nv);
    VNL_FOR(i, nv) {
      vreal acc = vreal(0.);
      for (int j = 0; j < nv; j++) acc += s[LO(newt_M) + i * nv + j] * s[v + j];
      s[out + i] = acc;
    }
    VNL_SYNC();
  }

  // solver.solve (CG / Newton).  One env per wave: the while loops run with this env's own trip counts.
  VNL_HD void solve() const {
    const int nv = MI(nv), ne = MI(nefc);
    // --- warm start selection: cost at qacc_warmstart vs qacc_smooth.
    // On entry: Jaref holds -aref (make_constraint), mv holds M*warm, qacc holds warm (forward()).
    fresh().jac_mul(LO(qacc_smooth), LO(Jaref), true);  // Jaref(qacc_smooth) = J qacc_smooth - aref
    vreal cost_s = fresh().constraint_cost(LO(Jaref));  // gauss term vanishes: M qacc_smooth = qfrc_smooth
    VNL_FOR(d, nv) s[LO(tmp) + d] = s[LO(qacc) + d] - s[LO(qacc_smooth) + d];
    VNL_SYNC();
    fresh().jac_mul(LO(tmp), LO(jv), false);  // J (warm - smooth)
    for_live_rows([&](int r) { s[LO(jv) + r] += s[LO(Jaref) + r]; });  // Jaref(warm)
    vreal gw = vreal(0.);
    VNL_FOR(d, nv) gw += (s[LO(mv) + d] - s[LO(smooth) + d]) * s[LO(tmp) + d];
    gw = vnl_wave_sum(gw);
    VNL_SYNC();
    vreal cost_w = fresh().constraint_cost(LO(jv)) + vreal(0.5) * gw;
    bool use_warm = cost_w < cost_s;
    if (trace) {
      VNL_FOR(k, VNL_TRACE_ROWS) trace[k] = k == 0 ? (int)use_warm : 0;
    }
    VNL_FOR(d, nv) {
      s[LO(qacc) + d] = use_warm ? s[LO(qacc) + d] : s[LO(qacc_smooth) + d];
      s[LO(Ma) + d] = use_warm ? s[LO(mv) + d] : s[LO(smooth) + d];
    }
    if (use_warm) for_live_rows([&](int r) { s[LO(Jaref) + r] = s[LO(jv) + r]; });
    VNL_SYNC();
    VNL_PROF(14);
    vreal gauss = use_warm ? vreal(0.5) * gw : vreal(0.);
    vreal cost = fresh().constraint_force() + gauss;
    vreal prev_cost = INFINITY;
    // The three dot products the iteration's head needs -- |grad|^2, |search|^2 and grad . Mgrad -- are accumulated in the
    // loops that PRODUCE those vectors (same lane order, same wave sum: the same bits as separate passes), not in passes
    // of their own at the top of every iteration.
    vreal gg = vreal(0.), ss = vreal(0.), gp = vreal(0.);
    VNL_FOR(d, nv) {
      vreal g = s[LO(Ma) + d] - s[LO(smooth) + d] - s[LO(qfrc_c) + d];
      s[LO(grad) + d] = g, s[LO(Mgrad) + d] = g;
      gg += g * g;
    }
    gg = vnl_wave_sum(gg);
    VNL_SYNC();
    const bool newton = MI(solver_newton) != 0;
    if (newton) {
      fresh().newton_jacobian();
      fresh().newton_solve(LO(Mgrad));
    } else {
      fresh().solve_inplace(LO(Mgrad));
    }
    VNL_FOR(d, nv) {
      const vreal mg = s[LO(Mgrad) + d], gr = s[LO(grad) + d];
      s[LO(search) + d] = -mg;
      s[LO(mv) + d] = -gr;  // M search (CG: search = -M^-1 grad)
      ss += mg * mg, gp += gr * mg;
    }
    ss = vnl_wave_sum(ss), gp = vnl_wave_sum(gp);
    VNL_SYNC();
    if (newton) fresh().newton_mass_mul(LO(search), LO(mv));
    VNL_PROF(15);

    for (int it = 0; it < MI(iterations); it++) {
      vreal improvement = (prev_cost - cost) / m.scale;
      vreal gradient = sqrt(gg) / m.scale;
      if (improvement < m.tolerance || gradient < m.tolerance) break;
      // ---- line search
      vreal smag = sqrt(ss) * m.scale;
      vreal gtol = m.tolerance * m.ls_tolerance * smag;
      VNL_PROF(16);
      fresh().jac_mul(LO(search), LO(jv), false);
      VNL_PROF(17);
      vreal qg1 = vreal(0.), qg2 = vreal(0.);
      VNL_FOR(d, nv) {
        vreal sd = s[LO(search) + d];
        qg1 += sd * s[LO(Ma) + d] - sd * s[LO(smooth) + d];
        qg2 += sd * s[LO(mv) + d];
      }
      qg1 = vnl_wave_sum(qg1), qg2 = vreal(0.5) * vnl_wave_sum(qg2);
      VNL_PROF(18);
      int* tr = (trace && it < VNL_TRACE_ITERS) ? trace + 8 + VNL_TRACE_REC * it : nullptr;
      if (trace) {
        VNL_SYNC();  // (the zero fill above is by all lanes, the entries by lane 0)
        VNL_SERIAL { trace[1] = it + 1; }
      }
      // (rows that exist: typically a dozen to a few dozen of the 303 -- one per lane; along a free-running rollout a tenth of
      // the envs has 65 .. 130 -- two per lane; only beyond that every row slot of the model is visited)
      const int nlive = num_live_rows();
      vreal alpha = (nlive <= (VNL_LANES < VNL_LIVE_MAX ? VNL_LANES : VNL_LIVE_MAX))
                        ? fresh().template line_search<1, true>(gauss, qg1, qg2, gtol, tr)
                        : (nlive <= (2 * VNL_LANES < VNL_LIVE_MAX ? 2 * VNL_LANES : VNL_LIVE_MAX))
                        ? fresh().template line_search<2, true>(gauss, qg1, qg2, gtol, tr)
                        : ((MI(nefc) <= 5 * VNL_LANES) ? fresh().template line_search<VNL_ROWS_SMALL>(gauss, qg1, qg2, gtol, tr)
                                                     : fresh().template line_search<VNL_ROWS_PER_LANE>(gauss, qg1, qg2, gtol, tr));
      VNL_FOR(d, nv) {
        s[LO(qacc) + d] += alpha * s[LO(search) + d];
        s[LO(Ma) + d] += alpha * s[LO(mv) + d];
      }
      VNL_PROF(19);
      for_live_rows([&](int r) { s[LO(Jaref) + r] += alpha * s[LO(jv) + r]; });
      VNL_SYNC();
      VNL_PROF(20);
      // ---- constraint + gradient update   (gp = grad . Mgrad of the vectors as they stand: carried, see above)
      vreal g = vreal(0.);
      VNL_FOR(d, nv) g += (s[LO(Ma) + d] - s[LO(smooth) + d]) * (s[LO(qacc) + d] - s[LO(qacc_smooth) + d]);
      g = vnl_wave_sum(g);
      VNL_PROF(21);
      vreal ncost = fresh().constraint_force() + vreal(0.5) * g;
      VNL_PROF(22);
      prev_cost = cost, cost = ncost, gauss = vreal(0.5) * g;
      vreal d1 = vreal(0.);
      gg = vreal(0.);
      VNL_FOR(d, nv) {
        vreal gn = s[LO(Ma) + d] - s[LO(smooth) + d] - s[LO(qfrc_c) + d];
        d1 += gn * s[LO(Mgrad) + d];
        s[LO(grad) + d] = gn, s[LO(tmp) + d] = gn;
        gg += gn * gn;
      }
      d1 = vnl_wave_sum(d1), gg = vnl_wave_sum(gg);
      VNL_SYNC();
      VNL_PROF(23);
      if (newton) fresh().newton_solve(LO(tmp));
      else fresh().solve_inplace(LO(tmp));
      VNL_PROF(24);
      vreal d2 = vdot(LO(grad), LO(tmp));
      vreal beta = fmax(vreal(0.), (d2 - d1) / fmax(VNL_MINVAL, gp));
      if (newton) beta = vreal(0.);  // solver.solve: search = -Mgrad
      gp = d2;  // grad . Mgrad with Mgrad := tmp below
      ss = vreal(0.);
      VNL_FOR(d, nv) {
        vreal mg = s[LO(tmp) + d];
        s[LO(Mgrad) + d] = mg;
        const vreal sn = -mg + beta * s[LO(search) + d];
        s[LO(search) + d] = sn;
        s[LO(mv) + d] = -s[LO(grad) + d] + beta * s[LO(mv) + d];
        ss += sn * sn;
      }
      ss = vnl_wave_sum(ss);
      VNL_SYNC();
      if (newton) fresh().newton_mass_mul(LO(search), LO(mv));
      VNL_PROF(25);
    }
  }

  // forward.forward.  `warm` = qacc_warmstart (HBM on the first substep, LDS qacc afterwards).
  VNL_HD void forward(const vreal* warm) const {
    VNL_FOR(d, MI(nv)) s[LO(tmp) + d] = warm[d];
    VNL_SYNC();
    VNL_FOR(d, MI(nv)) s[LO(qacc) + d] = s[LO(tmp) + d];
    VNL_SYNC();
    // Timing knob (VNL_DBG_REPEAT=stage:count at env creation; 0 in normal use): run one stage
    // `count` extra times on data that is recomputed afterwards, so results are unchanged and the
    // wall-time difference prices that stage in the real (uninstrumented) build.
#ifdef VNL_STAGE_KNOBS  // diagnostic library only (csrc/build.py --knobs): five extra inlined copies of the stages
    for (int rep = 0; rep < m.dbg_count; rep++) {
      if (m.dbg_stage == 1) {
        kinematics();
        body_inertias(false);
        mass_matrix(vreal(0.));
        factor();
        invert_factor();
      } else if (m.dbg_stage == 2) {
        kinematics();
      } else if (m.dbg_stage == 3) {
        kinematics();
        body_inertias(false);
        mass_matrix(vreal(0.));
      } else if (m.dbg_stage == 4) {
        kinematics();
        body_inertias(false);
        mass_matrix(vreal(0.));
        factor();
      } else if (m.dbg_stage == 5) {
        kinematics();
        body_inertias(true);
        (void)bias_forces();
      } else if (m.dbg_stage == 14) {
        kinematics();
        body_inertias(false);
        mass_matrix(vreal(0.));
        factor(false);
      } else if (m.dbg_stage == 15) {
        kinematics();
        body_inertias(false);
        tree_accumulate(LO(P), 10);
        if (factor_pair_ok()) factor_both(m.dt);
      } else if (m.dbg_stage == 17) {
        kinematics();
        body_inertias(false);
        tree_accumulate(LO(P), 10);
        if (factor_pair_ok()) {
          factor_both(m.dt);
          invert_both(fac2());
        }
      } else if (m.dbg_stage == 16) {  // euler()'s second-factor route: reload, apply
        const vreal* g2 = fac2();
        VNL_FOR(k, MI(nM) + MI(nv)) s[LO(P) + k] = g2[k];
        VNL_SYNC();
        solve_inplace(LO(tmp), LO(P), LO(P) + MI(nM));
      } else if (m.dbg_stage == 18) {
        tree_accumulate(LO(P), 10);
      } else if (m.dbg_stage == 19) {
        body_inertias(false);
      }
    }
#endif
    fresh().kinematics();
    VNL_PROF(0);
    fresh().body_inertias(true);
    VNL_PROF(1);
    int cvel = fresh().bias_forces();
    if (factor_pair_ok()) {
      // (the factors come from the composite inertias directly; the matrix itself only where the Newton solver wants its dense copy)
      if (MI(solver_newton)) fresh().mass_matrix(vreal(0.));
      else fresh().tree_accumulate(LO(P), 10);
      VNL_PROF(5);
      VNL_PROF(6);
      fresh().factor_both(m.dt);
      fresh().mass_mul_factor(LO(qacc), LO(mv));  // M * qacc_warmstart, needs L1 (before N1 takes its place)
      VNL_PROF(10);
      fresh().invert_both(fac2());
    } else {
      fresh().mass_matrix(vreal(0.));
      fresh().factor();
      fresh().mass_mul_factor(LO(qacc), LO(mv));
      VNL_PROF(10);
      fresh().invert_factor();
    }
    VNL_PROF(11);
    fresh().smooth_forces();
    VNL_PROF(12);
    fresh().make_constraint(cvel);
    VNL_PROF(13);
#ifdef VNL_STAGE_KNOBS
    for (int rep = 0; rep < m.dbg_count; rep++) {
      if (m.dbg_stage == 6) {
        jac_mul(LO(qacc_smooth), LO(jv), false);
      } else if (m.dbg_stage == 7) {
        VNL_FOR(d, MI(nv)) s[LO(tmp) + d] = s[LO(smooth) + d];
        VNL_SYNC();
        solve_inplace(LO(tmp));
      } else if (m.dbg_stage == 8) {
        vreal a3[3] = {vreal(0.), vreal(1e-4), vreal(2e-4)};
        LsPoint p[3];
        LsRows<VNL_ROWS_SMALL> rows;
        ls_load(rows, false);
        ls_eval<3>(rows, a3, vreal(0.), vreal(0.), vreal(0.), p);
        if (p[0].cost == vreal(-1.)) s[LO(tmp)] = p[1].cost;  // keep the result alive
      } else if (m.dbg_stage == 9) {
        make_constraint(cvel);
      } else if (m.dbg_stage == 10) {
        vreal c = constraint_force();
        if (c == vreal(-1.)) s[LO(tmp)] = c;
      } else if (m.dbg_stage == 11) {
        smooth_forces();
      } else if (m.dbg_stage == 12) {
        vreal c = vdot(LO(smooth), LO(qacc_smooth)) + vdot(LO(qacc), LO(smooth)) + vdot(LO(qacc), LO(qacc)) + vdot(LO(smooth), LO(smooth));
        if (c == vreal(-1.)) s[LO(tmp)] = c;
      } else if (m.dbg_stage == 13) {
        LsRows<VNL_ROWS_SMALL> rows;
        ls_load(rows, false);
        vreal a1[1] = {vreal(1e-4)};
        LsPoint p;
        ls_eval<1>(rows, a1, vreal(0.), vreal(0.), vreal(0.), &p);
        if (p.cost == vreal(-1.)) s[LO(tmp)] = p.cost;
      }
    }
#endif
    fresh().solve();
  }

  // forward.euler + _advance; leaves qacc (warm start) in LO(qacc) and the new state in LO(qpos)/qvel/act
  VNL_HD void euler() const {
    const int nv = MI(nv);
    VNL_PROF(26);  // the tail of solve()
    VNL_FOR(d, nv) s[LO(tmp) + d] = MI(eulerdamp) ? s[LO(smooth) + d] + s[LO(qfrc_c) + d] : s[LO(qacc) + d];
    VNL_SYNC();
    if (MI(eulerdamp) && factor_pair_ok()) {
      // the INVERTED factor of M + h diag(damping) was made beside M's by forward() (factor_aba + invert_aba): bring it
      // into the pool (the constraint rows are dead now) and apply it
      const vreal* g2 = fac2();
      const int n2 = MI(nM) + MI(nv);
      VNL_SYNC_GLOBAL();  // (written by invert_aba, one row per lane; read here element by element)
      VNL_FOR(k, n2) s[LO(P) + k] = g2[k];
      VNL_SYNC();
      fresh().solve_inplace(LO(tmp), LO(P), LO(P) + MI(nM));
      VNL_PROF(27);
    } else if (MI(eulerdamp)) {
      fresh().body_inertias(false);
      VNL_PROF(1);
      fresh().mass_matrix(m.dt);
      if (!fresh().factor_solve(LO(tmp))) {
        fresh().factor();
        fresh().invert_factor();
        fresh().solve_inplace(LO(tmp));
      }
      VNL_PROF(27);
    }
    VNL_FOR(i, MI(nu)) {
      if (m.act_tau[i] >= vreal(0.)) s[LO(act) + i] += s[LO(actdot) + i] * m.dt;
    }
    VNL_FOR(d, nv) s[LO(qvel) + d] += s[LO(tmp) + d] * m.dt;
    VNL_SYNC();
    VNL_FOR(j, MI(njnt)) {
      int qa = m.jnt_qposadr[j], da = m.jnt_dofadr[j];
      if (m.jnt_type[j] == VNL_JNT_FREE) {
        for (int t = 0; t < 3; t++) s[LO(qpos) + qa + t] += m.dt * s[LO(qvel) + da + t];
        V3 w = ld3(LO(qvel) + da + 3);
        vreal nrm = sqrt(dot(w, w));
        V3 ax = nrm > vreal(0.) ? w * (vreal(1.) / nrm) : w;
        vreal ang = m.dt * nrm, sn = sin(vreal(0.5) * ang), cs = cos(vreal(0.5) * ang);
        Q4 r = qmul(ld4(LO(qpos) + qa + 3), Q4{cs, ax.x * sn, ax.y * sn, ax.z * sn});
        vreal n2 = sqrt(r.w * r.w + r.x * r.x + r.y * r.y + r.z * r.z);
        vreal inv = n2 > vreal(0.) ? vreal(1.) / n2 : vreal(1.);
        s[LO(qpos) + qa + 3] = r.w * inv, s[LO(qpos) + qa + 4] = r.x * inv, s[LO(qpos) + qa + 5] = r.y * inv,
                        s[LO(qpos) + qa + 6] = r.z * inv;
      } else {
        s[LO(qpos) + qa] += m.dt * s[LO(qvel) + da];
      }
    }
    VNL_SYNC();
    VNL_PROF(28);
  }

  // ------------------------------------------------------------------ env glue
  VNL_HD static int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
  // jp.nan_to_num
  VNL_HD static vreal nan0(vreal x) {
    if (x != x) return vreal(0.);
    if (x > vreal(3.4028235e38)) return vreal(3.4028235e38);
    if (x < -vreal(3.4028235e38)) return -vreal(3.4028235e38);
    return x;
  }

  // rodent.py:241-264 (matrix 1-norm over the tracked bodies, L1 over joints); qpos / xpos given
  // as pointers so that both the carried (global) and the fresh (LDS) state can be scored
  VNL_HD vreal termination(int clip, int frame, const vreal* qpos, const vreal* xpos) const {
    int f = clampi(frame, 0, ev.T - 1), nj = MI(nq) - 7;
    const float* cj = ev.joints + ((size_t)clip * ev.T + f) * nj;
    const float* cb = ev.body_positions + ((size_t)clip * ev.T + f) * ev.nb * 3;
    vreal ej = vreal(0.), cx = vreal(0.), cy = vreal(0.), cz = vreal(0.);
    VNL_FOR(i, nj) ej += fabs(vreal(cj[i]) - qpos[7 + i]);
    VNL_FOR(k, ev.nb) {
      int bd = ev.body_idxs[k];
      cx += fabs(vreal(cb[3 * k]) - xpos[3 * bd]);
      cy += fabs(vreal(cb[3 * k + 1]) - xpos[3 * bd + 1]);
      cz += fabs(vreal(cb[3 * k + 2]) - xpos[3 * bd + 2]);
    }
    ej = vnl_wave_sum(ej), cx = vnl_wave_sum(cx), cy = vnl_wave_sum(cy), cz = vnl_wave_sum(cz);
    vreal eb = fmax(cx, fmax(cy, cz));
    if (ev.flags & VNL_ENV_TERM_MEAN) {  // humanoid.py:256-260: jp.mean(jp.abs(.)) over the (nb, 3) and (nj,) differences
      eb = (cx + cy + cz) / vreal(3 * ev.nb);
      ej = ej / vreal(nj);
    }
    vreal err = vreal(0.5) * ev.body_err_mult * eb + vreal(0.5) * ej;
    return vreal(1.) - err * ev.inv_term_threshold;
  }

  // carried state: HBM -> LDS
  VNL_HD void load_state() const {
    const vreal* gq = st.qpos + (size_t)e * MI(nq);
    const vreal* gv = st.qvel + (size_t)e * MI(nv);
    const vreal* ga = st.act + (size_t)e * MI(nu);
    VNL_FOR(i, MI(nq)) s[LO(qpos) + i] = gq[i];
    VNL_FOR(i, MI(nv)) s[LO(qvel) + i] = gv[i];
    VNL_FOR(i, MI(nu)) s[LO(act) + i] = ga[i];
    VNL_SYNC();
  }

  // new state + derived quantities: LDS -> HBM; returns true if anything is NaN
  VNL_HD bool store_state() const {
    vreal* gq = st.qpos + (size_t)e * MI(nq);
    vreal* gv = st.qvel + (size_t)e * MI(nv);
    vreal* ga = st.act + (size_t)e * MI(nu);
    vreal* gw = st.warm + (size_t)e * MI(nv);
    const vreal* gf = gqfrc_act();
    const vreal* gx = gxpos();
    bool bad = false;
    VNL_FOR(i, MI(nq)) {
      vreal v = s[LO(qpos) + i];
      gq[i] = v, bad |= v != v;
    }
    VNL_FOR(i, MI(nv)) {
      vreal a = s[LO(qvel) + i], b = s[LO(qacc) + i], c = gf[i];
      gv[i] = a, gw[i] = b;
      bad |= (a != a) | (b != b) | (c != c);
    }
    VNL_FOR(i, MI(nu)) {
      vreal v = s[LO(act) + i];
      ga[i] = v, bad |= v != v;
    }
    VNL_FOR(i, 3 * MI(nbody)) {
      vreal v = gx[i];
      bad |= v != v;
    }
    VNL_FOR(i, 3) {
      vreal v = s[LO(com) + i];
      st.com1[(size_t)e * 3 + i] = v, bad |= v != v;
    }
    return vnl_wave_any(bad);
  }

  // rodent.py:318-344
  VNL_HD void write_obs() const {
    vreal* o = st.obs + (size_t)e * ev.obs_size;
    VNL_FOR(i, MI(nq)) o[i] = nan0(s[LO(qpos) + i]);
    VNL_FOR(i, MI(nv)) o[MI(nq) + i] = nan0(s[LO(qvel) + i]);
    if (ev.flags & VNL_ENV_OBS_QPOS_QVEL) return;  // humanoid.py:354-368
    const vreal* gf = gqfrc_act();
    const vreal* gx = gxpos();
    VNL_FOR(i, MI(nv)) o[MI(nq) + MI(nv) + i] = nan0(gf[i]);
    VNL_FOR(k, 3 * ev.nee) o[MI(nq) + 2 * MI(nv) + k] = nan0(gx[3 * ev.end_eff_idx[k / 3] + k % 3]);
  }

  // rodent.py:346-448; local frame = v @ xmat[1]
  VNL_HD void write_traj(int clip, int frame) const {
    int Lr = ev.ref_len, s0 = clampi(frame + 1, 0, ev.T - Lr), nj = MI(nq) - 7;
    M3 R = qmat(gquat4(1));
    const vreal* gx = gxpos();
    size_t fb = (size_t)clip * ev.T + s0;
    vreal* tr = st.traj + (size_t)e * ev.traj_size;
    int n_app = Lr * ev.napp * 3, n_bod = Lr * ev.nb * 3, n_root = Lr * 3, n_j = Lr * ev.njc;
    VNL_FOR(k, n_app) {  // get_reference_appendages_pos
      int t = k / (ev.napp * 3), a = (k / 3) % ev.napp, i = k % 3;
      tr[k] = ev.body_positions[((fb + t) * ev.nb + ev.app_ref_col[a]) * 3 + i];
    }
    VNL_FOR(k, Lr * ev.nb) {  // bodies: local block then global block
      int t = k / ev.nb, b = k % ev.nb, bd = ev.body_idxs[b];
      const float* cb = ev.body_positions + ((fb + t) * ev.nb + b) * 3;
      V3 v = V3{vreal(cb[0]) - gx[3 * bd], vreal(cb[1]) - gx[3 * bd + 1], vreal(cb[2]) - gx[3 * bd + 2]};
      vreal* lo = tr + n_app + 3 * k;
      lo[0] = v.x * R.a[0] + v.y * R.a[3] + v.z * R.a[6];
      lo[1] = v.x * R.a[1] + v.y * R.a[4] + v.z * R.a[7];
      lo[2] = v.x * R.a[2] + v.y * R.a[5] + v.z * R.a[8];
      vreal* gl = tr + n_app + n_bod + 3 * k;
      gl[0] = v.x, gl[1] = v.y, gl[2] = v.z;
    }
    VNL_FOR(t, Lr) {  // root, local
      const float* cp = ev.position + (fb + t) * 3;
      V3 v = V3{vreal(cp[0]) - s[LO(qpos)], vreal(cp[1]) - s[LO(qpos) + 1], vreal(cp[2]) - s[LO(qpos) + 2]};
      vreal* o = tr + n_app + 2 * n_bod + 3 * t;
      o[0] = v.x * R.a[0] + v.y * R.a[3] + v.z * R.a[6];
      o[1] = v.x * R.a[1] + v.y * R.a[4] + v.z * R.a[7];
      o[2] = v.x * R.a[2] + v.y * R.a[5] + v.z * R.a[8];
    }
    VNL_FOR(k, n_j) {  // joints
      int t = k / ev.njc, col = ev.joint_cols[k % ev.njc];
      tr[n_app + 2 * n_bod + n_root + k] = vreal(ev.joints[(fb + t) * nj + col]) - s[LO(qpos) + 7 + col];
    }
  }

  // RodentTracking.reset, rodent.py:119-176 (start_frame / noise supplied by the caller)
  VNL_HD int* trace_of(int* base, int f) const {
    return base ? base + ((size_t)e * ev.n_frames + f) * VNL_TRACE_INTS : nullptr;
  }
  // Forward kinematics only, of a caller-supplied qpos row per env: smooth.kinematics + the root's subtree centre of mass.
  // The batched FK of clip preprocessing (reference preprocessing/mjx_preprocess.py:85-107 scans mjx.kinematics over the
  // frames of a clip): xpos / xquat / subtree_com1 of the state buffers are the outputs.
  VNL_HD void fk(const vreal* qpos_in) const {
    load_tables();
    const vreal* q = qpos_in + (size_t)e * MI(nq);
    VNL_FOR(k, MI(nq)) s[LO(qpos) + k] = q[k];
    VNL_FOR(d, MI(nv)) s[LO(qvel) + d] = vreal(0.);
    VNL_SYNC();
    fresh().kinematics();
    fresh().body_inertias(true);
    VNL_FOR(i, 3) st.com1[(size_t)e * 3 + i] = s[LO(com) + i];
    VNL_FOR(k, 4) st.qpos[(size_t)e * MI(nq) + 3 + k] = s[LO(qpos) + 3 + k];  // the root quaternion as kinematics normalised it
  }

  VNL_HD void reset(const int* start_frame, const vreal* noise, int* trace_base) const {
    load_tables();
    int clip = st.clip_id[e], sf = start_frame[e];
    int f = clampi(sf, 0, ev.T - 1), nj = MI(nq) - 7;
    size_t fb = (size_t)clip * ev.T + f;
    const vreal* nz = noise + (size_t)e * MI(nq);
    VNL_FOR(k, 3) s[LO(qpos) + k] = vreal(ev.position[fb * 3 + k]) + nz[k];
    VNL_FOR(k, 4) s[LO(qpos) + 3 + k] = vreal(ev.quaternion[fb * 4 + k]) + nz[3 + k];
    VNL_FOR(k, nj) s[LO(qpos) + 7 + k] = vreal(ev.joints[fb * nj + k]) + nz[7 + k];
    VNL_FOR(k, 3) s[LO(qvel) + k] = ev.velocity[fb * 3 + k];
    VNL_FOR(k, 3) s[LO(qvel) + 3 + k] = ev.angular_velocity[fb * 3 + k];
    VNL_FOR(k, nj) s[LO(qvel) + 6 + k] = ev.joints_velocity[fb * nj + k];
    VNL_FOR(i, MI(nu)) s[LO(act) + i] = vreal(0.), s[LO(ctrl) + i] = vreal(0.);
    VNL_FOR(d, MI(nv)) s[LO(qacc) + d] = vreal(0.);
    VNL_SYNC();
    with_trace(trace_of(trace_base, 0)).forward(s + LO(qacc));  // qacc_warmstart = 0 (mjx.make_data)
    VNL_SYNC_GLOBAL();
    store_state();
    write_traj(clip, sf);
    write_obs();
    vreal term = termination(clip, sf, s + LO(qpos), gxpos());
    VNL_SERIAL {
      st.reward[e] = vreal(0.), st.done[e] = vreal(0.);
      for (int k = 0; k < 7; k++) st.metrics[(size_t)e * 7 + k] = vreal(0.);
      st.cur_frame[e] = sf, st.sub_clip_frame[e] = 0;
      st.term_err[e] = term;
    }
  }

  // _calculate_reward (rodent.py:266-316 / humanoid.py:264-311) on a pipeline state given by pointers, against the clip
  // row at `frame`; unscaled terms
  struct RewardTerms {
    vreal rcom, rvel, rquat, ract, rapp, healthy;
  };
  VNL_HD RewardTerms reward_terms(int clip, int frame, const vreal* qpos, const vreal* qvel, const vreal* com, const vreal* qfrc,
                                  const vreal* xpos, const vreal* action) const {
    int fo = clampi(frame, 0, ev.T - 1), nj = MI(nq) - 7;
    size_t fb = (size_t)clip * ev.T + fo;
    const float* cb = ev.body_positions + fb * ev.nb * 3;
    const float* cref = ev.center_of_mass ? ev.center_of_mass + fb * 3 : cb + 3 * ev.com_ref_col;
    V3 dc = V3{com[0] - vreal(cref[0]), com[1] - vreal(cref[1]), com[2] - vreal(cref[2])};
    RewardTerms r;
    r.rcom = exp(vreal(-100.) * sqrt(dot(dc, dc)));
    vreal acc = vreal(0.);
    VNL_FOR(k, MI(nv)) {
      vreal ref = k < 3 ? ev.velocity[fb * 3 + k] : (k < 6 ? ev.angular_velocity[fb * 3 + k - 3] : ev.joints_velocity[fb * nj + k - 6]);
      vreal a = qvel[k] - ref;
      acc += a * a;
    }
    r.rvel = exp(vreal(-0.1) * sqrt(vnl_wave_sum(acc)));
    vreal nc = vreal(0.), nr = vreal(0.), dq = vreal(0.);
    for (int k = 0; k < 4; k++) {
      vreal a = qpos[3 + k], b = ev.quaternion[fb * 4 + k];
      nc += a * a, nr += b * b, dq += a * b;
    }
    dq = dq / (sqrt(nc) * sqrt(nr));
    vreal dist = fmin(vreal(1.), vreal(2.) * dq * dq - vreal(1.));
    r.rquat = exp(vreal(-2.) * fabs(vreal(0.5) * acos(dist)));
    acc = vreal(0.);
    if (ev.flags & VNL_ENV_RACT_ACTION) {  // ant.py:277: 0.01 * -0.015 * sum(action^2) / len(action)
      VNL_FOR(i, MI(nu)) acc += action[i] * action[i];
      r.ract = vreal(0.01) * vreal(-0.015) * vnl_wave_sum(acc) / (vreal)MI(nu);
    } else {
      VNL_FOR(d, MI(nv)) acc += qfrc[d] * qfrc[d];
      r.ract = vreal(-0.015) * (vnl_wave_sum(acc) / (vreal)MI(nv));
    }
    r.rapp = vreal(0.);
    if (!(ev.flags & VNL_ENV_NO_RAPP)) {
      acc = vreal(0.);
      VNL_FOR(k, 3 * ev.napp) {
        int a = k / 3, i = k % 3;
        vreal x = xpos[3 * ev.app_body[a] + i] - vreal(cb[3 * ev.app_ref_col[a] + i]);
        acc += x * x;
      }
      r.rapp = exp(vreal(-400.) * sqrt(vnl_wave_sum(acc)));
    }
    vreal z = qpos[2];
    r.healthy = (z < ev.healthy_lo || z > ev.healthy_hi) ? vreal(0.) : vreal(1.);
    return r;
  }

  // RodentTracking.step, rodent.py:178-239 (HumanoidTracking.step, humanoid.py:185-239, by the env flags)
  // dump_mid / trace_base: debug only (vnl_env_debug), null in normal use
  VNL_HD void step(const vreal* action, vreal* dump_mid, int* trace_base) const {
    prof_begin();
    int clip = st.clip_id[e], old_frame = st.cur_frame[e], old_sub = st.sub_clip_frame[e];
    load_tables();
    load_state();
    // rtrunk from the OLD pipeline state and OLD frame (rodent.py:250-262, 296)
    vreal rtrunk = termination(clip, old_frame, s + LO(qpos), gxpos());
    const vreal* ac = action + (size_t)e * MI(nu);
    VNL_FOR(i, MI(nu)) {
      vreal c = ac[i];
      // jnp.clip semantics: a NaN control stays NaN (and ends the episode below); fmin / fmax would swallow it
      if (m.act_limited[i]) c = c < m.act_lo[i] ? m.act_lo[i] : (c > m.act_hi[i] ? m.act_hi[i] : c);
      s[LO(ctrl) + i] = c;
    }
    VNL_SYNC();
    const vreal* gw = st.warm + (size_t)e * MI(nv);
    if (ev.flags & VNL_ENV_REWARD_OLD_STATE) {  // humanoid.py:195,264-311: every term from the state BEFORE the step
      // (parked in this env's metrics row across the substeps: six values kept in registers over the whole physics would
      // raise the kernel's register allocation past two waves per SIMD)
      const RewardTerms r0 = reward_terms(clip, old_frame, s + LO(qpos), s + LO(qvel), st.com1 + (size_t)e * 3, gqfrc_act(), gxpos(), ac);
      VNL_SERIAL {
        vreal* mt = st.metrics + (size_t)e * 7;
        mt[0] = r0.rcom, mt[1] = r0.rvel, mt[2] = r0.rquat, mt[3] = r0.ract, mt[4] = r0.rapp, mt[5] = r0.healthy;
      }
      VNL_SYNC_GLOBAL();
    }
    VNL_PROF(29);  // tables, state load, rtrunk
    for (int f = 0; f < ev.n_frames; f++) {
      fresh().with_trace(trace_of(trace_base, f)).forward(f == 0 ? gw : s + LO(qacc));
      if (dump_mid && f == ev.n_frames - 1) dump(dump_mid);  // the LDS image as the last forward pass leaves it
      fresh().euler();
    }
    int new_frame = old_frame + 1, new_sub = old_sub + 1;
    VNL_SYNC_GLOBAL();  // (the glue reads xpos / xquat / qfrc_actuator of the last forward pass across lanes)
    RewardTerms rw;
    if (!(ev.flags & VNL_ENV_REWARD_OLD_STATE)) {  // rodent.py:195: _calculate_reward(state, data) -- the NEW data
      rw = reward_terms(clip, old_frame, s + LO(qpos), s + LO(qvel), s + LO(com), gqfrc_act(), gxpos(), ac);
    } else {
      const vreal* mt = st.metrics + (size_t)e * 7;
      rw = RewardTerms{mt[0], mt[1], mt[2], mt[3], mt[4], mt[5]};
      VNL_SYNC_GLOBAL();  // (every lane has read the parked values before lane 0 rewrites the row below)
    }
    const vreal done_trunk = rtrunk < ev.done_threshold ? vreal(1.) : vreal(0.);  // on the unscaled value (rodent.py:213: < 0)
    // weighted terms, then their sum in the reference's order (rodent.py:203-210; ant.py:182-188)
    const vreal wcom = rw.rcom * ev.w_reward[0], wvel = rw.rvel * ev.w_reward[1], wtrunk = rtrunk * ev.w_reward[2];
    const vreal wquat = rw.rquat * ev.w_reward[3], wact = rw.ract * ev.w_reward[4], wapp = rw.rapp * ev.w_reward[5];
    const vreal total = wcom + wvel + wtrunk + wquat + wact + wapp;
    // the metrics hold the weighted terms (rodent.py:228-236), the ant's the raw ones (ant.py:216-225)
    const bool raw = ev.flags & VNL_ENV_METRICS_UNSCALED;
    const vreal rcom = raw ? rw.rcom : wcom, rvel = raw ? rw.rvel : wvel, rquat = raw ? rw.rquat : wquat;
    const vreal ract = raw ? rw.ract : wact, rapp = raw ? rw.rapp : wapp;
    if (!raw) rtrunk = wtrunk;
    vreal done = fmax(vreal(1.) - rw.healthy, done_trunk);
    done = fmax(new_sub < ev.sub_clip_length ? vreal(0.) : vreal(1.), done);
    if (store_state()) done = vreal(1.);
    write_obs();
    write_traj(clip, (ev.flags & VNL_ENV_TRAJ_OLD_FRAME) ? old_frame : new_frame);
    VNL_SERIAL {
      st.reward[e] = nan0(total), st.done[e] = done;
      vreal* mt = st.metrics + (size_t)e * 7;
      mt[0] = rcom, mt[1] = rvel, mt[2] = rtrunk, mt[3] = rquat, mt[4] = ract, mt[5] = rapp, mt[6] = rtrunk;
      st.cur_frame[e] = new_frame, st.sub_clip_frame[e] = new_sub;
      st.term_err[e] = rtrunk;
    }
    VNL_PROF(30);  // reward, termination, obs / traj, state store
    prof_end();
  }

  // bisection hook: copy this env's LDS image to a global dump [env][LO(total)]
  VNL_HD void dump(vreal* out) const {
    VNL_SYNC();
    VNL_FOR(k, LO(total)) out[(size_t)e * LO(total) + k] = s[k];
  }
};
#undef MI
#undef LO
typedef EnvWaveT<VnlSpecGeneric> EnvWave;
