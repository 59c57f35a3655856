// PPO minibatch step, forward AND backward, hand-written for gfx950 (include/vnl.h: vnl_ppo_update_*).
//
// What it replaces: the gradient of compute_ppo_intention_loss (reference ppo_imitation/intention_losses.py:91-202)
// w.r.t. the policy and value parameters, i.e. what `jax.grad` does inside brax's gradient_update_fn for
// ppo_imitation/train.py:255-268 -- the intention network (intention_policy_network.py:20-105: Dense -> ReLU ->
// LayerNorm encoder / decoder, reparameterised latent) and the value MLP (ppo_networks.py:114-118, swish) applied to a
// (T, B) minibatch, the loss head (vnl_ppo_head) and every parameter gradient, written straight into one flat
// gradient buffer laid out like the parameter buffer.
//
// All GEMMs are exact float32 on the matrix cores (v_mfma_f32_32x32x2_f32: same rounding as an fmaf chain), LDS-tiled
// with register-staged double buffering; bias / activation / activation-derivative are fused into the epilogues, the
// weight gradients (reduction over the ~2.6 k samples) are split over K into slabs that a second pass sums in a fixed
// order (deterministic: no float atomics).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <vector>

#include "../../include/vnl.h"
#include "vnl_policy_train.h"

void vnl_set_error_(const char* msg);
int vnl_ppo_head_phase_(const vnl_ppo_head_args* a, float* workspace, void* stream, int phase);  // vnl_lib.hip
static int pfail(int code, const char* msg) {
  vnl_set_error_(msg);
  return code;
}
#define PCHK(call)                                             \
  do {                                                         \
    hipError_t e_ = (call);                                    \
    if (e_ != hipSuccess) return pfail(VNL_ERR_HIP, hipGetErrorString(e_)); \
  } while (0)

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------------------------------------------ GEMM
struct GemmArgs {
  const float* A;    // !TA: [M][lda] (k contiguous)   TA: [K][lda] (m contiguous)
  const float* B;    // !TB: [K][ldb] (n contiguous)   TB: [N][ldb] (k contiguous)
  float* C;          // [M][ldc]; with splits > 1: slab s at C + s * M * ldc
  const float* bias; // [N] or null
  const float* aux;  // EPI_MUL_DSWISH: pre-activations [M][ldaux] whose swish' multiplies the result
  float* zout;       // EPI_SWISH: pre-activations out [M][ldc]
  float* bias_out;   // ones_row (TA only): row M-1 of the result is the column sum of B -> bias gradient [N]
  int M, N, K, lda, ldb, ldc, ldaux, k_chunk, accumulate, vecA, vecB, ones_row, direct;
  int prio;  // s_setprio level of the launch's waves (0 .. 3): the intention network's small, latency-bound GEMMs run beside the
             // value MLP's chip-filling ones and take the matrix cores first
  size_t slab_stride;  // floats between the slabs of a split-K launch (blockIdx.z)
};
enum { EPI_NONE = 0, EPI_RELU = 1, EPI_SWISH = 2, EPI_MUL_DSWISH = 3 };

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + __expf(-x)); }

template <int BM, int BN, bool TA, bool TB, int EPI>
__device__ __forceinline__ void gemm_body(const GemmArgs& g, const int bx, const int by, const int bz) {
  // K depth of a staged slab: 32 for the 64 x 64 tile (1024 MFMA cycles per wave and slab cover the global-load latency of
  // the next one; 35 KB of LDS), 16 for the 128 x 128 tile (2048 cycles, 34 KB)
  constexpr int BK = BM == 64 ? 32 : 16, LDA = BM + 4, LDB = BN + 4;
  constexpr int WTM = BM / 2, WTN = BN / 2, MI = WTM / 32, NJ = WTN / 32;
  constexpr int NA = BM * BK / 4 / 256, NB = BN * BK / 4 / 256;  // float4 per thread per tile
  // LDS image of an operand slab.  An operand whose k index is contiguous in global memory (A of NN / NT, B of NT) keeps
  // that orientation -- [row][BK + 4]: float4 stores straight from the float4 loads (no transposing scalar stores, whose
  // bank conflicts were 33-50 % of these kernels' LDS cycles) and the lane's whole k range of the slab in BK / 8
  // ds_read_b128.  The other orientation ([k][row + 4]) is read with one ds_read per k-pair as before.  Either way the
  // MFMA of step j pairs k = j (lanes 0-31) with k = j + BK / 2 (lanes 32-63): any pairing is valid if A and B agree.
  constexpr bool AK = !TA, BKC = TB;
  constexpr int LDK = BK + 4;
  constexpr int A_FLOATS = AK ? BM * LDK : BK * LDA, B_FLOATS = BKC ? BN * LDK : BK * LDB;
  __shared__ __align__(16) float As[2][A_FLOATS];
  __shared__ __align__(16) float Bs[2][B_FLOATS];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
  if (g.prio == 3) __builtin_amdgcn_s_setprio(3);
  const int m0 = by * BM, n0 = bx * BN;
  const int kbeg = bz * g.k_chunk, kend = min(g.K, kbeg + g.k_chunk);
  float* C = g.C + (size_t)bz * g.slab_stride;

  f32x16 acc[MI][NJ];
#pragma unroll
  for (int i = 0; i < MI; i++)
#pragma unroll
    for (int j = 0; j < NJ; j++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;

  f32x4 ra[NA], rb[NB];
  auto load_tiles = [&](int k0) {
#pragma unroll
    for (int i = 0; i < NA; i++) {
      const int idx = tid + 256 * i;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if constexpr (!TA) {  // rows of A, 4 consecutive k
        const int r = idx / (BK / 4), kq = (idx % (BK / 4)) * 4, m = m0 + r, k = k0 + kq;
        if (m < g.M) {
          const float* p = g.A + (size_t)m * g.lda + k;
          if (g.vecA && k + 3 < kend) v = *(const f32x4*)p;
          else {
            if (k < kend) v.x = p[0];
            if (k + 1 < kend) v.y = p[1];
            if (k + 2 < kend) v.z = p[2];
            if (k + 3 < kend) v.w = p[3];
          }
        }
      } else {  // rows of k, 4 consecutive m
        const int kk = idx / (BM / 4), mq = (idx % (BM / 4)) * 4, m = m0 + mq, k = k0 + kk;
        if (k < kend) {
          // (ones_row: the operand has one more, virtual, column of ones -- [X | 1]' dZ puts the bias gradient, the
          // column sums of dZ, into the last row of the weight gradient for free)
          const int Mr = g.M - g.ones_row;
          const float* p = g.A + (size_t)k * g.lda + m;
          if (g.vecA && m + 3 < Mr) v = *(const f32x4*)p;
          else {
            v.x = m < Mr ? p[0] : (m == Mr && g.ones_row ? 1.f : 0.f);
            v.y = m + 1 < Mr ? p[1] : (m + 1 == Mr && g.ones_row ? 1.f : 0.f);
            v.z = m + 2 < Mr ? p[2] : (m + 2 == Mr && g.ones_row ? 1.f : 0.f);
            v.w = m + 3 < Mr ? p[3] : (m + 3 == Mr && g.ones_row ? 1.f : 0.f);
          }
        }
      }
      ra[i] = v;
    }
#pragma unroll
    for (int i = 0; i < NB; i++) {
      const int idx = tid + 256 * i;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if constexpr (!TB) {  // rows of k, 4 consecutive n
        const int kk = idx / (BN / 4), nq = (idx % (BN / 4)) * 4, n = n0 + nq, k = k0 + kk;
        if (k < kend) {
          const float* p = g.B + (size_t)k * g.ldb + n;
          if (g.vecB && n + 3 < g.N) v = *(const f32x4*)p;
          else {
            if (n < g.N) v.x = p[0];
            if (n + 1 < g.N) v.y = p[1];
            if (n + 2 < g.N) v.z = p[2];
            if (n + 3 < g.N) v.w = p[3];
          }
        }
      } else {  // rows of n, 4 consecutive k
        const int r = idx / (BK / 4), kq = (idx % (BK / 4)) * 4, n = n0 + r, k = k0 + kq;
        if (n < g.N) {
          const float* p = g.B + (size_t)n * g.ldb + k;
          if (g.vecB && k + 3 < kend) v = *(const f32x4*)p;
          else {
            if (k < kend) v.x = p[0];
            if (k + 1 < kend) v.y = p[1];
            if (k + 2 < kend) v.z = p[2];
            if (k + 3 < kend) v.w = p[3];
          }
        }
      }
      rb[i] = v;
    }
  };
  auto store_tiles = [&](int buf) {
#pragma unroll
    for (int i = 0; i < NA; i++) {
      const int idx = tid + 256 * i;
      if constexpr (AK) {
        const int r = idx / (BK / 4), kq = (idx % (BK / 4)) * 4;
        *(f32x4*)&As[buf][r * LDK + kq] = ra[i];
      } else {
        const int kk = idx / (BM / 4), mq = (idx % (BM / 4)) * 4;
        *(f32x4*)&As[buf][kk * LDA + mq] = ra[i];
      }
    }
#pragma unroll
    for (int i = 0; i < NB; i++) {
      const int idx = tid + 256 * i;
      if constexpr (!BKC) {
        const int kk = idx / (BN / 4), nq = (idx % (BN / 4)) * 4;
        *(f32x4*)&Bs[buf][kk * LDB + nq] = rb[i];
      } else {
        const int r = idx / (BK / 4), kq = (idx % (BK / 4)) * 4;
        *(f32x4*)&Bs[buf][r * LDK + kq] = rb[i];
      }
    }
  };

  const int nk = (kend - kbeg + BK - 1) / BK;
  int buf = 0;
  if (nk > 0) {
    load_tiles(kbeg);
    store_tiles(0);
  }
  __syncthreads();
  // Measured and dropped (tools/ppo_update_bench.py, 0.478 ms per step as is): a second register set so that the global
  // loads of slab kt + 2 are issued before slab kt is computed (0.478); requesting every LDS fragment of a slab before
  // its first MFMA instead of the compiler's read / s_waitcnt / MFMA pairs (0.497); the value MLP's big weight gradient
  // on a third stream beside the input-gradient GEMM of the same layer (0.58: two chip-filling GEMMs thrash); slab depth
  // 64 for the 64 x 64 tile (two workgroups per CU: 0.525) and 16 (0.483) against 32 (0.473).
  for (int kt = 0; kt < nk; kt++) {
    if (kt + 1 < nk) load_tiles(kbeg + (kt + 1) * BK);  // global loads in flight under the MFMAs
    const int kh = lane >> 5, c = lane & 31;
    constexpr int H = BK / 2;  // k-steps of a slab; this lane's k range is [kh * H, kh * H + H)
    float av[AK ? MI : 1][AK ? H : 1], bv[BKC ? NJ : 1][BKC ? H : 1];
    if constexpr (AK) {
#pragma unroll
      for (int i = 0; i < MI; i++)
#pragma unroll
        for (int q = 0; q < H / 4; q++) {
          const f32x4 v = *(const f32x4*)&As[buf][(wm * WTM + i * 32 + c) * LDK + kh * H + 4 * q];
          av[i][4 * q] = v.x, av[i][4 * q + 1] = v.y, av[i][4 * q + 2] = v.z, av[i][4 * q + 3] = v.w;
        }
    }
    if constexpr (BKC) {
#pragma unroll
      for (int j = 0; j < NJ; j++)
#pragma unroll
        for (int q = 0; q < H / 4; q++) {
          const f32x4 v = *(const f32x4*)&Bs[buf][(wn * WTN + j * 32 + c) * LDK + kh * H + 4 * q];
          bv[j][4 * q] = v.x, bv[j][4 * q + 1] = v.y, bv[j][4 * q + 2] = v.z, bv[j][4 * q + 3] = v.w;
        }
    }
#pragma unroll
    for (int t = 0; t < H; t++) {
      float a[MI], b[NJ];
#pragma unroll
      for (int i = 0; i < MI; i++) {
        if constexpr (AK) a[i] = av[i][t];
        else a[i] = As[buf][(t + kh * H) * LDA + wm * WTM + i * 32 + c];
      }
#pragma unroll
      for (int j = 0; j < NJ; j++) {
        if constexpr (BKC) b[j] = bv[j][t];
        else b[j] = Bs[buf][(t + kh * H) * LDB + wn * WTN + j * 32 + c];
      }
#pragma unroll
      for (int i = 0; i < MI; i++)
#pragma unroll
        for (int j = 0; j < NJ; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    if (kt + 1 < nk) store_tiles(buf ^ 1);
    __syncthreads();
    buf ^= 1;
  }

  // epilogue: C/D map of the 32x32 tile: column = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
#pragma unroll
  for (int i = 0; i < MI; i++)
#pragma unroll
    for (int j = 0; j < NJ; j++) {
      const int col = n0 + wn * WTN + j * 32 + (lane & 31);
      if (col >= g.N) continue;
      const float bv = g.bias ? g.bias[col] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; r++) {
        const int row = m0 + wm * WTM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (row >= g.M) continue;
        float v = acc[i][j][r] + bv;
        const size_t o = (size_t)row * g.ldc + col;
        if constexpr (EPI == EPI_RELU) v = fmaxf(v, 0.f);
        if constexpr (EPI == EPI_SWISH) {
          g.zout[o] = v;
          v = v * sigmoidf_(v);
        }
        if constexpr (EPI == EPI_MUL_DSWISH) {
          const float z = g.aux[(size_t)row * g.ldaux + col], s = sigmoidf_(z);
          v *= s * (1.f + z * (1.f - s));
        }
        if (g.direct && g.ones_row && row == g.M - 1) {
          g.bias_out[col] = v;
          continue;
        }
        if (g.accumulate) v += C[o];
        C[o] = v;
      }
    }
}

template <int BM, int BN, bool TA, bool TB, int EPI>
__global__ void __launch_bounds__(256) gemm_f32_kernel(GemmArgs g) {
  gemm_body<BM, BN, TA, TB, EPI>(g, blockIdx.x, blockIdx.y, blockIdx.z);
}

// Several independent GEMMs of one flavour as ONE launch (the seven small weight gradients of the intention network:
// 20 us each on their own, latency-bound at 50-100 workgroups; together they fill the chip).  Workgroup b of the launch
// is tile (b - first[j]) of problem j; the tile's (x, y, z) follow the grid of a stand-alone launch.
#define VNL_MAX_GROUP 8
struct GemmGroup {
  int n;
  int first[VNL_MAX_GROUP + 1];  // prefix sums of the problems' workgroup counts
  int gx[VNL_MAX_GROUP], gy[VNL_MAX_GROUP];
  GemmArgs g[VNL_MAX_GROUP];
};
template <int BM, int BN, bool TA, bool TB, int EPI>
__global__ void __launch_bounds__(256) gemm_group_kernel(GemmGroup grp) {
  int j = 0;
  while (j + 1 < grp.n && (int)blockIdx.x >= grp.first[j + 1]) j++;
  const int t = blockIdx.x - grp.first[j], gx = grp.gx[j], gy = grp.gy[j];
  const int bz = t / (gx * gy), r = t - bz * gx * gy;
  gemm_body<BM, BN, TA, TB, EPI>(grp.g[j], r % gx, r / gx, bz);
}

// out[i] = sum_s part[s * stride + i]   (fixed order: deterministic)
__global__ void __launch_bounds__(256) sum_slabs_kernel(float* out, const float* part, int S, size_t stride, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    float a = 0.f;
    for (int s = 0; s < S; s++) a += part[(size_t)s * stride + i];
    out[i] = a;
  }
}

// All the split-K weight gradients of a minibatch step are summed by ONE launch at its end: job j owns the float4 groups
// [start4[j], start4[j+1]) of the launch; slab s of a job is `part + s * rows * cols`; the rows below `wrows` go to the
// weight gradient, the last one (the ones-row of the operand) to the bias gradient.
#define VNL_MAX_JOBS 28
struct ReduceJobs {
  int njobs, prio;
  unsigned start4[VNL_MAX_JOBS + 1];
  float* out[VNL_MAX_JOBS];
  float* bias_out[VNL_MAX_JOBS];
  const float* part[VNL_MAX_JOBS];
  int S[VNL_MAX_JOBS];
  unsigned wn[VNL_MAX_JOBS];    // weight elements (rows * cols without the ones row)
  unsigned tot[VNL_MAX_JOBS];   // floats between slabs (multiple of 4)
  unsigned real[VNL_MAX_JOBS];  // elements of one slab that exist (rows * cols with the ones row)
};
__global__ void __launch_bounds__(256) reduce_jobs_kernel(ReduceJobs J) {
  if (J.prio == 3) __builtin_amdgcn_s_setprio(3);
  const unsigned total4 = J.start4[J.njobs];
  for (unsigned i4 = blockIdx.x * 256 + threadIdx.x; i4 < total4; i4 += gridDim.x * 256) {
    int j = 0;
    while (j + 1 < J.njobs && i4 >= J.start4[j + 1]) j++;
    const unsigned e = (i4 - J.start4[j]) * 4, tot = J.tot[j];
    const float* p = J.part[j] + e;
    f32x4 a = *(const f32x4*)p;
    // eight slabs' loads in flight at once, summed in slab order (a load per trip made every slab a memory round trip)
    const int S = J.S[j];
    for (int s0 = 1; s0 < S; s0 += 8) {
      f32x4 b[8];
#pragma unroll
      for (int k = 0; k < 8; k++) b[k] = *(const f32x4*)(p + (size_t)(s0 + k < S ? s0 + k : 0) * tot);
#pragma unroll
      for (int k = 0; k < 8; k++)
        if (s0 + k < S) a.x += b[k].x, a.y += b[k].y, a.z += b[k].z, a.w += b[k].w;
    }
    const float v[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const unsigned idx = e + k;
      if (idx < J.wn[j]) J.out[j][idx] = v[k];
      else if (J.bias_out[j] && idx < J.real[j]) J.bias_out[j][idx - J.wn[j]] = v[k];
    }
  }
}

template <int BM, int BN>
static void launch_gemm_cfg(hipStream_t st, bool TA, bool TB, int epi, const GemmArgs& g, int splits) {
  dim3 grid((g.N + BN - 1) / BN, (g.M + BM - 1) / BM, splits), block(256);
#define LG(ta, tb, e) hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, ta, tb, e>), grid, block, 0, st, g)
  if (!TA && !TB) {
    if (epi == EPI_NONE) LG(false, false, EPI_NONE);
    else if (epi == EPI_RELU) LG(false, false, EPI_RELU);
    else LG(false, false, EPI_SWISH);
  } else if (!TA && TB) {
    if (epi == EPI_MUL_DSWISH) LG(false, true, EPI_MUL_DSWISH);
    else LG(false, true, EPI_NONE);
  } else {
    LG(true, false, EPI_NONE);
  }
#undef LG
}

struct SlabPool {  // host-side bump allocator over the slab buffer + the list of deferred reductions of one step
  float* base;
  size_t cap, used;
  ReduceJobs jobs;
  int tile;  // 0: by shape, 64 / 128: forced (tuning knob of tools/ppo_update_bench.py)
  int wg_target = 256;
  // large minibatches (rows > 4096): the weight gradients take [X | 1]' from this arena (transpose_ones_kernel) instead of
  // reading X transposed on the fly; `done` remembers what a step has transposed already (the two latent heads share an input)
  float* tarena = nullptr;
  size_t tcap = 0, tused = 0;
  int ldt = 0, ndone = 0;
  const float* done_src[16];
  float* done_xt[16];
};

__global__ void __launch_bounds__(256) transpose_ones_kernel(const float* X, int ldx, int rows, int cols, float* XT, int ldt);

static void launch_wgrad_group(hipStream_t st, GemmGroup& grp, bool nn = false) {
  if (grp.n == 0) return;
  if (nn) hipLaunchKernelGGL((gemm_group_kernel<64, 64, false, false, EPI_NONE>), dim3(grp.first[grp.n]), dim3(256), 0, st, grp);
  else hipLaunchKernelGGL((gemm_group_kernel<64, 64, true, false, EPI_NONE>), dim3(grp.first[grp.n]), dim3(256), 0, st, grp);
  grp.n = 0;
}

struct Gemm {  // C[M][N] (+)= op(A) op(B) with fused epilogue on stream `st`
  hipStream_t st;
  SlabPool* pool;
  GemmGroup* group = nullptr;  // non-null: 64 x 64 weight gradients are collected here and launched together by flush()
  int prio = 0;
  GemmGroup* group_nn = nullptr;  // .. and the ones that take a transposed input from the pool's arena
  void flush() {
    if (group) launch_wgrad_group(st, *group);
    if (group_nn) launch_wgrad_group(st, *group_nn, true);
  }
  void run(bool TA, bool TB, int epi, const float* A, int lda, const float* B, int ldb, float* C, int ldc, int M, int N, int K,
           const float* bias = nullptr, const float* aux = nullptr, int ldaux = 0, float* zout = nullptr, int accumulate = 0) {
    GemmArgs g{A, B, C, bias, aux, zout, nullptr, M, N, K, lda, ldb, ldc, ldaux, K, accumulate, 0, 0, 0, 1, prio, 0};
    g.vecA = (lda % 4 == 0) && (((uintptr_t)A) % 16 == 0);
    g.vecB = (ldb % 4 == 0) && (((uintptr_t)B) % 16 == 0);
    const long t128 = (long)((M + 127) / 128) * ((N + 127) / 128);
    (void)t128;
    const bool big = pool->tile == 128;  // 64 x 64 tiles measured faster at these sizes (0.69 vs 0.73 ms per step)
    if (big) launch_gemm_cfg<128, 128>(st, TA, TB, epi, g, 1);
    else launch_gemm_cfg<64, 64>(st, TA, TB, epi, g, 1);
  }
  // weight + bias gradient of a Dense layer: [dW; db] = [X | 1]' dZ, reduction over the `rows` samples split into slabs
  // that reduce_jobs_kernel sums at the end of the step (or written directly when one slab suffices)
  void wgrad(const float* X, int ldx, const float* dZ, int lddz, float* dW, float* db, int in, int out, int rows) {
    const int M = in + 1, N = out, K = rows;
    GemmArgs g{X, dZ, dW, nullptr, nullptr, nullptr, db, M, N, K, ldx, lddz, N, 0, K, 0, 0, 0, 1, 1, prio, 0};
    // large minibatch: [X | 1]' materialised once (k-contiguous first operand), then the plain product
    bool tr = false;
    // (wide layers only: measured at 20,480 rows, the 1024 x 1024 layer gains 25 % this way -- 2.82 against 2.98 ms per step --
    // while the intention network's narrow ones lose: all layers transposed 2.88 ms ..
    // .. and from 8,192 rows on: 5,120 rows 0.88 against 0.85 ms, 10,240 rows 1.53 / 1.54, 20,480 rows 2.81 / 2.97)
    if (pool->tarena && rows > 8192 && in >= 512 && out >= 512 && (size_t)M * pool->ldt <= pool->tcap - pool->tused) {
      float* xt = nullptr;
      for (int k = 0; k < pool->ndone; k++)
        if (pool->done_src[k] == X) xt = pool->done_xt[k];
      if (!xt && pool->ndone < 16) {
        xt = pool->tarena + pool->tused;
        pool->tused += (size_t)M * pool->ldt;
        hipLaunchKernelGGL(transpose_ones_kernel, dim3((rows + 31) / 32, (in + 31) / 32), dim3(256), 0, st, X, ldx, rows, in, xt, pool->ldt);
        pool->done_src[pool->ndone] = X, pool->done_xt[pool->ndone] = xt, pool->ndone++;
      }
      if (xt) g.A = xt, g.lda = pool->ldt, g.ones_row = 1, tr = true;
    }
    g.vecA = tr ? true : (ldx % 4 == 0) && (((uintptr_t)X) % 16 == 0);
    g.vecB = (lddz % 4 == 0) && (((uintptr_t)dZ) % 16 == 0);
    const long t128 = (long)((M + 127) / 128) * ((N + 127) / 128), t64 = (long)((M + 63) / 64) * ((N + 63) / 64);
    const int max_splits = K / 256 > 16 ? 16 : (K / 256 < 1 ? 1 : K / 256);  // slabs at least 256 deep
    // the one big weight gradient: 128 x 128 tiles x 4 slabs at the reference's 2,560 rows; with many more rows (the reference's
    // own proportions: 20,480) the 64 x 64 tile is the faster one again (3.01 vs 3.08 ms per minibatch step)
    const bool big = pool->tile ? pool->tile == 128 : (in >= 512 && out >= 512 && rows <= 4096);
    const long tiles = big ? t128 : t64;
    int splits = 1;
    while (splits * 2 <= max_splits && tiles * splits < pool->wg_target) splits *= 2;
    const size_t tot = ((size_t)M * N + 3) & ~(size_t)3;
    if (splits > 1 && (pool->used + splits * tot > pool->cap || pool->jobs.njobs >= VNL_MAX_JOBS)) splits = 1;
    if (splits > 1) {
      ReduceJobs& J = pool->jobs;
      const int j = J.njobs++;
      J.out[j] = dW, J.bias_out[j] = db, J.part[j] = pool->base + pool->used, J.S[j] = splits;
      J.wn[j] = (unsigned)((size_t)in * N), J.tot[j] = (unsigned)tot, J.real[j] = (unsigned)((size_t)M * N);
      J.start4[j + 1] = J.start4[j] + (unsigned)(tot / 4);
      g.C = pool->base + pool->used, g.direct = 0, g.slab_stride = tot;
      g.k_chunk = ((K + splits - 1) / splits + 31) / 32 * 32;
      pool->used += splits * tot;
    }
    GemmGroup* grp = tr ? group_nn : group;
    if (big && !tr) {
      launch_gemm_cfg<128, 128>(st, true, false, EPI_NONE, g, splits);
    } else if (grp) {
      if (grp->n == VNL_MAX_GROUP) flush();
      const int j = grp->n++;
      if (j == 0) grp->first[0] = 0;
      grp->gx[j] = (N + 63) / 64, grp->gy[j] = (M + 63) / 64;
      grp->first[j + 1] = grp->first[j] + grp->gx[j] * grp->gy[j] * splits;
      grp->g[j] = g;
    } else {
      launch_gemm_cfg<64, 64>(st, !tr, false, EPI_NONE, g, splits);
    }
  }
};

// XT[c][r] = X[r][c] (c < cols, r < rows), XT[cols][r] = 1: a layer's input with the SAMPLE index contiguous (+ the row of ones
// that carries the bias gradient), so that the layer's weight gradient [X | 1]' dZ runs as a product whose first operand is
// k-contiguous -- the form with vector LDS traffic (74-78 TFLOP/s at 20,480 rows against 53 for the transposed-operand form)
__global__ void __launch_bounds__(256) transpose_ones_kernel(const float* X, int ldx, int rows, int cols, float* XT, int ldt) {
  __shared__ float tile[32][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  const int r0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const int r = r0 + ty + 8 * k, c = c0 + tx;
    tile[ty + 8 * k][tx] = (r < rows && c < cols) ? X[(size_t)r * ldx + c] : 0.f;
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const int c = c0 + ty + 8 * k, r = r0 + tx;
    if (c < cols && r < rows) XT[(size_t)c * ldt + r] = tile[tx][ty + 8 * k];
  }
  if (blockIdx.y == 0 && ty == 0 && r0 + tx < rows) XT[(size_t)cols * ldt + r0 + tx] = 1.f;
}

// ------------------------------------------------------------------------------------------------ small kernels
// observation normalisation (running_statistics.normalize) for the T*B rows + the B bootstrap rows, and the trajectory
// copied to a row stride that is a multiple of 4 floats (vector loads in the GEMM)
__global__ void __launch_bounds__(256) prep_kernel(const float* obs, const float* next_last, const float* mean, const float* stdv,
                                                   const float* traj, float* obsn, float* trajp, int N, int Nv, int no, int nt,
                                                   int ntp) {
  const size_t total_o = (size_t)Nv * no, total_t = (size_t)N * ntp;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total_o + total_t; i += (size_t)gridDim.x * 256) {
    if (i < total_o) {
      const int r = (int)(i / no), c = (int)(i % no);
      float v = r < N ? obs[(size_t)r * no + c] : next_last[(size_t)(r - N) * no + c];
      if (mean) v = (v - mean[c]) / stdv[c];
      obsn[i] = v;
    } else {
      const size_t k = i - total_o;
      const int r = (int)(k / ntp), c = (int)(k % ntp);
      trajp[k] = c < nt ? traj[(size_t)r * nt + c] : 0.f;
    }
  }
}

// LayerNorm forward (flax eps 1e-6) over rows of H (already ReLU'd): one wave per row, the row held in registers
// (h <= 64 PER; one pass over memory, every load of a wave in flight at once)
template <int PER>
__global__ void __launch_bounds__(256) ln_fwd_kernel(const float* H, const float* gamma, const float* beta, float* Y, float* stats,
                                                     int rows, int h) {
  __builtin_amdgcn_s_setprio(3);  // (the intention network's chain of small launches runs beside chip-filling GEMMs)
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  float gm[PER], bt[PER];
#pragma unroll
  for (int k = 0; k < PER; k++) {
    const int c = lane + 64 * k;
    gm[k] = c < h ? gamma[c] : 0.f, bt[k] = c < h ? beta[c] : 0.f;
  }
  for (int r = blockIdx.x * 4 + w; r < rows; r += gridDim.x * 4) {
    const float* x = H + (size_t)r * h;
    float xv[PER], s = 0.f;
#pragma unroll
    for (int k = 0; k < PER; k++) {
      const int c = lane + 64 * k;
      xv[k] = c < h ? x[c] : 0.f;
      s += xv[k];
    }
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    const float mu = s / (float)h;
    float q = 0.f;
#pragma unroll
    for (int k = 0; k < PER; k++) {
      const float d = lane + 64 * k < h ? xv[k] - mu : 0.f;
      q += d * d;
    }
    for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o);
    const float rstd = rsqrtf(q / (float)h + 1e-6f);
#pragma unroll
    for (int k = 0; k < PER; k++) {
      const int c = lane + 64 * k;
      if (c < h) Y[(size_t)r * h + c] = (xv[k] - mu) * rstd * gm[k] + bt[k];
    }
    if (lane == 0) stats[2 * r] = mu, stats[2 * r + 1] = rstd;
  }
}
static void ln_fwd(hipStream_t st, const float* H, const float* gamma, const float* beta, float* Y, float* stats, int rows, int h) {
  const int blocks = (rows + 3) / 4;  // one row per wave
  if (h <= 128) hipLaunchKernelGGL((ln_fwd_kernel<2>), dim3(blocks), dim3(256), 0, st, H, gamma, beta, Y, stats, rows, h);
  else if (h <= 256) hipLaunchKernelGGL((ln_fwd_kernel<4>), dim3(blocks), dim3(256), 0, st, H, gamma, beta, Y, stats, rows, h);
  else hipLaunchKernelGGL((ln_fwd_kernel<16>), dim3(blocks), dim3(256), 0, st, H, gamma, beta, Y, stats, rows, h);
}

// LayerNorm + ReLU backward: dZ = relu'(H) * LN'(dY); per-block partial sums of d gamma, d beta -> part[blk][2][h]
// (NW waves per block: sixteen for h <= 256 -- 80 blocks x 16 waves leave two rows per wave, so the rows' load -> reduce ->
// store chains run side by side instead of ten deep)
template <int HMAX, int NW>
__global__ void __launch_bounds__(64 * NW) ln_bwd_kernel(const float* dY, const float* H, const float* stats, const float* gamma,
                                                         float* dZ, float* part, int rows, int h, int part_stride) {
  __builtin_amdgcn_s_setprio(3);
  constexpr int PER = HMAX / 64;
  __shared__ float red[NW][2][HMAX];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  float dg[PER], db[PER], gm[PER];
#pragma unroll
  for (int k = 0; k < PER; k++) {
    dg[k] = 0.f, db[k] = 0.f;
    const int c = lane + 64 * k;
    gm[k] = c < h ? gamma[c] : 0.f;
  }
  for (int r = blockIdx.x * NW + w; r < rows; r += gridDim.x * NW) {
    const float mu = stats[2 * r], rstd = stats[2 * r + 1];
    float xh[PER], gy[PER], hv[PER], s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int k = 0; k < PER; k++) {
      const int c = lane + 64 * k;
      const bool ok = c < h;
      hv[k] = ok ? H[(size_t)r * h + c] : 0.f;
      const float d = ok ? dY[(size_t)r * h + c] : 0.f;
      xh[k] = ok ? (hv[k] - mu) * rstd : 0.f;
      gy[k] = d * gm[k];
      s1 += gy[k], s2 += gy[k] * xh[k];
      dg[k] += d * xh[k], db[k] += d;
    }
    for (int o = 32; o > 0; o >>= 1) s1 += __shfl_xor(s1, o), s2 += __shfl_xor(s2, o);
    const float m1 = s1 / (float)h, m2 = s2 / (float)h;
#pragma unroll
    for (int k = 0; k < PER; k++) {
      const int c = lane + 64 * k;
      if (c < h) dZ[(size_t)r * h + c] = hv[k] > 0.f ? rstd * (gy[k] - m1 - xh[k] * m2) : 0.f;
    }
  }
#pragma unroll
  for (int k = 0; k < PER; k++) red[w][0][lane + 64 * k] = dg[k], red[w][1][lane + 64 * k] = db[k];
  __syncthreads();
  for (int c = threadIdx.x; c < 2 * h; c += 64 * NW) {
    const int which = c / h, cc = c % h;
    float a = 0.f;
#pragma unroll
    for (int q = 0; q < NW; q++) a += red[q][which][cc];
    part[(size_t)blockIdx.x * part_stride + (size_t)which * h + cc] = a;
  }
}

// part[rb][c] = sum over the rows of chunk rb of w[r] * X[r][c]  (w null: plain column sums)
__global__ void __launch_bounds__(256) colsum_kernel(const float* X, const float* w, float* part, int rows, int cols, int ldx,
                                                     int rows_per_chunk) {
  const int c = blockIdx.x * 256 + threadIdx.x, r0 = blockIdx.y * rows_per_chunk, r1 = min(rows, r0 + rows_per_chunk);
  if (c >= cols) return;
  float a = 0.f;
  for (int r = r0; r < r1; r++) a += (w ? w[r] : 1.f) * X[(size_t)r * ldx + c];
  part[(size_t)blockIdx.y * cols + c] = a;
}

// v[r] = A[r][:] . w + b : one wave per row
__global__ void __launch_bounds__(256) rowdot_kernel(const float* A, const float* w, const float* b, float* v, int rows, int h) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const bool vec = (h & 3) == 0 && (((uintptr_t)A | (uintptr_t)w) & 15) == 0;
  for (int r = blockIdx.x * 4 + wv; r < rows; r += gridDim.x * 4) {  // launched with one row per wave
    float s = 0.f;
    if (vec) {
      const f32x4* a4 = (const f32x4*)(A + (size_t)r * h);
      const f32x4* w4 = (const f32x4*)w;
      for (int c = lane; c < h / 4; c += 64) {
        const f32x4 x = a4[c], y = w4[c];
        s += x.x * y.x + x.y * y.y + x.z * y.z + x.w * y.w;
      }
    } else {
      for (int c = lane; c < h; c += 64) s += A[(size_t)r * h + c] * w[c];
    }
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) v[r] = s + b[0];
  }
}

// dZ[r][c] = gv[r] * w[c] * swish'(Z[r][c])
__global__ void __launch_bounds__(256) value_seed_kernel(const float* gv, const float* w, const float* Z, float* dZ, int rows, int h) {
  const size_t n = (size_t)rows * h;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const int r = (int)(i / h), c = (int)(i % h);
    const float z = Z[i], s = sigmoidf_(z);
    dZ[i] = gv[r] * w[c] * s * (1.f + z * (1.f - s));
  }
}

// z = mean + eps exp(0.5 logvar); D0 = [z | normalised obs]
__global__ void __launch_bounds__(256) reparam_kernel(const float* ml /* [N][2 lat]: mean | logvar */, const float* eps, const float* obsn,
                                                      float* D0, int N, int lat, int no) {
  __builtin_amdgcn_s_setprio(3);
  const int w = lat + no;
  const size_t n = (size_t)N * w;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const int r = (int)(i / w), c = (int)(i % w);
    D0[i] = c < lat ? ml[(size_t)r * 2 * lat + c] + eps[(size_t)r * lat + c] * __expf(0.5f * ml[(size_t)r * 2 * lat + lat + c])
                    : obsn[(size_t)r * no + (c - lat)];
  }
}

// d[mean | logvar] = through z (dD0[:, :lat]) + the KL term's own gradient
__global__ void __launch_bounds__(256) latent_bwd_kernel(const float* dD0, int ldd, const float* ml, const float* eps, const float* gklm,
                                                         const float* gkll, float* dml, int N, int lat) {
  __builtin_amdgcn_s_setprio(3);
  const size_t n = (size_t)N * lat;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const int r = (int)(i / lat), c = (int)(i % lat);
    const float dz = dD0[(size_t)r * ldd + c], lv = ml[(size_t)r * 2 * lat + lat + c];
    dml[(size_t)r * 2 * lat + c] = dz + gklm[i];
    dml[(size_t)r * 2 * lat + lat + c] = dz * eps[i] * 0.5f * __expf(0.5f * lv) + gkll[i];
  }
}

// the two heads' outputs side by side [N][2 lat] -> contiguous mean / logvar arrays for the loss head
// metrics["prediction_corr"] (intention_losses.py:186-188): the mean of jnp.corrcoef over the 2T rows [vs ; reward *
// scaling], each a variable with B observations.  One workgroup: centred rows in LDS, then the (2T)^2 normalised dot
// products, clamped to [-1, 1] as jnp.corrcoef does; a constant row gives NaN, as there.
#define VNL_CORR_THREADS 1024 /* one workgroup; sixteen waves so that the LDS latency of the pair loop is hidden */
__global__ void __launch_bounds__(VNL_CORR_THREADS) prediction_corr_kernel(const float* vs, const float* reward, float scale, int T, int B,
                                                              float* out) {
  extern __shared__ float xs[];  // [2T][B] centred, then [2T] norms
  __shared__ float red[VNL_CORR_THREADS];
  const int R = 2 * T, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  float* nrm = xs + (size_t)R * B;
  // rows into LDS first, every load of the block in flight at once (ten rows per wave, one after the other, each with two
  // passes over global memory, was 30 of this kernel's 36 us); then the means / norms from LDS
  for (int e = tid; e < R * B; e += VNL_CORR_THREADS) {
    const int r = e / B, c = e - r * B;
    xs[e] = r < T ? vs[(size_t)r * B + c] : reward[(size_t)(r - T) * B + c] * scale;
  }
  __syncthreads();
  for (int r = w; r < R; r += VNL_CORR_THREADS / 64) {
    float* x = xs + (size_t)r * B;
    float s1 = 0.f;
    for (int c = lane; c < B; c += 64) s1 += x[c];
    for (int o = 32; o > 0; o >>= 1) s1 += __shfl_xor(s1, o);
    const float mu = s1 / (float)B;
    float q = 0.f;
    for (int c = lane; c < B; c += 64) {
      const float d = x[c] - mu;
      x[c] = d;
      q += d * d;
    }
    for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o);
    if (lane == 0) nrm[r] = sqrtf(q);
  }
  __syncthreads();
  float acc = 0.f;
  for (int pq = tid; pq < R * R; pq += VNL_CORR_THREADS) {
    const int i = pq / R, j = pq - i * R;
    // (every lane starts at a column of its own: rows are B floats apart, so equal columns would share one LDS bank)
    // four independent partial sums, the column computed from k (no loop-carried address): the loop was one LDS latency
    // per step (35 us for T = 20, B = 128) while its chain ran through `d` and the wrapped column
    const int col0 = tid % B;
    const float* xi = xs + (size_t)i * B;
    const float* xj = xs + (size_t)j * B;
    float d4[4] = {0.f, 0.f, 0.f, 0.f};
    int k = 0;
    for (; k + 4 <= B; k += 4) {
#pragma unroll
      for (int u = 0; u < 4; u++) {
        int col = col0 + k + u;
        col = col >= B ? col - B : col;
        d4[u] += xi[col] * xj[col];
      }
    }
    for (; k < B; k++) {
      int col = col0 + k;
      col = col >= B ? col - B : col;
      d4[0] += xi[col] * xj[col];
    }
    const float d = (d4[0] + d4[1]) + (d4[2] + d4[3]);
    const float c = d / (nrm[i] * nrm[j]);
    acc += c != c ? c : fminf(fmaxf(c, -1.f), 1.f);
  }
  red[tid] = acc;
  __syncthreads();
  for (int k = VNL_CORR_THREADS / 2; k > 0; k >>= 1) {
    if (tid < k) red[tid] += red[tid + k];
    __syncthreads();
  }
  if (tid == 0) *out = red[0] / (float)(R * R);
}

__global__ void __launch_bounds__(256) split_ml_kernel(const float* ml, float* mean, float* logvar, int N, int lat) {
  __builtin_amdgcn_s_setprio(3);
  const size_t n = (size_t)N * lat;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const int r = (int)(i / lat), c = (int)(i % lat);
    mean[i] = ml[(size_t)r * 2 * lat + c], logvar[i] = ml[(size_t)r * 2 * lat + lat + c];
  }
}

// ------------------------------------------------------------------------------------------------ handle
struct DenseP {
  int in, out;
  size_t w, b, g, be;  // offsets in the flat buffer: kernel, bias, LayerNorm scale / bias (g == 0 && be == 0: none)
  bool ln;
};

struct vnl_ppo_update {
  int device = 0, T = 0, B = 0, N = 0, Nv = 0;
  vnl_ppo_net_spec spec{};
  std::vector<DenseP> enc, dec, val;
  size_t mean_w = 0, mean_b = 0, lv_w = 0, lv_b = 0, n_policy = 0, n_total = 0;
  int ntp = 0, wmax = 0;
  std::vector<void*> allocs;
  // workspace (device)
  float *obsn = nullptr, *trajp = nullptr, *D0 = nullptr, *ml = nullptr, *mean = nullptr, *logvar = nullptr, *logits = nullptr;
  float *v = nullptr, *gl = nullptr, *gb = nullptr, *gklm = nullptr, *gkll = nullptr, *vs = nullptr, *adv = nullptr, *headws = nullptr;
  float *dml = nullptr, *dA = nullptr, *dB = nullptr, *dPa = nullptr, *dPb = nullptr, *slabs = nullptr, *part = nullptr;
  float* dzarena = nullptr;  // d loss / d (pre-activations) of every LayerNorm layer: each keeps its own buffer until the
                             // step's grouped weight-gradient launch has read it
  size_t dz_floats = 0;
  hipStream_t s2 = nullptr;
  hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
  int tile = 0, wg_target = 512;  // (wg_target: workgroups a split-K weight gradient is split up to; 512 measured best)
  size_t slab_floats = 0, slab_floats_p = 0, part_floats = 0;  // (slab_floats_p: the intention network's share, at the end)
  std::vector<float*> encH, encY, encS, decH, decY, decS, valZ, valA;
  float* tarena = nullptr;  // transposed layer inputs of a large minibatch's weight gradients (SlabPool::tarena): value part | policy part
  size_t tarena_v = 0, tarena_p = 0;
  int ldt = 0;
  bool wgrad_t = true;      // tuning knob (tile = -6 switches it off)
  vnl_policy* fused = nullptr;  // the intention network's forward pass as ONE launch (csrc/vnl_policy.hip in its training form);
                                // null: the network is outside that kernel's limits -> layer by layer
  int fwd_mode = 2;  // the intention network's forward: 0 layer by layer, 1 ONE fused launch, 2 first Dense as a GEMM + the rest fused
                     // (tools/ppo_update_bench.py --fwd-mode; measured per minibatch step: see DESIGN section 5)
  bool fwd_mode_forced = false;  // (tools/ppo_update_bench.py --fwd-mode: also beyond the row count the default route is used up to)
  int prio = 3;                 // wave priority of the intention network's GEMMs (tuning knob: tile = -2 switches it off)
};

static int dalloc(vnl_ppo_update* u, float** p, size_t n) {
  void* q = nullptr;
  PCHK(hipMalloc(&q, (n ? n : 1) * sizeof(float)));
  u->allocs.push_back(q);
  *p = (float*)q;
  return VNL_OK;
}

extern "C" void vnl_ppo_update_destroy(vnl_ppo_update* u) {
  if (!u) return;
  for (void* p : u->allocs) (void)hipFree(p);
  if (u->s2) (void)hipStreamDestroy(u->s2);
  if (u->fused) vnl_policy_destroy(u->fused);
  for (hipEvent_t e : u->ev)
    if (e) (void)hipEventDestroy(e);
  delete u;
}

// tuning knob of tools/ppo_update_bench.py (not part of include/vnl.h): force the GEMM tile (64 / 128; 0 = by shape)
extern "C" int vnl_ppo_update_tune(vnl_ppo_update* u, int tile, int wg_target) {
  if (u && tile <= -10 && tile >= -12) {  // the intention network's forward: -10 layer by layer, -11 one fused launch, -12 GEMM + fused
    u->fwd_mode = -10 - tile, u->fwd_mode_forced = true;
    return VNL_OK;
  }
  if (u && (tile == -256 || tile == -512 || tile == -1024)) {  // threads of the fused part of the intention network's forward
    return u->fused ? vnl_policy_set_threads2_(u->fused, -tile) : VNL_OK;
  }
  if (u && tile == -6) {
    u->wgrad_t = false;
    return VNL_OK;
  }
  if (u && tile == -2) {
    u->prio = 0;
    return VNL_OK;
  }
  if (!u || (tile != 0 && tile != 64 && tile != 128)) return pfail(VNL_ERR_ARG, "vnl_ppo_update_tune: tile must be 0, 64 or 128");
  u->tile = tile;
  if (wg_target > 0) u->wg_target = wg_target;
  return VNL_OK;
}

extern "C" int64_t vnl_ppo_update_num_params(const vnl_ppo_update* u) { return u ? (int64_t)u->n_total : 0; }

extern "C" int vnl_ppo_update_buffer(const vnl_ppo_update* u, const char* name, float** dev_ptr, int64_t* count) {
  if (!u || !name || !dev_ptr || !count) return pfail(VNL_ERR_ARG, "vnl_ppo_update_buffer: null argument");
  const size_t N = u->N;
  struct {
    const char* n;
    float* p;
    size_t c;
  } tab[] = {{"vs", u->vs, N}, {"advantages", u->adv, N}, {"values", u->v, (size_t)u->Nv},
             {"logits", u->logits, N * 2 * u->spec.action_size}, {"latent_mean", u->mean, N * u->spec.latent_size},
             {"latent_logvar", u->logvar, N * u->spec.latent_size}};
  for (auto& t : tab)
    if (strcmp(name, t.n) == 0) {
      *dev_ptr = t.p, *count = (int64_t)t.c;
      return VNL_OK;
    }
  return pfail(VNL_ERR_ARG, "vnl_ppo_update_buffer: unknown buffer");
}

extern "C" int vnl_ppo_update_create(const vnl_ppo_net_spec* sp, int32_t T, int32_t B, int32_t device, vnl_ppo_update** out) {
  if (!sp || !out || T <= 0 || B <= 0) return pfail(VNL_ERR_ARG, "vnl_ppo_update_create: bad argument");
  if (sp->num_encoder_layers < 1 || sp->num_encoder_layers > 8 || sp->num_decoder_layers < 1 || sp->num_decoder_layers > 8 ||
      sp->num_value_layers < 1 || sp->num_value_layers > 8)
    return pfail(VNL_ERR_ARG, "vnl_ppo_update_create: layer counts must be 1..8");
  if (sp->decoder_layers[sp->num_decoder_layers - 1] != 2 * sp->action_size)
    return pfail(VNL_ERR_ARG, "vnl_ppo_update_create: the last decoder layer must have 2 * action_size outputs");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return pfail(VNL_ERR_NO_DEVICE, "no HIP device: no CPU fallback");
  if (device < 0 || device >= ndev) return pfail(VNL_ERR_ARG, "device ordinal out of range");
  int prev = 0;
  PCHK(hipGetDevice(&prev));
  PCHK(hipSetDevice(device));
  vnl_ppo_update* u = new vnl_ppo_update();
  u->device = device, u->T = T, u->B = B, u->N = T * B, u->Nv = T * B + B, u->spec = *sp;
  // parameter layout: exactly ppo_imitation/intention_policy_network.py (policy) then ppo_networks.ValueMLP (value)
  size_t o = 0;
  int fan = sp->traj_size, wmax = 2 * sp->latent_size;
  for (int i = 0; i < sp->num_encoder_layers; i++) {
    DenseP d{fan, sp->encoder_layers[i], 0, 0, 0, 0, true};
    d.w = o, o += (size_t)d.in * d.out, d.b = o, o += d.out, d.g = o, o += d.out, d.be = o, o += d.out;
    u->enc.push_back(d), fan = d.out, wmax = d.out > wmax ? d.out : wmax;
  }
  u->mean_w = o, o += (size_t)fan * sp->latent_size, u->mean_b = o, o += sp->latent_size;
  u->lv_w = o, o += (size_t)fan * sp->latent_size, u->lv_b = o, o += sp->latent_size;
  fan = sp->latent_size + sp->obs_size;
  wmax = fan > wmax ? fan : wmax;
  for (int i = 0; i < sp->num_decoder_layers; i++) {
    DenseP d{fan, sp->decoder_layers[i], 0, 0, 0, 0, i != sp->num_decoder_layers - 1};
    d.w = o, o += (size_t)d.in * d.out, d.b = o, o += d.out;
    if (d.ln) d.g = o, o += d.out, d.be = o, o += d.out;
    u->dec.push_back(d), fan = d.out, wmax = d.out > wmax ? d.out : wmax;
  }
  u->n_policy = o;
  fan = sp->obs_size;
  for (int i = 0; i <= sp->num_value_layers; i++) {
    DenseP d{fan, i < sp->num_value_layers ? sp->value_layers[i] : 1, 0, 0, 0, 0, false};
    d.w = o, o += (size_t)d.in * d.out, d.b = o, o += d.out;
    u->val.push_back(d), fan = d.out, wmax = d.out > wmax ? d.out : wmax;
  }
  u->n_total = o, u->wmax = wmax;
  if (wmax > 1024) {
    vnl_ppo_update_destroy(u);
    (void)hipSetDevice(prev);
    return pfail(VNL_ERR_UNSUPPORTED, "layer width above 1024 (LayerNorm backward keeps a row in registers)");
  }
  u->ntp = (sp->traj_size + 3) & ~3;
  const size_t N = u->N, Nv = u->Nv;
  int rc = VNL_OK;
#define AL(p, n) if (rc == VNL_OK) rc = dalloc(u, &(p), (n))
  AL(u->obsn, Nv * sp->obs_size);
  AL(u->trajp, N * u->ntp);
  AL(u->D0, N * (sp->latent_size + sp->obs_size));
  AL(u->ml, N * 2 * sp->latent_size);
  AL(u->mean, N * sp->latent_size);
  AL(u->logvar, N * sp->latent_size);
  AL(u->logits, N * 2 * sp->action_size);
  AL(u->v, Nv);
  AL(u->gl, N * 2 * sp->action_size);
  AL(u->gb, N);
  AL(u->gklm, N * sp->latent_size);
  AL(u->gkll, N * sp->latent_size);
  AL(u->vs, N);
  AL(u->adv, N);
  AL(u->headws, VNL_PPO_HEAD_WORKSPACE_FLOATS);
  AL(u->dml, N * 2 * sp->latent_size);
  AL(u->dA, Nv * wmax);
  AL(u->dB, Nv * wmax);
  AL(u->dPa, N * wmax);
  AL(u->dPb, N * wmax);
  for (auto& d : u->enc) u->dz_floats += N * d.out;
  for (auto& d : u->dec) u->dz_floats += N * d.out;
  AL(u->dzarena, u->dz_floats);
  for (auto& d : u->enc) {
    float *h = nullptr, *y = nullptr, *s = nullptr;
    AL(h, N * d.out);
    AL(y, N * d.out);
    AL(s, N * 2);
    u->encH.push_back(h), u->encY.push_back(y), u->encS.push_back(s);
  }
  for (auto& d : u->dec) {
    float *h = nullptr, *y = nullptr, *s = nullptr;
    if (d.ln) {
      AL(h, N * d.out);
      AL(y, N * d.out);
      AL(s, N * 2);
    }
    u->decH.push_back(h), u->decY.push_back(y), u->decS.push_back(s);
  }
  for (size_t i = 0; i + 1 < u->val.size(); i++) {
    float *z = nullptr, *a = nullptr;
    AL(z, Nv * u->val[i].out);
    AL(a, Nv * u->val[i].out);
    u->valZ.push_back(z), u->valA.push_back(a);
  }
  if (N > 8192) {  // (the reference's own proportions: 20,480 rows; at the 2,560 rows of the primary config the arena is not used)
    u->ldt = (int)((N + 3) & ~(size_t)3);
    size_t rows_v = 0, rows_p = 0;
    for (const DenseP& d : u->val) rows_v += d.in + 1;
    for (const DenseP& d : u->enc) rows_p += d.in + 1;
    for (const DenseP& d : u->dec) rows_p += d.in + 1;
    rows_p += u->enc.back().out + 1;       // the latent heads' shared input
    rows_p += u->val.back().in + 1;        // the value head's gradient rides with the intention network's group
    u->tarena_v = rows_v * u->ldt, u->tarena_p = rows_p * u->ldt;
    AL(u->tarena, u->tarena_v + u->tarena_p);
  }
  // split-K slabs of every weight gradient of a step (wgrad falls back to one slab) + the LayerNorm backward's per-block
  // partial sums (80 blocks x 2 h each)
  u->slab_floats_p = 8 * u->n_policy + 4096 + (size_t)80 * 2 * wmax * (u->enc.size() + u->dec.size());
  u->slab_floats = 8 * (u->n_total - u->n_policy) + 4096 + u->slab_floats_p;
  AL(u->slabs, u->slab_floats);
  if (rc == VNL_OK && hipStreamCreateWithFlags(&u->s2, hipStreamNonBlocking) != hipSuccess) rc = pfail(VNL_ERR_HIP, "hipStreamCreate");
  for (int k = 0; k < 4 && rc == VNL_OK; k++)
    if (hipEventCreateWithFlags(&u->ev[k], hipEventDisableTiming) != hipSuccess) rc = pfail(VNL_ERR_HIP, "hipEventCreate");
  u->part_floats = 256 * 2 * (size_t)wmax;
  AL(u->part, u->part_floats);
#undef AL
  if (rc == VNL_OK) {  // the fused forward of the intention network, if it fits that kernel (else layer by layer: not an error)
    vnl_policy_spec ps{};
    ps.traj_size = sp->traj_size, ps.obs_size = sp->obs_size, ps.action_size = sp->action_size, ps.latent_size = sp->latent_size;
    ps.num_encoder_layers = sp->num_encoder_layers, ps.num_decoder_layers = sp->num_decoder_layers;
    for (int i = 0; i < 8; i++) ps.encoder_layers[i] = sp->encoder_layers[i], ps.decoder_layers[i] = sp->decoder_layers[i];
    vnl_policy* pol = nullptr;
    if (vnl_policy_create(&ps, N, device, &pol) == VNL_OK) {
      if ((size_t)vnl_policy_num_params(pol) == u->n_policy) u->fused = pol;
      else vnl_policy_destroy(pol);
    }
  }
  (void)hipSetDevice(prev);
  if (rc != VNL_OK) {
    vnl_ppo_update_destroy(u);
    return rc;
  }
  *out = u;
  return VNL_OK;
}

// LayerNorm + ReLU backward of one layer; the per-block partial sums of d gamma | d beta (contiguous in the flat layout:
// scale then bias) go to a slab region of their own and are summed by the step's final reduce_jobs launch
static void ln_bwd(vnl_ppo_update* u, SlabPool* pool, hipStream_t st, const float* dY, const float* H, const float* stats,
                   const float* gamma, float* dZ, float* dgamma, int rows, int h) {
  const int blocks = h <= 256 ? 80 : 64;
  const size_t tot = ((size_t)2 * h + 3) & ~(size_t)3;
  float* part = u->part;
  bool deferred = false;
  if (pool->used + blocks * tot <= pool->cap && pool->jobs.njobs < VNL_MAX_JOBS) {
    ReduceJobs& J = pool->jobs;
    const int j = J.njobs++;
    part = pool->base + pool->used;
    J.out[j] = dgamma, J.bias_out[j] = nullptr, J.part[j] = part, J.S[j] = blocks;
    J.wn[j] = (unsigned)(2 * h), J.tot[j] = (unsigned)tot, J.real[j] = (unsigned)(2 * h);
    J.start4[j + 1] = J.start4[j] + (unsigned)(tot / 4);
    pool->used += blocks * tot;
    deferred = true;
  }
  if (h <= 128) hipLaunchKernelGGL((ln_bwd_kernel<128, 16>), dim3(blocks), dim3(1024), 0, st, dY, H, stats, gamma, dZ, part, rows, h, (int)tot);
  else if (h <= 256) hipLaunchKernelGGL((ln_bwd_kernel<256, 16>), dim3(blocks), dim3(1024), 0, st, dY, H, stats, gamma, dZ, part, rows, h, (int)tot);
  else hipLaunchKernelGGL((ln_bwd_kernel<1024, 4>), dim3(blocks), dim3(256), 0, st, dY, H, stats, gamma, dZ, part, rows, h, (int)tot);
  if (!deferred)
    hipLaunchKernelGGL(sum_slabs_kernel, dim3((2 * h + 255) / 256), dim3(256), 0, st, dgamma, (const float*)part, blocks, tot,
                       (size_t)2 * h);
}

extern "C" int vnl_ppo_minibatch_grad(vnl_ppo_update* u, const float* params, const vnl_ppo_batch* bt, const vnl_ppo_hparams* hp,
                                      float* grads, float* metrics, void* stream) {
  return vnl_ppo_minibatch_grad_part(u, params, bt, hp, grads, metrics, stream, 0);
}

/* part 0: the whole step.  part 1: forward of both networks, the loss head, the VALUE network's backward -- on return (in
 * stream order) the value segment of `grads` [policy params .. end) is final.  part 2: the policy network's backward (after
 * part 1 with the same arguments): the policy segment.  Data-parallel training issues the all-reduce of the value segment
 * between the two, so that it overlaps part 2 (reference ppo_imitation/train.py:251-268: one pmean per minibatch step). */
extern "C" int vnl_ppo_minibatch_grad_part(vnl_ppo_update* u, const float* params, const vnl_ppo_batch* bt,
                                           const vnl_ppo_hparams* hp, float* grads, float* metrics, void* stream, int part) {
  if (part < 0 || part > 2) return pfail(VNL_ERR_ARG, "vnl_ppo_minibatch_grad_part: part must be 0, 1 or 2");
  if (!u || !params || !bt || !hp || !grads || !metrics) return pfail(VNL_ERR_ARG, "vnl_ppo_minibatch_grad: null argument");
  const void* need[] = {bt->traj, bt->obs, bt->next_obs_last, bt->raw_action, bt->behaviour_log_prob, bt->reward, bt->truncation,
                        bt->discount, bt->eps_latent, bt->eps_entropy};
  for (const void* p : need)
    if (!p) return pfail(VNL_ERR_ARG, "vnl_ppo_minibatch_grad: null batch buffer");
  if ((bt->obs_mean == nullptr) != (bt->obs_std == nullptr)) return pfail(VNL_ERR_ARG, "obs_mean / obs_std: both or neither");
  int prev = 0;
  PCHK(hipGetDevice(&prev));
  if (prev != u->device) PCHK(hipSetDevice(u->device));
  // Every return path -- the error returns of PCHK / rc included -- restores the caller's device and joins the second
  // stream back into the caller's (an unjoined fork would invalidate a hipGraph capture of the caller's stream).
  struct Scope {
    int prev, device;
    hipStream_t st, s2;
    hipEvent_t ev;
    bool forked = false, done = false;
    ~Scope() {
      if (forked && !done && hipEventRecord(ev, s2) == hipSuccess) (void)hipStreamWaitEvent(st, ev, 0);
      if (prev != device) (void)hipSetDevice(prev);
    }
  } scope{prev, u->device, (hipStream_t)stream, u->s2, u->ev[3]};
  // Two chains per step: the value MLP (three big GEMMs forward, four backward) on the caller's stream, the intention
  // network (twenty small, latency-bound launches) beside it on the handle's second stream; they meet at the loss head and
  // at the final reduction.  The fork / join events make the second stream part of a hipGraph capture of the caller's.
  hipStream_t st = (hipStream_t)stream, sp2 = u->s2;
  const vnl_ppo_net_spec& sp = u->spec;
  const int N = u->N, Nv = u->Nv, no = sp.obs_size, lat = sp.latent_size, A2 = 2 * sp.action_size;
  // split-K slabs and the deferred reductions, one pool per chain: the intention network's are summed on its own stream
  // as soon as its grouped weight-gradient launch is done, beside the value MLP's last GEMMs
  SlabPool pool{u->slabs, u->slab_floats - u->slab_floats_p, 0, {}, u->tile};
  SlabPool poolP{u->slabs + (u->slab_floats - u->slab_floats_p), u->slab_floats_p, 0, {}, u->tile};
  pool.jobs.njobs = 0, pool.jobs.start4[0] = 0, pool.jobs.prio = 0;
  poolP.jobs.njobs = 0, poolP.jobs.start4[0] = 0, poolP.jobs.prio = u->prio;
  pool.wg_target = poolP.wg_target = u->wg_target;
  if (u->tarena && u->wgrad_t) {
    pool.tarena = u->tarena, pool.tcap = u->tarena_v, pool.ldt = u->ldt;
    poolP.tarena = u->tarena + u->tarena_v, poolP.tcap = u->tarena_p, poolP.ldt = u->ldt;
  }
  GemmGroup group, group_nn;
  group.n = 0, group_nn.n = 0;
  Gemm GV{st, &pool}, GP{sp2, &poolP, &group, u->prio, &group_nn};
  auto reduce_pool = [](SlabPool& pl, hipStream_t s) {
    if (pl.jobs.njobs == 0) return;
    const unsigned total4 = pl.jobs.start4[pl.jobs.njobs];
    unsigned blocks = (total4 + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(reduce_jobs_kernel, dim3(blocks), dim3(256), 0, s, pl.jobs);
  };
  size_t dz_used = 0;
  auto dz_alloc = [&](size_t n) {  // (sized in create for every layer's N x out: cannot run out)
    float* q = u->dzarena + dz_used;
    dz_used += n;
    return q;
  };
  const float* P = params;
  float* Gr = grads;

  const int nvl = (int)u->val.size();  // hidden layers + the output layer
  if (part == 2) {  // the policy network's backward alone: fork the second stream from the caller's
    PCHK(hipEventRecord(u->ev[0], st));
    PCHK(hipStreamWaitEvent(sp2, u->ev[0], 0));
    scope.forked = true;
  } else {
    // ---------------- forward
    {
      const size_t tot = (size_t)Nv * no + (size_t)N * u->ntp;
      hipLaunchKernelGGL(prep_kernel, dim3((unsigned)((tot + 255) / 256 > 4096 ? 4096 : (tot + 255) / 256)), dim3(256), 0, st, bt->obs,
                         bt->next_obs_last, bt->obs_mean, bt->obs_std, bt->traj, u->obsn, u->trajp, N, Nv, no, sp.traj_size, u->ntp);
    }
    PCHK(hipEventRecord(u->ev[0], st));
    PCHK(hipStreamWaitEvent(sp2, u->ev[0], 0));
    scope.forked = true;
    // value MLP over the T*B rows + the B bootstrap rows (ppo_networks.py:114-118; swish)
    {
      const float* x = u->obsn;
      int ldx = no;
      for (int i = 0; i + 1 < nvl; i++) {
        const DenseP& d = u->val[i];
        GV.run(false, false, EPI_SWISH, x, ldx, P + d.w, d.out, u->valA[i], d.out, Nv, d.out, d.in, P + d.b, nullptr, 0, u->valZ[i]);
        x = u->valA[i], ldx = d.out;
      }
      const DenseP& d = u->val[nvl - 1];
      hipLaunchKernelGGL(rowdot_kernel, dim3((Nv + 3) / 4), dim3(256), 0, st, x, P + d.w, P + d.b, u->v, Nv, d.in);
    }
    if (u->fused && u->fwd_mode != 0 && (u->fwd_mode_forced || N <= 4096)) {  // (more rows: layer by layer is faster -- 20,480 rows: 3.08 vs 3.34 ms)
      // the whole intention network as ONE launch: a 16-row tile goes through encoder, latent heads, reparameterisation and
      // decoder without leaving LDS (csrc/vnl_policy.hip, the acting path's kernel in its training form), writing what
      // the backward pass reads; it takes the raw trajectory / observation, so it does not wait for prep_kernel
      PolicyTrainOut to{};
      for (size_t i = 0; i < u->enc.size(); i++) to.encH[i] = u->encH[i], to.encS[i] = u->encS[i], to.encY[i] = u->encY[i];
      for (size_t i = 0; i < u->dec.size(); i++)
        if (u->dec[i].ln) to.decH[i] = u->decH[i], to.decS[i] = u->decS[i], to.decY[i] = u->decY[i];
      to.ml = u->ml, to.D0 = u->D0;
      if (u->fwd_mode == 2) {  // the first Dense (K = traj_size: most of the network's arithmetic) as a GEMM launch, the rest fused
        const DenseP& d = u->enc[0];
        GP.run(false, false, EPI_RELU, u->trajp, u->ntp, P + d.w, d.out, u->encH[0], d.out, N, d.out, d.in, P + d.b);
      }
      const int rc = vnl_policy_forward_train_(u->fused, P, bt->obs_mean, bt->obs_std, bt->traj, bt->obs, bt->eps_latent, N, u->logits,
                                               u->mean, u->logvar, &to, u->fwd_mode == 2, sp2);
      if (rc != VNL_OK) return rc;
    } else {
    // encoder (intention_policy_network.py:20-44)
    {
      const float* x = u->trajp;
      int ldx = u->ntp;
      for (size_t i = 0; i < u->enc.size(); i++) {
        const DenseP& d = u->enc[i];
        GP.run(false, false, EPI_RELU, x, ldx, P + d.w, d.out, u->encH[i], d.out, N, d.out, d.in, P + d.b);
        ln_fwd(sp2, u->encH[i], P + d.g, P + d.be, u->encY[i], u->encS[i], N, d.out);
        x = u->encY[i], ldx = d.out;
      }
      const int fan = u->enc.back().out;
      GP.run(false, false, EPI_NONE, x, ldx, P + u->mean_w, lat, u->ml, 2 * lat, N, lat, fan, P + u->mean_b);
      GP.run(false, false, EPI_NONE, x, ldx, P + u->lv_w, lat, u->ml + lat, 2 * lat, N, lat, fan, P + u->lv_b);
      const size_t nz = (size_t)N * (lat + no);
      hipLaunchKernelGGL(reparam_kernel, dim3((unsigned)((nz + 255) / 256)), dim3(256), 0, sp2, (const float*)u->ml, bt->eps_latent,
                         (const float*)u->obsn, u->D0, N, lat, no);
      hipLaunchKernelGGL(split_ml_kernel, dim3((N * lat + 255) / 256), dim3(256), 0, sp2, (const float*)u->ml, u->mean, u->logvar, N, lat);
    }
    // decoder (intention_policy_network.py:47-70)
    {
      const float* x = u->D0;
      int ldx = lat + no;
      for (size_t i = 0; i < u->dec.size(); i++) {
        const DenseP& d = u->dec[i];
        if (d.ln) {
          GP.run(false, false, EPI_RELU, x, ldx, P + d.w, d.out, u->decH[i], d.out, N, d.out, d.in, P + d.b);
          ln_fwd(sp2, u->decH[i], P + d.g, P + d.be, u->decY[i], u->decS[i], N, d.out);
          x = u->decY[i], ldx = d.out;
        } else {
          GP.run(false, false, EPI_NONE, x, ldx, P + d.w, d.out, u->logits, d.out, N, d.out, d.in, P + d.b);
        }
      }
    }
    }
    // ---------------- loss head: GAE, clipped surrogate, value / entropy / KL terms and d loss / d (network outputs)
    {
      vnl_ppo_head_args a{};
      a.T = u->T, a.B = u->B, a.act = sp.action_size, a.latent = lat;
      a.logits = u->logits, a.baseline = u->v, a.bootstrap = u->v + N, a.lat_mean = u->mean, a.lat_logvar = u->logvar;
      a.raw_action = bt->raw_action, a.behaviour_log_prob = bt->behaviour_log_prob, a.reward = bt->reward;
      a.truncation = bt->truncation, a.discount = bt->discount, a.eps_entropy = bt->eps_entropy;
      a.entropy_cost = hp->entropy_cost, a.discounting = hp->discounting, a.reward_scaling = hp->reward_scaling;
      a.gae_lambda = hp->gae_lambda, a.clipping_epsilon = hp->clipping_epsilon, a.kl_weight = hp->kl_weight;
      a.min_std = hp->min_std, a.var_scale = hp->var_scale, a.normalize_advantage = hp->normalize_advantage;
      a.g_logits = u->gl, a.g_baseline = u->gb, a.g_lat_mean = u->gklm, a.g_lat_logvar = u->gkll;
      a.vs = u->vs, a.advantages = u->adv, a.metrics = metrics;
      // GAE, the advantage statistics and d v_loss / d baseline need the value outputs only: they run on the value stream as
      // soon as the value MLP's forward is done, and its BACKWARD pass follows at once (below) -- the three big GEMMs of that
      // pass no longer wait for the intention network's forward
      int rc = vnl_ppo_head_phase_(&a, u->headws, stream, 1);
      if (rc != VNL_OK) return rc;
      PCHK(hipEventRecord(u->ev[2], st));
      // metrics[8] = prediction_corr (a metric only), in the same slack; NaN ("not computed", never a fake 0.0) when the 2T
      // rows do not fit in LDS -- the same rule as the torch backend (intention_losses.py: _corr_fits)
      const size_t lds = ((size_t)2 * u->T * u->B + 2 * u->T) * sizeof(float);
      if (lds <= 60 * 1024)
        hipLaunchKernelGGL(prediction_corr_kernel, dim3(1), dim3(VNL_CORR_THREADS), lds, st, (const float*)u->vs, bt->reward, hp->reward_scaling,
                           u->T, u->B, metrics + 8);
      else
        PCHK(hipMemsetAsync(metrics + 8, 0xff, sizeof(float), st));  // 0xffffffff: a quiet NaN
      // the per-sample head (policy / entropy / KL terms, d loss / d logits, d loss / d latent heads) on the intention
      // network's stream, right behind its forward pass and in front of its backward pass
      PCHK(hipStreamWaitEvent(sp2, u->ev[2], 0));
      rc = vnl_ppo_head_phase_(&a, u->headws, sp2, 2);
      if (rc != VNL_OK) return rc;
    }
  }
  // ---------------- backward: value MLP (the bootstrap rows carry no gradient: stop_gradient, intention_losses.py:137)
  if (part != 2) {
    const DenseP& dl = u->val[nvl - 1];
    const float* Alast = nvl >= 2 ? u->valA[nvl - 2] : u->obsn;
    // d w_out = A' gb, d b_out = sum gb: a 1025 x 1 product, 16 us as a launch of its own at the head of this chain -- it
    // joins the intention network's grouped weight-gradient launch on the other stream instead (operands ready since the head)
    // (split step: it must be final with the rest of the value segment, so it takes a launch of its own on this stream)
    if (part == 0) GP.wgrad(Alast, dl.in, u->gb, 1, Gr + dl.w, Gr + dl.b, dl.in, 1, N);
    else GV.wgrad(Alast, dl.in, u->gb, 1, Gr + dl.w, Gr + dl.b, dl.in, 1, N);
    float *dz = u->dA, *dz_other = u->dB;
    if (nvl >= 2) {
      const int h = u->val[nvl - 2].out;
      const size_t n = (size_t)N * h;
      hipLaunchKernelGGL(value_seed_kernel, dim3((unsigned)((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256)), dim3(256), 0, st,
                         (const float*)u->gb, P + dl.w, (const float*)u->valZ[nvl - 2], dz, N, h);
    }
    for (int i = nvl - 2; i >= 0; i--) {  // dz = d loss / d Z_i  [N][out_i]
      const DenseP& d = u->val[i];
      const float* xin = i > 0 ? u->valA[i - 1] : u->obsn;
      GV.wgrad(xin, d.in, dz, d.out, Gr + d.w, Gr + d.b, d.in, d.out, N);  // [dW; db] = [X | 1]' dZ
      if (i > 0)  // dZ_{i-1} = (dZ_i W_i') * swish'(Z_{i-1})
        GV.run(false, true, EPI_MUL_DSWISH, dz, d.out, P + d.w, d.out, dz_other, d.in, N, d.in, d.out, nullptr, u->valZ[i - 1], d.in);
      float* t = dz;
      dz = dz_other, dz_other = t;
    }
  }
  // ---------------- backward: decoder, latent, encoder (second stream)
  if (part != 1) {
    // The chain of input gradients runs first, launch after launch; the layers' weight gradients [X | 1]' dZ are only
    // collected (GP.wgrad defers them) and run as one grouped launch at the end, so every dZ keeps a buffer of its own.
    float* dcur = u->dPa;     // d loss / d (a layer's input), consumed by the next launch
    const float* dz = u->gl;  // d loss / d logits [N][2 act]
    int dzw = A2;
    for (int i = (int)u->dec.size() - 1; i >= 0; i--) {
      const DenseP& d = u->dec[i];
      const float* xin = i > 0 ? u->decY[i - 1] : u->D0;
      const int ldx = i > 0 ? u->dec[i - 1].out : lat + no;
      GP.wgrad(xin, ldx, dz, dzw, Gr + d.w, Gr + d.b, d.in, d.out, N);
      // d input: all of it for a hidden layer; only the latent columns of the first layer's input [z | obs]
      const int nin = i > 0 ? d.in : lat;
      GP.run(false, true, EPI_NONE, dz, dzw, P + d.w, d.out, dcur, nin, N, nin, d.out);
      if (i > 0) {
        const DenseP& pd = u->dec[i - 1];
        float* dnext = dz_alloc((size_t)N * pd.out);
        ln_bwd(u, &poolP, sp2, dcur, u->decH[i - 1], u->decS[i - 1], P + pd.g, dnext, Gr + pd.g, N, pd.out);
        dz = dnext, dzw = pd.out;
      }
    }
    // dcur = d loss / d z  [N][lat]
    hipLaunchKernelGGL(latent_bwd_kernel, dim3((N * lat + 255) / 256), dim3(256), 0, sp2, (const float*)dcur, lat, (const float*)u->ml,
                       bt->eps_latent, (const float*)u->gklm, (const float*)u->gkll, u->dml, N, lat);
    const DenseP& le = u->enc.back();
    const float* ylast = u->encY.back();
    GP.wgrad(ylast, le.out, u->dml, 2 * lat, Gr + u->mean_w, Gr + u->mean_b, le.out, lat, N);
    GP.wgrad(ylast, le.out, u->dml + lat, 2 * lat, Gr + u->lv_w, Gr + u->lv_b, le.out, lat, N);
    float* dy = u->dPb;  // d loss / d (last encoder output) = dmean Wm' + dlogvar Wlv'
    GP.run(false, true, EPI_NONE, u->dml, 2 * lat, P + u->mean_w, lat, dy, le.out, N, le.out, lat);
    GP.run(false, true, EPI_NONE, u->dml + lat, 2 * lat, P + u->lv_w, lat, dy, le.out, N, le.out, lat, nullptr, nullptr, 0, nullptr, 1);
    for (int i = (int)u->enc.size() - 1; i >= 0; i--) {
      const DenseP& d = u->enc[i];
      float* dzi = dz_alloc((size_t)N * d.out);
      ln_bwd(u, &poolP, sp2, dy, u->encH[i], u->encS[i], P + d.g, dzi, Gr + d.g, N, d.out);
      const float* xin = i > 0 ? u->encY[i - 1] : u->trajp;
      const int ldx = i > 0 ? u->enc[i - 1].out : u->ntp;
      GP.wgrad(xin, ldx, dzi, d.out, Gr + d.w, Gr + d.b, d.in, d.out, N);
      if (i > 0) GP.run(false, true, EPI_NONE, dzi, d.out, P + d.w, d.out, dy, d.in, N, d.in, d.out);
    }
    // (measured and dropped: the decoder's weight gradients as a launch of their own as soon as their operands are ready --
    // on a third stream hipGraph capture faulted in hipStreamEndCapture, on the caller's stream behind the value MLP's GEMMs
    // the replayed graph started this chain late: 0.49 ms per step against 0.46)
    GP.flush();  // every weight gradient of the intention network: one grouped launch
    reduce_pool(poolP, sp2);
  }
  PCHK(hipEventRecord(u->ev[3], sp2));
  PCHK(hipStreamWaitEvent(st, u->ev[3], 0));
  reduce_pool(pool, st);  // the value MLP's split-K weight / bias gradients, summed in a fixed order by one launch
  scope.done = true;  // joined above
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return pfail(VNL_ERR_HIP, hipGetErrorString(e));
  return VNL_OK;
}
