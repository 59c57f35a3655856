// Fused intention-policy inference for gfx950 (C-ABI: vnl_policy_* in include/vnl.h).
//
// Replaces, for the acting path, reference ppo_imitation/ppo_networks.py:45-83 (make_inference_fn.policy)
// and intention_policy_network.py:20-105 (Encoder -> reparameterize -> Decoder), plus the tanh-Normal
// sampling / log-prob of brax NormalTanhDistribution [UPSTREAM].
//
// One 1024-thread workgroup owns a tile of 16 envs for the whole network (256 workgroups at the benchmark's 4096
// envs: every CU busy; 32-env tiles left half of the chip idle):
//   traj tile (16 x 795) -> LDS -> [Dense+ReLU+LayerNorm] x len(encoder) -> fc2_mean | fc2_logvar (one pass)
//   -> z = mean + eps * exp(logvar / 2) -> [z | (obs - mu) / sigma] -> decoder -> logits (16 x 60)
//   -> scale = softplus(s) + 1e-3, raw = loc + scale * eps, action = tanh(raw), log_prob.
// Every Dense is a sequence of v_mfma_f32_16x16x4_f32 (exact fp32, the reference's precision).  The sixteen waves
// form a (64-column group) x (K slice) grid; A fragments come from LDS (leading dimension = 2 mod 32 -> the
// ds_read_b32 is conflict-free), B fragments stream from the L2-resident weight buffer (row-major (in, out), as
// Flax) with one dwordx4 per lane, two sets of four loads in flight per wave; the K slices' partial sums are added
// in LDS in slice order.  Activations never leave LDS; the only HBM traffic is the inputs, the weights and the
// outputs.  What bounds it: every workgroup streams all 1.36 MB of weights from L2 (347 MB per launch at 4096 envs,
// ~20 us at the L2's ~17 TB/s) beside 17.6 us of MFMA issue per CU; measured phase times: tools/policy_stage_profile.py.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/vnl.h"
#include "vnl_policy_train.h"

#define PT 16          /* envs per workgroup = rows of the 16x16x4 MFMA tile */
#define PTHREADS 1024  /* 16 waves: four per SIMD, some compute while the others wait for their weight loads */
#define PNT ((int)blockDim.x) /* threads of THIS launch: 1024, or 256 for the training form that starts at the first Dense output */
#define LN_EPS 1e-6f   /* flax.linen.LayerNorm default */

typedef float v4f __attribute__((ext_vector_type(4)));


// Y[16 x N] (LDS, ld = ldy) = X[16 x K] (LDS, ld = ldx) @ W[K x N] (global, row-major) + b, optional ReLU.
//
// The sixteen waves form a (column group) x (K slice) grid: a wave owns 64 consecutive output columns -- four MFMA
// tiles, tile t holding columns c0 + t of every lane's float4 -- so one global_load_dwordx4 per lane fetches four k-rows
// of 256 contiguous bytes (whole cache lines; a 16-column tile per wave used half of every line it touched and was
// bound by the L1's line rate).  With N = 256 that leaves 4 K slices, N = 128 -> 8, N <= 64 -> 16; the slices' partial
// sums meet in P (LDS) and are added in slice order, bias first: deterministic.
// Fragment layout of v_mfma_f32_16x16x4f32:  A: lane l holds X[l % 16][k0 + l / 16];  B: lane l holds
// W[k0 + l / 16][column of l % 16];  D: lane l, register i holds Y[(l / 16) * 4 + i][column of l % 16].
__device__ __forceinline__ int dense_col_groups(int N) {  // power of two <= 16 covering ceil(N / 64)
  int G = (N + 63) >> 6, Gp = 1;
  while (Gp < G && Gp < PNT / 64) Gp <<= 1;
  return Gp;
}

template <bool VEC>  // VEC: N % 4 == 0 and W 16-byte aligned -> one dwordx4 load per lane and k-row
__device__ __forceinline__ void dense_tile_t(const float* X, int ldx, int K, const float* __restrict__ params, int w_off,
                                             int b_off, int N, float* Y, int ldy, bool relu, float* P, int split,
                                             int w2_off, int b2_off) {
  const float* __restrict__ bias = params + b_off;
  const float* __restrict__ bias2 = params + b2_off;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = lane & 15, kq = lane >> 4;
  const int G = (N + 63) >> 6, Gp = dense_col_groups(N), S = (PNT / 64) / Gp;
  const int g0 = wave & (Gp - 1), s = wave / Gp;
  const int N4 = (N + 3) & ~3;
  const int T = (K + 3) >> 2, Ts = (T + S - 1) / S;  // k-steps of 4, per slice
  const int t0 = s * Ts, t1 = min(T, t0 + Ts);
  const float* xr = X + j * ldx + kq;
  // this thread's share of the bias (sum phase below), requested now so that its L2 latency passes under the MFMA loop
  float bias_r[4];
#pragma unroll
  for (int q = 0; q < 4; q++) {
    const int e = tid + q * PNT;
    const int c = e % N;
    bias_r[q] = e < PT * N ? (c < split ? bias[c] : bias2[c - split]) : 0.f;
  }
  const int ldw = split < N ? split : N;  // row stride of the weight matrices
  for (int g = g0; g < G; g += Gp) {
    const int c0 = g * 64 + 4 * j;
    // W[4 step + kq][c0 .. c0 + 3].  Outside the slice / the matrix the address is clamped to valid weights and the
    // A operand is zeroed instead (no select on the loaded value: it would pin the wait right behind the load);
    // columns >= N compute garbage that is never read back.
    auto load_b = [&](int step) -> v4f {
      const int k = 4 * step + kq;
      const float* w = params + (c0 < split ? w_off : w2_off - split) + (size_t)((step < t1 && k < K) ? k : 0) * ldw;
      v4f b;
      if (VEC) {
        b = *(const v4f*)(w + (c0 < N ? c0 : 0));
      } else {
#pragma unroll
        for (int t = 0; t < 4; t++) b[t] = w[c0 + t < N ? c0 + t : 0];
      }
      return b;
    };
    v4f acc[4];
#pragma unroll
    for (int t = 0; t < 4; t++) acc[t] = v4f{0.f, 0.f, 0.f, 0.f};
    // four k-steps (16 MFMAs) on the weights in b, A fragments from LDS
    auto mma4 = [&](const v4f (&b)[4], int t) {
      float a[4];
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const int step = t + u;
        const bool ok = step < t1 && 4 * step + kq < K;  // columns of X past K hold stale activations
        float v = xr[ok ? 4 * step : 0];
        a[u] = ok ? v : 0.f;
      }
#pragma unroll
      for (int u = 0; u < 4; u++)
#pragma unroll
        for (int tt = 0; tt < 4; tt++) acc[tt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u], b[u][tt], acc[tt], 0, 0, 0);
    };
    // ping-pong weight registers (no copies: a copy of a prefetched value would pin the wait for it at the loop's end):
    // the MFMAs on one set run under the loads of the other
    v4f b0[4], b1[4];
    if (t0 < t1) {
#pragma unroll
      for (int u = 0; u < 4; u++) b0[u] = load_b(t0 + u);
    }
    for (int t = t0; t < t1; t += 8) {  // the conditions are wave-uniform; no load is issued for nothing
      const bool more1 = t + 4 < t1, more0 = t + 8 < t1;
      if (more1) {
#pragma unroll
        for (int u = 0; u < 4; u++) b1[u] = load_b(t + 4 + u);
      }
      __builtin_amdgcn_sched_barrier(0);  // keep the loads in front of the MFMAs they are to run under
      mma4(b0, t);
      __builtin_amdgcn_sched_barrier(0);
      if (more0) {
#pragma unroll
        for (int u = 0; u < 4; u++) b0[u] = load_b(t + 8 + u);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (more1) mma4(b1, t + 4);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (c0 < N4) {
#pragma unroll
      for (int i = 0; i < 4; i++)
        *(v4f*)(P + (size_t)(s * PT + kq * 4 + i) * N4 + c0) = v4f{acc[0][i], acc[1][i], acc[2][i], acc[3][i]};
    }
  }
  __syncthreads();
  auto finish = [&](int e, float v) {
    const int r = e / N, c = e - r * N;
    for (int ss = 0; ss < S; ss++) v += P[(size_t)(ss * PT + r) * N4 + c];
    Y[r * ldy + c] = relu ? fmaxf(v, 0.f) : v;
  };
#pragma unroll
  for (int q = 0; q < 4; q++) {
    const int e = tid + q * PNT;
    if (e < PT * N) finish(e, bias_r[q]);
  }
  for (int e = tid + 4 * PNT; e < PT * N; e += PNT) finish(e, e % N < split ? bias[e % N] : bias2[e % N - split]);
  __syncthreads();
}

// Weights and biases are addressed as offsets into the flat parameter vector.
// `split` < N: columns [split, N) come from a second Dense (w2_off, b2_off) of the same input -- the two latent heads run as
// one pass; needs split % 64 == 0 (a column group reads one matrix) and N == 2 * split (one row stride).
__device__ __forceinline__ void dense_tile(const float* X, int ldx, int K, const float* __restrict__ params, int w_off,
                                           int b_off, int N, float* Y, int ldy, bool relu, float* P, int split = 1 << 30,
                                           int w2_off = 0, int b2_off = 0) {
  // branch once, outside the pipelined loop: control flow between the prefetch loads and the MFMAs makes the compiler
  // wait for every outstanding load (s_waitcnt vmcnt(0)) before the first MFMA of a trip
  const int ldw = split < N ? split : N;
  if ((ldw & 3) == 0 && ((uintptr_t)params & 15) == 0 && (w_off & 3) == 0 && (w2_off & 3) == 0)
    dense_tile_t<true>(X, ldx, K, params, w_off, b_off, N, Y, ldy, relu, P, split, w2_off, b2_off);
  else
    dense_tile_t<false>(X, ldx, K, params, w_off, b_off, N, Y, ldy, relu, P, split, w2_off, b2_off);
}

// the tile's rows (LDS, ld) -> dst[(e0 + r) * width + c], r < nrow: what the training form leaves for the backward pass
__device__ __forceinline__ void store_tile(const float* src, int ld, int nrow, int width, float* __restrict__ dst, int e0) {
  for (int i = threadIdx.x; i < nrow * width; i += PNT) {
    const int r = i / width, c = i - r * width;
    dst[(size_t)(e0 + r) * width + c] = src[r * ld + c];
  }
}

// in-place LayerNorm over the N columns of each of the PT rows (one row per wave); the training form computes the variance
// in two passes (as csrc/vnl_ppo.hip's ln_fwd_kernel does) and leaves mean | 1 / sqrt(var + eps) of the rows in stats
template <bool TRAIN>
__device__ __forceinline__ void layer_norm_rows(float* Y, int ldy, int N, const float* __restrict__ g,
                                                const float* __restrict__ be, float* __restrict__ stats = nullptr, int e0 = 0,
                                                int nrow = 0) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int RPW = PT / (PNT / 64);  // rows per wave
  // scale / bias of the first 256 columns are requested before the row statistics (their latency passes under them)
  float gr[4], br[4];
#pragma unroll
  for (int q = 0; q < 4; q++) {
    const int c = lane + 64 * q;
    gr[q] = c < N ? g[c] : 0.f, br[q] = c < N ? be[c] : 0.f;
  }
  for (int r = wave * RPW; r < wave * RPW + RPW; r++) {
    float* y = Y + r * ldy;
    float s = 0.f, ss = 0.f;
    for (int c = lane; c < N; c += 64) {
      float v = y[c];
      s += v, ss += v * v;
    }
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o), ss += __shfl_xor(ss, o);
    float mean = s / (float)N;
    float var = fmaxf(ss / (float)N - mean * mean, 0.f);
    if (TRAIN) {
      float q = 0.f;
      for (int c = lane; c < N; c += 64) {
        const float d = y[c] - mean;
        q += d * d;
      }
      for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o);
      var = q / (float)N;
    }
    float inv = rsqrtf(var + LN_EPS);
    if (TRAIN && lane == 0 && r < nrow) stats[2 * (size_t)(e0 + r)] = mean, stats[2 * (size_t)(e0 + r) + 1] = inv;
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const int c = lane + 64 * q;
      if (c < N) y[c] = (y[c] - mean) * inv * gr[q] + br[q];
    }
    for (int c = lane + 256; c < N; c += 64) y[c] = (y[c] - mean) * inv * g[c] + be[c];
  }
}

// dst[r][c] (LDS, ld) = src[r * width + c] for r < nrow, 0 for nrow <= r < PT; optionally (v - mean[c]) / std[c].
// src is the tile's contiguous chunk of PT * width floats; float4 loads, four per thread in flight, when it is 16-byte
// aligned (a load -> LDS store loop of scalars costs one HBM latency per trip: 13 trips for the traj tile).
__device__ __forceinline__ void load_tile(const float* __restrict__ src, int nrow, int width, float* dst, int ld,
                                          const float* __restrict__ mean, const float* __restrict__ stdv) {
  const int tid = threadIdx.x, n = nrow * width;
  auto put = [&](int i, float v) {
    const int r = i / width, c = i - r * width;
    if (mean) v = (v - mean[c]) / stdv[c];
    dst[r * ld + c] = v;
  };
  int done = 0;
  if (((uintptr_t)src & 15) == 0) {
    const v4f* s4 = (const v4f*)src;
    const int n4 = n >> 2;
    for (int base = 0; base < n4; base += 4 * PNT) {
      v4f v[4];
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const int i = base + u * PNT + tid;
        v[u] = s4[i < n4 ? i : 0];
      }
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const int i = base + u * PNT + tid;
        if (i < n4) {
#pragma unroll
          for (int q = 0; q < 4; q++) put(4 * i + q, v[u][q]);
        }
      }
    }
    done = n4 << 2;
  }
  for (int i = done + tid; i < n; i += PNT) put(i, src[i]);
  for (int i = n + tid; i < PT * width; i += PNT) {
    const int r = i / width, c = i - r * width;
    dst[r * ld + c] = 0.f;
  }
}

#ifdef VNL_PROFILE  // prof variant only: wall-clock (100 MHz) stamps of workgroup 0 at the phase boundaries
__device__ long long g_policy_stamps[32];
#define POL_STAMP(i)                                                               \
  do {                                                                             \
    if (blockIdx.x == 0 && threadIdx.x == 0) g_policy_stamps[i] = wall_clock64(); \
  } while (0)
extern "C" int vnl_policy_profile_stamps(long long* out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_policy_stamps), sizeof(long long) * 32) == hipSuccess ? 0 : -1;
}
#else
#define POL_STAMP(i)
#endif

__device__ __forceinline__ float softplusf(float x) { return x > 20.f ? x : log1pf(expf(x)); }

// MODE 0: acting (sampling, no intermediates).  MODE 1: training form -- no sampling, every intermediate the backward pass reads
// goes to t.  MODE 2: training form from the FIRST encoder layer's Dense output on (t.encH[0], made by a GEMM launch of
// csrc/vnl_ppo.hip): without the 16 x 795 trajectory tile the workgroup's LDS drops from 132 KB to 97 KB, so that workgroups of
// the value MLP's GEMMs stay resident beside it.
template <int MODE>
__global__ void __launch_bounds__(PTHREADS) vnl_policy_kernel_t(PolicyDev p, const float* __restrict__ params,
                                                              const float* __restrict__ obs_mean,
                                                              const float* __restrict__ obs_std,
                                                              const float* __restrict__ traj, const float* __restrict__ obs,
                                                              const float* __restrict__ eps_latent,
                                                              const float* __restrict__ eps_action, int batch,
                                                              int deterministic, float* __restrict__ action,
                                                              float* __restrict__ raw_action, float* __restrict__ log_prob,
                                                              float* __restrict__ logits, float* __restrict__ lat_mean,
                                                              float* __restrict__ lat_logvar,
                                                              const float* __restrict__ rand_action,
                                                              float* __restrict__ rand_log_prob, PolicyTrainOut t) {
  extern __shared__ __align__(16) float lds[];
  float* A = lds;
  float* B = lds + PT * p.ldA;
  float* P = B + PT * p.ldB;  // K-slice partial sums of the Dense in flight
  const int e0 = blockIdx.x * PT, tid = threadIdx.x;
  const int nrow = min(PT, batch - e0);

  constexpr bool TRAIN = MODE != 0;
  if (TRAIN) __builtin_amdgcn_s_setprio(3);  // (beside the value MLP's chip-filling GEMMs: this chain is the step's critical path)
  POL_STAMP(0);
  float *X = A, *Y = B;
  int ldx = p.ldA, ldy = p.ldB, K = p.traj_size;
  if (MODE == 2) {
    load_tile(t.encH[0] + (size_t)e0 * p.enc[0], nrow, p.enc[0], Y, ldy, nullptr, nullptr);
  } else {
    // traj tile -> A (rows beyond the batch are zero).  The tile's rows are one contiguous chunk of the input.
    load_tile(traj + (size_t)e0 * p.traj_size, nrow, p.traj_size, A, p.ldA, nullptr, nullptr);
  }
  __syncthreads();
  POL_STAMP(1);
  // ---- encoder: Dense -> ReLU -> LayerNorm   (intention_policy_network.py:29-41)
  for (int l = 0; l < p.n_enc; l++) {
    if (!(MODE == 2 && l == 0)) {
      dense_tile(X, ldx, K, params, p.enc_w[l], p.enc_b[l], p.enc[l], Y, ldy, true, P);
      POL_STAMP(2 + 2 * l);
      if (TRAIN) {
        store_tile(Y, ldy, nrow, p.enc[l], t.encH[l], e0);
        __syncthreads();  // (LayerNorm works in place, a wave per row)
      }
    }
    layer_norm_rows<TRAIN>(Y, ldy, p.enc[l], params + p.enc_g[l], params + p.enc_be[l], t.encS[l], e0, nrow);
    __syncthreads();
    if (TRAIN) store_tile(Y, ldy, nrow, p.enc[l], t.encY[l], e0);
    POL_STAMP(3 + 2 * l);
    float* t = X;
    X = Y, Y = t;
    int tl = ldx;
    ldx = ldy, ldy = tl;
    K = p.enc[l];
  }
  // ---- heads (ipn:42-44): mean -> Y[:, 0:latent], logvar -> Y[:, latent:2 latent]
  if ((p.latent & 63) == 0) {
    dense_tile(X, ldx, K, params, p.mean_w, p.mean_b, 2 * p.latent, Y, ldy, false, P, p.latent, p.lv_w, p.lv_b);
  } else {
    dense_tile(X, ldx, K, params, p.mean_w, p.mean_b, p.latent, Y, ldy, false, P);
    dense_tile(X, ldx, K, params, p.lv_w, p.lv_b, p.latent, Y + p.latent, ldy, false, P);
  }
  POL_STAMP(10);
  if (TRAIN) store_tile(Y, ldy, nrow, 2 * p.latent, t.ml, e0);
  // ---- z = mean + eps * exp(logvar / 2) (ipn:73-76); decoder input [z | normalised obs] -> X
  for (int i = tid; i < PT * p.latent; i += PNT) {
    int r = i / p.latent, c = i - r * p.latent;
    float mu = Y[r * ldy + c], lv = Y[r * ldy + p.latent + c];
    float z = 0.f;
    if (r < nrow) {
      size_t g = (size_t)(e0 + r) * p.latent + c;
      lat_mean[g] = mu, lat_logvar[g] = lv;
      z = mu + eps_latent[g] * expf(0.5f * lv);
    }
    X[r * ldx + c] = z;
  }
  // normalised obs (running_statistics.normalize; traj is NOT normalised)
  load_tile(obs + (size_t)e0 * p.obs_size, nrow, p.obs_size, X + p.latent, ldx, obs_mean, obs_std);
  __syncthreads();
  if (TRAIN) store_tile(X, ldx, nrow, p.latent + p.obs_size, t.D0, e0);
  POL_STAMP(11);
  // ---- decoder (ipn:56-70): [Dense -> ReLU -> LayerNorm] x (n-1), last Dense linear
  K = p.latent + p.obs_size;
  for (int l = 0; l < p.n_dec; l++) {
    bool last = l == p.n_dec - 1;
    dense_tile(X, ldx, K, params, p.dec_w[l], p.dec_b[l], p.dec[l], Y, ldy, !last, P);
    POL_STAMP(12 + 2 * l);
    if (!last) {
      if (TRAIN) {
        store_tile(Y, ldy, nrow, p.dec[l], t.decH[l], e0);
        __syncthreads();
      }
      layer_norm_rows<TRAIN>(Y, ldy, p.dec[l], params + p.dec_g[l], params + p.dec_be[l], t.decS[l], e0, nrow);
      __syncthreads();
      if (TRAIN) store_tile(Y, ldy, nrow, p.dec[l], t.decY[l], e0);
    }
    float* t = X;
    X = Y, Y = t;
    int tl = ldx;
    ldx = ldy, ldy = tl;
    K = p.dec[l];
  }
  POL_STAMP(20);
  // ---- distribution (brax NormalTanhDistribution; ppo_networks.py:56-83): X holds logits [loc | s]
  const int na = p.act_size;
  for (int i = tid; i < PT * 2 * na; i += PNT) {
    int r = i / (2 * na), c = i - r * 2 * na;
    if (r < nrow) logits[(size_t)(e0 + r) * 2 * na + c] = X[r * ldx + c];
  }
  if (TRAIN) return;  // (sampling belongs to acting; the loss head works from the logits)
  float* lp = Y;  // per-(env, action) log-prob terms
  for (int i = tid; i < PT * na; i += PNT) {
    int r = i / na, c = i - r * na;
    float term = 0.f;
    if (r < nrow) {
      size_t g = (size_t)(e0 + r) * na + c;
      float loc = X[r * ldx + c];
      if (deterministic) {
        action[g] = tanhf(loc);  // mode()
      } else {
        float scale = softplusf(X[r * ldx + na + c]) + 0.001f;
        float eps = eps_action[g];
        float raw = loc + scale * eps;
        raw_action[g] = raw, action[g] = tanhf(raw);
        // Normal log-pdf minus the tanh log-det-Jacobian 2 (log 2 - x - softplus(-2x))
        term = -0.5f * eps * eps - 0.9189385332046727f - logf(scale) - 2.f * (0.6931471805599453f - raw - softplusf(-2.f * raw));
        if (rand_action) {  // log-prob of ONE random pre-tanh action shared by all envs (ppo_networks.py:67-73)
          float x = rand_action[c], z = (x - loc) / scale;
          lp[r * ldy + na + c] = -0.5f * z * z - 0.9189385332046727f - logf(scale) - 2.f * (0.6931471805599453f - x - softplusf(-2.f * x));
        }
      }
    }
    lp[r * ldy + c] = term;
  }
  __syncthreads();
  if (!deterministic && tid < nrow) {
    float s = 0.f;
    for (int c = 0; c < na; c++) s += lp[tid * ldy + c];
    log_prob[e0 + tid] = s;
    if (rand_action) {
      float s2 = 0.f;
      for (int c = 0; c < na; c++) s2 += lp[tid * ldy + na + c];
      rand_log_prob[e0 + tid] = s2;
    }
  }
  POL_STAMP(21);
}

// ----------------------------------------------------------------------------- host side
void vnl_set_error_(const char* msg);  // vnl_lib.hip: feeds vnl_last_error()
static int pfail(int code, const char* msg) {
  vnl_set_error_(msg);
  return code;
}

struct vnl_policy {
  PolicyDev d;
  int device, max_batch;
  int64_t nparams;
  size_t lds_bytes;
  PolicyDev d2;       // MODE 2 (training form from the first Dense output on): no trajectory tile, both activation buffers narrow
  size_t lds_bytes2;
  int threads2;
};

// threads of the MODE 2 launch (256 / 512 / 1024) and the LDS it then needs: the partial-sum buffer shrinks with the wave count
// (tools/ppo_update_bench.py --fused-threads; measured per PPO minibatch step: 1024 -> 0.430 ms, 512 -> 0.451, 256 -> 0.458)
int vnl_policy_set_threads2_(vnl_policy* p, int threads) {
  if (!p || (threads != 256 && threads != 512 && threads != 1024)) return VNL_ERR_ARG;
  const PolicyDev& d = p->d;
  int pw2 = 0;
  bool ok2 = true;
  auto upd2 = [&](int n) {
    int G = (n + 63) / 64, Gp = 1;
    while (Gp < G && Gp < threads / 64) Gp <<= 1;
    if (G > Gp) ok2 = false;
    int w = (threads / 64 / Gp) * ((n + 3) & ~3);
    pw2 = w > pw2 ? w : pw2;
  };
  for (int l = 0; l < d.n_enc; l++) upd2(d.enc[l]);
  for (int l = 0; l < d.n_dec; l++) upd2(d.dec[l]);
  upd2(d.latent), upd2(2 * d.latent);
  if (!ok2) return VNL_ERR_UNSUPPORTED;
  p->threads2 = threads;
  p->lds_bytes2 = (size_t)PT * (2 * d.ldB + pw2) * sizeof(float);
  return VNL_OK;
}

extern "C" int vnl_policy_create(const vnl_policy_spec* s, int32_t max_batch, int32_t device, vnl_policy** out) {
  if (!s || !out) return pfail(VNL_ERR_ARG, "vnl_policy_create: null argument");
  if (s->num_encoder_layers < 1 || s->num_encoder_layers > 8 || s->num_decoder_layers < 1 || s->num_decoder_layers > 8)
    return pfail(VNL_ERR_UNSUPPORTED, "1..8 encoder and decoder layers supported");
  if (s->decoder_layers[s->num_decoder_layers - 1] != 2 * s->action_size)
    return pfail(VNL_ERR_ARG, "last decoder layer must be 2 * action_size (NormalTanh parameters)");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return pfail(VNL_ERR_NO_DEVICE, "no HIP device (no CPU fallback)");
  if (device < 0 || device >= ndev) return pfail(VNL_ERR_ARG, "device ordinal out of range");
  vnl_policy* p = new vnl_policy();
  PolicyDev& d = p->d;
  d.traj_size = s->traj_size, d.obs_size = s->obs_size, d.act_size = s->action_size, d.latent = s->latent_size;
  d.n_enc = s->num_encoder_layers, d.n_dec = s->num_decoder_layers;
  int off = 0, fan = d.traj_size;
  for (int l = 0; l < d.n_enc; l++) {  // order of IntentionNetwork.layout (intention_policy_network.py)
    int h = d.enc[l] = s->encoder_layers[l];
    d.enc_w[l] = off, off += fan * h;
    d.enc_b[l] = off, off += h;
    d.enc_g[l] = off, off += h;
    d.enc_be[l] = off, off += h;
    fan = h;
  }
  d.mean_w = off, off += fan * d.latent;
  d.mean_b = off, off += d.latent;
  d.lv_w = off, off += fan * d.latent;
  d.lv_b = off, off += d.latent;
  fan = d.latent + d.obs_size;
  for (int l = 0; l < d.n_dec; l++) {
    int h = d.dec[l] = s->decoder_layers[l];
    d.dec_w[l] = off, off += fan * h;
    d.dec_b[l] = off, off += h;
    if (l != d.n_dec - 1) {
      d.dec_g[l] = off, off += h;
      d.dec_be[l] = off, off += h;
    }
    fan = h;
  }
  p->nparams = off;
  // A holds the traj tile and every second activation, B the others; odd leading dimensions
  int wa = d.traj_size, wb = 2 * d.latent;
  bool toB = true;  // encoder layer 0 writes B
  auto put = [&](int w) {
    if (toB) wb = w > wb ? w : wb; else wa = w > wa ? w : wa;
    toB = !toB;
  };
  for (int l = 0; l < d.n_enc; l++) put(d.enc[l]);
  // heads write 2*latent into the "other" buffer, decoder input goes to the current X
  {
    int w_heads = 2 * d.latent, w_in = d.latent + d.obs_size;
    if (toB) wb = w_heads > wb ? w_heads : wb; else wa = w_heads > wa ? w_heads : wa;
    if (toB) wa = w_in > wa ? w_in : wa; else wb = w_in > wb ? w_in : wb;
  }
  for (int l = 0; l < d.n_dec; l++) put(d.dec[l]);
  wa = wa > wb ? wa : wb;  // keep it simple: both buffers sized for the widest activation except the traj tile
  int wide = 0;
  for (int l = 0; l < d.n_enc; l++) wide = d.enc[l] > wide ? d.enc[l] : wide;
  for (int l = 0; l < d.n_dec; l++) wide = d.dec[l] > wide ? d.dec[l] : wide;
  wide = wide > d.latent + d.obs_size ? wide : d.latent + d.obs_size;
  wide = wide > 2 * d.latent ? wide : 2 * d.latent;
  int la = d.traj_size > wide ? d.traj_size : wide;
  // ds_read_b32 banks are (address / 4) % 32 per 32-lane half; a half reads rows 0..15 x two k: ld = 2 mod 32 -> 32 banks
  auto pad = [](int w) { return ((w + 29) / 32) * 32 + 2; };
  d.ldA = pad(la), d.ldB = pad(wide);
  // partial sums of the widest Dense: (K slices) x 16 rows x N; slices x N <= 1024 whenever N <= 1024
  int pw = 0;
  auto part = [&](int n) {
    int G = (n + 63) / 64, Gp = 1;
    while (Gp < G && Gp < PTHREADS / 64) Gp <<= 1;
    int w = (PTHREADS / 64 / Gp) * ((n + 3) & ~3);
    if (G > Gp) w = -1;  // a wave would loop over column groups with one slice: needs N <= 1024
    return w;
  };
  bool too_wide = false;
  auto upd = [&](int n) {
    int w = part(n);
    if (w < 0) too_wide = true;
    pw = w > pw ? w : pw;
  };
  for (int l = 0; l < d.n_enc; l++) upd(d.enc[l]);
  for (int l = 0; l < d.n_dec; l++) upd(d.dec[l]);
  upd(d.latent), upd(2 * d.latent);
  if (too_wide) {
    delete p;
    return pfail(VNL_ERR_UNSUPPORTED, "layers wider than 1024 are not supported by the fused policy kernel");
  }
  p->lds_bytes = (size_t)PT * (d.ldA + d.ldB + pw) * sizeof(float);
  if (p->lds_bytes > 160 * 1024 - 512) {
    delete p;
    return pfail(VNL_ERR_UNSUPPORTED, "network too wide for the 16-env LDS tile");
  }
  p->d2 = d, p->d2.ldA = d.ldB;
  (void)vnl_policy_set_threads2_(p, PTHREADS);
  p->device = device, p->max_batch = max_batch;
  if (p->lds_bytes > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute((const void*)vnl_policy_kernel_t<0>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)p->lds_bytes);
    if (e == hipSuccess)
      e = hipFuncSetAttribute((const void*)vnl_policy_kernel_t<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)p->lds_bytes);
    if (e == hipSuccess)
      e = hipFuncSetAttribute((const void*)vnl_policy_kernel_t<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)p->lds_bytes2);
    if (e != hipSuccess) {
      delete p;
      return pfail(VNL_ERR_HIP, hipGetErrorString(e));
    }
  }
  *out = p;
  return VNL_OK;
}

extern "C" void vnl_policy_destroy(vnl_policy* p) { delete p; }
extern "C" int64_t vnl_policy_num_params(const vnl_policy* p) { return p ? p->nparams : -1; }

extern "C" int vnl_policy_forward(vnl_policy* p, const float* params, const float* obs_mean, const float* obs_std,
                                  const float* traj, const float* obs, const float* eps_latent, const float* eps_action,
                                  int32_t batch, int32_t deterministic, float* action, float* raw_action, float* log_prob,
                                  float* logits, float* latent_mean, float* latent_logvar, const float* rand_action,
                                  float* rand_log_prob, void* stream) {
  if (!p || !params || !traj || !obs || !eps_latent || !action || !logits || !latent_mean || !latent_logvar)
    return pfail(VNL_ERR_ARG, "vnl_policy_forward: null argument");
  if (!deterministic && (!eps_action || !raw_action || !log_prob))
    return pfail(VNL_ERR_ARG, "vnl_policy_forward: stochastic mode needs eps_action, raw_action, log_prob");
  if ((rand_action == nullptr) != (rand_log_prob == nullptr) || (rand_action && deterministic))
    return pfail(VNL_ERR_ARG, "rand_action / rand_log_prob: both or neither, stochastic mode only");
  if ((obs_mean == nullptr) != (obs_std == nullptr)) return pfail(VNL_ERR_ARG, "obs_mean / obs_std must both be given or both null");
  if (batch <= 0 || batch > p->max_batch) return pfail(VNL_ERR_ARG, "batch out of range");
  int grid = (batch + PT - 1) / PT;
  hipLaunchKernelGGL(vnl_policy_kernel_t<0>, dim3(grid), dim3(PTHREADS), p->lds_bytes, (hipStream_t)stream, p->d, params,
                     obs_mean, obs_std, traj, obs, eps_latent, eps_action, (int)batch, (int)deterministic, action,
                     raw_action, log_prob, logits, latent_mean, latent_logvar, rand_action, rand_log_prob, PolicyTrainOut{});
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return pfail(VNL_ERR_HIP, hipGetErrorString(e));
  return VNL_OK;
}

// The same kernel in its TRAINING form (csrc/vnl_ppo.hip: the intention network's forward pass of a PPO minibatch step):
// no sampling; besides the logits and the latent heads it leaves every intermediate the backward pass reads.
int vnl_policy_forward_train_(vnl_policy* p, const float* params, const float* obs_mean, const float* obs_std, const float* traj,
                              const float* obs, const float* eps_latent, int32_t batch, float* logits, float* latent_mean,
                              float* latent_logvar, const PolicyTrainOut* out, int from_first_dense, void* stream) {
  if (!p || !params || !traj || !obs || !eps_latent || !logits || !latent_mean || !latent_logvar || !out)
    return pfail(VNL_ERR_ARG, "vnl_policy_forward_train_: null argument");
  if (batch <= 0 || batch > p->max_batch) return pfail(VNL_ERR_ARG, "batch out of range");
  const int grid = (batch + PT - 1) / PT;
  if (from_first_dense)
    hipLaunchKernelGGL(vnl_policy_kernel_t<2>, dim3(grid), dim3(p->threads2), p->lds_bytes2, (hipStream_t)stream, p->d2, params,
                       obs_mean, obs_std, traj, obs, eps_latent, (const float*)nullptr, (int)batch, 1, (float*)nullptr, (float*)nullptr,
                       (float*)nullptr, logits, latent_mean, latent_logvar, (const float*)nullptr, (float*)nullptr, *out);
  else
  hipLaunchKernelGGL(vnl_policy_kernel_t<1>, dim3(grid), dim3(PTHREADS), p->lds_bytes, (hipStream_t)stream, p->d, params,
                     obs_mean, obs_std, traj, obs, eps_latent, (const float*)nullptr, (int)batch, 1, (float*)nullptr, (float*)nullptr,
                     (float*)nullptr, logits, latent_mean, latent_logvar, (const float*)nullptr, (float*)nullptr, *out);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return pfail(VNL_ERR_HIP, hipGetErrorString(e));
  return VNL_OK;
}
