// Fused intention-policy inference for gfx950 (C-ABI: vnl_policy_* in include/vnl.h).
//
// Replaces, for the acting path, reference ppo_imitation/ppo_networks.py:45-83 (make_inference_fn.policy)
// and intention_policy_network.py:20-105 (Encoder -> reparameterize -> Decoder), plus the tanh-Normal
// sampling / log-prob of brax NormalTanhDistribution [UPSTREAM].
//
// One 256-thread workgroup owns a tile of 32 envs for the whole network:
//   traj tile (32 x 795) -> LDS -> [Dense+ReLU+LayerNorm] x len(encoder) -> fc2_mean / fc2_logvar
//   -> z = mean + eps * exp(logvar / 2) -> [z | (obs - mu) / sigma] -> decoder -> logits (32 x 60)
//   -> scale = softplus(s) + 1e-3, raw = loc + scale * eps, action = tanh(raw), log_prob.
// Every Dense is a sequence of v_mfma_f32_32x32x2_f32 (exact fp32, the reference's precision): the four
// waves split the 32-column output tiles, A fragments come from LDS (odd leading dimension -> no bank
// conflicts), B fragments stream from the L2-resident weight buffer (row-major (in, out), as Flax).
// Activations never leave LDS; the only HBM traffic is the inputs, the weights and the outputs.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/vnl.h"

#define PT 32          /* envs per workgroup */
#define PTHREADS 1024  /* 16 waves: four per SIMD, some compute while the others wait for their weight loads */
#define LN_EPS 1e-6f   /* flax.linen.LayerNorm default */

typedef float v16f __attribute__((ext_vector_type(16)));

struct PolicyDev {
  int traj_size, obs_size, act_size, latent;
  int n_enc, n_dec;
  int enc[8], dec[8];
  // parameter offsets (floats) inside the flat buffer
  int enc_w[8], enc_b[8], enc_g[8], enc_be[8];
  int mean_w, mean_b, lv_w, lv_b;
  int dec_w[8], dec_b[8], dec_g[8], dec_be[8];
  int ldA, ldB;  // leading dimensions (odd) of the two LDS activation buffers
};

// Y[32 x N] (LDS, ld = ldy) = X[32 x K] (LDS, ld = ldx) @ W[K x N] (global) + b, optional ReLU.
// Wave w takes column tiles w, w+4, ...  Fragment layout of v_mfma_f32_32x32x2f32:
//   A: lane l holds X[l % 32][k0 + l / 32];  B: lane l holds W[k0 + l / 32][n0 + l % 32];
//   D: lane l, register i holds Y[(i / 4) * 8 + (l / 32) * 4 + i % 4][n0 + l % 32].
__device__ __forceinline__ void dense_tile(const float* X, int ldx, int K, const float* __restrict__ W,
                                           const float* __restrict__ bias, int N, float* Y, int ldy, bool relu) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int row = lane & 31, kh = lane >> 5;
  const int ntiles = (N + 31) / 32;
  for (int tile = wave; tile < ntiles; tile += PTHREADS / 64) {
    const int col = tile * 32 + row;
    const bool cok = col < N;
    v16f acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const float* xr = X + row * ldx + kh;
    const float* wc = W + (size_t)kh * N + (cok ? col : 0);
    int k0 = 0;
    for (; k0 + 16 <= K; k0 += 16) {  // 8 MFMAs per trip; the 8 weight loads are issued up front
      float b[8], a[8];
#pragma unroll
      for (int u = 0; u < 8; u++) b[u] = cok ? wc[(size_t)(k0 + 2 * u) * N] : 0.f;
#pragma unroll
      for (int u = 0; u < 8; u++) a[u] = xr[k0 + 2 * u];
#pragma unroll
      for (int u = 0; u < 8; u++) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], b[u], acc, 0, 0, 0);
    }
    for (; k0 < K; k0 += 2) {
      bool kok = k0 + kh < K;
      float a = kok ? xr[k0] : 0.f;
      float b = (kok && cok) ? wc[(size_t)k0 * N] : 0.f;
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    }
    if (cok) {
      float bv = bias[col];
#pragma unroll
      for (int i = 0; i < 16; i++) {
        int r = (i / 4) * 8 + kh * 4 + (i % 4);
        float v = acc[i] + bv;
        Y[r * ldy + col] = relu ? fmaxf(v, 0.f) : v;
      }
    }
  }
}

// in-place LayerNorm over the N columns of each of the 32 rows (wave w: rows 8w .. 8w+7)
__device__ __forceinline__ void layer_norm_rows(float* Y, int ldy, int N, const float* __restrict__ g,
                                                const float* __restrict__ be) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  constexpr int RPW = PT / (PTHREADS / 64);  // rows per wave
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
    float inv = rsqrtf(var + LN_EPS);
    for (int c = lane; c < N; c += 64) y[c] = (y[c] - mean) * inv * g[c] + be[c];
  }
}

__device__ __forceinline__ float softplusf(float x) { return x > 20.f ? x : log1pf(expf(x)); }

__global__ void __launch_bounds__(PTHREADS) vnl_policy_kernel(PolicyDev p, const float* __restrict__ params,
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
                                                              float* __restrict__ rand_log_prob) {
  extern __shared__ __align__(16) float lds[];
  float* A = lds;
  float* B = lds + PT * p.ldA;
  const int e0 = blockIdx.x * PT, tid = threadIdx.x;
  const int nrow = min(PT, batch - e0);

  // traj tile -> A (rows beyond the batch are zero)
  for (int i = tid; i < PT * p.traj_size; i += PTHREADS) {
    int r = i / p.traj_size, c = i - r * p.traj_size;
    A[r * p.ldA + c] = r < nrow ? traj[(size_t)(e0 + r) * p.traj_size + c] : 0.f;
  }
  __syncthreads();
  // ---- encoder: Dense -> ReLU -> LayerNorm   (intention_policy_network.py:29-41)
  float *X = A, *Y = B;
  int ldx = p.ldA, ldy = p.ldB, K = p.traj_size;
  for (int l = 0; l < p.n_enc; l++) {
    dense_tile(X, ldx, K, params + p.enc_w[l], params + p.enc_b[l], p.enc[l], Y, ldy, true);
    __syncthreads();
    layer_norm_rows(Y, ldy, p.enc[l], params + p.enc_g[l], params + p.enc_be[l]);
    __syncthreads();
    float* t = X;
    X = Y, Y = t;
    int tl = ldx;
    ldx = ldy, ldy = tl;
    K = p.enc[l];
  }
  // ---- heads (ipn:42-44): mean -> Y[:, 0:latent], logvar -> Y[:, latent:2 latent]
  dense_tile(X, ldx, K, params + p.mean_w, params + p.mean_b, p.latent, Y, ldy, false);
  dense_tile(X, ldx, K, params + p.lv_w, params + p.lv_b, p.latent, Y + p.latent, ldy, false);
  __syncthreads();
  // ---- z = mean + eps * exp(logvar / 2) (ipn:73-76); decoder input [z | normalised obs] -> X
  for (int i = tid; i < PT * p.latent; i += PTHREADS) {
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
  for (int i = tid; i < PT * p.obs_size; i += PTHREADS) {
    int r = i / p.obs_size, c = i - r * p.obs_size;
    float v = 0.f;
    if (r < nrow) {
      v = obs[(size_t)(e0 + r) * p.obs_size + c];
      if (obs_mean) v = (v - obs_mean[c]) / obs_std[c];  // running_statistics.normalize; traj is NOT normalised
    }
    X[r * ldx + p.latent + c] = v;
  }
  __syncthreads();
  // ---- decoder (ipn:56-70): [Dense -> ReLU -> LayerNorm] x (n-1), last Dense linear
  K = p.latent + p.obs_size;
  for (int l = 0; l < p.n_dec; l++) {
    bool last = l == p.n_dec - 1;
    dense_tile(X, ldx, K, params + p.dec_w[l], params + p.dec_b[l], p.dec[l], Y, ldy, !last);
    __syncthreads();
    if (!last) {
      layer_norm_rows(Y, ldy, p.dec[l], params + p.dec_g[l], params + p.dec_be[l]);
      __syncthreads();
    }
    float* t = X;
    X = Y, Y = t;
    int tl = ldx;
    ldx = ldy, ldy = tl;
    K = p.dec[l];
  }
  // ---- distribution (brax NormalTanhDistribution; ppo_networks.py:56-83): X holds logits [loc | s]
  const int na = p.act_size;
  for (int i = tid; i < PT * 2 * na; i += PTHREADS) {
    int r = i / (2 * na), c = i - r * 2 * na;
    if (r < nrow) logits[(size_t)(e0 + r) * 2 * na + c] = X[r * ldx + c];
  }
  float* lp = Y;  // per-(env, action) log-prob terms
  for (int i = tid; i < PT * na; i += PTHREADS) {
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
};

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
  d.ldA = la | 1, d.ldB = wide | 1;
  p->lds_bytes = (size_t)PT * (d.ldA + d.ldB) * sizeof(float);
  if (p->lds_bytes > 160 * 1024 - 512) {
    delete p;
    return pfail(VNL_ERR_UNSUPPORTED, "network too wide for the 32-env LDS tile");
  }
  p->device = device, p->max_batch = max_batch;
  if (p->lds_bytes > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute((const void*)vnl_policy_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)p->lds_bytes);
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
  hipLaunchKernelGGL(vnl_policy_kernel, dim3(grid), dim3(PTHREADS), p->lds_bytes, (hipStream_t)stream, p->d, params,
                     obs_mean, obs_std, traj, obs, eps_latent, eps_action, (int)batch, (int)deterministic, action,
                     raw_action, log_prob, logits, latent_mean, latent_logvar, rand_action, rand_log_prob);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return pfail(VNL_ERR_HIP, hipGetErrorString(e));
  return VNL_OK;
}
