// Shared by csrc/vnl_policy.hip (the fused intention-network kernel) and csrc/vnl_ppo.hip (the PPO minibatch step, whose
// forward pass of the intention network is that same kernel in its training form).  Internal to the library: not part of
// the C-ABI of include/vnl.h.
#pragma once
#include <stdint.h>

struct PolicyDev {
  int traj_size, obs_size, act_size, latent;
  int n_enc, n_dec;
  int enc[8], dec[8];
  // parameter offsets (floats) inside the flat buffer
  int enc_w[8], enc_b[8], enc_g[8], enc_be[8];
  int mean_w, mean_b, lv_w, lv_b;
  int dec_w[8], dec_b[8], dec_g[8], dec_be[8];
  int ldA, ldB;  // leading dimensions (= 2 mod 32: the A-fragment ds_read_b32 is conflict-free) of the two LDS activation buffers
};

// What the backward pass of the PPO step needs from the forward pass (reference intention_policy_network.py:20-105 under
// jax.grad): per hidden layer the Dense output after ReLU (H), the LayerNorm row statistics (S: mean | 1 / sqrt(var + eps))
// and the LayerNorm output (Y, the next layer's input); the two latent heads side by side; the decoder's input.
struct PolicyTrainOut {
  float *encH[8], *encS[8], *encY[8];
  float *decH[8], *decS[8], *decY[8];
  float* ml;  // [N][2 latent]: mean | logvar
  float* D0;  // [N][latent + obs]: z | normalised observation
};

struct vnl_policy;
int vnl_policy_set_threads2_(vnl_policy* p, int threads);
int vnl_policy_forward_train_(vnl_policy* p, const float* params, const float* obs_mean, const float* obs_std, const float* traj,
                              const float* obs, const float* eps_latent, int32_t batch, float* logits, float* latent_mean,
                              float* latent_logvar, const PolicyTrainOut* out, int from_first_dense /* encH[0] is given (a GEMM made it): start at its LayerNorm */,
                              void* stream);
