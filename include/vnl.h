/*
 * vnl.h -- C-ABI of the MI355X-native rodent-imitation rollout.
 *
 * The reference (talmolab/VNL-Brax-Imitation) is pure Python on JAX/MJX/Brax and
 * has no FFI of its own; this ABI is the boundary a maintainer binds (ctypes, see
 * INTEGRATION.md) to replace, for the hot path only:
 *
 *   vnl_env_reset  <->  RodentTracking.reset          reference envs/rodent.py:119-176
 *                       (+ brax PipelineEnv.pipeline_init = mjx.forward, rodent.py:148)
 *   vnl_env_step   <->  RodentTracking.step           reference envs/rodent.py:178-239
 *                       (+ PipelineEnv.pipeline_step = n_frames x mjx.step, rodent.py:181)
 *   vnl_policy_*   <->  make_inference_fn(...).policy reference ppo_imitation/ppo_networks.py:45-83
 *                       IntentionNetwork.__call__     reference ppo_imitation/intention_policy_network.py:91-105
 *   vnl_rollout_post <-> brax Episode/AutoReset wrappers (train.py:204-214) + Transition of actor_step (acting.py:34-57)
 *   vnl_gather_rows / vnl_ppo_head / vnl_adam_step
 *                  <->  the minibatch gather, loss head (intention_losses.py:26-87,131-202) and optimiser update of one
 *                       PPO minibatch step (ppo_imitation/train.py:231-291)
 *
 * Conventions
 *   - plain pointers and sizes only; no torch / C++ types cross this boundary;
 *   - every state / observation / action buffer is CALLER-OWNED device memory,
 *     float32 (int32 for frame counters), row-major [env][feature] (what a
 *     torch (B, n) tensor is).  One 64-lane wavefront owns one env, so lane i
 *     touches element env*n + i: contiguous per wave, and obs/traj come out in
 *     the layout the policy GEMMs consume;
 *   - the library owns model constants, the clip copy and per-env scratch,
 *     released by vnl_env_destroy / vnl_model_destroy;
 *   - all work is enqueued asynchronously on the caller's HIP stream (passed as
 *     void* = hipStream_t); no hidden synchronisation; a handle is not
 *     thread-safe (one handle per GPU / process);
 *   - return 0 on success, negative error code otherwise; message through
 *     vnl_last_error() (thread-local).  Nothing throws across the ABI.
 *   - there is NO CPU fallback: without a HIP device every compute entry point
 *     fails with VNL_ERR_NO_DEVICE.
 */
#ifndef VNL_H_
#define VNL_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VNL_OK 0
#define VNL_ERR_ARG -1
#define VNL_ERR_BLOB -2
#define VNL_ERR_HIP -3
#define VNL_ERR_NO_DEVICE -4
#define VNL_ERR_UNSUPPORTED -5

typedef struct vnl_model vnl_model;
typedef struct vnl_env vnl_env;
typedef struct vnl_policy vnl_policy;

/* Environment description: mirrors the constructor state of RodentTracking
 * (reference envs/rodent.py:17-117).  Index arrays are the EFFECTIVE indices
 * after the reference's (buggy) use of MuJoCo ids on filtered clip axes and
 * JAX's clamp-on-gather semantics; the Python
 * mirror derives them from names exactly as the reference does. */
typedef struct vnl_envspec {
  int32_t clip_frames;     /* T: frames per clip */
  int32_t num_clips;       /* C: clips resident (1 for the single-clip env) */
  int32_t ref_traj_length; /* rodent.py:34 */
  int32_t sub_clip_length; /* rodent.py:33 */
  int32_t n_frames;        /* physics substeps per control step, rodent.py:97-99 */
  int32_t num_track_bodies;/* width of the filtered clip.body_positions, rodent.py:113-115 */
  int32_t num_end_eff;
  int32_t num_appendages;
  int32_t num_joint_cols;
  int32_t com_ref_col;
  const int32_t* body_idxs;   /* [num_track_bodies] model body ids, rodent.py:80-85 */
  const int32_t* end_eff_idx; /* [num_end_eff] model body ids, rodent.py:65-70 */
  const int32_t* app_body;    /* [num_appendages] model body ids, rodent.py:71-76,307 */
  const int32_t* app_ref_col; /* [num_appendages] clamped columns of the filtered clip axis */
  const int32_t* joint_cols;  /* [num_joint_cols] clamped columns of the (nq-7)-wide joint axis */
  float healthy_z_lo, healthy_z_hi; /* rodent.py:31 */
  float termination_threshold;      /* rodent.py:35 */
  float body_error_multiplier;      /* rodent.py:36 */
  /* clip tensors, HOST pointers, float32, row-major (C, T, n); copied at create.
   * Fields of ReferenceClip, reference preprocessing/mjx_preprocess.py:21-40. */
  const float* position;         /* (C,T,3) */
  const float* quaternion;       /* (C,T,4) */
  const float* joints;           /* (C,T,nq-7) */
  const float* body_positions;   /* (C,T,num_track_bodies,3), already filtered */
  const float* velocity;         /* (C,T,3) */
  const float* angular_velocity; /* (C,T,3) */
  const float* joints_velocity;  /* (C,T,nq-7) */
  /* Which tracking env's glue.  0 = RodentTracking (envs/rodent.py:178-316).  HumanoidTracking (envs/humanoid.py:185-311)
   * = VNL_ENV_REWARD_OLD_STATE | VNL_ENV_TERM_MEAN | VNL_ENV_NO_RAPP | VNL_ENV_OBS_QPOS_QVEL with done_threshold 0.5. */
  int32_t flags;
  float done_threshold;          /* done when the UNSCALED rtrunk is below it: 0 (rodent.py:213), 0.5 (humanoid.py:199) */
  const float* center_of_mass;   /* (C,T,3) reference for rcom (humanoid.py:273), or NULL: body_positions[com_ref_col] (rodent.py:279) */
  /* with VNL_ENV_WEIGHTS: weights of rcom, rvel, rtrunk, rquat, ract, rapp in the total reward (ant.py:182-188:
   * 0.05, 0.01, 0.20, 0.01, 0.001, -); without it 0.01, 0.01, 0.01, 0.01, 0.0001, 0.01 (rodent.py:203-209) */
  float reward_weights[6];
} vnl_envspec;
#define VNL_ENV_REWARD_OLD_STATE 1 /* reward terms from the state BEFORE the step (humanoid.py:195: _calculate_reward(state, ..)) */
#define VNL_ENV_TERM_MEAN 2        /* termination error = means of |.| (humanoid.py:256-260), not the matrix-1 / L1 norms */
#define VNL_ENV_NO_RAPP 4          /* no appendage reward term */
#define VNL_ENV_OBS_QPOS_QVEL 8    /* observation = [qpos, qvel] only (humanoid.py:354-368) */
/* AntTracking (envs/ant.py:172-291) = the four above | the four below, done_threshold 0 */
#define VNL_ENV_WEIGHTS 16         /* reward_weights[] replace the built-in weights */
#define VNL_ENV_RACT_ACTION 32     /* ract = 0.01 * -0.015 * sum(action^2) / nu (ant.py:277), not the actuator forces */
#define VNL_ENV_METRICS_UNSCALED 64 /* metrics / termination_error hold the UNWEIGHTED terms (ant.py:197,216-225) */
#define VNL_ENV_TRAJ_OLD_FRAME 128 /* the step's reference slice starts at the OLD cur_frame + 1: ant.py:178 builds the
                                    * observation from the un-incremented info */

/* Caller-owned device buffers, row-major [num_envs][count]. */
typedef struct vnl_state {
  /* brax State.pipeline_state (mjx.Data) -- carried */
  float* qpos;           /* [nq] */
  float* qvel;           /* [nv] */
  float* act;            /* [nu] */
  float* qacc_warmstart; /* [nv] */
  /* derived by the last mjx.forward (lag qpos by one substep) */
  float* xpos;           /* [3*nbody] */
  float* xquat;          /* [4*nbody] */
  float* subtree_com1;   /* [3]  = data.subtree_com[1] */
  float* qfrc_actuator;  /* [nv] */
  /* brax State.{obs,reward,done,metrics} */
  float* obs;     /* [obs_size]  */
  float* reward;  /* [1] */
  float* done;    /* [1] */
  float* metrics; /* [7]: rcom rvel rtrunk rquat ract rapp termination_error */
  /* brax State.info */
  float* traj;              /* [traj_size] */
  float* termination_error; /* [1] */
  int32_t* cur_frame;       /* [1] */
  int32_t* sub_clip_frame;  /* [1] */
  int32_t* clip_id;         /* [1] which resident clip this env tracks (0 for single clip) */
} vnl_state;

/* Model dimensions, for sizing caller buffers. */
typedef struct vnl_dims {
  int32_t nq, nv, nu, nbody, njnt, ngeom_collide, ncon, nefc, obs_size, traj_size;
  int32_t workspace_floats_per_env; /* per-env LDS working set, in floats */
  int32_t workgroups_per_cu;        /* occupancy of the step kernel as reported by the HIP runtime */
  int32_t nbody_dynamic;            /* bodies the dynamics works on: the model's welded (jointless) bodies are folded into their
                                       parents; every one of the nbody bodies still has its row of xpos / xquat */
  int32_t kernel_specialised;       /* 1: the env runs the kernels specialised at compile time for its dimensions (the reference's
                                       rodent); 0: the generic kernels */
} vnl_dims;

const char* vnl_last_error(void);
int vnl_version(void);

/* model: blob produced by vnl_brax_imitation_amd.model.blob.to_blob (host memory) */
int vnl_model_create(const void* blob, size_t nbytes, vnl_model** out);
void vnl_model_destroy(vnl_model*);

/* env: device = HIP ordinal (>= 0) */
int vnl_env_create(const vnl_model*, const vnl_envspec*, int32_t num_envs, int32_t device, vnl_env** out);
void vnl_env_destroy(vnl_env*);
int vnl_env_dims(const vnl_env*, vnl_dims* out);

/* reset: start_frame [num_envs] int32, clip_id written by caller into state->clip_id,
 * noise [num_envs][nq] (already scaled by reset_noise_scale; rodent.py:131-147). */
int vnl_env_reset(vnl_env*, const int32_t* start_frame, const float* noise, const vnl_state* state, void* stream);

/* step: action [num_envs][nu]; state updated in place. */
int vnl_env_step(vnl_env*, const float* action, const vnl_state* state, void* stream);
/* Forward kinematics only (reference preprocessing/mjx_preprocess.py:85-107: `mjx.kinematics` scanned over the frames of a
 * clip): env e takes row e of qpos [num_envs][nq] and fills state->xpos, xquat, subtree_com1 and the
 * normalised root quaternion in state->qpos[e][3..7); nothing else of the state is touched. */
int vnl_env_fk(vnl_env* env, const float* qpos, const vnl_state* state, void* stream);

/* Bisection hooks.  The per-env working set lives in LDS; with debug on, every reset/step
 * also copies it to a device dump [num_envs][row_stride]: enable = 1 at the end of the kernel,
 * enable = 2 (step only) as the LAST forward pass of the step leaves it, i.e. before the Euler
 * update reuses the space of the constraint rows; 0 = off.  vnl_env_scratch returns the device
 * pointer of a named section inside row 0 ("qLD", "qfrc_smooth", "qacc_smooth", "qacc",
 * "efc_D", "Jaref", "qfrc_constraint", ...) and its element count; env e is at +e*row_stride.
 * The section "solver_trace" is separate from the image: int32 [num_envs][n_frames][536], the
 * discrete decisions (warm start, iteration counts, line-search bracket decisions, active-row
 * counts) of the solver call of every substep, layout in csrc/vnl_types.h (VNL_TRACE_*). */
int vnl_env_debug(vnl_env*, int32_t enable, int32_t* row_stride);
int vnl_env_scratch(const vnl_env*, const char* name, float** dev_ptr, int32_t* count);

/* ---- policy forward (intention network), ppo_networks.py:45-83 ----------------
 * params: flat float32 device buffer in the order documented in INTEGRATION.md.
 * Inputs traj/obs row-major as produced by vnl_env_step; normaliser mean/std [obs_size].
 * eps_latent [B][latent] and eps_action [B][act] are caller-supplied N(0,1) draws
 * (the JAX threefry stream is not reproduced).  Outputs row-major. */
typedef struct vnl_policy_spec {
  int32_t traj_size, obs_size, action_size, latent_size;
  int32_t num_encoder_layers, num_decoder_layers;
  int32_t encoder_layers[8], decoder_layers[8];
} vnl_policy_spec;

int vnl_policy_create(const vnl_policy_spec*, int32_t max_batch, int32_t device, vnl_policy** out);
void vnl_policy_destroy(vnl_policy*);
int64_t vnl_policy_num_params(const vnl_policy*);
int vnl_policy_forward(vnl_policy*, const float* params, const float* obs_mean, const float* obs_std,
                       const float* traj, const float* obs, const float* eps_latent, const float* eps_action,
                       int32_t batch, int32_t deterministic, float* action, float* raw_action, float* log_prob,
                       float* logits, float* latent_mean, float* latent_logvar,
                       const float* rand_action /* [act] pre-tanh action or NULL */,
                       float* rand_log_prob /* [batch] its log-prob under every env's distribution
                                               (the reference's policy extras, ppo_networks.py:67-73), or NULL */,
                       void* stream);

/* ---- rollout post-processing: the training wrappers and the Transition logging in ONE launch ----
 * brax EpisodeWrapper + AutoResetWrapper as applied by the reference's wrap_for_training
 * (ppo_imitation/train.py:204-214) and the per-step Transition of actor_step
 * (ppo_imitation/acting.py:34-57).  Per env e, in this order:
 *   steps = (prev_done ? 0 : steps) + action_repeat;  over = steps >= episode_length
 *   truncation = over ? 1 - done : 0;  done = over ? 1 : done;  prev_done = done
 *   log_reward = reward;  log_discount = 1 - done;  log_truncation = truncation   (each optional)
 *   every op:  v = (first && done) ? first[e] : src[e];  dst[e] = v (if dst);  log[e] = v (if log)
 * All arrays are row-major [num_envs][width] device buffers of 32-bit words (float32; int32 frame
 * counters are moved bit-exactly).  dst may alias src. */
#define VNL_POST_MAX_OPS 24
typedef struct vnl_post_op {
  float* dst;
  const float* src;
  const float* first;
  float* log;
  int32_t width;
  int32_t pad_;
} vnl_post_op;
typedef struct vnl_post_desc {
  float *steps, *prev_done, *done, *truncation;
  const float* reward;
  float *log_reward, *log_discount, *log_truncation;
  int32_t episode_length, action_repeat, num_ops, pad_;
  vnl_post_op ops[VNL_POST_MAX_OPS];
} vnl_post_desc;
int vnl_rollout_post(const vnl_post_desc*, int32_t num_envs, void* stream);

/* ---- PPO loss head: everything of compute_ppo_intention_loss after the network forward passes ----
 * (reference ppo_imitation/intention_losses.py:26-87 compute_gae, :131-202 loss) in three small launches: GAE,
 * advantage normalisation, tanh-Normal log-prob and entropy, clipped surrogate, value loss, latent KL,
 * their sum, AND the gradients of that sum w.r.t. the four network outputs.  Arrays are time-major
 * [T][B] (index t*B + b) float32 device buffers; logits [T*B][2*act], latents [T*B][latent]. */
typedef struct vnl_ppo_head_args {
  int32_t T, B, act, latent;
  const float *logits, *baseline, *bootstrap, *lat_mean, *lat_logvar;       /* network outputs */
  const float *raw_action, *behaviour_log_prob, *reward, *truncation, *discount, *eps_entropy;
  float entropy_cost, discounting, reward_scaling, gae_lambda, clipping_epsilon, kl_weight, min_std, var_scale;
  int32_t normalize_advantage, pad_;
  float *g_logits, *g_baseline, *g_lat_mean, *g_lat_logvar; /* d total_loss / d (network outputs) */
  float *vs, *advantages;                                   /* [T*B] GAE outputs (advantages before normalisation) */
  float *metrics; /* [8]: total_loss, policy_loss, v_loss, entropy_loss, kl_loss_intention, explained_variance,
                   * advantage mean, advantage std (before normalisation) */
} vnl_ppo_head_args;
/* workspace: 4 + 4 * 256 floats of device memory (statistics and per-block partial sums) */
#define VNL_PPO_HEAD_WORKSPACE_FLOATS (4 + 4 * 256)
int vnl_ppo_head(const vnl_ppo_head_args*, float* workspace, void* stream);

/* ---- minibatch gather: dst_k[t][j][:] = src_k[t][idx[j]][:] for every array k of a time-major Transition
 * (the index_select of brax's sgd_step, reference ppo_imitation/train.py:270-291), one launch for all arrays.
 * src_k is [T_k][N][width_k], dst_k [T_k][M][width_k], 32-bit words; idx [M] int64 on the device. */
typedef struct vnl_gather_op {
  float* dst;
  const float* src;
  int32_t T, width;
} vnl_gather_op;
typedef struct vnl_gather_desc {
  const int64_t* idx;
  int32_t N, M, num_ops, pad_;
  vnl_gather_op ops[VNL_POST_MAX_OPS];
} vnl_gather_desc;
int vnl_gather_rows(const vnl_gather_desc*, void* stream);

/* ---- Adam on one flat buffer (optax.adam as the reference builds it, ppo_imitation/train.py:231-233:
 * b1 0.9, b2 0.999, eps 1e-8, no weight decay): mu, nu, params updated in place in ONE launch.
 * `count` is the device-resident step number AFTER this step (the caller increments it first), so that
 * the launch can sit inside a captured hipGraph. */
int vnl_adam_step(float* params, const float* grads, float* mu, float* nu, const int64_t* count, int64_t n,
                  double lr, double b1, double b2, double eps, void* stream); /* hyper-parameters as the Python doubles they are:
                  1 - b2 is formed in double before the cast to float32, as optax does */

/* ---- PPO minibatch step, forward AND backward: the gradient of compute_ppo_intention_loss (reference
 * ppo_imitation/intention_losses.py:91-202) w.r.t. the policy and value parameters, i.e. what jax.grad computes inside
 * brax's gradient_update_fn for ppo_imitation/train.py:255-268.  Networks: the intention policy
 * (intention_policy_network.py:20-105; encoder / decoder of Dense -> ReLU -> LayerNorm, heads fc2_mean / fc2_logvar,
 * z = mean + eps exp(logvar / 2), decoder on [z | normalised obs], last decoder layer linear) and the value MLP
 * (ppo_networks.py:114-118, swish, scalar output).  `params` / `grads`: flat float32 [policy | value] in the layout of
 * INTEGRATION.md (Flax tensor order); grads is overwritten.  Minibatch arrays are time-major (index t*B + b).
 * Everything is enqueued on `stream`; no host synchronisation (capturable in a hipGraph). */
typedef struct vnl_ppo_net_spec {
  int32_t traj_size, obs_size, action_size, latent_size;
  int32_t num_encoder_layers, num_decoder_layers /* incl. the output layer of 2*action_size */, num_value_layers /* hidden */;
  int32_t encoder_layers[8], decoder_layers[8], value_layers[8];
} vnl_ppo_net_spec;
typedef struct vnl_ppo_batch {
  const float *traj;          /* [T*B][traj_size]   Transition.extras.state_extras.traj */
  const float *obs;           /* [T*B][obs_size]    Transition.observation (raw, not normalised) */
  const float *next_obs_last; /* [B][obs_size]      Transition.next_observation[-1] (bootstrap) */
  const float *raw_action, *behaviour_log_prob, *reward, *truncation, *discount;
  const float *eps_latent;    /* [T*B][latent]  N(0,1) draws of the reparameterisation */
  const float *eps_entropy;   /* [T*B][act]     N(0,1) draws of the entropy estimate */
  const float *obs_mean, *obs_std; /* [obs_size] running-statistics normaliser, or both NULL (identity) */
} vnl_ppo_batch;
typedef struct vnl_ppo_hparams {
  float entropy_cost, discounting, reward_scaling, gae_lambda, clipping_epsilon, kl_weight, min_std, var_scale;
  int32_t normalize_advantage, pad_;
} vnl_ppo_hparams;
typedef struct vnl_ppo_update vnl_ppo_update;
int vnl_ppo_update_create(const vnl_ppo_net_spec*, int32_t T, int32_t B, int32_t device, vnl_ppo_update** out);
void vnl_ppo_update_destroy(vnl_ppo_update*);
int64_t vnl_ppo_update_num_params(const vnl_ppo_update*);
/* device pointer of an intermediate of the LAST vnl_ppo_minibatch_grad call (valid until the next one; read it on the same
 * stream): "vs", "advantages" [T*B], "values" [T*B + B] (baseline then bootstrap), "logits" [T*B][2 act],
 * "latent_mean", "latent_logvar" [T*B][latent] */
int vnl_ppo_update_buffer(const vnl_ppo_update*, const char* name, float** dev_ptr, int64_t* count);
/* metrics [9]: [0..8) as vnl_ppo_head, [8] prediction_corr -- the mean of corrcoef([vs ; reward * reward_scaling]) over
 * its (2T)^2 entries (reference intention_losses.py:186-188); 0 when 2T * B floats exceed 60 KB of LDS */
int vnl_ppo_minibatch_grad(vnl_ppo_update*, const float* params, const vnl_ppo_batch*, const vnl_ppo_hparams*, float* grads,
                           float* metrics, void* stream);
/* The same step in two parts, for data-parallel training (reference ppo_imitation/train.py:251-268 averages the gradient over the
 * devices once per minibatch step): part 1 = forward of both networks + loss head + the VALUE network's backward -- in stream
 * order the value segment grads[policy_params .. num_params) is then final; part 2 = the policy network's backward (same
 * arguments, after part 1): the policy segment.  The caller all-reduces the value segment between the two, so that the
 * exchange overlaps part 2.  part 0 = the whole step (== vnl_ppo_minibatch_grad). */
int vnl_ppo_minibatch_grad_part(vnl_ppo_update*, const float* params, const vnl_ppo_batch*, const vnl_ppo_hparams*, float* grads,
                                float* metrics, void* stream, int part);

#ifdef __cplusplus
}
#endif
#endif /* VNL_H_ */
