"""ctypes binding of the C-ABI in include/vnl.h (libvnl.so, built by csrc/build.py).

There is no CPU fallback: if the HIP library is missing, `load_library` raises.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT_LIB = os.path.join(_HERE, "csrc", "libvnl.so")

i32p = C.POINTER(C.c_int32)
f32p = C.POINTER(C.c_float)


class EnvSpec(C.Structure):
    _fields_ = [
        ("clip_frames", C.c_int32),
        ("num_clips", C.c_int32),
        ("ref_traj_length", C.c_int32),
        ("sub_clip_length", C.c_int32),
        ("n_frames", C.c_int32),
        ("num_track_bodies", C.c_int32),
        ("num_end_eff", C.c_int32),
        ("num_appendages", C.c_int32),
        ("num_joint_cols", C.c_int32),
        ("com_ref_col", C.c_int32),
        ("body_idxs", i32p),
        ("end_eff_idx", i32p),
        ("app_body", i32p),
        ("app_ref_col", i32p),
        ("joint_cols", i32p),
        ("healthy_z_lo", C.c_float),
        ("healthy_z_hi", C.c_float),
        ("termination_threshold", C.c_float),
        ("body_error_multiplier", C.c_float),
        ("position", f32p),
        ("quaternion", f32p),
        ("joints", f32p),
        ("body_positions", f32p),
        ("velocity", f32p),
        ("angular_velocity", f32p),
        ("joints_velocity", f32p),
        ("flags", C.c_int32),
        ("done_threshold", C.c_float),
        ("center_of_mass", f32p),
        ("reward_weights", C.c_float * 6),
    ]


ENV_REWARD_OLD_STATE, ENV_TERM_MEAN, ENV_NO_RAPP, ENV_OBS_QPOS_QVEL = 1, 2, 4, 8  # include/vnl.h VNL_ENV_*
ENV_WEIGHTS, ENV_RACT_ACTION, ENV_METRICS_UNSCALED, ENV_TRAJ_OLD_FRAME = 16, 32, 64, 128


STATE_FLOAT_FIELDS = ("qpos", "qvel", "act", "qacc_warmstart", "xpos", "xquat", "subtree_com1", "qfrc_actuator",
                      "obs", "reward", "done", "metrics", "traj", "termination_error")
STATE_INT_FIELDS = ("cur_frame", "sub_clip_frame", "clip_id")


class StatePtrs(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in STATE_FLOAT_FIELDS + STATE_INT_FIELDS]


class Dims(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("nq", "nv", "nu", "nbody", "njnt", "ngeom_collide", "ncon", "nefc",
                                          "obs_size", "traj_size", "workspace_floats_per_env", "workgroups_per_cu",
                                          "nbody_dynamic", "kernel_specialised")]


class PolicySpec(C.Structure):
    _fields_ = [
        ("traj_size", C.c_int32),
        ("obs_size", C.c_int32),
        ("action_size", C.c_int32),
        ("latent_size", C.c_int32),
        ("num_encoder_layers", C.c_int32),
        ("num_decoder_layers", C.c_int32),
        ("encoder_layers", C.c_int32 * 8),
        ("decoder_layers", C.c_int32 * 8),
    ]


POST_MAX_OPS = 24


class PostOp(C.Structure):  # include/vnl.h: vnl_post_op
    _fields_ = [("dst", C.c_void_p), ("src", C.c_void_p), ("first", C.c_void_p), ("log", C.c_void_p),
                ("width", C.c_int32), ("pad_", C.c_int32)]


class PostDesc(C.Structure):  # include/vnl.h: vnl_post_desc
    _fields_ = [(n, C.c_void_p) for n in ("steps", "prev_done", "done", "truncation", "reward", "log_reward",
                                          "log_discount", "log_truncation")] + \
               [("episode_length", C.c_int32), ("action_repeat", C.c_int32), ("num_ops", C.c_int32),
                ("pad_", C.c_int32), ("ops", PostOp * POST_MAX_OPS)]


class GatherOp(C.Structure):  # include/vnl.h: vnl_gather_op
    _fields_ = [("dst", C.c_void_p), ("src", C.c_void_p), ("T", C.c_int32), ("width", C.c_int32)]


class GatherDesc(C.Structure):  # include/vnl.h: vnl_gather_desc
    _fields_ = [("idx", C.c_void_p), ("N", C.c_int32), ("M", C.c_int32), ("num_ops", C.c_int32), ("pad_", C.c_int32),
                ("ops", GatherOp * POST_MAX_OPS)]


class PPOHeadArgs(C.Structure):  # include/vnl.h: vnl_ppo_head_args
    _fields_ = [(n, C.c_int32) for n in ("T", "B", "act", "latent")] + \
               [(n, C.c_void_p) for n in ("logits", "baseline", "bootstrap", "lat_mean", "lat_logvar", "raw_action",
                                          "behaviour_log_prob", "reward", "truncation", "discount", "eps_entropy")] + \
               [(n, C.c_float) for n in ("entropy_cost", "discounting", "reward_scaling", "gae_lambda",
                                         "clipping_epsilon", "kl_weight", "min_std", "var_scale")] + \
               [("normalize_advantage", C.c_int32), ("pad_", C.c_int32)] + \
               [(n, C.c_void_p) for n in ("g_logits", "g_baseline", "g_lat_mean", "g_lat_logvar", "vs", "advantages",
                                          "metrics")]


PPO_HEAD_WORKSPACE_FLOATS = 4 + 4 * 256


class PPONetSpec(C.Structure):  # include/vnl.h: vnl_ppo_net_spec
    _fields_ = [(n, C.c_int32) for n in ("traj_size", "obs_size", "action_size", "latent_size", "num_encoder_layers",
                                          "num_decoder_layers", "num_value_layers")] + \
               [("encoder_layers", C.c_int32 * 8), ("decoder_layers", C.c_int32 * 8), ("value_layers", C.c_int32 * 8)]


class PPOBatch(C.Structure):  # include/vnl.h: vnl_ppo_batch
    _fields_ = [(n, C.c_void_p) for n in ("traj", "obs", "next_obs_last", "raw_action", "behaviour_log_prob", "reward",
                                          "truncation", "discount", "eps_latent", "eps_entropy", "obs_mean", "obs_std")]


class PPOHParams(C.Structure):  # include/vnl.h: vnl_ppo_hparams
    _fields_ = [(n, C.c_float) for n in ("entropy_cost", "discounting", "reward_scaling", "gae_lambda",
                                         "clipping_epsilon", "kl_weight", "min_std", "var_scale")] + \
               [("normalize_advantage", C.c_int32), ("pad_", C.c_int32)]

EXPORTS = (
    "vnl_last_error", "vnl_version", "vnl_model_create", "vnl_model_destroy", "vnl_env_create", "vnl_env_destroy",
    "vnl_env_dims", "vnl_env_reset", "vnl_env_step", "vnl_env_fk", "vnl_env_debug", "vnl_env_scratch", "vnl_policy_create", "vnl_policy_destroy",
    "vnl_policy_num_params", "vnl_policy_forward", "vnl_rollout_post", "vnl_ppo_head", "vnl_adam_step", "vnl_gather_rows",
    "vnl_ppo_update_create", "vnl_ppo_update_destroy", "vnl_ppo_update_num_params", "vnl_ppo_update_buffer",
    "vnl_ppo_minibatch_grad",
    "vnl_ppo_minibatch_grad_part",
)
_HIP_ONLY = ("vnl_policy_", "vnl_ppo_update_", "vnl_ppo_minibatch_")  # not in the test-only host simulation


class VnlError(RuntimeError):
    pass


def _declare(lib: C.CDLL) -> C.CDLL:
    vp = C.c_void_p
    lib.vnl_last_error.restype = C.c_char_p
    lib.vnl_version.restype = C.c_int
    lib.vnl_model_create.argtypes = [vp, C.c_size_t, C.POINTER(vp)]
    lib.vnl_model_destroy.argtypes = [vp]
    lib.vnl_model_destroy.restype = None
    lib.vnl_env_create.argtypes = [vp, C.POINTER(EnvSpec), C.c_int32, C.c_int32, C.POINTER(vp)]
    lib.vnl_env_destroy.argtypes = [vp]
    lib.vnl_env_destroy.restype = None
    lib.vnl_env_dims.argtypes = [vp, C.POINTER(Dims)]
    lib.vnl_env_reset.argtypes = [vp, vp, vp, C.POINTER(StatePtrs), vp]
    lib.vnl_env_step.argtypes = [vp, vp, C.POINTER(StatePtrs), vp]
    lib.vnl_env_fk.argtypes = [vp, vp, C.POINTER(StatePtrs), vp]
    lib.vnl_env_debug.argtypes = [vp, C.c_int32, C.POINTER(C.c_int32)]
    lib.vnl_env_scratch.argtypes = [vp, C.c_char_p, C.POINTER(vp), C.POINTER(C.c_int32)]
    lib.vnl_rollout_post.argtypes = [C.POINTER(PostDesc), C.c_int32, vp]
    lib.vnl_ppo_head.argtypes = [C.POINTER(PPOHeadArgs), vp, vp]
    lib.vnl_gather_rows.argtypes = [C.POINTER(GatherDesc), vp]
    lib.vnl_adam_step.argtypes = [vp, vp, vp, vp, vp, C.c_int64] + [C.c_double] * 4 + [vp]
    if hasattr(lib, "vnl_ppo_update_create"):
        lib.vnl_ppo_update_create.argtypes = [C.POINTER(PPONetSpec), C.c_int32, C.c_int32, C.c_int32, C.POINTER(vp)]
        lib.vnl_ppo_update_destroy.argtypes = [vp]
        lib.vnl_ppo_update_destroy.restype = None
        lib.vnl_ppo_update_num_params.argtypes = [vp]
        lib.vnl_ppo_update_num_params.restype = C.c_int64
        lib.vnl_ppo_update_buffer.argtypes = [vp, C.c_char_p, C.POINTER(vp), C.POINTER(C.c_int64)]
        lib.vnl_ppo_minibatch_grad.argtypes = [vp, vp, C.POINTER(PPOBatch), C.POINTER(PPOHParams), vp, vp, vp]
        lib.vnl_ppo_minibatch_grad_part.argtypes = [vp, vp, C.POINTER(PPOBatch), C.POINTER(PPOHParams), vp, vp, vp, C.c_int]
    if hasattr(lib, "vnl_policy_create"):
        lib.vnl_policy_create.argtypes = [C.POINTER(PolicySpec), C.c_int32, C.c_int32, C.POINTER(vp)]
        lib.vnl_policy_destroy.argtypes = [vp]
        lib.vnl_policy_destroy.restype = None
        lib.vnl_policy_num_params.argtypes = [vp]
        lib.vnl_policy_num_params.restype = C.c_int64
        lib.vnl_policy_forward.argtypes = [vp] + [vp] * 7 + [C.c_int32, C.c_int32] + [vp] * 6 + [vp, vp] + [vp]
    return lib


_cache = {}


def load_library(path: str | None = None, env_only: bool = False) -> C.CDLL:
    """Load libvnl.so.  Raises (loudly) if it has not been built.  `env_only` is for the test-only host
    simulation, which contains the env entry points but not the MFMA policy kernel."""
    path = os.path.abspath(path or os.environ.get("VNL_LIB", DEFAULT_LIB))
    if path not in _cache:
        if not os.path.exists(path):
            raise VnlError(
                f"HIP extension not found at {path}. Build it with `python -c 'import __graft_entry__ as g; "
                "g.build()'` (hipcc --offload-arch=gfx950). There is no CPU fallback for the rollout.")
        lib = C.CDLL(path)
        missing = [s for s in EXPORTS if not hasattr(lib, s) and not (env_only and s.startswith(_HIP_ONLY))]
        if missing:
            raise VnlError(f"{path} lacks symbols {missing}")
        _cache[path] = _declare(lib)
    return _cache[path]


def check(lib: C.CDLL, rc: int) -> None:
    if rc != 0:
        raise VnlError(f"vnl error {rc}: {lib.vnl_last_error().decode()}")
