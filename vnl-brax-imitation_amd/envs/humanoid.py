"""HumanoidTracking: drop-in for reference envs/humanoid.py:25-430 on the same kernels as RodentTracking.

What differs from the rodent env (all of it selected by the env flags of include/vnl.h, VNL_ENV_*):
  * model assets/humanoid.xml (17 bodies, 27 dofs, 21 motors, five capsule-floor <pair>s, timestep 0.005, implicit
    damping off: humanoid.py:39-52), no rescale;
  * observation = [qpos, qvel] (humanoid.py:354-368); the reference trajectory has no appendage block and uses ALL bodies
    (world included) and all hinge joints (humanoid.py:319-352);
  * termination error = 1 - (0.5 mult mean|d bodies| + 0.5 mean|d joints|) / threshold (humanoid.py:241-262), threshold 0.9;
  * every reward term is computed from the state BEFORE the step (`_calculate_reward(state, action)`, humanoid.py:195),
    rcom against the clip's center_of_mass (humanoid.py:273), no rapp; done = rtrunk < 0.5 (on the unscaled value,
    humanoid.py:199) or unhealthy or NaN; no sub-clip term;
  * reset adds no noise (humanoid.py:79-133).
The constructor keeps the reference's signature (`params` dict with solver / iterations / ls_iterations / clip_path).
"""
from __future__ import annotations

import os
from typing import Any, Optional

import numpy as np
import torch

from .. import _lib
from ..model import mjcf as _mjcf
from ..preprocessing import mjx_preprocess as _pp
from .rodent import RodentTracking, _load_model


def standing_clip(model, clip_length: int = 250) -> "_pp.ReferenceClip":
    """The reference's clips/humanoid_traj_stand.p is not shipped (.gitignore:19): qpos0 tiled, zero velocities."""
    return _pp.process_qpos(model, np.tile(np.asarray(model.arrays["qpos0"], dtype=np.float64), (clip_length, 1)))


class HumanoidTracking(RodentTracking):
    _env_flags = _lib.ENV_REWARD_OLD_STATE | _lib.ENV_TERM_MEAN | _lib.ENV_NO_RAPP | _lib.ENV_OBS_QPOS_QVEL
    _done_threshold = 0.5
    _use_clip_com = True

    def __init__(self, params, healthy_z_range=(1.0, 2.0), reset_noise_scale=1e-2, clip_length: int = 250,
                 episode_length: int = 150, ref_traj_length: int = 5, termination_threshold: float = 0.9,
                 body_error_multiplier: float = 1.0, num_envs: int = 1, device: Any = "cuda", reference_clip=None,
                 model: Optional[_mjcf.CompiledModel] = None, mjcf_path: str = "./assets/humanoid.xml", **kwargs):
        params = dict(params or {})
        self.sys = model if model is not None else _load_model(
            mjcf_path, None, params.get("solver", "cg"), int(params.get("iterations", 6)), int(params.get("ls_iterations", 6)))
        m = self.sys
        self._n_frames = int(kwargs.get("n_frames", 5))  # humanoid.py:54-56
        self.backend = "mjx"
        nbody, nj = int(m.scalars["nbody"]), int(m.scalars["nq"]) - 7
        self._end_eff_idx = np.zeros(0, dtype=np.int32)
        self._app_idx = np.zeros(0, dtype=np.int32)
        self._com_idx = 1
        self._body_idxs = np.arange(nbody, dtype=np.int32)       # ref_traj.body_positions - data.xpos: every body
        self._joint_idxs = np.arange(nj, dtype=np.int32)         # ref_traj.joints - data.qpos[7:]: every hinge
        self._healthy_z_range = healthy_z_range
        self._reset_noise_scale = 0.0                            # reset() adds none (humanoid.py:79-100)
        self._termination_threshold = float(termination_threshold)
        self._body_error_multiplier = float(body_error_multiplier)
        self._clip_length, self._episode_length = int(clip_length), int(episode_length)
        self._sub_clip_length = 1 << 30                          # no sub-clip term in `done`
        self._ref_traj_length = int(ref_traj_length)
        if self._episode_length > self._clip_length:
            raise ValueError("episode_length cannot be greater than clip_length!")  # humanoid.py:76-77
        if reference_clip is None:
            path = params.get("clip_path")
            reference_clip = _pp.ReferenceClip.load(path) if path and os.path.exists(path) else standing_clip(m, clip_length)
        self._build(reference_clip, num_envs, device)

    def reset(self, rng=None, *, start_frame=None, noise=None, clip_id=None, out=None):
        """humanoid.py:79-133: start_frame ~ U[0, clip_length - episode_length - ref_traj_length), no noise."""
        B, nq = self.num_envs, int(self.dims.nq)
        if start_frame is None:
            gen = rng if isinstance(rng, torch.Generator) else (self._gen if rng is None else torch.Generator().manual_seed(int(rng)))
            hi = self._clip_length - self._episode_length - self._ref_traj_length
            start_frame = torch.randint(0, max(hi, 1), (B,), generator=gen, dtype=torch.int32)
        if noise is None:
            noise = torch.zeros((B, nq), dtype=torch.float32)
        return super().reset(rng, start_frame=start_frame, noise=noise, clip_id=clip_id, out=out)
