"""AntTracking: drop-in for reference envs/ant.py:25-438 on the same kernels as RodentTracking.

What differs from the rodent env (selected by the env flags of include/vnl.h, VNL_ENV_*):
  * model assets/ant.xml through brax's `mjcf.load`, which fuses the four joint-less leg bodies into the torso: the env
    sees 10 bodies (world, torso, aux_i, ankle bodies) -- here the tracked-body list; implicit damping
    off (ant.py:47); the reference's config asks for the Newton solver, this library has CG only (same constraint model,
    `H = M + J'DJ` Cholesky instead of the `M^-1` preconditioner) and says so at construction;
  * observation = [reference-trajectory features | qpos | qvel] (ant.py:293-338), the features built from the
    UN-incremented frame counter (ant.py:178), without the appendage block;
  * every reward term from the state BEFORE the step (ant.py:180), `ract` from the action (ant.py:277), weights 0.05 /
    0.01 / 0.20 / 0.01 / 0.001 (ant.py:182-188), metrics and `termination_error` unweighted (ant.py:197,216-225);
    termination error = mean-based (ant.py:228-248), done = rtrunk < 0 or unhealthy or NaN (ant.py:200-201,210);
  * reset: frame 0, no noise (ant.py:90-171).
The reference's AntTracking keeps `info["cur_frame"]` only (it cannot be driven by `acting.actor_step`);
this mirror keeps `info["traj"]` as well.  The reference ships no ant clip: `standing_clip` is the stand-in.
"""
from __future__ import annotations

import os
import warnings
from typing import Any, Optional

import numpy as np
import torch

from .. import _lib
from ..model import mjcf as _mjcf
from ..preprocessing import mjx_preprocess as _pp
from .base import State
from .humanoid import standing_clip
from .rodent import RodentTracking, _load_model

_METRICS_ANT = ("rcom", "rvel", "rtrunk", "rquat", "ract", "termination_error")
_SLOT = {"rcom": 0, "rvel": 1, "rtrunk": 2, "rquat": 3, "ract": 4, "termination_error": 6}


class AntTracking(RodentTracking):
    _env_flags = (_lib.ENV_REWARD_OLD_STATE | _lib.ENV_TERM_MEAN | _lib.ENV_NO_RAPP | _lib.ENV_OBS_QPOS_QVEL |
                  _lib.ENV_WEIGHTS | _lib.ENV_RACT_ACTION | _lib.ENV_METRICS_UNSCALED | _lib.ENV_TRAJ_OLD_FRAME)
    _done_threshold = 0.0
    _use_clip_com = True
    _reward_weights = (0.05, 0.01, 0.20, 0.01, 0.001, 0.0)  # ant.py:182-188 (no rapp)

    def __init__(self, params, healthy_z_range=(0.2, 1.0), reset_noise_scale=1e-2, clip_length: int = 250,
                 episode_length: int = 150, ref_traj_length: int = 5, termination_threshold: float = 0.9,
                 body_error_multiplier: float = 1.0, num_envs: int = 1, device: Any = "cuda", reference_clip=None,
                 model: Optional[_mjcf.CompiledModel] = None, mjcf_path: str = "./assets/ant.xml", **kwargs):
        params = dict(params or {})
        solver = str(params.get("solver", "cg")).lower()  # ant.py:40-47; configs/env_config.yaml:16-21 selects newton, 1 / 4
        self.sys = model if model is not None else _load_model(
            mjcf_path, None, solver, int(params.get("iterations", 6)), int(params.get("ls_iterations", 6)))
        m = self.sys
        if model is not None:  # a pre-compiled model: the env's params still decide the solver options, as in the reference
            m.scalars.update(iterations=int(params.get("iterations", m.scalars["iterations"])),
                             ls_iterations=int(params.get("ls_iterations", m.scalars["ls_iterations"])),
                             solver_newton=1 if solver == "newton" else 0)
        m.scalars["eulerdamp"] = 0  # ant.py:47: mjDSBL_EULERDAMP
        self._n_frames = int(kwargs.get("n_frames", 5))  # ant.py:54-56
        self.backend = "mjx"
        # brax's mjcf.load fuses bodies without joints into their parent: what is left is the world, the torso and every
        # body that carries a joint, in model order (10 for ant.xml)
        has_joint = set(int(b) for b in np.asarray(m.arrays["jnt_bodyid"]))
        self._body_idxs = np.array([b for b in range(int(m.scalars["nbody"])) if b == 0 or b in has_joint], dtype=np.int32)
        self._end_eff_idx = np.zeros(0, dtype=np.int32)
        self._app_idx = np.zeros(0, dtype=np.int32)
        self._com_idx = 1
        self._joint_idxs = np.arange(int(m.scalars["nq"]) - 7, dtype=np.int32)
        self._healthy_z_range = healthy_z_range
        self._reset_noise_scale = 0.0
        self._termination_threshold = float(termination_threshold)
        self._body_error_multiplier = float(body_error_multiplier)
        self._clip_length, self._episode_length = int(clip_length), int(episode_length)
        self._sub_clip_length = 1 << 30  # no sub-clip term in `done`
        self._ref_traj_length = int(ref_traj_length)
        if self._episode_length > self._clip_length:
            raise ValueError("episode_length cannot be greater than clip_length!")  # ant.py:73-74
        if reference_clip is None:
            path = params.get("clip_path")
            reference_clip = _pp.ReferenceClip.load(path) if path and os.path.exists(path) else standing_clip(m, clip_length)
        self._build(reference_clip, num_envs, device)

    @property
    def observation_size(self) -> int:
        return int(self.dims.traj_size) + int(self.dims.obs_size)

    def _present(self, state: State) -> State:
        """obs = [traj features | qpos | qvel] (ant.py:322-338), assembled from the two buffers the kernel fills."""
        raw = state.info["_raw"]
        torch.cat((raw["traj"], raw["obs"]), dim=1, out=state.obs)
        return state

    def _alloc_state(self) -> State:
        st = super()._alloc_state()
        raw = st.info["_raw"]
        obs = torch.zeros((self.num_envs, self.observation_size), dtype=self._dtype, device=self.device)
        metrics = {k: raw["metrics"][:, _SLOT[k]] for k in _METRICS_ANT}
        return State(st.pipeline_state, obs, st.reward, st.done, metrics, st.info)

    def reset(self, rng=None, *, start_frame=None, noise=None, clip_id=None, out=None):
        """ant.py:90-171: frame 0, no noise."""
        B, nq = self.num_envs, int(self.dims.nq)
        if start_frame is None:
            start_frame = torch.zeros((B,), dtype=torch.int32)
        if noise is None:
            noise = torch.zeros((B, nq), dtype=torch.float32)
        return self._present(super().reset(rng, start_frame=start_frame, noise=noise, clip_id=clip_id, out=out))

    def step(self, state: State, action: torch.Tensor) -> State:
        """ant.py:172-226."""
        return self._present(super().step(state, action))
