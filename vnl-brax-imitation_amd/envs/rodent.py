"""RodentTracking: drop-in for reference envs/rodent.py:16-470, natively batched.

Same constructor keywords (rodent.py:17-38; configs/env_config.yaml:32-106), same
reset/step protocol and State fields, same numbers (bug-compatible: SURVEY.md
Appendix C).  Differences a caller sees:

  * the env is batched (it subsumes brax's VmapWrapper): every State field has a
    leading env dimension B = num_envs;
  * the JAX PRNG stream is not reproduced: `reset` draws start frames and noise
    from a torch.Generator, or takes them explicitly (`start_frame=`, `noise=`);
  * `step` updates the State's buffers in place (the kernels own no state);
  * physics + obs/traj/reward/termination run in ONE HIP kernel launch per
    control step through the C-ABI in include/vnl.h (no CPU fallback).
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Any, Dict, List, Optional, Sequence

import numpy as np
import torch

from .. import _lib
from ..model import blob as _blob
from ..model import mjcf as _mjcf
from .base import Env, PipelineState, State

_PKG = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# Test seam, outside every public signature: (library, dtype) that the envs constructed while it is set bind to instead of
# csrc/libvnl.so -- the host-compiled PRODUCT source of tests/hostsim, or a regression build of it.  Set and cleared by
# tests/helpers.py `backend(...)`; None in any product use, where construction fails loudly without a HIP device.
_TEST_BACKEND = None
_METRICS = ("rcom", "rvel", "rtrunk", "rquat", "ract", "rapp", "termination_error")


def packaged_model_path(name: str = "rodent", scale_factor: Optional[float] = 0.9) -> str:
    if scale_factor is None:  # models compiled without the dm_control rescale (humanoid, ant)
        return os.path.join(_PKG, "data", f"{name}.npz")
    return os.path.join(_PKG, "data", f"{name}_scale{scale_factor:g}.npz")


def _load_model(mjcf_path: str, scale_factor: float, solver: str, iterations: int, ls_iterations: int):
    if os.path.exists(mjcf_path):
        return _mjcf.compile_mjcf(mjcf_path, scale_factor=scale_factor, solver=solver, iterations=iterations,
                                  ls_iterations=ls_iterations)
    stem = os.path.splitext(os.path.basename(mjcf_path))[0]
    packaged = packaged_model_path(stem, scale_factor)
    if os.path.exists(packaged):
        model = _mjcf.CompiledModel.load(packaged)
        model.scalars.update(iterations=iterations, ls_iterations=ls_iterations,
                             solver_newton=1 if solver.lower() == "newton" else 0)
        return model
    raise FileNotFoundError(f"{mjcf_path} not found and no packaged compiled model {packaged}")


def _clamped_take(x: np.ndarray, idx: Sequence[int], axis: int) -> np.ndarray:
    """x[..., idx, ...] with JAX's clamp-out-of-range gather semantics."""
    idx = np.clip(np.asarray(idx, dtype=np.int64), 0, x.shape[axis] - 1)
    return np.take(x, idx, axis=axis)


class RodentTracking(Env):
    def __init__(
        self,
        reference_clip,
        end_eff_names: List[str],
        appendage_names: List[str],
        walker_body_names: List[str],
        joint_names: List[str],
        center_of_mass: str,
        mjcf_path: str = "./assets/rodent.xml",
        scale_factor: float = 0.9,
        solver: str = "cg",
        iterations: int = 6,
        ls_iterations: int = 6,
        healthy_z_range=(0.05, 0.5),
        reset_noise_scale=1e-3,
        clip_length: int = 250,
        sub_clip_length: int = 10,
        ref_traj_length: int = 5,
        termination_threshold: float = 5,
        body_error_multiplier: float = 1.0,
        num_envs: int = 1,
        device: Any = "cuda",
        model: Optional[_mjcf.CompiledModel] = None,
        **kwargs,
    ):
        # --- model (rodent.py:39-63) --------------------------------------------------
        self.sys = model if model is not None else _load_model(mjcf_path, scale_factor, solver, iterations,
                                                               ls_iterations)
        m = self.sys
        self._n_frames = int(kwargs.get("n_frames", 5))  # rodent.py:95-99
        self.backend = "mjx"
        # --- name -> id (rodent.py:65-93) ---------------------------------------------
        self._end_eff_idx = np.array([m.body_id(b) for b in end_eff_names], dtype=np.int32)
        self._app_idx = np.array([m.body_id(b) for b in appendage_names], dtype=np.int32)
        self._com_idx = int(m.body_id(center_of_mass))
        self._body_idxs = np.array([m.body_id(b) for b in walker_body_names], dtype=np.int32)
        self._joint_idxs = np.array([m.joint_id(j) for j in joint_names], dtype=np.int32)

        self._healthy_z_range = healthy_z_range
        self._reset_noise_scale = float(reset_noise_scale)
        self._termination_threshold = float(termination_threshold)
        self._body_error_multiplier = float(body_error_multiplier)
        self._clip_length = int(clip_length)
        self._sub_clip_length = int(sub_clip_length)
        self._ref_traj_length = int(ref_traj_length)
        if self._sub_clip_length > self._clip_length:
            raise ValueError("episode_length cannot be greater than clip_length!")  # rodent.py:116-117

        self._build(reference_clip, num_envs, device)

    # env-variant knobs (include/vnl.h VNL_ENV_*): RodentTracking's glue by default
    _env_flags = 0
    _done_threshold = 0.0
    _reward_weights = None  # with ENV_WEIGHTS: (rcom, rvel, rtrunk, rquat, ract, rapp)
    _use_clip_com = False

    def _build(self, reference_clip, num_envs, device, _library=None, _dtype=torch.float32):
        # everything set so far is configuration; what follows is per-instance device state (with_num_envs)
        if _library is None and _TEST_BACKEND is not None:  # tests/helpers.py `backend(...)` only; never set by the product
            _library, _dtype = _TEST_BACKEND
        self._config_attrs = dict(self.__dict__)
        self._build_args = (reference_clip, device, _library, _dtype)
        m = self.sys
        healthy_z_range = self._healthy_z_range
        # --- clip (rodent.py:112-115): filter body_positions to the tracked bodies -----
        clip = reference_clip.as_multi() if hasattr(reference_clip, "as_multi") else reference_clip
        f32 = lambda a: np.ascontiguousarray(np.asarray(a), dtype=np.float32)  # noqa: E731
        bp = f32(clip.body_positions)
        self._clip = dict(
            position=f32(clip.position), quaternion=f32(clip.quaternion), joints=f32(clip.joints),
            body_positions=np.ascontiguousarray(_clamped_take(bp, self._body_idxs, axis=-2)),
            velocity=f32(clip.velocity), angular_velocity=f32(clip.angular_velocity),
            joints_velocity=f32(clip.joints_velocity),
        )
        if self._use_clip_com:
            self._clip["center_of_mass"] = f32(clip.center_of_mass)
        self._num_clips, self._T = self._clip["position"].shape[:2]
        nb = len(self._body_idxs)
        nj = int(m.scalars["nq"]) - 7
        # effective indices after the reference's id-vs-column mix-ups + JAX clamping (Appendix C.4/C.5)
        self._app_ref_col = np.clip(self._app_idx, 0, nb - 1).astype(np.int32)
        self._com_ref_col = int(np.clip(self._com_idx, 0, nb - 1))
        self._joint_cols = np.clip(self._joint_idxs, 0, nj - 1).astype(np.int32)

        # --- device library -----------------------------------------------------------------
        self.num_envs = int(num_envs)
        self.device = torch.device(device)
        if _library is None:
            if self.device.type != "cuda":
                raise _lib.VnlError("RodentTracking runs on a HIP device only (device='cuda[:i]'); no CPU fallback")
            if self.device.index is None:
                self.device = torch.device("cuda", torch.cuda.current_device())
            _library = _lib.load_library()
        self._L = _library
        self._dtype = _dtype  # float64 only with the test-only host simulation built with -DVNL_REAL=double
        blob = _blob.to_blob(m)
        self._blob = C.create_string_buffer(blob, len(blob))
        self._model_h = C.c_void_p()
        _lib.check(self._L, self._L.vnl_model_create(self._blob, len(blob), C.byref(self._model_h)))
        spec = _lib.EnvSpec()
        spec.clip_frames, spec.num_clips = self._T, self._num_clips
        spec.ref_traj_length, spec.sub_clip_length, spec.n_frames = self._ref_traj_length, self._sub_clip_length, self._n_frames
        spec.num_track_bodies, spec.num_end_eff = nb, len(self._end_eff_idx)
        spec.num_appendages, spec.num_joint_cols, spec.com_ref_col = len(self._app_idx), len(self._joint_cols), self._com_ref_col
        self._keep = [self._body_idxs, self._end_eff_idx, self._app_idx, self._app_ref_col, self._joint_cols]
        ip = lambda a: a.ctypes.data_as(_lib.i32p)  # noqa: E731
        fp = lambda a: a.ctypes.data_as(_lib.f32p)  # noqa: E731
        spec.body_idxs, spec.end_eff_idx, spec.app_body = ip(self._body_idxs), ip(self._end_eff_idx), ip(self._app_idx)
        spec.app_ref_col, spec.joint_cols = ip(self._app_ref_col), ip(self._joint_cols)
        spec.healthy_z_lo, spec.healthy_z_hi = float(healthy_z_range[0]), float(healthy_z_range[1])
        spec.termination_threshold, spec.body_error_multiplier = self._termination_threshold, self._body_error_multiplier
        spec.flags, spec.done_threshold = int(self._env_flags), float(self._done_threshold)
        for i, w in enumerate(self._reward_weights or ()):
            spec.reward_weights[i] = float(w)
        for k, v in self._clip.items():
            setattr(spec, k, fp(v))
        self._env_h = C.c_void_p()
        dev_index = self.device.index if self.device.type == "cuda" else 0
        _lib.check(self._L, self._L.vnl_env_create(self._model_h, C.byref(spec), self.num_envs, dev_index,
                                                   C.byref(self._env_h)))
        self.dims = _lib.Dims()
        _lib.check(self._L, self._L.vnl_env_dims(self._env_h, C.byref(self.dims)))
        self._gen = torch.Generator(device="cpu")
        self._gen.manual_seed(0)

    def with_num_envs(self, num_envs: int) -> "RodentTracking":
        """A second env of the same configuration (model, clip, reward parameters) with its own batch of `num_envs`
        envs on the device -- the evaluator's env when num_eval_envs differs from the training batch."""
        new = object.__new__(type(self))
        new.__dict__.update(self._config_attrs)
        new._build(self._build_args[0], int(num_envs), *self._build_args[1:])
        return new

    def __del__(self):
        try:
            if getattr(self, "_env_h", None):
                self._L.vnl_env_destroy(self._env_h)
            if getattr(self, "_model_h", None):
                self._L.vnl_model_destroy(self._model_h)
        except Exception:
            pass

    # --- brax Env protocol -----------------------------------------------------------------
    @property
    def observation_size(self) -> int:
        return int(self.dims.obs_size)

    @property
    def traj_size(self) -> int:
        return int(self.dims.traj_size)

    @property
    def action_size(self) -> int:
        return int(self.dims.nu)

    @property
    def dt(self) -> float:
        return float(self.sys.scalars["timestep"]) * self._n_frames

    def env_spec(self) -> Dict[str, Any]:
        """Effective env description (also what tests hand to the CPU oracle)."""
        return dict(
            T=self._T, ref_len=self._ref_traj_length, sub_clip_length=self._sub_clip_length, n_frames=self._n_frames,
            nb=len(self._body_idxs), nee=len(self._end_eff_idx), napp=len(self._app_idx), njc=len(self._joint_cols),
            com_ref_col=self._com_ref_col, body_idxs=self._body_idxs.tolist(), end_eff_idx=self._end_eff_idx.tolist(),
            app_body=self._app_idx.tolist(), app_ref_col=self._app_ref_col.tolist(),
            joint_cols=self._joint_cols.tolist(), healthy_z_lo=self._healthy_z_range[0],
            healthy_z_hi=self._healthy_z_range[1], termination_threshold=self._termination_threshold,
            body_error_multiplier=self._body_error_multiplier, flags=int(self._env_flags),
            done_threshold=float(self._done_threshold),
            reward_weights=list(self._reward_weights) if self._reward_weights else None,
        )

    def clip_arrays(self, clip: int = 0) -> Dict[str, np.ndarray]:
        return {k: v[clip] for k, v in self._clip.items()}

    def _stream(self):
        if self.device.type == "cuda":
            return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        return C.c_void_p(0)

    def _alloc_state(self) -> State:
        B, dv, d = self.num_envs, self.device, self.dims
        ps = PipelineState.allocate(d, B, dv, self._dtype)
        z = lambda n, dt=self._dtype: torch.zeros((B, n) if n else (B,), dtype=dt, device=dv)  # noqa: E731
        raw = dict(obs=z(d.obs_size), reward=z(0), done=z(0), metrics=z(7), traj=z(d.traj_size),
                   termination_error=z(0), cur_frame=z(0, torch.int32), sub_clip_frame=z(0, torch.int32),
                   clip_id=z(0, torch.int32))
        metrics = {k: raw["metrics"][:, i] for i, k in enumerate(_METRICS)}
        info = dict(cur_frame=raw["cur_frame"], sub_clip_frame=raw["sub_clip_frame"], traj=raw["traj"],
                    termination_error=raw["termination_error"], clip_id=raw["clip_id"], _raw=raw)
        return State(ps, raw["obs"], raw["reward"], raw["done"], metrics, info)

    def _ptrs(self, state: State) -> _lib.StatePtrs:
        raw, ps = state.info["_raw"], state.pipeline_state
        p = _lib.StatePtrs()
        for k in PipelineState._FIELDS:
            setattr(p, k, ps.raw(k).data_ptr())
        for k in ("obs", "reward", "done", "metrics", "traj", "termination_error", "cur_frame", "sub_clip_frame",
                  "clip_id"):
            setattr(p, k, raw[k].data_ptr())
        return p

    def reset(self, rng=None, *, start_frame: Optional[torch.Tensor] = None, noise: Optional[torch.Tensor] = None,
              clip_id: Optional[torch.Tensor] = None, out: Optional[State] = None) -> State:
        """rodent.py:119-176.  `rng`: None (internal generator), an int seed or a torch.Generator (CPU).
        Explicit `start_frame` (B,) int and `noise` (B, nq) (already scaled) override the draws."""
        B, nq = self.num_envs, int(self.dims.nq)
        gen = self._gen
        if isinstance(rng, torch.Generator):
            gen = rng
        elif rng is not None:
            gen = torch.Generator(device="cpu")
            gen.manual_seed(int(rng))
        if start_frame is None:
            hi = self._clip_length - self._sub_clip_length - self._ref_traj_length  # rodent.py:123-128
            start_frame = torch.randint(0, max(hi, 1), (B,), generator=gen, dtype=torch.int32)
        if noise is None:
            noise = self._reset_noise_scale * torch.randn((B, nq), generator=gen, dtype=torch.float32)
        if clip_id is None:
            clip_id = (torch.zeros(B, dtype=torch.int32) if self._num_clips == 1 else
                       torch.randint(0, self._num_clips, (B,), generator=gen, dtype=torch.int32))
        state = out if out is not None else self._alloc_state()
        sf = torch.as_tensor(start_frame, dtype=torch.int32).to(self.device).contiguous()
        nz = torch.as_tensor(noise).to(device=self.device, dtype=self._dtype).contiguous()  # [B][nq]
        state.info["_raw"]["clip_id"].copy_(torch.as_tensor(clip_id, dtype=torch.int32).to(self.device))
        p = self._ptrs(state)
        _lib.check(self._L, self._L.vnl_env_reset(self._env_h, sf.data_ptr(), nz.data_ptr(), C.byref(p),
                                                  self._stream()))
        self._hold = (sf, nz)  # keep inputs alive until the stream has consumed them
        return state

    def step(self, state: State, action: torch.Tensor) -> State:
        """rodent.py:178-239.  action: (B, nu)."""
        nu, B = int(self.dims.nu), self.num_envs
        if tuple(action.shape) != (B, nu):
            raise ValueError(f"action must be ({B},{nu}), got {tuple(action.shape)}")
        a = action.to(device=self.device, dtype=self._dtype).contiguous()
        p = self._ptrs(state)
        ev = getattr(self, "kernel_events", None)  # (start, end) torch.cuda.Event pair, used by bench.py
        if ev is not None:
            ev[0].record()
        _lib.check(self._L, self._L.vnl_env_step(self._env_h, a.data_ptr(), C.byref(p), self._stream()))
        if ev is not None:
            ev[1].record()
        self._hold = (a,)
        return state

    def forward_kinematics(self, qpos: torch.Tensor, out: Optional[State] = None) -> State:
        """`mjx.kinematics` of one qpos row per env (B, nq) on the device (vnl_env_fk): returns a State whose
        pipeline_state.xpos / xquat / subtree_com_root are filled and whose qpos carries the normalised root quaternion.
        Used by the device route of clip preprocessing (preprocessing/mjx_preprocess.py: process_qpos(..., fk=...))."""
        B, nq = self.num_envs, int(self.dims.nq)
        if tuple(qpos.shape) != (B, nq):
            raise ValueError(f"qpos must be ({B},{nq}), got {tuple(qpos.shape)}")
        state = out if out is not None else self._alloc_state()
        q = qpos.to(device=self.device, dtype=self._dtype).contiguous()
        state.pipeline_state.qpos.copy_(q)
        p = self._ptrs(state)
        _lib.check(self._L, self._L.vnl_env_fk(self._env_h, q.data_ptr(), C.byref(p), self._stream()))
        self._hold = (q,)
        return state

    # --- bisection hook -----------------------------------------------------------------------
    def debug(self, mode=True) -> None:
        """Bisection hooks.  mode 1 (True): every reset/step also dumps the per-env LDS image at the end of the
        kernel; mode 2: `step` dumps it as its LAST forward pass leaves it (before the Euler update reuses the
        constraint rows' space), `reset` at its end as before.  Either mode also records the solver's discrete
        decisions (`solver_trace()`).  0 / False: off."""
        stride = C.c_int32()
        _lib.check(self._L, self._L.vnl_env_debug(self._env_h, int(mode), C.byref(stride)))
        self._dump_stride = stride.value

    def solver_trace(self) -> torch.Tensor:
        """(B, n_frames, VNL_TRACE_INTS) int32: discrete decisions of the solver call of every substep of the last
        step (row 0 only after a reset); layout in csrc/vnl_types.h."""
        ptr, cnt = C.c_void_p(), C.c_int32()
        _lib.check(self._L, self._L.vnl_env_scratch(self._env_h, b"solver_trace", C.byref(ptr), C.byref(cnt)))
        B, n = self.num_envs, cnt.value
        if self.device.type == "cuda":
            torch.cuda.synchronize(self.device)
            flat = torch.empty(B * n, dtype=torch.int32, device=self.device)
            rc = C.CDLL("libamdhip64.so").hipMemcpy(C.c_void_p(flat.data_ptr()), ptr, C.c_size_t(4 * B * n), C.c_int(3))
            if rc != 0:
                raise _lib.VnlError(f"hipMemcpy failed: {rc}")
            flat = flat.cpu()
        else:
            flat = torch.from_numpy(np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_int32)), shape=(B * n,)).copy())
        return flat.view(B, max(self._n_frames, 1), -1)

    def scratch(self, name: str) -> torch.Tensor:
        """(B, count) copy of a named per-env scratch section left by the last reset/step (debug on)."""
        ptr, cnt = C.c_void_p(), C.c_int32()
        _lib.check(self._L, self._L.vnl_env_scratch(self._env_h, name.encode(), C.byref(ptr), C.byref(cnt)))
        B, n, stride = self.num_envs, cnt.value, self._dump_stride
        esz = 8 if self._dtype == torch.float64 else 4
        total = (B - 1) * stride + n
        if self.device.type == "cuda":
            torch.cuda.synchronize(self.device)
            flat = torch.empty(total, dtype=self._dtype, device=self.device)
            hip = C.CDLL("libamdhip64.so")
            rc = hip.hipMemcpy(C.c_void_p(flat.data_ptr()), ptr, C.c_size_t(esz * total), C.c_int(3))
            if rc != 0:
                raise _lib.VnlError(f"hipMemcpy failed: {rc}")
        else:
            ct = C.c_double if self._dtype == torch.float64 else C.c_float
            flat = torch.from_numpy(np.ctypeslib.as_array(C.cast(ptr, C.POINTER(ct)), shape=(total,)).copy())
        return torch.as_strided(flat, (B, n), (stride, 1)).contiguous()


class RodentMultiClipTracking(RodentTracking):
    """Multi-clip variant.  The reference only has a stub (rodent.py:473-475); here the clip
    container carries a leading clip axis and each env tracks `info['clip_id']`."""
