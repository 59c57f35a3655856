"""Clip preprocessing: STAC qpos -> ReferenceClip features.

Counterpart of reference preprocessing/mjx_preprocess.py:21-193.  Offline, once per
clip, NumPy float32 (the reference runs the same arithmetic under JAX's float32
default): per-frame forward kinematics for xpos/xquat of every body, finite-
difference velocities with the quaternion log map, joint velocities clipped to
+-max_qvel, last velocity row zero (the pad-with-last-frame rule, :93).
"""
from __future__ import annotations

import dataclasses
import pickle
from typing import Optional, Sequence

import numpy as np

from ..model import mjcf
from . import transformations as tr


@dataclasses.dataclass
class ReferenceClip:
    """Fields of the reference's ReferenceClip (mjx_preprocess.py:21-40).

    Arrays are (T, ...) for one clip or (C, T, ...) for a multi-clip container.
    """

    position: Optional[np.ndarray] = None
    quaternion: Optional[np.ndarray] = None
    joints: Optional[np.ndarray] = None
    body_positions: Optional[np.ndarray] = None
    velocity: Optional[np.ndarray] = None
    joints_velocity: Optional[np.ndarray] = None
    angular_velocity: Optional[np.ndarray] = None
    body_quaternions: Optional[np.ndarray] = None

    def replace(self, **kw) -> "ReferenceClip":
        return dataclasses.replace(self, **kw)

    @property
    def num_clips(self) -> int:
        return 1 if self.position.ndim == 2 else self.position.shape[0]

    def as_multi(self) -> "ReferenceClip":
        """View with a leading clip axis (C, T, ...)."""
        if self.position.ndim == 3:
            return self
        return ReferenceClip(**{f.name: (None if getattr(self, f.name) is None else getattr(self, f.name)[None])
                                for f in dataclasses.fields(self)})

    @staticmethod
    def stack(clips: Sequence["ReferenceClip"]) -> "ReferenceClip":
        return ReferenceClip(**{f.name: np.stack([getattr(c, f.name) for c in clips])
                                for f in dataclasses.fields(ReferenceClip)})

    def save(self, path: str) -> None:
        np.savez_compressed(path, **{f.name: getattr(self, f.name) for f in dataclasses.fields(self)
                                     if getattr(self, f.name) is not None})

    @staticmethod
    def load(path: str) -> "ReferenceClip":
        z = np.load(path, allow_pickle=False)
        return ReferenceClip(**{k: z[k] for k in z.files if k in {f.name for f in dataclasses.fields(ReferenceClip)}})


def compute_velocity_from_kinematics(qpos_trajectory: np.ndarray, dt: float) -> np.ndarray:
    """mjx_preprocess.py:170-193: (T+1, nq) -> (T, nv), float32 arithmetic."""
    q = np.asarray(qpos_trajectory, dtype=np.float32)
    dt32 = np.float32(dt)
    qvel_translation = (q[1:, :3] - q[:-1, :3]) / dt32
    # quaternion log in float64, then cast: float32 arccos near w=1 loses ~1e-3 rad/s; the
    # float64 route reproduces the shipped clip's angular_velocity golden to 3e-7.
    q64 = q[:, 3:7].astype(np.float64)
    diff = tr.quat_diff(q64[:-1], q64[1:])
    diff = diff / np.linalg.norm(diff, axis=-1, keepdims=True)
    qvel_gyro = (tr.quat_to_axisangle(diff) / float(dt)).astype(np.float32)
    qvel_joints = (q[1:, 7:] - q[:-1, 7:]) / dt32
    return np.concatenate([qvel_translation, qvel_gyro, qvel_joints], axis=1).astype(np.float32)


def process_qpos(model: mjcf.CompiledModel, mocap_qpos: np.ndarray, max_qvel: float = 20.0,
                 dt: float = 0.02) -> ReferenceClip:
    """Feature extraction for one clip given its (T, nq) qpos rows (mjx_preprocess.py:85-107,110-134)."""
    mocap_qpos = np.asarray(mocap_qpos, dtype=np.float32)
    T = mocap_qpos.shape[0]
    nbody = int(model.scalars["nbody"])
    xpos = np.zeros((T, nbody, 3), np.float32)
    xquat = np.zeros((T, nbody, 4), np.float32)
    qn = np.zeros_like(mocap_qpos)
    for t in range(T):
        fk = mjcf.forward_kinematics(model, mocap_qpos[t].astype(np.float64))
        xpos[t], xquat[t], qn[t] = fk["xpos"], fk["xquat"], fk["qpos"]
    padded = np.concatenate([mocap_qpos, mocap_qpos[-1:]], axis=0)
    qvel = compute_velocity_from_kinematics(padded, dt)
    qvel[:, 6:] = np.clip(qvel[:, 6:], -max_qvel, max_qvel)
    return ReferenceClip(
        position=qn[:, :3], quaternion=qn[:, 3:7], joints=qn[:, 7:], body_positions=xpos, body_quaternions=xquat,
        velocity=qvel[:, :3], angular_velocity=qvel[:, 3:6], joints_velocity=qvel[:, 6:],
    )


def process_clip(stac_path: str, scale_factor: float = 0.9, start_step: int = 0, clip_length: int = 250,
                 max_qvel: float = 20.0, dt: float = 0.02, mjcf_path: str = "./assets/rodent.xml",
                 model: Optional[mjcf.CompiledModel] = None) -> ReferenceClip:
    """Same signature as the reference's process_clip (mjx_preprocess.py:43-107).

    `stac_path` may be the reference's pickle of a plain dict {"qpos": (N, nq), ...}
    (only builtin/numpy types are accepted by the restricted loader) or an .npz with a
    `qpos` array.
    """
    if stac_path.endswith(".npz"):
        qpos = np.load(stac_path, allow_pickle=False)["qpos"]
    else:
        qpos = _load_stac_pickle(stac_path)["qpos"]
    qpos = np.asarray(qpos)[start_step:start_step + clip_length]
    if model is None:
        model = mjcf.compile_mjcf(mjcf_path, scale_factor=scale_factor)
    return process_qpos(model, qpos, max_qvel=max_qvel, dt=dt)


class _NumpyOnlyUnpickler(pickle.Unpickler):
    _OK = {("numpy.core.multiarray", "_reconstruct"), ("numpy._core.multiarray", "_reconstruct"),
           ("numpy", "ndarray"), ("numpy", "dtype"), ("numpy.core.multiarray", "scalar"),
           ("numpy._core.multiarray", "scalar")}

    def find_class(self, module, name):
        if (module, name) in self._OK:
            return getattr(__import__(module, fromlist=[name]), name)
        raise pickle.UnpicklingError(f"refused global {module}.{name} in STAC file")


def _load_stac_pickle(path: str) -> dict:
    with open(path, "rb") as f:
        return _NumpyOnlyUnpickler(f).load()
