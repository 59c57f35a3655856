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
    center_of_mass: Optional[np.ndarray] = None  # subtree_com of the root body (mocap_preprocess.py:334; envs/humanoid.py:273)

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
        return ReferenceClip(**{f.name: (None if getattr(clips[0], f.name) is None else
                                         np.stack([getattr(c, f.name) for c in clips]))
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


def _qmul(u: np.ndarray, v: np.ndarray) -> np.ndarray:
    return tr.quat_mul(u, v)


def _qrot(v: np.ndarray, q: np.ndarray) -> np.ndarray:
    """MJX math.rotate, batched: v (..., 3) by q (..., 4)."""
    s, u = q[..., :1], q[..., 1:]
    return 2.0 * np.sum(u * v, -1, keepdims=True) * u + (s * s - np.sum(u * u, -1, keepdims=True)) * v + \
        2.0 * s * np.cross(u, v)


def forward_kinematics_batch(model: mjcf.CompiledModel, qpos: np.ndarray):
    """smooth.kinematics for N frames at once (float64): the per-frame loop of the reference's process_clip
    (mjx_preprocess.py:85-107 scans mjx.kinematics over the frames) vectorised over the frame axis; same arithmetic as
    model.mjcf.forward_kinematics.  Returns (qpos with normalised root quaternion, xpos (N, nbody, 3), xquat (N, nbody, 4))."""
    a = model.arrays
    nbody = int(model.scalars["nbody"])
    q = np.array(qpos, dtype=np.float64)
    N = q.shape[0]
    xpos = np.zeros((N, nbody, 3))
    xquat = np.zeros((N, nbody, 4))
    xquat[:, 0, 0] = 1.0
    for b in range(1, nbody):
        p = int(a["body_parentid"][b])
        pos = xpos[:, p] + _qrot(np.broadcast_to(a["body_pos"][b], (N, 3)), xquat[:, p])
        quat = _qmul(xquat[:, p], np.broadcast_to(a["body_quat"][b], (N, 4)))
        for k in range(int(a["body_jntnum"][b])):
            j = int(a["body_jntadr"][b]) + k
            qa = int(a["jnt_qposadr"][j])
            if a["jnt_type"][j] == mjcf.JNT_FREE:
                pos = q[:, qa:qa + 3].copy()
                quat = q[:, qa + 3:qa + 7] / np.linalg.norm(q[:, qa + 3:qa + 7], axis=1, keepdims=True)
                q[:, qa + 3:qa + 7] = quat
            else:
                jp, ax = np.broadcast_to(a["jnt_pos"][j], (N, 3)), a["jnt_axis"][j]
                anchor = _qrot(jp, quat) + pos
                half = 0.5 * (q[:, qa] - a["qpos0"][qa])
                dq = np.concatenate([np.cos(half)[:, None], np.sin(half)[:, None] * ax[None, :]], axis=1)
                quat = _qmul(quat, dq)
                pos = anchor - _qrot(jp, quat)
        xpos[:, b] = pos
        xquat[:, b] = quat / np.linalg.norm(quat, axis=1, keepdims=True)
    return q, xpos, xquat


def forward_kinematics_device(model: mjcf.CompiledModel, qpos: np.ndarray, device="cuda", chunk: int = 4096):
    """The same as forward_kinematics_batch on the GPU (float32): the frames go through `vnl_env_fk`, one frame per
    wavefront, `chunk` frames per launch (the batched FK kernel of SURVEY 8(f) f1).  Returns (qpos with normalised root
    quaternion, xpos (N, nbody, 3), xquat (N, nbody, 4), subtree centre of mass of the root body (N, 3))."""
    import torch

    from ..envs.rodent import RodentTracking

    q = np.ascontiguousarray(qpos, dtype=np.float32)
    N, nq = q.shape
    nbody, names = int(model.scalars["nbody"]), model.names
    B = min(int(chunk), N)
    z = lambda *s: np.zeros(s, dtype=np.float32)  # noqa: E731  (a one-frame placeholder clip: FK reads no clip data)
    nj = nq - 7
    dummy = ReferenceClip(position=z(1, 3), quaternion=np.tile(np.array([1, 0, 0, 0], np.float32), (1, 1)), joints=z(1, nj),
                          body_positions=z(1, nbody, 3), body_quaternions=z(1, nbody, 4), velocity=z(1, 3),
                          angular_velocity=z(1, 3), joints_velocity=z(1, nj), center_of_mass=z(1, 3))
    body = [n for n in names["body"] if n != "world"]
    env = RodentTracking(dummy, end_eff_names=[], appendage_names=[], walker_body_names=body[:1], joint_names=[],
                         center_of_mass=body[0], model=model, clip_length=1, sub_clip_length=1, ref_traj_length=1,
                         num_envs=B, device=device)
    out_q, xpos, xquat, com = q.copy(), z(N, nbody, 3), z(N, nbody, 4), z(N, 3)
    st = None
    for a in range(0, N, B):
        rows = q[a:a + B]
        n = len(rows)
        if n < B:
            rows = np.concatenate([rows, np.repeat(rows[-1:], B - n, axis=0)])
        st = env.forward_kinematics(torch.from_numpy(rows), out=st)
        ps = st.pipeline_state
        out_q[a:a + n, 3:7] = ps.qpos[:n, 3:7].cpu().numpy()
        xpos[a:a + n] = ps.xpos[:n].reshape(n, nbody, 3).cpu().numpy()
        xquat[a:a + n] = ps.xquat[:n].reshape(n, nbody, 4).cpu().numpy()
        com[a:a + n] = ps.subtree_com_root[:n].cpu().numpy()
    return out_q, xpos, xquat, com


def process_qpos(model: mjcf.CompiledModel, mocap_qpos: np.ndarray, max_qvel: float = 20.0,
                 dt: float = 0.02, fk_device=None) -> ReferenceClip:
    """Feature extraction for one clip given its (T, nq) qpos rows (mjx_preprocess.py:85-107,110-134).  `fk_device`
    ("cuda", ...): run the forward kinematics on the GPU (forward_kinematics_device) instead of the float64 NumPy pass."""
    mocap_qpos = np.asarray(mocap_qpos, dtype=np.float32)
    if fk_device is not None:
        qn, xpos, xquat, com = forward_kinematics_device(model, mocap_qpos, device=fk_device)
        padded = np.concatenate([mocap_qpos, mocap_qpos[-1:]], axis=0)
        qvel = compute_velocity_from_kinematics(padded, dt)
        qvel[:, 6:] = np.clip(qvel[:, 6:], -max_qvel, max_qvel)
        return ReferenceClip(
            position=qn[:, :3], quaternion=qn[:, 3:7], joints=qn[:, 7:], body_positions=xpos, body_quaternions=xquat,
            velocity=qvel[:, :3], angular_velocity=qvel[:, 3:6], joints_velocity=qvel[:, 6:], center_of_mass=com)
    qn, xpos, xquat = forward_kinematics_batch(model, mocap_qpos.astype(np.float64))
    qn, xpos, xquat = qn.astype(np.float32), xpos.astype(np.float32), xquat.astype(np.float32)
    padded = np.concatenate([mocap_qpos, mocap_qpos[-1:]], axis=0)
    qvel = compute_velocity_from_kinematics(padded, dt)
    qvel[:, 6:] = np.clip(qvel[:, 6:], -max_qvel, max_qvel)
    # subtree centre of mass of the root body (body 1): mass-weighted mean of the bodies' inertial-frame origins
    a = model.arrays
    mass = np.asarray(a["body_mass"], dtype=np.float64)
    xipos = xpos.astype(np.float64) + _qrot(np.broadcast_to(a["body_ipos"], xpos.shape), xquat.astype(np.float64))
    sub = np.asarray(a["body_rootid"]) == 1
    com = (mass[sub, None] * xipos[:, sub]).sum(1) / mass[sub].sum()
    return ReferenceClip(
        position=qn[:, :3], quaternion=qn[:, 3:7], joints=qn[:, 7:], body_positions=xpos, body_quaternions=xquat,
        velocity=qvel[:, :3], angular_velocity=qvel[:, 3:6], joints_velocity=qvel[:, 6:], center_of_mass=com.astype(np.float32),
    )


def synthesize_clips(model: mjcf.CompiledModel, qpos: np.ndarray, num_clips: int, seed: int = 0,
                     max_qvel: float = 20.0, dt: float = 0.02) -> ReferenceClip:
    """Multi-clip container of `num_clips` clips synthesised from ONE clip's (T, nq) qpos rows (SURVEY 8(d) config 4:
    the reference has only a stub for its multi-clip env, rodent.py:473-475, and its 842-clip file is not shipped): clip 0
    is the source itself; every further clip is the source under a random yaw rotation about the vertical axis, a random
    horizontal translation and, for about half of them, time reversal (all seeded), re-run through process_qpos."""
    rng = np.random.default_rng(seed)
    src = np.asarray(qpos, dtype=np.float64)
    clips = []
    for c in range(num_clips):
        q = src.copy()
        if c > 0:
            yaw, shift, rev = rng.uniform(-np.pi, np.pi), rng.uniform(-0.5, 0.5, 2), bool(rng.integers(0, 2))
            if rev:
                q = q[::-1].copy()
            cz, sz = np.cos(yaw), np.sin(yaw)
            x, y = q[:, 0].copy(), q[:, 1].copy()
            q[:, 0], q[:, 1] = cz * x - sz * y + shift[0], sz * x + cz * y + shift[1]
            qy = np.array([np.cos(0.5 * yaw), 0.0, 0.0, np.sin(0.5 * yaw)])
            q[:, 3:7] = tr.quat_mul(np.broadcast_to(qy, (len(q), 4)), q[:, 3:7])
        clips.append(process_qpos(model, q, max_qvel=max_qvel, dt=dt))
    return ReferenceClip.stack(clips)


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
