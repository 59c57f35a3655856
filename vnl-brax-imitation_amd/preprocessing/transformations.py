"""Quaternion helpers used by clip preprocessing (NumPy, host-side, once per clip).

Counterpart of reference preprocessing/transformations.py:30-139 (quat_mul, quat_conj,
quat_diff, quat_to_axisangle), restated for batches of quaternions [w, x, y, z].
"""
from __future__ import annotations

import numpy as np

_TOL = 1e-10  # transformations.py:8


def quat_mul(q1: np.ndarray, q2: np.ndarray) -> np.ndarray:
    """Hamilton product, any leading batch dims (transformations.py:30-52)."""
    w1, x1, y1, z1 = np.moveaxis(q1, -1, 0)
    w2, x2, y2, z2 = np.moveaxis(q2, -1, 0)
    return np.stack(
        [
            w1 * w2 - x1 * x2 - y1 * y2 - z1 * z2,
            w1 * x2 + x1 * w2 + y1 * z2 - z1 * y2,
            w1 * y2 - x1 * z2 + y1 * w2 + z1 * x2,
            w1 * z2 + x1 * y2 - y1 * x2 + z1 * w2,
        ],
        axis=-1,
    )


def quat_conj(q: np.ndarray) -> np.ndarray:
    """transformations.py:83-99."""
    return q * np.array([1.0, -1.0, -1.0, -1.0], dtype=q.dtype)


def quat_diff(source: np.ndarray, target: np.ndarray) -> np.ndarray:
    """Rotation from source to target, body frame: conj(source) * target (transformations.py:102-115)."""
    return quat_mul(quat_conj(source), target)


def quat_to_axisangle(q: np.ndarray) -> np.ndarray:
    """Axis * angle, with the reference's `angle < 1e-10 -> 0` branch and the
    (angle + pi) mod 2pi - pi wrap (transformations.py:118-139).  Batched."""
    angle = 2.0 * np.arccos(np.clip(q[..., 0], -1.0, 1.0))
    qn = np.sin(angle / 2.0)
    wrapped = np.mod(angle + np.pi, 2.0 * np.pi) - np.pi
    safe = np.where(np.abs(qn) > 0, qn, 1.0)
    out = q[..., 1:4] / safe[..., None] * wrapped[..., None]
    return np.where((angle < _TOL)[..., None], 0.0, out)
