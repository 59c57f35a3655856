"""Batched forward kinematics of clip preprocessing on the kernels (vnl_env_fk): reference
preprocessing/mjx_preprocess.py:85-107 scans mjx.kinematics over the frames of a clip.  Checked against the float64 NumPy
pass (which tests/test_model_golden.py pins to the reference's shipped clip data) -- CPU tier on the float32 host build,
`-m gpu` on the device incl. a multi-launch batch."""
import numpy as np
import pytest

import helpers as H
from vnl_brax_imitation_amd.preprocessing import mjx_preprocess as pp


def _check(device, lib=None, frames=250, chunk=4096):
    m, q = H.model(), H.golden_qpos()[:frames]
    ref = pp.process_qpos(m, q)
    with H.backend(lib):
        qn, xpos, xquat, com = pp.forward_kinematics_device(m, q, device=device, chunk=chunk)
    scale = np.abs(ref.body_positions).max()
    assert np.abs(xpos - ref.body_positions).max() / scale < 2e-6
    assert np.abs(com - ref.center_of_mass).max() / scale < 2e-6
    # quaternions up to sign, normalised root quaternion written back
    dq = np.minimum(np.abs(xquat - ref.body_quaternions).max(-1), np.abs(xquat + ref.body_quaternions).max(-1))
    assert dq[:, 1:].max() < 2e-6
    assert np.abs(qn[:, 3:7] - ref.quaternion).max() < 1e-6 and np.array_equal(qn[:, :3], ref.position)
    return ref


def test_device_fk_matches_numpy_fk_on_host_build():
    _check("cpu", lib=H.hostsim_library("float"), frames=40, chunk=16)  # 3 launches, the last one ragged


@pytest.mark.gpu
def test_device_fk_matches_numpy_fk_on_gpu():
    ref = _check("cuda:0", frames=250, chunk=96)  # 3 launches
    clip = pp.process_qpos(H.model(), H.golden_qpos(), fk_device="cuda:0")
    for k in ("position", "quaternion", "joints", "velocity", "angular_velocity", "joints_velocity"):
        assert np.abs(getattr(clip, k) - getattr(ref, k)).max() < 1e-6, k
    assert np.abs(clip.body_positions - ref.body_positions).max() < 1e-6 * 10
