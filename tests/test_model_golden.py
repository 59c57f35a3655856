"""Model compiler + clip preprocessing against the reference's only shipped data file
(clips/transform_snips_groom.p -> tests/golden/groom_clip.npz, made by tools/make_fixtures.py).
KATs of SURVEY.md Appendix D."""
import os

import numpy as np
import pytest

import helpers as H
from vnl_brax_imitation_amd.model import mjcf
from vnl_brax_imitation_amd.preprocessing import mjx_preprocess as pp

REF_XML = "/root/reference/assets/rodent.xml"


def _qpos(g, t):
    return np.concatenate([g["position"][t], g["quaternion"][t], g["joints"][t]]).astype(np.float64)


def test_dimensions():
    s = H.model().scalars
    assert (s["nq"], s["nv"], s["nu"], s["nbody"], s["njnt"]) == (74, 73, 30, 66, 68)
    assert (s["ncg"], s["ncon"], s["nlimit"], s["nefc"]) == (32, 59, 67, 303)
    assert abs(H.model().body_mass.sum() - 0.1867913) < 1e-6
    assert s["timestep"] == 0.002 and s["iterations"] == 6 and s["ls_iterations"] == 6 and s["eulerdamp"] == 1
    m = H.model()
    names = ["torso", "pelvis", "upper_leg_L", "lower_leg_L", "foot_L", "upper_leg_R", "lower_leg_R", "foot_R", "skull",
             "jaw", "scapula_L", "upper_arm_L", "lower_arm_L", "finger_L", "scapula_R", "upper_arm_R", "lower_arm_R",
             "finger_R"]
    assert [m.body_id(n) for n in names] == H.BODY_IDXS


def test_fk_and_com_match_clip_goldens():
    m, g = H.model(), H.golden_clip()
    fk_err = q_err = com_err = app_err = 0.0
    app = [m.body_id(n) for n in ("lower_arm_R", "lower_arm_L", "foot_R", "foot_L", "skull")]  # walker.py:183-190,365
    for t in range(0, 250, 3):
        fk = mjcf.forward_kinematics(m, _qpos(g, t))
        fk_err = max(fk_err, np.abs(fk["xpos"][H.BODY_IDXS] - g["body_positions"][t]).max())
        a, b = fk["xquat"][H.BODY_IDXS], g["body_quaternions"][t]
        q_err = max(q_err, np.minimum(np.abs(a - b).max(-1), np.abs(a + b).max(-1)).max())
        com = mjcf.subtree_com(m, fk["xipos"])[1]
        com_err = max(com_err, np.abs(com - g["center_of_mass"][t]).max())
        ego = (fk["xpos"][app] - fk["xpos"][1]) @ fk["xmat"][1]
        app_err = max(app_err, np.abs(ego - g["appendages"][t]).max())
    assert fk_err < 5e-8 and q_err < 2e-7 and com_err < 5e-8 and app_err < 5e-8, (fk_err, q_err, com_err, app_err)


def test_velocity_goldens_bit_exact():
    g = H.golden_clip()
    q = np.concatenate([g["position"], g["quaternion"], g["joints"]], axis=1)
    v = pp.compute_velocity_from_kinematics(np.concatenate([q, q[-1:]]), 0.02)
    v[:, 6:] = np.clip(v[:, 6:], -20.0, 20.0)
    assert np.array_equal(v[:, :3], g["velocity"])
    assert np.array_equal(v[:, 6:], g["joints_velocity"])
    assert np.abs(v[:, 3:6] - g["angular_velocity"]).max() < 1e-6
    assert np.all(v[-1] == 0)  # pad-with-last-frame rule, mjx_preprocess.py:93


def test_reference_clip_has_all_bodies_and_matches_goldens():
    c, g = H.reference_clip(), H.golden_clip()
    assert c.body_positions.shape == (250, 66, 3) and c.body_quaternions.shape == (250, 66, 4)
    assert np.abs(c.body_positions[:, H.BODY_IDXS] - g["body_positions"]).max() < 5e-8
    assert np.array_equal(c.body_positions[:, 1], c.position)  # torso is the free-joint body
    assert np.abs(np.linalg.norm(c.quaternion, axis=1) - 1).max() < 1e-6


@pytest.mark.skipif(not os.path.exists(REF_XML), reason="reference assets only exist in the build container")
def test_packaged_model_is_the_compiled_reference_xml():
    fresh = mjcf.compile_mjcf(REF_XML)
    pk = H.model()
    for k, v in fresh.arrays.items():
        assert np.array_equal(v, pk.arrays[k]), k
    # the explicit-only rescale reading is the discriminating one
    wrong = mjcf.compile_mjcf(REF_XML, rescale_defaults=True)
    g = H.golden_clip()
    fk = mjcf.forward_kinematics(wrong, _qpos(g, 0))
    assert np.abs(fk["xpos"][H.BODY_IDXS] - g["body_positions"][0]).max() > 5e-5


def test_blob_roundtrip_through_abi_parser():
    from vnl_brax_imitation_amd.model import blob

    b = blob.to_blob(H.model())
    assert b[:8] == b"VNLMDL01"
    from oracle.oracle import Oracle

    o = Oracle(b, "f64")  # parses every section it needs or raises
    assert o.field("qpos").shape == (74,)
