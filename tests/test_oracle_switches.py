"""The oracle's named switches for the items of the MJX restatement that are recalled rather than verified (SURVEY.md
Appendix B, least-certain items 1-5 and 8).  Dynamics parity is unpinned (no MJX here, the reference ships no tests), so
each switch documents -- and this file asserts -- WHICH outputs move under the alternative reading and by how much: a later
run against real MJX can then settle the items one at a time.  Defaults = the reading the product implements."""
import numpy as np
import pytest

import helpers as H


@pytest.fixture(scope="module")
def setup():
    env = H.hostsim_env(2, "double")  # only for the env spec / clip / model
    rng = np.random.default_rng(3)
    B = 24
    sf = rng.integers(0, 235, B).astype(np.int32)
    noise = 1e-3 * rng.standard_normal((B, 74))
    acts = np.clip(0.3 * rng.standard_normal((2, B, 30)), -1, 1)
    return env, sf, noise, acts


def _run(env, sf, noise, acts, **opts):
    o = H.make_oracle(env, "f64")
    for k, v in opts.items():
        o.set_option(k, v)
    try:
        st = o.env_reset(sf, noise)
        first = {k: v.copy() for k, v in st.items()}
        for a in acts:
            o.env_step(st, a)
    finally:
        for k in opts:  # the options are process-wide for the library: restore the defaults
            o.set_option(k, dict(quat_writeback=1, capsule_frame_axis=1, ls_mid_first=0, ls_tie_lo=0, inactive_pos_zero=1,
                                 contact_rows_by_type=0, reset_warmstart_zero=0)[k])
    return first, st


def _moved(a, b, keys):
    return {k: float(np.abs(np.asarray(a[k], float) - np.asarray(b[k], float)).max() / max(np.abs(np.asarray(b[k], float)).max(), 1e-30))
            for k in keys}


KEYS = ("qpos", "qvel", "qacc_warmstart", "obs", "traj", "reward")


def test_defaults_are_what_the_product_implements(setup):
    env, *_ = setup
    o = H.make_oracle(env, "f64")
    assert [o.get_option(k) for k in o.OPTIONS] == [1, 1, 0, 0, 1, 0, 0]


def test_item1_quaternion_writeback_moves_only_the_quaternion_slots_after_reset(setup):
    """B.1: does kinematics write the normalised free-joint quaternion back into qpos?  Reset adds noise to the quaternion
    components (C.12), so right after reset obs[3:7] differs by the normalisation (~1e-3); the physics does not care."""
    env, sf, noise, acts = setup
    (r0, s0), (r1, s1) = _run(env, sf, noise, acts), _run(env, sf, noise, acts, quat_writeback=0)
    d = np.abs(r0["qpos"] - r1["qpos"])
    assert d[:, 3:7].max() > 1e-5 and np.delete(d, [3, 4, 5, 6], axis=1).max() == 0.0
    assert np.abs(r0["obs"][:, 3:7] - r1["obs"][:, 3:7]).max() > 1e-5
    m = _moved(r0, r1, ("xpos", "qacc_warmstart", "traj"))
    assert max(m.values()) < 1e-12, m  # normalised inside kinematics either way
    m2 = _moved(s0, s1, ("qpos", "qvel"))  # after steps: the integrator renormalises -> identical again up to rounding
    print("item 1 (quat write-back off): after reset qpos[3:7] moves by", d[:, 3:7].max(), "; after 2 steps", m2)
    assert max(m2.values()) < 1e-3  # (a 1e-16 difference of the normalisation order, amplified over two control steps)


def test_item2_capsule_tangent_frame_moves_nothing_but_rounding(setup):
    """Tangent frame of plane-capsule contacts: capsule axis projected on the plane vs make_frame(n).  mu1 == mu2, so the
    friction pyramid is the same set of directions rotated about the normal: the unconverged CG result changes."""
    env, sf, noise, acts = setup
    (r0, s0), (r1, s1) = _run(env, sf, noise, acts), _run(env, sf, noise, acts, capsule_frame_axis=0)
    m0, m = _moved(r0, r1, KEYS[:3]), _moved(s0, s1, KEYS)
    print("item 2 (capsule frame = make_frame(n)): reset", m0, "after 2 steps", m)
    assert m0["qpos"] == 0.0  # reset state itself is the clip frame
    assert m["qvel"] > 0  # a rotated pyramid is a different (equally valid) discretisation of the cone


def test_item3_line_search_test_order_and_tie_break(setup):
    env, sf, noise, acts = setup
    base = _run(env, sf, noise, acts)
    for opt in ("ls_mid_first", "ls_tie_lo"):
        alt = _run(env, sf, noise, acts, **{opt: 1})
        m = _moved(base[1], alt[1], KEYS)
        print(f"item 3 ({opt}): after 2 steps", m)
        # exact cost ties do occur (a bracket whose two ends evaluate to the same float64 cost); either way the effect after two
        # control steps stays small
        assert m["qpos"] < 1e-2 and m["qvel"] < 1e-1, m


def test_item4_inactive_row_pos_moves_nothing(setup):
    """Rows masked out by make_constraint: pos zeroed or kept.  Their Jacobian is zero and their Jaref is >= 0 either way, so
    they never contribute: outputs identical."""
    env, sf, noise, acts = setup
    (r0, s0), (r1, s1) = _run(env, sf, noise, acts), _run(env, sf, noise, acts, inactive_pos_zero=0)
    m = _moved(s0, s1, KEYS)
    print("item 4 (inactive rows keep pos):", m)
    assert max(m.values()) == 0.0


def test_item5_contact_row_order_changes_only_summation_order(setup):
    env, sf, noise, acts = setup
    (r0, s0), (r1, s1) = _run(env, sf, noise, acts), _run(env, sf, noise, acts, contact_rows_by_type=1)
    m0, m = _moved(r0, r1, KEYS[:3]), _moved(s0, s1, KEYS)
    print("item 5 (contact rows grouped by geom type): reset", m0, "after 2 steps", m)
    assert m0["qacc_warmstart"] < 1e-9  # float64 summation order only ... amplified by the dynamics afterwards
    assert m["qpos"] < 1e-3


def test_item8_warmstart_after_reset_affects_the_first_substep_only_through_the_start_point(setup):
    """qacc_warmstart after pipeline_init: the init solve's qacc (default, MJX solver.solve writes it) vs zeros (make_data)."""
    env, sf, noise, acts = setup
    (r0, s0), (r1, s1) = _run(env, sf, noise, acts), _run(env, sf, noise, acts, reset_warmstart_zero=1)
    assert np.abs(r1["qacc_warmstart"]).max() == 0.0 and np.abs(r0["qacc_warmstart"]).max() > 1.0
    m0 = _moved(r0, r1, ("qpos", "qvel", "xpos", "obs", "traj"))
    assert max(m0.values()) == 0.0  # nothing else of the reset state depends on it
    m = _moved(s0, s1, KEYS)
    print("item 8 (warm start zero after reset): after 2 steps", m)
    assert m["qvel"] > 0  # a different starting point of an unconverged 6-iteration CG
