"""GPU parity: HIP kernels (through the C-ABI) vs the CPU oracle on identical inputs.

  * one forward pass (reset): every output within 2e-5 of its array scale (2e-6 for kinematics / obs / traj);
  * one control step: test_step_matches_oracle below; the finer-grained checks (single substeps, per-stage outputs,
    glue on the product's own state, 4096 envs) are in tests/test_stage_parity.py; the method is in tests/parity.py.
"""
import numpy as np
import pytest
import torch

import helpers as H

pytestmark = pytest.mark.gpu


def _inputs(B, seed=0):
    rng = np.random.default_rng(seed)
    sf = rng.integers(0, 235, B).astype(np.int32)
    noise = (1e-3 * rng.standard_normal((B, 74))).astype(np.float32)
    act = np.clip(0.3 * rng.standard_normal((B, 30)), -1, 1).astype(np.float32)
    return sf, noise, act


@pytest.fixture(scope="module")
def env():
    from vnl_brax_imitation_amd.envs.rodent import RodentTracking

    assert torch.cuda.is_available()
    return RodentTracking(H.reference_clip(), num_envs=256, device="cuda:0", **H.env_kwargs())


def test_native_library_loaded(env):
    import ctypes

    assert isinstance(env._L, ctypes.CDLL) and "libvnl.so" in env._L._name


def test_reset_matches_oracle(env):
    B = env.num_envs
    sf, noise, _ = _inputs(B)
    st = env.reset(start_frame=torch.from_numpy(sf), noise=torch.from_numpy(noise))
    o = H.make_oracle(env, "f64")
    ost = o.env_reset(sf, noise)
    ps = st.pipeline_state
    for k, tol in [("qpos", 1e-6), ("qvel", 1e-6), ("xpos", 2e-6), ("qacc_warmstart", 2e-5), ("qfrc_actuator", 1e-6)]:
        got = getattr(ps, k).reshape(B, -1).cpu().numpy()
        assert H.scaled_err(got, ost[k]) < tol, (k, H.scaled_err(got, ost[k]))
    assert H.scaled_err(st.obs.cpu().numpy(), ost["obs"]) < 1e-6
    assert H.scaled_err(st.info["traj"].cpu().numpy(), ost["traj"]) < 2e-6
    assert H.scaled_err(st.info["termination_error"].cpu().numpy(), ost["termination_error"]) < 1e-5
    assert H.scaled_err(ps.subtree_com_root.cpu().numpy(), ost["com1"]) < 2e-6


def test_step_matches_oracle(env):
    """One control step (5 substeps + glue) against the oracles made to follow the product's solver decisions
    (tests/parity.py): every env within max(1e-5 of the array's scale, 50 x the float32 oracle's own deviation on that
    env), no quantiles; counters / done exact; rtrunk (computed from the OLD state) tight."""
    import parity as P

    B = env.num_envs
    sf, noise, act = _inputs(B, seed=1)
    o64, o32 = H.make_oracle(env, "f64"), H.make_oracle(env, "f32")
    st, err, dev, rep, ost = P.control_step_follow(env, o64, o32, sf, noise, act)
    print("\n[control step, 256 envs] " + ", ".join(f"{k}: max {v.max():.2e} median {np.median(v):.2e}" for k, v in err.items()))
    P.check_control_step(err, dev, rep)
    assert np.median(err["qvel"]) < 1e-5 and np.median(err["qacc_warmstart"]) < 1e-5
    assert np.array_equal(st.done.cpu().numpy(), ost["done"].astype(np.float32))
    assert np.array_equal(st.info["cur_frame"].cpu().numpy(), ost["cur_frame"])
    assert H.scaled_err(st.metrics["rtrunk"].cpu().numpy(), ost["metrics"][:, 2]) < 1e-5
    # absolute bound on the reward (the weighted sum is ~0.03; 2e-4 is what an env with a flipped decision may move it by),
    # and the median env at the float32 conditioning of its exp / arccos terms (measured on the device: max 2.4e-6, median 1.3e-7)
    rerr = np.abs(st.reward.cpu().numpy() - ost["reward"])
    assert rerr.max() < 2e-4 and np.median(rerr) < 5e-7, (rerr.max(), np.median(rerr))
    print("   vs the natural oracle:", P.natural_check(st, o64, o32, act))


def test_rollout_stays_finite_and_resets(env):
    from vnl_brax_imitation_amd.envs.wrappers import wrap

    w = wrap(env, episode_length=150)
    st = w.reset(3)
    g = torch.Generator().manual_seed(0)
    for _ in range(30):
        a = torch.clamp(0.3 * torch.randn((env.num_envs, 30), generator=g), -1, 1)
        st = w.step(st, a)
    assert torch.isfinite(st.obs).all() and torch.isfinite(st.reward).all()
    assert (st.info["sub_clip_frame"] == 30).all()
    assert st.done.min().item() == 1.0  # info is not reset -> done every step after 10
