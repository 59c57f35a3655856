"""Size-independent properties of the HIP rollout at BASELINE's full size (4096 envs per GPU), where the
CPU oracle would take minutes: determinism, independence of an env from its position in the batch and from
the batch size, per-env clip selection, and the end-of-clip edge case against the oracle."""
import numpy as np
import pytest
import torch

import helpers as H

pytestmark = pytest.mark.gpu


def _make(B, **kw):
    from vnl_brax_imitation_amd.envs.rodent import RodentTracking

    args = H.env_kwargs()
    args.update(kw)
    clip = args.pop("reference_clip", None) or H.reference_clip()
    return RodentTracking(clip, num_envs=B, device="cuda:0", **args)


def _inputs(n, seed=3):
    rng = np.random.default_rng(seed)
    sf = rng.integers(0, 235, n).astype(np.int32)
    noise = (1e-3 * rng.standard_normal((n, 74))).astype(np.float32)
    acts = np.clip(0.3 * rng.standard_normal((3, n, 30)), -1, 1).astype(np.float32)
    return sf, noise, acts


def _run(env, sf, noise, acts):
    st = env.reset(start_frame=torch.from_numpy(sf), noise=torch.from_numpy(noise))
    for a in acts:
        st = env.step(st, torch.from_numpy(a))
    ps = st.pipeline_state
    return {"qpos": ps.qpos.clone(), "qvel": ps.qvel.clone(), "warm": ps.qacc_warmstart.clone(), "obs": st.obs.clone(),
            "traj": st.info["traj"].clone(), "reward": st.reward.clone(), "done": st.done.clone(),
            "frame": st.info["cur_frame"].clone()}


def test_full_size_is_deterministic_and_position_independent():
    B, n = 4096, 64
    sf, noise, acts = _inputs(n)
    rep = lambda a: np.tile(a, (B // n,) + (1,) * (a.ndim - 1))  # noqa: E731  env k, k+64, k+128 ... share inputs
    big = _make(B)
    r1 = _run(big, rep(sf), rep(noise), np.stack([rep(a) for a in acts]))
    r2 = _run(big, rep(sf), rep(noise), np.stack([rep(a) for a in acts]))
    small = _run(_make(n), sf, noise, acts)
    for k, v in r1.items():
        assert torch.equal(v, r2[k]), f"{k}: two identical runs differ"
        blocks = v.reshape(B // n, n, *v.shape[1:])
        assert torch.equal(blocks, blocks[:1].expand_as(blocks)), f"{k}: result depends on the position in the batch"
        assert torch.equal(blocks[0], small[k]), f"{k}: result depends on the batch size"
    assert torch.isfinite(r1["obs"]).all() and torch.isfinite(r1["qvel"]).all()


def test_multi_clip_selects_per_env_clip_on_gpu():
    from vnl_brax_imitation_amd.preprocessing.mjx_preprocess import ReferenceClip

    c0 = H.reference_clip()
    shift = np.array([0.05, -0.02, 0.0], dtype=np.float32)
    c1 = c0.replace(position=c0.position + shift, body_positions=c0.body_positions + shift)
    env = _make(128, reference_clip=ReferenceClip.stack([c0, c1]))
    sf = torch.full((128,), 10, dtype=torch.int32)
    cid = (torch.arange(128) % 2).to(torch.int32)
    st = env.reset(start_frame=sf, noise=torch.zeros(128, 74), clip_id=cid)
    q = st.pipeline_state.qpos.cpu().numpy()
    assert np.allclose(q[1::2, :3] - q[0::2, :3], shift, atol=1e-7)
    tr = st.info["traj"].cpu()
    assert torch.allclose(tr[0::2, 75:345], tr[1::2, 75:345], atol=1e-6)  # egocentric features: translation-free
    st = env.step(st, torch.zeros(128, 30))
    assert torch.isfinite(st.obs).all() and (st.info["cur_frame"].cpu() == 11).all()


def test_end_of_clip_matches_oracle():
    """start frames at the very end of the clip: the reference-trajectory window and the reward row clamp
    (JAX gather semantics); frame counters keep counting."""
    B = 64
    env = _make(B)
    T = int(env.clip_arrays(0)["position"].shape[0])
    sf = (T - 1 - (np.arange(B) % 4)).astype(np.int32)
    noise = np.zeros((B, 74), np.float32)
    act = np.zeros((B, 30), np.float32)
    st = env.reset(start_frame=torch.from_numpy(sf), noise=torch.from_numpy(noise))
    o = H.make_oracle(env, "f64")
    ost = o.env_reset(sf, noise)
    assert H.scaled_err(st.info["traj"].cpu().numpy(), ost["traj"]) < 2e-6
    for _ in range(2):
        st = env.step(st, torch.from_numpy(act))
        o.env_step(ost, act)
    assert np.array_equal(st.info["cur_frame"].cpu().numpy(), ost["cur_frame"])
    assert np.array_equal(st.done.cpu().numpy(), ost["done"].astype(np.float32))
    assert H.scaled_err(st.info["traj"].cpu().numpy(), ost["traj"]) < 1e-3
    assert torch.isfinite(st.obs).all()


def test_multi_clip_at_full_size_64_clips_4096_envs():
    """SURVEY 8(d) config 4, per-GPU share: 64 synthesised clips (yaw / shift / time reversal of the groom clip, seeded,
    through process_qpos), 4096 envs with a random clip each.  At full size: every env must behave exactly like a single-clip
    env of its own clip (bit for bit -- the clip index only selects rows); on a 64-env subsample: reset + control step against
    the oracle bound to that env's clip."""
    import parity as P
    from vnl_brax_imitation_amd.preprocessing import mjx_preprocess as pp

    C, B = 64, 4096
    multi = pp.synthesize_clips(H.model(), H.golden_qpos(), C, seed=0)
    rng = np.random.default_rng(11)
    cid = rng.integers(0, C, B).astype(np.int32)
    sf = rng.integers(0, 235, B).astype(np.int32)
    noise = (1e-3 * rng.standard_normal((B, 74))).astype(np.float32)
    act = np.clip(0.3 * rng.standard_normal((B, 30)), -1, 1).astype(np.float32)
    env = _make(B, reference_clip=multi)
    st = env.reset(start_frame=torch.from_numpy(sf), noise=torch.from_numpy(noise), clip_id=torch.from_numpy(cid))
    st = env.step(st, torch.from_numpy(act))
    assert torch.isfinite(st.obs).all() and torch.isfinite(st.info["traj"]).all()
    assert torch.equal(st.info["clip_id"].cpu(), torch.from_numpy(cid))
    # three clips, every env that tracks them: identical to a single-clip env
    for c in (0, int(cid[1]), C - 1):
        idx = np.where(cid == c)[0]
        if len(idx) == 0:
            continue
        single = pp.ReferenceClip(**{f: (None if getattr(multi, f) is None else getattr(multi, f)[c]) for f in
                                     ("position", "quaternion", "joints", "body_positions", "velocity", "joints_velocity",
                                      "angular_velocity", "body_quaternions", "center_of_mass")})
        e1 = _make(len(idx), reference_clip=single)
        s1 = e1.reset(start_frame=torch.from_numpy(sf[idx]), noise=torch.from_numpy(noise[idx]))
        s1 = e1.step(s1, torch.from_numpy(act[idx]))
        ti = torch.from_numpy(idx).to(st.obs.device)
        for name, a, b in (("qpos", st.pipeline_state.qpos, s1.pipeline_state.qpos), ("obs", st.obs, s1.obs),
                           ("traj", st.info["traj"], s1.info["traj"]), ("reward", st.reward, s1.reward)):
            assert torch.equal(a[ti], b), (c, name)
        if c == int(cid[1]):  # against the oracle bound to this clip, following the product's decisions
            o64, o32 = H.make_oracle(e1, "f64"), H.make_oracle(e1, "f32")
            _, err, dev, rep, _ = P.control_step_follow(e1, o64, o32, sf[idx], noise[idx], act[idx])
            P.check_control_step(err, dev, rep, verbose=False)
