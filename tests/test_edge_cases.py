"""Edge cases of the env entry points: NaN in the inputs (reference envs/rodent.py:216-226: reward / obs through nan_to_num,
done = 1 when the new pipeline state holds a NaN), a batch of one, ragged batch sizes, argument validation.
CPU tier on the float32 host build of the kernels; the same checks on the device under -m gpu."""
import numpy as np
import pytest
import torch

import helpers as H


def _env(B, device="cpu"):
    from vnl_brax_imitation_amd.envs.rodent import RodentTracking

    if device == "cpu":
        return H.hostsim_env(B)
    return RodentTracking(H.reference_clip(), num_envs=B, device=device, **H.env_kwargs())


def _inputs(B, seed=5):
    rng = np.random.default_rng(seed)
    sf = rng.integers(0, 200, B).astype(np.int32)
    noise = (1e-3 * rng.standard_normal((B, 74))).astype(np.float32)
    act = np.clip(0.3 * rng.standard_normal((B, 30)), -1, 1).astype(np.float32)
    return torch.from_numpy(sf), torch.from_numpy(noise), torch.from_numpy(act)


def _snapshot(st):
    ps = st.pipeline_state
    return {k: v.detach().cpu().clone() for k, v in dict(qpos=ps.qpos, qvel=ps.qvel, warm=ps.qacc_warmstart, obs=st.obs,
                                                         traj=st.info["traj"], reward=st.reward, done=st.done).items()}


def _nan_action_case(device):
    B, bad = 8, 2
    sf, noise, act = _inputs(B)
    env = _env(B, device)
    clean = _snapshot(env.step(env.reset(start_frame=sf, noise=noise), act))
    act_nan = act.clone()
    act_nan[bad, 3] = float("nan")
    got = _snapshot(env.step(env.reset(start_frame=sf, noise=noise), act_nan))
    # the poisoned env: done, finite reward and observation (nan_to_num), NaN left in the physics state as in the reference
    assert got["done"][bad] == 1.0 and torch.isfinite(got["reward"][bad]) and torch.isfinite(got["obs"][bad]).all()
    assert torch.isnan(got["qvel"][bad]).any()
    # every other env is untouched, bit for bit
    keep = [i for i in range(B) if i != bad]
    for k in clean:
        assert torch.equal(got[k][keep], clean[k][keep]), k
    # the oracle takes the same decisions on the same inputs
    o = H.make_oracle(env, "f32")
    ost = o.env_reset(sf.numpy(), noise.numpy())
    o.env_step(ost, act_nan.numpy())
    assert ost["done"][bad] == 1.0 and np.isfinite(ost["reward"][bad]) and np.isfinite(ost["obs"][bad]).all()
    assert np.array_equal(got["done"].numpy(), ost["done"].astype(np.float32))


def _ragged_sizes_case(device):
    sf, noise, act = _inputs(64)
    ref = _snapshot((lambda e: e.step(e.reset(start_frame=sf, noise=noise), act))(_env(64, device)))
    for B in (1, 3, 17):  # one env; sizes that are not multiples of anything the kernels tile by
        env = _env(B, device)
        got = _snapshot(env.step(env.reset(start_frame=sf[:B], noise=noise[:B]), act[:B]))
        for k in ref:
            assert torch.equal(got[k], ref[k][:B]), (B, k)


def _validation_case(device):
    env = _env(4, device)
    sf, noise, act = _inputs(4)
    st = env.reset(start_frame=sf, noise=noise)
    with pytest.raises(ValueError):
        env.step(st, act[:3])  # wrong batch
    with pytest.raises(ValueError):
        env.step(st, act[:, :29])  # wrong action width
    # start frames beyond the clip are clamped like a JAX gather (SURVEY C.4/C.5 semantics), not an error
    far = torch.full((4,), 10_000, dtype=torch.int32)
    st = env.reset(start_frame=far, noise=noise)
    assert torch.isfinite(st.obs).all() and (st.info["cur_frame"] == 10_000).all()


def test_nan_action_is_contained_to_its_env():
    _nan_action_case("cpu")


def test_batch_of_one_and_ragged_batch_sizes():
    _ragged_sizes_case("cpu")


def test_argument_validation_and_out_of_range_frames():
    _validation_case("cpu")


@pytest.mark.gpu
def test_edge_cases_on_the_device():
    _nan_action_case("cuda:0")
    _ragged_sizes_case("cuda:0")
    _validation_case("cuda:0")


def test_non_default_env_parameters_match_the_oracle():
    """Constructor parameters away from configs/env_config.yaml's values: 3 substeps, 3 reference frames, sub-clip 4, a
    tight healthy band (some envs fall out of it), a strict termination threshold with a body-error multiplier.  float64
    build of the kernels vs the float64 oracle, resynchronised every control step (tests/test_hostsim_parity.py: why)."""
    import parity as P

    B = 12
    env = H.hostsim_env(B, "double", n_frames=3, ref_traj_length=3, sub_clip_length=4, healthy_z_range=(0.068, 0.10),
                        termination_threshold=0.3, body_error_multiplier=2.5)
    assert env.traj_size == 3 * (env.traj_size // 3) and env._n_frames == 3
    o = H.make_oracle(env, "f64")
    rng = np.random.default_rng(11)
    sf = rng.integers(0, 200, B).astype(np.int32)
    noise = 5e-3 * rng.standard_normal((B, 74))
    st = env.reset(start_frame=torch.from_numpy(sf), noise=torch.from_numpy(noise))
    ost = o.env_reset(sf, noise)
    assert H.scaled_err(st.obs.numpy(), ost["obs"]) < 1e-12 and H.scaled_err(st.info["traj"].numpy(), ost["traj"]) < 1e-11
    assert H.scaled_err(st.info["termination_error"].numpy(), ost["termination_error"]) < 1e-7  # (0.3f != 0.3)
    seen_done, seen_alive = False, False
    for step in range(5):
        ost = P.oracle_state_from(env, o, st)
        act = np.clip(0.5 * rng.standard_normal((B, 30)), -1, 1)
        st = env.step(st, torch.from_numpy(act))
        o.env_step(ost, act)
        e = P.state_errors(st, ost)
        assert max(v.max() for v in e.values()) < 1e-8, {k: v.max() for k, v in e.items()}
        m = np.stack([st.metrics[k].numpy() for k in st.metrics], 1)
        # (thresholds and the z band are float32 in the C-ABI)
        assert np.abs(m - ost["metrics"]).max() < 1e-7 and np.abs(st.reward.numpy() - ost["reward"]).max() < 1e-7
        assert np.array_equal(st.done.numpy(), ost["done"]), (step, st.done.numpy(), ost["done"])
        assert np.array_equal(st.info["sub_clip_frame"].numpy(), ost["sub_clip_frame"])
        assert H.scaled_err(st.info["traj"].numpy(), ost["traj"]) < 1e-9
        seen_done |= bool((st.done.numpy() == 1).any())
        seen_alive |= bool((st.done.numpy() == 0).any())
        if step >= 3:
            assert (st.done.numpy() == 1).all()  # sub_clip_length 4 reached (rodent.py:207-215)
    assert seen_done and seen_alive  # both outcomes of the termination logic were exercised


def test_training_step_on_a_multi_clip_env():
    """BASELINE configs[3]'s code path at toy size: the trainer on an env whose clip container holds several clips, clip
    ids drawn per env at reset and carried through auto-resets (reference stub envs/rodent.py:473-475)."""
    import functools

    from vnl_brax_imitation_amd.ppo_imitation import ppo_networks
    from vnl_brax_imitation_amd.ppo_imitation import train as ppo
    from vnl_brax_imitation_amd.preprocessing import mjx_preprocess as pp

    multi = pp.synthesize_clips(H.model(), H.golden_qpos(), 3, seed=0)
    env = H.hostsim_env(8, reference_clip=multi)
    assert env._num_clips == 3
    st = env.reset(7)
    ids = st.info["clip_id"].clone()
    assert len(set(ids.tolist())) > 1  # several clips in one batch
    nf = functools.partial(ppo_networks.make_intention_ppo_networks, intention_latent_size=16, encoder_layer_sizes=(32, 24),
                           decoder_layer_sizes=(32, 24), value_hidden_layer_sizes=(32,))
    log = []
    ppo.train(environment=env, num_timesteps=2 * 8 * 5, episode_length=150, num_envs=8, learning_rate=1e-3, entropy_cost=1e-3,
              discounting=0.99, unroll_length=5, batch_size=2, num_minibatches=4, num_updates_per_batch=1, num_evals=1,
              normalize_observations=True, network_factory=nf, num_eval_envs=0, eval_env=None, seed=1,
              progress_fn=lambda s, m: log.append(m))
    assert log and np.isfinite(log[-1]["training/total_loss"])
