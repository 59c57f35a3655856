"""vnl_rollout_post (Episode + AutoReset wrappers + Transition logging in one launch) against the generic
wrapper / logging path, on the host simulation of the same kernels."""
import pytest
import torch

import helpers as H
from vnl_brax_imitation_amd.envs.wrappers import AutoResetWrapper, EpisodeWrapper
from vnl_brax_imitation_amd.ppo_imitation import acting, ppo_networks, running_statistics


def _setup(B, reset_info, episode_length, make_env, device):
    base = make_env(B)
    env = AutoResetWrapper(EpisodeWrapper(base, episode_length=episode_length, action_repeat=1),
                           reset_info_on_autoreset=reset_info)
    nets = ppo_networks.make_intention_ppo_networks(base.traj_size, base.observation_size, base.action_size,
                                                    preprocess_observations_fn=running_statistics.normalize,
                                                    intention_latent_size=16, encoder_layer_sizes=(32,),
                                                    decoder_layer_sizes=(32,))
    flat = nets.policy_network.init(torch.Generator().manual_seed(0)).to(device)
    norm = running_statistics.init_state(base.observation_size, device=device)
    return env, ppo_networks.make_inference_fn(nets)((norm, flat))


def _compare(make_env, device, reset_info, B=6, T=7):
    out = []
    for fused in (False, True):
        env, policy = _setup(B, reset_info, 3, make_env, device)
        torch.manual_seed(123)  # rand_log_prob's uniform draw comes from the global generator on HIP devices
        state = env.reset(torch.Generator().manual_seed(5))
        key = torch.Generator(device=device).manual_seed(11)
        state, data = acting.generate_unroll(env, state, policy, key, T, extra_fields=("truncation", "traj"), fused=fused)
        # a second unroll continues from the carried state (prev_done hand-over between calls)
        state, data2 = acting.generate_unroll(env, state, policy, key, 3, extra_fields=("truncation", "traj"), fused=fused)
        out.append((state, data, data2))
    (s0, d0, e0), (s1, d1, e1) = out
    for a, b in zip(acting._leaves(d0) + acting._leaves(e0), acting._leaves(d1) + acting._leaves(e1)):
        assert a.shape == b.shape and torch.equal(a, b)
    assert torch.equal(s0.obs, s1.obs) and torch.equal(s0.done, s1.done) and torch.equal(s0.reward, s1.reward)
    for k in ("steps", "truncation", "traj", "cur_frame", "sub_clip_frame"):
        assert torch.equal(s0.info[k], s1.info[k]), k
    for n in s0.pipeline_state._FIELDS:
        assert torch.equal(s0.pipeline_state.raw(n), s1.pipeline_state.raw(n)), n
    assert float(d0.extras["state_extras"]["truncation"].sum()) > 0  # episodes of 3 steps did end
    assert float((1 - d0.discount).sum()) > 0


@pytest.mark.parametrize("reset_info", [False, True])
def test_fused_post_matches_generic_wrappers_hostsim(reset_info):
    _compare(lambda B: H.hostsim_env(B, "float"), torch.device("cpu"), reset_info)


@pytest.mark.gpu
@pytest.mark.parametrize("reset_info", [False, True])
def test_fused_post_matches_generic_wrappers_gpu(reset_info):
    from vnl_brax_imitation_amd.envs.rodent import RodentTracking

    dev = torch.device("cuda:0")
    _compare(lambda B: RodentTracking(H.reference_clip(), num_envs=B, device=dev, **H.env_kwargs()), dev, reset_info,
             B=130, T=8)


@pytest.mark.gpu
@pytest.mark.parametrize("reset_info", [False, True])
def test_graphed_unroll_equals_the_eager_unroll_bit_for_bit(reset_info):
    """acting.GraphedUnroll: the unroll captured into ONE hipGraph and replayed (policy launches, env step kernels,
    vnl_rollout_post, the noise draws) against the eager fused loop: every Transition leaf and the carried env state of
    three consecutive unrolls, bit for bit -- same kernels, same arguments, same Philox stream."""
    from vnl_brax_imitation_amd.envs.rodent import RodentTracking

    dev = torch.device("cuda:0")
    make_env = lambda B: RodentTracking(H.reference_clip(), num_envs=B, device=dev, **H.env_kwargs())  # noqa: E731
    B, T, extra = 130, 6, ("truncation", "traj")
    out = []
    for graphed in (False, True):
        env, policy = _setup(B, reset_info, 4, make_env, dev)
        torch.manual_seed(123)
        state = env.reset(torch.Generator().manual_seed(5))
        key = torch.Generator(device=dev).manual_seed(11)
        g = acting.GraphedUnroll(env, state, policy, key, T, extra_fields=extra) if graphed else None
        datas = []
        for _ in range(3):
            if graphed:
                state, data = g()
            else:
                state, data = acting.generate_unroll(env, state, policy, key, T, extra_fields=extra, fused=True)
            datas.append([x.clone() for x in acting._leaves(data)])  # (the graph's buffers are overwritten by the next replay)
        out.append((state, datas, key.get_state().clone()))
    (s0, d0, k0), (s1, d1, k1) = out
    for a_, b_ in zip(d0, d1):
        for a, b in zip(a_, b_):
            assert a.shape == b.shape and torch.equal(a, b)
    assert torch.equal(k0, k1)  # the generator's stream advanced exactly as in the eager calls
    assert torch.equal(s0.obs, s1.obs) and torch.equal(s0.done, s1.done) and torch.equal(s0.reward, s1.reward)
    for k in ("steps", "truncation", "traj", "cur_frame", "sub_clip_frame"):
        assert torch.equal(s0.info[k], s1.info[k]), k
    for n in s0.pipeline_state._FIELDS:
        assert torch.equal(s0.pipeline_state.raw(n), s1.pipeline_state.raw(n)), n
