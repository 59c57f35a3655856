"""PPO training step on the GPU: the hipGraph-captured minibatch step must reproduce the eager one."""
import functools

import pytest
import torch

import helpers as H
from vnl_brax_imitation_amd.envs.rodent import RodentTracking
from vnl_brax_imitation_amd.ppo_imitation import ppo_networks
from vnl_brax_imitation_amd.ppo_imitation import train as ppo

pytestmark = pytest.mark.gpu


def _run(capture: bool):
    dev = torch.device("cuda:0")
    env = RodentTracking(H.reference_clip(), num_envs=64, device=dev, **H.env_kwargs())
    nf = functools.partial(ppo_networks.make_intention_ppo_networks, intention_latent_size=60,
                           encoder_layer_sizes=(128, 128), decoder_layer_sizes=(128, 128))
    log = []
    _, (norm, flat), _ = ppo.train(
        environment=env, num_timesteps=3 * 64 * 5, episode_length=150, num_envs=64, learning_rate=1e-3,
        entropy_cost=1e-2, discounting=0.95, unroll_length=5, batch_size=16, num_minibatches=4,
        num_updates_per_batch=2, num_evals=1, normalize_observations=True, network_factory=nf, num_eval_envs=0,
        eval_env=None, seed=3, capture_graph=capture, progress_fn=lambda s, m: log.append(m))
    return norm, flat, log[-1], ppo.train.last_training_state


def test_graph_capture_matches_eager():
    n0, f0, m0, ts0 = _run(False)
    n1, f1, m1, ts1 = _run(True)
    assert int(ts0.optimizer_state["count"]) == int(ts1.optimizer_state["count"]) == 3 * 2 * 4
    assert torch.equal(n0.mean, n1.mean) and torch.equal(n0.std, n1.std)
    # same kernels on the same inputs; the only difference is how they are launched
    assert torch.allclose(f0, f1, rtol=0, atol=1e-6), float((f0 - f1).abs().max())
    assert torch.allclose(ts0.params, ts1.params, rtol=0, atol=1e-6)
    for k in ("training/total_loss", "training/v_loss", "training/policy_loss", "training/kl_loss_intention"):
        assert abs(m0[k] - m1[k]) <= 1e-5 * max(1.0, abs(m0[k])), (k, m0[k], m1[k])
    assert torch.isfinite(f1).all()
