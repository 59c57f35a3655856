"""Checkpoint I/O: inference pair with the reference's Flax tensor names, full state for resume."""
import functools

import numpy as np
import pytest
import torch

import helpers as H
from vnl_brax_imitation_amd.ppo_imitation import checkpoint, ppo_networks
from vnl_brax_imitation_amd.ppo_imitation import train as ppo


def _train(tmp=None, **kw):
    env = H.hostsim_env(8)
    nf = functools.partial(ppo_networks.make_intention_ppo_networks, intention_latent_size=16,
                           encoder_layer_sizes=(32, 24), decoder_layer_sizes=(32, 24), value_hidden_layer_sizes=(32,))
    return ppo.train(environment=env, num_timesteps=8 * 5, episode_length=150, num_envs=8, learning_rate=1e-3,
                     entropy_cost=1e-3, discounting=0.99, unroll_length=5, batch_size=2, num_minibatches=4,
                     num_updates_per_batch=1, num_evals=1, normalize_observations=True, network_factory=nf,
                     num_eval_envs=0, eval_env=None, **kw)


def test_round_trip_with_flax_names(tmp_path):
    _, params, _ = _train()
    net, ts = ppo.train.last_ppo_network, ppo.train.last_training_state
    n_pol = net.policy_network.layout.size
    path = checkpoint.save_params(str(tmp_path / "ck"), params, net, value_params=ts.params.detach()[n_pol:],
                                  optimizer_state=ts.optimizer_state, env_steps=ts.env_steps)
    z = np.load(path)
    for k in ("policy/params/encoder/hidden_0/kernel", "policy/params/encoder/LayerNorm_1/scale",
              "policy/params/encoder/fc2_mean/bias", "policy/params/encoder/fc2_logvar/kernel",
              "policy/params/decoder/hidden_2/kernel", "value/params/hidden_0/kernel", "normalizer/mean",
              "optimizer/mu", "meta/env_steps"):
        assert k in z.files, k
    assert z["policy/params/encoder/hidden_0/kernel"].shape == (795, 32)  # Dense kernels stay (in, out) as in Flax
    ck = checkpoint.load_params(path, net)
    assert torch.equal(ck["params"][1], params[1]) and torch.equal(ck["params"][0].mean, params[0].mean)
    assert torch.equal(ck["value"], ts.params.detach()[n_pol:]) and ck["env_steps"] == ts.env_steps
    assert torch.equal(ck["optimizer"]["nu"], ts.optimizer_state["nu"])
    # nested-dict (Flax pytree) form
    tree = checkpoint.to_flax_tree(net.policy_network.layout, params[1])
    assert set(tree["params"]) == {"encoder", "decoder"}
    assert torch.equal(checkpoint.from_flax_tree(net.policy_network.layout, tree), params[1])
    bad = checkpoint.to_flax_tree(net.policy_network.layout, params[1])
    del bad["params"]["encoder"]["fc2_mean"]
    with pytest.raises(KeyError):
        checkpoint.from_flax_tree(net.policy_network.layout, bad)


def test_training_resumes_from_a_checkpoint(tmp_path):
    _, params, _ = _train()
    net, ts = ppo.train.last_ppo_network, ppo.train.last_training_state
    n_pol = net.policy_network.layout.size
    path = checkpoint.save_params(str(tmp_path / "ck"), params, net, value_params=ts.params.detach()[n_pol:],
                                  optimizer_state=ts.optimizer_state, env_steps=ts.env_steps)
    count0, steps0 = int(ts.optimizer_state["count"]), ts.env_steps
    _, params2, _ = _train(restore_from=path)
    ts2 = ppo.train.last_training_state
    assert int(ts2.optimizer_state["count"]) == 2 * count0 and ts2.env_steps == 2 * steps0
    assert float(params2[0].count) == 2 * float(params[0].count)  # the normaliser kept accumulating
    assert not torch.equal(params2[1], params[1])
