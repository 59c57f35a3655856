"""PPO-side hot path (networks forward, distribution, GAE, loss, normaliser, trainer) against the
independent NumPy float64 restatement in oracle/ppo_numpy.py."""
import functools

import numpy as np
import pytest
import torch

import helpers as H
from oracle import ppo_numpy as O
from vnl_brax_imitation_amd.ppo_imitation import acting, intention_losses, ppo_networks, running_statistics
from vnl_brax_imitation_amd.ppo_imitation import train as ppo

ENC, DEC = (256, 128), (128, 256)


@pytest.fixture(scope="module")
def nets():
    n = ppo_networks.make_intention_ppo_networks(795, 232, 30, preprocess_observations_fn=running_statistics.normalize,
                                                 intention_latent_size=64, encoder_layer_sizes=ENC,
                                                 decoder_layer_sizes=DEC, value_hidden_layer_sizes=(64, 32))
    g = torch.Generator().manual_seed(0)
    return n, n.policy_network.init(g), n.value_network.init(g)


def _np_params(layout, flat):
    return {k: v.double().numpy() for k, v in layout.views(flat).items()}


def test_parameter_counts_match_reference_architecture():
    n = ppo_networks.make_intention_ppo_networks(795, 232, 30, intention_latent_size=64, encoder_layer_sizes=ENC,
                                                 decoder_layer_sizes=DEC)
    assert n.policy_network.layout.size == 341_180 and n.value_network.layout.size == 1_289_217
    names = list(n.policy_network.layout.entries)
    assert names[0] == "encoder/hidden_0/kernel" and "encoder/fc2_logvar/bias" in names
    assert "decoder/LayerNorm_1/scale" in names and "decoder/LayerNorm_2/scale" not in names


def test_policy_and_value_forward(nets):
    n, pf, vf = nets
    rng = np.random.default_rng(0)
    traj, obs = rng.standard_normal((7, 795)), rng.standard_normal((7, 232))
    eps = rng.standard_normal((7, 64))
    st = running_statistics.init_state(232)
    st = running_statistics.update(st, torch.tensor(rng.standard_normal((50, 232)) * 3 + 1, dtype=torch.float32))
    lg, mu, lv = n.policy_network.apply(st, pf, torch.tensor(traj, dtype=torch.float32),
                                        torch.tensor(obs, dtype=torch.float32), torch.tensor(eps, dtype=torch.float32))
    P = _np_params(n.policy_network.layout, pf)
    obs_n = (obs - st.mean.double().numpy()) / st.std.double().numpy()
    rl, rm, rv = O.policy_forward(P, ENC, list(DEC) + [60], traj, obs_n, eps)
    assert np.abs(lg.numpy() - rl).max() < 2e-5 and np.abs(mu.numpy() - rm).max() < 2e-5
    assert np.abs(lv.numpy() - rv).max() < 2e-5
    v = n.value_network.apply(st, vf, torch.tensor(obs, dtype=torch.float32))
    rv_ = O.value_forward(_np_params(n.value_network.layout, vf), 3, obs_n)
    assert v.shape == (7,) and np.abs(v.numpy() - rv_).max() < 1e-5


def test_distribution(nets):
    d = nets[0].parametric_action_distribution
    rng = np.random.default_rng(1)
    logits, raw, eps = rng.standard_normal((5, 60)), rng.standard_normal((5, 30)), rng.standard_normal((5, 30))
    t = lambda a: torch.tensor(a, dtype=torch.float64)  # noqa: E731
    assert np.abs(d.log_prob(t(logits), t(raw)).numpy() - O.tanh_normal_log_prob(logits, raw)).max() < 1e-10
    assert np.abs(d.entropy(t(logits), t(eps)).numpy() - O.tanh_normal_entropy(logits, eps)).max() < 1e-10
    assert torch.allclose(d.mode(t(logits)), torch.tanh(t(logits)[:, :30]))


def test_running_statistics_matches_restatement():
    rng = np.random.default_rng(2)
    st = running_statistics.init_state(6)
    c, m, s = 0.0, np.zeros(6), np.zeros(6)
    for _ in range(3):
        b = rng.standard_normal((4, 5, 6)) * 2 + 0.5
        st = running_statistics.update(st, torch.tensor(b, dtype=torch.float32))
        c, m, s, sd = O.running_update(c, m, s, b)
    assert float(st.count) == c and np.abs(st.mean.numpy() - m).max() < 1e-5 and np.abs(st.std.numpy() - sd).max() < 1e-5
    assert running_statistics.init_state(3).std.tolist() == [1.0, 1.0, 1.0]


def _fake_batch(B, T, rng):
    return dict(observation=rng.standard_normal((B, T, 232)), next_observation=rng.standard_normal((B, T, 232)),
                reward=rng.random((B, T)) * 0.03, discount=(rng.random((B, T)) > 0.1).astype(float),
                truncation=(rng.random((B, T)) > 0.9).astype(float), traj=rng.standard_normal((B, T, 795)) * 0.1,
                raw_action=rng.standard_normal((B, T, 30)) * 0.5, log_prob=rng.standard_normal((B, T)) - 20)


def test_gae_and_ppo_loss_match_restatement(nets):
    n, pf, vf = nets
    rng = np.random.default_rng(3)
    B, T = 6, 5
    d = _fake_batch(B, T, rng)
    f = lambda a: torch.tensor(a, dtype=torch.float32)  # noqa: E731
    data = acting.Transition(f(d["observation"]), torch.zeros(B, T, 30), f(d["reward"]), f(d["discount"]),
                             f(d["next_observation"]),
                             {"policy_extras": {"raw_action": f(d["raw_action"]), "log_prob": f(d["log_prob"])},
                              "state_extras": {"truncation": f(d["truncation"]), "traj": f(d["traj"])}})
    eps_l, eps_e = rng.standard_normal((T, B, 64)), rng.standard_normal((T, B, 30))
    st = running_statistics.init_state(232)
    kw = dict(entropy_cost=1e-3, discounting=0.99, reward_scaling=1.0, gae_lambda=0.95, clipping_epsilon=0.2,
              normalize_advantage=True, kl_weight=1e-4)
    params = intention_losses.PPONetworkParams(policy=pf.clone().requires_grad_(True), value=vf.clone().requires_grad_(True))
    loss, m = intention_losses.compute_ppo_intention_loss(params, st, data, None, ppo_network=n,
                                                          noise={"latent": f(eps_l), "entropy": f(eps_e)}, **kw)
    ref = O.ppo_intention_loss(_np_params(n.policy_network.layout, pf), _np_params(n.value_network.layout, vf), ENC,
                               list(DEC) + [60], 3, np.zeros(232), np.ones(232), d, eps_l, eps_e, **kw)
    for k in ("total_loss", "policy_loss", "v_loss", "entropy_loss", "kl_loss_intention"):
        assert abs(float(m[k]) - ref[k]) < 2e-4 * max(1.0, abs(ref[k])), (k, float(m[k]), ref[k])
    loss.backward()
    assert torch.isfinite(params.policy.grad).all() and params.value.grad.abs().sum() > 0
    # gradient check of the torch loss against central differences of the NumPy restatement (one weight)
    name, idx = "decoder/hidden_2/bias", 3
    off = n.policy_network.layout.entries[name][0] + idx
    h = 1e-4
    vals = []
    for s in (+h, -h):
        P = _np_params(n.policy_network.layout, pf)
        P[name] = P[name].copy()
        P[name][idx] += s
        vals.append(O.ppo_intention_loss(P, _np_params(n.value_network.layout, vf), ENC, list(DEC) + [60], 3,
                                         np.zeros(232), np.ones(232), d, eps_l, eps_e, **kw)["total_loss"])
    fd = (vals[0] - vals[1]) / (2 * h)
    assert abs(float(params.policy.grad[off]) - fd) < 5e-3 * max(1.0, abs(fd)), (float(params.policy.grad[off]), fd)


def test_gae_simple_case():
    T, B = 4, 2
    z = torch.zeros(T, B)
    vs, adv = intention_losses.compute_gae(z, z, torch.ones(T, B), z, torch.zeros(B), lambda_=1.0, discount=1.0)
    assert vs[:, 0].tolist() == [4.0, 3.0, 2.0, 1.0] and adv[:, 0].tolist() == [4.0, 3.0, 2.0, 1.0]


def test_generate_unroll_shapes_and_reference_quirk():
    env = H.hostsim_env(4)
    from vnl_brax_imitation_amd.envs.wrappers import wrap

    w = wrap(env, episode_length=150)
    st = w.reset(0)
    n = ppo_networks.make_intention_ppo_networks(795, 232, 30, intention_latent_size=64, encoder_layer_sizes=ENC,
                                                 decoder_layer_sizes=DEC, value_hidden_layer_sizes=(32,))
    g = torch.Generator().manual_seed(0)
    policy = ppo_networks.make_inference_fn(n)((None, n.policy_network.init(g)))
    obs0 = st.obs.clone()
    st, data = acting.generate_unroll(w, st, policy, g, 3, extra_fields=("truncation", "traj"))
    assert data.observation.shape == (3, 4, 232) and data.extras["state_extras"]["traj"].shape == (3, 4, 795)
    assert data.extras["policy_extras"]["logits"].shape == (3, 4, 60)
    assert torch.equal(data.observation[0], obs0) and torch.equal(data.next_observation[0], data.observation[1])
    assert torch.equal(data.extras["state_extras"]["traj"][2], st.info["traj"])  # AFTER-step traj (acting.py:49)
    assert (data.action.abs() <= 1).all()


def test_train_runs_and_returns_reference_triple():
    env = H.hostsim_env(8)
    nf = functools.partial(ppo_networks.make_intention_ppo_networks, intention_latent_size=16,
                           encoder_layer_sizes=(32,), decoder_layer_sizes=(32,), value_hidden_layer_sizes=(32,))
    log = []
    make_policy, params, metrics = ppo.train(
        environment=env, num_timesteps=8 * 5 * 4, episode_length=150, num_envs=8, learning_rate=6e-4,
        entropy_cost=1e-3, discounting=0.99, unroll_length=5, batch_size=2, num_minibatches=4,
        num_updates_per_batch=2, num_evals=2, normalize_observations=True, network_factory=nf,
        progress_fn=lambda s, m: log.append((s, m)), kl_weight=1e-4, clipping_epsilon=0.2, eval_env=H.hostsim_env(4))
    assert [s for s, _ in log] == [0, 160]
    for k in ("training/sps", "training/walltime", "training/total_loss", "training/policy_loss", "training/v_loss",
              "training/entropy_loss", "training/kl_loss_intention", "eval/episode_reward", "eval/sps",
              "eval/avg_episode_length"):
        assert k in metrics, k
    norm, flat = params
    assert float(norm.count) == 160 and flat.dim() == 1  # 4 training steps x 8 envs x 5 steps
    act, extras = make_policy(params, deterministic=True)(torch.zeros(2, 795), torch.zeros(2, 232), None)
    assert act.shape == (2, 30) and extras == {}


def _loss_both_ways(n, pf, vf, device, lib, B=16, T=7, seed=5):
    rng = np.random.default_rng(seed)
    d = _fake_batch(B, T, rng)
    d["log_prob"] = d["log_prob"] + 20 - 25  # keep rho = exp(target - behaviour) near the clipping range ...
    f = lambda a: torch.tensor(a, dtype=torch.float32, device=device)  # noqa: E731
    data = acting.Transition(f(d["observation"]), torch.zeros(B, T, 30, device=device), f(d["reward"]), f(d["discount"]),
                             f(d["next_observation"]),
                             {"policy_extras": {"raw_action": f(d["raw_action"]), "log_prob": f(d["log_prob"])},
                              "state_extras": {"truncation": f(d["truncation"]), "traj": f(d["traj"])}})
    noise = {"latent": f(rng.standard_normal((T, B, 64))), "entropy": f(rng.standard_normal((T, B, 30)))}
    st = running_statistics.init_state(232, device=device)
    kw = dict(entropy_cost=1e-2, discounting=0.97, reward_scaling=2.0, gae_lambda=0.9, clipping_epsilon=0.3,
              normalize_advantage=True, kw_dummy=None)
    kw.pop("kw_dummy")
    # ... by making the behaviour log-prob the current one plus noise (as in a real PPO epoch)
    with torch.no_grad():
        lg, _, _ = n.policy_network.apply(st, pf.to(device), data.extras["state_extras"]["traj"].transpose(0, 1),
                                          data.observation.transpose(0, 1), noise["latent"])
        cur = n.parametric_action_distribution.log_prob(lg, data.extras["policy_extras"]["raw_action"].transpose(0, 1))
        data.extras["policy_extras"]["log_prob"] = (cur + 0.4 * f(rng.standard_normal((T, B)))).transpose(0, 1).contiguous()
    out = []
    for head in ("torch", lib):
        params = intention_losses.PPONetworkParams(policy=pf.clone().to(device).requires_grad_(True),
                                                   value=vf.clone().to(device).requires_grad_(True))
        loss, m = intention_losses.compute_ppo_intention_loss(params, st, data, None, ppo_network=n, noise=noise,
                                                              kl_weight=3e-3, head=head, **kw)
        loss.backward()
        out.append((loss.detach(), m, params.policy.grad.clone(), params.value.grad.clone()))
    return out


def _check_head(out):
    (l0, m0, gp0, gv0), (l1, m1, gp1, gv1) = out
    for k in m0:
        a, b = float(m0[k]), float(m1[k])
        assert abs(a - b) < 2e-5 * max(1.0, abs(a)), (k, a, b)
    assert abs(float(l0) - float(l1)) < 2e-5 * max(1.0, abs(float(l0)))
    for g0, g1 in ((gp0, gp1), (gv0, gv1)):
        scale = float(g0.abs().max())
        assert scale > 0 and float((g0 - g1).abs().max()) < 2e-4 * scale, (scale, float((g0 - g1).abs().max()))
    assert 1e-3 < float((m0["policy_loss"]).abs()) < 10  # a non-degenerate surrogate (rho spread around the clip range)


def test_fused_loss_head_matches_autograd_hostsim(nets):
    """vnl_ppo_head (loss head + gradients in one launch) against the op-by-op torch loss and ITS autograd."""
    n, pf, vf = nets
    _check_head(_loss_both_ways(n, pf, vf, torch.device("cpu"), H.hostsim_library("float")))


@pytest.mark.gpu
def test_fused_loss_head_matches_autograd_gpu(nets):
    from vnl_brax_imitation_amd import _lib

    n, pf, vf = nets
    _check_head(_loss_both_ways(n, pf, vf, torch.device("cuda:0"), _lib.load_library(), B=128, T=20))


def test_adam_kernel_matches_the_torch_update():
    """vnl_adam_step (one launch) against FlatAdam's op-by-op update (optax.adam semantics), host simulation."""
    g = torch.Generator().manual_seed(0)
    p0 = torch.randn(1000, generator=g)
    outs = []
    for lib in (None, H.hostsim_library("float")):
        opt = ppo.FlatAdam(3e-3, lib=lib)
        p, st = p0.clone(), None
        st = opt.init(p)
        gg = torch.Generator().manual_seed(1)
        for _ in range(5):
            opt.update(torch.randn(1000, generator=gg) * 0.1, st, p)
        outs.append((p, st))
    (pa, sa), (pb, sb) = outs
    assert int(sa["count"]) == int(sb["count"]) == 5
    assert torch.allclose(pa, pb, rtol=0, atol=2e-7) and torch.allclose(sa["nu"], sb["nu"], rtol=1e-6, atol=0)
    assert not torch.equal(pa, p0)


def test_train_with_the_references_batch_proportions():
    """configs/train_config.yaml:4-11 has batch_size x num_minibatches = 8 x num_envs (32 x 32 / 128): brax's training step then
    collects EIGHT unrolls before its SGD epochs (reference ppo_imitation/train.py:293-330, `batch_size * num_minibatches //
    num_envs` unrolls).  Same here, at a ratio of 2: every training step advances 2 x num_envs x unroll_length env steps, the
    normaliser has seen them all, and the minibatches hold batch_size trajectories."""
    env = H.hostsim_env(8)
    nf = functools.partial(ppo_networks.make_intention_ppo_networks, intention_latent_size=16,
                           encoder_layer_sizes=(32,), decoder_layer_sizes=(32,), value_hidden_layer_sizes=(32,))
    log = []
    _, params, metrics = ppo.train(
        environment=env, num_timesteps=2 * 8 * 5 * 2, episode_length=150, num_envs=8, learning_rate=6e-4,
        entropy_cost=1e-3, discounting=0.99, unroll_length=5, batch_size=4, num_minibatches=4,
        num_updates_per_batch=2, num_evals=2, normalize_observations=True, network_factory=nf,
        progress_fn=lambda s, m: log.append((s, m)), kl_weight=1e-4, clipping_epsilon=0.2, num_eval_envs=0, eval_env=None)
    assert [s for s, _ in log][-1] == 160  # 2 training steps x (2 unrolls x 8 envs x 5 steps)
    norm, flat = params
    assert float(norm.count) == 160 and torch.isfinite(flat).all()
    assert int(ppo.train.last_training_state.optimizer_state["count"]) == 2 * 2 * 4  # training steps x updates x minibatches
    assert all(np.isfinite(float(metrics[k])) for k in ("training/total_loss", "training/v_loss", "training/policy_loss"))
