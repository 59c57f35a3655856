"""The hand-written PPO minibatch step (forward + backward, csrc/vnl_ppo.hip through the C-ABI) against
 (1) torch autograd through the op-by-op loss (the reference's formulation, itself checked against oracle/ppo_numpy.py on
     the CPU tier), float64, and
 (2) the float64 NumPy restatement of the forward pass and the loss terms,
at the reference's network sizes (configs/train_config.yaml) and at a small odd-sized network."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _make(traj, obs, act, latent, enc, dec, val, T, B, seed=0):
    from vnl_brax_imitation_amd.ppo_imitation import acting, ppo_networks, running_statistics

    nets = ppo_networks.make_intention_ppo_networks(traj, obs, act, preprocess_observations_fn=running_statistics.normalize,
                                                    intention_latent_size=latent, encoder_layer_sizes=enc,
                                                    decoder_layer_sizes=dec, value_hidden_layer_sizes=val)
    g = torch.Generator().manual_seed(seed)
    flat = torch.cat([nets.policy_network.init(g), nets.value_network.init(g)])
    # LayerNorm scale / bias and the biases away from their trivial initial values, so that their gradients matter
    flat = flat + 0.05 * torch.randn(flat.shape, generator=g)
    r = lambda *s: torch.randn(*s, generator=g)  # noqa: E731
    data = acting.Transition(
        observation=2.0 * r(T, B, obs) + 0.5, action=torch.tanh(r(T, B, act)), reward=0.1 * r(T, B).abs(),
        discount=(torch.rand(T, B, generator=g) > 0.1).float(), next_observation=2.0 * r(T, B, obs) + 0.5,
        extras={"policy_extras": {"raw_action": 0.5 * r(T, B, act), "log_prob": -20.0 + r(T, B)},
                "state_extras": {"truncation": (torch.rand(T, B, generator=g) > 0.9).float(), "traj": 0.3 * r(T, B, traj)}})
    norm = running_statistics.init_state(obs)
    norm = running_statistics.update(norm, data.observation)
    noise = {"latent": r(T, B, latent), "entropy": r(T, B, act)}
    return nets, flat, data, norm, noise


HP = dict(entropy_cost=1e-3, discounting=0.99, reward_scaling=1.0, gae_lambda=0.95, clipping_epsilon=0.2,
          normalize_advantage=True, kl_weight=1e-4)


@pytest.mark.parametrize("cfg", [
    dict(traj=795, obs=232, act=30, latent=64, enc=(256, 128), dec=(128, 256), val=(1024, 1024), T=20, B=128),
    dict(traj=45, obs=19, act=5, latent=6, enc=(40, 24), dec=(24, 40), val=(72, 56), T=5, B=9),
    # other loss coefficients: unnormalised advantages, scaled rewards, strong KL / entropy terms, wide clip, three value layers
    dict(traj=45, obs=19, act=5, latent=6, enc=(40, 24), dec=(24, 40), val=(72, 56, 40), T=7, B=11,
         hp=dict(entropy_cost=0.05, discounting=0.9, reward_scaling=2.5, gae_lambda=0.8, clipping_epsilon=0.35,
                 normalize_advantage=False, kl_weight=0.3)),
    # a large minibatch with wide value layers: above 8,192 rows the wide layer's weight gradient takes its input transposed
    # ([X | 1]' materialised by transpose_ones_kernel, then a plain product) and the forward goes layer by layer
    dict(traj=45, obs=19, act=5, latent=6, enc=(40, 24), dec=(24, 40), val=(512, 512), T=20, B=420),
], ids=["reference-sizes", "small-odd", "other-coefficients", "large-minibatch-wide-value"])
def test_hip_update_matches_autograd_and_numpy(cfg):
    from vnl_brax_imitation_amd.ppo_imitation import hip_update, intention_losses

    cfg = dict(cfg)
    HP = cfg.pop("hp", globals()["HP"])
    T, B = cfg["T"], cfg["B"]
    nets, flat, data, norm, noise = _make(**cfg)
    dev = torch.device("cuda:0")
    n_pol = nets.policy_network.layout.size
    # --- reference: float64 autograd through the op-by-op loss ------------------------------------------------
    p64 = flat.double().requires_grad_(True)
    d64 = data.map(lambda x: x.double())
    from vnl_brax_imitation_amd.ppo_imitation import running_statistics
    n64 = running_statistics.RunningStatisticsState(norm.count.double(), norm.mean.double(), norm.summed_variance.double(),
                                                    norm.std.double())
    params = intention_losses.PPONetworkParams(policy=p64[:n_pol], value=p64[n_pol:])
    loss, m_ref = intention_losses.compute_ppo_intention_loss(
        params, n64, d64, None, ppo_network=nets, noise={k: v.double() for k, v in noise.items()}, head="torch",
        time_major=True, **HP)
    loss.backward()
    g_ref = p64.grad.numpy()
    # --- product ----------------------------------------------------------------------------------------------
    upd = hip_update.HipPPOUpdate(nets, T, B, dev, **HP)
    to = lambda t: t.to(dev)  # noqa: E731
    grads = torch.full((flat.numel(),), float("nan"), device=dev)
    ndev = running_statistics.RunningStatisticsState(to(norm.count), to(norm.mean), to(norm.summed_variance), to(norm.std))
    mt = upd.grad(to(flat).contiguous(), ndev, data.map(to), {k: to(v) for k, v in noise.items()}, grads)
    torch.cuda.synchronize()
    g = grads.cpu().numpy().astype(np.float64)
    assert np.isfinite(g).all()
    mt = mt.cpu().numpy()
    for i, k in enumerate(("total_loss", "policy_loss", "v_loss", "entropy_loss", "kl_loss_intention", "explained_variance")):
        ref = float(m_ref[k])
        assert abs(mt[i] - ref) <= 2e-5 * max(abs(ref), abs(float(m_ref["total_loss"]))), (k, mt[i], ref)
    # metrics[8]: mean of corrcoef([vs ; reward * scaling]) (intention_losses.py:186-188), float64 autograd path as reference
    # (NaN on BOTH sides = "not computed": the 2T x B rows of a large minibatch do not fit the one-workgroup kernel, and the
    # torch path applies the same rule, intention_losses.py: _corr_fits)
    pc_ref = float(m_ref["prediction_corr"])
    assert (np.isnan(mt[8]) and np.isnan(pc_ref)) or abs(mt[8] - pc_ref) < 2e-5, (mt[8], pc_ref)
    # gradients per tensor, relative to that tensor's largest gradient entry
    worst = 0.0
    for lay, off0 in ((nets.policy_network.layout, 0), (nets.value_network.layout, n_pol)):
        for name, (off, shape) in lay.entries.items():
            n = int(np.prod(shape))
            a, b = g[off0 + off: off0 + off + n], g_ref[off0 + off: off0 + off + n]
            scale = max(np.abs(b).max(), 1e-12)
            e = np.abs(a - b).max() / scale
            worst = max(worst, e)
            # (float32 sums over the T x B rows: the bound grows with the square root of the row count beyond the reference's 2,560)
            assert e < 2e-5 * max(1.0, np.sqrt(T * B / 2560.0)), (name, e, scale)
    print(f"\n[ppo update, {cfg['enc']}/{cfg['val']}] worst per-tensor gradient error {worst:.2e}; losses {mt[:5]}")
    # forward intermediates against the float64 NumPy restatement
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from oracle import ppo_numpy as PN

    views_p = {k: v.numpy().astype(np.float64) for k, v in nets.policy_network.layout.views(flat[:n_pol]).items()}
    views_v = {k: v.numpy().astype(np.float64) for k, v in nets.value_network.layout.views(flat[n_pol:]).items()}
    npd = lambda t: t.numpy().astype(np.float64)  # noqa: E731
    sw = lambda x: np.swapaxes(npd(x), 0, 1)  # noqa: E731  (the restatement takes [B, T, ...] like the reference)
    ddict = dict(observation=sw(data.observation), next_observation=sw(data.next_observation),
                 traj=sw(data.extras["state_extras"]["traj"]), raw_action=sw(data.extras["policy_extras"]["raw_action"]),
                 log_prob=sw(data.extras["policy_extras"]["log_prob"]), reward=sw(data.reward), discount=sw(data.discount),
                 truncation=sw(data.extras["state_extras"]["truncation"]))
    enc_l, dec_l = list(cfg["enc"]), list(cfg["dec"]) + [2 * cfg["act"]]
    parts = PN.ppo_intention_loss(views_p, views_v, enc_l, dec_l, len(cfg["val"]) + 1, npd(norm.mean), npd(norm.std), ddict,
                                  npd(noise["latent"]), npd(noise["entropy"]), **HP)
    for i, k in enumerate(("total_loss", "policy_loss", "v_loss", "entropy_loss", "kl_loss_intention")):
        assert abs(mt[i] - parts[k]) <= 2e-5 * max(abs(parts[k]), abs(parts["total_loss"])), ("numpy", k, mt[i], parts[k])
    obs_n = (npd(data.observation) - npd(norm.mean)) / npd(norm.std)
    logits, mean, logvar = PN.policy_forward(views_p, enc_l, dec_l, npd(data.extras["state_extras"]["traj"]), obs_n, npd(noise["latent"]))
    values = PN.value_forward(views_v, len(cfg["val"]) + 1, obs_n)
    for name, ref in (("logits", logits), ("latent_mean", mean), ("latent_logvar", logvar)):
        got = upd.buffer(name).cpu().numpy().astype(np.float64).reshape(ref.shape)
        assert np.abs(got - ref).max() / max(np.abs(ref).max(), 1e-12) < 1e-5, name
    got = upd.buffer("values").cpu().numpy().astype(np.float64)[: T * B].reshape(T, B)
    assert np.abs(got - values).max() / max(np.abs(values).max(), 1e-12) < 1e-5
    for name in ("vs", "advantages"):
        got = upd.buffer(name).cpu().numpy().astype(np.float64).reshape(T, B)
        ref = parts[name] if name == "vs" else None
        if ref is not None:
            assert np.abs(got - ref).max() / max(np.abs(ref).max(), 1e-12) < 1e-5, name


def test_forward_routes_of_the_intention_network_agree():
    """The intention network's forward pass has three routes (csrc/vnl_ppo.hip: layer by layer; ONE fused launch of the acting
    path's kernel in its training form; first Dense as a GEMM + the rest fused -- the default).  Same network, same inputs:
    losses and every gradient tensor must agree to float32 rounding (the fused routes accumulate a Dense in K slices)."""
    from vnl_brax_imitation_amd.ppo_imitation import hip_update, running_statistics

    cfg = dict(traj=795, obs=232, act=30, latent=64, enc=(256, 128), dec=(128, 256), val=(1024, 1024), T=20, B=128)
    nets, flat, data, norm, noise = _make(**cfg)
    dev = torch.device("cuda:0")
    to = lambda t: t.to(dev)  # noqa: E731
    ndev = running_statistics.RunningStatisticsState(to(norm.count), to(norm.mean), to(norm.summed_variance), to(norm.std))
    fl, dd, nn = to(flat).contiguous(), data.map(to), {k: to(v) for k, v in noise.items()}
    out = {}
    for mode in (0, 1, 2):
        upd = hip_update.HipPPOUpdate(nets, cfg["T"], cfg["B"], dev, **HP)
        assert upd.lib.vnl_ppo_update_tune(upd.h, -10 - mode, 0) == 0
        grads = torch.full((flat.numel(),), float("nan"), device=dev)
        mt = upd.grad(fl, ndev, dd, nn, grads)
        torch.cuda.synchronize()
        out[mode] = (grads.cpu().numpy().astype(np.float64), mt.cpu().numpy().astype(np.float64))
    n_pol = nets.policy_network.layout.size
    for mode in (0, 1):
        g, m = out[mode]
        gr, mr = out[2]
        assert np.isfinite(g).all()
        assert np.abs(m[:5] - mr[:5]).max() <= 2e-6 * max(1.0, np.abs(mr[:5]).max()), (mode, m[:5], mr[:5])
        worst = 0.0
        for lay, off0 in ((nets.policy_network.layout, 0), (nets.value_network.layout, n_pol)):
            for name, (off, shape) in lay.entries.items():
                n = int(np.prod(shape))
                a, b = g[off0 + off: off0 + off + n], gr[off0 + off: off0 + off + n]
                e = np.abs(a - b).max() / max(np.abs(b).max(), 1e-12)
                worst = max(worst, e)
                assert e < 1e-5, (mode, name, e)
        print(f"\n[forward route {mode} vs default] worst per-tensor gradient difference {worst:.2e}")
