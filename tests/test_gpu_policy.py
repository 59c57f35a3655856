"""Fused HIP policy kernel (vnl_policy_forward) vs the torch path and the NumPy restatement."""
import numpy as np
import pytest
import torch

import helpers as H
from oracle import ppo_numpy as O
from vnl_brax_imitation_amd.ppo_imitation import ppo_networks, running_statistics

pytestmark = pytest.mark.gpu
ENC, DEC = (256, 128), (128, 256)


@pytest.mark.parametrize("B", [32, 100, 4096])
def test_hip_policy_matches_torch_and_numpy(B):
    dev = torch.device("cuda:0")
    n = ppo_networks.make_intention_ppo_networks(795, 232, 30, preprocess_observations_fn=running_statistics.normalize,
                                                 intention_latent_size=64, encoder_layer_sizes=ENC, decoder_layer_sizes=DEC)
    g = torch.Generator().manual_seed(B)
    flat = n.policy_network.init(g)
    flat += 0.05 * torch.randn(flat.shape, generator=g)  # non-trivial biases / LayerNorm parameters
    st = running_statistics.init_state(232)
    st = running_statistics.update(st, torch.randn((64, 232), generator=g) * 2 + 0.3)
    traj, obs = torch.randn((B, 795), generator=g) * 0.2, torch.randn((B, 232), generator=g)
    eps_l, eps_a = torch.randn((B, 64), generator=g), torch.randn((B, 30), generator=g)
    from vnl_brax_imitation_amd.ppo_imitation.hip_policy import HipIntentionPolicy

    hp = HipIntentionPolicy(n.policy_module, 30, B, dev)
    d = lambda t: t.to(dev)  # noqa: E731
    act, ex = hp.forward(d(flat), d(st.mean), d(st.std), d(traj), d(obs), d(eps_l), d(eps_a), deterministic=False)
    # torch reference (CPU, float64 NumPy restatement for the network)
    P = {k: v.double().numpy() for k, v in n.policy_network.layout.views(flat).items()}
    obs_n = (obs.double().numpy() - st.mean.double().numpy()) / st.std.double().numpy()
    rl, rm, rv = O.policy_forward(P, ENC, list(DEC) + [60], traj.double().numpy(), obs_n, eps_l.double().numpy())
    # float32 accumulation over K <= 1027 against the float64 restatement: 2e-5 absolute on O(1) outputs
    errs = {k: np.abs(ex[k].cpu().numpy() - r).max() for k, r in (("logits", rl), ("latent_mean", rm), ("latent_logvar", rv))}
    print("policy forward max abs error vs float64:", errs)
    assert max(errs.values()) < 2e-5, errs
    dist = n.parametric_action_distribution
    lg = ex["logits"].cpu()
    raw = dist.sample_no_postprocessing(lg, eps_a)
    assert torch.allclose(ex["raw_action"].cpu(), raw, atol=1e-5)
    assert torch.allclose(act.cpu(), torch.tanh(raw), atol=1e-5)
    lp_ref = dist.log_prob(lg.double(), raw.double())  # float64 elementwise math on the kernel's own logits / raw actions
    lp_err = float((ex["log_prob"].cpu().double() - lp_ref).abs().max() / lp_ref.abs().max())
    print(f"policy log_prob max error relative to its scale ({float(lp_ref.abs().max()):.1f}): {lp_err:.2e}")
    assert lp_err < 2e-6, lp_err  # a sum of 30 float32 terms of O(1..10)
    # deterministic mode
    act_d, ex_d = hp.forward(d(flat), d(st.mean), d(st.std), d(traj), d(obs), d(eps_l), None, deterministic=True)
    assert torch.allclose(act_d.cpu(), torch.tanh(lg[:, :30]), atol=1e-5)


def test_make_policy_uses_hip_kernel_on_gpu():
    dev = torch.device("cuda:0")
    n = ppo_networks.make_intention_ppo_networks(795, 232, 30, preprocess_observations_fn=running_statistics.normalize,
                                                 intention_latent_size=64, encoder_layer_sizes=ENC, decoder_layer_sizes=DEC)
    g = torch.Generator().manual_seed(0)
    flat = n.policy_network.init(g).to(dev)
    st = running_statistics.init_state(232, device=dev)
    mk = ppo_networks.make_inference_fn(n)
    pol_hip, pol_torch = mk((st, flat)), mk((st, flat), backend="torch")
    assert pol_hip.__name__ == "policy_hip" and pol_torch.__name__ == "policy"
    traj, obs = torch.randn((64, 795), device=dev) * 0.1, torch.randn((64, 232), device=dev)
    a1, e1 = pol_hip(traj, obs, torch.Generator(device=dev).manual_seed(5))
    a2, e2 = pol_torch(traj, obs, torch.Generator(device=dev).manual_seed(5))
    assert set(e1) == set(e2) == {"log_prob", "rand_log_prob", "raw_action", "logits"}
    assert torch.allclose(e1["logits"], e2["logits"], atol=2e-4) and torch.allclose(a1, a2, atol=2e-4)
    # same generator, same draw order: the kernel's rand_log_prob is that of the same random action
    # (two float32 implementations of the network: the logits differ by ~5e-6, and the log-prob's sensitivity to them is
    # |d log_prob / d logits| ~ z / scale, up to ~1e2 for a random action five standard deviations out)
    sc = float(e2["log_prob"].abs().max())
    for k in ("rand_log_prob", "log_prob"):
        err = float((e1[k] - e2[k]).abs().max()) / max(float(e2[k].abs().max()), 1.0)
        print(f"{k}: hip vs torch max error relative to its scale: {err:.2e}")
        assert err < 2e-5, (k, err)
