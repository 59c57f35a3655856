#!/usr/bin/env python3
"""Gradient check of the hand-written PPO minibatch step ALONG a training run, not only at initialisation: train a few
epochs with it, then take a fresh rollout with the trained policy, cut one minibatch out of it and compare
vnl_ppo_minibatch_grad with float64 torch autograd through the op-by-op loss on exactly those inputs (trained parameters,
real env data, moved normaliser).  Prints the per-tensor relative gradient errors and the loss terms."""
import argparse
import functools
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import helpers as H  # noqa: E402
from vnl_brax_imitation_amd import configs  # noqa: E402
from vnl_brax_imitation_amd.envs.rodent import RodentTracking  # noqa: E402
from vnl_brax_imitation_amd.envs.wrappers import wrap  # noqa: E402
from vnl_brax_imitation_amd.ppo_imitation import acting, hip_update, intention_losses, ppo_networks, running_statistics  # noqa: E402
from vnl_brax_imitation_amd.ppo_imitation import train as ppo  # noqa: E402


def check(envs: int = 4096, epochs: int = 4, backend: str = "hip", updates=None) -> dict:
    a = argparse.Namespace(envs=envs, epochs=epochs, backend=backend)
    dev = torch.device("cuda", 0)
    B = a.envs
    env = RodentTracking(H.reference_clip(), num_envs=B, device=dev, **H.env_kwargs())
    c = configs.TRAIN_CONFIG
    nf = functools.partial(ppo_networks.make_intention_ppo_networks, intention_latent_size=c["intention_latent_size"],
                           encoder_layer_sizes=c["encoder_layer_sizes"], decoder_layer_sizes=c["decoder_layer_sizes"])
    unroll, nmb = c["unroll_length"], c["num_minibatches"]
    hp = dict(entropy_cost=c["entropy_cost"], discounting=c["discounting"], reward_scaling=1.0, gae_lambda=0.95,
              clipping_epsilon=c["clipping_epsilon"], normalize_advantage=True, kl_weight=c["kl_weight"])
    make_policy, (norm, _), _ = ppo.train(
        environment=env, num_timesteps=a.epochs * B * unroll, episode_length=c["episode_length"], num_envs=B,
        learning_rate=c["learning_rate"], entropy_cost=c["entropy_cost"], discounting=c["discounting"], unroll_length=unroll,
        batch_size=B // nmb, num_minibatches=nmb, num_updates_per_batch=updates or c["num_updates_per_batch"], num_evals=1,
        normalize_observations=True, network_factory=nf, num_eval_envs=0, eval_env=None, kl_weight=c["kl_weight"],
        clipping_epsilon=c["clipping_epsilon"], update_backend=a.backend)
    nets, ts = ppo.train.last_ppo_network, ppo.train.last_training_state
    flat = ts.params.detach().clone()
    n_pol = nets.policy_network.layout.size
    # a fresh rollout with the trained policy, time-major [T, B, ...]; minibatch = the first 128 envs
    w = wrap(env, episode_length=c["episode_length"])
    st = w.reset(123)
    policy = make_policy((norm, flat[:n_pol]))
    g = torch.Generator(device=dev).manual_seed(5)
    st, data = acting.generate_unroll(w, st, policy, g, unroll, extra_fields=("truncation", "traj"))
    mb = min(128, B)
    tm = data.map(lambda x: x[:, :mb].contiguous())
    T = unroll
    noise = {"latent": torch.randn((T, mb, nets.policy_module.latents), generator=g, device=dev),
             "entropy": torch.randn((T, mb, nets.parametric_action_distribution.event_size), generator=g, device=dev)}
    upd = hip_update.HipPPOUpdate(nets, T, mb, dev, **hp)
    grads = torch.full((flat.numel(),), float("nan"), device=dev)
    mt = upd.grad(flat.contiguous(), norm, tm, noise, grads).cpu().numpy()
    gh = grads.cpu().numpy().astype(np.float64)
    # float64 autograd on the CPU
    p64 = flat.cpu().double().requires_grad_(True)
    n64 = running_statistics.RunningStatisticsState(*(getattr(norm, k).cpu().double() for k in ("count", "mean", "summed_variance", "std")))
    params = intention_losses.PPONetworkParams(policy=p64[:n_pol], value=p64[n_pol:])
    loss, m_ref = intention_losses.compute_ppo_intention_loss(
        params, n64, tm.map(lambda x: x.cpu().double()), None, ppo_network=nets, noise={k: v.cpu().double() for k, v in noise.items()},
        head="torch", time_major=True, **hp)
    loss.backward()
    gr = p64.grad.numpy()
    out = {}
    for lay, off0, tag in ((nets.policy_network.layout, 0, "policy"), (nets.value_network.layout, n_pol, "value")):
        for name, (off, shape) in lay.entries.items():
            n = int(np.prod(shape))
            x, y = gh[off0 + off: off0 + off + n], gr[off0 + off: off0 + off + n]
            out[f"{tag}/{name}"] = float(np.abs(x - y).max() / max(np.abs(y).max(), 1e-30))
    worst = max(out, key=out.get)
    return {"epochs": a.epochs, "worst tensor": worst, "worst rel err": out[worst],
            "median rel err": float(np.median(list(out.values()))), "losses hip": [float(v) for v in mt[:6]],
            "losses f64": [float(m_ref[k]) for k in ("total_loss", "policy_loss", "v_loss", "entropy_loss", "kl_loss_intention",
                                                      "explained_variance")], "per tensor": out}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=4096)
    ap.add_argument("--epochs", type=int, default=4)
    ap.add_argument("--backend", default="hip")
    a = ap.parse_args()
    r = check(a.envs, a.epochs, a.backend)
    out = r.pop("per tensor")
    print(json.dumps(r))
    for k in sorted(out, key=out.get, reverse=True)[:6]:
        print(f"   {k:45s} {out[k]:.2e}")


if __name__ == "__main__":
    main()
