#!/usr/bin/env python3
"""Train the same run twice on one GPU -- hand-written update vs torch autograd -- and print the training metrics of every
epoch side by side (same seeds: the rollouts are identical until the parameters differ).  A diagnostic for systematic
differences between the two minibatch steps that a single-step comparison would not show."""
import argparse
import functools
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

import helpers as H  # noqa: E402
from vnl_brax_imitation_amd import configs  # noqa: E402
from vnl_brax_imitation_amd.envs.rodent import RodentTracking  # noqa: E402
from vnl_brax_imitation_amd.ppo_imitation import ppo_networks  # noqa: E402
from vnl_brax_imitation_amd.ppo_imitation import train as ppo  # noqa: E402


def run(backend, args):
    dev = torch.device("cuda", 0)
    B = args.envs
    env = RodentTracking(H.reference_clip(), num_envs=B, device=dev, **H.env_kwargs())
    c = configs.TRAIN_CONFIG
    nf = functools.partial(ppo_networks.make_intention_ppo_networks, intention_latent_size=c["intention_latent_size"],
                           encoder_layer_sizes=c["encoder_layer_sizes"], decoder_layer_sizes=c["decoder_layer_sizes"])
    unroll, nmb = c["unroll_length"], c["num_minibatches"]
    log = []
    _, (norm, flat), _ = ppo.train(
        environment=env, num_timesteps=args.epochs * B * unroll, episode_length=c["episode_length"], num_envs=B,
        learning_rate=c["learning_rate"], entropy_cost=c["entropy_cost"], discounting=c["discounting"], unroll_length=unroll,
        batch_size=B // nmb, num_minibatches=nmb, num_updates_per_batch=args.updates, num_evals=args.epochs + 1,
        normalize_observations=True, network_factory=nf, num_eval_envs=0, eval_env=None, kl_weight=c["kl_weight"],
        clipping_epsilon=c["clipping_epsilon"], update_backend=backend, seed=args.seed,
        progress_fn=lambda s, m: log.append((s, {k: float(v) for k, v in m.items() if k.startswith("training/")})))
    return log, flat.detach().cpu(), ppo.train.last_training_state.params.detach().cpu()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=4096)
    ap.add_argument("--epochs", type=int, default=6)
    ap.add_argument("--updates", type=int, default=16)
    ap.add_argument("--seed", type=int, default=0)
    a = ap.parse_args()
    la, pa, fa = run("hip", a)
    lb, pb, fb = run("torch", a)
    keys = ("training/total_loss", "training/policy_loss", "training/v_loss", "training/entropy_loss", "training/kl_loss_intention",
            "training/explained_variance", "training/prediction_corr")
    for (s, ma), (_, mb) in zip(la, lb):
        if not ma:
            continue
        print(json.dumps({"env_steps": s, **{k.split("/")[1]: [round(ma.get(k, float("nan")), 8), round(mb.get(k, float("nan")), 8)] for k in keys}}))
    n_pol = pa.numel()
    d = (fa - fb).abs()
    print(json.dumps({"param diff max (all)": float(d.max()), "policy part": float(d[:n_pol].max()), "value part": float(d[n_pol:].max()),
                      "param scale": float(fb.abs().max())}))


if __name__ == "__main__":
    main()
