#!/usr/bin/env python3
"""Short PPO training run on one GPU with periodic evaluation: shows that the whole hot path (fused rollout ->
GAE / PPO head -> graph-captured minibatch step -> Adam) learns.  Rodent, 4096 envs, reference hyper-parameters
(configs.TRAIN_CONFIG), 512 evaluation envs on the same clip.  Prints one JSON line per evaluation.
    python tools/train_curve.py [--train-steps 48] [--evals 7]"""
import argparse
import functools
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

import helpers as H  # noqa: E402
from vnl_brax_imitation_amd import configs  # noqa: E402
from vnl_brax_imitation_amd.envs.rodent import RodentTracking  # noqa: E402
from vnl_brax_imitation_amd.ppo_imitation import ppo_networks  # noqa: E402
from vnl_brax_imitation_amd.ppo_imitation import train as ppo  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--train-steps", type=int, default=48)
    ap.add_argument("--evals", type=int, default=7)
    ap.add_argument("--envs", type=int, default=4096)
    ap.add_argument("--eval-envs", type=int, default=512)
    ap.add_argument("--updates", type=int, default=None)
    ap.add_argument("--backend", default="auto", help="minibatch step: hip (hand-written) | torch (autograd) | auto")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--reset-info", action="store_true", help="reset info (frame counters, traj) on auto-reset: the fix of the "
                    "reference's quirk C.20 (without it every env is `done` at every step after its first ten)")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    B = args.envs
    env = RodentTracking(H.reference_clip(), num_envs=B, device=dev, **H.env_kwargs())
    eval_env = RodentTracking(H.reference_clip(), num_envs=args.eval_envs, device=dev, **H.env_kwargs())
    c = configs.TRAIN_CONFIG
    nf = functools.partial(ppo_networks.make_intention_ppo_networks, intention_latent_size=c["intention_latent_size"],
                           encoder_layer_sizes=c["encoder_layer_sizes"], decoder_layer_sizes=c["decoder_layer_sizes"])
    unroll, nmb = c["unroll_length"], c["num_minibatches"]
    t0 = time.time()

    def progress(step, m):
        keep = {k: (float(v) if hasattr(v, "__float__") else v) for k, v in m.items()
                if k in ("eval/episode_reward", "eval/episode_reward_std", "eval/avg_episode_length", "training/sps",
                         "training/total_loss", "training/policy_loss", "training/v_loss", "training/kl_loss")}
        print(json.dumps({"env_steps": int(step), "wall_s": round(time.time() - t0, 1), **keep}), flush=True)

    ppo.train(environment=env, num_timesteps=args.train_steps * B * unroll, episode_length=c["episode_length"],
              num_envs=B, learning_rate=c["learning_rate"], entropy_cost=c["entropy_cost"], discounting=c["discounting"],
              unroll_length=unroll, batch_size=B // nmb, num_minibatches=nmb,
              num_updates_per_batch=args.updates or c["num_updates_per_batch"], num_evals=args.evals,
              normalize_observations=True, network_factory=nf, num_eval_envs=args.eval_envs, eval_env=eval_env,
              kl_weight=c["kl_weight"], clipping_epsilon=c["clipping_epsilon"], progress_fn=progress,
              update_backend=args.backend, seed=args.seed, reset_info_on_autoreset=args.reset_info)


if __name__ == "__main__":
    main()
