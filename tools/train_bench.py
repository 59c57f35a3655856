#!/usr/bin/env python3
"""Full PPO training step on the GPU (BASELINE config 5, per-GPU share): rodent, 4096 envs/GPU,
unroll_length 20, batch_size 128/GPU, 32 minibatches, 16 updates per batch, reference network sizes.
Single GPU:   python tools/train_bench.py [--steps N]
Multi GPU:    python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 tools/train_bench.py
Prints env-steps/s of the whole train step (rollout + normaliser + SGD) as reported by training/sps."""
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
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--envs-per-gpu", type=int, default=4096)
    ap.add_argument("--updates", type=int, default=16)
    ap.add_argument("--clips", type=int, default=1, help="BASELINE configs[3]: synthesised multi-clip reference, clip id per env")
    ap.add_argument("--backend", default="auto", help="minibatch step: hip (hand-written fwd+bwd) | torch (autograd) | auto")
    ap.add_argument("--force-dist", action="store_true", help="initialise the RCCL process group even with one rank "
                    "(exercises the data-parallel code path: eager all-reduce + Adam between graph replays)")
    args = ap.parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 or args.force_dist:
        import torch.distributed as dist

        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"), os.environ.setdefault("MASTER_PORT", "29677")
        os.environ.setdefault("RANK", "0"), os.environ.setdefault("WORLD_SIZE", "1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    dev = torch.device("cuda", local_rank)
    B = args.envs_per_gpu
    clip = H.reference_clip()
    if args.clips > 1:
        from vnl_brax_imitation_amd.preprocessing import mjx_preprocess as pp

        clip = pp.synthesize_clips(H.model(), H.golden_qpos(), args.clips, seed=0)
    env = RodentTracking(clip, num_envs=B, device=dev, **H.env_kwargs())
    c = configs.TRAIN_CONFIG
    nf = functools.partial(ppo_networks.make_intention_ppo_networks, intention_latent_size=c["intention_latent_size"],
                           encoder_layer_sizes=c["encoder_layer_sizes"], decoder_layer_sizes=c["decoder_layer_sizes"])
    unroll, nmb = c["unroll_length"], c["num_minibatches"]
    batch = B * world // nmb  # batch_size * num_minibatches == num_envs: one unroll per training step
    log = []
    t0 = time.time()
    ppo.train(environment=env, num_timesteps=args.steps * B * world * unroll, episode_length=c["episode_length"],
              num_envs=B * world, learning_rate=c["learning_rate"], entropy_cost=c["entropy_cost"],
              discounting=c["discounting"], unroll_length=unroll, batch_size=batch, num_minibatches=nmb,
              num_updates_per_batch=args.updates, num_evals=1, normalize_observations=True, network_factory=nf,
              num_eval_envs=0, eval_env=None, kl_weight=c["kl_weight"], clipping_epsilon=c["clipping_epsilon"],
              update_backend=args.backend, progress_fn=lambda s, m: log.append((s, m)))
    if int(os.environ.get("RANK", "0")) == 0:
        s, m = log[-1]
        print(json.dumps({"env_steps": s, "training/sps": m["training/sps"], "wall_s": time.time() - t0,
                          "total_loss": m["training/total_loss"], "v_loss": m["training/v_loss"], "n_gpus": world,
                          "steps": args.steps, "updates_per_batch": args.updates, "backend": args.backend, "clips": args.clips}))
    if world > 1 or args.force_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
