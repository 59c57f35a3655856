#!/usr/bin/env python3
"""Headline benchmark: rodent-imitation rollout, env-steps/sec, 4096 envs per GPU.

One "step" = one pass of the hot path of SURVEY.md 3.1 over all envs of a rank:
intention-policy forward (fused HIP kernel) -> RodentTracking.step (ONE HIP kernel: 5 physics
substeps + obs / traj / reward / termination) -> brax Episode/AutoReset wrappers -> Transition
row written into the unroll buffers (acting.actor_step / generate_unroll, unroll_length 20).
Weights are freshly initialised (no checkpoints offline); the state lives in HBM throughout.
`--random-actions` replaces the policy by pre-generated clip(0.3*N(0,1), -1, 1) actions (the
protocol of the reference's notebooks/test_rodent.ipynb) and skips the Transition logging.

    python bench.py --gpus 1 --steps 100 --warmup 10
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line on rank 0 (see DESIGN.md "Measurement").
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402
import torch  # noqa: E402

ENVS_PER_GPU = 4096
# algorithmic HBM bytes per env-step of the rollout (state in/out, action,
# obs, traj, reward/done/metrics/info), and tree-sparse algorithmic flops per env-step.
B_ALG = 6284.0
F_ALG = 3.9e6 - 0.68e6  # step kernel only (the policy forward is a separate kernel)
# HBM bytes per step-kernel launch from the PMC counters (profiles/r01f_full_path_summary.md: separate
# `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes of this command, 4096 envs): 11,244 KB fetched
# (x2 gfx950 read correction = 22.5 MB, an upper bound for 4-B/lane accesses) + 32.0 MB written.
TRAFFIC_PMC_BYTES_4096 = 22.5e6 + 32.0e6
HBM_PEAK_GBS = 8000.0
VALU_PEAK_TFLOPS = 157.3


def cpu_baseline(env, num_envs: int, max_steps: int = 200, budget_s: float = 12.0, seed: int = 1) -> dict:
    """Time the CPU oracle (float32 build, OpenMP over envs) on a bounded sample of the same workload: control
    steps of `num_envs` envs until `budget_s` seconds of CPU work are spent (at most `max_steps`)."""
    import helpers as H

    o = H.make_oracle(env, "f32")
    rng = np.random.default_rng(seed)
    sf = rng.integers(0, 235, num_envs).astype(np.int32)
    noise = (1e-3 * rng.standard_normal((num_envs, 74))).astype(np.float32)
    st = o.env_reset(sf, noise)
    acts = np.clip(0.3 * rng.standard_normal((max_steps, num_envs, 30)), -1, 1).astype(np.float32)
    o.env_step(st, acts[0])  # warm-up (thread pool, page faults)
    t0 = time.perf_counter()
    steps = 0
    while steps < max_steps:
        o.env_step(st, acts[steps])
        steps += 1
        if time.perf_counter() - t0 >= budget_s:
            break
    dt = time.perf_counter() - t0
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    return dict(value=num_envs * steps / dt, unit="env-steps/s", cores=cores, kind="port",
                sample=f"{num_envs} envs x {steps} control steps, C restatement (oracle/vnl_oracle.c, float32, "
                       f"OpenMP over envs), {dt:.1f} s")


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--envs-per-gpu", type=int, default=ENVS_PER_GPU)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-autoreset", action="store_true")
    ap.add_argument("--random-actions", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    distributed = world > 1
    if distributed:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    assert args.gpus == world, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    dev = torch.device("cuda", local_rank)

    import helpers as H
    from vnl_brax_imitation_amd.envs.rodent import RodentTracking
    from vnl_brax_imitation_amd.envs.wrappers import AutoResetWrapper, EpisodeWrapper

    B = args.envs_per_gpu
    base = RodentTracking(H.reference_clip(), num_envs=B, device=dev, **H.env_kwargs())
    env = base if args.no_autoreset else AutoResetWrapper(EpisodeWrapper(base, episode_length=150, action_repeat=1))
    gen = torch.Generator(device="cpu")
    gen.manual_seed(1234 + rank)  # independent stream per rank: envs shard, nothing crosses the links
    state = env.reset(gen)
    total = args.steps + args.warmup
    unroll = 20
    if args.random_actions:
        actions = torch.clamp(0.3 * torch.randn((total, B, 30), generator=gen), -1.0, 1.0).to(dev)

        def run(k0, n, timed):
            nonlocal state
            for k in range(n):
                if timed:
                    base.kernel_events = ev[k]
                state = env.step(state, actions[k0 + k])
    else:
        from vnl_brax_imitation_amd import configs
        from vnl_brax_imitation_amd.ppo_imitation import acting, ppo_networks, running_statistics

        c = configs.TRAIN_CONFIG
        nets = ppo_networks.make_intention_ppo_networks(
            base.traj_size, base.observation_size, base.action_size, preprocess_observations_fn=running_statistics.normalize,
            intention_latent_size=c["intention_latent_size"], encoder_layer_sizes=c["encoder_layer_sizes"],
            decoder_layer_sizes=c["decoder_layer_sizes"])
        flat = nets.policy_network.init(torch.Generator().manual_seed(0)).to(dev)
        norm = running_statistics.init_state(base.observation_size, device=dev)
        policy = ppo_networks.make_inference_fn(nets)((norm, flat))
        assert policy.__name__ == "policy_hip"
        gdev = torch.Generator(device=dev).manual_seed(99 + rank)

        class _Timed:  # records the (start, end) events of every step-kernel launch of the timed region
            def __init__(self):
                self.k = 0

        tm = _Timed()
        orig_step = base.step

        def step_hook(st_, a_):
            if tm.k is not None and tm.k < len(ev):
                base.kernel_events = ev[tm.k]
                tm.k += 1
            return orig_step(st_, a_)

        extra = ("traj",) if args.no_autoreset else ("truncation", "traj")  # truncation comes from EpisodeWrapper

        def run(k0, n, timed):
            nonlocal state
            tm.k = 0 if timed else None
            base.step = step_hook if timed else orig_step
            done = 0
            while done < n:
                chunk = min(unroll, n - done)
                state, _ = acting.generate_unroll(env, state, policy, gdev, chunk, extra_fields=extra)
                done += chunk
            base.step = orig_step
    torch.cuda.synchronize(dev)

    ev = []
    run(0, args.warmup, False)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    base.kernel_events = None
    torch.cuda.synchronize(dev)
    if distributed:
        dist.barrier()
        torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    run(args.warmup, args.steps, True)
    torch.cuda.synchronize(dev)
    if distributed:
        dist.barrier()
        torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t0
    base.kernel_events = None
    if distributed:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    kernel_ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))
    finite = bool(torch.isfinite(state.obs).all().item())

    if rank == 0:
        value = world * B * args.steps / dt
        per_gpu_kernel_rate = B / (kernel_ms * 1e-3)
        achieved_gbs = B_ALG * B / (kernel_ms * 1e-3) / 1e9
        out = {
            "metric": "env-steps/sec (whole node), rodent imitation, num_envs=4096/GPU",
            "value": value,
            "unit": "env-steps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": ("synthetic: " + ("random actions clip(0.3*N(0,1),-1,1)" if args.random_actions else
                                      "freshly initialised intention network (seed 0)") +
                     "; reference clip = the shipped groom clip re-processed to 66 bodies; model = compiled rodent.xml "
                     "(scale 0.9); start frames / reset noise from torch Philox"),
            "config": {
                "workload": "rodent imitation rollout, single groom clip, "
                            f"{B} envs/GPU: " + ("random actions -> " if args.random_actions else
                                                 "intention-policy forward (HIP) -> ") +
                            "RodentTracking.step (5 substeps, CG 6/6) -> " +
                            f"auto-reset {'off' if args.no_autoreset else 'on'}" +
                            ("" if args.random_actions else " -> Transition logging (unroll 20)"),
                "envs_per_gpu": B,
                "parallelism": f"env-sharded x{world}, no data-path collective",
            },
            "roofline": {
                "bound": "hbm",
                "kernel": "vnl_step_kernel",
                "achieved": achieved_gbs,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved_gbs / HBM_PEAK_GBS,
                "traffic": TRAFFIC_PMC_BYTES_4096 if B == 4096 else None,
                "kernel_ms": kernel_ms,
                "algorithmic_bytes_per_launch": B_ALG * B,
                "note": "the fused step is FP32-VALU/latency bound, not HBM bound (SURVEY 8d); valu_frac is the "
                        "honest figure",
                "valu_achieved_tflops": per_gpu_kernel_rate * F_ALG / 1e12,
                "valu_peak_tflops": VALU_PEAK_TFLOPS,
                "valu_frac": per_gpu_kernel_rate * F_ALG / 1e12 / VALU_PEAK_TFLOPS,
            },
            "finite": finite,
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(base, num_envs=1024)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out))
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
