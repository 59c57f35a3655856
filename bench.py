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
    python bench.py --gpus 8 ...      # starts the 8 ranks itself (torch.distributed.run) before touching a GPU
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line on rank 0 (see DESIGN.md "Measurement").  Besides the contract's keys the line carries
`free_running`: the same measurement with the auto-reset wrapper off (with the reference's C.20 wrapper bug every
env of the default workload is reset to its cached first state at every step after the tenth, so the default
figure re-steps that first state; the free-running figure lets the envs evolve).
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

ENVS_PER_GPU = 4096
# algorithmic HBM bytes per env-step of the rollout (state in/out, action, obs, traj,
# reward/done/metrics/info), and the survey's tree-sparse flop estimate per env-step of the step kernel (worst
# case: every CG / line-search iteration runs; an estimate, not a count of executed instructions).
B_ALG = 6284.0
F_ALG = 3.9e6 - 0.68e6  # step kernel only (the policy forward is a separate kernel)
# HBM bytes per step-kernel launch cannot be measured from inside this script (PMC counters need rocprofv3):
# `roofline.traffic` is read from profiles/traffic_latest.json, written by tools/pmc_traffic.py from separate
# `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes of this very command (the file names its source);
# null when that file is absent or was taken at another env count.
TRAFFIC_FILE = os.path.join(ROOT, "profiles", "traffic_latest.json")
HBM_PEAK_GBS = 8000.0
VALU_PEAK_TFLOPS = 157.3


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--envs-per-gpu", type=int, default=ENVS_PER_GPU)
    ap.add_argument("--clips", type=int, default=1, help="number of resident reference clips (synthesised from the "
                    "groom clip when > 1: SURVEY 8(d) config 4), random clip per env")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-autoreset", action="store_true", help="main measurement without the auto-reset wrapper")
    ap.add_argument("--no-free-running", action="store_true", help="skip the additional free-running measurement")
    ap.add_argument("--random-actions", action="store_true")
    ap.add_argument("--eager", action="store_true", help="issue the unroll's launches from Python instead of replaying the "
                    "captured hipGraph (the round-2 form of the hot path)")
    ap.add_argument("--no-train-step", action="store_true", help="skip the `train_step` sub-record (full PPO training step)")
    ap.add_argument("--no-train-secondary", action="store_true", help="skip `train_step.reference_proportions` (batch_size x num_minibatches = 8 x num_envs)")
    ap.add_argument("--train-steps", type=int, default=3, help="training steps timed for the `train_step` sub-record")
    ap.add_argument("--train-step-multi", action="store_true", help="also run the `train_step` leg with --gpus > 1 (RCCL "
                    "gradient all-reduce; off by default so that a collective problem cannot take the headline line down)")
    ap.add_argument("--no-parity", action="store_true", help="skip the oracle compliance check of the cpu_baseline leg")
    ap.add_argument("--config", choices=("rodent", "humanoid"), default="rodent", help="humanoid: BASELINE config 'Humanoid "
                    "imitation, num_envs=1024' (HumanoidTracking, synthetic standing clip, random actions; not the headline metric)")
    return ap.parse_args()


def spawn_ranks(args) -> int:
    """`--gpus N` outside a torchrun launch: start the N ranks as children of torch.distributed.run.  This parent
    has not imported torch and never touches the GPU; rank 0's JSON line goes straight to our stdout."""
    import socket
    import subprocess

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


def cpu_baseline(env, num_envs: int, max_steps: int = 200, budget_s: float = 12.0, seed: int = 1) -> dict:
    """Time the CPU oracle (float32 build, OpenMP over envs) on a bounded sample of the same workload: control
    steps of `num_envs` envs until `budget_s` seconds of CPU work are spent (at most `max_steps`)."""
    import numpy as np

    import helpers as H

    o = H.make_oracle(env, "f32")
    rng = np.random.default_rng(seed)
    nq, nu = int(env.dims.nq), int(env.dims.nu)
    sf = rng.integers(0, max(env._T - 15, 1), num_envs).astype(np.int32)
    noise = (float(env._reset_noise_scale) * rng.standard_normal((num_envs, nq))).astype(np.float32)
    st = o.env_reset(sf, noise)
    acts = np.clip(0.3 * rng.standard_normal((max_steps, num_envs, nu)), -1, 1).astype(np.float32)
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


def parity_compliance(env_cls, clip, kwargs, dev, num_envs: int = 512, seed: int = 3) -> dict:
    """Checker leg (rank 0, N = 1, beside cpu_baseline): the fraction of envs whose state after ONE physics substep and
    after ONE control step is within 1e-5 of the float64 oracle (following the product's solver decisions; errors per env
    and per field group, tests/parity.py) -- for the product on the device and, on the same inputs, for the float32 build
    of the oracle.  north_star's "within 1e-5 rel" holds for that fraction of the envs, not for every env: the
    6-iteration CG of the reference's configuration is not converged and float32 rounding decides its discrete path."""
    import numpy as np
    import torch

    import helpers as H
    import parity as P

    rng = np.random.default_rng(seed)
    sf = rng.integers(0, 235, num_envs).astype(np.int32)
    noise = (1e-3 * rng.standard_normal((num_envs, 74))).astype(np.float32)
    act = np.clip(0.3 * rng.standard_normal((num_envs, 30)), -1, 1).astype(np.float32)
    out = {"envs": num_envs, "tolerance": 1e-5, "scaling": "per env and per field group (root position / quaternion / joint "
           "angles; linear / angular / joint velocities)"}
    for name, nf in (("one_substep", 1), ("one_control_step", 5)):
        env = env_cls(clip, num_envs=num_envs, device=dev, **{**kwargs, "n_frames": nf})
        o64, o32 = H.make_oracle(env, "f64"), H.make_oracle(env, "f32")
        _, err, dev32, _, _ = P.control_step_follow(env, o64, o32, sf, noise, act)
        torch.cuda.synchronize(dev)
        out[name] = {"product": {k: P.compliance(err)[k] for k in ("qpos", "qvel")},
                     "float32_oracle": {k: P.compliance(dev32)[k] for k in ("qpos", "qvel")},
                     "median_err_qvel": float(np.median(err["qvel"])), "max_err_qvel": float(err["qvel"].max())}
    return out


def train_step_record(clip, kwargs, dev, B: int, world: int, steps: int, ratio: int = 1) -> dict:
    """Full PPO training step (BASELINE config 5, per-GPU share; reference ppo_imitation/train.py:293-394): 4096 envs,
    unroll 20, 32 minibatches x 16 updates of 128 trajectories, reference network sizes, the hand-written update inside
    the captured hipGraph.  Two epochs of `steps` training steps: the first contains the graph capture, the second is the
    steady state whose `training/sps` (as the reference computes it) is reported.
    `ratio`: batch_size x num_minibatches = ratio x num_envs -- 1 is SURVEY 8(d)'s primary config 5 (one unroll per training
    step), 8 the reference's own proportions (configs/train_config.yaml:4-11: 32 x 32 / 128): eight unrolls per training
    step, minibatches of 1024 trajectories."""
    import functools

    import torch

    from vnl_brax_imitation_amd import configs
    from vnl_brax_imitation_amd.envs.rodent import RodentTracking
    from vnl_brax_imitation_amd.ppo_imitation import ppo_networks
    from vnl_brax_imitation_amd.ppo_imitation import train as ppo

    env = RodentTracking(clip, num_envs=B, device=dev, **kwargs)
    c = configs.TRAIN_CONFIG
    nf = functools.partial(ppo_networks.make_intention_ppo_networks, intention_latent_size=c["intention_latent_size"],
                           encoder_layer_sizes=c["encoder_layer_sizes"], decoder_layer_sizes=c["decoder_layer_sizes"])
    unroll, nmb, upd = c["unroll_length"], c["num_minibatches"], c["num_updates_per_batch"]
    log = []
    ppo.train(environment=env, num_timesteps=2 * steps * ratio * B * world * unroll, episode_length=c["episode_length"],
              num_envs=B * world, learning_rate=c["learning_rate"], entropy_cost=c["entropy_cost"],
              discounting=c["discounting"], unroll_length=unroll, batch_size=ratio * B * world // nmb, num_minibatches=nmb,
              num_updates_per_batch=upd, num_evals=3, normalize_observations=True, network_factory=nf, num_eval_envs=0,
              eval_env=None, kl_weight=c["kl_weight"], clipping_epsilon=c["clipping_epsilon"], update_backend="hip",
              progress_fn=lambda s_, m_: log.append((s_, dict(m_))))
    torch.cuda.synchronize(dev)
    if not log:
        return {}
    m = log[-1][1]
    sps = float(m["training/sps"])
    step_ms = ratio * B * world * unroll / sps * 1e3
    return {"env_steps_per_s": sps, "ms_per_training_step": step_ms, "training_steps_timed": steps,
            "minibatch_steps_per_training_step": nmb * upd, "total_loss": float(m["training/total_loss"]),
            "config": f"rodent, {B} envs/GPU x {world} GPU, unroll {unroll}, {nmb} minibatches x {upd} updates of "
                      f"{ratio * B // nmb} trajectories/GPU ({ratio} unroll(s) per training step), intention net {c['encoder_layer_sizes']}/{c['intention_latent_size']}/"
                      f"{c['decoder_layer_sizes']}, value (1024, 1024), hand-written update in a captured hipGraph, "
                      "steady state (second epoch)"}


def update_record(dev, iters: int = 60) -> dict:
    """The hand-written minibatch step alone (vnl_ppo_minibatch_grad, graph replay, synthetic minibatch of the
    reference's sizes T = 20 x 128 trajectories): ms per minibatch step and fp32-MFMA TFLOP/s (2 x MACs of every GEMM,
    forward + dX + dW)."""
    import torch

    from test_gpu_ppo_update import HP, _make
    from vnl_brax_imitation_amd.ppo_imitation import hip_update, running_statistics

    cfg = dict(traj=795, obs=232, act=30, latent=64, enc=(256, 128), dec=(128, 256), val=(1024, 1024), T=20, B=128)
    nets, flat, data, norm, noise = _make(**cfg)
    to = lambda t: t.to(dev)  # noqa: E731
    flat, data, noise = to(flat).contiguous(), data.map(to), {k: to(v) for k, v in noise.items()}
    ndev = running_statistics.RunningStatisticsState(to(norm.count), to(norm.mean), to(norm.summed_variance), to(norm.std))
    grads = torch.zeros_like(flat)
    upd = hip_update.HipPPOUpdate(nets, cfg["T"], cfg["B"], dev, **HP)
    for _ in range(3):
        upd.grad(flat, ndev, data, noise, grads)
    torch.cuda.synchronize(dev)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        upd.grad(flat, ndev, data, noise, grads)
    for _ in range(5):
        g.replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        g.replay()
    e1.record()
    torch.cuda.synchronize(dev)
    ms = e0.elapsed_time(e1) / iters
    N = cfg["T"] * cfg["B"]
    macs_p = 795 * 256 + 256 * 128 + 2 * 128 * 64 + 296 * 128 + 128 * 256 + 256 * 60
    macs_v = 232 * 1024 + 1024 * 1024 + 1024
    flop = 2.0 * 3.0 * (N * macs_p + N * macs_v) + 2.0 * cfg["B"] * macs_v
    return {"ms_per_minibatch_step": ms, "mfma_tflops": flop / ms / 1e9, "mfma_frac": flop / ms / 1e9 / VALU_PEAK_TFLOPS,
            "mfma_peak_tflops": VALU_PEAK_TFLOPS, "gflop_per_minibatch_step": flop / 1e9,
            "note": "fp32 MFMA (v_mfma_f32_32x32x2_f32) peak = the fp32 vector peak on gfx950"}


def main() -> None:
    args = parse_args()
    if args.gpus > 1 and "RANK" not in os.environ:
        sys.exit(spawn_ranks(args))

    import numpy as np
    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    distributed = world > 1
    dist = None
    if distributed:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # VNL_BENCH_REHEARSAL=1: every rank on cuda:0 with gloo for the barrier / MAX -- rehearses the N-rank control flow on a
        # one-GPU box (RCCL refuses two ranks on one device); the numbers of such a run mean nothing
        rehearsal = os.environ.get("VNL_BENCH_REHEARSAL") == "1"
        if rehearsal:
            local_rank = 0
            dist.init_process_group("gloo")
        else:
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    assert args.gpus == world, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    dev = torch.device("cuda", local_rank)
    red_dev = torch.device("cpu") if (distributed and os.environ.get("VNL_BENCH_REHEARSAL") == "1") else dev

    import helpers as H
    from vnl_brax_imitation_amd.envs.rodent import RodentTracking
    from vnl_brax_imitation_amd.envs.wrappers import AutoResetWrapper, EpisodeWrapper

    B = args.envs_per_gpu
    if args.config == "humanoid":
        from vnl_brax_imitation_amd.envs.humanoid import HumanoidTracking
        from vnl_brax_imitation_amd.model import mjcf

        if B == ENVS_PER_GPU:
            B = 1024
        args.random_actions = True
        hm = mjcf.CompiledModel.load(os.path.join(ROOT, "vnl-brax-imitation_amd", "data", "humanoid.npz"))
        base = HumanoidTracking(dict(solver="cg", iterations=6, ls_iterations=6), model=hm, num_envs=B, device=dev)
    else:
        clip = H.reference_clip()
        if args.clips > 1:
            from vnl_brax_imitation_amd.preprocessing import mjx_preprocess as pp

            clip = pp.synthesize_clips(H.model(), H.golden_qpos(), args.clips, seed=0)
        base = RodentTracking(clip, num_envs=B, device=dev, **H.env_kwargs())
    unroll = 20
    policy = gdev = acting = None
    if not args.random_actions:
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

    def measure(autoreset: bool, steps: int, warmup: int) -> dict:
        """W untimed + K timed steps of the hot path, barrier + synchronize on both sides, MAX over ranks; the step
        kernel's own duration from HIP events recorded around its launch on the launch stream."""
        env = AutoResetWrapper(EpisodeWrapper(base, episode_length=150, action_repeat=1)) if autoreset else base
        gen = torch.Generator(device="cpu")
        gen.manual_seed(1234 + rank)  # independent stream per rank: envs shard, nothing crosses the links
        state = env.reset(gen)
        ev = []
        orig_step = base.step
        k_ev = [None]

        def step_hook(st_, a_):
            if k_ev[0] is not None and k_ev[0] < len(ev):
                base.kernel_events = ev[k_ev[0]]
                k_ev[0] += 1
            return orig_step(st_, a_)

        if args.random_actions:
            actions = torch.clamp(0.3 * torch.randn((steps + warmup, B, base.action_size), generator=gen), -1.0, 1.0).to(dev)

            graphed = None

            def run(k0, n, timed, events=False):
                nonlocal state
                k_ev[0] = 0 if events else None
                base.step = step_hook if events else orig_step
                for k in range(n):
                    state = env.step(state, actions[k0 + k])
                base.step = orig_step
        else:
            extra = ("truncation", "traj") if autoreset else ("traj",)  # truncation comes from EpisodeWrapper
            # the unroll of 20 control steps as the trainer runs it: ONE hipGraph replay (acting.GraphedUnroll, bit-identical to
            # the eager loop: tests/test_fused_rollout.py); a remainder of steps % 20 and the free-running leg run eagerly
            graphed = acting.GraphedUnroll(env, state, policy, gdev, unroll, extra_fields=extra) if (autoreset and not args.eager) else None

            def run(k0, n, timed, events=False):
                nonlocal state
                k_ev[0] = 0 if events else None
                base.step = step_hook if events else orig_step
                done = 0
                while done < n:
                    chunk = min(unroll, n - done)
                    if graphed is not None and chunk == unroll and not events:
                        state, _ = graphed()
                    else:
                        state, _ = acting.generate_unroll(env, state, policy, gdev, chunk, extra_fields=extra)
                    done += chunk
                base.step = orig_step

        torch.cuda.synchronize(dev)
        run(0, warmup, False)
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
        base.kernel_events = None
        torch.cuda.synchronize(dev)
        if distributed:
            dist.barrier()
            torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        run(warmup, steps, True, events=graphed is None)
        torch.cuda.synchronize(dev)
        if distributed:
            dist.barrier()
            torch.cuda.synchronize(dev)
        dt = time.perf_counter() - t0
        base.kernel_events = None
        if graphed is not None:
            # HIP events recorded inside a captured graph cannot be timed: the step kernel's launch duration comes from an
            # eager re-run of the same hot path on the same envs right after the timed region (`kernel_ms_source`)
            nk = min(steps, 40)
            ev = ev[:nk]
            if args.random_actions:
                run(0, nk, False, events=True)
            else:
                run(warmup, nk, False, events=True)
            torch.cuda.synchronize(dev)
            base.kernel_events = None
        if distributed:
            t = torch.tensor([dt], dtype=torch.float64, device=red_dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        kernel_ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))
        return dict(dt=dt, kernel_ms=kernel_ms, finite=bool(torch.isfinite(state.obs).all().item()),
                    done_frac=float(state.done.float().mean().item()), graphed=graphed is not None)

    main_m = measure(not args.no_autoreset, args.steps, args.warmup)
    free_m = None
    if not args.no_free_running and not args.no_autoreset:
        free_m = measure(False, args.steps, args.warmup)

    if rank == 0:
        dt, kernel_ms = main_m["dt"], main_m["kernel_ms"]
        value = world * B * args.steps / dt
        per_gpu_kernel_rate = B / (kernel_ms * 1e-3)
        hbm_gbs = B_ALG * B / (kernel_ms * 1e-3) / 1e9
        valu_tflops = per_gpu_kernel_rate * F_ALG / 1e12
        traffic, traffic_src = None, None
        tpath = next((p_ for p_ in (os.path.join(ROOT, "gpurun_out", "traffic_latest.json"), TRAFFIC_FILE) if os.path.exists(p_)), None)
        if tpath:
            import hashlib

            tf = json.load(open(tpath))
            hsh = hashlib.sha256()
            for f in ("vnl_body.h", "vnl_lib.hip", "vnl_types.h"):
                hsh.update(open(os.path.join(ROOT, "vnl-brax-imitation_amd", "csrc", f), "rb").read())
            # a figure taken with other kernel sources (or another env count) is stale: reported as null, never as measured
            if int(tf.get("envs", 0)) == B and tf.get("kernel_sources_sha256") == hsh.hexdigest():
                traffic, traffic_src = float(tf["bytes_per_launch"]), tf.get("source")
        workload = ("humanoid imitation rollout (HumanoidTracking, synthetic standing clip), " if args.config == "humanoid" else
                    "rodent imitation rollout, ") + (
                    ("" if args.config == "humanoid" else "single groom clip, " if args.clips == 1 else
                     f"{args.clips} synthesised clips (random clip per env), ") +
                    f"{B} envs/GPU: " + ("random actions -> " if args.random_actions else "intention-policy forward (HIP) -> ") +
                    f"{type(base).__name__}.step (5 substeps, CG 6/6) -> " + f"auto-reset {'off' if args.no_autoreset else 'on'}" +
                    ("" if args.random_actions else " -> Transition logging (unroll 20" +
                     (", the 20 steps replayed as one captured hipGraph)" if main_m["graphed"] else ")")))
        out = {
            "metric": ("env-steps/sec (whole node), rodent imitation, num_envs=4096/GPU" if args.config == "rodent" else
                       "env-steps/sec (whole node), humanoid imitation, num_envs=1024/GPU (not the headline metric)"),
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
            "config": {"workload": workload, "envs_per_gpu": B, "clips": args.clips,
                       "parallelism": f"env-sharded x{world}, no data-path collective"},
            # The fused step kernel is bound by the FP32 vector ALU / dependent on-chip latency, not by HBM or MFMA
            # (~500 flop per HBM byte), so the roofline is priced against the fp32 VALU peak; the HBM
            # figures (what north_star asks to see) are kept beside it.
            "roofline": {
                "bound": "valu",
                "kernel": "vnl_step_kernel",
                "achieved": valu_tflops,
                "peak": VALU_PEAK_TFLOPS,
                "unit": "TFLOP/s",
                "frac": valu_tflops / VALU_PEAK_TFLOPS,
                "traffic": traffic,
                "traffic_source": traffic_src,
                "kernel_ms": kernel_ms,
                "kernel_ms_source": ("HIP events around the step-kernel launches of an eager re-run of the hot path right after "
                                     "the timed region (events inside the replayed hipGraph cannot be timed)" if main_m["graphed"]
                                     else "HIP events around the step-kernel launches of the timed region, on the launch stream"),
                "algorithmic_flops_per_launch": F_ALG * B,
                "algorithmic_flops_note": "SURVEY 8(d) tree-sparse ESTIMATE at worst-case iteration counts, not a count of executed "
                                          "flops: `achieved` and `frac` are upper bounds of the useful fp32 rate",
                "algorithmic_bytes_per_launch": B_ALG * B,
                "hbm_achieved_gbs": hbm_gbs,
                "hbm_peak_gbs": HBM_PEAK_GBS,
                "hbm_frac": hbm_gbs / HBM_PEAK_GBS,
            },
            "finite": main_m["finite"],
            "done_frac_last_step": main_m["done_frac"],
        }
        if free_m is not None:
            out["free_running"] = {
                "value": world * B * args.steps / free_m["dt"], "unit": "env-steps/s",
                "ms_per_step": free_m["dt"] / args.steps * 1e3, "kernel_ms": free_m["kernel_ms"],
                "finite": free_m["finite"], "done_frac_last_step": free_m["done_frac"],
                "note": "same hot path with the auto-reset wrapper off: envs evolve freely for warmup + steps control steps",
            }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(base, num_envs=1024)
            if not args.no_parity and args.config == "rodent" and args.clips == 1:
                try:
                    out["cpu_baseline"]["compliance"] = parity_compliance(RodentTracking, clip, H.env_kwargs(), dev)
                except Exception as e:  # the checker must not take the measurement down
                    out["cpu_baseline"]["compliance"] = {"error": repr(e)}
        else:
            out["cpu_baseline"] = None
    # ---- full training step ("report rollout-only and full-train-step numbers separately") -------------
    train_rec = None
    do_train = (not args.no_train_step and args.config == "rodent" and not args.random_actions and args.clips == 1 and
                (world == 1 or args.train_step_multi))
    if do_train:
        torch.cuda.empty_cache()
        try:
            train_rec = train_step_record(clip, H.env_kwargs(), dev, B, world, args.train_steps)
            if rank == 0 and world == 1:
                train_rec.update(update_record(dev))
            if world == 1 and not args.no_train_secondary:
                # SURVEY 8(d) config 5 "secondary": the reference's own proportions, batch_size x num_minibatches = 8 x num_envs
                torch.cuda.empty_cache()
                try:
                    train_rec["reference_proportions"] = train_step_record(clip, H.env_kwargs(), dev, B, world, 1, ratio=8)
                except Exception as e:
                    train_rec["reference_proportions"] = {"error": repr(e)}
        except Exception as e:
            if world > 1:
                raise
            train_rec = {"error": repr(e)}
    if rank == 0:
        out["train_step"] = train_rec
        print(json.dumps(out), flush=True)
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
