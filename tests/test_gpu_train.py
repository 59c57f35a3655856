"""PPO training step on the GPU: the hipGraph-captured minibatch step must reproduce the eager one."""
import functools

import numpy as np
import pytest
import torch

import helpers as H
from vnl_brax_imitation_amd.envs.rodent import RodentTracking
from vnl_brax_imitation_amd.ppo_imitation import ppo_networks
from vnl_brax_imitation_amd.ppo_imitation import train as ppo

pytestmark = pytest.mark.gpu


def _run(capture: bool, backend: str = "auto", steps: int = 3):
    dev = torch.device("cuda:0")
    env = RodentTracking(H.reference_clip(), num_envs=64, device=dev, **H.env_kwargs())
    nf = functools.partial(ppo_networks.make_intention_ppo_networks, intention_latent_size=60,
                           encoder_layer_sizes=(128, 128), decoder_layer_sizes=(128, 128))
    log = []
    _, (norm, flat), _ = ppo.train(
        environment=env, num_timesteps=steps * 64 * 5, episode_length=150, num_envs=64, learning_rate=1e-3,
        entropy_cost=1e-2, discounting=0.95, unroll_length=5, batch_size=16, num_minibatches=4,
        num_updates_per_batch=2, num_evals=1, normalize_observations=True, network_factory=nf, num_eval_envs=0,
        eval_env=None, seed=3, capture_graph=capture, update_backend=backend, progress_fn=lambda s, m: log.append(m))
    return norm, flat, log[-1], ppo.train.last_training_state


def test_graph_capture_matches_eager():
    n0, f0, m0, ts0 = _run(False)
    n1, f1, m1, ts1 = _run(True)
    assert int(ts0.optimizer_state["count"]) == int(ts1.optimizer_state["count"]) == 3 * 2 * 4
    assert torch.equal(n0.mean, n1.mean) and torch.equal(n0.std, n1.std)
    # same kernels on the same inputs; the only difference is how they are launched
    assert torch.allclose(f0, f1, rtol=0, atol=1e-6), float((f0 - f1).abs().max())
    assert torch.allclose(ts0.params, ts1.params, rtol=0, atol=1e-6)
    for k in ("training/total_loss", "training/v_loss", "training/policy_loss", "training/kl_loss_intention"):
        assert abs(m0[k] - m1[k]) <= 1e-5 * max(1.0, abs(m0[k])), (k, m0[k], m1[k])
    assert torch.isfinite(f1).all()


def test_hand_written_update_trains_like_the_autograd_update():
    """The same short training run with the hand-written forward + backward (vnl_ppo_minibatch_grad) and with torch
    autograd through the op-by-op loss: ONE training step (the same rollout: later ones would differ, the policies no
    longer being bit-identical), 8 Adam steps on it; parameters and losses must agree."""
    n0, f0, m0, ts0 = _run(True, "hip", steps=1)
    n1, f1, m1, ts1 = _run(True, "torch", steps=1)
    assert torch.equal(n0.mean, n1.mean)
    d = float((ts0.params - ts1.params).abs().max())
    print(f"\n[hip vs torch update, 8 Adam steps] max |param difference| {d:.2e} (lr 1e-3)")
    # Adam turns a 1e-6 relative gradient difference into at most ~lr of parameter difference per step (8e-3 over these 8);
    # measured 2e-5 .. 5.5e-5 depending on the rollout the env kernels of the day produce
    assert d < 1e-4, d
    for k in ("training/total_loss", "training/v_loss", "training/policy_loss", "training/kl_loss_intention"):
        assert abs(m0[k] - m1[k]) <= 1e-4 * max(1.0, abs(m0[k])), (k, m0[k], m1[k])


def test_training_with_evaluation_on_gpu():
    """Full train() contract on the device: evaluator (EvalWrapper around the training wrappers, deterministic HIP
    policy), progress / checkpoint callbacks, resumable state."""
    dev = torch.device("cuda:0")
    env = RodentTracking(H.reference_clip(), num_envs=64, device=dev, **H.env_kwargs())
    eval_env = RodentTracking(H.reference_clip(), num_envs=32, device=dev, **H.env_kwargs())
    nf = functools.partial(ppo_networks.make_intention_ppo_networks, intention_latent_size=60,
                           encoder_layer_sizes=(128, 128), decoder_layer_sizes=(128, 128))
    log, saved = [], []
    make_policy, params, metrics = ppo.train(
        environment=env, num_timesteps=2 * 64 * 5, episode_length=20, num_envs=64, learning_rate=1e-3,
        entropy_cost=1e-2, discounting=0.95, unroll_length=5, batch_size=16, num_minibatches=4,
        num_updates_per_batch=2, num_evals=3, normalize_observations=True, network_factory=nf, num_eval_envs=32,
        eval_env=eval_env, seed=1, progress_fn=lambda s, m: log.append((s, m)),
        policy_params_fn=lambda s, mk, p: saved.append(s))
    assert [s for s, _ in log] == [0, 320, 640] and saved == [320, 640]
    for k in ("eval/episode_reward", "eval/avg_episode_length", "eval/sps", "training/sps", "training/total_loss"):
        assert k in metrics and torch.isfinite(torch.tensor(float(metrics[k]))), k
    assert 0 < metrics["eval/avg_episode_length"] <= 20
    act, extras = make_policy(params, deterministic=True)(torch.zeros(4, 795, device=dev), torch.zeros(4, 232, device=dev), None)
    assert act.shape == (4, 30) and extras == {} and torch.isfinite(act).all()


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs (the round's GPU box has one)")
def test_two_rank_rccl_training_step():
    """train() over RCCL with two ranks, one per GPU (reference train.py:485-487 asserts identical replicas; so does
    ours at the end of train()): gradient all-reduce between graph replays, normaliser statistics over the global batch."""
    import json
    import os
    import subprocess
    import sys

    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29731", os.path.join(H.ROOT, "tools", "train_bench.py"), "--steps", "1", "--envs-per-gpu", "256",
           "--updates", "2"]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
    assert p.returncode == 0, p.stderr[-2000:]
    line = [ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1]
    r = json.loads(line)
    assert r["n_gpus"] == 2 and np.isfinite(r["total_loss"]) and r["env_steps"] == 2 * 256 * 20


def test_hand_written_gradient_stays_exact_along_training():
    """tests/test_gpu_ppo_update.py checks the minibatch step at initialisation on synthetic data.  Here: after two training
    epochs with it (parameters, normaliser and the policy's standard deviations have moved), on a fresh rollout of the
    trained policy, vnl_ppo_minibatch_grad against float64 autograd through the op-by-op loss: every gradient tensor
    within 1e-4 of its largest entry, the loss terms within 1e-5.  (Two training RUNS, hand-written vs autograd update,
    drift apart after the first epoch -- Adam turns rounding-level gradient differences into sign flips on near-zero
    coordinates -- so runs cannot be compared; gradients on identical inputs can.  tools/compare_backends.py shows the
    former, tools/grad_check_trained.py the latter at the reference's sizes.)"""
    import os
    import sys

    sys.path.insert(0, os.path.join(H.ROOT, "tools"))
    import grad_check_trained as G

    r = G.check(envs=512, epochs=2, updates=4)
    print("\n[gradient along training] worst", r["worst tensor"], f"{r['worst rel err']:.2e}", "median", f"{r['median rel err']:.2e}")
    assert r["worst rel err"] < 1e-4, (r["worst tensor"], r["worst rel err"])
    for a, b in zip(r["losses hip"][:5], r["losses f64"][:5]):
        assert abs(a - b) <= 1e-5 * max(abs(b), abs(r["losses f64"][0])), (r["losses hip"], r["losses f64"])


def test_full_size_training_step_on_one_gpu():
    """BASELINE configs[4]'s per-GPU share: 4096 envs, unroll 20, reference network sizes, 32 minibatches of 128 trajectories
    (2560 samples); two updates per batch instead of sixteen to keep the test short.  The hand-written update inside the
    captured graph, finite metrics, and the step count the reference's bookkeeping gives."""
    import os
    import subprocess
    import sys
    import json

    cmd = [sys.executable, os.path.join(H.ROOT, "tools", "train_bench.py"), "--steps", "2", "--updates", "2", "--backend", "hip"]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    r = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    assert r["env_steps"] == 2 * 4096 * 20 and np.isfinite(r["total_loss"]) and np.isfinite(r["v_loss"]) and r["training/sps"] > 0
    print(f"\n[full-size training step] {r['training/sps']:.0f} env-steps/s with 2 updates per batch (graph capture included)")


def test_split_step_with_a_one_rank_process_group_matches_the_single_gpu_step():
    """The data-parallel form of the minibatch step on the hardware that is there: a ONE-rank RCCL process group makes the
    trainer take the split route -- graph 1 (gather, forward, head, value backward), all-reduce of the value segment on the
    communication stream, graph 2 (policy backward) beside it, all-reduce of the policy segment, Adam per segment -- whose
    result must agree with the unsplit single-GPU step (same seeds; the value network's output-layer gradient is summed in
    another launch, so agreement is at rounding level, not bitwise)."""
    import json
    import os
    import subprocess
    import sys

    out = []
    for extra in ([], ["--force-dist"]):
        cmd = [sys.executable, os.path.join(H.ROOT, "tools", "train_bench.py"), "--steps", "2", "--updates", "2", "--backend", "hip",
               "--envs-per-gpu", "512"] + extra
        p = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=dict(os.environ, MASTER_PORT="29713"))
        assert p.returncode == 0, p.stderr[-2000:]
        out.append(json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1]))
    a, b = out
    print(f"\n[split step, one-rank RCCL group] total_loss {b['total_loss']:.6f} vs unsplit {a['total_loss']:.6f}; "
          f"{b['training/sps']:.0f} vs {a['training/sps']:.0f} env-steps/s")
    assert a["env_steps"] == b["env_steps"] and np.isfinite(b["total_loss"])
    assert abs(a["total_loss"] - b["total_loss"]) <= 2e-3 * max(abs(a["total_loss"]), 1e-3), (a["total_loss"], b["total_loss"])
    assert abs(a["v_loss"] - b["v_loss"]) <= 2e-3 * max(abs(a["v_loss"]), 1e-3)
