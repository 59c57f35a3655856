#!/usr/bin/env python3
"""Time the hand-written PPO minibatch step (vnl_ppo_minibatch_grad) at the reference's sizes: T=20, b=128 trajectories,
intention net 795 -> 256 -> 128 -> 64|64, [64 | 232] -> 128 -> 256 -> 60, value MLP 232 -> 1024 -> 1024 -> 1.
Usage (GPU box): python tools/ppo_update_bench.py [--iters N] [--torch]   (under rocprofv3 --kernel-trace --stats for the
per-kernel split).  FLOP count: 2 x MACs of every GEMM, forward + dX + dW (9.75 MFLOP per sample)."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

import vnl_brax_imitation_amd  # noqa: E402,F401
from test_gpu_ppo_update import HP, _make  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--torch", action="store_true", help="time the autograd path instead (hipBLASLt)")
    ap.add_argument("--tile", type=int, default=0, help="force the GEMM tile of the hand-written path (64 / 128)")
    ap.add_argument("--wg-target", type=int, default=0, help="workgroups a split-K weight gradient is split up to (default 256)")
    ap.add_argument("--fwd-mode", type=int, default=-1, help="the intention network's forward: 0 layer by layer, 1 one fused launch (csrc/vnl_policy.hip, training form), 2 first Dense as a GEMM + the rest fused (default)")
    ap.add_argument("--fused-threads", type=int, default=0, help="threads of the fused part of the forward (256 / 512 / 1024)")
    ap.add_argument("--trajectories", type=int, default=128, help="trajectories per minibatch (128: batch_size x num_minibatches = num_envs; 1024: the reference's proportions, 8 x num_envs)")
    ap.add_argument("--no-wgrad-transpose", action="store_true", help="large minibatches: weight gradients in the transposed-operand form (no [X | 1]' arena)")
    ap.add_argument("--noprio", action="store_true", help="no raised wave priority for the intention network's GEMMs")
    a = ap.parse_args()
    from vnl_brax_imitation_amd.ppo_imitation import hip_update, intention_losses, running_statistics
    from vnl_brax_imitation_amd.ppo_imitation.intention_policy_network import LeafParams

    cfg = dict(traj=795, obs=232, act=30, latent=64, enc=(256, 128), dec=(128, 256), val=(1024, 1024), T=20, B=a.trajectories)
    nets, flat, data, norm, noise = _make(**cfg)
    dev = torch.device("cuda:0")
    to = lambda t: t.to(dev)  # noqa: E731
    flat, data, noise = to(flat).contiguous(), data.map(to), {k: to(v) for k, v in noise.items()}
    ndev = running_statistics.RunningStatisticsState(to(norm.count), to(norm.mean), to(norm.summed_variance), to(norm.std))
    grads = torch.zeros_like(flat)
    n_pol = nets.policy_network.layout.size
    if a.torch:
        lp = intention_losses.PPONetworkParams(policy=LeafParams(nets.policy_network.layout, flat[:n_pol], grads[:n_pol]),
                                               value=LeafParams(nets.value_network.layout, flat[n_pol:], grads[n_pol:]))

        def step():
            lp.policy.zero_grad(), lp.value.zero_grad()
            loss, _ = intention_losses.compute_ppo_intention_loss(lp, ndev, data, None, ppo_network=nets, noise=noise,
                                                                  time_major=True, **HP)
            loss.backward()
            lp.policy.gather_grads(), lp.value.gather_grads()
    else:
        upd = hip_update.HipPPOUpdate(nets, cfg["T"], cfg["B"], dev, **HP)
        if a.tile or a.wg_target:
            assert upd.lib.vnl_ppo_update_tune(upd.h, a.tile, a.wg_target) == 0
        if a.fwd_mode >= 0:
            assert upd.lib.vnl_ppo_update_tune(upd.h, -10 - a.fwd_mode, 0) == 0
        if a.fused_threads:
            assert upd.lib.vnl_ppo_update_tune(upd.h, -a.fused_threads, 0) == 0
        if a.no_wgrad_transpose:
            assert upd.lib.vnl_ppo_update_tune(upd.h, -6, 0) == 0
        if a.noprio:
            assert upd.lib.vnl_ppo_update_tune(upd.h, -2, 0) == 0

        def step():
            upd.grad(flat, ndev, data, noise, grads)
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    # replay as a graph (what the trainer does): launch overhead out of the picture
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        step()
    torch.cuda.current_stream().wait_stream(s)
    with torch.cuda.graph(g):
        step()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.iters):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / a.iters
    N = cfg["T"] * cfg["B"]
    macs_p = 795 * 256 + 256 * 128 + 2 * 128 * 64 + 296 * 128 + 128 * 256 + 256 * 60
    macs_v = 232 * 1024 + 1024 * 1024 + 1024
    flop = 2.0 * 3.0 * (N * macs_p + N * macs_v) + 2.0 * cfg["B"] * macs_v  # fwd + dX + dW (+ the bootstrap rows' forward)
    print(json.dumps({"backend": "torch" if a.torch else "hip", "tile": a.tile, "wg_target": a.wg_target, "ms_per_minibatch_step": ms, "tflops": flop / ms / 1e9,
                      "gflop_per_step": flop / 1e9, "samples": N}))


if __name__ == "__main__":
    main()
