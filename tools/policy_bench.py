#!/usr/bin/env python3
"""Time the intention-policy forward at B=4096: fused HIP kernel vs torch (hipBLASLt) path. GPU only."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import vnl_brax_imitation_amd  # noqa: F401,E402
from vnl_brax_imitation_amd.ppo_imitation import ppo_networks, running_statistics  # noqa: E402

dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
n = ppo_networks.make_intention_ppo_networks(795, 232, 30, preprocess_observations_fn=running_statistics.normalize,
                                             intention_latent_size=64, encoder_layer_sizes=(256, 128),
                                             decoder_layer_sizes=(128, 256))
flat = n.policy_network.init(torch.Generator().manual_seed(0)).to(dev)
st = running_statistics.init_state(232, device=dev)
mk = ppo_networks.make_inference_fn(n)
traj, obs = torch.randn((B, 795), device=dev) * 0.1, torch.randn((B, 232), device=dev)
g = torch.Generator(device=dev).manual_seed(0)
only = sys.argv[2] if len(sys.argv) > 2 else None  # "hip" / "torch": time one backend only (rocprofv3 runs)
for name, pol in (("hip", mk((st, flat))), ("torch", mk((st, flat), backend="torch"))):
    if only and name != only:
        continue
    for _ in range(5):
        pol(traj, obs, g)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(50):
        pol(traj, obs, g)
    b.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 50
    print(f"{name:6s} B={B}: {ms * 1e3:8.1f} us per forward (incl. noise generation), {B * 0.68e6 / (ms * 1e-3) / 1e12:.2f} TFLOP/s")
