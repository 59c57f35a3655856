#!/usr/bin/env python3
"""Phase times of the policy kernel's workgroup 0 (prof build: wall-clock stamps at the phase boundaries). GPU only."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["VNL_LIB"] = os.path.join(ROOT, "vnl-brax-imitation_amd", "csrc", "libvnl_prof.so")
import torch  # noqa: E402

import vnl_brax_imitation_amd  # noqa: F401,E402
from vnl_brax_imitation_amd import _lib  # noqa: E402
from vnl_brax_imitation_amd.ppo_imitation import ppo_networks, running_statistics  # noqa: E402

dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
n = ppo_networks.make_intention_ppo_networks(795, 232, 30, preprocess_observations_fn=running_statistics.normalize,
                                             intention_latent_size=64, encoder_layer_sizes=(256, 128),
                                             decoder_layer_sizes=(128, 256))
flat = n.policy_network.init(torch.Generator().manual_seed(0)).to(dev)
st = running_statistics.init_state(232, device=dev)
pol = ppo_networks.make_inference_fn(n)((st, flat))
traj, obs = torch.randn((B, 795), device=dev) * 0.1, torch.randn((B, 232), device=dev)
g = torch.Generator(device=dev).manual_seed(0)
for _ in range(5):
    pol(traj, obs, g)
torch.cuda.synchronize()
lib = _lib.load_library()
out = (ctypes.c_longlong * 32)()
f = lib.vnl_policy_profile_stamps
f.argtypes, f.restype = [ctypes.POINTER(ctypes.c_longlong)], ctypes.c_int
assert f(out) == 0
names = {0: "start", 1: "traj tile", 2: "enc0 dense", 3: "enc0 LN", 4: "enc1 dense", 5: "enc1 LN", 10: "heads", 11: "z + obs tile",
         12: "dec0 dense", 13: "dec0 LN", 14: "dec1 dense", 15: "dec1 LN", 16: "dec2 dense", 20: "pre-dist", 21: "distribution"}
prev = out[0]
for i in sorted(names):
    if out[i]:
        print(f"{names[i]:14s} +{(out[i] - prev) * 10:6d} ns   (t = {(out[i] - out[0]) * 10} ns)")
        prev = out[i]
