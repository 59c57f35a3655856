#!/usr/bin/env python3
"""Per-stage share of the step kernel's wave cycles (diagnostic build libvnl_prof.so, GPU only).
Usage (on the GPU box): python tools/stage_profile.py [num_envs] [steps]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

import helpers as H  # noqa: E402
from vnl_brax_imitation_amd import _lib  # noqa: E402
from vnl_brax_imitation_amd.csrc import build as hb  # noqa: E402
from vnl_brax_imitation_amd.envs.rodent import RodentTracking  # noqa: E402

NAMES = ["kinematics", "crb+M", "factor(M)", "bias(rne)", "smooth+solveM", "collision+rows", "solver init",
         "ls: twists+Jv+Mv", "ls: row passes", "update: J'f+cost", "solveM(grad)", "euler pre", "factor(M+hB)",
         "euler solve+integrate", "env glue", "-"]


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    lib = _lib.load_library(hb.build(profile=True))
    env = RodentTracking(H.reference_clip(), num_envs=B, device="cuda:0", _library=lib, **H.env_kwargs())
    st = env.reset(0)
    g = torch.Generator().manual_seed(0)
    buf = (C.c_ulonglong * 16)()
    env.step(st, torch.zeros(B, 30))
    torch.cuda.synchronize()
    lib.vnl_prof_read(buf)
    t0 = torch.cuda.Event(enable_timing=True)
    t1 = torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(steps):
        a = torch.clamp(0.3 * torch.randn((B, 30), generator=g), -1, 1)
        env.step(st, a)
    t1.record()
    torch.cuda.synchronize()
    lib.vnl_prof_read(buf)
    tot = float(sum(buf))
    print(f"B={B} steps={steps} ms/step={t0.elapsed_time(t1) / steps:.2f} (diagnostic build; read shares only)")
    for n, v in zip(NAMES, buf):
        print(f"  {n:24s} {100.0 * v / tot:6.2f} %")


if __name__ == "__main__":
    main()
