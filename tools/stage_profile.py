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

NAMES = ["kinematics", "body inertias (x2)", "bias: velocity prefix", "bias: acceleration prefix", "bias: cfrc + subtree sums + project",
         "M: crb subtree sums (x2)", "M: f_i + entries (x2)", "factor: load rows (x2)", "factor: steps (x2)", "factor: store (x2)",
         "M*warm (1st only) [after factor]", "inversion (x2)", "smooth forces", "constraint rows (collision + limits)",
         "solve: warm-start choice (2 J*v + costs)", "solve: J'f + first M^-1 grad", "iter: convergence test + |search|",
         "iter: J*search", "iter: qg sums", "iter: line search", "iter: qacc / Ma / Jaref update", "iter: gauss terms",
         "iter: J'f + cost", "iter: gradient", "iter: M^-1 grad", "iter: beta + search update", "solve tail / loop exit",
         "euler: M^-1 rhs", "euler: integrate", "step: tables + state load + rtrunk", "step: reward / obs / traj / store"]
NAMES += ["-"] * (40 - len(NAMES))


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    lib = _lib.load_library(hb.build(profile=True))
    with H.backend(lib):
        env = RodentTracking(H.reference_clip(), num_envs=B, device="cuda:0", **H.env_kwargs())
    st = env.reset(0)
    g = torch.Generator().manual_seed(0)
    buf = (C.c_ulonglong * 40)()
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
        if v:
            print(f"  {n:44s} {100.0 * v / tot:6.2f} %")


if __name__ == "__main__":
    main()
