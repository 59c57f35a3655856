#!/usr/bin/env python3
"""Histogram of the number of LIVE constraint rows (D != 0) per env after a few control steps of the benchmark's rollout
(which line-search variant an env takes depends on it: <= 64 rows -> one row per lane).  GPU only."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import helpers as H  # noqa: E402
from vnl_brax_imitation_amd.envs.rodent import RodentTracking  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    env = RodentTracking(H.reference_clip(), num_envs=B, device="cuda:0", **H.env_kwargs())
    env.debug(2)
    st = env.reset(0)
    g = torch.Generator().manual_seed(0)
    ncon = int(env.sys.scalars["ncon"])
    for k in range(int(sys.argv[2]) if len(sys.argv) > 2 else 6):
        a = torch.clamp(0.3 * torch.randn((B, 30), generator=g), -1, 1)
        st = env.step(st, a)
        raw = np.ascontiguousarray(env.scratch("act_list").cpu().numpy()).view(np.int32)
        nl = raw[:, (ncon + 3) // 4 + 1]
        na = raw[:, (ncon + 3) // 4]
        print(f"step {k}: live rows min {nl.min()} median {int(np.median(nl))} mean {nl.mean():.1f} max {nl.max()}  "
              f"> 64: {(nl > 64).mean() * 100:.1f} %  > 128: {(nl > 128).mean() * 100:.1f} %   active contacts median {int(np.median(na))} max {na.max()}")


if __name__ == "__main__":
    main()
