#!/usr/bin/env python3
"""Price the stages of the step kernel in the REAL build: rerun the bench loop with
VNL_DBG_REPEAT=stage:count and report the extra kernel time per repetition (GPU only)."""
import os
import subprocess
import sys
import json

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NAMES = {1: "kin+inertia+M+factor+invert", 2: "kinematics", 3: "kin+inertia+M", 4: "kin+inertia+M+factor",
         5: "kin+inertia+bias", 6: "twists+Jv", 7: "solve (M^-1 x)", 8: "ls row pass (3 alphas)",
         18: "subtree sums (10 wide)", 19: "body inertias"}
if os.environ.get("STAGES"):
    NAMES = {int(k): NAMES.get(int(k), "?") for k in os.environ["STAGES"].split()}


def run(stage, count):
    env = dict(os.environ, VNL_LIB=os.path.join(ROOT, "vnl-brax-imitation_amd", "csrc", "libvnl_knobs.so"))  # build.py --knobs
    if count:
        env["VNL_DBG_REPEAT"] = f"{stage}:{count}"
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "10", "--warmup", "2",
                          "--no-cpu-baseline", "--no-autoreset", "--random-actions"], env=env, capture_output=True, text=True)
    if out.returncode != 0 or not out.stdout.strip():
        sys.exit("bench.py failed:\n" + out.stderr[-3000:])
    out = out.stdout
    return json.loads(out.strip().splitlines()[-1])["roofline"]["kernel_ms"]


base = run(0, 0)
import torch
print(f"base kernel_ms {base:.3f}  (5 substeps; per-substep extra cost below)")
for st, name in NAMES.items():
    n = 4
    t = run(st, n)
    print(f"  stage {st} {name:32s} {1e3 * (t - base) / (5 * n):8.1f} us per call per substep-slot  ({t:.2f} ms)")
