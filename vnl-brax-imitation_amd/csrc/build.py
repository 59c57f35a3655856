"""Build libvnl.so for gfx950 with hipcc (in-tree, next to the sources)."""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SOURCES = ["vnl_lib.hip", "vnl_policy.hip"]
DEPS = ["vnl_lib.hip", "vnl_policy.hip", "vnl_body.h", "vnl_types.h", "../../include/vnl.h"]
OUT = os.path.join(HERE, "libvnl.so")


def build(force: bool = False, verbose: bool = False, profile: bool = False) -> str:
    """profile=True builds the DIAGNOSTIC library libvnl_prof.so (per-stage stamps, -DVNL_PROFILE);
    it is only ever loaded by tools/stage_profile.py."""
    out = os.path.join(HERE, "libvnl_prof.so") if profile else OUT
    newest = max(os.path.getmtime(os.path.join(HERE, d)) for d in DEPS)
    if not force and os.path.exists(out) and os.path.getmtime(out) >= newest:
        return out
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-fno-slp-vectorize", "-std=c++17", "-shared", "-fPIC", "-o", out] + \
          (["-DVNL_PROFILE"] if profile else []) + [os.path.join(HERE, s) for s in SOURCES]
    if verbose:
        cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    return out


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True, profile="--profile" in sys.argv))
