"""Build libvnl.so for gfx950 with hipcc (in-tree, next to the sources)."""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SOURCES = ["vnl_lib.hip"]
DEPS = ["vnl_lib.hip", "vnl_body.h", "vnl_types.h", "vnl_policy.h", "vnl_policy_impl.h", "../../include/vnl.h"]
OUT = os.path.join(HERE, "libvnl.so")


def build(force: bool = False, verbose: bool = False) -> str:
    newest = max(os.path.getmtime(os.path.join(HERE, d)) for d in DEPS)
    if not force and os.path.exists(OUT) and os.path.getmtime(OUT) >= newest:
        return OUT
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-o", OUT] + \
          [os.path.join(HERE, s) for s in SOURCES]
    if verbose:
        cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
