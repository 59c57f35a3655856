"""Build libvnl.so for gfx950 with hipcc (in-tree, next to the sources)."""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SOURCES = ["vnl_lib.hip", "vnl_policy.hip"]
DEPS = ["vnl_lib.hip", "vnl_policy.hip", "vnl_body.h", "vnl_types.h", "../../include/vnl.h"]
OUT = os.path.join(HERE, "libvnl.so")


def build(force: bool = False, verbose: bool = False, profile: bool = False, knobs: bool = False) -> str:
    """profile=True builds the DIAGNOSTIC library libvnl_prof.so (per-stage stamps, -DVNL_PROFILE), only ever
    loaded by tools/stage_profile.py; knobs=True builds libvnl_knobs.so (-DVNL_STAGE_KNOBS: the VNL_DBG_REPEAT
    stage-repeat knob of tools/stage_cost.py / tools/pmc_stage.sh).  Neither is the product library."""
    out = os.path.join(HERE, "libvnl_prof.so") if profile else (os.path.join(HERE, "libvnl_knobs.so") if knobs else OUT)
    newest = max(os.path.getmtime(os.path.join(HERE, d)) for d in DEPS)
    if not force and os.path.exists(out) and os.path.getmtime(out) >= newest:
        return out
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-fno-slp-vectorize", "-std=c++17", "-shared", "-fPIC", "-o", out] + \
          (["-DVNL_PROFILE"] if profile else []) + (["-DVNL_STAGE_KNOBS"] if knobs else []) + \
          os.environ.get("VNL_HIPCC_EXTRA", "").split() + [os.path.join(HERE, s) for s in SOURCES]
    cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    r = subprocess.run(cmd, stderr=subprocess.PIPE, text=True)
    if verbose or r.returncode:
        sys.stderr.write(r.stderr)
    if r.returncode:
        raise subprocess.CalledProcessError(r.returncode, cmd)
    # The env kernels must keep two waves per SIMD (8 workgroups per CU) and no private-memory stack: a
    # build that slipped to 256 VGPRs + AGPR/scratch spills produced wrong results on the GPU (r01).
    name, bad = None, []
    for line in r.stderr.splitlines():
        if "Function Name:" in line:
            name = line.split("Function Name:")[1].split()[0]
        elif name and ("vnl_step_kernel" in name or "vnl_reset_kernel" in name) and not profile and not knobs:
            if "ScratchSize" in line and int(line.split("]:")[1].split()[0]) != 0:
                bad.append(f"{name}: {line.split('remark:')[1].strip()}")
            if "Occupancy [waves/SIMD]" in line and int(line.split("]:")[1].split()[0]) < 2:
                bad.append(f"{name}: {line.split('remark:')[1].strip()}")
    if bad:
        os.remove(out)
        raise RuntimeError("register budget exceeded, refusing to ship this build:\n  " + "\n  ".join(bad))
    return out


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True, profile="--profile" in sys.argv, knobs="--knobs" in sys.argv))
