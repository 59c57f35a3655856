#!/usr/bin/env python3
"""Turn gpurun_out/pmc_stage/all.txt (tools/pmc_stage.sh) into per-call, per-wave costs of each stage."""
import re
import sys

NAMES = {1: "kin+inertia+M+factor+invert", 2: "kinematics", 3: "kin+inertia+M", 4: "kin+inertia+M+factor",
         5: "kin+inertia+bias", 6: "twists+Jv", 7: "solve (M^-1 x)", 8: "ls row pass (3 alphas)",
         9: "make_constraint", 10: "constraint_force", 11: "smooth_forces", 12: "4 x vdot", 13: "ls_load + 1-alpha pass", 14: "kin+inertia+M+factor(load/store only)", 15: "kin+inertia+M+factor_pair", 16: "euler: reload+invert+apply"}
st, data = None, {}
for line in open(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/pmc_stage/all.txt"):
    m = re.match(r"stage (\d+)", line)
    if m:
        st = int(m.group(1))
        data[st] = {}
        continue
    m = re.search(r"(SQ_\w+)\s+calls=\s*\d+ avg=([\d.e+]+)", line)
    if m:
        data[st][m.group(1)] = float(m.group(2))
W, SUB, REP = 4096, 5, 4
b = data[0]
print("base per wave per substep: " + "  ".join(f"{k[3:]}={v / W / SUB:.0f}" for k, v in sorted(b.items())))
for st in sorted(NAMES):
    if st in data:
        d = {k: (data[st][k] - b[k]) / (W * SUB * REP) for k in b}
        print(f"stage {st} {NAMES[st]:30s} " + "  ".join(f"{k[3:]}={v:.0f}" for k, v in sorted(d.items())))
