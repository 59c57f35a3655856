#!/usr/bin/env python3
"""Parity sweep over seeds (test infrastructure, run by hand on the GPU box: `python tests/parity_sweep.py [--seeds 8]`): the
checks of tests/test_stage_parity.py / test_gpu_parity.py -- one control step and single resynchronised substeps at 4096
envs, product on the device vs the oracles following its solver decisions -- for several input seeds, with the per-seed
statistics printed.  Any violation raises, exactly as in the tests."""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import helpers as H  # noqa: E402
import parity as P  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", type=int, default=8)
    ap.add_argument("--envs", type=int, default=4096)
    ap.add_argument("--first", type=int, default=100, help="first seed")
    a = ap.parse_args()
    from vnl_brax_imitation_amd.envs.rodent import RodentTracking

    B = a.envs
    env = RodentTracking(H.reference_clip(), num_envs=B, device="cuda:0", **H.env_kwargs())
    env1 = RodentTracking(H.reference_clip(), num_envs=B, device="cuda:0", **{**H.env_kwargs(), "n_frames": 1})
    o64, o32 = H.make_oracle(env, "f64"), H.make_oracle(env, "f32")
    o64_1, o32_1 = H.make_oracle(env1, "f64"), H.make_oracle(env1, "f32")
    for seed in range(a.first, a.first + a.seeds):
        rng = np.random.default_rng(seed)
        sf = rng.integers(0, 235, B).astype(np.int32)
        noise = (1e-3 * rng.standard_normal((B, 74))).astype(np.float32)
        act = np.clip(0.3 * rng.standard_normal((B, 30)), -1, 1).astype(np.float32)
        st, err, dev, rep, ost = P.control_step_follow(env, o64, o32, sf, noise, act)
        flipped = P.check_control_step(err, dev, rep, verbose=False)
        done_eq = bool(np.array_equal(st.done.cpu().numpy(), ost["done"].astype(np.float32)))
        sub = []
        for k, (e1, d1, r1) in enumerate(P.resync_substeps(env1, o64_1, o32_1, sf, noise, act, nsub=3)):
            viol = sum(len(v) for v in P.bound_violations(e1, d1).values())
            r = r1.reshape(len(r1), -1, r1.shape[-1])[:, 0]
            sub.append({"substep": k, "qvel max": float(e1["qvel"].max()), "violations": int(viol),
                        "non-tie solves (natural-oracle reference)": int(P.non_tie(r).sum())})
            assert viol == 0
        print(json.dumps({"seed": seed, "envs": B, "control step": {k: [float(np.median(v)), float(v.max())] for k, v in err.items()
                                                                     if k in ("qpos", "qvel", "qacc_warmstart", "xpos")},
                          "envs with a flipped later decision": int(flipped), "done equal": done_eq, "single substeps": sub}), flush=True)
        assert done_eq


if __name__ == "__main__":
    main()
