#!/usr/bin/env python3
"""Builds tests/golden/parity_cases.npz: the envs on which the parity method of tests/parity.py was refined (TEST
INFRASTRUCTURE; the fixture is data -- inputs, the device's recorded solver decisions and, where kept, its outputs).

Two sources, both product runs on an MI355X:
  * the dumps the failing sweeps left behind (`--from-dumps`, default: gpurun_out/parity_fail_substep{1,2,3}_B4096.npz,
    gpurun_out/control_step_hip_B4096.npz): every solve whose followed decision was not a tie for the oracle, the worst
    envs of the control step, plus a few ordinary envs as controls;
  * fresh runs of given seeds on the device (`--seeds 122 207`, needs a GPU): the same selection, with the device's
    post-step state recorded.

A case = the state BEFORE one product step (n_frames = 1 or 5), the action, the device's solver trace, the follow
report the oracle gave for it and (when recorded) the device's post-step qpos / qvel / qacc_warmstart.
tests/test_parity_cases.py replays every case on the CPU tier."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "golden", "parity_cases.npz")
BEFORE = ("qpos", "qvel", "act", "qacc_warmstart", "xpos", "xmat1", "com1", "qfrc_actuator", "cur_frame", "sub_clip_frame")
AFTER = ("qpos", "qvel", "qacc_warmstart")
TRACE_INTS = 536


def _tie(r):
    return np.maximum.reduce([r[..., 0], r[..., 1], r[..., 2]])


def select(report, err_qvel=None, n_worst=8, n_control=4):
    """env indices worth keeping: a followed decision that was not a tie, an active-set mismatch, the largest trial-point
    gaps, the worst envs by error, and a few ordinary ones."""
    r = report.reshape(len(report), -1, report.shape[-1])
    odd = ((_tie(r) > 1.0) | (r[..., 3] > 0)).any(axis=1)
    keep = set(np.where(odd)[0].tolist())
    keep |= set(np.argsort(-r[..., 8].max(axis=1))[:3].tolist())  # trial step lengths furthest apart
    keep |= set(np.argsort(-_tie(r).max(axis=1))[:3].tolist())
    if err_qvel is not None:
        keep |= set(np.argsort(-err_qvel)[:n_worst].tolist())
    keep |= set(range(0, len(report), max(len(report) // n_control, 1)))  # controls
    return sorted(keep)


class Cases:
    def __init__(self):
        self.rows = []

    def add(self, source, n_frames, env, z_before, action, trace, report, after=None):
        row = {"source": source, "n_frames": n_frames, "env": env, "action": np.asarray(action, np.float32)}
        for k in BEFORE:
            row["before_" + k] = np.asarray(z_before[k])
        tr = np.zeros((5, TRACE_INTS), np.int32)
        tr[:n_frames] = np.asarray(trace).reshape(n_frames, -1)
        rp = np.zeros((5, 12))
        rp[:n_frames] = np.asarray(report).reshape(n_frames, -1)
        row["trace"], row["report"] = tr, rp
        for k in AFTER:
            row["after_" + k] = np.full_like(row["before_" + k], np.nan, dtype=np.float32) if after is None else np.asarray(after[k], np.float32)
        self.rows.append(row)

    def save(self, path, merge=True):
        rows = self.rows
        if merge and os.path.exists(path):
            z = np.load(path, allow_pickle=False)
            old = [{k: z[k][i] for k in z.files} for i in range(len(z["env"]))]
            have = {(str(r["source"]), int(r["env"])) for r in rows}
            rows = [r for r in old if (str(r["source"]), int(r["env"])) not in have] + rows
        keys = rows[0].keys()
        np.savez_compressed(path, **{k: np.stack([np.asarray(r[k]) for r in rows]) for k in keys})
        print(f"{path}: {len(rows)} cases")


def from_dumps(cases, paths):
    for p in paths:
        if not os.path.exists(p):
            print("missing", p)
            continue
        z = np.load(p)
        rep = z["report"]
        if rep.shape[-1] != 12 or z["trace"].shape[-1] != TRACE_INTS:
            print("skipping (old trace / report format)", p)
            continue
        nf = z["trace"].shape[1]
        before = {k: z["before_" + k] for k in BEFORE}
        after = {k: z["after_" + k] for k in AFTER} if "after_qpos" in z.files else None
        idx = select(rep, z["err_qvel"] if "err_qvel" in z.files else None)
        for i in idx:
            cases.add(os.path.basename(p), nf, i, {k: v[i] for k, v in before.items()}, z["action"][i], z["trace"][i], rep[i],
                      None if after is None else {k: v[i] for k, v in after.items()})
        print(f"{p}: {len(idx)} envs kept of {len(rep)}")


def from_seeds(cases, seeds, B):
    import torch

    import helpers as H
    import parity as P
    from vnl_brax_imitation_amd.envs.rodent import RodentTracking

    env = RodentTracking(H.reference_clip(), num_envs=B, device="cuda:0", **H.env_kwargs())
    env1 = RodentTracking(H.reference_clip(), num_envs=B, device="cuda:0", **{**H.env_kwargs(), "n_frames": 1})
    o64, o32 = H.make_oracle(env, "f64"), H.make_oracle(env, "f32")
    o64_1, o32_1 = H.make_oracle(env1, "f64"), H.make_oracle(env1, "f32")
    for seed in seeds:
        rng = np.random.default_rng(seed)
        sf = rng.integers(0, 235, B).astype(np.int32)
        noise = (1e-3 * rng.standard_normal((B, 74))).astype(np.float32)
        act = np.clip(0.3 * rng.standard_normal((B, 30)), -1, 1).astype(np.float32)
        st, err, dev, rep, _ = P.control_step_follow(env, o64, o32, sf, noise, act)
        ps = st.pipeline_state
        after = {k: P.to_np(getattr(ps, k)) for k in AFTER}
        idx = select(rep, err["qvel"])
        for i in idx:
            cases.add(f"seed{seed}_control_step", env._n_frames, i, {k: v[i] for k, v in P.LAST["before"].items()}, act[i],
                      P.LAST["trace"][i], rep[i], {k: v[i] for k, v in after.items()})
        print(f"seed {seed} control step: {len(idx)} envs kept")
        env1.debug(2)
        s1 = env1.reset(start_frame=torch.from_numpy(sf), noise=torch.from_numpy(noise))
        for sub in range(3):
            new, e1, d1, r1, _ = P.follow_compare(env1, o64_1, o32_1, lambda o: P.oracle_state_from(env1, o, s1),
                                                  lambda: env1.step(s1, torch.from_numpy(act)), act, 1)
            ps = new.pipeline_state
            after = {k: P.to_np(getattr(ps, k)) for k in AFTER}
            idx = select(r1, None, n_control=2)
            for i in idx:
                cases.add(f"seed{seed}_substep{sub}", 1, i, {k: v[i] for k, v in P.LAST["before"].items()}, act[i],
                          P.LAST["trace"][i], r1[i], {k: v[i] for k, v in after.items()})
            s1 = new
        env1.debug(0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--from-dumps", nargs="*", default=None)
    ap.add_argument("--seeds", nargs="*", type=int, default=[])
    ap.add_argument("--envs", type=int, default=4096)
    ap.add_argument("--out", default=OUT)
    a = ap.parse_args()
    cases = Cases()
    if a.from_dumps is not None:
        paths = a.from_dumps or [os.path.join(ROOT, "gpurun_out", f) for f in
                                 ("parity_fail_substep1_B4096.npz", "parity_fail_substep2_B4096.npz",
                                  "parity_fail_substep3_B4096.npz", "control_step_hip_B4096.npz")]
        from_dumps(cases, paths)
    if a.seeds:
        from_seeds(cases, a.seeds, a.envs)
    cases.save(a.out)


if __name__ == "__main__":
    main()
