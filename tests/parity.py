"""Discriminating parity checks: product (HIP through the C-ABI, or the host simulation of the same source) vs the
float64 oracle on IDENTICAL inputs, at the granularity where "within 1e-5" is a well-posed statement.

The 6-iteration CG solve of the reference's configuration is not converged, so its result depends on a chain of
discrete decisions (warm start, iteration counts, line-search bracket replacements, which rows are active at each
trial step length).  A float32 implementation reproduces the float64 result to rounding as long as every decision
comes out the same; where one flips, the result jumps by orders of magnitude more -- in ANY float32 implementation,
the float32 build of the oracle included.  The checks therefore separate the two cases with the solver traces both
sides record (csrc/vnl_types.h VNL_TRACE_*, oracle/vnl_oracle.c ORC_TRACE_*):

  * substep resynchronised comparison (`resync_substeps`): ONE physics substep at a time, the oracle restarted from
    the product's own state each time, so nothing compounds: every env whose decisions agree must match the oracle
    within TOL_* of the array's scale; an env whose outputs deviate more WITHOUT a flipped decision is a failure;
  * per-stage outputs of the forward pass from the LDS dump (`stage_errors`);
  * the glue (obs / traj / reward terms / termination) on the product's own post-step state (`glue_errors`);
  * the full control step with the per-substep traces (`control_step_flip_analysis`).
"""
from __future__ import annotations

import os

import numpy as np
import torch

import helpers as H

# decision entries of one solver-iteration record (64 ints): [1] ls iterations, [2] swap bits, [3] pick, [4..24) counts
_ITER_DECISIONS = slice(1, 24)


def decisions_equal(tr_a: np.ndarray, tr_b: np.ndarray) -> np.ndarray:
    """(..., TRACE_INTS) int arrays -> bool (...): same warm-start choice, iteration count and per-iteration decisions."""
    a, b = np.asarray(tr_a), np.asarray(tr_b)
    ok = (a[..., 0] == b[..., 0]) & (a[..., 1] == b[..., 1])
    ia = a[..., 8:8 + 64 * 8].reshape(*a.shape[:-1], -1, 64)[..., _ITER_DECISIONS]
    ib = b[..., 8:8 + 64 * 8].reshape(*b.shape[:-1], -1, 64)[..., _ITER_DECISIONS]
    return ok & np.all(ia == ib, axis=(-1, -2))


def describe_flip(tr_a: np.ndarray, tr_b: np.ndarray) -> str:
    """First differing decision of two single-solve traces, human readable."""
    if tr_a[0] != tr_b[0]:
        return f"warm-start choice {tr_a[0]} vs {tr_b[0]}"
    names = {1: "line-search iterations", 2: "bracket replacement bits", 3: "final pick (0 none / 1 lo / 2 hi)"}
    for it in range(8):
        ra, rb = tr_a[8 + 64 * it: 72 + 64 * it], tr_b[8 + 64 * it: 72 + 64 * it]
        if it >= max(tr_a[1], tr_b[1]):
            break
        if it >= min(tr_a[1], tr_b[1]):
            return f"CG exit: {tr_a[1]} vs {tr_b[1]} iterations"
        for k in range(4, 24):
            if ra[k] != rb[k]:
                return f"CG iteration {it}: active rows at trial step {k - 4}: {ra[k]} vs {rb[k]}"
        for k in (1, 2, 3):
            if ra[k] != rb[k]:
                return f"CG iteration {it}: {names[k]} {ra[k]} vs {rb[k]}"
    if tr_a[1] != tr_b[1]:
        return f"CG exit: {tr_a[1]} vs {tr_b[1]} iterations"
    return "none"


def to_np(t) -> np.ndarray:
    return t.detach().cpu().numpy() if isinstance(t, torch.Tensor) else np.asarray(t)


def oracle_state_from(env, oracle, state) -> dict:
    """Oracle batch state carrying the product's CURRENT pipeline state and frame counters (float64 copies)."""
    B = env.num_envs
    ost = oracle.new_state(B)
    ps = state.pipeline_state
    for k in ("qpos", "qvel", "act", "qacc_warmstart", "qfrc_actuator"):
        ost[k][:] = to_np(getattr(ps, k)).astype(np.float64)
    ost["xpos"][:] = to_np(ps.xpos).reshape(B, -1).astype(np.float64)
    ost["xmat1"][:] = to_np(ps.xmat[:, 1]).reshape(B, 9).astype(np.float64)
    ost["com1"][:] = to_np(ps.subtree_com_root).astype(np.float64)
    ost["cur_frame"][:] = to_np(state.info["cur_frame"])
    ost["sub_clip_frame"][:] = to_np(state.info["sub_clip_frame"])
    return ost


def per_env_scaled(a, ref) -> np.ndarray:
    """max_k |a - ref| per env, divided by the scale of the whole reference array (used for the glue outputs, whose
    entries share one unit; the physics state goes through per_env_grouped)."""
    a, ref = np.asarray(a, np.float64).reshape(len(ref), -1), np.asarray(ref, np.float64).reshape(len(ref), -1)
    return np.max(np.abs(a - ref), axis=1) / max(float(np.max(np.abs(ref))), 1e-30)


# Field groups of the physics state (free-joint root first, as in every model of the reference): quantities of one unit
# and one physical meaning are scaled together, PER ENV -- not by the largest entry of the whole batch (a batch-wide qvel
# scale of 22.7 rad/s made "1e-5" mean 5e-4 of a median entry).
GROUPS = {
    "qpos": (("root position", slice(0, 3)), ("root quaternion", slice(3, 7)), ("joint angles", slice(7, None))),
    "qvel": (("linear velocity", slice(0, 3)), ("angular velocity", slice(3, 6)), ("joint velocities", slice(6, None))),
    "qacc_warmstart": (("linear acceleration", slice(0, 3)), ("angular acceleration", slice(3, 6)),
                       ("joint accelerations", slice(6, None))),
}


def group_scales(ref, key: str) -> list:
    """[(name, slice, scale (B,))]: the scale of group g in env e is max(largest |ref| of the group in that env, the batch
    MEDIAN of that quantity): an env at rest in some group is measured against a typical env, never against the batch's
    worst one."""
    ref = np.asarray(ref, np.float64).reshape(len(ref), -1)
    out = []
    for name, sl in GROUPS.get(key, (("all", slice(0, None)),)):
        own = np.max(np.abs(ref[:, sl]), axis=1) if ref[:, sl].shape[1] else np.zeros(len(ref))
        out.append((name, sl, np.maximum(np.maximum(own, float(np.median(own))), 1e-30)))
    return out


def per_env_grouped(a, ref, key: str) -> np.ndarray:
    """max over the field groups of `key` of (max_k in group |a - ref|) / (scale of that group in that env)."""
    a, ref = np.asarray(a, np.float64).reshape(len(ref), -1), np.asarray(ref, np.float64).reshape(len(ref), -1)
    err = np.zeros(len(ref))
    for _, sl, scale in group_scales(ref, key):
        if ref[:, sl].shape[1]:
            err = np.maximum(err, np.max(np.abs(a[:, sl] - ref[:, sl]), axis=1) / scale)
    return err


STATE_KEYS = ("qpos", "qvel", "act", "qacc_warmstart", "xpos", "qfrc_actuator")


def state_errors(state, ost) -> dict:
    B = len(ost["qpos"])
    ps = state.pipeline_state
    out = {k: per_env_grouped(to_np(getattr(ps, k)).reshape(B, -1), ost[k], k) for k in STATE_KEYS}
    out["com1"] = per_env_grouped(to_np(ps.subtree_com_root), ost["com1"], "com1")
    return out


def compliance(err: dict, tol: float = 1e-5) -> dict:
    """{field: fraction of the envs whose error is within `tol` of the env's own group scales}"""
    return {k: float(np.mean(v <= tol)) for k, v in err.items()}


def _as_f32_state(ost: dict) -> dict:
    return {k: (v.astype(np.float32) if v.dtype == np.float64 else v.copy()) for k, v in ost.items()}


# Legitimacy bounds for followed decisions (oracle/vnl_oracle.c ORC_FOLLOW_REPORT), and the per-env error bound:
# a product output may deviate from the float64 oracle (same decisions) by at most max(TOL, K_SENS x the largest of
# N_SENS deviations of the float32 build of the ORACLE on the same env with the same decisions: inputs as given, and
# moved by one float32 rounding) -- i.e. north_star's 1e-5 of the array's
# scale wherever the env is well conditioned, and a bounded multiple of the env's own float32 rounding sensitivity
# where the unconverged, stiff solve amplifies rounding beyond that.
TOL = 1e-5
K_SENS = 50.0
N_SENS = 6
# the three tie measures are in units of 1e-6 x the magnitude of the terms summed (~16 float32 roundings): <= 1 means a
# float32 evaluation cannot tell the two sides of the comparison apart
LS_EXCESS_MAX = 1.0      # how much more the followed line-search step may cost than the oracle's own
EXIT_TIE_MAX = 1.0       # CG-exit disagreements: distance of the natural exit test from its threshold
WARM_TIE_MAX = 1.0       # warm-start disagreements: cost difference of the two starting points
# (report [4], informative: how close to its switching point the row sits whose activity differs at a trial step, relative to
# the magnitudes of the terms of its value J.qacc - aref + alpha J.search.  The product forms contact rows of J.v as
# DIFFERENCES of prefix sums over the dofs (vnl_body.h jac_mul), whose rounding error scales with the prefix, not with the
# row: 1.4e-5 typical, up to 4e-4 seen in tests/parity_sweep.py)
KINK_MARGIN_MAX = 1e-3
FLIP_CAP = 0.1  # scaled error an env with a flipped decision may show after a multi-substep comparison
ROW_DEPTH_MAX = 2e-6     # a limit / contact whose presence differs must be violated by less than this (m or rad)


LAST = {}  # inputs of the most recent follow_compare (state before, action, product trace): saved by the tests on failure


def save_last(path: str, **extra) -> None:
    import os

    os.makedirs(os.path.dirname(path), exist_ok=True)
    np.savez_compressed(path, action=LAST["action"], trace=LAST["trace"], **{"before_" + k: v for k, v in LAST["before"].items()},
                        **extra)


def follow_compare(env, o64, o32, state_before_fn, step_fn, action, n_frames):
    """One product step (`step_fn()` -> new State, traces recorded) against the float64 and float32 oracles started
    from the product's own pre-step state and made to FOLLOW the product's solver decisions.  Returns (err, dev32,
    report): per-env scaled errors of the product vs o64, of o32 vs o64, and the legitimacy report [B][n_frames][8]."""
    s64 = state_before_fn(o64)
    s32 = _as_f32_state(s64)
    LAST["before"] = {k: v.copy() for k, v in s64.items()}
    st = step_fn()
    ptr = env.solver_trace().numpy()
    LAST["trace"], LAST["action"] = ptr.copy(), np.asarray(action).copy()
    s64, _, rep = o64.env_step_follow(s64, action.astype(np.float64), ptr)
    err = state_errors(st, s64)
    # float32 sensitivity of every env: the float32 oracle on the same inputs and on inputs moved by one float32
    # rounding (N_SENS samples of the env's rounding noise; the largest deviation from the float64 result counts)
    dev = {k: np.zeros(len(err[k])) for k in err}
    rng = np.random.default_rng(12345)
    for n in range(N_SENS):
        t32 = {k: v.copy() for k, v in s32.items()}
        if n > 0:
            for k in ("qpos", "qvel", "act", "qacc_warmstart"):
                t32[k] = (t32[k] * (1 + np.float32(2.0 ** -23) * rng.integers(-1, 2, t32[k].shape).astype(np.float32))).astype(np.float32)
        t32, _, _ = o32.env_step_follow(t32, action.astype(np.float32), ptr)
        for k in STATE_KEYS:
            dev[k] = np.maximum(dev[k], per_env_grouped(t32[k].astype(np.float64), s64[k], k))
        dev["com1"] = np.maximum(dev["com1"], per_env_grouped(t32["com1"].astype(np.float64), s64["com1"], "com1"))
    if n_frames == 1:
        # Envs in whose solve a followed decision was NOT a tie for the oracle (non_tie): a row on the other side of its
        # switching point at one of the product's trial step lengths (a kink), or a bracket / exit / warm-start decision
        # that the oracle's own arithmetic does not find marginal.  Either the product is wrong there, or -- what the sweep
        # over seeds found every time -- the oracle's REPLAY of the product's decisions is distorted: after one differing
        # trial point the two line searches run on different brackets, and replaying bracket decisions on other trial
        # points leaves the oracle off its minimiser while both searches, left alone, end at the same one.  The two cases
        # are told apart by the NATURAL float64 oracle: these envs are held to the same bound against it, with the natural
        # float32 oracle's deviation as the sensitivity.
        r = rep.reshape(len(rep), -1, rep.shape[-1])[:, 0]
        kink = np.where(non_tie(r))[0]
        if len(kink):
            before = LAST["before"]
            n64 = o64.env_step({k: v.copy() for k, v in before.items()}, action.astype(np.float64))
            e_nat = state_errors(st, n64)
            d_nat = {k: np.zeros(len(err[k])) for k in err}
            rng = np.random.default_rng(54321)
            for n in range(N_SENS):
                t32 = _as_f32_state({k: v.copy() for k, v in before.items()})
                if n > 0:
                    for k in ("qpos", "qvel", "act", "qacc_warmstart"):
                        t32[k] = (t32[k] * (1 + np.float32(2.0 ** -23) * rng.integers(-1, 2, t32[k].shape).astype(np.float32))).astype(np.float32)
                t32 = o32.env_step(t32, action.astype(np.float32))
                for k in d_nat:
                    d_nat[k] = np.maximum(d_nat[k], per_env_grouped(t32[k].astype(np.float64), n64[k], k))
            for k in err:
                err[k][kink], dev[k][kink] = e_nat[k][kink], d_nat[k][kink]
            LAST["kink_envs"] = kink
    return st, err, dev, rep, s64


def assert_no_less_accurate_than_f32_oracle(err: dict, dev: dict, keep=None) -> None:
    """Distribution-level companion of the per-env bound (the per-env ratio of two samples of a heavy-tailed rounding
    process is noisy): over the batch, the product must be no less accurate than the float32 build of the oracle --
    median error within 1.5 x, and no more envs beyond TOL than the float32 oracle has (+25 %, + Poisson noise)."""
    # `keep`: envs that take part (a multi-substep comparison leaves out the envs with a flipped decision: the float32
    # oracle FOLLOWS the product's decisions, so its deviation does not contain what the flip costs, the product's does)
    if keep is None:
        keep = np.ones(len(err["qpos"]), dtype=bool)
    for f in ("qpos", "qvel", "qacc_warmstart", "xpos"):
        e, d = err[f][keep], dev[f][keep]
        assert np.median(e) <= 1.5 * np.median(d) + 1e-7, (f, np.median(e), np.median(d))
        # counts of rare events: 25 % more than the float32 oracle's, plus three standard deviations of a Poisson count of that
        # size (7 against 3 in one of 100 batch comparisons of tests/parity_sweep.py is noise; 250 against 100 is not)
        ne, nd = int((e > TOL).sum()), int((d > TOL).sum())
        assert ne <= 1.25 * nd + 3.0 * np.sqrt(nd + 1.0) + 3, (f, ne, nd)


def bound_violations(err: dict, dev: dict, tol=TOL, k=K_SENS) -> dict:
    """{field: indices of envs with err > max(tol, k * dev)}"""
    return {f: np.where(err[f] > np.maximum(tol, k * dev[f]))[0] for f in err}


ALPHA_GAP_MAX = 1e-2  # largest relative distance between a trial step length of the oracle and the followed side's


def faithful(r: np.ndarray) -> np.ndarray:
    """rows of a follow report (..., 12) -> bool: the oracle's REPLAY of the other side's decisions was faithful -- every
    trial step length of its line searches was (within ALPHA_GAP_MAX) the other side's, and the same number of rows was
    active at each.  Only then do the recorded bracket decisions refer to the oracle's own trial points, and only then is a
    followed decision that is not a tie evidence against the product.  (With one differing trial point the two searches
    run on different brackets from there on; replaying decisions taken on other points leaves the oracle off its
    minimiser although both searches, left alone, end at the same one: those solves are re-checked against the NATURAL
    oracle instead, see follow_compare.  Recorded data: of 32,768 solves in the four 4096-env dumps behind
    tests/golden/parity_cases.npz, every faithful replay had all three tie measures <= 0.07.)"""
    return (r[..., 3] == 0) & (r[..., 8] <= ALPHA_GAP_MAX)


def non_tie(r: np.ndarray) -> np.ndarray:
    """rows of a follow report (..., 12) -> bool: a followed decision of that solve was not a tie for the oracle."""
    return (r[..., 0] > LS_EXCESS_MAX) | (r[..., 1] > EXIT_TIE_MAX) | (r[..., 2] > WARM_TIE_MAX) | (r[..., 3] > 0)


def legitimacy_summary(rep: np.ndarray) -> dict:
    """Tie measures over the solves whose followed decisions WERE ties for the oracle, and how many were not (non_tie:
    those envs are checked against the natural oracle instead, see follow_compare; they must be rare)."""
    r = rep.reshape(-1, rep.shape[-1])
    odd = non_tie(r)
    tie = ~odd
    kink = r[:, 3] > 0
    mx = lambda col, m: float(r[m, col].max()) if m.any() else 0.0  # noqa: E731
    return dict(ls_excess=mx(0, tie), exit_tie=mx(1, tie), warm_tie=mx(2, tie), non_tie_solves=int(odd.sum()), solves=int(len(r)),
                ls_excess_non_tie=mx(0, odd), kink_solves=int(kink.sum()), kink_margin=mx(4, kink),
                decisions_differing_mean=float(r[:, 5].mean()), rows_followed=int(r[:, 6].sum()), row_depth=mx(7, np.ones(len(r), bool)),
                trial_alpha_gap=mx(8, np.ones(len(r), bool)))


def drifted(rep: np.ndarray) -> np.ndarray:
    """(B, n_frames, 12) report -> bool (B,): in some substep a followed decision was NOT a tie for the oracle (a step
    length worse than rounding explains, an exit / warm-start choice off its threshold, an active-set difference with no
    row on its switching point, a constraint row present on one side only beyond rounding).  In a multi-substep
    comparison this is the evidence that the two sides' states had drifted apart far enough for a discrete decision to
    flip -- the oracle was then made to follow a decision that belongs to a (slightly) different state."""
    return (non_tie(rep) | (rep[..., 7] > ROW_DEPTH_MAX)).any(axis=-1)


def check_control_step(err: dict, dev: dict, rep: np.ndarray, verbose: bool = True, max_flipped: int | None = None) -> int:
    """Assertion of a multi-substep follow comparison.  Every env within max(TOL, K_SENS x its float32 sensitivity);
    an env outside that bound must SHOW a flipped decision in its report (`drifted`) and stay within max(1000 x its
    sensitivity, FLIP_CAP); at most 1 % (at least 2) of the envs may be in that state.  Returns the number of such envs.
    (Legitimacy of the decisions themselves is asserted by the resynchronised single-substep test: over several
    substeps the two sides' states drift apart, and in a badly conditioned env far enough for a later decision of the
    product to stop being a tie for the oracle, which then follows a decision that belongs to a different state.)"""
    B = rep.shape[0]
    flipped = drifted(rep)
    viol = bound_violations(err, dev)
    if B >= 32:
        assert_no_less_accurate_than_f32_oracle(err, dev, keep=~flipped)
    if verbose:
        print(f"   envs whose later decisions flipped against the oracle's drifted state: {int(flipped.sum())} of {B}")
    # (1 % for the rodent; a model whose contacts sit at their activation threshold passes its own allowance)
    assert flipped.sum() <= (max(2, B // 100) if max_flipped is None else max_flipped), int(flipped.sum())
    for f, idx in viol.items():
        unexplained = [int(i) for i in idx if not flipped[i]]
        assert not unexplained, (f, unexplained[:8], err[f][unexplained[:8]], dev[f][unexplained[:8]])
        for i in idx:
            if verbose:
                print(f"   env {i} {f}: err {err[f][i]:.2e}, float32 sensitivity {dev[f][i]:.2e}; (mismatched trial "
                      f"points, nearest-row margin) per substep: {[(int(r[3]), float(r[4])) for r in rep[i]]}")
            # (what a flipped decision costs is not tied to the env's rounding sensitivity -- it is a different discrete
            # path; the float32 ORACLE, left to its own decisions, is up to 3e-2 of the scale off the float64 one in the worst env
            # of a batch (smoke()).  The cap only rejects garbage.)
            assert err[f][i] <= max(1000 * dev[f][i], FLIP_CAP), (f, i, err[f][i], dev[f][i])
    return int(flipped.sum())


def natural_check(state, o64, o32, action, q50=1.5, q90=3.0, fields=("qpos", "qvel", "qacc_warmstart")) -> dict:
    """Decision-INDEPENDENT companion of a follow comparison: the product's post-step state against the
    NATURAL float64 oracle (its own decisions) started from the pre-step state of the last follow_compare, with the
    natural float32 oracle as the yardstick.  A wrong line-search or active-set decision of the product would be replayed
    by a following oracle, but not by this one.  Per-env agreement is not expected (different discrete paths in the
    ill-conditioned envs); the DISTRIBUTION must be no worse than the float32 oracle's: median within `q50` x and 90 %
    quantile within `q90` x the float32 oracle's (+ TOL)."""
    before = LAST["before"]
    n64 = o64.env_step({k: v.copy() for k, v in before.items()}, np.asarray(action, np.float64))
    n32 = o32.env_step(_as_f32_state({k: v.copy() for k, v in before.items()}), np.asarray(action, np.float32))
    e = state_errors(state, n64)
    out = {}
    for f in fields:
        d = per_env_grouped(n32[f].astype(np.float64), n64[f], f)
        out[f] = dict(median=float(np.median(e[f])), q90=float(np.quantile(e[f], 0.9)), median_f32=float(np.median(d)),
                      q90_f32=float(np.quantile(d, 0.9)))
        assert out[f]["median"] <= q50 * out[f]["median_f32"] + TOL, (f, out[f])
        assert out[f]["q90"] <= q90 * out[f]["q90_f32"] + TOL, (f, out[f])
    return out


def flip_causes(rep: np.ndarray) -> dict:
    """For the envs `drifted` flags: what the evidence is -- a constraint row present on one side only (report [6]), an
    active-set mismatch at a trial step (report [3]) or a line-search / exit / warm-start decision off its tie."""
    fl = np.where(drifted(rep))[0]
    rows = rep[fl][..., 6].sum(axis=-1) > 0
    kinks = rep[fl][..., 3].sum(axis=-1) > 0
    return dict(envs=fl, row_presence=rows, active_set=kinks & ~rows, other=~rows & ~kinks)


def assert_legitimate(rep: np.ndarray) -> dict:
    """Single-substep comparison: decisions that are not ties for the oracle must be rare (measured over 100 seeds x 3
    substeps x 4096 envs: 0-2 per 4096 solves) -- their envs are checked against the natural oracle by follow_compare --
    and a constraint row present on one side only must be within rounding of its threshold."""
    s = legitimacy_summary(rep)
    # hard: wherever the replay was faithful (same trial points, same active sets), every followed decision is a tie
    r = rep.reshape(-1, rep.shape[-1])
    f = faithful(r)
    s["faithful_solves"] = int(f.sum())
    s["faithful_worst_tie"] = float(np.max(np.maximum.reduce([r[f, 0], r[f, 1], r[f, 2]]))) if f.any() else 0.0
    assert s["faithful_worst_tie"] <= 1.0, s
    # the rest: rare, and held to the same bound against the natural oracle (follow_compare)
    assert s["non_tie_solves"] <= max(2, s["solves"] // 500), s
    assert s["row_depth"] <= ROW_DEPTH_MAX, s
    return s


def resync_substeps(env1, o64, o32, sf, noise, action, nsub=5, assert_legit=True):
    """env1 / oracles: n_frames = 1.  Runs `nsub` consecutive substeps of the product; before each one the oracles are
    reloaded with the product's own state, so every comparison is ONE substep on identical inputs (nothing compounds).
    Returns one (err, dev32, report) per substep."""
    env1.debug(2)
    st = [env1.reset(start_frame=torch.from_numpy(sf), noise=torch.from_numpy(noise))]
    act_t = torch.from_numpy(action)
    out = []
    for _ in range(nsub):
        new, err, dev, rep, _ = follow_compare(env1, o64, o32, lambda o: oracle_state_from(env1, o, st[0]),
                                               lambda: env1.step(st[0], act_t), action, 1)
        st[0] = new
        out.append((err, dev, rep))
        try:
            if assert_legit:
                assert_legitimate(rep)
        except AssertionError:
            save_last(os.path.join(H.ROOT, "gpurun_out", f"parity_fail_substep{len(out) - 1}_B{env1.num_envs}.npz"), report=rep)
            env1.debug(0)
            raise
    env1.debug(0)
    return out


def stage_errors(env, oracle, state_before, action, envs):
    """Forward-pass stage outputs of the product (LDS image left by the last forward of a step, debug mode 2, taken
    by the caller) against the oracle's forward on the same inputs, for the env indices `envs`.  `state_before`:
    dict of numpy arrays (qpos, qvel, act, qacc_warmstart) the step started from; the env must have n_frames = 1.
    Returns {name: per-env scaled error} plus the discrete sets (constraint-row presence, active contacts)."""
    m = env.sys
    nv, nefc, nlimit = int(m.scalars["nv"]), int(m.scalars["nefc"]), int(m.scalars["nlimit"])
    ncon = int(m.scalars["ncon"])
    sec = {k: to_np(env.scratch(k)) for k in ("qfrc_smooth", "qacc_smooth", "qacc", "qfrc_constraint", "efc_D", "Jaref",
                                              "act_list")}
    ctrl = np.clip(np.asarray(action, np.float64), -1.0, 1.0)
    res = {k: [] for k in ("qfrc_smooth", "qacc_smooth", "efc_D", "Jaref", "qacc", "qfrc_constraint")}
    rows_equal, contacts_equal, traces = [], [], []
    ref_scale = {}
    refs = {k: [] for k in res}
    for i in envs:
        oracle.set(qpos=state_before["qpos"][i], qvel=state_before["qvel"][i], act=state_before["act"][i],
                   ctrl=ctrl[i], qacc_warmstart=state_before["qacc_warmstart"][i])
        oracle.call("forward")
        J = oracle.field("efc_J").reshape(nefc, nv)
        present = np.abs(J).sum(1) > 0
        D = np.abs(sec["efc_D"][i])  # limit rows carry the Jacobian sign on efc_D
        rows_equal.append(bool(np.array_equal(D != 0, present)))
        # active contact list: the first `na` bytes of the section are the contacts with D != 0, then their count
        raw = np.ascontiguousarray(sec["act_list"][i]).view(np.uint8)
        na = int(raw[4 * ((ncon + 3) // 4): 4 * ((ncon + 3) // 4) + 4].view(np.int32)[0])
        want = [c for c in range(ncon) if present[nlimit + 4 * c]]
        contacts_equal.append(list(raw[:na]) == want)
        traces.append(oracle.solver_trace)
        ref = dict(qfrc_smooth=oracle.field("qfrc_smooth").copy(), qacc_smooth=oracle.field("qacc_smooth").copy(),
                   qacc=oracle.field("qacc").copy(), qfrc_constraint=oracle.field("qfrc_constraint").copy(),
                   efc_D=np.where(present, oracle.field("efc_D"), 0.0),
                   Jaref=np.where(present, J @ oracle.field("qacc") - oracle.field("efc_aref"), 0.0))
        got = dict(qfrc_smooth=sec["qfrc_smooth"][i], qacc_smooth=sec["qacc_smooth"][i], qacc=sec["qacc"][i],
                   qfrc_constraint=sec["qfrc_constraint"][i], efc_D=np.where(present, D, 0.0),
                   Jaref=np.where(present, sec["Jaref"][i], 0.0))
        for k in res:
            res[k].append(np.max(np.abs(got[k] - ref[k])))
            refs[k].append(np.max(np.abs(ref[k])))
    for k in res:
        ref_scale[k] = max(float(np.max(refs[k])), 1e-30)
        res[k] = np.asarray(res[k]) / ref_scale[k]
    return res, np.asarray(rows_equal), np.asarray(contacts_equal), np.asarray(traces)


GLUE_KEYS = ("obs", "traj", "reward", "done", "termination_error")
METRIC_NAMES = ("rcom", "rvel", "rtrunk", "rquat", "ract", "rapp", "termination_error")


def _glue_run(env, oracle, state, old_qpos, old_xpos, old_cur_frame, old_sub_frame, f32=False, old_extra=None):
    ost = oracle_state_from(env, oracle, state)
    if f32:
        ost = _as_f32_state(ost)
    ost["cur_frame"][:] = old_cur_frame
    ost["sub_clip_frame"][:] = old_sub_frame
    oracle.env_glue(ost, old_qpos, old_xpos, *(old_extra or ()))
    return ost


def glue_errors(env, o64, o32, state, old_qpos, old_xpos, old_cur_frame, old_sub_frame, old_extra=None):
    # old_extra: (old qvel, old com1, old qfrc_actuator[, action]) for the envs whose reward reads the state before the step
    """obs / traj / the reward terms / done / counters of the product's step against the oracle's glue evaluated on
    the product's OWN post-step pipeline state (so physics sensitivity cannot mask a glue error).  Returns (err, dev32,
    flags): per-env errors vs the float64 oracle scaled by each array's scale, the float32 oracle's own deviation from
    the float64 one (rquat goes through arccos near 1, whose float32 conditioning is part of the reference's own
    arithmetic), and the exact-equality flags."""
    a = _glue_run(env, o64, state, old_qpos, old_xpos, old_cur_frame, old_sub_frame, old_extra=old_extra)
    b = _glue_run(env, o32, state, old_qpos, old_xpos, old_cur_frame, old_sub_frame, f32=True, old_extra=old_extra)
    raw = state.info["_raw"]  # the buffers the kernel fills (AntTracking presents obs = [traj | raw obs] and six metrics)
    err = {"obs": per_env_scaled(to_np(raw["obs"]), a["obs"]), "traj": per_env_scaled(to_np(state.info["traj"]), a["traj"])}
    dev = {"obs": per_env_scaled(b["obs"], a["obs"]), "traj": per_env_scaled(b["traj"], a["traj"])}

    def rel(x, ref):
        return np.abs(np.asarray(x, np.float64) - ref) / max(float(np.max(np.abs(ref))), 1e-30)

    for j, name in enumerate(METRIC_NAMES):
        err[name], dev[name] = rel(to_np(raw["metrics"][:, j]), a["metrics"][:, j]), rel(b["metrics"][:, j], a["metrics"][:, j])
    err["reward"], dev["reward"] = rel(to_np(state.reward), a["reward"]), rel(b["reward"], a["reward"])
    err["info.termination_error"] = rel(to_np(state.info["termination_error"]), a["termination_error"])
    dev["info.termination_error"] = rel(b["termination_error"], a["termination_error"])
    flags = dict(done_equal=np.array_equal(to_np(state.done).astype(np.float64), a["done"]),
                 frames_equal=(np.array_equal(to_np(state.info["cur_frame"]), a["cur_frame"]) and
                               np.array_equal(to_np(state.info["sub_clip_frame"]), a["sub_clip_frame"])))
    return err, dev, flags


def control_step_follow(env, o64, o32, sf, noise, action, dump_to=None):
    """One full control step (n_frames substeps + glue) from a reset, product vs oracles following the product's
    per-substep decisions.  Returns (state, err, dev32, report, oracle state)."""
    env.debug(1)
    st0 = env.reset(start_frame=torch.from_numpy(sf), noise=torch.from_numpy(noise))
    res = follow_compare(env, o64, o32, lambda o: oracle_state_from(env, o, st0),
                         lambda: env.step(st0, torch.from_numpy(action)), action, env._n_frames)
    env.debug(0)
    if dump_to:
        st, err, dev, rep, s64 = res
        ps = st.pipeline_state
        save_last(dump_to, report=rep, **{"after_" + k: to_np(getattr(ps, k)) for k in ("qpos", "qvel", "qacc_warmstart")},
                  **{"err_" + k: v for k, v in err.items()}, **{"dev_" + k: v for k, v in dev.items()})
    return res
