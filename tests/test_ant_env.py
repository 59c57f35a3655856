"""AntTracking (reference envs/ant.py:25-438) on the same kernels: SURVEY 8(f) f4.

CPU tier: the float64 host build of the kernels against the dense oracle, and the statements of ant.py that differ from
the rodent env, checked directly.  GPU tier: one control step through the C-ABI, following the solver decisions, and the
glue on the product's own state.  The reference's ant clip is not shipped: the clip is a seeded gait around the init pose."""
import os

import numpy as np
import pytest
import torch

import helpers as H
import parity as P
from vnl_brax_imitation_amd.model import mjcf
from vnl_brax_imitation_amd.preprocessing import mjx_preprocess as pp

ANT_NPZ = os.path.join(H.ROOT, "vnl-brax-imitation_amd", "data", "ant.npz")
PARAMS = dict(solver="cg", iterations=6, ls_iterations=6)


def _clip(m, T=60):
    t = np.arange(T)[:, None] * 0.02
    q = np.zeros((T, 15))
    q[:, 2], q[:, 3] = 0.55, 1.0
    q[:, 0] = 0.2 * t[:, 0]
    q[:, 7:] = np.array([0.0, 1.0, 0.0, -1.0, 0.0, -1.0, 0.0, 1.0]) + 0.15 * np.sin(2 * np.pi * 1.5 * t + np.arange(8))
    return pp.process_qpos(m, q, max_qvel=20.0, dt=0.02)


NEWTON = dict(solver="newton", iterations=1, ls_iterations=4)  # reference configs/env_config.yaml:16-21


def _env(B, real="float", device="cpu", params=None, **kw):
    from vnl_brax_imitation_amd import envs

    m = mjcf.CompiledModel.load(ANT_NPZ)
    import contextlib

    with (H.hostsim_backend(real) if device == "cpu" else contextlib.nullcontext()):
        return envs.get_environment("ant", params=params or PARAMS, clip_length=60, episode_length=20, reference_clip=_clip(m), model=m,
                                    num_envs=B, device=device, **kw)


def _oracle(env, precision="f64"):
    from oracle.oracle import Oracle
    from vnl_brax_imitation_amd.model import blob

    m = env.sys
    o = Oracle(blob.to_blob(m), precision)
    o.bind_env(env.env_spec(), env.clip_arrays(0), int(m.scalars["nbody"]), int(m.scalars["nq"]), int(m.scalars["nv"]),
               int(m.scalars["nu"]))
    return o


def test_ant_env_float64_build_matches_oracle_and_reference_semantics():
    B = 6
    env = _env(B, "double")
    nb = 10  # brax's fused body list: world, torso and the eight bodies that carry a hinge
    assert env.env_spec()["nb"] == nb and int(env.sys.scalars["eulerdamp"]) == 0
    ntraj = 5 * (2 * nb * 3 + 3 + 8)
    assert env.traj_size == ntraj and env.observation_size == ntraj + 15 + 14
    st = env.reset()
    assert (st.info["cur_frame"] == 0).all()  # ant.py:103: start_frame = 0
    o = _oracle(env)
    ost = o.env_reset(np.zeros(B, np.int32), np.zeros((B, 15)))
    ps, raw = st.pipeline_state, st.info["_raw"]
    for k in ("qpos", "qvel", "xpos", "qacc_warmstart"):
        assert H.scaled_err(getattr(ps, k).reshape(B, -1).numpy(), ost[k]) < 1e-10, k
    # obs = [traj features | qpos | qvel] (ant.py:322-338)
    assert torch.equal(st.obs, torch.cat((st.info["traj"], ps.qpos, ps.qvel), 1))
    assert H.scaled_err(raw["obs"].numpy(), ost["obs"]) < 1e-12 and H.scaled_err(st.info["traj"].numpy(), ost["traj"]) < 1e-11
    assert sorted(st.metrics) == sorted(("rcom", "rvel", "rtrunk", "rquat", "ract", "termination_error"))
    rng = np.random.default_rng(0)
    c = env.clip_arrays(0)
    for step in range(3):
        old = {k: getattr(ps, k).clone() for k in ("qpos", "qvel", "subtree_com_root")}
        old_frame = st.info["cur_frame"].clone()
        ost = P.oracle_state_from(env, o, st)
        act = np.clip(0.5 * rng.standard_normal((B, 8)), -1, 1)
        st = env.step(st, torch.from_numpy(act))
        o.env_step(ost, act)
        assert H.scaled_err(ps.qpos.numpy(), ost["qpos"]) < 1e-9 and H.scaled_err(ps.qvel.numpy(), ost["qvel"]) < 1e-8
        # (the reward weights and the termination threshold are float32 in the C-ABI: 0.05f != 0.05, 0.9f != 0.9)
        assert np.abs(raw["metrics"].numpy() - ost["metrics"]).max() < 1e-8 and np.abs(st.reward.numpy() - ost["reward"]).max() < 1e-7
        assert np.array_equal(st.done.numpy(), ost["done"]) and H.scaled_err(st.info["traj"].numpy(), ost["traj"]) < 1e-8
        assert torch.equal(st.obs, torch.cat((st.info["traj"], ps.qpos, ps.qvel), 1))
        # --- the statements of ant.py, directly -------------------------------------------------------------------
        mt = {k: v.numpy() for k, v in st.metrics.items()}
        f = old_frame.numpy()
        # every term from the state BEFORE the step (ant.py:180), unweighted in the metrics (ant.py:216-225)
        ref = np.concatenate([c["velocity"][f], c["angular_velocity"][f], c["joints_velocity"][f]], 1)
        assert np.abs(mt["rvel"] - np.exp(-0.1 * np.linalg.norm(old["qvel"].numpy() - ref, axis=1))).max() < 1e-12
        assert np.abs(mt["rcom"] - np.exp(-100 * np.linalg.norm(old["subtree_com_root"].numpy() - c["center_of_mass"][f], axis=1))).max() < 1e-9
        # ract from the action (ant.py:277)
        assert np.abs(mt["ract"] - 0.01 * -0.015 * (act ** 2).sum(1) / 8).max() < 1e-15
        # total (ant.py:182-188), termination_error = rtrunk unweighted (ant.py:197), done = rtrunk < 0 | unhealthy (:200-201)
        total = 0.05 * mt["rcom"] + 0.01 * mt["rvel"] + 0.20 * mt["rtrunk"] + 0.01 * mt["rquat"] + 0.001 * mt["ract"]
        assert np.abs(st.reward.numpy() - total).max() < 1e-7
        assert np.array_equal(mt["termination_error"], mt["rtrunk"]) and np.array_equal(st.info["termination_error"].numpy(), mt["rtrunk"])
        z = old["qpos"][:, 2].numpy()
        assert np.array_equal(st.done.numpy() >= 1, (mt["rtrunk"] < 0) | (z < 0.2) | (z > 1.0))
        # the trajectory features in the observation start at the OLD frame + 1 (ant.py:178: the un-incremented info)
        want = c["joints"][np.clip(f[:, None] + 1 + np.arange(5), 0, 59)] - ps.qpos[:, 7:].numpy()[:, None, :]
        assert np.abs(st.info["traj"].numpy()[:, -40:] - want.reshape(B, -1)).max() < 1e-9
        assert (st.info["cur_frame"].numpy() == f + 1).all()


@pytest.mark.gpu
def test_ant_env_on_gpu():
    B = 256
    env = _env(B, device="cuda:0")
    rng = np.random.default_rng(3)
    sf, noise = np.zeros(B, np.int32), np.zeros((B, 15), np.float32)
    act = np.clip(0.5 * rng.standard_normal((B, 8)), -1, 1).astype(np.float32)
    o64, o32 = _oracle(env, "f64"), _oracle(env, "f32")
    st0 = env.reset()
    ps = st0.pipeline_state
    old = [P.to_np(x).astype(np.float64).copy() for x in (ps.qpos, ps.xpos, ps.qvel, ps.subtree_com_root, ps.qfrc_actuator)]
    st, err, dev, rep, ost = P.control_step_follow(env, o64, o32, sf, noise, act)
    print("\n[ant env control step, 256 envs] " + ", ".join(f"{k}: max {v.max():.2e} median {np.median(v):.2e}" for k, v in err.items()))
    P.check_control_step(err, dev, rep, max_flipped=B // 8)  # (feet at their contact margin: tests/test_generic_model.py)
    causes = P.flip_causes(rep)  # .. and every flipped env beyond the rodent's 1 % must be a contact row on its switching point
    print(f"   flipped envs {len(causes['envs'])}: contact-row presence {int(causes['row_presence'].sum())}, active set at a trial "
          f"step {int(causes['active_set'].sum())}, other {int(causes['other'].sum())}")
    assert int(causes["other"].sum()) <= max(2, B // 50), causes
    print("   vs the natural oracle:", P.natural_check(st, o64, o32, act))  # decision-independent companion
    # glue on the product's own post-step state
    gerr, gdev, flags = P.glue_errors(env, o64, o32, st, old[0], old[1], np.zeros(B, np.int32), np.zeros(B, np.int32),
                                      old_extra=old[2:] + [act.astype(np.float64)])
    print("[ant env glue] " + ", ".join(f"{k} {v.max():.2e}" for k, v in gerr.items()))
    assert flags["done_equal"] and flags["frames_equal"]
    for name, v in gerr.items():
        assert (v <= np.maximum(1e-5, 10 * gdev[name])).all(), (name, v.max(), gdev[name].max())
    assert torch.equal(st.obs, torch.cat((st.info["traj"], st.pipeline_state.qpos, st.pipeline_state.qvel), 1))


def _config0_rollout(device, steps=250, B=64, seed=0):
    """BASELINE.json configs[0]: envs/ant.py random-action rollout, 64 envs (actions clip(0.3 N(0, I), -1, 1),
    seed 0, 250 control steps; pass = runs, finite, deterministic)."""
    from vnl_brax_imitation_amd.envs.wrappers import wrap

    env = _env(B, device=device, params=NEWTON)  # the reference's ant solver: Newton, 1 iteration, 4 line-search iterations
    w = wrap(env, episode_length=20)
    st = w.reset(seed)
    g = torch.Generator().manual_seed(seed)
    rew, dones = [], 0.0
    for _ in range(steps):
        a = torch.clamp(0.3 * torch.randn((B, 8), generator=g), -1.0, 1.0).to(env.device)
        st = w.step(st, a)
        rew.append(st.reward.detach().cpu().clone())
        dones += float(st.done.sum())
    assert torch.isfinite(st.obs).all() and torch.isfinite(torch.stack(rew)).all()
    return torch.stack(rew), st.obs.detach().cpu().clone(), dones


def test_config0_ant_random_action_rollout_runs_finite_and_deterministic():
    r1, o1, d1 = _config0_rollout("cpu")
    r2, o2, d2 = _config0_rollout("cpu")
    assert torch.equal(r1, r2) and torch.equal(o1, o2) and d1 == d2  # deterministic
    assert d1 > 0 and float(r1.abs().max()) > 0  # episodes end (20-step truncation / falls) and rewards are produced
    assert o1.shape == (64, 5 * (2 * 10 * 3 + 3 + 8) + 15 + 14)


@pytest.mark.gpu
def test_config0_ant_random_action_rollout_on_gpu():
    r1, o1, d1 = _config0_rollout("cuda:0")
    r2, o2, d2 = _config0_rollout("cuda:0")
    assert torch.equal(r1, r2) and torch.equal(o1, o2) and d1 == d2
    assert d1 > 0


def _newton_inputs(B, seed=4):
    rng = np.random.default_rng(seed)
    return [np.clip(0.5 * rng.standard_normal((B, 8)), -1, 1) for _ in range(4)]


def test_newton_oracle_solves_the_hessian_system_and_differs_from_cg():
    """The oracle's Newton route on its own terms (solver.py _update_gradient, NEWTON): after a forward pass with one
    iteration, Mgrad-based search direction = -H^-1 grad with H = qM + J' diag(D active) J, checked by a dense NumPy solve of
    the same system at the START point, and the step taken differs from the CG step (a different algorithm, not a no-op)."""
    B = 1
    en, ec = _env(B, "double", params=NEWTON), _env(B, "double", params=dict(solver="cg", iterations=1, ls_iterations=4))
    on, oc = _oracle(en), _oracle(ec)
    acts = _newton_inputs(B)
    sn, sc = on.env_reset(np.zeros(B, np.int32), np.zeros((B, 15))), oc.env_reset(np.zeros(B, np.int32), np.zeros((B, 15)))
    for a in acts:
        on.env_step(sn, a), oc.env_step(sc, a)
    assert np.isfinite(sn["qpos"]).all()
    assert np.abs(sn["qvel"] - sc["qvel"]).max() > 1e-6  # Newton and CG with 1 iteration take different steps
    # dense check of one Newton solve: set the oracle's single-env data to the last state and run forward
    on.set(qpos=sn["qpos"][0], qvel=sn["qvel"][0], act=np.zeros(8), ctrl=acts[-1][0], qacc_warmstart=sn["qacc_warmstart"][0])
    on.call("forward")
    nv, ne = 14, int(en.sys.scalars["nefc"])
    M, J = on.field("qM").reshape(nv, nv), on.field("efc_J").reshape(ne, nv)
    D, aref = on.field("efc_D"), on.field("efc_aref")
    fs, a0, aw = on.field("qfrc_smooth"), on.field("qacc_smooth"), sn["qacc_warmstart"][0]

    def cost(a):
        r = J @ a - aref
        return 0.5 * np.sum(D * r * r * (r < 0)) + 0.5 * (M @ a - fs) @ (a - a0)

    start = aw if cost(aw) < cost(a0) else a0
    r = J @ start - aref
    act_ = (r < 0) & (D != 0)
    grad = M @ start - fs - J.T @ (D * -r * act_)
    Hm = M + (J.T * (D * act_)) @ J
    search = -np.linalg.solve(Hm, grad)
    step = on.field("qacc") - start  # one iteration: qacc = start + alpha * search
    if np.abs(step).max() > 0:
        cosang = step @ search / (np.linalg.norm(step) * np.linalg.norm(search))
        assert cosang > 1 - 1e-9, cosang  # the step is along the Newton direction
    assert cost(on.field("qacc")) <= cost(start) + 1e-12


def test_ant_env_newton_float64_build_matches_oracle():
    """BASELINE configs[0]'s solver (Newton, 1 iteration, 4 line-search iterations; reference envs/ant.py:40-47 reading
    configs/env_config.yaml:16-21): the product source compiled for the host in float64 against the dense oracle."""
    B = 6
    env = _env(B, "double", params=NEWTON)
    assert int(env.sys.scalars["solver_newton"]) == 1 and int(env.sys.scalars["iterations"]) == 1
    st = env.reset()
    o = _oracle(env)
    ost = o.env_reset(np.zeros(B, np.int32), np.zeros((B, 15)))
    ps = st.pipeline_state
    for k in ("qpos", "qvel", "xpos", "qacc_warmstart"):
        assert H.scaled_err(getattr(ps, k).reshape(B, -1).numpy(), ost[k]) < 1e-10, k
    for act in _newton_inputs(B):
        st = env.step(st, torch.from_numpy(act))
        o.env_step(ost, act)
        assert H.scaled_err(ps.qpos.numpy(), ost["qpos"]) < 1e-9 and H.scaled_err(ps.qvel.numpy(), ost["qvel"]) < 1e-8
        assert H.scaled_err(ps.qacc_warmstart.numpy(), ost["qacc_warmstart"]) < 1e-7
        assert np.abs(st.reward.numpy() - ost["reward"]).max() < 1e-7 and np.array_equal(st.done.numpy(), ost["done"])
    assert float(np.abs(ps.qvel.numpy()).max()) > 1e-2


@pytest.mark.gpu
def test_ant_env_newton_on_gpu():
    """The Newton branch of EnvWave::solve on the device: one control step against the oracles following the product's
    line-search decisions (same bounds as the CG test above) and three more steps finite and deterministic."""
    B = 256
    env = _env(B, device="cuda:0", params=NEWTON)
    rng = np.random.default_rng(3)
    sf, noise = np.zeros(B, np.int32), np.zeros((B, 15), np.float32)
    act = np.clip(0.5 * rng.standard_normal((B, 8)), -1, 1).astype(np.float32)
    o64, o32 = _oracle(env, "f64"), _oracle(env, "f32")
    st, err, dev, rep, ost = P.control_step_follow(env, o64, o32, sf, noise, act)
    print("\n[ant env, Newton 1/4, control step, 256 envs] " + ", ".join(f"{k}: max {v.max():.2e} median {np.median(v):.2e}" for k, v in err.items()))
    P.check_control_step(err, dev, rep, max_flipped=B // 8)
    causes = P.flip_causes(rep)
    assert int(causes["other"].sum()) <= max(2, B // 50), causes
    print("   vs the natural oracle:", P.natural_check(st, o64, o32, act))
    outs = []
    for _ in range(2):
        s = env.reset()
        for k in range(3):
            s = env.step(s, torch.from_numpy(act))
        outs.append(s.pipeline_state.qvel.clone())
    assert torch.isfinite(outs[0]).all() and torch.equal(outs[0], outs[1])
