"""The kernels are model-generic: the reference's assets/ant.xml (14 dofs, 4 leaf chains, motor actuators, sphere
feet) through the same source, against the dense oracle in float64.  Covers the code paths the rodent does not take
(single lane set, the LDS factorisation fallback for models whose CG vectors are too small to hold the scratch
lines).  BASELINE.json config 0 names this model; the reference's ant clip is not shipped, so the reference
trajectory here is synthetic (forward kinematics of a smooth qpos sequence)."""
import os

import numpy as np
import pytest
import torch

import helpers as H
from vnl_brax_imitation_amd.model import mjcf
from vnl_brax_imitation_amd.preprocessing import mjx_preprocess as P

ANT = "/root/reference/assets/ant.xml"
ANT_COMPILED = os.path.join(H.ROOT, "vnl-brax-imitation_amd", "data", "ant.npz")  # numeric constants of the compiled model


def _ant_model():
    return mjcf.CompiledModel.load(ANT_COMPILED)


@pytest.mark.skipif(not os.path.exists(ANT), reason="reference checkout not present (GPU box)")
def test_packaged_ant_model_is_the_compiled_reference_xml():
    a, b = mjcf.compile_mjcf(ANT, scale_factor=1.0), _ant_model()
    assert a.names == b.names and a.scalars.keys() == b.scalars.keys()
    for k, v in a.arrays.items():
        assert np.array_equal(np.asarray(v), np.asarray(b.arrays[k])), k


def _ant_env(B, real, device="cpu"):
    from vnl_brax_imitation_amd.envs.rodent import RodentTracking

    m = _ant_model()
    T = 40
    t = np.arange(T)[:, None] * 0.02
    qpos = np.zeros((T, 15))
    qpos[:, 2], qpos[:, 3] = 0.55, 1.0
    qpos[:, 0] = 0.2 * t[:, 0]
    base = np.array([0.0, 1.0, 0.0, -1.0, 0.0, -1.0, 0.0, 1.0])
    qpos[:, 7:] = base + 0.15 * np.sin(2 * np.pi * 1.5 * t + np.arange(8))
    clip = P.process_qpos(m, qpos, max_qvel=20.0, dt=0.02)
    names = m.names
    import contextlib

    with (H.hostsim_backend(real) if device == "cpu" else contextlib.nullcontext()):
        env = RodentTracking(
            clip, end_eff_names=["aux_1", "aux_2", "aux_3", "aux_4"], appendage_names=["aux_1", "aux_2", "aux_3", "aux_4", "torso"],
            walker_body_names=[n for n in names["body"] if n != "world"], joint_names=names["joint"][1:],
            center_of_mass="torso", model=m, clip_length=T, sub_clip_length=10, ref_traj_length=5, healthy_z_range=(0.2, 1.0),
            num_envs=B, device=device)
    return env


def test_ant_model_matches_dense_oracle_float64():
    B = 8
    env = _ant_env(B, "double")
    d = env.dims
    assert (d.nq, d.nv, d.nu, d.nbody) == (15, 14, 8, 14)
    rng = np.random.default_rng(0)
    sf = rng.integers(0, 30, B).astype(np.int32)
    noise = 1e-3 * rng.standard_normal((B, 15))
    st = env.reset(start_frame=torch.from_numpy(sf), noise=torch.from_numpy(noise))
    o = H.make_oracle(env, "f64")
    ost = o.env_reset(sf, noise)
    ps = st.pipeline_state
    for k in ("qpos", "qvel", "xpos", "qacc_warmstart"):
        assert H.scaled_err(getattr(ps, k).reshape(B, -1).numpy(), ost[k]) < 1e-11, k
    assert H.scaled_err(st.obs.numpy(), ost["obs"]) < 1e-11 and H.scaled_err(st.info["traj"].numpy(), ost["traj"]) < 1e-11
    for _ in range(3):  # ground contact of the feet, joint limits, motor actuation
        act = np.clip(0.5 * rng.standard_normal((B, 8)), -1, 1)
        st = env.step(st, torch.from_numpy(act))
        o.env_step(ost, act)
    assert H.scaled_err(ps.qpos.numpy(), ost["qpos"]) < 1e-7 and H.scaled_err(ps.qvel.numpy(), ost["qvel"]) < 1e-6
    assert np.abs(st.reward.numpy() - ost["reward"]).max() < 1e-8
    assert np.array_equal(st.done.numpy(), ost["done"]) and np.array_equal(st.info["cur_frame"].numpy(), ost["cur_frame"])
    assert float(np.abs(ps.qvel.numpy()).max()) > 1e-2  # something moved


@pytest.mark.gpu
def test_ant_model_on_gpu_matches_oracle():
    """The single-lane-set instantiations and the LDS factorisation fallback on the real device."""
    B = 128
    env = _ant_env(B, "float", device="cuda:0")
    rng = np.random.default_rng(1)
    sf = rng.integers(0, 30, B).astype(np.int32)
    noise = (1e-3 * rng.standard_normal((B, 15))).astype(np.float32)
    st = env.reset(start_frame=torch.from_numpy(sf), noise=torch.from_numpy(noise))
    o = H.make_oracle(env, "f64")
    ost = o.env_reset(sf, noise)
    ps = st.pipeline_state
    for k, tol in (("qpos", 1e-6), ("xpos", 2e-6)):
        assert H.scaled_err(getattr(ps, k).reshape(B, -1).cpu().numpy(), ost[k]) < tol, k
    assert H.scaled_err(st.obs.cpu().numpy(), ost["obs"]) < 1e-6
    # decision-independent: the reset solve against the NATURAL oracle.  qacc passes through the 6-iteration constraint
    # solve and an env at a contact-activation threshold is sensitive to float32 rounding (the float32 host build shows
    # the same), so per-env quantiles rather than the maximum
    qa = ps.qacc_warmstart.cpu().numpy()
    per_env_a = np.array([H.scaled_err(qa[i], ost["qacc_warmstart"][i]) for i in range(B)])
    assert np.median(per_env_a) < 2e-6 and np.quantile(per_env_a, 0.9) < 2e-5, (np.median(per_env_a), per_env_a.max())
    # one control step against the oracles made to follow the product's solver decisions (tests/parity.py): every env
    # within max(1e-5 of the array's scale, 50 x the float32 oracle's own deviation on that env) -- no quantiles
    import parity as P

    act = np.clip(0.5 * rng.standard_normal((B, 8)), -1, 1).astype(np.float32)
    o32 = H.make_oracle(env, "f32")
    st, err, dev, rep, ost = P.control_step_follow(env, o, o32, sf, noise, act)
    print("\n[ant control step, 128 envs] " + ", ".join(f"{k}: max {v.max():.2e} median {np.median(v):.2e}" for k, v in err.items()))
    # the ant's four sphere feet rest AT their contact margin in the synthetic clip, so a contact row's presence is a
    # float32 tie far more often than for the rodent (9 of 128 envs measured on the device, 1 of 32 in the host build):
    # those envs must still show the flipped decision and stay within 1000 x their sensitivity (check_control_step)
    P.check_control_step(err, dev, rep, max_flipped=B // 8)
    # every flipped env beyond the rodent's allowance (1 %) must be explained by that mechanism -- a contact row on its
    # switching point: present on one side only (report [6]) or active on one side only at a trial step (report [3]; measured
    # on the device: all 10 flipped envs of this batch are of the second kind) -- not by a line-search or exit decision off
    # its tie with the same rows active on both sides
    causes = P.flip_causes(rep)
    print(f"   flipped envs {len(causes['envs'])}: contact-row presence {int(causes['row_presence'].sum())}, active set at a trial "
          f"step {int(causes['active_set'].sum())}, other {int(causes['other'].sum())}")
    assert int(causes["other"].sum()) <= max(2, B // 50), causes
    # and decision-independent: the product against the NATURAL float64 oracle over the control step, no worse than the
    # natural float32 oracle in distribution (a wrong decision of the product would be replayed by a following oracle)
    print("   vs the natural oracle:", P.natural_check(st, o, o32, act))
    assert np.array_equal(st.done.cpu().numpy(), ost["done"].astype(np.float32))
