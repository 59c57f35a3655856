"""Kernel LOGIC vs oracle on CPU (no GPU needed).

The product source csrc/vnl_lib.hip is compiled by g++ against tests/hostsim/stub (kernel
launch = serial loop).  Two builds:
  * -DVNL_REAL=double : same algorithms in float64 -> must agree with the dense float64
    oracle far below float32 rounding.  This is the algorithm-equivalence gate: the kernels
    use tree-sparse L'DL, matrix-free J*v / J'*f / M*v and a different spatial reference
    point; the oracle is dense and literal.
  * float (product arithmetic): agreement within the float32 sensitivity of the dynamics.
The same comparisons run against the real HIP build in tests/test_gpu_parity.py.
"""
import numpy as np
import pytest
import torch

import helpers as H

KEYS = ("qpos", "qvel", "act", "qacc_warmstart", "xpos", "qfrc_actuator")


def _inputs(B, seed=0):
    rng = np.random.default_rng(seed)
    sf = rng.integers(0, 235, B).astype(np.int32)
    noise = 1e-3 * rng.standard_normal((B, 74))
    acts = np.clip(0.3 * rng.standard_normal((3, B, 30)), -1, 1)
    return sf, noise, acts


def _cmp(st, ost, B):
    out = {k: H.scaled_err(getattr(st.pipeline_state, k).reshape(B, -1).numpy(), ost[k]) for k in KEYS}
    out["obs"] = H.scaled_err(st.obs.numpy(), ost["obs"])
    out["traj"] = H.scaled_err(st.info["traj"].numpy(), ost["traj"])
    out["reward"] = H.scaled_err(st.reward.numpy(), ost["reward"]) if np.abs(ost["reward"]).max() > 0 else 0.0
    out["com"] = H.scaled_err(st.pipeline_state.subtree_com_root.numpy(), ost["com1"])
    return out


def test_float64_build_is_algorithmically_identical_to_oracle():
    B = 16
    env = H.hostsim_env(B, "double")
    sf, noise, acts = _inputs(B)
    st = env.reset(start_frame=torch.from_numpy(sf), noise=torch.from_numpy(noise))
    o = H.make_oracle(env, "f64")
    ost = o.env_reset(sf, noise)
    e = _cmp(st, ost, B)
    assert max(e.values()) < 1e-11, e
    assert H.scaled_err(st.info["termination_error"].numpy(), ost["termination_error"]) < 1e-12
    st = env.step(st, torch.from_numpy(acts[0]))
    o.env_step(ost, acts[0])
    e = _cmp(st, ost, B)
    assert max(e.values()) < 1e-8, e  # 5 substeps of chaotic amplification of ~1e-15
    assert np.array_equal(st.done.numpy(), ost["done"])
    m = np.stack([st.metrics[k].numpy() for k in st.metrics], 1)
    assert np.abs(m - ost["metrics"]).max() < 1e-10


def test_float64_stage_outputs_match_dense_oracle():
    """Bisection hook (vnl_env_scratch): sparse qM, qacc_smooth, constraint rows, contacts, qacc."""
    env = H.hostsim_env(1, "double")
    env.debug(True)
    sf, noise, _ = _inputs(1, seed=7)
    env.reset(start_frame=torch.from_numpy(sf), noise=torch.from_numpy(noise))
    o = H.make_oracle(env, "f64")
    c = env.clip_arrays(0)
    f = int(sf[0])
    o.set(qpos=np.concatenate([c["position"][f], c["quaternion"][f], c["joints"][f]]) + noise[0],
          qvel=np.concatenate([c["velocity"][f], c["angular_velocity"][f], c["joints_velocity"][f]]),
          act=np.zeros(30), ctrl=np.zeros(30), qacc_warmstart=np.zeros(73))
    o.call("forward")
    m = env.sys
    par = m.dof_parentid
    # the kernel keeps only the in-place, INVERTED factor of the tree-sparse qM (N = L^-1, D^-1):
    # rebuild M^-1 = N D^-1 N' ... i.e. M = (N^-1)' D (N^-1), and compare with the oracle's dense qM
    Md = o.field("qM").reshape(73, 73)
    LD, dinv = env.scratch("qLD")[0].numpy(), env.scratch("qLDiagInv")[0].numpy()
    Lm, k = np.eye(73), 0
    for i in range(73):
        j = i
        while j >= 0:
            if j != i:
                Lm[i, j] = LD[k]
            k += 1
            j = par[j]
    assert k == len(LD) == 1119
    Minv = Lm @ np.diag(dinv) @ Lm.T  # Lm holds N = L^-1 (unit lower, ancestor pattern)
    assert np.abs(Minv @ Md - np.eye(73)).max() < 1e-9
    for name, ref in [("qfrc_smooth", "qfrc_smooth"), ("qacc_smooth", "qacc_smooth"), ("qacc", "qacc"),
                      ("qfrc_constraint", "qfrc_constraint")]:
        assert H.scaled_err(env.scratch(name)[0].numpy(), o.field(ref)) < 1e-10, name
    D, Dref = np.abs(env.scratch("efc_D")[0].numpy()), o.field("efc_D")  # limit rows carry the Jacobian sign
    present = np.abs(o.field("efc_J").reshape(303, 73)).sum(1) > 0
    assert np.array_equal(D != 0, present)
    assert H.scaled_err(D[present], Dref[present]) < 1e-12
    # after the solve Jaref = J qacc - aref on the present rows
    Jaref = o.field("efc_J").reshape(303, 73) @ o.field("qacc") - o.field("efc_aref")
    assert H.scaled_err(env.scratch("Jaref")[0].numpy()[present], Jaref[present]) < 1e-9


def test_float32_build_within_float32_sensitivity():
    B = 32
    env = H.hostsim_env(B, "float")
    sf, noise, acts = _inputs(B, seed=2)
    noise = noise.astype(np.float32)
    acts = acts.astype(np.float32)
    st = env.reset(start_frame=torch.from_numpy(sf), noise=torch.from_numpy(noise))
    o = H.make_oracle(env, "f64")
    ost = o.env_reset(sf, noise)
    e = _cmp(st, ost, B)
    assert e["qacc_warmstart"] < 2e-5 and max(v for k, v in e.items() if k != "qacc_warmstart") < 5e-6, e
    st = env.step(st, torch.from_numpy(acts[0]))
    o.env_step(ost, acts[0])
    qv = st.pipeline_state.qvel.numpy()
    per_env = np.array([H.scaled_err(qv[i], ost["qvel"][i]) for i in range(B)])
    assert np.median(per_env) < 1e-4 and np.quantile(per_env, 0.9) < 1e-2, per_env
    # the float32 ORACLE deviates from the float64 oracle by the same order: the spread is the
    # dynamics' sensitivity, not an implementation difference
    o32 = H.make_oracle(env, "f32")
    ost32 = o32.env_reset(sf, noise)
    o32.env_step(ost32, acts[0])
    per_env32 = np.array([H.scaled_err(ost32["qvel"][i], ost["qvel"][i]) for i in range(B)])
    assert np.median(per_env) < 10 * np.median(per_env32) + 1e-6


def test_step_in_place_and_frame_counters():
    env = H.hostsim_env(4)
    st = env.reset(5)
    q0 = st.pipeline_state.qpos.clone()
    st2 = env.step(st, torch.zeros(4, 30))
    assert st2 is st and not torch.equal(q0, st.pipeline_state.qpos)
    assert (st.info["sub_clip_frame"] == 1).all()
    assert st.obs.shape == (4, 232) and st.info["traj"].shape == (4, 795)
    assert torch.equal(st.obs[:, :74], st.pipeline_state.qpos)


def test_multi_clip_container_selects_per_env_clip():
    from vnl_brax_imitation_amd.preprocessing.mjx_preprocess import ReferenceClip

    c0 = H.reference_clip()
    shift = np.array([0.05, -0.02, 0.0], dtype=np.float32)
    c1 = c0.replace(position=c0.position + shift, body_positions=c0.body_positions + shift)
    multi = ReferenceClip.stack([c0, c1])
    env = H.hostsim_env(2, reference_clip=multi)
    sf = torch.tensor([10, 10], dtype=torch.int32)
    st = env.reset(start_frame=sf, noise=torch.zeros(2, 74), clip_id=torch.tensor([0, 1], dtype=torch.int32))
    d = (st.pipeline_state.qpos[1, :3] - st.pipeline_state.qpos[0, :3]).numpy()
    assert np.allclose(d, shift, atol=1e-7)
    # a pure translation of body + reference leaves the egocentric features unchanged
    assert torch.allclose(st.info["traj"][0, 75:75 + 270], st.info["traj"][1, 75:75 + 270], atol=1e-6)


def test_welded_bodies_are_folded_into_their_parents_and_keep_their_poses():
    """The rodent's 13 jointless bodies move rigidly with their parents; the library folds them into those for the dynamics
    (csrc/vnl_lib.hip: fuse_welded_bodies -- merged mass / centre of mass / inertia, re-attached children and collision geoms)
    and still reports every body's pose.  The oracle works on the model as given: after two control steps the float64 host
    build must agree with it on ALL 66 rows of xpos, and in particular on the rows of the welded bodies."""
    B = 8
    env = H.hostsim_env(B, "double")
    m = env.sys
    nb = int(m.scalars["nbody"])
    assert int(env.dims.nbody) == nb == 66 and int(env.dims.nbody_dynamic) == 53
    welded = [b for b in range(2, nb) if int(m.arrays["body_jntnum"][b]) == 0]
    assert len(welded) == nb - int(env.dims.nbody_dynamic) == 13
    sf, noise, acts = _inputs(B)
    st = env.reset(start_frame=torch.from_numpy(sf), noise=torch.from_numpy(noise))
    o = H.make_oracle(env, "f64")
    ost = o.env_reset(sf, noise)
    for k in range(2):
        st = env.step(st, torch.from_numpy(acts[k]))
        o.env_step(ost, acts[k])
    xp = st.pipeline_state.xpos.numpy().reshape(B, nb, 3)
    ox = ost["xpos"].reshape(B, nb, 3)
    assert np.abs(xp - ox).max() < 1e-8  # (two control steps of chaotic amplification of ~1e-15)
    assert np.abs(xp[:, welded] - ox[:, welded]).max() < 1e-8
    # a welded body really is offset from its parent (the rows are not copies of the parents' rows)
    par = np.asarray(m.arrays["body_parentid"])[welded]
    assert np.abs(xp[:, welded] - xp[:, par]).max() > 1e-3
