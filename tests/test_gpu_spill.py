"""Results must not depend on register spilling.

Round 1 saw a build of the step kernel that had slipped to 256 VGPRs + scratch give wrong results on the GPU and
answered by refusing such builds.  A well-defined program computes the same thing whether or not the compiler spills,
so this test compiles the SAME sources under a 128-VGPR cap (csrc/build.py --spill: hundreds of bytes of scratch and ~130-230
spilled registers per lane in the step kernel) and holds that build to the same oracle bounds as the product build,
and to agreement with the product build itself."""
import numpy as np
import pytest
import torch

import helpers as H
import parity as P

pytestmark = pytest.mark.gpu


def _env(B, lib=None):
    from vnl_brax_imitation_amd.envs.rodent import RodentTracking

    with H.backend(lib):
        return RodentTracking(H.reference_clip(), num_envs=B, device="cuda:0", **H.env_kwargs())


def test_spilling_build_matches_oracle_and_product_build():
    from vnl_brax_imitation_amd import _lib
    from vnl_brax_imitation_amd.csrc import build as hip_build

    path = hip_build.build(variant="spill")
    res = dict(l.split(" ", 1) for l in open(path + ".resources.txt").read().splitlines())
    step = next(v for k, v in res.items() if "vnl_step_kernel" in k)
    assert "ScratchSize [bytes/lane]=0" not in step, "the regression build no longer spills: lower its VGPR cap"
    B = 256
    rng = np.random.default_rng(21)
    sf = rng.integers(0, 235, B).astype(np.int32)
    noise = (1e-3 * rng.standard_normal((B, 74))).astype(np.float32)
    acts = np.clip(0.3 * rng.standard_normal((3, B, 30)), -1, 1).astype(np.float32)
    spill = _env(B, _lib.load_library(path))
    o64, o32 = H.make_oracle(spill, "f64"), H.make_oracle(spill, "f32")
    st, err, dev, rep, ost = P.control_step_follow(spill, o64, o32, sf, noise, acts[0])
    print("\n[spill build, control step] " + ", ".join(f"{k} max {v.max():.2e}" for k, v in err.items()))
    P.check_control_step(err, dev, rep)
    # three steps on both builds: same instructions modulo spill code, so the outputs should agree bit for bit; a
    # difference beyond float32 rounding of one step would be a spill-dependent result
    prod = _env(B)
    outs = []
    for env in (spill, prod):
        s = env.reset(start_frame=torch.from_numpy(sf), noise=torch.from_numpy(noise))
        for a in acts:
            s = env.step(s, torch.from_numpy(a))
        ps = s.pipeline_state
        outs.append({k: getattr(ps, k).cpu().numpy().copy() for k in ("qpos", "qvel", "qacc_warmstart")} |
                    {"obs": s.obs.cpu().numpy().copy(), "reward": s.reward.cpu().numpy().copy()})
    same = {k: bool(np.array_equal(outs[0][k], outs[1][k])) for k in outs[0]}
    worst = {k: float(P.per_env_scaled(outs[0][k], outs[1][k].astype(np.float64)).max()) for k in outs[0]}
    print("[spill vs product build, 3 steps] bitwise equal:", same, "worst scaled difference:", worst)
    assert all(same.values()), (same, worst)  # bit for bit, as DESIGN section 2 item 3 states


def test_blocked_products_agree_with_the_lane_per_row_products():
    """The products with the factor and its inverse (M^-1 x, M x) run BLOCKED on the device: rows / columns cut into blocks of
    eight entries dealt out over the lanes, partial sums combined by DPP shifts (EnvWave::blk_apply).  csrc/build.py --noblk
    builds the same sources with one lane per row / column: the same sums in another order.  One substep from identical
    states may differ by the rounding of those sums and what six CG iterations make of it -- nothing more: the median env
    within 1e-6 of scale on the velocities, and no more envs beyond 1e-5 than the product has against the float64 oracle."""
    from vnl_brax_imitation_amd import _lib
    from vnl_brax_imitation_amd.csrc import build as hip_build
    from vnl_brax_imitation_amd.envs.rodent import RodentTracking

    path = hip_build.build(variant="noblk")
    B = 1024
    rng = np.random.default_rng(33)
    sf = rng.integers(0, 235, B).astype(np.int32)
    noise = (1e-3 * rng.standard_normal((B, 74))).astype(np.float32)
    act = np.clip(0.3 * rng.standard_normal((B, 30)), -1, 1).astype(np.float32)
    kw = dict(H.env_kwargs(), n_frames=1)
    outs = []
    for lib in (_lib.load_library(path), None):
        with H.backend(lib):
            env = RodentTracking(H.reference_clip(), num_envs=B, device="cuda:0", **kw)
        s = env.reset(start_frame=torch.from_numpy(sf), noise=torch.from_numpy(noise))
        ps0 = {k: getattr(s.pipeline_state, k).cpu().numpy().copy() for k in ("qpos", "qvel", "qacc_warmstart")}
        s = env.step(s, torch.from_numpy(act))
        outs.append((ps0, {k: getattr(s.pipeline_state, k).cpu().numpy().astype(np.float64) for k in ("qpos", "qvel", "qacc_warmstart")}))
    # (the reset runs a forward pass with the products in it: the two builds start the substep from states that agree to rounding)
    for k in ("qpos", "qvel"):
        assert P.per_env_grouped(outs[0][0][k].astype(np.float64), outs[1][0][k].astype(np.float64), k).max() < 1e-6, k
    diff = {k: P.per_env_grouped(outs[0][1][k], outs[1][1][k], k) for k in ("qpos", "qvel", "qacc_warmstart")}
    print("\n[blocked vs lane-per-row products, one substep, 1024 envs] " +
          ", ".join(f"{k}: median {np.median(v):.2e} max {v.max():.2e} over 1e-5: {(v > 1e-5).sum()}" for k, v in diff.items()))
    assert np.median(diff["qvel"]) < 1e-6 and np.median(diff["qpos"]) < 1e-6
    assert (diff["qvel"] > 1e-5).mean() < 0.08  # (the product against the float64 oracle: 3-4 % of envs, bench.py compliance)


def test_specialised_kernels_equal_the_generic_kernels_bit_for_bit():
    """The rodent runs kernels SPECIALISED at compile time for its dimensions and LDS layout (csrc/vnl_types.h VnlSpecRodent:
    loop bounds fold, offsets become instruction immediates); csrc/build.py --nospec builds the same sources with the
    specialisation switched off, so that the generic kernels (every dimension read from the constant block: what any other
    model runs) take the rodent too.  Same operations on the same operands in the same order: bit-identical outputs."""
    from vnl_brax_imitation_amd import _lib
    from vnl_brax_imitation_amd.csrc import build as hip_build

    path = hip_build.build(variant="nospec")
    B = 512
    # the product runs the specialised kernels on the rodent (a mismatch of the compile-time constants would fall back to the
    # generic ones silently), the regression build the generic ones
    assert int(_env(4).dims.kernel_specialised) == 1 and int(_env(4, _lib.load_library(path)).dims.kernel_specialised) == 0
    rng = np.random.default_rng(44)
    sf = rng.integers(0, 235, B).astype(np.int32)
    noise = (1e-3 * rng.standard_normal((B, 74))).astype(np.float32)
    acts = np.clip(0.3 * rng.standard_normal((3, B, 30)), -1, 1).astype(np.float32)
    outs = []
    for env in (_env(B, _lib.load_library(path)), _env(B)):
        s = env.reset(start_frame=torch.from_numpy(sf), noise=torch.from_numpy(noise))
        for a in acts:
            s = env.step(s, torch.from_numpy(a))
        ps = s.pipeline_state
        outs.append({k: getattr(ps, k).cpu().numpy().copy() for k in ("qpos", "qvel", "qacc_warmstart", "xpos")} |
                    {"obs": s.obs.cpu().numpy().copy(), "reward": s.reward.cpu().numpy().copy(),
                     "traj": s.info["traj"].cpu().numpy().copy()})
    same = {k: bool(np.array_equal(outs[0][k], outs[1][k])) for k in outs[0]}
    worst = {k: float(np.abs(outs[0][k].astype(np.float64) - outs[1][k]).max()) for k in outs[0]}
    print("\n[specialised vs generic kernels, 3 steps, 512 envs] bitwise equal:", same, "largest difference:", worst)
    assert all(same.values()), (same, worst)
