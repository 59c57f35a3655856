"""brax training-wrapper semantics on the host simulation."""
import torch

import helpers as H
from vnl_brax_imitation_amd.envs.wrappers import EpisodeWrapper, EvalWrapper, wrap


def test_autoreset_restores_pipeline_state_but_not_info():
    env = wrap(H.hostsim_env(3, healthy_z_range=(0.0, 0.5)), episode_length=150)
    st = env.reset(1)
    first_q = st.info["first_pipeline_state"].qpos.clone()
    dones = []
    for _ in range(12):
        st = env.step(st, torch.zeros(3, 30))
        dones.append(st.done.clone())
    assert all(d.sum() == 0 for d in dones[:9]) and all(d.sum() == 3 for d in dones[9:])
    assert torch.equal(st.pipeline_state.qpos, first_q)  # restored on done
    assert torch.equal(st.obs[:, :74], first_q)
    assert (st.info["sub_clip_frame"] == 12).all()  # info NOT reset (C.20)
    assert (st.info["steps"] == 1).all()  # zeroed on the call after done, then +1


def test_autoreset_with_info_fix():
    env = wrap(H.hostsim_env(2, healthy_z_range=(0.0, 0.5)), episode_length=150, reset_info_on_autoreset=True)
    st = env.reset(1)
    f0 = st.info["cur_frame"].clone()
    for _ in range(10):
        st = env.step(st, torch.zeros(2, 30))
    assert (st.done == 1).all() and torch.equal(st.info["cur_frame"], f0) and (st.info["sub_clip_frame"] == 0).all()
    st = env.step(st, torch.zeros(2, 30))
    assert (st.done == 0).all()


def test_episode_truncation():
    env = EpisodeWrapper(H.hostsim_env(2, sub_clip_length=100), episode_length=4, action_repeat=1)
    st = env.reset(2)
    for i in range(4):
        st = env.step(st, torch.zeros(2, 30))
    assert (st.done == 1).all() and (st.info["truncation"] == 1).all() and (st.info["steps"] == 4).all()


def test_eval_wrapper_accumulates_first_episode_only():
    env = EvalWrapper(wrap(H.hostsim_env(2, healthy_z_range=(0.0, 0.5)), episode_length=150))
    st = env.reset(3)
    total = torch.zeros(2)
    for i in range(14):
        st = env.step(st, torch.zeros(2, 30))
        if i < 10:
            total += st.reward
    em = st.info["eval_metrics"]
    assert torch.allclose(em.episode_metrics["reward"], total, atol=1e-6)
    assert (em.active_episodes == 0).all() and (em.episode_steps == 10).all()
    assert set(em.episode_metrics) == {"rcom", "rvel", "rtrunk", "rquat", "ract", "rapp", "termination_error", "reward"}
