"""Training wrappers with brax.envs.wrappers.training semantics [UPSTREAM], as applied
by the reference at ppo_imitation/train.py:204-214 (wrap_for_training) and
ppo_imitation/acting.py:109 (EvalWrapper).  The Vmap wrapper is not needed: the
envs are natively batched.

State buffers are updated in place by the kernels, so the wrappers keep their own
copies where brax relies on immutability (first_pipeline_state / first_obs).
"""
from __future__ import annotations

import dataclasses
from typing import Dict

import torch

from .base import Env, State


class Wrapper(Env):
    def __init__(self, env: Env):
        self.env = env

    def __getattr__(self, name):
        if name == "env":
            raise AttributeError(name)
        return getattr(self.env, name)

    @property
    def unwrapped(self) -> Env:
        return self.env.unwrapped

    @property
    def observation_size(self) -> int:
        return self.env.observation_size

    @property
    def action_size(self) -> int:
        return self.env.action_size

    def reset(self, rng=None, **kw) -> State:
        return self.env.reset(rng, **kw)

    def step(self, state: State, action: torch.Tensor) -> State:
        return self.env.step(state, action)


class EpisodeWrapper(Wrapper):
    """Maintains episode step count and sets done at episode end (brax EpisodeWrapper)."""

    def __init__(self, env: Env, episode_length: int, action_repeat: int):
        super().__init__(env)
        self.episode_length = int(episode_length)
        self.action_repeat = int(action_repeat)

    def reset(self, rng=None, **kw) -> State:
        state = self.env.reset(rng, **kw)
        state.info["steps"] = torch.zeros_like(state.done)
        state.info["truncation"] = torch.zeros_like(state.done)
        return state

    def step(self, state: State, action: torch.Tensor) -> State:
        if self.action_repeat == 1:
            state = self.env.step(state, action)
        else:
            total = torch.zeros_like(state.reward)
            for _ in range(self.action_repeat):
                state = self.env.step(state, action)
                total += state.reward
            state.reward.copy_(total)
        steps = state.info["steps"]
        steps += self.action_repeat
        over = steps >= self.episode_length
        state.info["truncation"].copy_(torch.where(over, 1.0 - state.done, torch.zeros_like(state.done)))
        state.done.copy_(torch.where(over, torch.ones_like(state.done), state.done))
        return state


class AutoResetWrapper(Wrapper):
    """Automatically resets done envs to the cached first state (brax AutoResetWrapper).

    As in brax, only `pipeline_state` and `obs` are restored; `info` (cur_frame,
    sub_clip_frame, traj) is NOT unless reset_info_on_autoreset=True.
    """

    def __init__(self, env: Env, reset_info_on_autoreset: bool = False):
        super().__init__(env)
        self.reset_info = reset_info_on_autoreset

    def reset(self, rng=None, **kw) -> State:
        state = self.env.reset(rng, **kw)
        state.info["first_pipeline_state"] = state.pipeline_state.clone()
        state.info["first_obs"] = state.obs.clone()
        if self.reset_info:
            state.info["first_info"] = {k: state.info[k].clone() for k in ("cur_frame", "sub_clip_frame", "traj")}
        return state

    def step(self, state: State, action: torch.Tensor) -> State:
        if "steps" in state.info:
            steps = state.info["steps"]
            steps.copy_(torch.where(state.done.bool(), torch.zeros_like(steps), steps))
        state.done.zero_()
        state = self.env.step(state, action)
        done = state.done.bool()
        state.pipeline_state.copy_(state.info["first_pipeline_state"], mask=done)
        state.obs.copy_(torch.where(done[:, None], state.info["first_obs"], state.obs))
        if self.reset_info:
            for k, v in state.info["first_info"].items():
                cur = state.info[k]
                cur.copy_(torch.where(done[:, None] if cur.dim() == 2 else done, v, cur))
        return state


@dataclasses.dataclass
class EvalMetrics:
    """brax.envs.wrappers.training.EvalMetrics."""

    episode_metrics: Dict[str, torch.Tensor]
    active_episodes: torch.Tensor
    episode_steps: torch.Tensor


class EvalWrapper(Wrapper):
    """Accumulates per-episode metrics of the first episode of every env (brax EvalWrapper)."""

    def reset(self, rng=None, **kw) -> State:
        state = self.env.reset(rng, **kw)
        names = list(state.metrics.keys()) + ["reward"]
        state.info["eval_metrics"] = EvalMetrics(
            episode_metrics={k: torch.zeros_like(state.reward) for k in names},
            active_episodes=torch.ones_like(state.reward),
            episode_steps=torch.zeros_like(state.reward),
        )
        return state

    def step(self, state: State, action: torch.Tensor) -> State:
        em: EvalMetrics = state.info["eval_metrics"]
        nstate = self.env.step(state, action)
        active = em.active_episodes
        em.episode_steps.copy_(torch.where(active.bool(), nstate.info["steps"], em.episode_steps))
        for k, acc in em.episode_metrics.items():
            acc += (nstate.reward if k == "reward" else nstate.metrics[k]) * active
        active *= 1.0 - nstate.done
        return nstate


def wrap(env: Env, episode_length: int = 1000, action_repeat: int = 1, randomization_fn=None,
         reset_info_on_autoreset: bool = False) -> Wrapper:
    """brax.envs.training.wrap minus the VmapWrapper (natively batched env)."""
    if randomization_fn is not None:
        raise NotImplementedError("domain randomisation is outside the hot path")
    return AutoResetWrapper(EpisodeWrapper(env, episode_length, action_repeat), reset_info_on_autoreset)
