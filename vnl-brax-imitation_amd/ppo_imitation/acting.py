"""Acting: counterpart of reference ppo_imitation/acting.py:34-156.

actor_step = policy call + env step + Transition row; generate_unroll = unroll_length of those
written into preallocated [T, B, ...] buffers (the env updates its State in place, so the
pre-step observation is copied into the buffer before stepping)."""
from __future__ import annotations

import dataclasses
import time
from typing import Any, Callable, Dict, Optional, Sequence, Tuple

import numpy as np
import torch

from ..envs.base import State
from ..envs.wrappers import EvalWrapper


@dataclasses.dataclass
class Transition:
    """brax.training.types.Transition [UPSTREAM]."""

    observation: torch.Tensor
    action: torch.Tensor
    reward: torch.Tensor
    discount: torch.Tensor
    next_observation: torch.Tensor
    extras: Dict[str, Dict[str, torch.Tensor]]

    def map(self, fn: Callable[[torch.Tensor], torch.Tensor]) -> "Transition":
        return Transition(fn(self.observation), fn(self.action), fn(self.reward), fn(self.discount),
                          fn(self.next_observation),
                          {k: {kk: fn(vv) for kk, vv in v.items()} for k, v in self.extras.items()})


def actor_step(env, env_state: State, policy, key, extra_fields: Sequence[str] = ()) -> Tuple[State, Transition]:
    """acting.py:34-57."""
    obs = env_state.obs.clone()
    traj = env_state.info["traj"]
    actions, policy_extras = policy(traj, obs, key)
    nstate = env.step(env_state, actions)
    state_extras = {x: nstate.info[x] for x in extra_fields}
    return nstate, Transition(observation=obs, action=actions, reward=nstate.reward, discount=1 - nstate.done,
                              next_observation=nstate.obs,
                              extras={"policy_extras": policy_extras, "state_extras": state_extras})


def generate_unroll(env, env_state: State, policy, key, unroll_length: int,
                    extra_fields: Sequence[str] = ()) -> Tuple[State, Transition]:
    """acting.py:60-80: Transitions stacked on a leading time axis [T, B, ...], written row by row into
    buffers allocated once per call (the env mutates its State in place).
    NB (reference behaviour): state_extras['traj'] is nstate.info['traj'], i.e. the reference
    trajectory features AFTER the step (acting.py:49), not the ones the policy saw."""
    data: Optional[Transition] = None
    for t in range(unroll_length):
        env_state, tr = actor_step(env, env_state, policy, key, extra_fields=extra_fields)
        if data is None:
            alloc = lambda x: torch.empty((unroll_length, *x.shape), dtype=x.dtype, device=x.device)  # noqa: E731
            data = tr.map(alloc)
        for dst, src in zip(_leaves(data), _leaves(tr)):
            dst[t].copy_(src)
    return env_state, data


def _leaves(tr: Transition):
    out = [tr.observation, tr.action, tr.reward, tr.discount, tr.next_observation]
    for g in sorted(tr.extras):
        out += [tr.extras[g][k] for k in sorted(tr.extras[g])]
    return out


class Evaluator:
    """acting.py:84-156."""

    def __init__(self, eval_env, eval_policy_fn, num_eval_envs: int, episode_length: int, action_repeat: int,
                 key: Optional[torch.Generator]):
        self._key = key
        self._eval_walltime = 0.0
        self._env = EvalWrapper(eval_env)
        self._policy_fn = eval_policy_fn
        self._unroll = episode_length // action_repeat
        self._steps_per_unroll = episode_length * num_eval_envs

    def run_evaluation(self, policy_params, training_metrics: Dict[str, Any], aggregate_episodes: bool = True):
        t = time.time()
        state = self._env.reset(self._key)
        state, _ = generate_unroll(self._env, state, self._policy_fn(policy_params), self._key, self._unroll)
        em = state.info["eval_metrics"]
        if em.active_episodes.is_cuda:
            torch.cuda.synchronize(em.active_episodes.device)
        epoch_eval_time = time.time() - t
        metrics = {}
        for fn in (np.mean, np.std):
            suffix = "_std" if fn is np.std else ""
            metrics.update({f"eval/episode_{name}{suffix}": (fn(v.cpu().numpy()) if aggregate_episodes else v.cpu().numpy())
                            for name, v in em.episode_metrics.items()})
        metrics["eval/avg_episode_length"] = float(np.mean(em.episode_steps.cpu().numpy()))
        metrics["eval/epoch_eval_time"] = epoch_eval_time
        metrics["eval/sps"] = self._steps_per_unroll / epoch_eval_time
        self._eval_walltime += epoch_eval_time
        return {"eval/walltime": self._eval_walltime, **training_metrics, **metrics}
