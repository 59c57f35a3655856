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


def _fusable(env):
    """AutoReset(Episode(RodentTracking)) with action_repeat 1 on the native library: the wrappers and the
    Transition logging of one step then collapse into ONE launch of vnl_rollout_post."""
    from ..envs import wrappers as W

    if not isinstance(env, W.AutoResetWrapper) or not isinstance(env.env, W.EpisodeWrapper):
        return None
    ep, base = env.env, env.env.env
    if ep.action_repeat != 1 or base is not base.unwrapped or not hasattr(getattr(base, "_L", None), "vnl_rollout_post"):
        return None
    return env, ep, base


def _generate_unroll_fused(fz, env_state: State, policy, key, unroll_length: int, extra_fields) -> Tuple[State, Transition]:
    """Same results as the generic loop below (tests compare the two), 2 launches per step besides the
    policy: the env step kernel and vnl_rollout_post (which also writes the Transition's observation row)."""
    import ctypes as C

    from .. import _lib

    ar, ep, base = fz
    st = env_state
    T, info, dev = unroll_length, env_state.info, env_state.obs.device
    B = st.obs.shape[0]
    new = lambda *shape, like: torch.empty((T, *shape), dtype=like.dtype, device=dev)  # noqa: E731
    # Transition.observation[t + 1] IS next_observation[t] (the auto-reset wrapper's obs is what the next step sees), so only
    # the first row is copied; the policy reads the live observation buffer
    obs0, nobs_log = st.obs.clone(), new(*st.obs.shape, like=st.obs)
    rew_log, disc_log = new(B, like=st.reward), new(B, like=st.reward)
    sx_log = {x: new(*info[x].shape, like=info[x]) for x in extra_fields}
    prev_done = st.done.clone()
    stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream) if dev.type == "cuda" else C.c_void_p(0)
    ptr = lambda t: C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)  # noqa: E731
    width = lambda t: 1 if t.dim() == 1 else t.shape[1]  # noqa: E731

    d = _lib.PostDesc()
    d.steps, d.prev_done, d.done, d.truncation, d.reward = (ptr(info["steps"]), ptr(prev_done), ptr(st.done),
                                                            ptr(info["truncation"]), ptr(st.reward))
    d.episode_length, d.action_repeat = ep.episode_length, 1
    ops = []  # (dst, src, first, log rows [T, B, w] or None)
    fps = info["first_pipeline_state"]
    for name in st.pipeline_state._FIELDS:
        cur = st.pipeline_state.raw(name)
        ops.append((cur, cur, fps.raw(name), None))
    ops.append((st.obs, st.obs, info["first_obs"], nobs_log))
    logged = set()
    if ar.reset_info:
        for k2, v in info["first_info"].items():
            ops.append((info[k2], info[k2], v, sx_log.get(k2)))
            logged.add(k2)
    for x in extra_fields:
        if x not in logged and x != "truncation":
            ops.append((None, info[x], None, sx_log[x]))
    n_static = len(ops)
    px_log: Optional[dict] = None
    act_log = None
    for t in range(T):
        actions, pex = policy(info["traj"], st.obs, key)
        base.step(st, actions)
        if px_log is None:
            act_log = new(*actions.shape, like=actions)
            px_log = {k2: new(*v.shape, like=v) for k2, v in pex.items()}
        step_ops = ops[:n_static] + [(None, actions.contiguous(), None, act_log)] + \
            [(None, v.contiguous(), None, px_log[k2]) for k2, v in pex.items()]
        assert len(step_ops) <= _lib.POST_MAX_OPS
        d.num_ops = len(step_ops)
        keep = []
        for i, (dst, src, first, log) in enumerate(step_ops):
            assert src.dtype in (torch.float32, torch.int32) and src.is_contiguous()
            o = d.ops[i]
            o.dst, o.src, o.first, o.width = ptr(dst), ptr(src), ptr(first), width(src)
            o.log = ptr(log[t]) if log is not None else C.c_void_p(0)
            keep.append(src)
        d.log_reward, d.log_discount = ptr(rew_log[t]), ptr(disc_log[t])
        d.log_truncation = ptr(sx_log["truncation"][t]) if "truncation" in sx_log else C.c_void_p(0)
        _lib.check(base._L, base._L.vnl_rollout_post(C.byref(d), B, stream))
        _generate_unroll_fused.hold = keep  # the launch is asynchronous: keep this step's sources alive
    obs_log = torch.cat((obs0[None], nobs_log[:-1]), dim=0)
    data = Transition(observation=obs_log, action=act_log, reward=rew_log, discount=disc_log, next_observation=nobs_log,
                      extras={"policy_extras": px_log, "state_extras": sx_log})
    return st, data


class GraphedUnroll:
    """acting.generate_unroll (reference acting.py:60-80) captured ONCE into a hipGraph and replayed: the `unroll_length`
    policy launches, env step kernels, wrapper / logging launches and noise draws of an unroll are one graph launch instead
    of ~11 launches per step issued from Python (the eager loop leaves ~0.13 ms of launch gaps and small copies per control
    step on an MI355X).  Bit-identical to the eager fused unroll: same kernels, same arguments, same Philox stream
    (tests/test_fused_rollout.py).

    The graph works on fixed buffers: `env_state` (the env mutates it in place), the policy's parameter tensors (update them
    IN PLACE between replays: the trainer's flat parameter buffer already is; a normaliser state must be copied into the one
    given here) and the returned Transition, whose tensors are overwritten by the next replay.  `key` must be a generator
    on the device; it is registered with the graph, so every replay continues its stream as eager calls would."""

    def __init__(self, env, env_state: State, policy, key: torch.Generator, unroll_length: int,
                 extra_fields: Sequence[str] = ()):
        fz = _fusable(env)
        dev = env_state.obs.device
        if fz is None or dev.type != "cuda":
            raise ValueError("GraphedUnroll needs AutoResetWrapper(EpisodeWrapper(<env on a HIP device>)), action_repeat 1")
        if key is None or key.device.type != "cuda":
            raise ValueError("GraphedUnroll needs a torch.Generator on the device (its stream is captured with the graph)")
        self._args = (fz, env_state, policy, key, int(unroll_length), tuple(extra_fields))
        self.state = env_state
        # warm-up on a side stream (library handles, the policy's kernel object, allocator pools), with the generator and the
        # env state put back afterwards: building the graph must not consume randomness or advance the envs
        gen_state = key.get_state()
        saved = _snapshot_state(env_state)
        side = torch.cuda.Stream(dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            _generate_unroll_fused(*self._args)
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        _restore_state(env_state, saved)
        key.set_state(gen_state)
        self.graph = torch.cuda.CUDAGraph()
        self.graph.register_generator_state(key)
        with torch.cuda.graph(self.graph):
            _, self.data = _generate_unroll_fused(*self._args)
        self._hold = getattr(_generate_unroll_fused, "hold", None)
        _restore_state(env_state, saved)  # (capture does not execute, but keep the contract explicit)

    def __call__(self) -> Tuple[State, Transition]:
        self.graph.replay()
        return self.state, self.data


def _state_tensors(st: State):
    out = [st.pipeline_state.raw(n) for n in st.pipeline_state._FIELDS] + [st.obs, st.reward, st.done]
    for k in sorted(st.info):
        v = st.info[k]
        if isinstance(v, torch.Tensor):
            out.append(v)
        elif isinstance(v, dict):
            out += [v[k2] for k2 in sorted(v) if isinstance(v[k2], torch.Tensor)]
    seen, uniq = set(), []
    for t in out:  # (info["_raw"] aliases obs / reward / ...: each storage once)
        if t.data_ptr() not in seen:
            seen.add(t.data_ptr())
            uniq.append(t)
    return uniq


def _snapshot_state(st: State):
    return [t.clone() for t in _state_tensors(st)]


def _restore_state(st: State, saved) -> None:
    for t, s in zip(_state_tensors(st), saved):
        t.copy_(s)


def generate_unroll(env, env_state: State, policy, key, unroll_length: int,
                    extra_fields: Sequence[str] = (), fused: Optional[bool] = None) -> Tuple[State, Transition]:
    """acting.py:60-80: Transitions stacked on a leading time axis [T, B, ...], written row by row into
    buffers allocated once per call (the env mutates its State in place).
    NB (reference behaviour): state_extras['traj'] is nstate.info['traj'], i.e. the reference
    trajectory features AFTER the step (acting.py:49), not the ones the policy saw.
    `fused` (default: whenever possible) routes the wrappers + logging through vnl_rollout_post."""
    fz = _fusable(env) if fused is not False else None
    if fused is True and fz is None:
        raise ValueError("fused rollout needs AutoResetWrapper(EpisodeWrapper(RodentTracking)), action_repeat 1")
    if fz is not None:
        return _generate_unroll_fused(fz, env_state, policy, key, unroll_length, tuple(extra_fields))
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
