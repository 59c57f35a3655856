"""Brax-shaped containers for the natively batched environments.

`State` mirrors brax.envs.base.State (fields pipeline_state, obs, reward, done,
metrics, info) as used at reference envs/rodent.py:166,237-239; `PipelineState`
exposes the mjx.Data fields the reference env reads (qpos, qvel, xpos, xmat,
subtree_com, qfrc_actuator, q, qd: rodent.py:250-314,335-341).

Storage is row-major [env][feature] -- plain (B, n) tensors, exactly what the kernels
read and write (one wavefront per env, lanes across the feature axis).
"""
from __future__ import annotations

import dataclasses
from typing import Any, Dict

import torch

from .. import _lib


class PipelineState:
    """Carried physics state + quantities derived by the last forward pass."""

    _FIELDS = ("qpos", "qvel", "act", "qacc_warmstart", "xpos", "xquat", "subtree_com1", "qfrc_actuator")

    def __init__(self, soa: Dict[str, torch.Tensor]):
        self._soa = soa

    @staticmethod
    def allocate(dims, B: int, device, dtype=torch.float32) -> "PipelineState":
        n = dict(qpos=dims.nq, qvel=dims.nv, act=dims.nu, qacc_warmstart=dims.nv, xpos=3 * dims.nbody,
                 xquat=4 * dims.nbody, subtree_com1=3, qfrc_actuator=dims.nv)
        return PipelineState({k: torch.zeros((B, v), dtype=dtype, device=device) for k, v in n.items()})

    def raw(self, name: str) -> torch.Tensor:
        return self._soa[name]

    qpos = property(lambda s: s._soa["qpos"])
    qvel = property(lambda s: s._soa["qvel"])
    q = qpos
    qd = qvel
    act = property(lambda s: s._soa["act"])
    qacc_warmstart = property(lambda s: s._soa["qacc_warmstart"])
    qfrc_actuator = property(lambda s: s._soa["qfrc_actuator"])

    @property
    def xpos(self) -> torch.Tensor:  # (B, nbody, 3)
        t = self._soa["xpos"]
        return t.view(t.shape[0], -1, 3)

    @property
    def xquat(self) -> torch.Tensor:  # (B, nbody, 4)
        t = self._soa["xquat"]
        return t.view(t.shape[0], -1, 4)

    @property
    def xmat(self) -> torch.Tensor:  # (B, nbody, 3, 3), computed on demand from xquat
        q = self.xquat
        w, x, y, z = q.unbind(-1)
        m = torch.stack([
            w * w + x * x - y * y - z * z, 2 * (x * y - w * z), 2 * (x * z + w * y),
            2 * (x * y + w * z), w * w - x * x + y * y - z * z, 2 * (y * z - w * x),
            2 * (x * z - w * y), 2 * (y * z + w * x), w * w - x * x - y * y + z * z], dim=-1)
        return m.reshape(*q.shape[:-1], 3, 3)

    @property
    def subtree_com_root(self) -> torch.Tensor:  # (B, 3) == data.subtree_com[1]
        return self._soa["subtree_com1"]

    def clone(self) -> "PipelineState":
        return PipelineState({k: v.clone() for k, v in self._soa.items()})

    def copy_(self, other: "PipelineState", mask: torch.Tensor | None = None) -> None:
        for k, v in self._soa.items():
            if mask is None:
                v.copy_(other._soa[k])
            else:
                torch.where(mask[:, None], other._soa[k], v, out=v)


@dataclasses.dataclass
class State:
    """brax.envs.base.State counterpart.  obs/reward/done are env-major views."""

    pipeline_state: PipelineState
    obs: torch.Tensor  # (B, obs_size)
    reward: torch.Tensor  # (B,)
    done: torch.Tensor  # (B,)
    metrics: Dict[str, torch.Tensor]
    info: Dict[str, Any]

    def replace(self, **kw) -> "State":
        return dataclasses.replace(self, **kw)


class Env:
    """Minimal brax.envs.Env protocol (reset / step / sizes)."""

    def reset(self, rng) -> State:  # pragma: no cover - interface
        raise NotImplementedError

    def step(self, state: State, action: torch.Tensor) -> State:  # pragma: no cover - interface
        raise NotImplementedError

    @property
    def observation_size(self) -> int:
        raise NotImplementedError

    @property
    def action_size(self) -> int:
        raise NotImplementedError

    @property
    def unwrapped(self) -> "Env":
        return self


def soa_ptr(t: torch.Tensor) -> int:
    assert t.is_contiguous()
    return t.data_ptr()
