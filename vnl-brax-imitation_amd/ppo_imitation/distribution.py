"""tanh-squashed diagonal Normal, as brax.training.distribution.NormalTanhDistribution
[UPSTREAM], which the reference builds at ppo_imitation/ppo_networks.py:102-104.

parameters = [loc | pre-softplus scale]; scale = softplus(s) + min_std.
log_prob takes RAW (pre-tanh) actions and subtracts the tanh log-det-Jacobian
2 (log 2 - x - softplus(-2x)).
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F

_HALF_LOG_2PI = 0.5 * math.log(2.0 * math.pi)


class NormalTanhDistribution:
    def __init__(self, event_size: int, min_std: float = 0.001, var_scale: float = 1.0):
        self.param_size = 2 * event_size
        self.event_size = event_size
        self._min_std = min_std
        self._var_scale = var_scale

    def _loc_scale(self, parameters: torch.Tensor):
        loc, s = torch.chunk(parameters, 2, dim=-1)
        return loc, (F.softplus(s) + self._min_std) * self._var_scale

    @staticmethod
    def _fldj(x: torch.Tensor) -> torch.Tensor:  # TanhBijector.forward_log_det_jacobian
        return 2.0 * (math.log(2.0) - x - F.softplus(-2.0 * x))

    def sample_no_postprocessing(self, parameters: torch.Tensor, eps: torch.Tensor) -> torch.Tensor:
        loc, scale = self._loc_scale(parameters)
        return loc + scale * eps

    @staticmethod
    def postprocess(x: torch.Tensor) -> torch.Tensor:
        return torch.tanh(x)

    def mode(self, parameters: torch.Tensor) -> torch.Tensor:
        return torch.tanh(self._loc_scale(parameters)[0])

    def log_prob(self, parameters: torch.Tensor, raw_actions: torch.Tensor) -> torch.Tensor:
        loc, scale = self._loc_scale(parameters)
        z = (raw_actions - loc) / scale
        lp = -0.5 * z * z - _HALF_LOG_2PI - torch.log(scale)
        return (lp - self._fldj(raw_actions)).sum(-1)

    def entropy(self, parameters: torch.Tensor, eps: torch.Tensor) -> torch.Tensor:
        loc, scale = self._loc_scale(parameters)
        ent = 0.5 + _HALF_LOG_2PI + torch.log(scale)
        return (ent + self._fldj(loc + scale * eps)).sum(-1)
