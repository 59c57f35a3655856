"""Running observation normaliser with brax.training.acme.running_statistics semantics
[UPSTREAM] (used at reference ppo_imitation/train.py:220-222,330-334,405-407).

update(): Welford batch update; with a process group the three batch sums (count,
sum(x - mean_old), sum((x - mean_old)(x - mean_new))) are all-reduced -- collective C2 of
SURVEY.md 2.3: one tiny all-reduce per training step.
"""
from __future__ import annotations

import dataclasses
from typing import Optional

import torch


@dataclasses.dataclass
class RunningStatisticsState:
    count: torch.Tensor  # scalar (float64 on CPU side semantics; kept float32 like brax)
    mean: torch.Tensor
    summed_variance: torch.Tensor
    std: torch.Tensor

    def clone(self) -> "RunningStatisticsState":
        return RunningStatisticsState(*(t.clone() for t in (self.count, self.mean, self.summed_variance, self.std)))


def init_state(size: int, device=None) -> RunningStatisticsState:
    z = lambda: torch.zeros(size, dtype=torch.float32, device=device)  # noqa: E731
    return RunningStatisticsState(torch.zeros((), dtype=torch.float32, device=device), z(), z(),
                                  torch.ones(size, dtype=torch.float32, device=device))


def update(state: RunningStatisticsState, batch: torch.Tensor, *, std_min_value: float = 1e-6,
           std_max_value: float = 1e6, process_group=None, distributed: bool = False) -> RunningStatisticsState:
    """Welford-style update of brax.training.acme.running_statistics.update.  Data-parallel: TWO all-reduces per call --
    [count | sum of (x - old mean)] in one buffer, then the sum of (x - old mean)(x - new mean), which needs the new,
    global, mean (the same two dependent psums as brax)."""
    x = batch.reshape(-1, batch.shape[-1])
    diff_old = x - state.mean
    packed = torch.cat([torch.full((1,), float(x.shape[0]), dtype=torch.float32, device=x.device), diff_old.sum(0)])
    if distributed:
        import torch.distributed as dist

        dist.all_reduce(packed, group=process_group)
    count = state.count + packed[0]
    mean = state.mean + packed[1:] / count
    s2 = (diff_old * (x - mean)).sum(0)
    if distributed:
        dist.all_reduce(s2, group=process_group)
    summed_variance = state.summed_variance + s2
    std = torch.sqrt(torch.clamp(summed_variance, min=0.0) / count).clamp(std_min_value, std_max_value)
    return RunningStatisticsState(count, mean, summed_variance, std)


def normalize(batch: torch.Tensor, mean_std: Optional[RunningStatisticsState],
              max_abs_value: Optional[float] = None) -> torch.Tensor:
    if mean_std is None:
        return batch
    out = (batch - mean_std.mean) / mean_std.std
    if max_abs_value is not None:
        out = out.clamp(-max_abs_value, max_abs_value)
    return out
