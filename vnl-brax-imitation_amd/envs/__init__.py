"""Environment registry, mirroring how the reference registers and constructs envs
(reference train.py:65-68: envs.register_environment(...); train.py:86-90:
envs.get_environment(name, reference_clip=..., **env_args))."""
from __future__ import annotations

from typing import Callable, Dict

from .ant import AntTracking  # noqa: F401
from .base import Env, PipelineState, State  # noqa: F401
from .humanoid import HumanoidTracking  # noqa: F401
from .rodent import RodentMultiClipTracking, RodentTracking  # noqa: F401

_envs: Dict[str, Callable[..., Env]] = {}


def register_environment(env_name: str, env_class: Callable[..., Env]) -> None:
    _envs[env_name] = env_class


def get_environment(env_name: str, **kwargs) -> Env:
    return _envs[env_name](**kwargs)


register_environment("rodent", RodentTracking)
register_environment("rodent_multiclip", RodentMultiClipTracking)
register_environment("humanoidtracking", HumanoidTracking)  # reference train.py:65
register_environment("ant", AntTracking)  # reference train.py:66
