"""PPO networks with the intention policy: counterpart of reference ppo_imitation/ppo_networks.py:27-124.

`make_intention_ppo_networks(traj_size, observation_size, action_size, preprocess_observations_fn=...)`
keeps the reference's factory signature (it is what ppo.train calls through `network_factory`,
reference ppo_imitation/train.py:225-230).  Parameters are flat float32 buffers (see
intention_policy_network.py); "params" tuples are (normalizer_params, flat_policy_params) exactly
like the reference's `(normalizer_params, params.policy)` (train.py:299-301).
"""
from __future__ import annotations

import dataclasses
import math
from typing import Any, Callable, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

from . import distribution, running_statistics
from . import intention_policy_network as ipn


def identity_observation_preprocessor(obs, params):
    return obs


@dataclasses.dataclass
class FeedForwardNetwork:
    init: Callable[[torch.Generator], torch.Tensor]
    apply: Callable[..., Any]
    layout: ipn.ParamLayout


class ValueMLP:
    """brax.training.networks.make_value_network [UPSTREAM]: MLP obs -> hidden... -> 1, swish, lecun_uniform,
    output squeezed (used at ppo_networks.py:114-118)."""

    def __init__(self, obs_size: int, hidden: Sequence[int]):
        self.sizes = list(hidden) + [1]
        L = ipn.ParamLayout()
        fan = obs_size
        for i, h in enumerate(self.sizes):
            L.add(f"hidden_{i}/kernel", (fan, h)), L.add(f"hidden_{i}/bias", (h,))
            fan = h
        self.layout = L

    def init(self, gen: torch.Generator) -> torch.Tensor:
        flat = torch.zeros(self.layout.size, dtype=torch.float32)
        for name, (off, shape) in self.layout.entries.items():
            if name.endswith("/kernel"):
                ipn.lecun_uniform_(self.layout.view(flat, name), shape[0], gen)
        return flat

    def apply(self, flat: torch.Tensor, obs: torch.Tensor) -> torch.Tensor:
        lead = obs.shape[:-1]
        x = obs.reshape(-1, obs.shape[-1])
        n = len(self.sizes)
        for i in range(n):  # addmm: bias fused into the GEMM epilogue
            x = torch.addmm(self.layout.view(flat, f"hidden_{i}/bias"), x, self.layout.view(flat, f"hidden_{i}/kernel"))
            if i != n - 1:
                x = F.silu(x)
        return x.reshape(*lead, -1).squeeze(-1)


@dataclasses.dataclass
class PPOImitationNetworks:
    policy_network: FeedForwardNetwork
    value_network: FeedForwardNetwork
    parametric_action_distribution: distribution.NormalTanhDistribution
    policy_module: ipn.IntentionNetwork = None
    value_module: ValueMLP = None
    normalizes: bool = False  # preprocess_observations_fn is running_statistics.normalize
    hip_ok: bool = False      # the fused HIP inference kernel computes the same function


def make_intention_ppo_networks(
    traj_size: int,
    observation_size: int,
    action_size: int,
    preprocess_observations_fn=identity_observation_preprocessor,
    intention_latent_size: int = 64,
    encoder_layer_sizes: Sequence[int] = (1024,) * 2,
    decoder_layer_sizes: Sequence[int] = (1024,) * 2,
    value_hidden_layer_sizes: Sequence[int] = (1024,) * 2,
) -> PPOImitationNetworks:
    """ppo_networks.py:91-124."""
    dist = distribution.NormalTanhDistribution(event_size=action_size)
    pol = ipn.make_intention_policy(dist.param_size, latent_size=intention_latent_size, traj_size=traj_size,
                                    obs_size=observation_size, encoder_layer_sizes=encoder_layer_sizes,
                                    decoder_layer_sizes=decoder_layer_sizes)
    val = ValueMLP(observation_size, value_hidden_layer_sizes)

    def policy_apply(processor_params, policy_params, traj, obs, eps_latent):
        obs = preprocess_observations_fn(obs, processor_params)  # traj is NOT normalised (ipn:125-127)
        return pol.apply(policy_params, traj, obs, eps_latent)

    def value_apply(processor_params, value_params, obs):
        return val.apply(value_params, preprocess_observations_fn(obs, processor_params))

    return PPOImitationNetworks(
        policy_network=FeedForwardNetwork(init=pol.init, apply=policy_apply, layout=pol.layout),
        value_network=FeedForwardNetwork(init=val.init, apply=value_apply, layout=val.layout),
        parametric_action_distribution=dist,
        policy_module=pol,
        value_module=val,
        normalizes=preprocess_observations_fn is running_statistics.normalize,
        hip_ok=preprocess_observations_fn in (running_statistics.normalize, identity_observation_preprocessor),
    )


def _randn(shape, key: Optional[torch.Generator], device) -> torch.Tensor:
    if key is not None and key.device.type != torch.device(device).type:
        return torch.randn(shape, generator=key, dtype=torch.float32).to(device)
    return torch.randn(shape, generator=key, dtype=torch.float32, device=device)


def _rand_action(n: int, key: Optional[torch.Generator], device) -> torch.Tensor:
    """One U(-1, 1) pre-tanh action shared by the whole batch (ppo_networks.py:67-69), drawn on the device the
    generator lives on (a host draw + copy would synchronise the rollout loop every step)."""
    if key is not None and key.device.type != torch.device(device).type:
        return torch.empty((n,), dtype=torch.float32).uniform_(-1.0, 1.0, generator=key).to(device)
    return torch.empty((n,), dtype=torch.float32, device=device).uniform_(-1.0, 1.0, generator=key)  # one launch, not three


def make_inference_fn(ppo_networks: PPOImitationNetworks):
    """ppo_networks.py:35-87.  `key_sample` is a torch.Generator (or None for the global one);
    the JAX threefry stream is not reproduced, the sampling structure is:
      eps_latent ~ N(0,I) for the encoder's reparameterisation (ipn:96-101),
      eps_action ~ N(0,I) for sample_no_postprocessing (ppo_networks.py:60-62),
      one (action_size,) U(-1,1) draw broadcast over the batch for rand_log_prob (:67-73)."""

    hip_cache = {}

    def make_policy(params, deterministic: bool = False, backend: str = "auto"):
        """backend: "hip" = fused MFMA inference kernel (vnl_policy_forward), "torch" = hipBLASLt ops,
        "auto" = hip on a HIP device when the observation preprocessor is the running-statistics
        normaliser or the identity."""
        normalizer_params, policy_params = params
        dist = ppo_networks.parametric_action_distribution
        latent = ppo_networks.policy_module.latents
        use_hip = backend == "hip" or (backend == "auto" and policy_params.is_cuda and ppo_networks.hip_ok)

        @torch.no_grad()
        def policy_hip(trajectories, observations, key_sample=None):
            from .hip_policy import HipIntentionPolicy

            dev, B = observations.device, observations.shape[0]
            k = (dev, B)
            if k not in hip_cache:
                hip_cache[k] = HipIntentionPolicy(ppo_networks.policy_module, dist.event_size, B, dev)
            eps_latent = _randn((B, latent), key_sample, dev)
            eps_action = None if deterministic else _randn((B, dist.event_size), key_sample, dev)
            mean = std = None
            if ppo_networks.normalizes and normalizer_params is not None:
                mean, std = normalizer_params.mean, normalizer_params.std
            random_actions = None if deterministic else _rand_action(dist.event_size, key_sample, dev)
            action, extras = hip_cache[k].forward(policy_params.detach(), mean, std, trajectories, observations,
                                                  eps_latent, eps_action, deterministic, rand_action=random_actions)
            if deterministic:
                return action, {}
            return action, {k2: extras[k2] for k2 in ("log_prob", "rand_log_prob", "raw_action", "logits")}

        if use_hip:
            return policy_hip

        @torch.no_grad()
        def policy(trajectories: torch.Tensor, observations: torch.Tensor,
                   key_sample: Optional[torch.Generator] = None) -> Tuple[torch.Tensor, dict]:
            dev = observations.device
            lead = observations.shape[:-1]
            eps_latent = _randn((*lead, latent), key_sample, dev)
            logits, _, _ = ppo_networks.policy_network.apply(normalizer_params, policy_params, trajectories,
                                                             observations, eps_latent)
            if deterministic:
                return dist.mode(logits), {}
            eps_action = _randn((*lead, dist.event_size), key_sample, dev)
            raw_actions = dist.sample_no_postprocessing(logits, eps_action)
            log_prob = dist.log_prob(logits, raw_actions)
            random_actions = _rand_action(dist.event_size, key_sample, dev)
            rand_log_prob = dist.log_prob(logits, random_actions.expand_as(raw_actions))
            return dist.postprocess(raw_actions), {
                "log_prob": log_prob,
                "rand_log_prob": rand_log_prob,
                "raw_action": raw_actions,
                "logits": logits,
            }

        return policy

    return make_policy
