"""GAE + clipped-surrogate PPO loss with latent KL: counterpart of reference
ppo_imitation/intention_losses.py:21-202."""
from __future__ import annotations

import dataclasses
from typing import Any, Dict, Optional, Tuple

import torch

from .acting import Transition


@dataclasses.dataclass
class PPONetworkParams:
    """intention_losses.py:12-18: flat policy / value parameter buffers."""

    policy: torch.Tensor
    value: torch.Tensor


def kl_divergence(mean: torch.Tensor, logvar: torch.Tensor) -> torch.Tensor:
    """intention_losses.py:21-23 (a MEAN over batch x latent)."""
    return -0.5 * torch.mean(1 + logvar - mean.square() - logvar.exp())


def _corrcoef(x: torch.Tensor) -> torch.Tensor:
    """jnp.corrcoef of the rows of x (torch.corrcoef reads back to the host, which a hipGraph capture forbids)."""
    xm = x - x.mean(dim=1, keepdim=True)
    c = xm @ xm.T / (x.shape[1] - 1)
    d = torch.sqrt(torch.diagonal(c))
    return (c / (d[:, None] * d[None, :])).clamp(-1.0, 1.0)


def compute_gae(truncation, termination, rewards, values, bootstrap_value, lambda_: float = 1.0,
                discount: float = 0.99):
    """intention_losses.py:26-87; inputs time-major [T, B]; returns (vs, advantages), no gradient."""
    with torch.no_grad():
        truncation_mask = 1 - truncation
        values_t_plus_1 = torch.cat([values[1:], bootstrap_value[None]], dim=0)
        deltas = (rewards + discount * (1 - termination) * values_t_plus_1 - values) * truncation_mask
        # acc_t = deltas_t + coef_t * acc_{t+1} (intention_losses.py:69-79, a reverse lax.scan) as a scan by
        # doubling: log2(T) rounds of whole-array ops instead of T tiny sequential steps (launch-bound on a GPU)
        coef = discount * (1 - termination) * truncation_mask * lambda_
        out = deltas.clone()
        T = values.shape[0]
        k = 1
        while k < T:
            out = torch.cat([out[:T - k] + coef[:T - k] * out[k:], out[T - k:]], dim=0)
            coef = torch.cat([coef[:T - k] * coef[k:], coef[T - k:]], dim=0)
            k *= 2
        vs = out + values
        vs_t_plus_1 = torch.cat([vs[1:], bootstrap_value[None]], dim=0)
        advantages = (rewards + discount * (1 - termination) * vs_t_plus_1 - values) * truncation_mask
    return vs, advantages


def compute_ppo_intention_loss(
    params: PPONetworkParams,
    normalizer_params: Any,
    data: Transition,
    rng: Optional[torch.Generator],
    ppo_network,
    entropy_cost: float = 1e-4,
    discounting: float = 0.9,
    reward_scaling: float = 1.0,
    gae_lambda: float = 0.95,
    clipping_epsilon: float = 0.3,
    normalize_advantage: bool = True,
    kl_weight: float = 1e-4,
    noise: Optional[Dict[str, torch.Tensor]] = None,
) -> Tuple[torch.Tensor, Dict[str, torch.Tensor]]:
    """intention_losses.py:91-202.  `data` has leading dims [B, T]; `noise` optionally supplies the
    two Gaussian draws ('latent' [T,B,latent], 'entropy' [T,B,act]) for reproducible tests."""
    dist = ppo_network.parametric_action_distribution
    policy_apply = ppo_network.policy_network.apply
    value_apply = ppo_network.value_network.apply

    data = data.map(lambda x: x.transpose(0, 1))  # time-major, :131
    obs = data.observation
    dev = obs.device
    T, B = obs.shape[:2]
    latent = ppo_network.policy_module.latents

    def draw(name, shape):
        if noise is not None and name in noise:
            return noise[name]
        if rng is not None and rng.device.type != dev.type:
            return torch.randn(shape, generator=rng).to(dev)
        return torch.randn(shape, generator=rng, device=dev)

    policy_logits, intention_mean, intention_logvar = policy_apply(
        normalizer_params, params.policy, data.extras["state_extras"]["traj"], obs, draw("latent", (T, B, latent)))
    baseline = value_apply(normalizer_params, params.value, obs)
    bootstrap_value = value_apply(normalizer_params, params.value, data.next_observation[-1])

    rewards = data.reward * reward_scaling
    truncation = data.extras["state_extras"]["truncation"]
    termination = (1 - data.discount) * (1 - truncation)

    target_action_log_probs = dist.log_prob(policy_logits, data.extras["policy_extras"]["raw_action"])
    behaviour_action_log_probs = data.extras["policy_extras"]["log_prob"]

    vs, advantages = compute_gae(truncation=truncation, termination=termination, rewards=rewards,
                                 values=baseline.detach(), bootstrap_value=bootstrap_value.detach(),
                                 lambda_=gae_lambda, discount=discounting)
    if normalize_advantage:
        advantages = (advantages - advantages.mean()) / (advantages.std(unbiased=False) + 1e-8)
    rho_s = torch.exp(target_action_log_probs - behaviour_action_log_probs)
    surrogate_loss1 = rho_s * advantages
    surrogate_loss2 = rho_s.clamp(1 - clipping_epsilon, 1 + clipping_epsilon) * advantages
    policy_loss = -torch.mean(torch.minimum(surrogate_loss1, surrogate_loss2))

    v_error = vs - baseline
    v_loss = torch.mean(v_error * v_error) * 0.5 * 0.5  # :181-182

    entropy = torch.mean(dist.entropy(policy_logits, draw("entropy", (T, B, dist.event_size))))
    entropy_loss = entropy_cost * -entropy
    kl_intention = kl_weight * kl_divergence(intention_mean, intention_logvar)

    total_loss = policy_loss + v_loss + entropy_loss + kl_intention
    with torch.no_grad():
        explained_variance = 1.0 - (v_loss / rewards.var(unbiased=False))
        # jnp.corrcoef(vs, rewards) is a (2T x 2T) matrix that the trainer later averages (:189, C.14)
        prediction_corr = _corrcoef(torch.cat([vs, rewards], dim=0)).mean() if T * 2 <= 256 else torch.zeros((), device=dev)
    return total_loss, {
        "total_loss": total_loss.detach(),
        "policy_loss": policy_loss.detach(),
        "v_loss": v_loss.detach(),
        "entropy_loss": entropy_loss.detach(),
        "kl_loss_intention": kl_intention.detach(),
        "prediction_corr": prediction_corr,
        "explained_variance": explained_variance,
    }
