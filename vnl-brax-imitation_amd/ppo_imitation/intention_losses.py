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


def _corr_fits(T: int, B: int) -> bool:
    """prediction_corr (a 2T x 2T correlation matrix later averaged, reference intention_losses.py:186-188) is computed when
    the 2T rows of B samples fit the HIP kernel's LDS budget -- ONE rule for both update backends (csrc/vnl_ppo.hip,
    prediction_corr_kernel); otherwise the metric is NaN ("not computed"), never a fake 0.0."""
    return (2 * T * B + 2 * T) * 4 <= 60 * 1024


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


class _PPOHead(torch.autograd.Function):
    """vnl_ppo_head (csrc/vnl_lib.hip): loss head + its gradients in one launch.  forward() returns the total
    loss and stashes d loss / d (logits, baseline, latent mean, latent logvar) for backward()."""

    @staticmethod
    def forward(ctx, logits, baseline, lat_mean, lat_logvar, bootstrap, raw_action, behaviour_lp, reward, truncation,
                discount, eps_entropy, cfg, lib):
        import ctypes as C

        from .. import _lib

        T, B = baseline.shape
        c = lambda t: t.detach().contiguous()  # noqa: E731
        ins = [c(t) for t in (logits, baseline, bootstrap, lat_mean, lat_logvar, raw_action, behaviour_lp, reward,
                              truncation, discount, eps_entropy)]
        g = [torch.empty_like(ins[0]), torch.empty_like(ins[1]), torch.empty_like(ins[3]), torch.empty_like(ins[4])]
        vs, adv = torch.empty_like(ins[1]), torch.empty_like(ins[1])
        metrics = torch.empty(8, dtype=torch.float32, device=baseline.device)
        work = torch.empty(_lib.PPO_HEAD_WORKSPACE_FLOATS, dtype=torch.float32, device=baseline.device)
        a = _lib.PPOHeadArgs()
        a.T, a.B, a.act, a.latent = T, B, raw_action.shape[-1], lat_mean.shape[-1]
        ptr = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
        (a.logits, a.baseline, a.bootstrap, a.lat_mean, a.lat_logvar, a.raw_action, a.behaviour_log_prob, a.reward,
         a.truncation, a.discount, a.eps_entropy) = [ptr(t) for t in ins]
        for k, v in cfg.items():
            setattr(a, k, v)
        a.g_logits, a.g_baseline, a.g_lat_mean, a.g_lat_logvar = [ptr(t) for t in g]
        a.vs, a.advantages, a.metrics = ptr(vs), ptr(adv), ptr(metrics)
        dev = baseline.device
        stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream) if dev.type == "cuda" else C.c_void_p(0)
        _lib.check(lib, lib.vnl_ppo_head(C.byref(a), ptr(work), stream))
        ctx.save_for_backward(*g)
        ctx.hold = ins + [work]  # buffers of asynchronous launches
        ctx.mark_non_differentiable(vs, metrics)
        return metrics[0].clone(), vs, metrics

    @staticmethod
    def backward(ctx, g_loss, _g_vs, _g_metrics):
        gl, gb, gm, gv = ctx.saved_tensors
        return (gl * g_loss, gb * g_loss, gm * g_loss, gv * g_loss) + (None,) * 9


def _head_library(dev):
    """The native library when it can run the loss head on `dev` (HIP device, float32); None otherwise."""
    if dev.type != "cuda":
        return None
    from .. import _lib

    lib = _lib.load_library()
    return lib if hasattr(lib, "vnl_ppo_head") else None


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
    head: Any = "auto",
    time_major: bool = False,
) -> Tuple[torch.Tensor, Dict[str, torch.Tensor]]:
    """intention_losses.py:91-202.  `data` has leading dims [B, T]; `noise` optionally supplies the
    two Gaussian draws ('latent' [T,B,latent], 'entropy' [T,B,act]) for reproducible tests.
    `head`: "auto" = the fused HIP loss head (vnl_ppo_head) on a HIP device, "torch" = op by op (the
    reference's formulation, also what autograd-checks the kernel in the tests), or a loaded library."""
    dist = ppo_network.parametric_action_distribution
    policy_apply = ppo_network.policy_network.apply
    value_apply = ppo_network.value_network.apply

    if not time_major:  # (the trainer's captured step hands over [T, B, ...] directly: no strided copies)
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
    # one value-network pass over the T observations and the bootstrap observation (:135-137 calls it twice)
    values_all = value_apply(normalizer_params, params.value, torch.cat([obs, data.next_observation[-1:]], dim=0))
    baseline, bootstrap_value = values_all[:-1], values_all[-1].detach()

    truncation = data.extras["state_extras"]["truncation"]
    lib = _head_library(dev) if head == "auto" else (None if head == "torch" else head)
    if lib is not None:
        cfg = dict(entropy_cost=entropy_cost, discounting=discounting, reward_scaling=reward_scaling,
                   gae_lambda=gae_lambda, clipping_epsilon=clipping_epsilon, kl_weight=kl_weight, min_std=dist._min_std,
                   var_scale=dist._var_scale, normalize_advantage=int(normalize_advantage))
        total_loss, vs, mt = _PPOHead.apply(
            policy_logits, baseline, intention_mean, intention_logvar, bootstrap_value,
            data.extras["policy_extras"]["raw_action"], data.extras["policy_extras"]["log_prob"], data.reward, truncation,
            data.discount, draw("entropy", (T, B, dist.event_size)), cfg, lib)
        with torch.no_grad():
            rewards = data.reward * reward_scaling
            prediction_corr = _corrcoef(torch.cat([vs, rewards], dim=0)).mean() if _corr_fits(T, vs.shape[1]) else torch.full((), float('nan'), device=dev)
        return total_loss, {"total_loss": mt[0], "policy_loss": mt[1], "v_loss": mt[2], "entropy_loss": mt[3],
                            "kl_loss_intention": mt[4], "prediction_corr": prediction_corr, "explained_variance": mt[5]}

    rewards = data.reward * reward_scaling
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
        prediction_corr = _corrcoef(torch.cat([vs, rewards], dim=0)).mean() if _corr_fits(T, vs.shape[1]) else torch.full((), float('nan'), device=dev)
    return total_loss, {
        "total_loss": total_loss.detach(),
        "policy_loss": policy_loss.detach(),
        "v_loss": v_loss.detach(),
        "entropy_loss": entropy_loss.detach(),
        "kl_loss_intention": kl_intention.detach(),
        "prediction_corr": prediction_corr,
        "explained_variance": explained_variance,
    }
