"""NumPy float64 restatement of the PPO-side arithmetic (TEST INFRASTRUCTURE ONLY).

Follows, line by line:
  * IntentionNetwork forward      reference ppo_imitation/intention_policy_network.py:20-105
  * NormalTanhDistribution        brax.training.distribution [UPSTREAM], built at ppo_networks.py:102
  * value MLP                     brax.training.networks.make_value_network [UPSTREAM], ppo_networks.py:114
  * compute_gae                   reference ppo_imitation/intention_losses.py:26-87
  * compute_ppo_intention_loss    reference ppo_imitation/intention_losses.py:91-202
  * running_statistics.update     brax.training.acme.running_statistics [UPSTREAM]
Parity status: unpinned by reference data (the reference has no tests / stored outputs for these);
the formulas are restated independently of the torch product code (explicit loops, no autograd).
"""
from __future__ import annotations

import numpy as np

LN_EPS = 1e-6


def layer_norm(x, scale, bias):
    mu = x.mean(-1, keepdims=True)
    var = ((x - mu) ** 2).mean(-1, keepdims=True)
    return (x - mu) / np.sqrt(var + LN_EPS) * scale + bias


def softplus(x):
    return np.logaddexp(0.0, x)


def policy_forward(P: dict, enc_layers, dec_layers, traj, obs, eps_latent):
    """P: name -> ndarray with the Flax names (encoder/hidden_0/kernel ...)."""
    x = traj
    for i in range(len(enc_layers)):
        x = np.maximum(x @ P[f"encoder/hidden_{i}/kernel"] + P[f"encoder/hidden_{i}/bias"], 0.0)
        x = layer_norm(x, P[f"encoder/LayerNorm_{i}/scale"], P[f"encoder/LayerNorm_{i}/bias"])
    mean = x @ P["encoder/fc2_mean/kernel"] + P["encoder/fc2_mean/bias"]
    logvar = x @ P["encoder/fc2_logvar/kernel"] + P["encoder/fc2_logvar/bias"]
    z = mean + eps_latent * np.exp(0.5 * logvar)
    x = np.concatenate([z, obs], axis=-1)
    n = len(dec_layers)
    for i in range(n):
        x = x @ P[f"decoder/hidden_{i}/kernel"] + P[f"decoder/hidden_{i}/bias"]
        if i != n - 1:
            x = layer_norm(np.maximum(x, 0.0), P[f"decoder/LayerNorm_{i}/scale"], P[f"decoder/LayerNorm_{i}/bias"])
    return x, mean, logvar


def value_forward(V: dict, n_layers: int, obs):
    x = obs
    for i in range(n_layers):
        x = x @ V[f"hidden_{i}/kernel"] + V[f"hidden_{i}/bias"]
        if i != n_layers - 1:
            x = x / (1.0 + np.exp(-x))  # swish
    return x[..., 0]


def tanh_normal_log_prob(logits, raw, min_std=0.001):
    loc, s = np.split(logits, 2, axis=-1)
    scale = softplus(s) + min_std
    lp = -0.5 * ((raw - loc) / scale) ** 2 - 0.5 * np.log(2 * np.pi) - np.log(scale)
    lp = lp - 2.0 * (np.log(2.0) - raw - softplus(-2.0 * raw))
    return lp.sum(-1)


def tanh_normal_entropy(logits, eps, min_std=0.001):
    loc, s = np.split(logits, 2, axis=-1)
    scale = softplus(s) + min_std
    x = loc + scale * eps
    ent = 0.5 + 0.5 * np.log(2 * np.pi) + np.log(scale)
    return (ent + 2.0 * (np.log(2.0) - x - softplus(-2.0 * x))).sum(-1)


def compute_gae(truncation, termination, rewards, values, bootstrap_value, lambda_, discount):
    T = rewards.shape[0]
    mask = 1 - truncation
    v_next = np.concatenate([values[1:], bootstrap_value[None]], 0)
    deltas = (rewards + discount * (1 - termination) * v_next - values) * mask
    acc = np.zeros_like(bootstrap_value)
    vs_minus_v = np.zeros_like(values)
    for t in reversed(range(T)):
        acc = deltas[t] + discount * (1 - termination[t]) * mask[t] * lambda_ * acc
        vs_minus_v[t] = acc
    vs = vs_minus_v + values
    vs_next = np.concatenate([vs[1:], bootstrap_value[None]], 0)
    adv = (rewards + discount * (1 - termination) * vs_next - values) * mask
    return vs, adv


def ppo_intention_loss(P, V, enc_layers, dec_layers, n_val, norm_mean, norm_std, data: dict, eps_latent, eps_entropy,
                       entropy_cost, discounting, reward_scaling, gae_lambda, clipping_epsilon, normalize_advantage,
                       kl_weight):
    """data: dict of [B, T, ...] arrays (observation, next_observation, reward, discount, truncation, traj,
    raw_action, log_prob); eps_* are time-major [T, B, ...]."""
    d = {k: np.swapaxes(v, 0, 1) for k, v in data.items()}
    nrm = lambda o: (o - norm_mean) / norm_std  # noqa: E731
    logits, mean, logvar = policy_forward(P, enc_layers, dec_layers, d["traj"], nrm(d["observation"]), eps_latent)
    baseline = value_forward(V, n_val, nrm(d["observation"]))
    bootstrap = value_forward(V, n_val, nrm(d["next_observation"][-1]))
    rewards = d["reward"] * reward_scaling
    termination = (1 - d["discount"]) * (1 - d["truncation"])
    target_lp = tanh_normal_log_prob(logits, d["raw_action"])
    vs, adv = compute_gae(d["truncation"], termination, rewards, baseline, bootstrap, gae_lambda, discounting)
    if normalize_advantage:
        adv = (adv - adv.mean()) / (adv.std() + 1e-8)
    rho = np.exp(target_lp - d["log_prob"])
    policy_loss = -np.mean(np.minimum(rho * adv, np.clip(rho, 1 - clipping_epsilon, 1 + clipping_epsilon) * adv))
    v_loss = np.mean((vs - baseline) ** 2) * 0.5 * 0.5
    entropy_loss = entropy_cost * -np.mean(tanh_normal_entropy(logits, eps_entropy))
    kl = kl_weight * (-0.5 * np.mean(1 + logvar - mean ** 2 - np.exp(logvar)))
    return dict(total_loss=policy_loss + v_loss + entropy_loss + kl, policy_loss=policy_loss, v_loss=v_loss,
                entropy_loss=entropy_loss, kl_loss_intention=kl, vs=vs, advantages=adv)


def running_update(count, mean, summed_variance, batch, std_min=1e-6, std_max=1e6):
    x = batch.reshape(-1, batch.shape[-1])
    count = count + x.shape[0]
    d_old = x - mean
    mean = mean + d_old.sum(0) / count
    summed_variance = summed_variance + (d_old * (x - mean)).sum(0)
    std = np.clip(np.sqrt(summed_variance / count), std_min, std_max)
    return count, mean, summed_variance, std
