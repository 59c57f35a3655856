"""Hand-written HIP forward + backward of one PPO minibatch step (C-ABI vnl_ppo_update_*, csrc/vnl_ppo.hip): the
gradient of compute_ppo_intention_loss (reference ppo_imitation/intention_losses.py:91-202) w.r.t. the flat
[policy | value] parameter buffer, as the reference obtains it from jax.grad inside brax's gradient_update_fn
(ppo_imitation/train.py:255-268).  No CPU fallback: construction fails without the HIP library / a HIP device."""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional

import torch

from .. import _lib
from .acting import Transition


def supported(ppo_network) -> bool:
    """The fused update implements exactly the networks of make_intention_ppo_networks with the running-statistics
    normaliser or the identity as observation preprocessor, within the limits vnl_ppo_update_create checks (csrc/vnl_ppo.hip):
    1..8 layers per stack, widths <= 1024, the decoder's last layer = the distribution's 2 x action_size parameters."""
    if not (bool(getattr(ppo_network, "hip_ok", False)) and ppo_network.policy_module is not None and
            ppo_network.value_module is not None):
        return False
    pol, val, dist = ppo_network.policy_module, ppo_network.value_module, ppo_network.parametric_action_distribution
    stacks = (list(pol.encoder_layers), list(pol.decoder_layers), list(val.sizes[:-1]))
    if any(not (1 <= len(s) <= 8) for s in stacks):
        return False
    if max(w for s in stacks for w in s) > 1024:
        return False
    return pol.decoder_layers[-1] == 2 * dist.event_size


class HipPPOUpdate:
    def __init__(self, ppo_network, T: int, B: int, device, *, entropy_cost, discounting, reward_scaling, gae_lambda,
                 clipping_epsilon, normalize_advantage, kl_weight):
        dev = torch.device(device)
        if dev.type != "cuda":
            raise _lib.VnlError("HipPPOUpdate runs on a HIP device only; no CPU fallback")
        self.lib = _lib.load_library()
        pol, val, dist = ppo_network.policy_module, ppo_network.value_module, ppo_network.parametric_action_distribution
        sp = _lib.PPONetSpec()
        sp.traj_size, sp.obs_size, sp.action_size, sp.latent_size = pol.traj_size, pol.obs_size, dist.event_size, pol.latents
        sp.num_encoder_layers, sp.num_decoder_layers = len(pol.encoder_layers), len(pol.decoder_layers)
        sp.num_value_layers = len(val.sizes) - 1
        if dev.index is None:  # a bare 'cuda': the CURRENT device, not GPU 0
            dev = torch.device("cuda", torch.cuda.current_device())
        for i, h in enumerate(pol.encoder_layers):
            sp.encoder_layers[i] = h
        for i, h in enumerate(pol.decoder_layers):
            sp.decoder_layers[i] = h
        for i, h in enumerate(val.sizes[:-1]):
            sp.value_layers[i] = h
        self.T, self.B, self.device = T, B, dev
        self.h = C.c_void_p()
        _lib.check(self.lib, self.lib.vnl_ppo_update_create(C.byref(sp), T, B, dev.index, C.byref(self.h)))
        self.num_params = int(self.lib.vnl_ppo_update_num_params(self.h))
        assert self.num_params == ppo_network.policy_network.layout.size + ppo_network.value_network.layout.size
        self.normalizes = bool(ppo_network.normalizes)
        hp = _lib.PPOHParams()
        hp.entropy_cost, hp.discounting, hp.reward_scaling, hp.gae_lambda = entropy_cost, discounting, reward_scaling, gae_lambda
        hp.clipping_epsilon, hp.kl_weight, hp.min_std, hp.var_scale = clipping_epsilon, kl_weight, dist._min_std, dist._var_scale
        hp.normalize_advantage = int(normalize_advantage)
        self.hp = hp
        self.metrics = torch.zeros(9, dtype=torch.float32, device=dev)

    def __del__(self):
        try:
            if getattr(self, "h", None):
                self.lib.vnl_ppo_update_destroy(self.h)
        except Exception:
            pass

    def buffer(self, name: str) -> torch.Tensor:
        """Copy of an intermediate of the last call ("vs", "advantages", "values", "logits", "latent_mean", ...)."""
        ptr, cnt = C.c_void_p(), C.c_int64()
        _lib.check(self.lib, self.lib.vnl_ppo_update_buffer(self.h, name.encode(), C.byref(ptr), C.byref(cnt)))
        out = torch.empty(cnt.value, dtype=torch.float32, device=self.device)
        stream = torch.cuda.current_stream(self.device).cuda_stream
        rc = C.CDLL("libamdhip64.so").hipMemcpyAsync(C.c_void_p(out.data_ptr()), ptr, C.c_size_t(4 * cnt.value), C.c_int(3),
                                                     C.c_void_p(stream))
        if rc != 0:
            raise _lib.VnlError(f"hipMemcpyAsync failed: {rc}")
        return out

    def grad(self, params: torch.Tensor, normalizer_params, data: Transition, noise: Dict[str, torch.Tensor],
             grads: torch.Tensor, part: int = 0) -> torch.Tensor:
        """`data`: TIME-MAJOR Transition [T, B, ...] (next_observation: at least its last row [.., B, obs]); `noise`:
        {"latent": [T,B,latent], "entropy": [T,B,act]} N(0,1) draws.  Writes d loss / d params into `grads` (flat, same
        layout) and returns the metrics tensor [9] (total, policy, value, entropy, KL losses, explained variance, advantage mean / std,
        prediction_corr).  `part`: 0 = the whole step; 1 = forward + loss head + the value network's backward (the value
        segment of `grads` is final afterwards), 2 = the policy network's backward (after part 1, same arguments): the
        data-parallel trainer all-reduces the value segment between the two."""
        T, B = self.T, self.B
        assert params.is_contiguous() and grads.is_contiguous() and params.numel() == grads.numel() == self.num_params
        c = lambda t: t if t.is_contiguous() else t.contiguous()  # noqa: E731
        ex = data.extras
        keep = [c(ex["state_extras"]["traj"]), c(data.observation), c(data.next_observation[-1]),
                c(ex["policy_extras"]["raw_action"]), c(ex["policy_extras"]["log_prob"]), c(data.reward),
                c(ex["state_extras"]["truncation"]), c(data.discount), c(noise["latent"]), c(noise["entropy"])]
        assert keep[1].shape[:2] == (T, B), (keep[1].shape, T, B)
        b = _lib.PPOBatch()
        (b.traj, b.obs, b.next_obs_last, b.raw_action, b.behaviour_log_prob, b.reward, b.truncation, b.discount, b.eps_latent,
         b.eps_entropy) = [C.c_void_p(t.data_ptr()) for t in keep]
        if self.normalizes and normalizer_params is not None:
            keep += [c(normalizer_params.mean), c(normalizer_params.std)]
            b.obs_mean, b.obs_std = C.c_void_p(keep[-2].data_ptr()), C.c_void_p(keep[-1].data_ptr())
        stream = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        _lib.check(self.lib, self.lib.vnl_ppo_minibatch_grad_part(self.h, C.c_void_p(params.data_ptr()), C.byref(b), C.byref(self.hp),
                                                                  C.c_void_p(grads.data_ptr()), C.c_void_p(self.metrics.data_ptr()),
                                                                  stream, int(part)))
        self._hold = keep  # buffers of asynchronous launches
        return self.metrics
