"""Fused HIP inference kernel for the intention policy (C-ABI vnl_policy_*, csrc/vnl_policy.hip).

Used by `make_inference_fn` for acting on a HIP device; training (autograd) keeps the torch path,
which computes the same function (tests/test_gpu_policy.py compares the two)."""
from __future__ import annotations

import ctypes as C
from typing import Optional, Tuple

import torch

from .. import _lib
from .intention_policy_network import IntentionNetwork


class HipIntentionPolicy:
    def __init__(self, net: IntentionNetwork, action_size: int, max_batch: int, device: torch.device):
        self.lib = _lib.load_library()
        self.net, self.action_size, self.device = net, action_size, torch.device(device)
        spec = _lib.PolicySpec()
        spec.traj_size, spec.obs_size, spec.action_size, spec.latent_size = net.traj_size, net.obs_size, action_size, net.latents
        spec.num_encoder_layers, spec.num_decoder_layers = len(net.encoder_layers), len(net.decoder_layers)
        for i, h in enumerate(net.encoder_layers):
            spec.encoder_layers[i] = h
        for i, h in enumerate(net.decoder_layers):
            spec.decoder_layers[i] = h
        self.h = C.c_void_p()
        idx = self.device.index if self.device.index is not None else torch.cuda.current_device()
        _lib.check(self.lib, self.lib.vnl_policy_create(C.byref(spec), int(max_batch), idx, C.byref(self.h)))
        n = self.lib.vnl_policy_num_params(self.h)
        if n != net.num_params:
            raise _lib.VnlError(f"parameter layout mismatch: kernel expects {n}, network has {net.num_params}")
        self.max_batch = int(max_batch)

    def __del__(self):
        try:
            if getattr(self, "h", None):
                self.lib.vnl_policy_destroy(self.h)
        except Exception:
            pass

    @torch.no_grad()
    def forward(self, params: torch.Tensor, obs_mean: Optional[torch.Tensor], obs_std: Optional[torch.Tensor],
                traj: torch.Tensor, obs: torch.Tensor, eps_latent: torch.Tensor, eps_action: Optional[torch.Tensor],
                deterministic: bool = False, rand_action: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, dict]:
        B, na, nl = obs.shape[0], self.action_size, self.net.latents
        f32 = dict(dtype=torch.float32, device=obs.device)
        c = lambda t: t.contiguous()  # noqa: E731
        traj, obs, eps_latent, params = c(traj), c(obs), c(eps_latent), c(params)
        action = torch.empty((B, na), **f32)
        logits = torch.empty((B, 2 * na), **f32)
        lat_mean, lat_logvar = torch.empty((B, nl), **f32), torch.empty((B, nl), **f32)
        raw = torch.empty((B, na), **f32) if not deterministic else None
        lp = torch.empty((B,), **f32) if not deterministic else None
        rlp = torch.empty((B,), **f32) if (rand_action is not None and not deterministic) else None
        if rlp is None:
            rand_action = None
        else:
            rand_action = c(rand_action)
        if not deterministic:
            eps_action = c(eps_action)
        if obs_mean is not None:
            obs_mean, obs_std = c(obs_mean), c(obs_std)
        p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)  # noqa: E731
        stream = C.c_void_p(torch.cuda.current_stream(obs.device).cuda_stream)
        _lib.check(self.lib, self.lib.vnl_policy_forward(
            self.h, p(params), p(obs_mean), p(obs_std), p(traj), p(obs), p(eps_latent), p(eps_action), B,
            int(deterministic), p(action), p(raw), p(lp), p(logits), p(lat_mean), p(lat_logvar), p(rand_action),
            p(rlp), stream))
        self._hold = (traj, obs, eps_latent, eps_action, params, obs_mean, obs_std, rand_action)
        extras = {} if deterministic else {"log_prob": lp, "raw_action": raw, "logits": logits}
        if rlp is not None:
            extras["rand_log_prob"] = rlp
        return action, {**extras, "latent_mean": lat_mean, "latent_logvar": lat_logvar}
