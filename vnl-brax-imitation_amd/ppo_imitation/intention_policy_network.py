"""Intention (VAE-style) policy network: counterpart of reference
ppo_imitation/intention_policy_network.py:20-136.

  Encoder (ipn:20-44)  traj -> [Dense -> ReLU -> LayerNorm] x len(layers) -> fc2_mean, fc2_logvar
  reparameterize (:73-76)  z = mean + eps * exp(0.5 logvar)
  Decoder (ipn:47-70)  [z | obs] -> [Dense -> ReLU -> LayerNorm] x (n-1) -> Dense   (no activation on the last)

Parameters live in ONE flat float32 buffer (views carved by `layout`), so the data-parallel
gradient exchange is a single contiguous RCCL all-reduce and the buffer can be handed to the
C-ABI policy kernel as-is.  Names follow the Flax tree of the reference (`encoder/hidden_0/kernel`,
`encoder/LayerNorm_0/scale`, `encoder/fc2_mean/kernel`, `decoder/hidden_2/bias`, ...) so a
reference checkpoint maps 1:1.  Dense kernels are stored (in, out) as Flax does.
"""
from __future__ import annotations

import math
from collections import OrderedDict
from typing import Dict, List, Sequence, Tuple

import torch
import torch.nn.functional as F

LN_EPS = 1e-6  # flax.linen.LayerNorm default epsilon

Layout = "OrderedDict[str, Tuple[int, Tuple[int, ...]]]"


class ParamLayout:
    """name -> (offset, shape) inside a flat buffer."""

    def __init__(self):
        self.entries: "OrderedDict[str, Tuple[int, Tuple[int, ...]]]" = OrderedDict()
        self.size = 0

    def add(self, name: str, shape: Sequence[int]) -> None:
        n = int(math.prod(shape))
        self.entries[name] = (self.size, tuple(shape))
        self.size += n

    def view(self, flat, name: str) -> torch.Tensor:
        if isinstance(flat, LeafParams):
            return flat.leaves[name]
        off, shape = self.entries[name]
        return flat[off:off + int(math.prod(shape))].view(shape)

    def views(self, flat: torch.Tensor) -> Dict[str, torch.Tensor]:
        return {k: self.view(flat, k) for k in self.entries}


class LeafParams:
    """A flat parameter buffer presented to autograd as one LEAF per tensor.

    Differentiating through `flat[off:off+n].view(shape)` makes every tensor's backward allocate and add a
    zero-filled copy of the WHOLE buffer (34 tensors x 6.5 MB x fill/copy/add per minibatch step).  Here each
    tensor is a leaf that aliases its slice of `flat`; after backward() the per-tensor gradients are moved into
    one flat gradient buffer by a single multi-tensor copy, so the optimiser / all-reduce see flat buffers."""

    def __init__(self, layout: "ParamLayout", flat: torch.Tensor, flat_grad: torch.Tensor):
        assert not flat.requires_grad and flat.shape == flat_grad.shape == (layout.size,)
        self.layout, self.flat, self.flat_grad = layout, flat, flat_grad
        self.leaves: Dict[str, torch.Tensor] = {}
        self.grad_views = []
        for name, (off, shape) in layout.entries.items():
            n = int(math.prod(shape))
            leaf = flat[off:off + n].view(shape).detach().requires_grad_(True)
            self.leaves[name] = leaf
            self.grad_views.append(flat_grad[off:off + n].view(shape))

    def zero_grad(self) -> None:
        """Before backward(): with .grad unset autograd hands each leaf its gradient tensor as is (no add)."""
        for leaf in self.leaves.values():
            leaf.grad = None

    def gather_grads(self) -> None:
        """After backward(): all per-tensor gradients into the flat gradient buffer, one multi-tensor copy."""
        leaves = list(self.leaves.values())
        if any(leaf.grad is None for leaf in leaves):
            self.flat_grad.zero_()
        pairs = [(v, leaf.grad) for v, leaf in zip(self.grad_views, leaves) if leaf.grad is not None]
        torch._foreach_copy_([d for d, _ in pairs], [g for _, g in pairs])


def lecun_uniform_(t: torch.Tensor, fan_in: int, gen: torch.Generator) -> None:
    lim = math.sqrt(3.0 / fan_in)
    t.copy_((torch.rand(t.shape, generator=gen) * 2 - 1) * lim)


def lecun_normal_(t: torch.Tensor, fan_in: int, gen: torch.Generator) -> None:
    # jax variance_scaling(1.0, "fan_in", "truncated_normal"): stddev = sqrt(1/fan_in) / .87962566103423978
    std = math.sqrt(1.0 / fan_in) / 0.87962566103423978
    x = torch.empty(t.shape)
    torch.nn.init.trunc_normal_(x, mean=0.0, std=1.0, a=-2.0, b=2.0, generator=gen)
    t.copy_(x * std)


class IntentionNetwork:
    """Functional network over a flat parameter buffer (ipn:79-105)."""

    def __init__(self, traj_size: int, obs_size: int, encoder_layers: Sequence[int], decoder_layers: Sequence[int],
                 latents: int = 60):
        self.traj_size, self.obs_size, self.latents = traj_size, obs_size, latents
        self.encoder_layers, self.decoder_layers = list(encoder_layers), list(decoder_layers)
        L = ParamLayout()
        fan = traj_size
        for i, h in enumerate(self.encoder_layers):
            L.add(f"encoder/hidden_{i}/kernel", (fan, h)), L.add(f"encoder/hidden_{i}/bias", (h,))
            L.add(f"encoder/LayerNorm_{i}/scale", (h,)), L.add(f"encoder/LayerNorm_{i}/bias", (h,))
            fan = h
        L.add("encoder/fc2_mean/kernel", (fan, latents)), L.add("encoder/fc2_mean/bias", (latents,))
        L.add("encoder/fc2_logvar/kernel", (fan, latents)), L.add("encoder/fc2_logvar/bias", (latents,))
        fan = latents + obs_size
        for i, h in enumerate(self.decoder_layers):
            L.add(f"decoder/hidden_{i}/kernel", (fan, h)), L.add(f"decoder/hidden_{i}/bias", (h,))
            if i != len(self.decoder_layers) - 1:
                L.add(f"decoder/LayerNorm_{i}/scale", (h,)), L.add(f"decoder/LayerNorm_{i}/bias", (h,))
            fan = h
        self.layout = L

    @property
    def num_params(self) -> int:
        return self.layout.size

    def init(self, gen: torch.Generator) -> torch.Tensor:
        flat = torch.zeros(self.layout.size, dtype=torch.float32)
        for name, (off, shape) in self.layout.entries.items():
            v = self.layout.view(flat, name)
            if name.endswith("/kernel"):
                if "fc2_" in name:
                    lecun_normal_(v, shape[0], gen)  # nn.Dense default kernel_init (ipn:42-43)
                else:
                    lecun_uniform_(v, shape[0], gen)  # ipn:25,52
            elif name.endswith("/scale"):
                v.fill_(1.0)
        return flat

    def encode(self, flat: torch.Tensor, traj: torch.Tensor):
        P = self.layout
        x = traj
        for i, h in enumerate(self.encoder_layers):
            x = F.relu(torch.addmm(P.view(flat, f"encoder/hidden_{i}/bias"), x.reshape(-1, x.shape[-1]),
                                   P.view(flat, f"encoder/hidden_{i}/kernel")).view(*x.shape[:-1], h))
            x = F.layer_norm(x, (h,), P.view(flat, f"encoder/LayerNorm_{i}/scale"),
                             P.view(flat, f"encoder/LayerNorm_{i}/bias"), LN_EPS)
        mean = x @ P.view(flat, "encoder/fc2_mean/kernel") + P.view(flat, "encoder/fc2_mean/bias")
        logvar = x @ P.view(flat, "encoder/fc2_logvar/kernel") + P.view(flat, "encoder/fc2_logvar/bias")
        return mean, logvar

    def decode(self, flat: torch.Tensor, x: torch.Tensor) -> torch.Tensor:
        P = self.layout
        n = len(self.decoder_layers)
        for i, h in enumerate(self.decoder_layers):
            x = torch.addmm(P.view(flat, f"decoder/hidden_{i}/bias"), x.reshape(-1, x.shape[-1]),
                            P.view(flat, f"decoder/hidden_{i}/kernel")).view(*x.shape[:-1], h)
            if i != n - 1:
                x = F.layer_norm(F.relu(x), (h,), P.view(flat, f"decoder/LayerNorm_{i}/scale"),
                                 P.view(flat, f"decoder/LayerNorm_{i}/bias"), LN_EPS)
        return x

    def apply(self, flat: torch.Tensor, traj: torch.Tensor, obs: torch.Tensor, eps_latent: torch.Tensor):
        """-> (action logits, intention_mean, intention_logvar)   (ipn:91-105)"""
        mean, logvar = self.encode(flat, traj)
        z = mean + eps_latent * torch.exp(0.5 * logvar)  # reparameterize, ipn:73-76
        return self.decode(flat, torch.cat([z, obs], dim=-1)), mean, logvar


def make_intention_policy(param_size: int, latent_size: int, obs_size: int, traj_size: int,
                          encoder_layer_sizes: Sequence[int] = (1024, 1024),
                          decoder_layer_sizes: Sequence[int] = (1024, 1024)) -> IntentionNetwork:
    """ipn:108-136."""
    return IntentionNetwork(traj_size, obs_size, list(encoder_layer_sizes), list(decoder_layer_sizes) + [param_size],
                            latents=latent_size)
