"""Checkpoint I/O for the PPO trainer.

The reference only saves, and only the inference params: `model.save_params(path, (normalizer_params,
policy_params))` from `policy_params_fn` and at the end of training (reference train.py:154-156,337-338;
brax.io.model pickles the pytree).  Here the same pair goes to one `.npz` (NumPy only, no pickle):

    normalizer/{count,mean,summed_variance,std}
    policy/params/<flax path>            e.g. policy/params/encoder/hidden_0/kernel   (Dense kernels stay (in, out))
    [value/params/<flax path>, optimizer/{mu,nu,count}, meta/env_steps]   full training state, for resume

The tensor names are the reference's Flax tree (intention_policy_network.py:20-136), so a reference checkpoint
converted to nested dicts of arrays maps 1:1 through `from_flax_tree` / `to_flax_tree`.
"""
from __future__ import annotations

from typing import Any, Dict, Optional, Tuple

import numpy as np
import torch

from . import running_statistics
from .intention_policy_network import ParamLayout

_NORM = ("count", "mean", "summed_variance", "std")


def to_flax_tree(layout: ParamLayout, flat: torch.Tensor) -> Dict[str, Any]:
    """Flat buffer -> {'params': nested dict} with the reference's module / parameter names."""
    tree: Dict[str, Any] = {}
    for name, (off, shape) in layout.entries.items():
        node = tree
        *path, leaf = name.split("/")
        for p in path:
            node = node.setdefault(p, {})
        node[leaf] = flat.detach()[off:off + int(np.prod(shape))].reshape(shape).cpu().numpy().copy()
    return {"params": tree}


def from_flax_tree(layout: ParamLayout, tree: Dict[str, Any]) -> torch.Tensor:
    """Inverse of `to_flax_tree`; raises on missing / extra / mis-shaped tensors."""
    tree = tree.get("params", tree)
    flat = torch.empty(layout.size, dtype=torch.float32)
    seen = set()
    for name, (off, shape) in layout.entries.items():
        node = tree
        for p in name.split("/"):
            if not isinstance(node, dict) or p not in node:
                raise KeyError(f"checkpoint lacks {name}")
            node = node[p]
        a = np.asarray(node, dtype=np.float32)
        if tuple(a.shape) != tuple(shape):
            raise ValueError(f"{name}: shape {a.shape}, expected {shape}")
        flat[off:off + a.size] = torch.from_numpy(a.reshape(-1))
        seen.add(name)

    def walk(node, prefix=""):
        for k, v in node.items():
            if isinstance(v, dict):
                yield from walk(v, prefix + k + "/")
            else:
                yield prefix + k
    extra = set(walk(tree)) - seen
    if extra:
        raise KeyError(f"checkpoint has tensors the network does not: {sorted(extra)[:5]}")
    return flat


def _flatten(prefix: str, layout: ParamLayout, flat: torch.Tensor, out: Dict[str, np.ndarray]) -> None:
    for name, (off, shape) in layout.entries.items():
        out[f"{prefix}/params/{name}"] = flat.detach()[off:off + int(np.prod(shape))].reshape(shape).cpu().numpy()


def save_params(path: str, params: Tuple[running_statistics.RunningStatisticsState, torch.Tensor], ppo_network,
                *, value_params: Optional[torch.Tensor] = None, optimizer_state: Optional[Dict[str, torch.Tensor]] = None,
                env_steps: Optional[int] = None) -> str:
    """`params` = the inference pair the reference saves; the keyword extras make the file resumable."""
    norm, policy = params
    out: Dict[str, np.ndarray] = {f"normalizer/{k}": getattr(norm, k).detach().cpu().numpy() for k in _NORM}
    _flatten("policy", ppo_network.policy_network.layout, policy, out)
    if value_params is not None:
        _flatten("value", ppo_network.value_network.layout, value_params, out)
    if optimizer_state is not None:
        for k, v in optimizer_state.items():
            out[f"optimizer/{k}"] = v.detach().cpu().numpy()
    if env_steps is not None:
        out["meta/env_steps"] = np.asarray(env_steps, dtype=np.int64)
    if not path.endswith(".npz"):
        path += ".npz"
    np.savez(path, **out)
    return path


def _unflatten(prefix: str, layout: ParamLayout, z) -> torch.Tensor:
    tree: Dict[str, Any] = {}
    for key in z.files:
        if key.startswith(prefix + "/params/"):
            node = tree
            *p, leaf = key[len(prefix) + 8:].split("/")
            for q in p:
                node = node.setdefault(q, {})
            node[leaf] = z[key]
    return from_flax_tree(layout, tree)


# ---- reference checkpoints (brax.io.model.save_params: pickle.dumps of (RunningStatisticsState, flax params)) -----------
class _Bag:
    """Inert stand-in for a pickled dataclass / flax struct: takes its fields, runs no code of the pickled class."""

    def __init__(self, *args, **kwargs):
        self._args = args
        self.__dict__.update(kwargs)

    def __setstate__(self, state):
        if isinstance(state, dict):
            self.__dict__.update(state)
        else:
            self._state = state


def _jax_array(fun, args, arr_state, aval_state):
    """jax._src.array._reconstruct_array(fun, args, arr_state, aval_state): fun / args rebuild the NumPy ndarray."""
    arr = fun(*args)
    arr.__setstate__(arr_state)
    return np.asarray(arr)


class _BraxUnpickler(__import__("pickle").Unpickler):
    """NumPy reconstruction and plain containers only; every other global (flax structs, FrozenDict, jax Arrays) becomes
    an inert bag or a NumPy array.  Nothing of jax / flax / brax is imported or executed."""

    _NUMPY = {("numpy.core.multiarray", "_reconstruct"), ("numpy._core.multiarray", "_reconstruct"), ("numpy", "ndarray"),
              ("numpy", "dtype"), ("numpy.core.multiarray", "scalar"), ("numpy._core.multiarray", "scalar")}

    def find_class(self, module, name):
        if (module, name) in self._NUMPY:
            return getattr(__import__(module, fromlist=[name]), name)
        if (module, name) == ("jax._src.array", "_reconstruct_array"):
            return _jax_array
        if (module, name) == ("collections", "OrderedDict"):
            import collections

            return collections.OrderedDict
        if name == "FrozenDict":  # flax.core.frozen_dict.FrozenDict(dict)
            return lambda *a, **k: dict(*a, **k)
        return _Bag


def _as_tree(x):
    if isinstance(x, _Bag):
        d = {k: v for k, v in vars(x).items() if not k.startswith("_")}
        if not d and getattr(x, "_args", None):
            return _as_tree(x._args[0]) if len(x._args) == 1 else [_as_tree(a) for a in x._args]
        return {k: _as_tree(v) for k, v in d.items()}
    if isinstance(x, dict):
        return {k: _as_tree(v) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_as_tree(v) for v in x]
    return x


def convert_brax_params(pickle_path: str, ppo_network, npz_path: Optional[str] = None):
    """A checkpoint of the REFERENCE trainer -- `brax.io.model.save_params(path, (normalizer_params, policy_params))`,
    reference train.py:154-156,337-338 -- to this package's params `(RunningStatisticsState, policy_flat)`, optionally
    written as the `.npz` of `save_params`.  The pickle is read with a restricted unpickler (no jax / flax / brax needed or
    executed); the policy tree must carry the reference's Flax names (`from_flax_tree` checks names and shapes)."""
    import io

    with open(pickle_path, "rb") as f:
        obj = _as_tree(_BraxUnpickler(io.BytesIO(f.read())).load())
    if not (isinstance(obj, (list, tuple)) and len(obj) == 2):
        raise ValueError("expected the pair (normalizer_params, policy_params)")
    norm_t, pol_t = obj
    arr = lambda v: torch.from_numpy(np.array(v, dtype=np.float32))  # noqa: E731
    norm = running_statistics.RunningStatisticsState(*(arr(norm_t[k]) for k in _NORM))
    flat = from_flax_tree(ppo_network.policy_network.layout, pol_t if "params" in pol_t else {"params": pol_t})
    if npz_path:
        save_params(npz_path, (norm, flat), ppo_network)
    return norm, flat


def load_params(path: str, ppo_network, device=None) -> Dict[str, Any]:
    """-> {'params': (normalizer, policy_flat), and when present 'value', 'optimizer', 'env_steps'}."""
    z = np.load(path if path.endswith(".npz") else path + ".npz", allow_pickle=False)
    t = lambda a: torch.from_numpy(np.asarray(a)).to(device) if device is not None else torch.from_numpy(np.asarray(a))  # noqa: E731
    norm = running_statistics.RunningStatisticsState(*(t(z[f"normalizer/{k}"]) for k in _NORM))
    out: Dict[str, Any] = {"params": (norm, _unflatten("policy", ppo_network.policy_network.layout, z).to(device))}
    if any(k.startswith("value/") for k in z.files):
        out["value"] = _unflatten("value", ppo_network.value_network.layout, z).to(device)
    opt = {k[len("optimizer/"):]: t(z[k]) for k in z.files if k.startswith("optimizer/")}
    if opt:
        out["optimizer"] = opt
    if "meta/env_steps" in z.files:
        out["env_steps"] = int(z["meta/env_steps"])
    return out
