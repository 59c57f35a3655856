"""PPO training with the intention policy: counterpart of reference ppo_imitation/train.py:62-491.

Same keyword signature and return triple `(make_policy, params, metrics)`; same loop structure
(unroll -> normaliser update -> num_updates_per_batch x num_minibatches SGD steps).

MI355X mapping of the reference's jax.pmap data parallelism (train.py:363):
  * one process per GPU (torch.distributed, backend "nccl" = RCCL over xGMI); `environment`
    is this rank's shard of the envs (its `num_envs` = per-rank envs);
  * C1: gradients live in ONE flat float32 buffer (policy | value), all-reduced (mean) per minibatch step in the two segments
    it becomes final in -- value network first, overlapping the policy network's backward, then the policy's, each followed
    by Adam on that segment (brax gradient_update_fn pmean, train.py:251-253,264-266: one pmean after the backward pass);
  * C2: normaliser statistics all-reduced once per training step (train.py:330-334);
  * C3: parameters broadcast from rank 0 at init (train.py:410-412);
  * C4: replicas are checked bit-identical at the end (train.py:485-487).
Nothing else crosses the links; rollout, GAE, advantage normalisation and shuffling are local.
"""
from __future__ import annotations

import dataclasses
import functools
import logging
import time
from typing import Any, Callable, Dict, Optional, Tuple

import numpy as np
import torch

from ..envs import wrappers as env_wrappers
from . import acting, ppo_networks, running_statistics
from . import intention_losses as ppo_losses

Metrics = Dict[str, Any]


@dataclasses.dataclass
class TrainingState:
    """train.py:38-45."""

    optimizer_state: Dict[str, torch.Tensor]
    params: torch.Tensor  # flat [policy | value]
    normalizer_params: running_statistics.RunningStatisticsState
    env_steps: int


class FlatAdam:
    """optax.adam(lr) [UPSTREAM] on one flat buffer: b1=0.9, b2=0.999, eps=1e-8, no weight decay."""

    def __init__(self, lr: float, b1: float = 0.9, b2: float = 0.999, eps: float = 1e-8, lib=None):
        self.lr, self.b1, self.b2, self.eps, self.lib = lr, b1, b2, eps, lib

    def init(self, params: torch.Tensor) -> Dict[str, torch.Tensor]:
        return {"mu": torch.zeros_like(params), "nu": torch.zeros_like(params),
                "count": torch.zeros((), dtype=torch.int64, device=params.device)}

    @torch.no_grad()
    def update(self, grads: torch.Tensor, state: Dict[str, torch.Tensor], params: torch.Tensor,
               segment: Optional[slice] = None, bump: bool = True) -> None:
        """No host read-back (the step count stays on the device), so the update can sit inside a hipGraph.
        On a HIP device (or with `self.lib` set) the update itself is one launch of vnl_adam_step.
        `segment`: update only that slice of the flat buffer (the data-parallel step updates the value segment as soon as its
        all-reduce is in, while the policy segment's is still in flight); `bump=False` on every call but the first of an
        optimiser step, so that the step count advances once."""
        if bump:
            state["count"] += 1
        if segment is not None:
            sub = {"mu": state["mu"][segment], "nu": state["nu"][segment], "count": state["count"]}
            return self.update(grads[segment], sub, params[segment], bump=False)
        lib = self.lib
        if lib is None and params.is_cuda:
            from .. import _lib

            lib = self.lib = _lib.load_library()
        if lib is not None and params.dtype == torch.float32 and params.is_contiguous() and grads.is_contiguous():
            import ctypes as C

            from .. import _lib

            ptr = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
            stream = C.c_void_p(torch.cuda.current_stream(params.device).cuda_stream) if params.is_cuda else C.c_void_p(0)
            _lib.check(lib, lib.vnl_adam_step(ptr(params), ptr(grads), ptr(state["mu"]), ptr(state["nu"]),
                                              ptr(state["count"]), params.numel(), self.lr, self.b1, self.b2, self.eps,
                                              stream))
            return
        t = state["count"].to(torch.float64)
        bc1 = (1 - torch.pow(torch.full_like(t, self.b1), t)).to(params.dtype)
        bc2 = (1 - torch.pow(torch.full_like(t, self.b2), t)).to(params.dtype)
        state["mu"].mul_(self.b1).add_(grads, alpha=1 - self.b1)
        state["nu"].mul_(self.b2).addcmul_(grads, grads, value=1 - self.b2)
        denom = (state["nu"] / bc2).sqrt_().add_(self.eps)
        params.sub_((state["mu"] / bc1).div_(denom).mul_(self.lr))


def _dist_info():
    import torch.distributed as dist

    if dist.is_available() and dist.is_initialized():
        return dist, dist.get_rank(), dist.get_world_size()
    return None, 0, 1


def train(
    environment,
    num_timesteps: int,
    episode_length: int,
    action_repeat: int = 1,
    num_envs: int = 1,
    max_devices_per_host: Optional[int] = None,
    num_eval_envs: int = 128,
    learning_rate: float = 1e-4,
    entropy_cost: float = 1e-4,
    discounting: float = 0.9,
    seed: int = 0,
    unroll_length: int = 10,
    batch_size: int = 32,
    num_minibatches: int = 16,
    num_updates_per_batch: int = 2,
    num_evals: int = 1,
    num_resets_per_eval: int = 0,
    normalize_observations: bool = False,
    reward_scaling: float = 1.0,
    clipping_epsilon: float = 0.3,
    gae_lambda: float = 0.95,
    deterministic_eval: bool = False,
    network_factory=ppo_networks.make_intention_ppo_networks,
    progress_fn: Callable[[int, Metrics], None] = lambda *args: None,
    normalize_advantage: bool = True,
    eval_env=None,
    policy_params_fn: Callable[..., None] = lambda *args: None,
    randomization_fn=None,
    kl_weight: float = 1e-4,
    reset_info_on_autoreset: bool = False,
    capture_graph: Optional[bool] = None,
    restore_from: Optional[str] = None,
    update_backend: str = "auto",
):
    """PPO training (train.py:62-491).

    `num_envs`, `batch_size` are GLOBAL counts as in the reference (train.py:128-129 scales them by
    the device count); each rank owns `num_envs // world` envs -- `environment.num_envs` must equal
    that -- and `batch_size // world` trajectories per minibatch.

    `restore_from`: a file written by `checkpoint.save_params` (full training state if it holds one, else
    the inference pair only) -- the reference cannot resume.

    `update_backend`: "hip" = the hand-written forward + backward of the minibatch step (csrc/vnl_ppo.hip through
    vnl_ppo_minibatch_grad: fp32 MFMA GEMMs with fused epilogues, gradients straight into the flat buffer), "torch" =
    autograd through the op-by-op loss (hipBLASLt), "auto" = hip on a HIP device for the networks of
    make_intention_ppo_networks.

    `capture_graph` (default: on for HIP devices): the minibatch step (gather -> loss -> backward
    [-> Adam when single-GPU]) is captured once into a hipGraph and replayed -- the eager step is
    ~500 launches of microsecond kernels and purely launch-bound.
    """
    dist, rank, world = _dist_info()
    assert batch_size * num_minibatches % num_envs == 0
    assert num_envs % world == 0 and batch_size % world == 0
    local_envs, local_batch = num_envs // world, batch_size // world
    base = environment.unwrapped
    assert base.num_envs == local_envs, f"environment.num_envs={base.num_envs}, expected {local_envs} per rank"
    device = base.device
    xt = time.time()

    env_step_per_training_step = batch_size * unroll_length * num_minibatches * action_repeat
    num_evals_after_init = max(num_evals - 1, 1)
    num_training_steps_per_epoch = int(np.ceil(
        num_timesteps / (num_evals_after_init * env_step_per_training_step * max(num_resets_per_eval, 1))))

    # keys: networks global (identical on all ranks), env / sgd streams per rank (train.py:178-187)
    g_net = torch.Generator(device="cpu").manual_seed(seed)
    g_env = torch.Generator(device="cpu").manual_seed(seed * 1000003 + 17 + rank)
    g_dev = torch.Generator(device=device).manual_seed(seed * 7919 + 101 + rank) if device.type == "cuda" else g_env
    g_eval = torch.Generator(device="cpu").manual_seed(seed * 31 + 7)

    env = env_wrappers.wrap(environment, episode_length=episode_length, action_repeat=action_repeat,
                            randomization_fn=randomization_fn, reset_info_on_autoreset=reset_info_on_autoreset)
    env_state = env.reset(g_env)

    normalize = (lambda x, y: x)
    if normalize_observations:
        normalize = running_statistics.normalize
    ppo_network = network_factory(env_state.info["traj"].shape[-1], env_state.obs.shape[-1], env.action_size,
                                  preprocess_observations_fn=normalize)
    make_policy = ppo_networks.make_inference_fn(ppo_network)

    n_pol, n_val = ppo_network.policy_network.layout.size, ppo_network.value_network.layout.size
    flat = torch.cat([ppo_network.policy_network.init(g_net), ppo_network.value_network.init(g_net)]).to(device)
    if dist is not None:
        dist.broadcast(flat, src=0)  # C3
    flat_grad = torch.zeros_like(flat)  # d loss / d flat; the per-tensor .grad of the leaves alias it
    optimizer = FlatAdam(learning_rate)
    training_state = TrainingState(
        optimizer_state=optimizer.init(flat.detach()),
        params=flat,
        normalizer_params=running_statistics.init_state(env_state.obs.shape[-1], device=device),
        env_steps=0,
    )

    if restore_from is not None:
        from . import checkpoint

        ck = checkpoint.load_params(restore_from, ppo_network, device=device)
        with torch.no_grad():
            flat[:n_pol].copy_(ck["params"][1])
            if "value" in ck:
                flat[n_pol:].copy_(ck["value"])
            for k, v in ck.get("optimizer", {}).items():
                training_state.optimizer_state[k].copy_(v)
        training_state.normalizer_params = ck["params"][0]
        training_state.env_steps = ck.get("env_steps", 0)

    from .intention_policy_network import LeafParams

    leaf_params = ppo_losses.PPONetworkParams(
        policy=LeafParams(ppo_network.policy_network.layout, flat[:n_pol], flat_grad[:n_pol]),
        value=LeafParams(ppo_network.value_network.layout, flat[n_pol:], flat_grad[n_pol:]))

    loss_fn = functools.partial(
        ppo_losses.compute_ppo_intention_loss, ppo_network=ppo_network, entropy_cost=entropy_cost,
        discounting=discounting, reward_scaling=reward_scaling, gae_lambda=gae_lambda,
        clipping_epsilon=clipping_epsilon, normalize_advantage=normalize_advantage, kl_weight=kl_weight)

    from . import hip_update

    assert update_backend in ("auto", "hip", "torch")
    use_hip_update = update_backend == "hip" or (update_backend == "auto" and device.type == "cuda" and
                                                  hip_update.supported(ppo_network))
    hip_upd: Dict[Any, Any] = {}
    if use_hip_update and update_backend == "auto":
        # "auto" never raises for a network the library turns down: the handle of the minibatch shape is made up front and the
        # torch path taken if vnl_ppo_update_create refuses it ("hip" still fails loudly)
        from .. import _lib as _vnl_lib

        try:
            hip_upd[(unroll_length, local_batch)] = hip_update.HipPPOUpdate(
                ppo_network, unroll_length, local_batch, device, entropy_cost=entropy_cost, discounting=discounting,
                reward_scaling=reward_scaling, gae_lambda=gae_lambda, clipping_epsilon=clipping_epsilon,
                normalize_advantage=normalize_advantage, kl_weight=kl_weight)
        except _vnl_lib.VnlError as e:
            logging.warning("hand-written PPO update unavailable for this network (%s): using the torch update", e)
            use_hip_update = False
    _METRIC_KEYS = ("total_loss", "policy_loss", "v_loss", "entropy_loss", "kl_loss_intention", "explained_variance")

    def hip_grad(data_tm: acting.Transition, normalizer_params, noise, part: int = 0) -> Metrics:
        """d loss / d flat -> flat_grad by the hand-written kernels; `data_tm` time-major [T, mb, ...]."""
        T, mb = data_tm.reward.shape[:2]
        if (T, mb) not in hip_upd:
            hip_upd[(T, mb)] = hip_update.HipPPOUpdate(
                ppo_network, T, mb, device, entropy_cost=entropy_cost, discounting=discounting, reward_scaling=reward_scaling,
                gae_lambda=gae_lambda, clipping_epsilon=clipping_epsilon, normalize_advantage=normalize_advantage,
                kl_weight=kl_weight)
        upd = hip_upd[(T, mb)]
        mt = upd.grad(training_state.params, normalizer_params, data_tm, noise, flat_grad, part=part)
        hip_upd["last_metrics"] = mt
        metrics = {k: mt[i] for i, k in enumerate(_METRIC_KEYS)}
        metrics["prediction_corr"] = mt[8]  # computed by the library beside the backward passes
        return metrics

    # C1, in the order the segments of the flat buffer become final: the value network's gradients (its chain joins first,
    # csrc/vnl_ppo.hip part 1), then the policy's.  Each segment is all-reduced on a communication stream of its own as soon
    # as it is final and Adam updates it as soon as its exchange is in: the value exchange (79 % of the bytes) overlaps the
    # policy network's backward, the policy exchange overlaps the value segment's Adam.  (reference train.py:251-268: one
    # pmean of the whole gradient per minibatch step, after the backward pass.)
    segments = (("value", slice(n_pol, n_pol + n_val)), ("policy", slice(0, n_pol)))
    comm_stream = torch.cuda.Stream(device) if (dist is not None and device.type == "cuda") else None

    def exchange(seg: slice):
        """all-reduce(mean) of one segment of flat_grad, enqueued behind what the current stream holds; returns a handle"""
        view = flat_grad[seg]
        if comm_stream is None:
            return dist.all_reduce(view, async_op=True)
        comm_stream.wait_stream(torch.cuda.current_stream(device))
        with torch.cuda.stream(comm_stream):
            work = dist.all_reduce(view, async_op=True)
        return work

    def finish(work, seg: slice, first: bool) -> None:
        """wait for a segment's exchange, average it, apply Adam to it"""
        work.wait()  # (on a HIP device: makes the current stream wait for the collective, no host block)
        if comm_stream is not None:
            torch.cuda.current_stream(device).wait_stream(comm_stream)
        flat_grad[seg].div_(world)
        optimizer.update(flat_grad, training_state.optimizer_state, training_state.params, segment=seg, bump=first)

    def sync_and_step(between=None) -> None:
        """C1 + Adam for one minibatch step.  `between`: work to enqueue after the value segment's exchange has been issued
        (the policy network's backward when the step is split: hip_update part 2)."""
        (_, vseg), (_, pseg) = segments
        wv = exchange(vseg)
        if between is not None:
            between()
        wp = exchange(pseg)
        finish(wv, vseg, True)
        finish(wp, pseg, False)

    def minibatch_step(data: acting.Transition, normalizer_params) -> Metrics:
        """train.py:255-268 + brax gradient_update_fn: grad, all-reduce(mean), adam."""
        if use_hip_update:
            tm = data.map(lambda x: x.transpose(0, 1).contiguous())
            T, mb = tm.reward.shape[:2]
            noise = {"latent": torch.randn((T, mb, ppo_network.policy_module.latents), generator=g_dev, device=device),
                     "entropy": torch.randn((T, mb, ppo_network.parametric_action_distribution.event_size), generator=g_dev,
                                            device=device)}
            if dist is not None:  # split step: the value segment's exchange overlaps the policy network's backward
                metrics = hip_grad(tm, normalizer_params, noise, part=1)
                sync_and_step(between=lambda: hip_grad(tm, normalizer_params, noise, part=2))
                return metrics
            metrics = hip_grad(tm, normalizer_params, noise)
            optimizer.update(flat_grad, training_state.optimizer_state, training_state.params)
            return metrics
        leaf_params.policy.zero_grad(), leaf_params.value.zero_grad()
        loss, metrics = loss_fn(leaf_params, normalizer_params, data, g_dev)
        loss.backward()
        leaf_params.policy.gather_grads(), leaf_params.value.gather_grads()
        if dist is not None:
            sync_and_step()  # C1 (autograd delivers both segments at once: nothing to overlap with, same call structure)
        else:
            optimizer.update(flat_grad, training_state.optimizer_state, training_state.params)
        return metrics

    if capture_graph is None:
        capture_graph = device.type == "cuda"
    graphed: Dict[str, Any] = {}

    def build_graphed(data: acting.Transition, normalizer_params) -> None:
        """Static buffers + one captured minibatch step.  Inputs of a replay: `full` (the whole batch),
        `idx` (this minibatch's rows), `norm`, `noise`; outputs: the parameter update and `acc`."""
        n, T = data.reward.shape[:2]
        mb = n // num_minibatches
        g = graphed
        g["full"] = data.map(lambda x: x.transpose(0, 1).contiguous())  # time-major [T, N, ...]: gathers along dim 1
        g["idx"] = torch.zeros(mb, dtype=torch.int64, device=device)
        g["norm"] = normalizer_params.clone()
        g["noise"] = {"latent": torch.zeros(T, mb, ppo_network.policy_module.latents, device=device),
                      "entropy": torch.zeros(T, mb, ppo_network.parametric_action_distribution.event_size,
                                             device=device)}
        p = training_state.params
        saved = (p.detach().clone(), {k: v.clone() for k, v in training_state.optimizer_state.items()})

        # the minibatch: static buffers filled by ONE vnl_gather_rows launch (brax's index_select per leaf)
        import ctypes as C

        from .. import _lib

        full = g["full"]
        src = acting.Transition(full.observation, full.action, full.reward, full.discount, full.next_observation[-1:],
                                full.extras)  # only the bootstrap row of next_observation is read
        g["mb"] = src.map(lambda x: torch.empty((x.shape[0], mb, *x.shape[2:]), dtype=x.dtype, device=device))
        gd = _lib.GatherDesc()
        gd.idx, gd.N, gd.M = C.c_void_p(g["idx"].data_ptr()), n, mb
        pairs = list(zip(acting._leaves(g["mb"]), acting._leaves(src)))
        assert len(pairs) <= _lib.POST_MAX_OPS
        gd.num_ops = len(pairs)
        for o, (dst, sr) in zip(gd.ops, pairs):
            assert sr.is_contiguous() and sr.dtype in (torch.float32, torch.int32)
            o.dst, o.src, o.T = C.c_void_p(dst.data_ptr()), C.c_void_p(sr.data_ptr()), sr.shape[0]
            o.width = int(sr[0, 0].numel())
        g["gather_desc"], lib = gd, base._L

        split = dist is not None and use_hip_update  # two graphs: [gather, forward, head, value backward] | [policy backward]
        g["split"] = split

        def body2():
            hip_grad(g["mb"], g["norm"], g["noise"], part=2)

        def body():
            _lib.check(lib, lib.vnl_gather_rows(C.byref(gd), C.c_void_p(torch.cuda.current_stream(device).cuda_stream)))
            mbd = g["mb"]
            if use_hip_update:
                metrics = hip_grad(mbd, g["norm"], g["noise"], part=1 if split else 0)
            else:
                leaf_params.policy.zero_grad(), leaf_params.value.zero_grad()
                loss, metrics = loss_fn(leaf_params, g["norm"], mbd, None, noise=g["noise"], time_major=True)
                loss.backward()
                leaf_params.policy.gather_grads(), leaf_params.value.gather_grads()
            if dist is None:
                optimizer.update(flat_grad, training_state.optimizer_state, p)
            return metrics

        side = torch.cuda.Stream(device)
        side.wait_stream(torch.cuda.current_stream(device))
        with torch.cuda.stream(side):
            for _ in range(3):  # warm-up off the capture stream (library handles, autograd buffers)
                m = body()
                if split:
                    body2()
        torch.cuda.current_stream(device).wait_stream(side)
        g["keys"] = sorted(m.keys())
        g["acc"] = torch.zeros(len(g["keys"]), device=device)
        if use_hip_update:  # the library's metric vector, gathered into key order by one index_add of nine floats
            slot = {k: i for i, k in enumerate(_METRIC_KEYS)}
            slot["prediction_corr"] = 8
            g["slots"] = torch.tensor([slot[k] for k in g["keys"]], dtype=torch.int64, device=device)
        g["graph"] = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g["graph"]):
            m = body()
            if use_hip_update:
                g["acc"] += hip_upd["last_metrics"][g["slots"]]
            else:
                g["acc"] += torch.stack([m[k].to(torch.float32).reshape(()) for k in g["keys"]])
        if split:
            g["graph2"] = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g["graph2"]):
                body2()
        with torch.no_grad():  # the warm-up iterations must not count as training
            p.copy_(saved[0])
            for k, v in saved[1].items():
                training_state.optimizer_state[k].copy_(v)

    def sgd_step_graphed(data: acting.Transition, normalizer_params) -> Metrics:
        g = graphed
        n = data.reward.shape[0]
        mb = n // num_minibatches
        perm = torch.randperm(n, generator=g_env).to(device)
        g["acc"].zero_()
        for i in range(num_minibatches):
            g["idx"].copy_(perm[i * mb:(i + 1) * mb])
            for name in ("latent", "entropy"):  # same draws, same order as the eager loss
                torch.randn(g["noise"][name].shape, generator=g_dev, device=device, out=g["noise"][name])
            g["graph"].replay()
            if dist is not None:  # C1 + Adam, eager between the replays (segments: see sync_and_step)
                sync_and_step(between=g["graph2"].replay if g["split"] else None)
        acc = g["acc"] / num_minibatches
        return {k: acc[i] for i, k in enumerate(g["keys"])}

    def sgd_step(data: acting.Transition, normalizer_params) -> Metrics:
        """train.py:270-291: one shared permutation, num_minibatches steps."""
        if capture_graph:
            return sgd_step_graphed(data, normalizer_params)
        n = data.reward.shape[0]
        perm = torch.randperm(n, generator=g_env).to(device)
        acc: Dict[str, torch.Tensor] = {}
        mb = n // num_minibatches
        for i in range(num_minibatches):
            idx = perm[i * mb:(i + 1) * mb]
            m = minibatch_step(data.map(lambda x: x.index_select(0, idx)), normalizer_params)
            for k, v in m.items():
                acc[k] = acc.get(k, 0) + v
        return {k: v / num_minibatches for k, v in acc.items()}

    graphed_roll: Dict[str, Any] = {}

    def unroll_chunks(n_chunks: int):
        """acting.generate_unroll x n_chunks (train.py:304-320).  With `capture_graph` on a HIP device the whole unroll is ONE
        hipGraph replay (acting.GraphedUnroll): the policy reads the flat parameter buffer (updated in place by Adam) and a
        static copy of the normaliser state, refreshed here before every replay."""
        nonlocal env_state
        use_graph = (capture_graph and device.type == "cuda" and g_dev is not g_env and acting._fusable(env) is not None)
        if not use_graph:
            policy = make_policy((training_state.normalizer_params, training_state.params.detach()[:n_pol]))
            out = []
            for _ in range(n_chunks):
                env_state, data = acting.generate_unroll(env, env_state, policy, g_dev, unroll_length,
                                                         extra_fields=("truncation", "traj"))
                out.append(data)
            return out
        g = graphed_roll
        if not g or g["unroll"].state is not env_state:  # (a fresh State after `env.reset`: the graph is bound to its buffers)
            g["norm"] = training_state.normalizer_params.clone()
            g["unroll"] = acting.GraphedUnroll(env, env_state, make_policy((g["norm"], training_state.params.detach()[:n_pol])),
                                               g_dev, unroll_length, extra_fields=("truncation", "traj"))
        for f in ("count", "mean", "summed_variance", "std"):
            getattr(g["norm"], f).copy_(getattr(training_state.normalizer_params, f))
        out = []
        for k in range(n_chunks):
            env_state, data = g["unroll"]()
            out.append(data if n_chunks == 1 else data.map(torch.clone))  # the next replay overwrites the graph's buffers
        return out

    def training_step() -> Metrics:
        """train.py:293-349."""
        nonlocal env_state
        chunks = unroll_chunks(batch_size * num_minibatches // num_envs)
        # [U, T, B, ...] -> swapaxes(1,2) -> reshape(-1, T, ...)   (train.py:323-327)
        data = chunks[0].map(lambda x: x) if len(chunks) == 1 else None
        if data is None:
            keys = chunks[0]
            data = acting.Transition(
                *[torch.stack([getattr(c, f) for c in chunks]) for f in
                  ("observation", "action", "reward", "discount", "next_observation")],
                extras={g: {k: torch.stack([c.extras[g][k] for c in chunks]) for k in keys.extras[g]}
                        for g in keys.extras})
            data = data.map(lambda x: x.transpose(1, 2).reshape(-1, *x.shape[1:2], *x.shape[3:]))
        else:
            data = data.map(lambda x: x.transpose(0, 1))
        assert data.discount.shape[1:] == (unroll_length,)
        normalizer_params = running_statistics.update(training_state.normalizer_params, data.observation,
                                                      distributed=dist is not None)  # C2
        if capture_graph:
            if not graphed:
                build_graphed(data, normalizer_params)
            for dst, src in zip(acting._leaves(graphed["full"]), acting._leaves(data)):
                dst.copy_(src.transpose(0, 1))
            for f in ("count", "mean", "summed_variance", "std"):
                getattr(graphed["norm"], f).copy_(getattr(normalizer_params, f))
        acc: Dict[str, torch.Tensor] = {}
        for _ in range(num_updates_per_batch):
            m = sgd_step(data, normalizer_params)
            for k, v in m.items():
                acc[k] = acc.get(k, 0) + v
        training_state.normalizer_params = normalizer_params
        training_state.env_steps += env_step_per_training_step
        return {k: v / num_updates_per_batch for k, v in acc.items()}

    training_walltime = 0.0

    def training_epoch_with_timing() -> Metrics:
        """train.py:351-394."""
        nonlocal training_walltime
        t = time.time()
        acc: Dict[str, torch.Tensor] = {}
        for _ in range(num_training_steps_per_epoch):
            m = training_step()
            for k, v in m.items():
                acc[k] = acc.get(k, 0) + v
        metrics = {k: float(v / num_training_steps_per_epoch) for k, v in acc.items()}  # blocks (like :376)
        if device.type == "cuda":
            torch.cuda.synchronize(device)
        epoch_training_time = time.time() - t
        training_walltime += epoch_training_time
        sps = (num_training_steps_per_epoch * env_step_per_training_step * max(num_resets_per_eval, 1)) / epoch_training_time
        return {"training/sps": sps, "training/walltime": training_walltime,
                **{f"training/{name}": value for name, value in metrics.items()}}

    if not eval_env:
        eval_env = environment
    evaluator = None
    if rank == 0 and eval_env is not None and num_eval_envs > 0:
        # num_eval_envs is honoured (train.py:433-441 resets the eval env with that many keys): the device env has a fixed
        # batch, so a different count gets an env of its own with the same configuration
        if eval_env.unwrapped.num_envs != num_eval_envs:
            if not hasattr(eval_env.unwrapped, "with_num_envs"):
                raise ValueError(f"eval_env has {eval_env.unwrapped.num_envs} envs, num_eval_envs = {num_eval_envs}")
            eval_env = eval_env.unwrapped.with_num_envs(num_eval_envs)
        ev_wrapped = env_wrappers.wrap(eval_env, episode_length=episode_length, action_repeat=action_repeat,
                                       reset_info_on_autoreset=reset_info_on_autoreset)
        evaluator = acting.Evaluator(ev_wrapped, functools.partial(make_policy, deterministic=deterministic_eval),
                                     num_eval_envs=num_eval_envs, episode_length=episode_length,
                                     action_repeat=action_repeat, key=g_eval)

    def inference_params():
        return (training_state.normalizer_params.clone(), training_state.params.detach()[:n_pol].clone())

    metrics: Metrics = {}
    if rank == 0 and num_evals > 1 and evaluator is not None:
        metrics = evaluator.run_evaluation(inference_params(), training_metrics={})
        logging.info(metrics)
        progress_fn(0, metrics)

    training_metrics: Metrics = {}
    current_step = 0
    for it in range(num_evals_after_init):
        logging.info("starting iteration %s %s", it, time.time() - xt)
        for _ in range(max(num_resets_per_eval, 1)):
            training_metrics = training_epoch_with_timing()
            current_step = training_state.env_steps
            if num_resets_per_eval > 0:
                env_state = env.reset(g_env)
        if rank == 0:
            metrics = training_metrics
            if evaluator is not None:
                metrics = evaluator.run_evaluation(inference_params(), training_metrics)  # (its own State: the
                # training env_state is untouched, as in the reference)
            logging.info(metrics)
            progress_fn(current_step, metrics)
            policy_params_fn(current_step, make_policy, inference_params())
        if dist is not None:
            dist.barrier()

    total_steps = current_step
    assert total_steps >= num_timesteps
    # replicas must still be identical (train.py:485-487)
    if dist is not None:
        ref = training_state.params.detach().clone()
        dist.broadcast(ref, src=0)
        assert torch.equal(ref, training_state.params.detach()), "parameter replicas diverged"
        dist.barrier()
    params = inference_params()
    train.last_training_state = training_state  # value net / optimiser for checkpoint-resume (beyond the reference)
    train.last_ppo_network = ppo_network
    return make_policy, params, metrics
