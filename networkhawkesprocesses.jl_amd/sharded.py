"""ONE log-likelihood / gradient evaluation over the GPUs of a node (second way).

The continuous log-likelihood (src/continuous.jl:216-237, 360-389; recursive twin :241-276, 407-442) is a sum over
child nodes c of

    -∫λ0_c  -  Σ_p cnt[p]·[A·]W[p,c]  +  Σ_{i: c_i = c} log λ_i ,

and its gradient is block-separable in the same columns (λ0_c, θ[:,c] | μ[:,c], τ[:,c], W[:,c]).  So a single
evaluation shards by *column range*: every rank keeps all events (each is a parent of children on any node; 16 bytes
per event, nothing next to 288 GB) but builds work only for the children on its own nodes
(`nhp_cont_dataset_create_columns`), and the exchange is one all-reduce of a scalar (log-likelihood) or of the P-vector
(gradient; other ranks' columns are exact zeros) -- `torch.distributed`, backend "nccl" (RCCL over xGMI) on GPUs,
"gloo" in the CPU tests.  The time axis is not cut: the recursive formulation carries its state over all history
(SURVEY 8e), columns work for both formulations, and the per-column work is what the kernels already partition by.

The Gibbs sweep of mcmc! separates by the same columns (the parents of the children on c, column c's statistics and
conjugate draws, the sweep of A[:, c]), so `mcmc_` on a ShardedDataset runs ONE chain on all ranks, each its columns,
with one scalar exchanged per step -- and, the random streams being keyed by global indices, it is the single-GPU chain
value for value (inference.mcmc_).

The reference has no distributed code (README.md:42); independent chains / restarts shard without any exchange
(chains.py) and remain the primary way to use several GPUs.
"""
import numpy as np

from . import _lib


def column_costs(events, nodes, nnodes, Δtmax):
    """Per child node: parent-child pairs inside the look-back window plus a per-child constant -- what the
    windowed kernels' time is proportional to."""
    events = np.asarray(events, dtype=np.float64)
    nodes = np.asarray(nodes, dtype=np.int64)
    if len(events) == 0:
        return np.ones(nnodes)
    # window of event i: parents j < i with t_j > t_i - Δtmax (strict, src/continuous.jl:291)
    first = np.searchsorted(events, events - Δtmax, side="right") if np.isfinite(Δtmax) else np.zeros(len(events), dtype=np.int64)
    k = np.maximum(np.arange(len(events)) - first, 0)
    return np.bincount(nodes - 1, weights=k + 8.0, minlength=nnodes) + 1.0


def column_ranges(costs, n_shards, align=4):
    """Cut [0, N) into `n_shards` contiguous, non-empty ranges of roughly equal cost.  Cuts fall on multiples of
    `align` where that leaves every shard non-empty (keeps the XCD-aware item groups of cont_data.hip whole)."""
    costs = np.asarray(costs, dtype=np.float64)
    N = len(costs)
    if not 1 <= n_shards <= N:
        raise ValueError(f"need 1 <= n_shards <= nnodes, got {n_shards} shards for {N} nodes")
    cum = np.concatenate([[0.0], np.cumsum(costs)])
    cuts = [0]
    for s in range(1, n_shards):
        c = int(np.searchsorted(cum, cum[-1] * s / n_shards, side="left"))
        if align > 1 and N >= 2 * align * n_shards:
            c = int(round(c / align)) * align
        c = max(c, cuts[-1] + 1)                 # non-empty ...
        c = min(c, N - (n_shards - s))           # ... and leaves room for the shards after it
        cuts.append(c)
    cuts.append(N)
    return [(cuts[s], cuts[s + 1]) for s in range(n_shards)]


def _world():
    try:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            return dist.get_rank(), dist.get_world_size()
    except ImportError:                       # pragma: no cover
        pass
    return 0, 1


def _all_reduce_sum(x, ctx=None):
    """Sum a float64 numpy vector over the ranks of the default process group (identity without one).  On GPUs ("nccl"
    group) the exchange is the library's own RCCL communicator of `ctx` (nhp_allreduce_sum: staged on ctx's device,
    reduced over xGMI); the "gloo" rehearsal reduces on the CPU."""
    rank, world = _world()
    if world == 1:
        return x
    comm = _lib.comm_for(ctx)
    if comm is not None:
        return comm.allreduce_sum(x)
    import torch
    import torch.distributed as dist
    if dist.get_backend() == "nccl":          # (unreachable: comm_for serves every nccl group) stage on OUR device
        t = torch.from_numpy(np.ascontiguousarray(x, dtype=np.float64)).to(torch.device("cuda", (ctx or _lib.default_context()).device))
    else:
        t = torch.from_numpy(np.ascontiguousarray(x, dtype=np.float64).copy())
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t.cpu().numpy()


class ShardedDataset:
    """This rank's column shard of (events, nodes, duration): built once, reused by every evaluation."""

    def __init__(self, process, data, ctx=None, rank=None, world=None, ranges=None):
        from .continuous import DeviceDataset
        r, w = _world()
        self.rank = r if rank is None else int(rank)
        self.world = w if world is None else int(world)
        events, nodes, duration = data
        N, Δtmax = process.ndims(), process.impulses.Δtmax
        self.ranges = ranges if ranges is not None else column_ranges(column_costs(events, nodes, N, Δtmax), self.world)
        if len(self.ranges) != self.world:
            raise ValueError("one column range per rank")
        self.ctx = ctx or _lib.default_context()
        self.local = DeviceDataset(self.ctx, data, N, Δtmax, columns=self.ranges[self.rank])


def sharded_loglikelihood(process, data, recursive=True, ctx=None, model=None):
    """loglikelihood(process, data; recursive) evaluated by all ranks together: each rank its columns, one scalar
    all-reduce.  `data` is the (events, nodes, duration) tuple or a ShardedDataset; every rank returns the same value.
    Without a process group this is `loglikelihood`."""
    import ctypes as C
    from .continuous import loglikelihood, _check_recursive
    sd = data if isinstance(data, ShardedDataset) else ShardedDataset(process, data, ctx)
    comm = _lib.comm_for(sd.ctx)
    if comm is not None:                      # partial result all-reduced where the kernel left it (device, RCCL), fetched once
        model = model or process.device_model(sd.ctx)
        ll = C.c_double()
        _lib.check(_lib.lib().nhp_cont_loglik_allreduce(sd.ctx.h, comm.h, sd.local.h, model.h, _check_recursive(process, recursive),
                                                        C.byref(ll)), sd.ctx.h)
        return ll.value
    part = loglikelihood(process, sd.local, recursive=recursive, ctx=sd.ctx, model=model)
    return float(_all_reduce_sum(np.array([part]), sd.ctx)[0])


def sharded_loglikelihood_gradient(process, data, recursive=True, ctx=None, model=None):
    """(ll, ∇ll) in params! order, each rank its columns (the rest of its gradient is exactly 0), one all-reduce of
    [ll; ∇ll]."""
    import ctypes as C
    from .continuous import loglikelihood_gradient, _check_recursive, gradient_length
    sd = data if isinstance(data, ShardedDataset) else ShardedDataset(process, data, ctx)
    comm = _lib.comm_for(sd.ctx)
    if comm is not None:                      # [ll; ∇ll] all-reduced on the device: one download, already summed
        model = model or process.device_model(sd.ctx)
        P = gradient_length(process)
        g, ll = np.empty(P), C.c_double()
        _lib.check(_lib.lib().nhp_cont_loglik_grad_allreduce(sd.ctx.h, comm.h, sd.local.h, model.h, _check_recursive(process, recursive),
                                                             C.byref(ll), _lib.dptr(g), P), sd.ctx.h)
        return ll.value, g
    ll, g = loglikelihood_gradient(process, sd.local, recursive=recursive, ctx=sd.ctx, model=model)
    tot = _all_reduce_sum(np.concatenate([[ll], g]), sd.ctx)
    return float(tot[0]), tot[1:]
