"""Independent chains / restarts across the GPUs of one node.

mcmc! has no cross-chain term (src/inference.jl:49-70) and mle! restarts are independent
(src/continuous.jl:185,200), so the path shards by *unit*: chain k runs on rank k mod world with
the dataset replicated once per GPU, and there is no data-path collective.  The only exchange is
one gather of per-chain summaries at the end -- `torch.distributed` with backend "nccl" (RCCL over
xGMI) when the tensors live on the GPU, "gloo" in CPU tests.  The reference has no distributed
code at all (README.md:42 lists it as future work).
"""
import numpy as np


def chains_for_rank(n_chains, rank, world):
    """Chain ids owned by `rank`: k with k mod world == rank (round-robin, stable under world=1)."""
    return [k for k in range(n_chains) if k % world == rank]


def chain_seed(base_seed, chain):
    """Seeds of different chains must give independent Philox streams and host generators."""
    return int(base_seed) * 1_000_003 + int(chain)


def summarize_chain(samples, burn=0):
    """Posterior mean and second moment of a chain's samples: the O(P) payload that is gathered
    (full sample histories stay on their rank)."""
    x = np.asarray(samples[burn:], dtype=np.float64)
    return {"n": np.array([float(len(x))]), "mean": x.mean(axis=0), "m2": (x ** 2).mean(axis=0)}


def gather_summaries(local, n_chains, device=None):
    """All-gather {chain id: summary} dicts so every rank sees every chain.

    `local` maps the chain ids this rank ran to summarize_chain() results.  Without an initialised
    process group (single process) it is returned as is.  Payloads are packed into one tensor per
    rank so the exchange is a single collective."""
    try:
        import torch
        import torch.distributed as dist
    except ImportError:                       # pragma: no cover
        return dict(local)
    if not (dist.is_available() and dist.is_initialized()):
        return dict(local)
    world, rank = dist.get_world_size(), dist.get_rank()
    if n_chains < world:          # checked from the arguments, identically on every rank, BEFORE any collective: a rank
        raise ValueError("every rank must own at least one chain (n_chains >= world size)")   # without chains would
    per_rank = max(len(chains_for_rank(n_chains, r, world)) for r in range(world))             # leave the others blocked
    P = len(next(iter(local.values()))["mean"])
    width = 2 + 2 * P                                       # [chain id, n, mean(P), m2(P)]
    buf = np.full((per_rank, width), -1.0)
    for slot, (k, s) in enumerate(sorted(local.items())):
        buf[slot, 0], buf[slot, 1] = k, s["n"][0]
        buf[slot, 2:2 + P], buf[slot, 2 + P:] = s["mean"], s["m2"]
    if device is None and dist.get_backend() == "nccl":     # stage on THIS rank's GPU (the context's), not on torch's current one
        from . import _lib
        device = torch.device("cuda", _lib.default_context().device)
    dev = device if device is not None else "cpu"
    mine = torch.from_numpy(buf).to(dev)
    parts = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(parts, mine)
    out = {}
    for part in parts:
        for row in part.cpu().numpy():
            if row[0] >= 0:
                out[int(row[0])] = {"n": np.array([row[1]]), "mean": row[2:2 + P].copy(), "m2": row[2 + P:].copy()}
    assert sorted(out) == list(range(n_chains)), (rank, sorted(out))
    return out


def run_chains(make_process, data, n_chains, nsteps, base_seed=0, burn=0, ctx=None, **mcmc_kwargs):
    """Run this rank's share of `n_chains` independent mcmc! chains and gather every chain's summary.

    make_process(chain) must build a fresh process (its parameters are overwritten by the chain).
    Works single-process (world = 1) and under torch.distributed.run (one rank per GPU)."""
    from .inference import mcmc_
    from .continuous import device_dataset
    rank, world = 0, 1
    try:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            rank, world = dist.get_rank(), dist.get_world_size()
    except ImportError:                       # pragma: no cover
        pass
    if n_chains < world:
        raise ValueError("every rank must own at least one chain (n_chains >= world size)")
    from . import _lib
    ctx = ctx or _lib.default_context()
    comm = _lib.comm_for(ctx)
    local = {}
    ds = None
    device_models = {}
    for k in chains_for_rank(n_chains, rank, world):
        process = make_process(k)
        if ds is None:
            ds = device_dataset(process, data, ctx)         # uploaded once per GPU, shared by its chains
        # posterior summaries accumulate on the device when the whole sweep runs there (homogeneous baseline): no
        # per-step transfer of params(process) (33.5 MB at N = 1024); otherwise from the kept samples
        from .components import HomogeneousProcess
        on_device = isinstance(process.baseline, HomogeneousProcess) and mcmc_kwargs.get("device_draws", True)
        if on_device:
            kw = {a: v for a, v in mcmc_kwargs.items() if a not in ("keep_samples", "moments", "burn")}
            res = mcmc_(process, ds, nsteps=nsteps, seed=chain_seed(base_seed, k), ctx=ctx, keep_samples=False,
                        moments=True, burn=burn, **kw)
            local[k] = {"n": np.array([float(res.n)]), "mean": res.mean, "m2": res.m2}
            device_models[k] = (process, process._dev)
        else:
            res = mcmc_(process, ds, nsteps=nsteps, seed=chain_seed(base_seed, k), ctx=ctx, **mcmc_kwargs)
            local[k] = summarize_chain(res.samples, burn)
    if comm is not None and n_chains % world == 0 and len(device_models) == len(local):
        return gather_device_summaries(device_models, n_chains, ctx, comm)
    return gather_summaries(local, n_chains, device=None)


def gather_device_summaries(device_models, n_chains, ctx, comm):
    """BASELINE config 5's exchange through the library: the chains' running sums are all-gathered device to device over
    RCCL / xGMI (nhp_gather_moments) -- one collective per chain slot, every rank contributing its slot-th chain -- and
    divided into means on the host afterwards.  Needs the same number of chains on every rank."""
    import ctypes as C
    from . import _lib
    from .inference import _moments_in_params_order, moments_length
    from .continuous import ContinuousNetworkHawkesProcess
    world, rank = comm.world, comm.rank
    out = {}
    for slot, k in enumerate(sorted(device_models)):
        process, model = device_models[k]
        L = moments_length(process)
        s, q = np.empty((world, L)), np.empty((world, L))
        counts, rho = np.empty(world, dtype=np.int64), np.empty((world, 3))
        _lib.check(_lib.lib().nhp_gather_moments(ctx.h, comm.h, model.h, _lib.dptr(s), _lib.dptr(q), L, _lib.iptr(counts), _lib.dptr(rho)), ctx.h)
        network = isinstance(process, ContinuousNetworkHawkesProcess)
        for r in range(world):
            chain = chains_for_rank(n_chains, r, world)[slot]
            mean, m2 = _moments_in_params_order(process, s[r], q[r], int(counts[r]), network, rho[r, 1], rho[r, 2])
            out[chain] = {"n": np.array([float(counts[r])]), "mean": mean, "m2": m2}
    assert sorted(out) == list(range(n_chains)), (rank, sorted(out))
    return out
