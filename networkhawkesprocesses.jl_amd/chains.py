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
    per_rank = max(len(chains_for_rank(n_chains, r, world)) for r in range(world))
    if not local:
        raise ValueError("every rank must own at least one chain (n_chains >= world size)")
    P = len(next(iter(local.values()))["mean"])
    width = 2 + 2 * P                                       # [chain id, n, mean(P), m2(P)]
    buf = np.full((per_rank, width), -1.0)
    for slot, (k, s) in enumerate(sorted(local.items())):
        buf[slot, 0], buf[slot, 1] = k, s["n"][0]
        buf[slot, 2:2 + P], buf[slot, 2 + P:] = s["mean"], s["m2"]
    dev = device if device is not None else ("cuda" if dist.get_backend() == "nccl" else "cpu")
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
    local = {}
    ds = None
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
        else:
            res = mcmc_(process, ds, nsteps=nsteps, seed=chain_seed(base_seed, k), ctx=ctx, **mcmc_kwargs)
            local[k] = summarize_chain(res.samples, burn)
    return gather_summaries(local, n_chains)
