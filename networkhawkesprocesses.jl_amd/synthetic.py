"""Seeded synthetic workloads.  numpy PCG64, one seed per array, so the same
inputs are regenerated anywhere (tests, bench.py, the GPU box) without shipping data.

The reference's own generator (`rand(process, duration)`, src/continuous.jl:16-48) is a
recursive branching simulator used for N=2 examples; `branching_sample` below is an
independent iterative implementation of the same generative model for the README-sized case,
and `s_metric` draws i.i.d. uniform event times for the large throughput configurations.
"""
import numpy as np

from .components import (DenseWeightModel, ExponentialImpulseResponse, HomogeneousProcess,
                         LogitNormalImpulseResponse)
from .continuous import ContinuousNetworkHawkesProcess, ContinuousStandardHawkesProcess
from .components import BernoulliNetworkModel


def s_metric_data(N=1024, M=1_000_000, kbar=8.0, dt_max=1.0):
    """times = sort(T·U[0,1)) seed 0, nodes ~ UniformInt{1..N} seed 1, T = M·Δtmax/K̄."""
    T = M * dt_max / kbar
    times = np.sort(np.random.default_rng(0).uniform(0.0, T, M))
    nodes = np.random.default_rng(1).integers(1, N + 1, M).astype(np.int64)
    return times, nodes, float(T)


def s_metric_process(N, M, T, kind="exponential", dt_max=1.0, network=False):
    """λ0 ~ U(.5,1.5)·0.5·M/(N·T); W ~ U(0,1)/N; θ ~ U(1,5)/Δtmax; μ ~ N(0,1); τ ~ U(.5,2);
    A ~ Bernoulli(0.5).  Seeds 2-6."""
    lam0 = np.random.default_rng(2).uniform(0.5, 1.5, N) * 0.5 * M / (N * T)
    W = np.random.default_rng(3).uniform(0.0, 1.0, (N, N)) / N
    baseline = HomogeneousProcess(lam0)
    weights = DenseWeightModel(W)
    if kind == "exponential":
        theta = np.random.default_rng(4).uniform(1.0, 5.0, (N, N)) / (dt_max if np.isfinite(dt_max) else 1.0)
        impulses = ExponentialImpulseResponse(theta, 1.0, 1.0, dt_max)
    else:
        mu = np.random.default_rng(4).normal(0.0, 1.0, (N, N))
        tau = np.random.default_rng(5).uniform(0.5, 2.0, (N, N))
        impulses = LogitNormalImpulseResponse(mu, tau, dt_max)
    if not network:
        return ContinuousStandardHawkesProcess(baseline, impulses, weights)
    A = (np.random.default_rng(6).uniform(size=(N, N)) < 0.5).astype(np.float64)
    return ContinuousNetworkHawkesProcess(baseline, impulses, weights, A, BernoulliNetworkModel(0.5, N))


def branching_sample(lam0, W, theta, duration, seed=0, max_events=2_000_000):
    """Exponential standard Hawkes process by generation-wise branching (the generative model of
    src/continuous.jl:16-48,131-142: Poisson(W[p,c]) children at Exponential(1/θ[p,c]) delays)."""
    rng = np.random.default_rng(seed)
    lam0, W, theta = np.asarray(lam0, float), np.asarray(W, float), np.asarray(theta, float)
    N = len(lam0)
    times, nodes = [], []
    gen_t, gen_n = [], []
    for c in range(N):
        n = rng.poisson(lam0[c] * duration)
        gen_t.append(rng.uniform(0.0, duration, n))
        gen_n.append(np.full(n, c))
    gen_t, gen_n = np.concatenate(gen_t), np.concatenate(gen_n)
    while len(gen_t):
        times.append(gen_t)
        nodes.append(gen_n)
        if sum(len(t) for t in times) > max_events:
            raise RuntimeError("branching process exploded (unstable weights?)")
        nt, nn = [], []
        for c in range(N):
            k = rng.poisson(W[gen_n, c])
            rep_t, rep_p = np.repeat(gen_t, k), np.repeat(gen_n, k)
            child = rep_t + rng.exponential(1.0 / theta[rep_p, c])
            keep = child <= duration
            nt.append(child[keep])
            nn.append(np.full(int(keep.sum()), c))
        gen_t, gen_n = np.concatenate(nt), np.concatenate(nn)
    times, nodes = np.concatenate(times), np.concatenate(nodes)
    order = np.argsort(times, kind="stable")
    return times[order], (nodes[order] + 1).astype(np.int64), float(duration)


def simulated_data(process, duration, seed=0, max_events=20_000_000):
    """Events drawn from `process` itself (exponential impulses, homogeneous baseline) at benchmark sizes: the same
    generative model as branching_sample -- Poisson(W[p,c]) children of every event at Exponential(1/θ[p,c]) delays,
    src/continuous.jl:16-48 -- with the N Poisson draws per event merged into one Poisson(Σ_c W[p,c]) count and a
    categorical child node (Poisson splitting), so a generation costs O(events), not O(events·N).  Children are
    clustered behind their parents: look-back windows are burstier than the uniform S-metric times."""
    rng = np.random.default_rng(seed)
    lam0 = np.asarray(process.baseline.λ, float)
    N = len(lam0)
    W = np.asarray(process.weights.W, float) * getattr(process, "adjacency_matrix", np.ones((N, N)))
    theta = np.asarray(process.impulses.θ, float)
    rows = W.sum(axis=1)
    G = np.cumsum(W.ravel())                                   # row-major running sum: row p is G[p*N : (p+1)*N]
    start = np.concatenate([[0.0], G[N - 1::N][:-1]])           # running sum before row p
    n0 = rng.poisson(lam0 * duration)
    gen_t = rng.uniform(0.0, duration, int(n0.sum()))
    gen_n = np.repeat(np.arange(N), n0)
    times, nodes, total = [], [], 0
    while len(gen_t):
        times.append(gen_t)
        nodes.append(gen_n)
        total += len(gen_t)
        if total > max_events:
            raise RuntimeError("branching process exploded (unstable weights?)")
        k = rng.poisson(rows[gen_n])
        par_t, par_n = np.repeat(gen_t, k), np.repeat(gen_n, k)
        u = rng.uniform(size=len(par_n))
        flat = np.searchsorted(G, start[par_n] + u * rows[par_n], side="right")
        child_n = np.clip(flat - par_n * N, 0, N - 1)
        child_t = par_t + rng.exponential(1.0 / theta[par_n, child_n])
        keep = child_t <= duration
        gen_t, gen_n = child_t[keep], child_n[keep]
    times, nodes = np.concatenate(times), np.concatenate(nodes)
    order = np.argsort(times, kind="stable")
    return times[order], (nodes[order] + 1).astype(np.int64), float(duration)


def readme_case(seed=0):
    """C1: the README model verbatim (README.md:27-34): N=2, λ0=1, W=0.1, θ=1, Δtmax=Inf, T=1000."""
    N, T = 2, 1000.0
    process = ContinuousStandardHawkesProcess(HomogeneousProcess(np.ones(N)),
                                              ExponentialImpulseResponse(np.ones((N, N))),
                                              DenseWeightModel(0.1 * np.ones((N, N))))
    data = branching_sample(np.ones(N), 0.1 * np.ones((N, N)), np.ones((N, N)), T, seed)
    return process, data


# ---- rand(process, duration): host-side simulators of the generative models -------------------------
# The reference's `rand` methods (src/continuous.jl:16-48,131-142,335-348; src/discrete.jl:31-68,
# src/baselines.jl:189-210,521-527) are host data generators outside the hot path; these are independent
# numpy implementations of the same models (generation-wise branching in continuous time, bin-by-bin
# Poisson autoregression in discrete time), drawn from numpy's PCG64 -- not Julia's streams.
def _baseline_events(baseline, duration, rng):
    from .components import LogGaussianCoxProcess
    out = []
    for c in range(baseline.ndims()):
        if isinstance(baseline, LogGaussianCoxProcess):
            if baseline.length() != duration:
                raise ValueError("Sample duration does not match process duration.")       # src/baselines.jl:190
            y = baseline.λ[c]
            top = float(np.max(y))
            cand = rng.uniform(0.0, duration, rng.poisson(top * duration))                   # thinning
            keep = rng.uniform(0.0, top, len(cand)) < np.interp(cand, baseline.x, y)
            out.append(cand[keep])
        else:
            out.append(rng.uniform(0.0, duration, rng.poisson(baseline.λ[c] * duration)))
    return out


def _delays(impulses, p, c, n, rng):
    if isinstance(impulses, ExponentialImpulseResponse):
        return rng.exponential(1.0 / impulses.θ[p, c], n)
    z = impulses.μ[p, c] + rng.standard_normal(n) / np.sqrt(impulses.τ[p, c])                # logit-normal on (0, Δtmax)
    return impulses.Δtmax / (1.0 + np.exp(-z))


def rand_continuous(process, duration, seed=0, max_events=5_000_000):
    """rand(process::ContinuousHawkesProcess, duration) -> (events, nodes, duration)."""
    rng = np.random.default_rng(seed)
    N = process.ndims()
    W = process.weights.W * getattr(process, "adjacency_matrix", np.ones((N, N)))
    base = _baseline_events(process.baseline, duration, rng)
    gen_t = np.concatenate(base)
    gen_n = np.concatenate([np.full(len(b), c) for c, b in enumerate(base)]).astype(np.int64)
    times, nodes, total = [], [], 0
    while len(gen_t):
        times.append(gen_t)
        nodes.append(gen_n)
        total += len(gen_t)
        if total > max_events:
            raise RuntimeError("branching process exploded (unstable weights?)")
        nt, nn = [], []
        for p in range(N):
            tp = gen_t[gen_n == p]
            for c in range(N):
                k = rng.poisson(W[p, c], len(tp))
                child = np.repeat(tp, k) + _delays(process.impulses, p, c, int(k.sum()), rng)
                child = child[child <= duration]
                nt.append(child)
                nn.append(np.full(len(child), c, dtype=np.int64))
        gen_t, gen_n = np.concatenate(nt), np.concatenate(nn)
    times, nodes = np.concatenate(times), np.concatenate(nodes)
    order = np.argsort(times, kind="stable")
    return times[order], (nodes[order] + 1).astype(np.int64), float(duration)


def rand_discrete(process, duration, seed=0):
    """rand(process::DiscreteHawkesProcess, T) -> N x T count matrix: bin t is Poisson with the
    intensity of src/discrete.jl:115-129 given the bins before it."""
    rng = np.random.default_rng(seed)
    N, T, dt = process.ndims(), int(duration), process.dt
    phi = process.impulses.basis()                        # L x B
    L, B = phi.shape
    A = getattr(process, "adjacency_matrix", None)
    eta = process.weights.W[:, :, None] * process.impulses.θ * dt
    if A is not None:
        eta = eta * A[:, :, None]
    b = process.baseline
    base = b.intensity(np.arange(1, T + 1, dtype=np.float64))                                # T x N (λ·dt per bin)
    data = np.zeros((N, T), dtype=np.int64)
    for t in range(T):
        lo = max(0, t - L)
        hist = data[:, lo:t][:, ::-1]                                                        # lag 1 first
        shat = hist @ phi[: t - lo]                                                          # N x B
        lam = base[t] + np.einsum("pb,pcb->c", shat, eta)
        data[:, t] = rng.poisson(lam)
    return data


def rand(process, duration, seed=0):
    """rand(process, duration): simulate data from a continuous or a discrete process."""
    from .discrete import DiscreteHawkesProcess
    if isinstance(process, DiscreteHawkesProcess):
        return rand_discrete(process, duration, seed)
    return rand_continuous(process, duration, seed)
