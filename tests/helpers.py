"""Shared builders for the parity tests: the same seeded model is produced as (a) the
reference-mirroring process object that drives the HIP path and (b) the oracle's ContModel."""
import numpy as np


def random_case(N, M, T, kind="exponential", dt_max=1.0, network=False, lgcp=False, seed=0, nhp=None, orc=None):
    rng = np.random.default_rng(seed)
    times = np.sort(rng.uniform(0.0, T, M))
    nodes = rng.integers(1, N + 1, M).astype(np.int64)
    W = rng.uniform(0.0, 1.0, (N, N)) / max(N, 2) * 2.0
    A = (rng.uniform(size=(N, N)) < 0.5).astype(np.float64) if network else None
    if lgcp:
        G = 17
        gx = np.linspace(0.0, T, G)
        lam0 = np.exp(rng.normal(0.0, 0.5, (N, G)))
    else:
        gx = None
        lam0 = rng.uniform(0.5, 1.5, N)
    scale = dt_max if np.isfinite(dt_max) else 1.0
    theta = rng.uniform(1.0, 5.0, (N, N)) / scale
    mu = rng.normal(0.0, 1.0, (N, N))
    tau = rng.uniform(0.5, 2.0, (N, N))
    out = {"times": times, "nodes": nodes, "T": float(T), "data": (times, nodes, float(T))}
    if orc is not None:
        if kind == "exponential":
            out["om"] = orc.ContModel(lam0, W, theta=theta, dt_max=dt_max, A=A, grid_x=gx)
        else:
            out["om"] = orc.ContModel(lam0, W, mu=mu, tau=tau, dt_max=dt_max, A=A, grid_x=gx)
    if nhp is not None:
        baseline = nhp.LogGaussianCoxProcess(gx, list(lam0)) if lgcp else nhp.HomogeneousProcess(lam0)
        if kind == "exponential":
            impulses = nhp.ExponentialImpulseResponse(theta, 1.0, 1.0, dt_max)
        else:
            impulses = nhp.LogitNormalImpulseResponse(mu, tau, dt_max)
        weights = nhp.DenseWeightModel(W)
        if network:
            out["proc"] = nhp.ContinuousNetworkHawkesProcess(baseline, impulses, weights, A,
                                                             nhp.BernoulliNetworkModel(0.5, N))
        else:
            out["proc"] = nhp.ContinuousStandardHawkesProcess(baseline, impulses, weights)
    return out


def rel(a, b):
    return abs(a - b) / max(abs(b), 1e-300)
