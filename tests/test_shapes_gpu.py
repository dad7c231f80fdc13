"""GPU parity on awkward shapes: N = 1, non-power-of-two N, N above one staging sweep, skewed node
populations, windows from empty to everything, every impulse / baseline / adjacency combination."""
import numpy as np
import pytest

from helpers import rel

pytestmark = pytest.mark.gpu


def build(nhp, orc, N, M, T, kind, dt_max, network, seed, skew=False):
    rng = np.random.default_rng(seed)
    times = np.sort(rng.uniform(0.0, T, M))
    if skew:                                   # 80 % of the events on node 1, some nodes empty
        nodes = np.where(rng.uniform(size=M) < 0.8, 1, rng.integers(1, max(2, N // 2) + 1, M)).astype(np.int64)
    else:
        nodes = rng.integers(1, N + 1, M).astype(np.int64)
    lam0 = rng.uniform(0.5, 1.5, N)
    W = rng.uniform(0.0, 1.0, (N, N)) / max(N, 1)
    A = (rng.uniform(size=(N, N)) < 0.5).astype(np.float64) if network else None
    sc = dt_max if np.isfinite(dt_max) else 1.0
    if kind == "exponential":
        th = rng.uniform(1.0, 5.0, (N, N)) / sc
        om = orc.ContModel(lam0, W, theta=th, dt_max=dt_max, A=A)
        imp = nhp.ExponentialImpulseResponse(th, 1.0, 1.0, dt_max)
    else:
        mu, tau = rng.normal(0, 1, (N, N)), rng.uniform(0.5, 2.0, (N, N))
        om = orc.ContModel(lam0, W, mu=mu, tau=tau, dt_max=dt_max, A=A)
        imp = nhp.LogitNormalImpulseResponse(mu, tau, dt_max)
    base, w = nhp.HomogeneousProcess(lam0), nhp.DenseWeightModel(W)
    proc = (nhp.ContinuousNetworkHawkesProcess(base, imp, w, A, nhp.BernoulliNetworkModel(0.5, N)) if network
            else nhp.ContinuousStandardHawkesProcess(base, imp, w))
    return proc, om, (times, nodes, float(T))


CASES = [  # N, M, T, kind, dt_max, network, skew
    (1, 500, 50.0, "exponential", 2.0, False, False),
    (1, 500, 50.0, "logitnormal", 2.0, False, False),
    (3, 2000, 40.0, "exponential", np.inf, True, False),
    (37, 5000, 300.0, "logitnormal", 1.0, True, True),
    (300, 20000, 800.0, "exponential", 1.0, False, True),
    (1300, 30000, 900.0, "exponential", 0.5, True, False),       # N > 1024: more than one staging sweep per thread
    (257, 9000, 30.0, "logitnormal", 5.0, False, False),         # ~1500 parents per window
    (64, 6000, 1e7, "exponential", 1e-3, False, False),          # every window empty
    (4500, 40000, 2000.0, "exponential", 0.2, False, False),     # columns above the default 64 KiB LDS carve-out
    (3000, 30000, 1500.0, "exponential", 0.3, False, False),     # recursive kernel's 512-thread variant (2048 < N <= 4096)
]


@pytest.mark.parametrize("N,M,T,kind,dt_max,network,skew", CASES)
def test_loglik_sampler_and_gradient_on_awkward_shapes(nhp, orc, N, M, T, kind, dt_max, network, skew):
    proc, om, data = build(nhp, orc, N, M, T, kind, dt_max, network, seed=N + M, skew=skew)
    t, n, dur = data
    got = nhp.loglikelihood(proc, data, recursive=False)
    want = orc.loglik_windowed(om, t, n, dur, flags=orc.FAST_INTEGRAL)
    assert rel(got, want) < 1e-11
    if kind == "exponential" and N <= 4096:                       # recursive kernel: register-resident state, N <= 4096
        got = nhp.loglikelihood(proc, data, recursive=True)
        want = orc.loglik_recursive(om, t, n, dur, flags=orc.FAST_INTEGRAL)
        assert rel(got, want) < 1e-11
    u = np.random.default_rng(1).uniform(size=M)
    p, pn, st = nhp.resample_parents(proc, data, u=u, with_stats=True)
    wp, wpn = orc.resample_parents(om, t, n, u, flags=orc.MATH_DET)
    assert np.array_equal(p, wp) and np.array_equal(pn, wpn)
    assert np.array_equal(st["Mnm"], orc.parent_counts(n, pn, N))
    assert np.array_equal(st["cnt0"], orc.baseline_node_counts(n, pn, N))
    if kind == "exponential":
        assert np.array_equal(st["Xnm"], orc.duration_mean(t, n, p, N))
    lam = nhp.total_intensity(proc, data)
    assert np.max(np.abs(lam - orc.total_intensity(om, t, n)) / lam) < 1e-12
    if (N <= 64 or N == 3000) and not network:                    # N = 3000: gradient columns above 64 KiB of LDS
        # (the recursive gradient keeps 8 N-vectors in LDS: N <= 2550)
        for rec in ((False, True) if kind == "exponential" and N <= 2550 else (False,)):
            ll, g = nhp.loglikelihood_gradient(proc, data, recursive=rec)
            wll, wg = orc.loglik_grad(om, t, n, dur, recursive=rec)
            assert np.max(np.abs(g - wg) / np.maximum(1.0, np.abs(wg))) < 1e-9
    if network and N <= 40:
        uA = np.random.default_rng(2).uniform(size=(N, N))
        wantA = orc.resample_adjacency(om, t, n, dur, 0.5, uA)
        nhp.resample_adjacency_matrix_(proc, data, u=uA)
        assert np.array_equal(proc.adjacency_matrix, wantA)


def test_device_dataset_cache_sees_in_place_refills(nhp, orc):
    """device_dataset caches the upload per (arrays, Δtmax); refilling the SAME buffers with another dataset of the same
    length must not evaluate the stale device copy (the cache key carries a content fingerprint), and two datasets of equal
    length keep separate truncation windows for the recursive formulation (keyed on the dataset, not on its size)."""
    rng = np.random.default_rng(0)
    N, M, T = 4, 3000, 100.0
    times = np.sort(rng.uniform(0, T, M))
    nodes = rng.integers(1, N + 1, M).astype(np.int64)
    proc = nhp.ContinuousStandardHawkesProcess(nhp.HomogeneousProcess(rng.uniform(0.5, 1.5, N)),
                                               nhp.ExponentialImpulseResponse(rng.uniform(20, 40, (N, N)), 1.0, 1.0, 0.5),
                                               nhp.DenseWeightModel(rng.uniform(0, 0.2, (N, N))))
    om = orc.ContModel(proc.baseline.λ, proc.weights.W, theta=proc.impulses.θ, dt_max=0.5)
    data = (times, nodes, T)
    for rec in (False, True):
        assert abs(nhp.loglikelihood(proc, data, recursive=rec) - orc.loglik(om, times, nodes, T, recursive=rec)) < 1e-9
    times[:] = np.sort(rng.uniform(0, T, M) ** 1.7 / T ** 0.7)     # refill in place: another (burstier) dataset
    nodes[:] = rng.integers(1, N + 1, M)
    for rec in (False, True):
        assert abs(nhp.loglikelihood(proc, data, recursive=rec) - orc.loglik(om, times, nodes, T, recursive=rec)) < 1e-9
