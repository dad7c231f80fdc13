"""GPU parity: nhp_cont_lgcp_loglik vs the oracle, and the elliptical-slice baseline update built on it."""
import numpy as np
import pytest

from helpers import random_case

pytestmark = pytest.mark.gpu


def _lgcp_case(nhp, N=6, M=5000, T=40.0, G=33, seed=0):
    c = random_case(N, M, T, "exponential", 1.0, lgcp=False, seed=seed, nhp=nhp)
    rng = np.random.default_rng(seed + 100)
    kernel = nhp.SquaredExponentialKernel(1.0, 5.0)
    x = np.linspace(0.0, T, G)
    base = nhp.LogGaussianCoxProcess(x, [np.exp(rng.normal(0, 0.4, G)) for _ in range(N)], kernel, 0.0)
    proc = nhp.ContinuousStandardHawkesProcess(base, c["proc"].impulses, c["proc"].weights)
    return proc, c["data"], rng


def test_candidate_loglik_matches_oracle(nhp, orc):
    proc, data, rng = _lgcp_case(nhp)
    times, nodes, T = data
    N, G = proc.ndims(), len(proc.baseline.x)
    ds = nhp.device_dataset(proc, data)
    pn = (rng.integers(0, N + 1, len(times)) * (rng.uniform(size=len(times)) < 0.5)).astype(np.int64)
    Y = rng.normal(0, 0.6, (N, G))
    got = proc.baseline.candidate_loglikelihood(ds, Y, parentnodes=pn)
    want = orc.lgcp_loglik(times, nodes, pn, N, proc.baseline.x, np.exp(proc.baseline.m + Y))
    assert np.max(np.abs(got - want) / np.abs(want)) < 1e-12            # contract: 1e-6
    Y2 = rng.normal(0, 0.6, (N, G))                                     # the attribution stays on the device
    got2 = proc.baseline.candidate_loglikelihood(ds, Y2)
    want2 = orc.lgcp_loglik(times, nodes, pn, N, proc.baseline.x, np.exp(proc.baseline.m + Y2))
    assert np.max(np.abs(got2 - want2) / np.abs(want2)) < 1e-12


def test_uses_the_sampler_attribution(nhp, orc):
    proc, data, rng = _lgcp_case(nhp, seed=4)
    times, nodes, T = data
    N, G = proc.ndims(), len(proc.baseline.x)
    ds = nhp.device_dataset(proc, data)
    _, pnodes = nhp.resample_parents(proc, ds, seed=9, step=2)
    Y = rng.normal(0, 0.3, (N, G))
    got = proc.baseline.candidate_loglikelihood(ds, Y)                  # no parentnodes: what the sweep left behind
    want = orc.lgcp_loglik(times, nodes, pnodes, N, proc.baseline.x, np.exp(proc.baseline.m + Y))
    assert np.max(np.abs(got - want) / np.abs(want)) < 1e-12


def test_error_conventions(nhp):
    proc, data, rng = _lgcp_case(nhp, M=300, seed=5)
    times, nodes, T = data
    N, G = proc.ndims(), len(proc.baseline.x)
    fresh = nhp.DeviceDataset(nhp.default_context(), data, N, 1.0)
    with pytest.raises(nhp.NhpError):
        proc.baseline.candidate_loglikelihood(fresh, np.zeros((N, G)))  # nothing attributed yet
    proc.baseline.x = proc.baseline.x * 0.5                             # grid ends before the last event
    with pytest.raises(nhp.DomainError):
        proc.baseline.candidate_loglikelihood(fresh, np.zeros((N, G)), parentnodes=np.zeros(len(times), np.int64))


def test_elliptical_slice_tracks_a_rate_change(nhp):
    # Poisson data whose rate drops from 20 to 2 half way, no excitation: the posterior curve must follow
    rng = np.random.default_rng(1)
    N, T, G = 3, 20.0, 21
    ev, nd = [], []
    for c in range(N):
        a = rng.uniform(0, T / 2, rng.poisson(20 * T / 2))
        b = rng.uniform(T / 2, T, rng.poisson(2 * T / 2))
        ev.append(np.concatenate([a, b]))
        nd.append(np.full(len(a) + len(b), c + 1))
    ev, nd = np.concatenate(ev), np.concatenate(nd)
    o = np.argsort(ev, kind="stable")
    data = (ev[o], nd[o].astype(np.int64), T)
    x = np.linspace(0, T, G)
    base = nhp.LogGaussianCoxProcess(x, [np.full(G, 6.0)] * N, nhp.SquaredExponentialKernel(1.5, 2.0), np.log(6.0))
    proc = nhp.ContinuousStandardHawkesProcess(base, nhp.ExponentialImpulseResponse(np.ones((N, N)), 1.0, 1.0, 1.0),
                                               nhp.DenseWeightModel(np.full((N, N), 1e-9)))
    ds = nhp.device_dataset(proc, data)
    pn0 = np.zeros(len(ev), np.int64)                                   # every event is a baseline event
    curves = []
    for s in range(60):
        base.resample_(ds, rng, parentnodes=pn0 if s == 0 else None)
        if s >= 20:
            curves.append(np.vstack(base.λ))
    mean = np.mean(curves, axis=0)
    early, late = mean[:, 2:9].mean(), mean[:, 12:19].mean()
    assert 12.0 < early < 30.0 and 0.8 < late < 4.5, (early, late)


def test_mcmc_with_lgcp_baseline(nhp):
    proc, data, rng = _lgcp_case(nhp, M=2000, seed=7)
    res = nhp.mcmc_(proc, data, nsteps=5, seed=3)
    assert res.steps == 5 and len(res.samples) == 5
    assert len(res.samples[0]) == len(proc.params())
    assert all(np.all(np.isfinite(s)) and np.all(s[: proc.ndims() * len(proc.baseline.x)] > 0) for s in res.samples)
    assert not np.allclose(res.samples[0], res.samples[-1])
