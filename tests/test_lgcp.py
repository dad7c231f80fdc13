"""LGCP baseline update: host pieces and the oracle, CPU only.

Fixtures: the reference's own split_extract cases (test/baselines.jl:29-40) and constructor checks
(:43-56); the oracle's ll is pinned by closed forms (constant and linear intensities)."""
import numpy as np
import pytest


def test_split_extract_reference_fixtures(nhp):
    # test/baselines.jl:29-40
    out = nhp.split_extract(([], [], 1.0), ([], []), 1)
    assert len(out) == 1 and len(out[0][0]) == 0 and out[0][2] == 1.0
    assert len(nhp.split_extract(([], [], 1.0), ([], []), 2)) == 2
    data = ([0.1, 0.2, 0.3, 0.4], [1, 1, 2, 2], 1.0)
    a, b = nhp.split_extract(data, ([], [0, 1, 0, 2]), 2)
    assert a[0].tolist() == [0.1] and a[1].tolist() == [1] and a[2] == 1.0
    assert b[0].tolist() == [0.3] and b[1].tolist() == [2]
    a, b = nhp.split_extract(data, ([], [1, 1, 2, 2]), 2)
    assert len(a[0]) == 0 and len(b[0]) == 0


def test_constructor_and_kernels(nhp):
    # test/baselines.jl:43-56
    rng = np.random.default_rng(0)
    kernel = nhp.SquaredExponentialKernel(1.0, 1.0)
    gp = nhp.GaussianProcess(kernel)
    x = np.arange(0.0, 1.05, 0.1)
    p = nhp.LogGaussianCoxProcess(x, [np.exp(gp.rand(x, rng))], kernel, 0.0)
    assert p.ndims() == 1 and p.length() == 1.0
    p = nhp.LogGaussianCoxProcess(x, [np.exp(gp.rand(x, rng)) for _ in range(2)], kernel, 0.0)
    assert p.ndims() == 2 and p.length() == 1.0
    with pytest.raises(nhp.DomainError):
        nhp.LogGaussianCoxProcess(x + 0.5, [np.ones(len(x))], kernel, 0.0)      # src/baselines.jl:156
    for k in (kernel, nhp.OrnsteinUhlenbeckKernel(2.0, 0.5), nhp.PeriodicKernel(1.0, 1.0, 0.5)):
        S = k(x)
        assert np.allclose(S, S.T) and np.min(np.linalg.eigvalsh(S)) > -1e-12   # posdef! (3 diagonal shifts at most)
        assert np.isclose(k(0.2, 0.7), k(0.7, 0.2))
    q = nhp.LogGaussianCoxProcess.from_gp(gp, 1.0, 10.0, 20, 3, rng)
    assert q.ndims() == 3 and len(q.x) == 21 and q.length() == 10.0
    v = q.params()
    q.params_(2 * v)
    assert np.allclose(q.params(), 2 * v)


def test_oracle_lgcp_loglik_closed_forms(orc):
    rng = np.random.default_rng(3)
    N, M, T, G = 3, 400, 8.0, 9
    times = np.sort(rng.uniform(0, T, M))
    nodes = rng.integers(1, N + 1, M)
    pn = rng.integers(0, N + 1, M) * (rng.uniform(size=M) < 0.6)
    gx = np.linspace(0, T, G)
    lam = np.tile(np.array([2.0, 0.5, 3.0])[:, None], (1, G))                   # constant: -λT + n0 log λ
    got = orc.lgcp_loglik(times, nodes, pn, N, gx, lam)
    for c in range(N):
        n0 = np.sum((nodes == c + 1) & (pn == 0))
        assert np.isclose(got[c], -lam[c, 0] * T + n0 * np.log(lam[c, 0]), rtol=1e-13)
    lam = np.exp(rng.normal(0, 0.7, (N, G)))                                    # piecewise linear vs numpy
    got = orc.lgcp_loglik(times, nodes, pn, N, gx, lam)
    for c in range(N):
        sel = (nodes == c + 1) & (pn == 0)
        want = -np.sum(0.5 * (lam[c, 1:] + lam[c, :-1]) * np.diff(gx)) + np.sum(np.log(np.interp(times[sel], gx, lam[c])))
        assert np.isclose(got[c], want, rtol=1e-12)
    with pytest.raises(Exception):
        orc.lgcp_loglik(times + 1.0, nodes, pn, N, gx, lam)                     # an event beyond x[end]: DomainError
