"""Pins the CPU oracle for the discrete path (CPU only): reference fixtures
(test/baselines.jl:60-88), scipy cross-checks of the third-party pieces, and a brute-force
numpy evaluation of the VB step straight from the reference's formulas."""
import numpy as np
import pytest
from scipy import signal, special, stats


def case(N=3, T=60, B=3, L=5, seed=0, dt=1.0):
    rng = np.random.default_rng(seed)
    data = rng.poisson(0.4, (N, T)).astype(np.int64)
    W = rng.uniform(0.05, 0.3, (N, N))
    th = rng.dirichlet(np.ones(B), (N, N))
    lam0 = rng.uniform(0.2, 1.0, N)
    A = (rng.uniform(size=(N, N)) < 0.6).astype(float)
    return data, lam0, W, th, A, L, B, dt


def test_reference_discrete_baseline_fixture(orc):
    # test/baselines.jl:69-81: λ = ones(2), dt = 0.5 -> intensity 0.5; integrated_intensity(2.0) == [1, 1]
    conv = np.zeros((11, 2, 1))
    lam = orc.disc_intensity(conv, np.ones(2), np.zeros((2, 2)), np.ones((2, 2, 1)), dt=0.5)
    assert np.array_equal(lam, 0.5 * np.ones((11, 2)))


def test_basis_matches_formula(orc):
    for L, B in ((4, 3), (5, 5), (32, 8), (3, 4)):
        phi = orc.disc_basis(L, B, 0.5)
        sigma = L / (B - 1)
        mu = np.linspace(1, L, B + 2)[1:-1] if B < L else np.linspace(1, L, B)
        raw = np.exp(-(np.arange(1, L + 1)[:, None] - mu[None, :]) ** 2 / (4 * sigma))
        want = raw / (raw.sum(axis=0) * 0.5)
        assert np.allclose(phi, want, rtol=1e-13, atol=0)
        assert np.allclose(phi.sum(axis=0) * 0.5, 1.0)


def test_convolve_vs_scipy(orc):
    data, *_rest, L, B, dt = case(T=200, L=7)
    phi = orc.disc_basis(L, B, dt)
    conv = orc.disc_convolve(data, phi)
    T = data.shape[1]
    for b in range(B):
        full = signal.convolve(data.T.astype(float), np.concatenate([[0.0], phi[:, b]])[:, None])[:T]
        assert np.allclose(conv[:, :, b], np.maximum(full, 0.0), atol=1e-12)
    assert np.all(conv[0] == 0.0)                       # lag 0 excluded, nothing before the first bin


def test_intensity_and_loglik_vs_numpy_scipy(orc):
    data, lam0, W, th, A, L, B, dt = case(dt=0.5)
    phi = orc.disc_basis(L, B, dt)
    conv = orc.disc_convolve(data, phi)
    for adj in (None, A):
        lam = orc.disc_intensity(conv, lam0, W, th, dt=dt, A=adj)
        Weff = W if adj is None else adj * W
        want = lam0[None, :] * dt + np.einsum("tpb,pcb->tc", conv, Weff[:, :, None] * th * dt)
        assert np.allclose(lam, want, rtol=1e-13)
        ll = orc.disc_loglik(data, lam)
        assert abs(ll - stats.poisson.logpmf(data.T, lam).sum()) < 1e-9 * abs(ll)


def test_digamma_vs_scipy(orc):
    xs = np.concatenate([np.linspace(1e-3, 12, 500), np.exp(np.linspace(0, 15, 100))])
    got = np.array([orc.digamma(x) for x in xs])
    assert np.max(np.abs(got - special.digamma(xs)) / np.maximum(1.0, np.abs(special.digamma(xs)))) < 1e-14


def test_vb_step_vs_bruteforce(orc):
    data, lam0, W, th, A, L, B, dt = case(N=3, T=40, B=2, L=4, seed=3)
    N, T = data.shape
    phi = orc.disc_basis(L, B, dt)
    conv = orc.disc_convolve(data, phi)
    rng = np.random.default_rng(1)
    av, bv = rng.uniform(0.5, 3, N), rng.uniform(0.5, 3, N)
    kv, nv, gv = rng.uniform(0.5, 3, (N, N)), rng.uniform(0.5, 3, (N, N)), rng.uniform(0.5, 3, (N, N, B))
    a0, b0, kap, nu, gam = 1.5, 2.0, 1.2, 0.7, 1.1
    got = orc.disc_vb_step(data, conv, dt, a0, b0, kap, nu, gam, av, bv, kv, nv, gv)
    # brute force from src/parents.jl:136-177 and the component update!s
    u = np.zeros((T, N, 1 + N * B))
    for c in range(N):
        u[:, c, 0] = np.exp(special.digamma(av[c]) - np.log(bv[c]))
        for p in range(N):
            elt = special.digamma(gv[p, c]) - special.digamma(gv[p, c].sum())
            elw = special.digamma(kv[p, c]) - np.log(nv[p, c])
            u[:, c, 1 + p * B:1 + (p + 1) * B] = conv[:, p, :] * np.exp(elt + elw)
    u /= u.sum(axis=2, keepdims=True)
    want_av = a0 + (u[:, :, 0] * data.T).sum(axis=0)
    want_bv = 1 / b0 + T * dt * np.ones(N)
    want_g = np.zeros((N, N, B))
    for p in range(N):
        for c in range(N):
            want_g[p, c] = (data[c][:, None] * u[:, c, 1 + p * B:1 + (p + 1) * B]).sum(axis=0)
    assert np.allclose(got[0], want_av, rtol=1e-12)
    assert np.allclose(got[1], want_bv, rtol=1e-15)
    assert np.allclose(got[2], kap + want_g.sum(axis=2), rtol=1e-12)
    assert np.allclose(got[3], nu + data.sum(axis=1)[:, None] * np.ones((N, N)), rtol=1e-15)
    assert np.allclose(got[4], gam + want_g, rtol=1e-12)


def test_gibbs_parent_counts_are_a_multinomial_draw(orc):
    # src/parents.jl:82-116 reduced over time: every event gets exactly one parent; the mean of the
    # counts is Σ_t n[c,t]·μ_k(t,c); a bin with weight only on one category puts all its events there
    data, lam0, W, th, A, L, B, dt = case(N=3, T=300, B=3, L=6, seed=5)
    N, T = data.shape
    conv = orc.disc_convolve(data, orc.disc_basis(L, B, dt))
    lam = orc.disc_intensity(conv, lam0, W, th, dt)
    S = 150
    acc = np.zeros((N, 1 + N * B))
    for s in range(S):
        c = orc.disc_resample_parents(data, conv, lam0, W, th, dt, seed=4, step=s)
        assert np.array_equal(c.sum(axis=1), data.sum(axis=1))
        acc += c
    mu = np.zeros((N, 1 + N * B))
    for c in range(N):
        mu[c, 0] = (data[c] * lam0[c] * dt / lam[:, c]).sum()
        for p in range(N):
            for b in range(B):
                mu[c, 1 + p * B + b] = (data[c] * conv[:, p, b] * W[p, c] * th[p, c, b] * dt / lam[:, c]).sum()
    assert np.all(np.abs(acc / S - mu) < 5 * np.sqrt(mu / S) + 0.5)
    none = orc.disc_resample_parents(data, conv, lam0, np.zeros((N, N)), th, dt, seed=1, step=0)
    assert np.array_equal(none[:, 0], data.sum(axis=1)) and not none[:, 1:].any()      # W = 0: baseline takes all


def test_discrete_adjacency_sweep_limits(orc):
    # src/discrete.jl:445-480: with W = 0 the likelihood does not see A, so u <= ρ decides every entry;
    # with a strongly excitatory, well-supported link the entry is kept whatever the prior odds
    data, lam0, W, th, A, L, B, dt = case(N=3, T=200, B=2, L=5, seed=8)
    N, T = data.shape
    conv = orc.disc_convolve(data, orc.disc_basis(L, B, dt))
    u = np.random.default_rng(2).uniform(size=(N, N))
    got = orc.disc_resample_adjacency(data, conv, lam0, np.zeros((N, N)), th, np.ones((N, N)), 0.37, u, dt)
    assert np.array_equal(got, (u <= 0.37).astype(float))
    rng = np.random.default_rng(4)
    data2 = np.zeros((2, 400), dtype=np.int64)
    data2[0, ::10] = 1
    data2[1, 1::10] = 3                                  # node 2 fires right after node 1, every time
    conv2 = orc.disc_convolve(data2, orc.disc_basis(3, 2, 1.0))
    th2 = np.full((2, 2, 2), 0.5)
    got = orc.disc_resample_adjacency(data2, conv2, np.full(2, 0.01), np.full((2, 2), 2.0), th2, np.zeros((2, 2)), 0.5,
                                      np.full((2, 2), 0.999), 1.0)
    assert got[0, 1] == 1.0


def test_discrete_lgcp_baseline_pieces(orc, nhp):
    # intensity(p::DiscreteLogGaussianCoxProcess, times) = interpolation of (x, λ[:, n]·dt): src/baselines.jl:531-537;
    # its likelihood on range(p): :571-584 -- vs numpy / scipy
    from scipy.special import gammaln
    rng = np.random.default_rng(6)
    G, N, T, dt = 9, 3, 64, 0.5
    x = np.linspace(0.0, T * dt, G)
    lam = np.exp(rng.normal(0, 0.5, (G, N)))
    times = np.arange(1, 17, dtype=np.float64)
    got = orc.disc_lgcp_intensity(x, lam, dt, times)
    want = np.column_stack([np.interp(times, x, lam[:, n] * dt) for n in range(N)])
    assert np.allclose(got, want, rtol=1e-14)
    with pytest.raises(Exception):
        orc.disc_lgcp_intensity(x, lam, dt, np.array([T * dt + 1.0]))            # outside the support: DomainError
    s0 = rng.poisson(0.7, (T, N))
    ll = orc.disc_lgcp_loglik(s0, x, lam, dt)
    ts = x[0] + dt * np.arange(T)
    for n in range(N):
        l = np.interp(ts, x, lam[:, n] * dt)
        assert np.isclose(ll[n], np.sum(s0[:, n] * np.log(l) - l - gammaln(s0[:, n] + 1.0)), rtol=1e-12)
    b = nhp.DiscreteLogGaussianCoxProcess(x, lam, nhp.SquaredExponentialKernel(1.0, 4.0), 0.0, dt)
    assert b.ndims() == N and b.nsteps() == T and np.allclose(b.range(), ts)
    assert np.allclose(b.intensity(times), want) and np.isclose(b.intensity(2, 3.0), want[2, 1])
    v = b.params()
    b.params_(2 * v)
    assert np.allclose(b.params(), 2 * v)
    assert np.allclose(b.integrated_intensity(), b.intensity(ts).sum(axis=0))
