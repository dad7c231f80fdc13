"""GPU parity for the DiscreteLogGaussianCoxProcess baseline (src/baselines.jl:461-609): per-bin baseline in
the GEMM epilogues, in the Gibbs parent counts (bit-exact) and in the elliptical-slice likelihood."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def make(nhp, N=4, T=600, B=3, L=7, G=13, seed=0, dt=1.0):
    rng = np.random.default_rng(seed)
    data = rng.poisson(0.5, (N, T)).astype(np.int64)
    th = rng.dirichlet(np.ones(B), (N, N))
    th[:, :, -1] = 1.0 - th[:, :, :-1].sum(axis=2)
    while not np.all(th.sum(axis=2) == 1.0):
        th = rng.dirichlet(np.ones(B), (N, N))
        th[:, :, -1] = 1.0 - th[:, :, :-1].sum(axis=2)
    x = np.linspace(0.0, T * dt, G)
    lam = np.exp(rng.normal(-1.0, 0.5, (G, N)))
    base = nhp.DiscreteLogGaussianCoxProcess(x, lam, nhp.SquaredExponentialKernel(1.0, T / 6.0), -1.0, dt)
    proc = nhp.DiscreteStandardHawkesProcess(base, nhp.DiscreteGaussianImpulseResponse(th, L, dt),
                                             nhp.DenseWeightModel(rng.uniform(0.0, 0.6, (N, N)) / N), dt)
    return proc, data, rng


def test_intensity_loglik_and_counts_with_a_per_bin_baseline(nhp, orc):
    proc, data, rng = make(nhp)
    N, T = data.shape
    b = proc.baseline
    ds, conv = nhp.convolve(proc, data, fetch=True)
    base_tn = orc.disc_lgcp_intensity(b.x, b.λ, b.dt, np.arange(1, T + 1, dtype=np.float64))      # intensity(baseline, 1:T)
    assert np.allclose(base_tn, b.intensity(np.arange(1, T + 1)), rtol=1e-14)
    lam = nhp.intensity(proc, ds)
    want = orc.disc_intensity_b(conv, base_tn, proc.weights.W, proc.impulses.θ, proc.dt)
    assert np.max(np.abs(lam - want) / want) < 1e-12
    ll = nhp.loglikelihood(proc, data, convolved=ds)
    assert abs(ll - orc.disc_loglik(data, want)) < 1e-11 * abs(ll)
    got = nhp.resample_parent_counts(proc, convolved=ds, seed=3, step=1)
    wc, wb = orc.disc_resample_parents_b(data, conv, base_tn, proc.weights.W, proc.impulses.θ, proc.dt, seed=3, step=1)
    assert np.array_equal(got, wc)
    # the per-bin baseline counts stay on the device: score candidate curves against them
    Y = rng.normal(0, 0.4, b.λ.shape)
    gl = b.candidate_loglikelihood(ds, Y)
    wl = orc.disc_lgcp_loglik(wb, b.x, np.exp(b.m + Y), b.dt)
    assert np.max(np.abs(gl - wl) / np.abs(wl)) < 1e-12


def test_gradient_with_lgcp_baseline_matches_finite_differences(nhp, orc):
    proc, data, rng = make(nhp, N=3, T=300, B=2, L=5, G=7, seed=4)
    N, T = data.shape
    b = proc.baseline
    ds, conv = nhp.convolve(proc, data, fetch=True)
    ll, g = nhp.loglikelihood_gradient(proc, data, convolved=ds)
    x = proc.params()
    G = len(b.x)
    assert len(g) == len(x) == G * N + N * N * 2
    times = np.arange(1, T + 1, dtype=np.float64)

    def f(v):
        lam = v[:G * N].reshape((G, N), order="F")
        eta = v[G * N:].reshape((N, N, 2), order="F")
        W = eta.sum(axis=2)
        base_tn = orc.disc_lgcp_intensity(b.x, lam, b.dt, times)
        return orc.disc_loglik(data, orc.disc_intensity_b(conv, base_tn, W, eta / W[:, :, None], proc.dt))

    assert abs(ll - f(x)) < 1e-10 * abs(ll)
    for k in list(range(0, G * N, 3)) + list(range(G * N, len(x), 2)):
        h = 1e-6 * max(1.0, abs(x[k]))
        xp, xm = x.copy(), x.copy()
        xp[k] += h
        xm[k] -= h
        fd = (f(xp) - f(xm)) / (2 * h)
        assert abs(g[k] - fd) < 1e-5 * max(1.0, abs(fd)), (k, g[k], fd)


def test_mcmc_and_mle_run_with_the_discrete_lgcp_baseline(nhp):
    # examples/discrete-gaussian-standard-hawkes-gp.jl:40,48
    proc, data, rng = make(nhp, N=3, T=800, B=2, L=5, G=9, seed=8)
    ll0 = nhp.loglikelihood(proc, data)
    res = nhp.mcmc_(proc, data, nsteps=6, seed=4)
    assert res.steps == 6 and all(np.all(np.isfinite(s)) and np.all(s > 0) for s in res.samples)
    assert len(res.samples[0]) == len(proc.params())
    fit = nhp.mle_(proc, data, guess=np.clip(proc.params(), 1e-3, 5.0), max_steps=40)
    assert np.isfinite(fit.maximum) and fit.maximum >= nhp.loglikelihood(proc, data) - 1e-6 * abs(fit.maximum)
    assert fit.maximum > ll0 - abs(ll0)            # finite, sane
    with pytest.raises(NotImplementedError):
        nhp.update_(proc, data, nhp.convolve(proc, data))


def test_adjacency_sweep_with_a_per_bin_baseline(nhp, monkeypatch):
    """resample_adjacency_matrix! on a network process whose baseline is the per-bin LGCP curve: the sweep's starting λ of the
    occupied bins comes from the entry lists (k_dadj_lambda0: base read per bin) or from the intensity GEMM's epilogue
    (NHP_DADJ_LAMBDA0=0) -- the same decisions for the same uniforms; B = 4 so that the list route is taken."""
    proc, data, rng = make(nhp, N=6, T=900, B=4, L=6, G=11, seed=9)
    N = data.shape[0]
    A0 = (rng.uniform(size=(N, N)) < 0.5).astype(np.float64)
    u = rng.uniform(size=(N, N))
    got = {}
    for name, env in (("lists", "1"), ("gemm", "0")):
        monkeypatch.setenv("NHP_DADJ_LAMBDA0", env)
        net = nhp.DiscreteNetworkHawkesProcess(proc.baseline, proc.impulses, nhp.DenseWeightModel(proc.weights.W * N * 1.5), A0.copy(),
                                               nhp.BernoulliNetworkModel(0.3, N), proc.dt)
        ds = nhp.convolve(net, data)
        nhp.disc_resample_adjacency_matrix_(net, convolved=ds, u=u)
        got[name] = net.adjacency_matrix.copy()
    monkeypatch.delenv("NHP_DADJ_LAMBDA0", raising=False)
    assert np.array_equal(got["lists"], got["gemm"]) and not np.array_equal(got["lists"], A0)
