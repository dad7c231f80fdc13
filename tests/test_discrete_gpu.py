"""GPU parity for the discrete path: convolve, intensity (fp64-MFMA GEMM-1), Poisson
log-likelihood and the fused VB step (GEMM-1 + GEMM-2), reference src/discrete.jl:86-151,369-385,
src/parents.jl:136-177."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def make(nhp, N, T, B, L, seed=0, dt=1.0, network=False, rate=0.3):
    rng = np.random.default_rng(seed)
    data = rng.poisson(rate, (N, T)).astype(np.int64)
    W = rng.uniform(0.05, 0.3, (N, N)) / max(1, N // 4)
    th = rng.dirichlet(np.ones(B), (N, N))
    th[:, :, -1] = 1.0 - th[:, :, :-1].sum(axis=2)
    th = np.where(th.sum(axis=2, keepdims=True) == 1.0, th, th)   # constructor demands exact unit sums
    lam0 = rng.uniform(0.2, 1.0, N)
    A = (rng.uniform(size=(N, N)) < 0.6).astype(float)
    base = nhp.DiscreteHomogeneousProcess(lam0, dt)
    imp = nhp.DiscreteGaussianImpulseResponse.__new__(nhp.DiscreteGaussianImpulseResponse)
    imp.θ, imp.γ, imp.γv, imp.nlags, imp.dt, imp.ϕ = th, 1.0, np.ones_like(th), L, dt, None
    wts = nhp.DenseWeightModel(W)
    if network:
        proc = nhp.DiscreteNetworkHawkesProcess(base, imp, wts, A, nhp.BernoulliNetworkModel(0.6, N), dt)
    else:
        proc = nhp.DiscreteStandardHawkesProcess(base, imp, wts, dt)
    return proc, data, lam0, W, th, (A if network else None)


@pytest.mark.parametrize("N,T,B,L", [(3, 50, 2, 4), (5, 300, 3, 7), (16, 1000, 8, 32), (130, 257, 2, 3)])
def test_convolve_bit_exact(nhp, orc, N, T, B, L):
    proc, data, *_ = make(nhp, N, T, B, L, seed=N)
    phi = proc.impulses.basis()
    assert np.array_equal(phi, orc.disc_basis(L, B, 1.0))
    ds, conv = nhp.convolve(proc, data, fetch=True)
    assert conv.shape == (T, N, B)
    assert np.array_equal(conv, orc.disc_convolve(data, phi))      # same summation order, no contraction


@pytest.mark.parametrize("N,T,B,L,rate", [(4, 1001, 3, 64, 0.05), (3, 777, 2, 70, 0.3), (6, 1500, 11, 9, 2.0), (2, 513, 1, 2, 0.0)])
def test_convolve_sparse_and_dense_paths_bit_exact(nhp, orc, N, T, B, L, rate, monkeypatch):
    """The convolution walks only the NONZERO counts of a bin's lag window (bitmap, L <= 64), odd T takes 8-byte stores, L > 64
    takes the dense kernel, B > 8 takes two basis passes, an all-zero matrix gives all zeros: each bit for bit the oracle's
    direct-form sum; and the dense kernel forced on a sparse case agrees with the sparse one."""
    proc, data, *_ = make(nhp, N, T, max(B, 2), L, seed=3 * N + L, rate=rate)
    if B != proc.impulses.θ.shape[2]:
        th = np.ones((N, N, B)) / B
        th[:, :, -1] = 1.0 - th[:, :, :-1].sum(axis=2)
        proc.impulses.θ = th
    phi = proc.impulses.basis()
    want = orc.disc_convolve(data, phi)
    _, conv = nhp.convolve(proc, data, fetch=True)
    assert np.array_equal(conv, want)
    monkeypatch.setenv("NHP_CONV_DENSE", "1")
    _, conv2 = nhp.convolve(proc, data, fetch=True)
    assert np.array_equal(conv2, want)


@pytest.mark.parametrize("N,T,B,L,network", [(3, 50, 2, 4, False), (5, 300, 3, 7, True), (16, 1000, 8, 32, False),
                                             (130, 257, 2, 3, True), (64, 2000, 4, 8, False)])
def test_intensity_and_loglik(nhp, orc, N, T, B, L, network):
    dt = 0.5
    proc, data, lam0, W, th, A = make(nhp, N, T, B, L, seed=N + 1, dt=dt, network=network)
    conv = orc.disc_convolve(data, orc.disc_basis(L, B, dt))
    want = orc.disc_intensity(conv, lam0, W, th, dt=dt, A=A)
    ds = nhp.convolve(proc, data)
    got = nhp.intensity(proc, ds)
    assert got.shape == (T, N)
    assert np.max(np.abs(got - want) / want) < 1e-12
    ll = nhp.loglikelihood(proc, data)                 # loglikelihood(process, data) convolves itself
    ll2 = nhp.loglikelihood(proc, data, convolved=ds)  # loglikelihood(process, data, convolved)
    wll = orc.disc_loglik(data, want)
    assert abs(ll - wll) < 1e-11 * abs(wll) and ll == ll2


@pytest.mark.parametrize("N,T,B,L", [(3, 40, 2, 4), (6, 500, 3, 5), (20, 3000, 4, 8), (130, 300, 2, 3)])
def test_vb_step(nhp, orc, N, T, B, L):
    proc, data, lam0, W, th, _ = make(nhp, N, T, B, L, seed=7 * N)
    rng = np.random.default_rng(5)
    proc.baseline.αv, proc.baseline.βv = rng.uniform(0.5, 3, N), rng.uniform(0.5, 3, N)
    proc.weights.κv, proc.weights.νv = rng.uniform(0.5, 3, (N, N)), rng.uniform(0.5, 3, (N, N))
    proc.impulses.γv = rng.uniform(0.5, 3, (N, N, B))
    conv = orc.disc_convolve(data, orc.disc_basis(L, B, 1.0))
    want = orc.disc_vb_step(data, conv, 1.0, proc.baseline.α0, proc.baseline.β0, proc.weights.κ, proc.weights.ν,
                            proc.impulses.γ, proc.baseline.αv, proc.baseline.βv, proc.weights.κv, proc.weights.νv,
                            proc.impulses.γv)
    ds = nhp.convolve(proc, data)
    vp = nhp.update_(proc, data, ds)
    got = (proc.baseline.αv, proc.baseline.βv, proc.weights.κv, proc.weights.νv, proc.impulses.γv)
    for g, w in zip(got, want):
        assert np.allclose(g, w, rtol=1e-10, atol=1e-12)
    assert len(vp) == 2 * N + N * N * B + 2 * N * N
    # a second step starts from the updated parameters (update order: u from the OLD parameters)
    want2 = orc.disc_vb_step(data, conv, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0, *want)
    nhp.update_(proc, data, ds)
    assert np.allclose(proc.impulses.γv, want2[4], rtol=1e-9)


def test_vb_driver_and_errors(nhp):
    proc, data, *_ = make(nhp, 4, 200, 2, 4, seed=2)
    res = nhp.vb_(proc, data, max_steps=5)
    assert res.step == 5 and len(res.trace) == 5
    proc2, _d, *_ = make(nhp, 4, 200, 2, 4, seed=2)
    res2 = nhp.vb_(proc2, data, max_steps=5, keep_trace=False)      # 5 steps resident on the device
    assert res2.step == 5 and np.allclose(res2.trace[-1], res.trace[-1], rtol=1e-12)
    assert np.all(proc.weights.νv == 1.0 + data.sum(axis=1)[:, None])
    netproc, *_ = make(nhp, 4, 200, 2, 4, seed=2, network=True)
    with pytest.raises(NotImplementedError):
        nhp.update_(netproc, data, None)               # network VB is broken in the reference (D6)
    with pytest.raises(nhp.DomainError):
        nhp.DiscreteHomogeneousProcess([1.0, -1.0])    # test/baselines.jl:62
    with pytest.raises(ValueError):
        nhp.DiscreteHomogeneousProcess(np.ones(2), 0.5).update_(np.zeros((2, 10)), np.zeros((2, 10, 1)))


def test_config4_shape_properties(nhp):
    # BASELINE configs[3] scaled to fit a test (N=512, B=8, L=32, T=4096): GEMM linearity and
    # the closed form for W = 0
    N, T, B, L = 512, 4096, 8, 32
    proc, data, lam0, W, th, _ = make(nhp, N, T, B, L, seed=1, rate=0.05)
    ds = nhp.convolve(proc, data)
    lam = nhp.intensity(proc, ds)
    proc.weights.W = 2.0 * W
    lam2 = nhp.intensity(proc, ds)
    assert np.allclose(lam2 - lam0[None, :], 2.0 * (lam - lam0[None, :]), rtol=1e-12, atol=1e-15)
    proc.weights.W = np.zeros((N, N))
    from scipy.special import gammaln
    closed = (data.T * np.log(lam0)[None, :]).sum() - T * lam0.sum() - gammaln(data + 1.0).sum()
    assert abs(nhp.loglikelihood(proc, data, convolved=ds) - closed) < 1e-10 * abs(closed)


def test_gradient_matches_finite_differences_of_the_oracle(nhp, orc):
    # ∂ll/∂[λ0; vec(W .* θ)] (params order of src/discrete.jl:174-182) vs central differences of the oracle's ll
    rng = np.random.default_rng(12)
    N, T, B, L, dt = 3, 400, 2, 5, 0.5
    data = rng.poisson(0.4, (N, T)).astype(np.int64)
    th = rng.dirichlet(np.ones(B), (N, N))
    th[:, :, -1] = 1.0 - th[:, :, :-1].sum(axis=2)
    proc = nhp.DiscreteStandardHawkesProcess(nhp.DiscreteHomogeneousProcess(rng.uniform(0.3, 0.8, N), dt),
                                             nhp.DiscreteGaussianImpulseResponse(th, L, dt),
                                             nhp.DenseWeightModel(rng.uniform(0.05, 0.3, (N, N))), dt)
    ds, conv = nhp.convolve(proc, data, fetch=True)
    ll, g = nhp.loglikelihood_gradient(proc, data, convolved=ds)
    assert abs(ll - nhp.loglikelihood(proc, data, convolved=ds)) < 1e-9 * abs(ll)
    x = proc.params()
    assert len(g) == len(x) == N + N * N * B

    def f(v):
        eta = v[N:].reshape((N, N, B), order="F")
        W = eta.sum(axis=2)
        return orc.disc_loglik(data, orc.disc_intensity(conv, v[:N], W, eta / W[:, :, None], dt))

    for k in range(len(x)):
        h = 1e-6 * max(1.0, abs(x[k]))
        xp, xm = x.copy(), x.copy()
        xp[k] += h
        xm[k] -= h
        fd = (f(xp) - f(xm)) / (2 * h)
        assert abs(g[k] - fd) < 1e-5 * max(1.0, abs(fd)), (k, g[k], fd)


def test_discrete_mle_finds_the_rates_of_independent_poisson_data(nhp):
    rng = np.random.default_rng(3)
    N, T, B, L = 2, 20000, 2, 4
    true = np.array([0.2, 0.6])
    data = rng.poisson(true[:, None] * np.ones((N, T))).astype(np.int64)
    th = np.full((N, N, B), 0.5)
    proc = nhp.DiscreteStandardHawkesProcess(nhp.DiscreteHomogeneousProcess(np.ones(N), 1.0),
                                             nhp.DiscreteGaussianImpulseResponse(th, L, 1.0),
                                             nhp.DenseWeightModel(np.full((N, N), 0.1)), 1.0)
    ll0 = nhp.loglikelihood(proc, data)
    res = nhp.mle_(proc, data, seed=1)
    assert res.status == "success" and res.maximum > ll0
    assert abs(res.maximum - nhp.loglikelihood(proc, data)) < 1e-6 * abs(res.maximum)    # process holds the estimate
    W = proc.weights.W
    # rate = λ0 / (1 - excitation) roughly; with no true excitation the fitted weights are small
    assert np.all(W < 0.1), W
    assert np.all(np.abs(proc.baseline.λ / (1 - W.sum(axis=0)) - true) / true < 0.1), (proc.baseline.λ, W)
    assert np.allclose(proc.impulses.θ.sum(axis=2), 1.0)
    with pytest.raises(NotImplementedError):
        nhp.mle_(proc, data, regularize=True)


def test_discrete_device_mle_reaches_a_maximum_the_host_optimizer_cannot_improve(nhp, orc):
    # nhp_disc_mle_run (projected L-BFGS with its state in HBM; params!'s split of x into W and θ redone on the device) against
    # scipy's L-BFGS-B on the same analytic gradient and box: the answer is a local maximum -- zero projected gradient, the host
    # optimizer started there gains nothing -- the process holds it, its value is the oracle's, the independent-Poisson rates
    # are recovered, and a problem with real excitation (N = 6, B = 3) ends as high as the host route from the same start.
    rng = np.random.default_rng(3)
    N, T, B, L = 2, 20000, 2, 4
    true = np.array([0.2, 0.6])
    data = rng.poisson(true[:, None] * np.ones((N, T))).astype(np.int64)

    def fresh(N, B, L):
        return nhp.DiscreteStandardHawkesProcess(nhp.DiscreteHomogeneousProcess(np.ones(N), 1.0),
                                                 nhp.DiscreteGaussianImpulseResponse(np.full((N, N, B), 1.0 / B), L, 1.0),
                                                 nhp.DenseWeightModel(np.full((N, N), 0.1)), 1.0)
    proc = fresh(N, B, L)
    res = nhp.mle_(proc, data, seed=1, optimizer="device", f_abstol=1e-9)
    assert res.status == "success"
    x = res.maximizer
    assert np.all(x >= 1e-6) and np.all(x <= 10.0) and np.array_equal(proc.params(), nhp.discrete.disc_params_(fresh(N, B, L), x))
    ll, g = nhp.loglikelihood_gradient(proc, data)
    assert abs(ll - res.maximum) < 1e-9 * abs(ll)
    pg = np.where(((x <= 1e-6) & (g < 0)) | ((x >= 10.0) & (g > 0)), 0.0, g)
    assert np.max(np.abs(pg)) < 5e-2 * abs(ll) ** 0.5
    W = proc.weights.W
    assert np.all(W < 0.1) and np.all(np.abs(proc.baseline.λ / (1 - W.sum(axis=0)) - true) / true < 0.1)
    conv = orc.disc_convolve(data, proc.impulses.basis())
    assert abs(orc.disc_loglik(data, orc.disc_intensity(conv, proc.baseline.λ, W, proc.impulses.θ, 1.0)) - res.maximum) < 1e-9 * abs(ll)
    polish = nhp.mle_(fresh(N, B, L), data, guess=x, f_abstol=1e-9)
    assert polish.maximum - res.maximum < 1e-4 * abs(ll) ** 0.5
    # excitation present
    p6, d6, *_ = make(nhp, 6, 4000, 3, 5, seed=8)
    guess = np.concatenate([rng.uniform(0.05, 0.3, 6), rng.uniform(0.005, 0.05, 6 * 6 * 3)])
    import copy
    a, b = copy.deepcopy(p6), copy.deepcopy(p6)
    dev = nhp.mle_(a, d6, guess=guess, optimizer="device", f_abstol=1e-9, max_steps=3000)
    host = nhp.mle_(b, d6, guess=guess, f_abstol=1e-9, max_steps=3000)
    assert dev.status == "success" and dev.maximum >= host.maximum - 1e-3 * abs(host.maximum) ** 0.5


@pytest.mark.parametrize("N,T,B,L", [(5, 300, 3, 7), (130, 997, 2, 3)])
def test_both_gemm_tile_heights_agree(nhp, orc, N, T, B, L, monkeypatch):
    """GEMM-1 picks 128- or 160-row output tiles by how they fill the last round of workgroups (gemm1_tile_m); both give
    the oracle's intensity / log-likelihood, and the same gradient and VB step (ragged T: partial tiles in both)."""
    proc, data, lam0, W, th, _ = make(nhp, N, T, B, L, seed=N + 1)
    want = orc.disc_intensity(orc.disc_convolve(data, proc.impulses.basis()), lam0, W, th, 1.0)
    got = {}
    for bm in ("128", "160"):
        monkeypatch.setenv("NHP_GEMM_BM", bm)
        ds = nhp.convolve(proc, data)
        lam = nhp.intensity(proc, ds)
        assert np.max(np.abs(lam - want) / want) < 1e-12
        ll, g = nhp.loglikelihood_gradient(proc, data, convolved=ds)
        import copy
        p2 = copy.deepcopy(proc)
        nhp.update_(p2, data, ds)
        got[bm] = (nhp.loglikelihood(proc, data, convolved=ds), ll, g, p2.weights.κv.copy(), p2.impulses.γv.copy(), p2.baseline.αv.copy())
    a, b = got["128"], got["160"]
    assert abs(a[0] - b[0]) < 1e-12 * abs(a[0]) and abs(a[1] - b[1]) < 1e-12 * abs(a[1])
    for x, y in zip(a[2:], b[2:]):
        assert np.allclose(x, y, rtol=1e-12, atol=1e-13)


@pytest.mark.parametrize("N,T,B,L,bm", [(128, 1280, 2, 4, "128"), (128, 1280, 2, 4, "160"), (256, 640, 4, 5, "160")])
def test_whole_tile_gemm_variant_matches_the_oracle_and_the_general_one(nhp, orc, N, T, B, L, bm, monkeypatch):
    """Shapes made of whole tiles (T a multiple of the tile height, N of 128, N·B and the VB reduction chunks of 16) take
    the branch-free, instruction-scheduled main loop (k_gemm_f64<.., WHOLE = true>); NHP_GEMM_PLAIN forces the general
    loop.  Both give the oracle's intensity, log-likelihood, gradient and VB step."""
    proc, data, lam0, W, th, _ = make(nhp, N, T, B, L, seed=N + 3)
    conv = orc.disc_convolve(data, proc.impulses.basis())
    want = orc.disc_intensity(conv, lam0, W, th, 1.0)
    wll = orc.disc_loglik(data, want)
    monkeypatch.setenv("NHP_GEMM_BM", bm)
    got = {}
    for plain in (False, True):
        if plain:
            monkeypatch.setenv("NHP_GEMM_PLAIN", "1")
        else:
            monkeypatch.delenv("NHP_GEMM_PLAIN", raising=False)
        ds = nhp.convolve(proc, data)
        lam = nhp.intensity(proc, ds)
        assert np.max(np.abs(lam - want) / want) < 1e-12
        ll = nhp.loglikelihood(proc, data, convolved=ds)
        assert abs(ll - wll) < 1e-11 * abs(wll)
        ll2, g = nhp.loglikelihood_gradient(proc, data, convolved=ds)
        import copy
        p2 = copy.deepcopy(proc)
        nhp.update_(p2, data, ds)
        got[plain] = (ll, ll2, g, p2.weights.κv.copy(), p2.impulses.γv.copy(), p2.baseline.αv.copy())
    a, b = got[False], got[True]
    assert abs(a[0] - b[0]) < 1e-12 * abs(a[0]) and abs(a[1] - b[1]) < 1e-12 * abs(a[1])
    for x, y in zip(a[2:], b[2:]):
        assert np.allclose(x, y, rtol=1e-11, atol=1e-13)
    p3 = copy.deepcopy(proc)
    want_vb = orc.disc_vb_step(data, conv, 1.0, proc.baseline.α0, proc.baseline.β0, proc.weights.κ, proc.weights.ν,
                               proc.impulses.γ, proc.baseline.αv, proc.baseline.βv, proc.weights.κv, proc.weights.νv,
                               proc.impulses.γv)
    monkeypatch.delenv("NHP_GEMM_PLAIN", raising=False)
    nhp.update_(p3, data, nhp.convolve(p3, data))
    assert np.allclose(p3.impulses.γv, want_vb[4], rtol=1e-10, atol=1e-12)
