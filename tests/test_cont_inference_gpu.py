"""GPU tests for intensity(process, data, times), the analytic gradient, mle! and mcmc!
(reference: src/continuous.jl:76-96,144-208,350-358; src/inference.jl:49-70)."""
import numpy as np
import pytest

from helpers import random_case, rel

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("kind,network,lgcp", [("exponential", False, False), ("logitnormal", True, False),
                                               ("exponential", True, True)])
def test_intensity_matches_oracle(nhp, orc, kind, network, lgcp):
    c = random_case(7, 1500, 80.0, kind, 1.5, network=network, lgcp=lgcp, seed=3, nhp=nhp, orc=orc)
    q = np.concatenate([[0.0, 80.0], np.random.default_rng(1).uniform(0, 80, 300), c["times"][[5, 100, 1499]]])
    got = nhp.intensity(c["proc"], c["data"], q)
    want = orc.intensity(c["om"], c["times"], c["nodes"], q)
    assert got.shape == (len(q), 7)
    assert np.max(np.abs(got - want) / want) < 1e-12
    one = nhp.intensity(c["proc"], c["data"], float(q[7]))          # scalar time -> length-N vector
    assert one.shape == (7,) and np.array_equal(one, got[7])
    if not lgcp:
        with pytest.raises(nhp.DomainError):
            nhp.intensity(c["proc"], c["data"], np.array([-1.0]))   # src/baselines.jl:111
    else:
        with pytest.raises(nhp.DomainError):
            nhp.intensity(c["proc"], c["data"], np.array([80.5]))   # outside the interpolation support


@pytest.mark.parametrize("kind,recursive,network", [("exponential", False, False), ("exponential", True, False),
                                                    ("exponential", False, True), ("exponential", True, True),
                                                    ("logitnormal", False, False), ("logitnormal", False, True)])
def test_gradient_matches_oracle(nhp, orc, kind, recursive, network):
    dtm = np.inf if recursive else 1.5
    c = random_case(6, 2500, 150.0, kind, dtm, network=network, seed=17, nhp=nhp, orc=orc)
    ll, g = nhp.loglikelihood_gradient(c["proc"], c["data"], recursive=recursive) if not network else \
        _grad_network(nhp, c, recursive)
    wll, wg = orc.loglik_grad(c["om"], c["times"], c["nodes"], c["T"], recursive=recursive)
    assert rel(ll, wll) < 1e-11
    assert np.max(np.abs(g - wg) / np.maximum(1.0, np.abs(wg))) < 1e-9


@pytest.mark.parametrize("network,lgcp", [(False, False), (True, False), (False, True)])
def test_gradient_over_the_slices_equals_the_two_pass_route(nhp, orc, network, lgcp, monkeypatch):
    """Exponential impulses on the dataset's own windows: log-likelihood and gradient come from ONE launch over the child
    slices (λ_k, 1/λ_k in LDS) and the parent slices (lane = parent node: no atomics) -- cont_slices.hip.  Against the oracle
    and against the two-pass route (NHP_GRAD_SLICES=0), with columns cut into several items (the kernel adds to k_grad_init's
    terms), ties and a burst in the data, and every (workgroup, rows per request) shape."""
    c = random_case(7, 5000, 300.0, "exponential", 1.0, network=network, lgcp=lgcp, seed=23, nhp=nhp, orc=orc)
    t = c["times"].copy()
    t[700:730:2] = t[701:731:2]
    t[2000:2090] = np.sort(np.random.default_rng(5).uniform(t[2000], t[2000] + 0.6, 90))
    t = np.sort(t)
    data = (t, c["nodes"], c["T"])
    wll, wg = orc.loglik_grad(c["om"], t, c["nodes"], c["T"], recursive=False)

    def run():
        nhp.invalidate_device_datasets()
        if network:
            return _grad_network(nhp, dict(c, data=data), False)
        return nhp.loglikelihood_gradient(c["proc"], data, recursive=False)

    got = {}
    for name, env in (("slices", {}), ("two-pass", {"NHP_GRAD_SLICES": "0"})):
        for k in ("NHP_GRAD_SLICES", "NHP_SLICES_CFG"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        ll, g = run()
        assert rel(ll, wll) < 1e-11, name
        assert np.max(np.abs(g - wg) / np.maximum(1.0, np.abs(wg))) < 1e-9, name
        got[name] = g
    assert np.max(np.abs(got["slices"] - got["two-pass"]) / np.maximum(1.0, np.abs(wg))) < 1e-10
    monkeypatch.delenv("NHP_GRAD_SLICES", raising=False)
    for cfg in ("64,2", "128,4", "256,8", "512,2", "1024,4"):
        monkeypatch.setenv("NHP_SLICES_CFG", cfg)
        ll, g = run()
        assert rel(ll, wll) < 1e-11, cfg
        assert np.max(np.abs(g - wg) / np.maximum(1.0, np.abs(wg))) < 1e-9, cfg
    monkeypatch.delenv("NHP_SLICES_CFG", raising=False)


def test_gradient_over_the_slices_one_item_per_node(nhp, orc, monkeypatch):
    """At N >= 1024 every node is one item: the slice kernel then STORES every entry of the gradient (the parameter-
    independent terms -T, -cnt[p] included) and k_grad_init does not run.  A node without events and one with a single
    event are in the data."""
    N, M = 1040, 60000
    monkeypatch.delenv("NHP_SLICES", raising=False)                 # (this test is about the slices' own route)
    monkeypatch.setenv("NHP_CHUNK", "4096")                         # (58 events a node: the 1.3 x mean item bound would cut some nodes in two)
    c = random_case(N, M, 4000.0, "exponential", 1.0, seed=31, nhp=nhp, orc=orc)
    n = c["nodes"].copy()
    n[n == 7] = 8
    n[n == 11] = 12
    n[4321] = 11
    data = (c["times"], n, c["T"])
    wll, wg = orc.loglik_grad(c["om"], c["times"], n, c["T"], recursive=False)
    for env in ({}, {"NHP_GRAD_SLICES": "0"}):
        monkeypatch.delenv("NHP_GRAD_SLICES", raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        nhp.invalidate_device_datasets()
        g0 = np.full(len(wg), np.nan)                               # (the call fills every entry)
        ll, g = nhp.loglikelihood_gradient(c["proc"], data, recursive=False)
        assert rel(ll, wll) < 1e-11
        assert np.all(np.isfinite(g)) and g.shape == g0.shape
        assert np.max(np.abs(g - wg) / np.maximum(1.0, np.abs(wg))) < 1e-9
    monkeypatch.delenv("NHP_GRAD_SLICES", raising=False)
    # no atomics on this path and the parent slices are sorted lane by lane after the ticketed fill: two builds of the
    # dataset give the same bits
    nhp.invalidate_device_datasets()
    ll1, g1 = nhp.loglikelihood_gradient(c["proc"], data, recursive=False)
    nhp.invalidate_device_datasets()
    ll2, g2 = nhp.loglikelihood_gradient(c["proc"], data, recursive=False)
    assert ll1 == ll2 and np.array_equal(g1, g2)
    monkeypatch.delenv("NHP_CHUNK", raising=False)
    nhp.invalidate_device_datasets()


def _grad_network(nhp, c, recursive):
    # the network process has no params!/mle! in the reference (src/continuous.jl:325-333), but the
    # kernels differentiate it all the same; P is the standard process's [λ0; θ|μ;τ; W]
    import ctypes as C
    from nhp_amd import _lib
    proc = c["proc"]
    ctx = nhp.default_context()
    ds = nhp.device_dataset(proc, c["data"], ctx)
    model = proc.device_model(ctx)
    N = proc.ndims()
    P = N + N * N * (2 if isinstance(proc.impulses, nhp.ExponentialImpulseResponse) else 3)
    g = np.empty(P)
    ll = C.c_double()
    flags = _lib.LL_RECURSIVE if recursive else 0
    _lib.check(_lib.lib().nhp_cont_loglik_grad(ctx.h, ds.h, model.h, flags, C.byref(ll), _lib.dptr(g), P), ctx.h)
    return ll.value, g


def test_mle_recovers_the_maximum(nhp, orc):
    # C1-sized problem: N=2 README model, simulated by the branching sampler
    proc, data = nhp.synthetic.readme_case(seed=1)
    truth = proc.params().copy()
    ll_truth = nhp.loglikelihood(proc, data)
    res = nhp.mle_(proc, data, seed=0)
    assert res.status == "success"
    assert res.maximum >= ll_truth - 1e-6                  # the optimum beats the generating parameters
    assert np.array_equal(proc.params(), res.maximizer)    # process overwritten in place
    assert np.all(res.maximizer >= 1e-6) and np.all(res.maximizer <= 10.0)
    # gradient ~ 0 at interior coordinates of the maximiser
    _, g = nhp.loglikelihood_gradient(proc, data)
    interior = (res.maximizer > 1e-5) & (res.maximizer < 9.99)
    assert np.max(np.abs(g[interior])) < 5e-2 * max(1.0, abs(res.maximum)) ** 0.5
    lam_hat = res.maximizer[:2]
    assert np.all(np.abs(lam_hat - truth[:2]) < 0.25)      # baseline rates ≈ 1.0 with T = 1000
    # regularised objective runs and returns a finite optimum
    res2 = nhp.mle_(proc, data, regularize=True, guess=res.maximizer)
    assert np.isfinite(res2.maximum)


def test_mcmc_runs_and_is_reproducible(nhp):
    c = random_case(4, 3000, 300.0, "logitnormal", 1.0, network=True, seed=23, nhp=nhp)
    c2 = random_case(4, 3000, 300.0, "logitnormal", 1.0, network=True, seed=23, nhp=nhp)
    r1 = nhp.mcmc_(c["proc"], c["data"], nsteps=25, seed=7)
    r2 = nhp.mcmc_(c2["proc"], c2["data"], nsteps=25, seed=7)
    assert r1.steps == 25 and len(r1.samples) == 25 and r1.status == "complete"
    assert all(np.array_equal(a, b) for a, b in zip(r1.samples, r2.samples))   # same seed -> same chain
    r3 = nhp.mcmc_(c2["proc"], c2["data"], nsteps=5, seed=8)
    assert not np.array_equal(r3.samples[0], r1.samples[0])
    assert len(r1.samples[0]) == 1 + 4 + 16 + 32 + 16                           # [ρ; λ0; W; μ; τ; vec(A)]
    assert all(np.all(np.isfinite(s)) for s in r1.samples)


def test_device_draws_match_their_distributions(nhp):
    # one device sweep on a model whose statistics we also fetch: the N² conjugate draws must have the
    # moments of Gamma(shape, scale) / Normal the reference's resample! bodies prescribe
    import ctypes as C
    from nhp_amd import _lib, inference
    N = 48
    c = random_case(N, 40000, 2000.0, "logitnormal", 1.0, seed=31, nhp=nhp)
    proc, ctx = c["proc"], nhp.default_context()
    ds = nhp.device_dataset(proc, c["data"], ctx)
    _, _, st = nhp.resample_parents(proc, ds, seed=9, step=4, with_stats=True, want_parents=False)
    model = proc.device_model(ctx)
    pri = inference._priors(proc)
    _lib.check(_lib.lib().nhp_cont_gibbs_step(ctx.h, ds.h, model.h, C.byref(pri), 9, 4), ctx.h)
    inference._pull_params(proc, model, ctx)
    Mnm, Mn, X, V = st["Mnm"], st["Mn"], st["Xnm"], st["Vnm"]
    # W ~ Gamma(κ + Mnm, 1/(ν + Mn[p])): standardised draws have mean 0, variance 1
    shape, rate = 1.0 + Mnm, (1.0 + Mn)[:, None] * np.ones((N, N))
    z = (proc.weights.W - shape / rate) / (np.sqrt(shape) / rate)
    assert abs(z.mean()) < 4 / N and abs(z.var() - 1.0) < 0.15
    lam_shape, lam_rate = 1.0 + st["cnt0"], 1.0 + c["T"]
    zl = (proc.baseline.λ - lam_shape / lam_rate) / (np.sqrt(lam_shape) / lam_rate)
    assert abs(zl.mean()) < 0.6 and 0.4 < zl.var() < 2.0
    with np.errstate(invalid="ignore"):
        bnm = V / 2 + Mnm * 1.0 / (Mnm + 1.0) * (X - 1.0) ** 2 / 2
        bnm[np.isnan(bnm)] = 1.0
        mnm = (1.0 + Mnm * X) / (1.0 + Mnm)
        mnm[np.isnan(mnm)] = 1.0
    tshape = 1.0 + Mnm / 2
    zt = (proc.impulses.τ - tshape / bnm) / (np.sqrt(tshape) / bnm)
    assert abs(zt.mean()) < 4 / N and abs(zt.var() - 1.0) < 0.15
    zm = (proc.impulses.μ - mnm) * np.sqrt((1.0 + Mnm) * proc.impulses.τ)
    assert abs(zm.mean()) < 4 / N and abs(zm.var() - 1.0) < 0.1
    assert np.all(proc.weights.W > 0) and np.all(proc.impulses.τ > 0)


def test_mcmc_posterior_concentrates(nhp):
    # exponential standard process, data simulated from known parameters: posterior means land near them
    lam0, W, th = np.array([0.8, 1.2]), np.array([[0.3, 0.1], [0.05, 0.25]]), np.full((2, 2), 2.0)
    data = nhp.synthetic.branching_sample(lam0, W, th, 3000.0, seed=3)
    proc = nhp.ContinuousStandardHawkesProcess(nhp.HomogeneousProcess(np.ones(2)),
                                               nhp.ExponentialImpulseResponse(np.ones((2, 2)), 1.0, 1.0, 10.0),
                                               nhp.DenseWeightModel(0.5 * np.ones((2, 2))))
    for device_draws in (True, False):
        proc.params_(np.concatenate([np.ones(2), np.ones(4), 0.5 * np.ones(4)]))
        res = nhp.mcmc_(proc, data, nsteps=300, seed=1, device_draws=device_draws)
        post = np.mean(res.samples[100:], axis=0)
        assert np.all(np.abs(post[:2] - lam0) < 0.12)
        assert np.all(np.abs(post[6:].reshape((2, 2), order="F") - W) < 0.08)


@pytest.mark.parametrize("kind,lgcp", [("exponential", False), ("logitnormal", False), ("exponential", True)])
def test_adjacency_gibbs_matches_oracle(nhp, orc, kind, lgcp):
    # resample_adjacency_matrix! (src/continuous.jl:444-519) with explicit Bernoulli uniforms: the
    # GPU sweep (every pair evaluated once, incremental λ updates) must take the same N² decisions as
    # the oracle's literal restatement (two full event scans per entry)
    N = 7
    c = random_case(N, 1500, 120.0, kind, 1.5, network=True, lgcp=lgcp, seed=41, nhp=nhp, orc=orc)
    u = np.random.default_rng(3).uniform(size=(N, N))
    c["proc"].network.ρ = 0.35
    want = orc.resample_adjacency(c["om"], c["times"], c["nodes"], c["T"], 0.35, u)
    links = nhp.resample_adjacency_matrix_(c["proc"], c["data"], u=u)
    assert np.array_equal(c["proc"].adjacency_matrix, want)
    assert links == want.sum()
    assert 0 < want.sum() < N * N                                    # a non-trivial draw
    # ρ = 1 (DenseNetworkModel): log(1-ρ) = -Inf -> every link present
    c["proc"].network = nhp.DenseNetworkModel(N)
    nhp.resample_adjacency_matrix_(c["proc"], c["data"], u=u)
    assert np.all(c["proc"].adjacency_matrix == 1.0)


def test_network_mcmc_recovers_structure(nhp):
    # two blocks: node 1 excites node 2 strongly, nothing else; the adjacency posterior should find it
    lam0 = np.array([1.0, 0.2, 0.5])
    W = np.array([[0.0, 0.8, 0.0], [0.0, 0.0, 0.0], [0.0, 0.0, 0.0]])
    th = np.full((3, 3), 3.0)
    data = nhp.synthetic.branching_sample(lam0, W, th, 2500.0, seed=5)
    proc = nhp.ContinuousNetworkHawkesProcess(nhp.HomogeneousProcess(np.ones(3)),
                                              nhp.ExponentialImpulseResponse(np.ones((3, 3)), 1.0, 1.0, 5.0),
                                              nhp.DenseWeightModel(0.3 * np.ones((3, 3))), np.ones((3, 3)),
                                              nhp.BernoulliNetworkModel(0.5, 3))
    res = nhp.mcmc_(proc, data, nsteps=250, seed=2)
    S = np.array(res.samples[80:])
    A_mean = S[:, -9:].mean(axis=0).reshape((3, 3), order="F")
    W_mean = S[:, 4:13].mean(axis=0).reshape((3, 3), order="F")
    assert A_mean[0, 1] > 0.95 and (A_mean * W_mean)[0, 1] > 0.5
    eff = A_mean * W_mean
    eff[0, 1] = 0.0
    assert eff.max() < 0.15


@pytest.mark.parametrize("kind", ["exponential", "logitnormal"])
def test_gradient_with_lgcp_baseline(nhp, orc, kind):
    # params order [vcat(λ...); θ | μ; τ; W] (src/baselines.jl:173, src/continuous.jl:116-119): GPU vs oracle,
    # and the oracle itself vs central differences of its log-likelihood
    from helpers import random_case
    c = random_case(4, 1500, 60.0, kind, 1.5, lgcp=True, seed=31, nhp=nhp, orc=orc)
    proc, om = c["proc"], c["om"]
    for rec in ((False, True) if kind == "exponential" else (False,)):
        ll, g = nhp.loglikelihood_gradient(proc, c["data"], recursive=rec)
        wll, wg = orc.loglik_grad(om, c["times"], c["nodes"], c["T"], recursive=rec)
        assert len(g) == len(proc.params()) == len(wg)
        assert abs(ll - wll) < 1e-10 * abs(wll)
        assert np.max(np.abs(g - wg) / np.maximum(1.0, np.abs(wg))) < 1e-9
    G = len(proc.baseline.x)
    lam = np.vstack(proc.baseline.λ)
    for (node, k) in ((0, 0), (1, 5), (3, G - 1), (2, 9)):
        h = 1e-6
        vals = []
        for sgn in (+1, -1):
            l2 = lam.copy()
            l2[node, k] += sgn * h
            m2 = orc.ContModel(l2, proc.weights.W, theta=getattr(proc.impulses, "θ", None), mu=getattr(proc.impulses, "μ", None),
                               tau=getattr(proc.impulses, "τ", None), dt_max=1.5, grid_x=proc.baseline.x)
            vals.append(orc.loglik_windowed(m2, c["times"], c["nodes"], c["T"]))
        fd = (vals[0] - vals[1]) / (2 * h)
        _, wg = orc.loglik_grad(om, c["times"], c["nodes"], c["T"], recursive=False)
        assert abs(wg[node * G + k] - fd) < 1e-5 * max(1.0, abs(fd))


def test_mle_with_lgcp_baseline_improves_the_likelihood(nhp):
    # examples/continuous-logit-normal-standard-hawkes-gp.jl:38 runs mle! on an LGCP-baseline process
    from helpers import random_case
    c = random_case(3, 1200, 40.0, "exponential", 1.0, lgcp=True, seed=5, nhp=nhp)
    proc = c["proc"]
    ll0 = nhp.loglikelihood(proc, c["data"])
    res = nhp.mle_(proc, c["data"], guess=np.clip(proc.params(), 1e-3, 5.0), max_steps=60)
    assert res.maximum > ll0
    assert abs(nhp.loglikelihood(proc, c["data"]) - res.maximum) < 1e-8 * abs(res.maximum)
    assert len(res.maximizer) == len(proc.params())


def test_gibbs_entry_points_on_empty_data(nhp):
    # zero events: the sweep must still run (statistics all zero, adjacency decided by the prior alone)
    N = 3
    proc = nhp.ContinuousNetworkHawkesProcess(nhp.HomogeneousProcess(np.ones(N)),
                                              nhp.ExponentialImpulseResponse(np.ones((N, N)), 1.0, 1.0, 1.0),
                                              nhp.DenseWeightModel(np.full((N, N), 0.2)), np.ones((N, N)),
                                              nhp.BernoulliNetworkModel(0.4, N))
    data = (np.array([]), np.array([], dtype=np.int64), 5.0)
    p, pn, st = nhp.resample_parents(proc, data, with_stats=True)
    assert len(p) == 0 and not st["Mnm"].any() and not st["cnt0"].any()
    u = np.random.default_rng(1).uniform(size=(N, N))
    nhp.resample_adjacency_matrix_(proc, data, u=u)
    assert np.array_equal(proc.adjacency_matrix, (u <= 0.4).astype(float))
    res = nhp.mcmc_(proc, data, nsteps=3, seed=0)
    assert res.steps == 3 and all(np.all(np.isfinite(s)) for s in res.samples)


@pytest.mark.parametrize("kind,network", [("exponential", False), ("logitnormal", True)])
def test_device_moments_equal_the_moments_of_the_kept_samples(nhp, kind, network):
    """The on-device sample store (nhp_cont_model_moments_*): mean and mean square of params(process) over the steps
    after burn-in, with no per-step transfer, equal numpy's over the samples mcmc! keeps (same chain, same order)."""
    c = random_case(5, 2500, 250.0, kind, 1.0, network=network, seed=31, nhp=nhp)
    res = nhp.mcmc_(c["proc"], c["data"], nsteps=40, seed=3, keep_samples=True, moments=True, burn=10)
    S = np.array(res.samples[10:])
    assert res.n == 30 and res.mean.shape == S[0].shape
    assert np.allclose(res.mean, S.mean(axis=0), rtol=1e-12, atol=1e-14)
    assert np.allclose(res.m2, (S ** 2).mean(axis=0), rtol=1e-12, atol=1e-14)
    # without kept samples the chain and its moments are the same
    c2 = random_case(5, 2500, 250.0, kind, 1.0, network=network, seed=31, nhp=nhp)
    r2 = nhp.mcmc_(c2["proc"], c2["data"], nsteps=40, seed=3, keep_samples=False, moments=True, burn=10)
    assert np.array_equal(r2.mean, res.mean) and np.array_equal(r2.m2, res.m2)
    assert np.array_equal(r2.samples[-1], res.samples[-1])
    with pytest.raises(ValueError):
        nhp.mcmc_(c2["proc"], c2["data"], nsteps=2, device_draws=False, moments=True)


def test_run_chains_summaries(nhp):
    """chains.run_chains (one process: every chain on this GPU): each chain's summary equals the mean / mean square of
    the samples the same chain gives when it is run by hand and its samples are kept."""
    from nhp_amd import chains

    def make(k):
        return random_case(4, 2000, 200.0, "logitnormal", 1.0, network=True, seed=41, nhp=nhp)["proc"]
    data = random_case(4, 2000, 200.0, "logitnormal", 1.0, network=True, seed=41, nhp=nhp)["data"]
    out = chains.run_chains(make, data, n_chains=3, nsteps=30, base_seed=5, burn=10)
    assert sorted(out) == [0, 1, 2]
    for k in range(3):
        res = nhp.mcmc_(make(k), data, nsteps=30, seed=chains.chain_seed(5, k))
        S = np.array(res.samples[10:])
        assert out[k]["n"][0] == 20
        assert np.allclose(out[k]["mean"], S.mean(axis=0), rtol=1e-12, atol=1e-14)
        assert np.allclose(out[k]["m2"], (S ** 2).mean(axis=0), rtol=1e-12, atol=1e-14)
    assert not np.array_equal(out[0]["mean"], out[1]["mean"])


def test_sampler_error_is_reported_immediately_or_one_sweep_late(nhp):
    """Weights that sum to zero (λ0 = 0, W = 0): resample_parents raises at once; nhp_cont_gibbs_step does not drain the
    GPU inside a sweep, so its flag surfaces at the next sweep or at the next call that synchronises -- and only once."""
    import ctypes as C
    from nhp_amd import _lib, inference
    ctx = nhp.Context(0)
    rng = np.random.default_rng(3)
    N, M, T = 3, 400, 50.0
    data = (np.sort(rng.uniform(0, T, M)), rng.integers(1, N + 1, M).astype(np.int64), T)

    def make(zero):
        lam0 = np.zeros(N) if zero else np.ones(N)
        W = np.zeros((N, N)) if zero else np.full((N, N), 0.1)
        return nhp.ContinuousStandardHawkesProcess(nhp.HomogeneousProcess(lam0), nhp.ExponentialImpulseResponse(np.ones((N, N)), 1.0, 1.0, 1.0),
                                                   nhp.DenseWeightModel(W))
    bad, good = make(True), make(False)
    with pytest.raises(nhp.DomainError, match="positive finite"):
        nhp.resample_parents(bad, data, seed=1, step=0, ctx=ctx)
    ds = nhp.device_dataset(bad, data, ctx)
    mb, mg = bad.device_model(ctx), good.device_model(ctx)
    pri = inference._priors(bad)
    lib = _lib.lib()
    assert lib.nhp_cont_gibbs_step(ctx.h, ds.h, mb.h, C.byref(pri), 1, 0) == _lib.OK          # enqueued, not yet judged
    assert lib.nhp_cont_gibbs_step(ctx.h, ds.h, mg.h, C.byref(pri), 1, 1) == _lib.EDOMAIN     # the sweep before this one
    assert b"earlier sweep" in lib.nhp_last_error(ctx.h)
    assert lib.nhp_cont_gibbs_step(ctx.h, ds.h, mg.h, C.byref(pri), 1, 2) == _lib.OK          # reported once
    ctx.synchronize()
    mb = bad.device_model(ctx)                           # (the first sweep drew new parameters into the device model)
    assert lib.nhp_cont_gibbs_step(ctx.h, ds.h, mb.h, C.byref(pri), 1, 3) == _lib.OK
    with pytest.raises(nhp.DomainError, match="earlier sweep"):
        ctx.synchronize()                                                                     # a synchronising call reports it too
    ctx.synchronize()
    res = nhp.mcmc_(make(False), data, nsteps=5, seed=2, ctx=ctx)                             # the context is usable afterwards
    assert res.steps == 5 and np.all(np.isfinite(res.samples[-1]))


@pytest.mark.parametrize("kind,recursive,lgcp", [("exponential", True, False), ("exponential", False, False),
                                                 ("logitnormal", False, False), ("exponential", True, True)])
def test_device_mle_reaches_a_maximum_the_host_optimizer_cannot_improve(nhp, orc, kind, recursive, lgcp):
    # nhp_cont_mle_run (projected L-BFGS, state in HBM) against scipy's L-BFGS-B fed the same gradient, same objective, same
    # box [1e-6, 10] (src/continuous.jl:185-186).  The log-likelihood is not concave in θ, so two methods started at one guess
    # may end in different local maxima; what is checked is that the device route's answer IS one: inside the box, the
    # projected gradient vanishes, the host optimizer started there finds nothing better, the device optimizer started at the
    # host route's optimum does not lose it, the value is the oracle's, and the process is overwritten in place.
    def case():
        return random_case(5, 3000, 250.0, kind, 1.5, lgcp=lgcp, seed=31, nhp=nhp, orc=orc)
    c = case()
    guess = np.random.default_rng(5).uniform(0.2, 0.8, len(c["proc"].params()))
    ll0 = nhp.loglikelihood(_set(case()["proc"], guess), c["data"], recursive=recursive)
    dev = nhp.mle_(c["proc"], c["data"], guess=guess, recursive=recursive, f_abstol=1e-9, max_steps=5000, optimizer="device")
    scale = max(1.0, abs(dev.maximum)) ** 0.5
    assert dev.status == "success" and dev.maximum > ll0
    x = dev.maximizer
    assert np.all(x >= 1e-6) and np.all(x <= 10.0)
    assert np.array_equal(c["proc"].params(), x)
    ll, g = nhp.loglikelihood_gradient(c["proc"], c["data"], recursive=recursive)
    assert ll == pytest.approx(dev.maximum, rel=1e-12)
    pg = np.where(((x <= 1e-6) & (g < 0)) | ((x >= 10.0) & (g > 0)), 0.0, g)       # ascent directions the box allows
    assert np.max(np.abs(pg)) < 5e-2 * scale
    assert rel(dev.maximum, orc.loglik(_with_params(orc, c["om"], c["proc"], lgcp), c["times"], c["nodes"], c["T"], recursive=recursive)) < 1e-10
    # (the host route runs without scipy's own relative-decrease test since round 3, so its polish goes on where a flat direction
    #  still yields 1e-9 a step; the device run's path -- and where it meets |Δf| < 1e-9 -- varies with the order of the
    #  gradient's atomics at this size)
    polish = nhp.mle_(case()["proc"], c["data"], guess=x, recursive=recursive, f_abstol=1e-9, max_steps=3000)
    assert -1e-9 * scale <= polish.maximum - dev.maximum < 1e-3 * scale
    host = nhp.mle_(case()["proc"], c["data"], guess=guess, recursive=recursive, f_abstol=1e-9, max_steps=3000)
    back = nhp.mle_(case()["proc"], c["data"], guess=host.maximizer, recursive=recursive, f_abstol=1e-9, max_steps=3000, optimizer="device")
    assert back.maximum >= host.maximum - 1e-9 * scale


def test_device_mle_with_most_weights_on_the_lower_bound(nhp, orc):
    """A sparse truth: three quarters of W are zero, so at the optimum most weights sit on the box's lower bound 1e-6 and
    the projection clips most quasi-Newton steps.  The line search accepts a trial only along a path that descends to first
    order and never with a larger objective (csrc/nhp_lbfgs.h): the run must end at a point of the box where the projected
    gradient vanishes, not below its start, with the clipped coordinates exactly on the bound, and the host optimizer
    started there must find nothing better."""
    N, T = 8, 600.0
    rng = np.random.default_rng(12)
    W = rng.uniform(0.1, 0.4, (N, N)) * (rng.uniform(size=(N, N)) < 0.25)          # 24 links of 64, spectral radius 0.72
    proc = nhp.ContinuousStandardHawkesProcess(nhp.HomogeneousProcess(rng.uniform(0.5, 1.0, N)),
                                               nhp.ExponentialImpulseResponse(rng.uniform(2.0, 4.0, (N, N)), 1.0, 1.0, 2.0),
                                               nhp.DenseWeightModel(W))
    data = nhp.synthetic.rand(proc, T, seed=4)
    assert len(data[0]) > 3000
    guess = np.random.default_rng(6).uniform(0.3, 0.9, len(proc.params()))
    ll0 = nhp.loglikelihood(_set(proc, guess), data, recursive=False)
    # (the θ of a link without weight is a flat direction: 5-10 thousand steps to |Δf| < 1e-10 from this start, with a plateau
    #  of ~3000 steps near 2418.14 before a link leaves the bound, for the device optimizer and for scipy's L-BFGS-B with its
    #  own tests off alike -- 0.7 s)
    dev = nhp.mle_(proc, data, guess=guess, recursive=False, f_abstol=1e-10, max_steps=40000, optimizer="device")
    x = dev.maximizer
    assert dev.status == "success" and dev.maximum >= ll0 and np.all(x >= 1e-6) and np.all(x <= 10.0)
    Wfit = x[N + N * N:].reshape((N, N), order="F")
    on_bound = Wfit == 1e-6
    assert on_bound.sum() >= N * N // 4                             # (24 of the 40 true zeros end exactly on the bound)
    ll, g = nhp.loglikelihood_gradient(proc, data, recursive=False)
    assert ll == pytest.approx(dev.maximum, rel=1e-12)
    scale = max(1.0, abs(dev.maximum)) ** 0.5
    pg = np.where(((x <= 1e-6) & (g < 0)) | ((x >= 10.0) & (g > 0)), 0.0, g)
    assert np.max(np.abs(pg)) < 5e-2 * scale
    # the host optimizer from there: nothing lower, and nothing better beyond what the landscape's other stationary points
    # are apart (runs end between 2716.3 and 2718.45 here: which links leave the bound depends on the path, and the path on
    # the summation order of the gradient's atomics)
    polish = nhp.mle_(proc, data, guess=x, recursive=False, f_abstol=1e-10, max_steps=3000)
    assert -1e-9 * scale <= polish.maximum - dev.maximum < 1e-3 * abs(dev.maximum)


def test_device_optimizer_on_quadratics_against_scipy(nhp):
    """csrc/nhp_lbfgs.h alone (nhp_probe_lbfgs: f = ½ Σ h_i (x_i - c_i)² on the box [1e-6, 10]^n), against scipy's L-BFGS-B
    with the same eight pairs, the same start and the same stopping rule (|f_k - f_{k-1}| < 1e-10, scipy's own tests off):
    the same minimiser -- clipped coordinates exactly on the bound -- in a comparable number of steps."""
    import ctypes as C
    from scipy import optimize
    from nhp_amd import _lib
    ctx = _lib.default_context()
    rng = np.random.default_rng(0)
    n = 136
    for cond, outside in ((1e2, 0.0), (1e2, 0.3), (1e4, 0.0), (1e4, 0.3)):
        h = np.exp(rng.uniform(0.0, np.log(cond), n))
        c = rng.uniform(1.0, 5.0, n)
        below = rng.uniform(size=n) < outside
        c[below] = -1.0                                             # these coordinates' minimiser lies below the box
        x0 = rng.uniform(0.5, 9.0, n)
        x = x0.copy()
        loss, steps, conv, ev = C.c_double(), C.c_int32(), C.c_int32(), C.c_int32()
        _lib.check(_lib.lib().nhp_probe_lbfgs(ctx.h, n, _lib.dptr(h), _lib.dptr(c), 1e-6, 10.0, 1e-10, 20000, _lib.dptr(x), C.byref(loss),
                                              C.byref(steps), C.byref(conv), C.byref(ev)), ctx.h)

        def f(z):
            d = z - c
            return 0.5 * np.sum(h * d * d), h * d
        state = {"prev": np.inf, "it": 0}

        def cb(z):
            v = f(z)[0]
            state["it"] += 1
            if abs(v - state["prev"]) < 1e-10:
                raise StopIteration
            state["prev"] = v
        optimize.minimize(f, x0, jac=True, method="L-BFGS-B", bounds=[(1e-6, 10.0)] * n, callback=cb,
                          options=dict(maxiter=20000, ftol=0.0, gtol=0.0, maxcor=8))
        xs = np.clip(c, 1e-6, 10.0)
        fopt = f(xs)[0]
        assert conv.value == 1 and loss.value == pytest.approx(f(x)[0], rel=1e-12)
        assert loss.value - fopt < 1e-7 * max(1.0, cond / 1e4)      # (scipy ends 2e-10 ... 8e-9 above the minimum on these)
        assert np.all(x[below] == 1e-6) and np.max(np.abs(x - xs)) < 1e-3
        assert steps.value <= 1.5 * state["it"] + 20, (cond, outside, steps.value, state["it"])
        assert ev.value <= 1.15 * steps.value + 10                  # the unit step is accepted almost always


def _set(proc, x):
    proc.params_(np.asarray(x, dtype=np.float64))
    return proc


def _with_params(orc, om, proc, lgcp):
    base = proc.baseline
    lam0 = np.array(base.λ) if lgcp else base.λ
    gx = base.x if lgcp else None
    if hasattr(proc.impulses, "θ"):
        return orc.ContModel(lam0, proc.weights.W, theta=proc.impulses.θ, dt_max=proc.impulses.Δtmax, grid_x=gx)
    return orc.ContModel(lam0, proc.weights.W, mu=proc.impulses.μ, tau=proc.impulses.τ, dt_max=proc.impulses.Δtmax, grid_x=gx)


def test_device_mle_error_behaviour(nhp):
    # the reference's mle! is defined for the standard process only (src/continuous.jl:144); a wrong-length guess is the
    # reference's "Parameter vector length does not match model parameter length." (src/continuous.jl:121-129)
    import ctypes as C
    from nhp_amd import _lib
    c = random_case(4, 1500, 120.0, "exponential", 1.0, network=True, seed=3, nhp=nhp)
    with pytest.raises(TypeError):
        nhp.mle_(c["proc"], c["data"], optimizer="device")
    s = random_case(4, 1500, 120.0, "exponential", 1.0, seed=3, nhp=nhp)
    ctx = nhp.default_context()
    ds, model = nhp.device_dataset(s["proc"], s["data"], ctx), s["proc"].device_model(ctx)
    x = np.full(7, 0.5)
    out = (C.c_double(), C.c_int32(), C.c_int32(), C.c_int32())
    rc = _lib.lib().nhp_cont_mle_run(ctx.h, None, ds.h, model.h, 0, 1e-6, 10.0, 1e-6, 10, _lib.dptr(x), len(x), C.byref(out[0]),
                                     C.byref(out[1]), C.byref(out[2]), C.byref(out[3]))
    assert rc == 3 and b"Parameter vector length" in _lib.lib().nhp_last_error(ctx.h)        # NHP_ESHAPE
    net_model = c["proc"].device_model(ctx)
    xn = np.full(4 + 2 * 16, 0.5)
    rc = _lib.lib().nhp_cont_mle_run(ctx.h, None, nhp.device_dataset(c["proc"], c["data"], ctx).h, net_model.h, 0, 1e-6, 10.0, 1e-6, 10,
                                     _lib.dptr(xn), len(xn), C.byref(out[0]), C.byref(out[1]), C.byref(out[2]), C.byref(out[3]))
    assert rc != 0 and b"ContinuousStandardHawkesProcess" in _lib.lib().nhp_last_error(ctx.h)
    with pytest.raises(NotImplementedError):
        nhp.mle_(s["proc"], s["data"], optimizer="device", regularize=True)
    # max_steps = 0: the clamped guess comes back, with its log-likelihood
    g = np.random.default_rng(2).uniform(0.2, 0.8, len(s["proc"].params()))
    r = nhp.mle_(s["proc"], s["data"], guess=g, optimizer="device", max_steps=0)
    assert r.steps == 0 and np.array_equal(r.maximizer, g) and r.maximum == pytest.approx(nhp.loglikelihood(s["proc"], s["data"]), rel=1e-12)
