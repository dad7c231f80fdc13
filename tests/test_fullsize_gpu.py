"""BASELINE.json configs 2, 3 and 4 at their REAL sizes on the GPU, checked the way
`test_metric_size_properties` checks the metric config: the oracle on random slices (it cannot run
the whole problem in seconds) plus size-independent properties.

config 2  continuous exponential standard, N=128, M=1e5: log-likelihood + analytic gradient vs the oracle
          (reference: src/continuous.jl:144-198,210-239).
config 3  continuous logit-normal NETWORK, N=1024, M=1e6: parent indices of random 1000-event slices
          bit-equal to the oracle's (src/parents.jl:1-46), count statistics exact (src/parents.jl:61-79,
          src/baselines.jl:87-96), adjacency decisions of whole columns equal to the oracle's literal
          restatement (src/continuous.jl:444-519).
config 4  discrete Gaussian basis, N=512, B=8, L=32, T=1e5: intensity rows of random 64-bin slices vs the
          oracle (src/discrete.jl:115-131,146-151), the Poisson log-likelihood from the returned
          intensity (src/discrete.jl:86-102), linearity of the contraction, and the invariants of one
          VB step (src/discrete.jl:369-375, src/parents.jl:136-177).
"""
import numpy as np
import pytest

from helpers import rel

pytestmark = pytest.mark.gpu


# ---------------------------------------------------------------------------------------------- config 2
def test_config2_gradient_matches_oracle_at_full_size(nhp, orc):
    N, M = 128, 100_000
    times, nodes, T = nhp.synthetic.s_metric_data(N, M, kbar=16.0)
    proc = nhp.synthetic.s_metric_process(N, M, T, "exponential", 1.0)
    om = orc.ContModel(proc.baseline.λ, proc.weights.W, theta=proc.impulses.θ, dt_max=1.0)
    data = (times, nodes, T)
    ll, g = nhp.loglikelihood_gradient(proc, data, recursive=False)
    wll, wg = orc.loglik_grad(om, times, nodes, T, recursive=False)
    assert len(g) == len(wg) == N + 2 * N * N
    assert rel(ll, wll) < 1e-6                      # contractual tolerance (BASELINE.json north_star)
    assert rel(ll, wll) < 1e-11
    assert np.max(np.abs(g - wg) / np.maximum(1.0, np.abs(wg))) < 1e-9
    # the recursive formulation's gradient (full history: O(M²) in the oracle) on the first 20 000 events
    m = 20_000
    sub = (times[:m], nodes[:m], float(times[m]))
    ll, g = nhp.loglikelihood_gradient(proc, sub, recursive=True)
    wll, wg = orc.loglik_grad(om, sub[0], sub[1], sub[2], recursive=True)
    assert rel(ll, wll) < 1e-11
    assert np.max(np.abs(g - wg) / np.maximum(1.0, np.abs(wg))) < 1e-9


# ---------------------------------------------------------------------------------------------- config 3
@pytest.fixture(scope="module")
def config3(nhp, orc):
    N, M = 1024, 1_000_000
    times, nodes, T = nhp.synthetic.s_metric_data(N, M, kbar=8.0)
    proc = nhp.synthetic.s_metric_process(N, M, T, "logitnormal", 1.0, network=True)
    om = orc.ContModel(proc.baseline.λ, proc.weights.W, mu=proc.impulses.μ, tau=proc.impulses.τ, dt_max=1.0,
                       A=proc.adjacency_matrix)
    return dict(N=N, M=M, times=times, nodes=nodes, T=T, data=(times, nodes, T), proc=proc, om=om)


def test_config3_parents_and_statistics_at_full_size(nhp, orc, config3):
    c = config3
    N, M, times, nodes = c["N"], c["M"], c["times"], c["nodes"]
    seed, step = 17, 3
    p, pn, st = nhp.resample_parents(c["proc"], c["data"], seed=seed, step=step, with_stats=True)
    u = orc.uniform_stream(seed, step, M)
    for i0 in (1, 271_828, 500_000, 998_999):
        i1 = i0 + 1000
        # the slice with its look-back prefix: the oracle sees events [a, i1); every window of [i0, i1) lies inside
        a = max(int(np.searchsorted(times, times[i0] - 1.0, side="right")) - 1, 0)
        wp, wpn = orc.resample_parents(c["om"], times[a:i1], nodes[a:i1], u[a:i1], flags=orc.MATH_DET)
        wp = np.where(wp > 0, wp + a, 0)[i0 - a:]
        assert np.array_equal(p[i0:i1], wp)                     # bit-exact indices (north_star)
        assert np.array_equal(pn[i0:i1], wpn[i0 - a:])
    assert p[0] == 0 and pn[0] == 0
    # a parent lies inside the look-back window and before its child; parentnodes are the parents' nodes
    has = p > 0
    idx = np.arange(M)
    assert np.all(p[has] - 1 < idx[has])
    assert np.all(times[p[has] - 1] > times[has] - 1.0)
    assert np.array_equal(pn[has], nodes[p[has] - 1]) and not pn[~has].any()
    # count statistics: exact integers (reference helpers src/parents.jl:61-79, src/baselines.jl:87-96)
    assert st["Mnm"].sum() + st["cnt0"].sum() == M
    assert np.array_equal(st["Mn"], np.bincount(nodes - 1, minlength=N).astype(float))
    assert np.array_equal(st["cnt0"], np.bincount(nodes[~has] - 1, minlength=N).astype(float))
    want = np.zeros((N, N))
    np.add.at(want, (pn[has] - 1, nodes[has] - 1), 1.0)
    assert np.array_equal(st["Mnm"], want)
    # logit-duration statistics of three (parent node, child node) cells recomputed from the returned parents
    d = times[has] - times[p[has] - 1]
    ell = np.log(d / (1.0 - d))
    cells = np.stack([pn[has] - 1, nodes[has] - 1])
    for k in (0, 12345, 99999):
        pp, cc = cells[:, k]
        sel = (cells[0] == pp) & (cells[1] == cc)
        x = ell[sel].sum() / sel.sum()
        assert abs(st["Xnm"][pp, cc] - x) < 1e-12 * max(1.0, abs(x))
        assert abs(st["Vnm"][pp, cc] - ((ell[sel] - x) ** 2).sum()) < 1e-10


def test_config3_adjacency_columns_at_full_size(nhp, orc, config3):
    c = config3
    N = c["N"]
    proc = nhp.synthetic.s_metric_process(N, c["M"], c["T"], "logitnormal", 1.0, network=True)   # same seeds: same model
    proc.network.ρ = 0.35
    u = np.random.default_rng(5).uniform(size=(N, N))
    links = nhp.resample_adjacency_matrix_(proc, c["data"], u=u)
    got = proc.adjacency_matrix
    assert links == got.sum() and 0 < links < N * N
    for col in (5, 777):
        want = orc.resample_adjacency_columns(c["om"], c["times"], c["nodes"], c["T"], 0.35, u, col, col + 1)
        assert np.array_equal(got[:, col], want[:, col])
        assert 0 < want[:, col].sum() < N


def test_config5_chains_at_full_size_on_one_gpu(nhp, config3):
    """BASELINE config 5 is 8 chains of config 3, one per GPU, gathered once.  What one GPU can rehearse at the REAL size: two
    chains of the N=1024, M=1e6 logit-normal network model run by chains.run_chains (the library's chain driver, posterior
    moments on the device) -- each chain's gathered summary equals what the same chain gives when it is run by hand, the two
    chains differ, and the sweep leaves a valid state (A binary, ρ in (0, 1), positive rates)."""
    from nhp_amd import chains
    c = config3

    def make(k):
        return nhp.synthetic.s_metric_process(c["N"], c["M"], c["T"], "logitnormal", 1.0, network=True)
    out = chains.run_chains(make, c["data"], n_chains=2, nsteps=6, base_seed=3, burn=2)
    assert sorted(out) == [0, 1] and out[0]["n"][0] == out[1]["n"][0] == 4
    P = 1 + c["N"] + 4 * c["N"] ** 2                             # [ρ; λ0; W; μ; τ; vec(A)]  src/continuous.jl:325-333
    assert out[0]["mean"].shape == (P,) and not np.array_equal(out[0]["mean"], out[1]["mean"])
    proc = make(0)
    res = nhp.mcmc_(proc, c["data"], nsteps=6, seed=chains.chain_seed(3, 0), keep_samples=False, moments=True, burn=2)
    assert np.array_equal(res.mean, out[0]["mean"]) and np.array_equal(res.m2, out[0]["m2"])
    A = proc.adjacency_matrix
    assert set(np.unique(A)) <= {0.0, 1.0} and 0.0 < proc.network.ρ < 1.0
    assert abs(A.mean() - proc.network.ρ) < 0.01                  # ρ | A ~ Beta(1 + ΣA, 1 + N² - ΣA), N² = 10⁶ entries
    assert np.all(proc.baseline.λ > 0) and np.all(proc.weights.W > 0) and np.all(proc.impulses.τ > 0)
    mean_A = out[0]["mean"][-c["N"] ** 2:]
    assert 0.0 <= mean_A.min() and mean_A.max() <= 1.0


def test_config3_loglikelihood_slices_at_full_size(nhp, orc, config3):
    # the logit-normal network kernel at full size: per-event intensities of random slices vs the oracle, and the
    # log-likelihood rebuilt from them (masked integral term, src/continuous.jl:368-371)
    c = config3
    N, times, nodes, T, proc = c["N"], c["times"], c["nodes"], c["T"], c["proc"]
    lam = nhp.total_intensity(proc, c["data"])
    for i0 in (0, 424_242, 999_000):
        want = orc.total_intensity(c["om"], times, nodes, i0, i0 + 1000)
        assert np.max(np.abs(lam[i0:i0 + 1000] - want) / want) < 1e-12
    AW = proc.adjacency_matrix * proc.weights.W
    integral = -(proc.baseline.λ * T).sum() - (np.bincount(nodes - 1, minlength=N) @ AW.sum(axis=1))
    assert rel(nhp.loglikelihood(proc, c["data"]), integral + np.log(lam).sum()) < 1e-12


# ---------------------------------------------------------------------------------------------- config 4
@pytest.fixture(scope="module")
def config4(nhp):
    N, B, L, T = 512, 8, 32, 100_000
    rng = np.random.default_rng(7)
    data = rng.poisson(0.05, (N, T)).astype(np.int64)
    th = rng.dirichlet(np.ones(B), (N, N))
    th[:, :, -1] = 1.0 - th[:, :, :-1].sum(axis=2)
    th = np.asfortranarray(th)
    lam0 = rng.uniform(0.02, 0.08, N)
    W = np.asfortranarray(rng.uniform(0, 1, (N, N)) / N)
    imp = nhp.DiscreteGaussianImpulseResponse.__new__(nhp.DiscreteGaussianImpulseResponse)
    imp.θ, imp.γ, imp.γv, imp.nlags, imp.dt, imp.ϕ = th, 1.0, np.ones_like(th), L, 1.0, None
    proc = nhp.DiscreteStandardHawkesProcess(nhp.DiscreteHomogeneousProcess(lam0, 1.0), imp, nhp.DenseWeightModel(W), 1.0)
    ds = nhp.convolve(proc, data)
    return dict(N=N, B=B, L=L, T=T, data=data, proc=proc, ds=ds, lam0=lam0, W=W, th=th)


def test_config4_intensity_and_loglik_at_full_size(nhp, orc, config4):
    from scipy.special import gammaln
    c = config4
    N, B, L, T, data, proc, ds = c["N"], c["B"], c["L"], c["T"], c["data"], c["proc"], c["ds"]
    phi = orc.disc_basis(L, B, 1.0)
    lam = nhp.intensity(proc, ds)
    assert lam.shape == (T, N)
    for t0 in (0, 31_415, T - 64):
        lo = max(0, t0 - L)
        conv = orc.disc_convolve(data[:, lo:t0 + 64], phi)[t0 - lo:]       # rows with their full L-lag history
        want = orc.disc_intensity(conv, c["lam0"], c["W"], c["th"], dt=1.0)
        assert np.max(np.abs(lam[t0:t0 + 64] - want) / want) < 1e-12
    # Poisson log-likelihood (fused GEMM epilogue) == the same sum over the intensity the other epilogue returned
    ll = nhp.loglikelihood(proc, data, convolved=ds)
    want = (data.T * np.log(lam)).sum() - lam.sum() - gammaln(data + 1.0).sum()
    assert abs(ll - want) < 1e-11 * abs(want)
    # the contraction is linear in W: λ(2W) - λ0 = 2 (λ(W) - λ0)
    proc.weights.W = 2.0 * c["W"]
    lam2 = nhp.intensity(proc, ds)
    proc.weights.W = c["W"]
    base = c["lam0"][None, :]
    assert np.allclose(lam2 - base, 2.0 * (lam - base), rtol=1e-12, atol=1e-16)


def test_parent_counts_through_the_large_tile(nhp, orc):
    """The parent-count kernel as config 4 runs it -- 128 x 128 bins x nodes a workgroup of 512 threads, two bins per thread from
    the column-major list, the second walk entered at a checkpoint, the node tiles of a bin range on one XCD (ids cy·8 + i)
    -- which the small cases of tests/test_discrete_gibbs_gpu.py never reach (N >= 256 and T >= 32768 select it).  N = 256,
    B = 8 (2048 categories), T = 32768 at 5 % occupancy: counts equal the oracle's, bit for bit; the other list and walk
    variants give the same counts."""
    import os
    N, B, L, T = 256, 8, 12, 32768
    rng = np.random.default_rng(5)
    data = rng.poisson(0.05, (N, T)).astype(np.int64)
    data[7, 1000] = 9                                                   # a bin with many events: thresholds in several eighths
    th = rng.dirichlet(np.ones(B), (N, N))
    th[:, :, -1] = 1.0 - th[:, :, :-1].sum(axis=2)
    imp = nhp.DiscreteGaussianImpulseResponse.__new__(nhp.DiscreteGaussianImpulseResponse)
    imp.θ, imp.γ, imp.γv, imp.nlags, imp.dt, imp.ϕ = np.asfortranarray(th), 1.0, np.ones_like(th), L, 1.0, None
    proc = nhp.DiscreteStandardHawkesProcess(nhp.DiscreteHomogeneousProcess(rng.uniform(0.02, 0.08, N), 1.0), imp,
                                             nhp.DenseWeightModel(np.asfortranarray(rng.uniform(0, 1, (N, N)) / N)), 1.0)
    ds, conv = nhp.convolve(proc, data, fetch=True)
    got = nhp.resample_parent_counts(proc, convolved=ds, seed=3, step=2)
    assert np.array_equal(got.sum(axis=1), data.sum(axis=1))
    want = orc.disc_resample_parents(data, conv, proc.baseline.λ, proc.weights.W, proc.impulses.θ, proc.dt, seed=3, step=2)
    assert np.array_equal(got, want)
    for env in ({"NHP_RP_CHK": "0"}, {"NHP_RP_COLM": "0"}, {"NHP_RP_XCD": "0"}, {"NHP_RP_TILE": "128,256,1024"}):
        os.environ.update(env)
        try:
            assert np.array_equal(nhp.resample_parent_counts(proc, convolved=ds, seed=3, step=2), want), env
        finally:
            for k in env:
                del os.environ[k]


def test_config4_adjacency_sweep_rows_at_full_size(nhp, orc, config4):
    """One sweep of the discrete adjacency matrix at config-4 scale (N = 512, B = 8, T = 1e5: 2.56e6 occupied bins, 513 launches
    of k_dadj_step over 512 spans of ~195 bins), explicit uniforms.  The literal restatement costs N²·T·N·B = 1e14 terms, so the
    definition is applied to single entries: rows 0 and 1 of a few columns from the intensity under the OLD matrix, and the
    LAST row from the intensity under the sweep's own earlier decisions of that column (the GEMM of the intensity test) --
    which is where an error of the λ carried through 511 steps would show."""
    c = config4
    N, B, L, T, data = c["N"], c["B"], c["L"], c["T"], c["data"]
    rng = np.random.default_rng(21)
    A0 = (rng.uniform(size=(N, N)) < 0.5).astype(np.float64)
    W = np.asfortranarray(c["W"] * 2.0)                                # (links that matter a little more)
    imp = c["proc"].impulses
    net = nhp.DiscreteNetworkHawkesProcess(nhp.DiscreteHomogeneousProcess(c["lam0"], 1.0), imp, nhp.DenseWeightModel(W), A0.copy(),
                                           nhp.BernoulliNetworkModel(0.3, N), 1.0)
    u = rng.uniform(size=(N, N))
    lam_old = nhp.intensity(net, c["ds"])                              # T x N under A0
    nhp.disc_resample_adjacency_matrix_(net, convolved=c["ds"], u=u)
    A1 = net.adjacency_matrix.copy()
    assert set(np.unique(A1)) <= {0.0, 1.0} and not np.array_equal(A1, A0)
    phi = orc.disc_basis(L, B, 1.0)
    prior = np.log(0.3) - np.log(0.7)
    logit_u = np.log(u / (1.0 - u))

    def xs(p, col):                                                    # x_t = W dt Σ_b Ŝ[t, p, b] θ[p, c, b]
        conv_p = orc.disc_convolve(data[p:p + 1], phi)[:, 0, :]        # T x B
        return W[p, col] * (conv_p @ c["th"][p, col, :])

    def decide(lam_col, a_old, x, s, p, col):
        l0 = lam_col - a_old * x
        occ = s > 0
        d = float(np.sum(s[occ] * (np.log(l0[occ] + x[occ]) - np.log(l0[occ]))) - x.sum() + prior)
        return d, (1.0 if logit_u[p, col] <= d else 0.0)

    checked = 0
    for col in (0, 77, 300, 511):
        s = data[col].astype(np.float64)
        lam_col = lam_old[:, col].copy()
        for p in (0, 1):
            x = xs(p, col)
            d, a_new = decide(lam_col, A0[p, col], x, s, p, col)
            if abs(d - logit_u[p, col]) > 1e-6:                        # (a draw this close to the odds is not a test of anything)
                assert A1[p, col] == a_new, (p, col, d, logit_u[p, col])
                checked += 1
            lam_col += (A1[p, col] - A0[p, col]) * x                   # what the sweep carried into the next row
    # the last row: λ under the sweep's decisions for rows < N-1 of the column and the old entry of row N-1
    state = A1.copy()
    state[N - 1, :] = A0[N - 1, :]
    net.adjacency_matrix = state
    lam_state = nhp.intensity(net, c["ds"])
    for col in (0, 77, 300, 511):
        x = xs(N - 1, col)
        d, a_new = decide(lam_state[:, col], A0[N - 1, col], x, data[col].astype(np.float64), N - 1, col)
        if abs(d - logit_u[N - 1, col]) > 1e-6:
            assert A1[N - 1, col] == a_new, (col, d, logit_u[N - 1, col])
            checked += 1
    assert checked >= 8


def test_config4_vb_step_invariants_at_full_size(nhp, config4):
    import copy
    from scipy.special import digamma
    c = config4
    N, B, T, data, ds = c["N"], c["B"], c["T"], c["data"], c["ds"]
    proc = copy.deepcopy(c["proc"])
    rng = np.random.default_rng(11)
    b, w, imp = proc.baseline, proc.weights, proc.impulses
    b.αv, b.βv = rng.uniform(0.5, 3, N), rng.uniform(0.5, 3, N)
    w.κv, w.νv = rng.uniform(0.5, 3, (N, N)), rng.uniform(0.5, 3, (N, N))
    imp.γv = rng.uniform(0.5, 3, (N, N, B))
    old = (b.αv.copy(), b.βv.copy(), w.κv.copy(), w.νv.copy(), imp.γv.copy())
    nhp.update_(proc, data, ds)
    counts = data.sum(axis=1).astype(float)
    # responsibilities sum to one over (baseline, parents × bases) for every (t, c): every event of node c is
    # attributed exactly once  (src/parents.jl:165-166)
    attributed = (b.αv - b.α0) + (imp.γv - imp.γ).sum(axis=(0, 2))
    assert np.allclose(attributed, counts, rtol=1e-11)
    assert np.allclose(w.κv, w.κ + (imp.γv - imp.γ).sum(axis=2), rtol=1e-12)          # src/weights.jl:70-91
    assert np.array_equal(w.νv, w.ν + counts[:, None] * np.ones((N, N)))
    assert np.allclose(b.βv, 1.0 / b.β0 + T * 1.0)                                      # SURVEY D14, literal
    # one cell against the definition: γv[p,c,b] = γ + E[p,c,b] Σ_t Ŝ[t,p,b] data[c,t] / Z[t,c], Z from the OLD
    # parameters -- with Z recovered from an intensity call whose effective weights are the VB expectations E
    αv, βv, κv, νv, γv = old
    E = np.exp(digamma(γv) - digamma(γv.sum(axis=2))[:, :, None] + (digamma(κv) - np.log(νv))[:, :, None])
    e0 = np.exp(digamma(αv) - np.log(βv))
    probe = copy.deepcopy(c["proc"])
    Wp = E.sum(axis=2)
    probe.baseline.λ, probe.weights.W = e0, np.asfortranarray(Wp)
    probe.impulses.θ = np.asfortranarray(E / Wp[:, :, None])
    Z = nhp.intensity(probe, ds)                                                          # T x N
    assert np.allclose(b.αv, b.α0 + e0 * (data.T / Z).sum(axis=0), rtol=1e-10)          # src/baselines.jl:444-452


# ---------------------------------------------------------------------------------------------- metric size, default dispatch
def test_default_dispatch_routes_agree_at_the_metric_size(nhp, orc):
    """`loglikelihood(process, data)` is `recursive=true` in the reference (src/continuous.jl:210-214,241-276): the O(M·N)
    recursion over the full history.  At N = 1024, M = 1e6 the oracle needs ~20 s per evaluation, so the check is by
    properties: (1) the recursion itself (k_recursive_waves, forced through the C ABI) and the library's default route
    (the 2^-60 truncated window) give the same log-likelihood and the same 2.1e6-entry gradient; (2) the recursion has no
    look-ahead, so a prefix of the events is a complete problem: its value and gradient equal the oracle's literal
    recursion on that prefix."""
    import ctypes as C
    from nhp_amd import _lib
    N, M = 1024, 1_000_000
    times, nodes, T = nhp.synthetic.s_metric_data(N, M, kbar=8.0)
    proc = nhp.synthetic.s_metric_process(N, M, T, "exponential", 1.0)
    ctx = nhp.default_context()
    P = N + 2 * N * N
    full = _lib.LL_RECURSIVE | _lib.LL_FULL_RECURSION

    def both(data, flags):
        ds = nhp.device_dataset(proc, data, ctx)
        model = proc.device_model(ctx)
        ll, g = C.c_double(), np.empty(P)
        _lib.check(_lib.lib().nhp_cont_loglik_grad(ctx.h, ds.h, model.h, flags, C.byref(ll), _lib.dptr(g), P), ctx.h)
        l2 = C.c_double()
        _lib.check(_lib.lib().nhp_cont_loglik(ctx.h, ds.h, model.h, flags, C.byref(l2)), ctx.h)
        assert rel(l2.value, ll.value) < 1e-13          # the log-likelihood pass alone = the one inside the gradient call
        return ll.value, g

    data = (times, nodes, T)
    ll_rec, g_rec = both(data, full)
    ll_def, g_def = both(data, _lib.LL_RECURSIVE)
    assert rel(ll_rec, ll_def) < 1e-12
    assert np.max(np.abs(g_rec - g_def) / np.maximum(1.0, np.abs(g_def))) < 1e-9
    assert np.isfinite(g_rec).all()

    m = 20_000                                                       # 2·m·N = 4e7 exponentials in the oracle
    sub = (times[:m].copy(), nodes[:m].copy(), float(times[m - 1]))
    om = orc.ContModel(proc.baseline.λ, proc.weights.W, theta=proc.impulses.θ, dt_max=1.0)
    ll_sub, g_sub = both(sub, full)
    wll, wg = orc.loglik_grad(om, sub[0], sub[1], sub[2], recursive=True)
    assert rel(ll_sub, orc.loglik_recursive(om, *sub)) < 1e-11
    assert rel(ll_sub, wll) < 1e-11
    assert np.max(np.abs(g_sub - wg) / np.maximum(1.0, np.abs(wg))) < 1e-9


# ------------------------------------------------------------ the other workloads bench.py times, at their real sizes
@pytest.mark.parametrize("workload", ["windowed_k64", "windowed_k512", "simulated_k32"])
def test_bench_workloads_at_full_size(nhp, orc, workload):
    """`bench.py` times the exponential log-likelihood at N = 1024, M = 1e6 for mean windows of 64 and 512 parents and on
    events drawn from the model itself (`simulated_k32`, burstier windows), each through its own kernel route (child slices
    up to the middle windows, 8- / 16-byte event records beyond).  The same data the bench builds, checked like the metric
    size: per-event intensities of three 1000-event slices against the oracle (src/continuous.jl:286-300), the
    log-likelihood = integral + Σ log λ of the returned intensities (src/continuous.jl:216-237), and the parent sampler's
    indices on one slice bit-equal to the oracle's (src/parents.jl:25-46)."""
    import bench
    w = bench.WORKLOADS[workload]
    N, M = 1024, 1_000_000
    times, nodes, T = nhp.synthetic.s_metric_data(N, M, kbar=w["kbar"])
    proc = nhp.synthetic.s_metric_process(N, M, T, w["kind"], 1.0)
    if w.get("simulated"):
        times, nodes, T = nhp.synthetic.simulated_data(proc, T, seed=0)
        M = len(times)
    data = (times, nodes, T)
    lam0, W = proc.baseline.λ, proc.weights.W
    om = orc.ContModel(lam0, W, theta=proc.impulses.θ, dt_max=1.0)
    lam = nhp.total_intensity(proc, data)
    starts = (0, M // 3 + 17, M - 1000)
    for i0 in starts:
        want = orc.total_intensity(om, times, nodes, i0, i0 + 1000)
        assert np.max(np.abs(lam[i0:i0 + 1000] - want) / want) < 1e-12
    integral = -(lam0 * T).sum() - (np.bincount(nodes - 1, minlength=N) @ W.sum(axis=1))
    ll = nhp.loglikelihood(proc, data, recursive=False)
    assert rel(ll, integral + np.log(lam).sum()) < 1e-6             # contractual tolerance (BASELINE.json north_star)
    assert rel(ll, integral + np.log(lam).sum()) < 1e-11
    # parent sampler: one slice with its look-back prefix (the oracle sees events [a, i1))
    seed, step = 5, 2
    p, pn = nhp.resample_parents(proc, data, seed=seed, step=step)
    u = orc.uniform_stream(seed, step, M)
    i0 = starts[1]
    i1 = i0 + 1000
    a = max(int(np.searchsorted(times, times[i0] - 1.0, side="right")) - 1, 0)
    wp, wpn = orc.resample_parents(om, times[a:i1], nodes[a:i1], u[a:i1], flags=orc.MATH_DET)
    wp = np.where(wp > 0, wp + a, 0)[i0 - a:]
    assert np.array_equal(p[i0:i1], wp)                             # bit-exact indices (north_star)
    assert np.array_equal(pn[i0:i1], wpn[i0 - a:])
