"""GPU parity: discrete Gibbs parent counts (nhp_disc_resample_parents) vs the oracle -- integer work,
bit-exact -- and the discrete mcmc! built on it."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def make(nhp, N, T, B, L, rate, seed, dt=1.0):
    rng = np.random.default_rng(seed)
    data = rng.poisson(rate, (N, T)).astype(np.int64)
    th = rng.dirichlet(np.ones(B), (N, N))
    th[:, :, -1] = 1.0 - th[:, :, :-1].sum(axis=2)
    while not np.all(th.sum(axis=2) == 1.0):                    # the constructor checks the sum exactly
        th = rng.dirichlet(np.ones(B), (N, N))
        th[:, :, -1] = 1.0 - th[:, :, :-1].sum(axis=2)
    proc = nhp.DiscreteStandardHawkesProcess(nhp.DiscreteHomogeneousProcess(rng.uniform(0.05, 0.3, N), dt),
                                             nhp.DiscreteGaussianImpulseResponse(th, L, dt),
                                             nhp.DenseWeightModel(rng.uniform(0.0, 0.5, (N, N)) / N), dt)
    return proc, data


@pytest.mark.parametrize("N,T,B,L,rate", [(3, 500, 4, 10, 0.3), (5, 3000, 3, 7, 2.5), (130, 700, 2, 5, 0.05),
                                          (17, 129, 5, 9, 0.8), (2, 64, 2, 3, 40.0)])
def test_counts_equal_the_oracle(nhp, orc, N, T, B, L, rate):
    proc, data = make(nhp, N, T, B, L, rate, seed=N + T)
    ds, conv = nhp.convolve(proc, data, fetch=True)
    got = nhp.resample_parent_counts(proc, convolved=ds, seed=11, step=4)
    want = orc.disc_resample_parents(data, conv, proc.baseline.λ, proc.weights.W, proc.impulses.θ, proc.dt, seed=11, step=4)
    assert got.shape == (N, 1 + N * B)
    assert np.array_equal(got.sum(axis=1), data.sum(axis=1))            # every event gets exactly one parent
    assert np.array_equal(got, want)
    again = nhp.resample_parent_counts(proc, convolved=ds, seed=11, step=4)
    assert np.array_equal(got, again)                                   # keyed draws: reproducible
    other = nhp.resample_parent_counts(proc, convolved=ds, seed=11, step=5)
    assert not np.array_equal(got, other)


def test_counts_follow_the_multinomial_mean(nhp):
    # E[counts[c, k]] = Σ_t n[c,t]·μ_k(t, c): compare the baseline column and the per-parent totals
    N, T, B, L = 4, 4000, 3, 8
    proc, data = make(nhp, N, T, B, L, 0.4, seed=9)
    ds = nhp.convolve(proc, data)
    lam = nhp.intensity(proc, ds)                                        # T x N
    exp0 = (data.T * (proc.baseline.λ * proc.dt)[None, :] / lam).sum(axis=0)
    acc = np.zeros((N, 1 + N * B))
    S = 60
    for s in range(S):
        acc += nhp.resample_parent_counts(proc, convolved=ds, seed=2, step=s)
    mean0 = acc[:, 0] / S
    assert np.all(np.abs(mean0 - exp0) < 5 * np.sqrt(exp0 / S) + 1.0)
    assert np.allclose(acc.sum(axis=1) / S, data.sum(axis=1))


def test_discrete_mcmc_recovers_the_baseline_when_there_is_no_excitation(nhp):
    rng = np.random.default_rng(0)
    N, T, B, L = 3, 20000, 3, 6
    true = np.array([0.2, 0.5, 1.0])
    data = rng.poisson(true[:, None] * np.ones((N, T))).astype(np.int64)
    th = np.full((N, N, B), 1.0 / 4); th[:, :, -1] = 0.5
    proc = nhp.DiscreteStandardHawkesProcess(nhp.DiscreteHomogeneousProcess(np.ones(N), 1.0),
                                             nhp.DiscreteGaussianImpulseResponse(th, L, 1.0),
                                             nhp.DenseWeightModel(np.full((N, N), 0.1)), 1.0)
    res = nhp.mcmc_(proc, data, nsteps=60, seed=1)
    assert res.steps == 60 and len(res.samples) == 60
    lam = np.mean([s[:N] for s in res.samples[20:]], axis=0)
    W = proc.weights.W
    # the Gamma(1, 1) prior on W keeps a little excitation alive, so the baseline sits slightly below the rate
    assert np.all(np.abs(lam - true) / true < 0.25) and np.all(np.diff(lam) > 0), lam
    assert np.all(W < 0.2)                                               # independent Poisson data: weights shrink
    assert np.allclose(proc.impulses.θ.sum(axis=2), 1.0)


def make_network(nhp, N, T, B, L, rate, seed):
    proc, data = make(nhp, N, T, B, L, rate, seed)
    rng = np.random.default_rng(seed + 50)
    A = (rng.uniform(size=(N, N)) < 0.5).astype(np.float64)
    net = nhp.DiscreteNetworkHawkesProcess(proc.baseline, proc.impulses, proc.weights, A,
                                           nhp.BernoulliNetworkModel(0.3, N), proc.dt)
    return net, data


@pytest.mark.parametrize("N,T,B,L,rate", [(3, 300, 3, 6, 0.5), (5, 257, 2, 4, 1.5), (4, 1000, 4, 9, 0.1)])
def test_adjacency_sweep_equals_the_literal_restatement(nhp, orc, N, T, B, L, rate):
    # explicit uniforms: the decisions of the incremental GPU sweep vs the reference's two full
    # conditional log-likelihoods per entry (src/discrete.jl:445-480)
    proc, data = make_network(nhp, N, T, B, L, rate, seed=3 * N + T)
    proc.weights.W = proc.weights.W * N * 1.5                       # links that matter
    ds, conv = nhp.convolve(proc, data, fetch=True)
    u = np.random.default_rng(7).uniform(size=(N, N))
    A0 = proc.adjacency_matrix.copy()
    want = orc.disc_resample_adjacency(data, conv, proc.baseline.λ, proc.weights.W, proc.impulses.θ, A0, 0.3, u, proc.dt)
    links = nhp.disc_resample_adjacency_matrix_(proc, convolved=ds, u=u)
    assert np.array_equal(proc.adjacency_matrix, want)
    assert links == want.sum()
    assert not np.array_equal(want, A0)                             # the sweep did something


@pytest.mark.parametrize("N,T,B,L,rate,spans,big", [(6, 1000, 8, 5, 0.2, 4, False), (24, 600, 4, 4, 0.05, 3, False),
                                                     (4, 700, 8, 6, 0.3, 0, True), (5, 900, 3, 4, 0.4, 5, False),
                                                     (5, 520, 8, 6, 0.3, 3, True)])
def test_adjacency_sweep_over_long_spans(nhp, orc, monkeypatch, N, T, B, L, rate, spans, big):
    """The step kernel's layouts (csrc/disc_gibbs.hip k_dadj_step): spans of up to 256 bins sorted by node (NHP_DADJ_SPANS cuts
    the time axis as a 256-CU device cuts T = 1e5: ~200 bins a span), B = 8 / 4 compiled in and any other B at run time, and
    a count past 255, which takes the unpacked entry arrays -- decisions against the oracle's literal restatement."""
    if spans:
        monkeypatch.setenv("NHP_DADJ_SPANS", str(spans))
    if spans == 5:
        monkeypatch.setenv("NHP_DADJ_VLDS", "0")                    # the row of V read from global memory (large N·B)
        monkeypatch.setenv("NHP_DADJ_THREADS", "256")
    proc, data = make_network(nhp, N, T, B, L, rate, seed=5 * N + T)
    if big:
        # (the node's own rate goes up with it: the reference takes log(pdf(Poisson(λ), s)), and a count of 300 under a rate
        #  of 0.3 underflows the pdf -- ll0 = ll1 = -Inf, Bernoulli(NaN) -- which is not a case to reproduce)
        data = data.copy()
        data[N // 2, T // 3] = 300
        lam0 = proc.baseline.λ.copy()
        lam0[N // 2] = 280.0
        proc.baseline.λ = lam0
    proc.weights.W = proc.weights.W * N * 1.5
    ds, conv = nhp.convolve(proc, data, fetch=True)
    u = np.random.default_rng(11).uniform(size=(N, N))
    A0 = proc.adjacency_matrix.copy()
    want = orc.disc_resample_adjacency(data, conv, proc.baseline.λ, proc.weights.W, proc.impulses.θ, A0, 0.3, u, proc.dt)
    links = nhp.disc_resample_adjacency_matrix_(proc, convolved=ds, u=u)
    assert np.array_equal(proc.adjacency_matrix, want)
    assert links == want.sum()
    assert not np.array_equal(want, A0)
    # a second sweep from the first one's result (flips of the last row carried, tables made anew)
    u2 = np.random.default_rng(12).uniform(size=(N, N))
    want2 = orc.disc_resample_adjacency(data, conv, proc.baseline.λ, proc.weights.W, proc.impulses.θ, want, 0.3, u2, proc.dt)
    nhp.disc_resample_adjacency_matrix_(proc, convolved=ds, u=u2)
    assert np.array_equal(proc.adjacency_matrix, want2)


def test_network_counts_respect_the_mask_and_mcmc_runs(nhp, orc):
    proc, data = make_network(nhp, 6, 2000, 3, 6, 0.4, seed=21)
    ds, conv = nhp.convolve(proc, data, fetch=True)
    got = nhp.resample_parent_counts(proc, convolved=ds, seed=5, step=0)
    want = orc.disc_resample_parents(data, conv, proc.baseline.λ, proc.weights.W, proc.impulses.θ, proc.dt,
                                     A=proc.adjacency_matrix, seed=5, step=0)
    assert np.array_equal(got, want)
    Mnm = nhp.disc_parent_counts(got, 6, 3)
    assert not Mnm[proc.adjacency_matrix == 0].any()                # no parents through absent links
    res = nhp.mcmc_(proc, data, nsteps=8, seed=2)
    assert res.steps == 8 and all(np.all(np.isfinite(s)) for s in res.samples)
    assert set(np.unique(proc.adjacency_matrix)) <= {0.0, 1.0}
    assert 0.0 < proc.network.ρ < 1.0


def test_empty_and_tiny_data(nhp, orc):
    # no events at all; a single occupied bin; one node
    N, T, B, L = 3, 50, 2, 4
    proc, _ = make_network(nhp, N, T, B, L, 0.5, seed=1)
    zero = np.zeros((N, T), dtype=np.int64)
    ds = nhp.convolve(proc, zero)
    assert not nhp.resample_parent_counts(proc, convolved=ds, seed=1, step=0).any()
    u = np.random.default_rng(0).uniform(size=(N, N))
    nhp.disc_resample_adjacency_matrix_(proc, convolved=ds, u=u)
    assert np.array_equal(proc.adjacency_matrix, (u <= 0.3).astype(float))          # no data: the prior decides
    one = zero.copy()
    one[1, 7] = 4
    ds1, conv1 = nhp.convolve(proc, one, fetch=True)
    got = nhp.resample_parent_counts(proc, convolved=ds1, seed=1, step=0)
    want = orc.disc_resample_parents(one, conv1, proc.baseline.λ, proc.weights.W, proc.impulses.θ, proc.dt,
                                     A=proc.adjacency_matrix, seed=1, step=0)
    assert np.array_equal(got, want) and got.sum() == 4
    p1, d1 = make(nhp, 1, 300, 2, 4, 0.8, seed=2)
    dsa, conva = nhp.convolve(p1, d1, fetch=True)
    assert np.array_equal(nhp.resample_parent_counts(p1, convolved=dsa, seed=2, step=3),
                          orc.disc_resample_parents(d1, conva, p1.baseline.λ, p1.weights.W, p1.impulses.θ, p1.dt, seed=2, step=3))


def test_device_draws_follow_their_conjugate_posteriors(nhp):
    # nhp_disc_gibbs_step: with parameters that make the counts (nearly) deterministic -- W = 0: every event is a
    # baseline event -- the draws must match Gamma(α0 + Σ data, 1/(β0 + T dt)), Gamma(κ, 1/(ν + Σ data[p])), Dirichlet(γ)
    rng = np.random.default_rng(5)
    N, T, B, L = 4, 4000, 3, 5
    data = rng.poisson(0.3, (N, T)).astype(np.int64)
    draws_l, draws_W, draws_t = [], [], []
    for s in range(200):
        th = np.full((N, N, B), 0.25); th[:, :, -1] = 0.5
        proc = nhp.DiscreteStandardHawkesProcess(nhp.DiscreteHomogeneousProcess(np.full(N, 0.3), 1.0),
                                                 nhp.DiscreteGaussianImpulseResponse(th, L, 1.0),
                                                 nhp.DenseWeightModel(np.zeros((N, N))), 1.0)
        nhp.resample_(proc, data, None, rng, seed=7, step=s)
        draws_l.append(proc.baseline.λ.copy()); draws_W.append(proc.weights.W.copy()); draws_t.append(proc.impulses.θ.copy())
    lam, W, th = np.mean(draws_l, axis=0), np.mean(draws_W, axis=0), np.mean(draws_t, axis=0)
    cnt = data.sum(axis=1)
    assert np.all(np.abs(lam - (1 + cnt) / (1 + T)) / ((1 + cnt) / (1 + T)) < 0.02)
    want_W = 1.0 / (1.0 + cnt)[:, None] * np.ones((N, N))                    # Gamma(1, 1/(1 + Mn[p])) mean
    assert np.all(np.abs(W - want_W) / want_W < 0.25)
    assert np.all(np.abs(th - 1.0 / B) < 0.06)                               # Dirichlet(1,1,1) mean
    assert np.allclose(np.sum(draws_t[0], axis=2), 1.0)
    # the host-draw path samples the same posteriors
    draws = []
    for s_ in range(100):
        th = np.full((N, N, B), 0.25); th[:, :, -1] = 0.5
        proc = nhp.DiscreteStandardHawkesProcess(nhp.DiscreteHomogeneousProcess(np.full(N, 0.3), 1.0),
                                                 nhp.DiscreteGaussianImpulseResponse(th, L, 1.0),
                                                 nhp.DenseWeightModel(np.zeros((N, N))), 1.0)
        nhp.resample_(proc, data, None, rng, seed=9, step=s_, device_draws=False)
        draws.append(proc.baseline.λ.copy())
    lam2 = np.mean(draws, axis=0)
    assert np.all(np.abs(lam2 - (1 + cnt) / (1 + T)) / ((1 + cnt) / (1 + T)) < 0.02)
