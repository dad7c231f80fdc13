"""GPU parity: HIP log-likelihood / intensity kernels (through the C ABI) vs the CPU oracle.

Tolerance: BASELINE.json asks for 1e-6 relative on the fp64 log-likelihood; the kernels
differ from the oracle only in summation order and 1-ulp exp/log, so the tests hold them to
1e-11 and state the contractual 1e-6 where it is asserted.
"""
import math
import os

import numpy as np
import pytest

from helpers import random_case, rel

pytestmark = pytest.mark.gpu
TOL = 1e-11           # observed ~1e-14; contract (BASELINE.json north_star): 1e-6


def both(nhp, orc, c, recursive):
    got = nhp.loglikelihood(c["proc"], c["data"], recursive=recursive)
    want = orc.loglik(c["om"], c["times"], c["nodes"], c["T"], recursive=recursive)
    return got, want


@pytest.mark.parametrize("kind", ["exponential", "logitnormal"])
@pytest.mark.parametrize("network", [False, True])
@pytest.mark.parametrize("lgcp", [False, True])
def test_windowed_matches_oracle(nhp, orc, kind, network, lgcp):
    c = random_case(8, 3000, 200.0, kind, 1.0, network=network, lgcp=lgcp, seed=1, nhp=nhp, orc=orc)
    got, want = both(nhp, orc, c, recursive=False)
    assert rel(got, want) < TOL


@pytest.mark.parametrize("network", [False, True])
@pytest.mark.parametrize("lgcp", [False, True])
def test_recursive_matches_oracle(nhp, orc, network, lgcp):
    c = random_case(8, 3000, 200.0, "exponential", 1.0, network=network, lgcp=lgcp, seed=2, nhp=nhp, orc=orc)
    got, want = both(nhp, orc, c, recursive=True)
    assert rel(got, want) < TOL


def test_readme_example_c1(nhp, orc):
    # BASELINE configs[0]: README.md:25-38, N=2, Δtmax=Inf so both formulations agree (D8)
    proc, data = nhp.synthetic.readme_case(seed=0)
    om = orc.ContModel(proc.baseline.λ, proc.weights.W, theta=proc.impulses.θ, dt_max=np.inf)
    a = nhp.loglikelihood(proc, data)                       # default recursive=true
    b = nhp.loglikelihood(proc, data, recursive=False)
    assert math.isfinite(a)
    assert rel(a, orc.loglik_recursive(om, *data)) < TOL
    assert rel(b, orc.loglik_windowed(om, *data)) < TOL
    assert rel(a, b) < 1e-10


@pytest.mark.parametrize("group", [1, 2, 4, 8, 16, 32, 64])
def test_every_group_width(nhp, orc, group, monkeypatch):
    # the lanes-per-child width is chosen from the mean window; force each instantiation
    monkeypatch.setenv("NHP_GROUP", str(group))
    for kind in ("exponential", "logitnormal"):
        c = random_case(5, 2000, 60.0, kind, 1.0, seed=group, nhp=nhp, orc=orc)
        got, want = both(nhp, orc, c, recursive=False)
        assert rel(got, want) < TOL


def test_edge_cases(nhp, orc):
    lam0, W, th = np.array([0.7, 1.3, 0.9]), np.full((3, 3), 0.2), np.full((3, 3), 1.5)

    def run(times, nodes, T, dtm, recursive):
        proc = nhp.ContinuousStandardHawkesProcess(nhp.HomogeneousProcess(lam0),
                                                   nhp.ExponentialImpulseResponse(th, 1.0, 1.0, dtm),
                                                   nhp.DenseWeightModel(W))
        om = orc.ContModel(lam0, W, theta=th, dt_max=dtm)
        t, n = np.asarray(times, float), np.asarray(nodes, np.int64)
        got = nhp.loglikelihood(proc, (t, n, T), recursive=recursive)
        want = orc.loglik(om, t, n, T, recursive=recursive)
        assert rel(got, want) < TOL, (times, recursive, got, want)

    for rec in (False, True):
        run([], [], 5.0, 1.0, rec)                              # empty data: ll = -Σλ0·T
        run([2.0], [3], 5.0, 1.0, rec)                          # single event
        run([1.0, 1.0, 1.0, 2.0], [1, 2, 1, 2], 5.0, 1.0, rec)  # ties: Δt = 0 parents are included
        run([0.0, 0.0, 0.5, 0.9], [1, 2, 2, 1], 5.0, 1.0, rec)  # events at t = 0.0 (D9 on the recursive path)
        run([0.5, 1.5, 2.5, 3.5], [1, 1, 1, 1], 5.0, 1.0, rec)  # Δt == Δtmax exactly: excluded (strict); nodes 2,3 empty
        run(np.linspace(0.1, 4.9, 700), np.ones(700, int), 5.0, 1e-9, rec)   # empty windows, one crowded node
        run(np.sort(np.random.default_rng(0).uniform(0, 5, 900)), np.random.default_rng(1).integers(1, 4, 900),
            5.0, np.inf, rec)                                   # Δtmax = Inf: full history


def test_windowed_inf_equals_recursive_on_gpu(nhp):
    c = random_case(16, 4000, 50.0, "exponential", np.inf, seed=5, nhp=nhp)
    a = nhp.loglikelihood(c["proc"], c["data"], recursive=True)
    b = nhp.loglikelihood(c["proc"], c["data"], recursive=False)
    assert rel(a, b) < 1e-10


def test_error_conventions(nhp):
    c = random_case(4, 100, 10.0, "exponential", 1.0, seed=0, nhp=nhp)
    t, n, T = c["data"]
    with pytest.raises(nhp.DomainError):
        nhp.loglikelihood(c["proc"], (t, n, -1.0))                 # duration < 0  (src/baselines.jl:100)
    with pytest.raises(nhp.DomainError):
        nhp.loglikelihood(c["proc"], (t, np.where(n == 1, 9, n), T))   # node outside 1..N
    with pytest.raises(nhp.DomainError):
        nhp.loglikelihood(c["proc"], (t - 5.0, n, T))              # negative time (src/baselines.jl:116)
    with pytest.raises(nhp.NhpError):
        nhp.loglikelihood(c["proc"], (t[::-1].copy(), n, T))       # unsorted
    c["proc"].weights.W = np.ones((3, 3))
    with pytest.raises(ValueError):
        nhp.loglikelihood(c["proc"], c["data"])                    # parameter shape mismatch


def test_event_intensity_matches_oracle(nhp, orc):
    c = random_case(8, 2500, 100.0, "logitnormal", 2.0, network=True, seed=8, nhp=nhp, orc=orc)
    got = nhp.total_intensity(c["proc"], c["data"])
    want = orc.total_intensity(c["om"], c["times"], c["nodes"])
    assert np.max(np.abs(got - want) / want) < 1e-12


def test_c2_size_against_oracle(nhp, orc):
    # BASELINE configs[1]: N=128, ~1e5 events
    N, M = 128, 100_000
    times, nodes, T = nhp.synthetic.s_metric_data(N, M, kbar=16.0)
    proc = nhp.synthetic.s_metric_process(N, M, T, "exponential", 1.0)
    om = orc.ContModel(proc.baseline.λ, proc.weights.W, theta=proc.impulses.θ, dt_max=1.0)
    for rec in (False, True):
        got = nhp.loglikelihood(proc, (times, nodes, T), recursive=rec)
        want = orc.loglik(om, times, nodes, T, recursive=rec, flags=orc.FAST_INTEGRAL)
        assert rel(got, want) < 1e-6        # contractual tolerance
        assert rel(got, want) < TOL


def test_metric_size_properties(nhp, orc):
    # BASELINE metric config: N=1024, M=1e6.  The oracle cannot run this in seconds, so check
    # size-independent properties: (1) W = 0 gives the closed form -Σλ0·T + Σ log λ0[c_i];
    # (2) per-event intensities of random slices of events agree with the oracle;
    # (3) windowed and recursive agree when θ·Δtmax is so large that truncation is below 1 ulp.
    N, M = 1024, 1_000_000
    times, nodes, T = nhp.synthetic.s_metric_data(N, M, kbar=8.0)
    data = (times, nodes, T)
    proc = nhp.synthetic.s_metric_process(N, M, T, "exponential", 1.0)
    lam0 = proc.baseline.λ
    W_saved = proc.weights.W.copy()
    proc.weights.W = np.zeros((N, N))
    closed = -(lam0 * T).sum() + np.log(lam0[nodes - 1]).sum()
    assert rel(nhp.loglikelihood(proc, data, recursive=False), closed) < 1e-12
    assert rel(nhp.loglikelihood(proc, data, recursive=True), closed) < 1e-12
    proc.weights.W = W_saved
    lam = nhp.total_intensity(proc, data)
    om = orc.ContModel(lam0, W_saved, theta=proc.impulses.θ, dt_max=1.0)
    for i0 in (0, 123_456, 999_000):
        want = orc.total_intensity(om, times, nodes, i0, i0 + 1000)
        assert np.max(np.abs(lam[i0:i0 + 1000] - want) / want) < 1e-12
    integral = -(lam0 * T).sum() - (np.bincount(nodes - 1, minlength=N) @ W_saved.sum(axis=1))
    assert rel(nhp.loglikelihood(proc, data, recursive=False), integral + np.log(lam).sum()) < 1e-12
    proc.impulses.θ = proc.impulses.θ * 12.0          # θ·Δtmax >= 12·... make tails negligible: θ in [12, 60]
    proc.impulses.θ += 30.0                           # θ >= 42 -> e^{-42} < 1e-18 relative
    a = nhp.loglikelihood(proc, data, recursive=False)
    b = nhp.loglikelihood(proc, data, recursive=True)
    assert rel(a, b) < 1e-9


def test_batch_of_models(nhp, orc):
    # nhp_cont_loglik_batch: several parameter sets on one dataset, one synchronisation
    import ctypes as C
    from nhp_amd import _lib
    ctx = nhp.default_context()
    cases = [random_case(6, 1500, 90.0, "exponential", 1.0, seed=s, nhp=nhp, orc=orc) for s in (50, 51, 52)]
    data = cases[0]["data"]
    ds = nhp.device_dataset(cases[0]["proc"], data, ctx)
    models = [c["proc"].device_model(ctx) for c in cases]
    arr = (C.c_void_p * 3)(*[m.h for m in models])
    out = np.empty(3)
    _lib.check(_lib.lib().nhp_cont_loglik_batch(ctx.h, ds.h, arr, 3, 0, _lib.dptr(out)), ctx.h)
    for c, got in zip(cases, out):
        assert rel(got, orc.loglik_windowed(c["om"], data[0], data[1], data[2])) < TOL


@pytest.mark.parametrize("network", [False, True])
@pytest.mark.parametrize("lgcp", [False, True])
def test_recursive_through_the_truncated_window_matches_the_recursion(nhp, orc, network, lgcp):
    # fast-decaying impulses: the full-history sum is evaluated through a window beyond which the tail is below
    # 2^-60 of every λ_i (csrc/cont_recursive.hip); the oracle runs the literal O(M·N) recursion.  Events at exactly
    # t = 0 (skipped by the recursion, D9) and the unmasked integral (D7) are part of the comparison.
    N, M, T = 64, 4000, 400.0
    rng = np.random.default_rng(77)
    times = np.sort(rng.uniform(0.0, T, M))
    times[:3] = 0.0
    nodes = rng.integers(1, N + 1, M).astype(np.int64)
    th = rng.uniform(20.0, 40.0, (N, N))
    W = rng.uniform(0.0, 1.0, (N, N)) / N
    A = (rng.uniform(size=(N, N)) < 0.5).astype(np.float64) if network else None
    if lgcp:
        gx = np.linspace(0.0, T, 9)
        lam0 = np.exp(rng.normal(0.0, 0.3, (N, 9)))
        base = nhp.LogGaussianCoxProcess(gx, list(lam0))
    else:
        gx, lam0 = None, rng.uniform(0.5, 1.5, N)
        base = nhp.HomogeneousProcess(lam0)
    imp, w = nhp.ExponentialImpulseResponse(th, 1.0, 1.0, 0.05), nhp.DenseWeightModel(W)
    proc = (nhp.ContinuousNetworkHawkesProcess(base, imp, w, A, nhp.BernoulliNetworkModel(0.5, N)) if network
            else nhp.ContinuousStandardHawkesProcess(base, imp, w))
    om = orc.ContModel(lam0, W, theta=th, dt_max=0.05, A=A, grid_x=gx)
    got = nhp.loglikelihood(proc, (times, nodes, T), recursive=True)
    want = orc.loglik_recursive(om, times, nodes, T)
    assert rel(got, want) < 1e-12
    if not network:                                       # (mle! is defined for the standard process)
        ll, g = nhp.loglikelihood_gradient(proc, (times, nodes, T), recursive=True)
        wll, wg = orc.loglik_grad(om, times, nodes, T, recursive=True)
        assert rel(ll, wll) < 1e-12
        assert np.max(np.abs(g - wg) / np.maximum(1.0, np.abs(wg))) < 1e-10
    # and it is NOT the windowed (Δtmax = 0.05) value: the recursion ignores Δtmax (D8)
    assert rel(nhp.loglikelihood(proc, (times, nodes, T), recursive=False), want) > 1e-6


@pytest.mark.parametrize("kind,network,lgcp", [("exponential", False, False), ("logitnormal", True, False), ("exponential", False, True)])
def test_fused_batches_of_models(nhp, orc, kind, network, lgcp):
    # nhp_cont_loglik_batch takes compatible models four (or two) at a time through one pass over the data
    import ctypes as C
    from nhp_amd import _lib
    ctx = nhp.default_context()
    cases = [random_case(9, 2500, 120.0, kind, 1.0, network=network, lgcp=lgcp, seed=s, nhp=nhp, orc=orc) for s in range(60, 67)]
    data = cases[0]["data"]
    ds = nhp.device_dataset(cases[0]["proc"], data, ctx)
    models = [c["proc"].device_model(ctx) for c in cases]
    arr = (C.c_void_p * 7)(*[m.h for m in models])
    out = np.empty(7)
    _lib.check(_lib.lib().nhp_cont_loglik_batch(ctx.h, ds.h, arr, 7, 0, _lib.dptr(out)), ctx.h)      # 4 + 2 + 1
    for c, got in zip(cases, out):
        assert rel(got, orc.loglik_windowed(c["om"], data[0], data[1], data[2])) < TOL



def test_batch_lanes_follow_parameter_updates(nhp, orc):
    # the batch alternates its launches between the context's two internal streams: many launches per call, calls
    # back to back, and a parameter upload between two calls (it must be seen by both streams)
    import ctypes as C
    from nhp_amd import _lib
    ctx = nhp.default_context()
    cases = [random_case(7, 3000, 150.0, "exponential", 1.0, seed=s, nhp=nhp, orc=orc) for s in range(80, 103)]
    data = cases[0]["data"]
    ds = nhp.device_dataset(cases[0]["proc"], data, ctx)
    models = [c["proc"].device_model(ctx) for c in cases]
    n = len(cases)
    arr = (C.c_void_p * n)(*[m.h for m in models])
    out = np.empty(n)
    want = np.array([orc.loglik_windowed(c["om"], data[0], data[1], data[2]) for c in cases])
    for _ in range(3):
        out[:] = 0.0
        _lib.check(_lib.lib().nhp_cont_loglik_batch(ctx.h, ds.h, arr, n, 0, _lib.dptr(out)), ctx.h)
        assert np.max(np.abs(out - want) / np.abs(want)) < TOL
    # models 1, 2, ... take model 0's parameters: every entry must now be model 0's value
    x0 = cases[0]["proc"].params()
    for m in models[1:]:
        m.set_params(x0)
    _lib.check(_lib.lib().nhp_cont_loglik_batch(ctx.h, ds.h, arr, n, 0, _lib.dptr(out)), ctx.h)
    assert np.max(np.abs(out - want[0]) / abs(want[0])) < TOL
    # and a plain evaluation right after a batch uses the main stream again
    assert rel(nhp.loglikelihood(cases[3]["proc"], data, recursive=False), want[3]) < TOL


@pytest.mark.parametrize("kind,network,lgcp", [("exponential", False, False), ("logitnormal", True, False), ("exponential", True, True)])
@pytest.mark.parametrize("N,M,T", [(9, 2500, 120.0), (70, 9000, 40.0)])
def test_batches_of_eight_share_one_pass(nhp, orc, kind, network, lgcp, N, M, T, monkeypatch):
    """nhp_cont_loglik_batch with 19 compatible models: 8 + 8 + 2 + 1 through k_windowed_batch (one lane per (child, model), the
    S columns in LDS, windows staged once per child) -- every value equals the oracle's windowed log-likelihood of its own
    model, also with ragged windows (dense second case: windows of ~50 parents, several record chunks per child), and the
    older 4-model kernel (NHP_BATCH_KERNEL=0) gives the same."""
    import ctypes as C
    from nhp_amd import _lib
    ctx = nhp.default_context()
    cases = [random_case(N, M, T, kind, 1.0, network=network, lgcp=lgcp, seed=s, nhp=nhp, orc=orc) for s in range(200, 219)]
    data = cases[0]["data"]
    ds = nhp.device_dataset(cases[0]["proc"], data, ctx)
    models = [c["proc"].device_model(ctx) for c in cases]
    n = len(cases)
    arr = (C.c_void_p * n)(*[m.h for m in models])
    out = np.empty(n)
    _lib.check(_lib.lib().nhp_cont_loglik_batch(ctx.h, ds.h, arr, n, 0, _lib.dptr(out)), ctx.h)
    want = np.array([orc.loglik_windowed(c["om"], data[0], data[1], data[2]) for c in cases])
    assert np.max(np.abs(out - want) / np.abs(want)) < TOL
    single = np.array([nhp.loglikelihood(c["proc"], data, recursive=False) for c in cases])
    assert np.max(np.abs(out - single) / np.abs(single)) < 1e-13
    # exponential models with flat baselines go four (or two) at a time through one pass over the child slices
    # (k_slices_batch); NHP_BATCH_SLICES=0 keeps them on k_windowed_batch: same values, for every workgroup shape
    monkeypatch.setenv("NHP_BATCH_SLICES", "0")
    out0 = np.empty(n)
    _lib.check(_lib.lib().nhp_cont_loglik_batch(ctx.h, ds.h, arr, n, 0, _lib.dptr(out0)), ctx.h)
    assert np.max(np.abs(out0 - want) / np.abs(want)) < TOL
    monkeypatch.delenv("NHP_BATCH_SLICES", raising=False)
    for cfg in ("256,2", "256,4", "512,2", "512,4", "1024,2", "1024,4"):
        monkeypatch.setenv("NHP_SETS_CFG", cfg)
        out1 = np.empty(n)
        _lib.check(_lib.lib().nhp_cont_loglik_batch(ctx.h, ds.h, arr, n, 0, _lib.dptr(out1)), ctx.h)
        assert np.max(np.abs(out1 - want) / np.abs(want)) < TOL, cfg
    monkeypatch.delenv("NHP_SETS_CFG", raising=False)


def test_batch_handles_empty_windows_and_degenerate_intensities(nhp, orc):
    """Edge cases of the (child, model) kernel: children without any parent (Δtmax tiny), a node without events, fewer
    children than one wave group, and a model whose λ is 0 somewhere (log-likelihood -Inf, as the per-child logs give)."""
    import ctypes as C
    from nhp_amd import _lib
    ctx = nhp.default_context()
    rng = np.random.default_rng(5)
    N, M, T = 5, 40, 30.0
    times = np.sort(rng.uniform(0, T, M))
    nodes = rng.integers(1, N, M).astype(np.int64)           # node N never fires
    procs, oms = [], []
    for k in range(8):
        lam0, W, th = rng.uniform(0.2, 1.0, N), rng.uniform(0, 0.3, (N, N)), rng.uniform(1, 4, (N, N))
        if k == 3:
            lam0[:] = 0.0
            W[:] = 0.0                                       # λ = 0 at every event
        procs.append(nhp.ContinuousStandardHawkesProcess(nhp.HomogeneousProcess(lam0), nhp.ExponentialImpulseResponse(th, 1.0, 1.0, 1e-3),
                                                         nhp.DenseWeightModel(W)))
        oms.append(orc.ContModel(lam0, W, theta=th, dt_max=1e-3))
    ds = nhp.device_dataset(procs[0], (times, nodes, T), ctx)
    models = [p.device_model(ctx) for p in procs]
    arr = (C.c_void_p * 8)(*[m.h for m in models])
    out = np.empty(8)
    _lib.check(_lib.lib().nhp_cont_loglik_batch(ctx.h, ds.h, arr, 8, 0, _lib.dptr(out)), ctx.h)
    for k in range(8):
        if k == 3:
            assert out[k] == -np.inf
        else:
            assert rel(out[k], orc.loglik_windowed(oms[k], times, nodes, T)) < TOL


@pytest.mark.parametrize("network,lgcp", [(False, False), (True, True)])
def test_pair_list_and_packed_records_agree_with_the_exact_records(nhp, orc, network, lgcp, monkeypatch):
    """Short-window exponential evaluations run through the cached pair list (k_windowed_pairs: Δt as a 48-bit fraction of
    Δtmax) and, failing that, through the 8-byte event records (k_windowed<.., PACK>); NHP_PLIST=0 / NHP_EV8=0 switch them
    off one after the other.  All three equal the oracle on the ORIGINAL times, for every (lanes per child, children in
    flight, workgroup size) the pair kernel is built with, with ties (Δt = 0) and ragged windows in the data."""
    c = random_case(12, 6000, 500.0, "exponential", 1.0, network=network, lgcp=lgcp, seed=11, nhp=nhp, orc=orc)
    t = c["times"].copy()
    t[1000:1040:2] = t[1001:1041:2]                                  # ties: parents at Δt = 0
    t[3000:3120] = np.sort(np.random.default_rng(3).uniform(t[3000], t[3000] + 0.7, 120))   # a burst: windows of ~100 parents
    t = np.sort(t)
    data = (t, c["nodes"], c["T"])
    want = orc.loglik(c["om"], t, c["nodes"], c["T"], recursive=False)
    got = {}
    for name, env in (("slices", {}), ("pairs", {"NHP_SLICES": "0"}), ("ev8", {"NHP_PLIST": "0"}),
                      ("exact", {"NHP_PLIST": "0", "NHP_EV8": "0"})):
        for k in ("NHP_PLIST", "NHP_EV8", "NHP_SLICES"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        nhp.invalidate_device_datasets()
        got[name] = nhp.loglikelihood(c["proc"], data, recursive=False)
        assert rel(got[name], want) < TOL, name
    assert rel(got["pairs"], got["exact"]) < 1e-13 and rel(got["ev8"], got["exact"]) < 1e-13
    assert rel(got["slices"], got["exact"]) < 1e-12                  # one lane per child: 6-byte records (cont_slices.hip)
    for k in ("NHP_PLIST", "NHP_EV8", "NHP_SLICES"):
        monkeypatch.delenv(k, raising=False)
    nhp.invalidate_device_datasets()
    for cfg in ("64,2", "64,4", "128,2", "128,4", "256,2", "256,4", "512,2", "512,4", "1024,2", "1024,4"):   # (waves per item, rows per request)
        monkeypatch.setenv("NHP_SLICES_CFG", cfg)
        assert rel(nhp.loglikelihood(c["proc"], data, recursive=False), want) < TOL, cfg
    monkeypatch.delenv("NHP_SLICES_CFG", raising=False)
    monkeypatch.setenv("NHP_SLICES", "0")
    nhp.invalidate_device_datasets()
    for cfg in ("1,1,256", "2,4,256", "4,2,512", "8,4,512", "4,1,1024", "8,2,1024", "2,2,512"):
        monkeypatch.setenv("NHP_PAIRS_CFG", cfg)
        assert rel(nhp.loglikelihood(c["proc"], data, recursive=False), want) < TOL, cfg
    monkeypatch.delenv("NHP_PAIRS_CFG", raising=False)
    monkeypatch.delenv("NHP_SLICES", raising=False)
    # logit-normal impulses: the pair kernel reads {logit(x), 1/(x(1-x))} made once per dataset -- the same operations, so
    # the same bits per term as the kernel that evaluates the whole pdf
    cl = random_case(12, 6000, 500.0, "logitnormal", 1.0, network=network, lgcp=lgcp, seed=11, nhp=nhp, orc=orc)
    wantl = orc.loglik(cl["om"], t, cl["nodes"], cl["T"], recursive=False)
    # (round 3: first in line is the slices' twin, k_windowed_slices_ln -- one lane per child over the planes of logit(x) and
    #  1/(x(1-x)) the parent sampler keeps; NHP_SLICES_LN=0 gives the pair kernel, NHP_PLIST=0 the whole pdf per record)
    gl = {}
    for name, env in (("slices", {}), ("pairs", {"NHP_SLICES_LN": "0"}), ("records", {"NHP_PLIST": "0"})):
        for k in ("NHP_PLIST", "NHP_SLICES_LN"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        nhp.invalidate_device_datasets()
        gl[name] = nhp.loglikelihood(cl["proc"], (t, cl["nodes"], cl["T"]), recursive=False)
        assert rel(gl[name], wantl) < TOL, name
    assert rel(gl["pairs"], gl["records"]) < 1e-13 and rel(gl["slices"], gl["records"]) < 1e-13
    monkeypatch.delenv("NHP_PLIST", raising=False)
    monkeypatch.delenv("NHP_SLICES_LN", raising=False)
    for cfg in ("64,2", "64,4", "256,2", "256,4", "512,2", "512,4"):
        monkeypatch.setenv("NHP_SLICES_LN_CFG", cfg)
        assert rel(nhp.loglikelihood(cl["proc"], (t, cl["nodes"], cl["T"]), recursive=False), wantl) < TOL, cfg
    monkeypatch.delenv("NHP_SLICES_LN_CFG", raising=False)
    monkeypatch.setenv("NHP_SLICES_LN", "0")
    for cfg in ("2,2,256", "4,1,512", "8,2,1024"):
        monkeypatch.setenv("NHP_PAIRS_CFG", cfg)
        assert rel(nhp.loglikelihood(cl["proc"], (t, cl["nodes"], cl["T"]), recursive=False), wantl) < TOL, cfg
    monkeypatch.delenv("NHP_PAIRS_CFG", raising=False)
    monkeypatch.delenv("NHP_SLICES_LN", raising=False)


def test_child_slices_at_a_thousand_nodes(nhp, orc, monkeypatch):
    """N = 1024 is where the 6-byte records of the child slices are tightest (11 bits of node, 37 bits of delay: 3.6e-12 of
    Δtmax): the log-likelihood through them, through the 8-byte pair list and through the exact records against the oracle on
    the original times, standard and network model, with a column that has no child and an item with a single one."""
    for network in (False, True):
        c = random_case(1024, 40000, 5000.0, "exponential", 1.0, network=network, seed=77, nhp=nhp, orc=orc)
        n = c["nodes"].copy()
        n[n == 5] = 6
        n[n == 9] = 10
        n[123] = 9
        data = (c["times"], n, c["T"])
        want = orc.loglik(c["om"], c["times"], n, c["T"], recursive=False)
        for name, env in (("slices", {}), ("pairs", {"NHP_SLICES": "0"}), ("exact", {"NHP_PLIST": "0", "NHP_EV8": "0"})):
            for k in ("NHP_PLIST", "NHP_EV8", "NHP_SLICES"):
                monkeypatch.delenv(k, raising=False)
            for k, v in env.items():
                monkeypatch.setenv(k, v)
            nhp.invalidate_device_datasets()
            assert rel(nhp.loglikelihood(c["proc"], data, recursive=False), want) < TOL, (name, network)
    for k in ("NHP_PLIST", "NHP_EV8", "NHP_SLICES"):
        monkeypatch.delenv(k, raising=False)
    nhp.invalidate_device_datasets()


def test_derived_layouts_agree_on_awkward_data(nhp, monkeypatch):
    """The child slices, the pair list and the 8-byte records are derived from the data once per dataset: time origins far from zero,
    ties, a node without events, one node with all of them, tiny and huge Δtmax, N = 1 -- each evaluated through the pair
    list, the 8-byte records and the exact 16-byte records (the three agree to 1e-12; the exact path is the one the oracle tests
    pin), for exponential and logit-normal impulses."""
    rng = np.random.default_rng(123)
    cases = []
    for (N, M, t_lo, t_hi, dtm) in ((5, 3000, 1e6, 1e6 + 400.0, 1.0), (7, 2500, 0.25, 350.0, 0.7), (1, 800, 0.0, 90.0, 0.5),
                                    (9, 4000, 0.0, 2.0, 1e-3), (6, 1500, 10.0, 60.0, 3.0), (12, 5000, 0.0, 5e5, 400.0)):
        t = np.sort(rng.uniform(t_lo, t_hi, M))
        t[10:40:3] = t[11:41:3]                                     # ties
        t = np.sort(t)
        n = rng.integers(1, N + 1, M).astype(np.int64)
        if N > 2:
            n[n == 2] = 1                                           # node 2 has no events
        cases.append((N, t, n, float(t_hi + 1.0), dtm))
    for N, t, n, T, dtm in cases:
        lam0 = rng.uniform(0.5, 1.5, N)
        W = rng.uniform(0.0, 1.0, (N, N)) / max(N, 2) * 2.0
        for kind in ("exponential", "logitnormal"):
            if kind == "exponential":
                imp = nhp.ExponentialImpulseResponse(rng.uniform(1.0, 5.0, (N, N)) / dtm, 1.0, 1.0, dtm)
            else:
                imp = nhp.LogitNormalImpulseResponse(rng.normal(0.0, 1.0, (N, N)), rng.uniform(0.5, 2.0, (N, N)), dtm)
            proc = nhp.ContinuousStandardHawkesProcess(nhp.HomogeneousProcess(lam0), imp, nhp.DenseWeightModel(W))
            got = {}
            for name, env in (("slices", {}), ("pairs", {"NHP_SLICES": "0"}), ("ev8", {"NHP_PLIST": "0"}),
                              ("exact", {"NHP_PLIST": "0", "NHP_EV8": "0"})):
                for k in ("NHP_PLIST", "NHP_EV8", "NHP_SLICES"):
                    monkeypatch.delenv(k, raising=False)
                for k, v in env.items():
                    monkeypatch.setenv(k, v)
                nhp.invalidate_device_datasets()
                got[name] = nhp.loglikelihood(proc, (t, n, T), recursive=False)
            assert np.isfinite(got["exact"])
            assert rel(got["slices"], got["exact"]) < 1e-12, (N, kind, dtm, got)
            assert rel(got["pairs"], got["exact"]) < 1e-12, (N, kind, dtm, got)
            assert rel(got["ev8"], got["exact"]) < 1e-12, (N, kind, dtm, got)
    for k in ("NHP_PLIST", "NHP_EV8", "NHP_SLICES"):
        monkeypatch.delenv(k, raising=False)


@pytest.mark.parametrize("N,network,lgcp", [(3, False, False), (70, True, False), (200, False, True), (300, False, False),
                                           (600, True, False), (1100, False, False), (2100, False, False)])
def test_full_recursion_every_part_shape(nhp, orc, N, network, lgcp):
    # NHP_LL_FULL_RECURSION: the O(M·N) recursion itself (k_recursive_waves: one wave per (column, part of the parent
    # nodes)), never the truncated window.  The node counts walk through every (parents per lane, parts) shape the launcher
    # picks -- 1x1, 1x2, 1x4, 2x4, 4x4 (ragged last part), 4x8, 4x16 -- incl. parts with no node, columns with no child,
    # events at exactly t = 0 (D9) and the unmasked integral of the network twin (D7).
    import ctypes as C
    from nhp_amd import _lib
    M = 2500
    c = random_case(N, M, 300.0, "exponential", 1.0, network=network, lgcp=lgcp, seed=100 + N, nhp=nhp, orc=orc)
    c["times"][:2] = 0.0
    data = (c["times"], c["nodes"], c["T"])
    ctx = nhp.default_context()
    ds = nhp.device_dataset(c["proc"], data, ctx)
    model = c["proc"].device_model(ctx)
    ll = C.c_double()
    _lib.check(_lib.lib().nhp_cont_loglik(ctx.h, ds.h, model.h, _lib.LL_RECURSIVE | _lib.LL_FULL_RECURSION, C.byref(ll)), ctx.h)
    assert rel(ll.value, orc.loglik_recursive(c["om"], *data)) < TOL
    # the default route (truncated window where it applies) gives the same number
    assert rel(nhp.loglikelihood(c["proc"], data, recursive=True), ll.value) < TOL
    # ... and the gradient of the same recursion (k_recursive_waves leaves 1/λ of every child, k_grad_recursive_waves
    # accumulates the sums per (column, part)) against the oracle's; P = [λ0 | grid intensities; θ; W]
    nb = N * 17 if lgcp else N
    P = nb + 2 * N * N
    g = np.empty(P)
    _lib.check(_lib.lib().nhp_cont_loglik_grad(ctx.h, ds.h, model.h, _lib.LL_RECURSIVE | _lib.LL_FULL_RECURSION, C.byref(ll),
                                               _lib.dptr(g), P), ctx.h)
    wll, wg = orc.loglik_grad(c["om"], *data, recursive=True)
    assert rel(ll.value, wll) < TOL
    assert np.max(np.abs(g - wg) / np.maximum(1.0, np.abs(wg))) < 1e-9
