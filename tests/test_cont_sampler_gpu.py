"""GPU parity for the parent sampler (reference src/parents.jl:1-79) and its fused Gibbs
statistics.  Parent indices must be BIT-EXACT against the oracle for a fixed uniform stream
(BASELINE.json north_star); that holds by construction because kernel and oracle evaluate
every weight with the same IEEE operation sequence -- checked bitwise first."""
import ctypes as C

import numpy as np
import pytest

from helpers import random_case

pytestmark = pytest.mark.gpu


def probe(nhp, op, x, y=None):
    from nhp_amd import _lib
    ctx = nhp.default_context()
    x = np.ascontiguousarray(x, dtype=np.float64)
    y = None if y is None else np.ascontiguousarray(y, dtype=np.float64)
    out = np.empty_like(x)
    _lib.check(_lib.lib().nhp_probe_math(ctx.h, op, _lib.dptr(x), _lib.dptr(y), len(x), _lib.dptr(out)), ctx.h)
    return out


def test_det_math_is_bitwise_identical(nhp, orc):
    rng = np.random.default_rng(0)
    x = np.concatenate([rng.uniform(-708.0, 0.0, 200_000), -np.exp(rng.uniform(-40, 6, 100_000)), [0.0, -708.0, -745.0, -1e300]])
    lib = orc.lib()
    want = np.array([lib.orc_det_exp(float(v)) for v in x])
    assert np.array_equal(probe(nhp, 4, x).view(np.uint64), want.view(np.uint64))          # exp, x <= 0
    assert np.array_equal(probe(nhp, 0, x).view(np.uint64), want.view(np.uint64))          # general exp
    y = np.concatenate([np.exp(rng.uniform(-700, 700, 200_000)), rng.uniform(0.5, 2.0, 100_000), [1.0, 5e-324, 2.2e-308]])
    want = np.array([lib.orc_det_log(float(v)) for v in y])
    assert np.array_equal(probe(nhp, 1, y).view(np.uint64), want.view(np.uint64))
    # IEEE sqrt and division are correctly rounded on both sides
    assert np.array_equal(probe(nhp, 2, y).view(np.uint64), np.sqrt(y).view(np.uint64))
    z = np.exp(rng.uniform(-300, 300, len(y)))
    assert np.array_equal(probe(nhp, 3, y, z).view(np.uint64), (y / z).view(np.uint64))
    # the two pair evaluators, det mode
    th, dt = rng.uniform(0.1, 9.0, 100_000), rng.uniform(0.0, 3.0, 100_000)
    want = np.array([lib.orc_impulse_exponential(float(a), float(b), 1) for a, b in zip(th, dt)])
    assert np.array_equal(probe(nhp, 5, th, dt).view(np.uint64), want.view(np.uint64))
    tau, dt = rng.uniform(0.3, 3.0, 100_000), rng.uniform(0.0, 2.0, 100_000)
    want = np.array([lib.orc_impulse_logitnormal(0.25, float(a), 2.0, float(b), 1) for a, b in zip(tau, dt)])
    assert np.array_equal(probe(nhp, 6, tau, dt).view(np.uint64), want.view(np.uint64))


def test_table_driven_exp_of_the_loglik_kernels_is_within_two_ulp(nhp):
    """nhp_exp_neg_tab (64-entry 2^(j/64) table in LDS + degree-5 polynomial; csrc/nhp_math.h) is what the log-likelihood
    kernels evaluate e^{-θΔt} with: not bit-identical to the det-math exp the sampler shares with the oracle, but within
    2 ulp of the correctly rounded value over the whole range, underflowing gradually like exp itself."""
    import mpmath as mp
    rng = np.random.default_rng(7)
    x = np.concatenate([-rng.uniform(0.0, 745.0, 150_000), -np.exp(rng.uniform(-40, 3, 100_000)), [0.0, -1e-300, -708.0, -744.0, -745.2, -800.0, -1e6]])
    got = probe(nhp, 7, x)
    want = np.exp(x)                                    # (glibc: < 1 ulp)
    big = want > 1e-300
    assert np.max(np.abs(got[big] - want[big]) / want[big]) < 3.0 * 2.0 ** -53
    assert np.all(np.abs(got[~big] - want[~big]) <= 4e-308 * 1e-15 + 3.0 * 2.0 ** -53 * want[~big] + 5e-324 * 4)
    assert got[-1] == 0.0 and got[-2] == 0.0 and got[len(x) - 7] == 1.0
    mp.mp.prec = 120                                    # a few points against 120-bit arithmetic
    for v in (-0.3, -5.25, -37.0, -123.456, -700.5):
        g = float(probe(nhp, 7, np.array([v]))[0])
        assert abs((mp.mpf(g) - mp.exp(mp.mpf(v))) / mp.exp(mp.mpf(v))) < 2.3 * mp.mpf(2) ** -53


def test_uniform_stream_matches_oracle(nhp, orc):
    for seed, step in ((0, 0), (1, 7), (2 ** 63 + 5, 2 ** 40 + 3)):
        assert np.array_equal(nhp.uniform_stream(seed, step, 1000), orc.uniform_stream(seed, step, 1000))


@pytest.mark.parametrize("kind", ["exponential", "logitnormal"])
@pytest.mark.parametrize("network,lgcp", [(False, False), (True, False), (True, True)])
def test_parents_bit_exact(nhp, orc, kind, network, lgcp):
    c = random_case(8, 6000, 300.0, kind, 1.0, network=network, lgcp=lgcp, seed=4, nhp=nhp, orc=orc)
    u = np.random.default_rng(99).uniform(size=6000)
    got_p, got_n = nhp.resample_parents(c["proc"], c["data"], u=u)
    want_p, want_n = orc.resample_parents(c["om"], c["times"], c["nodes"], u, flags=orc.MATH_DET)
    assert np.array_equal(got_p, want_p)
    assert np.array_equal(got_n, want_n)
    assert got_p[0] == 0 and got_n[0] == 0                       # event 1 -> (0, 0)  src/parents.jl:26-28
    # Philox stream generated inside the kernel == host-supplied stream
    p2, n2 = nhp.resample_parents(c["proc"], c["data"], seed=11, step=3)
    w2, wn2 = orc.resample_parents(c["om"], c["times"], c["nodes"], orc.uniform_stream(11, 3, 6000), flags=orc.MATH_DET)
    assert np.array_equal(p2, w2) and np.array_equal(n2, wn2)
    # libm-evaluated reference-faithful weights give the same indices on this stream
    w3, _ = orc.resample_parents(c["om"], c["times"], c["nodes"], u, flags=orc.MATH_LIBM)
    assert np.array_equal(got_p, w3)


def test_pairwise_sum_for_windows_above_1024(nhp, orc):
    # Δtmax = Inf: event i has i+1 weights; Julia's sum switches to midpoint-split pairwise above 1024
    c = random_case(4, 5000, 50.0, "exponential", np.inf, seed=6, nhp=nhp, orc=orc)
    u = np.random.default_rng(5).uniform(size=5000)
    got_p, got_n = nhp.resample_parents(c["proc"], c["data"], u=u)
    want_p, want_n = orc.resample_parents(c["om"], c["times"], c["nodes"], u, flags=orc.MATH_DET)
    assert np.array_equal(got_p, want_p) and np.array_equal(got_n, want_n)


def test_extreme_uniforms_and_empty_windows(nhp):
    c = random_case(5, 3000, 100.0, "exponential", 2.0, seed=2, nhp=nhp)
    M = 3000
    p, n = nhp.resample_parents(c["proc"], c["data"], u=np.zeros(M))
    # u = 0 picks the first entry: the most recent parent (i-1) whenever the window is not empty
    t = c["times"]
    has_parent = np.concatenate([[False], t[:-1] > t[1:] - 2.0])
    idx = np.arange(M)
    assert np.array_equal(p[has_parent], idx[has_parent])        # 1-based index of event i-1 is i
    assert np.all(p[~has_parent] == 0)
    p, n = nhp.resample_parents(c["proc"], c["data"], u=np.full(M, np.nextafter(1.0, 0.0)))
    assert np.all(p == 0) and np.all(n == 0)                     # u -> 1-: the last entry, the baseline
    c = random_case(5, 500, 1e6, "exponential", 1e-3, seed=3, nhp=nhp)     # every window empty
    p, n = nhp.resample_parents(c["proc"], c["data"], seed=1)
    assert np.all(p == 0) and np.all(n == 0)


def test_sampling_frequencies_chi2(nhp):
    # one child with 3 possible parents + baseline, replicated through the step counter
    from oracle import mp_eval
    from oracle import oracle as orc
    c = random_case(3, 4, 1.0, "exponential", 10.0, seed=7, nhp=nhp, orc=orc)
    probs = np.array([float(v) for v in mp_eval.parent_probabilities(c["om"], c["times"], c["nodes"], 3)])
    counts = np.zeros(4)
    R = 4000
    for step in range(R):
        p, _ = nhp.resample_parents(c["proc"], c["data"], seed=123, step=step)
        counts[3 if p[3] == 0 else 3 - p[3]] += 1              # parent index 3,2,1 -> slot 0,1,2; baseline -> 3
    chi2 = np.sum((counts - R * probs) ** 2 / (R * probs))
    assert chi2 < 21.1                                           # χ²(3) at p = 1e-4


@pytest.mark.parametrize("kind", ["exponential", "logitnormal"])
def test_fused_statistics_match_reference_helpers(nhp, orc, kind):
    N, M = 6, 8000
    c = random_case(N, M, 400.0, kind, 1.5, network=True, seed=13, nhp=nhp, orc=orc)
    p, pn, st = nhp.resample_parents(c["proc"], c["data"], seed=5, step=2, with_stats=True)
    nodes, times = c["nodes"], c["times"]
    assert np.array_equal(st["cnt0"], orc.baseline_node_counts(nodes, pn, N))
    assert np.array_equal(st["Mn"], orc.node_counts(nodes, N))
    assert np.array_equal(st["Mnm"], orc.parent_counts(nodes, pn, N))
    assert st["Mnm"].sum() + st["cnt0"].sum() == M
    if kind == "exponential":
        # same accumulation order as the reference's serial loop -> identical bits
        assert np.array_equal(st["Xnm"], orc.duration_mean(times, nodes, p, N))
    else:
        X, V = orc.log_duration_stats(times, nodes, p, N, 1.5)
        assert np.array_equal(np.isnan(st["Xnm"]), np.isnan(X))
        m = ~np.isnan(X)
        assert np.allclose(st["Xnm"][m], X[m], rtol=1e-12, atol=1e-13)
        assert np.allclose(st["Vnm"], V, rtol=1e-11, atol=1e-12)
    # statistics only (no parent vectors copied back)
    _, _, st2 = nhp.resample_parents(c["proc"], c["data"], seed=5, step=2, with_stats=True, want_parents=False)
    assert np.array_equal(st2["Mnm"], st["Mnm"])


@pytest.mark.parametrize("which", ["1", "8"])
@pytest.mark.parametrize("kind", ["exponential", "logitnormal"])
def test_both_sampler_kernels_give_the_oracle_indices(nhp, orc, which, kind, monkeypatch):
    # the single-lane kernel and the 8-lanes-per-child kernel (exact sequential DPP chains) are chosen by mean
    # window; force each and compare with the oracle on short, medium and ragged windows, LGCP baseline, network
    monkeypatch.setenv("NHP_SAMPLER", which)
    from helpers import random_case
    for (N, M, T, dtm, net, lg, seed) in ((5, 3000, 100.0, 0.3, False, False, 1), (7, 4000, 40.0, 1.5, True, True, 2),
                                          (3, 2500, 10.0, 2.0, False, False, 3)):
        c = random_case(N, M, T, kind, dtm, network=net, lgcp=lg, seed=seed, nhp=nhp, orc=orc)
        u = np.random.default_rng(seed).uniform(size=M)
        p, pn = nhp.resample_parents(c["proc"], c["data"], u=u)
        wp, wpn = orc.resample_parents(c["om"], c["times"], c["nodes"], u, flags=orc.MATH_DET)
        assert np.array_equal(p, wp) and np.array_equal(pn, wpn)



@pytest.mark.parametrize("kind,network,lgcp", [("logitnormal", False, False), ("logitnormal", True, True),
                                               ("exponential", False, False), ("exponential", True, True)])
def test_slice_sampler_gives_the_oracle_indices(nhp, orc, kind, network, lgcp, monkeypatch):
    """A sliced dataset draws its parents one lane per child over the slice planes -- logit-normal impulses through the planes
    of logit(x) and 1/(x(1-x)), exponential ones through the plane of exact delays
    (k_sampler_slices: rows coalesced, the first 8 or 16 weights of a child kept in LDS for the scan, later ones evaluated
    again).  Windows from empty to ~60 parents -- well past the cache -- ties and a burst in the data: indices, parent nodes
    and statistics equal the oracle's and the lane-per-child kernel's (NHP_SAMPLER_SLICES=0), for every (workgroup, cache) shape,
    with explicit uniforms and with the Philox stream."""
    c = random_case(9, 7000, 350.0, kind, 1.2, network=network, lgcp=lgcp, seed=41, nhp=nhp, orc=orc)
    monkeypatch.delenv("NHP_SAMPLER_EXPO_SLICES", raising=False)
    t = c["times"].copy()
    t[500:530:2] = t[501:531:2]
    t[3000:3060] = np.sort(np.random.default_rng(8).uniform(t[3000], t[3000] + 0.9, 60))
    t = np.sort(t)
    data = (t, c["nodes"], c["T"])
    M = len(t)
    u = np.random.default_rng(17).uniform(size=M)
    u[::97] = 0.0
    u[5::89] = np.nextafter(1.0, 0.0)
    wp, wpn = orc.resample_parents(c["om"], t, c["nodes"], u, flags=orc.MATH_DET)
    ref = None
    for cfg in (None, "256,8", "256,16", "512,8", "512,16", "off"):
        monkeypatch.delenv("NHP_SAMPLER_SLICES", raising=False)
        monkeypatch.delenv("NHP_SAMPLER_CFG", raising=False)
        if cfg == "off":
            monkeypatch.setenv("NHP_SAMPLER_SLICES", "0")
            monkeypatch.setenv("NHP_SAMPLER_EXPO_SLICES", "0")
        elif cfg:
            monkeypatch.setenv("NHP_SAMPLER_CFG", cfg)
        p, pn, st = nhp.resample_parents(c["proc"], data, u=u, with_stats=True)
        assert np.array_equal(p, wp) and np.array_equal(pn, wpn), cfg
        if ref is None:
            ref = st
        else:
            for k in ("cnt0", "Mn", "Mnm"):
                assert np.array_equal(st[k], ref[k]), (cfg, k)
            assert np.array_equal(np.nan_to_num(st["Xnm"]), np.nan_to_num(ref["Xnm"])), cfg
        p2, pn2 = nhp.resample_parents(c["proc"], data, seed=3, step=9)
        w2, wn2 = orc.resample_parents(c["om"], t, c["nodes"], orc.uniform_stream(3, 9, M), flags=orc.MATH_DET)
        assert np.array_equal(p2, w2) and np.array_equal(pn2, wn2), cfg
    monkeypatch.delenv("NHP_SAMPLER_SLICES", raising=False)
    monkeypatch.delenv("NHP_SAMPLER_CFG", raising=False)
    monkeypatch.delenv("NHP_SAMPLER_EXPO_SLICES", raising=False)


def test_logitnormal_pair_cache_keeps_every_index(nhp, monkeypatch):
    """The logit-normal sampler reads {logit(x), 1/(x(1-x))} made once per pair (k_plq_build) instead of taking the logarithm
    and the division per weight: the same operations in the same order, so parents and statistics are identical with the cache
    switched off (NHP_PLQ=0) -- ties, a far time origin, a network mask and a time-varying baseline in the data."""
    for (N, M, t_lo, t_hi, dtm, net, lg, seed) in ((6, 4000, 5e5, 5e5 + 300.0, 1.0, True, False, 1), (9, 5000, 0.0, 40.0, 0.3, False, True, 2),
                                                   (3, 2500, 0.0, 900.0, 2.5, True, True, 3)):
        c = random_case(N, M, t_hi - t_lo, "logitnormal", dtm, network=net, lgcp=lg, seed=seed, nhp=nhp)
        t = np.sort(c["times"] + t_lo)
        t[100:160:3] = t[101:161:3]
        t = np.sort(t)
        if lg:                                                       # the baseline grid of random_case spans [0, T]
            t = t - t_lo
        data = (t, c["nodes"], float(t[-1] + 1.0))
        u = np.random.default_rng(seed).uniform(size=M)
        out = {}
        for name, off in (("cache", None), ("plain", "0")):
            if off is None:
                monkeypatch.delenv("NHP_PLQ", raising=False)
            else:
                monkeypatch.setenv("NHP_PLQ", off)
            nhp.invalidate_device_datasets()
            p, pn, st = nhp.resample_parents(c["proc"], data, u=u, with_stats=True)
            out[name] = (p, pn, st)
        a, b = out["cache"], out["plain"]
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
        for k in ("cnt0", "Mn", "Mnm"):
            assert np.array_equal(a[2][k], b[2][k])
        assert np.array_equal(np.isnan(a[2]["Xnm"]), np.isnan(b[2]["Xnm"]))
        m = ~np.isnan(a[2]["Xnm"])
        assert np.array_equal(a[2]["Xnm"][m], b[2]["Xnm"][m])
    monkeypatch.delenv("NHP_PLQ", raising=False)
