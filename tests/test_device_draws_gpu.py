"""The device-side random variates of the Gibbs kernels (csrc/nhp_rng.h) held to

 (1) KNOWN ANSWERS: the Philox4x32-10 -> uniform -> Marsaglia-Tsang Gamma / Box-Muller normal / Beta chain recomputed
     here in plain Python integers and floats for fixed (seed, step, element) counters, and
 (2) their DISTRIBUTIONS by Kolmogorov-Smirnov tests of >= 10^4 draws per family through the probability integral
     transform against scipy.stats cdfs (a wrong squeeze in the Gamma sampler or a biased Box-Muller tail moves the
     KS statistic; two-moment bands do not see either).

Julia's own samplers cannot be matched bit for bit ([3P] Distributions / Random): parity of the conjugate draws with the
reference (src/baselines.jl:72-77, src/weights.jl:59-64, src/impulses.jl:68-73,204-214, src/networks.jl:70-78) is
distributional, and this file is where it is checked.
"""
import ctypes as C
import math

import numpy as np
import pytest
from scipy import stats

from helpers import random_case

pytestmark = pytest.mark.gpu

M32 = 0xFFFFFFFF


# ---- plain-Python mirror of csrc/nhp_rng.h ---------------------------------------------------------------------------
def philox(key, step, e, attempt):
    c0, c1 = e & M32, ((e >> 32) ^ (attempt << 8)) & M32
    c2, c3 = step & M32, (step >> 32) & M32
    k0, k1 = key & M32, (key >> 32) & M32
    for _ in range(10):
        p0, p1 = 0xD2511F53 * c0, 0xCD9E8D57 * c2
        c0, c1, c2, c3 = ((p1 >> 32) ^ c1 ^ k0) & M32, p1 & M32, ((p0 >> 32) ^ c3 ^ k1) & M32, p0 & M32
        k0, k1 = (k0 + 0x9E3779B9) & M32, (k1 + 0xBB67AE85) & M32
    return c0, c1, c2, c3


def two_uniforms(key, step, e, attempt):
    c0, c1, c2, c3 = philox(key, step, e, attempt)
    return ((((c0 << 32) | c1) >> 11) + 1.0) * 2.0 ** -53, ((((c2 << 32) | c3) >> 11) + 1.0) * 2.0 ** -53


def normal(key, step, e, attempt=0):
    ua, ub = two_uniforms(key, step, e, attempt)
    return math.sqrt(-2.0 * math.log(ua)) * math.cos(2.0 * math.pi * int(ub * 4294967296.0) / 4294967296.0)


def gamma(shape, scale, key, step, e):
    boost, attempt = 1.0, 0
    if shape < 1.0:
        ua, _ = two_uniforms(key, step, e, attempt)
        attempt += 1
        boost = ua ** (1.0 / shape)
        shape += 1.0
    d = shape - 1.0 / 3.0
    c = 1.0 / math.sqrt(9.0 * d)
    while True:
        c0, c1, c2, c3 = philox(key, step, e, attempt)
        attempt += 1
        ua = ((((c0 << 32) | c1) >> 11) + 1.0) * 2.0 ** -53
        x = math.sqrt(-2.0 * math.log(ua)) * math.cos(2.0 * math.pi * c2 / 4294967296.0)
        u = (c3 + 0.5) * 2.0 ** -32
        t = 1.0 + c * x
        v = t * t * t
        if v > 0.0 and (u < 1.0 - 0.0331 * x ** 4 or math.log(u) < 0.5 * x * x + d - d * v + d * math.log(v)):
            return d * v * scale * boost


def draws(nhp, kind, seed, step, a=None, b=None, n=None):
    from nhp_amd import _lib
    ctx = nhp.default_context()
    n = len(a) if a is not None else n
    a = None if a is None else np.ascontiguousarray(a, dtype=np.float64)
    b = None if b is None else np.ascontiguousarray(b, dtype=np.float64)
    out = np.empty(n)
    _lib.check(_lib.lib().nhp_probe_draws(ctx.h, kind, seed, step, n, _lib.dptr(a), _lib.dptr(b), _lib.dptr(out)), ctx.h)
    return out


# ---- (1) known answers -----------------------------------------------------------------------------------------------
def test_philox_gamma_normal_beta_known_answers(nhp):
    # Philox4x32-10 itself: the Random123 known-answer vectors (counter, key) -> output
    assert philox(0, 0, 0, 0) == (0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8)
    # eight fixed counters per family, shapes below 1 (the boost path), near 1, moderate, and count-sized
    shapes = np.array([0.3, 0.9, 1.0, 2.5, 7.0, 50.0, 1e3, 5e5])
    scales = np.array([1.0, 0.5, 2.0, 1.0, 0.1, 3.0, 1e-3, 1.0])
    for seed, step in ((0, 0), (12345, 7), (2 ** 63 + 11, 2 ** 33 + 5)):
        got = draws(nhp, 0, seed, step, shapes, scales)
        want = np.array([gamma(float(s), float(c), seed, step, i) for i, (s, c) in enumerate(zip(shapes, scales))])
        assert np.allclose(got, want, rtol=1e-11, atol=0.0), (seed, step, got, want)
        gn = draws(nhp, 1, seed, step, n=8)
        wn = np.array([normal(seed, step, i) for i in range(8)])
        assert np.allclose(gn, wn, rtol=1e-10, atol=1e-13)
        gb = draws(nhp, 2, seed, step, shapes, shapes[::-1].copy())
        wb = []
        for i, (a, b) in enumerate(zip(shapes, shapes[::-1])):
            x, y = gamma(float(a), 1.0, seed, step, 2 * i), gamma(float(b), 1.0, seed, step, 2 * i + 1)
            wb.append(x / (x + y))
        assert np.allclose(gb, wb, rtol=1e-11, atol=0.0)


def test_device_rho_draw_known_answer(nhp):
    """resample!(network, A) on the device (k_rho_draw): ρ = X/(X+Y), X ~ Gamma(α + ΣA), Y ~ Gamma(β + N² - ΣA), keyed
    (seed ^ 0x3F84D5B5B5470917, step, 0 | 1) -- src/networks.jl:70-78."""
    from nhp_amd import _lib
    c = random_case(4, 300, 50.0, "exponential", 1.0, network=True, seed=1, nhp=nhp)
    ctx = nhp.default_context()
    model = c["proc"].device_model(ctx)
    lib = _lib.lib()
    _lib.check(lib.nhp_cont_model_set_rho(ctx.h, model.h, 0.5), ctx.h)
    r3 = np.empty(3)
    for seed, step, links, a, b in ((7, 0, 5.0, 1.0, 1.0), (99, 41, 11.0, 2.5, 0.5), (2 ** 40 + 3, 2 ** 35, 0.0, 1.0, 3.0)):
        _lib.check(lib.nhp_cont_network_rho(ctx.h, model.h, a, b, links, 16.0, seed, step), ctx.h)
        _lib.check(lib.nhp_cont_model_get_rho(ctx.h, model.h, _lib.dptr(r3)), ctx.h)
        key = seed ^ 0x3F84D5B5B5470917
        x, y = gamma(a + links, 1.0, key, step, 0), gamma(b + 16.0 - links, 1.0, key, step, 1)
        assert abs(r3[0] - x / (x + y)) < 1e-11


# ---- (2) distributions: Kolmogorov-Smirnov through the probability integral transform ------------------------------------
P_MIN = 1e-4          # a correct sampler fails one such test in 10^4 runs; the statistics below are deterministic (fixed seeds)


def ks_uniform(u):
    return stats.kstest(u, "uniform").pvalue


@pytest.mark.parametrize("shape", [0.3, 1.0, 2.5, 40.0, 1e4, 6e5])
def test_gamma_draws_ks(nhp, shape):
    n = 40_000
    rng = np.random.default_rng(int(shape * 10))
    scale = rng.uniform(0.2, 3.0, n)
    x = draws(nhp, 0, 1234 + int(shape), 5, np.full(n, shape), scale)
    assert np.all(x > 0) and np.all(np.isfinite(x))
    assert ks_uniform(stats.gamma.cdf(x / scale, shape)) > P_MIN
    # and the tails specifically (where a wrong squeeze or a truncated normal would bite): upper / lower 1 % counts
    u = stats.gamma.cdf(x / scale, shape)
    for tail in ((u < 0.01).sum(), (u > 0.99).sum()):
        assert abs(tail - 0.01 * n) < 5.0 * math.sqrt(0.01 * n)


def test_gamma_draws_with_mixed_shapes_ks(nhp):
    # the shapes a Gibbs sweep actually sees: prior + small counts, element by element different
    n = 60_000
    rng = np.random.default_rng(3)
    shape = 1.0 + rng.poisson(1.5, n) * rng.choice([0.5, 1.0], n)
    scale = 1.0 / (1.0 + rng.poisson(900, n))
    x = draws(nhp, 0, 77, 123456789, shape, scale)
    assert ks_uniform(stats.gamma.cdf(x / scale, shape)) > P_MIN


def test_normal_draws_ks(nhp):
    n = 100_000
    z = draws(nhp, 1, 31337, 2, n=n)
    assert ks_uniform(stats.norm.cdf(z)) > P_MIN
    # Box-Muller's tail: P(|z| > 3.5) = 4.65e-4 -> 46.5 expected of 10^5
    assert abs((np.abs(z) > 3.5).sum() - 46.5) < 5.0 * math.sqrt(46.5)
    assert abs(stats.skew(z)) < 0.03 and abs(stats.kurtosis(z)) < 0.06


@pytest.mark.parametrize("a,b", [(1.0, 1.0), (0.5, 3.0), (30.0, 70.0), (5e5 + 1, 5.5e5 + 1)])
def test_beta_draws_ks(nhp, a, b):
    n = 30_000
    r = draws(nhp, 2, 4242, 9, np.full(n, a), np.full(n, b))
    assert np.all((r > 0) & (r < 1))
    assert ks_uniform(stats.beta.cdf(r, a, b)) > P_MIN


def test_device_rho_chain_draws_are_beta(nhp):
    """ρ as a network mcmc! step draws it (nhp_cont_network_rho, the second half of nhp_cont_network_step): 3000 steps with
    the link count of a N = 1024 half-full matrix -- Beta(α + ΣA, β + N² - ΣA)."""
    from nhp_amd import _lib
    c = random_case(4, 300, 50.0, "exponential", 1.0, network=True, seed=2, nhp=nhp)
    ctx = nhp.default_context()
    model = c["proc"].device_model(ctx)
    lib = _lib.lib()
    _lib.check(lib.nhp_cont_model_set_rho(ctx.h, model.h, 0.5), ctx.h)
    links, nn, r3, rho = 524_000.0, 1024.0 ** 2, np.empty(3), []
    for step in range(3000):
        _lib.check(lib.nhp_cont_network_rho(ctx.h, model.h, 1.0, 1.0, links, nn, 5, step), ctx.h)
        _lib.check(lib.nhp_cont_model_get_rho(ctx.h, model.h, _lib.dptr(r3)), ctx.h)
        rho.append(r3[0])
    assert ks_uniform(stats.beta.cdf(np.array(rho), 1.0 + links, 1.0 + nn - links)) > P_MIN


def test_gibbs_sweep_draws_ks(nhp):
    """One device sweep (nhp_cont_gibbs_step) on a model whose sufficient statistics we also fetch: every family of the N²
    conjugate draws, transformed by ITS OWN posterior cdf (element by element different shapes / rates), is uniform --
    src/baselines.jl:72-77, src/weights.jl:59-64, src/impulses.jl:204-214."""
    from nhp_amd import _lib, inference
    N = 100
    c = random_case(N, 80_000, 2000.0, "logitnormal", 1.0, seed=31, nhp=nhp)
    proc, ctx = c["proc"], nhp.default_context()
    ds = nhp.device_dataset(proc, c["data"], ctx)
    _, _, st = nhp.resample_parents(proc, ds, seed=9, step=4, with_stats=True, want_parents=False)
    model = proc.device_model(ctx)
    pri = inference._priors(proc)
    _lib.check(_lib.lib().nhp_cont_gibbs_step(ctx.h, ds.h, model.h, C.byref(pri), 9, 4), ctx.h)
    inference._pull_params(proc, model, ctx)
    Mnm, Mn, X, V = st["Mnm"], st["Mn"], st["Xnm"], st["Vnm"]
    # W[p,c] ~ Gamma(κ + Mnm, 1/(ν + Mn[p]))
    shape, rate = 1.0 + Mnm, (1.0 + Mn)[:, None] * np.ones((N, N))
    assert ks_uniform(stats.gamma.cdf(proc.weights.W * rate, shape).ravel()) > P_MIN
    with np.errstate(invalid="ignore"):
        bnm = V / 2 + Mnm * 1.0 / (Mnm + 1.0) * (X - 1.0) ** 2 / 2
        bnm[np.isnan(bnm)] = 1.0
        mnm = (1.0 + Mnm * X) / (1.0 + Mnm)
        mnm[np.isnan(mnm)] = 1.0
    tshape = 1.0 + Mnm / 2
    assert ks_uniform(stats.gamma.cdf(proc.impulses.τ * bnm, tshape).ravel()) > P_MIN
    zm = (proc.impulses.μ - mnm) * np.sqrt((1.0 + Mnm) * proc.impulses.τ)
    assert ks_uniform(stats.norm.cdf(zm).ravel()) > P_MIN
    lam_shape, lam_rate = 1.0 + st["cnt0"], 1.0 + c["T"]
    assert ks_uniform(stats.gamma.cdf(proc.baseline.λ * lam_rate, lam_shape)) > 1e-3      # N = 100 draws
    assert np.all(proc.weights.W > 0) and np.all(proc.impulses.τ > 0)
