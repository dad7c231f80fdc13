"""Pins the CPU oracle for the continuous path (CPU only).

The reference's tests hold no vector for loglikelihood / intensity / resample_parents
(SURVEY.md 4, 8c), so the oracle is pinned by: the reference fixtures that do exist
(test/baselines.jl:8-25, test/interpolation.jl:7-14), closed forms, a 50-digit mpmath
evaluator, equality of the two independent formulations, and finite differences.
"""
import math

import numpy as np
import pytest

from helpers import random_case, rel


# ---- reference fixtures, verbatim --------------------------------------------------------
def test_reference_node_counts_fixture(orc):
    # test/baselines.jl:8-25
    assert list(orc.baseline_node_counts([1, 1, 2, 2], [0, 1, 0, 2], 2)) == [1, 1]
    assert list(orc.baseline_node_counts([], [], 2)) == [0, 0]
    assert list(orc.baseline_node_counts([1, 1, 2, 2], [1, 2, 1, 2], 2)) == [0, 0]


def test_reference_interpolator_fixture(orc):
    # test/interpolation.jl:7-14: x = 0:0.5:2pi, y = sin.(x)
    import ctypes as C
    x = np.arange(0.0, 2 * math.pi, 0.5)
    y = np.sin(x)
    lib = orc.lib()
    out = C.c_double()
    dp = C.POINTER(C.c_double)

    def f(x0):
        rc = lib.orc_linear_interpolate(x.ctypes.data_as(dp), y.ctypes.data_as(dp), C.c_int32(len(x)),
                                        C.c_double(x0), C.byref(out))
        return rc, out.value

    assert f(0.0) == (0, 0.0)
    assert f(0.1) == (0, math.sin(0.5) / .5 * .1)
    assert f(2 * math.pi)[0] == 2          # DomainError
    assert f(-1.0)[0] == 2
    integ = lib.orc_linear_integrate(x.ctypes.data_as(dp), y.ctypes.data_as(dp), C.c_int32(len(x)))
    assert integ == np.sum(y[1:] + y[:-1]) * 0.5 ** 2


# ---- closed forms --------------------------------------------------------------------------
def test_empty_and_single_event(orc):
    lam0 = np.array([0.7, 1.3])
    W = np.array([[0.1, 0.2], [0.3, 0.4]])
    th = np.ones((2, 2))
    m = orc.ContModel(lam0, W, theta=th, dt_max=2.0)
    T = 10.0
    assert orc.loglik_windowed(m, [], [], T) == -(lam0.sum() * T)
    assert orc.loglik_recursive(m, [], [], T) == -(lam0.sum() * T)
    # one event on node 2: ll = -Σλ0·T - ΣW[2,:] + log λ0[2]
    want = -(lam0 * T).sum() - W[1].sum() + math.log(lam0[1])
    assert rel(orc.loglik_windowed(m, [3.0], [2], T), want) < 1e-15
    assert rel(orc.loglik_recursive(m, [3.0], [2], T), want) < 1e-15


def test_two_events_by_hand(orc):
    lam0 = np.array([0.5, 0.25])
    W = np.array([[0.3, 0.6], [0.2, 0.1]])
    th = np.array([[2.0, 3.0], [4.0, 5.0]])
    m = orc.ContModel(lam0, W, theta=th, dt_max=10.0)
    t1, t2, T = 1.0, 1.5, 4.0
    lam2 = lam0[1] + W[0, 1] * th[0, 1] * math.exp(-th[0, 1] * (t2 - t1))
    want = -(lam0 * T).sum() - W[0].sum() - W[1].sum() + math.log(lam0[0]) + math.log(lam2)
    assert rel(orc.loglik_windowed(m, [t1, t2], [1, 2], T), want) < 1e-15
    assert rel(orc.loglik_recursive(m, [t1, t2], [1, 2], T), want) < 1e-15


def test_window_is_strict_and_stops(orc):
    # parent exactly Δtmax old is excluded (events[j] > t - Δtmax is strict), src/continuous.jl:291
    lam0, W, th = np.array([1.0]), np.array([[0.5]]), np.array([[1.0]])
    m = orc.ContModel(lam0, W, theta=th, dt_max=1.0)
    lam = orc.total_intensity(m, [1.0, 2.0, 2.5], [1, 1, 1])
    assert lam[1] == 1.0                                        # Δt = 1.0 = Δtmax -> excluded
    assert rel(lam[2], 1.0 + 0.5 * math.exp(-0.5)) < 1e-15      # only the event at 2.0


def test_ozaki_univariate_recursion(orc):
    # N = 1: λ_i = λ0 + Wθ·R_i with R_i = e^{-θΔ}(1 + R_{i-1})  (Ozaki 1979)
    rng = np.random.default_rng(5)
    t = np.sort(rng.uniform(0.1, 30.0, 200))
    lam0, W, th, T = 0.8, 0.6, 1.7, 30.0
    m = orc.ContModel([lam0], [[W]], theta=[[th]], dt_max=np.inf)
    R, ll = 0.0, -lam0 * T - len(t) * W
    for i in range(len(t)):
        if i > 0:
            R = math.exp(-th * (t[i] - t[i - 1])) * (1.0 + R)
        ll += math.log(lam0 + W * th * R)
    assert rel(orc.loglik_recursive(m, t, np.ones(len(t), int), T), ll) < 1e-13
    assert rel(orc.loglik_windowed(m, t, np.ones(len(t), int), T), ll) < 1e-13


# ---- two formulations agree -------------------------------------------------------
@pytest.mark.parametrize("network", [False, True])
def test_windowed_inf_equals_recursive(orc, network):
    c = random_case(6, 800, 100.0, "exponential", np.inf, network=False, seed=3, orc=orc)
    a = orc.loglik_windowed(c["om"], c["times"], c["nodes"], c["T"])
    b = orc.loglik_recursive(c["om"], c["times"], c["nodes"], c["T"])
    assert rel(a, b) < 1e-12
    if network:
        # the recursive network twin does NOT mask the integral term (D7): they differ by exactly
        # Σ_i Σ_c (1-A)[n_i,c]·W[n_i,c]
        c = random_case(6, 800, 100.0, "exponential", np.inf, network=True, seed=3, orc=orc)
        a = orc.loglik_windowed(c["om"], c["times"], c["nodes"], c["T"])
        b = orc.loglik_recursive(c["om"], c["times"], c["nodes"], c["T"])
        om = c["om"]
        gap = sum(((1 - om.A[n - 1]) * om.W[n - 1]).sum() for n in c["nodes"])
        assert rel(a - b, gap) < 1e-9


def test_recursive_drops_events_at_time_zero(orc):
    # D9: parenttimes > 0.0 is the "seen" flag, so an event at exactly 0.0 never becomes a parent
    lam0, W, th = np.array([1.0, 1.0]), np.full((2, 2), 0.4), np.full((2, 2), 1.5)
    m = orc.ContModel(lam0, W, theta=th, dt_max=np.inf)
    t, n, T = [0.0, 0.5, 0.9], [1, 2, 1], 2.0
    rec = orc.loglik_recursive(m, t, n, T)
    win = orc.loglik_windowed(m, t, n, T)
    lam2 = 1.0                                       # event at 0.5: its only parent is at t=0 -> dropped
    lam3 = 1.0 + 0.4 * 1.5 * math.exp(-1.5 * 0.4)    # event at 0.9: parent at 0.5 only
    want = -2.0 * T - 3 * 0.8 + math.log(lam2) + math.log(lam3)
    assert rel(rec, want) < 1e-15
    assert win > rec                                 # the windowed path keeps the t=0 parent


# ---- mpmath, 50 digits ----------------------------------------------------------------------
@pytest.mark.parametrize("kind,network,lgcp", [("exponential", False, False), ("exponential", True, False),
                                               ("logitnormal", False, False), ("logitnormal", True, True),
                                               ("exponential", False, True)])
def test_against_mpmath(orc, kind, network, lgcp):
    from oracle import mp_eval
    c = random_case(3, 150, 30.0, kind, 1.5, network=network, lgcp=lgcp, seed=11, orc=orc)
    want = mp_eval.loglik(c["om"], c["times"], c["nodes"], c["T"])
    for flags in (orc.MATH_LIBM, orc.MATH_DET, orc.FAST_INTEGRAL):
        got = orc.loglik_windowed(c["om"], c["times"], c["nodes"], c["T"], flags=flags)
        assert abs(got - float(want)) / abs(float(want)) < 1e-12
    lam = orc.total_intensity(c["om"], c["times"], c["nodes"])
    for i in (0, 1, 57, 149):
        assert abs(lam[i] - float(mp_eval.event_intensity(c["om"], c["times"], c["nodes"], i))) < 1e-12 * lam[i]


def test_recursive_against_mpmath(orc):
    from oracle import mp_eval
    c = random_case(3, 150, 30.0, "exponential", np.inf, network=True, seed=12, orc=orc)
    want = float(mp_eval.loglik(c["om"], c["times"], c["nodes"], c["T"], recursive=True))
    assert rel(orc.loglik_recursive(c["om"], c["times"], c["nodes"], c["T"]), want) < 1e-12
    assert rel(orc.loglik_recursive(c["om"], c["times"], c["nodes"], c["T"], flags=orc.MATH_DET), want) < 1e-12


# ---- [3P] pieces vs scipy -----------------------------------------------------------------------
def test_logitnormal_pdf_vs_scipy(orc):
    from scipy.stats import norm
    rng = np.random.default_rng(2)
    for _ in range(200):
        mu, tau, dtm = rng.normal(), rng.uniform(0.3, 3), rng.uniform(0.5, 4)
        dt = rng.uniform(0, dtm)
        x = dt / dtm
        want = norm.pdf(math.log(x / (1 - x)), mu, tau ** -0.5) / (x * (1 - x))
        for flags in (0, 1):
            got = orc.lib().orc_impulse_logitnormal(mu, tau, dtm, dt, flags)
            assert rel(got, want) < 1e-12        # tails: exp(-z²/2) amplifies 1-ulp errors of logit(x)
    assert orc.lib().orc_impulse_logitnormal(0.0, 1.0, 1.0, 1.0, 0) == 0.0     # x = 1 -> 0
    assert orc.lib().orc_impulse_logitnormal(0.0, 1.0, 1.0, 0.0, 1) == 0.0     # x = 0 -> 0


def test_det_math_accuracy(orc):
    rng = np.random.default_rng(9)
    for x in rng.uniform(-708.0, 0.0, 5000):
        assert rel(orc.det_exp(x), math.exp(x)) < 4.5e-16
    assert orc.det_exp(-800.0) == 0.0 and orc.det_exp(0.0) == 1.0
    for x in np.exp(rng.uniform(-300, 300, 5000)):
        assert abs(orc.det_log(x) - math.log(x)) <= 2.3e-16 * max(1.0, abs(math.log(x)))
    assert orc.det_log(1.0) == 0.0


# ---- intensity(process, data, times) ---------------------------------------------------------------
def test_intensity_matches_total_intensity_just_before_events(orc):
    c = random_case(4, 300, 40.0, "logitnormal", 2.0, network=True, seed=4, orc=orc)
    q = np.array([0.0, 5.0, 17.3, 39.9])
    lam = orc.intensity(c["om"], c["times"], c["nodes"], q)
    assert lam.shape == (4, 4)
    from oracle import mp_eval
    # build the same quantity from the definition: strict window on both sides
    for k, t in enumerate(q):
        for ch in range(4):
            s = c["om"].lambda0[ch]
            for tj, nj in zip(c["times"], c["nodes"]):
                if t - 2.0 < tj < t:
                    s += float(mp_eval._weight(c["om"], nj - 1, ch) * mp_eval._pdf(c["om"], nj - 1, ch, t - tj))
            assert rel(lam[k, ch], s) < 1e-12


# ---- analytic gradient vs central finite differences ----------------------------------------------
@pytest.mark.parametrize("kind,recursive", [("exponential", False), ("exponential", True), ("logitnormal", False)])
def test_gradient_finite_differences(orc, kind, recursive):
    dtm = np.inf if recursive else 1.5
    c = random_case(3, 120, 25.0, kind, dtm, seed=21, orc=orc)
    om = c["om"]
    ll, g = orc.loglik_grad(om, c["times"], c["nodes"], c["T"], recursive=recursive)
    assert rel(ll, orc.loglik(om, c["times"], c["nodes"], c["T"], recursive=recursive)) < 1e-12
    x0 = om.params_vector()
    N = 3

    def f(x):
        lam0 = x[:N]
        if kind == "exponential":
            th = x[N:N + 9].reshape((N, N), order="F")
            W = x[N + 9:].reshape((N, N), order="F")
            m = orc.ContModel(lam0, W, theta=th, dt_max=dtm)
        else:
            mu = x[N:N + 9].reshape((N, N), order="F")
            tau = x[N + 9:N + 18].reshape((N, N), order="F")
            W = x[N + 18:].reshape((N, N), order="F")
            m = orc.ContModel(lam0, W, mu=mu, tau=tau, dt_max=dtm)
        return orc.loglik(m, c["times"], c["nodes"], c["T"], recursive=recursive)

    for k in range(len(x0)):
        h = 1e-6 * max(1.0, abs(x0[k]))
        xp, xm = x0.copy(), x0.copy()
        xp[k] += h
        xm[k] -= h
        fd = (f(xp) - f(xm)) / (2 * h)
        assert abs(fd - g[k]) < 1e-5 * max(1.0, abs(g[k])), (k, fd, g[k])


@pytest.mark.parametrize("kind", ["exponential", "logitnormal"])
def test_threaded_branch_equals_serial_branch(orc, kind):
    """src/continuous.jl:224-232 (Threads.@threads + atomic add) against :233-237 (serial): the same terms in another
    association; the all-cores CPU baseline of bench.py."""
    rng = np.random.default_rng(8)
    N, M, T = 6, 4000, 300.0
    t = np.sort(rng.uniform(0, T, M))
    n = rng.integers(1, N + 1, M)
    kw = dict(theta=rng.uniform(1, 5, (N, N))) if kind == "exponential" else dict(mu=rng.normal(0, 1, (N, N)), tau=rng.uniform(0.5, 2, (N, N)))
    A = (rng.uniform(size=(N, N)) < 0.5).astype(np.float64)
    om = orc.ContModel(rng.uniform(0.5, 1.5, N), rng.uniform(0, 1, (N, N)) / N, dt_max=1.5, A=A, **kw)
    want = orc.loglik_windowed(om, t, n, T)
    for threads in (1, 2, 0):
        got = orc.loglik_windowed_mt(om, t, n, T, threads=threads)
        assert abs(got - want) <= 1e-12 * abs(want)
    assert orc.loglik_windowed_mt(om, t[:0], n[:0], T) == orc.loglik_windowed(om, t[:0], n[:0], T)
    assert orc.max_threads() >= 1
    with pytest.raises(Exception):                         # validation errors still surface from inside the threads' caller
        orc.loglik_windowed_mt(om, t[::-1].copy(), n, T)
