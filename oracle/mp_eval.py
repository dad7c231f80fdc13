"""50-digit mpmath evaluation of the reference's formulas for tiny cases.

TEST INFRASTRUCTURE: an evaluator independent of oracle/nhp_oracle.c (different language,
arbitrary precision, straight from the mathematical definitions) used to pin the C oracle.
Indexing follows the reference: W[p, c] with p the parent node, c the child node; node ids
1-based in `nodes`.
"""
import mpmath as mp

mp.mp.dps = 50


def _pdf(model, p, c, dt):
    dt = mp.mpf(dt)
    if model.theta is not None:
        th = mp.mpf(float(model.theta[p, c]))
        return th * mp.e ** (-th * dt)
    x = dt / mp.mpf(model.dt_max)
    if not (0 < x < 1):
        return mp.mpf(0)
    mu, tau = mp.mpf(float(model.mu[p, c])), mp.mpf(float(model.tau[p, c]))
    z = (mp.log(x / (1 - x)) - mu) * mp.sqrt(tau)
    return mp.e ** (-z * z / 2) * mp.sqrt(tau / (2 * mp.pi)) / (x * (1 - x))


def _weight(model, p, c):
    w = mp.mpf(float(model.W[p, c]))
    if model.A is not None:
        w *= mp.mpf(float(model.A[p, c]))
    return w


def _baseline(model, c, t):
    if model.grid_x is None:
        return mp.mpf(float(model.lambda0[c]))
    x, y = model.grid_x, model.lambda0[c]
    for i in range(len(x) - 1):
        if x[i] <= t < x[i + 1]:
            x0, x1, y0, y1 = (mp.mpf(float(v)) for v in (x[i], x[i + 1], y[i], y[i + 1]))
            return (y1 * (mp.mpf(t) - x0) + y0 * (x1 - mp.mpf(t))) / (x1 - x0)
    return mp.mpf(float(y[-1]))


def _baseline_integral(model, duration):
    if model.grid_x is None:
        return sum(mp.mpf(float(v)) * mp.mpf(duration) for v in model.lambda0)
    s = mp.mpf(0)
    x = model.grid_x
    for c in range(model.N):
        y = model.lambda0[c]
        for i in range(len(x) - 1):
            s += (mp.mpf(float(y[i])) + mp.mpf(float(y[i + 1]))) / 2 * (mp.mpf(float(x[i + 1])) - mp.mpf(float(x[i])))
    return s


def event_intensity(model, times, nodes, i, recursive=False):
    """λ_{c_i}(t_i) with the windowed rule (t_j > t_i - Δtmax, strict, in fp64 as the reference
    evaluates it) or the recursive rule (every earlier event with t_j > 0)."""
    t, c = times[i], int(nodes[i]) - 1
    lam = _baseline(model, c, t)
    thr = float(t) - model.dt_max
    for j in range(i - 1, -1, -1):
        if recursive:
            if not times[j] > 0.0:
                continue
        elif not times[j] > thr:
            break
        p = int(nodes[j]) - 1
        lam += _weight(model, p, c) * _pdf(model, p, c, mp.mpf(float(t)) - mp.mpf(float(times[j])))
    return lam


def loglik(model, times, nodes, duration, recursive=False):
    ll = -_baseline_integral(model, duration)
    masked = model.A is not None and not recursive
    for n in nodes:
        p = int(n) - 1
        for c in range(model.N):
            w = mp.mpf(float(model.W[p, c]))
            if masked:
                w *= mp.mpf(float(model.A[p, c]))
            ll -= w
    for i in range(len(times)):
        ll += mp.log(event_intensity(model, times, nodes, i, recursive))
    return ll


def parent_probabilities(model, times, nodes, i):
    """Normalised categorical weights of event i, most-recent-first, baseline last."""
    t, c = times[i], int(nodes[i]) - 1
    thr = float(t) - model.dt_max
    w = []
    for j in range(i - 1, -1, -1):
        if not times[j] > thr:
            break
        p = int(nodes[j]) - 1
        w.append(_weight(model, p, c) * _pdf(model, p, c, mp.mpf(float(t)) - mp.mpf(float(times[j]))))
    w.append(_baseline(model, c, t))
    s = sum(w)
    return [v / s for v in w]
