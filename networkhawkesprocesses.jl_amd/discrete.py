"""Discrete-time process models: host mirror of src/discrete.jl (+ the discrete components of
src/baselines.jl:358-456 and src/impulses.jl:272-375) on top of libnhp.so.

`data` is the reference's N x T Int64 count matrix (src/discrete.jl:18,80); `convolved` is the
T x N x B array of basis-filtered counts, kept on the device inside a DiscreteDataset handle.
"""
import ctypes as C
import time
import weakref

import numpy as np

from . import _lib
from ._lib import DomainError
from .components import DenseWeightModel
from .continuous import HawkesProcess


class DiscreteBaseline:
    pass


class DiscreteHomogeneousProcess(DiscreteBaseline):
    """DiscreteHomogeneousProcess(λ[, dt]) or (λ, α0, β0, αv, βv, dt) -- src/baselines.jl:358-382."""

    def __init__(self, λ, *args):
        λ = np.array(λ, dtype=np.float64)
        if len(args) <= 1:
            α0, β0, αv, βv, dt = 1.0, 1.0, np.ones_like(λ), np.ones_like(λ), (args[0] if args else 1.0)
        elif len(args) == 5:
            α0, β0, αv, βv, dt = args
        else:
            raise TypeError("DiscreteHomogeneousProcess(λ[, dt]) or (λ, α0, β0, αv, βv, dt)")
        αv, βv = np.array(αv, dtype=np.float64), np.array(βv, dtype=np.float64)
        if np.any(λ < 0):
            raise DomainError("DiscreteHomogeneousProcess: intensity parameter λ must be non-negative")
        if not α0 > 0:
            raise DomainError("DiscreteHomogeneousProcess: shape parameter α0 must be positive")
        if not β0 > 0:
            raise DomainError("DiscreteHomogeneousProcess: rate parameter β0 must be positive")
        if not np.all(αv > 0):
            raise DomainError("DiscreteHomogeneousProcess: shape parameter αv must be positive")
        if not np.all(βv > 0):
            raise DomainError("DiscreteHomogeneousProcess: rate parameter βv must be positive")
        if not dt > 0.0:
            raise DomainError("DiscreteHomogeneousProcess: time step dt must be non-negative")
        self.λ, self.α0, self.β0, self.αv, self.βv, self.dt = λ, float(α0), float(β0), αv, βv, float(dt)

    def ndims(self):
        return len(self.λ)

    def params(self):
        return self.λ.copy()

    def variational_params(self):
        return np.concatenate([self.αv, self.βv])

    def intensity(self, *args):
        """intensity(p, ts) -> len(ts) x N, or intensity(p, node, time) -- src/baselines.jl:402-411"""
        if len(args) == 1:
            ts = np.atleast_1d(np.asarray(args[0], dtype=np.float64))
            if np.any(ts < 0.0):
                raise DomainError("intensity: times ts must be non-negative")
            return np.tile(self.λ, (len(ts), 1)) * self.dt
        node, t = args
        if node < 1 or node > self.ndims():
            raise DomainError("intensity: node must be between one and ndims")
        if t < 0.0:
            raise DomainError("intensity: time must be non-negative")
        return self.λ[node - 1] * self.dt

    def sufficient_statistics(self, data):
        """src/baselines.jl:421-425 (pinned by test/baselines.jl:77-78)"""
        data = np.asarray(data)
        return data.sum(axis=1), data.shape[1]

    def integrated_intensity(self, *args):
        """src/baselines.jl:427-439"""
        if len(args) == 1:
            (duration,) = args
            if duration < 0.0:
                raise DomainError("intensity: duration must be non-negative")
            return self.λ * self.dt * duration
        node, duration = args
        if node < 1 or node > self.ndims():
            raise DomainError("intensity: node must be between one and ndims")
        if duration < 0.0:
            raise DomainError("intensity: duration must be non-negative")
        return self.λ[node - 1] * self.dt * duration

    def update_(self, data, parents):
        """The reference's argument check (src/baselines.jl:447, test/baselines.jl:88); the update
        itself is fused into the GPU VB step."""
        data, parents = np.asarray(data), np.asarray(parents)
        if data.shape != (parents.shape[1], parents.shape[0]):
            raise ValueError("update!: data and parent dimensions do not conform")
        N, T = data.shape
        self.αv = self.α0 + np.sum(parents[:, :, 0] * data.T, axis=0)
        self.βv = 1.0 / self.β0 + T * self.dt * np.ones(N)
        return self.αv.copy(), self.βv.copy()


class DiscreteLogGaussianCoxProcess(DiscreteBaseline):
    """DiscreteLogGaussianCoxProcess(x, λ, Σ | kernel, m, dt) -- src/baselines.jl:461-509: λ is G x N
    (λ[i, n] = λ_n(x[i])), intensity = linear interpolation of (x, λ[:, n]·dt)."""

    def __init__(self, x, λ, Σ, m, dt):
        from .components import Kernel
        x = np.asarray(x, dtype=np.float64)
        if isinstance(Σ, Kernel):
            if x[0] != 0.0:
                raise ValueError("Grid points must start at 0.")            # src/baselines.jl:493
            Σ = Σ(x)
        self.x = x
        self.λ = np.array(λ, dtype=np.float64)
        if self.λ.ndim != 2 or self.λ.shape[0] != len(x):
            raise ValueError("λ must be a (grid points) x (nodes) matrix")
        self.Σ = None if Σ is None else np.asarray(Σ, dtype=np.float64)
        self.m, self.dt = float(m), float(dt)

    @classmethod
    def from_gp(cls, gp, m, T, n, k, dt, rng):
        """DiscreteLogGaussianCoxProcess(gp, m, T, n, k, dt) -- src/baselines.jl:497-505"""
        if T % n != 0:
            raise ValueError("Duration must be divisible by number of steps.")
        x = np.linspace(0.0, T, n + 1)
        Σ = gp.cov(x)
        return cls(x, np.column_stack([np.exp(m + gp.rand(x, rng, sigma=Σ)) for _ in range(k)]), Σ, m, dt)

    def ndims(self):
        return self.λ.shape[1]

    def range(self):
        """range(p) = x[1] : dt : x[end] - dt  -- src/baselines.jl:509"""
        return self.x[0] + self.dt * np.arange(self.nsteps())

    def nsteps(self):
        return int(np.floor((self.x[-1] - self.dt - self.x[0]) / self.dt + 1e-9)) + 1

    def params(self):
        return self.λ.ravel(order="F").copy()

    def params_(self, x):
        """params!: src/baselines.jl:514-520"""
        if len(x) != self.λ.size:
            raise ValueError("Parameter vector length does not match model parameter length.")
        self.λ = np.asarray(x, dtype=np.float64).reshape(self.λ.shape, order="F").copy()

    def intensity(self, *args):
        """intensity(p, times) -> len(times) x N, intensity(p, node, times): src/baselines.jl:523-541
        (piecewise-linear through (x, λ[:, n]·dt); DomainError outside the grid, last value at x[end])"""
        if len(args) == 2:
            node, ts = args
            return self.intensity(ts)[:, node - 1] if np.ndim(ts) else self.intensity(np.array([ts]))[0, node - 1]
        ts = np.atleast_1d(np.asarray(args[0], dtype=np.float64))
        if np.any(ts < self.x[0]) or np.any(ts > self.x[-1]):
            raise DomainError("Value is outside interpolation support")
        return np.column_stack([np.interp(ts, self.x, self.λ[:, n] * self.dt) for n in range(self.ndims())])

    def integrated_intensity(self, duration=None):
        """src/baselines.jl:543"""
        return self.intensity(self.range()).sum(axis=0)

    def attach(self, ds):
        """Put intensity(p, 1:T) on the device with the dataset (calls then pass lambda0 = NULL)."""
        lam = np.asfortranarray(self.λ).ravel(order="K")
        _lib.check(_lib.lib().nhp_disc_set_lgcp_baseline(ds.ctx.h, ds.h, _lib.dptr(self.x), len(self.x), _lib.dptr(lam), self.dt),
                   ds.ctx.h)

    def candidate_loglikelihood(self, ds, Y):
        """loglikelihood(p, data, node, y) (src/baselines.jl:571-584) of latent curves Y [G, N], one per node, in
        one GPU call, on the baseline counts parents[:, :, 1] the latest parent sweep left on the device."""
        cand = np.asfortranarray(np.exp(self.m + np.asarray(Y, dtype=np.float64))).ravel(order="K")
        out = np.empty(self.ndims())
        _lib.check(_lib.lib().nhp_disc_lgcp_loglik(ds.ctx.h, ds.h, _lib.dptr(cand), self.dt, _lib.dptr(out)), ds.ctx.h)
        return out

    def resample_(self, ds, rng, max_attempts=100):
        """resample!(process, parents; sampler=elliptical_slice) -- src/baselines.jl:589-609,640-679; the N slice
        loops advance in lock step, one GPU likelihood call per round (as for the continuous LGCP)."""
        if self.Σ is None:
            raise ValueError("DiscreteLogGaussianCoxProcess needs Σ to be resampled")
        G, N = self.λ.shape
        L = np.linalg.cholesky(self.Σ)
        Y = np.log(self.λ) - self.m
        V = L @ rng.standard_normal((G, N))
        lly = self.candidate_loglikelihood(ds, Y) + np.log(rng.uniform(size=N))
        θ = 2 * np.pi * rng.uniform(size=N)
        θmin, θmax = θ - 2 * np.pi, θ.copy()
        Ynew = Y * np.cos(θ)[None, :] + V * np.sin(θ)[None, :]
        done = self.candidate_loglikelihood(ds, Ynew) >= lly
        attempts = 1
        while not done.all():
            if attempts >= max_attempts:
                raise RuntimeError("Elliptical slice sampling reached maximum attempts.")
            attempts += 1
            todo = ~done
            neg = θ < 0.0
            θmin = np.where(todo & neg, θ, θmin)
            θmax = np.where(todo & ~neg, θ, θmax)
            θ = np.where(todo, θmin + (θmax - θmin) * rng.uniform(size=N), θ)
            cand = Y * np.cos(θ)[None, :] + V * np.sin(θ)[None, :]
            Ynew = np.where(todo[None, :], cand, Ynew)
            done = done | (todo & (self.candidate_loglikelihood(ds, Ynew) >= lly))
        self.λ = np.exp(self.m + Ynew)
        return self.λ.copy()


class DiscreteImpulseResponse:
    pass


class DiscreteGaussianImpulseResponse(DiscreteImpulseResponse):
    """DiscreteGaussianImpulseResponse(θ, nlags[, dt]) -- src/impulses.jl:272-288; θ is N x N x B
    with Σ_b θ[p,c,·] = 1."""

    def __init__(self, θ, nlags, dt=1.0):
        θ = np.array(θ, dtype=np.float64, order="K")
        if not np.all(θ.sum(axis=2) == 1.0):
            raise ValueError("Invalid discrete basis parameter.")
        self.θ, self.γ, self.γv, self.nlags, self.dt = θ, 1.0, np.ones_like(θ), int(nlags), float(dt)
        self.ϕ = None

    def ndims(self):
        return self.θ.shape[0]

    def nbasis(self):
        return self.θ.shape[2]

    def params(self):
        return self.θ.ravel(order="F").copy()

    def variational_params(self):
        return self.γv.ravel(order="F").copy()

    def basis(self):
        """basis(impulse) -> L x B matrix (column b = ϕ_b) -- src/impulses.jl:321-335"""
        L, B = self.nlags, self.nbasis()
        phi = np.empty((B, L))
        _lib.check(_lib.lib().nhp_disc_basis(L, B, self.dt, _lib.dptr(phi)))
        return phi.T.copy()


class DiscreteDataset:
    """nhp_disc_dataset handle: the count matrix (uploaded once, transposed on the device) and,
    after convolve(), the T x N x B basis-filtered counts."""

    def __init__(self, ctx, data):
        data = np.asarray(data)
        if data.ndim != 2:
            raise ValueError("data must be an N x T matrix")
        self.N, self.T = data.shape
        self.ctx = ctx
        self.node_counts = data.sum(axis=1).astype(np.float64)    # node_counts(data): src/parents.jl:118-121
        d = np.asfortranarray(data.astype(np.int64, copy=False)).ravel(order="K")
        h = C.c_void_p()
        _lib.check(_lib.lib().nhp_disc_dataset_create(ctx.h, _lib.iptr(d), self.N, self.T, C.byref(h)), ctx.h)
        self.h = h
        self.B = 0
        self._fin = weakref.finalize(self, _lib.lib().nhp_disc_dataset_destroy, h)


class DiscreteHawkesProcess(HawkesProcess):
    def ndims(self):
        return self.baseline.ndims()

    def nlags(self):
        return self.impulses.nlags

    def _lowered(self):
        A = getattr(self, "adjacency_matrix", None)
        l0 = None if isinstance(self.baseline, DiscreteLogGaussianCoxProcess) else _lib.f64(self.baseline.λ)
        return (l0, _lib.colmajor(self.weights.W), _lib.colmajor(self.impulses.θ),
                None if A is None else _lib.colmajor(A))


class DiscreteStandardHawkesProcess(DiscreteHawkesProcess):
    """DiscreteStandardHawkesProcess(baseline, impulses, weights, dt) -- src/discrete.jl:161-170."""

    def __init__(self, baseline, impulses, weights, dt):
        if baseline.dt != dt or impulses.dt != dt:
            raise ValueError("Baseline and impulse response time step must match process time step.")
        self.baseline, self.impulses, self.weights, self.dt = baseline, impulses, weights, float(dt)

    def isstable(self):
        return np.max(np.abs(np.linalg.eigvals(self.weights.W))) < 1.0

    def params(self):
        """[λ0; vec(W .* θ)] -- src/discrete.jl:174-182"""
        return np.concatenate([self.baseline.params(), (self.weights.W[:, :, None] * self.impulses.θ).ravel(order="F")])

    def variational_params(self):
        """src/discrete.jl:204-209"""
        return np.concatenate([self.baseline.variational_params(), self.impulses.variational_params(),
                               self.weights.variational_params()])


class DiscreteNetworkHawkesProcess(DiscreteHawkesProcess):
    """DiscreteNetworkHawkesProcess(baseline, impulses, weights, adjacency_matrix, network, dt)
    -- src/discrete.jl:395-402."""

    def __init__(self, baseline, impulses, weights, adjacency_matrix, network, dt):
        self.baseline, self.impulses, self.weights = baseline, impulses, weights
        self.adjacency_matrix, self.network, self.dt = np.array(adjacency_matrix, dtype=np.float64), network, float(dt)

    def isstable(self):
        return np.max(np.abs(np.linalg.eigvals(self.adjacency_matrix * self.weights.W))) < 1.0

    def params(self):
        """[ρ; λ0; W; θ; vec(A)] -- src/discrete.jl:406-414"""
        return np.concatenate([self.network.params(), self.baseline.params(), self.weights.params(),
                               self.impulses.params(), self.adjacency_matrix.ravel(order="F")])


def convolve(process, data, ctx=None, fetch=False):
    """convolve(process, data) -- src/discrete.jl:146-151.  Returns a DiscreteDataset whose device
    copy holds Ŝ (T x N x B); with fetch=True also returns the array itself."""
    ctx = ctx or _lib.default_context()
    ds = data if isinstance(data, DiscreteDataset) else DiscreteDataset(ctx, data)
    phi = process.impulses.basis()
    L, B = phi.shape
    ph = np.asfortranarray(phi).ravel(order="K")
    out = np.empty(ds.T * ds.N * B) if fetch else None
    _lib.check(_lib.lib().nhp_disc_convolve(ctx.h, ds.h, _lib.dptr(ph), L, B, _lib.dptr(out)), ctx.h)
    ds.B = B
    if fetch:
        return ds, out.reshape((ds.T, ds.N, B), order="F")
    return ds


def _convolved(process, data, convolved, ctx):
    ds = convolved if convolved is not None else convolve(process, data, ctx)
    if isinstance(process.baseline, DiscreteLogGaussianCoxProcess):
        process.baseline.attach(ds)           # intensity(baseline, 1:T) follows the current grid values
    return ds


def disc_intensity(process, data=None, convolved=None, ctx=None):
    """intensity(process, convolved) / intensity(process, data) -> T x N -- src/discrete.jl:115-131"""
    ctx = ctx or _lib.default_context()
    ds = _convolved(process, data, convolved, ctx)
    l0, W, th, A = process._lowered()
    out = np.empty(ds.T * ds.N)
    _lib.check(_lib.lib().nhp_disc_intensity(ctx.h, ds.h, _lib.dptr(l0), _lib.dptr(W), _lib.dptr(th), _lib.dptr(A),
                                             process.dt, _lib.dptr(out)), ctx.h)
    return out.reshape((ds.T, ds.N), order="F")


def disc_loglikelihood(process, data=None, convolved=None, ctx=None):
    """loglikelihood(process, data[, convolved]) -- src/discrete.jl:86-102"""
    ctx = ctx or _lib.default_context()
    ds = _convolved(process, data, convolved, ctx)
    l0, W, th, A = process._lowered()
    ll = C.c_double()
    _lib.check(_lib.lib().nhp_disc_loglik(ctx.h, ds.h, _lib.dptr(l0), _lib.dptr(W), _lib.dptr(th), _lib.dptr(A),
                                          process.dt, C.byref(ll)), ctx.h)
    return ll.value


def disc_loglikelihood_gradient(process, data=None, convolved=None, ctx=None):
    """(ll, ∂ll/∂[λ0; vec(W .* θ)]) in one GPU call: the analytic gradient of mle!'s objective
    (src/discrete.jl:211-296 uses finite differences of loglikelihood, 2P calls per gradient)."""
    ctx = ctx or _lib.default_context()
    ds = _convolved(process, data, convolved, ctx)
    l0, W, th, _ = process._lowered()
    P = len(process.baseline.params()) + ds.N * ds.N * ds.B
    g = np.empty(P)
    ll = C.c_double()
    _lib.check(_lib.lib().nhp_disc_loglik_grad(ctx.h, ds.h, _lib.dptr(l0), _lib.dptr(W), _lib.dptr(th), process.dt,
                                               C.byref(ll), _lib.dptr(g), P), ctx.h)
    return ll.value, g


def disc_params_(process, x):
    """params!(process::DiscreteStandardHawkesProcess, x): x = [λ0; vec(W .* θ)], W = Σ_b η, θ = η ./ W
    -- src/discrete.jl:183-201"""
    N, B = process.ndims(), process.impulses.nbasis()
    nb = len(process.baseline.params())
    if len(x) != nb + N * N * B:
        raise ValueError("Parameter vector length does not match model parameter length.")
    η = np.asarray(x[nb:], dtype=np.float64).reshape((N, N, B), order="F")
    W = η.sum(axis=2)
    if isinstance(process.baseline, DiscreteLogGaussianCoxProcess):
        process.baseline.params_(x[:nb])
    else:
        process.baseline.λ = np.array(x[:nb], dtype=np.float64)
    process.weights.W = np.asfortranarray(W)
    process.impulses.θ = np.asfortranarray(η / W[:, :, None])
    return process.params()


def disc_mle_(process, data, optimizer="L-BFGS-B", verbose=False, f_abstol=1e-6, regularize=False, guess=None,
              seed=None, max_steps=1000, ctx=None):
    """mle!(process::DiscreteStandardHawkesProcess, data) -- src/discrete.jl:211-296: same objective
    (-loglikelihood(process, data, convolved)), same parameter vector [λ0; vec(W .* θ)], same box [1e-6, 10],
    same stable random start (:346-360) and |f - f_prev| < f_abstol stopping rule.  The reference runs
    Optim's Fminbox(BFGS) on finite differences; here scipy's L-BFGS-B gets the analytic gradient from the GPU
    (three fp64-MFMA GEMMs per objective + gradient)."""
    import time
    from scipy import optimize
    from .inference import MaximumLikelihood
    if not isinstance(process, DiscreteStandardHawkesProcess):
        raise TypeError("mle! is defined for DiscreteStandardHawkesProcess (src/discrete.jl:211)")
    if regularize:
        raise NotImplementedError("logprior(::DiscreteStandardHawkesProcess) reads fields that do not exist "
                                  "(src/discrete.jl:316-322, SURVEY D5)")
    ctx = ctx or _lib.default_context()
    ds = convolve(process, data, ctx)
    N = process.ndims()
    rng = np.random.default_rng(seed)
    if guess is None:                                    # _rand_init_: src/discrete.jl:346-360
        for _ in range(100):
            x0 = rng.uniform(size=len(process.params()))
            nb = len(process.baseline.params())
            x0[nb:] /= 10
            W0 = x0[nb:].reshape((N, N, -1), order="F").sum(axis=2)
            if np.max(np.abs(np.linalg.eigvals(W0))) < 1.0:
                break
        else:
            raise RuntimeError("Random initialization reached max attempts.")
    else:
        x0 = np.array(guess, dtype=np.float64)
    lower, upper = 1e-6, 1e1
    state = {"minloss": np.inf, "steps": 0, "converged": False, "last": None}
    start = time.time()

    if optimizer in ("device", "LBFGS-device"):
        # the optimizer's state on the device (nhp_disc_mle_run: projected L-BFGS in HBM; params!'s split of x into W and θ
        # redone on the device per evaluation): no parameter upload, gradient download or host-side update per objective call
        if isinstance(process.baseline, DiscreteLogGaussianCoxProcess):
            raise NotImplementedError("optimizer='device' takes the homogeneous baseline; use the host optimizer with an LGCP baseline")
        x = np.ascontiguousarray(np.clip(x0, lower, upper), dtype=np.float64)
        loss, steps, conv, evals = C.c_double(), C.c_int32(), C.c_int32(), C.c_int32()
        _lib.check(_lib.lib().nhp_disc_mle_run(ctx.h, ds.h, process.dt, lower, upper, float(f_abstol), int(max_steps), _lib.dptr(x), len(x),
                                               C.byref(loss), C.byref(steps), C.byref(conv), C.byref(evals)), ctx.h)
        if verbose:
            print(f" > steps: {steps.value}, objective evaluations: {evals.value}, loss: {loss.value}, elapsed: {time.time() - start}")
        disc_params_(process, x)
        res = MaximumLikelihood(x.copy(), -float(loss.value), int(steps.value), time.time() - start, "success" if conv.value else "failure")
        res.evaluations = int(evals.value)
        return res

    def fg(x):
        disc_params_(process, x)
        ll, g = disc_loglikelihood_gradient(process, convolved=ds, ctx=ctx)
        state["last"] = -ll
        return -ll, -g

    def status_update(xk):
        state["steps"] += 1
        value = state["last"]
        if verbose:
            print(f" > step: {state['steps']}, loss: {value}, elapsed: {time.time() - start}")
        if abs(value - state["minloss"]) < f_abstol:
            state["converged"] = True
            raise StopIteration
        state["minloss"] = value

    options = {"maxiter": max_steps}
    if optimizer == "L-BFGS-B":       # scipy's own relative-decrease test off, as in inference.mle_ (Optim's g_tol = 1e-8 kept)
        options.update(ftol=0.0, gtol=1e-8, maxfun=20 * max_steps + 1000)
    res = optimize.minimize(fg, np.clip(x0, lower, upper), jac=True, method=optimizer,
                            bounds=[(lower, upper)] * len(x0), callback=status_update, options=options)
    disc_params_(process, res.x)
    return MaximumLikelihood(res.x.copy(), -float(res.fun), state["steps"], time.time() - start,
                             "success" if (state["converged"] or res.success) else "failure")


def resample_parent_counts(process, data=None, convolved=None, seed=0, step=0, ctx=None):
    """Σ_t resample_parents(process, data, convolved)[t, :, :] -> N x (1 + N·B) integer counts
    (src/parents.jl:82-116): column 0 the baseline, column 1 + p·B + b parent node p through basis b
    (0-based p, b) -- the reduction every discrete resample! applies to the T x N x (1+NB) array, which is
    never materialised here.  Draws are keyed (seed, step), reproducible, and equal to the oracle."""
    ctx = ctx or _lib.default_context()
    ds = _convolved(process, data, convolved, ctx)
    l0, W, th, A = process._lowered()
    N, B = ds.N, ds.B
    out = np.empty(N * (1 + N * B), dtype=np.int64)
    _lib.check(_lib.lib().nhp_disc_resample_parents(ctx.h, ds.h, _lib.dptr(l0), _lib.dptr(W), _lib.dptr(th), _lib.dptr(A),
                                                    process.dt, seed, step, _lib.iptr(out)), ctx.h)
    return out.reshape((N, 1 + N * B), order="F")


def disc_parent_counts(counts, ndims, nbasis):
    """parent_counts(parents, ndims, nbasis) on the time-reduced array -- src/parents.jl:123-134"""
    return counts[:, 1:].reshape((ndims, ndims, nbasis)).sum(axis=2).T.astype(np.float64)      # [parent, child]


def disc_resample_adjacency_matrix_(process, data=None, convolved=None, u=None, seed=0, step=0, ctx=None):
    """resample_adjacency_matrix!(process, data, convolved) -- src/discrete.jl:424-480: one Gibbs sweep of
    the adjacency matrix on the GPU.  `u` (N x N, [parent, child]) supplies the Bernoulli uniforms
    explicitly; otherwise Philox keyed (seed, step).  Updates process.adjacency_matrix, returns ΣA."""
    from .components import BernoulliNetworkModel, DenseNetworkModel
    ctx = ctx or _lib.default_context()
    ds = _convolved(process, data, convolved, ctx)
    l0, W, th, _ = process._lowered()
    N = ds.N
    if isinstance(process.network, BernoulliNetworkModel):
        scalar, rho_m = float(process.network.ρ), None
    elif isinstance(process.network, DenseNetworkModel):
        scalar, rho_m = 1.0, None
    else:
        scalar, rho_m = 0.5, _lib.colmajor(np.asarray(process.network.link_probability(), dtype=np.float64))
    A = _lib.colmajor(process.adjacency_matrix).copy()
    uu = None if u is None else _lib.colmajor(np.asarray(u, dtype=np.float64))
    nl = C.c_double()
    _lib.check(_lib.lib().nhp_disc_resample_adjacency(ctx.h, ds.h, _lib.dptr(l0), _lib.dptr(W), _lib.dptr(th), _lib.dptr(A),
                                                      process.dt, _lib.dptr(rho_m), scalar, _lib.dptr(uu), seed, step,
                                                      C.byref(nl)), ctx.h)
    process.adjacency_matrix = A.reshape((N, N), order="F")
    return nl.value


def disc_resample_(process, data, convolved, rng, seed=0, step=0, ctx=None, device_draws=True):
    """resample!(process::DiscreteStandardHawkesProcess, data, convolved) -- src/discrete.jl:362-368, and the
    network twin :416-424 (adds the adjacency sweep and the network's ρ).

    Parent counts come from the GPU; the conjugate draws are numpy (statistical, not bitwise, parity with
    Julia's samplers).  The reference's baseline update is broken for the homogeneous process (it passes
    the T x N slice to a helper that expects N x T: SURVEY D2); the intended update is applied:
    λ ~ Gamma(α0 + Σ_t parents[t, c, 1], 1 / (β0 + T·dt))  (src/baselines.jl:413-419)."""
    if not isinstance(process.weights, DenseWeightModel):
        raise NotImplementedError("discrete Gibbs is built for DenseWeightModel (SparseWeightModel: SURVEY 2.1)")
    ctx = ctx or _lib.default_context()
    ds = _convolved(process, data, convolved, ctx)
    N, B = ds.N, ds.B
    b, w, imp = process.baseline, process.weights, process.impulses
    if device_draws and isinstance(b, DiscreteHomogeneousProcess):
        # parents and conjugate draws in one GPU call (nhp_disc_gibbs_step): only the new parameters come back
        l0, W, th, A = process._lowered()
        l0, W, th = l0.copy(), W.copy(), th.copy()
        _lib.check(_lib.lib().nhp_disc_gibbs_step(ctx.h, ds.h, _lib.dptr(l0), _lib.dptr(W), _lib.dptr(th), _lib.dptr(A), process.dt,
                                                  b.α0, b.β0, w.κ, w.ν, imp.γ, seed, step), ctx.h)
        b.λ, w.W, imp.θ = l0, W.reshape((N, N), order="F"), th.reshape((N, N, B), order="F")
        if isinstance(process, DiscreteNetworkHawkesProcess):
            links = disc_resample_adjacency_matrix_(process, convolved=ds, seed=seed, step=step, ctx=ctx)
            process.network.resample_links_(links, N * N, rng)
        return process.params()
    counts = resample_parent_counts(process, convolved=ds, seed=seed, step=step, ctx=ctx)
    if isinstance(b, DiscreteLogGaussianCoxProcess):
        b.resample_(ds, rng)              # elliptical slice on parents[:, :, 1], left on the device by the sweep above
    else:
        b.λ = rng.gamma(b.α0 + counts[:, 0], 1.0 / (b.β0 + ds.T * b.dt))
    Mnm = disc_parent_counts(counts, N, B)
    w.W = rng.gamma(w.κ + Mnm, 1.0 / (w.ν + ds.node_counts)[:, None] * np.ones((N, N)))        # src/weights.jl:59-64
    γ = imp.γ + counts[:, 1:].reshape((N, N, B)).transpose(1, 0, 2)                                  # [parent, child, basis]
    g = rng.gamma(γ, 1.0)
    imp.θ = g / g.sum(axis=2, keepdims=True)                                                         # Dirichlet: src/impulses.jl:337-353
    if isinstance(process, DiscreteNetworkHawkesProcess):
        links = disc_resample_adjacency_matrix_(process, convolved=ds, seed=seed, step=step, ctx=ctx)
        process.network.resample_links_(links, N * N, rng)
    return process.params()


def disc_mcmc_(process, data, nsteps=1000, log_freq=100, verbose=False, seed=0, ctx=None, device_draws=True):
    """mcmc!(process::DiscreteHawkesProcess, data) -- src/inference.jl:49-70: convolve once, then
    resample!(process, data, convolved) per step."""
    import time
    from .inference import MarkovChainMonteCarlo
    ctx = ctx or _lib.default_context()
    ds = convolve(process, data, ctx)
    rng = np.random.default_rng(seed)
    res = MarkovChainMonteCarlo()
    start = time.time()
    while res.steps < nsteps:
        res.samples.append(disc_resample_(process, None, ds, rng, seed=seed, step=res.steps, ctx=ctx, device_draws=device_draws))
        res.steps += 1
        if res.steps % log_freq == 0 and verbose:
            res.elapsed = time.time() - start
            print(f" > step: {res.steps}, elapsed: {res.elapsed}")
    res.elapsed = time.time() - start
    res.status = "complete"
    return res


def update_(process, data, convolved, ctx=None, n_steps=1):
    """update!(process, data, convolved) -- src/discrete.jl:369-375: one mean-field step (or n_steps
    of them with the parameters resident on the device in between); the variational parameters of
    baseline, weights and impulses are overwritten in place."""
    from .components import SparseWeightModel
    if (not isinstance(process, DiscreteStandardHawkesProcess) or not isinstance(process.weights, DenseWeightModel)
            or isinstance(process.weights, SparseWeightModel)):
        raise NotImplementedError("VB exists only for DiscreteStandardHawkesProcess + DenseWeightModel "
                                  "(the reference's network / sparse variants are broken: SURVEY D6)")
    if not isinstance(process.baseline, DiscreteHomogeneousProcess):
        raise NotImplementedError("update! is defined for DiscreteHomogeneousProcess baselines only (src/baselines.jl:444-456)")
    ctx = ctx or _lib.default_context()
    ds = _convolved(process, data, convolved, ctx)
    b, w, imp = process.baseline, process.weights, process.impulses
    N, B = process.ndims(), imp.nbasis()
    av, bv = _lib.f64(b.αv).copy(), _lib.f64(b.βv).copy()
    kv, nv, gv = _lib.colmajor(w.κv).copy(), _lib.colmajor(w.νv).copy(), _lib.colmajor(imp.γv).copy()
    _lib.check(_lib.lib().nhp_disc_vb_run(ctx.h, ds.h, process.dt, b.α0, b.β0, w.κ, w.ν, imp.γ, n_steps,
                                          _lib.dptr(av), _lib.dptr(bv), _lib.dptr(kv), _lib.dptr(nv), _lib.dptr(gv)),
               ctx.h)
    b.αv, b.βv = av, bv
    w.κv, w.νv = kv.reshape((N, N), order="F"), nv.reshape((N, N), order="F")
    imp.γv = gv.reshape((N, N, B), order="F")
    return process.variational_params()


class VariationalInference:
    """src/inference.jl:78-92"""

    def __init__(self):
        self.trace, self.step, self.elapsed, self.status = [], 0, 0.0, "incomplete"

    def __repr__(self):
        return f"\n* Status: {self.status}\n    step: {self.step}\n    elapsed: {self.elapsed}"


def vb_(process, data, max_steps=1000, Δx_thresh=1e-6, Δq_thresh=1e-2, verbose=False, keep_trace=True, ctx=None):
    """vb!(process, data; max_steps, Δx_thresh, Δq_thresh, verbose) -- src/inference.jl:153-181.
    Like the reference (whose convergence test is commented out, :163-176) it runs max_steps updates.
    keep_trace=False runs them back to back on the device and records only the final parameters
    (the reference's per-step trace is 2N + N²B + 2N² doubles a step)."""
    ctx = ctx or _lib.default_context()
    start = time.time()
    convolved = convolve(process, data, ctx)
    res = VariationalInference()
    if not keep_trace and max_steps > 0:
        update_(process, data, convolved, ctx, n_steps=max_steps)
        res.trace.append(process.variational_params())
        res.step = max_steps
    while res.step < max_steps:
        update_(process, data, convolved, ctx)
        res.trace.append(process.variational_params())
        res.step += 1
    res.elapsed = time.time() - start
    if verbose:
        print(" ** maximum steps reached **")
    return res
