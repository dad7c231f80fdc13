"""Plug-in components: the reference's Baseline / ImpulseResponse / Weights / Network types
with the same names, fields and constructor defaults, as plain parameter holders.

The reference dispatches a scalar method per (parent, child) pair on these types inside its
innermost loops; here they only carry arrays, which `lower()` in
continuous.py turns into the nhp_cont_model_desc blob the HIP kernels read.  Field names
keep the reference's Unicode spelling (θ, λ, Δtmax ...) so code written against the Julia
package reads the same.  Matrices are indexed [parent, child] (src/continuous.jl:303).

Host-side random draws (Gibbs conjugate updates, src/baselines.jl:72-77,
src/weights.jl:59-64, src/impulses.jl:68-73,204-214, src/networks.jl:65-78) use numpy's
Generator: statistically, not bitwise, the same as Julia's samplers.
"""
import numpy as np

from ._lib import DomainError


def _fillna(x, value):
    """fillna!(X, value): src/utils/helpers.jl:18-25"""
    x = np.array(x, dtype=np.float64)
    x[np.isnan(x)] = value
    return x


# ------------------------------------------------------------------------------ baselines
class Baseline:
    pass


class HomogeneousProcess(Baseline):
    """HomogeneousProcess(λ[, α0, β0]) -- src/baselines.jl:27-39."""

    def __init__(self, λ, α0=1.0, β0=1.0):
        λ = np.array(λ, dtype=np.float64)
        if np.any(λ < 0):
            raise DomainError("HomogeneousProcess: intensity parameter λ must be non-negative")
        if not α0 > 0:
            raise DomainError("HomogeneousProcess: shape parameter α0 must be positive")
        if not β0 > 0:
            raise DomainError("HomogeneousProcess: rate parameter β0 must be positive")
        self.λ, self.α0, self.β0 = λ, float(α0), float(β0)

    def ndims(self):
        return len(self.λ)

    def params(self):
        return self.λ.copy()

    def params_(self, x):
        """params!: src/baselines.jl:44-50"""
        if len(x) != len(self.λ):
            raise ValueError("Parameter vector length does not match model parameter length.")
        self.λ[:] = x

    def intensity(self, node=None, time=0.0):
        """src/baselines.jl:110-118"""
        if time < 0:
            raise DomainError("time must be non-negative")
        return self.λ.copy() if node is None else self.λ[node - 1]

    def integrated_intensity(self, duration, node=None):
        """src/baselines.jl:98-108"""
        if duration < 0:
            raise DomainError("duration must be non-negative")
        return self.λ * duration if node is None else self.λ[node - 1] * duration

    def resample_(self, cnt0, duration, rng):
        """resample!: λ ~ Gamma(α0 + counts, 1/(β0 + T)) -- src/baselines.jl:72-77"""
        self.λ = rng.gamma(self.α0 + cnt0, 1.0 / (self.β0 + duration))
        return self.λ


# ---- Gaussian-process plumbing of the LGCP baseline (src/utils/gaussian.jl, src/utils/helpers.jl:1-11)
def posdef_(sigma, maxiter=3):
    """posdef!: shift the diagonal until the smallest eigenvalue is positive -- src/utils/helpers.jl:1-11"""
    sigma = np.array(sigma, dtype=np.float64)
    for _ in range(maxiter):
        eps = 2 * np.min(np.linalg.eigvalsh(sigma))
        if eps > 0.0:
            return sigma
        sigma[np.diag_indices_from(sigma)] -= eps
    print("WARNING: failed to make sigma positive definite")
    return sigma


class Kernel:
    def __call__(self, x, y=None):
        """kernel(x, y) on scalars; kernel(x::Vector) -> posdef!(Σ) -- src/utils/gaussian.jl:8-17"""
        if y is not None:
            return self.k(x, y)
        x = np.asarray(x, dtype=np.float64)
        return posdef_(self.k(x[:, None], x[None, :]))


class SquaredExponentialKernel(Kernel):
    """σ² exp(-((x-y)/η)²/2) -- src/utils/gaussian.jl:19-24"""

    def __init__(self, σ, η):
        self.σ, self.η = σ, η

    def k(self, x, y):
        return self.σ ** 2 * np.exp(-((x - y) / self.η) ** 2 / 2)


class OrnsteinUhlenbeckKernel(Kernel):
    """σ² exp(-|x-y|/η) -- src/utils/gaussian.jl:26-31"""

    def __init__(self, σ, η):
        self.σ, self.η = σ, η

    def k(self, x, y):
        return self.σ ** 2 * np.exp(-np.abs(x - y) / self.η)


class PeriodicKernel(Kernel):
    """σ² exp(-2 (sin(π|x-y|/θ)/η)²) -- src/utils/gaussian.jl:33-39"""

    def __init__(self, σ, η, θ):
        self.σ, self.η, self.θ = σ, η, θ

    def k(self, x, y):
        return self.σ ** 2 * np.exp(-2 * (np.sin(np.pi * np.abs(x - y) / self.θ) / self.η) ** 2)


class GaussianProcess:
    """GaussianProcess(kernel) / GaussianProcess(mu, kernel) -- src/utils/gaussian.jl:60-83"""

    def __init__(self, *args):
        self.mu, self.kernel = (np.zeros_like, args[0]) if len(args) == 1 else args

    def cov(self, x):
        return self.kernel(np.asarray(x, dtype=np.float64))

    def rand(self, x, rng, sigma=None):
        x = np.asarray(x, dtype=np.float64)
        sigma = self.cov(x) if sigma is None else sigma
        return np.asarray(self.mu(x), dtype=np.float64) + np.linalg.cholesky(sigma) @ rng.standard_normal(len(x))


def split_extract(data, parents, nnodes):
    """Events attributed to the baseline (parent node 0), split by node -- src/baselines.jl:227-238."""
    events, nodes, duration = data
    events, nodes = np.asarray(events, dtype=np.float64), np.asarray(nodes, dtype=np.int64)
    parentnodes = np.asarray(parents[1], dtype=np.int64)
    out = []
    for node in range(1, nnodes + 1):
        idx = np.flatnonzero((nodes == node) & (parentnodes == 0)) if len(nodes) else np.array([], dtype=np.int64)
        out.append((events[idx], nodes[idx], duration))
    return out


class LogGaussianCoxProcess(Baseline):
    """LogGaussianCoxProcess(x, λ, Σ | kernel, m) -- src/baselines.jl:148-173: piecewise-linear
    intensity through (x, λ[k]) per node (src/utils/interpolation.jl), λ = exp(m + y), y ~ N(0, Σ)."""

    def __init__(self, x, λ, Σ=None, m=0.0):
        x = np.asarray(x, dtype=np.float64)
        if x[0] != 0.0:
            raise DomainError("Grid points x must start at 0.")
        self.x = x
        self.λ = [np.asarray(v, dtype=np.float64) for v in λ]
        for v in self.λ:
            if len(v) != len(x):
                raise ValueError("intensity vectors must match the grid")
        self.Σ = Σ(x) if isinstance(Σ, Kernel) else (None if Σ is None else np.asarray(Σ, dtype=np.float64))
        self.m = float(m)

    @classmethod
    def from_gp(cls, gp, m, T, n, k, rng):
        """LogGaussianCoxProcess(gp, m, T, n, k): random intensities on n+1 grid points -- src/baselines.jl:164-171"""
        x = np.linspace(0.0, T, n + 1)
        Σ = gp.cov(x)
        return cls(x, [np.exp(m + gp.rand(x, rng, sigma=Σ)) for _ in range(k)], Σ, m)

    def ndims(self):
        return len(self.λ)

    def length(self):
        """length(process) = x[end], the longest duration the process supports -- src/baselines.jl:173"""
        return float(self.x[-1])

    def params(self):
        return np.concatenate(self.λ)

    def params_(self, x):
        """params!: src/baselines.jl:175-188"""
        if len(x) != sum(len(v) for v in self.λ):
            raise ValueError("Parameter vector length does not match model parameter length.")
        G = len(self.x)
        self.λ = [np.array(x[k * G:(k + 1) * G], dtype=np.float64) for k in range(len(self.λ))]

    def integrated_intensity(self, duration=None):
        """integrate.(LinearInterpolator) -- src/baselines.jl:336; ignores duration"""
        dx = np.diff(self.x)
        return np.array([np.sum(0.5 * (y[:-1] + y[1:]) * dx) for y in self.λ])

    def candidate_loglikelihood(self, ds, Y, parentnodes=None):
        """loglikelihood(process, data, node, y) (src/baselines.jl:247-254) of latent curves Y [N, G],
        one per node, in a single GPU call.  The baseline-attributed events are those the latest
        parent sweep left on the device with `ds`, unless `parentnodes` is given."""
        from . import _lib
        lam = _lib.f64(np.exp(self.m + np.asarray(Y, dtype=np.float64)).ravel())
        pn = None if parentnodes is None else np.ascontiguousarray(parentnodes, dtype=np.int64)
        out = np.empty(self.ndims())
        _lib.check(_lib.lib().nhp_cont_lgcp_loglik(ds.ctx.h, ds.h, _lib.iptr(pn), _lib.dptr(self.x), len(self.x),
                                                   _lib.dptr(lam), _lib.dptr(out)), ds.ctx.h)
        return out

    def resample_(self, ds, rng, parentnodes=None, max_attempts=100):
        """resample!(process, data, parents; sampler=elliptical_slice) -- src/baselines.jl:212-245,287-326
        (Murray et al. 2010).  The reference runs one slice loop per node; the loops are independent,
        so they advance in lock step here and each round scores the pending candidates of all nodes
        with one GPU call."""
        if self.Σ is None:
            raise ValueError("LogGaussianCoxProcess needs Σ to be resampled")
        N, G = self.ndims(), len(self.x)
        L = np.linalg.cholesky(self.Σ)
        Y = np.log(np.vstack(self.λ)) - self.m                      # init_y: :241
        V = rng.standard_normal((N, G)) @ L.T                       # v ~ N(0, Σ)
        lly = self.candidate_loglikelihood(ds, Y, parentnodes) + np.log(rng.uniform(size=N))
        θ = 2 * np.pi * rng.uniform(size=N)
        θmin, θmax = θ - 2 * np.pi, θ.copy()
        Ynew = Y * np.cos(θ)[:, None] + V * np.sin(θ)[:, None]
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
            cand = Y * np.cos(θ)[:, None] + V * np.sin(θ)[:, None]
            Ynew = np.where(todo[:, None], cand, Ynew)              # accepted nodes keep their draw
            done = done | (todo & (self.candidate_loglikelihood(ds, Ynew) >= lly))
        self.λ = [np.exp(self.m + Ynew[k]) for k in range(N)]       # resample_node!: :242-244
        return [v.copy() for v in self.λ]


# ------------------------------------------------------------------------------ impulses
class ImpulseResponse:
    pass


class ExponentialImpulseResponse(ImpulseResponse):
    """ExponentialImpulseResponse(θ[, α, β, Δtmax]); 1-arg default Δtmax = Inf -- src/impulses.jl:30-37."""

    def __init__(self, θ, α=1.0, β=1.0, Δtmax=np.inf):
        self.θ = np.array(θ, dtype=np.float64)
        self.α, self.β, self.Δtmax = float(α), float(β), float(Δtmax)

    def size(self):
        return self.θ.shape[0]

    def params(self):
        return self.θ.ravel(order="F").copy()

    def params_(self, x):
        """params!: src/impulses.jl:43-51"""
        if len(x) != self.θ.size:
            raise ValueError("Parameter vector length does not match model parameter length.")
        n = self.size()
        self.θ = np.asarray(x, dtype=np.float64).reshape((n, n), order="F").copy(order="F")   # (column-major like the vector: a plain copy)

    def resample_(self, Mnm, Xnm, rng):
        """resample!: θ ~ Gamma(α + Mnm, 1/(β + Mnm·Xnm)) -- src/impulses.jl:68-73"""
        self.θ = rng.gamma(self.α + Mnm, 1.0 / (self.β + Mnm * Xnm))
        return self.θ


class LogitNormalImpulseResponse(ImpulseResponse):
    """LogitNormalImpulseResponse(μ, τ, [μμ, κμ, α0, β0,] Δtmax) -- src/impulses.jl:138-148."""

    def __init__(self, μ, τ, *args):
        if len(args) == 1:
            μμ, κμ, α0, β0, Δtmax = 1.0, 1.0, 1.0, 1.0, args[0]
        elif len(args) == 5:
            μμ, κμ, α0, β0, Δtmax = args
        else:
            raise TypeError("LogitNormalImpulseResponse(μ, τ, Δtmax) or (μ, τ, μμ, κμ, α0, β0, Δtmax)")
        self.μ = np.array(μ, dtype=np.float64)
        self.τ = np.array(τ, dtype=np.float64)
        self.μμ, self.κμ, self.α0, self.β0, self.Δtmax = float(μμ), float(κμ), float(α0), float(β0), float(Δtmax)

    def size(self):
        return self.μ.shape[0]

    def params(self):
        return np.concatenate([self.μ.ravel(order="F"), self.τ.ravel(order="F")])

    def params_(self, x):
        """params!: src/impulses.jl:154-162"""
        if len(x) != self.μ.size + self.τ.size:
            raise ValueError("Parameter vector length does not match model parameter length.")
        n = self.size()
        x = np.asarray(x, dtype=np.float64)
        self.μ = x[: n * n].reshape((n, n), order="F").copy(order="F")
        self.τ = x[n * n:].reshape((n, n), order="F").copy(order="F")

    def resample_(self, Mnm, Xnm, Vnm, rng):
        """resample!: normal-gamma conjugate draw -- src/impulses.jl:204-214"""
        with np.errstate(invalid="ignore", divide="ignore"):
            αnm = self.α0 + Mnm / 2
            βnm = _fillna(Vnm / 2 + Mnm * self.κμ / (Mnm + self.κμ) * (Xnm - self.μμ) ** 2 / 2, self.β0)
            self.τ = rng.gamma(αnm, 1.0 / βnm)
            κnm = self.κμ + Mnm
            μnm = _fillna((self.κμ * self.μμ + Mnm * Xnm) / (self.κμ + Mnm), self.μμ)
            σ = (1.0 / (κnm * self.τ)) ** 0.5
            self.μ = rng.normal(μnm, σ)
        return self.μ.copy(), self.τ.copy()


# ------------------------------------------------------------------------------ weights
class Weights:
    pass


class DenseWeightModel(Weights):
    """DenseWeightModel(W[, κ, ν, κv, νv]) -- src/weights.jl:47-55."""

    def __init__(self, W, κ=1.0, ν=1.0, κv=None, νv=None):
        self.W = np.array(W, dtype=np.float64, order="K")
        self.κ, self.ν = float(κ), float(ν)
        self.κv = np.ones_like(self.W) if κv is None else np.array(κv, dtype=np.float64)
        self.νv = np.ones_like(self.W) if νv is None else np.array(νv, dtype=np.float64)

    def size(self):
        return self.W.shape[0]

    def params(self):
        return self.W.ravel(order="F").copy()

    def params_(self, x):
        """params!: src/weights.jl:9-15"""
        if len(x) != self.W.size:
            raise ValueError("Parameter vector length does not match model parameter length.")
        self.W = np.asarray(x, dtype=np.float64).reshape(self.W.shape, order="F").copy(order="F")

    def resample_(self, Mn, Mnm, rng):
        """resample!: W ~ Gamma(κ + Mnm, 1/(ν + Mn[p])) -- src/weights.jl:59-64"""
        self.W = rng.gamma(self.κ + Mnm, 1.0 / (self.ν + Mn)[:, None] * np.ones_like(Mnm))
        return self.W

    def variational_params(self):
        return np.concatenate([self.κv.ravel(order="F"), self.νv.ravel(order="F")])


class SparseWeightModel(DenseWeightModel):
    """SparseWeightModel(W) -- src/weights.jl:104-127: the weights of a network process, with separate Gamma priors
    for absent (κ0, ν0) and present (κ1, ν1) links.  Its Gibbs update (src/weights.jl:133-139) is the dense one under
    the present-link prior -- W ~ Gamma(κ1 + Mnm, 1/(ν1 + Mn[p])) -- so every kernel sees it as a DenseWeightModel with
    (κ, ν) = (κ1, ν1).  The variational methods of the reference are broken (undefined `p`, `ν1`, `ρ`, field-name
    mismatches: src/weights.jl:141-173) and raise here."""

    def __init__(self, W, κ0=1.0, ν0=1.0, κ1=1.0, ν1=1.0):
        super().__init__(W, κ1, ν1)
        self.κ0, self.ν0, self.κ1, self.ν1 = float(κ0), float(ν0), float(κ1), float(ν1)
        self.κv0, self.νv0 = np.ones_like(self.W), np.ones_like(self.W)
        self.κv1, self.νv1 = np.ones_like(self.W), np.ones_like(self.W)

    def variational_params(self):
        """src/weights.jl:131"""
        return np.concatenate([v.ravel(order="F") for v in (self.κv0, self.νv0, self.κv1, self.νv1)])


# ------------------------------------------------------------------------------ networks
class Network:
    pass


class DenseNetworkModel(Network):
    """src/networks.jl:13-31: every link present."""

    def __init__(self, nnodes):
        self.nnodes = int(nnodes)

    def params(self):
        return np.array([])

    def link_probability(self):
        return np.ones((self.nnodes, self.nnodes))

    def rand(self, rng):
        return np.ones((self.nnodes, self.nnodes))

    def resample_(self, A, rng):
        return None

    def resample_links_(self, nlinks, size, rng):
        return None


class BernoulliNetworkModel(Network):
    """BernoulliNetworkModel(ρ, N) with Beta(α, β) prior -- src/networks.jl:34-78."""

    def __init__(self, ρ, nnodes, α=1.0, β=1.0):
        self.ρ, self.α, self.β, self.nnodes = float(ρ), float(α), float(β), int(nnodes)

    def params(self):
        return np.array([self.ρ])

    def link_probability(self):
        return self.ρ * np.ones((self.nnodes, self.nnodes))

    def rand(self, rng):
        return (rng.uniform(size=(self.nnodes, self.nnodes)) < self.ρ).astype(np.float64)

    def resample_(self, A, rng):
        """resample!: ρ ~ Beta(α + ΣA, β + N² - ΣA) -- src/networks.jl:70-78"""
        return self.resample_links_(float(np.sum(A)), A.size, rng)

    def resample_links_(self, nlinks, size, rng):
        self.ρ = rng.beta(self.α + nlinks, self.β + size - nlinks)
        return self.ρ
