"""Inference drivers: host mirror of src/inference.jl and of the mle! / resample! bodies in
src/continuous.jl:144-208,350-358.  Every O(M·K̄) or O(M·N) loop runs on the GPU; what stays
on the host is O(N²) bookkeeping and the conjugate random draws (out of scope).

Result containers keep the reference's field names (src/inference.jl:6-12,24-29,78-83).
"""
import time

import numpy as np
from scipy import optimize
from scipy.special import gammaln

from . import _lib
from .components import (ExponentialImpulseResponse, HomogeneousProcess, LogitNormalImpulseResponse)
from .continuous import (ContinuousNetworkHawkesProcess, ContinuousStandardHawkesProcess,
                         device_dataset, loglikelihood, loglikelihood_gradient)
from .parents import resample_parents


class MaximumLikelihood:
    """src/inference.jl:6-12"""

    def __init__(self, maximizer, maximum, steps, elapsed, status):
        self.maximizer, self.maximum, self.steps, self.elapsed, self.status = maximizer, maximum, steps, elapsed, status

    def __repr__(self):
        return f"\n* Status: {self.status}\n    steps: {self.steps}\n    elapsed: {self.elapsed}\n    loss: {self.maximum}"


class MarkovChainMonteCarlo:
    """src/inference.jl:24-37"""

    def __init__(self):
        self.samples, self.steps, self.elapsed, self.status = [], 0, 0.0, "incomplete"

    def __repr__(self):
        return f"\n* Status: complete\n    steps: {self.steps}\n    elapsed: {self.elapsed}"


# ---- log priors and their gradients (host, O(N²)) ---------------------------------------------
def _gamma_logpdf(x, shape, rate):
    return shape * np.log(rate) - gammaln(shape) + (shape - 1.0) * np.log(x) - rate * x


def logprior(process):
    """logprior(process) -- src/continuous.jl:278-284: Gamma(α0, 1/β0) on λ (src/baselines.jl:120-122),
    Gamma(κ, 1/ν) on W (src/weights.jl:66-68), Gamma(α, 1/β) on θ (src/impulses.jl:110-112) or
    normal-gamma on (μ, τ) (src/impulses.jl:254-259)."""
    b, w, imp = process.baseline, process.weights, process.impulses
    lp = np.sum(_gamma_logpdf(b.λ, b.α0, b.β0)) + np.sum(_gamma_logpdf(w.W, w.κ, w.ν))
    if isinstance(imp, ExponentialImpulseResponse):
        lp += np.sum(_gamma_logpdf(imp.θ, imp.α, imp.β))
    else:
        lp += np.sum(_gamma_logpdf(imp.τ, imp.α0, imp.β0))
        prec = imp.κμ * imp.τ
        lp += np.sum(0.5 * np.log(prec / (2 * np.pi)) - 0.5 * prec * (imp.μ - imp.μμ) ** 2)
    return float(lp)


def _logprior_gradient(process):
    b, w, imp = process.baseline, process.weights, process.impulses
    g = [(b.α0 - 1.0) / b.λ - b.β0]
    if isinstance(imp, ExponentialImpulseResponse):
        g.append(((imp.α - 1.0) / imp.θ - imp.β).ravel(order="F"))
    else:
        g.append((-imp.κμ * imp.τ * (imp.μ - imp.μμ)).ravel(order="F"))
        g.append(((imp.α0 - 1.0) / imp.τ - imp.β0 + 0.5 / imp.τ - 0.5 * imp.κμ * (imp.μ - imp.μμ) ** 2).ravel(order="F"))
    g.append(((w.κ - 1.0) / w.W - w.ν).ravel(order="F"))
    return np.concatenate(g)


def _rand_init_(process, rng):
    """src/continuous.jl:200: rand(length(params(process)))"""
    return rng.uniform(size=len(process.params()))


def mle_(process, data, optimizer="L-BFGS-B", verbose=False, f_abstol=1e-6, regularize=False, guess=None,
         recursive=True, seed=None, max_steps=1000, ctx=None):
    """mle!(process, data; optimizer, verbose, f_abstol, regularize, guess) -- src/continuous.jl:144-198.

    Same objective (-loglikelihood [- logprior]), same box [1e-6, 10] on every coordinate, same
    random start and the same |f - f_prev| < f_abstol stopping rule; `process` is overwritten with
    the estimate.  The reference runs Optim's Fminbox(BFGS) on finite differences (2P objective
    calls per gradient); here a box-constrained quasi-Newton method (scipy L-BFGS-B) is fed the
    analytic gradient computed on the GPU, so iterates differ while the optimum is the same.
    optimizer="device": the whole iteration on the GPU (nhp_cont_mle_run, projected L-BFGS with its state in HBM) -- at
    2.1e6 parameters the host route spends its time moving x and ∇ll over PCIe and in the host-side update."""
    if not isinstance(process, ContinuousStandardHawkesProcess):
        raise TypeError("mle! is defined for ContinuousStandardHawkesProcess (src/continuous.jl:144)")
    if regularize and not isinstance(process.baseline, HomogeneousProcess):
        raise NotImplementedError("logprior is not defined for LogGaussianCoxProcess in the reference (src/baselines.jl)")
    from .sharded import ShardedDataset, _all_reduce_sum
    shard = data if isinstance(data, ShardedDataset) else None       # every rank runs the same optimizer on all-reduced values
    ctx = (shard.ctx if shard else ctx) or _lib.default_context()
    ds = shard.local if shard else device_dataset(process, data, ctx)
    rng = np.random.default_rng(seed)
    x0 = _rand_init_(process, rng) if guess is None else np.asarray(guess, dtype=np.float64)
    if shard is not None and shard.world > 1:                        # rank 0's start everywhere
        x0 = _all_reduce_sum(x0 if shard.rank == 0 else np.zeros_like(x0), ctx)
    lower, upper = 1e-6, 1e1
    state = {"minloss": np.inf, "steps": 0, "converged": False, "last": None}
    start = time.time()

    import ctypes as C
    from .continuous import _check_recursive
    flags = _check_recursive(process, recursive)
    P = len(x0)
    comm = _lib.comm_for(ctx) if shard is not None else None

    if optimizer in ("device", "LBFGS-device"):
        # the optimizer's state on the device (nhp_cont_mle_run: projected L-BFGS in HBM, the host reads scalars): no
        # parameter upload, gradient download or host-side quasi-Newton update per objective call
        if regularize:
            raise NotImplementedError("optimizer='device' minimises -loglikelihood only; use the host optimizer with regularize=True")
        if shard is not None and shard.world > 1 and comm is None:
            raise NotImplementedError("optimizer='device' on a sharded dataset needs an RCCL clique (one GPU per rank)")
        x = np.clip(x0, lower, upper)                               # (a new float64 vector: the caller's guess is not written to)
        # mle! overwrites the process anyway: taking the start into it first leaves its tables column-major like the
        # vector, so lowering them to the device model is a plain copy (a fresh process holds row-major numpy arrays, whose
        # lowering transposes 16 MB at N = 1024); a wrong-length guess raises the reference's error here
        process.params_(x)
        model = process.device_model(ctx)
        loss, steps, conv, evals = C.c_double(), C.c_int32(), C.c_int32(), C.c_int32()
        _lib.check(_lib.lib().nhp_cont_mle_run(ctx.h, comm.h if comm is not None else None, ds.h, model.h, flags, lower, upper,
                                               float(f_abstol), int(max_steps), _lib.dptr(x), P, C.byref(loss), C.byref(steps),
                                               C.byref(conv), C.byref(evals)), ctx.h)
        if verbose:
            print(f" > steps: {steps.value}, objective evaluations: {evals.value}, loss: {loss.value}, elapsed: {time.time() - start}")
        process.params_(x)
        res = MaximumLikelihood(x, -float(loss.value), int(steps.value), time.time() - start,
                                "success" if conv.value else "failure")
        res.evaluations = int(evals.value)
        return res

    model = process.device_model(ctx)

    def fg(x):
        # params!(process, x) straight into the device-resident model: x already is the reference's
        # column-major parameter vector, so nothing is re-packed on the host per objective call
        model.set_params(x)
        g = np.empty(P)
        ll_c = C.c_double()
        if comm is not None:           # the other ranks' columns: [ll; ∇ll] summed on the device over RCCL, one download
            _lib.check(_lib.lib().nhp_cont_loglik_grad_allreduce(ctx.h, comm.h, ds.h, model.h, flags, C.byref(ll_c), _lib.dptr(g), P), ctx.h)
            ll = ll_c.value
        else:
            _lib.check(_lib.lib().nhp_cont_loglik_grad(ctx.h, ds.h, model.h, flags, C.byref(ll_c), _lib.dptr(g), P), ctx.h)
            ll = ll_c.value
            if shard is not None:                                    # (gloo rehearsal) the other ranks' columns through the host
                tot = _all_reduce_sum(np.concatenate([[ll], g]), ctx)
                ll, g = float(tot[0]), tot[1:]
        if regularize:
            process.params_(x)
            ll += logprior(process)
            g = g + _logprior_gradient(process)
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
    if optimizer == "L-BFGS-B":
        # scipy's own tests would end the run long before the reference's rule does (its default is a RELATIVE decrease of
        # 2.2e-9, i.e. 2e-5 at |f| ~ 1e4): off, but for Optim's default gradient tolerance (g_tol = 1e-8)
        options.update(ftol=0.0, gtol=1e-8, maxfun=20 * max_steps + 1000)
    res = optimize.minimize(fg, np.clip(x0, lower, upper), jac=True, method=optimizer,
                            bounds=[(lower, upper)] * len(x0), callback=status_update,
                            options=options)
    process.params_(res.x)
    return MaximumLikelihood(res.x.copy(), -float(res.fun), state["steps"], time.time() - start,
                             "success" if (state["converged"] or res.success) else "failure")


def resample_adjacency_matrix_(process, data, u=None, seed=0, step=0, model=None, fetch=True, ctx=None):
    """resample_adjacency_matrix!(process, data) -- src/continuous.jl:444-470: one Gibbs sweep of the
    adjacency matrix, columns in parallel, entries of a column in sequence (resample_column! :472-487).
    `u` (N x N, [parent, child]) supplies the Bernoulli uniforms explicitly; otherwise they come from
    Philox keyed (seed, step).  Updates process.adjacency_matrix in place and returns the link count."""
    import ctypes as C
    ctx = ctx or _lib.default_context()
    ds = device_dataset(process, data, ctx)
    model = model or process.device_model(ctx)
    N = process.ndims()
    from .components import BernoulliNetworkModel, DenseNetworkModel
    if isinstance(process.network, BernoulliNetworkModel):       # link_probability = ρ .* ones  (src/networks.jl:65-68)
        scalar, rho_m = float(process.network.ρ), None
    elif isinstance(process.network, DenseNetworkModel):
        scalar, rho_m = 1.0, None
    else:
        scalar, rho_m = None, _lib.colmajor(np.asarray(process.network.link_probability(), dtype=np.float64))
    uu = None if u is None else _lib.colmajor(np.asarray(u, dtype=np.float64))
    A = np.empty(N * N) if fetch else None
    nl = C.c_double()
    _lib.check(_lib.lib().nhp_cont_resample_adjacency(ctx.h, ds.h, model.h, _lib.dptr(rho_m), scalar if scalar is not None else 0.5,
                                                      _lib.dptr(uu), seed, step, _lib.dptr(A), C.byref(nl)), ctx.h)
    if fetch:
        process.adjacency_matrix = A.reshape((N, N), order="F")
    return nl.value


def resample_(process, data, rng, step=0, seed=0, ctx=None):
    """resample!(process, data) -- src/continuous.jl:202-208,350-358: one Gibbs sweep.

    Parents and every sufficient statistic come from one GPU call; the conjugate draws are host
    numpy; the network process then resamples its adjacency matrix on the GPU and redraws ρ."""
    _, _, st = resample_parents(process, data, seed=seed, step=step, with_stats=True, want_parents=False, ctx=ctx)
    duration = data.duration if hasattr(data, "duration") else data[2]
    if isinstance(process.baseline, HomogeneousProcess):
        process.baseline.resample_(st["cnt0"], duration, rng)
    else:       # LGCP: elliptical slice on the events the sweep above attributed to the baseline
        process.baseline.resample_(device_dataset(process, data, ctx), rng)
    process.weights.resample_(st["Mn"], st["Mnm"], rng)
    if isinstance(process.impulses, ExponentialImpulseResponse):
        process.impulses.resample_(st["Mnm"], st["Xnm"], rng)
    else:
        process.impulses.resample_(st["Mnm"], st["Xnm"], st["Vnm"], rng)
    if isinstance(process, ContinuousNetworkHawkesProcess):
        resample_adjacency_matrix_(process, data, seed=seed, step=step, ctx=ctx)
        process.network.resample_(process.adjacency_matrix, rng)
    return process.params()


def _priors(process):
    b, w, imp = process.baseline, process.weights, process.impulses
    if isinstance(imp, ExponentialImpulseResponse):
        return _lib.GibbsPriors(b.α0, b.β0, w.κ, w.ν, imp.α, imp.β, 0.0, 1.0)
    return _lib.GibbsPriors(b.α0, b.β0, w.κ, w.ν, imp.α0, imp.β0, imp.μμ, imp.κμ)


def _pull_params(process, model, ctx):
    """Copy the device-resident parameters back into the component structs (in-place semantics of
    the reference: "all inference methods overwrite model parameters", docs/src/index.md:109-111)."""
    import ctypes as C
    N = process.ndims()
    nimp = N * N * (1 if isinstance(process.impulses, ExponentialImpulseResponse) else 2)
    x = np.empty(N + nimp + N * N)
    _lib.check(_lib.lib().nhp_cont_model_get_params(ctx.h, model.h, _lib.dptr(x), len(x)), ctx.h)
    process.baseline.λ = x[:N].copy()
    process.impulses.params_(x[N:N + nimp])
    process.weights.params_(x[N + nimp:])


def _fetch_moments(process, model, ctx, network, rho_sum, rho_sq):
    """Mean and mean square of params(process) from the device-side running sums, in params(process) order."""
    import ctypes as C
    N = process.ndims()
    nimp = N * N * (1 if isinstance(process.impulses, ExponentialImpulseResponse) else 2)
    L = N + nimp + N * N + (N * N if network else 0)
    s, q = np.empty(L), np.empty(L)
    cnt = C.c_int64()
    _lib.check(_lib.lib().nhp_cont_model_moments_fetch(ctx.h, model.h, _lib.dptr(s), _lib.dptr(q), L, C.byref(cnt)), ctx.h)
    mean, m2 = _moments_in_params_order(process, s, q, cnt.value, network, rho_sum, rho_sq)
    return mean, m2, cnt.value


def moments_length(process):
    """Length of the device-side running sums of nhp_cont_model_moments_*: [λ0; impulses; W; vec(A) if any]."""
    N = process.ndims()
    nimp = N * N * (1 if isinstance(process.impulses, ExponentialImpulseResponse) else 2)
    return N + nimp + N * N + (N * N if isinstance(process, ContinuousNetworkHawkesProcess) else 0)


def _moments_in_params_order(process, s, q, count, network, rho_sum, rho_sq):
    """Σx, Σx² in the device order -> mean and mean square in params(process) order."""
    N = process.ndims()
    nimp = N * N * (1 if isinstance(process.impulses, ExponentialImpulseResponse) else 2)
    n = max(1, count)
    mean, m2 = s / n, q / n
    if network:        # device order [λ0; impulses; W; vec(A)] -> params(process) = [ρ; λ0; W; impulses; vec(A)] (src/continuous.jl:325-333)
        k = len(process.network.params())
        a, b, c = N, N + nimp, N + nimp + N * N

        def order(v, rho):
            return np.concatenate([np.full(k, rho), v[:a], v[b:c], v[a:b], v[c:]])
        mean, m2 = order(mean, rho_sum / n), order(m2, rho_sq / n)
    return mean, m2


def _owned_mask(process, shard, network):
    """1 on the entries of params(process) this rank owns (its columns; rank 0 also the network's ρ), 0 elsewhere."""
    N = process.ndims()
    c0, c1 = shard.ranges[shard.rank]
    col = np.zeros(N)
    col[c0:c1] = 1.0
    mat = np.repeat(col, N)                                   # vec of an N x N matrix, column-major: index p + c·N
    nmat = 1 if isinstance(process.impulses, ExponentialImpulseResponse) else 2
    if not network:                                           # [λ0; impulses; W]
        return np.concatenate([col] + [mat] * (nmat + 1))
    k = len(process.network.params())                         # [ρ; λ0; W; impulses; vec(A)]
    return np.concatenate([np.full(k, 1.0 if shard.rank == 0 else 0.0), col] + [mat] * (nmat + 2))


def _merge_shards(process, shard, network):
    """After a sharded chain every rank holds the final values of its own columns: put the full state on every rank."""
    from .sharded import _all_reduce_sum
    N = process.ndims()
    full = _all_reduce_sum(process.params() * _owned_mask(process, shard, network), shard.ctx)
    k = len(process.network.params()) if network else 0
    nimp = N * N * (1 if isinstance(process.impulses, ExponentialImpulseResponse) else 2)
    process.baseline.λ = full[k:k + N].copy()
    if network:
        process.weights.params_(full[k + N:k + N + N * N])
        process.impulses.params_(full[k + N + N * N:k + N + N * N + nimp])
        process.adjacency_matrix = full[k + N + N * N + nimp:].reshape((N, N), order="F").copy()
    else:
        process.impulses.params_(full[N:N + nimp])
        process.weights.params_(full[N + nimp:])


def mcmc_(process, data, nsteps=1000, log_freq=100, verbose=False, seed=0, keep_samples=True, device_draws=True,
          ctx=None, moments=False, burn=0):
    """mcmc!(process, data; nsteps, log_freq, verbose) -- src/inference.jl:49-70.

    With `device_draws` (default) a whole sweep -- parents, statistics, conjugate draws -- stays on
    the GPU (nhp_cont_gibbs_step): parameters never cross PCIe unless samples are kept.  With
    device_draws=False the statistics come back and numpy draws the parameters (same
    distributions).  `seed` keys every random stream, so a chain is reproducible and chains with
    different seeds are independent (one per GPU: chains.py).

    `moments=True` (device draws only) keeps the chain's running sums on the device (nhp_cont_model_moments_*): after the
    run `res.mean` and `res.m2` hold the mean and the mean square of params(process) over the steps >= `burn` -- the
    summaries chains.py gathers -- with no per-step transfer; combine with keep_samples=False for long chains at large N
    (a sample is 4N²+N doubles, 33.5 MB at N = 1024).

    ONE chain over several GPUs: pass a `sharded.ShardedDataset` (device draws, keep_samples=False).  A sweep is
    separable by child-node column -- the parents of the children on c, column c's statistics and conjugate draws and
    the sweep of A[:, c] touch column c only, and every random stream is keyed by global event / entry indices -- so
    each rank sweeps its own columns and the chain is the single-GPU chain, value for value; the only exchange per step
    is the scalar link count the network's ρ update needs, and at the end the ranks' columns (and moments) are merged."""
    import ctypes as C
    from .sharded import ShardedDataset, _all_reduce_sum
    from .components import BernoulliNetworkModel, DenseNetworkModel
    shard = data if isinstance(data, ShardedDataset) else None
    if not isinstance(process.baseline, HomogeneousProcess):
        device_draws = False      # nhp_cont_gibbs_step draws the homogeneous λ0; the LGCP curve is a host slice loop
    if moments and not device_draws:
        raise ValueError("moments=True needs the device-side draws (homogeneous baseline, device_draws=True)")
    if shard is not None and (not device_draws or keep_samples):
        raise ValueError("a sharded chain runs with the device-side draws and keep_samples=False (use moments=True)")
    ctx = (shard.ctx if shard else ctx) or _lib.default_context()
    ds = shard.local if shard else device_dataset(process, data, ctx)
    rng = np.random.default_rng(seed)
    res = MarkovChainMonteCarlo()
    start = time.time()
    lib = _lib.lib()
    model = process.device_model(ctx) if device_draws else None
    pri = _priors(process) if device_draws else None
    network = isinstance(process, ContinuousNetworkHawkesProcess)
    # the network's link probability lives on the device too (nhp_cont_model_set_rho): ρ ~ Beta(α + ΣA, β + N² - ΣA) is
    # drawn there (src/networks.jl:70-78), so a network step needs no synchronisation; DenseNetworkModel keeps ρ = 1
    net = process.network if network else None
    net_a, net_b = (net.α, net.β) if isinstance(net, BernoulliNetworkModel) else (0.0, 0.0)
    device_net = device_draws and isinstance(net, (BernoulliNetworkModel, DenseNetworkModel))
    if device_net:
        _lib.check(lib.nhp_cont_model_set_rho(ctx.h, model.h, net.ρ if isinstance(net, BernoulliNetworkModel) else 1.0), ctx.h)
    comm = _lib.comm_for(ctx) if shard is not None else None
    host_exchange = shard is not None and shard.world > 1 and comm is None        # gloo rehearsal: link counts through the host
    if moments:
        _lib.check(lib.nhp_cont_model_moments_reset(ctx.h, model.h), ctx.h)

    def pull_rho():
        if device_net and isinstance(net, BernoulliNetworkModel):
            r3 = np.empty(3)
            _lib.check(lib.nhp_cont_model_get_rho(ctx.h, model.h, _lib.dptr(r3)), ctx.h)
            net.ρ = float(r3[0])
            return r3
        return np.zeros(3)

    def pull_adjacency():
        N = process.ndims()
        A = np.empty(N * N)
        _lib.check(lib.nhp_cont_model_get_adjacency(ctx.h, model.h, _lib.dptr(A), N * N), ctx.h)
        process.adjacency_matrix = A.reshape((N, N), order="F")

    # The whole chain inside the library (nhp_cont_mcmc_run: the body of src/inference.jl:55-62, one synchronisation per
    # call) when no step needs the host: no samples kept, device-side network.  `verbose` cuts it at the log points.
    resident = device_draws and not keep_samples and (not network or device_net) and not host_exchange
    if resident:
        while res.steps < nsteps:
            n = min(nsteps - res.steps, log_freq if verbose else nsteps)
            _lib.check(lib.nhp_cont_mcmc_run(ctx.h, comm.h if comm is not None else None, ds.h, model.h, C.byref(pri), net_a, net_b,
                                             seed, res.steps, n, burn if moments else -1), ctx.h)
            res.steps += n
            if verbose and res.steps % log_freq == 0:
                res.elapsed = time.time() - start
                print(f" > step: {res.steps}, elapsed: {res.elapsed}")
        _pull_params(process, model, ctx)
        if network:
            pull_adjacency()
            pull_rho()
    while res.steps < nsteps:
        if device_draws:
            _lib.check(lib.nhp_cont_gibbs_step(ctx.h, ds.h, model.h, C.byref(pri), seed, res.steps), ctx.h)
            last = keep_samples or res.steps == nsteps - 1
            if network and device_net and not host_exchange:
                _lib.check(lib.nhp_cont_network_step(ctx.h, comm.h if comm is not None else None, ds.h, model.h, net_a, net_b,
                                                     seed, res.steps), ctx.h)
            elif network and device_net:                     # ranks without an RCCL clique: the one exchange of a step, by hand
                nl = C.c_double()
                _lib.check(lib.nhp_cont_network_sweep(ctx.h, ds.h, model.h, seed, res.steps, C.byref(nl)), ctx.h)
                links = float(_all_reduce_sum(np.array([nl.value]), ctx)[0])
                _lib.check(lib.nhp_cont_network_rho(ctx.h, model.h, net_a, net_b, links, float(process.ndims()) ** 2, seed, res.steps), ctx.h)
            elif network:                                    # a network model the library does not hold: host-side ρ
                links = resample_adjacency_matrix_(process, ds, seed=seed, step=res.steps, model=model, fetch=last, ctx=ctx)
                process.network.resample_links_(links, process.ndims() ** 2, rng)
            if moments and res.steps >= burn:
                _lib.check(lib.nhp_cont_model_moments_accumulate(ctx.h, model.h), ctx.h)
            if last:
                _pull_params(process, model, ctx)
                if network and device_net:
                    pull_adjacency()
                    pull_rho()
            x = process.params() if keep_samples else None
        else:
            x = resample_(process, ds, rng, step=res.steps, seed=seed, ctx=ctx)
        if keep_samples:
            res.samples.append(x)
        res.steps += 1
        if res.steps % log_freq == 0 and verbose:
            res.elapsed = time.time() - start
            print(f" > step: {res.steps}, elapsed: {res.elapsed}")
    res.elapsed = time.time() - start
    res.status = "complete"
    if shard is not None and shard.world > 1:
        _merge_shards(process, shard, network)
    if not keep_samples:
        res.samples.append(process.params())
    if moments:
        r3 = pull_rho() if network else np.zeros(3)
        res.mean, res.m2, res.n = _fetch_moments(process, model, ctx, network, r3[1], r3[2])
        if shard is not None and shard.world > 1:
            mask = _owned_mask(process, shard, network)
            res.mean, res.m2 = _all_reduce_sum(res.mean * mask, ctx), _all_reduce_sum(res.m2 * mask, ctx)
    return res
