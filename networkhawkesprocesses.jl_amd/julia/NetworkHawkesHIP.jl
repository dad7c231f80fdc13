# NetworkHawkesHIP.jl -- the reference-side binding of libnhp.so (include/nhp.h, ABI 2).
#
# Drop-in for the hot path only: `using NetworkHawkesProcesses; include("NetworkHawkesHIP.jl")` gives GPU methods with
# the package's own names, argument meaning, keyword arguments and result structs
#     loglikelihood / intensity / resample_parents / mle! / mcmc!                (src/continuous.jl, src/parents.jl)
#     convolve / intensity / loglikelihood / update! / vb! / mle! / mcmc!        (src/discrete.jl, src/inference.jl)
# that take the package's process structs and data and `ccall` the C ABI.  Nothing else of the package changes.
# Julia is not installed in the build image, so this file is NOT executed by the test-suite; every entry point it binds
# is exercised through the identical ctypes binding (networkhawkesprocesses.jl_amd/_lib.py), and the struct layouts it
# assumes are checked against the loaded library when the module initialises (ABI_LAYOUT below; the same numbers are
# asserted from ctypes in tests/test_abi_and_host.py).  Keep it declarative: all logic lives behind the C ABI.
module NetworkHawkesHIP

using NetworkHawkesProcesses
import Optim                      # already a dependency of NetworkHawkesProcesses (Project.toml)
const NHP = NetworkHawkesProcesses

const libnhp = get(ENV, "NHP_LIB", joinpath(@__DIR__, "..", "libnhp.so"))

# sizeof / offsetof of nhp_cont_model_desc (80 bytes), nhp_gibbs_priors (64), nhp_cont_stats (40), then NHP_MAX_SLOTS and
# NHP_COMM_ID_BYTES -- what nhp_abi_layout() of the library must return for the structs below to be passed by reference
const ABI_LAYOUT = Int32[80, 0, 4, 8, 16, 24, 28, 32, 40, 48, 56, 64, 72,
                         64, 0, 8, 16, 24, 32, 40, 48, 56,
                         40, 0, 8, 16, 24, 32,
                         4096, 128]

function __init__()
    v = ccall((:nhp_abi_version, libnhp), Int32, ())
    v == 2 || error("libnhp.so has ABI version $v, this binding expects 2")
    got = zeros(Int32, 64)
    n = ccall((:nhp_abi_layout, libnhp), Int32, (Ptr{Int32}, Int32), got, 64)
    got[1:n] == ABI_LAYOUT || error("struct layout mismatch between NetworkHawkesHIP.jl and libnhp.so: $(got[1:n])")
    sizeof(ModelDesc) == 80 && sizeof(Priors) == 64 && sizeof(Stats) == 40 || error("Julia struct sizes differ from the C ABI")
end

# --- status -> exception (include/nhp.h: nhp_status) ---------------------------------------
function check(rc::Int32, ctx::Ptr{Cvoid}=C_NULL)
    rc == 0 && return
    msg = unsafe_string(ccall((:nhp_last_error, libnhp), Cstring, (Ptr{Cvoid},), ctx))
    rc == 2 && throw(DomainError(msg))            # NHP_EDOMAIN  (src/baselines.jl:100,106,111,116)
    rc == 3 && error(msg)                         # NHP_ESHAPE   (src/impulses.jl:44-45, src/weights.jl:10-11)
    rc == 1 && throw(ArgumentError(msg))          # NHP_EINVAL
    error("libnhp status $rc: $msg")              # NHP_ENOMEM / NHP_EHIP / NHP_ENOTIMPL / NHP_ERCCL
end

mutable struct Context
    h::Ptr{Cvoid}
    function Context(device::Integer=parse(Int, get(ENV, "NHP_DEVICE", get(ENV, "LOCAL_RANK", "0"))))
        r = Ref{Ptr{Cvoid}}(C_NULL)
        check(ccall((:nhp_ctx_create, libnhp), Int32, (Int32, Ref{Ptr{Cvoid}}), device, r))
        ctx = new(r[])
        finalizer(c -> ccall((:nhp_ctx_destroy, libnhp), Cvoid, (Ptr{Cvoid},), c.h), ctx)
    end
end
const DEFAULT = Ref{Union{Nothing,Context}}(nothing)
context() = (DEFAULT[] === nothing && (DEFAULT[] = Context()); DEFAULT[])

# --- nhp_cont_model_desc: the lowered (Baseline, ImpulseResponse, Weights, A); offsets in ABI_LAYOUT[2:13] -----------
struct ModelDesc
    n_nodes::Int32; baseline_kind::Int32; lambda0::Ptr{Float64}; grid_x::Ptr{Float64}
    grid_n::Int32; impulse_kind::Int32; theta::Ptr{Float64}; mu::Ptr{Float64}; tau::Ptr{Float64}
    dt_max::Float64; W::Ptr{Float64}; A::Ptr{Float64}
end
struct Priors   # nhp_gibbs_priors; offsets in ABI_LAYOUT[15:22]
    α0::Float64; β0::Float64; κ::Float64; ν::Float64; a::Float64; b::Float64; μμ::Float64; κμ::Float64
end
struct Stats    # nhp_cont_stats; offsets in ABI_LAYOUT[24:28]
    cnt0::Ptr{Float64}; Mn::Ptr{Float64}; Mnm::Ptr{Float64}; Xnm::Ptr{Float64}; Vnm::Ptr{Float64}
end

# Julia arrays are already column-major [parent, child]: pointers are passed untouched.
function lower(p::NHP.ContinuousHawkesProcess)
    keep = Any[]
    f64(x) = (a = Array{Float64}(x); push!(keep, a); pointer(a))
    N = NHP.ndims(p)
    if p.baseline isa NHP.HomogeneousProcess
        bk, l0, gx, gn = Int32(0), f64(p.baseline.λ), Ptr{Float64}(C_NULL), Int32(0)
    else   # LogGaussianCoxProcess evaluator: grid x, λ[k] per node (src/baselines.jl:148-173)
        bk, l0, gx, gn = Int32(1), f64(vcat(p.baseline.λ...)), f64(p.baseline.x), Int32(length(p.baseline.x))
    end
    nul = Ptr{Float64}(C_NULL)
    if p.impulses isa NHP.ExponentialImpulseResponse
        ik, th, mu, tau = Int32(0), f64(p.impulses.θ), nul, nul
    else
        ik, th, mu, tau = Int32(1), nul, f64(p.impulses.μ), f64(p.impulses.τ)
    end
    A = p isa NHP.ContinuousNetworkHawkesProcess ? f64(p.adjacency_matrix) : nul
    ModelDesc(N, bk, l0, gx, gn, ik, th, mu, tau, Float64(p.impulses.Δtmax), f64(p.weights.W), A), keep
end

mutable struct Dataset          # (events, nodes, duration) uploaded once; pre-pass for Δtmax done
    h::Ptr{Cvoid}
end
# columns = 1-based node range this process evaluates (one loglikelihood / chain over several GPUs: `comm` keyword of the
# entry points below); the default is the whole dataset
function Dataset(ctx::Context, data, N::Integer, Δtmax::Real; columns::UnitRange{Int}=1:N)
    events, nodes, duration = data
    ev, nd = Vector{Float64}(events), Vector{Int64}(nodes)
    r = Ref{Ptr{Cvoid}}(C_NULL)
    GC.@preserve ev nd check(ccall((:nhp_cont_dataset_create_columns, libnhp), Int32,
        (Ptr{Cvoid}, Ptr{Float64}, Ptr{Int64}, Int64, Int32, Float64, Float64, Int32, Int32, Ref{Ptr{Cvoid}}),
        ctx.h, ev, nd, length(ev), N, duration, Δtmax, first(columns) - 1, last(columns), r), ctx.h)
    ds = Dataset(r[])
    finalizer(d -> ccall((:nhp_cont_dataset_destroy, libnhp), Cvoid, (Ptr{Cvoid},), d.h), ds)
end

function with_model(f, ctx::Context, p)
    desc, keep = lower(p)
    r = Ref{Ptr{Cvoid}}(C_NULL)
    GC.@preserve keep begin
        check(ccall((:nhp_cont_model_create, libnhp), Int32, (Ptr{Cvoid}, Ref{ModelDesc}, Ref{Ptr{Cvoid}}),
                    ctx.h, Ref(desc), r), ctx.h)
    end
    try f(r[]) finally ccall((:nhp_cont_model_destroy, libnhp), Cvoid, (Ptr{Cvoid},), r[]) end
end

llflags(p, recursive) = Int32(recursive && p.impulses isa NHP.ExponentialImpulseResponse ? 1 : 0)

# ---- several GPUs: RCCL over xGMI through the library (include/nhp.h, multi-GPU section) ------------------------------
# Rank 0: id = unique_id(); ship the 128 bytes to the other ranks (Distributed.remotecall, a file, MPI.bcast ...); every
# rank: comm = Comm(ctx, id, rank, world).  `comm` is then a keyword of loglikelihood / loglikelihood_gradient / mle! /
# mcmc! (ONE evaluation or chain over all ranks, each its column range), and gather_moments collects the independent
# chains of BASELINE config 5.
mutable struct Comm
    h::Ptr{Cvoid}; rank::Int; world::Int
end
function unique_id()
    id = zeros(UInt8, 128)
    check(ccall((:nhp_comm_unique_id, libnhp), Int32, (Ptr{UInt8},), id))
    id
end
function Comm(ctx::Context, id::Vector{UInt8}, rank::Integer, world::Integer)
    r = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:nhp_comm_create, libnhp), Int32, (Ptr{Cvoid}, Ptr{UInt8}, Int32, Int32, Ref{Ptr{Cvoid}}), ctx.h, id, rank, world, r), ctx.h)
    c = Comm(r[], rank, world)
    finalizer(x -> ccall((:nhp_comm_destroy, libnhp), Cvoid, (Ptr{Cvoid},), x.h), c)
end
commptr(c) = c === nothing ? Ptr{Cvoid}(C_NULL) : c.h
function allreduce_sum!(x::Vector{Float64}, comm::Comm; ctx=context())
    check(ccall((:nhp_allreduce_sum, libnhp), Int32, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Float64}, Int64), ctx.h, comm.h, x, length(x)), ctx.h)
    x
end

# --- loglikelihood(process, data; recursive=true)  src/continuous.jl:210,360 ----------------
function loglikelihood(p::NHP.ContinuousHawkesProcess, data; recursive=true, ctx=context(), comm=nothing,
                       ds=Dataset(ctx, data, NHP.ndims(p), p.impulses.Δtmax))
    ll = Ref{Float64}(0.0)
    with_model(ctx, p) do m
        if comm === nothing
            check(ccall((:nhp_cont_loglik, libnhp), Int32, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Int32, Ref{Float64}),
                        ctx.h, ds.h, m, llflags(p, recursive), ll), ctx.h)
        else    # ds is this rank's column shard: partial results summed on the device over RCCL
            check(ccall((:nhp_cont_loglik_allreduce, libnhp), Int32, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Int32, Ref{Float64}),
                        ctx.h, comm.h, ds.h, m, llflags(p, recursive), ll), ctx.h)
        end
    end
    ll[]
end

# --- intensity(process, data, times) -> length(times) x N; intensity(process, data, time) -> N  src/continuous.jl:76-96
function intensity(p::NHP.ContinuousHawkesProcess, data, times::Vector{Float64}; ctx=context(),
                   ds=Dataset(ctx, data, NHP.ndims(p), p.impulses.Δtmax))
    out = Matrix{Float64}(undef, length(times), NHP.ndims(p))
    with_model(ctx, p) do m
        check(ccall((:nhp_cont_intensity, libnhp), Int32,
                    (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Float64}, Int64, Ptr{Float64}),
                    ctx.h, ds.h, m, times, length(times), out), ctx.h)
    end
    out
end
intensity(p::NHP.ContinuousHawkesProcess, data, time::Float64; kw...) = vec(intensity(p, data, [time]; kw...))

# --- resample_parents(process, data) -> (parents, parentnodes)  src/parents.jl:1-23 ----------
# stats=true also returns the Gibbs sufficient statistics of the same sweep (src/parents.jl:61-79, src/baselines.jl:87-96,
# src/impulses.jl:84-96,216-252) as a NamedTuple of N / N x N Float64 arrays.
function resample_parents(p::NHP.ContinuousHawkesProcess, data; seed::UInt64=UInt64(0), step::UInt64=UInt64(0), stats=false,
                          ctx=context(), ds=Dataset(ctx, data, NHP.ndims(p), p.impulses.Δtmax))
    M, N = length(data[1]), NHP.ndims(p)
    parents, parentnodes = Vector{Int64}(undef, M), Vector{Int64}(undef, M)
    cnt0, Mn, Mnm, Xnm, Vnm = zeros(N), zeros(N), zeros(N, N), zeros(N, N), zeros(N, N)
    st = Ref(Stats(pointer(cnt0), pointer(Mn), pointer(Mnm), pointer(Xnm), pointer(Vnm)))
    with_model(ctx, p) do m
        GC.@preserve st cnt0 Mn Mnm Xnm Vnm check(ccall((:nhp_cont_resample_parents, libnhp), Int32,
                    (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Float64}, UInt64, UInt64, Ptr{Int64}, Ptr{Int64}, Ptr{Cvoid}),
                    ctx.h, ds.h, m, C_NULL, seed, step, parents, parentnodes, stats ? Base.unsafe_convert(Ptr{Cvoid}, st) : C_NULL), ctx.h)
    end
    stats ? (parents, parentnodes, (cnt0=cnt0, Mn=Mn, Mnm=Mnm, Xnm=Xnm, Vnm=Vnm)) : (parents, parentnodes)
end

# --- objective + analytic gradient of mle!  src/continuous.jl:144-198 -------------------------------------------------
function loglikelihood_gradient(p::NHP.ContinuousStandardHawkesProcess, data; recursive=true, ctx=context(), comm=nothing,
                                ds=Dataset(ctx, data, NHP.ndims(p), p.impulses.Δtmax))
    P = length(NHP.params(p))
    g, ll = Vector{Float64}(undef, P), Ref{Float64}(0.0)
    with_model(ctx, p) do m
        grad_call(ctx, comm, ds, m, llflags(p, recursive), ll, g)
    end
    ll[], g
end
function grad_call(ctx, comm, ds, m, flags, ll, g)
    if comm === nothing
        check(ccall((:nhp_cont_loglik_grad, libnhp), Int32, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Int32, Ref{Float64}, Ptr{Float64}, Int64),
                    ctx.h, ds.h, m, flags, ll, g, length(g)), ctx.h)
    else
        check(ccall((:nhp_cont_loglik_grad_allreduce, libnhp), Int32,
                    (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Int32, Ref{Float64}, Ptr{Float64}, Int64),
                    ctx.h, comm.h, ds.h, m, flags, ll, g, length(g)), ctx.h)
    end
end

# loglikelihood(process::LogGaussianCoxProcess, data, node, y) for every node at once
# (src/baselines.jl:247-254): Y is G x N, column c the candidate latent curve of node c.
# `parentnodes === nothing` reuses the attribution the latest resample_parents left on the device.
function lgcp_loglikelihood(b::NHP.LogGaussianCoxProcess, ds::Dataset, Y::Matrix{Float64};
                            parentnodes::Union{Nothing,Vector{Int64}}=nothing, ctx=context())
    lam = exp.(b.m .+ Y)
    ll = Vector{Float64}(undef, size(Y, 2))
    pn = parentnodes === nothing ? Ptr{Int64}(C_NULL) : pointer(parentnodes)
    GC.@preserve parentnodes lam ll check(ccall((:nhp_cont_lgcp_loglik, libnhp), Int32,
        (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Int64}, Ptr{Float64}, Int32, Ptr{Float64}, Ptr{Float64}),
        ctx.h, ds.h, pn, b.x, Int32(length(b.x)), lam, ll), ctx.h)
    ll
end

# ---- the reference's mle! driver around an objective/gradient pair ---------------------------------------------------
# Same keyword arguments, box, callback logic, printed banners and result struct as src/continuous.jl:144-198 and
# src/discrete.jl:211-296 (`max_increase_steps` is the discrete method's extra stop rule; `nothing` = not used).
# `fg!(G, x)` returns -loglikelihood [- logprior] and fills G with its gradient; the reference hands Optim no gradient
# (2P objective calls per finite-difference gradient), here it is analytic and comes from the same GPU call.
function run_mle(fg!, guess; optimizer, verbose, f_abstol, max_increase_steps=nothing)
    minloss, outer_iter, converged, steps, increase_steps = Inf, 0, false, 0, 0
    function banner(o, what)
        println("\n* Status: $what criteria reached!")
        println("    elapsed: $(o.metadata["time"])")
        println("    final loss: $(o.value)")
        println("    min. loss: $(minloss)")
        println("    outer iterations: $outer_iter")
        println("    inner iterations: $(o.iteration)\n")
    end
    function status_update(o)
        if o.iteration == 0
            verbose && println("* iteration (n=$outer_iter)")
            outer_iter += 1
            minloss = Inf
        end
        verbose && println(" > step: $(o.iteration), loss: $(o.value), elapsed: $(o.metadata["time"])")
        if abs(o.value - minloss) < f_abstol
            converged = true; steps = o.iteration
            banner(o, "f_abstol convergence")
            return true
        elseif max_increase_steps !== nothing && o.value > minloss
            increase_steps += 1
            if increase_steps >= max_increase_steps
                converged = true; steps = o.iteration
                banner(o, "loss increase")
                return true
            end
        else
            minloss = o.value
            increase_steps = 0
        end
        return false
    end
    lower, upper = fill(1e-6, size(guess)), fill(1e1, size(guess))
    res = Optim.optimize(Optim.only_fg!((F, G, x) -> fg!(G, x)), lower, upper, guess, Optim.Fminbox(optimizer()),
                         Optim.Options(callback=status_update))
    NHP.MaximumLikelihood(res.minimizer, -res.minimum, steps, res.time_run, converged ? "success" : "failure")
end

# d/dx of logprior(process) (src/continuous.jl:278-284) in params! order [λ0; θ | μ; τ; W]: the Gamma / normal-gamma
# log-densities of src/baselines.jl:120-122, src/impulses.jl:110-112,254-259, src/weights.jl:66-68 differentiated
function logprior_gradient(p::NHP.ContinuousStandardHawkesProcess)
    b, w, imp = p.baseline, p.weights, p.impulses
    g = [(b.α0 - 1) ./ b.λ .- b.β0]
    if imp isa NHP.ExponentialImpulseResponse
        push!(g, vec((imp.α - 1) ./ imp.θ .- imp.β))
    else
        push!(g, vec(-imp.κμ .* imp.τ .* (imp.μ .- imp.μμ)))
        push!(g, vec((imp.α0 - 1) ./ imp.τ .- imp.β0 .+ 0.5 ./ imp.τ .- 0.5 .* imp.κμ .* (imp.μ .- imp.μμ) .^ 2))
    end
    push!(g, vec((w.κ - 1) ./ w.W .- w.ν))
    vcat(g...)
end

# --- mle!(process, data; optimizer=BFGS, verbose=false, f_abstol=1e-6, regularize=false, guess=nothing)
#     -> MaximumLikelihood   src/continuous.jl:144-198 ------------------------------------------------------------
# Extra keywords (not in the reference): recursive (the loglikelihood dispatch), ctx / ds / comm (device handles), max_steps
# and optimizer=:device (the whole iteration inside the library, nhp_cont_mle_run).
function mle!(p::NHP.ContinuousStandardHawkesProcess, data; optimizer=Optim.BFGS, verbose=false, f_abstol=1e-6, regularize=false,
              guess=nothing, recursive=true, ctx=context(), comm=nothing, max_steps=1000,
              ds=Dataset(ctx, data, NHP.ndims(p), p.impulses.Δtmax))
    guess = guess === nothing ? NHP._rand_init_(p) : guess
    P = length(guess)
    flags = llflags(p, recursive)
    if optimizer === :device
        # the optimizer's state on the device (nhp_cont_mle_run: projected L-BFGS in HBM on the same box [1e-6, 10], the same
        # |f_k - f_{k-1}| < f_abstol rule; the host reads scalars).  At 2.1e6 parameters one step costs milliseconds instead
        # of the parameter upload + gradient download + host-side quasi-Newton update of the Optim route.
        regularize && error("optimizer=:device minimises -loglikelihood only; use an Optim optimizer with regularize=true")
        x = clamp.(Vector{Float64}(guess), 1e-6, 1e1)
        loss, steps, conv, evals = Ref{Float64}(0.0), Ref{Int32}(0), Ref{Int32}(0), Ref{Int32}(0)
        t0 = time()
        with_model(ctx, p) do m
            check(ccall((:nhp_cont_mle_run, libnhp), Int32,
                (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Int32, Float64, Float64, Float64, Int32, Ptr{Float64}, Int64,
                 Ref{Float64}, Ref{Int32}, Ref{Int32}, Ref{Int32}),
                ctx.h, comm === nothing ? C_NULL : comm.h, ds.h, m, flags, 1e-6, 1e1, f_abstol, Int32(max_steps), x, P,
                loss, steps, conv, evals), ctx.h)
        end
        verbose && println(" > steps: $(steps[]), objective evaluations: $(evals[]), loss: $(loss[])")
        NHP.params!(p, x)
        return NHP.MaximumLikelihood(x, -loss[], Int(steps[]), time() - t0, conv[] == 1 ? "success" : "failure")
    end
    res = with_model(ctx, p) do m
        ll, g = Ref{Float64}(0.0), Vector{Float64}(undef, P)
        function fg!(G, x)
            # params!(process, x) straight into the device-resident model (x already is the column-major parameter vector)
            check(ccall((:nhp_cont_model_set_params, libnhp), Int32, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Float64}, Int64), ctx.h, m, x, P), ctx.h)
            grad_call(ctx, comm, ds, m, flags, ll, g)
            f = -ll[]
            if regularize
                NHP.params!(p, x)
                f -= NHP.logprior(p)
                g .+= logprior_gradient(p)
            end
            G === nothing || (G .= .-g)
            f
        end
        run_mle(fg!, guess; optimizer=optimizer, verbose=verbose, f_abstol=f_abstol)
    end
    NHP.params!(p, res.maximizer)                      # "all inference methods overwrite model parameters"
    res
end

# --- mcmc!(process, data; nsteps=1000, log_freq=100, verbose=false) -> MarkovChainMonteCarlo  src/inference.jl:49-70
# A sweep -- parents, sufficient statistics, conjugate draws, (network) adjacency sweep and ρ -- stays on the device
# (nhp_cont_gibbs_step / nhp_cont_network_step); `push!(res.samples, params(process))` downloads the parameters every
# step exactly as the reference keeps them.  Extra keywords: seed (keys every Philox stream: reproducible chains,
# independent across seeds), keep_samples=false runs the chain inside the library (nhp_cont_mcmc_run: one
# synchronisation per log_freq steps, posterior moments accumulated on the device and returned by `moments`),
# ctx / ds / comm.  Draws are distributionally, not bitwise, those of Julia's samplers.
priors(p) = p.impulses isa NHP.ExponentialImpulseResponse ?
    Priors(p.baseline.α0, p.baseline.β0, p.weights.κ, p.weights.ν, p.impulses.α, p.impulses.β, 0.0, 1.0) :
    Priors(p.baseline.α0, p.baseline.β0, p.weights.κ, p.weights.ν, p.impulses.α0, p.impulses.β0, p.impulses.μμ, p.impulses.κμ)

function pull!(p::NHP.ContinuousHawkesProcess, ctx, m)     # device-resident model -> the mutable component structs
    N = NHP.ndims(p)
    nimp = N * N * (p.impulses isa NHP.ExponentialImpulseResponse ? 1 : 2)
    x = Vector{Float64}(undef, N + nimp + N * N)
    check(ccall((:nhp_cont_model_get_params, libnhp), Int32, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Float64}, Int64), ctx.h, m, x, length(x)), ctx.h)
    NHP.params!(p.baseline, x[1:N]); NHP.params!(p.impulses, x[N+1:N+nimp]); NHP.params!(p.weights, x[N+nimp+1:end])
    if p isa NHP.ContinuousNetworkHawkesProcess
        A = Matrix{Float64}(undef, N, N)
        check(ccall((:nhp_cont_model_get_adjacency, libnhp), Int32, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Float64}, Int64), ctx.h, m, A, N * N), ctx.h)
        p.adjacency_matrix = A
        if p.network isa NHP.BernoulliNetworkModel
            r = zeros(3)
            check(ccall((:nhp_cont_model_get_rho, libnhp), Int32, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Float64}), ctx.h, m, r), ctx.h)
            p.network.ρ = r[1]
        end
    end
end

# moments=true: the chain's posterior sums over the steps >= burn are kept on the device (nhp_cont_model_moments_*, what
# `gather_moments` exchanges between chains) and come back as the second value: (res, (sum, sumsq, count, rho)) -- fetched
# before the device model is released.  Without it no sum is accumulated (burn = -1 to the library).
function mcmc!(p::NHP.ContinuousHawkesProcess, data; nsteps=1000, log_freq=100, verbose=false, seed::UInt64=UInt64(0),
               keep_samples=true, moments=false, burn::Integer=0, ctx=context(), comm=nothing,
               ds=Dataset(ctx, data, NHP.ndims(p), p.impulses.Δtmax))
    p.baseline isa NHP.HomogeneousProcess || error("mcmc! on the device draws the homogeneous baseline; use the package's mcmc! with an LGCP baseline")
    res = NHP.MarkovChainMonteCarlo(p)
    mom = nothing
    start_time = time()
    network = p isa NHP.ContinuousNetworkHawkesProcess
    bern = network && p.network isa NHP.BernoulliNetworkModel
    na, nb = bern ? (Float64(p.network.α), Float64(p.network.β)) : (0.0, 0.0)      # 0, 0: ρ held at 1 (DenseNetworkModel)
    pr = Ref(priors(p))
    with_model(ctx, p) do m
        network && check(ccall((:nhp_cont_model_set_rho, libnhp), Int32, (Ptr{Cvoid}, Ptr{Cvoid}, Float64), ctx.h, m, bern ? p.network.ρ : 1.0), ctx.h)
        moments && check(ccall((:nhp_cont_model_moments_reset, libnhp), Int32, (Ptr{Cvoid}, Ptr{Cvoid}), ctx.h, m), ctx.h)
        while res.steps < nsteps
            n = keep_samples ? 1 : min(nsteps - res.steps, verbose ? log_freq : nsteps)
            check(ccall((:nhp_cont_mcmc_run, libnhp), Int32,
                        (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ref{Priors}, Float64, Float64, UInt64, UInt64, Int64, Int64),
                        ctx.h, commptr(comm), ds.h, m, pr, na, nb, seed, UInt64(res.steps), n, moments ? Int64(burn) : Int64(-1)), ctx.h)
            res.steps += n
            if keep_samples || res.steps == nsteps
                pull!(p, ctx, m)
                keep_samples && push!(res.samples, NHP.params(p))
            end
            if res.steps % log_freq == 0 && verbose
                res.elapsed = time() - start_time
                println(" > step: $(res.steps), elapsed: $(res.elapsed)")
            end
        end
        if moments                                          # while the model is alive
            N = NHP.ndims(p)
            len = N + N * N * (p.impulses isa NHP.ExponentialImpulseResponse ? 1 : 2) + N * N + (network ? N * N : 0)
            s, q, cnt, rho = Vector{Float64}(undef, len), Vector{Float64}(undef, len), Ref{Int64}(0), zeros(3)
            check(ccall((:nhp_cont_model_moments_fetch, libnhp), Int32, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Int64, Ref{Int64}),
                        ctx.h, m, s, q, len, cnt), ctx.h)
            network && check(ccall((:nhp_cont_model_get_rho, libnhp), Int32, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Float64}), ctx.h, m, rho), ctx.h)
            mom = (s, q, cnt[], rho)
        end
    end
    res.elapsed = time() - start_time
    return moments ? (res, mom) : res
end

# BASELINE config 5: after every rank ran its own chain with keep_samples=false, the per-chain posterior sums (still on
# the devices) all-gathered over RCCL: returns (sum, sumsq) as len x world matrices, the sample counts, and ρ's sums.
function gather_moments(ctx::Context, comm::Comm, m::Ptr{Cvoid}, len::Integer)
    s, q = Matrix{Float64}(undef, len, comm.world), Matrix{Float64}(undef, len, comm.world)
    counts, rho = Vector{Int64}(undef, comm.world), Matrix{Float64}(undef, 3, comm.world)
    check(ccall((:nhp_gather_moments, libnhp), Int32, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Int64, Ptr{Int64}, Ptr{Float64}),
                ctx.h, comm.h, m, s, q, len, counts, rho), ctx.h)
    s, q, counts, rho
end

# ======================================================================================================================
# Discrete half: src/discrete.jl:86-151,211-296,369-375; src/inference.jl:153-181; src/parents.jl:82-177
# ======================================================================================================================

# convolve(process, data) keeps Ŝ (T x N x B, 3.3 GB at BASELINE config 4) on the device: `Convolved` stands where the
# reference passes the `convolved` array, and `Array(c)` / `convolve(...; fetch=true)` gives the array itself.
mutable struct Convolved
    h::Ptr{Cvoid}; N::Int; T::Int; B::Int
    data::Matrix{Int64}
    host::Union{Nothing,Array{Float64,3}}
end

function basis_matrix(imp::NHP.DiscreteGaussianImpulseResponse)        # basis(impulse): L x B  src/impulses.jl:321-335
    L, B = imp.nlags, size(imp.θ, 3)
    phi = Matrix{Float64}(undef, L, B)
    check(ccall((:nhp_disc_basis, libnhp), Int32, (Int32, Int32, Float64, Ptr{Float64}), L, B, imp.dt, phi))
    phi
end

# --- convolve(process, data)  src/discrete.jl:146-151 ---------------------------------------------------------------
function convolve(p::NHP.DiscreteHawkesProcess, data::Matrix{Int64}; fetch=false, ctx=context())
    N, T = size(data)
    r = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:nhp_disc_dataset_create, libnhp), Int32, (Ptr{Cvoid}, Ptr{Int64}, Int32, Int64, Ref{Ptr{Cvoid}}), ctx.h, data, N, T, r), ctx.h)
    phi = basis_matrix(p.impulses)
    L, B = size(phi)
    host = fetch ? Array{Float64,3}(undef, T, N, B) : nothing
    GC.@preserve host check(ccall((:nhp_disc_convolve, libnhp), Int32, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Float64}, Int32, Int32, Ptr{Float64}),
                ctx.h, r[], phi, L, B, fetch ? pointer(host) : Ptr{Float64}(C_NULL)), ctx.h)
    c = Convolved(r[], N, T, B, data, host)
    finalizer(x -> ccall((:nhp_disc_dataset_destroy, libnhp), Cvoid, (Ptr{Cvoid},), x.h), c)
end

lowered(p::NHP.DiscreteHawkesProcess) = (Vector{Float64}(p.baseline.λ), Matrix{Float64}(p.weights.W), Array{Float64,3}(p.impulses.θ),
    p isa NHP.DiscreteNetworkHawkesProcess ? Matrix{Float64}(p.adjacency_matrix) : nothing)
aptr(A) = A === nothing ? Ptr{Float64}(C_NULL) : pointer(A)

# --- intensity(process, convolved) -> T x N  src/discrete.jl:115-131 --------------------------------------------------
function intensity(p::NHP.DiscreteHawkesProcess, c::Convolved; ctx=context())
    l0, W, θ, A = lowered(p)
    λ = Matrix{Float64}(undef, c.T, c.N)
    GC.@preserve A check(ccall((:nhp_disc_intensity, libnhp), Int32,
        (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Float64, Ptr{Float64}),
        ctx.h, c.h, l0, W, θ, aptr(A), p.dt, λ), ctx.h)
    λ
end

# --- loglikelihood(process, data[, convolved])  src/discrete.jl:86-102 ------------------------------------------------
function loglikelihood(p::NHP.DiscreteHawkesProcess, data::Matrix{Int64}, c::Convolved=convolve(p, data); ctx=context())
    l0, W, θ, A = lowered(p)
    ll = Ref{Float64}(0.0)
    GC.@preserve A check(ccall((:nhp_disc_loglik, libnhp), Int32,
        (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Float64, Ref{Float64}),
        ctx.h, c.h, l0, W, θ, aptr(A), p.dt, ll), ctx.h)
    ll[]
end

# --- update!(process, data, convolved): one mean-field step  src/discrete.jl:369-375 ----------------------------------
# (src/parents.jl:136-177 + src/baselines.jl:444-452, src/weights.jl:70-91, src/impulses.jl:355-371, fused: the
# T x N x (1+NB) responsibilities are never formed).  Overwrites the variational parameters of the components in place
# and returns variational_params(process) like the reference.  n_steps > 1 keeps them on the device in between.
function update!(p::NHP.DiscreteStandardHawkesProcess, data, c::Convolved; n_steps::Integer=1, ctx=context())
    b, w, imp = p.baseline, p.weights, p.impulses
    αv, βv = Vector{Float64}(b.αv), Vector{Float64}(b.βv)
    κv, νv, γv = Matrix{Float64}(w.κv), Matrix{Float64}(w.νv), Array{Float64,3}(imp.γv)
    check(ccall((:nhp_disc_vb_run, libnhp), Int32,
        (Ptr{Cvoid}, Ptr{Cvoid}, Float64, Float64, Float64, Float64, Float64, Float64, Int32,
         Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}),
        ctx.h, c.h, p.dt, b.α0, b.β0, w.κ, w.ν, imp.γ, n_steps, αv, βv, κv, νv, γv), ctx.h)
    b.αv, b.βv, w.κv, w.νv, imp.γv = αv, βv, κv, νv, γv
    NHP.variational_params(p)
end

# --- vb!(process, data; max_steps=1_000, Δx_thresh=1e-6, Δq_thresh=1e-2, verbose=false) -> VariationalInference
#     src/inference.jl:153-181 (its convergence test is commented out in the reference: max_steps updates always run)
function vb!(p::NHP.DiscreteStandardHawkesProcess, data::Matrix{Int64}; max_steps::Int64=1_000, Δx_thresh=1e-6, Δq_thresh=1e-2,
             verbose=false, ctx=context())
    convolved = convolve(p, data; ctx=ctx)
    res = NHP.VariationalInference(p)
    start_time = time()
    while res.step < max_steps
        push!(res.trace, update!(p, data, convolved; ctx=ctx))
        res.step += 1
    end
    res.elapsed = time() - start_time
    println(" ** maximum steps reached **")
    return res
end

# --- mle!(process::DiscreteStandardHawkesProcess, data; optimizer=BFGS, verbose=false, f_abstol=1e-6, regularize=false,
#          guess=nothing, max_increase_steps=3) -> MaximumLikelihood   src/discrete.jl:211-296 ---------------------------
# Parameter vector [λ0; vec(W .* θ)] (params / params!, src/discrete.jl:174-201).  regularize=true calls the reference's
# logprior(process), which reads fields the process does not have (SURVEY D5): it throws here as it does there.
function mle!(p::NHP.DiscreteStandardHawkesProcess, data::Matrix{Int64}; optimizer=Optim.BFGS, verbose=false, f_abstol=1e-6,
              regularize=false, guess=nothing, max_increase_steps=3, max_steps=1000, ctx=context())
    convolved = convolve(p, data; ctx=ctx)
    guess = guess === nothing ? NHP._rand_init_(p) : guess
    P = length(guess)
    if optimizer === :device
        # the whole iteration inside the library (nhp_disc_mle_run): x = params(process) stays on the device, params!'s split
        # into W and θ is redone there per evaluation; homogeneous baseline
        regularize && error("optimizer=:device minimises -loglikelihood only")
        x = clamp.(Vector{Float64}(guess), 1e-6, 1e1)
        loss, steps, conv, evals = Ref{Float64}(0.0), Ref{Int32}(0), Ref{Int32}(0), Ref{Int32}(0)
        t0 = time()
        check(ccall((:nhp_disc_mle_run, libnhp), Int32,
            (Ptr{Cvoid}, Ptr{Cvoid}, Float64, Float64, Float64, Float64, Int32, Ptr{Float64}, Int64, Ref{Float64}, Ref{Int32}, Ref{Int32}, Ref{Int32}),
            ctx.h, convolved.h, p.dt, 1e-6, 1e1, f_abstol, Int32(max_steps), x, P, loss, steps, conv, evals), ctx.h)
        verbose && println(" > steps: $(steps[]), objective evaluations: $(evals[]), loss: $(loss[])")
        NHP.params!(p, x)
        return NHP.MaximumLikelihood(x, -loss[], Int(steps[]), time() - t0, conv[] == 1 ? "success" : "failure")
    end
    ll, g = Ref{Float64}(0.0), Vector{Float64}(undef, P)
    function fg!(G, x)
        NHP.params!(p, x)
        l0, W, θ, _ = lowered(p)
        check(ccall((:nhp_disc_loglik_grad, libnhp), Int32,
            (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Float64, Ref{Float64}, Ptr{Float64}, Int64),
            ctx.h, convolved.h, l0, W, θ, p.dt, ll, g, P), ctx.h)
        f = -ll[]
        regularize && (f -= NHP.logprior(p))
        G === nothing || (G .= .-g)
        f
    end
    run_mle(fg!, guess; optimizer=optimizer, verbose=verbose, f_abstol=f_abstol, max_increase_steps=max_increase_steps)
end

# --- resample!(process, data, convolved): one discrete Gibbs sweep  src/discrete.jl:362-368,416-422 -------------------
# Parent counts (src/parents.jl:82-134, reduced straight to counts[N, 1+NB]) and the conjugate draws on the device; the
# network process then sweeps its adjacency matrix (src/discrete.jl:424-480) and redraws ρ (src/networks.jl:70-78).
function resample!(p::NHP.DiscreteHawkesProcess, data, c::Convolved; seed::UInt64=UInt64(0), step::UInt64=UInt64(0), ctx=context())
    b, w, imp = p.baseline, p.weights, p.impulses
    l0, W, θ, A = lowered(p)
    GC.@preserve A check(ccall((:nhp_disc_gibbs_step, libnhp), Int32,
        (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Float64, Float64, Float64, Float64, Float64, Float64, UInt64, UInt64),
        ctx.h, c.h, l0, W, θ, aptr(A), p.dt, b.α0, b.β0, w.κ, w.ν, imp.γ, seed, step), ctx.h)
    b.λ, w.W, imp.θ = l0, W, θ
    if p isa NHP.DiscreteNetworkHawkesProcess
        links = Ref{Float64}(0.0)
        ρ = p.network isa NHP.BernoulliNetworkModel ? Float64(p.network.ρ) : 1.0
        check(ccall((:nhp_disc_resample_adjacency, libnhp), Int32,
            (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Float64, Ptr{Float64}, Float64, Ptr{Float64}, UInt64, UInt64, Ref{Float64}),
            ctx.h, c.h, l0, W, θ, A, p.dt, C_NULL, ρ, C_NULL, seed, step, links), ctx.h)
        p.adjacency_matrix = A
        NHP.resample!(p.network, A)                                # ρ ~ Beta(α + ΣA, β + N² - ΣA): host, O(1)
    end
    NHP.params(p)
end

# --- mcmc!(process::DiscreteHawkesProcess, data; nsteps=1000, log_freq=100, verbose=false)  src/inference.jl:49-70 ----
function mcmc!(p::NHP.DiscreteHawkesProcess, data::Matrix{Int64}; nsteps=1000, log_freq=100, verbose=false, seed::UInt64=UInt64(0),
               ctx=context())
    res = NHP.MarkovChainMonteCarlo(p)
    start_time = time()
    convolved = convolve(p, data; ctx=ctx)
    while res.steps < nsteps
        push!(res.samples, resample!(p, data, convolved; seed=seed, step=UInt64(res.steps), ctx=ctx))
        res.steps += 1
        if res.steps % log_freq == 0 && verbose
            res.elapsed = time() - start_time
            println(" > step: $(res.steps), elapsed: $(res.elapsed)")
        end
    end
    res.elapsed = time() - start_time
    return res
end

end # module
