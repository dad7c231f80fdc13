# NetworkHawkesHIP.jl -- the reference-side binding of libnhp.so (include/nhp.h).
#
# Drop-in for the hot path only: `using NetworkHawkesProcesses; include("NetworkHawkesHIP.jl")`
# adds GPU methods that take the package's own process structs and data tuples and `ccall` the
# C ABI.  Nothing else of the package changes.  Julia is not installed in the build image, so
# this file is NOT executed by the test-suite; every entry point it binds is exercised through
# the identical ctypes binding (networkhawkesprocesses.jl_amd/_lib.py).  Keep it declarative.
module NetworkHawkesHIP

using NetworkHawkesProcesses
import Optim                      # already a dependency of NetworkHawkesProcesses (Project.toml)
const NHP = NetworkHawkesProcesses

const libnhp = get(ENV, "NHP_LIB", joinpath(@__DIR__, "..", "libnhp.so"))

# --- status -> exception (include/nhp.h: nhp_status) ---------------------------------------
function check(rc::Int32, ctx::Ptr{Cvoid}=C_NULL)
    rc == 0 && return
    msg = unsafe_string(ccall((:nhp_last_error, libnhp), Cstring, (Ptr{Cvoid},), ctx))
    rc == 2 && throw(DomainError(msg))            # NHP_EDOMAIN  (src/baselines.jl:100,106,111,116)
    rc == 3 && error(msg)                         # NHP_ESHAPE   (src/impulses.jl:44-45)
    error("libnhp status $rc: $msg")
end

mutable struct Context
    h::Ptr{Cvoid}
    function Context(device::Integer=parse(Int, get(ENV, "NHP_DEVICE", "0")))
        r = Ref{Ptr{Cvoid}}(C_NULL)
        check(ccall((:nhp_ctx_create, libnhp), Int32, (Int32, Ref{Ptr{Cvoid}}), device, r))
        ctx = new(r[])
        finalizer(c -> ccall((:nhp_ctx_destroy, libnhp), Cvoid, (Ptr{Cvoid},), c.h), ctx)
    end
end
const DEFAULT = Ref{Union{Nothing,Context}}(nothing)
context() = (DEFAULT[] === nothing && (DEFAULT[] = Context()); DEFAULT[])

# --- nhp_cont_model_desc: the lowered (Baseline, ImpulseResponse, Weights, A) ---------------
struct ModelDesc
    n_nodes::Int32; baseline_kind::Int32; lambda0::Ptr{Float64}; grid_x::Ptr{Float64}
    grid_n::Int32; impulse_kind::Int32; theta::Ptr{Float64}; mu::Ptr{Float64}; tau::Ptr{Float64}
    dt_max::Float64; W::Ptr{Float64}; A::Ptr{Float64}
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
# columns = 1-based node range this process evaluates (one loglikelihood over several GPUs: add the parts, e.g.
# MPI.Allreduce(ll, +, comm)); the default is the whole dataset
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

# --- loglikelihood(process, data; recursive=true)  src/continuous.jl:210,360 ----------------
function loglikelihood(p::NHP.ContinuousHawkesProcess, data; recursive=true, ctx=context(),
                       ds=Dataset(ctx, data, NHP.ndims(p), p.impulses.Δtmax))
    flags = Int32(recursive && p.impulses isa NHP.ExponentialImpulseResponse ? 1 : 0)
    ll = Ref{Float64}(0.0)
    with_model(ctx, p) do m
        check(ccall((:nhp_cont_loglik, libnhp), Int32, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Int32, Ref{Float64}),
                    ctx.h, ds.h, m, flags, ll), ctx.h)
    end
    ll[]
end

# --- intensity(process, data, times) -> length(times) x N  src/continuous.jl:76-96 -----------
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

# --- resample_parents(process, data) -> (parents, parentnodes)  src/parents.jl:1-23 ----------
function resample_parents(p::NHP.ContinuousHawkesProcess, data; seed::UInt64=UInt64(0), step::UInt64=UInt64(0),
                          ctx=context(), ds=Dataset(ctx, data, NHP.ndims(p), p.impulses.Δtmax))
    M = length(data[1])
    parents, parentnodes = Vector{Int64}(undef, M), Vector{Int64}(undef, M)
    with_model(ctx, p) do m
        check(ccall((:nhp_cont_resample_parents, libnhp), Int32,
                    (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Float64}, UInt64, UInt64, Ptr{Int64}, Ptr{Int64}, Ptr{Cvoid}),
                    ctx.h, ds.h, m, C_NULL, seed, step, parents, parentnodes, C_NULL), ctx.h)
    end
    parents, parentnodes
end

# --- gradient for mle!: objective/gradient pair for Optim.only_fg!  src/continuous.jl:144-198 -
function loglikelihood_gradient(p::NHP.ContinuousStandardHawkesProcess, data; recursive=true, ctx=context(),
                                ds=Dataset(ctx, data, NHP.ndims(p), p.impulses.Δtmax))
    P = length(NHP.params(p))
    g, ll = Vector{Float64}(undef, P), Ref{Float64}(0.0)
    flags = Int32(recursive && p.impulses isa NHP.ExponentialImpulseResponse ? 1 : 0)
    with_model(ctx, p) do m
        check(ccall((:nhp_cont_loglik_grad, libnhp), Int32,
                    (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Int32, Ref{Float64}, Ptr{Float64}, Int64),
                    ctx.h, ds.h, m, flags, ll, g, P), ctx.h)
    end
    ll[], g
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

# --- mle!(process, data; ...)  src/continuous.jl:144-198 --------------------------------------------------------
# Same objective, box [1e-6, 10] and stopping rule; the objective and its ANALYTIC gradient come from one call on a
# device-resident model whose parameters are overwritten in place with the optimiser's vector (the reference hands
# Optim no gradient, so Fminbox(BFGS) spends 2P log-likelihood calls on finite differences per step).
struct Priors   # nhp_gibbs_priors
    α0::Float64; β0::Float64; κ::Float64; ν::Float64; a::Float64; b::Float64; μμ::Float64; κμ::Float64
end

function mle!(p::NHP.ContinuousStandardHawkesProcess, data; f_abstol=1e-6, guess=nothing, recursive=true, ctx=context(),
              ds=Dataset(ctx, data, NHP.ndims(p), p.impulses.Δtmax), optimizer=nothing)
    x0 = guess === nothing ? NHP._rand_init_(p) : guess
    P = length(x0)
    flags = Int32(recursive && p.impulses isa NHP.ExponentialImpulseResponse ? 1 : 0)
    result = with_model(ctx, p) do m
        function fg!(F, G, x)
            check(ccall((:nhp_cont_model_set_params, libnhp), Int32, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Float64}, Int64),
                        ctx.h, m, x, P), ctx.h)
            ll, g = Ref{Float64}(0.0), Vector{Float64}(undef, P)
            check(ccall((:nhp_cont_loglik_grad, libnhp), Int32,
                        (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Int32, Ref{Float64}, Ptr{Float64}, Int64),
                        ctx.h, ds.h, m, flags, ll, g, P), ctx.h)
            G === nothing || (G .= .-g)
            return -ll[]
        end
        inner = optimizer === nothing ? Optim.LBFGS() : optimizer
        Optim.optimize(Optim.only_fg!(fg!), fill(1e-6, P), fill(10.0, P), clamp.(x0, 1e-6, 10.0), Optim.Fminbox(inner),
                       Optim.Options(f_abstol=f_abstol))
    end
    NHP.params!(p, Optim.minimizer(result))            # "all inference methods overwrite model parameters"
    result
end

# --- mcmc!(process, data; nsteps)  src/inference.jl:49-70 -------------------------------------------------------
# A sweep -- parents, sufficient statistics, conjugate draws, (network) adjacency -- stays on the device; what comes
# back per step is what the caller asks for: nothing (posterior moments accumulate on the device), or params(process).
priors(p) = p.impulses isa NHP.ExponentialImpulseResponse ?
    Priors(p.baseline.α0, p.baseline.β0, p.weights.κ, p.weights.ν, p.impulses.α, p.impulses.β, 0.0, 1.0) :
    Priors(p.baseline.α0, p.baseline.β0, p.weights.κ, p.weights.ν, p.impulses.α0, p.impulses.β0, p.impulses.μμ, p.impulses.κμ)

function mcmc!(p::NHP.ContinuousHawkesProcess, data; nsteps=1000, seed::UInt64=UInt64(0), keep_samples=false, ctx=context(),
               ds=Dataset(ctx, data, NHP.ndims(p), p.impulses.Δtmax))
    N = NHP.ndims(p)
    network = p isa NHP.ContinuousNetworkHawkesProcess
    nimp = N * N * (p.impulses isa NHP.ExponentialImpulseResponse ? 1 : 2)
    L = N + nimp + N * N
    samples = Vector{Vector{Float64}}()
    pr = Ref(priors(p))
    sum1, sum2, count = zeros(L + (network ? N * N : 0)), zeros(L + (network ? N * N : 0)), Ref{Int64}(0)
    with_model(ctx, p) do m
        check(ccall((:nhp_cont_model_moments_reset, libnhp), Int32, (Ptr{Cvoid}, Ptr{Cvoid}), ctx.h, m), ctx.h)
        for step in 0:nsteps-1
            check(ccall((:nhp_cont_gibbs_step, libnhp), Int32, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ref{Priors}, UInt64, UInt64),
                        ctx.h, ds.h, m, pr, seed, UInt64(step)), ctx.h)
            if network
                links = Ref{Float64}(0.0)
                last = keep_samples || step == nsteps - 1
                Aout = last ? Matrix{Float64}(undef, N, N) : nothing
                GC.@preserve Aout check(ccall((:nhp_cont_resample_adjacency, libnhp), Int32,
                            (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Float64}, Float64, Ptr{Float64}, UInt64, UInt64, Ptr{Float64}, Ref{Float64}),
                            ctx.h, ds.h, m, C_NULL, p.network.ρ, C_NULL, seed, UInt64(step),
                            last ? pointer(Aout) : Ptr{Float64}(C_NULL), links), ctx.h)
                last && (p.adjacency_matrix = Aout)
                # resample_connection_probability!: ρ ~ Beta(α + links, β + N² - links)  src/networks.jl:70-76
                p.network.ρ = rand(NHP.Beta(p.network.α + links[], p.network.β + N * N - links[]))
            end
            check(ccall((:nhp_cont_model_moments_accumulate, libnhp), Int32, (Ptr{Cvoid}, Ptr{Cvoid}), ctx.h, m), ctx.h)
            if keep_samples || step == nsteps - 1
                x = Vector{Float64}(undef, L)
                check(ccall((:nhp_cont_model_get_params, libnhp), Int32, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Float64}, Int64), ctx.h, m, x, L), ctx.h)
                keep_samples && push!(samples, x)
                if step == nsteps - 1                                # in place, component by component: [λ0; θ | μ; τ; W]
                    NHP.params!(p.baseline, x[1:N]); NHP.params!(p.impulses, x[N+1:N+nimp]); NHP.params!(p.weights, x[N+nimp+1:end])
                end
            end
        end
        check(ccall((:nhp_cont_model_moments_fetch, libnhp), Int32,
                    (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Int64, Ref{Int64}),
                    ctx.h, m, sum1, sum2, length(sum1), count), ctx.h)
    end
    (samples=samples, mean=sum1 ./ max(count[], 1), m2=sum2 ./ max(count[], 1), n=count[])
end

end # module
