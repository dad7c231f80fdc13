// Log-likelihood + analytic gradient in params! order [λ0; θ | μ; τ; W].
//
// The reference hands Optim no gradient (src/continuous.jl:190), so every BFGS gradient inside
// mle! costs it 2P finite-difference objective calls (P = N + 2N²).  These kernels
// produce the exact gradient in about two log-likelihood passes.  With g_i = 1/λ_i:
//   ∂/∂λ0[c]  = -T + Σ_{i on c} g_i
//   ∂/∂W[p,c] = -cnt[p]·mask[p,c] + a[p,c] Σ_{i on c} g_i Σ_{j∈win(i), n_j=p} ħ(Δt_ij)
//   ∂/∂θ[p,c] =  a·w Σ_i g_i Σ_j ∂ħ/∂θ,      ∂ħ/∂θ = (1 - θΔ) e^{-θΔ}                (exponential)
//   ∂ħ/∂μ = ħ·τ(ℓ-μ),  ∂ħ/∂τ = ħ·(1 - τ(ℓ-μ)²)/(2τ),  ℓ = logit(Δ/Δtmax)            (logit-normal)
// Validated against central finite differences of the oracle's log-likelihood in tests/.
//
// Windowed form: pass A is the log-likelihood kernel itself, which also stores λ_i; pass B
// revisits each child's window with g_i known and accumulates Σ g·ħ and Σ g·∂ħ per parent node
// in LDS (the child's node c is fixed per workgroup, so the accumulators are a column), then
// scales by the column's constants and adds the column into the gradient.
// Recursive form: the column-state recursion of cont_recursive.hip, extended with the
// derivative state R_pc(t) = Σ_j (t - t_j) e^{-θ(t - t_j)}:  ∂λ/∂θ[p,c] = a·w·(S - θR).
#include <algorithm>

#include "nhp_internal.h"
#include "nhp_math.h"

template <int G>
__device__ __forceinline__ double ggroup_sum(double v)
{
#pragma unroll
    for (int off = G / 2; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

__device__ __forceinline__ double grad_baseline_at(const nhp_cont_args &a, int c, double t)
{
#pragma clang fp contract(off)
    if (a.baseline_kind == NHP_BASELINE_HOMOGENEOUS) return a.lambda0[c];
    const double *x = a.grid;
    const double *y = a.lambda0 + (size_t)c * a.grid_n;
    int lo = 0, hi = a.grid_n - 1;
    if (!(t < x[hi])) return y[hi];
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (t >= x[mid]) lo = mid; else hi = mid;
    }
    return (y[lo + 1] * (t - x[lo]) + y[lo] * (x[lo + 1] - t)) / (x[lo + 1] - x[lo]);
}

// params(baseline) is λ (N) or vcat(λ...) (N·G grid intensities of the LGCP): src/baselines.jl:41,173
__device__ __forceinline__ size_t grad_nbase(const nhp_cont_args &a)
{
    return a.baseline_kind == NHP_BASELINE_HOMOGENEOUS ? (size_t)a.N : (size_t)a.N * (size_t)a.grid_n;
}

// ∂ log λ_i / ∂(grid intensities of node c) for an event at time t: g = 1/λ_i spread over the two grid
// neighbours with the interpolation weights (src/utils/interpolation.jl:26-35)
__device__ __forceinline__ void grad_lgcp_scatter(const nhp_cont_args &a, int c, double t, double g, double *grad)
{
    const double *x = a.grid;
    const int G = a.grid_n;
    double *gc = grad + (size_t)c * G;
    int lo = 0, hi = G - 1;
    if (!(t < x[hi])) { atomicAdd(&gc[hi], g); return; }
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (t >= x[mid]) lo = mid; else hi = mid;
    }
    const double w = x[lo + 1] - x[lo];
    atomicAdd(&gc[lo], g * (x[lo + 1] - t) / w);
    atomicAdd(&gc[lo + 1], g * (t - x[lo]) / w);
}

// grad <- the parameter-independent terms: -T (or minus the trapezoid weights), 0, -cnt[p]·mask
__global__ __launch_bounds__(256) void k_grad_init(nhp_cont_args a, int mask_integral, double *__restrict__ grad)
{
    const size_t N = (size_t)a.N, NN = N * N;
    const size_t nimp = a.impulse_kind == NHP_IMPULSE_EXPONENTIAL ? NN : 2 * NN;
    const size_t nb = grad_nbase(a);
    const size_t P = nb + nimp + NN;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < P; i += (size_t)gridDim.x * blockDim.x) {
        double v = 0.0;
        // a column shard owns the terms of its child nodes only; everything else stays 0 so that shards add up
        const size_t col = i < nb ? (a.baseline_kind == NHP_BASELINE_HOMOGENEOUS ? i : i / (size_t)a.grid_n)
                                  : ((i - nb) % NN) / N;
        if (col < (size_t)a.col_begin || col >= (size_t)a.col_end) {
            grad[i] = 0.0;
            continue;
        }
        if (i < nb) {
            if (a.baseline_kind == NHP_BASELINE_HOMOGENEOUS) {
                v = -a.duration;
            } else {                                             // d/dy_g of the trapezoid rule (ignores duration)
                const int g = (int)(i % (size_t)a.grid_n), G = a.grid_n;
                const double left = g > 0 ? a.grid[g] - a.grid[g - 1] : 0.0, right = g + 1 < G ? a.grid[g + 1] - a.grid[g] : 0.0;
                v = -0.5 * (left + right);
            }
        } else if (i >= nb + nimp) {
            const size_t k = i - nb - nimp;
            const double mk = (a.A && mask_integral) ? a.A[k] : 1.0;
            v = -a.cnt[k % N] * mk;
        }
        grad[i] = v;
    }
}

// TH threads per item: the pass is two LDS atomics and one exponential per pair behind a column staged once; at long windows
// (the recursive objective's truncated window: 300-600 pairs per event) a column keeps 512 threads busy.
template <int IMP, int G, int TH>
__global__ __launch_bounds__(TH) void k_grad_windowed(nhp_cont_args a, const double *__restrict__ lambda,
                                                             double *__restrict__ grad)
{
    extern __shared__ __align__(16) unsigned char smem[];
    double *red = reinterpret_cast<double *>(smem);
    double2 *col = reinterpret_cast<double2 *>(smem + 64);       // exp {θ, -θ·64/ln 2}; logit {μ, sqrt τ}
    double *accH = reinterpret_cast<double *>(col + a.N);        // Σ g·ħ
    double *acc1 = accH + a.N;                                   // Σ g·∂ħ/∂θ  |  Σ g·ħ·sqrtτ·z
    double *acc2 = acc1 + a.N;                                   // logit: Σ g·ħ·(1 - z²)
    double *etab = acc1 + a.N;                                   // exponential (no acc2): [64] 2^(j/64), 512 bytes counted by the launcher
    if (IMP == NHP_IMPULSE_EXPONENTIAL) nhp_exp_tab_init(etab);

    const nhp_item it = a.items[blockIdx.x];
    const int c = it.node, N = a.N, tid = threadIdx.x;
    for (int p = tid; p < N; p += TH) {
        const size_t k = (size_t)p + (size_t)c * N;
        if (IMP == NHP_IMPULSE_EXPONENTIAL) col[p] = make_double2(a.p1[k], -(a.p1[k] * 92.33248261689366));
        else col[p] = make_double2(a.p1[k], __builtin_sqrt(a.p2[k]));
        accH[p] = 0.0; acc1[p] = 0.0;
        if (IMP != NHP_IMPULSE_EXPONENTIAL) acc2[p] = 0.0;
    }
    __syncthreads();

    constexpr int GROUPS = TH / G;
    const int gid = tid / G, gl = tid % G;
    double gsum = 0.0;
    for (int k = it.kbeg + gid; k < it.kend; k += GROUPS) {
        const nhp_child ch = a.child[k];
        const double g = 1.0 / lambda[ch.idx];
        if (gl == 0) {
            if (a.baseline_kind == NHP_BASELINE_HOMOGENEOUS) gsum += g;
            else grad_lgcp_scatter(a, c, ch.t, g, grad);
        }
        if (IMP == NHP_IMPULSE_EXPONENTIAL) {
            // one 16-byte record per parent, the next one requested before this one's arithmetic; col = {θ, -θ·64/ln 2}
            int j = ch.idx - 1 - gl;
            nhp_event e0 = a.ev[j > 0 ? j : 0];
            for (; j >= ch.first; j -= G) {
                const nhp_event en = a.ev[j - G > 0 ? j - G : 0];
                asm volatile("" ::: "memory");
                const double dt = ch.t - e0.t;
                const int p = e0.node;
                const double2 q = col[p];
                const double e = nhp_exp_neg_tab_scaled(q.y * dt, etab);
                atomicAdd(&accH[p], g * (q.x * e));
                atomicAdd(&acc1[p], g * ((1.0 - q.x * dt) * e));
                e0 = en;
            }
        } else
        for (int j = ch.idx - 1 - gl; j >= ch.first; j -= G) {
            const double dt = ch.t - a.times[j];
            const int p = a.nodes[j];
            const double2 q = col[p];
            {
                const double x = dt * a.inv_dtmax;
                if (x > 0.0 && x < 1.0) {
                    const double o = 1.0 - x, qq = 1.0 / (x * o);
                    const double z = (nhp_log((x * x) * qq) - q.x) * q.y;
                    const double h = (nhp_exp_neg(-0.5 * (z * z)) * (NHP_INVSQRT2PI * q.y)) * qq;
                    atomicAdd(&accH[p], g * h);
                    atomicAdd(&acc1[p], g * h * (q.y * z));
                    atomicAdd(&acc2[p], g * h * (1.0 - z * z));
                }
            }
        }
    }
    __syncthreads();
    const size_t Nn = grad_nbase(a), NN = (size_t)N * (size_t)N;      // Nn: offset of the impulse block
    const size_t nimp = IMP == NHP_IMPULSE_EXPONENTIAL ? NN : 2 * NN;
    for (int p = tid; p < N; p += TH) {
        const size_t k = (size_t)p + (size_t)c * N;
        const double av = a.A ? a.A[k] : 1.0;
        const double aw = av * a.W[k];
        if (accH[p] != 0.0) atomicAdd(&grad[Nn + nimp + k], av * accH[p]);
        if (IMP == NHP_IMPULSE_EXPONENTIAL) {
            if (acc1[p] != 0.0) atomicAdd(&grad[Nn + k], aw * acc1[p]);
        } else {
            if (acc1[p] != 0.0) atomicAdd(&grad[Nn + k], aw * acc1[p]);
            if (acc2[p] != 0.0) atomicAdd(&grad[Nn + NN + k], aw * (0.5 / a.p2[k]) * acc2[p]);
        }
    }
    const double gs = nhp_block_sum_n<TH / 64>(gsum, red);
    if (tid == 0 && gs != 0.0 && a.baseline_kind == NHP_BASELINE_HOMOGENEOUS) atomicAdd(&grad[c], gs);
}

// ---- recursive exponential: ll and gradient in one pass ---------------------------------------
#define GR_RING 64
__global__ __launch_bounds__(NHP_BLOCK) void k_grad_recursive(nhp_cont_args a, double *__restrict__ partials,
                                                              double *__restrict__ grad)
{
    extern __shared__ __align__(16) unsigned char smem[];
    double *red = reinterpret_cast<double *>(smem);              // [4]
    double *wpart = red + 4;                                     // [NHP_WAVES]
    double *th = wpart + NHP_WAVES;                              // θ[p,c]
    double *wth = th + a.N;                                      // (a·w)·θ
    double *S = wth + a.N, *R = S + a.N;                         // state and derivative state at the last child
    double *nS = R + a.N, *nR = nS + a.N;                        // segment accumulators referenced to t_k
    double *GS = nR + a.N, *GR = GS + a.N;                       // Σ_k g_k S_p(t_k), Σ_k g_k R_p(t_k)

    const int c = a.col_begin + blockIdx.x, N = a.N, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    double integ = 0.0;
    for (int p = tid; p < N; p += NHP_BLOCK) {
        const size_t k = (size_t)p + (size_t)c * N;
        const double w = a.W[k], weff = a.A ? a.A[k] * w : w, t = a.p1[k];
        th[p] = t; wth[p] = weff * t;
        S[p] = R[p] = nS[p] = nR[p] = GS[p] = GR[p] = 0.0;
        integ += a.cnt[p] * w;
    }
    __syncthreads();
    const int kb = a.boff[c], ke = a.boff[c + 1];
    int prev_idx = 0;
    double prev_t = 0.0, logsum = 0.0, gsum = 0.0;
    for (int k = kb; k < ke; ++k) {
        const nhp_child ch = a.child[k];
        // four packed event records per thread in flight, exponentials evaluated unconditionally, atomics predicated
        // (the scheme of k_recursive's fold)
        for (int j0 = prev_idx + tid; j0 < ch.idx; j0 += 4 * NHP_BLOCK) {
            nhp_event ev[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) ev[u] = a.ev[j0 + u * NHP_BLOCK < ch.idx ? j0 + u * NHP_BLOCK : j0];
            double d[4], e[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { d[u] = ch.t - ev[u].t; e[u] = nhp_exp_neg(-(th[ev[u].node] * d[u])); }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (j0 + u * NHP_BLOCK < ch.idx && ev[u].t > 0.0) {
                    atomicAdd(&nS[ev[u].node], e[u]);
                    atomicAdd(&nR[ev[u].node], d[u] * e[u]);
                }
        }
        __syncthreads();
        const double gap = ch.t - prev_t;
        double part = 0.0;
        for (int p = tid; p < N; p += NHP_BLOCK) {
            double s = S[p], r = R[p];
            if (k != kb) {
                const double dec = nhp_exp_neg(-(th[p] * gap));
                r = dec * (r + gap * s);
                s = dec * s;
            }
            s += nS[p]; r += nR[p];
            nS[p] = 0.0; nR[p] = 0.0;
            S[p] = s; R[p] = r;
            part += wth[p] * s;
        }
        part = nhp_wave_sum(part);
        if (lane == 0) wpart[wave] = part;
        __syncthreads();
        double lam = grad_baseline_at(a, c, ch.t);
        for (int w = 0; w < NHP_WAVES; ++w) lam += wpart[w];
        const double g = 1.0 / lam;
        for (int p = tid; p < N; p += NHP_BLOCK) { GS[p] += g * S[p]; GR[p] += g * R[p]; }
        if (tid == 0) {
            logsum += nhp_log(lam);
            if (a.baseline_kind == NHP_BASELINE_HOMOGENEOUS) gsum += g;
            else grad_lgcp_scatter(a, c, ch.t, g, grad);
        }
        prev_idx = ch.idx;
        prev_t = ch.t;
    }
    __syncthreads();
    const size_t Nn = grad_nbase(a), NN = (size_t)N * (size_t)N;
    for (int p = tid; p < N; p += NHP_BLOCK) {
        const size_t k = (size_t)p + (size_t)c * N;
        const double av = a.A ? a.A[k] : 1.0, w = a.W[k], t = th[p];
        grad[Nn + NN + k] = -a.cnt[p] + av * t * GS[p];          // unmasked integral (D7)
        grad[Nn + k] = av * w * (GS[p] - t * GR[p]);
    }
    const double blk = nhp_block_sum(logsum, red);
    const double blk_int = nhp_block_sum(integ, red);
    if (tid == 0) {
        partials[2 * (size_t)blockIdx.x] = blk;
        partials[2 * (size_t)blockIdx.x + 1] = blk_int;
        if (a.baseline_kind == NHP_BASELINE_HOMOGENEOUS) grad[c] = -a.duration + gsum;
    }
}

// ---- recursive exponential, gradient pass of the wave-partitioned recursion -------------------------------------------
// k_recursive_waves (cont_recursive.hip) has evaluated the log-likelihood and left g_k = 1/λ_k of every child; with g known
// the gradient needs no sum across parents any more: wave h of column c runs the same recursion on ITS parents -- state
// S_p(t_k) = Σ_j e^{-θ(t_k - t_j)} and its θ-derivative state R_p(t_k) = Σ_j (t_k - t_j) e^{-θ(t_k - t_j)}, folded from the
// per-part event lists (one exp, two LDS atomics per event) and decayed per child (one exp per parent) -- and accumulates
// GS_p = Σ_k g_k S_p(t_k), GR_p = Σ_k g_k R_p(t_k) in registers.  No reduction, no ring, no barrier after the table is staged.
// Reference: the objective of mle! (src/continuous.jl:144-198) differentiated by hand; same sums as k_grad_recursive.
template <int PQ, int H>
__global__ __launch_bounds__(64 * H) void k_grad_recursive_waves(nhp_cont_args a, nhp_rec_parts rp, const double *__restrict__ ginv,
                                                                 double *__restrict__ grad)
{
    constexpr int NP = 64 * PQ;
    extern __shared__ __align__(16) unsigned char smem[];
    double *tab = reinterpret_cast<double *>(smem);             // [64] 2^(j/64)
    const int tid = threadIdx.x, lane = tid & 63, h = tid >> 6;
    double *th = tab + 64 + (size_t)h * 3 * NP;                  // [NP] -θ[p,c]·64/ln2 of this wave's parents
    double *accS = th + NP, *accR = accS + NP;                   // [NP] segment accumulators (this wave's alone)
    const int c = a.col_begin + blockIdx.x, N = a.N;
    const double K64 = 92.33248261689366;                        // 64 / ln 2

    nhp_exp_tab_init(tab);
    double S[PQ], R[PQ], thr[PQ], GS[PQ], GR[PQ];
#pragma unroll
    for (int q = 0; q < PQ; ++q) {
        const int pl = lane + 64 * q, p = h * NP + pl;
        S[q] = R[q] = GS[q] = GR[q] = 0.0;
        thr[q] = p < N ? -(a.p1[(size_t)p + (size_t)c * N] * K64) : 0.0;
        th[pl] = thr[q];
        accS[pl] = 0.0; accR[pl] = 0.0;
    }
    __syncthreads();                                             // the table

    const nhp_event *ev = rp.ev + rp.poff[h];
    const int32_t *rk = rp.rank + (size_t)h * (size_t)a.M;
    const int kb = a.boff[c], ke = a.boff[c + 1];

    auto term = [&](const nhp_event &e, bool live, double tk) {
        const double d = tk - e.t;
        const double v = nhp_exp_neg_tab_scaled(th[e.node] * d, tab);
        if (live) { atomicAdd(&accS[e.node], v); atomicAdd(&accR[e.node], d * v); }
    };
    auto fold = [&](nhp_event e, int j, int je, double tk) {      // as in k_recursive_waves
        int r = j - lane;
        while (r < je) {
            const nhp_event en = ev[j + 64];
            asm volatile("" ::: "memory");
            term(e, j < je, tk);
            r += 64;
            if (r >= je) break;
            e = ev[j + 128];
            asm volatile("" ::: "memory");
            term(en, j + 64 < je, tk);
            r += 64; j += 128;
        }
    };

    double ch_t = 0.0, nx_t = 0.0, gsum = 0.0;
    int ch_r = 0, nx_r = 0;
    if (kb < ke) { ch_t = a.child[kb].t; ch_r = rk[kb]; nx_t = ch_t; nx_r = ch_r; }
    if (kb + 1 < ke) { nx_t = a.child[kb + 1].t; nx_r = rk[kb + 1]; }
    if (kb < ke) fold(ev[lane], lane, ch_r, ch_t);
    double prev_t = ch_t;
    for (int k = kb; k < ke; ++k) {
        const int kn = k + 2 < ke ? k + 2 : ke - 1;
        const double nn_t = a.child[kn].t;
        const int nn_r = rk[kn];
        const double g = ginv[k];
        const bool more = k + 1 < ke;
        const int fj = ch_r + lane, fe = more ? nx_r : ch_r;
        const nhp_event pe = ev[fj];
        NHP_RECW_SYNC();
        const double gap = ch_t - prev_t;
#pragma unroll
        for (int q = 0; q < PQ; ++q) {
            const int pl = lane + 64 * q;
            const double dec = nhp_exp_neg_tab_scaled(thr[q] * gap, tab);
            const double r = dec * (R[q] + gap * S[q]) + accR[pl];
            const double s = dec * S[q] + accS[pl];
            accS[pl] = 0.0; accR[pl] = 0.0;
            S[q] = s; R[q] = r;
            GS[q] += g * s; GR[q] += g * r;
        }
        if (h == 0) {                                            // the baseline's share, once per child
            if (a.baseline_kind == NHP_BASELINE_HOMOGENEOUS) gsum += g;
            else if (lane == 0) grad_lgcp_scatter(a, c, ch_t, g, grad);
        }
        fold(pe, fj, fe, nx_t);
        prev_t = ch_t;
        ch_t = nx_t; ch_r = nx_r;
        nx_t = nn_t; nx_r = nn_r;
    }
    const size_t Nn = grad_nbase(a), NN = (size_t)N * (size_t)N;
#pragma unroll
    for (int q = 0; q < PQ; ++q) {
        const int p = h * NP + lane + 64 * q;
        if (p < N) {
            const size_t k = (size_t)p + (size_t)c * N;
            const double av = a.A ? a.A[k] : 1.0, w = a.W[k], t = a.p1[k];
            grad[Nn + NN + k] = -a.cnt[p] + av * t * GS[q];      // unmasked integral (D7)
            grad[Nn + k] = av * w * (GS[q] - t * GR[q]);
        }
    }
    if (tid == 0 && a.baseline_kind == NHP_BASELINE_HOMOGENEOUS) grad[c] = -a.duration + gsum;
}

template <int IMP, int TH>
static void launch_grad_group(int G, dim3 grid, size_t lds, hipStream_t st, const nhp_cont_args &a,
                              const double *lambda, double *grad)
{
#define NHP_CASE(g)                                                                                      \
    case g:                                                                                              \
        if (lds > 64 * 1024)                                                                             \
            (void)hipFuncSetAttribute((const void *)k_grad_windowed<IMP, g, TH>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        hipLaunchKernelGGL((k_grad_windowed<IMP, g, TH>), grid, dim3(TH), lds, st, a, lambda, grad);     \
        break;
    switch (G) {
        NHP_CASE(1) NHP_CASE(2) NHP_CASE(4) NHP_CASE(8) NHP_CASE(16) NHP_CASE(32)
    default:
        if (lds > 64 * 1024)
            (void)hipFuncSetAttribute((const void *)k_grad_windowed<IMP, 64, TH>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL((k_grad_windowed<IMP, 64, TH>), grid, dim3(TH), lds, st, a, lambda, grad);
    }
#undef NHP_CASE
}

// Enqueue log-likelihood (-> ctx->d_results[0]) and gradient (-> *d_grad_out, P doubles inside ctx->d_scratch; the two
// doubles in front of it are spare, so the multi-GPU path can put the log-likelihood at (*d_grad_out)[-1] and all-reduce
// [ll; grad] as one vector).
nhp_status nhp_grad_enqueue(nhp_ctx *ctx, const nhp_cont_dataset *ds, const nhp_cont_model *m, int32_t flags, int64_t grad_len,
                            double **d_grad_out)
{
    NHP_TRY(nhp_check_pair(ctx, ds, m));
    const size_t N = (size_t)ds->N, NN = N * N;
    const bool exp_imp = m->impulse_kind == NHP_IMPULSE_EXPONENTIAL;
    const bool lgcp = m->baseline_kind != NHP_BASELINE_HOMOGENEOUS;
    const size_t P = (lgcp ? N * (size_t)m->grid_n : N) + (exp_imp ? 2 : 3) * NN;
    if ((size_t)grad_len != P) { nhp_set_error(ctx, "Parameter vector length does not match model parameter length."); return NHP_ESHAPE; }
    NHP_HIP(ctx, hipSetDevice(ctx->device));
    const size_t M = (size_t)(ds->M > 0 ? ds->M : 1);
    NHP_TRY(nhp_ctx_reserve_scratch(ctx, 8 * (2 + P + M)));
    double *d_grad = (double *)ctx->d_scratch + 2, *d_lambda = d_grad + P;
    *d_grad_out = d_grad;
    nhp_cont_args a = nhp_make_args(ds, m);
    hipStream_t st = ctx->stream;
    // the recursive formulation through its truncated window when the bound allows (cont_recursive.hip): the windowed
    // gradient on the other window starts, with the recursion's unmasked integral
    const nhp_child *child_cut = nullptr;
    int cut_group = 0;
    if ((flags & NHP_LL_RECURSIVE) && !(flags & NHP_LL_FULL_RECURSION) && exp_imp)
        NHP_TRY(nhp_recursive_window(ctx, ds, m, &child_cut, &cut_group, 1.2));
    if ((flags & NHP_LL_RECURSIVE) && exp_imp && !child_cut) {
        // two passes of the wave-partitioned recursion: log-likelihood + 1/λ of every child, then the gradient sums
        int PQ = 0, H = 0;
        nhp_rec_parts rp{};
        NHP_TRY(nhp_rec_parts_for(ctx, ds, &PQ, &H, &rp));
        if (PQ) {
            bool launched = false;
            if (lgcp || nhp_is_column_shard(ds)) {
                hipLaunchKernelGGL(k_grad_init, dim3(1024), dim3(256), 0, st, a, 0, d_grad);
                NHP_HIP(ctx, hipGetLastError());
            }
            NHP_TRY(nhp_launch_recursive_waves(ctx, ds, m, ctx->d_results, d_lambda, &launched));
            if (launched) {
                const size_t wl = sizeof(double) * (64 + (size_t)H * 3 * 64 * PQ);
                const int ncol = ds->col_end - ds->col_begin;
#define NHP_GRW(Q, HH)                                                                                                           \
                do {                                                                                                             \
                    if (wl > 64 * 1024)                                                                                          \
                        (void)hipFuncSetAttribute((const void *)k_grad_recursive_waves<Q, HH>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)wl); \
                    hipLaunchKernelGGL((k_grad_recursive_waves<Q, HH>), dim3((unsigned)ncol), dim3(64 * HH), wl, st, a, rp, d_lambda, d_grad); \
                } while (0)
                NHP_REC_SHAPES(NHP_GRW, launched = false);
#undef NHP_GRW
                if (launched) {
                    NHP_HIP(ctx, hipGetLastError());
                    *d_grad_out = d_grad;
                    return NHP_OK;
                }
            }
        }
        const size_t lds = 8 * (4 + NHP_WAVES + 8 * N);
        if (lds > 160 * 1024) { nhp_set_error(ctx, "recursive gradient: n_nodes = %d exceeds the LDS budget", ds->N); return NHP_ENOTIMPL; }
        if (lds > 64 * 1024)
            NHP_HIP(ctx, hipFuncSetAttribute((const void *)k_grad_recursive, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        NHP_TRY(nhp_ctx_reserve_partials(ctx, 2 * N));
        // the grid-intensity block is accumulated with atomics: start it from the integral's derivative; a column shard
        // needs the zeros of the columns it does not own
        if (lgcp || nhp_is_column_shard(ds)) {
            hipLaunchKernelGGL(k_grad_init, dim3(1024), dim3(256), 0, st, a, 0, d_grad);
            NHP_HIP(ctx, hipGetLastError());
        }
        const int ncol = ds->col_end - ds->col_begin;
        hipLaunchKernelGGL(k_grad_recursive, dim3((unsigned)ncol), dim3(NHP_BLOCK), lds, st, a, ctx->d_partials, d_grad);
        NHP_HIP(ctx, hipGetLastError());
        NHP_TRY(nhp_launch_finalize(ctx, a, ncol, ctx->d_results));
    } else {
        const int mask = child_cut ? 0 : 1;
        const int G = child_cut ? cut_group : ds->group;
        // exponential impulses on the dataset's own short / middle windows: ONE launch over the child and the parent slices
        // (cont_slices.hip: no LDS atomics, no λ round trip through HBM; k_grad_init only where items share a column)
        if (!child_cut && exp_imp && ds->d_sl_row && !(getenv("NHP_SLICES") && atoi(getenv("NHP_SLICES")) == 0) &&
            !(getenv("NHP_GRAD_SLICES") && atoi(getenv("NHP_GRAD_SLICES")) == 0)) {
            if (!nhp_grad_slices_direct(ds, m)) {
                hipLaunchKernelGGL(k_grad_init, dim3(1024), dim3(256), 0, st, a, mask, d_grad);
                NHP_HIP(ctx, hipGetLastError());
            }
            bool launched = false;
            NHP_TRY(nhp_launch_grad_slices(ctx, ds, m, ctx->d_results, d_grad, &launched));
            if (launched) return NHP_OK;
        }
        if (child_cut) { a.child = child_cut; a.child_w = child_cut; }
        hipLaunchKernelGGL(k_grad_init, dim3(1024), dim3(256), 0, st, a, mask, d_grad);
        NHP_HIP(ctx, hipGetLastError());
        NHP_TRY(nhp_launch_event_intensity_as(ctx, ds, m, child_cut, G, mask, d_lambda));   // pass A: partials + λ_i
        NHP_TRY(nhp_launch_finalize(ctx, a, ds->n_items, ctx->d_results));
        const size_t lds = 64 + 16 * N + 8 * N * (exp_imp ? 2 : 3) + (exp_imp ? 512 : 0);     // + the exponential's 2^(j/64) table
        if (lds > 160 * 1024) { nhp_set_error(ctx, "gradient: n_nodes = %d exceeds the 160 KiB LDS budget", ds->N); return NHP_ENOTIMPL; }
        dim3 grid((unsigned)ds->n_items);
        // threads per item: 512 where an item has the pairs for them (NHP_GRAD_THREADS overrides)
        const double pairs_per_item = (double)(child_cut ? ds->cut_pairs : ds->pairs) / (double)std::max(1, ds->n_items);
        static const int forced = getenv("NHP_GRAD_THREADS") ? atoi(getenv("NHP_GRAD_THREADS")) : 0;
        const int TH = forced == 256 || forced == 512 ? forced : (pairs_per_item >= 4096.0 ? 512 : 256);
        if (exp_imp && TH == 512) launch_grad_group<NHP_IMPULSE_EXPONENTIAL, 512>(G, grid, lds, st, a, d_lambda, d_grad);
        else if (exp_imp) launch_grad_group<NHP_IMPULSE_EXPONENTIAL, 256>(G, grid, lds, st, a, d_lambda, d_grad);
        else if (TH == 512) launch_grad_group<NHP_IMPULSE_LOGITNORMAL, 512>(G, grid, lds, st, a, d_lambda, d_grad);
        else launch_grad_group<NHP_IMPULSE_LOGITNORMAL, 256>(G, grid, lds, st, a, d_lambda, d_grad);
        NHP_HIP(ctx, hipGetLastError());
    }
    return NHP_OK;
}

extern "C" nhp_status nhp_cont_loglik_grad(nhp_ctx *ctx, const nhp_cont_dataset *ds, const nhp_cont_model *m,
                                           int32_t flags, double *ll, double *grad, int64_t grad_len)
{
    if (!ll || !grad) return NHP_EINVAL;
    double *d_grad = nullptr;
    NHP_TRY(nhp_grad_enqueue(ctx, ds, m, flags, grad_len, &d_grad));
    NHP_TRY(nhp_download(ctx, grad, d_grad, 8 * (size_t)grad_len));
    return nhp_ctx_fetch(ctx, 0, 1, ll);
}
