// Discrete-time Gibbs sampling on the GPU: parent counts of a sweep and the adjacency-matrix
// sweep of the network process.  The shared pieces -- dataset, bump table staging, the intensity GEMM -- live
// in disc.hip (declared in nhp_internal.h).
#include <math.h>

#include <algorithm>
#include <vector>

#include "nhp_internal.h"
#include "nhp_math.h"
#include "nhp_rng.h"

// ---- discrete Gibbs parent counts (reference resample_parents / resample_parent
// src/parents.jl:82-116 reduced over time to counts[c + N·k] = Σ_t parents[t, c, k], which is all the
// discrete resample! methods read: src/baselines.jl:413-419, src/weights.jl:28-35,
// src/impulses.jl:337-353).  Only occupied bins draw anything, and a bin's Multinomial(n, μ) is n
// categorical draws over 1 + N·B categories, i.e. an inverse-CDF walk along the bin's row of
// Z = base ⊕ G·E.  The reference materialises parents[T, N, 1+NB]; here a workgroup takes a 64-bin x
// 128-node tile, lists its occupied bins, and walks the category axis twice in chunks staged through
// LDS (G rows and E rows shared by all the tile's bins): once for the row total, once comparing the
// running sum with the bin's ascending thresholds u_(1) < u_(2) < ... (order statistics generated one
// at a time from Philox, so n events cost one walk).  One lane owns a bin's running sum, in the
// reference's category order with separate multiply and add: counts equal the oracle's, bit for bit.
#define RP_KC 16
#define RP_KEY 0xD15C0DE5EEDC0FFEull

__device__ __attribute__((noinline)) double rp_next_u(double u_prev, int remaining, uint64_t seed, uint64_t step, uint64_t bin, int j)
{
#pragma clang fp contract(off)
    const double V = nhp_philox_uniform(seed ^ RP_KEY, step, (bin << 20) | (uint64_t)j);
    const double r = nhp_exp(nhp_log(V) / (double)remaining);
    const double w = 1.0 - r;
    return u_prev + (1.0 - u_prev) * w;
}

// RP_SLOTS = occupied bins a thread carries through one pair of walks: the host picks the smallest of 1, 2, 4 whose
// RP_TH·RP_SLOTS slots hold a tile's occupied bins (an empty slot costs the walk as much as a full one: at config 4 a
// 64 x 128 tile holds ~400, and two slots per thread take 14.7 ms where four took 20.9).
// RP_TT x RP_CT = the tile (bins x child nodes), RP_TH threads.  What a launch moves is the staging traffic: per chunk of RP_KC
// categories a tile fetches (RP_TT + RP_CT)·RP_KC doubles for RP_TT·RP_CT·rate occupied bins -- 78 GB per sweep at config 4 with
// 64 x 128 tiles (5.6 TB/s out of L2 / the Infinity Cache: the bound of the 14 ms launch), half of that with 128 x 256.
template <int RP_SLOTS, int RP_TT, int RP_CT, int RP_TH>
__global__ __launch_bounds__(RP_TH) void k_disc_resample_parents(const double *__restrict__ dataT, const double *__restrict__ conv,
                                                               const double *__restrict__ E2, const double *__restrict__ base,
                                                               const double *__restrict__ baseT, int64_t T, int N, int B,
                                                               unsigned b_magic, uint64_t seed, uint64_t step,
                                                               int *__restrict__ counts, int *__restrict__ base_counts)
{
#pragma clang fp contract(off)
    extern __shared__ __align__(16) unsigned char rp_smem[];
    double (*Gt)[RP_TT] = reinterpret_cast<double (*)[RP_TT]>(rp_smem);                                   // [RP_KC][RP_TT]
    double (*Et)[RP_CT + 1] = reinterpret_cast<double (*)[RP_CT + 1]>(rp_smem + 8 * RP_KC * RP_TT);      // [RP_KC][RP_CT + 1]
    unsigned short *list = reinterpret_cast<unsigned short *>(rp_smem + 8 * RP_KC * (RP_TT + RP_CT + 1)); // [RP_TT * RP_CT]
    __shared__ int nb, wcnt[RP_TH / 64];
    const int tid = threadIdx.x, K = N * B;
    const int64_t t0 = (int64_t)blockIdx.x * RP_TT;
    const int c0 = blockIdx.y * RP_CT;
    // Occupied bins, listed bin-row by bin-row (entry = tl·RP_CT + cl): consecutive lanes then share a
    // bin row, so a wave's reads of a G row collapse to a few broadcast addresses and its reads of an E
    // row hit distinct banks.  Flags are gathered with coalesced loads (t fastest), then compacted in order.
    unsigned char *occ = reinterpret_cast<unsigned char *>(&Et[0][0]);       // [RP_TT][RP_CT], before Et is used
    for (int i = tid; i < RP_TT * RP_CT; i += RP_TH) {
        const int tl_ = i % RP_TT, cl_ = i / RP_TT;
        const int64_t t = t0 + tl_;
        const int c = c0 + cl_;
        occ[tl_ * RP_CT + cl_] = (t < T && c < N && dataT[(size_t)t + (size_t)T * c] > 0.0) ? 1 : 0;
    }
    __syncthreads();
    {
        const int lane = tid & 63, wave = tid >> 6;
        constexpr int PER_WAVE = RP_TT * RP_CT / (RP_TH / 64);
        int cnt = 0;
        for (int i = wave * PER_WAVE + lane; i < (wave + 1) * PER_WAVE; i += 64) cnt += __popcll(__ballot(occ[i] != 0));
        if (lane == 0) wcnt[wave] = cnt;
        __syncthreads();
        int off = 0;
        for (int w = 0; w < wave; ++w) off += wcnt[w];
        for (int i = wave * PER_WAVE + lane; i < (wave + 1) * PER_WAVE; i += 64) {
            const bool f = occ[i] != 0;
            const unsigned long long m = __ballot(f);
            if (f) list[off + __popcll(m & ((1ull << lane) - 1ull))] = (unsigned short)i;
            off += __popcll(m);
        }
        if (tid == RP_TH - 1) nb = off;
    }
    __syncthreads();
    const int nbins = nb;

    // A chunk of RP_KC categories is fetched into registers one chunk ahead (under the previous chunk's
    // arithmetic) and written to LDS between two barriers.
    constexpr int GN = RP_KC * RP_TT / RP_TH, EN = RP_KC * RP_CT / RP_TH;
    static_assert(GN >= 1 && EN >= 1 && RP_TT * RP_CT <= 65536 && RP_TT * RP_CT <= 8 * RP_KC * (RP_CT + 1), "tile shape");
    static_assert(RP_TH % RP_TT == 0 && RP_TH % RP_KC == 0, "staging coordinates below assume these");
    double rg[GN], re[EN];
    // staging coordinates are fixed per thread: G element r is (category row tid/RP_TT + (RP_TH/RP_TT)·r,
    // bin tid % RP_TT), E element r is (category tid % RP_KC, node tid/RP_KC + (RP_TH/RP_KC)·r).
    // Category q = p·B + b reads Ŝ[t, p, b]; q / B is a multiply-high by the host-made reciprocal.
    const int g_tt = tid % RP_TT, g_k0 = tid / RP_TT, e_kk = tid % RP_KC, e_c0 = tid / RP_KC;
    const bool g_ok = t0 + g_tt < T;
    const double *g_base = conv + (size_t)(g_ok ? t0 + g_tt : 0);
    const unsigned T32 = (unsigned)T;
    // loads are unconditional from clamped (always valid) addresses and zeroed by a select afterwards:
    // predicated loads compile to one exec-mask branch each, and twelve of them per chunk spill SGPRs
    unsigned gmask = 0, emask = 0;            // which of the fetched values are real (applied when staged,
    auto fetch = [&](int q0) {                // so that the loads stay in flight under the arithmetic)
        gmask = 0; emask = 0;
#pragma unroll
        for (int r = 0; r < GN; ++r) {
            const int qr = q0 + g_k0 + (RP_TH / RP_TT) * r;
            const unsigned q = (unsigned)(qr < K ? qr : K - 1);
            const unsigned pq = B == 1 ? q : __umulhi(q, b_magic), bq = q - pq * (unsigned)B;
            rg[r] = g_base[(size_t)T32 * (size_t)(pq + (unsigned)N * bq)];
            gmask |= (qr < K && g_ok ? 1u : 0u) << r;
        }
#pragma unroll
        for (int r = 0; r < EN; ++r) {
            const int cr = c0 + e_c0 + (RP_TH / RP_KC) * r, qr = q0 + e_kk;
            re[r] = E2[(size_t)(qr < K ? qr : K - 1) + (size_t)(cr < N ? cr : N - 1) * K];
            emask |= (qr < K && cr < N ? 1u : 0u) << r;
        }
    };
    auto stage = [&]() {
#pragma unroll
        for (int r = 0; r < GN; ++r) { const int e = tid + RP_TH * r; Gt[e / RP_TT][e % RP_TT] = (gmask >> r) & 1u ? rg[r] : 0.0; }
#pragma unroll
        for (int r = 0; r < EN; ++r) { const int e = tid + RP_TH * r; Et[e % RP_KC][e / RP_KC] = (emask >> r) & 1u ? re[r] : 0.0; }
    };

    for (int b0 = 0; b0 < nbins; b0 += RP_TH * RP_SLOTS) {
        int tl[RP_SLOTS], cl[RP_SLOTS], n[RP_SLOTS], j[RP_SLOTS];
        double cum[RP_SLOTS], total[RP_SLOTS], thr[RP_SLOTS], u[RP_SLOTS];
#pragma unroll
        for (int s = 0; s < RP_SLOTS; ++s) {
            const int idx = b0 + tid + RP_TH * s;
            const int e = idx < nbins ? list[idx] : 0;
            tl[s] = e / RP_CT; cl[s] = e % RP_CT;
            n[s] = idx < nbins ? (int)dataT[(size_t)(t0 + tl[s]) + (size_t)T * (c0 + cl[s])] : 0;
            j[s] = 0;
            cum[s] = n[s] > 0 ? (baseT ? baseT[(size_t)(t0 + tl[s]) + (size_t)T * (c0 + cl[s])] : base[c0 + cl[s]]) : 0.0;
            total[s] = 0.0; thr[s] = 0.0; u[s] = 0.0;
        }
        // ---- walk 1: row totals
        fetch(0);
        for (int q0 = 0; q0 < K; q0 += RP_KC) {
            __syncthreads();
            stage();
            __syncthreads();
            if (q0 + RP_KC < K) fetch(q0 + RP_KC);
#pragma unroll 4
            for (int kk = 0; kk < RP_KC; ++kk) {
#pragma unroll
                for (int s = 0; s < RP_SLOTS; ++s) cum[s] = cum[s] + Gt[kk][tl[s]] * Et[kk][cl[s]];
            }
        }
        // first thresholds; the baseline category
#pragma unroll
        for (int s = 0; s < RP_SLOTS; ++s) {
            total[s] = cum[s];
            if (n[s] > 0) {
                const int c = c0 + cl[s];
                const uint64_t bin = (uint64_t)(t0 + tl[s]) + (uint64_t)T * (uint64_t)c;
                u[s] = rp_next_u(0.0, n[s], seed, step, bin, 0);
                thr[s] = u[s] * total[s];
                cum[s] = baseT ? baseT[(size_t)(t0 + tl[s]) + (size_t)T * c] : base[c];
                while (j[s] < n[s] && cum[s] > thr[s]) {
                    atomicAdd(&counts[c], 1);
                    if (base_counts) atomicAdd(&base_counts[(size_t)(t0 + tl[s]) + (size_t)T * c], 1);
                    if (++j[s] < n[s]) { u[s] = rp_next_u(u[s], n[s] - j[s], seed, step, bin, j[s]); thr[s] = u[s] * total[s]; }
                }
            }
            if (j[s] >= n[s]) thr[s] = __builtin_inf();                      // nothing (left) to place: walk 2 never stops here
        }
        // ---- walk 2: categories by inverse CDF
        fetch(0);
        for (int q0 = 0; q0 < K; q0 += RP_KC) {
            __syncthreads();
            stage();
            __syncthreads();
            if (q0 + RP_KC < K) fetch(q0 + RP_KC);
            for (int kk = 0; kk < RP_KC; ++kk) {
#pragma unroll
                for (int s = 0; s < RP_SLOTS; ++s) {
                    cum[s] = cum[s] + Gt[kk][tl[s]] * Et[kk][cl[s]];
                    // ONE test per multiply-add: a bin with nothing left to place carries thr = +inf, and categories
                    // past K are staged as zeros (the sum cannot pass a threshold there that it had not passed before)
                    if (cum[s] > thr[s]) {
                        const int c = c0 + cl[s];
                        const uint64_t bin = (uint64_t)(t0 + tl[s]) + (uint64_t)T * (uint64_t)c;
                        do {
                            atomicAdd(&counts[(size_t)c + (size_t)N * (1 + q0 + kk)], 1);
                            if (++j[s] < n[s]) { u[s] = rp_next_u(u[s], n[s] - j[s], seed, step, bin, j[s]); thr[s] = u[s] * total[s]; }
                            else thr[s] = __builtin_inf();
                        } while (cum[s] > thr[s]);
                    }
                }
            }
        }
#pragma unroll
        for (int s = 0; s < RP_SLOTS; ++s)                                    // capped at the last category
            if (j[s] < n[s]) atomicAdd(&counts[(size_t)(c0 + cl[s]) + (size_t)N * K], n[s] - j[s]);
    }
}

// Runs the parent-count sweep; *d_counts_out (int [N*(1+K)], index c + N*k) stays in ctx scratch until the next call.
static nhp_status disc_parent_counts(nhp_ctx *ctx, const nhp_disc_dataset *ds, const double *lambda0, const double *W,
                                     const double *theta, const double *A, double dt, uint64_t seed, uint64_t step,
                                     size_t extra_doubles, int **d_counts_out, double **extra_out, double **dW_out,
                                     double **dth_out, double **dl0_out)
{
    NHP_HIP(ctx, hipSetDevice(ctx->device));
    const size_t N = (size_t)ds->N, NN = N * N, K = N * (size_t)ds->B, NC = N * (1 + K);
    double *E2, *base, *extra;
    NHP_TRY(nhp_disc_stage_bump(ctx, ds, lambda0, W, theta, A, dt, &E2, &base, (NC + 1) / 2 + 1 + extra_doubles, &extra, 1));
    int *d_counts = reinterpret_cast<int *>(extra);
    hipStream_t st = ctx->stream;
    NHP_HIP(ctx, hipMemsetAsync(d_counts, 0, sizeof(int) * NC, st));
    if (ds->T >= ((int64_t)1 << 31) || K >= ((size_t)1 << 24)) { nhp_set_error(ctx, "resample_parents: T or N*B too large"); return NHP_ENOTIMPL; }
    // q / B for q < 2^24 as a multiply-high: exact with magic = floor(2^32 / B) + 1 while q·B < 2^32
    const unsigned b_magic = (unsigned)((((uint64_t)1 << 32) / (uint64_t)ds->B + 1) & 0xFFFFFFFFu);   // unused for B = 1
    if (ds->d_base_counts) NHP_HIP(ctx, hipMemsetAsync(ds->d_base_counts, 0, sizeof(int) * (size_t)ds->T * N, st));
    // tile (bins x nodes, threads): the larger one halves the staging traffic per occupied bin where the problem fills it;
    // NHP_RP_TILE = "TT,CT,THREADS" overrides (64,128,256 | 128,128,512 | 128,256,1024), NHP_RP_SLOTS the slots per thread
    int TT = 64, CT = 128, TH = 256;
    if (N >= 256 && ds->T >= 128 * 256) { TT = 128; CT = 256; TH = 1024; }
    if (const char *ts = getenv("NHP_RP_TILE")) sscanf(ts, "%d,%d,%d", &TT, &CT, &TH);
    if (!((TT == 64 && CT == 128 && TH == 256) || (TT == 128 && CT == 128 && TH == 512) || (TT == 128 && CT == 256 && TH == 1024))) { TT = 64; CT = 128; TH = 256; }
    dim3 grid((unsigned)((ds->T + TT - 1) / TT), (unsigned)((N + CT - 1) / CT));
    // occupied bins of a tile: the mean plus three standard deviations (a tile that overflows its slots walks twice)
    const double mean = (double)ds->nocc * (double)(TT * CT) / ((double)ds->T * (double)std::max<size_t>(N, (size_t)CT));
    const double need = mean + 3.0 * sqrt(mean);
    const char *fs = getenv("NHP_RP_SLOTS");
    const int slots = fs ? atoi(fs) : (need <= 1.0 * TH ? 1 : need <= 2.0 * TH ? 2 : 4);
    const size_t lds = 8 * (size_t)RP_KC * (size_t)(TT + CT + 1) + 2 * (size_t)TT * CT;
#define RP_LAUNCH(S, tt, ct, th)                                                                                                  \
    do {                                                                                                                          \
        if (lds > 64 * 1024)                                                                                                      \
            (void)hipFuncSetAttribute((const void *)k_disc_resample_parents<S, tt, ct, th>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        hipLaunchKernelGGL((k_disc_resample_parents<S, tt, ct, th>), grid, dim3(th), lds, st, ds->d_dataT, ds->d_conv, E2, base,  \
                           lambda0 ? nullptr : ds->d_baseT, ds->T, ds->N, ds->B, b_magic, seed, step, d_counts, ds->d_base_counts); \
    } while (0)
#define RP_TILE(tt, ct, th) do { if (slots == 1) RP_LAUNCH(1, tt, ct, th); else if (slots == 2) RP_LAUNCH(2, tt, ct, th); else RP_LAUNCH(4, tt, ct, th); } while (0)
    if (TT == 64) RP_TILE(64, 128, 256); else if (CT == 128) RP_TILE(128, 128, 512); else RP_TILE(128, 256, 1024);
#undef RP_TILE
#undef RP_LAUNCH
    if (ds->d_base_counts) const_cast<nhp_disc_dataset *>(ds)->base_counts_valid = true;
    NHP_HIP(ctx, hipGetLastError());
    *d_counts_out = d_counts;
    if (extra_out) *extra_out = extra + (NC + 1) / 2 + 1;
    // stage_bump's layout after E: base (N) | λ0 (N) | W (N²) | θ (N²B) | A (N²) | extra
    if (dl0_out) *dl0_out = base + N;
    if (dW_out) *dW_out = base + 2 * N;
    if (dth_out) *dth_out = base + 2 * N + NN;
    return NHP_OK;
}

extern "C" nhp_status nhp_disc_resample_parents(nhp_ctx *ctx, const nhp_disc_dataset *ds, const double *lambda0,
                                                const double *W, const double *theta, const double *A, double dt,
                                                uint64_t seed, uint64_t step, int64_t *counts)
{
    if (!ctx || !ds || !counts) return NHP_EINVAL;
    const size_t N = (size_t)ds->N, NC = N * (1 + N * (size_t)ds->B);
    int *d_counts;
    NHP_TRY(disc_parent_counts(ctx, ds, lambda0, W, theta, A, dt, seed, step, 0, &d_counts, nullptr, nullptr, nullptr, nullptr));
    hipStream_t st = ctx->stream;
    std::vector<int> h((size_t)NC);
    NHP_HIP(ctx, hipMemcpyAsync(h.data(), d_counts, sizeof(int) * NC, hipMemcpyDeviceToHost, st));
    NHP_HIP(ctx, hipStreamSynchronize(st));
    for (size_t i = 0; i < NC; ++i) counts[i] = h[i];
    return NHP_OK;
}

// ---- conjugate draws of the discrete resample! methods on the device (src/baselines.jl:413-419 as intended (D2),
// src/weights.jl:59-64, src/impulses.jl:337-353): λ0_c ~ Gamma(α0 + counts[c, 0], 1/(β0 + T·dt));
// W[p,c] ~ Gamma(κ + Σ_b counts[c, 1+pB+b], 1/(ν + Σ_t data[p, t])); θ[p,c,:] ~ Dirichlet(γ + counts[c, 1+pB+·])
// as normalised Gamma(·, 1) draws.  Philox-keyed like the continuous draws: distributional parity with Julia.
__global__ __launch_bounds__(256) void k_disc_gibbs_draw(int N, int B, double Tdt, double alpha0, double beta0, double kappa,
                                                         double nu, double gamma0, uint64_t seed, uint64_t step,
                                                         const int *__restrict__ counts, const double *__restrict__ node_counts,
                                                         double *__restrict__ lambda0, double *__restrict__ W,
                                                         double *__restrict__ theta)
{
    const size_t NN = (size_t)N * N;
    const size_t pc = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (pc < (size_t)N)
        lambda0[pc] = dev_gamma(alpha0 + (double)counts[pc], 1.0 / (beta0 + Tdt), seed ^ 0x243F6A8885A308D3ull, step, pc);
    if (pc >= NN) return;
    const size_t p = pc % N, c = pc / N;
    double m = 0.0, gs = 0.0;
    for (int b = 0; b < B; ++b) {
        const double cnt = (double)counts[c + (size_t)N * (1 + p * B + b)];
        m += cnt;
        const double gv = dev_gamma(gamma0 + cnt, 1.0, seed ^ 0xA4093822299F31D0ull, step, pc + (size_t)b * NN);
        theta[pc + (size_t)b * NN] = gv;
        gs += gv;
    }
    for (int b = 0; b < B; ++b) theta[pc + (size_t)b * NN] /= gs;
    W[pc] = dev_gamma(kappa + m, 1.0 / (nu + node_counts[p]), seed ^ 0x13198A2E03707344ull, step, pc);
}

// resample!(process::DiscreteStandardHawkesProcess, data, convolved) src/discrete.jl:362-368 in one call: parent
// counts, then the conjugate draws, all on the device; lambda0 [N], W [N*N], theta [N*N*B] are read and overwritten.
extern "C" nhp_status nhp_disc_gibbs_step(nhp_ctx *ctx, const nhp_disc_dataset *ds, double *lambda0, double *W,
                                          double *theta, const double *A, double dt, double alpha0, double beta0,
                                          double kappa, double nu, double gamma0, uint64_t seed, uint64_t step)
{
    if (!ctx || !ds || !lambda0 || !W || !theta) return NHP_EINVAL;
    const size_t N = (size_t)ds->N, NN = N * N, B = (size_t)ds->B;
    int *d_counts;
    double *dW, *dth, *dl0;
    NHP_TRY(disc_parent_counts(ctx, ds, lambda0, W, theta, A, dt, seed, step, 0, &d_counts, nullptr, &dW, &dth, &dl0));
    hipStream_t st = ctx->stream;
    // Σ_t data[p, t] = node_counts(data): the first half of the dataset's column statistics
    hipLaunchKernelGGL(k_disc_gibbs_draw, dim3((unsigned)((NN + 255) / 256)), dim3(256), 0, st, (int)N, (int)B, (double)ds->T * dt,
                       alpha0, beta0, kappa, nu, gamma0, seed, step, d_counts, ds->d_colsum, dl0, dW, dth);
    NHP_HIP(ctx, hipGetLastError());
    NHP_HIP(ctx, hipMemcpyAsync(lambda0, dl0, 8 * N, hipMemcpyDeviceToHost, st));
    NHP_HIP(ctx, hipMemcpyAsync(W, dW, 8 * NN, hipMemcpyDeviceToHost, st));
    NHP_HIP(ctx, hipMemcpyAsync(theta, dth, 8 * NN * B, hipMemcpyDeviceToHost, st));
    NHP_HIP(ctx, hipStreamSynchronize(st));
    return NHP_OK;
}

// ---- discrete adjacency Gibbs sweep (reference resample_adjacency_matrix! / resample_column!
// / conditional_loglikelihood src/discrete.jl:424-480).  For entry (p, c) the reference evaluates two full
// Poisson log-likelihoods of column c over all T bins and all N·B parent terms.  Their difference is
//     ll1 - ll0 = Σ_{t: s>0} s_tc [log(λ⁰_tc + x_t) - log λ⁰_tc] - Σ_t x_t + log ρ - log(1-ρ),
//     x_t = W[p,c] dt Σ_b Ŝ[t,p,b] θ[p,c,b],   λ⁰ = the intensity with A[p,c] = 0,
// where only OCCUPIED bins need a log and Σ_t x_t = Σ_b (W θ dt)[p,c,b] · Σ_t Ŝ[t,p,b] uses per-dataset column
// sums.  Columns are independent, entries of a column sequential in p -- so the sweep is N steps, each over
// all occupied bins of all columns at once: k_dadj_accum (time-tiled: the Ŝ[·, p, ·] slice of the tile sits
// in LDS, each lane owns occupied bins, per-column sums collect in LDS and leave as one row of partials per
// tile) then k_dadj_decide (adds the tiles in fixed order, draws A[p, ·], and prepares step p+1).  λ of the
// occupied bins is carried incrementally.
__global__ __launch_bounds__(256) void k_dadj_gather(const double *__restrict__ lam, const int32_t *__restrict__ occ_t,
                                                     const int32_t *__restrict__ occ_c, int64_t nocc, int64_t T,
                                                     double *__restrict__ lam_occ, double *__restrict__ xprev)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < nocc) { lam_occ[i] = lam[(size_t)occ_t[i] + (size_t)T * occ_c[i]]; xprev[i] = 0.0; }
}

// V[c*B + b] = W[p,c] θ[p,c,b] dt,  a_p[c] = A[p,c]   (row p of the tables, gathered once per step)
__device__ __forceinline__ void dadj_prep_row(int p, int c, int N, int B, double dt, const double *W, const double *theta,
                                              const double *A, double *V, double *a_p)
{
    const size_t pc = (size_t)p + (size_t)c * N;
    for (int b = 0; b < B; ++b) V[(size_t)c * B + b] = (W[pc] * theta[pc + (size_t)b * N * N]) * dt;
    a_p[c] = A[pc];
}

__global__ __launch_bounds__(256) void k_dadj_prep(int p, int N, int B, double dt, const double *__restrict__ W,
                                                   const double *__restrict__ theta, const double *__restrict__ A,
                                                   double *__restrict__ V, double *__restrict__ a_p, double *__restrict__ dprev)
{
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c < N) { dadj_prep_row(p, c, N, B, dt, W, theta, A, V, a_p); dprev[c] = 0.0; }
}

__global__ __launch_bounds__(256) void k_dadj_accum(int p, int N, int B, int64_t T, const double *__restrict__ conv,
                                                    const int32_t *__restrict__ occ_t, const int32_t *__restrict__ occ_c,
                                                    const double *__restrict__ occ_s, const int32_t *__restrict__ occ_off,
                                                    const double *__restrict__ V, const double *__restrict__ a_p,
                                                    const double *__restrict__ dprev, double *__restrict__ lam_occ,
                                                    double *__restrict__ xprev, double *__restrict__ partial)
{
    extern __shared__ __align__(16) double dsm[];
    double *Gt = dsm;                    // [B][NHP_DA_TT]
    double *acc = dsm + (size_t)B * NHP_DA_TT;   // [N]
    const int tid = threadIdx.x;
    const int64_t t0 = (int64_t)blockIdx.x * NHP_DA_TT;
    for (int e = tid; e < B * NHP_DA_TT; e += 256) {
        const int b = e / NHP_DA_TT, tt = e % NHP_DA_TT;
        const int64_t t = t0 + tt;
        Gt[e] = t < T ? conv[(size_t)t + (size_t)T * ((size_t)p + (size_t)N * b)] : 0.0;
    }
    for (int c = tid; c < N; c += 256) acc[c] = 0.0;
    __syncthreads();
    for (int i = occ_off[blockIdx.x] + tid; i < occ_off[blockIdx.x + 1]; i += 256) {
        const int c = occ_c[i], tt = occ_t[i] - (int)t0;
        double lam = lam_occ[i];
        const double dp = dprev[c];
        if (dp != 0.0) { lam += dp * xprev[i]; lam_occ[i] = lam; }       // entry (p-1, c) flipped: carry it into λ
        double x = 0.0;
        const double *v = V + (size_t)c * B;
        for (int b = 0; b < B; ++b) x += Gt[b * NHP_DA_TT + tt] * v[b];
        xprev[i] = x;
        if (x > 0.0) {
            const double l0 = lam - a_p[c] * x;
            atomicAdd(&acc[c], occ_s[i] * (nhp_log(l0 + x) - nhp_log(l0)));
        }
    }
    __syncthreads();
    for (int c = tid; c < N; c += 256) partial[(size_t)blockIdx.x * N + c] = acc[c];
}

__global__ __launch_bounds__(256) void k_dadj_decide(int p, int N, int B, int ntiles, double dt, const double *__restrict__ W,
                                                     const double *__restrict__ theta, double *__restrict__ A,
                                                     const double *__restrict__ partial, const double *__restrict__ convsum,
                                                     const double *__restrict__ rho_mat, double rho_scalar,
                                                     const double *__restrict__ u, uint64_t seed, uint64_t step,
                                                     double *__restrict__ V, double *__restrict__ a_p, double *__restrict__ dprev)
{
    // 32 columns per workgroup, 8 lane-groups per column: group g adds tiles g, g+8, ... (four independent
    // chains in flight), then the eight group sums are added in order -- a fixed summation tree
    __shared__ double gsum[8][32];
    const int cl = threadIdx.x & 31, grp = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + cl;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    if (c < N) {
        int k = grp;
        for (; k + 24 < ntiles; k += 32) {
            s0 += partial[(size_t)k * N + c];
            s1 += partial[(size_t)(k + 8) * N + c];
            s2 += partial[(size_t)(k + 16) * N + c];
            s3 += partial[(size_t)(k + 24) * N + c];
        }
        for (; k < ntiles; k += 8) s0 += partial[(size_t)k * N + c];
    }
    gsum[grp][cl] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (grp != 0 || c >= N) return;
    double delta = 0.0;
    for (int g8 = 0; g8 < 8; ++g8) delta += gsum[g8][cl];
    double sx = 0.0;                                                                // Σ_t x_t
    for (int b = 0; b < B; ++b) sx += V[(size_t)c * B + b] * convsum[(size_t)p + (size_t)N * b];
    const size_t pc = (size_t)p + (size_t)c * N;
    const double rho = rho_mat ? rho_mat[pc] : rho_scalar;
    const double d = (delta - sx) + nhp_log(rho) - nhp_log(1.0 - rho);              // ll1 - ll0
    // rand(Bernoulli(exp(ll1 - logsumexp(ll0, ll1)))): u <= 1/(1 + e^{-d})  <=>  logit(u) <= d
    const double uu = u ? u[pc] : nhp_philox_uniform(seed ^ 0xAD7AC3117D15C0DEull, step, pc);
    const double anew = nhp_log(uu / (1.0 - uu)) <= d ? 1.0 : 0.0;
    const double aold = a_p[c];
    A[pc] = anew;
    if (p + 1 < N) dadj_prep_row(p + 1, c, N, B, dt, W, theta, A, V, a_p);
    dprev[c] = anew - aold;
}

extern "C" nhp_status nhp_disc_resample_adjacency(nhp_ctx *ctx, const nhp_disc_dataset *ds, const double *lambda0,
                                                  const double *W, const double *theta, double *A, double dt,
                                                  const double *rho_matrix, double rho, const double *u,
                                                  uint64_t seed, uint64_t step, double *n_links)
{
    if (!ctx || !ds || !A) return NHP_EINVAL;
    if (!rho_matrix && !(rho >= 0.0 && rho <= 1.0)) { nhp_set_error(ctx, "link probability must lie in [0, 1]"); return NHP_EDOMAIN; }
    NHP_HIP(ctx, hipSetDevice(ctx->device));
    const size_t N = (size_t)ds->N, NN = N * N, B = (size_t)ds->B, TN = (size_t)ds->T * N;
    const size_t nocc = (size_t)(ds->nocc > 0 ? ds->nocc : 1);
    const int ntiles = (int)((ds->T + NHP_DA_TT - 1) / NHP_DA_TT);
    const size_t lds = 8 * (B * NHP_DA_TT + N);
    if (lds > 160 * 1024) { nhp_set_error(ctx, "resample_adjacency: N = %d, B = %d exceed the LDS budget", ds->N, ds->B); return NHP_ENOTIMPL; }
    // scratch after stage_bump's own block: λ (T·N, for the initial gather) | λ_occ | xprev | partial | V | a_p | dprev | u | ρ
    const size_t extra = TN + 2 * nocc + (size_t)ntiles * N + N * B + 2 * N + 2 * NN;
    double *E, *base, *x;
    NHP_TRY(nhp_disc_stage_bump(ctx, ds, lambda0, W, theta, A, dt, &E, &base, extra, &x, 0));
    double *dlam = x; x += TN;
    double *lam_occ = x; x += nocc;
    double *xprev = x; x += nocc;
    double *partial = x; x += (size_t)ntiles * N;
    double *V = x; x += N * B;
    double *a_p = x; x += N;
    double *dprev = x; x += N;
    double *d_u = x; x += NN;
    double *d_rho = x;
    // stage_bump left W, θ, A on the device just before the extra block: recover the pointers
    double *dW = base + 2 * N, *dth = dW + NN, *dA = dth + NN * B;
    hipStream_t st = ctx->stream;
    if (u) NHP_HIP(ctx, hipMemcpyAsync(d_u, u, 8 * NN, hipMemcpyHostToDevice, st));
    if (rho_matrix) NHP_HIP(ctx, hipMemcpyAsync(d_rho, rho_matrix, 8 * NN, hipMemcpyHostToDevice, st));
    // λ under the current A (GEMM-1), gathered at the occupied bins
    NHP_TRY(nhp_disc_launch_intensity(ctx, ds, E, base, lambda0 == nullptr, dlam));
    hipLaunchKernelGGL(k_dadj_gather, dim3((unsigned)((nocc + 255) / 256)), dim3(256), 0, st, dlam, ds->d_occ_t, ds->d_occ_c,
                       ds->nocc, ds->T, lam_occ, xprev);
    const unsigned cb = (unsigned)((N + 255) / 256);
    hipLaunchKernelGGL(k_dadj_prep, dim3(cb), dim3(256), 0, st, 0, ds->N, ds->B, dt, dW, dth, dA, V, a_p, dprev);
    NHP_HIP(ctx, hipGetLastError());
    if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void *)k_dadj_accum, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    for (int p = 0; p < ds->N; ++p) {
        hipLaunchKernelGGL(k_dadj_accum, dim3((unsigned)ntiles), dim3(256), lds, st, p, ds->N, ds->B, ds->T, ds->d_conv, ds->d_occ_t,
                           ds->d_occ_c, ds->d_occ_s, ds->d_occ_off, V, a_p, dprev, lam_occ, xprev, partial);
        hipLaunchKernelGGL(k_dadj_decide, dim3((unsigned)((N + 31) / 32)), dim3(256), 0, st, p, ds->N, ds->B, ntiles, dt, dW, dth, dA, partial, ds->d_convsum,
                           rho_matrix ? d_rho : nullptr, rho, u ? d_u : nullptr, seed, step, V, a_p, dprev);
    }
    NHP_HIP(ctx, hipGetLastError());
    NHP_HIP(ctx, hipMemcpyAsync(A, dA, 8 * NN, hipMemcpyDeviceToHost, st));
    NHP_HIP(ctx, hipStreamSynchronize(st));
    if (n_links) { double s = 0.0; for (size_t i = 0; i < NN; ++i) s += A[i]; *n_links = s; }
    return NHP_OK;
}
