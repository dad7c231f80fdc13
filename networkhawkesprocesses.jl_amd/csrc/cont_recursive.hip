// Recursive exponential log-likelihood (reference: recursive_loglikelihood,
// src/continuous.jl:241-276, network twin :407-442).
//
// What the reference's O(M*N) recursion computes, for every event i on child node c:
//   λ_i = λ0_c(t_i) + Σ_p (A[p,c] W[p,c] θ[p,c]) * S_pc(t_i),
//   S_pc(t) = Σ_{j<i, n_j=p, t_j>0} exp(-θ[p,c] (t - t_j))
// i.e. ALL earlier events are parents -- Δtmax is ignored -- except events at
// exactly t = 0.0, which the `parenttimes > 0.0` seen-flag drops (D9); the integral term of
// the network twin is NOT masked by A (D7).  The reference walks events serially and keeps an
// N x N state; one event touches one row (update) and one column (read) of it.
//
// CDNA4 mapping: the state is partitioned by COLUMN.  Workgroup c owns S_{.,c} (N doubles in
// LDS) plus column c of θ and of A∘W∘θ, and visits only the children of node c (node
// buckets).  Between two consecutive children k-1 and k it folds the events of the segment
// [idx_{k-1}, idx_k) -- contiguous in the time-ordered arrays, so the loads are coalesced --
// into an accumulator with one exp and one LDS atomic each, then decays the state to t_k and
// takes the dot product with the weights (one exp per parent node), lanes across parent
// nodes.  Exp count per evaluation: M*N (segments) + M*N (decays) = the reference's 2*M*N,
// but N workgroups wide and 256 lanes deep instead of serial.
#include "nhp_internal.h"
#include "nhp_math.h"

#define NHP_RING 64   // children whose wave partials are buffered before the log pass

__device__ __forceinline__ double rec_baseline(const nhp_cont_args &a, int c, double t)
{
#pragma clang fp contract(off)
    if (a.baseline_kind == NHP_BASELINE_HOMOGENEOUS) return a.lambda0[c];
    const double *x = a.grid;
    const double *y = a.lambda0 + (size_t)c * a.grid_n;
    int lo = 0, hi = a.grid_n - 1;
    if (!(t < x[hi])) return y[hi];
    while (hi - lo > 1) {
        int mid = (lo + hi) >> 1;
        if (t >= x[mid]) lo = mid; else hi = mid;
    }
    return (y[lo + 1] * (t - x[lo]) + y[lo] * (x[lo + 1] - t)) / (x[lo + 1] - x[lo]);
}

// Layout of one workgroup (column c): thread `tid` owns parent nodes p = tid + 256 q; their state
// S_pc, θ_pc and (a·w·θ)_pc live in registers.  LDS holds θ[·,c] (gathered by node in the segment
// pass) and TWO segment accumulators, so that folding segment k+1 and consuming segment k share
// one barrier interval: one barrier per child instead of two, and two independent instruction
// streams for the scheduler to interleave.
// BLOCK threads, PQ parent nodes per thread (N <= BLOCK * PQ); the launcher picks 256 x 8 up to
// N = 2048 and 512 x 8 up to 4096.
template <int BLOCK, int PQ>
__global__ __launch_bounds__(BLOCK) void k_recursive(nhp_cont_args a, double *__restrict__ partials)
{
    constexpr int WAVES = BLOCK / 64, REC_PQ = PQ;
    extern __shared__ __align__(16) unsigned char smem[];
    double *red = reinterpret_cast<double *>(smem);              // [8]
    double *ring = red + 8;                                      // [2][NHP_RING * WAVES]
    double *th = ring + 2 * NHP_RING * WAVES;                // [N] θ[p,c]
    double *acc0 = th + a.N;                                     // [N] segment accumulators (double-buffered)
    double *acc1 = acc0 + a.N;

    const int c = blockIdx.x, N = a.N, tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;

    double S[REC_PQ], thr[REC_PQ], wthr[REC_PQ];
    double integ = 0.0;
#pragma unroll
    for (int q = 0; q < REC_PQ; ++q) {
        const int p = tid + q * BLOCK;
        S[q] = 0.0; thr[q] = 0.0; wthr[q] = 0.0;
        if (p < N) {
            const size_t k = (size_t)p + (size_t)c * N;
            const double w = a.W[k];
            const double weff = a.A ? a.A[k] * w : w;
            const double t = a.p1[k];
            thr[q] = t; wthr[q] = weff * t;
            th[p] = t; acc0[p] = 0.0; acc1[p] = 0.0;
            integ += a.cnt[p] * w;                               // unmasked: src/continuous.jl:247,413
        }
    }
    __syncthreads();

    const int kb = a.boff[c], ke = a.boff[c + 1];
    double logsum = 0.0;

    // fold the events of [jb, je) into acc, referenced to time tk  (t_j > 0: the D9 seen-flag)
    auto fold = [&](double *acc, int jb, int je, double tk) {
        for (int j = jb + tid; j < je; j += BLOCK) {
            const nhp_event e = a.ev[j];
            if (e.t > 0.0) atomicAdd(&acc[e.node], nhp_exp_neg(-(th[e.node] * (tk - e.t))));
        }
    };

    // child records (time, index) are kept in scalars and fetched two ahead, so their latency never
    // sits on the loop (a struct copy here makes hipcc spill the records to scratch)
    double ch_t = 0.0, nx_t = 0.0;
    int ch_idx = 0, nx_idx = 0;
    if (kb < ke) { ch_t = a.child[kb].t; ch_idx = a.child[kb].idx; nx_t = ch_t; nx_idx = ch_idx; }
    if (kb + 1 < ke) { nx_t = a.child[kb + 1].t; nx_idx = a.child[kb + 1].idx; }
    if (kb < ke) fold(acc0, 0, ch_idx, ch_t);
    __syncthreads();
    double prev_t = ch_t;
    for (int k = kb; k < ke; ++k) {
        const int par = (k - kb) & 1;
        double *accA = par ? acc0 : acc1;                        // filled now, consumed next iteration
        double *accB = par ? acc1 : acc0;                        // filled last iteration, consumed now
        const int kn = k + 2 < ke ? k + 2 : ke - 1;
        const double nn_t = a.child[kn].t;
        const int nn_idx = a.child[kn].idx;
        if (k + 1 < ke) fold(accA, ch_idx, nx_idx, nx_t);
        // decay the state to t_k, merge segment k, dot with the weights
        const double gap = ch_t - prev_t;
        double part = 0.0;
#pragma unroll
        for (int q = 0; q < REC_PQ; ++q) {
            const int p = tid + q * BLOCK;
            if (p < N) {
                double s = S[q];
                if (k != kb) s *= nhp_exp_neg(-(thr[q] * gap));
                s += accB[p];
                accB[p] = 0.0;
                S[q] = s;
                part += wthr[q] * s;
            }
        }
        part = nhp_wave_sum(part);
        const int slot = (k - kb) & (NHP_RING - 1);
        const int half = ((k - kb) / NHP_RING) & 1;
        if (lane == 0) ring[(half * NHP_RING + slot) * WAVES + wave] = part;
        __syncthreads();
        if (slot == NHP_RING - 1 || k == ke - 1) {
            if (tid <= slot) {
                const double tk = a.child[k - slot + tid].t;
                double lam = rec_baseline(a, c, tk);
                for (int w = 0; w < WAVES; ++w) lam += ring[(half * NHP_RING + tid) * WAVES + w];
                logsum += nhp_log(lam);
            }
        }
        prev_t = ch_t;
        ch_t = nx_t; ch_idx = nx_idx;
        nx_t = nn_t; nx_idx = nn_idx;
    }
    const double blk = nhp_block_sum_n<WAVES>(logsum, red);
    const double blk_int = nhp_block_sum_n<WAVES>(integ, red);
    if (tid == 0) {
        partials[2 * (size_t)c] = blk;
        partials[2 * (size_t)c + 1] = blk_int;
    }
}

nhp_status nhp_launch_recursive(nhp_ctx *ctx, const nhp_cont_dataset *ds, const nhp_cont_model *m, double *d_out)
{
    if (m->impulse_kind != NHP_IMPULSE_EXPONENTIAL) return NHP_EINVAL;
    NHP_HIP(ctx, hipSetDevice(ctx->device));
    if (ds->N > 4096) { nhp_set_error(ctx, "recursive ll: n_nodes = %d > 4096 not supported", ds->N); return NHP_ENOTIMPL; }
    NHP_TRY(nhp_ctx_reserve_partials(ctx, 2 * (size_t)ds->N));
    nhp_cont_args a = nhp_make_args(ds, m);
#define NHP_REC_LAUNCH(B, Q)                                                                                   \
    do {                                                                                                       \
        const size_t lds = sizeof(double) * (8 + 2 * NHP_RING * ((B) / 64) + 3 * (size_t)ds->N);               \
        if (lds > 64 * 1024)                                                                                   \
            (void)hipFuncSetAttribute((const void *)k_recursive<B, Q>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        hipLaunchKernelGGL((k_recursive<B, Q>), dim3((unsigned)ds->N), dim3(B), lds, ctx->stream, a, ctx->d_partials); \
    } while (0)
    // measured at N = 1024, M = 1e6: 256 x 8 3.16 ms, 256 x 4 3.18, 512 x 4 3.35, 128 x 8 4.19, 64 x 16 7.35
    if (ds->N <= 2048) NHP_REC_LAUNCH(256, 8);
    else NHP_REC_LAUNCH(512, 8);
#undef NHP_REC_LAUNCH
    NHP_HIP(ctx, hipGetLastError());
    return nhp_launch_finalize(ctx, a, ds->N, d_out);
}
