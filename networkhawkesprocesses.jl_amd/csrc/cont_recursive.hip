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

__global__ __launch_bounds__(NHP_BLOCK) void k_recursive(nhp_cont_args a, double *__restrict__ partials)
{
    extern __shared__ __align__(16) unsigned char smem[];
    double *red = reinterpret_cast<double *>(smem);              // [4]
    double *ring = red + 4;                                      // [NHP_RING * NHP_WAVES]
    double *th = ring + NHP_RING * NHP_WAVES;                    // [N] θ[p,c]
    double *wth = th + a.N;                                      // [N] (a*w)*θ
    double *S = wth + a.N;                                       // [N] state, as of the last child
    double *acc_new = S + a.N;                                   // [N] segment accumulator, as of t_k

    const int c = blockIdx.x, N = a.N, tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;

    double integ = 0.0;
    for (int p = tid; p < N; p += NHP_BLOCK) {
        const size_t k = (size_t)p + (size_t)c * N;
        const double w = a.W[k];
        const double weff = a.A ? a.A[k] * w : w;
        const double t = a.p1[k];
        th[p] = t;
        wth[p] = weff * t;
        S[p] = 0.0;
        acc_new[p] = 0.0;
        integ += a.cnt[p] * w;                                   // unmasked: src/continuous.jl:247,413
    }
    __syncthreads();

    const int kb = a.boff[c], ke = a.boff[c + 1];
    int prev_idx = 0;
    double prev_t = 0.0, logsum = 0.0;
    for (int k = kb; k < ke; ++k) {
        const nhp_child ch = a.child[k];
        // fold the segment's events into the accumulator, referenced to t_k
        for (int j = prev_idx + tid; j < ch.idx; j += NHP_BLOCK) {
            const double tj = a.times[j];
            if (tj > 0.0) {
                const int p = a.nodes[j];
                const double e = nhp_exp_neg(-(th[p] * (ch.t - tj)));
                atomicAdd(&acc_new[p], e);
            }
        }
        __syncthreads();
        // decay the state to t_k, merge, dot with the weights
        const double gap = ch.t - prev_t;
        double part = 0.0;
        for (int p = tid; p < N; p += NHP_BLOCK) {
            double s = S[p];
            if (k != kb) s *= nhp_exp_neg(-(th[p] * gap));
            s += acc_new[p];
            acc_new[p] = 0.0;
            S[p] = s;
            part += wth[p] * s;
        }
        part = nhp_wave_sum(part);
        const int slot = (k - kb) & (NHP_RING - 1);
        if (lane == 0) ring[slot * NHP_WAVES + wave] = part;
        __syncthreads();
        if (slot == NHP_RING - 1 || k == ke - 1) {
            if (tid <= slot) {
                const double tk = a.child[k - slot + tid].t;
                double lam = rec_baseline(a, c, tk);
                for (int w = 0; w < NHP_WAVES; ++w) lam += ring[tid * NHP_WAVES + w];
                logsum += nhp_log(lam);
            }
        }
        prev_idx = ch.idx;
        prev_t = ch.t;
    }
    const double blk = nhp_block_sum(logsum, red);
    const double blk_int = nhp_block_sum(integ, red);
    if (tid == 0) {
        partials[2 * (size_t)c] = blk;
        partials[2 * (size_t)c + 1] = blk_int;
    }
}

nhp_status nhp_launch_recursive(nhp_ctx *ctx, const nhp_cont_dataset *ds, const nhp_cont_model *m, double *d_out)
{
    if (m->impulse_kind != NHP_IMPULSE_EXPONENTIAL) return NHP_EINVAL;
    NHP_HIP(ctx, hipSetDevice(ctx->device));
    const size_t lds = sizeof(double) * (4 + NHP_RING * NHP_WAVES + 4 * (size_t)ds->N);
    if (lds > 64 * 1024) { nhp_set_error(ctx, "recursive ll: n_nodes = %d exceeds the 64 KiB LDS state budget", ds->N); return NHP_ENOTIMPL; }
    NHP_TRY(nhp_ctx_reserve_partials(ctx, 2 * (size_t)ds->N));
    nhp_cont_args a = nhp_make_args(ds, m);
    hipLaunchKernelGGL(k_recursive, dim3((unsigned)ds->N), dim3(NHP_BLOCK), lds, ctx->stream, a, ctx->d_partials);
    NHP_HIP(ctx, hipGetLastError());
    return nhp_launch_finalize(ctx, a, ds->N, d_out);
}
