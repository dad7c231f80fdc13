// Log Gaussian Cox baseline inside a Gibbs sweep (reference
// loglikelihood(process::LogGaussianCoxProcess, data, node, y) src/baselines.jl:247-254 on the
// events split_extract :227-238 keeps).  The elliptical-slice sampler (:287-326) proposes, per
// node, a new latent curve y and needs
//     ll_c = -trapezoid(λ_c) + Σ_{i: c_i = c, parent node of i = 0} log λ_c(t_i),   λ_c = exp(m + y_c)
// many times per sweep.  The reference re-splits the data and walks the grid linearly per event;
// here the attribution left on the device by the parent sampler (one int32 per child, bucket
// order) is reused, workgroup c owns node c with the candidate curve in LDS, and one call scores
// the candidates of all N nodes -- the slice loops of the nodes advance in lock step on the host.
#include "nhp_internal.h"
#include "nhp_math.h"

// pn[k] <- parent node (0-based, -1 = baseline) of bucket slot k from a caller's parentnodes vector
__global__ __launch_bounds__(256) void k_lgcp_attr(const nhp_child *__restrict__ child, const int64_t *__restrict__ parentnodes,
                                                   int64_t M, int32_t *__restrict__ pn)
{
    const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (k < M) pn[k] = (int32_t)parentnodes[child[k].idx] - 1;
}

__global__ __launch_bounds__(NHP_BLOCK) void k_lgcp_ll(const nhp_child *__restrict__ child, const int32_t *__restrict__ boff,
                                                       const int32_t *__restrict__ pn, const double *__restrict__ gx,
                                                       const double *__restrict__ lam, int G, double *__restrict__ ll)
{
#pragma clang fp contract(off)
    extern __shared__ __align__(16) unsigned char smem[];
    double *x = reinterpret_cast<double *>(smem), *y = x + G, *red = y + G;
    const int c = blockIdx.x, tid = threadIdx.x;
    for (int g = tid; g < G; g += NHP_BLOCK) { x[g] = gx[g]; y[g] = lam[(size_t)c * G + g]; }
    __syncthreads();
    double acc = 0.0;
    for (int k = boff[c] + tid; k < boff[c + 1]; k += NHP_BLOCK) {
        if (pn[k] >= 0) continue;                           // split_extract: parentnode == 0 only
        const double t = child[k].t;
        double f = y[G - 1];                                // interpolate(): src/utils/interpolation.jl:26-35
        if (t < x[G - 1]) {
            int lo = 0, hi = G - 1;
            while (hi - lo > 1) {
                const int mid = (lo + hi) >> 1;
                if (t >= x[mid]) lo = mid; else hi = mid;
            }
            f = (y[lo + 1] * (t - x[lo]) + y[lo] * (x[lo + 1] - t)) / (x[lo + 1] - x[lo]);
        }
        acc += nhp_log(f);
    }
    const double s = nhp_block_sum(acc, red);
    if (tid == 0) {
        double I = 0.0;                                     // integrate(): :40-48, same order
        for (int g = 0; g + 1 < G; ++g) I += 0.5 * (y[g] + y[g + 1]) * (x[g + 1] - x[g]);
        ll[c] = (0.0 - I) + s;
    }
}

extern "C" nhp_status nhp_cont_lgcp_loglik(nhp_ctx *ctx, const nhp_cont_dataset *ds, const int64_t *parentnodes,
                                           const double *grid_x, int32_t grid_n, const double *lam, double *ll)
{
    if (!ctx || !ds || !grid_x || !lam || !ll) { if (ctx) nhp_set_error(ctx, "lgcp_loglik: null argument"); return NHP_EINVAL; }
    if (ds->ctx != ctx) { nhp_set_error(ctx, "lgcp_loglik: dataset belongs to another context"); return NHP_EINVAL; }
    NHP_WHOLE_DATASET(ctx, ds, "lgcp_loglik");          // reads the parent assignment of every node
    if (grid_n < 2 || grid_n > 4096) { nhp_set_error(ctx, "lgcp_loglik: grid_n = %d outside [2, 4096]", grid_n); return NHP_ESHAPE; }
    for (int g = 0; g + 1 < grid_n; ++g)
        if (!(grid_x[g + 1] > grid_x[g])) { nhp_set_error(ctx, "lgcp_loglik: grid points must be strictly increasing"); return NHP_EDOMAIN; }
    // interpolate() throws DomainError outside [x[1], x[end]] (src/utils/interpolation.jl:27)
    if (ds->M > 0 && (ds->t_last > grid_x[grid_n - 1] || grid_x[0] > 0.0)) {
        nhp_set_error(ctx, "lgcp_loglik: events fall outside the grid support [%g, %g]", grid_x[0], grid_x[grid_n - 1]);
        return NHP_EDOMAIN;
    }
    if (!parentnodes && !ds->pn_valid) {
        nhp_set_error(ctx, "lgcp_loglik: no parent assignment on the device (run resample_parents first or pass parentnodes)");
        return NHP_EINVAL;
    }
    NHP_HIP(ctx, hipSetDevice(ctx->device));
    const size_t M = (size_t)ds->M, N = (size_t)ds->N, G = (size_t)grid_n;
    size_t off = 0;
    auto carve = [&](size_t bytes) { size_t r = off; off += (bytes + 255) & ~(size_t)255; return r; };
    const size_t o_x = carve(8 * G), o_lam = carve(8 * N * G), o_ll = carve(8 * N), o_pn = carve(parentnodes ? 8 * (M ? M : 1) : 8);
    NHP_TRY(nhp_ctx_reserve_scratch(ctx, off));
    char *base = (char *)ctx->d_scratch;
    hipStream_t st = ctx->stream;
    if (parentnodes && M) {
        for (size_t i = 0; i < M; ++i)
            if (parentnodes[i] < 0 || parentnodes[i] > (int64_t)N) { nhp_set_error(ctx, "lgcp_loglik: parentnodes[%zu] outside 0..N", i); return NHP_EDOMAIN; }
        NHP_HIP(ctx, hipMemcpyAsync(base + o_pn, parentnodes, 8 * M, hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(k_lgcp_attr, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, st, ds->d_child,
                           (const int64_t *)(base + o_pn), (int64_t)M, ds->d_pn);
        NHP_HIP(ctx, hipGetLastError());
        ds->pn_valid = true;
    }
    NHP_HIP(ctx, hipMemcpyAsync(base + o_x, grid_x, 8 * G, hipMemcpyHostToDevice, st));
    NHP_HIP(ctx, hipMemcpyAsync(base + o_lam, lam, 8 * N * G, hipMemcpyHostToDevice, st));
    const size_t lds = 8 * (2 * G + NHP_WAVES);
    if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void *)k_lgcp_ll, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(k_lgcp_ll, dim3((unsigned)N), dim3(NHP_BLOCK), lds, st, ds->d_child, ds->d_boff, ds->d_pn,
                       (const double *)(base + o_x), (const double *)(base + o_lam), grid_n, (double *)(base + o_ll));
    NHP_HIP(ctx, hipGetLastError());
    NHP_HIP(ctx, hipMemcpyAsync(ll, base + o_ll, 8 * N, hipMemcpyDeviceToHost, st));
    NHP_HIP(ctx, hipStreamSynchronize(st));
    return NHP_OK;
}
