// intensity(process, data, times): λ_c(t) for every child node c at arbitrary query times
// (reference src/continuous.jl:76-96): parents are the events with t - Δtmax < t_j < t
// (strict on both sides), every node is a child, result = λ0(t) .+ Σ.
//
// One workgroup per query time.  A parent event contributes to all N children through ROW
// n_j of the parameter tables, which is strided in the reference's column-major layout, so
// the tables are transposed once per call into row-major scratch (N² elements, coalesced both
// ways through an LDS tile); lanes then run across children and read contiguous rows, while
// the parent's (t_j, n_j) is wave-uniform.  Parents are folded in ascending time order, the
// order of the reference's loop.
#include "nhp_internal.h"
#include "nhp_math.h"

// out1/out2/out3 [p*N + c] <- exp: {θ, a*w};  logit-normal: {μ, sqrt(τ), a*w}
__global__ __launch_bounds__(256) void k_transpose_params(nhp_cont_args a, double *__restrict__ o1,
                                                          double *__restrict__ o2, double *__restrict__ o3)
{
    __shared__ double tile[3][32][33];
    const int N = a.N, tx = threadIdx.x & 31, ty = threadIdx.x >> 5;      // 32 x 8
    const int p0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    for (int r = ty; r < 32; r += 8) {
        const int p = p0 + tx, c = c0 + r;
        if (p < N && c < N) {
            const size_t k = (size_t)p + (size_t)c * N;
            double w = a.W[k];
            if (a.A) w = a.A[k] * w;
            if (a.impulse_kind == NHP_IMPULSE_EXPONENTIAL) {
                tile[0][r][tx] = a.p1[k];
                tile[1][r][tx] = w;
            } else {
                tile[0][r][tx] = a.p1[k];
                tile[1][r][tx] = __builtin_sqrt(a.p2[k]);
                tile[2][r][tx] = w;
            }
        }
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int p = p0 + r, c = c0 + tx;
        if (p < N && c < N) {
            const size_t k = (size_t)p * N + c;
            o1[k] = tile[0][tx][r];
            o2[k] = tile[1][tx][r];
            if (a.impulse_kind != NHP_IMPULSE_EXPONENTIAL) o3[k] = tile[2][tx][r];
        }
    }
}

__device__ __forceinline__ double int_baseline(const nhp_cont_args &a, int c, double t)
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

__global__ __launch_bounds__(NHP_BLOCK) void k_intensity(nhp_cont_args a, const double *__restrict__ q, int64_t Q,
                                                         const double *__restrict__ r1, const double *__restrict__ r2,
                                                         const double *__restrict__ r3, double *__restrict__ out)
{
    const int64_t iq = blockIdx.x;
    const double t = q[iq];
    const double thr = t - a.dt_max;
    // first event with t_j > thr, first event with t_j >= t  (events sorted ascending)
    int64_t lo = 0, hi = a.M;
    while (lo < hi) { int64_t mid = (lo + hi) >> 1; if (a.times[mid] > thr) hi = mid; else lo = mid + 1; }
    const int64_t jb = lo;
    lo = jb; hi = a.M;
    while (lo < hi) { int64_t mid = (lo + hi) >> 1; if (a.times[mid] >= t) hi = mid; else lo = mid + 1; }
    const int64_t je = lo;
    const int N = a.N;
    for (int c = threadIdx.x; c < N; c += NHP_BLOCK) {
        double lam = 0.0;
        for (int64_t j = jb; j < je; ++j) {
            const double dt = t - a.times[j];
            const size_t k = (size_t)a.nodes[j] * N + c;
            if (a.impulse_kind == NHP_IMPULSE_EXPONENTIAL)
                lam += r2[k] * nhp_pdf_exponential(r1[k], dt);
            else
                lam += r3[k] * nhp_pdf_logitnormal(r1[k], r2[k], a.inv_dtmax, dt);
        }
        out[iq + (size_t)c * Q] = int_baseline(a, c, t) + lam;
    }
}

extern "C" nhp_status nhp_cont_intensity(nhp_ctx *ctx, const nhp_cont_dataset *ds, const nhp_cont_model *m,
                                         const double *times, int64_t Q, double *out)
{
    NHP_TRY(nhp_check_pair(ctx, ds, m));
    if (Q < 0 || (Q > 0 && (!times || !out))) return NHP_EINVAL;
    if (Q == 0) return NHP_OK;
    for (int64_t i = 0; i < Q; ++i) {
        if (m->baseline_kind == NHP_BASELINE_HOMOGENEOUS) {
            if (times[i] < 0.0) { nhp_set_error(ctx, "time must be non-negative"); return NHP_EDOMAIN; }     // src/baselines.jl:111
        } else if (times[i] < 0.0 || times[i] > m->grid_end) {
            nhp_set_error(ctx, "Value is outside interpolation support (0, %g)", m->grid_end);               // interpolation.jl:29
            return NHP_EDOMAIN;
        }
    }
    NHP_HIP(ctx, hipSetDevice(ctx->device));
    const size_t N = (size_t)ds->N, NN = N * N;
    NHP_TRY(nhp_ctx_reserve_scratch(ctx, 8 * (3 * NN + (size_t)Q + (size_t)Q * N)));
    double *r1 = (double *)ctx->d_scratch, *r2 = r1 + NN, *r3 = r2 + NN, *dq = r3 + NN, *dout = dq + Q;
    nhp_cont_args a = nhp_make_args(ds, m);
    hipStream_t st = ctx->stream;
    NHP_HIP(ctx, hipMemcpyAsync(dq, times, 8 * (size_t)Q, hipMemcpyHostToDevice, st));
    dim3 tg((unsigned)((N + 31) / 32), (unsigned)((N + 31) / 32));
    hipLaunchKernelGGL(k_transpose_params, tg, dim3(256), 0, st, a, r1, r2, r3);
    NHP_HIP(ctx, hipGetLastError());
    hipLaunchKernelGGL(k_intensity, dim3((unsigned)Q), dim3(NHP_BLOCK), 0, st, a, dq, Q, r1, r2, r3, dout);
    NHP_HIP(ctx, hipGetLastError());
    NHP_HIP(ctx, hipMemcpyAsync(out, dout, 8 * (size_t)Q * N, hipMemcpyDeviceToHost, st));
    NHP_HIP(ctx, hipStreamSynchronize(st));
    return NHP_OK;
}
