// Diagnostics: evaluate the device det-math primitives on arrays, so tests can compare them
// bit for bit with the CPU evaluation of the same operation sequence.
#include "nhp_internal.h"
#include "nhp_math.h"

__global__ void k_probe(int op, const double *__restrict__ x, const double *__restrict__ y, int64_t n,
                        double *__restrict__ out)
{
#pragma clang fp contract(off)
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    __shared__ double etab[64];
    nhp_exp_tab_init(etab);
    __syncthreads();
    if (i >= n) return;
    double a = x[i], b = y ? y[i] : 0.0, r;
    switch (op) {
    case 7: r = nhp_exp_neg_tab(a, etab); break;                       // the log-likelihood kernels' table-driven exp, x <= 0
    case 8: r = nhp_pdf_exponential_tab(a, b, etab); break;            // (θ, Δt) through it
    case 0: r = nhp_exp(a); break;
    case 1: r = nhp_log(a); break;
    case 2: r = __builtin_sqrt(a); break;
    case 3: r = a / b; break;
    case 4: r = nhp_exp_neg(a); break;
    case 5: { double scale = 1.0 / a; r = nhp_pdf_exponential(1.0 / scale, b); } break;     // (θ, Δt)
    default: r = nhp_pdf_logitnormal(0.25, __builtin_sqrt(a), 1.0 / 2.0, b); break;         // (τ, Δt), μ=.25, Δtmax=2
    }
    out[i] = r;
}

extern "C" nhp_status nhp_probe_math(nhp_ctx *ctx, int32_t op, const double *x, const double *y, int64_t n, double *out)
{
    if (!ctx || !x || !out || n <= 0) return NHP_EINVAL;
    NHP_HIP(ctx, hipSetDevice(ctx->device));
    NHP_TRY(nhp_ctx_reserve_scratch(ctx, 3 * 8 * (size_t)n));
    double *dx = (double *)ctx->d_scratch, *dy = dx + n, *dout = dy + n;
    NHP_HIP(ctx, hipMemcpyAsync(dx, x, 8 * n, hipMemcpyHostToDevice, ctx->stream));
    if (y) NHP_HIP(ctx, hipMemcpyAsync(dy, y, 8 * n, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(k_probe, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, op, dx, y ? dy : nullptr, n, dout);
    NHP_HIP(ctx, hipGetLastError());
    NHP_HIP(ctx, hipMemcpyAsync(out, dout, 8 * n, hipMemcpyDeviceToHost, ctx->stream));
    NHP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return NHP_OK;
}

// Throughput calibration ("c_pair measured by an exp-only microkernel"): every lane
// evaluates `iters` exponential pair terms on register operands, no memory traffic in the loop.
// mode 0: the pair term w*θ*exp(-θΔ);  mode 1: bare fp64 fma chain (peak fp64 VALU check).
__global__ __launch_bounds__(256) void k_probe_rate(int mode, int iters, double *__restrict__ sink)
{
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    double th = 1.0 + 1e-3 * (gid & 1023), dt = 1e-2 * (1 + (gid & 63)), acc0 = 0.0, acc1 = 0.0, acc2 = 0.0, acc3 = 0.0;
    if (mode == 0) {
        for (int i = 0; i < iters; i += 4) {
            acc0 += 0.5 * nhp_pdf_exponential(th, dt);
            acc1 += 0.5 * nhp_pdf_exponential(th, dt + 0.25);
            acc2 += 0.5 * nhp_pdf_exponential(th, dt + 0.5);
            acc3 += 0.5 * nhp_pdf_exponential(th, dt + 0.75);
            dt += 1e-6;
        }
    } else {
        double a = th, b = dt;
        for (int i = 0; i < iters; i += 4) {
            acc0 = __builtin_fma(acc0, a, b);
            acc1 = __builtin_fma(acc1, a, b);
            acc2 = __builtin_fma(acc2, a, b);
            acc3 = __builtin_fma(acc3, a, b);
        }
    }
    if (acc0 + acc1 + acc2 + acc3 == 12345.678) sink[gid] = acc0;
}

extern "C" nhp_status nhp_probe_rate(nhp_ctx *ctx, int32_t mode, int32_t iters, int32_t blocks, double *ops_per_s)
{
    if (!ctx || !ops_per_s || iters < 4 || blocks < 1) return NHP_EINVAL;
    NHP_HIP(ctx, hipSetDevice(ctx->device));
    NHP_TRY(nhp_ctx_reserve_scratch(ctx, 8 * (size_t)blocks * 256));
    double *sink = (double *)ctx->d_scratch;
    hipLaunchKernelGGL(k_probe_rate, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, mode, iters, sink);   // warm-up
    NHP_HIP(ctx, hipEventRecord(ctx->ev0, ctx->stream));
    hipLaunchKernelGGL(k_probe_rate, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, mode, iters, sink);
    NHP_HIP(ctx, hipEventRecord(ctx->ev1, ctx->stream));
    NHP_HIP(ctx, hipEventSynchronize(ctx->ev1));
    float ms = 0.f;
    NHP_HIP(ctx, hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
    *ops_per_s = (double)blocks * 256.0 * (double)iters / ((double)ms * 1e-3);
    return NHP_OK;
}

// Gather calibration for the short-window regime: `n_windows` windows of `recs` consecutive
// 16-byte records at pseudo-random positions of an `array_recs`-record array, 8 lanes per window and four
// windows per lane group in flight -- the windowed kernel's access pattern with the arithmetic removed
// (one add per record).  Reports microseconds per launch.
__global__ __launch_bounds__(256) void k_probe_gather(const double2 *__restrict__ arr, unsigned array_recs, int recs,
                                                      int n_windows, double *__restrict__ sink)
{
    const int gid = (blockIdx.x * 256 + threadIdx.x) >> 3, gl = threadIdx.x & 7, ngroups = (gridDim.x * 256) >> 3;
    double acc = 0.0;
    for (int w0 = gid * 4; w0 < n_windows; w0 += ngroups * 4) {
        double2 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            unsigned h = (unsigned)(w0 + u) * 2654435761u;               // Knuth hash: a scattered start per window
            h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
            const unsigned start = h % (array_recs - (unsigned)recs);
            v[u] = make_double2(0.0, 0.0);
            for (int r = gl; r < recs; r += 8) { const double2 q = arr[start + r]; v[u].x += q.x; v[u].y += q.y; }
        }
        acc += (v[0].x + v[1].x) + (v[2].x + v[3].x) + (v[0].y + v[1].y) + (v[2].y + v[3].y);
    }
    if (acc == 12345.678) sink[0] = acc;                                // keep the loads alive
}

extern "C" nhp_status nhp_probe_gather(nhp_ctx *ctx, int32_t n_windows, int32_t recs, int64_t array_recs, int32_t blocks,
                                       double *us_per_launch)
{
    if (!ctx || !us_per_launch || n_windows < 1 || recs < 1 || array_recs <= recs || blocks < 1) return NHP_EINVAL;
    NHP_HIP(ctx, hipSetDevice(ctx->device));
    NHP_TRY(nhp_ctx_reserve_scratch(ctx, 16 * (size_t)array_recs + 64));
    double2 *arr = (double2 *)ctx->d_scratch;
    double *sink = (double *)(arr + array_recs);
    NHP_HIP(ctx, hipMemsetAsync(arr, 0, 16 * (size_t)array_recs, ctx->stream));
    const int reps = 20;
    hipLaunchKernelGGL(k_probe_gather, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, arr, (unsigned)array_recs, recs, n_windows, sink);
    NHP_HIP(ctx, hipEventRecord(ctx->ev0, ctx->stream));
    for (int r = 0; r < reps; ++r)
        hipLaunchKernelGGL(k_probe_gather, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, arr, (unsigned)array_recs, recs, n_windows, sink);
    NHP_HIP(ctx, hipEventRecord(ctx->ev1, ctx->stream));
    NHP_HIP(ctx, hipEventSynchronize(ctx->ev1));
    float ms = 0.f;
    NHP_HIP(ctx, hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
    *us_per_launch = 1e3 * (double)ms / reps;
    return NHP_OK;
}


// Streaming-read calibration for the child-slice kernels: every workgroup sweeps its own contiguous share of a
// `bytes`-byte buffer, one wave-wide row at a time, `ahead` rows in flight per wave -- mode 0: 16 bytes per lane (the copy
// kernels' shape), mode 1: the slices' own shape, a 4-byte plane and a 2-byte plane (256 + 128 bytes a row).  Launched back
// to back, so a buffer below 256 MB is served by the Infinity Cache like the log-likelihood's own repeated evaluations.
template <int MODE>
__global__ __launch_bounds__(512) void k_probe_stream(const uint32_t *__restrict__ a32, const uint16_t *__restrict__ a16, size_t rows_per_wave,
                                                      double *__restrict__ sink)
{
    const size_t wave = (size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    uint32_t acc = 0;
    if (MODE == 0) {
        const uint4 *p = reinterpret_cast<const uint4 *>(a32) + wave * rows_per_wave * 64 + lane;
        for (size_t r = 0; r < rows_per_wave; r += 4) {
            uint4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = p[(r + u) * 64];
#pragma unroll
            for (int u = 0; u < 4; ++u) acc += v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
        }
    } else {
        const uint32_t *p = a32 + wave * rows_per_wave * 64 + lane;
        const uint16_t *q = a16 + wave * rows_per_wave * 64 + lane;
        for (size_t r = 0; r < rows_per_wave; r += 8) {
            uint32_t v[8], h[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { v[u] = p[(r + u) * 64]; h[u] = q[(r + u) * 64]; }
#pragma unroll
            for (int u = 0; u < 8; ++u) acc += v[u] ^ h[u];
        }
    }
    if (acc == 0x12345678u) sink[0] = 1.0;
}

extern "C" nhp_status nhp_probe_stream(nhp_ctx *ctx, int32_t mode, int64_t bytes, int32_t blocks, int32_t threads, double *us_per_launch,
                                        int64_t *bytes_read)
{
    if (!ctx || !us_per_launch || bytes < 1 || blocks < 1 || (threads != 256 && threads != 512) || (mode != 0 && mode != 1)) return NHP_EINVAL;
    NHP_HIP(ctx, hipSetDevice(ctx->device));
    const size_t waves = (size_t)blocks * (size_t)(threads / 64);
    const size_t row_bytes = mode == 0 ? 1024 : 384;
    size_t rows_per_wave = (size_t)bytes / row_bytes / waves;
    rows_per_wave -= rows_per_wave % (mode == 0 ? 4 : 8);
    if (rows_per_wave < 4) { nhp_set_error(ctx, "probe_stream: buffer too small for the grid"); return NHP_EINVAL; }
    const size_t n32 = waves * rows_per_wave * 64 * (mode == 0 ? 4 : 1), n16 = mode == 0 ? 0 : waves * rows_per_wave * 64;
    NHP_TRY(nhp_ctx_reserve_scratch(ctx, 4 * n32 + 2 * n16 + 64));
    uint32_t *a32 = (uint32_t *)ctx->d_scratch;
    uint16_t *a16 = (uint16_t *)(a32 + n32);
    double *sink = (double *)((char *)ctx->d_scratch + ((4 * n32 + 2 * n16 + 7) & ~(size_t)7));
    NHP_HIP(ctx, hipMemsetAsync(a32, 0, 4 * n32 + 2 * n16, ctx->stream));
    const int reps = 20;
    for (int r = 0; r <= reps; ++r) {
        if (r == 1) NHP_HIP(ctx, hipEventRecord(ctx->ev0, ctx->stream));
        if (mode == 0) hipLaunchKernelGGL(k_probe_stream<0>, dim3((unsigned)blocks), dim3((unsigned)threads), 0, ctx->stream, a32, a16, rows_per_wave, sink);
        else hipLaunchKernelGGL(k_probe_stream<1>, dim3((unsigned)blocks), dim3((unsigned)threads), 0, ctx->stream, a32, a16, rows_per_wave, sink);
    }
    NHP_HIP(ctx, hipEventRecord(ctx->ev1, ctx->stream));
    NHP_HIP(ctx, hipEventSynchronize(ctx->ev1));
    float ms = 0.f;
    NHP_HIP(ctx, hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
    *us_per_launch = 1e3 * (double)ms / reps;
    if (bytes_read) *bytes_read = (int64_t)(waves * rows_per_wave * row_bytes);        // (the request rounded down to whole rows per wave)
    return NHP_OK;
}
