// Diagnostics: evaluate the device det-math primitives on arrays, so tests can compare them
// bit for bit with the CPU evaluation of the same operation sequence.
#include "nhp_internal.h"
#include "nhp_math.h"

__global__ void k_probe(int op, const double *__restrict__ x, const double *__restrict__ y, int64_t n,
                        double *__restrict__ out)
{
#pragma clang fp contract(off)
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double a = x[i], b = y ? y[i] : 0.0, r;
    switch (op) {
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
