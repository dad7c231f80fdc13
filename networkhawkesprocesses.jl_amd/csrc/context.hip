// Context, error plumbing and result slots of libnhp.so.
#include <stdarg.h>
#include <string.h>

#include "nhp_internal.h"
#include "nhp_math.h"

static thread_local std::string g_last_error;

void nhp_set_error(nhp_ctx *ctx, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_last_error = buf;
    if (ctx) ctx->err = buf;
}

extern "C" int32_t nhp_abi_version(void) { return 2; }

#include <stddef.h>
extern "C" int32_t nhp_abi_layout(int32_t *out, int32_t cap)
{
    const int32_t v[NHP_ABI_LAYOUT_LEN] = {
        (int32_t)sizeof(nhp_cont_model_desc),
        (int32_t)offsetof(nhp_cont_model_desc, n_nodes), (int32_t)offsetof(nhp_cont_model_desc, baseline_kind),
        (int32_t)offsetof(nhp_cont_model_desc, lambda0), (int32_t)offsetof(nhp_cont_model_desc, grid_x),
        (int32_t)offsetof(nhp_cont_model_desc, grid_n), (int32_t)offsetof(nhp_cont_model_desc, impulse_kind),
        (int32_t)offsetof(nhp_cont_model_desc, theta), (int32_t)offsetof(nhp_cont_model_desc, mu),
        (int32_t)offsetof(nhp_cont_model_desc, tau), (int32_t)offsetof(nhp_cont_model_desc, dt_max),
        (int32_t)offsetof(nhp_cont_model_desc, W), (int32_t)offsetof(nhp_cont_model_desc, A),
        (int32_t)sizeof(nhp_gibbs_priors),
        (int32_t)offsetof(nhp_gibbs_priors, alpha0), (int32_t)offsetof(nhp_gibbs_priors, beta0),
        (int32_t)offsetof(nhp_gibbs_priors, kappa), (int32_t)offsetof(nhp_gibbs_priors, nu),
        (int32_t)offsetof(nhp_gibbs_priors, a), (int32_t)offsetof(nhp_gibbs_priors, b),
        (int32_t)offsetof(nhp_gibbs_priors, mu_mu), (int32_t)offsetof(nhp_gibbs_priors, kappa_mu),
        (int32_t)sizeof(nhp_cont_stats),
        (int32_t)offsetof(nhp_cont_stats, cnt0), (int32_t)offsetof(nhp_cont_stats, Mn), (int32_t)offsetof(nhp_cont_stats, Mnm),
        (int32_t)offsetof(nhp_cont_stats, Xnm), (int32_t)offsetof(nhp_cont_stats, Vnm),
        NHP_MAX_SLOTS, NHP_COMM_ID_BYTES};
    for (int32_t i = 0; out && i < cap && i < NHP_ABI_LAYOUT_LEN; ++i) out[i] = v[i];
    return NHP_ABI_LAYOUT_LEN;
}

extern "C" const char *nhp_last_error(const nhp_ctx *ctx)
{
    return ctx ? ctx->err.c_str() : g_last_error.c_str();
}

extern "C" nhp_status nhp_ctx_create(int32_t device, nhp_ctx **out)
{
    if (!out) return NHP_EINVAL;
    *out = nullptr;
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0) {
        nhp_set_error(nullptr, "no HIP device available (%s)", e == hipSuccess ? "count = 0" : hipGetErrorString(e));
        return NHP_EHIP;
    }
    if (device < 0 || device >= count) {
        nhp_set_error(nullptr, "device %d out of range (0..%d)", device, count - 1);
        return NHP_EINVAL;
    }
    nhp_ctx *ctx = new nhp_ctx();
    ctx->device = device;
    NHP_HIP(ctx, hipSetDevice(device));
    hipDeviceProp_t prop;
    NHP_HIP(ctx, hipGetDeviceProperties(&prop, device));
    ctx->cu_count = prop.multiProcessorCount;
    NHP_HIP(ctx, hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
    NHP_HIP(ctx, hipEventCreate(&ctx->ev0));
    NHP_HIP(ctx, hipEventCreate(&ctx->ev1));
    NHP_HIP(ctx, hipMalloc(&ctx->d_results, sizeof(double) * NHP_MAX_SLOTS));
    NHP_HIP(ctx, hipHostMalloc(&ctx->h_results, sizeof(double) * NHP_MAX_SLOTS));
    NHP_HIP(ctx, hipMalloc(&ctx->d_counter, 128 * 80));
    NHP_HIP(ctx, hipMemsetAsync(ctx->d_counter, 0, 128 * 80, ctx->stream));
    NHP_HIP(ctx, hipMalloc((void **)&ctx->d_err, sizeof(int)));
    NHP_HIP(ctx, hipHostMalloc((void **)&ctx->h_err, sizeof(int)));
    *ctx->h_err = 0;
    NHP_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_err, hipEventDisableTiming));
    NHP_HIP(ctx, hipStreamCreateWithFlags(&ctx->stream2, hipStreamNonBlocking));
    NHP_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming));
    NHP_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_join, hipEventDisableTiming));
    NHP_HIP(ctx, hipMalloc(&ctx->d_counter2, 128 * 80));
    NHP_HIP(ctx, hipMemsetAsync(ctx->d_counter2, 0, 128 * 80, ctx->stream));
    NHP_HIP(ctx, hipMemsetAsync(ctx->d_results, 0, sizeof(double) * NHP_MAX_SLOTS, ctx->stream));
    NHP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    *out = ctx;
    return NHP_OK;
}

extern "C" void nhp_ctx_destroy(nhp_ctx *ctx)
{
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    if (ctx->stream2) (void)hipStreamSynchronize(ctx->stream2);
    if (ctx->d_partials2) (void)hipFree(ctx->d_partials2);
    if (ctx->d_counter2) (void)hipFree(ctx->d_counter2);
    if (ctx->ev_fork) (void)hipEventDestroy(ctx->ev_fork);
    if (ctx->ev_join) (void)hipEventDestroy(ctx->ev_join);
    if (ctx->stream2) (void)hipStreamDestroy(ctx->stream2);
    if (ctx->d_results) (void)hipFree(ctx->d_results);
    if (ctx->h_results) (void)hipHostFree(ctx->h_results);
    if (ctx->h_stage) (void)hipHostFree(ctx->h_stage);
    if (ctx->d_err) (void)hipFree(ctx->d_err);
    if (ctx->h_err) (void)hipHostFree(ctx->h_err);
    if (ctx->ev_err) (void)hipEventDestroy(ctx->ev_err);
    if (ctx->d_partials) (void)hipFree(ctx->d_partials);
    if (ctx->d_scratch) (void)hipFree(ctx->d_scratch);
    if (ctx->d_mle) (void)hipFree(ctx->d_mle);
    if (ctx->h_mle_scal) (void)hipHostFree(ctx->h_mle_scal);
    if (ctx->d_counter) (void)hipFree(ctx->d_counter);
    if (ctx->ev0) (void)hipEventDestroy(ctx->ev0);
    if (ctx->ev1) (void)hipEventDestroy(ctx->ev1);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

extern "C" nhp_status nhp_ctx_synchronize(nhp_ctx *ctx)
{
    if (!ctx) return NHP_EINVAL;
    NHP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return nhp_check_deferred(ctx);
}

extern "C" nhp_status nhp_ctx_timer_start(nhp_ctx *ctx)
{
    if (!ctx) return NHP_EINVAL;
    NHP_HIP(ctx, hipEventRecord(ctx->ev0, ctx->stream));
    return NHP_OK;
}

extern "C" nhp_status nhp_ctx_timer_stop(nhp_ctx *ctx, double *elapsed_ms)
{
    if (!ctx || !elapsed_ms) return NHP_EINVAL;
    NHP_HIP(ctx, hipEventRecord(ctx->ev1, ctx->stream));
    NHP_HIP(ctx, hipEventSynchronize(ctx->ev1));
    float ms = 0.f;
    NHP_HIP(ctx, hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
    *elapsed_ms = (double)ms;
    return NHP_OK;
}

extern "C" nhp_status nhp_ctx_fetch(nhp_ctx *ctx, int32_t first_slot, int32_t n, double *out)
{
    if (!ctx || !out || first_slot < 0 || n < 0 || first_slot + n > NHP_MAX_SLOTS) return NHP_EINVAL;
    NHP_HIP(ctx, hipMemcpyAsync(ctx->h_results + first_slot, ctx->d_results + first_slot,
                                sizeof(double) * (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
    NHP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    memcpy(out, ctx->h_results + first_slot, sizeof(double) * (size_t)n);
    return NHP_OK;
}

nhp_status nhp_ctx_reserve_partials(nhp_ctx *ctx, size_t n)
{
    if (n <= ctx->partials_cap) return NHP_OK;
    NHP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->d_partials) (void)hipFree(ctx->d_partials);
    ctx->d_partials = nullptr;
    ctx->partials_cap = 0;
    NHP_HIP(ctx, hipMalloc(&ctx->d_partials, sizeof(double) * n));
    ctx->partials_cap = n;
    return NHP_OK;
}

nhp_status nhp_check_deferred(nhp_ctx *ctx)
{
    if (!ctx->err_pending) return NHP_OK;
    ctx->err_pending = false;
    NHP_HIP(ctx, hipEventSynchronize(ctx->ev_err));
    if (*ctx->h_err) {
        *ctx->h_err = 0;
        nhp_set_error(ctx, "gibbs_step (an earlier sweep): weights of some event do not sum to a positive finite value");
        return NHP_EDOMAIN;
    }
    return NHP_OK;
}

nhp_status nhp_download(nhp_ctx *ctx, void *dst, const void *d_src, size_t bytes)
{
    NHP_TRY(nhp_check_deferred(ctx));
    if (bytes == 0) { NHP_HIP(ctx, hipStreamSynchronize(ctx->stream)); return NHP_OK; }
    if (bytes > ctx->stage_cap) {
        NHP_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (ctx->h_stage) (void)hipHostFree(ctx->h_stage);
        ctx->h_stage = nullptr; ctx->stage_cap = 0;
        const size_t cap = (bytes + ((size_t)1 << 20) - 1) & ~(((size_t)1 << 20) - 1);
        if (hipHostMalloc(&ctx->h_stage, cap) != hipSuccess) {       // no pinned memory: fall back to the direct copy
            ctx->h_stage = nullptr;
            NHP_HIP(ctx, hipMemcpyAsync(dst, d_src, bytes, hipMemcpyDeviceToHost, ctx->stream));
            NHP_HIP(ctx, hipStreamSynchronize(ctx->stream));
            return NHP_OK;
        }
        ctx->stage_cap = cap;
    }
    NHP_HIP(ctx, hipMemcpyAsync(ctx->h_stage, d_src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    NHP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    memcpy(dst, ctx->h_stage, bytes);
    return NHP_OK;
}

nhp_status nhp_ctx_reserve_scratch(nhp_ctx *ctx, size_t bytes)
{
    if (bytes <= ctx->scratch_cap) return NHP_OK;
    NHP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->d_scratch) (void)hipFree(ctx->d_scratch);
    ctx->d_scratch = nullptr;
    ctx->scratch_cap = 0;
    NHP_HIP(ctx, hipMalloc(&ctx->d_scratch, bytes));
    ctx->scratch_cap = bytes;
    return NHP_OK;
}

extern "C" void nhp_uniform_stream(uint64_t seed, uint64_t step, int64_t n, double *u)
{
    for (int64_t i = 0; i < n; ++i) u[i] = nhp_philox_uniform(seed, step, (uint64_t)i);
}
