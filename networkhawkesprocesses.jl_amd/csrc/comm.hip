// RCCL over xGMI behind the C ABI: one nhp_comm per rank, bound to that rank's nhp_ctx.  The library
// hands RCCL device pointers on the ctx stream -- the partial log-likelihood in ctx->d_results, the gradient in
// ctx->d_scratch, a chain's running moments in model->d_mom -- so reduced results cross PCIe once.
//
// librccl.so.1 is opened with dlopen on first use rather than linked: a single-GPU host (and this library's CPU-side
// symbol tests) need no RCCL, and in a process that already holds a copy (torch bundles one under the same soname) the
// loader hands back that copy instead of mapping a second one.
#include <dlfcn.h>
#include <string.h>

#include <rccl/rccl.h>

#include "nhp_internal.h"

namespace {
struct rccl_api {
    void *handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    bool ok = false;
    std::string why;            // why not, captured once: dlerror() hands its message out a single time
};

rccl_api &rccl()
{
    static rccl_api api = [] {
        rccl_api a;
        const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char *n : names) {
            if ((a.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL)) != nullptr) break;
            const char *e = dlerror();
            a.why = e ? e : "dlopen(librccl.so.1) failed";
        }
        if (!a.handle) return a;
        a.GetUniqueId = (decltype(a.GetUniqueId))dlsym(a.handle, "ncclGetUniqueId");
        a.CommInitRank = (decltype(a.CommInitRank))dlsym(a.handle, "ncclCommInitRank");
        a.CommDestroy = (decltype(a.CommDestroy))dlsym(a.handle, "ncclCommDestroy");
        a.AllReduce = (decltype(a.AllReduce))dlsym(a.handle, "ncclAllReduce");
        a.AllGather = (decltype(a.AllGather))dlsym(a.handle, "ncclAllGather");
        a.GetErrorString = (decltype(a.GetErrorString))dlsym(a.handle, "ncclGetErrorString");
        a.ok = a.GetUniqueId && a.CommInitRank && a.CommDestroy && a.AllReduce && a.AllGather && a.GetErrorString;
        if (!a.ok) {                                            // an RCCL without a symbol this library calls: say so, keep nothing open
            a.why = "librccl lacks a required symbol";
            (void)dlclose(a.handle);
            a.handle = nullptr;
        }
        return a;
    }();
    return api;
}

nhp_status need_rccl(nhp_ctx *ctx)
{
    if (rccl().ok) return NHP_OK;
    nhp_set_error(ctx, "RCCL is not available: %s", rccl().why.c_str());
    return NHP_ERCCL;
}
}   // namespace

#define NHP_RCCL(ctx, call)                                                                                   \
    do {                                                                                                      \
        ncclResult_t r_ = (call);                                                                             \
        if (r_ != ncclSuccess) {                                                                              \
            nhp_set_error(ctx, "%s failed: %s (%s:%d)", #call, rccl().GetErrorString(r_), __FILE__, __LINE__); \
            return NHP_ERCCL;                                                                                 \
        }                                                                                                     \
    } while (0)

static_assert(NHP_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "the C ABI carries ncclUniqueId as opaque bytes");

extern "C" nhp_status nhp_comm_unique_id(uint8_t *id)
{
    if (!id) return NHP_EINVAL;
    NHP_TRY(need_rccl(nullptr));
    ncclUniqueId u;
    NHP_RCCL(nullptr, rccl().GetUniqueId(&u));
    memcpy(id, u.internal, NHP_COMM_ID_BYTES);
    return NHP_OK;
}

extern "C" nhp_status nhp_comm_create(nhp_ctx *ctx, const uint8_t *id, int32_t rank, int32_t world, nhp_comm **out)
{
    if (!ctx || !id || !out) return NHP_EINVAL;
    *out = nullptr;
    if (world < 1 || rank < 0 || rank >= world) { nhp_set_error(ctx, "comm_create: rank %d of %d", rank, world); return NHP_EINVAL; }
    NHP_TRY(need_rccl(ctx));
    NHP_HIP(ctx, hipSetDevice(ctx->device));
    ncclUniqueId u;
    memcpy(u.internal, id, NHP_COMM_ID_BYTES);
    ncclComm_t c = nullptr;
    NHP_RCCL(ctx, rccl().CommInitRank(&c, world, u, rank));
    nhp_comm *comm = new nhp_comm();
    comm->ctx = ctx; comm->nccl = c; comm->rank = rank; comm->world = world;
    *out = comm;
    return NHP_OK;
}

extern "C" void nhp_comm_destroy(nhp_comm *comm)
{
    if (!comm) return;
    if (comm->ctx) {
        (void)hipSetDevice(comm->ctx->device);
        (void)hipStreamSynchronize(comm->ctx->stream);
    }
    if (comm->nccl && rccl().ok) (void)rccl().CommDestroy((ncclComm_t)comm->nccl);
    delete comm;
}

extern "C" int32_t nhp_comm_rank(const nhp_comm *comm) { return comm ? comm->rank : -1; }
extern "C" int32_t nhp_comm_world(const nhp_comm *comm) { return comm ? comm->world : -1; }

static nhp_status check_comm(nhp_ctx *ctx, const nhp_comm *comm)
{
    if (!ctx || !comm) return NHP_EINVAL;
    if (comm->ctx != ctx) { nhp_set_error(ctx, "communicator belongs to another ctx"); return NHP_EINVAL; }
    return NHP_OK;
}

nhp_status nhp_comm_allreduce_dev(nhp_ctx *ctx, nhp_comm *comm, double *d_buf, size_t n)
{
    NHP_TRY(check_comm(ctx, comm));
    if (n == 0) return NHP_OK;
    NHP_RCCL(ctx, rccl().AllReduce(d_buf, d_buf, n, ncclDouble, ncclSum, (ncclComm_t)comm->nccl, ctx->stream));
    return NHP_OK;
}

nhp_status nhp_comm_allgather_dev(nhp_ctx *ctx, nhp_comm *comm, const double *d_mine, size_t n, double *d_all)
{
    NHP_TRY(check_comm(ctx, comm));
    if (n == 0) return NHP_OK;
    NHP_RCCL(ctx, rccl().AllGather(d_mine, d_all, n, ncclDouble, (ncclComm_t)comm->nccl, ctx->stream));
    return NHP_OK;
}

// ---- host vectors through the context's scratch (control data: link counts, traces) -------------------------------
extern "C" nhp_status nhp_allreduce_sum(nhp_ctx *ctx, nhp_comm *comm, double *x, int64_t n)
{
    NHP_TRY(check_comm(ctx, comm));
    if (n < 0 || (n > 0 && !x)) return NHP_EINVAL;
    if (n == 0) return NHP_OK;
    NHP_HIP(ctx, hipSetDevice(ctx->device));
    NHP_TRY(nhp_ctx_reserve_scratch(ctx, 8 * (size_t)n));
    double *d = (double *)ctx->d_scratch;
    NHP_HIP(ctx, hipMemcpyAsync(d, x, 8 * (size_t)n, hipMemcpyHostToDevice, ctx->stream));
    NHP_TRY(nhp_comm_allreduce_dev(ctx, comm, d, (size_t)n));
    return nhp_download(ctx, x, d, 8 * (size_t)n);
}

extern "C" nhp_status nhp_allgather(nhp_ctx *ctx, nhp_comm *comm, const double *mine, int64_t n, double *all)
{
    NHP_TRY(check_comm(ctx, comm));
    if (n < 0 || (n > 0 && (!mine || !all))) return NHP_EINVAL;
    if (n == 0) return NHP_OK;
    NHP_HIP(ctx, hipSetDevice(ctx->device));
    const size_t w = (size_t)comm->world;
    NHP_TRY(nhp_ctx_reserve_scratch(ctx, 8 * (size_t)n * (w + 1)));
    double *d_mine = (double *)ctx->d_scratch, *d_all = d_mine + n;
    NHP_HIP(ctx, hipMemcpyAsync(d_mine, mine, 8 * (size_t)n, hipMemcpyHostToDevice, ctx->stream));
    NHP_TRY(nhp_comm_allgather_dev(ctx, comm, d_mine, (size_t)n, d_all));
    return nhp_download(ctx, all, d_all, 8 * (size_t)n * w);
}

// ---- one evaluation over all ranks (column shards; DESIGN.md 7, second way) --------------------------------------
extern "C" nhp_status nhp_cont_loglik_allreduce(nhp_ctx *ctx, nhp_comm *comm, const nhp_cont_dataset *ds,
                                                const nhp_cont_model *m, int32_t flags, double *ll)
{
    NHP_TRY(check_comm(ctx, comm));
    if (!ll) return NHP_EINVAL;
    NHP_TRY(nhp_cont_loglik_enqueue(ctx, ds, m, flags, 0));
    NHP_TRY(nhp_comm_allreduce_dev(ctx, comm, ctx->d_results, 1));
    return nhp_ctx_fetch(ctx, 0, 1, ll);
}


__global__ void k_pack_ll(const double *__restrict__ res, double *__restrict__ dst) { *dst = *res; }

// log-likelihood -> ctx->d_results[0] and gradient -> *d_grad, both left on the device; with a communicator the shards'
// [ll; grad] are summed over the ranks first (one collective of P + 1 doubles).  Asynchronous.
nhp_status nhp_grad_enqueue_reduced(nhp_ctx *ctx, nhp_comm *comm, const nhp_cont_dataset *ds, const nhp_cont_model *m, int32_t flags,
                                    int64_t grad_len, double **d_grad_out)
{
    double *d_grad = nullptr;
    NHP_TRY(nhp_grad_enqueue(ctx, ds, m, flags, grad_len, &d_grad));
    *d_grad_out = d_grad;
    if (!comm) return NHP_OK;
    NHP_TRY(check_comm(ctx, comm));
    hipLaunchKernelGGL(k_pack_ll, dim3(1), dim3(1), 0, ctx->stream, ctx->d_results, d_grad - 1);
    NHP_HIP(ctx, hipGetLastError());
    NHP_TRY(nhp_comm_allreduce_dev(ctx, comm, d_grad - 1, (size_t)grad_len + 1));
    NHP_HIP(ctx, hipMemcpyAsync(ctx->d_results, d_grad - 1, 8, hipMemcpyDeviceToDevice, ctx->stream));
    return NHP_OK;
}

extern "C" nhp_status nhp_cont_loglik_grad_allreduce(nhp_ctx *ctx, nhp_comm *comm, const nhp_cont_dataset *ds,
                                                     const nhp_cont_model *m, int32_t flags, double *ll, double *grad,
                                                     int64_t grad_len)
{
    NHP_TRY(check_comm(ctx, comm));
    if (!ll || !grad) return NHP_EINVAL;
    double *d_grad = nullptr;
    NHP_TRY(nhp_grad_enqueue_reduced(ctx, comm, ds, m, flags, grad_len, &d_grad));
    NHP_TRY(nhp_download(ctx, grad, d_grad, 8 * (size_t)grad_len));
    return nhp_ctx_fetch(ctx, 0, 1, ll);
}

// ---- config 5: per-chain summaries, device to device ---------------------------------------------------------------
extern "C" nhp_status nhp_gather_moments(nhp_ctx *ctx, nhp_comm *comm, const nhp_cont_model *m, double *sum_all,
                                         double *sumsq_all, int64_t len, int64_t *counts, double *rho_all)
{
    NHP_TRY(check_comm(ctx, comm));
    if (!m || !sum_all || !sumsq_all || !counts) return NHP_EINVAL;
    if (m->ctx != ctx) { nhp_set_error(ctx, "model belongs to another ctx"); return NHP_EINVAL; }
    if (!m->d_mom) { nhp_set_error(ctx, "moments: nothing accumulated"); return NHP_EINVAL; }
    if (len != m->mom_len) { nhp_set_error(ctx, "Parameter vector length does not match model parameter length."); return NHP_ESHAPE; }
    NHP_HIP(ctx, hipSetDevice(ctx->device));
    const size_t w = (size_t)comm->world, L = (size_t)len;
    // scratch: [world][2 L] gathered moments | mine[4] | all[world][4] (count, ρ, Σρ, Σρ²)
    NHP_TRY(nhp_ctx_reserve_scratch(ctx, 8 * (2 * L * w + 4 + 4 * w)));
    double *d_all = (double *)ctx->d_scratch, *d_mine4 = d_all + 2 * L * w, *d_all4 = d_mine4 + 4;
    NHP_TRY(nhp_comm_allgather_dev(ctx, comm, m->d_mom, 2 * L, d_all));
    double mine4[4] = {(double)m->mom_count, 0.0, 0.0, 0.0};
    NHP_HIP(ctx, hipMemcpyAsync(d_mine4, mine4, 8, hipMemcpyHostToDevice, ctx->stream));
    if (m->d_rho) NHP_HIP(ctx, hipMemcpyAsync(d_mine4 + 1, m->d_rho, 24, hipMemcpyDeviceToDevice, ctx->stream));
    else NHP_HIP(ctx, hipMemsetAsync(d_mine4 + 1, 0, 24, ctx->stream));
    NHP_TRY(nhp_comm_allgather_dev(ctx, comm, d_mine4, 4, d_all4));
    for (size_t r = 0; r < w; ++r) {
        NHP_TRY(nhp_download(ctx, sum_all + r * L, d_all + r * 2 * L, 8 * L));
        NHP_TRY(nhp_download(ctx, sumsq_all + r * L, d_all + r * 2 * L + L, 8 * L));
    }
    std::vector<double> all4(4 * w);
    NHP_TRY(nhp_download(ctx, all4.data(), d_all4, 8 * 4 * w));
    for (size_t r = 0; r < w; ++r) {
        counts[r] = (int64_t)all4[4 * r];
        if (rho_all) { rho_all[3 * r] = all4[4 * r + 1]; rho_all[3 * r + 1] = all4[4 * r + 2]; rho_all[3 * r + 2] = all4[4 * r + 3]; }
    }
    return NHP_OK;
}
