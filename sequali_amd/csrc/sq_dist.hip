/*
 * sq_dist.hip -- the job's one exchange step (the final all-reduce of the count tables over xGMI) in the C ABI,
 * for a host that binds libsqgpu.so without torch.  sequali_amd/dist.py does the same through
 * torch.distributed's "nccl" backend, which IS RCCL on ROCm; here RCCL is called directly.
 *
 * librccl.so is opened when the first of these entry points is called (dlopen, the seven symbols below): a process
 * that runs on one GPU never maps it, and libsqgpu.so has no link-time dependency on it.
 *
 * One process per GPU.  The ranks make a communicator out of band: rank 0 calls sq_rccl_unique_id and hands the 128
 * bytes to the others by whatever it has (a file, MPI, a socket); every rank calls sq_rccl_comm_init.  Then, after
 * the pass over its shard:
 *     sq_qcmetrics_allreduce(m, comm)        -- max_length agreed (all-reduce max), tables padded, summed in place,
 *     sq_adaptercounter_allreduce(a, comm)      totals set: every rank holds the job's tables
 * PerTileQuality / OverrepresentedSequences / DedupEstimator / InsertSizeMetrics merge through the sq_*_shard_* entry
 * points (include/sqgpu.h "multi-GPU"); their collectives are ragged gathers of small candidate lists, which
 * sq_rccl_allgather_bytes serves.
 *
 * No N > 1 run over RCCL has happened yet (no node with more than one GPU has been available to this build).  What has
 * run: a communicator of ONE rank on one MI355X (tests/test_gpu_rccl.py: ncclCommInitRank, the grouped all-reduces of
 * both modules, the all-gather); the enum values below are pinned against rccl.h where the header is installed
 * (tests/test_boundary_cpu.py).
 */
#include <dlfcn.h>

#include <mutex>
#include <string>

#include "sq_common.h"

namespace {

/* the part of rccl.h this file needs (rccl/rccl.h:40-43, 448-470): kept here so that building libsqgpu.so does not
   need RCCL's headers either */
typedef struct ncclComm *ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef int ncclResult_t;
enum { ncclSuccess = 0 };
enum { ncclSum = 0, ncclMax = 2 };
enum { ncclUint8 = 1, ncclUint64 = 5, ncclFloat64 = 8 };

struct Rccl {
    void *handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
};

Rccl *rccl()
{
    static Rccl R;
    static std::once_flag once;
    static std::string why;   /* what dlopen / dlsym said, kept: dlerror() answers once and forgets */
    std::call_once(once, [] {
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            R.handle = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (R.handle) break;
            const char *e = dlerror();
            why = e ? e : "dlopen failed";
        }
        if (!R.handle) return;
        auto sym = [&](const char *n) { return dlsym(R.handle, n); };
        R.GetUniqueId = (decltype(R.GetUniqueId))sym("ncclGetUniqueId");
        R.CommInitRank = (decltype(R.CommInitRank))sym("ncclCommInitRank");
        R.CommDestroy = (decltype(R.CommDestroy))sym("ncclCommDestroy");
        R.AllReduce = (decltype(R.AllReduce))sym("ncclAllReduce");
        R.AllGather = (decltype(R.AllGather))sym("ncclAllGather");
        R.GroupStart = (decltype(R.GroupStart))sym("ncclGroupStart");
        R.GroupEnd = (decltype(R.GroupEnd))sym("ncclGroupEnd");
        R.GetErrorString = (decltype(R.GetErrorString))sym("ncclGetErrorString");
        if (!R.GetUniqueId || !R.CommInitRank || !R.CommDestroy || !R.AllReduce || !R.AllGather || !R.GroupStart || !R.GroupEnd) {
            dlclose(R.handle);
            R.handle = nullptr;
            why = "one of ncclGetUniqueId, ncclCommInitRank, ncclCommDestroy, ncclAllReduce, ncclAllGather, ncclGroupStart, ncclGroupEnd is missing";
        }
    });
    if (!R.handle) {
        sq_set_error("librccl.so could not be loaded: %s", why.c_str());
        return nullptr;
    }
    return &R;
}

#define SQ_RCCL(R, call)                                                                                       \
    do {                                                                                                       \
        ncclResult_t r_ = (call);                                                                              \
        if (r_ != ncclSuccess) {                                                                               \
            sq_set_error("%s: %s", #call, (R)->GetErrorString ? (R)->GetErrorString(r_) : "RCCL error");        \
            return SQ_ERR_HIP;                                                                                 \
        }                                                                                                      \
    } while (0)

}  // namespace

/* 1 when librccl.so and its entry points are there (nothing else is touched: no device, no communicator) */
SQ_EXPORT int sq_rccl_available(void) { return rccl() != nullptr; }

/* ncclGetUniqueId: 128 bytes that rank 0 hands to every rank of the job */
SQ_EXPORT int sq_rccl_unique_id(uint8_t *out128)
{
    Rccl *R = rccl();
    if (!R) return SQ_ERR_SYSTEM;
    ncclUniqueId id;
    SQ_RCCL(R, R->GetUniqueId(&id));
    memcpy(out128, id.internal, 128);
    return SQ_OK;
}

/* ncclCommInitRank on the context's device; NULL on failure (sq_last_error) */
SQ_EXPORT void *sq_rccl_comm_init(sq_ctx *ctx, int n_ranks, const uint8_t *id128, int rank)
{
    Rccl *R = rccl();
    if (!R) return nullptr;
    if (!ctx) { sq_set_error("sq_rccl_comm_init: no context (one process per GPU: sq_init first)"); return nullptr; }
    if (hipSetDevice(ctx->device) != hipSuccess) { sq_set_error("sq_rccl_comm_init: hipSetDevice(%d) failed", ctx->device); return nullptr; }
    ncclUniqueId id;
    memcpy(id.internal, id128, 128);
    ncclComm_t comm = nullptr;
    const ncclResult_t r = R->CommInitRank(&comm, n_ranks, id, rank);
    if (r != ncclSuccess) {
        sq_set_error("ncclCommInitRank: %s", R->GetErrorString ? R->GetErrorString(r) : "RCCL error");
        return nullptr;
    }
    return comm;
}

SQ_EXPORT void sq_rccl_comm_destroy(void *comm)
{
    Rccl *R = rccl();
    if (R && comm) (void)R->CommDestroy((ncclComm_t)comm);
}

/* In-place all-reduce of n device arrays of 8-byte elements over the communicator, on the context's stream (behind
 * the pass that filled them), as ONE group: op 0 = sum of u64 counters, 1 = sum of f64, 2 = max of u64.  Every rank
 * passes arrays of the same shapes. */
SQ_EXPORT int sq_rccl_allreduce_tables(sq_ctx *ctx, void *comm, void *const *ptrs, const uint64_t *counts, size_t n, int op)
{
    Rccl *R = rccl();
    if (!R) return SQ_ERR_SYSTEM;
    if (!ctx || !comm) { sq_set_error("sq_rccl_allreduce_tables: no context or no communicator"); return SQ_ERR_VALUE; }
    if (op < 0 || op > 2) { sq_set_error("sq_rccl_allreduce_tables: op must be 0 (sum u64), 1 (sum f64) or 2 (max u64)"); return SQ_ERR_VALUE; }
    SQ_RCCL(R, R->GroupStart());
    for (size_t i = 0; i < n; i++) {
        if (!ptrs[i] || !counts[i]) continue;
        const ncclResult_t r = R->AllReduce(ptrs[i], ptrs[i], (size_t)counts[i], op == 1 ? ncclFloat64 : ncclUint64, op == 2 ? ncclMax : ncclSum,
                                            (ncclComm_t)comm, ctx->stream);
        if (r != ncclSuccess) {
            (void)R->GroupEnd();
            sq_set_error("ncclAllReduce: %s", R->GetErrorString ? R->GetErrorString(r) : "RCCL error");
            return SQ_ERR_HIP;
        }
    }
    SQ_RCCL(R, R->GroupEnd());
    return SQ_OK;
}

/* every rank's `bytes` bytes at d_send, rank after rank, in d_recv (n_ranks * bytes) on every rank */
SQ_EXPORT int sq_rccl_allgather_bytes(sq_ctx *ctx, void *comm, const void *d_send, void *d_recv, size_t bytes)
{
    Rccl *R = rccl();
    if (!R) return SQ_ERR_SYSTEM;
    if (!ctx || !comm) { sq_set_error("sq_rccl_allgather_bytes: no context or no communicator"); return SQ_ERR_VALUE; }
    SQ_RCCL(R, R->AllGather(d_send, d_recv, bytes, ncclUint8, (ncclComm_t)comm, ctx->stream));
    return SQ_OK;
}
