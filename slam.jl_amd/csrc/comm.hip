// comm.hip -- RCCL collectives behind the C ABI (slam_comm_*), so that a non-Python host (the Julia shim) can drive the
// point-sharded bundle adjustment of SURVEY 8e: one all-reduce of the reduced camera system and one all-gather of the
// trial costs per LM iteration, enqueued on the context's own HIP stream -- no cross-stream synchronisation, no host sync.
//
// librccl is bound at run time (dlopen / dlsym): libslamhip has no link-time dependency on it, a process that never
// creates a communicator never loads it, and a host that already carries an RCCL (e.g. PyTorch's bundled copy) shares it.
#include "common.hpp"
#include <dlfcn.h>
#include <rccl/rccl.h>

struct slam_comm {
    ncclComm_t comm = nullptr;
    int nranks = 1, rank = 0, device = 0;
};

namespace {
struct Rccl {
    void *lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    std::string err;
};
Rccl &rccl()
{
    static Rccl r;
    static bool tried = false;
    if (tried) return r;
    tried = true;
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
    for (const char *nm : names) if ((r.lib = dlopen(nm, RTLD_NOW | RTLD_NOLOAD))) break;      // one already in the process (the host's)
    if (!r.lib) for (const char *nm : names) if ((r.lib = dlopen(nm, RTLD_NOW | RTLD_LOCAL))) break;
    if (!r.lib) { r.err = std::string("librccl not found: ") + (dlerror() ? dlerror() : ""); return r; }
#define BIND(field, sym) r.field = (decltype(r.field))dlsym(r.lib, sym); if (!r.field) { r.err = std::string("librccl lacks ") + sym; r.lib = nullptr; return r; }
    BIND(GetUniqueId, "ncclGetUniqueId") BIND(CommInitRank, "ncclCommInitRank") BIND(CommDestroy, "ncclCommDestroy")
    BIND(AllReduce, "ncclAllReduce") BIND(AllGather, "ncclAllGather") BIND(GetErrorString, "ncclGetErrorString")
#undef BIND
    return r;
}
}  // namespace

#define RCCL_TRY(ctx, expr)                                                                                          \
    do {                                                                                                             \
        ncclResult_t _r = (expr);                                                                                    \
        if (_r != ncclSuccess) return slam_fail((ctx), SLAM_ERR_HIP, "%s: %s", #expr, rccl().GetErrorString(_r));  \
    } while (0)

extern "C" {

int slam_comm_unique_id(void *id128)
{
    if (!id128) return slam_fail(nullptr, SLAM_ERR_ARG, "slam_comm_unique_id: id128 is NULL");
    Rccl &R = rccl();
    if (!R.lib) return slam_fail(nullptr, SLAM_ERR_HIP, "slam_comm_unique_id: %s", R.err.c_str());
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
    ncclUniqueId id;
    RCCL_TRY(nullptr, R.GetUniqueId(&id));
    memcpy(id128, &id, 128);
    return SLAM_OK;
}

int slam_comm_create(slam_ctx *ctx, int nranks, int rank, const void *id128, slam_comm **out)
{
    ARG_TRY(ctx, ctx != nullptr && out != nullptr && id128 != nullptr && nranks >= 1 && rank >= 0 && rank < nranks);
    Rccl &R = rccl();
    if (!R.lib) return slam_fail(ctx, SLAM_ERR_HIP, "slam_comm_create: %s", R.err.c_str());
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    ncclUniqueId id;
    memcpy(&id, id128, 128);
    slam_comm *c = new slam_comm();
    c->nranks = nranks; c->rank = rank; c->device = ctx->device;
    ncclResult_t r = R.CommInitRank(&c->comm, nranks, id, rank);
    if (r != ncclSuccess) { delete c; return slam_fail(ctx, SLAM_ERR_HIP, "ncclCommInitRank: %s", R.GetErrorString(r)); }
    *out = c;
    return SLAM_OK;
}

int slam_comm_destroy(slam_comm *c)
{
    if (!c) return SLAM_OK;
    (void)hipSetDevice(c->device);
    if (c->comm && rccl().lib) (void)rccl().CommDestroy(c->comm);
    delete c;
    return SLAM_OK;
}

int slam_comm_size(const slam_comm *c) { return c ? c->nranks : SLAM_ERR_ARG; }
int slam_comm_rank(const slam_comm *c) { return c ? c->rank : SLAM_ERR_ARG; }

// in-place sum over all ranks of buf_dev[0 .. count), Float64, enqueued on ctx's stream (returns after enqueueing)
int slam_comm_allreduce_sum(slam_ctx *ctx, slam_comm *c, double *buf_dev, int64_t count)
{
    ARG_TRY(ctx, ctx != nullptr && c != nullptr && buf_dev != nullptr && count >= 0);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (count == 0) return SLAM_OK;
    RCCL_TRY(ctx, rccl().AllReduce(buf_dev, buf_dev, (size_t)count, ncclDouble, ncclSum, c->comm, ctx->stream));
    return SLAM_OK;
}

// recv_dev[r * count .. (r + 1) * count) = rank r's send_dev[0 .. count), enqueued on ctx's stream
int slam_comm_allgather(slam_ctx *ctx, slam_comm *c, const double *send_dev, double *recv_dev, int64_t count)
{
    ARG_TRY(ctx, ctx != nullptr && c != nullptr && send_dev != nullptr && recv_dev != nullptr && count >= 0);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (count == 0) return SLAM_OK;
    RCCL_TRY(ctx, rccl().AllGather(send_dev, recv_dev, (size_t)count, ncclDouble, c->comm, ctx->stream));
    return SLAM_OK;
}

}  // extern "C"
