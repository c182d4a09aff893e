// Native transport for the one exchange step of the path (sharded single-group relax, ochip_relax_set_shard): RCCL
// all-gathers on the context's compute stream, in place on the solver's record arrays.  The solver's kernels before and
// after the exchange are on the same stream, so nothing waits on the host: the all-gather is one more item of the launch
// sequence.  xGMI is point to point, so a ring all-gather of 4 MB per evaluation (C3: 8 982 pair records of 56 doubles)
// is latency-bound, not link-bound; one ncclGroup of the three arrays keeps it at one ring setup per evaluation.
//
// librccl is resolved at run time (dlopen "librccl.so.1"): libochip.so loads on machines without it, and in a process
// that already holds PyTorch's RCCL the soname resolves to that instance instead of a second copy.
// Replaces: SURVEY.md section 8b's ochip_allreduce_normal_eq / ncclAllReduce (the all-gather of per-pair records moves
// 4 MB where an all-reduce of the dense normal equations would move 72 MB).
#include <dlfcn.h>

#include <cstring>

#include "ctx.hpp"

// The library is bound at run time (dlopen below), so the header is only needed for its types; a ROCm install without
// the RCCL development files still builds libochip.so from the few declarations of the public NCCL ABI used here.
#if __has_include(<rccl/rccl.h>)
#include <rccl/rccl.h>
static_assert(OCHIP_RCCL_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "ochip.h mirrors ncclUniqueId");
#else
typedef struct ncclComm *ncclComm_t;
typedef struct
{
    char internal[OCHIP_RCCL_ID_BYTES];
} ncclUniqueId;
typedef enum
{
    ncclSuccess = 0
} ncclResult_t;
typedef enum
{
    ncclInt8 = 0,
    ncclChar = 0
} ncclDataType_t;
#endif

namespace
{
struct rccl_api
{
    void *lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    std::string error;
};

rccl_api *load_rccl()
{
    // initialised once, by whichever thread comes first (a function-local static: the language serialises it)
    static rccl_api api = [] {
        rccl_api a;
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"})
            if ((a.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL)))
                break;
        if (!a.lib)
        {
            a.error = std::string("librccl not found: ") + dlerror();
            return a;
        }
        auto sym = [&](const char *n) -> void * {
            void *p = dlsym(a.lib, n);
            if (!p && a.error.empty())
                a.error = std::string("librccl lacks ") + n;
            return p;
        };
        a.GetUniqueId = (decltype(a.GetUniqueId))sym("ncclGetUniqueId");
        a.CommInitRank = (decltype(a.CommInitRank))sym("ncclCommInitRank");
        a.CommDestroy = (decltype(a.CommDestroy))sym("ncclCommDestroy");
        a.AllGather = (decltype(a.AllGather))sym("ncclAllGather");
        a.GroupStart = (decltype(a.GroupStart))sym("ncclGroupStart");
        a.GroupEnd = (decltype(a.GroupEnd))sym("ncclGroupEnd");
        a.GetErrorString = (decltype(a.GetErrorString))sym("ncclGetErrorString");
        return a;
    }();
    return &api;
}
} // namespace

struct ochip_rccl_comm
{
    ochip_ctx *ctx = nullptr;
    rccl_api *api = nullptr;
    ncclComm_t comm = nullptr;
    uint32_t rank = 0, world = 1;
    uint64_t exchanges = 0, bytes = 0;
};

#define OCHIP_RCCL(ctx, api, call)                                                                                          \
    do                                                                                                                       \
    {                                                                                                                        \
        const ncclResult_t r__ = (call);                                                                                     \
        if (r__ != ncclSuccess)                                                                                              \
            return ochip_fail((ctx), OCHIP_EHIP, "%s failed: %s", #call, (api)->GetErrorString(r__));                        \
    } while (0)

extern "C"
{

int ochip_rccl_unique_id(ochip_ctx *ctx, uint8_t *id)
{
    if (!ctx || !id)
        return OCHIP_EINVAL;
    rccl_api *api = load_rccl();
    if (!api->error.empty())
        return ochip_fail(ctx, OCHIP_EHIP, "%s", api->error.c_str());
    ncclUniqueId u;
    OCHIP_RCCL(ctx, api, api->GetUniqueId(&u));
    std::memcpy(id, u.internal, OCHIP_RCCL_ID_BYTES);
    return OCHIP_OK;
}

int ochip_rccl_comm_create(ochip_ctx *ctx, const uint8_t *id, uint32_t rank, uint32_t world, ochip_rccl_comm **out)
{
    if (!ctx || !id || !out || world == 0 || rank >= world)
        return OCHIP_EINVAL;
    rccl_api *api = load_rccl();
    if (!api->error.empty())
        return ochip_fail(ctx, OCHIP_EHIP, "%s", api->error.c_str());
    OCHIP_HIP(ctx, hipSetDevice(ctx->device));
    ncclUniqueId u;
    std::memcpy(u.internal, id, OCHIP_RCCL_ID_BYTES);
    ncclComm_t comm = nullptr;
    OCHIP_RCCL(ctx, api, api->CommInitRank(&comm, (int)world, u, (int)rank));
    ochip_rccl_comm *c = new ochip_rccl_comm();
    c->ctx = ctx;
    c->api = api;
    c->comm = comm;
    c->rank = rank;
    c->world = world;
    *out = c;
    return OCHIP_OK;
}

void ochip_rccl_comm_destroy(ochip_rccl_comm *c)
{
    if (!c)
        return;
    if (c->comm)
    {
        (void)hipStreamSynchronize(c->ctx->stream);
        c->api->CommDestroy(c->comm);
    }
    delete c;
}

int ochip_rccl_comm_stats(const ochip_rccl_comm *c, uint64_t *exchanges, uint64_t *bytes_gathered)
{
    if (!c)
        return OCHIP_EINVAL;
    if (exchanges)
        *exchanges = c->exchanges;
    if (bytes_gathered)
        *bytes_gathered = c->bytes;
    return OCHIP_OK;
}

// ochip_relax_exchange_fn with user = the ochip_rccl_comm: rank r's slice of every array sits at ptr + r * bytes, the
// all-gathers are in place (sendbuff inside recvbuff at the rank's offset, which RCCL special-cases) and enqueued on the
// context's stream; they are complete for every later item of that stream, which is all the solver needs.
int ochip_rccl_relax_exchange(void *user, void *acc_dev, uint64_t acc_bytes, void *cost_dev, uint64_t cost_bytes, void *fail_dev,
                              uint64_t fail_bytes)
{
    ochip_rccl_comm *c = static_cast<ochip_rccl_comm *>(user);
    if (!c || !c->comm)
        return OCHIP_EINVAL;
    ochip_ctx *ctx = c->ctx;
    rccl_api *api = c->api;
    struct part
    {
        void *p;
        uint64_t n;
    } parts[3] = {{acc_dev, acc_bytes}, {cost_dev, cost_bytes}, {fail_dev, fail_bytes}};
    OCHIP_RCCL(ctx, api, api->GroupStart());
    ncclResult_t first_error = ncclSuccess; // the group is closed whatever happens inside it
    for (const part &q : parts)
        if (q.p && q.n && first_error == ncclSuccess)
        {
            first_error = api->AllGather(static_cast<char *>(q.p) + (size_t)c->rank * q.n, q.p, (size_t)q.n, ncclChar, c->comm,
                                         ctx->stream);
            c->bytes += q.n * c->world;
        }
    const ncclResult_t end = api->GroupEnd();
    if (first_error != ncclSuccess || end != ncclSuccess)
        return ochip_fail(ctx, OCHIP_EHIP, "ncclAllGather of the relax records failed: %s",
                          api->GetErrorString(first_error != ncclSuccess ? first_error : end));
    c->exchanges++;
    return OCHIP_OK;
}

} // extern "C"
