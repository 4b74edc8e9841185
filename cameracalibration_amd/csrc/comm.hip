// comm.hip -- the one collective of the path: the arg-min exchange between the candidate shards
// of a one-process-per-GPU run, over RCCL (xGMI inside a node).
//
// RCCL has no MINLOC reduction, so every rank contributes one 16-byte (cost, index) record to an
// ncclAllGather on the context's stream and resolves the lexicographic minimum itself: the lowest
// index among equal costs wins on every rank, which is what the reference's single strict-'<' scan
// (src/camera_calibration_optimization.cpp:542) would have kept.  128 bytes at 8 ranks: the exchange
// is latency-bound, link bandwidth plays no part.
//
// librccl is opened on first use (dlopen), so the library loads -- and every single-GPU entry point
// works -- on a machine without RCCL, and programs that never exchange do not pay for loading it.
#include "ccal_internal.hpp"

#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <vector>

namespace {

struct Rccl {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
    ncclResult_t (*CommUserRank)(const ncclComm_t, int*) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    std::string error;
};

Rccl& rccl()
{
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, []() {
        for (const char* name : {"librccl.so.1", "/opt/rocm/lib/librccl.so.1", "librccl.so"}) {
            r.handle = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (r.handle) break;
        }
        if (!r.handle) {
            const char* e = dlerror();
            r.error = std::string("cannot load librccl: ") + (e ? e : "unknown error");
            return;
        }
        auto sym = [&](const char* n) -> void* {
            void* p = dlsym(r.handle, n);
            if (!p && r.error.empty()) r.error = std::string("librccl lacks ") + n;
            return p;
        };
        r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(sym("ncclGetUniqueId"));
        r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(sym("ncclCommInitRank"));
        r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(sym("ncclCommDestroy"));
        r.AllGather = reinterpret_cast<decltype(r.AllGather)>(sym("ncclAllGather"));
        r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(sym("ncclGetErrorString"));
        r.CommCount = reinterpret_cast<decltype(r.CommCount)>(sym("ncclCommCount"));
        r.CommUserRank = reinterpret_cast<decltype(r.CommUserRank)>(sym("ncclCommUserRank"));
    });
    return r;
}

struct Record {
    double cost;
    int64_t index;
};
static_assert(sizeof(Record) == 16, "the exchanged record is 16 bytes");

int comm_fail(ccal_ctx* ctx, int code, const std::string& msg)
{
    if (ctx) ctx->err = msg;
    return code;
}

} // namespace

struct ccal_comm {
    ncclComm_t comm = nullptr;
    int world = 1, rank = 0;
    ccal_ctx* ctx = nullptr;
    Record* send = nullptr;   // device, one record
    Record* recv = nullptr;   // device, `world` records
};

static_assert(CCAL_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "ccal.h states RCCL's id size");

extern "C" int ccal_comm_unique_id(char id_out[CCAL_COMM_ID_BYTES])
{
    if (!id_out) return CCAL_ERR_ARG;
    Rccl& r = rccl();
    if (!r.error.empty()) return CCAL_ERR_DEVICE;
    ncclUniqueId id;
    if (r.GetUniqueId(&id) != ncclSuccess) return CCAL_ERR_DEVICE;
    std::memcpy(id_out, id.internal, CCAL_COMM_ID_BYTES);
    return CCAL_OK;
}

extern "C" int ccal_comm_create(ccal_ctx* ctx, int world_size, int rank, const char id[CCAL_COMM_ID_BYTES],
                                ccal_comm** out)
{
    if (!ctx || !out) return CCAL_ERR_ARG;
    *out = nullptr;
    if (!id || world_size < 1 || rank < 0 || rank >= world_size)
        return comm_fail(ctx, CCAL_ERR_ARG, "comm_create: bad rank / world size / id");
    Rccl& r = rccl();
    if (!r.error.empty()) return comm_fail(ctx, CCAL_ERR_DEVICE, r.error);
    if (hipSetDevice(ctx->device) != hipSuccess) return comm_fail(ctx, CCAL_ERR_DEVICE, "comm_create: hipSetDevice failed");
    ccal_comm* c = new (std::nothrow) ccal_comm;
    if (!c) return comm_fail(ctx, CCAL_ERR_DEVICE, "comm_create: out of memory");
    c->world = world_size;
    c->rank = rank;
    c->ctx = ctx;
    ncclUniqueId uid;
    std::memcpy(uid.internal, id, CCAL_COMM_ID_BYTES);
    ncclResult_t rc = r.CommInitRank(&c->comm, world_size, uid, rank);
    if (rc != ncclSuccess) {
        delete c;
        return comm_fail(ctx, CCAL_ERR_DEVICE, std::string("ncclCommInitRank: ") + r.GetErrorString(rc));
    }
    if (hipMalloc(reinterpret_cast<void**>(&c->send), sizeof(Record)) != hipSuccess ||
        hipMalloc(reinterpret_cast<void**>(&c->recv), sizeof(Record) * (size_t)world_size) != hipSuccess) {
        (void)ccal_comm_destroy(c);
        return comm_fail(ctx, CCAL_ERR_DEVICE, "comm_create: hipMalloc failed");
    }
    *out = c;
    return CCAL_OK;
}

extern "C" int ccal_comm_destroy(ccal_comm* c)
{
    if (!c) return CCAL_ERR_ARG;
    if (c->ctx) (void)hipSetDevice(c->ctx->device);
    if (c->send) (void)hipFree(c->send);
    if (c->recv) (void)hipFree(c->recv);
    if (c->comm) (void)rccl().CommDestroy(c->comm);
    delete c;
    return CCAL_OK;
}

// What RCCL itself says about the communicator (not what the caller passed to ccal_comm_create).
extern "C" int ccal_comm_info(const ccal_comm* c, int* world_size, int* rank)
{
    if (!c || !c->comm) return CCAL_ERR_ARG;
    Rccl& r = rccl();
    int n = 0, k = 0;
    if (r.CommCount(c->comm, &n) != ncclSuccess || r.CommUserRank(c->comm, &k) != ncclSuccess) return CCAL_ERR_DEVICE;
    if (world_size) *world_size = n;
    if (rank) *rank = k;
    return CCAL_OK;
}

extern "C" int ccal_argmin_allgather(ccal_ctx* ctx, ccal_comm* c, double cost, int64_t index, double* best_cost,
                                     int64_t* best_index, int* winner_rank)
{
    if (!ctx || !c || c->ctx != ctx) return ctx ? comm_fail(ctx, CCAL_ERR_ARG, "argmin_allgather: communicator of another context") : CCAL_ERR_ARG;
    if (!best_cost || !best_index) return comm_fail(ctx, CCAL_ERR_ARG, "argmin_allgather: null argument");
    Rccl& r = rccl();
    if (hipSetDevice(ctx->device) != hipSuccess) return comm_fail(ctx, CCAL_ERR_DEVICE, "argmin_allgather: hipSetDevice failed");
    const Record mine{cost, index};
    std::vector<Record> all((size_t)c->world);
    hipError_t e = hipMemcpyAsync(c->send, &mine, sizeof mine, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) {
        const ncclResult_t rc = r.AllGather(c->send, c->recv, sizeof(Record), ncclUint8, c->comm, ctx->stream);
        if (rc != ncclSuccess) return comm_fail(ctx, CCAL_ERR_DEVICE, std::string("ncclAllGather: ") + r.GetErrorString(rc));
        e = hipMemcpyAsync(all.data(), c->recv, sizeof(Record) * all.size(), hipMemcpyDeviceToHost, ctx->stream);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) return comm_fail(ctx, CCAL_ERR_DEVICE, std::string("argmin_allgather: ") + hipGetErrorString(e));
    // a rank whose search failed still joins the exchange (nobody is left waiting) and says so
    for (int k = 0; k < c->world; ++k)
        if (all[k].index == CCAL_SHARD_FAILED)
            return comm_fail(ctx, CCAL_ERR_STATE, "argmin_allgather: rank " + std::to_string(k) + " reported a failed search");
    // lexicographic (cost, index) minimum, empty shards (index < 0) skipped: the same rule as ccal_argmin_merge
    int w = -1;
    for (int k = 0; k < c->world; ++k) {
        if (all[k].index < 0) continue;
        if (w < 0 || all[k].cost < all[w].cost || (all[k].cost == all[w].cost && all[k].index < all[w].index)) w = k;
    }
    *best_cost = w >= 0 ? all[w].cost : 1e9;      // :531
    *best_index = w >= 0 ? all[w].index : -1;
    if (winner_rank) *winner_rank = w;
    return CCAL_OK;
}
