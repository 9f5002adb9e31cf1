// ccx_rccl.hip -- the one collective of the multi-GPU layout, called on RCCL directly.
//
// Env shards never exchange data; the only reduction of a job is the SUM of the six u64
// throughput counters over the ranks, once per measurement window.  ccx_rccl_allreduce_counters enqueues
// it on the handle's stream: reduce the per-tile partial counters to the six totals, then ncclAllReduce
// those 48 bytes over xGMI -- device to device, no host round trip.  The reference has no counterpart
// (its parallelism is one env per RLlib EnvRunner process, examples/training_script.py:84).
//
// librccl is bound at run time (dlopen) rather than at link time: libccx.so must load on a box without
// RCCL, and inside a PyTorch process it must use the very librccl.so.1 torch has already loaded (same
// soname -> the loader hands back that copy) instead of dragging in a second RCCL.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cstring>
#include <mutex>

#include "ccx_internal.h"

using ccxi::fail;

namespace {

struct Rccl {
    void* lib = nullptr;
    ncclResult_t (*get_unique_id)(ncclUniqueId*) = nullptr;
    ncclResult_t (*comm_init_rank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*comm_destroy)(ncclComm_t) = nullptr;
    ncclResult_t (*comm_count)(const ncclComm_t, int*) = nullptr;
    ncclResult_t (*all_reduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*error_string)(ncclResult_t) = nullptr;
    char why[256] = "";
};

Rccl* rccl(const char** why = nullptr) {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char* n : names) {
            r.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
            if (r.lib) break;
        }
        if (!r.lib) {
            snprintf(r.why, sizeof(r.why), "librccl.so.1 not found (%s)", dlerror());
            return;
        }
        auto sym = [&](const char* name) {
            void* p = dlsym(r.lib, name);
            if (!p && !r.why[0]) snprintf(r.why, sizeof(r.why), "librccl lacks %s", name);
            return p;
        };
        r.get_unique_id = reinterpret_cast<decltype(r.get_unique_id)>(sym("ncclGetUniqueId"));
        r.comm_init_rank = reinterpret_cast<decltype(r.comm_init_rank)>(sym("ncclCommInitRank"));
        r.comm_destroy = reinterpret_cast<decltype(r.comm_destroy)>(sym("ncclCommDestroy"));
        r.comm_count = reinterpret_cast<decltype(r.comm_count)>(sym("ncclCommCount"));
        r.all_reduce = reinterpret_cast<decltype(r.all_reduce)>(sym("ncclAllReduce"));
        r.error_string = reinterpret_cast<decltype(r.error_string)>(sym("ncclGetErrorString"));
    });
    if (why) *why = r.why;
    return (r.lib && !r.why[0]) ? &r : nullptr;
}

int no_rccl() {
    const char* why = "";
    (void)rccl(&why);
    return fail(CCX_ENODEVICE, "RCCL unavailable: %s", why);
}

#define CCX_RCCL(R, call)                                                                          \
    do {                                                                                           \
        ncclResult_t r_ = (call);                                                                  \
        if (r_ != ncclSuccess)                                                                     \
            return fail(CCX_EHIP, "%s failed: %s", #call, (R)->error_string ? (R)->error_string(r_) : "?"); \
    } while (0)

}  // namespace

extern "C" {

int ccx_rccl_unique_id(void* id_out_128) {
    static_assert(sizeof(ncclUniqueId) == CCX_RCCL_UNIQUE_ID_BYTES, "ncclUniqueId size");
    if (!id_out_128) return fail(CCX_EINVAL, "NULL argument");
    Rccl* R = rccl();
    if (!R) return no_rccl();
    ncclUniqueId id;
    CCX_RCCL(R, R->get_unique_id(&id));
    memcpy(id_out_128, &id, sizeof(id));
    return CCX_OK;
}

int ccx_rccl_comm_create(int32_t num_ranks, const void* id_128, int32_t rank, int32_t device, void** comm_out) {
    if (!id_128 || !comm_out) return fail(CCX_EINVAL, "NULL argument");
    if (num_ranks < 1 || rank < 0 || rank >= num_ranks) return fail(CCX_EINVAL, "rank %d outside 0..%d", rank, num_ranks - 1);
    Rccl* R = rccl();
    if (!R) return no_rccl();
    CCX_HIP(hipSetDevice(device));
    ncclUniqueId id;
    memcpy(&id, id_128, sizeof(id));
    ncclComm_t comm = nullptr;
    CCX_RCCL(R, R->comm_init_rank(&comm, num_ranks, id, rank));
    *comm_out = comm;
    return CCX_OK;
}

int ccx_rccl_comm_destroy(void* comm) {
    if (!comm) return CCX_OK;
    Rccl* R = rccl();
    if (!R) return no_rccl();
    CCX_RCCL(R, R->comm_destroy(static_cast<ncclComm_t>(comm)));
    return CCX_OK;
}

int ccx_rccl_allreduce_counters(ccx_handle* h, void* rccl_comm, uint64_t* out_device, int32_t* num_ranks) {
    if (!h || !rccl_comm || !out_device) return fail(CCX_EINVAL, "NULL argument");
    Rccl* R = rccl();
    if (!R) return no_rccl();
    CCX_HIP(hipSetDevice(h->device));
    ncclComm_t comm = static_cast<ncclComm_t>(rccl_comm);
    if (num_ranks) {
        int n = 0;
        CCX_RCCL(R, R->comm_count(comm, &n));
        *num_ranks = n;
    }
    // the rollout kernel leaves per-tile partial counters: sum them into the six totals first
    CCX_HIP(ccx::launch_reduce_counters(h->stream, h->counters, h->E));
    CCX_RCCL(R, R->all_reduce(h->counters, out_device, 6, ncclUint64, ncclSum, comm, h->stream));
    return CCX_OK;
}

}  // extern "C"
