// The multi-GPU step of the draw-list description as native calls: RCCL's all-gather on the caller's communicator
// and stream, followed by the library's own rebase of the gathered firstTask values on the same stream.
// One process per GPU; the reference has no counterpart (single device, Engine.cpp:658-666).
//
// RCCL is looked up at run time (dlopen), not linked: the library a process has already loaded - PyTorch brings its
// own copy - is reused, a C++ host without PyTorch gets the system's, and libmcull.so itself loads where no RCCL is
// installed (the entry points below then return MCULL_E_STATE).  Types come from <rccl/rccl.h>; nothing else of it.
#include <dlfcn.h>
#include <cstdio>
#include <cstring>
#include <rccl/rccl.h>

#include <mutex>

#include "mcull_internal.h"

namespace {

struct RcclApi {
    void *lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int *) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    bool ok = false;
};

RcclApi g_rccl;
std::once_flag g_rccl_once;
thread_local char t_error[256];

void load_rccl() {
    const char *names[] = {"librccl.so.1", "librccl.so"};
    for (int pass = 0; pass < 2 && !g_rccl.lib; pass++)  // first whatever the process already holds, then the system's
        for (const char *n : names) {
            g_rccl.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL | (pass == 0 ? RTLD_NOLOAD : 0));
            if (g_rccl.lib) break;
        }
    if (!g_rccl.lib) return;
#define MCULL_SYM(field, name) g_rccl.field = reinterpret_cast<decltype(g_rccl.field)>(dlsym(g_rccl.lib, name))
    MCULL_SYM(GetUniqueId, "ncclGetUniqueId");
    MCULL_SYM(CommInitRank, "ncclCommInitRank");
    MCULL_SYM(CommDestroy, "ncclCommDestroy");
    MCULL_SYM(CommCount, "ncclCommCount");
    MCULL_SYM(AllGather, "ncclAllGather");
    MCULL_SYM(GetErrorString, "ncclGetErrorString");
#undef MCULL_SYM
    g_rccl.ok = g_rccl.GetUniqueId && g_rccl.CommInitRank && g_rccl.CommDestroy && g_rccl.AllGather;
}

const RcclApi *rccl() {
    std::call_once(g_rccl_once, load_rccl);
    if (!g_rccl.ok) snprintf(t_error, sizeof t_error, "no usable RCCL library (librccl.so.1) in this process or on the library path");
    return g_rccl.ok ? &g_rccl : nullptr;
}

int rccl_fail(const RcclApi *api, const char *what, ncclResult_t r) {
    snprintf(t_error, sizeof t_error, "%s failed: %s", what, api->GetErrorString ? api->GetErrorString(r) : "RCCL error");
    return MCULL_E_HIP;
}

}  // namespace

extern "C" {

const char *mcull_collective_last_error(void) { return t_error; }

int mcull_rccl_unique_id(void *id_128_bytes) {
    if (!id_128_bytes) return MCULL_E_INVALID;
    const RcclApi *api = rccl();
    if (!api) return MCULL_E_STATE;
    static_assert(sizeof(ncclUniqueId) == MCULL_RCCL_UNIQUE_ID_BYTES, "mcull.h states the size of an RCCL unique id");
    const ncclResult_t r = api->GetUniqueId(reinterpret_cast<ncclUniqueId *>(id_128_bytes));
    return r == ncclSuccess ? MCULL_OK : rccl_fail(api, "ncclGetUniqueId", r);
}

int mcull_rccl_comm_create(const void *id_128_bytes, int world, int rank, int device_ordinal, void **comm) {
    if (!id_128_bytes || !comm || world <= 0 || rank < 0 || rank >= world) return MCULL_E_INVALID;
    *comm = nullptr;
    const RcclApi *api = rccl();
    if (!api) return MCULL_E_STATE;
    if (hipSetDevice(device_ordinal) != hipSuccess) {
        snprintf(t_error, sizeof t_error, "hipSetDevice(%d) failed", device_ordinal);
        return MCULL_E_HIP;
    }
    ncclUniqueId id;
    std::memcpy(&id, id_128_bytes, sizeof id);
    ncclComm_t c = nullptr;
    const ncclResult_t r = api->CommInitRank(&c, world, id, rank);
    if (r != ncclSuccess) return rccl_fail(api, "ncclCommInitRank", r);
    *comm = c;
    return MCULL_OK;
}

int mcull_rccl_comm_destroy(void *comm) {
    if (!comm) return MCULL_OK;
    const RcclApi *api = rccl();
    if (!api) return MCULL_E_STATE;
    const ncclResult_t r = api->CommDestroy(reinterpret_cast<ncclComm_t>(comm));
    return r == ncclSuccess ? MCULL_OK : rccl_fail(api, "ncclCommDestroy", r);
}

int mcull_allgather_commands(void *comm, void *stream, const void *send_block, void *gathered, uint32_t world,
                             uint64_t rows_per_rank, uint64_t *totals, uint64_t *bases) {
    if (!comm || !send_block || !gathered || !totals || !bases || world == 0 || world > 4096) return MCULL_E_INVALID;
    const RcclApi *api = rccl();
    if (!api) return MCULL_E_STATE;
    if (api->CommCount) {
        int n = 0;
        if (api->CommCount(reinterpret_cast<ncclComm_t>(comm), &n) == ncclSuccess && (uint32_t)n != world) {
            snprintf(t_error, sizeof t_error, "the communicator has %d ranks, world is %u", n, world);
            return MCULL_E_INVALID;
        }
    }
    // rows_per_rank {taskCount, firstTask} records and one record that is the rank's visible count: 8 bytes each
    const size_t words = (size_t)(rows_per_rank + 1) * 2;
    const ncclResult_t r = api->AllGather(send_block, gathered, words, ncclUint32, reinterpret_cast<ncclComm_t>(comm), (hipStream_t)stream);
    if (r != ncclSuccess) return rccl_fail(api, "ncclAllGather", r);
    if (mcull::launch_rebase_gathered(gathered, world, rows_per_rank, totals, bases, (hipStream_t)stream) != hipSuccess) {
        snprintf(t_error, sizeof t_error, "the rebase launch failed");
        return MCULL_E_HIP;
    }
    return MCULL_OK;
}

}  // extern "C"
