// comm.hip.h — the ONE collective of the path: an RCCL all-gather of every rank's packed
// [nq_local x k] (id, distance) top-k over xGMI, issued on the context's stream right behind Refine.
//
// The reference has no collective at all (single JVM, FSA:636 serial loop); queries shard embarrassingly over GPUs
// with the index replicated, so this merge is the only exchange.  The message is tiny (k = 10, nq_local = 1024:
// 120 KB per rank) and therefore latency-bound: one ncclAllGather, no ring all-reduce, no second exchange.
//
// librccl is bound at run time (dlopen) so that single-GPU deployments do not need it and so that a process that
// already carries an RCCL (e.g. PyTorch's) shares that instance instead of loading a second one:
//   $FSPANN_RCCL_LIB, else an already-loaded librccl.so / librccl.so.1, else librccl.so.1 from the loader path / /opt/rocm/lib.
// The communicator is bootstrapped by the caller's own transport: rank 0 asks for the 128-byte unique id
// (fspann_comm_unique_id) and hands it to the other ranks however it likes (JVM: its RPC; Python: torch.distributed).
#pragma once
#include <dlfcn.h>

#include <mutex>

#include "fspann_common.h"

namespace fspann {

struct RcclApi {
    // the five entry points used, with the ABI of rccl.h (ncclUniqueId is a 128-byte struct passed BY VALUE)
    struct UniqueId { char internal[128]; };
    int (*GetUniqueId)(UniqueId*) = nullptr;
    int (*CommInitRank)(void**, int, UniqueId, int) = nullptr;
    int (*AllGather)(const void*, void*, size_t, int, void*, hipStream_t) = nullptr;
    int (*CommDestroy)(void*) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    void* handle = nullptr;
    std::string path;
};

inline RcclApi* rccl_api() {
    static RcclApi api;
    static std::once_flag once;
    std::call_once(once, [] {
        const char* env = getenv("FSPANN_RCCL_LIB");
        struct Try { const char* name; int flags; };
        const Try tries[] = {
            {env, RTLD_NOW | RTLD_LOCAL},
            {"librccl.so", RTLD_NOW | RTLD_NOLOAD},       // an instance the process already carries (PyTorch's)
            {"librccl.so.1", RTLD_NOW | RTLD_NOLOAD},
            {"librccl.so.1", RTLD_NOW | RTLD_LOCAL},
            {"/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_LOCAL},
            {"librccl.so", RTLD_NOW | RTLD_LOCAL},
        };
        for (const Try& t : tries) {
            if (!t.name || !*t.name) continue;
            void* h = dlopen(t.name, t.flags);
            if (!h) continue;
            RcclApi a;
            a.handle = h;
            a.path = t.name;
            a.GetUniqueId = reinterpret_cast<decltype(a.GetUniqueId)>(dlsym(h, "ncclGetUniqueId"));
            a.CommInitRank = reinterpret_cast<decltype(a.CommInitRank)>(dlsym(h, "ncclCommInitRank"));
            a.AllGather = reinterpret_cast<decltype(a.AllGather)>(dlsym(h, "ncclAllGather"));
            a.CommDestroy = reinterpret_cast<decltype(a.CommDestroy)>(dlsym(h, "ncclCommDestroy"));
            a.GetErrorString = reinterpret_cast<decltype(a.GetErrorString)>(dlsym(h, "ncclGetErrorString"));
            if (a.GetUniqueId && a.CommInitRank && a.AllGather && a.CommDestroy) { api = a; return; }
            dlclose(h);
        }
    });
    return api.handle ? &api : nullptr;
}

inline const char* rccl_err(RcclApi* a, int rc) { return (a && a->GetErrorString) ? a->GetErrorString(rc) : "?"; }

}  // namespace fspann

struct fspann_comm {
    fspann_ctx* ctx = nullptr;
    void* nccl = nullptr;
    int world = 1, rank = 0;
};
