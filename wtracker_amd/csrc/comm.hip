// The multi-GPU path's only collective for callers of the C ABI that do not go through PyTorch (SURVEY.md §8b export list, §8e):
// an RCCL communicator of its own + the all-gather of per-rank [n_local][4] track slices.  librccl is bound at run time (dlopen on
// the first wtk_comm_* call) so that single-GPU users of libwtk_hip.so do not load it.  The Python pipeline
// (wtracker_amd/pipeline.py) uses torch.distributed's RCCL communicator instead; both issue the same ncclAllGather.
#include "../../include/wtk_hip.h"

#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstring>
#include <string>

extern int wtk_set_error(const std::string &msg); // wtk_api.hip

namespace {
struct Rccl {
    void *lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
};
Rccl g_rccl;

int load_rccl() {
    if (g_rccl.lib) return 0;
    void *lib = nullptr;
    for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
        lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
        if (lib) break;
    }
    if (!lib) return wtk_set_error(std::string("wtk_comm: cannot load librccl: ") + dlerror());
    Rccl r;
    r.lib = lib;
    r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(dlsym(lib, "ncclGetUniqueId"));
    r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(dlsym(lib, "ncclCommInitRank"));
    r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(dlsym(lib, "ncclCommDestroy"));
    r.AllGather = reinterpret_cast<decltype(r.AllGather)>(dlsym(lib, "ncclAllGather"));
    r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(dlsym(lib, "ncclGetErrorString"));
    if (!r.GetUniqueId || !r.CommInitRank || !r.CommDestroy || !r.AllGather || !r.GetErrorString) return wtk_set_error("wtk_comm: librccl lacks an expected symbol");
    g_rccl = r;
    return 0;
}
int fail_nccl(const char *what, ncclResult_t e) { return wtk_set_error(std::string(what) + ": " + g_rccl.GetErrorString(e)); }
} // namespace

struct wtk_comm {
    ncclComm_t comm = nullptr;
    int device = 0, rank = 0, world = 1;
};

extern "C" int wtk_comm_unique_id(uint8_t *id_out, size_t cap) {
    if (!id_out || cap < WTK_COMM_ID_BYTES) return wtk_set_error("wtk_comm_unique_id: id buffer must hold WTK_COMM_ID_BYTES (128) bytes");
    static_assert(WTK_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "id size");
    if (load_rccl()) return 1;
    ncclUniqueId id;
    const ncclResult_t e = g_rccl.GetUniqueId(&id);
    if (e != ncclSuccess) return fail_nccl("ncclGetUniqueId", e);
    std::memcpy(id_out, id.internal, NCCL_UNIQUE_ID_BYTES);
    return 0;
}

extern "C" int wtk_comm_create(wtk_comm **out, int32_t device, int32_t rank, int32_t world, const uint8_t *id) {
    if (!out || !id) return wtk_set_error("wtk_comm_create: null argument");
    if (world < 1 || rank < 0 || rank >= world) return wtk_set_error("wtk_comm_create: need 0 <= rank < world");
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || device < 0 || device >= n) {
        (void)hipGetLastError();
        return wtk_set_error("wtk_comm_create: no such HIP device (is a GPU visible?)");
    }
    if (load_rccl()) return 1;
    int prev = -1;
    const bool have_prev = hipGetDevice(&prev) == hipSuccess;
    hipError_t he = hipSetDevice(device);
    if (he != hipSuccess) return wtk_set_error(std::string("wtk_comm_create: hipSetDevice: ") + hipGetErrorString(he));
    struct Restore { // the caller's current device is left as it was found (PyTorch tracks the same thread-local state)
        int prev; bool on;
        ~Restore() { if (on) (void)hipSetDevice(prev); }
    } restore{prev, have_prev && prev != device};
    ncclUniqueId uid;
    std::memcpy(uid.internal, id, NCCL_UNIQUE_ID_BYTES);
    wtk_comm *c = new wtk_comm();
    c->device = device, c->rank = rank, c->world = world;
    const ncclResult_t e = g_rccl.CommInitRank(&c->comm, world, uid, rank);
    if (e != ncclSuccess) {
        delete c;
        return fail_nccl("ncclCommInitRank", e);
    }
    *out = c;
    return 0;
}

extern "C" void wtk_comm_destroy(wtk_comm *c) {
    if (!c) return;
    if (c->comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(c->comm);
    delete c;
}

extern "C" int wtk_allgather_tracks(wtk_comm *c, const float *local_dev, int32_t n_local, float *all_dev, void *stream) {
    if (!c || !local_dev || !all_dev) return wtk_set_error("wtk_allgather_tracks: null argument");
    if (n_local < 0) return wtk_set_error("wtk_allgather_tracks: negative row count");
    if (n_local == 0) return 0;
    int prev = -1;
    if (hipGetDevice(&prev) == hipSuccess && prev != c->device) (void)hipSetDevice(c->device);
    // rank r's n_local rows land at all[r * n_local ...]: rank order == frame order of the interleaved super-batch (pipeline.py)
    const ncclResult_t e = g_rccl.AllGather(local_dev, all_dev, (size_t)n_local * 4, ncclFloat, c->comm, (hipStream_t)stream);
    if (prev >= 0 && prev != c->device) (void)hipSetDevice(prev);
    if (e != ncclSuccess) return fail_nccl("ncclAllGather", e);
    return 0;
}
