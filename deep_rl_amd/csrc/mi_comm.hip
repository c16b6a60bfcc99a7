// mi_comm.hip — the only cross-GPU exchange of the path, straight on RCCL (rccl.h): one communicator per process
// (one process per GPU), SUM all-reduces of two tiny buffers enqueued IN-STREAM between the kernels of mi_ppo_update_sharded, so a
// whole sharded outer update is ONE C call with no Python between launches.
//
// RCCL is bound at run time (dlopen), not at link time: libmirl.so stays loadable on a CPU-only box and single-GPU runs never touch
// it.  The instance already living in the process (torch's bundled librccl.so) is preferred, so both talk to the same transport.
#include <dlfcn.h>
#include <rccl/rccl.h>   // types / enums / prototypes only

#include "mi_common.h"

struct rccl_api_t {
    void* so;
    decltype(&ncclGetUniqueId) GetUniqueId;
    decltype(&ncclCommInitRank) CommInitRank;
    decltype(&ncclCommDestroy) CommDestroy;
    decltype(&ncclAllReduce) AllReduce;
    decltype(&ncclGetErrorString) GetErrorString;
    decltype(&ncclGetVersion) GetVersion;
    decltype(&ncclCommCount) CommCount;
};
static rccl_api_t g_rccl = {};

static int rccl_bind() {
    if (g_rccl.so) return MI_OK;
    const char* names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so.1"};
    void* so = dlopen(names[0], RTLD_NOW | RTLD_NOLOAD);          // the instance torch already mapped, if any
    if (!so) so = dlopen(names[1], RTLD_NOW | RTLD_NOLOAD);
    for (int i = 0; i < 3 && !so; ++i) so = dlopen(names[i], RTLD_NOW | RTLD_LOCAL);
    if (!so) { mi_set_error("mi_comm: cannot load RCCL (librccl.so): %s", dlerror()); return MI_ESTATE; }
    rccl_api_t a = {};
    a.so = so;
    a.GetUniqueId = (decltype(a.GetUniqueId))dlsym(so, "ncclGetUniqueId");
    a.CommInitRank = (decltype(a.CommInitRank))dlsym(so, "ncclCommInitRank");
    a.CommDestroy = (decltype(a.CommDestroy))dlsym(so, "ncclCommDestroy");
    a.AllReduce = (decltype(a.AllReduce))dlsym(so, "ncclAllReduce");
    a.GetErrorString = (decltype(a.GetErrorString))dlsym(so, "ncclGetErrorString");
    a.GetVersion = (decltype(a.GetVersion))dlsym(so, "ncclGetVersion");
    a.CommCount = (decltype(a.CommCount))dlsym(so, "ncclCommCount");
    if (!a.GetUniqueId || !a.CommInitRank || !a.CommDestroy || !a.AllReduce || !a.GetErrorString) {
        mi_set_error("mi_comm: librccl.so lacks an ncclGetUniqueId / ncclCommInitRank / ncclCommDestroy / ncclAllReduce entry point");
        return MI_ESTATE;
    }
    g_rccl = a;
    return MI_OK;
}

#define MI_RCCL(call)                                                                              \
    do {                                                                                           \
        ncclResult_t r_ = (call);                                                                  \
        if (r_ != ncclSuccess) {                                                                   \
            mi_set_error("%s: %s failed: %s", __func__, #call, g_rccl.GetErrorString(r_));         \
            return MI_EHIP;                                                                        \
        }                                                                                          \
    } while (0)

struct mi_comm { ncclComm_t comm; int world, rank, device; };

extern "C" int mi_comm_unique_id(void* id128) {
    MI_CHECK_ARG(id128 != nullptr, "id128 is NULL");
    static_assert(sizeof(ncclUniqueId) == MI_COMM_ID_BYTES, "ncclUniqueId size");
    int rc = rccl_bind();
    if (rc) return rc;
    MI_RCCL(g_rccl.GetUniqueId((ncclUniqueId*)id128));
    return MI_OK;
}

extern "C" int mi_comm_create(const void* id128, int world_size, int rank, void** out) {
    MI_CHECK_ARG(id128 && out, "NULL pointer");
    MI_CHECK_ARG(world_size >= 1 && rank >= 0 && rank < world_size, "rank / world_size out of range");
    int rc = rccl_bind();
    if (rc) return rc;
    mi_comm* c = (mi_comm*)calloc(1, sizeof(mi_comm));
    if (!c) { mi_set_error("mi_comm_create: out of host memory"); return MI_ENOMEM; }
    c->world = world_size; c->rank = rank;
    MI_HIP(hipGetDevice(&c->device));
    ncclUniqueId id;
    memcpy(&id, id128, sizeof(id));
    ncclResult_t r = g_rccl.CommInitRank(&c->comm, world_size, id, rank);   // collective over all ranks; binds the current device
    if (r != ncclSuccess) { mi_set_error("mi_comm_create: ncclCommInitRank failed: %s", g_rccl.GetErrorString(r)); free(c); return MI_EHIP; }
    *out = c;
    return MI_OK;
}

extern "C" int mi_comm_destroy(void* comm) {
    if (!comm) return MI_OK;
    mi_comm* c = (mi_comm*)comm;
    if (g_rccl.so && c->comm) (void)g_rccl.CommDestroy(c->comm);
    free(c);
    return MI_OK;
}

extern "C" int mi_comm_info(void* comm, int* world_size, int* rank, int* rccl_version, int* comm_count) {
    MI_CHECK_ARG(comm != nullptr, "comm is NULL");
    mi_comm* c = (mi_comm*)comm;
    if (world_size) *world_size = c->world;
    if (rank) *rank = c->rank;
    if (rccl_version) { int v = 0; if (g_rccl.GetVersion) (void)g_rccl.GetVersion(&v); *rccl_version = v; }
    if (comm_count) { int n = -1; if (g_rccl.CommCount && g_rccl.CommCount(c->comm, &n) != ncclSuccess) n = -1; *comm_count = n; }
    return MI_OK;
}

// in-place SUM all-reduce of n f32 (dtype 0) or f64 (dtype 1) elements, enqueued on `stream`
int mi_comm_allreduce_impl(void* comm, void* buf, size_t n, int dtype, hipStream_t s) {
    mi_comm* c = (mi_comm*)comm;
    MI_RCCL(g_rccl.AllReduce(buf, buf, n, dtype ? ncclDouble : ncclFloat, ncclSum, c->comm, s));
    return MI_OK;
}

extern "C" int mi_comm_allreduce_sum(void* comm, void* buf, size_t n, int dtype, void* stream) {
    MI_CHECK_ARG(comm && buf && n > 0 && (dtype == 0 || dtype == 1), "bad arguments");
    return mi_comm_allreduce_impl(comm, buf, n, dtype, (hipStream_t)stream);
}
