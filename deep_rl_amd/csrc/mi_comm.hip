// mi_comm.hip — the only cross-GPU exchange of the path: SUM all-reduces of tiny buffers (<= 36.6 KB for PPO, <= 539 KB for SAC's twin critics)
// enqueued IN-STREAM between the kernels of mi_ppo_update_sharded & co., so a whole sharded outer update is ONE C call with no Python between launches.
// Two carriers behind one handle (reference ppo.py:189-192, dqn.py:131-133, sac.py:185-210 with the gradient exchange between backward and the optimizer step):
//   RCCL  (mi_comm_create; default)       straight on rccl.h, one communicator per process (one process per GPU).
//   P2P   (mi_comm_p2p_alloc / _connect)  a one-shot exchange over hipIpc-mapped inboxes: every rank stores its share into its slot of EVERY rank's inbox, publishes a
//         sequence number, waits for the world's sequence numbers in its own inbox and sums the slots IN RANK ORDER — one launch per all-reduce, no ring, and the
//         same bits on every rank by construction (a ring's grouping depends on the rank).  Also the only carrier that takes two ranks on ONE device (RCCL refuses).
//
// RCCL is bound at run time (dlopen), not at link time: libmirl.so stays loadable on a CPU-only box and single-GPU runs never touch
// it.  The instance already living in the process (torch's bundled librccl.so) is preferred, so both talk to the same transport.
#include <dlfcn.h>
#include <stdlib.h>
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

// ---- the P2P carrier's inbox (device memory of its owner, mapped into every peer with hipIpcOpenMemHandle) ------------------------------------------------------
//   [flags u32 [2 parities][P2P_MAX_WORLD][P2P_MAX_GROUPS]] [status u32 ... padded to P2P_HDR_BYTES] [data [2 parities][world][cap bytes]]
// flag (parity, r, g) = sequence number of the last all-reduce whose workgroup g of rank r has stored its chunk into slot (parity, r) of THIS inbox.
#define P2P_MAX_WORLD 8
#define P2P_MAX_GROUPS 64
#define P2P_THREADS 256
#define P2P_FLAG_WORDS (2 * P2P_MAX_WORLD * P2P_MAX_GROUPS)
#define P2P_HDR_BYTES (P2P_FLAG_WORDS * 4 + 256)
enum { CARRIER_RCCL = 0, CARRIER_P2P = 1 };

struct mi_comm {
    int carrier;
    ncclComm_t comm;
    int world, rank, device;
    // P2P
    char* inbox;                   // own inbox
    char* peer[P2P_MAX_WORLD];     // every rank's inbox as mapped here (peer[rank] == inbox; synthetic: all == inbox)
    bool opened[P2P_MAX_WORLD];    // mapped with hipIpcOpenMemHandle (to be closed)
    size_t cap;                    // bytes per slot
    uint32_t seq;                  // sequence number of the last enqueued all-reduce (host side; every rank counts the same calls)
    int synthetic;                 // one process plays `world` ranks into its own inbox (slot 0 = its share, the others zeros): timing only
    int mem_kind;                  // 0 uncached, 1 fine-grained, 2 plain device memory
    unsigned long long budget;     // wait budget in 100 MHz ticks
    int connected;
};

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

// ================================================================ the P2P carrier ================================================================
struct p2p_args_t {
    char* peer[P2P_MAX_WORLD];
    unsigned long long budget;
    size_t cap;
    uint32_t seq;
    int world, rank, synthetic;
};
__device__ __forceinline__ uint32_t* p2p_flag(char* box, int parity, int r, int g) { return reinterpret_cast<uint32_t*>(box) + (parity * P2P_MAX_WORLD + r) * P2P_MAX_GROUPS + g; }
__device__ __forceinline__ uint32_t* p2p_status(char* box) { return reinterpret_cast<uint32_t*>(box) + P2P_FLAG_WORDS; }
__device__ __forceinline__ char* p2p_slot(char* box, int parity, int r, int world, size_t cap) { return box + P2P_HDR_BYTES + ((size_t)parity * world + r) * cap; }
__device__ __forceinline__ unsigned long long p2p_clock() {
    unsigned long long t;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
    return t;
}

// One launch = one all-reduce.  Workgroup g owns elements [g * per, (g + 1) * per): it stores them into slot (parity, rank) of every rank's inbox, publishes the
// sequence number (one flag per peer and workgroup), waits for the world's flags of ITS chunk in its own inbox and writes the rank-ordered sum back to buf.  A
// workgroup never waits for anything but peers' stores of the same chunk, which do not depend on any wait: no deadlock whatever the residency (two ranks on one
// device included).  Two parities: rank A reaches all-reduce k + 2 only after B has published k + 1, i.e. after B's launch k — the reader of parity k & 1 — is over
// (launches of one stream run in order; all collectives of a communicator must be enqueued on streams ordered with each other).
// The wait is bounded (budget, default 10 s): a peer that never arrives sets the status word, the launch leaves buf = the local share, every later launch on this
// communicator returns at once, and mi_comm_check / the next enqueue's host-side check report MI_ESTATE.
template <typename T, int V>
__global__ void __launch_bounds__(P2P_THREADS) p2p_allreduce_kernel(p2p_args_t a, T* __restrict__ buf, size_t n, size_t per) {
    typedef T vec_t __attribute__((ext_vector_type(V)));
    __shared__ int s_bad;
    const int tid = threadIdx.x, g = blockIdx.x, parity = a.seq & 1;
    char* const mine = a.peer[a.synthetic ? 0 : a.rank];
    if (tid == 0) s_bad = __hip_atomic_load(p2p_status(mine), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
    __syncthreads();
    if (s_bad) return;
    const size_t lo = (size_t)g * per, hi = lo + per < n ? lo + per : n;   // per is a multiple of V, buf is V-aligned
    // 1. publish
    for (int d = 0; d < a.world; ++d) {
        const int p = a.rank + d < a.world ? a.rank + d : a.rank + d - a.world;   // start with the own inbox, then rank + 1 ...: the peers' links are used side by side
        const bool zeros = a.synthetic && p != 0;
        T* dst = reinterpret_cast<T*>(p2p_slot(a.peer[p], parity, a.synthetic ? p : a.rank, a.world, a.cap));
        for (size_t i = lo + (size_t)tid * V; i < hi; i += (size_t)P2P_THREADS * V) {
            if (i + V <= hi) {
                vec_t v = *reinterpret_cast<const vec_t*>(buf + i);
                if (zeros) v = vec_t(0);
                *reinterpret_cast<vec_t*>(dst + i) = v;
            } else {
                for (size_t k = i; k < hi; ++k) dst[k] = zeros ? T(0) : buf[k];
            }
        }
    }
    __threadfence_system();   // this thread's stores are visible system-wide before ...
    __syncthreads();          // ... any flag of this workgroup is
    if (tid < a.world) __hip_atomic_store(p2p_flag(a.peer[tid], parity, a.synthetic ? tid : a.rank, g), a.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    // 2. wait for the world's chunk g
    if (tid < a.world) {
        const uint32_t* f = p2p_flag(mine, parity, tid, g);
        const unsigned long long t0 = p2p_clock();
        uint32_t spins = 0;
        while (__hip_atomic_load(f, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) != a.seq) {
            __builtin_amdgcn_s_sleep(1);
            if ((++spins & 31) == 0 && (p2p_clock() - t0 > a.budget || __hip_atomic_load(p2p_status(mine), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) {
                __hip_atomic_fetch_or(p2p_status(mine), 1u | (1u << (8 + tid)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // bit 8 + r: rank r never arrived
                s_bad = 1;
                break;
            }
        }
    }
    __syncthreads();
    if (s_bad) return;
    __threadfence_system();   // acquire for every thread's loads below
    // 3. the sum, in rank order on every rank
    for (size_t i = lo + (size_t)tid * V; i < hi; i += (size_t)P2P_THREADS * V) {
        if (i + V <= hi) {
            vec_t acc = __builtin_nontemporal_load(reinterpret_cast<const vec_t*>(reinterpret_cast<const T*>(p2p_slot(mine, parity, 0, a.world, a.cap)) + i));
            for (int r = 1; r < a.world; ++r)
                acc += __builtin_nontemporal_load(reinterpret_cast<const vec_t*>(reinterpret_cast<const T*>(p2p_slot(mine, parity, r, a.world, a.cap)) + i));
            *reinterpret_cast<vec_t*>(buf + i) = acc;
        } else {
            for (size_t k = i; k < hi; ++k) {
                T acc = __builtin_nontemporal_load(reinterpret_cast<const T*>(p2p_slot(mine, parity, 0, a.world, a.cap)) + k);
                for (int r = 1; r < a.world; ++r) acc += __builtin_nontemporal_load(reinterpret_cast<const T*>(p2p_slot(mine, parity, r, a.world, a.cap)) + k);
                buf[k] = acc;
            }
        }
    }
}

static int p2p_new(int world, int rank, size_t max_bytes, int synthetic, mi_comm** out) {
    mi_comm* c = (mi_comm*)calloc(1, sizeof(mi_comm));
    if (!c) { mi_set_error("mi_comm_p2p: out of host memory"); return MI_ENOMEM; }
    c->carrier = CARRIER_P2P; c->world = world; c->rank = rank; c->synthetic = synthetic;
    c->cap = (max_bytes + 255) & ~(size_t)255;
    unsigned long long ms = 10000;
    if (const char* e = getenv("MIRL_P2P_TIMEOUT_MS")) { const long long v = atoll(e); if (v > 0) ms = (unsigned long long)v; }
    c->budget = ms * 100000ull;   // s_memrealtime: 100 MHz
    hipError_t e = hipGetDevice(&c->device);
    if (e != hipSuccess) { mi_set_error("mi_comm_p2p: hipGetDevice failed: %s", hipGetErrorString(e)); free(c); return MI_EHIP; }
    const size_t bytes = P2P_HDR_BYTES + 2 * (size_t)world * c->cap;
    // uncached (or at least fine-grained) device memory: a peer's stores over xGMI must not meet stale lines in the owner's L2 while the owner polls inside a kernel
    const unsigned kinds[3] = {hipDeviceMallocUncached, hipDeviceMallocFinegrained, hipDeviceMallocDefault};
    const char* want = getenv("MIRL_P2P_MEM");   // "uncached" | "finegrained" | "plain": start of the fallback chain (diagnostic)
    int first = 0;
    if (want && !strcmp(want, "finegrained")) first = 1;
    if (want && !strcmp(want, "plain")) first = 2;
    void* box = nullptr;
    for (int k = first; k < 3 && !box; ++k) {
        e = hipExtMallocWithFlags(&box, bytes, kinds[k]);
        if (e != hipSuccess) { box = nullptr; (void)hipGetLastError(); continue; }
        if (!synthetic) {   // must be exportable
            hipIpcMemHandle_t h;
            if (hipIpcGetMemHandle(&h, box) != hipSuccess) { (void)hipGetLastError(); (void)hipFree(box); box = nullptr; continue; }
        }
        c->mem_kind = k;
    }
    if (!box) { mi_set_error("mi_comm_p2p: cannot allocate an exportable %zu-byte inbox: %s", bytes, hipGetErrorString(e)); free(c); return MI_ENOMEM; }
    if (hipMemset(box, 0, bytes) != hipSuccess || hipDeviceSynchronize() != hipSuccess) {
        mi_set_error("mi_comm_p2p: clearing the inbox failed"); (void)hipFree(box); free(c); return MI_EHIP;
    }
    c->inbox = (char*)box;
    c->peer[rank] = c->inbox;
    if (synthetic) { for (int r = 0; r < world; ++r) c->peer[r] = c->inbox; c->connected = 1; }
    *out = c;
    return MI_OK;
}

extern "C" int mi_comm_p2p_alloc(int world_size, int rank, size_t max_bytes, void** out, void* ipc_handle64) {
    MI_CHECK_ARG(out && ipc_handle64, "NULL pointer");
    MI_CHECK_ARG(world_size >= 1 && world_size <= P2P_MAX_WORLD && rank >= 0 && rank < world_size, "rank / world_size out of range (world_size <= 8)");
    MI_CHECK_ARG(max_bytes > 0, "max_bytes must be positive");
    static_assert(sizeof(hipIpcMemHandle_t) == MI_COMM_IPC_BYTES, "hipIpcMemHandle_t size");
    mi_comm* c = nullptr;
    int rc = p2p_new(world_size, rank, max_bytes, 0, &c);
    if (rc) return rc;
    hipIpcMemHandle_t h;
    hipError_t e = hipIpcGetMemHandle(&h, c->inbox);
    if (e != hipSuccess) { mi_set_error("mi_comm_p2p_alloc: hipIpcGetMemHandle failed: %s", hipGetErrorString(e)); (void)hipFree(c->inbox); free(c); return MI_EHIP; }
    memcpy(ipc_handle64, &h, sizeof(h));
    *out = c;
    return MI_OK;
}

extern "C" int mi_comm_p2p_connect(void* comm, const void* handles) {
    MI_CHECK_ARG(comm && handles, "NULL pointer");
    mi_comm* c = (mi_comm*)comm;
    MI_CHECK_ARG(c->carrier == CARRIER_P2P && !c->synthetic, "not an unconnected P2P communicator");
    MI_CHECK_ARG(!c->connected, "already connected");
    for (int r = 0; r < c->world; ++r) {
        if (r == c->rank) continue;
        hipIpcMemHandle_t h;
        memcpy(&h, (const char*)handles + (size_t)r * MI_COMM_IPC_BYTES, sizeof(h));
        void* p = nullptr;
        hipError_t e = hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess);
        if (e != hipSuccess) {
            mi_set_error("mi_comm_p2p_connect: hipIpcOpenMemHandle of rank %d's inbox failed on rank %d: %s", r, c->rank, hipGetErrorString(e));
            (void)hipGetLastError();
            return MI_EHIP;
        }
        c->peer[r] = (char*)p; c->opened[r] = true;
    }
    c->connected = 1;
    return MI_OK;
}

extern "C" int mi_comm_p2p_synthetic(int world_size, size_t max_bytes, void** out) {
    MI_CHECK_ARG(out != nullptr, "NULL pointer");
    MI_CHECK_ARG(world_size >= 1 && world_size <= P2P_MAX_WORLD && max_bytes > 0, "world_size must be 1..8, max_bytes positive");
    mi_comm* c = nullptr;
    int rc = p2p_new(world_size, 0, max_bytes, 1, &c);
    if (rc) return rc;
    *out = c;
    return MI_OK;
}

static int p2p_allreduce(mi_comm* c, void* buf, size_t n, int dtype, hipStream_t s) {
    const size_t esz = dtype ? 8 : 4;
    if (!c->connected) { mi_set_error("mi_comm (p2p): all-reduce before mi_comm_p2p_connect"); return MI_ESTATE; }
    if (n * esz > c->cap) { mi_set_error("mi_comm (p2p): a %zu-byte message does not fit the %zu-byte slots (max_bytes of mi_comm_p2p_alloc)", n * esz, c->cap); return MI_EINVAL; }
    const bool vec = ((uintptr_t)buf & 15) == 0;
    const size_t V = vec ? 16 / esz : 1;
    // >= 1024 elements per workgroup (one 16-byte vector per thread and pass), at most P2P_MAX_GROUPS workgroups
    size_t groups = (n + 1023) / 1024;
    if (groups > P2P_MAX_GROUPS) groups = P2P_MAX_GROUPS;
    size_t per = (n + groups - 1) / groups;
    per = (per + V - 1) / V * V;
    groups = (n + per - 1) / per;
    p2p_args_t a;
    for (int r = 0; r < P2P_MAX_WORLD; ++r) a.peer[r] = r < c->world ? c->peer[r] : nullptr;
    a.budget = c->budget; a.cap = c->cap; a.world = c->world; a.rank = c->rank; a.synthetic = c->synthetic;
    if (++c->seq == 0) c->seq = 1;   // 0 is the cleared inbox
    a.seq = c->seq;
    if (dtype) {
        if (vec) p2p_allreduce_kernel<double, 2><<<(unsigned)groups, P2P_THREADS, 0, s>>>(a, (double*)buf, n, per);
        else p2p_allreduce_kernel<double, 1><<<(unsigned)groups, P2P_THREADS, 0, s>>>(a, (double*)buf, n, per);
    } else {
        if (vec) p2p_allreduce_kernel<float, 4><<<(unsigned)groups, P2P_THREADS, 0, s>>>(a, (float*)buf, n, per);
        else p2p_allreduce_kernel<float, 1><<<(unsigned)groups, P2P_THREADS, 0, s>>>(a, (float*)buf, n, per);
    }
    MI_LAUNCH_CHECK();
    return MI_OK;
}

// Host-synchronising: MI_OK, or MI_ESTATE when a wait of the P2P carrier ran out (mi_last_error names the ranks that never arrived).  RCCL: always MI_OK.
extern "C" int mi_comm_check(void* comm) {
    MI_CHECK_ARG(comm != nullptr, "comm is NULL");
    mi_comm* c = (mi_comm*)comm;
    if (c->carrier != CARRIER_P2P) return MI_OK;
    uint32_t st = 0;
    MI_HIP(hipMemcpy(&st, c->inbox + (size_t)P2P_FLAG_WORDS * 4, 4, hipMemcpyDeviceToHost));
    if (st) {
        char who[64]; int k = 0;
        for (int r = 0; r < c->world; ++r) if (st & (1u << (8 + r))) k += snprintf(who + k, sizeof(who) - k, " %d", r);
        mi_set_error("mi_comm (p2p), rank %d: a wait ran out (MIRL_P2P_TIMEOUT_MS); ranks that never arrived:%s — the buffers of that and every later all-reduce hold "
                     "the LOCAL share only", c->rank, k ? who : " ?");
        return MI_ESTATE;
    }
    return MI_OK;
}

extern "C" int mi_comm_carrier(void* comm) { return comm ? ((mi_comm*)comm)->carrier : MI_EINVAL; }

extern "C" int mi_comm_destroy(void* comm) {
    if (!comm) return MI_OK;
    mi_comm* c = (mi_comm*)comm;
    if (c->carrier == CARRIER_P2P) {   // callers put a barrier in front: a peer may still be storing into this inbox
        (void)hipDeviceSynchronize();
        for (int r = 0; r < c->world; ++r) if (c->opened[r]) (void)hipIpcCloseMemHandle(c->peer[r]);
        if (c->inbox) (void)hipFree(c->inbox);
    } else if (g_rccl.so && c->comm) {
        (void)g_rccl.CommDestroy(c->comm);
    }
    free(c);
    return MI_OK;
}

extern "C" int mi_comm_info(void* comm, int* world_size, int* rank, int* rccl_version, int* comm_count) {
    MI_CHECK_ARG(comm != nullptr, "comm is NULL");
    mi_comm* c = (mi_comm*)comm;
    if (world_size) *world_size = c->world;
    if (rank) *rank = c->rank;
    if (c->carrier == CARRIER_P2P) {   // no RCCL behind it: version 0, count = the inboxes mapped
        if (rccl_version) *rccl_version = 0;
        if (comm_count) *comm_count = c->connected ? c->world : 0;
        return MI_OK;
    }
    if (rccl_version) { int v = 0; if (g_rccl.GetVersion) (void)g_rccl.GetVersion(&v); *rccl_version = v; }
    if (comm_count) { int n = -1; if (g_rccl.CommCount && g_rccl.CommCount(c->comm, &n) != ncclSuccess) n = -1; *comm_count = n; }
    return MI_OK;
}

// in-place SUM all-reduce of n f32 (dtype 0) or f64 (dtype 1) elements, enqueued on `stream`
int mi_comm_allreduce_impl(void* comm, void* buf, size_t n, int dtype, hipStream_t s) {
    mi_comm* c = (mi_comm*)comm;
    if (c->carrier == CARRIER_P2P) return p2p_allreduce(c, buf, n, dtype, s);
    MI_RCCL(g_rccl.AllReduce(buf, buf, n, dtype ? ncclDouble : ncclFloat, ncclSum, c->comm, s));
    return MI_OK;
}

extern "C" int mi_comm_allreduce_sum(void* comm, void* buf, size_t n, int dtype, void* stream) {
    MI_CHECK_ARG(comm && buf && n > 0 && (dtype == 0 || dtype == 1), "bad arguments");
    return mi_comm_allreduce_impl(comm, buf, n, dtype, (hipStream_t)stream);
}
