// mi_comm.hip — the only cross-GPU exchange of the path: SUM all-reduces of tiny buffers (<= 36.6 KB for PPO, <= 539 KB for SAC's twin critics)
// enqueued IN-STREAM between the kernels of mi_ppo_update_sharded & co., so a whole sharded outer update is ONE C call with no Python between launches.
// Two carriers behind one handle (reference ppo.py:189-192, dqn.py:131-133, sac.py:185-210 with the gradient exchange between backward and the optimizer step):
//   RCCL  (mi_comm_create; default)       straight on rccl.h, one communicator per process (one process per GPU).
//   P2P   (mi_comm_p2p_alloc / _connect)  a one-shot exchange over hipIpc-mapped inboxes: every rank stores its share — every 32-bit word together with the
//         all-reduce's sequence number in ONE 8-byte line — into its slot of EVERY rank's inbox, polls the world's lines in its own inbox and sums them IN RANK
//         ORDER: one launch per all-reduce, no ring, no fence, the same bits on every rank by construction (a ring's grouping depends on the rank).  Also the only
//         carrier that takes two ranks on ONE device (RCCL refuses).
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
//   [barrier lines u64 [2 parities][8 ranks] at byte 64, padded to P2P_HDR_BYTES] [lines u64 [2 parities][world][cap / 4]]
// (the status word — bit 0: a wait ran out, bit 8 + r: rank r never arrived — is NOT in the inbox: only this rank's own launches read it, so it lives in plain, cacheable
// device memory; in the uncached inbox its load cost every gated gradient launch 0.15 us)
// A LINE is one 8-byte word {payload word (low), sequence number (high)}, stored and loaded as ONE 8-byte access: the payload carries its own "arrived" flag, so an
// all-reduce needs no fence, no separate flag and no second round trip (the LL protocol idea).  Line i of slot (parity, r) = 32-bit word i of rank r's message.
// SEQUENCE NUMBERS AND PARITY.  The parity is a bit of its own that flips with every all-reduce (never derived from the sequence number); the sequence number counts
// 1 .. P2P_SEQ_LAST inside an EPOCH.  Messages have different lengths, so a slot keeps lines of OLDER all-reduces behind the end of a shorter one — harmless while
// sequence numbers never repeat, i.e. inside an epoch.  When the number would pass P2P_SEQ_LAST, every rank (they count the same calls) changes the epoch in-stream:
//   (1) clear the lines of the own inbox — the last all-reduce of the old epoch has completed here, so every store a peer aimed at this inbox in that epoch has landed;
//   (2) a barrier through the header's barrier lines (fixed length: one line per rank, its own counter, never cleared): a rank passes it only after EVERY rank has
//       cleared, so no store of the new epoch can meet a clear;  (3) go on with sequence number 1.
// (Round 5 derived the parity from the number and wrapped 0xFFFFFFFF -> 1: two consecutive all-reduces on one parity, and stale lines with a "right" number.)
#define P2P_MAX_GROUPS 64
#define P2P_THREADS 256
#define P2P_HDR_BYTES 256
#define P2P_BAR_OFF 64
#define P2P_SEQ_LAST 0xFFFFFFF0u
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
    size_t inbox_bytes;            // size of the inbox allocation (parked, never freed: see p2p_park)
    uint32_t seq;                  // sequence number of the last enqueued all-reduce inside the current epoch (host side; every rank counts the same calls)
    uint32_t parity;               // parity of the last enqueued all-reduce: flips with every one, whatever the sequence number does
    uint32_t bseq;                 // epoch changes so far (= the barrier lines' own sequence number)
    uint32_t* mirror;              // host-pinned, device-mapped copy of the status word (mi_comm_poll: no sync)
    char* status;                  // the status word: 64 bytes of PLAIN device memory (read by the waits of this rank's launches and by every gated optimizer step)
    int synthetic;                 // one process plays `world` ranks into its own inbox (slot 0 = its share, the others zeros): timing only
    int mem_kind;                  // 0 uncached, 1 fine-grained, 2 plain device memory
    unsigned long long budget;     // wait budget in 100 MHz ticks
    int connected;
    int colocated;                 // the LARGEST number of ranks of this communicator sharing one device (1 = one rank per GPU, the production placement)
    int fused_set;                 // mi_comm_p2p_set_fused: -1 / 0 (unset) / 1
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
// One launch = one all-reduce; a thread owns elements e = its global id, + the grid's threads, ...: it stores each of them as lines into slot (parity, rank) of EVERY
// rank's inbox (all stores of the launch go out before the first load), then polls the SAME lines of every rank's slot in its own inbox — all WORLD loads of an
// element in flight together — and writes the sum in RANK ORDER back to buf.  No barrier, no fence, no flag word: a thread waits for nothing but its peers' stores of its
// own elements, which depend on no wait — no deadlock whatever the residency (two ranks time-sharing one device included).  Two parities: rank A reaches all-reduce
// k + 2 only after B has stored k + 1, i.e. after B's launch k — the reader of parity k & 1 — is over (launches of one stream run in order; all collectives of a
// communicator must be enqueued on streams ordered with each other, in the same order on every rank).
// The wait is bounded (budget: MIRL_P2P_TIMEOUT_MS, default 30 s): a peer that never arrives sets the status word (and its host-pinned mirror), the elements concerned
// keep the LOCAL share, later launches give up after 64 polls — and nothing is lost but the update: every optimizer step behind the exchange reads the status word with
// its state and is withheld (mi_comm_gate), every later mi_*_sharded / mi_comm_allreduce_sum call returns MI_ESTATE at its entry (mi_comm_poll, no sync),
// mi_comm_check (synchronising) says which ranks were missing.
template <typename T, int WORLD>
__global__ void __launch_bounds__(P2P_THREADS) p2p_allreduce_kernel(p2p_args_t a, T* __restrict__ buf, size_t n) {
    constexpr int W = ll_elem<T>::W;
    const size_t gtid = (size_t)blockIdx.x * P2P_THREADS + threadIdx.x, nthreads = (size_t)gridDim.x * P2P_THREADS;
    // 1. publish
    for (size_t e = gtid; e < n; e += nthreads) {
        uint32_t w[W];
        ll_elem<T>::split(buf[e], w);
#pragma unroll
        for (int d = 0; d < WORLD; ++d)
#pragma unroll
            for (int k = 0; k < W; ++k) ll_store_nowait(a.dst[d] + e * W + k, (a.zeros >> d) & 1 ? 0u : w[k], a.seq);
    }
    // 2. poll the world's lines of the own elements, 3. sum in rank order
    for (size_t e = gtid; e < n; e += nthreads) {
        uint64_t v[WORLD][W];
        if (!ll_gather<WORLD, W>(a, e * W, v)) continue;   // the element keeps the local share
        T acc = ll_elem<T>::join(v[0]);
#pragma unroll
        for (int r = 1; r < WORLD; ++r) acc += ll_elem<T>::join(v[r]);
        buf[e] = acc;
    }
}

template <typename T>
static void p2p_launch(int world, unsigned groups, hipStream_t s, const p2p_args_t& a, T* buf, size_t n) {
    switch (world) {
        case 1: p2p_allreduce_kernel<T, 1><<<groups, P2P_THREADS, 0, s>>>(a, buf, n); break;
        case 2: p2p_allreduce_kernel<T, 2><<<groups, P2P_THREADS, 0, s>>>(a, buf, n); break;
        case 3: p2p_allreduce_kernel<T, 3><<<groups, P2P_THREADS, 0, s>>>(a, buf, n); break;
        case 4: p2p_allreduce_kernel<T, 4><<<groups, P2P_THREADS, 0, s>>>(a, buf, n); break;
        case 5: p2p_allreduce_kernel<T, 5><<<groups, P2P_THREADS, 0, s>>>(a, buf, n); break;
        case 6: p2p_allreduce_kernel<T, 6><<<groups, P2P_THREADS, 0, s>>>(a, buf, n); break;
        case 7: p2p_allreduce_kernel<T, 7><<<groups, P2P_THREADS, 0, s>>>(a, buf, n); break;
        default: p2p_allreduce_kernel<T, 8><<<groups, P2P_THREADS, 0, s>>>(a, buf, n); break;
    }
}

// ---- epoch change (see the inbox comment): clear the own lines, then a barrier over the header's barrier lines ----
__global__ void __launch_bounds__(256) p2p_clear_kernel(uint64_t* __restrict__ lines, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
        asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1" :: "v"(lines + i), "v"((uint64_t)0) : "memory");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
struct p2p_bar_t { uint64_t* dst[P2P_MAX_WORLD]; const uint64_t* src[P2P_MAX_WORLD]; char* mine; uint32_t* mirror; unsigned long long budget; uint32_t seq; int world; };
__global__ void __launch_bounds__(64) p2p_barrier_kernel(p2p_bar_t b) {
    const int r = threadIdx.x;
    if (r >= b.world) return;
    ll_store_nowait(b.dst[r], 0u, b.seq);
    unsigned long long t0 = 0;
    for (uint32_t spins = 0;; ++spins) {
        uint64_t v = ll_load_nowait(b.src[r]);
        ll_wait_loads(); ll_pin(v);
        if ((uint32_t)(v >> 32) == b.seq) return;
        if (spins == 0) t0 = p2p_clock();
        if ((spins & 63) == 63 && (p2p_clock() - t0 > b.budget || __hip_atomic_load(p2p_status(b.mine), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) {
            const uint32_t missing = 1u | (1u << (8 + r));
            __hip_atomic_fetch_or(p2p_status(b.mine), missing, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (b.mirror) __hip_atomic_fetch_or(b.mirror, missing, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            return;
        }
        __builtin_amdgcn_s_sleep(1);
    }
}
static int p2p_new_epoch(mi_comm* c, hipStream_t s) {
    const size_t n_lines = 2 * (size_t)c->world * 2 * c->cap / 8;
    p2p_clear_kernel<<<64, 256, 0, s>>>(reinterpret_cast<uint64_t*>(c->inbox + P2P_HDR_BYTES), n_lines);
    MI_LAUNCH_CHECK();
    p2p_bar_t b;
    memset(&b, 0, sizeof(b));
    if (++c->bseq == 0) c->bseq = 1;   // (2^32 epochs of 2^32 all-reduces each: fixed-length lines, rewritten at every use of their parity — a repeat is harmless)
    const int bp = c->bseq & 1;
    auto line = [&](char* box, int r) { return reinterpret_cast<uint64_t*>(box + P2P_BAR_OFF) + (size_t)bp * P2P_MAX_WORLD + r; };
    for (int d = 0; d < c->world; ++d) {
        const int p = (c->rank + d) % c->world;
        b.dst[d] = line(c->peer[p], c->synthetic ? p : c->rank);
        b.src[d] = line(c->inbox, d);
    }
    b.mine = c->status; b.mirror = c->mirror; b.budget = c->budget; b.seq = c->bseq; b.world = c->world;
    p2p_barrier_kernel<<<1, 64, 0, s>>>(b);
    MI_LAUNCH_CHECK();
    c->seq = 0;
    return MI_OK;
}

// INBOXES ARE NEVER RETURNED TO THE ALLOCATOR WHILE THE PROCESS LIVES (round 6, found by tests/test_gpu_synthetic_world.py).  An inbox is uncached memory, cleared with
// hipMemset and polled with sc0 sc1 loads; after hipFree its physical pages go to the next allocation — a torch tensor, cached memory — and on this chip readers of that
// tensor then met STALE lines of the inbox's life (zeros of the clear, in exactly the 87 KB the exchanged lines had occupied in each 2 MiB slot) in one XCD's L2 until
// something evicted them: a DQN slab-sum launch summed zeros for parts of six gradient slabs the TD launch had just written, once, in the first update behind a
// communicator's destruction; memory itself was right (a second read after a cache flush returned the data), and with the inboxes leaked instead of freed the effect was
// gone.  So mi_comm_destroy parks the inbox here and p2p_new takes a parked one of the same device, kind and size (cleared again, as a new one is); a process that
// creates one communicator per carrier — every production run — never sees the pool.  Beyond P2P_POOL entries an inbox is leaked rather than freed.
#define P2P_POOL 64
struct p2p_parked_t { void* p; size_t bytes; int kind, device; };
static p2p_parked_t g_parked[P2P_POOL];
static void* p2p_take_parked(size_t bytes, int device, int first_kind, int* kind) {
    for (int i = 0; i < P2P_POOL; ++i)
        if (g_parked[i].p && g_parked[i].bytes == bytes && g_parked[i].device == device && g_parked[i].kind >= first_kind) {
            void* p = g_parked[i].p;
            *kind = g_parked[i].kind;
            g_parked[i].p = nullptr;
            return p;
        }
    return nullptr;
}
static void p2p_park(void* p, size_t bytes, int kind, int device) {
    for (int i = 0; i < P2P_POOL; ++i)
        if (!g_parked[i].p) { g_parked[i].p = p; g_parked[i].bytes = bytes; g_parked[i].kind = kind; g_parked[i].device = device; return; }
    // pool full: the allocation stays with the process (a leak of address space is harmless; recycled pages were not)
}

static int p2p_new(int world, int rank, size_t max_bytes, int synthetic, mi_comm** out) {
    mi_comm* c = (mi_comm*)calloc(1, sizeof(mi_comm));
    if (!c) { mi_set_error("mi_comm_p2p: out of host memory"); return MI_ENOMEM; }
    c->carrier = CARRIER_P2P; c->world = world; c->rank = rank; c->synthetic = synthetic; c->colocated = 1;
    c->cap = (max_bytes + 255) & ~(size_t)255;
    unsigned long long ms = 30000;   // a stall this long on one rank (a checkpoint write, a debugger) fails the run LOUDLY (fail-safe below) instead of being waited out
    if (const char* e = getenv("MIRL_P2P_TIMEOUT_MS")) { const long long v = atoll(e); if (v > 0) ms = (unsigned long long)v; }
    c->budget = ms * 100000ull;   // s_memrealtime: 100 MHz
    hipError_t e = hipGetDevice(&c->device);
    if (e != hipSuccess) { mi_set_error("mi_comm_p2p: hipGetDevice failed: %s", hipGetErrorString(e)); free(c); return MI_EHIP; }
    const size_t bytes = P2P_HDR_BYTES + 2 * (size_t)world * 2 * c->cap;   // 2 parities x world slots x (cap payload bytes as 8-byte lines)
    // uncached (or at least fine-grained) device memory: a peer's stores over xGMI must not meet stale lines in the owner's L2 while the owner polls inside a kernel
    const unsigned kinds[3] = {hipDeviceMallocUncached, hipDeviceMallocFinegrained, hipDeviceMallocDefault};
    const char* want = getenv("MIRL_P2P_MEM");   // "uncached" | "finegrained" | "plain": start of the fallback chain (diagnostic)
    int first = 0;
    if (want && !strcmp(want, "finegrained")) first = 1;
    if (want && !strcmp(want, "plain")) first = 2;
    void* box = p2p_take_parked(bytes, c->device, first, &c->mem_kind);
    if (box && !synthetic) {   // (a parked inbox of a synthetic communicator was never checked for export)
        hipIpcMemHandle_t h;
        if (hipIpcGetMemHandle(&h, box) != hipSuccess) { (void)hipGetLastError(); p2p_park(box, bytes, c->mem_kind, c->device); box = nullptr; }
    }
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
    c->inbox_bytes = bytes;
    if (hipMemset(box, 0, bytes) != hipSuccess || hipDeviceSynchronize() != hipSuccess) {
        mi_set_error("mi_comm_p2p: clearing the inbox failed"); p2p_park(box, bytes, c->mem_kind, c->device); free(c); return MI_EHIP;
    }
    c->inbox = (char*)box;
    if (hipMalloc((void**)&c->status, 64) != hipSuccess || hipMemset(c->status, 0, 64) != hipSuccess || hipDeviceSynchronize() != hipSuccess) {
        mi_set_error("mi_comm_p2p: cannot allocate the status word"); (void)hipGetLastError(); p2p_park(box, bytes, c->mem_kind, c->device); free(c); return MI_ENOMEM;
    }
    {   // the status word's host-visible mirror: written (system scope) by the wait that runs out, read by the host without a sync
        uint32_t* h = nullptr;
        if (hipHostMalloc((void**)&h, 64, hipHostMallocMapped | hipHostMallocPortable) != hipSuccess) {
            mi_set_error("mi_comm_p2p: cannot allocate the pinned status mirror"); (void)hipGetLastError(); (void)hipFree(c->status); p2p_park(box, bytes, c->mem_kind, c->device); free(c); return MI_ENOMEM;
        }
        h[0] = 0u;
        c->mirror = h;
    }
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
    if (e != hipSuccess) { mi_set_error("mi_comm_p2p_alloc: hipIpcGetMemHandle failed: %s", hipGetErrorString(e)); p2p_park(c->inbox, c->inbox_bytes, c->mem_kind, c->device); (void)hipFree(c->status); (void)hipHostFree(c->mirror); free(c); return MI_EHIP; }
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

int mi_comm_p2p_next(void* comm, size_t n_words, p2p_args_t* a, int* world_out, hipStream_t s) {
    mi_comm* c = (mi_comm*)comm;
    if (!c || c->carrier != CARRIER_P2P) { mi_set_error("mi_comm: not a P2P communicator"); return MI_EINVAL; }
    if (!c->connected) { mi_set_error("mi_comm (p2p): all-reduce before mi_comm_p2p_connect"); return MI_ESTATE; }
    if (n_words * 4 > c->cap) { mi_set_error("mi_comm (p2p): a %zu-byte message does not fit the %zu-byte slots (max_bytes of mi_comm_p2p_alloc)", n_words * 4, c->cap); return MI_EINVAL; }
    if (c->seq >= P2P_SEQ_LAST) { const int rc = p2p_new_epoch(c, s); if (rc) return rc; }
    ++c->seq;            // 1 .. P2P_SEQ_LAST inside an epoch; 0 is the cleared inbox
    c->parity ^= 1u;     // its own bit: consecutive all-reduces never share a slot set, whatever the sequence number does
    memset(a, 0, sizeof(*a));
    const int parity = (int)c->parity, world = c->world;
    auto slot = [&](char* box, int r) { return reinterpret_cast<uint64_t*>(box + P2P_HDR_BYTES + ((size_t)parity * world + r) * 2 * c->cap); };
    for (int d = 0; d < world; ++d) {
        const int p = (c->rank + d) % world;
        a->dst[d] = slot(c->peer[p], c->synthetic ? p : c->rank);
        if (c->synthetic && p != 0) a->zeros |= 1u << d;
        a->src[d] = slot(c->inbox, d);
    }
    a->mine = c->status; a->mirror = c->mirror; a->budget = c->budget; a->seq = c->seq;
    if (world_out) *world_out = world;
    return MI_OK;
}

bool mi_comm_is_p2p(void* comm) { return comm && ((mi_comm*)comm)->carrier == CARRIER_P2P; }
const uint32_t* mi_comm_gate(void* comm) { return mi_comm_is_p2p(comm) ? reinterpret_cast<const uint32_t*>(((mi_comm*)comm)->status) : nullptr; }

static void p2p_describe_failure(mi_comm* c, uint32_t st, const char* tail) {
    char who[64]; int k = 0;
    who[0] = 0;
    for (int r = 0; r < c->world; ++r) if (st & (1u << (8 + r))) k += snprintf(who + k, sizeof(who) - k, " %d", r);
    mi_set_error("mi_comm (p2p), rank %d: a wait ran out (MIRL_P2P_TIMEOUT_MS); ranks that never arrived:%s — %s", c->rank, k ? who : " ?", tail);
}
// no synchronisation: a plain host load of the pinned mirror the timed-out wait wrote (system scope) — what every mi_*_sharded call does first
int mi_comm_poll_impl(void* comm) {
    if (!mi_comm_is_p2p(comm)) return MI_OK;
    mi_comm* c = (mi_comm*)comm;
    const uint32_t st = c->mirror ? __atomic_load_n(c->mirror, __ATOMIC_RELAXED) : 0u;
    if (!st) return MI_OK;
    p2p_describe_failure(c, st, "every optimizer step behind that exchange was WITHHELD (parameters, moments and targets are those in front of it); elements of that and "
                                "every later all-reduce hold the LOCAL share; the communicator is dead (destroy it)");
    return MI_ESTATE;
}
extern "C" int mi_comm_poll(void* comm) {
    MI_CHECK_ARG(comm != nullptr, "comm is NULL");
    return mi_comm_poll_impl(comm);
}
// TEST HOOK: the sequence number of the last all-reduce inside the epoch (every rank must set the same value, between all-reduces) — presets it just below the epoch
// change so that a test crosses it in a few exchanges (tests/test_gpu_p2p.py)
extern "C" int mi_comm_test_set_seq(void* comm, uint32_t seq) {
    MI_CHECK_ARG(mi_comm_is_p2p(comm), "not a P2P communicator");
    ((mi_comm*)comm)->seq = seq;
    return MI_OK;
}

// May a kernel of 145 x 1,024 threads (grad_reduce_kernel) spin-wait for its peers' exchange?  One rank per device: always.  Ranks SHARING a device (test placements):
// every waiting rank holds 145 workgroups' worth of wave slots and registers while it spins, and a peer's gradient launch needs whole CUs (512 VGPRs per SIMD lane): two
// ranks leave it 111 CUs, from three waiting ranks on the chip is full of waiters and the rank they wait for cannot be scheduled — a (bounded) deadlock, seen with 8 ranks
// x 4096 envs on one MI355X.  More than two colocated ranks therefore take the stand-alone all-reduce launch (18 small workgroups per rank).  MIRL_P2P_FUSED=0 / 1 overrides.
bool mi_comm_p2p_fused_ok(void* comm) {
    if (!mi_comm_is_p2p(comm)) return false;
    const mi_comm* c = (const mi_comm*)comm;
    if (c->fused_set) return c->fused_set > 0;     // agreed over the ranks by the caller (mi_comm_p2p_set_fused)
    if (const char* e = getenv("MIRL_P2P_FUSED")) return atoi(e) != 0;
    return c->colocated <= 2;
}
// The ranks of a communicator MUST take the same form of PPO's gradient exchange: the in-launch one publishes line p in SLAB order, the stand-alone one in PARAMETER
// order, both under the same sequence number — a mix sums permuted elements silently.  deep_rl_amd.dist agrees on one value over the process group
// (MAX of the per-device rank counts, the MIRL_P2P_FUSED settings compared) and fixes it here: -1 = stand-alone launch, 1 = in-launch, 0 = back to the rule above.
extern "C" int mi_comm_p2p_set_fused(void* comm, int mode) {
    MI_CHECK_ARG(mi_comm_is_p2p(comm), "not a P2P communicator");
    MI_CHECK_ARG(mode >= -1 && mode <= 1, "mode must be -1 (stand-alone), 0 (by colocation) or 1 (in-launch)");
    ((mi_comm*)comm)->fused_set = mode;
    return MI_OK;
}

extern "C" int mi_comm_p2p_set_colocated(void* comm, int ranks_on_this_device) {
    MI_CHECK_ARG(mi_comm_is_p2p(comm), "not a P2P communicator");
    MI_CHECK_ARG(ranks_on_this_device >= 1 && ranks_on_this_device <= ((mi_comm*)comm)->world, "ranks_on_this_device must be in [1, world_size]");
    ((mi_comm*)comm)->colocated = ranks_on_this_device;
    return MI_OK;
}

static int p2p_allreduce(mi_comm* c, void* buf, size_t n, int dtype, hipStream_t s) {
    p2p_args_t a;
    int world = 1;
    int rc = mi_comm_p2p_next(c, n * (dtype ? 2 : 1), &a, &world, s);
    if (rc) return rc;
    // two elements per thread while that gives <= P2P_MAX_GROUPS workgroups (PPO's 9,159 floats: 18 workgroups), more beyond (SAC's 134,660: 9 per thread)
    size_t groups = (n + 2 * P2P_THREADS - 1) / (2 * P2P_THREADS);
    if (groups > P2P_MAX_GROUPS) groups = P2P_MAX_GROUPS;
    if (dtype) p2p_launch<double>(world, (unsigned)groups, s, a, (double*)buf, n);
    else p2p_launch<float>(world, (unsigned)groups, s, a, (float*)buf, n);
    MI_LAUNCH_CHECK();
    return MI_OK;
}

// Host-synchronising: MI_OK, or MI_ESTATE when a wait of the P2P carrier ran out (mi_last_error names the ranks that never arrived).  RCCL: always MI_OK.
extern "C" int mi_comm_check(void* comm) {
    MI_CHECK_ARG(comm != nullptr, "comm is NULL");
    mi_comm* c = (mi_comm*)comm;
    if (c->carrier != CARRIER_P2P) return MI_OK;
    uint32_t st = 0;
    MI_HIP(hipMemcpy(&st, c->status, 4, hipMemcpyDeviceToHost));
    if (st) {
        p2p_describe_failure(c, st, "elements of that and every later all-reduce hold the LOCAL share, and every optimizer step behind it was withheld");
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
        if (c->inbox) p2p_park(c->inbox, c->inbox_bytes, c->mem_kind, c->device);   // never hipFree: its pages must not be recycled into cached memory (see p2p_park)
        if (c->status) (void)hipFree(c->status);
        if (c->mirror) (void)hipHostFree(c->mirror);
    } else if (g_rccl.so && c->comm) {
        (void)g_rccl.CommDestroy(c->comm);
    }
    free(c);
    return MI_OK;
}

extern "C" int mi_comm_info(void* comm, int* world_size, int* rank, int* rccl_version, int* comm_count) {
    MI_CHECK_ARG(comm != nullptr, "comm is NULL");
    mi_comm* c = (mi_comm*)comm;
    if (world_size) *world_size = c->synthetic ? 1 : c->world;   // a synthetic communicator is ONE rank (the caller's arithmetic is the single-rank one) playing comm_count ranks
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
    if (const int rc = mi_comm_poll_impl(comm)) return rc;
    return mi_comm_allreduce_impl(comm, buf, n, dtype, (hipStream_t)stream);
}
