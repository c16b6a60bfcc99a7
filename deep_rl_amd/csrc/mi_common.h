// mi_common.h — shared device/host helpers of libmirl (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/mi_rl.h"

// ---- flat parameter layout (== order of agent.parameters(), reference ppo.py:34-47) -------------
#define OBS 4
#define HID 64
#define NACT 2
#define A_W1 0
#define A_B1 256
#define A_W2 320
#define A_B2 4416
#define A_W3 4480
#define A_B3 4608
#define C_BASE 4610
#define NPARAMS 9155
// per-net offsets relative to the net's base
#define N_W1 0
#define N_B1 256
#define N_W2 320
#define N_B2 4416
#define N_W3 4480

// ---- error plumbing -----------------------------------------------------------------------------
void mi_set_error(const char* fmt, ...);
#define MI_CHECK_ARG(cond, msg)                                  \
    do {                                                         \
        if (!(cond)) {                                           \
            mi_set_error("%s: invalid argument: %s", __func__, msg); \
            return MI_EINVAL;                                    \
        }                                                        \
    } while (0)
#define MI_HIP(call)                                                                          \
    do {                                                                                      \
        hipError_t e_ = (call);                                                               \
        if (e_ != hipSuccess) {                                                               \
            mi_set_error("%s: %s failed: %s", __func__, #call, hipGetErrorString(e_));        \
            return MI_EHIP;                                                                   \
        }                                                                                     \
    } while (0)
#define MI_LAUNCH_CHECK() MI_HIP(hipGetLastError())

// in-library profiler (mi_env.hip): brackets a tagged launch with HIP events on its stream when armed
void mi_prof_mark(int tag, bool end, hipStream_t s);
struct mi_prof_scope {
    int tag; hipStream_t s;
    mi_prof_scope(int t, hipStream_t st) : tag(t), s(st) { mi_prof_mark(tag, false, s); }
    ~mi_prof_scope() { mi_prof_mark(tag, true, s); }
};

// ---- the launches seen from inside (diagnostic build -DMI_INSIDE, tools/inside_view.py) ------------------------------------------------------------------------
// Every wave of an instrumented kernel stores the 100 MHz wall clock (s_memrealtime) at its entry and at its exit (RAII: every return path), and mi_prof_mark puts a
// one-wave marker launch in front of and behind every tagged launch.  Per tag the tool prints: marker -> first entry (start-up), first entry -> last exit (span), last
// exit -> marker (how long the launch stays open after its last wave has left: memory-side atomics, write-backs ...).  rocprof, the counters and the stamp builds charge
// all three to "the kernel".  One array per translation unit (no relocatable device code in this build): MI_INSIDE_EXPORT(tu) exports its reader.
#ifdef MI_INSIDE
#define MI_INSIDE_SLOTS 16384   // wave slots per tag
static __device__ unsigned long long mi_inside_marks[18][2][MI_INSIDE_SLOTS];
struct mi_inside_guard {
    int tag;
    __device__ __forceinline__ static void mark(int tag, int which) {
        if ((threadIdx.x & 63) == 0) {
            unsigned long long t;
            asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
            const unsigned slot = (blockIdx.x + gridDim.x * blockIdx.y) * (blockDim.x >> 6) + (threadIdx.x >> 6);
            if (slot < MI_INSIDE_SLOTS) mi_inside_marks[tag][which][slot] = t;
        }
    }
    __device__ __forceinline__ explicit mi_inside_guard(int t) : tag(t) { mark(tag, 0); }
    __device__ __forceinline__ ~mi_inside_guard() { mark(tag, 1); }
};
#define MI_INSIDE_SCOPE(tag) mi_inside_guard mi_inside_guard_(tag)
#define MI_INSIDE_EXPORT(tu) \
    extern "C" int mi_debug_inside_##tu(unsigned long long* out, int clear) { \
        if (clear) { void* p = nullptr; if (hipGetSymbolAddress(&p, HIP_SYMBOL(mi_inside_marks)) != hipSuccess) return -2; return hipMemset(p, 0, sizeof(mi_inside_marks)) == hipSuccess ? 0 : -2; } \
        return hipMemcpyFromSymbol(out, HIP_SYMBOL(mi_inside_marks), sizeof(mi_inside_marks)) == hipSuccess ? 0 : -2; }
#else
#define MI_INSIDE_SCOPE(tag) do {} while (0)
#define MI_INSIDE_EXPORT(tu)
#endif

// ---- env handle ----------------------------------------------------------------------------------
struct mi_env {
    int kind, n, device;   // kind 1 (Pendulum): x = theta, x_dot = theta_dot; theta / theta_dot arrays unused
    uint64_t seed, env_id_base;
    // struct-of-arrays device state: fp64 dynamics, int counters
    double *x, *x_dot, *theta, *theta_dot;
    int32_t* elapsed;    // TimeLimit._elapsed_steps
    float* ep_ret;       // RecordEpisodeStatistics.episode_returns
    int32_t* ep_len;
    uint64_t* episode;   // resets so far (index of the next reset-noise draw)
    uint64_t* step_ctr;  // actions sampled so far (index of the next action uniform)
    // episode statistics WITHOUT same-address atomics: workgroup b of an acting / rollout launch stores {finished episodes, sum of their lengths, longest} at
    // stats_part[4 b ..] with plain stores; mi_env_stats_reduce sums the stats_n slots of the last such launch on request.  (Agent-scope atomics on one address are
    // performed one after the other at the memory side, ~8.5 ns each on an MI355X, and a launch is not over before the last one: 3 x 1024 of them kept
    // rollout_q4_kernel open for 26 us, 3 x 256 dqn_act4_kernel for 9 us — round 4, tools/dqn_act_stamps.py, profiles/r04h_*.)
    int32_t* stats_part;
    int stats_cap, stats_n;   // slots allocated; slots written by the last launch that kept its statistics here (host-side bookkeeping, -1: none yet)
};
#define MI_STATS_PART_MIN 32   // launches of at least this many workgroups keep their statistics per workgroup (below it the few atomics are cheaper than a reduction)
// sums the per-workgroup statistics of the last launch into out[0..3] = {episodes, sum of lengths, longest, 0} (device pointer; one small launch)
int mi_env_stats_reduce(mi_env* e, int32_t* out, hipStream_t s);

int mi_pend_reset_impl(mi_env* e, float* obs, const double* forced_state, hipStream_t s);
int mi_rollout_gae_internal(void* handle, const float* params, int T, float* obs_cur, float* observations, float* values, int64_t* actions, float* log_probs,
                            float* rewards, float* dones, mi_episode_t* episodes, int32_t* episode_stats, int max_ep, float gamma, float gae_lambda,
                            float* advantages, float* returns, double* zero_f64, int zero_n, int32_t* stats_next, void* stream);

int mi_comm_allreduce_impl(void* comm, void* buf, size_t n, int dtype /* 0 f32, 1 f64 */, hipStream_t s);   // mi_comm.hip

// ---- the P2P carrier's line protocol (mi_comm.hip; also spoken by grad_reduce_kernel, which exchanges the gradient it has just summed without a launch of its own) ----
// A LINE is one 8-byte word {payload word (low), sequence number of the all-reduce (high)}, stored and loaded as ONE 8-byte access: the payload carries its own
// "arrived" flag, so an exchange needs no fence, no flag word and no second round trip.  Line i of slot (parity, r) of an inbox = 32-bit word i of rank r's message.
#define P2P_MAX_WORLD 8
struct p2p_args_t {          // everything resolved on the host: the kernel does no address arithmetic beyond "+ element"
    uint64_t* dst[P2P_MAX_WORLD];         // d-th target: slot (parity, this rank) of rank (rank + d) % world's inbox — the own inbox first, then rank + 1 ...: the links are used side by side
    const uint64_t* src[P2P_MAX_WORLD];   // slot (parity, r) of the OWN inbox, r in rank order
    char* mine;                           // this rank's status word (plain device memory of its own: only local launches read it — peers never do — so its loads hit L2 like any other)
    uint32_t* mirror;                     // host-pinned, device-mapped copy of the status word: the host reads it at the entry of every call WITHOUT a sync (mi_comm_poll)
    unsigned long long budget;
    uint32_t seq;
    uint32_t zeros;                       // bit d: target d receives zeros (synthetic communicator: the ranks this process plays besides its own)
};
__device__ __forceinline__ uint32_t* p2p_status(char* box) { return reinterpret_cast<uint32_t*>(box); }
__device__ __forceinline__ unsigned long long p2p_clock() {
    unsigned long long t;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
    return t;
}
// system-scope 8-byte accesses, written through to the owner's memory / read past every cache (sc0 sc1), issued WITHOUT a wait: a thread puts all its stores, then all
// the loads of an element, in flight together (the compiler's own system-scope atomics wait for each one: 8 dependent round trips at world 8)
__device__ __forceinline__ void ll_store_nowait(uint64_t* p, uint32_t w, uint32_t seq) {
    const uint64_t v = ((uint64_t)seq << 32) | w;
    asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1" :: "v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ uint64_t ll_load_nowait(const uint64_t* p) {
    uint64_t v;
    asm volatile("global_load_dwordx2 %0, %1, off sc0 sc1" : "=v"(v) : "v"(p) : "memory");
    return v;
}
__device__ __forceinline__ void ll_wait_loads() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void ll_pin(uint64_t& v) { asm volatile("" : "+v"(v)); }   // every use of a loaded value sits behind this point, which sits behind ll_wait_loads()

// Gathers the W lines of one element from every rank's slot in the own inbox: all WORLD * W loads in flight together, re-issued TOGETHER until every line carries the
// all-reduce's sequence number (polling them one after the other costs a memory round trip per rank).  false when the budget ran out (or an earlier wait on this
// communicator did): the status word then names the ranks whose lines never arrived (bit 8 + r).
template <int WORLD, int W>
__device__ __forceinline__ bool ll_gather(const p2p_args_t& x, size_t line, uint64_t (&v)[WORLD][W]) {
    unsigned long long t0 = 0;
    for (uint32_t spins = 0;; ++spins) {
#pragma unroll
        for (int r = 0; r < WORLD; ++r)
#pragma unroll
            for (int k = 0; k < W; ++k) v[r][k] = ll_load_nowait(x.src[r] + line + k);
        ll_wait_loads();
        bool all = true;
#pragma unroll
        for (int r = 0; r < WORLD; ++r)
#pragma unroll
            for (int k = 0; k < W; ++k) { ll_pin(v[r][k]); all = all && (uint32_t)(v[r][k] >> 32) == x.seq; }
        if (all) return true;
        if (spins == 0) t0 = p2p_clock();
        if ((spins & 63) == 63 && (p2p_clock() - t0 > x.budget || __hip_atomic_load(p2p_status(x.mine), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) {
            uint32_t missing = 1u;
#pragma unroll
            for (int r = 0; r < WORLD; ++r)
#pragma unroll
                for (int k = 0; k < W; ++k) if ((uint32_t)(v[r][k] >> 32) != x.seq) missing |= 1u << (8 + r);
            __hip_atomic_fetch_or(p2p_status(x.mine), missing, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (x.mirror) __hip_atomic_fetch_or(x.mirror, missing, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            return false;
        }
        __builtin_amdgcn_s_sleep(1);
    }
}

template <typename T> struct ll_elem;   // an element as W payload words
template <> struct ll_elem<float> {
    static constexpr int W = 1;
    static __device__ __forceinline__ void split(float x, uint32_t* w) { w[0] = __float_as_uint(x); }
    static __device__ __forceinline__ float join(const uint64_t* v) { return __uint_as_float((uint32_t)v[0]); }
};
template <> struct ll_elem<double> {
    static constexpr int W = 2;
    static __device__ __forceinline__ void split(double x, uint32_t* w) { const uint64_t b = (uint64_t)__double_as_longlong(x); w[0] = (uint32_t)b; w[1] = (uint32_t)(b >> 32); }
    static __device__ __forceinline__ double join(const uint64_t* v) { return __longlong_as_double((long long)(((uint64_t)(uint32_t)v[1] << 32) | (uint32_t)v[0])); }
};

// One f32 element of an all-reduce carried INSIDE a launch that has just produced it (PPO's grad_reduce_kernel, DQN's slab-sum kernels): store it as line `line` into
// slot (parity, rank) of every rank's inbox, poll the WORLD lines of it in the own inbox, add them IN RANK ORDER.  WORLD = 0: no exchange.  A wait that runs out
// returns the local share (the status word is set: every optimizer step behind it is withheld).
template <int WORLD>
__device__ __forceinline__ float p2p_exchange(const p2p_args_t& x, int line, float t) {
    if constexpr (WORLD > 0) {
#pragma unroll
        for (int d = 0; d < WORLD; ++d) ll_store_nowait(x.dst[d] + line, (x.zeros >> d) & 1 ? 0u : __float_as_uint(t), x.seq);
        uint64_t v[WORLD][1];
        if (!ll_gather<WORLD, 1>(x, (size_t)line, v)) return t;
        float acc = __uint_as_float((uint32_t)v[0][0]);
#pragma unroll
        for (int r = 1; r < WORLD; ++r) acc += __uint_as_float((uint32_t)v[r][0]);
        return acc;
    } else {
        return t;
    }
}

// The same with the rank count as a RUN-TIME value (SAC's gradient assembly kernels: four sites, by-value kernel arguments — no room for nine instantiations of each):
// the loops run over P2P_MAX_WORLD with the ranks beyond `world` predicated off; stores and loads are issued together exactly as above.
__device__ __forceinline__ float p2p_exchange_rt(const p2p_args_t& x, int world, int line, float t) {
#pragma unroll
    for (int d = 0; d < P2P_MAX_WORLD; ++d) if (d < world) ll_store_nowait(x.dst[d] + line, (x.zeros >> d) & 1 ? 0u : __float_as_uint(t), x.seq);
    uint64_t v[P2P_MAX_WORLD];
    unsigned long long t0 = 0;
    for (uint32_t spins = 0;; ++spins) {
#pragma unroll
        for (int r = 0; r < P2P_MAX_WORLD; ++r) v[r] = r < world ? ll_load_nowait(x.src[r] + line) : 0ull;
        ll_wait_loads();
        bool all = true;
#pragma unroll
        for (int r = 0; r < P2P_MAX_WORLD; ++r) if (r < world) { ll_pin(v[r]); all = all && (uint32_t)(v[r] >> 32) == x.seq; }
        if (all) break;
        if (spins == 0) t0 = p2p_clock();
        if ((spins & 63) == 63 && (p2p_clock() - t0 > x.budget || __hip_atomic_load(p2p_status(x.mine), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) {
            uint32_t missing = 1u;
#pragma unroll
            for (int r = 0; r < P2P_MAX_WORLD; ++r) if (r < world && (uint32_t)(v[r] >> 32) != x.seq) missing |= 1u << (8 + r);
            __hip_atomic_fetch_or(p2p_status(x.mine), missing, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (x.mirror) __hip_atomic_fetch_or(x.mirror, missing, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            return t;   // the local share; every optimizer step behind it is withheld (gate)
        }
        __builtin_amdgcn_s_sleep(1);
    }
    float acc = __uint_as_float((uint32_t)v[0]);
#pragma unroll
    for (int r = 1; r < P2P_MAX_WORLD; ++r) if (r < world) acc += __uint_as_float((uint32_t)v[r]);
    return acc;
}

// Fills `a` for the NEXT all-reduce of `n_words` 32-bit words on a P2P communicator (advances its sequence number: the caller MUST launch exactly one kernel that
// publishes and consumes those lines on every rank, on stream `s`); MI_EINVAL when the message does not fit, MI_ESTATE when the communicator is not connected or an
// earlier wait on it ran out.  When the 32-bit sequence number is about to wrap, the call first enqueues the epoch change on `s` (inbox cleared, barrier: mi_comm.hip).
int mi_comm_p2p_next(void* comm, size_t n_words, p2p_args_t* a, int* world, hipStream_t s);
bool mi_comm_is_p2p(void* comm);
// The FAIL-SAFE of the P2P carrier.  A wait that runs out sets the rank's status word (plain device memory) and its host-pinned mirror; from then on
//   * every optimizer step that would consume an exchanged gradient is WITHHELD: its launch reads the word behind `mi_comm_gate` with its state and leaves parameters,
//     moments and targets as they are (the pattern of mi_sac.hip's fault word) — a stalled peer costs the update, never the parameters;
//   * every mi_*_sharded call and mi_comm_allreduce_sum returns MI_ESTATE at its entry (mi_comm_poll_impl: a plain host load of the mirror, no synchronisation).
const uint32_t* mi_comm_gate(void* comm);   // device pointer of the status word (nullptr: RCCL / no communicator — nothing to gate on)
int mi_comm_poll_impl(void* comm);          // MI_OK / MI_ESTATE, no synchronisation (NULL comm: MI_OK)
int mi_clip_adam_gated(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int n, int64_t step, double lr, double beta1, double beta2, double eps,
                       float max_norm, float* grad_norm, void* comm, void* stream);   // mi_update.hip: mi_clip_adam, withheld when comm's status word is set
#ifdef MI_NO_GATE   // A/B build only (tools/ab_headline.sh): what the fail-safe's loads and branches cost the launches that carry them
__device__ __forceinline__ bool mi_gate_closed(const uint32_t*) { return false; }
#else
__device__ __forceinline__ bool mi_gate_closed(const uint32_t* gate) { return gate != nullptr && __hip_atomic_load(gate, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u; }
#endif
bool mi_comm_p2p_fused_ok(void* comm);   // grad_reduce_kernel may carry the exchange (false when more than two ranks share this device: see mi_comm.hip)

// ---- RNG contract (include/mi_rl.h) ----------------------------------------------------------------
#define STREAM_RESET 0u
#define STREAM_ACTION 1u
#define STREAM_PERM 2u

__host__ __device__ inline void philox4x32_10(uint32_t c[4], uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
}

__host__ __device__ inline void mi_philox(uint64_t seed, uint64_t env, uint64_t idx, uint32_t stream, uint32_t out[4]) {
    out[0] = (uint32_t)env; out[1] = (uint32_t)(env >> 32); out[2] = (uint32_t)idx;
    out[3] = ((uint32_t)(idx >> 32) << 4) | stream;
    philox4x32_10(out, (uint32_t)seed, (uint32_t)(seed >> 32));
}

// reset noise in [-0.05, 0.05): numpy's low + (high-low)*u restated in f64 (no contraction)
__device__ inline void mi_reset_noise(uint64_t seed, uint64_t env, uint64_t episode, double s[4]) {
    uint32_t r[4];
    mi_philox(seed, env, episode, STREAM_RESET, r);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        double u = __dmul_rn(__dadd_rn((double)r[i], 0.5), 1.0 / 4294967296.0);
        s[i] = __dadd_rn(-0.05, __dmul_rn(0.05 - -0.05, u));
    }
}

// action draw of env-step `step`: word (step & 3) of the Philox block with idx = step >> 2 (one block feeds 4 steps)
__device__ inline float mi_u32_to_uniform(uint32_t w) { return (float)(w >> 8) * (1.0f / 16777216.0f); }
__device__ inline float mi_action_uniform(uint64_t seed, uint64_t env, uint64_t step) {
    uint32_t r[4];
    mi_philox(seed, env, step >> 2, STREAM_ACTION, r);
    return mi_u32_to_uniform(r[step & 3]);
}

__host__ __device__ inline uint32_t mi_mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

// keyed bijection on [0, n): 6-round alternating Feistel on `bits` bits (a = bits/2 low) + cycle walking
__host__ __device__ inline uint32_t mi_feistel(uint32_t i, uint32_t n, uint32_t a, uint32_t b, uint32_t k0, uint32_t k1) {
    const uint32_t ma = (1u << a) - 1u, mb = (1u << b) - 1u;
    uint32_t x = i;
    do {
        uint32_t l = x & ma, r = x >> a;
#pragma unroll
        for (uint32_t rd = 0; rd < 6; rd += 2) {
            l ^= mi_mix32(r * 0x9E3779B1u + k0 + rd) & ma;
            r ^= mi_mix32(l * 0x85EBCA77u + k1 + rd + 1u) & mb;
        }
        x = (r << a) | l;
    } while (x >= n);
    return x;
}

// ---- math ------------------------------------------------------------------------------------------
// tanh = 1 - 2/(2^(2x log2 e) + 1): 5 instructions (v_exp_f32, v_rcp_f32), branch-free, exact limits at +-inf.
// Absolute error <= ~2.5e-7 (tests/test_gpu_parity.py::test_device_tanh_accuracy); the RELATIVE error is not bounded
// near 0 — activations enter O(1) sums, so absolute accuracy is what the fp32 parity tolerances need.  The rollout,
// the forward API and the update kernel all use this one function, so new/old log-probs are consistent.
__device__ __forceinline__ float mi_tanhf(float x) {
    const float e = __builtin_amdgcn_exp2f(x * 2.8853900817779268f);
    return __builtin_fmaf(-2.0f, __builtin_amdgcn_rcpf(e + 1.0f), 1.0f);
}

// Categorical(logits=l) for 2 actions (reference ppo.py:52-54 / torch.distributions.Categorical):
// nl = l - logsumexp(l), p = softmax(nl), entropy = -sum(nl * p)
__device__ __forceinline__ void mi_categorical2(float l0, float l1, float& nl0, float& nl1, float& p0, float& p1, float& ent) {
    const float m = fmaxf(l0, l1);
    const float s = expf(l0 - m) + expf(l1 - m);
    const float lse = logf(s) + m;
    nl0 = l0 - lse; nl1 = l1 - lse;
    const float m2 = fmaxf(nl0, nl1);
    const float e0 = expf(nl0 - m2), e1 = expf(nl1 - m2);
    const float s2 = e0 + e1;
    p0 = e0 / s2; p1 = e1 / s2;
    ent = -(nl0 * p0 + nl1 * p1);
}

// sin/cos of the pole angle.  |theta| stays below ~0.25 rad while an episode is alive, so the fdlibm
// __kernel_sin/__kernel_cos polynomials (|x| < pi/4, error < 1 ulp, Sun Microsystems public algorithm)
// replace ocml's full-range routines; larger arguments (only reachable through forced states) fall back.
// gym evaluates math.sin/math.cos (glibc); any < 1 ulp implementation can differ from it in the last bit,
// which is why parity on the float64 state is asserted to tolerance and bit-exactness on obs_f32/done.
__device__ __forceinline__ void mi_sincos(double x, double& s, double& c) {
    if (fabs(x) < 0.5) {
        const double z = x * x;
        const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03, S3 = -1.98412698298579493134e-04,
                     S4 = 2.75573137070700676789e-06, S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
        const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03, C3 = 2.48015872894767294178e-05,
                     C4 = -2.75573143513906633035e-07, C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
        const double v = z * x;
        const double rs = __builtin_fma(z, __builtin_fma(z, __builtin_fma(z, __builtin_fma(z, S6, S5), S4), S3), S2);
        s = __builtin_fma(v, __builtin_fma(z, rs, S1), x);
        const double rc = z * __builtin_fma(z, __builtin_fma(z, __builtin_fma(z, __builtin_fma(z, __builtin_fma(z, C6, C5), C4), C3), C2), C1);
        c = 1.0 - (0.5 * z - z * rc);
    } else {
        s = sin(x);
        c = cos(x);
    }
}

// The same distribution for the hot kernels: one exp + one log for the log-sum-exp (exp(0) = 1 is exact), p = exp(nl)
// (softmax of already-normalised logits), hardware exp2/log2 (v_exp_f32 / v_log_f32, ~1 ulp) — absolute error ~1e-7,
// well inside the fp32 parity tolerances, at a third of the instructions.
__device__ __forceinline__ float mi_fast_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.4426950408889634f); }
__device__ __forceinline__ float mi_fast_log(float x) { return __builtin_amdgcn_logf(x) * 0.6931471805599453f; }
__device__ __forceinline__ void mi_categorical2_fast(float l0, float l1, float& nl0, float& nl1, float& p0, float& p1, float& ent) {
    const float m = fmaxf(l0, l1);
    const float s = 1.0f + mi_fast_exp(-fabsf(l0 - l1));
    const float lse = mi_fast_log(s) + m;
    nl0 = l0 - lse; nl1 = l1 - lse;
    p0 = mi_fast_exp(nl0); p1 = mi_fast_exp(nl1);
    ent = -(nl0 * p0 + nl1 * p1);
}

// full-range sin/cos (Pendulum's angle is unbounded): Cody-Waite reduction by pi/2 with two constants + the fdlibm kernels.
// Bit-identical to oracle/cpu_ref.c ref_sincos_full in its device-matched mode (same IEEE operation sequence).
__device__ __forceinline__ void mi_sincos_full(double x, double& s, double& c) {
    const double n = __builtin_rint(x * 6.36619772367581382433e-01);
    double r = __builtin_fma(-n, 1.57079632673412561417e+00, x);
    r = __builtin_fma(-n, 6.07710050650619224932e-11, r);
    const double z = r * r;
    const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03, S3 = -1.98412698298579493134e-04,
                 S4 = 2.75573137070700676789e-06, S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
    const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03, C3 = 2.48015872894767294178e-05,
                 C4 = -2.75573143513906633035e-07, C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
    const double v = z * r;
    const double rs = __builtin_fma(z, __builtin_fma(z, __builtin_fma(z, __builtin_fma(z, S6, S5), S4), S3), S2);
    const double ks = __builtin_fma(v, __builtin_fma(z, rs, S1), r);
    const double rc = z * __builtin_fma(z, __builtin_fma(z, __builtin_fma(z, __builtin_fma(z, __builtin_fma(z, C6, C5), C4), C3), C2), C1);
    const double kc = 1.0 - (0.5 * z - z * rc);
    const long long q = (long long)n & 3;
    s = q == 0 ? ks : q == 1 ? kc : q == 2 ? -ks : -kc;
    c = q == 0 ? kc : q == 1 ? -ks : q == 2 ? -kc : ks;
}

// ---- Pendulum-v1 step (gym 0.21 pendulum.py), fp64, no FMA contraction; obs = (cos, sin, theta_dot) of the NEW state ----
#define PEND_MAX_STEPS 200
__device__ __forceinline__ void mi_pendulum_step(double& th, double& thdot, float u_in, double& reward) {
    const double PI = 3.14159265358979323846, max_speed = 8.0, max_torque = 2.0, dt = 0.05, g = 10.0, m = 1.0, l = 1.0;
    double u = (double)u_in;
    u = u < -max_torque ? -max_torque : (u > max_torque ? max_torque : u);
    double an = fmod(th + PI, 2 * PI);
    if (an != 0.0 && an < 0.0) an += 2 * PI;
    an -= PI;
    reward = -(an * an + 0.1 * (thdot * thdot) + 0.001 * (u * u));
    double sn, cs;
    mi_sincos_full(th, sn, cs);
    double nd = thdot + (3 * g / (2 * l) * sn + 3.0 / (m * (l * l)) * u) * dt;
    nd = nd < -max_speed ? -max_speed : (nd > max_speed ? max_speed : nd);
    th = th + nd * dt;
    thdot = nd;
}
__device__ __forceinline__ void mi_pendulum_reset_noise(uint64_t seed, uint64_t env, uint64_t episode, double& th, double& thdot) {
    const double PI = 3.14159265358979323846;
    uint32_t r[4];
    mi_philox(seed, env, episode, STREAM_RESET, r);
    const double u0 = __dmul_rn(__dadd_rn((double)r[0], 0.5), 1.0 / 4294967296.0), u1 = __dmul_rn(__dadd_rn((double)r[1], 0.5), 1.0 / 4294967296.0);
    th = __dadd_rn(-PI, __dmul_rn(PI - -PI, u0));
    thdot = __dadd_rn(-1.0, __dmul_rn(1.0 - -1.0, u1));
}

// ---- CartPole-v1 step (gym 0.21 cartpole.py), fp64, no FMA contraction (-ffp-contract=off) ---------
#define CP_MAX_STEPS 500
__device__ __forceinline__ void mi_cartpole_step(double& x, double& x_dot, double& theta, double& theta_dot, int action,
                                                 int& terminated) {
    const double gravity = 9.8, masscart = 1.0, masspole = 0.1, length = 0.5, force_mag = 10.0, tau = 0.02;
    const double total_mass = masspole + masscart;
    const double polemass_length = masspole * length;
    const double theta_thr = 12 * 2 * 3.14159265358979323846 / 360;
    const double x_thr = 2.4;
    const double force = action == 1 ? force_mag : -force_mag;
    double costheta, sintheta;
    mi_sincos(theta, sintheta, costheta);
    // x / total_mass, bit-identical to the IEEE quotient: with y = RN(1/c), q = RN(x*y), r = x - c*q (exact, one FMA),
    // RN(q + r*y) is the correctly rounded x/c (Markstein 1990; 4e8 random operands checked on the CPU, and the
    // float64-state parity tests compare against the oracle's true divisions).  Saves three v_rcp_f64 sequences per step.
    const double inv_tm = 1.0 / total_mass;
    auto div_tm = [&](double v) { const double q = v * inv_tm; return __builtin_fma(__builtin_fma(-total_mass, q, v), inv_tm, q); };
    const double temp = div_tm(force + polemass_length * (theta_dot * theta_dot) * sintheta);
    const double thetaacc = (gravity * sintheta - costheta * temp) /
                            (length * (4.0 / 3.0 - div_tm(masspole * (costheta * costheta))));
    const double xacc = temp - div_tm(polemass_length * thetaacc * costheta);
    x = x + tau * x_dot;
    x_dot = x_dot + tau * xacc;
    theta = theta + tau * theta_dot;
    theta_dot = theta_dot + tau * thetaacc;
    terminated = (x < -x_thr || x > x_thr || theta < -theta_thr || theta > theta_thr) ? 1 : 0;
}

// ---- one Adam step of one element (torch single-tensor Adam: ppo.py:90,192; dqn.py:68,132; sac.py:117-122,185,200,209) -------------
// m <- lerp(m, g, 1 - beta1); v <- beta2 v + (1 - beta2) g^2; p <- p - (lr / bc1) * m / (sqrt(v) / sqrt(bc2) + eps), with the hardware
// sqrt / reciprocal (v_sqrt_f32, v_rcp_f32: 1 ulp each) and rbc2 = 1 / sqrt(bc2) from the host: 10 instructions instead of ~45 for the
// IEEE sqrt + two divisions.  The update term (<= ~lr) is off by <= 4 ulp of ITSELF, i.e. <= 3e-10 absolute — far below one ulp of a
// parameter; every kernel of this library that steps an optimizer uses this one function, so fused and unfused paths agree bit for bit.
__device__ __forceinline__ float mi_adam_elem(float p, float g, float& m, float& v, float w1, float b2, float w2, float step_size, float rbc2, float eps) {
    m = m + w1 * (g - m);
    v = v * b2 + w2 * (g * g);
    const float denom = __builtin_amdgcn_sqrtf(v) * rbc2 + eps;
    return p + (-step_size) * (m * __builtin_amdgcn_rcpf(denom));
}

// ---- wave-level reductions (64 lanes): DPP inside a 16-lane row, then two cross-row exchanges ------------
__device__ __forceinline__ float dpp_xor1(float v) { return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true)); }
__device__ __forceinline__ float dpp_xor2(float v) { return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true)); }
__device__ __forceinline__ float dpp_half_mirror(float v) { return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true)); }
__device__ __forceinline__ float dpp_mirror(float v) { return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true)); }

// sum over the 64 lanes of a wave on the VALU only (no LDS round trip): row reduction by DPP, then row_bcast:15 /
// row_bcast:31 carry the row totals upward so that lane 63 holds the wave total; v_readlane makes it wave-uniform.
__device__ __forceinline__ float wave_sum_uniform(float v) {
    v += dpp_xor1(v);
    v += dpp_xor2(v);
    v += dpp_half_mirror(v);
    v += dpp_mirror(v);
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x142, 0xA, 0xF, false));  // row_bcast:15 -> rows 1,3
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x143, 0xC, 0xF, false));  // row_bcast:31 -> rows 2,3
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

// sum over the 64 lanes of a wave, result in every lane
__device__ __forceinline__ float wave_sum(float v) {
    v += dpp_xor1(v);
    v += dpp_xor2(v);
    v += dpp_half_mirror(v);
    v += dpp_mirror(v);
    v += __shfl_xor(v, 16);
    v += __shfl_xor(v, 32);
    return v;
}


// sum over the 4 lane groups (lane ^ 16, lane ^ 32) on the VALU: v_permlane16_swap / v_permlane32_swap with both operands = v
// give {rows 0,0,2,2} / {rows 1,1,3,3} and {lo,lo} / {hi,hi}; (r0 + r1) + (r2 + r3) in every lane (what the two
// ds_bpermute-based __shfl_xor steps computed, without the LDS round trips).
__device__ __forceinline__ float groups_sum(float v) {
    const unsigned u = __builtin_bit_cast(unsigned, v);
    auto a = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    const float s = __builtin_bit_cast(float, (unsigned)a[0]) + __builtin_bit_cast(float, (unsigned)a[1]);
    const unsigned w = __builtin_bit_cast(unsigned, s);
    auto b = __builtin_amdgcn_permlane32_swap(w, w, false, false);
    return __builtin_bit_cast(float, (unsigned)b[0]) + __builtin_bit_cast(float, (unsigned)b[1]);
}


// ---- lane = unit / register = env building blocks of the 16-block 4x4x1 MFMA kernels (rollout_q4_kernel, dqn_act_q4_kernel) ----
typedef float rq_f32x4 __attribute__((ext_vector_type(4)));
// 4x4 transpose inside every quad of lanes: in: lane 4b + q holds v[e] = M[q][e]; out: lane 4b + q holds v[e] = M[e][q]
__device__ __forceinline__ rq_f32x4 quad_transpose(rq_f32x4 v, bool b0, bool b1) {
    const float x0 = dpp_xor1(b0 ? v[0] : v[1]), x1 = dpp_xor1(b0 ? v[2] : v[3]);
    if (b0) { v[0] = x0; v[2] = x1; } else { v[1] = x0; v[3] = x1; }
    const float x2 = dpp_xor2(b1 ? v[0] : v[2]), x3 = dpp_xor2(b1 ? v[1] : v[3]);
    if (b1) { v[0] = x2; v[1] = x3; } else { v[2] = x2; v[3] = x3; }
    return v;
}

__device__ __forceinline__ float dpp_row_ror4(float v) { return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xF, 0xF, true)); }
__device__ __forceinline__ float dpp_row_ror8(float v) { return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xF, 0xF, true)); }

// p[e] = this lane's contribution to env e's total (e = 0..3) -> the sum over all 64 lanes for env (lane & 3), in every lane.
// Fixed order: quad butterfly (lane ^ 1, lane ^ 2), the row's 4 quads (ror 4, ror 8), the 4 rows (permlane16 / 32 swaps).
__device__ __forceinline__ float quad_env_reduce(const rq_f32x4& p, bool b0, bool b1) {
    const float r01 = (b0 ? p[1] : p[0]) + dpp_xor1(b0 ? p[0] : p[1]);
    const float r23 = (b0 ? p[3] : p[2]) + dpp_xor1(b0 ? p[2] : p[3]);
    float r = (b1 ? r23 : r01) + dpp_xor2(b1 ? r01 : r23);
    r += dpp_row_ror4(r);
    r += dpp_row_ror8(r);
    return groups_sum(r);
}

// wave-private LDS traffic: LDS executes one wave's instructions in order, so only the COMPILER must be kept
// from moving accesses across this point.
__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}
