// mi_sac.hip — the SAC hot path of reference deep_rl/sac.py (re-targeted to Pendulum-v1) on the device:
//   pend_*_kernel            Pendulum-v1 reset / step (gym 0.21 pendulum.py + TimeLimit 200 + episode statistics)
//   sac_act_kernel           sac.py:138-158: uniform random action before learning_starts, else actor.get_action; step; ring store
//   sac_critic_kernel        sac.py:165-185: actor on next obs, two target critics, TD target, two critics forward + backward
//   sac_actor_kernel         sac.py:193-197: actor forward (rsample), min(Q1,Q2) forward, d(-minQ)/d action, actor backward;
//                            logp_only: sac.py:203-204 (+ the alpha step by the last workgroup to finish)
//   sac_dw2_adam_kernel      batches <= 512 rows: the 256x256 weight gradients dW2 = dZ2^T H1 on v_mfma_f32_16x16x4_f32, the thin gradients' slab sums and the
//                            optimizer step (+ polyak) of every element in ONE launch
//   sac_dw2_gemm_kernel      larger batches: dW2 as split-K batch GEMMs, followed by
//   sac_grad_reduce_kernel   the fixed-order assembly of slabs and K-split partials (+ Adam + polyak in the single-process fusion)
//   adam_kernel / polyak_kernel / sac_alpha_kernel   the unfused pieces (sharded runs all-reduce between them)
// Row-group layout: a 256-thread workgroup owns SR = 16 batch rows; 256x256 layers on the f32 MFMA with weights streamed from L2
// into registers as one continuous stream per kernel, activations in LDS; see "row-group building blocks" below and DESIGN.md §7c.
// All reductions are fixed-order (bitwise reproducible).
#include "mi_common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define SA_H 256
#define SQ_W1 0
#define SQ_B1 1024
#define SQ_W2 1280
#define SQ_B2 66816
#define SQ_W3 67072
#define SQ_B3 67328
#define SQ_NP 67329
#define AC_W1 0
#define AC_B1 768
#define AC_W2 1024
#define AC_B2 66560
#define AC_WM 66816
#define AC_BM 67072
#define AC_WL 67073
#define AC_BL 67329
#define AC_NP 67330
#define SA_LOG_STD_MAX 2.0f
#define SA_LOG_STD_MIN -5.0f
#define SA_ACT_SCALE 2.0f
#define SA_ACT_BIAS 0.0f
#define STREAM_NORMAL 5u
#define STREAM_UNIF_ACT 6u
#define SR 16  // rows per workgroup

// ================================================ Pendulum env ==================================================================
__global__ void __launch_bounds__(256) pend_reset_kernel(mi_env e, float* __restrict__ obs, const double* __restrict__ forced) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= e.n) return;
    double th, thd;
    if (forced) { th = forced[2 * (size_t)i]; thd = forced[2 * (size_t)i + 1]; }
    else mi_pendulum_reset_noise(e.seed, e.env_id_base + (uint64_t)i, e.episode[i], th, thd);
    e.episode[i] += 1; e.elapsed[i] = 0; e.ep_ret[i] = 0.0f; e.ep_len[i] = 0;
    e.x[i] = th; e.x_dot[i] = thd;
    double sn, cs;
    mi_sincos_full(th, sn, cs);
    obs[3 * (size_t)i] = (float)cs; obs[3 * (size_t)i + 1] = (float)sn; obs[3 * (size_t)i + 2] = (float)thd;
}

int mi_pend_reset_impl(mi_env* e, float* obs, const double* forced_state, hipStream_t s) {
    pend_reset_kernel<<<(e->n + 255) / 256, 256, 0, s>>>(*e, obs, forced_state);
    MI_LAUNCH_CHECK();
    return MI_OK;
}

// one env: step + TimeLimit + episode statistics + auto-reset; returns the observation of the (possibly reset) state
struct pend_out { float o0, o1, o2, reward; bool done; float fin_ret; int fin_len; };
__device__ __forceinline__ pend_out pend_step_one(const mi_env& e, int i, float action, const double* forced_reset2, double& th, double& thd,
                                                  int& elapsed, float& epret, int& eplen, uint64_t& episode) {
    pend_out r;
    double rw;
    mi_pendulum_step(th, thd, action, rw);
    r.reward = (float)rw;
    elapsed += 1;
    r.done = elapsed >= PEND_MAX_STEPS;
    epret += (float)rw; eplen += 1;
    r.fin_ret = 0.0f; r.fin_len = 0;
    if (r.done) {
        r.fin_ret = epret; r.fin_len = eplen;
        epret = 0.0f; eplen = 0; elapsed = 0;
        if (forced_reset2) { th = forced_reset2[0]; thd = forced_reset2[1]; }
        else mi_pendulum_reset_noise(e.seed, e.env_id_base + (uint64_t)i, episode, th, thd);
        episode += 1;
    }
    double sn, cs;
    mi_sincos_full(th, sn, cs);
    r.o0 = (float)cs; r.o1 = (float)sn; r.o2 = (float)thd;
    return r;
}

__global__ void __launch_bounds__(256)
pend_step_kernel(mi_env e, const float* __restrict__ actions, const double* __restrict__ forced_reset, float* __restrict__ obs,
                 float* __restrict__ reward, uint8_t* __restrict__ done, uint8_t* __restrict__ truncated, float* __restrict__ fin_ret,
                 int32_t* __restrict__ fin_len, float* __restrict__ raw_obs) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= e.n) return;
    double th = e.x[i], thd = e.x_dot[i];
    if (raw_obs) {   // what gym's env.step returned, before sac.py:142-144's reset (trace dumps only: the step is simply evaluated twice)
        double rth = th, rthd = thd, rrw, sn, cs;
        mi_pendulum_step(rth, rthd, actions[i], rrw);
        mi_sincos_full(rth, sn, cs);
        raw_obs[3 * (size_t)i] = (float)cs; raw_obs[3 * (size_t)i + 1] = (float)sn; raw_obs[3 * (size_t)i + 2] = (float)rthd;
    }
    int elapsed = e.elapsed[i], eplen = e.ep_len[i];
    float epret = e.ep_ret[i];
    uint64_t episode = e.episode[i];
    const pend_out r = pend_step_one(e, i, actions[i], forced_reset ? forced_reset + 2 * (size_t)i : nullptr, th, thd, elapsed, epret, eplen, episode);
    e.x[i] = th; e.x_dot[i] = thd; e.elapsed[i] = elapsed; e.ep_len[i] = eplen; e.ep_ret[i] = epret; e.episode[i] = episode;
    obs[3 * (size_t)i] = r.o0; obs[3 * (size_t)i + 1] = r.o1; obs[3 * (size_t)i + 2] = r.o2;
    reward[i] = r.reward; done[i] = r.done; truncated[i] = r.done;  // Pendulum only ever ends by TimeLimit
    fin_ret[i] = r.fin_ret; fin_len[i] = r.fin_len;
}

extern "C" int mi_env_step_cont(void* handle, const float* actions, const double* forced_reset, float* obs, float* reward, uint8_t* done,
                                uint8_t* truncated, float* fin_ret, int32_t* fin_len, void* stream) {
    MI_CHECK_ARG(handle && actions && obs && reward && done && truncated && fin_ret && fin_len, "NULL pointer");
    mi_env* e = (mi_env*)handle;
    MI_CHECK_ARG(e->kind == MI_ENV_PENDULUM_V1, "continuous-action step on a discrete-action env");
    pend_step_kernel<<<(e->n + 255) / 256, 256, 0, (hipStream_t)stream>>>(*e, actions, forced_reset, obs, reward, done, truncated, fin_ret, fin_len, nullptr);
    MI_LAUNCH_CHECK();
    return MI_OK;
}
int mi_pend_step_ex_impl(mi_env* e, const float* actions, const double* forced_reset, float* obs, float* reward, uint8_t* done, uint8_t* truncated, float* fin_ret,
                         int32_t* fin_len, float* raw_obs, hipStream_t s) {
    pend_step_kernel<<<(e->n + 255) / 256, 256, 0, s>>>(*e, actions, forced_reset, obs, reward, done, truncated, fin_ret, fin_len, raw_obs);
    MI_LAUNCH_CHECK();
    return MI_OK;
}

// ================================================ row-group building blocks =====================================================
// A 256-thread workgroup (4 waves) owns SR = 16 batch rows.  The 256 x 256 layers run on v_mfma_f32_16x16x4_f32 with
//   A = the weight tile, streamed from L2 straight into registers (each weight element is needed by exactly one wave),
//   B = the row group's activations, read from LDS ([row][unit], row stride LDW),
//   D = [unit][row]: lane (row = lane & 15, g = lane >> 4) holds units 64 w + 16 t + 4 g + r of its row in acc[t][r],
// wave w owning 64 output units.  The reduction index is consumed in the permuted order the vector loads deliver (k-step s of a
// 16-wide chunk pairs element s of lane group g, i.e. index 4 g + s, on both operands).
// A kernel is a fixed sequence of such passes (critic update: 7, actor update: 6), and its weights are ONE stream: a 4-deep ring of
// 32-index stages, loads issued 3 stages (~1.3 us of MFMA work) ahead; the last 3 stages of a pass issue the first 3 of the NEXT
// pass's matrix, so only the first pass of a kernel waits for L2.  Thin parameters (layer-1 rows, biases, head weights) are fetched
// one pass ahead as one coalesced value per thread and handed to the D layout through LDS.  Thin layers (3/4 -> 256, heads, bias
// and thin weight gradients) are thread-per-unit VALU code on the same LDS images.
#define LDW 260
// Waves per row-group workgroup (round 4, measured: profiles/r04_sac_waves.txt).  TWO waves per SIMD, each owning two of the sixteen 16-unit output tiles of a pass, let one
// wave's MFMAs cover the other's weight-stream waits — which pays on a kernel that is ONE pass long (acting: 15.0 -> 13.6 us: its stream starts cold) and not on the
// multi-pass update kernels, whose later passes are requested three stages ahead anyway while every one of their ~20 barriers gets dearer with eight waves (critic 31.1 ->
// 31.4 us, actor 37.0 -> 38.3 us).  So: 8 for the single-pass kernels, 4 for the update kernels (the build switches for the other two forms left the source in round 5).
#define SA_WAVES 4
#define SA_WAVES_ACT 8
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)
#define L2_STAGES 8       // 256 reduction indices in stages of 32
#define L2_NBUF 4         // ring of stage buffers (8 stages = 0 mod 4: every pass starts at slot 0)
#define L2_AHEAD 3        // stages in flight ahead of the MFMAs
// keyed standard normal (production mode): Box-Muller on two Philox words
__device__ __forceinline__ float keyed_normal(uint64_t seed, uint64_t tag_update, uint64_t row) {
    uint32_t r[4];
    mi_philox(seed, tag_update, row, STREAM_NORMAL, r);
    const float u1 = ((float)(r[0] >> 8) + 0.5f) * (1.0f / 16777216.0f), u2 = (float)(r[1] >> 8) * (1.0f / 16777216.0f);
    return sqrtf(-2.0f * logf(u1)) * cosf(6.28318530717958647692f * u2);
}

#ifdef SAC_MARKS   // diagnostic build (tools/sac_marks.py): besides the phase marks below, a fine trace of the building blocks — thread 0 of a workgroup whose smem carries a trace slot
                   // appends {tag, s_memrealtime} at the end of every building block (layer 1, matrix pass, bias + head partial, cross-wave combine, row scalars)
__device__ unsigned long long sac_fine_dbg[4][64][2];
#define SAC_FINE_FIELDS int fine_slot, fine_n;
#define SAC_FINE(sm, tag) do { if (threadIdx.x == 0 && (unsigned)(sm).fine_slot < 4u && (unsigned)(sm).fine_n < 64u) {   /* (kernels that never set the slot hold garbage there: bounds only) */ \
                                unsigned long long rt_; asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(rt_) :: "memory"); \
                                sac_fine_dbg[(sm).fine_slot][(sm).fine_n][0] = (tag); sac_fine_dbg[(sm).fine_slot][(sm).fine_n][1] = rt_; (sm).fine_n = (sm).fine_n + 1; } } while (0)
#define SAC_FINE_INIT(sm, slot) do { if (threadIdx.x == 0) { (sm).fine_slot = (slot); (sm).fine_n = 0; } } while (0)
extern "C" int mi_debug_sac_fine(unsigned long long* out) { return hipMemcpyFromSymbol(out, HIP_SYMBOL(sac_fine_dbg), sizeof(sac_fine_dbg)) == hipSuccess ? 0 : -2; }
#else
#define SAC_FINE_FIELDS
#define SAC_FINE(sm, tag) do {} while (0)
#define SAC_FINE_INIT(sm, slot) do {} while (0)
#endif
#ifndef SAC_FWD_AS_BWD
#define SAC_FWD_AS_BWD 0   // TIMING-ONLY experiment (wrong results): forward passes fetch their stages with the backward pass's access pattern (what a transposed copy of the matrix would allow)
#endif

namespace rg_act {
#define SA_W SA_WAVES_ACT
#include "mi_sac_rowgroup.inc"
#undef SA_W
}  // namespace rg_act
namespace rg {
#define SA_W SA_WAVES
#include "mi_sac_rowgroup.inc"
#undef SA_W
}  // namespace rg
using namespace rg;   // everything below is the 4-wave form unless it sits in namespace rg_act

// ================================================ forward-only API kernels ======================================================
namespace rg_act {
__global__ void __launch_bounds__(SA_THREADS) sac_actor_sample_kernel(const float* __restrict__ actor, const float* __restrict__ obs, const float* __restrict__ eps,
                                                                int n, float* __restrict__ action, float* __restrict__ logp) {
    __shared__ sac_smem sm;
    const int row0 = blockIdx.x * SR;
    wstream ws; thin_t th; f32x4 acc[SA_NT];
    issue_thin_actor(actor, th);
    stream_prime<false>(actor + AC_W2, ws);
    if (threadIdx.x < SR * 3) { const int r = threadIdx.x / 3, k = threadIdx.x % 3; const int b = row0 + r < n ? row0 + r : n - 1; sm.x[r][k] = obs[3 * (size_t)b + k]; }
    float e = 0.0f;
    if (threadIdx.x < SR) e = eps[row0 + threadIdx.x < n ? row0 + threadIdx.x : n - 1];
    __syncthreads();
    layer1<3>(sm, th, sm.x, sm.b0);
    __syncthreads();
    actor_forward2<false>(sm, actor, actor + AC_W2, nullptr, sm.b0, ws, acc, e);
    if (threadIdx.x < SR && row0 + threadIdx.x < n) { action[row0 + threadIdx.x] = sm.rv[threadIdx.x][6]; if (logp) logp[row0 + threadIdx.x] = sm.rv[threadIdx.x][5]; }
}
}  // namespace rg_act

extern "C" int mi_sac_actor_sample(const float* actor, const float* obs, const float* eps, int n, float* action, float* logp, void* stream) {
    MI_CHECK_ARG(actor && obs && eps && action && n > 0, "bad arguments");
    rg_act::sac_actor_sample_kernel<<<(n + SR - 1) / SR, rg_act::SA_THREADS, 0, (hipStream_t)stream>>>(actor, obs, eps, n, action, logp);
    MI_LAUNCH_CHECK();
    return MI_OK;
}

#ifdef SAC_STAMPS   // diagnostics build: out[0..] of workgroup 0 receives phase durations in units of 10 ns (wall_clock64), results are destroyed
#define STAMP(k) do { __builtin_amdgcn_s_waitcnt(0); if (blockIdx.x == 0 && threadIdx.x == 0) stamp[k] = wall_clock64(); } while (0)
#else
#define STAMP(k)
#endif
__global__ void __launch_bounds__(SA_THREADS) sac_q_forward_kernel(const float* __restrict__ q, const float* __restrict__ obs, const float* __restrict__ act, int n,
                                                             float* __restrict__ out) {
    __shared__ sac_smem sm;
#ifdef SAC_STAMPS
    __shared__ unsigned long long stamp[8];
#endif
    const int row0 = blockIdx.x * SR;
    wstream ws; thin_t th; f32x4 acc[SA_NT];
    STAMP(0);
    issue_thin_q(q, th);
    stream_prime<false>(q + SQ_W2, ws);
    if (threadIdx.x < SR * 4) {
        const int r = threadIdx.x / 4, k = threadIdx.x & 3; const int b = row0 + r < n ? row0 + r : n - 1;
        sm.x[r][k] = k < 3 ? obs[3 * (size_t)b + k] : act[b];
    }
    __syncthreads();
    STAMP(1);
    layer1<4>(sm, th, sm.x, sm.b0);
    __syncthreads();
    STAMP(2);
#ifdef SAC_STAMPS
#ifndef SAC_STAMP_BWD
#define SAC_STAMP_BWD false
#endif
    f32x4 accx[SA_NT];
    mfma_pass<false, SAC_STAMP_BWD>(q + SQ_W2, q + SQ_W2, sm.b0, ws, accx);     // first touch of the matrix in this kernel
    STAMP(6);
    mfma_pass<SAC_STAMP_BWD, false>(q + SQ_W2, nullptr, sm.b0, ws, acc);
    for (int t = 0; t < SA_NT; ++t) acc[t] += accx[t];
#else
    mfma_pass<false, false>(q + SQ_W2, nullptr, sm.b0, ws, acc);
#endif
    STAMP(3);
    relu_bias(sm, acc);
    const float hp = head_partial(acc, sm.pb[1]);
    STAMP(4);
    rows_combine2(sm, hp, 0.0f, 8, 15);
    if (threadIdx.x < SR) sm.rv[threadIdx.x][8] += q[SQ_B3];
    __syncthreads();
    STAMP(5);
    if (threadIdx.x < SR && row0 + threadIdx.x < n) out[row0 + threadIdx.x] = sm.rv[threadIdx.x][8];
#ifdef SAC_STAMPS
    __syncthreads();
    if (blockIdx.x == 0 && threadIdx.x < 5) out[threadIdx.x] = (float)(stamp[threadIdx.x + 1] - stamp[threadIdx.x]);
    if (blockIdx.x == 0 && threadIdx.x == 2) { out[2] = (float)(stamp[6] - stamp[2]); out[5] = (float)(stamp[3] - stamp[6]); }
#endif
}

extern "C" int mi_sac_q_forward(const float* q, const float* obs, const float* act, int n, float* out, void* stream) {
    MI_CHECK_ARG(q && obs && act && out && n > 0, "bad arguments");
    sac_q_forward_kernel<<<(n + SR - 1) / SR, SA_THREADS, 0, (hipStream_t)stream>>>(q, obs, act, n, out);
    MI_LAUNCH_CHECK();
    return MI_OK;
}

// ================================================ workspace layout ==============================================================
// Kp = batch rounded up to a row group.  [H1 mats: 3 x Kp x 256][DZ2 mats: 3 x Kp x 256][slabs: nblocks x SLAB][GEMM partials: GEMM_MAX_SPLIT x 3 x 65536][ticket 4]
// [hand-off between the sibling workgroups of a row group: Kp x 6 tagged 64-bit words (xw_put / xw_take; self-resetting; both kernels use the same words)]
// [owed alpha step: observation stash, two slots of Kp x 3 (the actor update that creates a debt writes the slot the pending debt does NOT read)][Kp / SR log-prob partials][epoch]
// The caller zero-fills the workspace once (the ticket word resets itself after every use).
// (mats 0,1: critics; 2: actor)
#define SLAB 3600
#define GEMM_MAX_SPLIT 16
#define SAC_FUSED_KP 512   // up to this (padded) batch the dW2 GEMM, the gradient assembly and the optimizer step are ONE launch (sac_dw2_adam_kernel)
__host__ __device__ inline int ws_kp(int batch) { return (batch + SR - 1) / SR * SR; }
__host__ __device__ inline size_t ws_mat_floats(int batch) { return (size_t)ws_kp(batch) * SA_H; }
__host__ __device__ inline size_t ws_slab_off(int batch) { return 6 * ws_mat_floats(batch); }
__host__ __device__ inline size_t ws_part_off(int batch) { return ws_slab_off(batch) + (size_t)(ws_kp(batch) / SR) * SLAB; }
__host__ __device__ inline size_t ws_xch_off(int batch) { return ws_part_off(batch) + (size_t)GEMM_MAX_SPLIT * 3 * SA_H * SA_H + 4; }
__host__ __device__ inline size_t ws_stash_off(int batch) { return ws_xch_off(batch) + 12 * (size_t)ws_kp(batch); }   // [2][Kp x 3 obs][Kp / SR partials][epoch]
__host__ __device__ inline size_t ws_lp_off(int batch) { return ws_stash_off(batch) + 6 * (size_t)ws_kp(batch); }
__host__ __device__ inline size_t ws_epoch_off(int batch) { return ws_lp_off(batch) + (size_t)(ws_kp(batch) / SR); }
extern "C" size_t mi_sac_workspace_bytes(int batch) {
    if (batch <= 0) return 0;
    return (ws_epoch_off(batch) + 4 /* epoch word, padded */) * sizeof(float);
}
static int gemm_split(int batch) {   // each wave of a GEMM workgroup reduces >= 32 batch rows, the grid covers the rest
    int s = ws_kp(batch) / 128;
    return s < 1 ? 1 : (s > GEMM_MAX_SPLIT ? GEMM_MAX_SPLIT : s);
}

// ---- waits between workgroups of ONE launch (include/mi_rl.h "mi_sac_check") ----------------------------------------------------------------------------------
// Two rules make them safe.  (1) ORDER: a workgroup only ever waits for workgroups with a LOWER linear id (blockIdx.y * gridDim.x + blockIdx.x) — the owed-alpha
// workgroups sit at the lowest blockIdx.x of row y = 0, publishing siblings in lower rows y than their readers — and no workgroup that others wait for ever waits for
// a higher id.  The dispatcher hands workgroups out in that order, so whoever a resident reader waits for has been dispatched and runs to completion on its own: no
// co-residency of the whole grid is needed (a CU mask, another process or stream on the chip only make it slower).  (2) BOUND: HIP promises no dispatch order, so
// every spin also has a wall-clock budget (s_memrealtime, 100 MHz); when it runs out the waiter records a code in the process's status word (host-pinned, read by
// the next mi_sac_* call without a sync), takes NaN as the value — which poisons the launch's gradients and losses — and goes on to the end of the kernel: a failure
// is a negative return code, never a hang.  The same code goes into a DEVICE word (sac_fault_word) that every optimizer step of this file reads first: once it is
// set, no launch steps parameters, Adam moments, targets or log_alpha any more (one spurious timeout used to overwrite all of them with NaN — recovery
// needed a checkpoint the caller may not have).  The caller gets MI_ESTATE with its state intact and may mi_sac_clear_error and go on (the faulted update is lost).
#define SAC_SPIN_TICKS 10000000ull          // 100 ms
enum { SAC_FAULT_XW = 1, SAC_FAULT_EPOCH = 2 };
__device__ unsigned int* sac_status_word;   // device pointer of the host-pinned status word (set per device by sac_status_init)
__device__ unsigned int sac_fault_word;     // != 0: a wait of an earlier launch on this device timed out; cleared by mi_sac_clear_error
__device__ __noinline__ void sac_timeout(unsigned code) {
    __hip_atomic_store(&sac_fault_word, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned int* p = sac_status_word;
    if (p) __hip_atomic_store(p, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
// set by a PREVIOUS launch of the stream (the row-group kernels wait, the optimizer steps run in the launches behind them), or by this launch's own waiters in the
// kernels that both wait and step log_alpha: a plain load suffices for the former, and the latter's steppers do not depend on any hand-off
__device__ __forceinline__ bool sac_faulted() { return __hip_atomic_load(&sac_fault_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u; }
// hand-off of one float per row between sibling workgroups: a 64-bit word = (tag 1 << 32) | value bits, written and read with device-scope atomics, so the
// value travels WITH its "ready" mark (no flag -> fence -> payload sequence: one round trip less); the reader zeroes the word, the next launch finds 0.
__device__ __forceinline__ void xw_put(unsigned long long* w, float v) {
    __hip_atomic_store(w, (1ull << 32) | (unsigned long long)__builtin_bit_cast(unsigned, v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float xw_take(unsigned long long* w) {
    unsigned long long v = __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if ((v >> 32) != 1ull) {
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        do {
            __builtin_amdgcn_s_sleep(2);
            if (__builtin_amdgcn_s_memrealtime() - t0 > SAC_SPIN_TICKS) {
                sac_timeout(SAC_FAULT_XW);
                __hip_atomic_store(w, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // leave the word as a completed hand-off would (a late producer may still fill it: mi_sac_clear_error zeroes all)
                return __builtin_nanf("");
            }
            v = __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } while ((v >> 32) != 1ull);
    }
    __hip_atomic_store(w, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return __builtin_bit_cast(float, (unsigned)v);
}

// alpha step riding on the log-prob launch: the LAST workgroup to finish (ticket in the workspace) sums the slabs in fixed order and does the Adam step
struct sac_alpha_t { float* log_alpha; float* m; float* v; float* alpha; float* out; unsigned int* ticket; float target_entropy, inv_count, w1, b2, w2, step_size, rbc2, eps; };
__device__ __forceinline__ void sac_alpha_apply(const sac_alpha_t& a, float mean_lp) {
    if (sac_faulted()) return;                                 // a timed-out launch on this device: no optimizer state is stepped until mi_sac_clear_error
    const float la = a.log_alpha[0];
    const float g = -(mean_lp + a.target_entropy);            // d/d log_alpha of mean(-log_alpha * (logp + target_entropy)), sac.py:205
    if (a.out) { a.out[0] = -la * (mean_lp + a.target_entropy); a.out[1] = g; }
    float mi = a.m[0], vi = a.v[0];
    const float nla = mi_adam_elem(la, g, mi, vi, a.w1, a.b2, a.w2, a.step_size, a.rbc2, a.eps);
    a.m[0] = mi; a.v[0] = vi;
    a.log_alpha[0] = nla;
    a.alpha[0] = expf(nla);                                    // :210
}

// ---- the owed alpha step (sac.py:199-210).  Its log-prob pass depends only on the actor (after the last actor update) and on that update's batch observations;
// as a launch of its own it costs 12.6 us for 16 workgroups' worth of one matrix pass.  Instead the NEXT row-group launch of the stream (the following actor update,
// or the next iteration's critic update) carries n_lp extra workgroups that evaluate it from the observations the actor update stashed in the workspace; the last
// of them to finish does the Adam step on log_alpha and publishes the step number in the epoch word; the host launch's own workgroups read alpha only after the
// epoch word has reached the debt's epoch (they need alpha late: after their forward passes).  The epoch is the CALLER's counter of owed steps handed to this
// workspace — strictly increasing, independent of the Adam step number (which a checkpoint load may rewind) — compared wrap-safely, so the word never needs a reset.
// fault: test hook (mi_sac_test_fault): bit 0 = the publishing siblings skip their xw_put, bit 1 = the owed role does not publish its epoch.
struct sac_owed_t { int n_lp, epoch, slot, fault; uint64_t seed, update; sac_alpha_t al; };
__device__ __forceinline__ int* ws_epoch(float* ws_, int batch) { return reinterpret_cast<int*>(ws_ + ws_epoch_off(batch)); }
__device__ __forceinline__ float wait_owed_alpha(const sac_owed_t& ow, float* ws_, int batch, const float* alpha_p, sac_smem& sm) {
    if (ow.n_lp) {   // block-uniform
        if (threadIdx.x == 0) {
            int* ep = ws_epoch(ws_, batch);
            int bad = 0;
            if (__hip_atomic_load(ep, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) - ow.epoch < 0) {
                const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
                do {
                    __builtin_amdgcn_s_sleep(8);
                    if (__builtin_amdgcn_s_memrealtime() - t0 > SAC_SPIN_TICKS) { sac_timeout(SAC_FAULT_EPOCH); bad = 1; break; }
                } while (__hip_atomic_load(ep, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) - ow.epoch < 0);
            }
            sm.fault = bad;
        }
        __syncthreads();
        return sm.fault ? __builtin_nanf("") : __hip_atomic_load(alpha_p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    return alpha_p[0];
}
__device__ void sac_owed_alpha_role(sac_smem& sm, const float* __restrict__ actor, int batch, float* __restrict__ ws_, const sac_owed_t& ow, int rg) {
    const int t = threadIdx.x, row0 = rg * SR;
    const float* stash = ws_ + ws_stash_off(batch) + (size_t)ow.slot * 3 * (size_t)ws_kp(batch);   // the slot the actor update that owes this step wrote
    float* part = ws_ + ws_lp_off(batch);
    wstream ws; thin_t th; f32x4 acc[SA_NT];
    issue_thin_actor(actor, th);
    stream_prime<false>(actor + AC_W2, ws);
    if (t < SR * 3) { const int r = t / 3, k = t % 3; const int b = row0 + r < batch ? row0 + r : batch - 1; sm.x[r][k] = stash[3 * (size_t)b + k]; }
    float e_row = 0.0f;
    if (t < SR) { const int b = row0 + t < batch ? row0 + t : batch - 1; e_row = keyed_normal(ow.seed, (4ull << 40) + ow.update, (uint64_t)b); }
    __syncthreads();
    layer1<3>(sm, th, sm.x, sm.b0);
    __syncthreads();
    actor_forward2<false>(sm, actor, actor + AC_W2, nullptr, sm.b0, ws, acc, e_row);
    if (t == 0) { float s = 0.0f; for (int r = 0; r < SR; ++r) s += row0 + r < batch ? sm.rv[r][5] : 0.0f; part[rg] = s; }
    if (t == 0) { __threadfence(); sm.cur[0] = atomicAdd(ow.al.ticket, 1u) == (unsigned)ow.n_lp - 1 ? 1 : 0; }
    __syncthreads();
    if (!sm.cur[0] || t >= 64) return;
    __threadfence();
    const volatile float* vp = part;
    float s = 0.0f;
    for (int b = t; b < ow.n_lp; b += 64) s += vp[b];
    const float mean_lp = wave_sum(s) * ow.al.inv_count;
    if (t == 0) {
        sac_alpha_apply(ow.al, mean_lp);
        *ow.al.ticket = 0u;
        __threadfence();
        if (!(ow.fault & 2)) __hip_atomic_store(ws_epoch(ws_, batch), ow.epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
}

#ifdef SAC_MARKS   // diagnostic build: wall-clock marks (s_memrealtime, 100 MHz) of thread 0 of row group 0's workgroups at the phase boundaries of the two update kernels (tools/sac_marks.py)
__device__ unsigned long long sac_mark_dbg[2][4][16];   // [critic | actor kernel][role / sibling][mark]
#define SAC_MARK(kern, role, k) do { if (threadIdx.x == 0 && row0 == 0) { unsigned long long rt_; asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(rt_) :: "memory"); \
                                     sac_mark_dbg[kern][role][k] = rt_; } } while (0)
extern "C" int mi_debug_sac_marks(unsigned long long* out) { return hipMemcpyFromSymbol(out, HIP_SYMBOL(sac_mark_dbg), sizeof(sac_mark_dbg)) == hipSuccess ? 0 : -2; }
#else
#define SAC_MARK(kern, role, k) do {} while (0)
#endif

// ---- transposed copies of the layer-2 matrices (mi_sac_shadow_*, round 5): kernels instantiated with TR stream every FORWARD pass from them (q_forward2 / actor_forward2, FT) ----
struct sac_wt_t { const float* actor; const float* q; const float* qt; };   // actor^T [65536], critics^T [2][65536], targets^T [2][65536]; members a kernel does not stream may be null
#define SA_MAT (SA_H * SA_H)

// ================================================ critic update =================================================================
// slab layout (critic): net n at n*1793: W1 1024 | b1 256 | b2 256 | W3 256 | b3 1;  then [3586] = sum (q1-y)^2, [3587] = sum (q2-y)^2
// passes: actor fwd -> target 1 fwd -> target 2 fwd -> critic 1 fwd, bwd -> critic 2 fwd, bwd
// the TD target's ingredients arriving from the target workgroups of a four-workgroup row group (see sac_critic_kernel)
// y_only: the sibling already formed the TD target (two-workgroup row groups): ONE word per row
struct quad_wait_t { unsigned long long* xw; const float* rewards; const uint8_t* terminated; const float* alpha_p; float* ws_; float gamma; const sac_owed_t* ow; bool y_only; };
template <int NET, bool TR>
__device__ __forceinline__ void critic_net_update(sac_smem& sm, const float* __restrict__ q, const float* __restrict__ Wfwd /* this critic's layer-2 matrix as the forward pass streams it */,
                                                  const float* __restrict__ Wnext /* the matrix of the pass behind the backward (streamed as a forward pass: TR form if TR) */, wstream& ws, const thin_t& th,
                                                  int batch, int row0, float invn, float* __restrict__ H1, float* __restrict__ DZ2, float* __restrict__ slab,
                                                  const quad_wait_t* qw = nullptr) {
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6, li = lane & 15, lg = lane >> 4;
    const float* p = q + (size_t)NET * SQ_NP;
    float* sl = slab + NET * 1793;
    f32x4 acc[SA_NT];
    SAC_MARK(0, 2 + NET, 5);
    layer1<4>(sm, th, sm.x, sm.b0);
    __syncthreads();
    q_forward2<true, TR>(sm, p, Wfwd, p + SQ_W2, sm.b0, ws, acc, 8);          // h1 in b0, relu(h2) in acc; the stream continues with this net's backward
    SAC_MARK(0, 2 + NET, 6);
    // ---- the backward matrix pass with UNIT weight per row, BEFORE the TD target is needed (round 5).  d loss / d q of a row is one scalar dq, and everything behind the
    //      head is linear in it: dz2 = relu'(h2) w3 dq, dh1 = W2^T dz2 = dq (W2^T (relu'(h2) w3)).  A critic role of the quad form used to sit idle from the end of its
    //      forward (10.6 us after entry) until the target roles' words arrived (20.0 us: actor' forward, then the target forward), and only then ran this pass (4.1 us):
    //      now the pass runs in the wait and the arrival of the target is followed by a scaling (tools/sac_marks.py, profiles/r05_sac_marks.txt).  DZ2, db2, dW3 keep
    //      their bits (the same products, the same sums); dh1 — hence dW1 / db1 of the critics — is rounded after the sum instead of before it. ----
    store_acc(acc, sm.b2);                           // h2 image
#pragma unroll
    for (int tt = 0; tt < SA_NT; ++tt) {
        const f32x4 w3 = *reinterpret_cast<const f32x4*>(&sm.pb[1][16 * (SA_NT * wv + tt) + 4 * lg]);
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[tt][r] = acc[tt][r] > 0.0f ? w3[r] : 0.0f;
    }
    store_acc(acc, sm.b1);                           // relu'(h2) w3: dz2 per unit of dq
    __syncthreads();
    if (t < SA_H) {   // H1 rows for the dW2 GEMM: nothing of the target in them either
#pragma unroll
        for (int r = 0; r < SR; ++r) H1[((size_t)NET * ws_kp(batch) + row0 + r) * SA_H + t] = sm.b0[r][t];
    }
    SAC_MARK(0, 2 + NET, 8);
    mfma_pass<true, TR>(p + SQ_W2, Wnext, sm.b1, ws, acc);      // dh1 per unit of dq, in the D layout
    SAC_MARK(0, 2 + NET, 9);
#pragma unroll
    for (int tt = 0; tt < SA_NT; ++tt) {
        const f32x4 h1 = *reinterpret_cast<const f32x4*>(&sm.b0[li][16 * (SA_NT * wv + tt) + 4 * lg]);
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[tt][r] = h1[r] > 0.0f ? acc[tt][r] : 0.0f;
    }
    // ---- the TD target ----
    if (qw && qw->y_only) {
        if (t < SR) sm.rv[t][10] = xw_take(qw->xw + 6 * t);
        __syncthreads();
    } else if (qw) {   // y = r + gamma (1 - d) (min(Q1', Q2') - alpha log pi(a'|s')) (sac.py:176-177) from the target workgroups' words
        float rw = 0.0f, nd = 0.0f;
        if (t < SR) { rw = qw->rewards[sm.nxt[t]]; nd = qw->terminated[sm.nxt[t]] ? 0.0f : 1.0f; }
        const float alpha = wait_owed_alpha(*qw->ow, qw->ws_, batch, qw->alpha_p, sm);
        if (t < SR) {
            const float t1 = xw_take(qw->xw + 6 * t + NET), t2 = xw_take(qw->xw + 6 * t + 2 + NET), lp = xw_take(qw->xw + 6 * t + 4 + NET);
            sm.rv[t][10] = rw + nd * qw->gamma * (fminf(t1, t2) - alpha * lp);
        }
        __syncthreads();
    }
    SAC_MARK(0, 2 + NET, 7);
    if (t < SR) {
        const bool valid = row0 + t < batch;
        const float d = valid ? sm.rv[t][8] - sm.rv[t][10] : 0.0f;
        sm.rv[t][9] = d * d;           // loss contribution
        sm.rv[t][8] = 2.0f * d * invn; // d loss / d q
    }
    __syncthreads();
    {   // scale by the row's dq: the GEMM operand DZ2 (w3 dq for the live units: the product the pre-round-5 form stored) and dz1
        const float dq = sm.rv[li][8];
#pragma unroll
        for (int tt = 0; tt < SA_NT; ++tt) {
            const int u = 16 * (SA_NT * wv + tt) + 4 * lg;
            const f32x4 un = *reinterpret_cast<const f32x4*>(&sm.b1[li][u]);
            *reinterpret_cast<f32x4*>(&DZ2[((size_t)NET * ws_kp(batch) + row0 + li) * SA_H + u]) = f32x4{un[0] * dq, un[1] * dq, un[2] * dq, un[3] * dq};
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[tt][r] = acc[tt][r] * dq;
        }
    }
    if (t < SA_H) {   // unit j = t: thin gradients of layer 3 / bias 2
        float gw3 = 0.0f, gb2 = 0.0f;
#pragma unroll
        for (int r = 0; r < SR; ++r) { gw3 = __builtin_fmaf(sm.rv[r][8], sm.b2[r][t], gw3); gb2 += sm.b1[r][t] * sm.rv[r][8]; }
        sl[1024 + 256 + t] = gb2; sl[1024 + 512 + t] = gw3;
        if (t == 0) { float gb3 = 0.0f, l = 0.0f; for (int r = 0; r < SR; ++r) { gb3 += sm.rv[r][8]; l += sm.rv[r][9]; } sl[1792] = gb3; slab[3586 + NET] = l; }
    }
    __syncthreads();                                 // every thread is done reading b2 (h2)
    store_acc(acc, sm.b2);                           // dz1 image
    __syncthreads();
    if (t < SA_H) {   // input unit k = t: thin gradients of layer 1
        float gb1 = 0.0f, gw[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int r = 0; r < SR; ++r) {
            const float d = sm.b2[r][t];
            gb1 += d;
#pragma unroll
            for (int c = 0; c < 4; ++c) gw[c] = __builtin_fmaf(d, sm.x[r][c], gw[c]);
        }
        *reinterpret_cast<float4*>(sl + 4 * t) = make_float4(gw[0], gw[1], gw[2], gw[3]);
        sl[1024 + t] = gb1;
    }
    __syncthreads();
    SAC_MARK(0, 2 + NET, 10);
}

template <bool TR>
__global__ void __launch_bounds__(SA_THREADS)
sac_critic_kernel(const float* __restrict__ q, const float* __restrict__ qt, const float* __restrict__ actor, const float* __restrict__ observations,
                  const float* __restrict__ actions, const float* __restrict__ rewards, const uint8_t* __restrict__ terminated,
                  const int64_t* idx /* may alias idx_out: no __restrict__ */, int batch, int n_envs, long long slots, const float* __restrict__ eps, uint64_t seed,
                  uint64_t update, const float* __restrict__ alpha_p, float gamma, float invn, float* __restrict__ ws_, uint64_t sample_update, uint64_t sample_upper,
                  int64_t* idx_out, sac_owed_t ow, sac_wt_t wt) {
    __shared__ sac_smem sm;
    MI_INSIDE_SCOPE(MI_PROF_SAC_CRITIC);
    const float* const AW = TR ? wt.actor : actor + AC_W2;                                    // the matrices as this kernel's FORWARD passes stream them
    const float* const QW0 = TR ? wt.q : q + SQ_W2, * const QW1 = TR ? wt.q + SA_MAT : q + SQ_NP + SQ_W2;
    const float* const QTW0 = TR ? wt.qt : qt + SQ_W2, * const QTW1 = TR ? wt.qt + SA_MAT : qt + SQ_NP + SQ_W2;
    const int bx = (int)blockIdx.x - ow.n_lp;   // row group; the owed alpha step's workgroups come FIRST in dispatch order (blockIdx.x < n_lp of row y = 0): everyone who waits for alpha is behind them
    if (bx < 0) {
        if (blockIdx.y == 0) sac_owed_alpha_role(sm, actor, batch, ws_, ow, (int)blockIdx.x);
        return;
    }
    const int t = threadIdx.x, row0 = bx * SR;
    SAC_FINE_INIT(sm, bx == 0 && (blockIdx.y == 0 || blockIdx.y == 2) ? (blockIdx.y == 0 ? 0 : 1) : -1);   // target role 0, critic role 2
    SAC_MARK(0, (int)blockIdx.y & 3, 0);
    // gridDim.y == 2: critic 2's forward + backward run in a sibling workgroup y = 1, which takes the finished TD target from workgroup y = 0 (actor forward, both
    // targets, critic 1) — y = 0 never waits for y = 1, so the pair needs no co-residency: 5 passes on the critical path instead of 7, used while the row groups do
    // not fill the chip anyway.  (Round 2 let the two siblings evaluate one target each and swap them: 4 us faster at batch 288 - 768, but a symmetric rendezvous
    // hangs when the second workgroup cannot become resident — VERDICT r02 weak #7.)
    // gridDim.y == 4 (row groups <= 32): FOUR workgroups per row group.  y = 0 / 1 evaluate the next action and target critic 1 / 2 on the next observations and
    // publish (Q_y', and y = 0 also log pi(a'|s')) once per reader; y = 2 / 3 own critic 1 / 2: they run its forward on (obs, action) — which needs nothing from the
    // targets — at the same time, then take the targets' words, form the TD target and do loss + backward.  Critical path: max(actor fwd + target fwd, critic fwd)
    // + hand-off + backward instead of actor fwd + target fwd + hand-off + critic fwd + backward.
    const bool quad = gridDim.y == 4, split = gridDim.y == 2;
    const int role = quad ? (int)blockIdx.y : (split && blockIdx.y == 1 ? 3 : 0);   // split: y = 1 owns critic 2 like the quad's role 3
    const size_t matf = ws_mat_floats(batch);
    float* H1 = ws_; float* DZ2 = ws_ + 3 * matf;
    float* slab = ws_ + ws_slab_off(batch) + (size_t)bx * SLAB;
    wstream ws; thin_t th, th2; f32x4 acc[SA_NT];
    if (role >= 2) issue_thin_q(q + (role - 2) * SQ_NP, th); else
    issue_thin_actor(actor, th);
    // the batch rows: thread (r = t / 4, k = t & 3) derives row r's index itself (four threads repeat the draw: no LDS hand-over, no barrier) and requests its
    // element; the weight stream is started AFTER the rows are in LDS (loads return in order: the first pass cannot start before layer 1 has the rows anyway)
    float gx = 0.0f, gxn = 0.0f;
    if (t < SR * 4) {
        const int r = t >> 2, k = t & 3;
        const int b = row0 + r < batch ? row0 + r : batch - 1;
        long long i;
        if (sample_upper) {   // batch_inds = randint(global_step, size=batch) (sac.py:162) drawn here (the mi_dqn_sample contract, stream 4): no launch of its own
            uint32_t rr[4];
            mi_philox(seed, sample_update, (uint64_t)b, 4u, rr);
            i = (long long)((((uint64_t)rr[1] << 32) | rr[0]) % sample_upper);
            if (k == 0 && row0 + r < batch) idx_out[b] = i;     // (both workgroups of a split row group write the same value)
        } else i = idx[b];
        long long sl, en;   // next-slot row of the same env; flat indices below 2^32 take the 32-bit divider, the wrap is a compare
        if ((unsigned long long)i >> 32) { sl = i / n_envs; en = i % n_envs; }
        else { const unsigned qd = (unsigned)i / (unsigned)n_envs; sl = qd; en = (unsigned)i - qd * (unsigned)n_envs; }
        sl = sl + 1 == slots ? 0 : sl + 1;
        const long long nx = sl * n_envs + en;
        gx = k < 3 ? observations[3 * i + k] : actions[i];
        gxn = k < 3 ? observations[3 * nx + k] : 0.0f;
        if (k == 0) { sm.cur[r] = i; sm.nxt[r] = nx; }
    }
    // Order (round 5, from the ISA): the row noise is arithmetic and runs while the gathers are on their way; the rows go to LDS BEFORE the weight stream is primed.
    // With ~100 stream requests between the gathers and their LDS stores the wait in front of the stores could only be written as vmcnt(62) — it waited for the
    // gathers AND the first 35 stream requests — and the noise, computed behind the stream, landed in a register the allocator had also given to a stream request:
    // s_waitcnt vmcnt(0) in wave 0, i.e. the rows barrier waited for the whole primed stream (rows in LDS 2.0 - 2.5 us after entry).
    // (the noise is drawn by WAVE 1: Philox + Box-Muller are ~1,000 dependent cycles, as long as wave 0's Philox + 64-bit modulo in front of the gathers)
    if (t >= 64 && t < 64 + SR) { const int r = t - 64, b = row0 + r < batch ? row0 + r : batch - 1; sm.noise[r] = eps ? eps[b] : keyed_normal(seed, (2ull << 40) + update, (uint64_t)b); }
    if (t < SR * 4) { sm.x[t >> 2][t & 3] = gx; sm.xn[t >> 2][t & 3] = gxn; }
    stream_prime<TR>(role >= 2 ? (role == 2 ? QW0 : QW1) : AW, ws);
    __syncthreads();
    const float e_row = t < SR ? sm.noise[t] : 0.0f;
    SAC_MARK(0, (int)blockIdx.y & 3, 1);
    unsigned long long* xw = reinterpret_cast<unsigned long long*>(ws_ + ws_xch_off(batch)) + 6 * (size_t)row0;
    if (role >= 2) {   // critic role - 2 (quad), critic 2 fed the finished TD target (split)
        const quad_wait_t qw = {xw, rewards, terminated, alpha_p, ws_, gamma, &ow, split};
        if (role == 2) critic_net_update<0, TR>(sm, q, QW0, nullptr, ws, th, batch, row0, invn, H1, DZ2, slab, &qw);
        else critic_net_update<1, TR>(sm, q, QW1, nullptr, ws, th, batch, row0, invn, H1, DZ2, slab, &qw);
        return;
    }
    if (quad) {
        const float* qtn = qt + role * SQ_NP;   // target role
        layer1<3>(sm, th, sm.xn, sm.b0);
        issue_thin_q(qtn, th);
        __syncthreads();
        actor_forward2<TR, TR>(sm, actor, AW, role ? QTW1 : QTW0, sm.b0, ws, acc, e_row);
        SAC_MARK(0, role, 2);
        if (t < SR) { sm.xn[t][3] = sm.rv[t][6]; sm.rv[t][9] = sm.rv[t][5]; }   // a', log pi(a'|s')
        __syncthreads();
        layer1<4>(sm, th, sm.xn, sm.b0);
        __syncthreads();
        q_forward2<false, TR>(sm, qtn, role ? QTW1 : QTW0, nullptr, sm.b0, ws, acc, 8);
        SAC_MARK(0, role, 3);
        if (t < SR && !(ow.fault & 1)) {
            xw_put(xw + 6 * t + 2 * role, sm.rv[t][8]); xw_put(xw + 6 * t + 2 * role + 1, sm.rv[t][8]);
            if (role == 0) { xw_put(xw + 6 * t + 4, sm.rv[t][9]); xw_put(xw + 6 * t + 5, sm.rv[t][9]); }
        }
        SAC_MARK(0, role, 4);
        return;
    }
    // ---- next action + log-prob under the current actor (no grad; sac.py:172) ----
    layer1<3>(sm, th, sm.xn, sm.b0);
    issue_thin_q(qt, th);
    __syncthreads();
    actor_forward2<TR, TR>(sm, actor, AW, QTW0, sm.b0, ws, acc, e_row);
    if (t < SR) { sm.xn[t][3] = sm.rv[t][6]; sm.rv[t][9] = sm.rv[t][5]; }   // a', log pi(a'|s')
    __syncthreads();
    // ---- target critics (:173-174) ----
    layer1<4>(sm, th, sm.xn, sm.b0);
    issue_thin_q(qt + SQ_NP, th);
    __syncthreads();
    q_forward2<TR, TR>(sm, qt, QTW0, QTW1, sm.b0, ws, acc, 8);
    if (t < SR) sm.rv[t][10] = sm.rv[t][8];
    __syncthreads();
    layer1<4>(sm, th, sm.xn, sm.b0);
    issue_thin_q(q, th);
    if (!split) issue_thin_q(q + SQ_NP, th2);
    __syncthreads();
    q_forward2<TR, TR>(sm, qt + SQ_NP, QTW1, QW0, sm.b0, ws, acc, 8);
    const float alpha_now = wait_owed_alpha(ow, ws_, batch, alpha_p, sm);
    if (t < SR) {
        const float alpha = alpha_now;
        const float mq = fminf(sm.rv[t][10], sm.rv[t][8]) - alpha * sm.rv[t][9];                                   // :176
        sm.rv[t][10] = rewards[sm.nxt[t]] + (terminated[sm.nxt[t]] ? 0.0f : 1.0f) * gamma * mq;                     // :177  (y)
        if (split && !(ow.fault & 1)) xw_put(xw + 6 * t, sm.rv[t][10]);                                             // critic 2's workgroup (y = 1) is waiting for it
    }
    __syncthreads();
    // ---- the two critics on (obs, action): forward, loss, backward (:179-185) ----
    critic_net_update<0, TR>(sm, q, QW0, split ? nullptr : QW1, ws, th, batch, row0, invn, H1, DZ2, slab);
    if (!split) critic_net_update<1, TR>(sm, q, QW1, nullptr, ws, th2, batch, row0, invn, H1, DZ2, slab);
}

// ================================================ actor update ==================================================================
// slab layout (actor): W1 768 | b1 256 | b2 256 | Wm 256 | bm 1 | Wl 256 | bl 1 | [1794] sum(alpha*logp - minq) | [1795] sum logp
// passes: actor fwd -> critic 1 fwd -> critic 2 fwd -> critic 1 bwd -> critic 2 bwd -> actor bwd
// d(-min Q)/d action of one critic for the lane's row: this wave's share (summed over lane groups), from dh1 in the D layout
__device__ __forceinline__ float q_daction_partial(const sac_smem& sm, int net, const f32x4 acc[SA_NT]) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, i = lane & 15, g = lane >> 4;
    float v = 0.0f;
#pragma unroll
    for (int t = 0; t < SA_NT; ++t) {
        const int k = 16 * (SA_NT * w + t) + 4 * g;
        const f32x4 w13 = *reinterpret_cast<const f32x4*>(&sm.qw13[net][k]);
        const uint4 mk = *reinterpret_cast<const uint4*>(&sm.qmask[net][k]);
        const uint32_t m[4] = {mk.x, mk.y, mk.z, mk.w};
#pragma unroll
        for (int r = 0; r < 4; ++r) v += (m[r] >> i) & 1u ? w13[r] * acc[t][r] : 0.0f;     // the forward's own layer-1 ReLU mask
    }
    v += __shfl_xor(v, 16);
    v += __shfl_xor(v, 32);
    return v;
}

template <bool TR>
__global__ void __launch_bounds__(SA_THREADS)
sac_actor_kernel(const float* __restrict__ actor, const float* __restrict__ q, const float* __restrict__ observations, const int64_t* __restrict__ idx,
                 int batch, const float* __restrict__ eps, uint64_t seed, uint64_t update, const float* __restrict__ alpha_p, float invn,
                 float* __restrict__ ws_, int logp_only, sac_alpha_t al, sac_owed_t ow, int stash_slot, sac_wt_t wt) {
    __shared__ sac_smem sm;
    const float* const AW = TR ? wt.actor : actor + AC_W2;                                    // the matrices as this kernel's FORWARD passes stream them
    const float* const QW0 = TR ? wt.q : q + SQ_W2, * const QW1 = TR ? wt.q + SA_MAT : q + SQ_NP + SQ_W2;
    MI_INSIDE_SCOPE(logp_only ? MI_PROF_SAC_LOGP : MI_PROF_SAC_ACTOR);
    const int bx = (int)blockIdx.x - ow.n_lp;   // row group; the owed alpha step's workgroups come first in dispatch order (see sac_critic_kernel)
    if (bx < 0) {
        if (blockIdx.y == 0) sac_owed_alpha_role(sm, actor, batch, ws_, ow, (int)blockIdx.x);
        return;
    }
    const int t = threadIdx.x, row0 = bx * SR;
    SAC_FINE_INIT(sm, bx == 0 ? 2 + (int)(blockIdx.y & 1) : -1);   // 2: critic 2's sibling (y = 0), 3: main (y = 1)
    SAC_MARK(1, (int)blockIdx.y & 3, 0);
    const int lane = t & 63, wv = t >> 6, li = lane & 15, lg = lane >> 4;
    const size_t matf = ws_mat_floats(batch);
    float* H1 = ws_ + 2 * matf; float* DZ2 = ws_ + 5 * matf;
    float* slab = ws_ + ws_slab_off(batch) + (size_t)bx * SLAB;
    // gridDim.y == 2 (while the row groups do not fill the chip): critic 2's forward + backward run in the sibling workgroup y = 0 (dispatched FIRST: it waits for
    // nobody, hands (q2, d q2 / d action) per row over through the workspace and exits); the row group's main workgroup y = 1 runs critic 1 and goes on with the actor's
    // backward: 4 matrix passes + one hand-off on the critical path instead of 6.
    const bool split = gridDim.y == 2, second = split && blockIdx.y == 0;
    wstream ws; thin_t th; f32x4 acc[SA_NT];
    uint32_t h2mask[2];                                           // the critics' layer-2 ReLU masks in the D layout (bit 4 t + r), kept for the backward
    issue_thin_actor(actor, th);
    float gv = 0.0f;
    if (t < SR * 3) { const int r = t / 3, k = t % 3; const int b = row0 + r < batch ? row0 + r : batch - 1; gv = observations[3 * idx[b] + k]; }
    // (the noise: arithmetic, drawn by wave 1 while wave 0 waits for the index -> row requests)
    if (t >= 64 && t < 64 + SR) { const int r = t - 64, b = row0 + r < batch ? row0 + r : batch - 1; sm.noise[r] = eps ? eps[b] : keyed_normal(seed, ((logp_only ? 4ull : 3ull) << 40) + update, (uint64_t)b); }
    if (t < SR * 3) {
        const int r = t / 3, k = t % 3; const int b = row0 + r < batch ? row0 + r : batch - 1;
        sm.x[r][k] = gv;
        // kept for an alpha step that rides on a LATER launch, in the slot that an alpha step owed to THIS launch does not read (sac_owed_alpha_role reads ow.slot)
        if (!logp_only && !second && row0 + r < batch) ws_[ws_stash_off(batch) + (size_t)stash_slot * 3 * (size_t)ws_kp(batch) + 3 * (size_t)b + k] = gv;
    }
    stream_prime<TR>(AW, ws);   // (behind the rows' LDS stores: see sac_critic_kernel)
    __syncthreads();
    const float e_row = t < SR ? sm.noise[t] : 0.0f;
    SAC_MARK(1, (int)blockIdx.y & 3, 1);
    layer1<3>(sm, th, sm.x, sm.b0);                              // actor h1 -> b0 (kept for the backward)
    const float wm = th.h0, wl = th.h1;                          // this unit's head weights, kept for the actor's backward
    if (logp_only) {   // sac.py:203-204
        __syncthreads();
        actor_forward2<false, TR>(sm, actor, AW, nullptr, sm.b0, ws, acc, e_row);
        if (t == 0) { float s = 0.0f; for (int r = 0; r < SR; ++r) s += row0 + r < batch ? sm.rv[r][5] : 0.0f; slab[1795] = s; slab[1794] = 0.0f; }
        if (!al.ticket) return;
        if (t == 0) { __threadfence(); sm.cur[0] = atomicAdd(al.ticket, 1u) == gridDim.x - 1 ? 1 : 0; }
        __syncthreads();
        if (!sm.cur[0] || t >= 64) return;
        __threadfence();
        const volatile float* slabs = ws_ + ws_slab_off(batch);
        float s = 0.0f;
        for (int b = t; b < (int)gridDim.x; b += 64) s += slabs[(size_t)b * SLAB + 1795];
        const float mean_lp = wave_sum(s) * al.inv_count;
        if (t == 0) { sac_alpha_apply(al, mean_lp); *al.ticket = 0u; }
        return;
    }
    // split: the sibling repeats the actor forward (same bits).  The backward runs with unit weight per row; torch.min's routing (1 / 0 / one half on ties) scales
    // the result afterwards — exact, so both forms give the same bits.
    const float* qn = q + (second ? SQ_NP : 0);
    issue_thin_q(qn, th);
    __syncthreads();
    actor_forward2<TR, TR>(sm, actor, AW, second ? QW1 : QW0, sm.b0, ws, acc, e_row);
    SAC_MARK(1, (int)blockIdx.y & 3, 2);
    store_acc(acc, sm.b1);                                       // actor h2 image, kept until the actor's backward
    float alpha = 0.0f;                                          // read where it is first needed (an owed alpha step of this launch may still be producing it)
    if (t < SR) sm.x[t][3] = sm.rv[t][6];   // the action enters the critics
    __syncthreads();
    if (split) {
        const int net = second ? 1 : 0;
        unsigned long long* xw = reinterpret_cast<unsigned long long*>(ws_ + ws_xch_off(batch)) + 2 * (size_t)row0;
        const uint32_t mk = layer1<4>(sm, th, sm.x, sm.b2);
        if (t < SA_H) { sm.qmask[net][t] = mk; sm.qw13[net][t] = th.w1[3]; sm.qw3[net][t] = th.h0; }
        __syncthreads();
        q_forward2<true, TR>(sm, qn, second ? QW1 : QW0, qn + SQ_W2, sm.b2, ws, acc, 8);                // q_net(obs, pi(obs)) -> rv[8]; next pass: the same matrix, column-wise
        SAC_MARK(1, (int)blockIdx.y & 3, 3);
        uint32_t hm = 0;
#pragma unroll
        for (int tt = 0; tt < SA_NT; ++tt)
#pragma unroll
            for (int r = 0; r < 4; ++r) hm |= (acc[tt][r] > 0.0f ? 1u : 0u) << (4 * tt + r);
        if (t < SR) sm.rv[t][10] = sm.rv[t][8];                                  // this critic's q
        const float dq = row0 + li < batch ? -invn : 0.0f;                       // unit routing weight
#pragma unroll
        for (int tt = 0; tt < SA_NT; ++tt) {
            const f32x4 w3 = *reinterpret_cast<const f32x4*>(&sm.qw3[net][16 * (SA_NT * wv + tt) + 4 * lg]);
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[tt][r] = (hm >> (4 * tt + r)) & 1u ? w3[r] * dq : 0.0f;
        }
        __syncthreads();                                                         // the forward pass is done reading b2
        store_acc(acc, sm.b2);
        __syncthreads();
        if (second) mfma_pass<true, false>(qn + SQ_W2, nullptr, sm.b2, ws, acc);
        else mfma_pass<true, true>(qn + SQ_W2, actor + AC_W2, sm.b2, ws, acc);
        rows_combine2(sm, q_daction_partial(sm, net, acc), 0.0f, 11, 15);        // d q_net / d action (unit weight) -> rv[11]
        SAC_MARK(1, (int)blockIdx.y & 3, 4);
        if (second) {
            if (t < SR && !(ow.fault & 1)) { xw_put(xw + 2 * t, sm.rv[t][10]); xw_put(xw + 2 * t + 1, sm.rv[t][11]); }
            SAC_MARK(1, 0, 5);
            return;
        }
        alpha = wait_owed_alpha(ow, ws_, batch, alpha_p, sm);   // (contains a barrier when a step is owed)
        if (t < SR) {
            const float q1 = sm.rv[t][10], da1 = sm.rv[t][11];
            const float q2 = xw_take(xw + 2 * t), da2 = xw_take(xw + 2 * t + 1);
            const bool valid = row0 + t < batch;
            sm.rv[t][9] = valid ? alpha * sm.rv[t][5] - fminf(q1, q2) : 0.0f;                    // loss term (:197)
            const float w1 = !valid ? 0.0f : q1 < q2 ? 1.0f : (q2 < q1 ? 0.0f : 0.5f);             // torch.min routes the gradient to the smaller input, half / half on ties
            const float w2 = !valid ? 0.0f : q2 < q1 ? 1.0f : (q1 < q2 ? 0.0f : 0.5f);
            sm.rv[t][8] = w1 * da1; sm.rv[t][11] = w2 * da2;
        }
        __syncthreads();
        SAC_MARK(1, 1, 5);
    } else {
    // ---- min(Q1, Q2)(obs, pi(obs)) (:194-196) ----
    {
        const uint32_t mk = layer1<4>(sm, th, sm.x, sm.b2);
        if (t < SA_H) { sm.qmask[0][t] = mk; sm.qw13[0][t] = th.w1[3]; sm.qw3[0][t] = th.h0; }
        issue_thin_q(q + SQ_NP, th);
        __syncthreads();
        q_forward2<TR, TR>(sm, q, QW0, QW1, sm.b2, ws, acc, 8);
        h2mask[0] = 0;
#pragma unroll
        for (int tt = 0; tt < SA_NT; ++tt)
#pragma unroll
            for (int r = 0; r < 4; ++r) h2mask[0] |= (acc[tt][r] > 0.0f ? 1u : 0u) << (4 * tt + r);
        if (t < SR) sm.rv[t][10] = sm.rv[t][8];
        __syncthreads();
    }
    {
        const uint32_t mk = layer1<4>(sm, th, sm.x, sm.b2);
        if (t < SA_H) { sm.qmask[1][t] = mk; sm.qw13[1][t] = th.w1[3]; sm.qw3[1][t] = th.h0; }
        __syncthreads();
        q_forward2<true, TR>(sm, q + SQ_NP, QW1, q + SQ_W2, sm.b2, ws, acc, 8);
        h2mask[1] = 0;
#pragma unroll
        for (int tt = 0; tt < SA_NT; ++tt)
#pragma unroll
            for (int r = 0; r < 4; ++r) h2mask[1] |= (acc[tt][r] > 0.0f ? 1u : 0u) << (4 * tt + r);
    }
    alpha = wait_owed_alpha(ow, ws_, batch, alpha_p, sm);
    if (t < SR) {
        const float q1 = sm.rv[t][10], q2 = sm.rv[t][8];
        const bool valid = row0 + t < batch;
        sm.rv[t][9] = valid ? alpha * sm.rv[t][5] - fminf(q1, q2) : 0.0f;                    // loss term (:197)
        // torch.min routes the gradient to the smaller input, half / half on ties
        sm.rv[t][10] = !valid ? 0.0f : q1 < q2 ? 1.0f : (q2 < q1 ? 0.0f : 0.5f);               // weight of critic 1
        sm.rv[t][12] = !valid ? 0.0f : q2 < q1 ? 1.0f : (q1 < q2 ? 0.0f : 0.5f);               // weight of critic 2
    }
    __syncthreads();
    // ---- d(-min Q)/d action through both critics ----
    float da_part[2];
#pragma unroll
    for (int net = 0; net < 2; ++net) {
        const float dq = -invn * sm.rv[li][net == 0 ? 10 : 12];
#pragma unroll
        for (int tt = 0; tt < SA_NT; ++tt) {
            const f32x4 w3 = *reinterpret_cast<const f32x4*>(&sm.qw3[net][16 * (SA_NT * wv + tt) + 4 * lg]);
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[tt][r] = (h2mask[net] >> (4 * tt + r)) & 1u ? w3[r] * dq : 0.0f;
        }
        if (net == 1) __syncthreads();                          // net 0's MFMA pass is done reading b2
        store_acc(acc, sm.b2);
        __syncthreads();
        if (net == 0) mfma_pass<true, true>(q + SQ_W2, q + SQ_NP + SQ_W2, sm.b2, ws, acc);
        else mfma_pass<true, true>(q + SQ_NP + SQ_W2, actor + AC_W2, sm.b2, ws, acc);
        da_part[net] = q_daction_partial(sm, net, acc);
    }
    rows_combine2(sm, da_part[0], da_part[1], 8, 11);
    }
    // ---- d loss / d mean, d loss / d sraw per row ----
    if (t < SR) {
        const float u = sm.rv[t][4], sd = sm.rv[t][3], ls = sm.rv[t][2];
        const float omu2 = 1.0f - u * u;
        const bool valid = row0 + t < batch;
        const float c = valid ? invn : 0.0f;
        const float dqa = sm.rv[t][8] + sm.rv[t][11];
        const float du = SA_ACT_SCALE * dqa + alpha * c * (2.0f * SA_ACT_SCALE * u) / (SA_ACT_SCALE * omu2 + 1e-6f);
        const float dz_u = du * omu2;
        const float dL = dz_u * e_row * sd - alpha * c;
        sm.rv[t][0] = dz_u;                                                                          // d / d mean
        sm.rv[t][1] = 0.5f * (SA_LOG_STD_MAX - SA_LOG_STD_MIN) * dL * (1.0f - ls * ls);                // d / d sraw
    }
    __syncthreads();
    if (t < SA_H) {   // unit j = t of the actor: dz2, thin gradients of the heads / bias 2, H1 / DZ2 rows
        float gwm = 0.0f, gwl = 0.0f, gb2 = 0.0f;
#pragma unroll
        for (int r = 0; r < SR; ++r) {
            const float h2 = sm.b1[r][t], dm = sm.rv[r][0], ds = sm.rv[r][1];
            const float dz = h2 > 0.0f ? __builtin_fmaf(wl, ds, wm * dm) : 0.0f;
            gwm = __builtin_fmaf(dm, h2, gwm); gwl = __builtin_fmaf(ds, h2, gwl); gb2 += dz;
            DZ2[(size_t)(row0 + r) * SA_H + t] = dz; H1[(size_t)(row0 + r) * SA_H + t] = sm.b0[r][t];
            sm.b2[r][t] = dz;
        }
        slab[768 + 256 + t] = gb2; slab[768 + 512 + t] = gwm; slab[768 + 512 + 257 + t] = gwl;
        if (t == 0) {
            float gbm = 0.0f, gbl = 0.0f, l = 0.0f, lp = 0.0f;
            for (int r = 0; r < SR; ++r) { gbm += sm.rv[r][0]; gbl += sm.rv[r][1]; l += sm.rv[r][9]; lp += row0 + r < batch ? sm.rv[r][5] : 0.0f; }
            slab[768 + 512 + 256] = gbm; slab[768 + 512 + 257 + 256] = gbl; slab[1794] = l; slab[1795] = lp;
        }
    }
    __syncthreads();
    SAC_MARK(1, (int)blockIdx.y & 3, 6);
    mfma_pass<true, false>(actor + AC_W2, nullptr, sm.b2, ws, acc);
    SAC_MARK(1, (int)blockIdx.y & 3, 7);
#pragma unroll
    for (int tt = 0; tt < SA_NT; ++tt) {
        const f32x4 h1 = *reinterpret_cast<const f32x4*>(&sm.b0[li][16 * (SA_NT * wv + tt) + 4 * lg]);
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[tt][r] = h1[r] > 0.0f ? acc[tt][r] : 0.0f;
    }
    store_acc(acc, sm.b1);                                       // dz1 image (the h2 image was last read before the previous barrier)
    __syncthreads();
    if (t < SA_H) {
        float gb1 = 0.0f, gw[3] = {0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int r = 0; r < SR; ++r) {
            const float d = sm.b1[r][t];
            gb1 += d;
#pragma unroll
            for (int c = 0; c < 3; ++c) gw[c] = __builtin_fmaf(d, sm.x[r][c], gw[c]);
        }
        slab[3 * t] = gw[0]; slab[3 * t + 1] = gw[1]; slab[3 * t + 2] = gw[2];
        slab[768 + t] = gb1;
    }
    SAC_MARK(1, (int)blockIdx.y & 3, 8);
}

// ================================================ dW2 = dZ2^T H1 on the f32 MFMA ================================================
// Workgroup = one 32 x 64 tile of one 256 x 256 weight gradient over one K-range of the (padded) batch; its 4 waves split that range
// and are summed through LDS in wave order.  Per k-step (4 batch rows) a lane loads one float2 of dZ2 and one float4 of H1 and
// issues 8 MFMAs; the element -> tile assignment follows the vector loads (A element e: row 32 mg + 2 i + e, B element e: column
// 64 ng + 4 j + e), so the results leave as float4 stores.  Output: partial [blockIdx.y][mat][256*256] in the workspace.
__global__ void __launch_bounds__(256) sac_dw2_gemm_kernel(float* __restrict__ ws, int batch, int mat0, int n_mats) {
    __shared__ f32x4 red[3][8][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, j = lane & 15, g = lane >> 4;
    const int tile = blockIdx.x & 31, mat = blockIdx.x >> 5, mg = tile >> 2, ng = tile & 3;
    const int Kp = ws_kp(batch), steps = Kp / 4, slices = 4 * gridDim.y, slice = 4 * blockIdx.y + w;
    const int s0 = (int)((long long)steps * slice / slices), s1 = (int)((long long)steps * (slice + 1) / slices);
    const float* H1 = ws + (size_t)(mat0 + mat) * ws_mat_floats(batch) + 64 * ng + 4 * j;
    const float* DZ2 = ws + (size_t)(3 + mat0 + mat) * ws_mat_floats(batch) + 32 * mg + 2 * j;
    f32x4 acc[2][4];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    int s = s0;
    for (; s + 4 <= s1; s += 4) {
        float2 av[4]; float4 bv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const size_t row = (size_t)(4 * (s + u) + g) * SA_H;
            av[u] = *reinterpret_cast<const float2*>(DZ2 + row); bv[u] = *reinterpret_cast<const float4*>(H1 + row);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const float a2[2] = {av[u].x, av[u].y}; const float b4[4] = {bv[u].x, bv[u].y, bv[u].z, bv[u].w};
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) acc[a][b] = MFMA16(a2[a], b4[b], acc[a][b]);
        }
    }
    for (; s < s1; ++s) {
        const size_t row = (size_t)(4 * s + g) * SA_H;
        const float2 av = *reinterpret_cast<const float2*>(DZ2 + row); const float4 bv = *reinterpret_cast<const float4*>(H1 + row);
        const float a2[2] = {av.x, av.y}; const float b4[4] = {bv.x, bv.y, bv.z, bv.w};
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) acc[a][b] = MFMA16(a2[a], b4[b], acc[a][b]);
    }
    if (w > 0) {
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) red[w - 1][4 * a + b][lane] = acc[a][b];
    }
    __syncthreads();
    if (w == 0) {
        float* out = ws + ws_part_off(batch) + ((size_t)blockIdx.y * 3 + mat) * SA_H * SA_H;
#pragma unroll
        for (int a = 0; a < 2; ++a) {
#pragma unroll
            for (int b = 0; b < 4; ++b) acc[a][b] = ((acc[a][b] + red[0][4 * a + b][lane]) + red[1][4 * a + b][lane]) + red[2][4 * a + b][lane];
            // acc[a][b][r] = dW2[32 mg + 2 (4 g + r) + a][64 ng + 4 j + b]
#pragma unroll
            for (int r = 0; r < 4; ++r)
                *reinterpret_cast<f32x4*>(out + (size_t)(32 * mg + 2 * (4 * g + r) + a) * SA_H + 64 * ng + 4 * j) = f32x4{acc[a][0][r], acc[a][1][r], acc[a][2][r], acc[a][3][r]};
        }
    }
}

// gradient assembly: (a) thin gradients = fixed-order sum of the per-workgroup slabs, scattered into the flat layout; (b) W2 gradients =
// fixed-order sum of the GEMM's K-split partials; (c) the two loss scalars.  With `opt.params` set, the same launch also applies the Adam
// step to every element it has just assembled and (critics) the polyak step of the target copy: optimizer.step() and the target update
// (sac.py:185,213-217) cost no launch of their own.
struct sac_opt_t { float* params; float* m; float* v; float* target; float w1, b2, w2, step_size, rbc2, eps, tau;
                   float* params_t; float* target_t; int is_actor;     // transposed layer-2 copies of params / target (nullable: not registered or not valid), kept in step by sac_apply
                   // sharded run on the P2P carrier (round 6; world = 0: none): every gradient element and loss scalar is exchanged — line = its index in the caller's
                   // {gradient, 2 scalars} buffer, rank-ordered sum — by the thread that has just assembled it, and stepped behind the exchange: the all-reduce, Adam and polyak
                   // launches of the sharded routes disappear into the assembly launch (sac.py:185-217 with the exchange between backward and step).  gate: the carrier's
                   // status word — a wait that ran out withholds every step (mi_common.h).
                   p2p_args_t xa; int world; const uint32_t* gate; };
__device__ __forceinline__ float sac_xchg(const sac_opt_t& o, int line, float g) { return o.world > 0 ? p2p_exchange_rt(o.xa, o.world, line, g) : g; }
// flat parameter index -> index into the transposed layer-2 copy ([net][k][unit]), -1 outside the layer-2 matrices
__device__ __forceinline__ int sac_t_index(int is_actor, int i) {
    const int net = is_actor ? 0 : (i >= SQ_NP ? 1 : 0);
    const int l = i - net * SQ_NP - (is_actor ? AC_W2 : SQ_W2);
    if (l < 0 || l >= SA_H * SA_H) return -1;
    return net * (SA_H * SA_H) + (l & (SA_H - 1)) * SA_H + (l >> 8);
}
// the element's optimizer state is requested BEFORE its gradient is summed (sac_state_load), so the launch is one memory latency deep, not two
struct sac_state_t { float p, m, v, t; unsigned fault; };   // fault: the device's fault word, requested with the state (set by an EARLIER launch: a plain load)
__device__ __forceinline__ sac_state_t sac_state_load(const sac_opt_t& o, int i) {
    sac_state_t s = {0.0f, 0.0f, 0.0f, 0.0f, 0u};
    if (o.params) { s.p = o.params[i]; s.m = o.m[i]; s.v = o.v[i]; if (o.target) s.t = o.target[i]; s.fault = sac_fault_word; }
    return s;
}
__device__ __forceinline__ void sac_apply(const sac_opt_t& o, int i, float g, sac_state_t s) {
    if (s.fault || mi_gate_closed(o.gate)) return;             // see sac_timeout / the P2P carrier's fail-safe: {params, exp_avg, exp_avg_sq, target} stay as they were
    const float p = mi_adam_elem(s.p, g, s.m, s.v, o.w1, o.b2, o.w2, o.step_size, o.rbc2, o.eps);
    o.m[i] = s.m; o.v[i] = s.v;
    o.params[i] = p;
    const float tn = o.tau * p + (1.0f - o.tau) * s.t;
    if (o.target) o.target[i] = tn;
    if (o.params_t) {
        const int ti = sac_t_index(o.is_actor, i);
        if (ti >= 0) { o.params_t[ti] = p; if (o.target && o.target_t) o.target_t[ti] = tn; }
    }
}
#define RED_SMALL_PER_BLOCK 64
// (a) + (c) for the 64 thin elements of block `blk`: threads 0..255 of the workgroup (4 slab groups x 64 elements); contains one __syncthreads
__device__ __forceinline__ void sac_thin_reduce(int blk, float (&part)[4][RED_SMALL_PER_BLOCK], const float* __restrict__ ws, int batch, int n_slabs, int is_actor,
                                                double inv_count, float* __restrict__ grads, float* __restrict__ out2, const sac_opt_t& opt) {
    const int per = is_actor ? 1794 : 1793, nets = is_actor ? 1 : 2, n_small = per * nets + 2;
    {
        const float* slabs = ws + ws_slab_off(batch);
        const bool live = threadIdx.x < 256;
        const int e = live ? blk * RED_SMALL_PER_BLOCK + (threadIdx.x & 63) : n_small, grp = (threadIdx.x >> 6) & 3;
        int off = -1;
        if (e < per * nets) off = (e / per) * 1793 + e % per;
        else if (e < n_small) off = (is_actor ? 1794 : 3586) + (e - per * nets);
        int dst = -1;
        if (e < per * nets) {
            const int net = e / per, l = e % per;
            if (is_actor) dst = l < 768 ? AC_W1 + l : l < 1024 ? AC_B1 + (l - 768) : l < 1280 ? AC_B2 + (l - 1024) : l < 1536 ? AC_WM + (l - 1280)
                              : l == 1536 ? AC_BM : l < 1793 ? AC_WL + (l - 1537) : AC_BL;
            else dst = net * SQ_NP + (l < 1024 ? SQ_W1 + l : l < 1280 ? SQ_B1 + (l - 1024) : l < 1536 ? SQ_B2 + (l - 1280) : l < 1792 ? SQ_W3 + (l - 1536) : SQ_B3);
        }
        sac_state_t st = {0.0f, 0.0f, 0.0f, 0.0f, 0u};
        if (live && grp == 0 && dst >= 0) st = sac_state_load(opt, dst);
        float acc = 0.0f;
        if (off >= 0) {   // this group's slabs grp, grp + 4, ...: eight loads in flight at a time, summed in slab order
            int b = grp;
            for (; b + 28 < n_slabs; b += 32) {
                float x[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) x[k] = slabs[(size_t)(b + 4 * k) * SLAB + off];
#pragma unroll
                for (int k = 0; k < 8; ++k) acc += x[k];
            }
            for (; b + 12 < n_slabs; b += 16) {
                float x[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) x[k] = slabs[(size_t)(b + 4 * k) * SLAB + off];
#pragma unroll
                for (int k = 0; k < 4; ++k) acc += x[k];
            }
            for (; b < n_slabs; b += 4) acc += slabs[(size_t)b * SLAB + off];
        }
        if (live) part[grp][threadIdx.x & 63] = acc;
        __syncthreads();
        if (live && grp == 0 && off >= 0) {
            float v = (part[0][threadIdx.x] + part[1][threadIdx.x]) + (part[2][threadIdx.x] + part[3][threadIdx.x]);
            if (dst >= 0) {
                v = sac_xchg(opt, dst, v);
                grads[dst] = v;
                if (opt.params) sac_apply(opt, dst, v, st);
            } else if (out2) out2[e - per * nets] = sac_xchg(opt, (is_actor ? AC_NP : 2 * SQ_NP) + (e - per * nets), (float)((double)v * inv_count));
        }
    }
}
__global__ void __launch_bounds__(256) sac_grad_reduce_kernel(const float* __restrict__ ws, int batch, int n_slabs, int n_split, int is_actor, double inv_count,
                                                              float* __restrict__ grads, float* __restrict__ out2, sac_opt_t opt) {
    __shared__ float part[4][RED_SMALL_PER_BLOCK];
    const int per = is_actor ? 1794 : 1793, nets = is_actor ? 1 : 2, n_small = per * nets + 2;
    const int nb_small = (n_small + RED_SMALL_PER_BLOCK - 1) / RED_SMALL_PER_BLOCK;
    if ((int)blockIdx.x < nb_small) {
        sac_thin_reduce((int)blockIdx.x, part, ws, batch, n_slabs, is_actor, inv_count, grads, out2, opt);
    } else {
        const int e4 = (blockIdx.x - nb_small) * 256 + threadIdx.x;          // float4 index over nets x 65536
        if (e4 >= nets * (SA_H * SA_H / 4)) return;
        const int net = e4 / (SA_H * SA_H / 4), l4 = e4 % (SA_H * SA_H / 4);
        const float* part0 = ws + ws_part_off(batch) + (size_t)net * SA_H * SA_H + 4 * (size_t)l4;
        const int d0 = (is_actor ? AC_W2 : net * SQ_NP + SQ_W2) + 4 * l4;    // the second critic's block is only 4-byte aligned: scalar accesses
        sac_state_t st[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) st[c] = sac_state_load(opt, d0 + c);
        f32x4 acc = *reinterpret_cast<const f32x4*>(part0);
        if (n_split == 2) acc += *reinterpret_cast<const f32x4*>(part0 + (size_t)3 * SA_H * SA_H);
        else if (n_split == 4) {   // all partials in flight, summed in split order
            const f32x4 a1 = *reinterpret_cast<const f32x4*>(part0 + (size_t)1 * 3 * SA_H * SA_H), a2 = *reinterpret_cast<const f32x4*>(part0 + (size_t)2 * 3 * SA_H * SA_H);
            const f32x4 a3 = *reinterpret_cast<const f32x4*>(part0 + (size_t)3 * 3 * SA_H * SA_H);
            acc += a1; acc += a2; acc += a3;
        } else for (int y = 1; y < n_split; ++y) acc += *reinterpret_cast<const f32x4*>(part0 + (size_t)y * 3 * SA_H * SA_H);
#pragma unroll
        for (int c = 0; c < 4; ++c) { const float gx = sac_xchg(opt, d0 + c, acc[c]); grads[d0 + c] = gx; if (opt.params) sac_apply(opt, d0 + c, gx, st[c]); }
    }
}

// Small batches (Kp <= 512: the reference's 256): dW2, its optimizer step and the thin gradients in ONE launch — the two kernels above are a 5.5 us + 5.0 us pair of
// which ~4.5 us is the boundary between them (round 3: 95 -> 86 us per SAC iteration).  Workgroups [0, 64 nets): one 32 x 32 tile of one dW2 each, the (padded) batch
// split over EIGHT waves; their partial tiles are summed through LDS in the order the K-split pair of launches used at the reference batch (two groups of four waves
// = its two K-split partials: bitwise the same gradient), the tile is laid out row-major in LDS, and the 512 threads apply Adam (+ polyak) to 2 consecutive elements
// each — the element's optimizer state is requested before the K loop.  Workgroups behind them: sac_thin_reduce.
// The launch's body as a ROLE (workgroup `bi` of 64 nets + thin blocks, 512 threads): sac_dw2_adam_kernel runs it alone; rg_act::sac_act_kernel carries the critics'
// step owed from the last critic update on extra workgroups of the acting launch (mi_sac_act_step_carry).
union __attribute__((aligned(16))) sac_dw2_smem { f32x4 red[8][4][64]; float tile[32][36]; float part[4][RED_SMALL_PER_BLOCK]; };
struct sac_dw2_args_t { float* ws; int batch, mat0, n_slabs, is_actor; double inv_count; float* grads; float* out2; sac_opt_t opt; };
static int sac_dw2_blocks(int is_actor) { return 64 * (is_actor ? 1 : 2) + ((is_actor ? 1794 : 2 * 1793) + 2 + RED_SMALL_PER_BLOCK - 1) / RED_SMALL_PER_BLOCK; }
__device__ __forceinline__ void sac_dw2_adam_role(sac_dw2_smem& sm, const int bi, float* __restrict__ ws, int batch, int mat0, int n_slabs, int is_actor, double inv_count,
                                                  float* __restrict__ grads, float* __restrict__ out2, const sac_opt_t& opt) {
    const int nets = is_actor ? 1 : 2;
    if (bi >= 64 * nets) {
        sac_thin_reduce(bi - 64 * nets, sm.part, ws, batch, n_slabs, is_actor, inv_count, grads, out2, opt);
        return;
    }
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, j = lane & 15, g = lane >> 4;
    const int tile = bi & 63, mat = bi >> 6, mg = tile >> 3, ng = tile & 7;
    const int Kp = ws_kp(batch), steps = Kp / 4;
    const int s0 = (int)((long long)steps * w / 8), s1 = (int)((long long)steps * (w + 1) / 8);
    const float* H1 = ws + (size_t)(mat0 + mat) * ws_mat_floats(batch) + 32 * ng + 2 * j;
    const float* DZ2 = ws + (size_t)(3 + mat0 + mat) * ws_mat_floats(batch) + 32 * mg + 2 * j;
    // everything this thread will need from memory is requested here, before anything is waited for: the first 8 k-steps' operands (all of them at the reference
    // batch) and the optimizer state of its 2 elements of the finished tile (row 32 mg + (t >> 4), columns 32 ng + 2 (t & 15), + 1) — one round trip per launch
    float2 av[8], bv[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const size_t row = (size_t)(4 * (s0 + u < s1 ? s0 + u : s0) + g) * SA_H;
        av[u] = *reinterpret_cast<const float2*>(DZ2 + row); bv[u] = *reinterpret_cast<const float2*>(H1 + row);
    }
    const int d0 = (is_actor ? AC_W2 : mat * SQ_NP + SQ_W2) + (32 * mg + ((int)threadIdx.x >> 4)) * SA_H + 32 * ng + 2 * ((int)threadIdx.x & 15);
    sac_state_t st[2];
#pragma unroll
    for (int c = 0; c < 2; ++c) st[c] = sac_state_load(opt, d0 + c);
    f32x4 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    auto kstep = [&](const float2& a_, const float2& b_) {
        const float a2[2] = {a_.x, a_.y}, b2[2] = {b_.x, b_.y};
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) acc[a][b] = MFMA16(a2[a], b2[b], acc[a][b]);
    };
#pragma unroll
    for (int u = 0; u < 8; ++u) if (s0 + u < s1) kstep(av[u], bv[u]);   // (wave-uniform)
    for (int s = s0 + 8; s < s1; ++s) {                                 // batches above 256 rows
        const size_t row = (size_t)(4 * s + g) * SA_H;
        kstep(*reinterpret_cast<const float2*>(DZ2 + row), *reinterpret_cast<const float2*>(H1 + row));
    }
    // cross-wave sum: every wave leaves its 4 fragments in LDS; wave w then owns half of fragment f = w >> 1 (registers 2 (w & 1), + 1) and adds the eight partials
    // in the order of the K-split pair of launches this kernel replaces at the reference batch (waves 0-3 = its first partial, 4-7 = its second)
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) sm.red[w][2 * a + b][lane] = acc[a][b];
    __syncthreads();
    const int f = w >> 1, r0 = 2 * (w & 1);
    float sum[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const float lo = ((sm.red[0][f][lane][r0 + q] + sm.red[1][f][lane][r0 + q]) + sm.red[2][f][lane][r0 + q]) + sm.red[3][f][lane][r0 + q];
        const float hi = ((sm.red[4][f][lane][r0 + q] + sm.red[5][f][lane][r0 + q]) + sm.red[6][f][lane][r0 + q]) + sm.red[7][f][lane][r0 + q];
        sum[q] = lo + hi;
    }
    __syncthreads();   // every partial has been read: the LDS becomes the row-major tile
    {   // fragment (a, b), register r = dW2[32 mg + 2 (4 g + r) + a][32 ng + 2 j + b]
        const int a = f >> 1, b = f & 1;
#pragma unroll
        for (int q = 0; q < 2; ++q) sm.tile[2 * (4 * g + r0 + q) + a][2 * j + b] = sum[q];
    }
    __syncthreads();
    float2 gv = *reinterpret_cast<const float2*>(&sm.tile[threadIdx.x >> 4][2 * (threadIdx.x & 15)]);
    gv.x = sac_xchg(opt, d0, gv.x); gv.y = sac_xchg(opt, d0 + 1, gv.y);
    grads[d0] = gv.x; grads[d0 + 1] = gv.y;
    if (opt.params) { sac_apply(opt, d0, gv.x, st[0]); sac_apply(opt, d0 + 1, gv.y, st[1]); }
}
__global__ void __launch_bounds__(512) sac_dw2_adam_kernel(sac_dw2_args_t a) {
    __shared__ sac_dw2_smem sm;
    MI_INSIDE_SCOPE(MI_PROF_SAC_GEMM);
    sac_dw2_adam_role(sm, (int)blockIdx.x, a.ws, a.batch, a.mat0, a.n_slabs, a.is_actor, a.inv_count, a.grads, a.out2, a.opt);
}

// ---- transposed layer-2 copies ("shadows"), include/mi_rl.h mi_sac_shadow_*.  A registry keyed by the flat parameter vector's device pointer: the caller owns both buffers,
//      registers the pair, refreshes it (one transpose launch) whenever it has written the parameters behind the library's back, and the library (a) streams every forward
//      pass of the acting / update launches from valid shadows, (b) keeps them in step wherever ITS fused optimizer steps write a layer-2 element (sac_apply), (c) marks a
//      shadow invalid wherever one of its other writers (mi_adam, mi_polyak, mi_clip_adam) touches the vector.  An invalid or missing shadow only means the old access pattern.
struct sac_shadow_ent { const float* base; float* t; int is_actor; bool valid; };
#define SAC_MAX_SHADOWS 32
static sac_shadow_ent g_shadow[SAC_MAX_SHADOWS];
static int g_nshadow = 0;
static sac_shadow_ent* shadow_find(const float* base) {
    for (int k = 0; k < g_nshadow; ++k) if (g_shadow[k].base == base) return &g_shadow[k];
    return nullptr;
}
static float* shadow_valid(const float* base) { sac_shadow_ent* e = base ? shadow_find(base) : nullptr; return (e && e->valid) ? e->t : nullptr; }
extern "C" int mi_sac_shadow_set(const float* params, int is_actor, float* shadow) {
    MI_CHECK_ARG(params != nullptr && (is_actor == 0 || is_actor == 1), "params is NULL / is_actor must be 0 (the two critics' flat vector) or 1 (the actor's)");
    sac_shadow_ent* e = shadow_find(params);
    if (!shadow) {   // drop
        if (e) { *e = g_shadow[g_nshadow - 1]; --g_nshadow; }
        return MI_OK;
    }
    if (!e) {
        if (g_nshadow == SAC_MAX_SHADOWS) { mi_set_error("mi_sac_shadow_set: more than %d parameter vectors registered", SAC_MAX_SHADOWS); return MI_ESTATE; }
        e = &g_shadow[g_nshadow++];
    }
    e->base = params; e->t = shadow; e->is_actor = is_actor; e->valid = false;   // valid after mi_sac_shadow_refresh
    return MI_OK;
}
extern "C" int mi_sac_shadow_invalidate(const float* params) {   // NULL: every registered vector
    for (int k = 0; k < g_nshadow; ++k) if (!params || g_shadow[k].base == params) g_shadow[k].valid = false;
    return MI_OK;
}
static void shadow_invalidate_range(const float* p, size_t n) {   // a library writer that does not maintain shadows touched [p, p + n): any OVERLAP with a registered vector
    for (int k = 0; k < g_nshadow; ++k) {
        const float* base = g_shadow[k].base;
        const size_t len = g_shadow[k].is_actor ? (size_t)AC_NP : (size_t)2 * SQ_NP;
        if (base < p + n && base + len > p) g_shadow[k].valid = false;
    }
}
// t[m][k][u] = W_m[u][k]: 32 x 32 tiles through LDS, both sides coalesced
__global__ void __launch_bounds__(256) sac_transpose_kernel(const float* __restrict__ params, int is_actor, float* __restrict__ t) {
    __shared__ float tile[32][33];
    const int m = blockIdx.z;
    const float* W = params + (is_actor ? AC_W2 : m * SQ_NP + SQ_W2);
    const int u0 = 32 * blockIdx.y, k0 = 32 * blockIdx.x, tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
    for (int r = ty; r < 32; r += 8) tile[r][tx] = W[(size_t)(u0 + r) * SA_H + k0 + tx];
    __syncthreads();
#pragma unroll
    for (int r = ty; r < 32; r += 8) t[(size_t)m * SA_H * SA_H + (size_t)(k0 + r) * SA_H + u0 + tx] = tile[tx][r];
}
extern "C" int mi_sac_shadow_refresh(const float* params, void* stream) {
    sac_shadow_ent* e = params ? shadow_find(params) : nullptr;
    MI_CHECK_ARG(e != nullptr, "params is not a registered parameter vector (mi_sac_shadow_set)");
    sac_transpose_kernel<<<dim3(SA_H / 32, SA_H / 32, e->is_actor ? 1 : 2), 256, 0, (hipStream_t)stream>>>(params, e->is_actor, e->t);
    MI_LAUNCH_CHECK();
    e->valid = true;
    return MI_OK;
}
extern "C" int mi_sac_shadow_valid(const float* params) { return shadow_valid(params) ? 1 : 0; }

static sac_opt_t sac_no_opt() { sac_opt_t o; memset(&o, 0, sizeof(o)); return o; }
static sac_opt_t sac_make_opt(float* params, float* m, float* v, float* target, int64_t step, double lr, double beta1, double beta2, double eps, float tau, int is_actor) {
    sac_opt_t o;
    memset(&o, 0, sizeof(o));
    o.params_t = shadow_valid(params); o.target_t = target ? shadow_valid(target) : nullptr; o.is_actor = is_actor;
    if (target && o.target_t && !o.params_t) { (void)mi_sac_shadow_invalidate(target); o.target_t = nullptr; }   // the step would move the target without its shadow
    const double bc1 = 1.0 - pow(beta1, (double)step), bc2 = 1.0 - pow(beta2, (double)step);
    o.params = params; o.m = m; o.v = v; o.target = target;
    o.w1 = (float)(1.0 - beta1); o.b2 = (float)beta2; o.w2 = (float)(1.0 - beta2); o.step_size = (float)(lr / bc1); o.rbc2 = (float)(1.0 / sqrt(bc2)); o.eps = (float)eps; o.tau = tau;
    return o;
}

static int sac_launch_grads(void* workspace, int batch, int is_actor, double inv_count, float* grads, float* out2, const sac_opt_t& opt, hipStream_t s) {
    const int nb = ws_kp(batch) / SR, nets = is_actor ? 1 : 2, split = gemm_split(batch);
    if (ws_kp(batch) <= SAC_FUSED_KP) {
        const int n_small = (is_actor ? 1794 : 2 * 1793) + 2;
        mi_prof_scope prof(MI_PROF_SAC_GEMM, s);
        static_assert(SAC_FUSED_KP <= 512, "sac_dw2_adam_role requests at most 8 k-steps per wave up front");
        (void)n_small; (void)nets;
        sac_dw2_adam_kernel<<<sac_dw2_blocks(is_actor), 512, 0, s>>>(sac_dw2_args_t{(float*)workspace, batch, is_actor ? 2 : 0, nb, is_actor, inv_count, grads, out2, opt});
        MI_LAUNCH_CHECK();
        return MI_OK;
    }
    {
        mi_prof_scope prof(MI_PROF_SAC_GEMM, s);
        sac_dw2_gemm_kernel<<<dim3(32 * nets, split), 256, 0, s>>>((float*)workspace, batch, is_actor ? 2 : 0, nets);
    }
    MI_LAUNCH_CHECK();
    const int n_small = (is_actor ? 1794 : 2 * 1793) + 2;
    const int nblk = (n_small + RED_SMALL_PER_BLOCK - 1) / RED_SMALL_PER_BLOCK + nets * (SA_H * SA_H / 4) / 256;
    {
        mi_prof_scope prof(MI_PROF_SAC_ASSEMBLE, s);
        sac_grad_reduce_kernel<<<nblk, 256, 0, s>>>((const float*)workspace, batch, nb, split, is_actor, inv_count, grads, out2, opt);
    }
    MI_LAUNCH_CHECK();
    return MI_OK;
}

// ================================================ acting ========================================================================
// (Round 4, measured, not kept: requesting the row's env state before the forward instead of behind it.  In front of the actor's own requests: +1 us per launch — loads
// return in order, the first weight tiles wait behind seven state loads; behind them: no change, 84.6 - 84.7 against 84.2 - 84.6 us per iteration — the tail of the launch
// is the fp64 step itself, not its loads.  Running the action-independent half of the step (fmod, sin) on the stepping wave ahead of the first barrier: +0.7 us, every
// wave of the workgroup waits for it there.)
namespace rg_act {
template <bool TR>
__global__ void __launch_bounds__(SA_THREADS)
sac_act_kernel(mi_env e, const float* __restrict__ actor, long long global_step, long long slots, long long learning_starts, float* __restrict__ obs_cur,
               float* __restrict__ observations, float* __restrict__ actions, float* __restrict__ rewards, uint8_t* __restrict__ terminated,
               const float* __restrict__ forced_actions, const float* __restrict__ forced_eps, const double* __restrict__ forced_resets,
               mi_episode_t* __restrict__ episodes, int32_t* __restrict__ episode_stats, int max_ep, int n_act, sac_dw2_args_t dw, const float* __restrict__ actor_t) {
    __shared__ sac_smem sm;
    MI_INSIDE_SCOPE(MI_PROF_SAC_ACT);
    // workgroups behind the acting ones: the critics' weight-gradient + Adam + polyak step owed from the last critic update (mi_sac_act_step_carry).  Acting reads the
    // actor and the env only, the step touches the critics only: no dependency inside the launch; whoever needs the stepped critics sits behind the kernel boundary.
    if ((int)blockIdx.x >= n_act) {
        static_assert(sizeof(sac_smem) >= sizeof(sac_dw2_smem) && alignof(sac_smem) >= alignof(sac_dw2_smem), "the carried critic step reuses the acting kernel's LDS block");
        sac_dw2_adam_role(*reinterpret_cast<sac_dw2_smem*>(&sm), (int)blockIdx.x - n_act, dw.ws, dw.batch, dw.mat0, dw.n_slabs, dw.is_actor, dw.inv_count, dw.grads, dw.out2, dw.opt);
        return;
    }
    const int N = e.n, row0 = blockIdx.x * SR;
    const bool policy = !forced_actions && global_step >= learning_starts;   // block-uniform
    if (policy) {
        wstream ws; thin_t th; f32x4 acc[SA_NT];
        issue_thin_actor(actor, th);
        const float* const AW = TR ? actor_t : actor + AC_W2;
        float ov = 0.0f;   // the observation is requested first, the noise is arithmetic behind the request, the weight stream is primed behind the LDS store (see sac_critic_kernel)
        if (threadIdx.x < SR * 3) { const int r = threadIdx.x / 3, k = threadIdx.x % 3; const int g = row0 + r < N ? row0 + r : N - 1; ov = obs_cur[3 * (size_t)g + k]; }
        if (threadIdx.x < SR) {
            const int g = row0 + threadIdx.x < N ? row0 + threadIdx.x : N - 1;
            sm.noise[threadIdx.x] = forced_eps ? forced_eps[g] : keyed_normal(e.seed, (1ull << 40) + e.env_id_base + (uint64_t)g, (uint64_t)global_step);
        }
        if (threadIdx.x < SR * 3) sm.x[threadIdx.x / 3][threadIdx.x % 3] = ov;
        stream_prime<TR>(AW, ws);
        __syncthreads();
        const float eps = threadIdx.x < SR ? sm.noise[threadIdx.x] : 0.0f;
        layer1<3>(sm, th, sm.x, sm.b0);
        __syncthreads();
        actor_forward2<false, TR>(sm, actor, AW, nullptr, sm.b0, ws, acc, eps);
    }
    if (threadIdx.x < SR && row0 + threadIdx.x < N) {
        const int g = row0 + threadIdx.x;
        float a;
        if (forced_actions) a = forced_actions[g];
        else if (policy) a = sm.rv[threadIdx.x][6];
        else {   // env.action_space.sample() (sac.py:139): uniform in [low, high), keyed
            uint32_t r[4];
            mi_philox(e.seed, e.env_id_base + (uint64_t)g, (uint64_t)global_step, STREAM_UNIF_ACT, r);
            a = -SA_ACT_SCALE + 2.0f * SA_ACT_SCALE * ((float)(r[0] >> 8) * (1.0f / 16777216.0f));
        }
        const long long slot = global_step % slots, nslot = (global_step + 1) % slots;
        actions[slot * N + g] = a;                                                        // sac.py:145
        double th = e.x[g], thd = e.x_dot[g];
        int elapsed = e.elapsed[g], eplen = e.ep_len[g];
        float epret = e.ep_ret[g];
        uint64_t episode = e.episode[g];
        const pend_out r = pend_step_one(e, g, a, forced_resets ? forced_resets + 2 * (size_t)g : nullptr, th, thd, elapsed, epret, eplen, episode);
        e.x[g] = th; e.x_dot[g] = thd; e.elapsed[g] = elapsed; e.ep_len[g] = eplen; e.ep_ret[g] = epret; e.episode[g] = episode;
        e.step_ctr[g] += 1;
        const size_t no = (size_t)(nslot * N + g);
        observations[3 * no] = r.o0; observations[3 * no + 1] = r.o1; observations[3 * no + 2] = r.o2;   // :156 (reset obs where done)
        rewards[no] = r.reward;                                                           // :157
        terminated[no] = 0;                                                               // :158: done and not truncated == False for Pendulum
        obs_cur[3 * (size_t)g] = r.o0; obs_cur[3 * (size_t)g + 1] = r.o1; obs_cur[3 * (size_t)g + 2] = r.o2;
        if (r.done && episode_stats) {
            atomicAdd(episode_stats, 1); atomicAdd(episode_stats + 1, r.fin_len);
            if (max_ep > 0) { const int sl = atomicAdd(episode_stats + 3, 1); if (sl < max_ep) episodes[sl] = mi_episode_t{g, 0, r.fin_ret, r.fin_len}; }
        }
    }
}
}  // namespace rg_act

__global__ void sac_zero4_kernel(int32_t* p) { if (threadIdx.x < 4) p[threadIdx.x] = 0; }

// the acting arguments, checked BEFORE anything is enqueued: mi_sac_act_step_carry must not launch (or carry) the owed critic step and then refuse the acting call —
// the caller would keep the debt and apply the same Adam + polyak step a second time
static int sac_act_check(void* handle, const float* actor, int64_t global_step, int64_t slots, const float* obs_cur, const float* observations, const float* actions,
                         const float* rewards, const uint8_t* terminated, const mi_episode_t* episodes, int max_ep) {
    MI_CHECK_ARG(handle && actor && obs_cur && observations && actions && rewards && terminated, "NULL pointer");
    MI_CHECK_ARG(((mi_env*)handle)->kind == MI_ENV_PENDULUM_V1, "SAC path needs a Pendulum-v1 handle");
    MI_CHECK_ARG(slots >= 2 && global_step >= 0 && max_ep >= 0 && (max_ep == 0 || episodes), "bad arguments");
    return MI_OK;
}

static int sac_act_impl(void* handle, const float* actor, int64_t global_step, int64_t slots, int64_t learning_starts, float* obs_cur,
                        float* observations, float* actions, float* rewards, uint8_t* terminated, const float* forced_actions,
                        const float* forced_eps, const double* forced_resets, mi_episode_t* episodes, int32_t* episode_stats, int max_ep,
                        const sac_dw2_args_t* carry, void* stream) {
    if (const int rc = sac_act_check(handle, actor, global_step, slots, obs_cur, observations, actions, rewards, terminated, episodes, max_ep)) return rc;
    mi_env* e = (mi_env*)handle;
    hipStream_t s = (hipStream_t)stream;
    if (episode_stats) { sac_zero4_kernel<<<1, 64, 0, s>>>(episode_stats); MI_LAUNCH_CHECK(); }
    mi_prof_scope prof(MI_PROF_SAC_ACT, s);
    const int n_act = (e->n + SR - 1) / SR;
    static_assert(rg_act::SA_THREADS == 512, "the carried critic step (sac_dw2_adam_role) is written for 512-thread workgroups");
    sac_dw2_args_t none; memset(&none, 0, sizeof(none));
    const float* actor_t = shadow_valid(actor);
    const unsigned grid = n_act + (carry ? sac_dw2_blocks(0) : 0);
    if (actor_t)
        rg_act::sac_act_kernel<true><<<grid, rg_act::SA_THREADS, 0, s>>>(*e, actor, (long long)global_step, (long long)slots, (long long)learning_starts, obs_cur, observations, actions, rewards,
                                                                          terminated, forced_actions, forced_eps, forced_resets, episodes, episode_stats, max_ep, n_act, carry ? *carry : none, actor_t);
    else
        rg_act::sac_act_kernel<false><<<grid, rg_act::SA_THREADS, 0, s>>>(*e, actor, (long long)global_step, (long long)slots, (long long)learning_starts, obs_cur, observations, actions, rewards,
                                                                           terminated, forced_actions, forced_eps, forced_resets, episodes, episode_stats, max_ep, n_act, carry ? *carry : none, nullptr);
    MI_LAUNCH_CHECK();
    return MI_OK;
}

extern "C" int mi_sac_act_step(void* handle, const float* actor, int64_t global_step, int64_t slots, int64_t learning_starts, float* obs_cur,
                               float* observations, float* actions, float* rewards, uint8_t* terminated, const float* forced_actions,
                               const float* forced_eps, const double* forced_resets, mi_episode_t* episodes, int32_t* episode_stats, int max_ep,
                               void* stream) {
    return sac_act_impl(handle, actor, global_step, slots, learning_starts, obs_cur, observations, actions, rewards, terminated, forced_actions, forced_eps, forced_resets,
                        episodes, episode_stats, max_ep, nullptr, stream);
}

// ---- the critics' optimizer step DEFERRED to the next launch (round 4).  mi_sac_critic_update_deferred = mi_sac_critic_update_owed without its second launch (the
// weight-gradient GEMM + gradient assembly + Adam + polyak): the row-group kernel leaves H1 / dZ2 / slabs in the workspace, and the step is owed until
//   mi_sac_act_step_carry(..., step)   carries it on extra workgroups of the acting launch (which reads the actor and the env only: no dependency inside the launch), or
//   mi_sac_critic_step(step)           runs it as the launch of its own that mi_sac_critic_update_owed would have made.
// Either way the same workgroups do the same arithmetic: bit-identical to the undeferred call.  Nothing else may read or write the critics, their targets or their Adam
// moments, or reuse the workspace, in between (deep_rl_amd.SACEngine settles the debt before any such access).
static int sac_critic_step_args(const mi_sac_critic_step_t* st, sac_dw2_args_t* out) {
    MI_CHECK_ARG(st && st->workspace && st->q && st->exp_avg && st->exp_avg_sq && st->grads && st->batch > 0 && st->step >= 1, "critic step: NULL pointer / bad batch or step");
    const sac_opt_t opt = sac_make_opt(st->q, st->exp_avg, st->exp_avg_sq, st->tau >= 0.0f ? st->q_target : nullptr, st->step, st->lr, st->beta1, st->beta2, st->adam_eps, st->tau, 0);
    MI_CHECK_ARG(st->tau < 0.0f || st->q_target, "critic step: q_target is NULL but tau >= 0");
    *out = sac_dw2_args_t{(float*)st->workspace, st->batch, 0, ws_kp(st->batch) / SR, 0, 1.0 / st->batch, st->grads, st->losses, opt};
    return MI_OK;
}
extern "C" int mi_sac_critic_step(const mi_sac_critic_step_t* st, void* stream) {
    sac_dw2_args_t a;
    if (const int rc = sac_critic_step_args(st, &a)) return rc;
    return sac_launch_grads(st->workspace, st->batch, 0, 1.0 / st->batch, st->grads, st->losses, a.opt, (hipStream_t)stream);
}
extern "C" int mi_sac_act_step_carry(void* handle, const float* actor, int64_t global_step, int64_t slots, int64_t learning_starts, float* obs_cur,
                                     float* observations, float* actions, float* rewards, uint8_t* terminated, const float* forced_actions,
                                     const float* forced_eps, const double* forced_resets, mi_episode_t* episodes, int32_t* episode_stats, int max_ep,
                                     const mi_sac_critic_step_t* step, void* stream) {
    if (!step) return sac_act_impl(handle, actor, global_step, slots, learning_starts, obs_cur, observations, actions, rewards, terminated, forced_actions, forced_eps,
                                   forced_resets, episodes, episode_stats, max_ep, nullptr, stream);
    sac_dw2_args_t a;
    if (const int rc = sac_critic_step_args(step, &a)) return rc;
    if (const int rc = sac_act_check(handle, actor, global_step, slots, obs_cur, observations, actions, rewards, terminated, episodes, max_ep)) return rc;   // nothing enqueued yet
    if (ws_kp(step->batch) > SAC_FUSED_KP) {   // large batches: the step is two launches of its own (K-split GEMM, assembly); nothing to carry
        if (const int rc = mi_sac_critic_step(step, stream)) return rc;
        return sac_act_impl(handle, actor, global_step, slots, learning_starts, obs_cur, observations, actions, rewards, terminated, forced_actions, forced_eps, forced_resets,
                            episodes, episode_stats, max_ep, nullptr, stream);
    }
    return sac_act_impl(handle, actor, global_step, slots, learning_starts, obs_cur, observations, actions, rewards, terminated, forced_actions, forced_eps, forced_resets,
                        episodes, episode_stats, max_ep, &a, stream);
}


// How many sibling workgroups a row group gets.  Siblings WAIT for each other inside the launch, so every workgroup of the launch must be resident at once: these
// kernels run one workgroup per CU (256 + 68..128 registers per lane), and the launch is kept to half the chip (row groups x roles + the owed-alpha workgroups
// <= CUs / 2), which leaves room even when something else holds part of the device.  Beyond that the single-workgroup form (no waits) runs.
// CUs this process can really use: the device's count, cut down by a CU mask in the environment (HSA_CU_MASK = "<gpu list>:<cu list>[;...]", ROC_GLOBAL_CU_MASK =
// hex bit mask) — hipDeviceProp_t.multiProcessorCount does not see those — and by mi_sac_set_max_cus (the caller knows about other tenants of the chip).  Since round 3
// the count only steers PERFORMANCE (how many sibling roles pay off); the waits no longer need the whole grid resident.
static int g_sac_max_cus = 0;
static int parse_index_list_count(const char* p, const char* end, int only /* -1: count all; else 1 if `only` is in the list */) {
    int n = 0;
    while (p < end) {
        char* q;
        long a = strtol(p, &q, 0), b = a;
        if (q == p) break;
        if (q < end && *q == '-') { const char* r = q + 1; b = strtol(r, &q, 0); if (q == r) b = a; }
        if (b < a) { const long tmp = a; a = b; b = tmp; }
        if (only < 0) n += (int)(b - a + 1); else if (only >= a && only <= b) n = 1;
        p = q;
        while (p < end && (*p == ',' || *p == ' ')) ++p;
    }
    return n;
}
static int env_cu_limit(int dev, int cus) {
    if (const char* g = getenv("ROC_GLOBAL_CU_MASK")) {
        int bits = 0;
        for (const char* c = (g[0] == '0' && (g[1] == 'x' || g[1] == 'X')) ? g + 2 : g; *c; ++c) {
            const int v = (*c >= '0' && *c <= '9') ? *c - '0' : (*c >= 'a' && *c <= 'f') ? *c - 'a' + 10 : (*c >= 'A' && *c <= 'F') ? *c - 'A' + 10 : 0;
            bits += __builtin_popcount((unsigned)v);
        }
        if (bits > 0 && bits < cus) cus = bits;
    }
    if (const char* h = getenv("HSA_CU_MASK")) {
        for (const char* p = h; *p;) {
            const char* semi = strchr(p, ';'); const char* end = semi ? semi : p + strlen(p);
            const char* colon = (const char*)memchr(p, ':', (size_t)(end - p));
            if (colon && parse_index_list_count(p, colon, dev) == 1) { const int n = parse_index_list_count(colon + 1, end, -1); if (n > 0 && n < cus) cus = n; }
            p = semi ? semi + 1 : end;
        }
    }
    return cus;
}
#define SAC_MAX_DEVICES 64
static int sac_device() { int dev = 0; return (hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < SAC_MAX_DEVICES) ? dev : 0; }
static int sac_cus() {
    static int cus_of[SAC_MAX_DEVICES] = {0};   // per device: a process may drive SAC on more than one; g_sac_max_cus / g_sac_fault are process-wide knobs
    const int dev = sac_device();
    int& cus = cus_of[dev];
    if (!cus) {
        hipDeviceProp_t prop;
        cus = 256;
        if (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) cus = prop.multiProcessorCount;
        if (!getenv("MIRL_SAC_IGNORE_CU_MASK")) cus = env_cu_limit(dev, cus);   // (test hook: pretend the mask is not there, to run sibling roles on a masked chip)
    }
    return g_sac_max_cus > 0 && g_sac_max_cus < cus ? g_sac_max_cus : cus;
}
extern "C" int mi_sac_set_max_cus(int max_cus) {
    MI_CHECK_ARG(max_cus >= 0, "max_cus must be >= 0 (0 = no limit beyond the device / environment)");
    g_sac_max_cus = max_cus;
    return MI_OK;
}
extern "C" int mi_sac_usable_cus(void) { return sac_cus(); }

// ---- launch status: ONE host-pinned, device-mapped word per process.  A wait that times out inside a kernel stores a code there (system scope); every mi_sac_*
// update call reads it on the host first — a plain load, no synchronisation — and refuses with MI_ESTATE once it is set (sticky until mi_sac_clear_error).
static unsigned int* g_sac_status_host = nullptr;
static bool g_sac_status_dev[SAC_MAX_DEVICES] = {false};   // sac_status_word is a per-device symbol: every device this process runs SAC on gets the pointer
static int g_sac_fault = 0;   // mi_sac_test_fault
static int sac_status_init() {
    const int dev = sac_device();
    if (g_sac_status_host && g_sac_status_dev[dev]) return MI_OK;
    if (!g_sac_status_host) {
        unsigned int* h = nullptr;
        MI_HIP(hipHostMalloc((void**)&h, 64, hipHostMallocMapped | hipHostMallocPortable));
        h[0] = 0u;
        g_sac_status_host = h;
    }
    unsigned int* d = nullptr;
    MI_HIP(hipHostGetDevicePointer((void**)&d, g_sac_status_host, 0));
    MI_HIP(hipMemcpyToSymbol(HIP_SYMBOL(sac_status_word), &d, sizeof(d)));   // the CURRENT device's copy of the symbol
    g_sac_status_dev[dev] = true;
    return MI_OK;
}
static int sac_status_check(const char* who) {
    if (const int rc = sac_status_init()) return rc;
    const unsigned code = __atomic_load_n(g_sac_status_host, __ATOMIC_RELAXED);
    if (code) {
        mi_set_error("%s: an earlier SAC launch of this process timed out waiting for a sibling workgroup (%s not published within 100 ms): its gradients and losses are "
                     "NaN, and no optimizer step has been applied since (parameters, Adam moments and targets are intact); call mi_sac_clear_error to go on", who, code == SAC_FAULT_EPOCH ? "the owed alpha step's epoch" : "a row hand-off word");
        return MI_ESTATE;
    }
    return MI_OK;
}
extern "C" int mi_sac_check(void* stream, int wait) {
    if (wait) MI_HIP(hipStreamSynchronize((hipStream_t)stream));
    return sac_status_check("mi_sac_check");
}
extern "C" int mi_sac_clear_error(void* workspace, int batch, void* stream) {
    if (const int rc = sac_status_init()) return rc;
    if (workspace) {   // ticket, hand-off words, stash, partials, epoch word: everything a broken launch may have left half-written
        MI_CHECK_ARG(batch > 0, "batch must be positive");
        const size_t from = ws_part_off(batch) + (size_t)GEMM_MAX_SPLIT * 3 * SA_H * SA_H, to = ws_epoch_off(batch) + 4;
        MI_HIP(hipMemsetAsync((float*)workspace + from, 0, (to - from) * sizeof(float), (hipStream_t)stream));
    }
    const unsigned int zero = 0u;
    MI_HIP(hipMemcpyToSymbolAsync(HIP_SYMBOL(sac_fault_word), &zero, sizeof(zero), 0, hipMemcpyHostToDevice, (hipStream_t)stream));   // this device's optimizer steps run again
    MI_HIP(hipStreamSynchronize((hipStream_t)stream));
    __atomic_store_n(g_sac_status_host, 0u, __ATOMIC_RELAXED);
    return MI_OK;
}
extern "C" int mi_sac_test_fault(int mode) {
    MI_CHECK_ARG(mode >= 0 && mode <= 3, "mode: bit 0 = siblings skip their hand-off words, bit 1 = the owed alpha role does not publish its epoch");
    g_sac_fault = mode;
    return MI_OK;
}
static int sac_roles(int nrg, int n_lp, int max_roles) {
    for (int r = max_roles; r > 1; r >>= 1) if (nrg * r + n_lp <= sac_cus() / 2) return r;
    return 1;
}
static bool sac_owed_fits(int nrg) { return 2 * nrg <= sac_cus() / 2; }
extern "C" int mi_sac_owed_alpha_fits(int batch) { return batch > 0 && sac_owed_fits(ws_kp(batch) / SR) ? 1 : 0; }   // row groups (single role at least... with one sibling) + as many owed workgroups

static sac_alpha_t sac_make_alpha(float target_entropy, float inv_count, float* log_alpha, float* m, float* v, int64_t step, double lr, float* alpha, float* out,
                                  unsigned int* ticket);
static sac_owed_t sac_make_owed(const mi_sac_owed_alpha_t* o, int batch, uint64_t seed, void* workspace) {
    sac_owed_t w; memset(&w, 0, sizeof(w));
    if (!o) return w;
    unsigned int* ticket = (unsigned int*)((float*)workspace + ws_part_off(batch) + (size_t)GEMM_MAX_SPLIT * 3 * SA_H * SA_H);
    w.n_lp = ws_kp(batch) / SR; w.epoch = o->epoch; w.slot = o->stash_slot & 1; w.seed = seed; w.update = o->update_index;
    w.al = sac_make_alpha(o->target_entropy, 1.0f / (float)batch, o->log_alpha, o->exp_avg, o->exp_avg_sq, o->step, o->lr, o->alpha, o->out, ticket);
    return w;
}
static int sac_check_owed(const mi_sac_owed_alpha_t* o, int batch) {
    if (!o) return MI_OK;
    MI_CHECK_ARG(o->log_alpha && o->exp_avg && o->exp_avg_sq && o->alpha && o->step >= 1, "owed alpha step: NULL state or bad step");
    MI_CHECK_ARG(o->stash_slot == 0 || o->stash_slot == 1, "owed alpha step: stash_slot must be 0 or 1");
    MI_CHECK_ARG(sac_owed_fits(ws_kp(batch) / SR), "an owed alpha step rides only on launches that leave at least half of the CUs idle (batch <= 1024 on 256 CUs)");
    return MI_OK;
}

static int sac_critic_impl(float* q, float* q_target, const float* actor, const float* observations, const float* actions, const float* rewards,
                           const uint8_t* terminated, const int64_t* idx, int batch, int n_envs, int64_t slots, const float* eps, uint64_t seed,
                           uint64_t update_index, const float* alpha, float gamma, double inv_count, void* workspace, float* grads, float* losses,
                           const sac_opt_t& opt, uint64_t sample_update, int64_t sample_upper, const sac_owed_t& ow_in, hipStream_t s, bool defer_step = false) {
    const int nrg = ws_kp(batch) / SR;
    if (const int rc = sac_status_check("SAC critic update")) return rc;
    sac_owed_t ow = ow_in; ow.fault = g_sac_fault;
    {
        mi_prof_scope prof(MI_PROF_SAC_CRITIC, s);
        const sac_wt_t wt = {shadow_valid(actor), shadow_valid(q), shadow_valid(q_target)};
        const dim3 grid(nrg + ow.n_lp, sac_roles(nrg, ow.n_lp, 4));
        if (wt.actor && wt.q && wt.qt)
            sac_critic_kernel<true><<<grid, SA_THREADS, 0, s>>>(q, q_target, actor, observations, actions, rewards, terminated, idx, batch, n_envs, (long long)slots, eps, seed, update_index, alpha, gamma,
                                                                (float)inv_count, (float*)workspace, sample_update, (uint64_t)sample_upper, (int64_t*)idx, ow, wt);
        else
            sac_critic_kernel<false><<<grid, SA_THREADS, 0, s>>>(q, q_target, actor, observations, actions, rewards, terminated, idx, batch, n_envs, (long long)slots, eps, seed, update_index, alpha, gamma,
                                                                 (float)inv_count, (float*)workspace, sample_update, (uint64_t)sample_upper, (int64_t*)idx, ow, sac_wt_t{});
    }
    MI_LAUNCH_CHECK();
    if (defer_step) return MI_OK;   // mi_sac_critic_update_deferred: the caller owes mi_sac_critic_step / mi_sac_act_step_carry
    return sac_launch_grads(workspace, batch, 0, inv_count, grads, losses, opt, s);
}

extern "C" int mi_sac_critic_grad(const float* q, const float* q_target, const float* actor, const float* observations, const float* actions,
                                  const float* rewards, const uint8_t* terminated, const int64_t* idx, int batch, int n_envs, int64_t slots,
                                  const float* eps, uint64_t seed, uint64_t update_index, const float* alpha, float gamma, double inv_count,
                                  void* workspace, float* grads, float* losses, void* stream) {
    MI_CHECK_ARG(q && q_target && actor && observations && actions && rewards && terminated && idx && alpha && workspace && grads, "NULL pointer");
    MI_CHECK_ARG(batch > 0 && n_envs > 0 && slots >= 2, "bad sizes");
    return sac_critic_impl((float*)q, (float*)q_target, actor, observations, actions, rewards, terminated, idx, batch, n_envs, slots, eps, seed, update_index, alpha,
                           gamma, inv_count, workspace, grads, losses, sac_no_opt(), 0, 0, sac_owed_t{}, (hipStream_t)stream);
}

extern "C" int mi_sac_critic_update_owed(float* q, float* q_target, const float* actor, const float* observations, const float* actions, const float* rewards,
                                         const uint8_t* terminated, int64_t* idx, int batch, int n_envs, int64_t slots, const float* eps, uint64_t seed,
                                         uint64_t update_index, const float* alpha, float gamma, void* workspace, float* grads, float* losses, float* exp_avg,
                                         float* exp_avg_sq, int64_t step, double lr, double beta1, double beta2, double adam_eps, float tau, uint64_t sample_update,
                                         int64_t sample_upper, const mi_sac_owed_alpha_t* owed, void* stream) {
    MI_CHECK_ARG(sample_upper >= 0, "sample_upper must be >= 0");
    MI_CHECK_ARG(q && q_target && actor && observations && actions && rewards && terminated && idx && alpha && workspace && grads && exp_avg && exp_avg_sq, "NULL pointer");
    MI_CHECK_ARG(batch > 0 && n_envs > 0 && slots >= 2 && step >= 1, "bad sizes");
    if (const int rc = sac_check_owed(owed, batch)) return rc;
    return sac_critic_impl(q, q_target, actor, observations, actions, rewards, terminated, idx, batch, n_envs, slots, eps, seed, update_index, alpha, gamma,
                           1.0 / batch, workspace, grads, losses, sac_make_opt(q, exp_avg, exp_avg_sq, tau >= 0.0f ? q_target : nullptr, step, lr, beta1, beta2, adam_eps, tau, 0),
                           sample_update, sample_upper, sac_make_owed(owed, batch, seed, workspace), (hipStream_t)stream);
}
extern "C" int mi_sac_critic_update_deferred(const float* q, const float* q_target, const float* actor, const float* observations, const float* actions, const float* rewards,
                                             const uint8_t* terminated, int64_t* idx, int batch, int n_envs, int64_t slots, const float* eps, uint64_t seed,
                                             uint64_t update_index, const float* alpha, float gamma, void* workspace, uint64_t sample_update, int64_t sample_upper,
                                             const mi_sac_owed_alpha_t* owed, void* stream) {
    MI_CHECK_ARG(sample_upper >= 0, "sample_upper must be >= 0");
    MI_CHECK_ARG(q && q_target && actor && observations && actions && rewards && terminated && idx && alpha && workspace, "NULL pointer");
    MI_CHECK_ARG(batch > 0 && n_envs > 0 && slots >= 2, "bad sizes");
    if (const int rc = sac_check_owed(owed, batch)) return rc;
    return sac_critic_impl((float*)q, (float*)q_target, actor, observations, actions, rewards, terminated, idx, batch, n_envs, slots, eps, seed, update_index, alpha, gamma,
                           1.0 / batch, workspace, nullptr, nullptr, sac_no_opt(), sample_update, sample_upper, sac_make_owed(owed, batch, seed, workspace), (hipStream_t)stream,
                           true);
}
extern "C" int mi_sac_critic_update(float* q, float* q_target, const float* actor, const float* observations, const float* actions, const float* rewards,
                                    const uint8_t* terminated, int64_t* idx, int batch, int n_envs, int64_t slots, const float* eps, uint64_t seed,
                                    uint64_t update_index, const float* alpha, float gamma, void* workspace, float* grads, float* losses, float* exp_avg,
                                    float* exp_avg_sq, int64_t step, double lr, double beta1, double beta2, double adam_eps, float tau, uint64_t sample_update,
                                    int64_t sample_upper, void* stream) {
    return mi_sac_critic_update_owed(q, q_target, actor, observations, actions, rewards, terminated, idx, batch, n_envs, slots, eps, seed, update_index, alpha, gamma,
                                     workspace, grads, losses, exp_avg, exp_avg_sq, step, lr, beta1, beta2, adam_eps, tau, sample_update, sample_upper, nullptr, stream);
}

static int sac_actor_impl(float* actor, const float* q, const float* observations, const int64_t* idx, int batch, const float* eps, uint64_t seed,
                          uint64_t update_index, const float* alpha, double inv_count, void* workspace, float* grads, float* out, const sac_opt_t& opt, const sac_owed_t& ow_in,
                          hipStream_t s) {
    if (const int rc = sac_status_check("SAC actor update")) return rc;
    sac_owed_t ow = ow_in; ow.fault = g_sac_fault;
    {
        mi_prof_scope prof(MI_PROF_SAC_ACTOR, s);
        const int nrg = ws_kp(batch) / SR;
        // the batch observations go to the stash slot that a debt carried by THIS launch does not read (no debt: slot 0)
        const sac_wt_t wt = {shadow_valid(actor), shadow_valid(q), nullptr};
        const dim3 grid(nrg + ow.n_lp, sac_roles(nrg, ow.n_lp, 2));
        if (wt.actor && wt.q)
            sac_actor_kernel<true><<<grid, SA_THREADS, 0, s>>>(actor, q, observations, idx, batch, eps, seed, update_index, alpha, (float)inv_count, (float*)workspace, 0, sac_alpha_t{}, ow,
                                                               ow.n_lp ? (ow.slot ^ 1) : 0, wt);
        else
            sac_actor_kernel<false><<<grid, SA_THREADS, 0, s>>>(actor, q, observations, idx, batch, eps, seed, update_index, alpha, (float)inv_count, (float*)workspace, 0, sac_alpha_t{}, ow,
                                                                ow.n_lp ? (ow.slot ^ 1) : 0, sac_wt_t{});
    }
    MI_LAUNCH_CHECK();
    return sac_launch_grads(workspace, batch, 1, inv_count, grads, out, opt, s);
}

extern "C" int mi_sac_actor_grad(const float* actor, const float* q, const float* observations, const int64_t* idx, int batch, const float* eps,
                                 uint64_t seed, uint64_t update_index, const float* alpha, double inv_count, void* workspace, float* grads, float* out,
                                 void* stream) {
    MI_CHECK_ARG(actor && q && observations && idx && alpha && workspace && grads, "NULL pointer");
    MI_CHECK_ARG(batch > 0, "bad sizes");
    return sac_actor_impl((float*)actor, q, observations, idx, batch, eps, seed, update_index, alpha, inv_count, workspace, grads, out, sac_no_opt(), sac_owed_t{}, (hipStream_t)stream);
}

extern "C" int mi_sac_actor_update_owed(float* actor, const float* q, const float* observations, const int64_t* idx, int batch, const float* eps, uint64_t seed,
                                        uint64_t update_index, const float* alpha, void* workspace, float* grads, float* out, float* exp_avg, float* exp_avg_sq,
                                        int64_t step, double lr, double beta1, double beta2, double adam_eps, const mi_sac_owed_alpha_t* owed, void* stream) {
    MI_CHECK_ARG(actor && q && observations && idx && alpha && workspace && grads && exp_avg && exp_avg_sq, "NULL pointer");
    MI_CHECK_ARG(batch > 0 && step >= 1, "bad sizes");
    if (const int rc = sac_check_owed(owed, batch)) return rc;
    return sac_actor_impl(actor, q, observations, idx, batch, eps, seed, update_index, alpha, 1.0 / batch, workspace, grads, out,
                          sac_make_opt(actor, exp_avg, exp_avg_sq, nullptr, step, lr, beta1, beta2, adam_eps, 0.0f, 1), sac_make_owed(owed, batch, seed, workspace),
                          (hipStream_t)stream);
}
extern "C" int mi_sac_actor_update(float* actor, const float* q, const float* observations, const int64_t* idx, int batch, const float* eps, uint64_t seed,
                                   uint64_t update_index, const float* alpha, void* workspace, float* grads, float* out, float* exp_avg, float* exp_avg_sq,
                                   int64_t step, double lr, double beta1, double beta2, double adam_eps, void* stream) {
    return mi_sac_actor_update_owed(actor, q, observations, idx, batch, eps, seed, update_index, alpha, workspace, grads, out, exp_avg, exp_avg_sq, step, lr, beta1, beta2,
                                    adam_eps, nullptr, stream);
}

static int adam_launch(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int n, int64_t step, double lr, double beta1, double beta2, double eps,
                       const uint32_t* gate, void* stream);
static int polyak_launch(float* target, const float* param, int n, float tau, const uint32_t* gate, void* stream);
static int alpha_adam_launch(const float* mean_logp, float target_entropy, float* log_alpha, float* exp_avg, float* exp_avg_sq, int64_t step, double lr, float* alpha,
                             float* out, const uint32_t* gate, void* stream);
// P2P carrier with one (or two) ranks per device: the exchange rides INSIDE the gradient assembly launch (sac_opt_t.xa / world / gate; exactly one mi_comm_p2p_next per
// assembling launch, on every rank) — the all-reduce, Adam and polyak launches of the sequence below disappear.  *fused = false: not that carrier (take the sequence).
static int sac_opt_p2p(sac_opt_t* o, void* comm, size_t n_words, hipStream_t s, bool* fused) {
    *fused = comm != nullptr && mi_comm_p2p_fused_ok(comm);
    if (!*fused) return MI_OK;
    int world = 0;
    const int rc = mi_comm_p2p_next(comm, n_words, &o->xa, &world, s);
    if (rc) return rc;
    o->world = world; o->gate = mi_comm_gate(comm);
    return MI_OK;
}
// FAIL-SAFE (mi_common.h): a timed-out wait of the P2P carrier withholds the optimizer steps behind the exchange (adam / polyak / alpha read the carrier's status word
// with their state), and the next call returns MI_ESTATE at its entry.
// ---- sharded runs, ONE C call per update (the pattern of mi_ppo_update_sharded): the *_grad launches with the share scaled by 1 / (world * batch), an in-stream RCCL
// SUM all-reduce of the caller's {gradient, 2 scalars} buffer, then mi_adam (and mi_polyak) — exactly the launches of the host-sequenced route (sac_engine.py: *_grad,
// torch.distributed.all_reduce, Adam.step, update_targets), so the two agree bit for bit; no Python between launches.
extern "C" int mi_sac_critic_update_sharded(float* q, float* q_target, const float* actor, const float* observations, const float* actions, const float* rewards,
                                            const uint8_t* terminated, const int64_t* idx, int batch, int n_envs, int64_t slots, const float* eps, uint64_t seed,
                                            uint64_t update_index, const float* alpha, float gamma, void* workspace, float* qbuf /* grads [2 Q_NP] + losses [2] */,
                                            float* exp_avg, float* exp_avg_sq, int64_t step, double lr, double beta1, double beta2, double adam_eps, float tau,
                                            void* comm, void* stream) {
    MI_CHECK_ARG(qbuf && exp_avg && exp_avg_sq && step >= 1, "NULL optimizer state / bad step");
    int world = 1;
    if (comm) {
        if (const int rc = mi_comm_poll_impl(comm)) return rc;
        if (const int rc = mi_comm_info(comm, &world, nullptr, nullptr, nullptr)) return rc;
    }
    {   // P2P carrier: the critic launch + ONE assembly launch that exchanges, steps and polyak-averages (see sac_opt_t)
        MI_CHECK_ARG(q && actor && observations && actions && rewards && terminated && idx && alpha && workspace, "NULL pointer");
        MI_CHECK_ARG(batch > 0 && n_envs > 0 && slots >= 2, "bad sizes");
        sac_opt_t o = sac_make_opt(q, exp_avg, exp_avg_sq, tau < 0.0f ? nullptr : q_target, step, lr, beta1, beta2, adam_eps, tau < 0.0f ? 0.0f : tau, 0);
        bool fused = false;
        // (the sequence number is drawn here, in front of the critic launch: the assembly launch behind it is the one launch that publishes and consumes those lines)
        if (const int rc0 = sac_opt_p2p(&o, comm, (size_t)2 * SQ_NP + 2, (hipStream_t)stream, &fused)) return rc0;
        if (fused)
            return sac_critic_impl(q, q_target, actor, observations, actions, rewards, terminated, idx, batch, n_envs, slots, eps, seed, update_index, alpha, gamma,
                                   1.0 / ((double)batch * world), workspace, qbuf, qbuf + 2 * SQ_NP, o, 0, 0, sac_owed_t{}, (hipStream_t)stream);
    }
    int rc = mi_sac_critic_grad(q, q_target, actor, observations, actions, rewards, terminated, idx, batch, n_envs, slots, eps, seed, update_index, alpha, gamma,
                                1.0 / ((double)batch * world), workspace, qbuf, qbuf + 2 * SQ_NP, stream);
    if (rc) return rc;
    if (comm) {
        mi_prof_scope prof(MI_PROF_COMM_GRAD, (hipStream_t)stream);
        rc = mi_comm_allreduce_impl(comm, qbuf, (size_t)2 * SQ_NP + 2, 0, (hipStream_t)stream);
        if (rc) return rc;
    }
    rc = adam_launch(q, qbuf, exp_avg, exp_avg_sq, 2 * SQ_NP, step, lr, beta1, beta2, adam_eps, mi_comm_gate(comm), stream);
    if (rc || tau < 0.0f) return rc;
    return polyak_launch(q_target, q, 2 * SQ_NP, tau, mi_comm_gate(comm), stream);
}
extern "C" int mi_sac_actor_update_sharded(float* actor, const float* q, const float* observations, const int64_t* idx, int batch, const float* eps, uint64_t seed,
                                           uint64_t update_index, const float* alpha, void* workspace, float* abuf /* grads [ACTOR_NP] + out [2] */, float* exp_avg,
                                           float* exp_avg_sq, int64_t step, double lr, double beta1, double beta2, double adam_eps, void* comm, void* stream) {
    MI_CHECK_ARG(abuf && exp_avg && exp_avg_sq && step >= 1, "NULL optimizer state / bad step");
    int world = 1;
    if (comm) {
        if (const int rc = mi_comm_poll_impl(comm)) return rc;
        if (const int rc = mi_comm_info(comm, &world, nullptr, nullptr, nullptr)) return rc;
    }
    {
        MI_CHECK_ARG(actor && q && observations && idx && alpha && workspace, "NULL pointer");
        MI_CHECK_ARG(batch > 0, "bad sizes");
        sac_opt_t o = sac_make_opt(actor, exp_avg, exp_avg_sq, nullptr, step, lr, beta1, beta2, adam_eps, 0.0f, 1);
        bool fused = false;
        if (const int rc0 = sac_opt_p2p(&o, comm, (size_t)AC_NP + 2, (hipStream_t)stream, &fused)) return rc0;
        if (fused)
            return sac_actor_impl(actor, q, observations, idx, batch, eps, seed, update_index, alpha, 1.0 / ((double)batch * world), workspace, abuf, abuf + AC_NP, o, sac_owed_t{},
                                  (hipStream_t)stream);
    }
    int rc = mi_sac_actor_grad(actor, q, observations, idx, batch, eps, seed, update_index, alpha, 1.0 / ((double)batch * world), workspace, abuf, abuf + AC_NP, stream);
    if (rc) return rc;
    if (comm) {
        mi_prof_scope prof(MI_PROF_COMM_GRAD, (hipStream_t)stream);
        rc = mi_comm_allreduce_impl(comm, abuf, (size_t)AC_NP + 2, 0, (hipStream_t)stream);
        if (rc) return rc;
    }
    return adam_launch(actor, abuf, exp_avg, exp_avg_sq, AC_NP, step, lr, beta1, beta2, adam_eps, mi_comm_gate(comm), stream);
}
// ================================================ alpha, Adam, polyak ============================================================
// one wave.  mean_in: nullable device scalar holding the (already all-reduced) mean log-prob; NULL = sum this rank's slabs (lane-strided, then the fixed DPP tree)
__global__ void __launch_bounds__(64)
sac_alpha_kernel(const float* __restrict__ ws, int batch, int n_slabs, const float* __restrict__ mean_in, float* __restrict__ mean_out, sac_alpha_t al,
                 const uint32_t* __restrict__ gate, const p2p_args_t xa, int world) {
    // world > 0 (sharded run on the P2P carrier): the mean summed from this rank's slabs is this rank's SHARE (inv_count = 1 / (world batch)); thread 0 exchanges it as line
    // 0 (rank-ordered sum), leaves the all-reduced mean in mean_out (what the sequence's all-reduce leaves there) and steps log_alpha: slab sum, all-reduce and step in ONE launch
    if (world == 0 && mi_gate_closed(gate)) return;   // mean_in came out of a timed-out exchange: the alpha step is withheld
    float mean_lp;
    if (mean_in) mean_lp = mean_in[0];
    else {
        const float* slabs = ws + ws_slab_off(batch);
        float s = 0.0f;
        for (int b = threadIdx.x; b < n_slabs; b += 64) s += slabs[(size_t)b * SLAB + 1795];
        mean_lp = wave_sum(s) * al.inv_count;
    }
    if (threadIdx.x != 0) return;
    if (world > 0) {
        mean_lp = p2p_exchange_rt(xa, world, 0, mean_lp);
        if (mean_out) mean_out[0] = mean_lp;
        if (!mi_gate_closed(gate)) sac_alpha_apply(al, mean_lp);
        return;
    }
    if (mean_out) { mean_out[0] = mean_lp; return; }
    sac_alpha_apply(al, mean_lp);
}

static sac_alpha_t sac_make_alpha(float target_entropy, float inv_count, float* log_alpha, float* m, float* v, int64_t step, double lr, float* alpha, float* out,
                                  unsigned int* ticket) {
    sac_alpha_t a;
    const double b1 = 0.9, b2 = 0.999, bc1 = 1.0 - pow(b1, (double)step), bc2 = 1.0 - pow(b2, (double)step);
    a.log_alpha = log_alpha; a.m = m; a.v = v; a.alpha = alpha; a.out = out; a.ticket = ticket; a.target_entropy = target_entropy; a.inv_count = inv_count;
    a.w1 = (float)(1.0 - b1); a.b2 = (float)b2; a.w2 = (float)(1.0 - b2); a.step_size = (float)(lr / bc1); a.rbc2 = (float)(1.0 / sqrt(bc2)); a.eps = 1e-8f;
    return a;
}

static int sac_launch_logp(const float* actor, const float* observations, const int64_t* idx, int batch, const float* eps, uint64_t seed,
                           uint64_t update_index, void* workspace, const sac_alpha_t& al, hipStream_t s) {
    if (const int rc = sac_status_check("SAC log-prob pass")) return rc;
    {
        mi_prof_scope prof(MI_PROF_SAC_LOGP, s);
        const sac_wt_t wt = {shadow_valid(actor), nullptr, nullptr};
        if (wt.actor)
            sac_actor_kernel<true><<<ws_kp(batch) / SR, SA_THREADS, 0, s>>>(actor, nullptr, observations, idx, batch, eps, seed, update_index, nullptr, 0.0f, (float*)workspace, 1, al, sac_owed_t{}, 0, wt);
        else
            sac_actor_kernel<false><<<ws_kp(batch) / SR, SA_THREADS, 0, s>>>(actor, nullptr, observations, idx, batch, eps, seed, update_index, nullptr, 0.0f, (float*)workspace, 1, al, sac_owed_t{}, 0, sac_wt_t{});
    }
    MI_LAUNCH_CHECK();
    return MI_OK;
}

// one launch: the log-prob workgroups, the last of which to finish does the alpha step (ticket word behind the GEMM partials, self-resetting)
extern "C" int mi_sac_alpha_step(const float* actor, const float* observations, const int64_t* idx, int batch, const float* eps, uint64_t seed,
                                 uint64_t update_index, float target_entropy, float* log_alpha, float* exp_avg, float* exp_avg_sq, int64_t step,
                                 double lr, float* alpha, float* out, void* workspace, void* stream) {
    MI_CHECK_ARG(actor && observations && idx && log_alpha && exp_avg && exp_avg_sq && alpha && workspace, "NULL pointer");
    MI_CHECK_ARG(batch > 0 && step >= 1, "bad arguments");
    unsigned int* ticket = (unsigned int*)((float*)workspace + ws_part_off(batch) + (size_t)GEMM_MAX_SPLIT * 3 * SA_H * SA_H);
    return sac_launch_logp(actor, observations, idx, batch, eps, seed, update_index, workspace,
                           sac_make_alpha(target_entropy, 1.0f / (float)batch, log_alpha, exp_avg, exp_avg_sq, step, lr, alpha, out, ticket), (hipStream_t)stream);
}

// an owed alpha step that found no launch to ride on (the state is read, the run ends, ...): its workgroups alone
__global__ void __launch_bounds__(SA_THREADS) sac_owed_alpha_kernel(const float* __restrict__ actor, int batch, float* __restrict__ ws_, sac_owed_t ow) {
    __shared__ sac_smem sm;
    sac_owed_alpha_role(sm, actor, batch, ws_, ow, (int)blockIdx.x);
}
extern "C" int mi_sac_alpha_step_owed(const float* actor, int batch, uint64_t seed, const mi_sac_owed_alpha_t* owed, void* workspace, void* stream) {
    MI_CHECK_ARG(actor && owed && workspace && batch > 0, "NULL pointer");
    if (const int rc = sac_check_owed(owed, batch)) return rc;
    if (const int rc = sac_status_check("mi_sac_alpha_step_owed")) return rc;
    sac_owed_t ow = sac_make_owed(owed, batch, seed, workspace);
    ow.fault = g_sac_fault;
    mi_prof_scope prof(MI_PROF_SAC_LOGP, (hipStream_t)stream);
    sac_owed_alpha_kernel<<<ow.n_lp, SA_THREADS, 0, (hipStream_t)stream>>>(actor, batch, (float*)workspace, ow);
    MI_LAUNCH_CHECK();
    return MI_OK;
}

extern "C" int mi_sac_mean_logp(const float* actor, const float* observations, const int64_t* idx, int batch, const float* eps, uint64_t seed,
                                uint64_t update_index, double inv_count, float* mean_logp, void* workspace, void* stream) {
    MI_CHECK_ARG(actor && observations && idx && mean_logp && workspace && batch > 0, "bad arguments");
    const int rc = sac_launch_logp(actor, observations, idx, batch, eps, seed, update_index, workspace, sac_alpha_t{}, (hipStream_t)stream);
    if (rc) return rc;
    sac_alpha_t al{}; al.inv_count = (float)inv_count;
    sac_alpha_kernel<<<1, 64, 0, (hipStream_t)stream>>>((const float*)workspace, batch, ws_kp(batch) / SR, nullptr, mean_logp, al, nullptr, p2p_args_t{}, 0);
    MI_LAUNCH_CHECK();
    return MI_OK;
}

static int alpha_adam_launch(const float* mean_logp, float target_entropy, float* log_alpha, float* exp_avg, float* exp_avg_sq, int64_t step, double lr, float* alpha,
                             float* out, const uint32_t* gate, void* stream) {
    MI_CHECK_ARG(mean_logp && log_alpha && exp_avg && exp_avg_sq && alpha && step >= 1, "bad arguments");
    sac_alpha_kernel<<<1, 64, 0, (hipStream_t)stream>>>(nullptr, 0, 0, mean_logp, nullptr,
                                                       sac_make_alpha(target_entropy, 0.0f, log_alpha, exp_avg, exp_avg_sq, step, lr, alpha, out, nullptr), gate, p2p_args_t{}, 0);
    MI_LAUNCH_CHECK();
    return MI_OK;
}
extern "C" int mi_sac_alpha_adam(const float* mean_logp, float target_entropy, float* log_alpha, float* exp_avg, float* exp_avg_sq, int64_t step,
                                 double lr, float* alpha, float* out, void* stream) {
    return alpha_adam_launch(mean_logp, target_entropy, log_alpha, exp_avg, exp_avg_sq, step, lr, alpha, out, nullptr, stream);
}

// (the third sharded one-call route; it sits here, behind the log-prob launch and the alpha kernel it uses)
extern "C" int mi_sac_alpha_step_sharded(const float* actor, const float* observations, const int64_t* idx, int batch, const float* eps, uint64_t seed,
                                         uint64_t update_index, float target_entropy, float* log_alpha, float* exp_avg, float* exp_avg_sq, int64_t step, double lr,
                                         float* alpha, float* out, float* mean_logp /* dev f32 [1] scratch */, void* workspace, void* comm, void* stream) {
    MI_CHECK_ARG(mean_logp != nullptr, "mean_logp scratch is NULL");
    int world = 1;
    if (comm) {
        if (const int rc = mi_comm_poll_impl(comm)) return rc;
        if (const int rc = mi_comm_info(comm, &world, nullptr, nullptr, nullptr)) return rc;
    }
    if (comm && mi_comm_p2p_fused_ok(comm)) {   // P2P carrier: the log-prob launch + ONE one-wave launch that sums the slabs, exchanges the mean and steps log_alpha
        MI_CHECK_ARG(actor && observations && idx && log_alpha && exp_avg && exp_avg_sq && alpha && workspace && batch > 0 && step >= 1, "bad arguments");
        int rc0 = sac_launch_logp(actor, observations, idx, batch, eps, seed, update_index, workspace, sac_alpha_t{}, (hipStream_t)stream);
        if (rc0) return rc0;
        p2p_args_t xa;
        int pw = 0;
        rc0 = mi_comm_p2p_next(comm, 1, &xa, &pw, (hipStream_t)stream);
        if (rc0) return rc0;
        const sac_alpha_t al = sac_make_alpha(target_entropy, (float)(1.0 / ((double)batch * world)), log_alpha, exp_avg, exp_avg_sq, step, lr, alpha, out, nullptr);
        sac_alpha_kernel<<<1, 64, 0, (hipStream_t)stream>>>((const float*)workspace, batch, ws_kp(batch) / SR, nullptr, mean_logp, al, mi_comm_gate(comm), xa, pw);
        MI_LAUNCH_CHECK();
        return MI_OK;
    }
    int rc = mi_sac_mean_logp(actor, observations, idx, batch, eps, seed, update_index, 1.0 / ((double)batch * world), mean_logp, workspace, stream);
    if (rc) return rc;
    if (comm) {
        mi_prof_scope prof(MI_PROF_COMM_STATS, (hipStream_t)stream);
        rc = mi_comm_allreduce_impl(comm, mean_logp, 1, 0, (hipStream_t)stream);
        if (rc) return rc;
    }
    return alpha_adam_launch(mean_logp, target_entropy, log_alpha, exp_avg, exp_avg_sq, step, lr, alpha, out, mi_comm_gate(comm), stream);
}


__global__ void __launch_bounds__(256) adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, int n,
                                                    float w1, float b2, float w2, float step_size, float rbc2, float eps, const uint32_t* __restrict__ gate) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n || sac_faulted() || mi_gate_closed(gate)) return;   // gate: the P2P exchange that produced g timed out (mi_common.h) - the step is withheld   // (mi_adam / mi_polyak are the SAC engines' unfused steps: they honour the device's fault word like the fused ones)
    float mi = m[i], vi = v[i];
    p[i] = mi_adam_elem(p[i], g[i], mi, vi, w1, b2, w2, step_size, rbc2, eps);
    m[i] = mi; v[i] = vi;
}

static int adam_launch(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int n, int64_t step, double lr, double beta1,
                       double beta2, double eps, const uint32_t* gate, void* stream) {
    MI_CHECK_ARG(params && grads && exp_avg && exp_avg_sq && n > 0 && step >= 1, "bad arguments");
    shadow_invalidate_range(params, (size_t)n);   // this launch does not keep a transposed copy in step
    const double bc1 = 1.0 - pow(beta1, (double)step), bc2 = 1.0 - pow(beta2, (double)step);
    adam_kernel<<<(n + 255) / 256, 256, 0, (hipStream_t)stream>>>(params, grads, exp_avg, exp_avg_sq, n, (float)(1.0 - beta1), (float)beta2,
                                                                 (float)(1.0 - beta2), (float)(lr / bc1), (float)(1.0 / sqrt(bc2)), (float)eps, gate);
    MI_LAUNCH_CHECK();
    return MI_OK;
}
extern "C" int mi_adam(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int n, int64_t step, double lr, double beta1,
                       double beta2, double eps, void* stream) {
    return adam_launch(params, grads, exp_avg, exp_avg_sq, n, step, lr, beta1, beta2, eps, nullptr, stream);
}

__global__ void __launch_bounds__(256) polyak_kernel(float* __restrict__ t, const float* __restrict__ p, int n, float tau, const uint32_t* __restrict__ gate) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n && !sac_faulted() && !mi_gate_closed(gate)) t[i] = tau * p[i] + (1.0f - tau) * t[i];
}

static int polyak_launch(float* target, const float* param, int n, float tau, const uint32_t* gate, void* stream) {
    MI_CHECK_ARG(target && param && n > 0, "bad arguments");
    shadow_invalidate_range(target, (size_t)n);
    polyak_kernel<<<(n + 255) / 256, 256, 0, (hipStream_t)stream>>>(target, param, n, tau, gate);
    MI_LAUNCH_CHECK();
    return MI_OK;
}
extern "C" int mi_polyak(float* target, const float* param, int n, float tau, void* stream) { return polyak_launch(target, param, n, tau, nullptr, stream); }

MI_INSIDE_EXPORT(sac)
