// mi_env.hip — batched CartPole-v1 stepper behind the gym-0.21 Env protocol (struct-of-arrays, fp64 state).
// Replaces TorchWrapper.step/reset + gym.make + TimeLimit + RecordEpisodeStatistics (reference ppo.py:10-22,79-84).
#include <stdarg.h>

#include "mi_common.h"

// ---- error text ------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";
void mi_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
extern "C" const char* mi_last_error(void) { return g_err; }
extern "C" int mi_version(void) { return MI_VERSION; }
#ifndef MI_SOURCE_ID
#define MI_SOURCE_ID "unknown"
#endif
extern "C" const char* mi_source_id(void) { return MI_SOURCE_ID; }   // sha256 (12 hex digits) of the sources this library was built from (csrc/Makefile: ALLSRC)

// ---- kernels (one lane per env; all accesses coalesced over the env axis) -------------------------------
__global__ void __launch_bounds__(256) env_reset_kernel(mi_env e, float* __restrict__ obs, const double* __restrict__ forced) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= e.n) return;
    double s[4];
    if (forced) {
#pragma unroll
        for (int k = 0; k < 4; ++k) s[k] = forced[4 * (size_t)i + k];
    } else {
        mi_reset_noise(e.seed, e.env_id_base + (uint64_t)i, e.episode[i], s);
    }
    e.episode[i] += 1;
    e.elapsed[i] = 0;
    e.ep_ret[i] = 0.0f;
    e.ep_len[i] = 0;
    e.x[i] = s[0]; e.x_dot[i] = s[1]; e.theta[i] = s[2]; e.theta_dot[i] = s[3];
    reinterpret_cast<float4*>(obs)[i] = make_float4((float)s[0], (float)s[1], (float)s[2], (float)s[3]);
}

__global__ void __launch_bounds__(256)
env_step_kernel(mi_env e, const int64_t* __restrict__ actions, const double* __restrict__ forced_reset, float* __restrict__ obs,
                float* __restrict__ reward, uint8_t* __restrict__ done, uint8_t* __restrict__ truncated,
                float* __restrict__ fin_ret, int32_t* __restrict__ fin_len, float* __restrict__ raw_obs) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= e.n) return;
    double s[4] = {e.x[i], e.x_dot[i], e.theta[i], e.theta_dot[i]};
    int term;
    mi_cartpole_step(s[0], s[1], s[2], s[3], (int)actions[i], term);
    if (raw_obs) reinterpret_cast<float4*>(raw_obs)[i] = make_float4((float)s[0], (float)s[1], (float)s[2], (float)s[3]);   // what gym's env.step returned, before ppo.py:128-129's reset
    const int el = e.elapsed[i] + 1;
    int trunc = 0, d = term;
    if (el >= CP_MAX_STEPS) { trunc = !term; d = 1; }
    const float ret = e.ep_ret[i] + 1.0f;
    const int len = e.ep_len[i] + 1;
    reward[i] = 1.0f;
    done[i] = (uint8_t)d;
    truncated[i] = (uint8_t)trunc;
    if (d) {
        fin_ret[i] = ret; fin_len[i] = len;
        e.ep_ret[i] = 0.0f; e.ep_len[i] = 0;
        if (forced_reset) {
#pragma unroll
            for (int k = 0; k < 4; ++k) s[k] = forced_reset[4 * (size_t)i + k];
        } else {
            mi_reset_noise(e.seed, e.env_id_base + (uint64_t)i, e.episode[i], s);
        }
        e.episode[i] += 1;
        e.elapsed[i] = 0;
    } else {
        fin_ret[i] = 0.0f; fin_len[i] = 0;
        e.ep_ret[i] = ret; e.ep_len[i] = len;
        e.elapsed[i] = el;
    }
    e.x[i] = s[0]; e.x_dot[i] = s[1]; e.theta[i] = s[2]; e.theta_dot[i] = s[3];
    reinterpret_cast<float4*>(obs)[i] = make_float4((float)s[0], (float)s[1], (float)s[2], (float)s[3]);
}

__global__ void __launch_bounds__(256) env_get_state_kernel(mi_env e, double* __restrict__ state, int32_t* __restrict__ elapsed) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= e.n) return;
    if (state && e.kind == MI_ENV_PENDULUM_V1) {   // [N,2]: theta, theta_dot
        state[2 * (size_t)i + 0] = e.x[i]; state[2 * (size_t)i + 1] = e.x_dot[i];
    } else if (state) {
        state[4 * (size_t)i + 0] = e.x[i]; state[4 * (size_t)i + 1] = e.x_dot[i];
        state[4 * (size_t)i + 2] = e.theta[i]; state[4 * (size_t)i + 3] = e.theta_dot[i];
    }
    if (elapsed) elapsed[i] = e.elapsed[i];
}

// ---- C ABI -------------------------------------------------------------------------------------------
extern "C" int mi_env_create(int kind, int n_envs, uint64_t seed, uint64_t env_id_base, void** handle) {
    MI_CHECK_ARG(handle != nullptr, "handle is NULL");
    *handle = nullptr;
    MI_CHECK_ARG(kind == MI_ENV_CARTPOLE_V1 || kind == MI_ENV_PENDULUM_V1, "unknown env kind (0 = CartPole-v1, 1 = Pendulum-v1)");
    MI_CHECK_ARG(n_envs > 0, "n_envs must be positive");
    mi_env* e = new (std::nothrow) mi_env();
    if (!e) { mi_set_error("mi_env_create: out of host memory"); return MI_ENOMEM; }
    memset(e, 0, sizeof(*e));
    e->kind = kind; e->n = n_envs; e->seed = seed; e->env_id_base = env_id_base;
    MI_HIP(hipGetDevice(&e->device));
    const size_t n = (size_t)n_envs;
    // one slab: 4 f64 + 2 u64 + 2 i32 + 1 f32 per env, and behind them the per-workgroup episode statistics (mi_common.h)
    char* slab = nullptr;
    const size_t bytes = n * (4 * 8 + 2 * 8 + 3 * 4);
    const int stats_cap = (n_envs + 3) / 4 + 1;   // the smallest workgroup of any acting kernel owns 4 envs (rollout_q4_kernel)
    const size_t bytes_all = (bytes + 15) / 16 * 16 + (size_t)stats_cap * 16;
    if (hipMalloc(&slab, bytes_all) != hipSuccess) {
        delete e;
        mi_set_error("mi_env_create: hipMalloc(%zu) failed", bytes_all);
        return MI_ENOMEM;
    }
    MI_HIP(hipMemset(slab, 0, bytes_all));
    e->x = (double*)slab; e->x_dot = e->x + n; e->theta = e->x_dot + n; e->theta_dot = e->theta + n;
    e->episode = (uint64_t*)(e->theta_dot + n); e->step_ctr = e->episode + n;
    e->elapsed = (int32_t*)(e->step_ctr + n); e->ep_len = e->elapsed + n;
    e->ep_ret = (float*)(e->ep_len + n);
    e->stats_part = (int32_t*)(slab + (bytes + 15) / 16 * 16); e->stats_cap = stats_cap; e->stats_n = -1;
    *handle = e;
    return MI_OK;
}

// ---- checkpointing (the reference has none): the whole env state is ONE slab (60 bytes per env: 4 f64 state,
// episode / step counters u64, TimeLimit / episode-length i32, episode return f32), so export / import are one device-to-device copy.
// With the counter-based RNG (keys = seed, global env id, these counters) a restored run continues bit for bit.
extern "C" size_t mi_env_state_bytes(void* handle) {
    if (!handle) return 0;
    return (size_t)((mi_env*)handle)->n * (4 * 8 + 2 * 8 + 3 * 4);
}

extern "C" int mi_env_export_state(void* handle, void* dst, void* stream) {
    MI_CHECK_ARG(handle && dst, "NULL pointer");
    mi_env* e = (mi_env*)handle;
    MI_HIP(hipMemcpyAsync(dst, e->x, mi_env_state_bytes(handle), hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return MI_OK;
}

extern "C" int mi_env_import_state(void* handle, const void* src, void* stream) {
    MI_CHECK_ARG(handle && src, "NULL pointer");
    mi_env* e = (mi_env*)handle;
    MI_HIP(hipMemcpyAsync(e->x, src, mi_env_state_bytes(handle), hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return MI_OK;
}

// ---- episode statistics kept per workgroup (mi_common.h) --------------------------------------------------------------------------
__global__ void __launch_bounds__(256) env_stats_reduce_kernel(const int32_t* __restrict__ part, int n, int32_t* __restrict__ out) {
    __shared__ int sc[4], sl[4], sm[4];
    int c = 0, l = 0, m = 0;
    for (int b = threadIdx.x; b < n; b += 256) {
        const int4 v = reinterpret_cast<const int4*>(part)[b];
        c += v.x; l += v.y; m = v.z > m ? v.z : m;
    }
#pragma unroll
    for (int sft = 1; sft < 64; sft <<= 1) { c += __shfl_xor(c, sft); l += __shfl_xor(l, sft); const int mo = __shfl_xor(m, sft); m = mo > m ? mo : m; }
    if ((threadIdx.x & 63) == 0) { sc[threadIdx.x >> 6] = c; sl[threadIdx.x >> 6] = l; sm[threadIdx.x >> 6] = m; }
    __syncthreads();
    if (threadIdx.x == 0) {
        out[0] = (sc[0] + sc[1]) + (sc[2] + sc[3]); out[1] = (sl[0] + sl[1]) + (sl[2] + sl[3]);
        const int a = sm[0] > sm[1] ? sm[0] : sm[1], b = sm[2] > sm[3] ? sm[2] : sm[3];
        out[2] = a > b ? a : b; out[3] = 0;
    }
}
int mi_env_stats_reduce(mi_env* e, int32_t* out, hipStream_t s) {
    env_stats_reduce_kernel<<<1, 256, 0, s>>>(e->stats_part, e->stats_n > 0 ? e->stats_n : 0, out);
    MI_LAUNCH_CHECK();
    return MI_OK;
}
extern "C" int mi_env_episode_stats(void* handle, int32_t* out, void* stream) {
    MI_CHECK_ARG(handle && out, "NULL pointer");
    mi_env* e = (mi_env*)handle;
    if (e->stats_n < 0) { mi_set_error("mi_env_episode_stats: no acting / rollout call with episode_stats == NULL has run on this handle yet"); return MI_ESTATE; }
    return mi_env_stats_reduce(e, out, (hipStream_t)stream);
}

extern "C" int mi_env_destroy(void* handle) {
    if (!handle) return MI_OK;
    mi_env* e = (mi_env*)handle;
    if (e->x) (void)hipFree(e->x);
    delete e;
    return MI_OK;
}

extern "C" int mi_env_reset(void* handle, float* obs, const double* forced_state, void* stream) {
    MI_CHECK_ARG(handle && obs, "handle/obs is NULL");
    mi_env* e = (mi_env*)handle;
    if (e->kind == MI_ENV_PENDULUM_V1) return mi_pend_reset_impl(e, obs, forced_state, (hipStream_t)stream);
    env_reset_kernel<<<(e->n + 255) / 256, 256, 0, (hipStream_t)stream>>>(*e, obs, forced_state);
    MI_LAUNCH_CHECK();
    return MI_OK;
}

extern "C" int mi_env_step(void* handle, const int64_t* actions, const double* forced_reset, float* obs, float* reward,
                           uint8_t* done, uint8_t* truncated, float* fin_ret, int32_t* fin_len, void* stream) {
    MI_CHECK_ARG(handle && actions && obs && reward && done && truncated && fin_ret && fin_len, "NULL pointer");
    mi_env* e = (mi_env*)handle;
    MI_CHECK_ARG(e->kind == MI_ENV_CARTPOLE_V1, "discrete-action step on a continuous-action env (use mi_env_step_cont)");
    env_step_kernel<<<(e->n + 255) / 256, 256, 0, (hipStream_t)stream>>>(*e, actions, forced_reset, obs, reward, done,
                                                                        truncated, fin_ret, fin_len, nullptr);
    MI_LAUNCH_CHECK();
    return MI_OK;
}

int mi_pend_step_ex_impl(mi_env* e, const float* actions, const double* forced_reset, float* obs, float* reward, uint8_t* done, uint8_t* truncated, float* fin_ret,
                         int32_t* fin_len, float* raw_obs, hipStream_t s);   // mi_sac.hip
extern "C" int mi_env_step_ex(void* handle, const void* actions, const double* forced_reset, float* obs, float* reward, uint8_t* done, uint8_t* truncated,
                              float* fin_ret, int32_t* fin_len, float* raw_obs, void* stream) {
    MI_CHECK_ARG(handle && actions && obs && reward && done && truncated && fin_ret && fin_len, "NULL pointer");
    mi_env* e = (mi_env*)handle;
    if (e->kind == MI_ENV_PENDULUM_V1)
        return mi_pend_step_ex_impl(e, (const float*)actions, forced_reset, obs, reward, done, truncated, fin_ret, fin_len, raw_obs, (hipStream_t)stream);
    env_step_kernel<<<(e->n + 255) / 256, 256, 0, (hipStream_t)stream>>>(*e, (const int64_t*)actions, forced_reset, obs, reward, done, truncated, fin_ret, fin_len, raw_obs);
    MI_LAUNCH_CHECK();
    return MI_OK;
}

extern "C" int mi_env_get_state(void* handle, double* state, int32_t* elapsed, void* stream) {
    MI_CHECK_ARG(handle != nullptr, "handle is NULL");
    mi_env* e = (mi_env*)handle;
    env_get_state_kernel<<<(e->n + 255) / 256, 256, 0, (hipStream_t)stream>>>(*e, state, elapsed);
    MI_LAUNCH_CHECK();
    return MI_OK;
}

// ---- HIP-event timer for bench.py ----------------------------------------------------------------------
struct mi_timer { hipEvent_t a, b; };
extern "C" int mi_timer_create(void** t) {
    MI_CHECK_ARG(t != nullptr, "timer is NULL");
    mi_timer* x = new (std::nothrow) mi_timer();
    if (!x) return MI_ENOMEM;
    MI_HIP(hipEventCreate(&x->a));
    MI_HIP(hipEventCreate(&x->b));
    *t = x;
    return MI_OK;
}
extern "C" int mi_timer_destroy(void* t) {
    if (!t) return MI_OK;
    mi_timer* x = (mi_timer*)t;
    (void)hipEventDestroy(x->a); (void)hipEventDestroy(x->b);
    delete x;
    return MI_OK;
}
extern "C" int mi_timer_start(void* t, void* stream) { MI_CHECK_ARG(t, "NULL"); MI_HIP(hipEventRecord(((mi_timer*)t)->a, (hipStream_t)stream)); return MI_OK; }
extern "C" int mi_timer_stop(void* t, void* stream) { MI_CHECK_ARG(t, "NULL"); MI_HIP(hipEventRecord(((mi_timer*)t)->b, (hipStream_t)stream)); return MI_OK; }
extern "C" int mi_timer_elapsed_ms(void* t, float* ms) {
    MI_CHECK_ARG(t && ms, "NULL");
    mi_timer* x = (mi_timer*)t;
    MI_HIP(hipEventSynchronize(x->b));
    MI_HIP(hipEventElapsedTime(ms, x->a, x->b));
    return MI_OK;
}

// ---- in-library kernel profiler ------------------------------------------------------------------------------
#include <vector>
static struct {
    bool armed = false;
    uint32_t mask = 0;
    std::vector<hipEvent_t> ev;   // pairs
    std::vector<int> tag;
    size_t used = 0;              // events handed out
} g_prof;

#ifdef MI_INSIDE
static __device__ unsigned long long mi_inside_marker[18][2][2];
__global__ void mi_inside_marker_kernel(int tag, int end) {
    unsigned long long t0, t1;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
    if (threadIdx.x == 0) { mi_inside_marker[tag][end][0] = t0; mi_inside_marker[tag][end][1] = t1; }
}
extern "C" int mi_debug_inside_markers(unsigned long long* out) { return hipMemcpyFromSymbol(out, HIP_SYMBOL(mi_inside_marker), sizeof(mi_inside_marker)) == hipSuccess ? 0 : -2; }
#endif
void mi_prof_mark(int tag, bool end, hipStream_t s) {
#ifdef MI_INSIDE
    mi_inside_marker_kernel<<<1, 64, 0, s>>>(tag, end ? 1 : 0);
#endif
    if (!g_prof.armed || !((g_prof.mask >> tag) & 1u)) return;
    if (!end) {
        if (g_prof.used + 2 > g_prof.ev.size()) return;  // pool exhausted: silently stop sampling
        g_prof.tag[g_prof.used / 2] = tag;
        (void)hipEventRecord(g_prof.ev[g_prof.used], s);
        g_prof.used += 1;
    } else {
        if ((g_prof.used & 1) == 0) return;  // begin was dropped
        (void)hipEventRecord(g_prof.ev[g_prof.used], s);
        g_prof.used += 1;
    }
}

extern "C" int mi_prof_begin(int max_launches, uint32_t tag_mask) {
    MI_CHECK_ARG(max_launches > 0 && max_launches <= (1 << 20), "max_launches out of range");
    for (hipEvent_t e : g_prof.ev) (void)hipEventDestroy(e);
    g_prof.ev.assign(2 * (size_t)max_launches, nullptr);
    g_prof.tag.assign(max_launches, 0);
    for (auto& e : g_prof.ev) MI_HIP(hipEventCreate(&e));
    g_prof.used = 0;
    g_prof.mask = tag_mask;
    g_prof.armed = true;
    return MI_OK;
}

// pause / resume the sampling between mi_prof_begin and mi_prof_end (a measurement that brackets EVERY launch of a 70 us kernel with two events costs the loop it
// measures 7.5 %, tools/prof_overhead.py: bench.py samples the launches of every 10th update)
extern "C" int mi_prof_pause(int paused) {
    if (g_prof.ev.empty()) return MI_OK;
    if ((g_prof.used & 1) != 0) return MI_OK;   // between a begin and an end mark (only reachable from a second host thread while a C call is inside a tagged scope): leave it
    g_prof.armed = !paused;
    return MI_OK;
}

extern "C" int mi_prof_end(float* total_ms, int32_t* count) {
    MI_CHECK_ARG(total_ms && count, "NULL pointer");
    g_prof.armed = false;
    for (int t = 0; t < MI_PROF_NTAGS; ++t) { total_ms[t] = 0.0f; count[t] = 0; }
    const size_t pairs = g_prof.used / 2;
    if (pairs) MI_HIP(hipEventSynchronize(g_prof.ev[2 * pairs - 1]));
    for (size_t i = 0; i < pairs; ++i) {
        float ms = 0.0f;
        MI_HIP(hipEventElapsedTime(&ms, g_prof.ev[2 * i], g_prof.ev[2 * i + 1]));
        const int t = g_prof.tag[i];
        if (t >= 0 && t < MI_PROF_NTAGS) { total_ms[t] += ms; count[t] += 1; }
    }
    for (hipEvent_t e : g_prof.ev) (void)hipEventDestroy(e);
    g_prof.ev.clear(); g_prof.tag.clear(); g_prof.used = 0;
    return MI_OK;
}
