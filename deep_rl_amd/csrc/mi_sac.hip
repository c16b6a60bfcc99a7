// mi_sac.hip — the SAC hot path of reference deep_rl/sac.py (re-targeted to Pendulum-v1) on the device:
//   pend_*_kernel           Pendulum-v1 reset / step (gym 0.21 pendulum.py + TimeLimit 200 + episode statistics)
//   sac_act_kernel          sac.py:138-158: uniform random action before learning_starts, else actor.get_action; step; ring store
//   sac_critic_kernel       sac.py:165-185: actor on next obs, two target critics, TD target, two critics forward + backward
//   sac_actor_kernel        sac.py:193-197: actor forward (rsample), min(Q1,Q2) forward, d(-minQ)/d action, actor backward
//   sac_logp_kernel         sac.py:203-204: log-probs of fresh actions for the alpha loss
//   sac_dw2_gemm_kernel     the 256x256 weight gradients as batch GEMMs dW2 = dZ2^T H1 on v_mfma_f32_16x16x4_f32
//   sac_small_reduce_kernel fixed-order sum of the per-workgroup slabs of the thin gradients
//   adam_kernel / polyak_kernel / sac_alpha_kernel
// Row-group layout: a 256-thread workgroup owns SR = 8 batch rows; thread j owns hidden unit j of whichever net is being
// evaluated; a row group's activations live in LDS, 256x256 weight matrices are streamed from L2 (forward: row j per thread;
// backward-data: column k per thread, coalesced).  All reductions are fixed-order (bitwise reproducible).
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
#define SR 8  // rows per workgroup

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
                 int32_t* __restrict__ fin_len) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= e.n) return;
    double th = e.x[i], thd = e.x_dot[i];
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
    pend_step_kernel<<<(e->n + 255) / 256, 256, 0, (hipStream_t)stream>>>(*e, actions, forced_reset, obs, reward, done, truncated, fin_ret, fin_len);
    MI_LAUNCH_CHECK();
    return MI_OK;
}

// ================================================ row-group building blocks =====================================================
struct __attribute__((aligned(16))) sac_smem {
    float x[SR][4];           // obs (3) + action
    float xn[SR][4];          // next obs (3) + next action
    float a1[SR][SA_H];       // activation buffers (post-ReLU), rows x units
    float a2[SR][SA_H];
    float s1[SR][SA_H];
    float s2[SR][SA_H];
    float red[4][SR][2];      // cross-wave partial sums
    float rv[SR][16];         // per-row scalars
    long long cur[SR], nxt[SR];
};

// keyed standard normal (production mode): Box-Muller on two Philox words
__device__ __forceinline__ float keyed_normal(uint64_t seed, uint64_t tag_update, uint64_t row) {
    uint32_t r[4];
    mi_philox(seed, tag_update, row, STREAM_NORMAL, r);
    const float u1 = ((float)(r[0] >> 8) + 0.5f) * (1.0f / 16777216.0f), u2 = (float)(r[1] >> 8) * (1.0f / 16777216.0f);
    return sqrtf(-2.0f * logf(u1)) * cosf(6.28318530717958647692f * u2);
}

// layer 1 of unit j for all rows: h[r][j] = relu(b[j] + W[j][:IN] . x[r][:IN])
template <int IN>
__device__ __forceinline__ void layer1(const float* __restrict__ W, const float* __restrict__ b, const float (*x)[4], float (*h)[SA_H], int j) {
    float w[IN];
#pragma unroll
    for (int k = 0; k < IN; ++k) w[k] = W[j * IN + k];
    const float bj = b[j];
#pragma unroll
    for (int r = 0; r < SR; ++r) {
        float z = 0.0f;
#pragma unroll
        for (int k = 0; k < IN; ++k) z = __builtin_fmaf(w[k], x[r][k], z);
        h[r][j] = fmaxf(z + bj, 0.0f);
    }
}

// layer 2 of unit j for all rows: out[r][j] = relu(b[j] + W[j][:] . in[r][:]); W row j streamed as float4 (1 KB contiguous per thread)
__device__ __forceinline__ void layer2(const float* __restrict__ W, const float* __restrict__ b, const float (*in)[SA_H], float (*out)[SA_H], int j) {
    float acc[SR];
#pragma unroll
    for (int r = 0; r < SR; ++r) acc[r] = 0.0f;
    const float4* wrow = reinterpret_cast<const float4*>(W + (size_t)j * SA_H);
#pragma unroll 4
    for (int k4 = 0; k4 < SA_H / 4; ++k4) {
        const float4 w = wrow[k4];
#pragma unroll
        for (int r = 0; r < SR; ++r) {
            const float4 hv = *reinterpret_cast<const float4*>(&in[r][4 * k4]);
            acc[r] = __builtin_fmaf(w.x, hv.x, acc[r]); acc[r] = __builtin_fmaf(w.y, hv.y, acc[r]);
            acc[r] = __builtin_fmaf(w.z, hv.z, acc[r]); acc[r] = __builtin_fmaf(w.w, hv.w, acc[r]);
        }
    }
    const float bj = b[j];
#pragma unroll
    for (int r = 0; r < SR; ++r) out[r][j] = fmaxf(acc[r] + bj, 0.0f);
}

// backward-data through layer 2: dh[r] (for input unit k = this thread) = sum_j W[j][k] * dz[r][j]   (column k: coalesced over threads)
__device__ __forceinline__ void layer2_bwd(const float* __restrict__ W, const float (*dz)[SA_H], float dh[SR], int k) {
#pragma unroll
    for (int r = 0; r < SR; ++r) dh[r] = 0.0f;
#pragma unroll 4
    for (int j4 = 0; j4 < SA_H / 4; ++j4) {
        const float w0 = W[(size_t)(4 * j4 + 0) * SA_H + k], w1 = W[(size_t)(4 * j4 + 1) * SA_H + k];
        const float w2 = W[(size_t)(4 * j4 + 2) * SA_H + k], w3 = W[(size_t)(4 * j4 + 3) * SA_H + k];
#pragma unroll
        for (int r = 0; r < SR; ++r) {
            const float4 d = *reinterpret_cast<const float4*>(&dz[r][4 * j4]);
            dh[r] = __builtin_fmaf(w0, d.x, dh[r]); dh[r] = __builtin_fmaf(w1, d.y, dh[r]);
            dh[r] = __builtin_fmaf(w2, d.z, dh[r]); dh[r] = __builtin_fmaf(w3, d.w, dh[r]);
        }
    }
}

// sum over the 256 threads of v0[r], v1[r] for every row; result readable by everyone in sm.rv[r][slot0], [slot1] after the call
__device__ __forceinline__ void block_rowsum2(sac_smem& sm, const float v0[SR], const float v1[SR], int slot0, int slot1) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
    for (int r = 0; r < SR; ++r) {
        const float a = wave_sum_uniform(v0[r]), b = wave_sum_uniform(v1[r]);
        if (lane == 0) { sm.red[wave][r][0] = a; sm.red[wave][r][1] = b; }
    }
    __syncthreads();
    if (threadIdx.x < SR) {
        const int r = threadIdx.x;
        sm.rv[r][slot0] = (sm.red[0][r][0] + sm.red[1][r][0]) + (sm.red[2][r][0] + sm.red[3][r][0]);
        sm.rv[r][slot1] = (sm.red[0][r][1] + sm.red[1][r][1]) + (sm.red[2][r][1] + sm.red[3][r][1]);
    }
    __syncthreads();
}

// SoftQNetwork forward on the rows of `x` (obs + action): activations -> h1, h2; q[r] -> sm.rv[r][slot]
__device__ __forceinline__ void q_forward(sac_smem& sm, const float* __restrict__ p, const float (*x)[4], float (*h1)[SA_H], float (*h2)[SA_H], int slot) {
    const int j = threadIdx.x;
    layer1<4>(p + SQ_W1, p + SQ_B1, x, h1, j);
    __syncthreads();
    layer2(p + SQ_W2, p + SQ_B2, h1, h2, j);
    float v[SR], zero[SR];
    const float w3 = p[SQ_W3 + j];
#pragma unroll
    for (int r = 0; r < SR; ++r) { v[r] = w3 * h2[r][j]; zero[r] = 0.0f; }
    block_rowsum2(sm, v, zero, slot, 15);
    if (threadIdx.x < SR) sm.rv[threadIdx.x][slot] += p[SQ_B3];
    __syncthreads();
}

// rv slots used by the actor: 0 mean, 1 sraw, 2 ls, 3 sd, 4 u, 5 logp, 6 action, 7 eps
__device__ __forceinline__ void actor_forward(sac_smem& sm, const float* __restrict__ p, const float (*x)[4], float (*h1)[SA_H], float (*h2)[SA_H],
                                              const float eps_row /*valid in threads < SR*/) {
    const int j = threadIdx.x;
    layer1<3>(p + AC_W1, p + AC_B1, x, h1, j);
    __syncthreads();
    layer2(p + AC_W2, p + AC_B2, h1, h2, j);
    float vm[SR], vl[SR];
    const float wm = p[AC_WM + j], wl = p[AC_WL + j];
#pragma unroll
    for (int r = 0; r < SR; ++r) { vm[r] = wm * h2[r][j]; vl[r] = wl * h2[r][j]; }
    block_rowsum2(sm, vm, vl, 0, 1);
    if (threadIdx.x < SR) {
        const int r = threadIdx.x;
        const float mean = sm.rv[r][0] + p[AC_BM], sraw = sm.rv[r][1] + p[AC_BL];
        const float ls = tanhf(sraw);
        const float L = SA_LOG_STD_MIN + 0.5f * (SA_LOG_STD_MAX - SA_LOG_STD_MIN) * (ls + 1.0f);   // sac.py:69
        const float sd = expf(L);
        const float z = mean + eps_row * sd;                                                     // :71
        const float u = tanhf(z);                                                                // :72
        const float d = z - mean;
        float lp = -(d * d) / (2.0f * (sd * sd)) - logf(sd) - 0.91893853320467274178f;           // :73
        lp -= logf(SA_ACT_SCALE * (1.0f - u * u) + 1e-6f);                                       // :75
        sm.rv[r][0] = mean; sm.rv[r][1] = sraw; sm.rv[r][2] = ls; sm.rv[r][3] = sd; sm.rv[r][4] = u; sm.rv[r][5] = lp;
        sm.rv[r][6] = u * SA_ACT_SCALE + SA_ACT_BIAS; sm.rv[r][7] = eps_row;                      // :77
    }
    __syncthreads();
}

// ================================================ forward-only API kernels ======================================================
__global__ void __launch_bounds__(256) sac_actor_sample_kernel(const float* __restrict__ actor, const float* __restrict__ obs, const float* __restrict__ eps,
                                                                int n, float* __restrict__ action, float* __restrict__ logp) {
    __shared__ sac_smem sm;
    const int row0 = blockIdx.x * SR;
    if (threadIdx.x < SR * 3) { const int r = threadIdx.x / 3, k = threadIdx.x % 3; const int b = row0 + r < n ? row0 + r : n - 1; sm.x[r][k] = obs[3 * (size_t)b + k]; }
    __syncthreads();
    float e = 0.0f;
    if (threadIdx.x < SR) e = eps[row0 + threadIdx.x < n ? row0 + threadIdx.x : n - 1];
    actor_forward(sm, actor, sm.x, sm.a1, sm.a2, e);
    if (threadIdx.x < SR && row0 + threadIdx.x < n) { action[row0 + threadIdx.x] = sm.rv[threadIdx.x][6]; if (logp) logp[row0 + threadIdx.x] = sm.rv[threadIdx.x][5]; }
}

extern "C" int mi_sac_actor_sample(const float* actor, const float* obs, const float* eps, int n, float* action, float* logp, void* stream) {
    MI_CHECK_ARG(actor && obs && eps && action && n > 0, "bad arguments");
    sac_actor_sample_kernel<<<(n + SR - 1) / SR, 256, 0, (hipStream_t)stream>>>(actor, obs, eps, n, action, logp);
    MI_LAUNCH_CHECK();
    return MI_OK;
}

__global__ void __launch_bounds__(256) sac_q_forward_kernel(const float* __restrict__ q, const float* __restrict__ obs, const float* __restrict__ act, int n,
                                                             float* __restrict__ out) {
    __shared__ sac_smem sm;
    const int row0 = blockIdx.x * SR;
    if (threadIdx.x < SR * 4) {
        const int r = threadIdx.x / 4, k = threadIdx.x & 3; const int b = row0 + r < n ? row0 + r : n - 1;
        sm.x[r][k] = k < 3 ? obs[3 * (size_t)b + k] : act[b];
    }
    __syncthreads();
    q_forward(sm, q, sm.x, sm.s1, sm.s2, 8);
    if (threadIdx.x < SR && row0 + threadIdx.x < n) out[row0 + threadIdx.x] = sm.rv[threadIdx.x][8];
}

extern "C" int mi_sac_q_forward(const float* q, const float* obs, const float* act, int n, float* out, void* stream) {
    MI_CHECK_ARG(q && obs && act && out && n > 0, "bad arguments");
    sac_q_forward_kernel<<<(n + SR - 1) / SR, 256, 0, (hipStream_t)stream>>>(q, obs, act, n, out);
    MI_LAUNCH_CHECK();
    return MI_OK;
}

// ================================================ acting ========================================================================
__global__ void __launch_bounds__(256)
sac_act_kernel(mi_env e, const float* __restrict__ actor, long long global_step, long long slots, long long learning_starts, float* __restrict__ obs_cur,
               float* __restrict__ observations, float* __restrict__ actions, float* __restrict__ rewards, uint8_t* __restrict__ terminated,
               const float* __restrict__ forced_actions, const float* __restrict__ forced_eps, const double* __restrict__ forced_resets,
               mi_episode_t* __restrict__ episodes, int32_t* __restrict__ episode_stats, int max_ep) {
    __shared__ sac_smem sm;
    const int N = e.n, row0 = blockIdx.x * SR;
    const bool policy = !forced_actions && global_step >= learning_starts;   // wave/block-uniform
    if (threadIdx.x < SR * 3) { const int r = threadIdx.x / 3, k = threadIdx.x % 3; const int g = row0 + r < N ? row0 + r : N - 1; sm.x[r][k] = obs_cur[3 * (size_t)g + k]; }
    __syncthreads();
    if (policy) {
        float eps = 0.0f;
        if (threadIdx.x < SR) {
            const int g = row0 + threadIdx.x < N ? row0 + threadIdx.x : N - 1;
            eps = forced_eps ? forced_eps[g] : keyed_normal(e.seed, (1ull << 40) + e.env_id_base + (uint64_t)g, (uint64_t)global_step);
        }
        actor_forward(sm, actor, sm.x, sm.a1, sm.a2, eps);
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

__global__ void sac_zero4_kernel(int32_t* p) { if (threadIdx.x < 4) p[threadIdx.x] = 0; }

extern "C" int mi_sac_act_step(void* handle, const float* actor, int64_t global_step, int64_t slots, int64_t learning_starts, float* obs_cur,
                               float* observations, float* actions, float* rewards, uint8_t* terminated, const float* forced_actions,
                               const float* forced_eps, const double* forced_resets, mi_episode_t* episodes, int32_t* episode_stats, int max_ep,
                               void* stream) {
    MI_CHECK_ARG(handle && actor && obs_cur && observations && actions && rewards && terminated, "NULL pointer");
    mi_env* e = (mi_env*)handle;
    MI_CHECK_ARG(e->kind == MI_ENV_PENDULUM_V1, "SAC path needs a Pendulum-v1 handle");
    MI_CHECK_ARG(slots >= 2 && global_step >= 0 && max_ep >= 0 && (max_ep == 0 || episodes), "bad arguments");
    hipStream_t s = (hipStream_t)stream;
    if (episode_stats) { sac_zero4_kernel<<<1, 64, 0, s>>>(episode_stats); MI_LAUNCH_CHECK(); }
    sac_act_kernel<<<(e->n + SR - 1) / SR, 256, 0, s>>>(*e, actor, (long long)global_step, (long long)slots, (long long)learning_starts, obs_cur, observations,
                                                       actions, rewards, terminated, forced_actions, forced_eps, forced_resets, episodes, episode_stats, max_ep);
    MI_LAUNCH_CHECK();
    return MI_OK;
}

// ================================================ workspace layout ==============================================================
// [H1 mats: 3 x batch x 256][DZ2 mats: 3 x batch x 256][slabs: nblocks x SLAB]   (mats 0,1: critics; 2: actor)
#define SLAB 3600
__host__ __device__ inline size_t ws_mat_floats(int batch) { return (size_t)batch * SA_H; }
extern "C" size_t mi_sac_workspace_bytes(int batch) {
    const size_t nb = (size_t)(batch + SR - 1) / SR;
    return (6 * ws_mat_floats(batch) + nb * SLAB) * sizeof(float);
}

// ================================================ critic update =================================================================
// slab layout (critic): net n at n*1793: W1 1024 | b1 256 | b2 256 | W3 256 | b3 1;  then [3586] = sum (q1-y)^2, [3587] = sum (q2-y)^2
__global__ void __launch_bounds__(256)
sac_critic_kernel(const float* __restrict__ q, const float* __restrict__ qt, const float* __restrict__ actor, const float* __restrict__ observations,
                  const float* __restrict__ actions, const float* __restrict__ rewards, const uint8_t* __restrict__ terminated,
                  const int64_t* __restrict__ idx, int batch, int n_envs, long long slots, const float* __restrict__ eps, uint64_t seed, uint64_t update,
                  const float* __restrict__ alpha_p, float gamma, float invn, float* __restrict__ ws) {
    __shared__ sac_smem sm;
    const int t = threadIdx.x, row0 = blockIdx.x * SR;
    float* H1 = ws; float* DZ2 = ws + 3 * ws_mat_floats(batch);
    float* slab = ws + 6 * ws_mat_floats(batch) + (size_t)blockIdx.x * SLAB;
    if (t < SR) {
        const int b = row0 + t < batch ? row0 + t : batch - 1;
        const long long i = idx[b];
        sm.cur[t] = i; sm.nxt[t] = ((i / n_envs + 1) % slots) * n_envs + i % n_envs;
    }
    __syncthreads();
    if (t < SR * 4) {
        const int r = t / 4, k = t & 3;
        sm.x[r][k] = k < 3 ? observations[3 * sm.cur[r] + k] : actions[sm.cur[r]];
        sm.xn[r][k] = k < 3 ? observations[3 * sm.nxt[r] + k] : 0.0f;
    }
    __syncthreads();
    // ---- next action + log-prob under the current actor (no grad; sac.py:172) ----
    float e_row = 0.0f;
    if (t < SR) { const int b = row0 + t < batch ? row0 + t : batch - 1; e_row = eps ? eps[b] : keyed_normal(seed, (2ull << 40) + update, (uint64_t)b); }
    actor_forward(sm, actor, sm.xn, sm.a1, sm.a2, e_row);
    if (t < SR) { sm.xn[t][3] = sm.rv[t][6]; sm.rv[t][9] = sm.rv[t][5]; }   // a', log pi(a'|s')
    __syncthreads();
    // ---- target critics (:173-174) ----
    q_forward(sm, qt, sm.xn, sm.s1, sm.s2, 8);
    if (t < SR) sm.rv[t][10] = sm.rv[t][8];
    __syncthreads();
    q_forward(sm, qt + SQ_NP, sm.xn, sm.s1, sm.s2, 8);
    if (t < SR) {
        const float alpha = alpha_p[0];
        const float mq = fminf(sm.rv[t][10], sm.rv[t][8]) - alpha * sm.rv[t][9];                                   // :176
        sm.rv[t][10] = rewards[sm.nxt[t]] + (terminated[sm.nxt[t]] ? 0.0f : 1.0f) * gamma * mq;                     // :177  (y)
    }
    __syncthreads();
    // ---- the two critics on (obs, action): forward, loss, backward (:179-185) ----
    for (int net = 0; net < 2; ++net) {
        const float* p = q + (size_t)net * SQ_NP;
        float* sl = slab + net * 1793;
        q_forward(sm, p, sm.x, sm.s1, sm.s2, 8);
        if (t < SR) {
            const bool valid = row0 + t < batch;
            const float d = valid ? sm.rv[t][8] - sm.rv[t][10] : 0.0f;
            sm.rv[t][9] = d * d;          // loss contribution
            sm.rv[t][8] = 2.0f * d * invn; // d loss / d q
        }
        __syncthreads();
        {   // unit j = t: dz2, thin gradients of layer 3 / bias 2, H1 / DZ2 rows for the GEMM
            const float w3 = p[SQ_W3 + t];
            float gw3 = 0.0f, gb2 = 0.0f;
#pragma unroll
            for (int r = 0; r < SR; ++r) {
                const float h2 = sm.s2[r][t], dq = sm.rv[r][8];
                const float dz = h2 > 0.0f ? w3 * dq : 0.0f;
                gw3 = __builtin_fmaf(dq, h2, gw3); gb2 += dz;
                if (row0 + r < batch) { DZ2[((size_t)net * batch + row0 + r) * SA_H + t] = dz; H1[((size_t)net * batch + row0 + r) * SA_H + t] = sm.s1[r][t]; }
                sm.s2[r][t] = dz;
            }
            sl[1024 + 256 + t] = gb2; sl[1024 + 512 + t] = gw3;
            if (t == 0) { float gb3 = 0.0f, l = 0.0f; for (int r = 0; r < SR; ++r) { gb3 += sm.rv[r][8]; l += sm.rv[r][9]; } sl[1792] = gb3; slab[3586 + net] = l; }
        }
        __syncthreads();
        {   // input unit k = t: dh1 -> dz1 -> thin gradients of layer 1
            float dh[SR];
            layer2_bwd(p + SQ_W2, sm.s2, dh, t);
            float gb1 = 0.0f, gw[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int r = 0; r < SR; ++r) {
                const float d = sm.s1[r][t] > 0.0f ? dh[r] : 0.0f;
                gb1 += d;
#pragma unroll
                for (int c = 0; c < 4; ++c) gw[c] = __builtin_fmaf(d, sm.x[r][c], gw[c]);
            }
            *reinterpret_cast<float4*>(sl + 4 * t) = make_float4(gw[0], gw[1], gw[2], gw[3]);
            sl[1024 + t] = gb1;
        }
        __syncthreads();
    }
}

// ================================================ actor update ==================================================================
// slab layout (actor): W1 768 | b1 256 | b2 256 | Wm 256 | bm 1 | Wl 256 | bl 1 | [1794] sum(alpha*logp - minq) | [1795] sum logp
__global__ void __launch_bounds__(256)
sac_actor_kernel(const float* __restrict__ actor, const float* __restrict__ q, const float* __restrict__ observations, const int64_t* __restrict__ idx,
                 int batch, const float* __restrict__ eps, uint64_t seed, uint64_t update, const float* __restrict__ alpha_p, float invn,
                 float* __restrict__ ws, int logp_only) {
    __shared__ sac_smem sm;
    const int t = threadIdx.x, row0 = blockIdx.x * SR;
    float* H1 = ws + 2 * ws_mat_floats(batch); float* DZ2 = ws + 5 * ws_mat_floats(batch);
    float* slab = ws + 6 * ws_mat_floats(batch) + (size_t)blockIdx.x * SLAB;
    if (t < SR * 3) { const int r = t / 3, k = t % 3; const int b = row0 + r < batch ? row0 + r : batch - 1; sm.x[r][k] = observations[3 * idx[b] + k]; }
    __syncthreads();
    float e_row = 0.0f;
    if (t < SR) { const int b = row0 + t < batch ? row0 + t : batch - 1; e_row = eps ? eps[b] : keyed_normal(seed, ((logp_only ? 4ull : 3ull) << 40) + update, (uint64_t)b); }
    actor_forward(sm, actor, sm.x, sm.a1, sm.a2, e_row);
    if (logp_only) {   // sac.py:203-204
        if (t == 0) { float s = 0.0f; for (int r = 0; r < SR; ++r) s += row0 + r < batch ? sm.rv[r][5] : 0.0f; slab[1795] = s; slab[1794] = 0.0f; }
        return;
    }
    const float alpha = alpha_p[0];
    if (t < SR) sm.x[t][3] = sm.rv[t][6];   // the action enters the critics
    __syncthreads();
    // ---- min(Q1, Q2)(obs, pi(obs)) and d(-min Q)/d action (:194-196) ----
    q_forward(sm, q, sm.x, sm.s1, sm.s2, 8);
    if (t < SR) sm.rv[t][10] = sm.rv[t][8];
    __syncthreads();
    q_forward(sm, q + SQ_NP, sm.x, sm.s1, sm.s2, 8);
    if (t < SR) {
        const float q1 = sm.rv[t][10], q2 = sm.rv[t][8];
        const bool valid = row0 + t < batch;
        sm.rv[t][9] = valid ? alpha * sm.rv[t][5] - fminf(q1, q2) : 0.0f;                    // loss term (:197)
        // torch.min routes the gradient to the smaller input, half / half on ties
        sm.rv[t][10] = !valid ? 0.0f : q1 < q2 ? 1.0f : (q2 < q1 ? 0.0f : 0.5f);               // weight of critic 1
        sm.rv[t][11] = 0.0f;                                                                 // d loss / d action accumulator
    }
    __syncthreads();
    // net 1 first (its activations are still live in s1 / s2), then net 0 is recomputed
    for (int pass = 0; pass < 2; ++pass) {
        const int net = pass == 0 ? 1 : 0;
        const float* p = q + (size_t)net * SQ_NP;
        if (pass == 1) q_forward(sm, p, sm.x, sm.s1, sm.s2, 8);
        {
            const float w3 = p[SQ_W3 + t];
#pragma unroll
            for (int r = 0; r < SR; ++r) {
                const float wgt = net == 0 ? sm.rv[r][10] : (row0 + r < batch ? 1.0f - sm.rv[r][10] : 0.0f);
                const float dq = -invn * wgt;
                sm.s2[r][t] = sm.s2[r][t] > 0.0f ? w3 * dq : 0.0f;
            }
        }
        __syncthreads();
        {
            float dh[SR], v[SR], zero[SR];
            layer2_bwd(p + SQ_W2, sm.s2, dh, t);
            const float w13 = p[SQ_W1 + 4 * t + 3];
#pragma unroll
            for (int r = 0; r < SR; ++r) { v[r] = sm.s1[r][t] > 0.0f ? w13 * dh[r] : 0.0f; zero[r] = 0.0f; }
            block_rowsum2(sm, v, zero, 8, 15);
            if (t < SR) { sm.rv[t][11] += sm.rv[t][8]; }
        }
        __syncthreads();
    }
    // ---- d loss / d mean, d loss / d sraw per row ----
    if (t < SR) {
        const float u = sm.rv[t][4], sd = sm.rv[t][3], ls = sm.rv[t][2];
        const float omu2 = 1.0f - u * u;
        const bool valid = row0 + t < batch;
        const float c = valid ? invn : 0.0f;
        const float du = SA_ACT_SCALE * sm.rv[t][11] + alpha * c * (2.0f * SA_ACT_SCALE * u) / (SA_ACT_SCALE * omu2 + 1e-6f);
        const float dz_u = du * omu2;
        const float dL = dz_u * e_row * sd - alpha * c;
        sm.rv[t][0] = dz_u;                                                                          // d / d mean
        sm.rv[t][1] = 0.5f * (SA_LOG_STD_MAX - SA_LOG_STD_MIN) * dL * (1.0f - ls * ls);                // d / d sraw
    }
    __syncthreads();
    {   // unit j = t of the actor: dz2, thin gradients of the heads / bias 2, H1 / DZ2 rows
        const float wm = actor[AC_WM + t], wl = actor[AC_WL + t];
        float gwm = 0.0f, gwl = 0.0f, gb2 = 0.0f;
#pragma unroll
        for (int r = 0; r < SR; ++r) {
            const float h2 = sm.a2[r][t], dm = sm.rv[r][0], ds = sm.rv[r][1];
            const float dz = h2 > 0.0f ? __builtin_fmaf(wl, ds, wm * dm) : 0.0f;
            gwm = __builtin_fmaf(dm, h2, gwm); gwl = __builtin_fmaf(ds, h2, gwl); gb2 += dz;
            if (row0 + r < batch) { DZ2[(size_t)(row0 + r) * SA_H + t] = dz; H1[(size_t)(row0 + r) * SA_H + t] = sm.a1[r][t]; }
            sm.a2[r][t] = dz;
        }
        slab[768 + 256 + t] = gb2; slab[768 + 512 + t] = gwm; slab[768 + 512 + 257 + t] = gwl;
        if (t == 0) {
            float gbm = 0.0f, gbl = 0.0f, l = 0.0f, lp = 0.0f;
            for (int r = 0; r < SR; ++r) { gbm += sm.rv[r][0]; gbl += sm.rv[r][1]; l += sm.rv[r][9]; lp += row0 + r < batch ? sm.rv[r][5] : 0.0f; }
            slab[768 + 512 + 256] = gbm; slab[768 + 512 + 257 + 256] = gbl; slab[1794] = l; slab[1795] = lp;
        }
    }
    __syncthreads();
    {
        float dh[SR];
        layer2_bwd(actor + AC_W2, sm.a2, dh, t);
        float gb1 = 0.0f, gw[3] = {0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int r = 0; r < SR; ++r) {
            const float d = sm.a1[r][t] > 0.0f ? dh[r] : 0.0f;
            gb1 += d;
#pragma unroll
            for (int c = 0; c < 3; ++c) gw[c] = __builtin_fmaf(d, sm.x[r][c], gw[c]);
        }
        slab[3 * t] = gw[0]; slab[3 * t + 1] = gw[1]; slab[3 * t + 2] = gw[2];
        slab[768 + t] = gb1;
    }
}

// ================================================ dW2 = dZ2^T H1 on the f32 MFMA ================================================
// One wave per 16 x 64 strip of the 256 x 256 output (A fragment shared by 4 column tiles), K = batch rows.
// v_mfma_f32_16x16x4_f32: A lane (i = lane&15, g = lane>>4) = dZ2[row 4s+g][16mt + i], B lane = H1[row 4s+g][16nt + i];
// D lane (j = lane&15, g), reg r = dW2[16mt + 4g + r][16nt + j].
__global__ void __launch_bounds__(256) sac_dw2_gemm_kernel(const float* __restrict__ ws, int batch, int mat0, int n_mats, float* __restrict__ grads,
                                                           int grads_stride, int w2_off) {
    const int wave = (blockIdx.x * 256 + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    const int strips = 16 * 4;                  // 16 row tiles x 4 column groups of 64
    const int mat = wave / strips, st = wave % strips;
    if (mat >= n_mats) return;
    const int mt = st >> 2, ng = st & 3, i = lane & 15, g = lane >> 4;
    const float* H1 = ws + (size_t)(mat0 + mat) * ws_mat_floats(batch);
    const float* DZ2 = ws + (size_t)(3 + mat0 + mat) * ws_mat_floats(batch);
    f32x4 acc[4];
#pragma unroll
    for (int n = 0; n < 4; ++n) acc[n] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    for (int s = 0; s < batch / 4; ++s) {
        const size_t row = (size_t)(4 * s + g) * SA_H;
        const float a = DZ2[row + 16 * mt + i];
#pragma unroll
        for (int n = 0; n < 4; ++n) acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, H1[row + 64 * ng + 16 * n + i], acc[n], 0, 0, 0);
    }
    float* out = grads + (size_t)mat * grads_stride + w2_off;
#pragma unroll
    for (int n = 0; n < 4; ++n)
#pragma unroll
        for (int r = 0; r < 4; ++r) out[(size_t)(16 * mt + 4 * g + r) * SA_H + 64 * ng + 16 * n + i] = acc[n][r];
}

// thin gradients: sum the slabs in slab order and scatter into the flat gradient; block.y selects the net
__global__ void __launch_bounds__(256) sac_small_reduce_kernel(const float* __restrict__ ws, int batch, int n_slabs, int is_actor, double inv_count,
                                                               float* __restrict__ grads, float* __restrict__ out2) {
    const float* slabs = ws + 6 * ws_mat_floats(batch);
    const int e = blockIdx.x * 256 + threadIdx.x;
    const int per = is_actor ? 1794 : 1793, nets = is_actor ? 1 : 2;
    if (e < per * nets) {
        const int net = e / per, l = e % per;
        float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        for (int b = 0; b < n_slabs; ++b) acc[b & 3] += slabs[(size_t)b * SLAB + net * 1793 + l];
        const float v = (acc[0] + acc[1]) + (acc[2] + acc[3]);
        int dst;
        if (is_actor) dst = l < 768 ? AC_W1 + l : l < 1024 ? AC_B1 + (l - 768) : l < 1280 ? AC_B2 + (l - 1024) : l < 1536 ? AC_WM + (l - 1280)
                          : l == 1536 ? AC_BM : l < 1793 ? AC_WL + (l - 1537) : AC_BL;
        else dst = net * SQ_NP + (l < 1024 ? SQ_W1 + l : l < 1280 ? SQ_B1 + (l - 1024) : l < 1536 ? SQ_B2 + (l - 1280) : l < 1792 ? SQ_W3 + (l - 1536) : SQ_B3);
        grads[dst] = v;
    } else if (e < per * nets + 2 && out2) {
        const int k = e - per * nets;
        double s = 0.0;
        const int off = is_actor ? 1794 + k : 3586 + k;
        for (int b = 0; b < n_slabs; ++b) s += slabs[(size_t)b * SLAB + off];
        out2[k] = (float)(s * inv_count);
    }
}

static int sac_launch_reduce(void* workspace, int batch, int is_actor, double inv_count, float* grads, float* out2, hipStream_t s) {
    const int nb = (batch + SR - 1) / SR;
    const int n_out = (is_actor ? 1794 : 2 * 1793) + 2;
    sac_small_reduce_kernel<<<(n_out + 255) / 256, 256, 0, s>>>((const float*)workspace, batch, nb, is_actor, inv_count, grads, out2);
    MI_LAUNCH_CHECK();
    return MI_OK;
}

extern "C" int mi_sac_critic_grad(const float* q, const float* q_target, const float* actor, const float* observations, const float* actions,
                                  const float* rewards, const uint8_t* terminated, const int64_t* idx, int batch, int n_envs, int64_t slots,
                                  const float* eps, uint64_t seed, uint64_t update_index, const float* alpha, float gamma, double inv_count,
                                  void* workspace, float* grads, float* losses, void* stream) {
    MI_CHECK_ARG(q && q_target && actor && observations && actions && rewards && terminated && idx && alpha && workspace && grads, "NULL pointer");
    MI_CHECK_ARG(batch > 0 && batch % 4 == 0 && n_envs > 0 && slots >= 2, "batch must be a positive multiple of 4");
    hipStream_t s = (hipStream_t)stream;
    const int nb = (batch + SR - 1) / SR;
    sac_critic_kernel<<<nb, 256, 0, s>>>(q, q_target, actor, observations, actions, rewards, terminated, idx, batch, n_envs, (long long)slots, eps, seed,
                                         update_index, alpha, gamma, (float)inv_count, (float*)workspace);
    MI_LAUNCH_CHECK();
    sac_dw2_gemm_kernel<<<(2 * 64 * 64 + 255) / 256, 256, 0, s>>>((const float*)workspace, batch, 0, 2, grads, SQ_NP, SQ_W2);
    MI_LAUNCH_CHECK();
    return sac_launch_reduce(workspace, batch, 0, inv_count, grads, losses, s);
}

extern "C" int mi_sac_actor_grad(const float* actor, const float* q, const float* observations, const int64_t* idx, int batch, const float* eps,
                                 uint64_t seed, uint64_t update_index, const float* alpha, double inv_count, void* workspace, float* grads, float* out,
                                 void* stream) {
    MI_CHECK_ARG(actor && q && observations && idx && alpha && workspace && grads, "NULL pointer");
    MI_CHECK_ARG(batch > 0 && batch % 4 == 0, "batch must be a positive multiple of 4");
    hipStream_t s = (hipStream_t)stream;
    const int nb = (batch + SR - 1) / SR;
    sac_actor_kernel<<<nb, 256, 0, s>>>(actor, q, observations, idx, batch, eps, seed, update_index, alpha, (float)inv_count, (float*)workspace, 0);
    MI_LAUNCH_CHECK();
    sac_dw2_gemm_kernel<<<(64 * 64 + 255) / 256, 256, 0, s>>>((const float*)workspace, batch, 2, 1, grads, AC_NP, AC_W2);
    MI_LAUNCH_CHECK();
    return sac_launch_reduce(workspace, batch, 1, inv_count, grads, out, s);
}

// ================================================ alpha, Adam, polyak ============================================================
// mean_in: nullable device scalar holding the (already all-reduced) mean log-prob; NULL = sum this rank's slabs
__global__ void sac_alpha_kernel(const float* __restrict__ ws, int batch, int n_slabs, const float* __restrict__ mean_in, float inv_count,
                                 float* __restrict__ mean_out, float target_entropy, float* __restrict__ log_alpha, float* __restrict__ m,
                                 float* __restrict__ v, float w1, float b2, float w2, float step_size, float bc2_sqrt, float eps,
                                 float* __restrict__ alpha, float* __restrict__ out) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    float mean_lp;
    if (mean_in) mean_lp = mean_in[0];
    else {
        const float* slabs = ws + 6 * ws_mat_floats(batch);
        double s = 0.0;
        for (int b = 0; b < n_slabs; ++b) s += slabs[(size_t)b * SLAB + 1795];
        mean_lp = (float)(s * (double)inv_count);
    }
    if (mean_out) { mean_out[0] = mean_lp; return; }
    const float la = log_alpha[0];
    const float g = -(mean_lp + target_entropy);              // d/d log_alpha of mean(-log_alpha * (logp + target_entropy)), sac.py:205
    if (out) { out[0] = -la * (mean_lp + target_entropy); out[1] = g; }
    const float mi = m[0] + w1 * (g - m[0]);
    const float vi = v[0] * b2 + w2 * (g * g);
    m[0] = mi; v[0] = vi;
    const float nla = la + (-step_size) * (mi / (sqrtf(vi) / bc2_sqrt + eps));
    log_alpha[0] = nla;
    alpha[0] = expf(nla);                                      // :210
}

static int sac_launch_logp(const float* actor, const float* observations, const int64_t* idx, int batch, const float* eps, uint64_t seed,
                           uint64_t update_index, void* workspace, hipStream_t s) {
    sac_actor_kernel<<<(batch + SR - 1) / SR, 256, 0, s>>>(actor, nullptr, observations, idx, batch, eps, seed, update_index, nullptr, 0.0f, (float*)workspace, 1);
    MI_LAUNCH_CHECK();
    return MI_OK;
}

static int sac_launch_alpha(const float* ws, int batch, const float* mean_in, float inv_count, float target_entropy, float* log_alpha, float* exp_avg,
                            float* exp_avg_sq, int64_t step, double lr, float* alpha, float* out, hipStream_t s) {
    const double b1 = 0.9, b2 = 0.999, bc1 = 1.0 - pow(b1, (double)step), bc2 = 1.0 - pow(b2, (double)step);
    sac_alpha_kernel<<<1, 64, 0, s>>>(ws, batch, (batch + SR - 1) / SR, mean_in, inv_count, nullptr, target_entropy, log_alpha, exp_avg, exp_avg_sq,
                                      (float)(1.0 - b1), (float)b2, (float)(1.0 - b2), (float)(lr / bc1), (float)sqrt(bc2), 1e-8f, alpha, out);
    MI_LAUNCH_CHECK();
    return MI_OK;
}

extern "C" int mi_sac_alpha_step(const float* actor, const float* observations, const int64_t* idx, int batch, const float* eps, uint64_t seed,
                                 uint64_t update_index, float target_entropy, float* log_alpha, float* exp_avg, float* exp_avg_sq, int64_t step,
                                 double lr, float* alpha, float* out, void* workspace, void* stream) {
    MI_CHECK_ARG(actor && observations && idx && log_alpha && exp_avg && exp_avg_sq && alpha && workspace, "NULL pointer");
    MI_CHECK_ARG(batch > 0 && step >= 1, "bad arguments");
    const int rc = sac_launch_logp(actor, observations, idx, batch, eps, seed, update_index, workspace, (hipStream_t)stream);
    if (rc) return rc;
    return sac_launch_alpha((const float*)workspace, batch, nullptr, 1.0f / (float)batch, target_entropy, log_alpha, exp_avg, exp_avg_sq, step, lr, alpha, out,
                            (hipStream_t)stream);
}

extern "C" int mi_sac_mean_logp(const float* actor, const float* observations, const int64_t* idx, int batch, const float* eps, uint64_t seed,
                                uint64_t update_index, double inv_count, float* mean_logp, void* workspace, void* stream) {
    MI_CHECK_ARG(actor && observations && idx && mean_logp && workspace && batch > 0, "bad arguments");
    const int rc = sac_launch_logp(actor, observations, idx, batch, eps, seed, update_index, workspace, (hipStream_t)stream);
    if (rc) return rc;
    sac_alpha_kernel<<<1, 64, 0, (hipStream_t)stream>>>((const float*)workspace, batch, (batch + SR - 1) / SR, nullptr, (float)inv_count, mean_logp, 0.0f, nullptr,
                                                       nullptr, nullptr, 0.0f, 0.0f, 0.0f, 0.0f, 1.0f, 0.0f, nullptr, nullptr);
    MI_LAUNCH_CHECK();
    return MI_OK;
}

extern "C" int mi_sac_alpha_adam(const float* mean_logp, float target_entropy, float* log_alpha, float* exp_avg, float* exp_avg_sq, int64_t step,
                                 double lr, float* alpha, float* out, void* stream) {
    MI_CHECK_ARG(mean_logp && log_alpha && exp_avg && exp_avg_sq && alpha && step >= 1, "bad arguments");
    return sac_launch_alpha(nullptr, 0, mean_logp, 0.0f, target_entropy, log_alpha, exp_avg, exp_avg_sq, step, lr, alpha, out, (hipStream_t)stream);
}

__global__ void __launch_bounds__(256) adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, int n,
                                                    float w1, float b2, float w2, float step_size, float bc2_sqrt, float eps) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float gi = g[i];
    const float mi = m[i] + w1 * (gi - m[i]);
    const float vi = v[i] * b2 + w2 * (gi * gi);
    m[i] = mi; v[i] = vi;
    p[i] = p[i] + (-step_size) * (mi / (sqrtf(vi) / bc2_sqrt + eps));
}

extern "C" int mi_adam(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int n, int64_t step, double lr, double beta1,
                       double beta2, double eps, void* stream) {
    MI_CHECK_ARG(params && grads && exp_avg && exp_avg_sq && n > 0 && step >= 1, "bad arguments");
    const double bc1 = 1.0 - pow(beta1, (double)step), bc2 = 1.0 - pow(beta2, (double)step);
    adam_kernel<<<(n + 255) / 256, 256, 0, (hipStream_t)stream>>>(params, grads, exp_avg, exp_avg_sq, n, (float)(1.0 - beta1), (float)beta2,
                                                                 (float)(1.0 - beta2), (float)(lr / bc1), (float)sqrt(bc2), (float)eps);
    MI_LAUNCH_CHECK();
    return MI_OK;
}

__global__ void __launch_bounds__(256) polyak_kernel(float* __restrict__ t, const float* __restrict__ p, int n, float tau) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) t[i] = tau * p[i] + (1.0f - tau) * t[i];
}

extern "C" int mi_polyak(float* target, const float* param, int n, float tau, void* stream) {
    MI_CHECK_ARG(target && param && n > 0, "bad arguments");
    polyak_kernel<<<(n + 255) / 256, 256, 0, (hipStream_t)stream>>>(target, param, n, tau);
    MI_LAUNCH_CHECK();
    return MI_OK;
}
