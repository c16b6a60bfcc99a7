// mi_rollout.hip — the whole rollout loop of reference ppo.py:110-141 in ONE launch (rollout_q4_kernel), plus the batch forward that backs
// the nn.Module-style API (forward_kernel).
//
// Envs are independent for the whole rollout (weights are frozen), so there is no inter-workgroup communication at all.
// Storage index convention (ppo.py:113-141): actions/log_probs at t, obs/values/rewards/dones at t+1; on done the stored obs/value
// are those of the RESET state.  critic(obs[t+1]) and actor(obs[t+1]) see the same observation, so both nets are evaluated once per
// step (ppo.py:139 and :120 of the next iteration).
//
// History (numbers in profiles/r02_rollout_stamps.txt; the code of the superseded forms is in git history only):
//   r01  rollout_kernel<E>     one wave per E envs, lane = hidden unit, VALU FMAs from register-resident rows          0.36-0.40 ms
//   r01  rollout_mfma_kernel   16 envs per 4-wave workgroup on 16x16x4 MFMA, one barrier per step                      0.245 ms
//   r02  rollout_q4_kernel     4 envs per actor / critic wave pair on the 16-block 4x4x1 MFMA, no barrier (below)      0.207-0.216 ms
#include "mi_common.h"

// ---- 4 envs per actor / critic wave pair on the 16-block 4x4x1 MFMA ---------------------------------------------------------------
// The r01 16-env workgroup spent 2,900 of its ~4,000 cycles per step in the two nets although their MFMAs need 1,150: four waves
// met at a barrier every step, each evaluated half a net, and the critic sat on the critical path although nothing waits for a value
// before the GAE scan.  This form cuts the dependency chain to what the algorithm needs:
//   * a workgroup = 2 waves owns 4 envs: wave 0 = ACTOR (forward, draw, env step — NO global memory traffic inside its loop), wave 1 =
//     CRITIC (forward on the observations the actor publishes through an LDS ring, ALL the storage writes, GAE scan at the end): a
//     vector store costs its issuing wave ~100 cycles whatever its lane count, and the actor's chain is what the launch lasts.  No barrier in the loop: the actor never waits for
//     the critic (ring depth T + 1 when T <= 128), the critic polls an LDS counter.  1024 workgroups at N = 4096: 8 waves per CU.
//   * v_mfma_f32_4x4x1_16b_f32 (16 blocks of 4x4, K = 1) with the A operand BROADCAST from one block (cbsz = 4, abid = b'):
//         D[b][i][j] += A[b'][i] * B[b][j]      i = env (4), block b / column j = output unit 4b + j = the LANE, one k per instruction
//     so the accumulator holds out[unit = lane][env = register], the B operand of k-step k is W[unit = lane][k] — every lane keeps ITS
//     row of W2 (64 VGPRs), W1 (4) and W3 resident for the whole launch — and a layer of 64 outputs x 64 inputs x 4 envs is 64 MFMAs of
//     8 cycles at full utilisation (the 16x16x4 form needs 16 envs per wave for that).  The A operand of k-step k = 4b' + jj is the
//     previous layer's activation h[k][env i] in lanes 4b' + i: a 4x4 transpose inside every quad of lanes (quad_transpose: 8 DPP moves)
//     turns the accumulator (lane 4b' + q, register e = h[4b' + q][e]) into exactly that (register jj, abid b').
//   * heads: per-lane products, then quad_env_reduce — a butterfly that halves the values per lane at each of its two quad stages, two
//     row rotations and the two cross-row permlane swaps: 18 instructions per output, the total of env (lane & 3) in every lane.
//   * every lane carries the fp64 state of env (lane & 3) and runs the scalar section (draw, CartPole step, TimeLimit, auto-reset)
//     redundantly, as before — identical IEEE sequences, identical bits; lanes 0..3 write.  The critic wave runs no physics at all.
// Layouts (CBSZ / ABID broadcast, D register = i) are verified on the device with exact integers: mi_selftest_mfma probe 4.
#define RQ_ENVS 4
#define RQ_GAE_T 128
#define RQ_RING (RQ_GAE_T + 1)
#define RQ_UNI_BLOCKS (RQ_GAE_T / 4 + 2)   // Philox blocks (4 steps each) a launch of <= RQ_GAE_T steps can touch, whatever its first step's phase

#define RQ_CHAINS 4   // independent accumulator chains of layer 2 (k mod RQ_CHAINS); 8 measured no faster: the wave is issue-bound, not MFMA-latency-bound
template <int K>
struct rq_layer2 {   // k-steps K..63
    static __device__ __forceinline__ void run(const float (&w2)[HID], const rq_f32x4& tr, rq_f32x4 (&acc)[RQ_CHAINS]) {
        acc[K % RQ_CHAINS] = __builtin_amdgcn_mfma_f32_4x4x1f32(tr[K & 3], w2[K], acc[K % RQ_CHAINS], 4, K >> 2, 0);
        rq_layer2<K + 1>::run(w2, tr, acc);
    }
};
template <>
struct rq_layer2<HID> {
    static __device__ __forceinline__ void run(const float (&)[HID], const rq_f32x4&, rq_f32x4 (&)[RQ_CHAINS]) {}
};

// one net's hidden layers on 4 envs: ob = this lane's env's observation (env = lane & 3); returns tanh(layer 2)[unit = lane][env = register]
__device__ __forceinline__ rq_f32x4 rq_hidden(const float4& ob, const float (&w1)[OBS], float b1, const float (&w2)[HID], float b2, bool q0, bool q1) {
    rq_f32x4 h = {b1, b1, b1, b1}, hb = {0.0f, 0.0f, 0.0f, 0.0f};       // two chains
    h = __builtin_amdgcn_mfma_f32_4x4x1f32(ob.x, w1[0], h, 4, 0, 0);   // A: lanes 0..3 of block 0 = obs component k of envs 0..3
    hb = __builtin_amdgcn_mfma_f32_4x4x1f32(ob.y, w1[1], hb, 4, 0, 0);
    h = __builtin_amdgcn_mfma_f32_4x4x1f32(ob.z, w1[2], h, 4, 0, 0);
    hb = __builtin_amdgcn_mfma_f32_4x4x1f32(ob.w, w1[3], hb, 4, 0, 0);
    h = h + hb;
#pragma unroll
    for (int e = 0; e < 4; ++e) h[e] = mi_tanhf(h[e]);
    const rq_f32x4 tr = quad_transpose(h, q0, q1);
    rq_f32x4 acc[RQ_CHAINS];
#pragma unroll
    for (int c = 0; c < RQ_CHAINS; ++c) acc[c] = rq_f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    acc[0] = rq_f32x4{b2, b2, b2, b2};
    rq_layer2<0>::run(w2, tr, acc);
#pragma unroll
    for (int st = RQ_CHAINS / 2; st > 0; st >>= 1)
#pragma unroll
        for (int c = 0; c < st; ++c) acc[c] = acc[c] + acc[c + st];
    rq_f32x4 z = acc[0];
#pragma unroll
    for (int e = 0; e < 4; ++e) z[e] = mi_tanhf(z[e]);
    return z;
}

#ifdef RQ_STAMPS   // diagnostic build: where the actor wave's cycles go (s_memtime), read back by mi_debug_rollout_stamps
__device__ unsigned long long rq_stamp_dbg[1024 * 8 + 1024];   // [8192 ..): per-step wall-clock marks of workgroup 517 (tools/rollout_timeline.py)
#define RQ_STAMP(k) do { __builtin_amdgcn_sched_barrier(0); { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); \
                         rq_acc[k] += t_ - rq_last; rq_last = t_; } __builtin_amdgcn_sched_barrier(0); } while (0)
extern "C" int mi_debug_rollout_stamps(unsigned long long* out, int n) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(rq_stamp_dbg), sizeof(unsigned long long) * (size_t)(n < 9216 ? n : 9216)) == hipSuccess ? 0 : -2;
}
// wall-clock mark k of workgroup 517 (tools/rollout_timeline.py): 600 kernel entry, 601 actor weights in registers, 700 critic loop + last flush done, 701 GAE scanned, 702 GAE written
#define RQ_MARK(k) do { if (blockIdx.x == 517 && (threadIdx.x & 63) == 0) { unsigned long long rt_; asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(rt_) :: "memory"); rq_stamp_dbg[8192 + (k)] = rt_; } } while (0)
#else
#define RQ_STAMP(k) do {} while (0)
#define RQ_MARK(k) do {} while (0)
#endif

// LDS mailbox words between the waves of a workgroup.  The accesses must be ds_read / ds_write: a `volatile int*` to __shared__ memory
// compiles to FLAT instructions with sc0 sc1 and an s_waitcnt vmcnt(0) behind every store, i.e. each publish would wait for all of the
// wave's outstanding GLOBAL stores (measured: 500 cycles per step).  The LDS executes one wave's accesses in program order, so data
// written before the counter is visible before it; the empty asm statements keep the COMPILER from reordering around them.
// Forward progress: both waves belong to ONE workgroup, so they are resident together by construction and each waits only for a counter the other one advances
// unconditionally (the actor never waits while T + 1 <= RQ_RING; with a wrapped ring it waits for a slot the critic frees after a bounded amount of work, and the
// critic for a slot the actor fills likewise): unlike the inter-workgroup waits of mi_sac.hip these polls need no co-residency argument and no time budget.
typedef __attribute__((address_space(3))) int rq_lds_int;
__device__ __forceinline__ void lds_publish(int* word, int v) {
    asm volatile("" ::: "memory");
    *(volatile rq_lds_int*)(rq_lds_int*)word = v;
    asm volatile("" ::: "memory");
}
__device__ __forceinline__ int lds_peek(int* word) {
    const int v = *(volatile rq_lds_int*)(rq_lds_int*)word;
    asm volatile("" ::: "memory");
    return v;
}

struct __attribute__((aligned(16))) rq_slot {   // what the actor publishes per step and env: slot (t + 1) % RQ_RING
    float4 ob;                  // obs[t+1] (the reset observation where done)
    float rew, dn, logp;        // rewards[t+1], dones[t+1]; log_probs[t]
    int act;                    // actions[t]
};
struct __attribute__((aligned(16))) rq_smem {
    rq_slot ring[RQ_RING][RQ_ENVS];
    union {
        float4 wstage[HID * HID / 4];   // launch start: W2 of one net on its way into the lanes' registers (rq_load_w2): the actor wave first, then the critic wave
        struct {                        // launch end: fused GAE — advantages / returns of the 4 envs, scanned by 4 lanes, written out by all 64
            float ga[RQ_GAE_T + 1][RQ_ENVS];
            float gr[RQ_GAE_T + 1][RQ_ENVS];
        } g;
    };
    float gv[RQ_GAE_T + 1][RQ_ENVS];    // the critic's values (T <= RQ_GAE_T: every row of the rollout; rewards / dones: the ring, which does not wrap then)
    float uni[RQ_UNI_BLOCKS * 4][RQ_ENVS];   // the action uniforms of the launch's steps (RNG contract stream 1), drawn up front by the critic wave: row = step - 4 (step0 >> 2)
    int produced;                       // slots published so far (actor -> critic)
    int consumed;                       // slots the critic is done with (critic -> actor; only read when T + 1 > RQ_RING)
};
// 37.2 KB per workgroup: 4 workgroups (8 waves) per CU, what the headline's 1,024 workgroups need on 256 CUs.

// A lane's row of W2 (64 floats) into its registers.  Round 2 let every lane read its own row straight from global memory — 16 float4 loads whose 64 lanes touch 64
// different 256-byte rows each: 5.6 us until the first step could start (profiles/r02_rollout_stamps.txt).  Here the wave reads the matrix as 16 fully coalesced 1 KB
// loads, drops it into LDS with the 16-byte chunks of row r XOR-swizzled by (r & 15) (stores and row reads are both conflict-free in the b128 lane groups),
// and every lane picks up its row with 16 ds_read_b128.
__device__ __forceinline__ void rq_load_w2(const float* __restrict__ W2, float (&w2)[HID], float4* stage, int lane) {
    float4 v[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = reinterpret_cast<const float4*>(W2)[i * 64 + lane];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int r = 4 * i + (lane >> 4), j = lane & 15;
        stage[r * 16 + (j ^ (r & 15))] = v[i];
    }
    wave_lds_fence();
#pragma unroll
    for (int jj = 0; jj < 16; ++jj) {
        const float4 q = stage[lane * 16 + (jj ^ (lane & 15))];
        w2[4 * jj] = q.x; w2[4 * jj + 1] = q.y; w2[4 * jj + 2] = q.z; w2[4 * jj + 3] = q.w;
    }
    wave_lds_fence();
}

// FORCED: parity mode (forced_actions / forced_uniforms / forced_resets may be given); EPLOG: the per-episode list is kept (max_ep > 0).
// Separate instantiations, not run-time branches: a conditional global LOAD (or returning atomic) inside the step loop makes the compiler
// place s_waitcnt vmcnt(0) at the join, which is executed on EVERY step and drains all of the wave's outstanding stores (~500 cycles).
template <bool FORCED, bool EPLOG>
__global__ void __launch_bounds__(128)
rollout_q4_kernel(mi_env e, const float* __restrict__ params, int T, float* __restrict__ obs_cur, float* __restrict__ observations,
                  float* __restrict__ values, int64_t* __restrict__ actions, float* __restrict__ log_probs,
                  float* __restrict__ rewards, float* __restrict__ dones, const int64_t* __restrict__ forced_actions,
                  const float* __restrict__ forced_uniforms, const double* __restrict__ forced_resets,
                  mi_episode_t* __restrict__ episodes, int32_t* __restrict__ episode_stats, int max_ep,
                  float* __restrict__ adv, float* __restrict__ returns, float gamma, float lam, double* __restrict__ zero_f64, int zero_n,
                  int32_t* __restrict__ zero_i32, int32_t* __restrict__ stats_part) {
    __shared__ rq_smem sm;
    MI_INSIDE_SCOPE(MI_PROF_ROLLOUT);
#ifdef RQ_STAMPS
    if (blockIdx.x == 517 && threadIdx.x == 0) { unsigned long long rt_; asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(rt_) :: "memory"); rq_stamp_dbg[8192 + 600] = rt_; }
#endif
    if (zero_f64 && blockIdx.x == 0) for (int k = threadIdx.x; k < zero_n; k += 128) zero_f64[k] = 0.0;   // scratch the next launches accumulate into
    if (zero_i32 && blockIdx.x == 0 && threadIdx.x < 4) zero_i32[threadIdx.x] = 0;                        // the NEXT rollout's episode statistics (double-buffered by the caller)
    // (wave 0 = actor, wave 1 = critic in EVERY workgroup: the dispatcher already places the 2,048 waves of the headline launch so that each SIMD hosts exactly one
    //  actor and one critic — tools/rollout_placement.py; swapping the roles by a block-index bit breaks that: 226 - 256 us against 230)
    const int lane = threadIdx.x & 63, net = threadIdx.x >> 6, en = lane & 3;
    const bool q0 = lane & 1, q1 = lane & 2;
    const int N = e.n;
    // XCD-aware env-group map.  Workgroups are dealt round-robin over the 8 XCDs (b % 8), each with an L2 of its own, and a group of 4 envs writes 16-byte pieces of
    // the (T+1, N) tensors' rows: with group = blockIdx the eight groups that share a 128-byte line sit on eight different XCDs and every L2 holds one dirty piece of
    // the line; with group = (b % 8) * (grid / 8) + b / 8 neighbouring groups share an XCD, whose L2 can merge their pieces into whole lines before they leave.
    // Measured neutral on the launch time (204.2 vs 204.4 us, profiles/r03_rollout_notes.txt: the launch does not wait for its stores) — kept for the HBM traffic.
    // Results do not depend on which workgroup carries an env (keyed RNG by global env id).
    const unsigned nblk = gridDim.x;
    const unsigned egrp = (nblk & 7u) ? blockIdx.x : (blockIdx.x & 7u) * (nblk >> 3) + (blockIdx.x >> 3);
    const int i = (int)egrp * RQ_ENVS + en;
    const bool mine = i < N, writer = mine && lane < RQ_ENVS;
    const int g = mine ? i : N - 1;   // lanes past the end shadow the last env and never write
    if (threadIdx.x == 0) { sm.produced = 0; sm.consumed = 0; }
#ifdef RQ_STAMPS
    if (lane == 0 && blockIdx.x < 1024) rq_stamp_dbg[blockIdx.x * 8 + 4 + (threadIdx.x >> 6)] = ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32) | __builtin_amdgcn_s_getreg((31 << 11) | 4);
#endif
    // ---- this wave's net: lane = hidden unit; its rows of W1 / W2, its column of W3 ----
    const float* p = params + (net ? C_BASE : 0);
    float w1[OBS], w2[HID];
    {
        const float4 a = *reinterpret_cast<const float4*>(p + N_W1 + 4 * lane);
        w1[0] = a.x; w1[1] = a.y; w1[2] = a.z; w1[3] = a.w;
    }
    const float b1 = p[N_B1 + lane], b2 = p[N_B2 + lane];
    const float w3a = p[N_W3 + lane], w3b = net == 0 ? p[N_W3 + HID + lane] : 0.0f;
    const float b3a0 = params[A_B3], b3a1 = params[A_B3 + 1], b3c = params[C_BASE + N_W3 + HID];
    const bool ringed = T + 1 > RQ_RING;   // long rollouts: the ring wraps and the actor must not overrun the critic
    // W2 through the one staging image: the actor (the chain the launch lasts) first; the critic takes its turn behind the barrier, which the actor only passes through
    if (net == 0) { rq_load_w2(p + N_W2, w2, sm.wstage, lane); RQ_MARK(601); }
    else if (!ringed && (!FORCED || (!forced_actions && !forced_uniforms))) {
        // the critic wave has nothing to do until the actor publishes the first observation: it draws the launch's action uniforms (one Philox block feeds 4 steps of an
        // env; lane l = (block l >> 2, env l & 3): all 64 lanes useful) — ~40 instructions per step that no longer sit in the actor wave's chain
        const uint64_t blk0 = e.step_ctr[g] >> 2;
#pragma unroll 1
        for (int j = lane >> 2; j < RQ_UNI_BLOCKS; j += 16) {
            uint32_t r4[4];
            mi_philox(e.seed, e.env_id_base + (uint64_t)g, blk0 + (uint64_t)j, STREAM_ACTION, r4);
#pragma unroll
            for (int w = 0; w < 4; ++w) sm.uni[4 * j + w][en] = mi_u32_to_uniform(r4[w]);
        }
    }
    __syncthreads();                       // counters initialised; the actor is done with the staging image
    if (net == 1) rq_load_w2(p + N_W2, w2, sm.wstage, lane);

    if (net == 0) {
        // ================================================= ACTOR wave =================================================
        __builtin_amdgcn_s_setprio(2);     // the dependent chain of the launch: win every issue arbitration against the critic waves
        double sx = e.x[g], sxd = e.x_dot[g], sth = e.theta[g], sthd = e.theta_dot[g];
        int elapsed = e.elapsed[g], eplen = e.ep_len[g];
        float epret = e.ep_ret[g];
        uint64_t episode = e.episode[g], stepctr = e.step_ctr[g];
        float4 ob = reinterpret_cast<const float4*>(obs_cur)[g];
        // Categorical(logits) depends on the logits only through d = l0 - l1 (log-softmax and probabilities are shift-invariant,
        // ppo.py:52-59): ONE head reduction over (W3[0] - W3[1]) . h2 instead of two; the distribution is evaluated at (d, 0).
        float my_d = 0.0f;
        const float w3d = w3a - w3b, b3d = b3a0 - b3a1;
        int st_cnt = 0, st_len = 0, st_max = 0;
        uint32_t urand[4] = {0, 0, 0, 0};
        const bool keyed_actions = !FORCED || (!forced_actions && !forced_uniforms);
        const uint64_t ubase = stepctr & ~3ull;   // sm.uni row 0 = the first step of the Philox block this launch starts in
        if (keyed_actions && ringed && (stepctr & 3)) mi_philox(e.seed, e.env_id_base + (uint64_t)g, stepctr >> 2, STREAM_ACTION, urand);
        // every value loaded so far is consumed HERE, so that no load is pending when the loop starts (a pending load at the loop head
        // becomes an s_waitcnt vmcnt(small) inside the body, which in steady state waits for the previous step's stores instead)
        asm volatile("" : "+v"(ob.x), "+v"(ob.y), "+v"(ob.z), "+v"(ob.w), "+v"(sx), "+v"(sxd), "+v"(sth), "+v"(sthd), "+v"(elapsed), "+v"(eplen), "+v"(epret), "+v"(episode), "+v"(stepctr));
#ifdef RQ_STAMPS
        unsigned long long rq_acc[4] = {0, 0, 0, 0}, rq_last;
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(rq_last) :: "memory");
#endif
        for (int t = -1; t < T; ++t) {
            float rew = 0.0f, dn = 0.0f, logp_t = 0.0f;
            int act_t = 0;
            if (t >= 0) {
                // ---- sample from the logits of obs[t], step the env (ppo.py:120-129) ----
                const size_t row = (size_t)t * N + g;
                float nl0, nl1, p0, p1, ent;
                mi_categorical2_fast(my_d, 0.0f, nl0, nl1, p0, p1, ent);
                int a;
                if (FORCED && forced_actions) a = (int)forced_actions[row];
                else {
                    float u;
                    if (FORCED && forced_uniforms) u = forced_uniforms[row];
                    else if (!ringed) u = sm.uni[(int)(stepctr - ubase)][en];   // drawn up front by the critic wave
                    else {
                        const uint32_t wd = (uint32_t)stepctr & 3u;
                        if (wd == 0) mi_philox(e.seed, e.env_id_base + (uint64_t)g, stepctr >> 2, STREAM_ACTION, urand);
                        u = mi_u32_to_uniform(wd == 0 ? urand[0] : wd == 1 ? urand[1] : wd == 2 ? urand[2] : urand[3]);
                    }
                    a = (u >= p0) ? 1 : 0;
                }
                stepctr += 1;
                act_t = a; logp_t = a ? nl1 : nl0;   // ppo.py:123-124 (stored by the critic wave)
                int term;
                mi_cartpole_step(sx, sxd, sth, sthd, a, term);
                elapsed += 1;
                const bool d = term || elapsed >= CP_MAX_STEPS;
                epret += 1.0f;
                eplen += 1;
                if (d) {
                    if (writer) {
                        st_cnt += 1; st_len += eplen; st_max = eplen > st_max ? eplen : st_max;
                        if (EPLOG && max_ep > 0 && episode_stats) {
                            const int slot = atomicAdd(episode_stats + 3, 1);
                            if (slot < max_ep) episodes[slot] = mi_episode_t{g, t, epret, eplen};
                        }
                    }
                    epret = 0.0f; eplen = 0; elapsed = 0;
                    double sr[4];
                    if (FORCED && forced_resets) {
#pragma unroll
                        for (int k = 0; k < 4; ++k) sr[k] = forced_resets[4 * row + k];
                    } else {
                        mi_reset_noise(e.seed, e.env_id_base + (uint64_t)g, episode, sr);
                    }
                    episode += 1;
                    sx = sr[0]; sxd = sr[1]; sth = sr[2]; sthd = sr[3];
                }
                ob = make_float4((float)sx, (float)sxd, (float)sth, (float)sthd);
                rew = 1.0f; dn = d ? 1.0f : 0.0f;
            }
            RQ_STAMP(0);   // scalar section
            // ---- publish obs[t+1] to the critic (LDS executes one wave's accesses in order: data first, then the counter) ----
            if (ringed) while (lds_peek(&sm.consumed) < t + 3 - RQ_RING) __builtin_amdgcn_s_sleep(2);
            // (the LDS executes one wave's accesses in program order, so the counter can never become visible before the data; a
            //  workgroup-scope RELEASE store would also wait for this wave's outstanding GLOBAL stores — 400 cycles per step — for nothing)
            if (lane < RQ_ENVS) {
                rq_slot& sl = sm.ring[(t + 1) % RQ_RING][lane];
                sl.ob = ob;
                *reinterpret_cast<float4*>(&sl.rew) = make_float4(rew, dn, logp_t, __builtin_bit_cast(float, act_t));
            }
            if (lane == 0) lds_publish(&sm.produced, t + 2);
            RQ_STAMP(1);   // publish + stores
            // ---- actor(obs[t+1]) (ppo.py:120 of the next iteration) ----
            const rq_f32x4 h2 = rq_hidden(ob, w1, b1, w2, b2, q0, q1);
            RQ_STAMP(2);   // hidden layers
            my_d = quad_env_reduce(h2 * w3d, q0, q1) + b3d;
            RQ_STAMP(3);   // head
#ifdef RQ_STAMPS
            if (blockIdx.x == 517 && lane == 0 && t + 1 < 200) { unsigned long long rt_; asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(rt_) :: "memory"); rq_stamp_dbg[8192 + t + 1] = rt_; }
#endif
        }
#ifdef RQ_STAMPS
        if (lane == 0 && blockIdx.x < 1024) { for (int k = 0; k < 4; ++k) rq_stamp_dbg[blockIdx.x * 8 + k] = rq_acc[k]; }
#endif
        __builtin_amdgcn_s_setprio(0);
        if (episode_stats || stats_part) {   // lanes 0..3: one flush per workgroup
            int c = writer ? st_cnt : 0, l = writer ? st_len : 0, m = writer ? st_max : 0;
#pragma unroll
            for (int sft = 1; sft < 4; sft <<= 1) { c += __shfl_xor(c, sft); l += __shfl_xor(l, sft); const int mo = __shfl_xor(m, sft); m = mo > m ? mo : m; }
            if (stats_part) { if (lane == 0) reinterpret_cast<int4*>(stats_part)[blockIdx.x] = make_int4(c, l, m, 0); }   // summed on request (mi_common.h: no atomics)
            else if (lane == 0 && c > 0) { atomicAdd(episode_stats, c); atomicAdd(episode_stats + 1, l); atomicMax(episode_stats + 2, m); }
        }
        if (writer) {   // carry-over `observation` and env state for the next rollout
            e.x[g] = sx; e.x_dot[g] = sxd; e.theta[g] = sth; e.theta_dot[g] = sthd;
            e.elapsed[g] = elapsed; e.ep_ret[g] = epret; e.ep_len[g] = eplen;
            e.episode[g] = episode; e.step_ctr[g] = stepctr;
            reinterpret_cast<float4*>(obs_cur)[g] = ob;
        }
    } else {
        // ================================================= CRITIC wave ================================================
        // Storage writes (ppo.py:113-141).  A vector store costs the issuing wave ~100 cycles whatever its lane count, and with 4 envs per workgroup a per-step store is 4
        // lanes wide (six of them per step in round 2: 600 cycles of this wave's issue on a SIMD it shares with an actor wave, 1.5x write amplification).  While the ring does
        // not wrap (T <= RQ_GAE_T) everything a row needs stays in LDS, so the rows are written 16 at a time by ALL 64 lanes — lane l = (row l >> 2, env l & 3): one
        // instruction per tensor and 16 rows, each row's 4 envs one contiguous 16 / 32 / 64-byte segment.
        const int rr = lane >> 2;                                  // row within a 16-row chunk (env = en = lane & 3)
        auto flush_rows = [&](int r0, int cnt) {
            wave_lds_fence();                                      // the values of these rows were written to sm.gv by lanes 0..3 of this wave
            const int r = r0 + rr;
            if (rr < cnt && mine) {
                const rq_slot& sl = sm.ring[r][en];
                const float4 ob = sl.ob;
                const float4 misc = *reinterpret_cast<const float4*>(&sl.rew);
                const size_t row = (size_t)r * N + g;
                reinterpret_cast<float4*>(observations)[row] = ob;      // :113,:137 (the reset obs where done)
                values[row] = sm.gv[r][en];                             // :115,:139
                if (r >= 1) {
                    rewards[row] = misc.x; dones[row] = misc.y;         // :140-141
                    log_probs[row - N] = misc.z; actions[row - N] = (int64_t)__builtin_bit_cast(int, misc.w);   // :123-124
                }
            }
        };
        if (!ringed) {
            for (int t = -1; t < T; ++t) {
                while (lds_peek(&sm.produced) < t + 2) __builtin_amdgcn_s_sleep(1);   // the ring read below is issued after the poll that saw the counter
                const rq_slot& sl = sm.ring[t + 1][en];
                const float4 ob = sl.ob;
                const rq_f32x4 h2 = rq_hidden(ob, w1, b1, w2, b2, q0, q1);
                const float val = quad_env_reduce(h2 * w3a, q0, q1) + b3c;   // ppo.py:115,:139
                if (lane < RQ_ENVS) sm.gv[t + 1][lane] = val;
                if (((t + 1) & 15) == 15) flush_rows(t + 1 - 15, 16);
#ifdef RQ_STAMPS
                if (blockIdx.x == 517 && lane == 0 && t + 1 < 200) { unsigned long long rt_; asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(rt_) :: "memory"); rq_stamp_dbg[8192 + 256 + t + 1] = rt_; }
#endif
            }
            RQ_MARK(703);
            if ((T + 1) & 15) flush_rows((T + 1) & ~15, (T + 1) & 15);
            RQ_MARK(700);
        } else {
            for (int t = -1; t < T; ++t) {
                while (lds_peek(&sm.produced) < t + 2) __builtin_amdgcn_s_sleep(1);
                const rq_slot& sl = sm.ring[(t + 1) % RQ_RING][en];
                const float4 ob = sl.ob;
                const float4 misc = *reinterpret_cast<const float4*>(&sl.rew);
                const rq_f32x4 h2 = rq_hidden(ob, w1, b1, w2, b2, q0, q1);
                const float val = quad_env_reduce(h2 * w3a, q0, q1) + b3c;
                if (writer) {   // the storage of time step t (action, log-prob) and t + 1 (everything that resulted from it)
                    const size_t row = (size_t)(t + 1) * N + g;
                    reinterpret_cast<float4*>(observations)[row] = ob;
                    values[row] = val;
                    if (t >= 0) {
                        rewards[row] = misc.x; dones[row] = misc.y;
                        log_probs[row - N] = misc.z; actions[row - N] = (int64_t)__builtin_bit_cast(int, misc.w);
                    }
                }
                if (lane == 0) lds_publish(&sm.consumed, t + 2);
            }
        }
        if (adv) {       // GAE (ppo.py:144-151; expression order of gae_kernel): T <= RQ_GAE_T here, every reward / done / value of the 4 envs is in LDS
            // The scan is sequential in t (the expression order is the parity contract): 4 lanes, 16 steps at a time — the block's 48 inputs are requested from LDS
            // together (as one dependent load -> compute -> store chain per step the scan took 7.9 us: round-3 timeline), then 16 x 5 dependent operations; the block's
            // 16 rows are then written by all 64 lanes while the next block is scanned.
            float last = 0.0f, vnext = 0.0f;
            if (lane < RQ_ENVS) vnext = sm.gv[T][lane];
            if (writer) { adv[(size_t)T * N + g] = 0.0f; returns[(size_t)T * N + g] = 0.0f + vnext; }
            for (int hi = T - 1; hi >= 0; hi -= 16) {
                const int lo = hi >= 15 ? hi - 15 : 0;
                if (lane < RQ_ENVS) {
                    float vc[16], dn[16], rw[16];
#pragma unroll
                    for (int k = 0; k < 16; ++k) {
                        const int t = hi - k >= lo ? hi - k : lo;
                        vc[k] = sm.gv[t][lane]; dn[k] = sm.ring[t + 1][lane].dn; rw[k] = sm.ring[t + 1][lane].rew;
                    }
#pragma unroll
                    for (int k = 0; k < 16; ++k) {
                        const int t = hi - k;
                        if (t >= lo) {
                            const float vcur = vc[k];
                            const float a = gamma * (1.0f - dn[k]);
                            const float b = vnext + lam * last;
                            float v = rw[k] + a * b;
                            v = v - vcur;
                            sm.g.ga[t][lane] = v;
                            sm.g.gr[t][lane] = v + vcur;
                            last = v;
                            vnext = vcur;
                        }
                    }
                }
                wave_lds_fence();
                if (hi == T - 1) RQ_MARK(705);
                const int r = lo + rr;
                if (r <= hi && mine) {
                    const size_t c = (size_t)r * N + g;
                    adv[c] = sm.g.ga[r][en];
                    returns[c] = sm.g.gr[r][en];
                }
                if (hi == T - 1) RQ_MARK(706);
            }
            RQ_MARK(702);
        }
    }
}

// Negative result (measured, removed; profiles/r02_rollout_stamps.txt): a THIRD wave per 4 envs that computes both successors of the
// current state while the actor evaluates the policy on it (speculation over CartPole's two actions; the positions and the termination
// test are action-independent).  Correct and bit-identical, but 0.258 ms against 0.216: the two-action physics costs 2,050 cycles per
// step (two IEEE fp64 divisions, ~130 fp64 operations at 3 waves per SIMD), more than the actor's whole step, and the launch is not purely
// latency-bound — per SIMD and step the two nets' MFMAs (1,088 cycles), their VALU work (~800) and one physics step (~600) already fill
// ~75 % of the issue slots, so work added to shorten the chain is paid in full.

// Negative result (round 4, measured, removed): the CRITIC wave preparing every env's next reset state (Philox + four fp64 conversions, ~300 instructions) in LDS so that
// the actor takes it on `done` instead of computing it on its dependent chain (with a fallback to computing it when the critic lags; bit-identical).  187.4 against 188.0 us
// at the initial policy (episodes of 22 steps: a reset in 18 % of a wave's steps), 183.5 against 182.4 us after 40 updates (tools/rollout_ab.py, two runs each): nothing.
// The "22 us for short episodes" of round 3 was mostly a cold-GPU artefact (202 us for the first 40 launches of a process, 188 us warm, whatever the build).

__global__ void zero_i32x4_kernel(int32_t* p) { if (threadIdx.x < 4) p[threadIdx.x] = 0; }

static int rollout_impl(void* handle, const float* params, int T, float* obs_cur, float* observations, float* values,
                        int64_t* actions, float* log_probs, float* rewards, float* dones, const int64_t* forced_actions,
                        const float* forced_uniforms, const double* forced_resets, mi_episode_t* episodes,
                        int32_t* episode_stats, int max_ep, float* advantages, float* returns, float gamma, float lam, double* zero_f64, int zero_n,
                        int32_t* stats_next, void* stream) {
    MI_CHECK_ARG(handle && params && obs_cur && observations && values && actions && log_probs && rewards && dones, "NULL pointer");
    MI_CHECK_ARG(!stats_next || (episode_stats && stats_next != episode_stats), "episode_stats_next must be a second statistics buffer");
    MI_CHECK_ARG(T > 0, "T must be positive");
    MI_CHECK_ARG(max_ep >= 0 && (max_ep == 0 || episodes), "episodes buffer missing");
    mi_env* e = (mi_env*)handle;
    hipStream_t s = (hipStream_t)stream;
    const bool forced = forced_actions || forced_uniforms || forced_resets, eplog = max_ep > 0 && episode_stats;
    const dim3 grid((e->n + RQ_ENVS - 1) / RQ_ENVS);
    const dim3 block(128);
    // episode statistics: episode_stats == NULL keeps them per workgroup in the handle (mi_env_episode_stats sums them on request: no atomics, no launch); with a
    // buffer, launches of >= MI_STATS_PART_MIN workgroups do the same and sum them into it right behind the launch, smaller ones (and the episode log) use atomics
    const bool part = !eplog && (!episode_stats || (int)grid.x >= MI_STATS_PART_MIN) && (int)grid.x <= e->stats_cap;
    // atomics: statistics double-buffered by the caller (stats_next): episode_stats is zero on entry and this launch zeroes the other buffer — no reset launch
    if (episode_stats && !part && !stats_next) { zero_i32x4_kernel<<<1, 64, 0, s>>>(episode_stats); MI_LAUNCH_CHECK(); }
    {
        mi_prof_scope prof(MI_PROF_ROLLOUT, s);
#define RQ_LAUNCH(F, L) rollout_q4_kernel<F, L><<<grid, block, 0, s>>>(*e, params, T, obs_cur, observations, values, actions, log_probs, rewards, dones, \
                                                                     forced_actions, forced_uniforms, forced_resets, episodes, part ? nullptr : episode_stats, max_ep, \
                                                                     advantages, returns, gamma, lam, zero_f64, zero_n, stats_next, part ? e->stats_part : nullptr)
        if (forced) { if (eplog) RQ_LAUNCH(true, true); else RQ_LAUNCH(true, false); }
        else { if (eplog) RQ_LAUNCH(false, true); else RQ_LAUNCH(false, false); }
#undef RQ_LAUNCH
        MI_LAUNCH_CHECK();
    }
    if (part) { e->stats_n = (int)grid.x; if (episode_stats) return mi_env_stats_reduce(e, episode_stats, s); }
    else if (!episode_stats) e->stats_n = 0;
    return MI_OK;
}

extern "C" int mi_ppo_rollout(void* handle, const float* params, int T, float* obs_cur, float* observations, float* values,
                              int64_t* actions, float* log_probs, float* rewards, float* dones, const int64_t* forced_actions,
                              const float* forced_uniforms, const double* forced_resets, mi_episode_t* episodes,
                              int32_t* episode_stats, int max_ep, void* stream) {
    return rollout_impl(handle, params, T, obs_cur, observations, values, actions, log_probs, rewards, dones, forced_actions, forced_uniforms, forced_resets,
                        episodes, episode_stats, max_ep, nullptr, nullptr, 0.0f, 0.0f, nullptr, 0, nullptr, stream);
}

// rollout + GAE (ppo.py:110-151): the rollout workgroups scan their own envs at the end of the launch (T <= RQ_GAE_T; otherwise mi_gae runs
// as a launch of its own).  advantages / returns: dev f32 [T+1, N], bit-identical to mi_gae's.
// internal (mi_ppo_update): additionally zero-fills an fp64 scratch the following launches accumulate into (saves the memset launch)
int mi_rollout_gae_internal(void* handle, const float* params, int T, float* obs_cur, float* observations, float* values,
                            int64_t* actions, float* log_probs, float* rewards, float* dones, mi_episode_t* episodes,
                            int32_t* episode_stats, int max_ep, float gamma, float gae_lambda, float* advantages, float* returns, double* zero_f64, int zero_n,
                            int32_t* stats_next, void* stream) {
    MI_CHECK_ARG(advantages && returns, "NULL advantages / returns");
    if (T <= RQ_GAE_T)
        return rollout_impl(handle, params, T, obs_cur, observations, values, actions, log_probs, rewards, dones, nullptr, nullptr, nullptr, episodes,
                            episode_stats, max_ep, advantages, returns, gamma, gae_lambda, zero_f64, zero_n, stats_next, stream);
    const int rc = rollout_impl(handle, params, T, obs_cur, observations, values, actions, log_probs, rewards, dones, nullptr, nullptr, nullptr, episodes,
                                episode_stats, max_ep, nullptr, nullptr, 0.0f, 0.0f, zero_f64, zero_n, stats_next, stream);
    if (rc) return rc;
    return mi_gae(rewards, dones, values, T, ((mi_env*)handle)->n, gamma, gae_lambda, advantages, returns, stream);
}

extern "C" int mi_ppo_rollout_gae(void* handle, const float* params, int T, float* obs_cur, float* observations, float* values,
                                  int64_t* actions, float* log_probs, float* rewards, float* dones, mi_episode_t* episodes,
                                  int32_t* episode_stats, int max_ep, float gamma, float gae_lambda, float* advantages, float* returns, void* stream) {
    return mi_rollout_gae_internal(handle, params, T, obs_cur, observations, values, actions, log_probs, rewards, dones, episodes, episode_stats, max_ep, gamma,
                                   gae_lambda, advantages, returns, nullptr, 0, nullptr, stream);
}

// ---- ActorCritic forward on an arbitrary batch (agent.get_value / get_action_distribution, ppo.py:49-54) ------
// One lane per row, all weights in LDS (every lane reads the same address -> broadcast).  Not on the training
// hot path (the rollout and update kernels fuse their own forwards); this backs the nn.Module-style API.
__global__ void __launch_bounds__(256) forward_kernel(const float* __restrict__ params, const float* __restrict__ obs, int n,
                                                      float* __restrict__ logits, float* __restrict__ value) {
    __shared__ __attribute__((aligned(16))) float w[NPARAMS + 5];
    for (int i = threadIdx.x; i < NPARAMS; i += blockDim.x) w[i] = params[i];
    __syncthreads();
    for (int row = blockIdx.x * blockDim.x + threadIdx.x; row < n; row += gridDim.x * blockDim.x) {
        const float4 o = reinterpret_cast<const float4*>(obs)[row];
        const float x[4] = {o.x, o.y, o.z, o.w};
#pragma unroll 1
        for (int net = 0; net < 2; ++net) {
            if (net == 0 ? !logits : !value) continue;
            const float* p = w + (net ? C_BASE : 0);
            float h1[HID];
#pragma unroll
            for (int j = 0; j < HID; ++j) {
                float z = p[N_B1 + j];
#pragma unroll
                for (int k = 0; k < OBS; ++k) z = __builtin_fmaf(p[N_W1 + 4 * j + k], x[k], z);
                h1[j] = mi_tanhf(z);
            }
            float o0 = 0.0f, o1 = 0.0f;
#pragma unroll 4
            for (int j = 0; j < HID; ++j) {
                float a0 = 0.0f, a1 = 0.0f;
#pragma unroll
                for (int k = 0; k < HID; k += 2) {
                    a0 = __builtin_fmaf(p[N_W2 + HID * j + k], h1[k], a0);
                    a1 = __builtin_fmaf(p[N_W2 + HID * j + k + 1], h1[k + 1], a1);
                }
                const float h2 = mi_tanhf((a0 + a1) + p[N_B2 + j]);
                o0 = __builtin_fmaf(p[N_W3 + j], h2, o0);
                if (net == 0) o1 = __builtin_fmaf(p[N_W3 + HID + j], h2, o1);
            }
            if (net == 0) {
                logits[2 * (size_t)row] = o0 + p[N_W3 + 2 * HID];
                logits[2 * (size_t)row + 1] = o1 + p[N_W3 + 2 * HID + 1];
            } else {
                value[row] = o0 + p[N_W3 + HID];
            }
        }
    }
}

extern "C" int mi_ppo_forward(const float* params, const float* obs, int n, float* logits, float* value, void* stream) {
    MI_CHECK_ARG(params && obs, "NULL pointer");
    MI_CHECK_ARG(n >= 0, "n must be >= 0");
    if (n == 0 || (!logits && !value)) return MI_OK;
    int blocks = (n + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    forward_kernel<<<blocks, 256, 0, (hipStream_t)stream>>>(params, obs, n, logits, value);
    MI_LAUNCH_CHECK();
    return MI_OK;
}

MI_INSIDE_EXPORT(rollout)
