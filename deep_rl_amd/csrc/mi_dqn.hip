// mi_dqn.hip — the DQN hot path of reference deep_rl/dqn.py on the device:
//   dqn_forward_kernel   QNetwork.forward 4->120->84->2 ReLU                                   (dqn.py:24-36)
//   dqn_act4_kernel      n_steps x {epsilon-greedy, env.step + auto-reset, ring store} per env (dqn.py:84-108)
//   dqn_sample_kernel    batch_inds = randint(upper, size=batch)                               (dqn.py:116)
//   dqn_td_kernel        gather, target max, TD target, MSE loss, backward                      (dqn.py:118-128)
//   dqn_reduce_kernel    fixed-order sum of the per-workgroup partial gradients
// Acting runs on the f32 MFMA (16 envs per workgroup, weights register-resident, 120 / 84 units zero-padded to 128 / 96); the TD
// update's batches are small (128 rows in the reference), so it is VALU + LDS: parameters (43.7 KB) stay L2-resident, activations of
// a row group live in LDS.
#include "mi_common.h"

#define DQ_H1 120
#define DQ_H2 84
#define DQ_W1 0
#define DQ_B1 480
#define DQ_W2 600
#define DQ_B2 10680
#define DQ_W3 10764
#define DQ_B3 10932
#define DQ_NP 10934
#define STREAM_EXPLORE 3u
#define STREAM_SAMPLE 4u

// ---- one row through the net, weights from LDS (every lane reads the same address: broadcast) ----------------------
__device__ __forceinline__ void dqn_row_forward(const float* __restrict__ w, const float x[4], float q[2]) {
    float h1[DQ_H1];
#pragma unroll
    for (int j = 0; j < DQ_H1; ++j) {
        float z = w[DQ_B1 + j];
#pragma unroll
        for (int k = 0; k < OBS; ++k) z = __builtin_fmaf(w[DQ_W1 + 4 * j + k], x[k], z);
        h1[j] = fmaxf(z, 0.0f);
        // h1 must stay a register array (constant indices => full unroll), but hipcc would then hoist all 600 LDS loads of
        // the layer to the top (600 live VGPRs, spills).  Fence the scheduler every 8 units.
        if ((j & 7) == 7) __builtin_amdgcn_sched_barrier(0);
    }
    float q0 = w[DQ_B3], q1 = w[DQ_B3 + 1];
#pragma unroll 1
    for (int j = 0; j < DQ_H2; ++j) {
        float a0 = 0.0f, a1 = 0.0f;
        const float4* wr = reinterpret_cast<const float4*>(w + DQ_W2 + DQ_H1 * j);  // 480-byte rows: 16-byte aligned
#pragma unroll
        for (int k4 = 0; k4 < DQ_H1 / 4; ++k4) {
            const float4 ww = wr[k4];
            a0 = __builtin_fmaf(ww.x, h1[4 * k4], a0); a1 = __builtin_fmaf(ww.y, h1[4 * k4 + 1], a1);
            a0 = __builtin_fmaf(ww.z, h1[4 * k4 + 2], a0); a1 = __builtin_fmaf(ww.w, h1[4 * k4 + 3], a1);
        }
        const float h2 = fmaxf((a0 + a1) + w[DQ_B2 + j], 0.0f);
        q0 = __builtin_fmaf(w[DQ_W3 + j], h2, q0);
        q1 = __builtin_fmaf(w[DQ_W3 + DQ_H2 + j], h2, q1);
    }
    q[0] = q0; q[1] = q1;
}

__device__ __forceinline__ void stage_params(float* __restrict__ w, const float* __restrict__ params) {
    for (int i = threadIdx.x; i < DQ_NP; i += blockDim.x) w[i] = params[i];
    __syncthreads();
}

__global__ void __launch_bounds__(256) dqn_forward_kernel(const float* __restrict__ params, const float* __restrict__ obs, int n, float* __restrict__ q) {
    __shared__ __attribute__((aligned(16))) float w[DQ_NP + 2];
    stage_params(w, params);
    for (int row = blockIdx.x * blockDim.x + threadIdx.x; row < n; row += gridDim.x * blockDim.x) {
        const float4 o = reinterpret_cast<const float4*>(obs)[row];
        const float x[4] = {o.x, o.y, o.z, o.w};
        float qq[2];
        dqn_row_forward(w, x, qq);
        q[2 * (size_t)row] = qq[0]; q[2 * (size_t)row + 1] = qq[1];
    }
}

extern "C" int mi_dqn_forward(const float* params, const float* obs, int n, float* q, void* stream) {
    MI_CHECK_ARG(params && obs && q, "NULL pointer");
    MI_CHECK_ARG(n >= 0, "n must be >= 0");
    if (n == 0) return MI_OK;
    int blocks = (n + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    dqn_forward_kernel<<<blocks, 256, 0, (hipStream_t)stream>>>(params, obs, n, q);
    MI_LAUNCH_CHECK();
    return MI_OK;
}

#define DQN_MAX_STEPS_PER_CALL 64
struct dqn_eps_tab { float v[DQN_MAX_STEPS_PER_CALL]; };  // epsilon(global_step + k), passed by value

// =====================================================================================================================================
// Prioritized replay (reference per.py): the pieces of its bookkeeping that RIDE on launches of the DQN path (round 6; the kernels they came from and the
// sampler's contract are in the "Prioritized replay" block further down).  One optimisation step used to be six launches sequenced from Python — act, marks,
// sampler, TD, slab sum + Adam, scatter + sums: 67.6 us per iteration, 29.5 us of it bookkeeping in three launches of their own.  Now
//   acting launch   + MARK workgroups: priorities[new rows] = max_priority (per.py:105), the write head, the touched level-1 groups' sums — max_priority only changes
//                     at an update (per.py:145), i.e. at a kernel boundary in front of this launch;
//                   + one OWED-SUMS workgroup: the chunk sums of the entries the LAST update re-prioritised (their scatter ran in that update's slab-sum launch);
//                     nothing reads a sum before the next sampler, which sits behind this launch's boundary;
//   sampler launch  unchanged (per_sample_kernel);
//   TD launch       unchanged (dqn_td_kernel);
//   slab sum + Adam + one SCATTER workgroup: "last duplicate wins" (atomicMax owner scheme), priorities[batch_inds] = |td|, max_priority (per.py:144-145) — |td| is final
//                     at the TD kernel's boundary and the slab sum does not read priorities.
// Every role executes the expression sequence of the stand-alone kernel it replaces (recompute, never adjust): indices, weights, priorities, sums and parameters are
// bit-identical to the six-launch sequence (tests/test_gpu_per.py).
#define PER_CHUNK 64
#define STREAM_PER 7u
struct per_ws_t { double* s0; double* a0; double* s1; double* a1; double* totals; };   // a0 / a1: the same sums of p^alpha; totals = {sum p, sum p^alpha}
struct per_mark_t { long long a[2], b[2]; int nb0; };   // the touched flat ranges [a, b) (the second one after the ring wrap) and piece 0's workgroup count
// p^alpha of a priority (per.py:131): 2^(alpha log2 p) on the hardware log2 / exp2 (3 instructions instead of ~150 for powf; relative
// error ~1e-6, it only enters the importance weights, which are compared to 2e-5).  p = 0 -> 0 for EVERY alpha, explicitly: never-written entries and
// the ring's write head must contribute +0 to the incremental sums, and alpha = 0 (uniform PER, legitimate in per.py) would otherwise give 0 * -inf = NaN.
// (torch's 0 ** 0 = 1 would add the count of never-written entries to sum p^alpha at alpha = 0; the weights of per.py:145-146 are normalised by their
// maximum and every sampled entry has p > 0, so they are exactly 1 either way.  the CPU oracle makes the same choice.)
__device__ __forceinline__ float per_pow(float p, float alpha) { return p == 0.0f ? 0.0f : __builtin_amdgcn_exp2f(alpha * __builtin_amdgcn_logf(p)); }
__host__ __device__ inline int64_t per_n0(int64_t n) { return (n + PER_CHUNK - 1) / PER_CHUNK; }

struct per_ride_t {            // what the acting launch carries for PER (all zero: nothing)
    float* prio; const float* max_prio; double *s0, *a0, *s1, *a1;
    per_mark_t mk; long long capacity, slots, gs; float alpha; int N, n_steps;
    int n_mark;                // MARK workgroups = level-1 groups the acting call's rows reach
    const int64_t* owed_idx; int owed_batch;   // OWED-SUMS workgroups (PER_OWED_WGS of them, behind the mark workgroups): the last update's batch indices; nullptr: nothing owed
};
struct per_scatter_t { float* prio; const int64_t* idx; const float* td_abs; int batch; int32_t* owner; float* max_prio; };   // the slab-sum launch's SCATTER workgroup

__host__ __device__ inline bool per_group_marked(const per_mark_t& mk, long long grp) {   // is level-1 group `grp` one a MARK workgroup of this launch rebuilds?
    constexpr long long G = (long long)PER_CHUNK * PER_CHUNK;
    for (int p = 0; p < 2; ++p) if (mk.b[p] > mk.a[p] && grp >= mk.a[p] / G && grp <= (mk.b[p] - 1) / G) return true;
    return false;
}

struct per_mark_smem { float pv[PER_CHUNK][PER_CHUNK + 1], pa[PER_CHUNK][PER_CHUNK + 1]; double cs[PER_CHUNK], ca[PER_CHUNK]; };
// MARK role, NT threads (a multiple of 128 dividing 4,096): one workgroup per level-1 group (64 chunks = 4,096 entries) that the touched rows reach — writes the marks,
// recomputes the group's 64 chunk sums and, owning the whole group, its level-1 sums (per_sums0_kernel's / per_sums1_kernel's loops, in their order).  The thread's
// 4,096 / NT priorities are requested TOGETHER, unconditionally (a `for (e = tid; e < 4096; e += blockDim.x)` loop with its load inside is one memory round trip per
// iteration: the first riding form of this role took 24 us for sixteen of them).
template <int NT>
__device__ __forceinline__ void per_mark_role(per_mark_smem& sm, int wg, const per_ride_t& r) {
    constexpr int IT = PER_CHUNK * PER_CHUNK / NT;
    const per_mark_t& mk = r.mk;
    const int piece = wg < mk.nb0 ? 0 : 1;
    const long long m = mk.a[piece] / (PER_CHUNK * PER_CHUNK) + (wg - (piece ? mk.nb0 : 0));   // this workgroup's level-1 group
    const long long base = m * PER_CHUNK * PER_CHUNK, head = r.gs % r.slots;
    const long long row0 = base / r.N;                 // 64-bit divisions once per workgroup; per entry only 32-bit arithmetic
    const int off0 = (int)(base - row0 * r.N);
    const float mp = r.max_prio[0];
    float pr[IT];
#pragma unroll
    for (int it = 0; it < IT; ++it) {
        const long long i = base + it * NT + (int)threadIdx.x;
        pr[it] = r.prio[i < r.capacity ? i : base];   // (base < capacity: the group exists)
    }
#pragma unroll
    for (int it = 0; it < IT; ++it) {
        const int e = it * NT + (int)threadIdx.x;
        const long long i = base + e;
        float p = 0.0f;
        if (i < r.capacity) {
            long long st = row0 + (unsigned)(off0 + e) / (unsigned)r.N - head;   // time steps after the first one written by the acting call
            st = st < 0 ? st + r.slots : st;
            if (st < r.n_steps) { p = mp; r.prio[i] = p; }                   // per.py:105
            else if (st == r.n_steps) { p = 0.0f; r.prio[i] = p; }           // the ring's write head: never sampled
            else p = pr[it];
        }
        sm.pv[e >> 6][e & 63] = p; sm.pa[e >> 6][e & 63] = per_pow(p, r.alpha);
    }
    __syncthreads();
    if (threadIdx.x < 2 * PER_CHUNK) {               // threads 0..63: chunk sums of p, 64..127: of p^alpha — per_sums0_kernel's loop
        const int q = threadIdx.x & 63;
        const float (*src)[PER_CHUNK + 1] = threadIdx.x < PER_CHUNK ? sm.pv : sm.pa;
        double sum = 0.0;
        for (int j = 0; j < PER_CHUNK; ++j) sum += (double)src[q][j];
        const long long k = m * PER_CHUNK + q;
        if (threadIdx.x < PER_CHUNK) { sm.cs[q] = sum; if (k * PER_CHUNK < r.capacity) r.s0[k] = sum; }
        else { sm.ca[q] = sum; if (k * PER_CHUNK < r.capacity) r.a0[k] = sum; }
    }
    __syncthreads();
    if (threadIdx.x == 0 || threadIdx.x == 64) {     // the group's level-1 sums (per_sums1_kernel's order; chunks beyond the ring are +0.0)
        const double* src = threadIdx.x == 0 ? sm.cs : sm.ca;
        double sum = 0.0;
        for (int j = 0; j < PER_CHUNK; ++j) sum += src[j];
        (threadIdx.x == 0 ? r.s1 : r.a1)[m] = sum;
    }
}

// OWED-SUMS role, NT threads, ROWS batch rows per pass through an LDS block of 2 * ROWS * 65 doubles: the level-0 sums of every chunk the last update's scatter touched,
// then the level-1 sums of their groups — per_scatter_sums_kernel's phases, its loops in their order.  Workgroup `part` of `nparts` takes the batch rows whose LEVEL-1
// GROUP is congruent to it (a group's chunk sums and its level-1 sum then have one writer, and phase 2 reads what phase 1 of the SAME workgroup wrote); with nparts > 1
// the rows are first compacted into `list` (LDS, >= batch ints; the order is whatever the LDS counter gives — every row's result depends on its own chunk only).
// `mk` (nullable): level-1 groups that MARK workgroups of the SAME launch rebuild from the (final) priorities are left to them — same values, one writer.
// (First riding form, one workgroup walking the batch 32 rows at a time with its loads inside runtime-strided loops: ~38 us, longer than the acting loop it was to hide
// behind — the PER iteration went from 69 to 86 us.  Now: eight workgroups, one pass each at the reference's batch, every pass two memory round trips deep — the rows'
// base indices first, then the thread's ROWS * 64 / NT values together.)
template <int NT, int ROWS>
__device__ __forceinline__ void per_owed_sums_role(unsigned char* raw, int* list, int part, int nparts, const float* __restrict__ prio, const int64_t* __restrict__ idx,
                                                   int batch, long long capacity, float alpha, double* __restrict__ s0, double* __restrict__ a0, double* __restrict__ s1,
                                                   double* __restrict__ a1, const per_mark_t* mk) {
    constexpr int IT = ROWS * PER_CHUNK / NT;
    static_assert(ROWS * PER_CHUNK % NT == 0 && 2 * ROWS <= NT, "thread map of the owed-sums role");
    __shared__ long long cb[ROWS];   // per row of the pass: the first flat index of its chunk (phase 1) / the first level-0 index of its group (phase 2); -1: no row
    __shared__ int cnt;
    float (*spv)[PER_CHUNK + 1] = reinterpret_cast<float (*)[PER_CHUNK + 1]>(raw);
    float (*spa)[PER_CHUNK + 1] = spv + ROWS;
    double (*ts)[PER_CHUNK + 1] = reinterpret_cast<double (*)[PER_CHUNK + 1]>(raw);
    double (*ta)[PER_CHUNK + 1] = ts + ROWS;
    constexpr long long G = (long long)PER_CHUNK * PER_CHUNK;
    auto live = [&](int b) { return b < batch && !(mk && per_group_marked(*mk, idx[b] / G)); };
    int n = batch;
    if (list) {   // compact this workgroup's rows
        if (threadIdx.x == 0) cnt = 0;
        __syncthreads();
        for (int b = threadIdx.x; b < batch; b += NT)
            if ((int)((idx[b] / G) % nparts) == part && live(b)) list[atomicAdd(&cnt, 1)] = b;
        __syncthreads();
        n = cnt;
    }
    auto row = [&](int k) { return list ? (k < n ? list[k] : -1) : (live(k) ? k : -1); };   // batch row of position k, -1: none
    for (int b0 = 0; b0 < n; b0 += ROWS) {
        __syncthreads();
        if (threadIdx.x < ROWS) { const int b = row(b0 + (int)threadIdx.x); cb[threadIdx.x] = b >= 0 ? (idx[b] / PER_CHUNK) * PER_CHUNK : -1; }
        __syncthreads();
        float pr[IT];
#pragma unroll
        for (int it = 0; it < IT; ++it) {
            const int e = it * NT + (int)threadIdx.x;
            const long long i = cb[e >> 6] + (e & 63);
            pr[it] = prio[cb[e >> 6] >= 0 && i < capacity ? i : 0];
        }
#pragma unroll
        for (int it = 0; it < IT; ++it) {
            const int e = it * NT + (int)threadIdx.x, q = e >> 6, j = e & 63;
            const float p = cb[q] >= 0 && cb[q] + j < capacity ? pr[it] : 0.0f;
            spv[q][j] = p; spa[q][j] = per_pow(p, alpha);
        }
        __syncthreads();
        if (threadIdx.x < ROWS && cb[threadIdx.x] >= 0) {
            double sum = 0.0, sa = 0.0;
            for (int j = 0; j < PER_CHUNK; ++j) { sum += (double)spv[threadIdx.x][j]; sa += (double)spa[threadIdx.x][j]; }
            const long long k = cb[threadIdx.x] / PER_CHUNK;
            s0[k] = sum; a0[k] = sa;   // (a chunk hit by several batch rows is recomputed by several threads: same inputs, same value)
        }
    }
    const long long n0c = per_n0(capacity);
    for (int b0 = 0; b0 < n; b0 += ROWS) {
        __syncthreads();   // (first pass: also orders this workgroup's level-0 stores before the loads below)
        if (threadIdx.x < ROWS) { const int b = row(b0 + (int)threadIdx.x); cb[threadIdx.x] = b >= 0 ? (idx[b] / G) * PER_CHUNK : -1; }
        __syncthreads();
        double vs[IT], va[IT];
#pragma unroll
        for (int it = 0; it < IT; ++it) {
            const int e = it * NT + (int)threadIdx.x;
            const long long k = cb[e >> 6] + (e & 63);
            const bool in = cb[e >> 6] >= 0 && k < n0c;
            vs[it] = s0[in ? k : 0]; va[it] = a0[in ? k : 0];
        }
#pragma unroll
        for (int it = 0; it < IT; ++it) {
            const int e = it * NT + (int)threadIdx.x, q = e >> 6, j = e & 63;
            const bool in = cb[q] >= 0 && cb[q] + j < n0c;
            ts[q][j] = in ? vs[it] : 0.0; ta[q][j] = in ? va[it] : 0.0;
        }
        __syncthreads();
        if (threadIdx.x < 2 * ROWS) {
            const int q = threadIdx.x % ROWS;
            if (cb[q] >= 0) {
                const double (*src)[PER_CHUNK + 1] = (int)threadIdx.x < ROWS ? ts : ta;
                double sum = 0.0;
                for (int j = 0; j < PER_CHUNK; ++j) sum += src[q][j];
                ((int)threadIdx.x < ROWS ? s1 : a1)[cb[q] / PER_CHUNK] = sum;
            }
        }
    }
}

// SCATTER role (ONE workgroup): priorities[idx[b]] = |td_b| with the LAST occurrence of a duplicated index winning (per.py:144 on the host is sequential), and
// max_priority = max(max_priority, surviving |td|) — every entry is <= the running max_priority at all times, so this equals max(torch.max(priorities), max_priority)
// (per.py:145) without a pass over the buffer.  owner: int32 per ring entry, all -1 between calls.  (per_scatter_kernel's body; wmax: >= blockDim.x / 64 floats of LDS.)
__device__ __forceinline__ void per_scatter_role(float* wmax, const per_scatter_t& sc) {
    const int nt = (int)blockDim.x;
    for (int b = threadIdx.x; b < sc.batch; b += nt) atomicMax(&sc.owner[sc.idx[b]], b);
    __syncthreads();
    float mx = 0.0f;
    for (int b = threadIdx.x; b < sc.batch; b += nt)
        if (sc.owner[sc.idx[b]] == b) { sc.prio[sc.idx[b]] = sc.td_abs[b]; mx = fmaxf(mx, sc.td_abs[b]); }
    __syncthreads();
    for (int b = threadIdx.x; b < sc.batch; b += nt) sc.owner[sc.idx[b]] = -1;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = mx;
    __syncthreads();
    if (threadIdx.x == 0) {
        float m = sc.max_prio[0];
        for (int k = 0; k < (nt >> 6); ++k) m = fmaxf(m, wmax[k]);
        sc.max_prio[0] = m;
    }
}
#define PER_RIDE_ROWS 32      // batch rows per pass of an owed-sums workgroup inside the acting launch: 2 x 32 x 65 doubles = 33 KB of the acting kernel's 46 KB LDS block
#define PER_OWED_WGS 8        // owed-sums workgroups of the acting launch (level-1 group % 8): ~16 rows each at the reference's batch of 128
#define PER_OWED_LIST 2048    // ... and their row lists (8 KB of the same block): larger batches settle their sums in a launch of their own


// ---- acting: n_steps iterations of {epsilon-greedy, env.step + auto-reset, ring store} in one launch (the online net is frozen
// between two updates).  A workgroup owns 16 envs: THREE forward waves run the Q-network on v_mfma_f32_16x16x4_f32 in the [unit][env] orientation
// with EVERY weight resident in registers as an A operand for the whole launch (forward wave w owns output tiles 2w, 2w+1 of layer 2):
//   layer 1  D1[unit][env] = W1[unit][0..3] . obs[env][0..3]            one k-step per 16-unit tile (8 tiles, 120 units padded to 128)
//   layer 2  D2[out][env] += W2[out][u] * h1[u][env]                    the layer-1 accumulators ARE the B operands: accumulator
//            register r of tile t in lane group g is unit 16 t + 4 g + r, so k-step (t, r) contracts over g with A = W2[out][16 t + 4 g + r]
//   layer 3  two dot products over the wave's 32 units, summed across lane groups (2 shuffles) and across the 3 waves (LDS, one barrier)
// and a FOURTH wave steps the fp64 dynamics (below).  72 MFMAs per forward wave and step.  (The 3-wave form of rounds 1 - 3, in which every forward wave carried
// the dynamics redundantly, lost the A/B in round 4 and left the source in round 5: docs/LEDGER.md.)
#define DA_ENVS 16
typedef float dq_f32x4 __attribute__((ext_vector_type(4)));
typedef dq_f32x4 f32x4_t;
#define DQ_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)
// FORCED (parity mode: forced_actions / forced_resets may be given) and EPLOG (per-episode list kept) are separate instantiations, not
// run-time branches: a conditional global load or returning atomic inside the step loop makes the compiler place s_waitcnt vmcnt(0) at
// the join, executed on EVERY step, where it waits for all of the wave's outstanding ring stores (mi_rollout.hip, rollout_q4_kernel).
// CartPole has two actions, so the dynamics wave computes BOTH successors of the current state (lane groups 0 / 1 take action 0 / 1, groups 2 / 3 idle
// replicas) while the three forward waves evaluate the Q-network on it, and publishes them as records in LDS; after the step's one barrier every wave
// derives the action from the partial sums and picks the chosen record — the forward waves only its observation.  The dynamics wave also draws the
// exploration words one step ahead, keeps the episode bookkeeping and writes the ring.  Same keyed draws, same IEEE sequence per env: bit-identical
// trajectories (tests/test_gpu_dqn.py runs unchanged).  Round 4 (tools/dqn_act_stamps.py, -DDA_STAMPS): a step is 4,100 cycles at 2.23 GHz = 1.85 us, of which the forward
// waves' chain is 3,450 (72 MFMAs = 2,304) + 170 at the barrier + 480 for the action and the chosen record; the dynamics wave waits 40 % of the step.  The launch is
// 30.6 us by rocprof for an 18.5 us loop: 2.3 us of weight prologue, <= 0.5 us of start skew between XCDs, and the rest around the first entry and the last exit.
// Nontemporal ring stores: +0.9 us per iteration; write-through (sc0 sc1) ring stores: -0.2 us, inside the noise — neither kept.
#ifdef DA_STAMPS   // diagnostic build: where a forward wave (0) and the dynamics wave (3) of workgroup DA_STAMP_WG spend a step (s_memtime), tools/dqn_act_stamps.py
#ifndef DA_STAMP_WG
#define DA_STAMP_WG 100
#endif
__device__ unsigned long long da_stamp_dbg[2][8];
__device__ unsigned long long da_mark_dbg[8][2][4][256];   // [forward wave 0 | dynamics wave][kernel entry, loop start, loop end, exit][workgroup]: s_memrealtime (100 MHz)
#define DA_MARK(k) do { if (lane == 0 && (w == 0 || w == 3) && blockIdx.x < 256) { unsigned long long rt_; asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(rt_) :: "memory"); \
                        da_mark_dbg[(global_step / n_steps) & 7][w == 3][k][blockIdx.x] = rt_; } } while (0)
extern "C" int mi_debug_dqn_act_marks(unsigned long long* out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(da_mark_dbg), sizeof(unsigned long long) * 8 * 2 * 4 * 256) == hipSuccess ? 0 : -2;
}
#define DA_STAMP(k) do { __builtin_amdgcn_sched_barrier(0); { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); \
                         da_acc[k] += t_ - da_last; da_last = t_; } __builtin_amdgcn_sched_barrier(0); } while (0)
extern "C" int mi_debug_dqn_act_stamps(unsigned long long* out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(da_stamp_dbg), sizeof(unsigned long long) * 16) == hipSuccess ? 0 : -2;
}
__device__ unsigned long long da_tiny_dbg[64][2];
__global__ void da_tiny_kernel(int i) {   // a one-wave launch that only marks its entry and exit: put between two acting launches it splits their boundary into an end and a start
    unsigned long long t0, t1;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
    if (threadIdx.x == 0) { da_tiny_dbg[i & 63][0] = t0; da_tiny_dbg[i & 63][1] = t1; }
}
extern "C" int mi_debug_tiny_mark(int i, void* stream) { da_tiny_kernel<<<1, 64, 0, (hipStream_t)stream>>>(i); return hipGetLastError() == hipSuccess ? 0 : -2; }
extern "C" int mi_debug_tiny_read(unsigned long long* out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(da_tiny_dbg), sizeof(unsigned long long) * 128) == hipSuccess ? 0 : -2;
}
#else
#define DA_STAMP(k) do {} while (0)
#define DA_MARK(k) do {} while (0)
#endif
struct __attribute__((aligned(16))) da4_rec {
    float4 ob;                      // successor observation (the reset observation where the step ends the episode)
    double x, xd, th, thd;          // successor state (after the reset where done)
    int elapsed, eplen; float epret; int flags;   // flags: bit 0 terminated, bit 1 done
    int fin_len; float fin_ret; unsigned ep_lo, ep_hi;
};
struct __attribute__((aligned(16))) da4_smem {
    float W2s[DQ_H2 * DQ_H1];       // layer 2 as it lies in memory, staged once per launch by coalesced loads: the forward waves pick their MFMA fragments from here
    da4_rec rec[2][2][DA_ENVS];     // [step parity][action][env]
    float qp[2][3][DA_ENVS][2];     // [step parity][forward wave][env][action]: partial head sums
    int expl[4][DA_ENVS];           // [step & 3][env]: bit 0 explore, bit 1 the random action
};
// (PER: at least 2 waves per SIMD, i.e. <= 256 registers.  A kernel's register allocation holds for ALL its workgroups: at the 268 registers the acting code takes
// when left alone, PER's riding workgroups could not share a SIMD with an acting wave — they ran BEHIND the acting workgroups, a 4.4 us tail on a 21.5 us launch.)
template <bool FORCED, bool EPLOG, bool PER>
__global__ void __launch_bounds__(256, PER ? 2 : 1)
dqn_act4_kernel(mi_env e, const float* __restrict__ params, int n_steps, long long global_step, long long slots, long long learning_starts,
                dqn_eps_tab eps, float* __restrict__ obs_cur, float* __restrict__ observations,
                int64_t* __restrict__ actions, float* __restrict__ rewards, uint8_t* __restrict__ terminated,
                const int64_t* __restrict__ forced_actions, const double* __restrict__ forced_resets, mi_episode_t* __restrict__ episodes,
                int32_t* __restrict__ episode_stats, int max_ep, int32_t* __restrict__ zero_next, int32_t* __restrict__ stats_part, int n_act, per_ride_t ride) {
    __shared__ da4_smem sm;
    MI_INSIDE_SCOPE(MI_PROF_DQN_ACT);
    if constexpr (PER) {
        // workgroups behind the acting ones: PER's bookkeeping (see per_ride_t).  Acting reads the online net and the env and writes the replay ring; the roles read and
        // write priorities and their sums only: no dependency inside the launch — whoever needs the sums (the sampler) sits behind the kernel boundary.
        if ((int)blockIdx.x >= n_act) {
            constexpr size_t IMG = 2 * PER_RIDE_ROWS * (PER_CHUNK + 1) * sizeof(double);
            static_assert(sizeof(da4_smem) >= sizeof(per_mark_smem) && sizeof(da4_smem) >= IMG + PER_OWED_LIST * sizeof(int) && alignof(da4_smem) >= 8,
                          "PER's riding roles reuse the acting kernel's LDS block");
            const int wg = (int)blockIdx.x - n_act;
            if (wg < ride.n_mark) per_mark_role<256>(*reinterpret_cast<per_mark_smem*>(&sm), wg, ride);
            else per_owed_sums_role<256, PER_RIDE_ROWS>(reinterpret_cast<unsigned char*>(&sm), reinterpret_cast<int*>(reinterpret_cast<unsigned char*>(&sm) + IMG), wg - ride.n_mark,
                                                        PER_OWED_WGS, ride.prio, ride.owed_idx, ride.owed_batch, ride.capacity, ride.alpha, ride.s0, ride.a0, ride.s1, ride.a1,
                                                        &ride.mk);
            return;
        }
        // the riders share CUs (and SIMDs) with acting workgroups, whose step is one wave's instruction issue: the acting waves take the issue port first (priority 3
        // against the riders' default 0) — the riders have the whole launch to finish in (round 6: 25.9 -> see docs/LEDGER.md)
        __builtin_amdgcn_s_setprio(3);
    }
    if (zero_next && blockIdx.x == 0 && threadIdx.x < 4) zero_next[threadIdx.x] = 0;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, j = lane & 15, lg = lane >> 4;
    const bool phys = w == 3;
    DA_MARK(0);
    const int N = e.n;
    const int i = blockIdx.x * DA_ENVS + j;
    const bool mine = i < N;
    const int g = mine ? i : N - 1;
    const int la = lg & 1;                                  // dynamics wave: the action this lane group speculates on
    const bool writer = phys && mine && lg == 0;
    // ---- prologue: ONE memory latency deep (round 5, the lesson of dqn_td_kernel's stamps: a guarded load is a basic block that ends in s_waitcnt vmcnt(0), and a
    //      float4 requested in MFMA-fragment layout — sixteen 64-byte segments 480 bytes apart — costs a CU's address unit ~40 cycles per instruction and wave).
    //      Every request is unconditional and the same in all four waves (the dynamics wave drops the operands, the forward waves the env state): the observation
    //      first, W2 as it lies in memory through LDS (10 fully coalesced float4 per thread), the thin operands, the env state; selects and LDS stores behind the last
    //      request.  Until then: weights (forward waves, one branch) -> wait -> env state (dynamics wave, another branch) -> observation -> wait: 2.3 us. ----
    float4 ob = reinterpret_cast<const float4*>(obs_cur)[g];
    float b30 = params[DQ_B3], b31 = params[DQ_B3 + 1];   // every wave derives the action
    constexpr int W2V = DQ_H2 * DQ_H1 / 4, STG = (W2V + 255) / 256;
    dq_f32x4 stg[STG];
#pragma unroll
    for (int q = 0; q < STG; ++q) {
        const int v = (int)threadIdx.x + 256 * q < W2V ? (int)threadIdx.x + 256 * q : W2V - 1;   // (threads past the end repeat the last element)
        stg[q] = reinterpret_cast<const dq_f32x4*>(params + DQ_W2)[v];
    }
    // thin operands of the forward waves.  120 = 30 x 4 and 84 = 21 x 4: a lane's four consecutive units are all inside or all outside.
    float w1a[8];
    dq_f32x4 b1v[8];
    float w2a[2][8][4];
    dq_f32x4 b2v[2], w3v[2][2];
    const int wf = phys ? 0 : w;   // (the dynamics wave requests wave 0's operands and zeroes them)
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        const int ua = 16 * t + j, u0 = 16 * t + 4 * lg;
        w1a[t] = params[DQ_W1 + 4 * (ua < DQ_H1 ? ua : DQ_H1 - 1) + lg];
        b1v[t] = *reinterpret_cast<const dq_f32x4*>(params + DQ_B1 + (u0 < DQ_H1 ? u0 : DQ_H1 - 4));
    }
#pragma unroll
    for (int T = 0; T < 2; ++T) {
        const int o0 = 16 * (2 * wf + T) + 4 * lg, oc = o0 < DQ_H2 ? o0 : DQ_H2 - 4;
        b2v[T] = *reinterpret_cast<const dq_f32x4*>(params + DQ_B2 + oc);
        w3v[T][0] = *reinterpret_cast<const dq_f32x4*>(params + DQ_W3 + oc);
        w3v[T][1] = *reinterpret_cast<const dq_f32x4*>(params + DQ_W3 + DQ_H2 + oc);
    }
    // ---- env state: the dynamics wave owns it (every lane group holds a copy) ----
    double sx = e.x[g], sxd = e.x_dot[g], sth = e.theta[g], sthd = e.theta_dot[g];
    int elapsed = e.elapsed[g], eplen = e.ep_len[g];
    float epret = e.ep_ret[g];
    uint64_t episode = e.episode[g], stepctr0 = e.step_ctr[g];
    // ---- nothing above this line reads a loaded value ----
    const dq_f32x4 z4 = dq_f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        const int ua = 16 * t + j, u0 = 16 * t + 4 * lg;
        if (phys || !(ua < DQ_H1)) w1a[t] = 0.0f;
        if (phys || !(u0 < DQ_H1)) b1v[t] = z4;
    }
#pragma unroll
    for (int T = 0; T < 2; ++T) {
        const int o0 = 16 * (2 * wf + T) + 4 * lg;
        if (phys || !(o0 < DQ_H2)) { b2v[T] = z4; w3v[T][0] = z4; w3v[T][1] = z4; }
    }
    if (!phys) { sx = 0.0; sxd = 0.0; sth = 0.0; sthd = 0.0; elapsed = 0; eplen = 0; epret = 0.0f; episode = 0; stepctr0 = 0; }
#pragma unroll
    for (int q = 0; q < STG; ++q) {
        const int v = (int)threadIdx.x + 256 * q < W2V ? (int)threadIdx.x + 256 * q : W2V - 1;
        reinterpret_cast<dq_f32x4*>(sm.W2s)[v] = stg[q];
    }
    int st_cnt = 0, st_len = 0, st_max = 0;
    auto draw = [&](int s) {   // exploration words of step s -> bit 0 explore, bit 1 random action (dqn.py:86-90)
        uint32_t r[4];
        mi_philox(e.seed, e.env_id_base + (uint64_t)g, stepctr0 + (uint64_t)s, STREAM_EXPLORE, r);
        const float u = mi_u32_to_uniform(r[0]);
        const bool explore = global_step + s < learning_starts || u < eps.v[s];
        return (explore ? 1 : 0) | (int)((r[1] & 1u) << 1);
    };
    if (phys && lg == 0) sm.expl[0][j] = draw(0);
    asm volatile("" : "+v"(ob.x), "+v"(ob.y), "+v"(ob.z), "+v"(ob.w), "+v"(sx), "+v"(sxd), "+v"(sth), "+v"(sthd), "+v"(elapsed), "+v"(eplen), "+v"(epret), "+v"(episode), "+v"(stepctr0), "+v"(b30), "+v"(b31));
    __syncthreads();
#pragma unroll
    for (int T = 0; T < 2; ++T) {   // the layer-2 fragments of this lane (zeros in the dynamics wave and in the padding)
        const int row = 16 * (2 * wf + T) + j, rowc = row < DQ_H2 ? row : DQ_H2 - 1;
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const int col = 16 * t + 4 * lg;
            const dq_f32x4 v = *reinterpret_cast<const dq_f32x4*>(&sm.W2s[DQ_H1 * rowc + (col < DQ_H1 ? col : DQ_H1 - 4)]);
            const bool in = !phys && row < DQ_H2 && col < DQ_H1;
            w2a[T][t][0] = in ? v[0] : 0.0f; w2a[T][t][1] = in ? v[1] : 0.0f; w2a[T][t][2] = in ? v[2] : 0.0f; w2a[T][t][3] = in ? v[3] : 0.0f;
        }
    }
    int a = 0;
    int ex = sm.expl[0][j];   // exploration word of the step about to run; the next step's is fetched together with the head sums behind the barrier (off the forward's chain)
#ifdef DA_STAMPS
    unsigned long long da_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, da_last;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(da_last) :: "memory");
    const unsigned long long da_first = da_last;
    unsigned long long da_rt0;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(da_rt0) :: "memory");
#endif
    // bookkeeping of one committed step (dynamics wave, lane group 0): dqn.py:95, :106-108 and the episode statistics
    long long slot = global_step % slots;   // ring slot of the step being committed; advanced by compare-and-wrap (a 64-bit modulo per step is ~150 scalar instructions)
    auto commit = [&](int s, int act, const da4_rec& rc) {
        const long long nslot = slot + 1 == slots ? 0 : slot + 1;
        sx = rc.x; sxd = rc.xd; sth = rc.th; sthd = rc.thd; elapsed = rc.elapsed; eplen = rc.eplen; epret = rc.epret;
        episode = ((uint64_t)rc.ep_hi << 32) | rc.ep_lo;
        if (writer) {
            actions[slot * N + g] = act;
            reinterpret_cast<float4*>(observations)[nslot * N + g] = rc.ob;
            rewards[nslot * N + g] = 1.0f;
            terminated[nslot * N + g] = (uint8_t)(rc.flags & 1);
            if (rc.flags & 2) {
                st_cnt += 1; st_len += rc.fin_len; st_max = rc.fin_len > st_max ? rc.fin_len : st_max;
                if (EPLOG && max_ep > 0 && episode_stats) {
                    const int sl = atomicAdd(episode_stats + 3, 1);
                    if (sl < max_ep) episodes[sl] = mi_episode_t{g, s, rc.fin_ret, rc.fin_len};
                }
            }
        }
        slot = nslot;
    };
    DA_MARK(1);
    for (int s = 0; s < n_steps; ++s) {
        const int par = s & 1;
        if (!phys) {
            const bool explore = (FORCED && forced_actions) ? true : ((ex & 1) != 0);
            if (__any(!explore)) {  // skip the forward while every env of the group explores (the same envs in all three waves)
                const float b0 = lg == 0 ? ob.x : lg == 1 ? ob.y : lg == 2 ? ob.z : ob.w;
                dq_f32x4 h1[8];
#pragma unroll
                for (int t = 0; t < 8; ++t) {
                    h1[t] = DQ_MFMA(w1a[t], b0, b1v[t]);
#pragma unroll
                    for (int r2 = 0; r2 < 4; ++r2) h1[t][r2] = fmaxf(h1[t][r2], 0.0f);
                }
                dq_f32x4 h2[2] = {b2v[0], b2v[1]};
#pragma unroll
                for (int t = 0; t < 8; ++t)
#pragma unroll
                    for (int r2 = 0; r2 < 4; ++r2) {
                        h2[0] = DQ_MFMA(w2a[0][t][r2], h1[t][r2], h2[0]);
                        h2[1] = DQ_MFMA(w2a[1][t][r2], h1[t][r2], h2[1]);
                    }
                float p0 = 0.0f, p1 = 0.0f;
#pragma unroll
                for (int T = 0; T < 2; ++T)
#pragma unroll
                    for (int r2 = 0; r2 < 4; ++r2) {
                        const float hv = fmaxf(h2[T][r2], 0.0f);
                        p0 = __builtin_fmaf(w3v[T][0][r2], hv, p0); p1 = __builtin_fmaf(w3v[T][1][r2], hv, p1);
                    }
                p0 += __shfl_xor(p0, 16); p0 += __shfl_xor(p0, 32);   // (measured: the VALU-only groups_sum and reading both candidates in the
                p1 += __shfl_xor(p1, 16); p1 += __shfl_xor(p1, 32);   //  post-barrier round trip are each ~1.5 us per launch SLOWER here)
                if (lg == 0) { sm.qp[par][w][j][0] = p0; sm.qp[par][w][j][1] = p1; }
            }
            DA_STAMP(0);
        } else {
            if (s > 0) commit(s - 1, a, sm.rec[par ^ 1][a][j]);
            DA_STAMP(0);
            // both successors of the committed state: this lane group's action
            double nx = sx, nxd = sxd, nth = sth, nthd = sthd;
            int term;
            mi_cartpole_step(nx, nxd, nth, nthd, la, term);
            int nel = elapsed + 1, nlen = eplen + 1;
            float nret = epret + 1.0f;
            uint64_t nep = episode;
            const bool trunc = !term && nel >= CP_MAX_STEPS;
            const bool d = term || trunc;
            const int fin_len = nlen; const float fin_ret = nret;
            if (d) {
                nret = 0.0f; nlen = 0; nel = 0;
                double rs[4];
                if (FORCED && forced_resets) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) rs[k] = forced_resets[4 * ((size_t)s * N + g) + k];
                } else {
                    mi_reset_noise(e.seed, e.env_id_base + (uint64_t)g, episode, rs);
                }
                nep += 1;
                nx = rs[0]; nxd = rs[1]; nth = rs[2]; nthd = rs[3];
            }
            if (lg < 2) {
                da4_rec& rc = sm.rec[par][la][j];
                rc.ob = make_float4((float)nx, (float)nxd, (float)nth, (float)nthd);
                rc.x = nx; rc.xd = nxd; rc.th = nth; rc.thd = nthd;
                rc.elapsed = nel; rc.eplen = nlen; rc.epret = nret; rc.flags = (term ? 1 : 0) | (d ? 2 : 0);
                rc.fin_len = fin_len; rc.fin_ret = fin_ret; rc.ep_lo = (unsigned)nep; rc.ep_hi = (unsigned)(nep >> 32);
            }
            DA_STAMP(1);
            if (lg == 0 && s + 1 < n_steps) sm.expl[(s + 1) & 3][j] = draw(s + 1);
            DA_STAMP(2);
        }
        __syncthreads();
        DA_STAMP(3);
        // every wave: the action of step s (dqn.py:86-92)
        if (FORCED && forced_actions) a = (int)forced_actions[(size_t)s * N + g];
        else {
            const int exn = sm.expl[(s + 1) & 3][j];   // (drawn before this step's barrier; past the last step: never used)
            a = (ex >> 1) & 1;
            if (!(ex & 1)) {
                const float q0 = ((sm.qp[par][0][j][0] + sm.qp[par][1][j][0]) + sm.qp[par][2][j][0]) + b30;
                const float q1 = ((sm.qp[par][0][j][1] + sm.qp[par][1][j][1]) + sm.qp[par][2][j][1]) + b31;
                a = q1 > q0 ? 1 : 0;                               // torch.argmax: first index on ties (dqn.py:92)
            }
            ex = exn;
        }
        if (!phys) ob = sm.rec[par][a][j].ob;
#ifdef DA_STAMPS
        asm volatile("" :: "v"(ob.x), "v"(a));
#endif
        DA_STAMP(4);
    }
    DA_MARK(2);
#ifdef DA_STAMPS
    if (blockIdx.x == DA_STAMP_WG && lane == 0 && (w == 0 || w == 3)) {
        for (int k = 0; k < 5; ++k) da_stamp_dbg[w == 3][k] = da_acc[k];
        unsigned long long da_rt1;
        asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(da_rt1) :: "memory");
        da_stamp_dbg[w == 3][5] = da_last - da_first; da_stamp_dbg[w == 3][6] = (unsigned long long)n_steps; da_stamp_dbg[w == 3][7] = da_rt1 - da_rt0;   // [7]: 100 MHz ticks
    }
#endif
    if (phys) {
        commit(n_steps - 1, a, sm.rec[(n_steps - 1) & 1][a][j]);
        if (writer) {
            e.x[g] = sx; e.x_dot[g] = sxd; e.theta[g] = sth; e.theta_dot[g] = sthd;
            e.elapsed[g] = elapsed; e.ep_ret[g] = epret; e.ep_len[g] = eplen; e.episode[g] = episode; e.step_ctr[g] = stepctr0 + (uint64_t)n_steps;
            reinterpret_cast<float4*>(obs_cur)[g] = make_float4((float)sx, (float)sxd, (float)sth, (float)sthd);
        }
        // one flush per workgroup — and no atomics when the launch has many workgroups: same-address agent-scope atomics are performed one after the other at the
        // memory side, and a launch is not over before the last of them (round 4, tools/dqn_act_stamps.py: 3 x 256 of them kept this launch open for 8.8 us after
        // its last wave had left; the statistics then go to the workgroup's slot in the handle and are summed on request, mi_common.h)
        if (episode_stats || stats_part) {
            int c = writer ? st_cnt : 0, l = writer ? st_len : 0, m = writer ? st_max : 0;
#pragma unroll
            for (int sft = 1; sft < 16; sft <<= 1) { c += __shfl_xor(c, sft); l += __shfl_xor(l, sft); const int mo = __shfl_xor(m, sft); m = mo > m ? mo : m; }
            if (stats_part) { if (lane == 0) reinterpret_cast<int4*>(stats_part)[blockIdx.x] = make_int4(c, l, m, 0); }
            else if (lane == 0 && c > 0) { atomicAdd(episode_stats, c); atomicAdd(episode_stats + 1, l); atomicMax(episode_stats + 2, m); }
        }
    }
    DA_MARK(3);
}

// Negative result (round 2, measured, removed): the PPO rollout's lane = unit formulation on the 16-block 4x4x1 MFMA (4 envs per 2-wave workgroup,
// weights as register-resident B operands) for THIS net: correct and parity-green, but 55 us per 10-step launch against 35 — 84 output units on
// 64-lane waves leave a third of every layer-2 MFMA empty (the 16x16x4 form pads 84 to 96), each wave needs 120 + 8 resident weights (255 VGPRs,
// spills), and the two waves still meet at a barrier per step for the head.

__global__ void dqn_zero_stats_kernel(int32_t* p) { if (threadIdx.x < 4) p[threadIdx.x] = 0; }

static int dqn_act_impl(void* handle, const float* params, int n_steps, int64_t global_step, int64_t slots, int64_t learning_starts,
                        double start_e, double end_e, double exploration_fraction, int64_t total_timesteps, float* obs_cur,
                        float* observations, int64_t* actions, float* rewards, uint8_t* terminated, const int64_t* forced_actions,
                        const double* forced_resets, mi_episode_t* episodes, int32_t* episode_stats, int max_ep, int32_t* zero_next, bool zero_now, void* stream,
                        const per_ride_t* ride = nullptr) {
    MI_CHECK_ARG(handle && params && obs_cur && observations && actions && rewards && terminated, "NULL pointer");
    MI_CHECK_ARG(n_steps > 0 && n_steps <= DQN_MAX_STEPS_PER_CALL, "n_steps must be in [1, 64]");
    MI_CHECK_ARG(slots >= 2 && global_step >= 0, "slots must be >= 2 and global_step >= 0");
    MI_CHECK_ARG(max_ep >= 0 && (max_ep == 0 || episodes), "episodes buffer missing");
    mi_env* e = (mi_env*)handle;
    hipStream_t s = (hipStream_t)stream;
    const bool forced = forced_actions || forced_resets, eplog = max_ep > 0 && episode_stats;
    const int n_wg = (e->n + DA_ENVS - 1) / DA_ENVS;
    // episode statistics: per workgroup in the handle when there is no buffer (mi_env_episode_stats) or when the launch is large (summed into the buffer right behind it)
    const bool part = !eplog && (!episode_stats || n_wg >= MI_STATS_PART_MIN) && n_wg <= e->stats_cap;
    if (episode_stats && !part && zero_now) { dqn_zero_stats_kernel<<<1, 64, 0, s>>>(episode_stats); MI_LAUNCH_CHECK(); }
    // epsilon = max(slope * global_step + start_e, end_e) in the reference's double arithmetic (dqn.py:47,86), evaluated on
    // the host for the n_steps of this call and handed over by value (no allocation, no copy to enqueue)
    dqn_eps_tab tab;
    const double slope = (end_e - start_e) / (exploration_fraction * (double)total_timesteps);
    for (int k = 0; k < DQN_MAX_STEPS_PER_CALL; ++k) {
        const double ev = slope * (double)(global_step + k) + start_e;
        tab.v[k] = (float)(ev > end_e ? ev : end_e);
    }
    {
    mi_prof_scope prof(MI_PROF_DQN_ACT, s);
    const int n_extra = ride ? ride->n_mark + (ride->owed_idx ? PER_OWED_WGS : 0) : 0;   // PER's riding workgroups behind the acting ones
    const dim3 grid(n_wg + n_extra), block(256);
    per_ride_t rd;
    if (ride) rd = *ride; else memset(&rd, 0, sizeof(rd));
#define DA_LAUNCH(F, L, P) dqn_act4_kernel<F, L, P><<<grid, block, 0, s>>>(*e, params, n_steps, (long long)global_step, (long long)slots, (long long)learning_starts, tab, \
                                                                  obs_cur, observations, actions, rewards, terminated, forced_actions, forced_resets, episodes, \
                                                                  part ? nullptr : episode_stats, max_ep, zero_next, part ? e->stats_part : nullptr, n_wg, rd)
    if (n_extra) {
        if (forced) { if (eplog) DA_LAUNCH(true, true, true); else DA_LAUNCH(true, false, true); }
        else { if (eplog) DA_LAUNCH(false, true, true); else DA_LAUNCH(false, false, true); }
    } else {
        if (forced) { if (eplog) DA_LAUNCH(true, true, false); else DA_LAUNCH(true, false, false); }
        else { if (eplog) DA_LAUNCH(false, true, false); else DA_LAUNCH(false, false, false); }
    }
#undef DA_LAUNCH
    MI_LAUNCH_CHECK();
    }
    if (part) { e->stats_n = n_wg; if (episode_stats) return mi_env_stats_reduce(e, episode_stats, s); }
    else if (!episode_stats) e->stats_n = 0;
    return MI_OK;
}

extern "C" int mi_dqn_act_steps(void* handle, const float* params, int n_steps, int64_t global_step, int64_t slots, int64_t learning_starts,
                                double start_e, double end_e, double exploration_fraction, int64_t total_timesteps, float* obs_cur,
                                float* observations, int64_t* actions, float* rewards, uint8_t* terminated, const int64_t* forced_actions,
                                const double* forced_resets, mi_episode_t* episodes, int32_t* episode_stats, int max_ep, void* stream) {
    return dqn_act_impl(handle, params, n_steps, global_step, slots, learning_starts, start_e, end_e, exploration_fraction, total_timesteps, obs_cur, observations,
                        actions, rewards, terminated, forced_actions, forced_resets, episodes, episode_stats, max_ep, nullptr, true, stream);
}

// the same with the statistics double-buffered by the caller: `episode_stats` must be zero on entry (it is NOT reset by a launch of its own);
// `zero_next` (dev i32 [4], a different buffer) is zeroed by this launch for the next acting call
extern "C" int mi_dqn_act_steps2(void* handle, const float* params, int n_steps, int64_t global_step, int64_t slots, int64_t learning_starts,
                                 double start_e, double end_e, double exploration_fraction, int64_t total_timesteps, float* obs_cur,
                                 float* observations, int64_t* actions, float* rewards, uint8_t* terminated, const int64_t* forced_actions,
                                 const double* forced_resets, mi_episode_t* episodes, int32_t* episode_stats, int max_ep, int32_t* zero_next, void* stream) {
    MI_CHECK_ARG(zero_next && zero_next != episode_stats, "zero_next must be a second statistics buffer");
    return dqn_act_impl(handle, params, n_steps, global_step, slots, learning_starts, start_e, end_e, exploration_fraction, total_timesteps, obs_cur, observations,
                        actions, rewards, terminated, forced_actions, forced_resets, episodes, episode_stats, max_ep, zero_next, false, stream);
}

// ---- sampling: idx[b] = (w0 | w1 << 32) mod upper of Philox(seed, env := update, idx := b, stream 4) -----------------------
__global__ void __launch_bounds__(256) dqn_sample_kernel(uint64_t seed, uint64_t update, uint64_t upper, int batch, int64_t* __restrict__ idx) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= batch) return;
    uint32_t r[4];
    mi_philox(seed, update, (uint64_t)b, STREAM_SAMPLE, r);
    idx[b] = (int64_t)((((uint64_t)r[1] << 32) | r[0]) % upper);
}

extern "C" int mi_dqn_sample(uint64_t seed, uint64_t update_index, int64_t upper_flat, int batch, int64_t* idx, void* stream) {
    MI_CHECK_ARG(idx && batch > 0 && upper_flat > 0, "bad arguments");
    dqn_sample_kernel<<<(batch + 255) / 256, 256, 0, (hipStream_t)stream>>>(seed, update_index, (uint64_t)upper_flat, batch, idx);
    MI_LAUNCH_CHECK();
    return MI_OK;
}

// ---- TD loss + gradient.  A 256-thread workgroup walks groups of R batch rows (group = blockIdx.x, + gridDim.x, ...): a thread owns a hidden unit in the thin
//      phases, the rows' activations sit in LDS (post-ReLU: relu(z) > 0 <=> z > 0, so the mask for the backward pass needs no second copy), the three 120 x 84
//      contractions run on v_mfma_f32_16x16x4_f32.  The shape of PPO's grad_kernel (round 4): at most one workgroup per CU, every weight operand
//      fetched ONCE per workgroup and kept in registers across the groups, the whole gradient accumulated in registers and written as ONE slab per workgroup at the
//      end — until round 3 every 8 rows got a workgroup of their own that re-read both nets and wrote a full 43.7 KB slab (batch 4,096: 512 slabs, 27.6 MB of
//      writes for 0.23 MB of gathers).  R = 8 while the batch gives at most TD_MAX_BLOCKS groups of 8 (one group per workgroup: the round-3 arithmetic, bit for bit);
//      beyond that R = 16, which fills the MFMAs' 16 row columns (at R = 8 half of every B operand is zero) -----------------------------------------------------
#ifdef TD_STAMPS   // diagnostic build: the phases of dqn_td_kernel on the wall clock (s_memrealtime, 100 MHz), wave 0 of every workgroup; tools/dqn_td_stamps.py
__device__ unsigned long long td_mark_dbg[16][16];   // [workgroup < 16][mark]
#define TD_MARK(k) do { __builtin_amdgcn_sched_barrier(0); if (threadIdx.x == 0 && blockIdx.x < 16) { unsigned long long rt_; \
                        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(rt_) :: "memory"); td_mark_dbg[blockIdx.x][k] = rt_; } \
                        __builtin_amdgcn_sched_barrier(0); } while (0)
#define TD_MARK_NOWAIT(k) do { __builtin_amdgcn_sched_barrier(0); if (threadIdx.x == 0 && blockIdx.x < 16) { unsigned long long rt_; \
                        asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(rt_) :: "memory"); td_mark_dbg[blockIdx.x][k] = rt_; } \
                        __builtin_amdgcn_sched_barrier(0); } while (0)
extern "C" int mi_debug_dqn_td_marks(unsigned long long* out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(td_mark_dbg), sizeof(unsigned long long) * 16 * 16) == hipSuccess ? 0 : -2;
}
#else
#define TD_MARK(k) do {} while (0)
#define TD_MARK_NOWAIT(k) do {} while (0)
#endif
#define TD_R 8                 // rows per group of the small-batch form; also the granule of mi_dqn_workspace_bytes
#define TD_MAX_BLOCKS 256      // workgroups (= slabs) per launch at most: one per CU of an MI355X
#define TD_SLAB (DQ_NP + 2)   // + loss
template <int R>
struct __attribute__((aligned(16))) td_smem {
    float W2s[2][DQ_H2 * DQ_H1];   // layer 2 of both nets as in memory (row-major [unit][k]), staged once per workgroup by fully coalesced loads: the MFMA fragments come from here
    float x[2][R][4];          // [0] obs, [1] next obs
    float h1[2][R][DQ_H1];     // [0] online on obs, [1] target on next obs
    float h2[2][R][DQ_H2];
    float q[2][R][2];
    float dq[R][2];
    float dz2[DQ_H2][R];       // [unit][row]: one broadcast b128 pair per unit in the dW2 pass
    float dz1[R][DQ_H1];
    float td_err[R];
    int act[R];
    float w3[2][2][DQ_H2];     // both nets' head weights, staged once per workgroup
    float b3[2][2];
    float rw[R], live[R];      // reward and (not terminated) of the successor row
};

template <int R>
__global__ void __launch_bounds__(256)
dqn_td_kernel(const float* __restrict__ params, const float* __restrict__ target_params, const float* __restrict__ observations,
              const int64_t* __restrict__ actions, const float* __restrict__ rewards, const uint8_t* __restrict__ terminated,
              const int64_t* idx /* may alias idx_out: no __restrict__ */, int batch, int n_envs, long long slots, float gamma, float invn,
              float* __restrict__ workspace, const float* __restrict__ row_w, float* __restrict__ td_abs, uint64_t sample_seed, uint64_t sample_update,
              uint64_t sample_upper, int64_t* idx_out) {
    static_assert(R == 8 || R == 16, "the MFMA passes take the rows as R / 4 k-steps (dW2) and as (part of) a 16-column B operand");
    __shared__ td_smem<R> sm;
    MI_INSIDE_SCOPE(MI_PROF_DQN_TD);
    const int t = threadIdx.x;
    TD_MARK_NOWAIT(0);   // entry
    const int n_groups = (batch + R - 1) / R;
    float* part = workspace + (size_t)blockIdx.x * TD_SLAB;
    // The batch rows of a group: thread (net nt = t / 4R, row r, component k = t & 3) < 8R derives row r's index ITSELF (the four threads of a row repeat the draw or
    // the load: no LDS hand-over, no barrier between index and gather) and fetches its element of obs (nt = 0) or next obs (nt = 1) plus the row's action / the
    // successor's reward and terminated flag.  The index of the NEXT group is requested a whole group ahead.
    // Round 5 (-DTD_STAMPS, tools/dqn_td_stamps.py): the prologue is ONE memory latency deep.  Before, rows were in LDS 4.3 us after entry although a cold load
    // returns in 0.4 us: every guarded load and every operation on a loaded value inside a divergent branch is a basic block that ends in s_waitcnt vmcnt(0) (loads
    // return in order), so the weights, the gathers, the dh1 operands and the thin parameters were four latencies in a row.  Now every request of the prologue is
    // UNCONDITIONAL (all 256 threads, clamped addresses, branch-free), nothing touches a loaded value before the last request has left, and the LDS stores (the
    // COMMIT of a gather) come behind everything.
    const int g_nt = (t / (R * 4)) & 1, g_r = (t >> 2) % R, g_k = t & 3;
    const bool small_ring = (unsigned long long)slots * (unsigned long long)n_envs <= 0xffffffffull;   // uniform: every flat index fits 32 bits (any ring below 4 G transitions)
    const uint64_t upper1 = sample_upper ? sample_upper : 1;
    auto row_b = [&](int grp) -> int { return grp * R + g_r < batch ? grp * R + g_r : batch - 1; };
    auto row_draw = [&](int grp) -> long long {   // batch_inds = randint(upper, size=batch) (dqn.py:116) drawn here: the contract of dqn_sample_kernel, no launch of its own
        uint32_t r4[4];
        mi_philox(sample_seed, sample_update, (uint64_t)row_b(grp), STREAM_SAMPLE, r4);
        return (long long)((((uint64_t)r4[1] << 32) | r4[0]) % upper1);
    };
    struct row_req { float xv; int a; float rw; unsigned term; long long i; };
    auto gather_request = [&](long long i) -> row_req {
        // next-slot row of the same env: ((i / N + 1) % slots) * N + i % N; the wrap is a compare (slot < slots always), not a second modulo
        long long sl, en;
        if (small_ring) { const unsigned qq = (unsigned)i / (unsigned)n_envs; sl = qq; en = (unsigned)i - qq * (unsigned)n_envs; }
        else { sl = i / n_envs; en = i % n_envs; }
        sl = sl + 1 == slots ? 0 : sl + 1;
        const long long nx = sl * n_envs + en;
        row_req q;
        q.i = i;
        q.xv = observations[4 * (g_nt ? nx : i) + g_k];
        q.a = reinterpret_cast<const int*>(actions)[2 * i];   // low dword of the int64 action
        q.rw = rewards[nx];
        q.term = terminated[nx];
        return q;
    };
    // the commit is unconditional too — threads t, t + 8R, ... hold the same (net, row, component) and store the same values — because a load whose only use sits
    // inside a branch is SUNK into that branch by the compiler (seen in the ISA: the action / reward / terminated loads had moved behind every other request)
    auto gather_commit = [&](int grp, const row_req& q) {
        sm.act[g_r] = q.a;
        sm.rw[g_r] = q.rw;
        sm.live[g_r] = q.term ? 0.0f : 1.0f;
        sm.x[g_nt][g_r][g_k] = q.xv;
        if (sample_upper && t < 4 * R && g_k == 0 && grp * R + g_r < batch) idx_out[grp * R + g_r] = q.i;
    };
    const bool first = (int)blockIdx.x < n_groups;
    const int grp0 = first ? (int)blockIdx.x : 0;
    // a caller's index list (teacher-forced runs, PER's sampler) is a load: first in the queue (with the keyed draw `idx` is the output list: a harmless read);
    // the keyed draw is arithmetic and runs while the weights are on their way
    const long long i_list = idx[row_b(grp0)];
    TD_MARK_NOWAIT(1);   // index requested (a caller's list)
    // Order of the requests = order of first use (loads return in order, and a CU's address unit takes ~1.2 us for the 98 KB of layer-2 operands of a workgroup):
    // thin parameters and the rows (layer 1 needs nothing else) in FRONT of the layer-2 operands, which stream in while layer 1 runs; the dh1 operands are requested
    // behind layer 2's matrix instructions and arrive during the thin phases that follow.
    // thin parameters: layer 1 of the thread's (net, unit), the online head's column of thread j < 84 (both only ever read by the threads inside those ranges); both
    // heads into LDS for the forward dot products
    const int net = t >> 7, u = t & 127;                 // threads 0..127: online net, 128..255: target net
    const float* p = net ? target_params : params;
    const int uc = u < DQ_H1 ? u : DQ_H1 - 1, tc = t < DQ_H2 ? t : DQ_H2 - 1;
    const float4 w1v = *reinterpret_cast<const float4*>(p + DQ_W1 + 4 * uc);
    const float b1v = p[DQ_B1 + uc];
    const float w30 = params[DQ_W3 + tc], w31 = params[DQ_W3 + DQ_H2 + tc];
    static_assert(2 * 2 * DQ_H2 <= 2 * 256 && 2 * 2 * DQ_H2 > 256, "two head values per thread");
    const int h3a = t, h3b = t + 256 < 2 * 2 * DQ_H2 ? t + 256 : 2 * 2 * DQ_H2 - 1;
    const float hv0 = (h3a / (2 * DQ_H2) ? target_params : params)[DQ_W3 + h3a % (2 * DQ_H2)];
    const float hv1 = (h3b / (2 * DQ_H2) ? target_params : params)[DQ_W3 + h3b % (2 * DQ_H2)];
    const float hb = ((t & 2) ? target_params : params)[DQ_B3 + (t & 1)];
    const long long i_phx = row_draw(grp0);
    const long long i_cur = sample_upper ? i_phx : i_list;
    row_req q_cur = gather_request(i_cur);
    // MFMA roles (layer 2, dh1, dW2): wave mw, lane (mj, mlg).  None of the weight operands depends on the batch: fetched once, they stay in registers for every
    // group of the workgroup (layer 2: 24 float4 + 12 bias dwords per lane; dh1: 42 dwords).
    // Round 5: the operands go THROUGH LDS.  Requested in fragment layout — a lane's float4 of row 16T + j: sixteen 64-byte segments 480 bytes apart per instruction —
    // a CU's address unit needed ~165 cycles per instruction and wave (-DTD_STAMPS: 28 such loads kept a wave in the issue stage for 2.1 us, and the 42 column-wise
    // dwords of the dh1 operands another 0.5).  Now the workgroup copies both nets' W2 (2 x 40,320 bytes) with 20 fully coalesced float4 loads per thread (1 KB
    // contiguous per instruction) into LDS as it lies in memory, and every lane picks its fragments from there (ds_read_b128 / ds_read_b32).
    const int mw = t >> 6, mj = t & 15, mlg = (t >> 4) & 3;
    const int mnet = mw >> 1;
    const float* mp = mnet ? target_params : params;
    constexpr int W2V = DQ_H2 * DQ_H1 / 4;               // float4 per net
    constexpr int STG = (2 * W2V + 255) / 256;           // float4 per thread (the last ones clamped: threads past the end repeat the last element)
    static_assert(DQ_H2 * DQ_H1 % 4 == 0 && DQ_W2 % 4 == 0, "W2 is float4-aligned");
    f32x4_t stg[STG];
#pragma unroll
    for (int q = 0; q < STG; ++q) {
        const int v = t + 256 * q < 2 * W2V ? t + 256 * q : 2 * W2V - 1;
        stg[q] = v < W2V ? reinterpret_cast<const f32x4_t*>(params + DQ_W2)[v] : reinterpret_cast<const f32x4_t*>(target_params + DQ_W2)[v - W2V];
    }
    f32x4_t wA[3][8], bias2[3];
#pragma unroll
    for (int T3 = 0; T3 < 3; ++T3) {
        const int o0 = 16 * (3 * (mw & 1) + T3) + 4 * mlg;   // 84 = 21 x 4: a lane's four units are all inside or all outside
        bias2[T3] = *reinterpret_cast<const f32x4_t*>(mp + DQ_B2 + (o0 < DQ_H2 ? o0 : DQ_H2 - 4));
    }
    float wa1[2][21];   // the dh1 pass's A operands (W2 read column-wise): requested in the first group, behind layer 2
    TD_MARK_NOWAIT(2);       // every prologue request issued
    // ---- nothing above this line reads a loaded value; the commit below waits for the rows and the thin parameters only ----
    gather_commit(grp0, q_cur);   // (a workgroup without a group commits row 0's values and never reads them: a guard here would sink the gather behind it)
#pragma unroll
    for (int q = 0; q < STG; ++q) {
        const int v = t + 256 * q < 2 * W2V ? t + 256 * q : 2 * W2V - 1;
        reinterpret_cast<f32x4_t*>(&sm.W2s[0][0])[v] = stg[q];
    }
    (&sm.w3[0][0][0])[h3a] = hv0;
    (&sm.w3[0][0][0])[h3b] = hv1;           // (threads past the end repeat the last element)
    sm.b3[(t >> 1) & 1][t & 1] = hb;        // (every thread: the same four values)
    // gradient accumulators over the workgroup's groups
    float g30 = 0.0f, g31 = 0.0f, gb2 = 0.0f;            // t < 84: dW3[0][t], dW3[1][t], db2[t];  t = 84, 85: g30 = db3[t - 84];  t = 86: g30 = loss
    float gb1 = 0.0f, gw1[4] = {0.0f, 0.0f, 0.0f, 0.0f};  // t < 120: db1[t], dW1[t][0..3]
    f32x4_t dW2acc[2][6];
#pragma unroll
    for (int U2 = 0; U2 < 2; ++U2)
#pragma unroll
        for (int T = 0; T < 6; ++T) dW2acc[U2][T] = f32x4_t{0.0f, 0.0f, 0.0f, 0.0f};

    // (between two groups one extra barrier — for x[0], which the dW1 phase reads last — in front of the next group's gather; everything else that gather writes was
    // last read before the final barrier of the group, and every later phase sits behind barriers of its own)
    for (int grp = blockIdx.x; grp < n_groups; grp += gridDim.x) {
        const int row0 = grp * R;
        __syncthreads();                                   // this group's rows are in LDS (gather_group below: before the loop / at the end of the previous group)
        TD_MARK_NOWAIT(3);   // rows in LDS (thread 0 gathers: its own loads have returned; loads return in order, so has every weight request in front of them)
        const int nxt_grp = grp + (int)gridDim.x;
        long long i_nxt = 0;
        if (nxt_grp < n_groups) i_nxt = sample_upper ? row_draw(nxt_grp) : idx[row_b(nxt_grp)];   // requested a whole group ahead (uniform branch)
        if (grp == (int)blockIdx.x) {   // (uniform) the layer-2 fragments of this lane: on their way from LDS while layer 1 runs
#pragma unroll
            for (int T3 = 0; T3 < 3; ++T3) {
                const int u2 = 16 * (3 * (mw & 1) + T3) + mj;
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    const int k = 16 * c + 4 * mlg;
                    wA[T3][c] = *reinterpret_cast<const f32x4_t*>(&sm.W2s[mnet][DQ_H1 * (u2 < DQ_H2 ? u2 : DQ_H2 - 1) + (k < DQ_H1 ? k : DQ_H1 - 4)]);
                }
            }
        }
        // ---- layer 1 ----
        if (u < DQ_H1) {
#pragma unroll
            for (int r = 0; r < R; ++r) {
                float z = b1v;
                z = __builtin_fmaf(w1v.x, sm.x[net][r][0], z); z = __builtin_fmaf(w1v.y, sm.x[net][r][1], z);
                z = __builtin_fmaf(w1v.z, sm.x[net][r][2], z); z = __builtin_fmaf(w1v.w, sm.x[net][r][3], z);
                sm.h1[net][r][u] = fmaxf(z, 0.0f);
            }
        }
        __syncthreads();
        TD_MARK_NOWAIT(4);   // layer 1 done
        // ---- layer 2 on v_mfma_f32_16x16x4_f32, D[unit][row] = W2[unit][k] h1[k][row]: 2 nets x 6 unit tiles over the 4 waves (wave w: net w >> 1, tiles
        //      3 (w & 1) ..+2).  Lane (j, lg) supplies k = 16c + 4lg + r in k-step (c, r): A = W2[16T + j][k] (register-resident), B = h1[row j][k] (one LDS float4
        //      per c; rows >= R are zero columns).
        //      (r01 form: thread per unit, its W2 row streamed as 30 dependent float4 loads from 64 different lines each — 12,100 of the kernel's 34,500 cycles.) ----
        {
            const int j = mj, lg = mlg;
            f32x4_t hB[8], acc2[3];
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const int k = 16 * c + 4 * lg;
                hB[c] = (j < R && k < DQ_H1) ? *reinterpret_cast<const f32x4_t*>(&sm.h1[mnet][j][k]) : f32x4_t{0.0f, 0.0f, 0.0f, 0.0f};
            }
#pragma unroll
            for (int T3 = 0; T3 < 3; ++T3) {   // zero padding of the operands (units >= 84, k >= 120; idempotent): selects on the loaded values, here at their first use
                const int T = 3 * (mw & 1) + T3, u2 = 16 * T + mj, o0 = 16 * T + 4 * mlg;
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    const int k = 16 * c + 4 * mlg;
                    if (!(u2 < DQ_H2 && k < DQ_H1)) wA[T3][c] = f32x4_t{0.0f, 0.0f, 0.0f, 0.0f};
                }
                if (!(o0 < DQ_H2)) bias2[T3] = f32x4_t{0.0f, 0.0f, 0.0f, 0.0f};
            }
#pragma unroll
            for (int T3 = 0; T3 < 3; ++T3) acc2[T3] = bias2[T3];
#pragma unroll
            for (int c = 0; c < 8; ++c)
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int T3 = 0; T3 < 3; ++T3) acc2[T3] = DQ_MFMA(wA[T3][c][r], hB[c][r], acc2[T3]);
            if (grp == (int)blockIdx.x) {   // (uniform) the dh1 operands (the online net's W2 read column-wise), from the staged copy
#pragma unroll
                for (int s2 = 0; s2 < 21; ++s2)
#pragma unroll
                    for (int U2 = 0; U2 < 2; ++U2) {
                        const int k = 16 * (2 * mw + U2) + mj;
                        wa1[U2][s2] = sm.W2s[0][DQ_H1 * (4 * s2 + mlg) + (k < DQ_H1 ? k : DQ_H1 - 1)];
                    }
            }
#pragma unroll
            for (int T3 = 0; T3 < 3; ++T3) {
                const int o = 16 * (3 * (mw & 1) + T3) + 4 * lg;
                if (j < R && o < DQ_H2)
                    *reinterpret_cast<f32x4_t*>(&sm.h2[mnet][j][o]) = f32x4_t{fmaxf(acc2[T3][0], 0.0f), fmaxf(acc2[T3][1], 0.0f), fmaxf(acc2[T3][2], 0.0f), fmaxf(acc2[T3][3], 0.0f)};
            }
        }
        __syncthreads();
        TD_MARK(5);          // layer 2 done (all of this thread's weight operands have arrived)
        // ---- layer 3: 2 nets x R rows x 2 actions dot products of length 84 ----
        if (t < 2 * R * 2) {   // (split over P = 64 / R lanes per dot + a butterfly: 0.80 -> 0.48 us of the launch, but another summation order than the oracle's chain
            // for a gain inside the noise of the iteration: measured in round 5 and not kept)
            const int n3 = t / (R * 2), r = (t >> 1) % R, a = t & 1;
            float acc = 0.0f;
            for (int j = 0; j < DQ_H2; ++j) acc = __builtin_fmaf(sm.w3[n3][a][j], sm.h2[n3][r][j], acc);
            sm.q[n3][r][a] = acc + sm.b3[n3][a];
        }
        __syncthreads();
        TD_MARK_NOWAIT(6);   // layer 3 done
        // ---- TD target, loss, d loss / d q (dqn.py:119-123) ----
        if (t < R) {
            const bool valid = row0 + t < batch;
            const float target_max = fmaxf(sm.q[1][t][0], sm.q[1][t][1]);
            const float td = sm.rw[t] + gamma * target_max * sm.live[t];
            const int a = sm.act[t];
            const float diff = valid ? td - sm.q[0][t][a] : 0.0f;
            const float wb = (row_w && valid) ? row_w[row0 + t] : 1.0f;      // importance weight (per.py:145-147); 1 for plain DQN
            if (td_abs && valid) td_abs[row0 + t] = fabsf(diff);               // the new priority (per.py:141)
            sm.td_err[t] = wb * (diff * diff);
            sm.dq[t][0] = a == 0 ? -2.0f * (wb * diff) * invn : 0.0f;
            sm.dq[t][1] = a == 1 ? -2.0f * (wb * diff) * invn : 0.0f;
        }
        __syncthreads();
        TD_MARK_NOWAIT(7);   // loss done
        // ---- backward through layer 3 (online net only): thread j < 84 ----
        if (t < DQ_H2) {
            float g0 = 0.0f, g1 = 0.0f, gb = 0.0f;
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const float h2 = sm.h2[0][r][t];
                const float d = h2 > 0.0f ? __builtin_fmaf(w31, sm.dq[r][1], w30 * sm.dq[r][0]) : 0.0f;
                sm.dz2[t][r] = d;
                g0 = __builtin_fmaf(sm.dq[r][0], h2, g0); g1 = __builtin_fmaf(sm.dq[r][1], h2, g1);
                gb += d;
            }
            g30 += g0; g31 += g1; gb2 += gb;
        } else if (t < DQ_H2 + 2) {
            const int a = t - DQ_H2;
            float gb = 0.0f;
#pragma unroll
            for (int r = 0; r < R; ++r) gb += sm.dq[r][a];
            g30 += gb;
        } else if (t == DQ_H2 + 2) {
            float l = 0.0f;
#pragma unroll
            for (int r = 0; r < R; ++r) l += sm.td_err[r];
            g30 += l;
        }
        __syncthreads();
        TD_MARK_NOWAIT(8);   // backward through layer 3 done
        // ---- dh1[k][row] = sum_j W2[j][k] dz2[j][row] on the MFMA: 8 k-tiles over the 4 waves (wave w: tiles 2w, 2w + 1), 21 k-steps over j = 4s + lg.
        //      A = W2[j][16U + kk] (register-resident), B = dz2[j][row] from LDS; D: lane (row, lg), register r <-> k = 16U + 4lg + r ----
        {
            const int w = mw, j = mj, lg = mlg;
            f32x4_t dh[2] = {f32x4_t{0.0f, 0.0f, 0.0f, 0.0f}, f32x4_t{0.0f, 0.0f, 0.0f, 0.0f}};
            float db[21];
#pragma unroll
            for (int s2 = 0; s2 < 21; ++s2) db[s2] = j < R ? sm.dz2[4 * s2 + lg][j] : 0.0f;
#pragma unroll
            for (int s2 = 0; s2 < 21; ++s2)   // zero padding (k >= 120; idempotent)
#pragma unroll
                for (int U2 = 0; U2 < 2; ++U2) if (!(16 * (2 * w + U2) + j < DQ_H1)) wa1[U2][s2] = 0.0f;
#pragma unroll
            for (int s2 = 0; s2 < 21; ++s2)
#pragma unroll
                for (int U2 = 0; U2 < 2; ++U2) dh[U2] = DQ_MFMA(wa1[U2][s2], db[s2], dh[U2]);
#pragma unroll
            for (int U2 = 0; U2 < 2; ++U2) {
                const int k = 16 * (2 * w + U2) + 4 * lg;
                if (j < R && k < DQ_H1) {
                    const f32x4_t h = *reinterpret_cast<const f32x4_t*>(&sm.h1[0][j][k]);
                    *reinterpret_cast<f32x4_t*>(&sm.dz1[j][k]) = f32x4_t{h[0] > 0.0f ? dh[U2][0] : 0.0f, h[1] > 0.0f ? dh[U2][1] : 0.0f, h[2] > 0.0f ? dh[U2][2] : 0.0f, h[3] > 0.0f ? dh[U2][3] : 0.0f};
                }
            }
        }
        // ---- dW2 on the MFMA, computed transposed so that a lane ends up with 4 consecutive k of one row: D[k][j] += sum_r h1[r][k] dz2[j][r] (K = the R rows =
        //      R / 4 k-steps); 8 x 6 output tiles, 12 per wave (k tiles 2w, 2w + 1).  A = h1[r = 4s + lg][16U + kk], B = dz2[16T + jj][r = 4s + lg];
        //      D: lane (jj, lg), register r <-> k = 16U + 4lg + r: accumulated over the groups, one float4 store per lane and tile at the end ----
        {
            const int w = mw, kk = mj, lg = mlg;
            float b2[6][R / 4];
#pragma unroll
            for (int T = 0; T < 6; ++T)
#pragma unroll
                for (int s2 = 0; s2 < R / 4; ++s2) { const int jj = 16 * T + kk; b2[T][s2] = jj < DQ_H2 ? sm.dz2[jj][4 * s2 + lg] : 0.0f; }
#pragma unroll
            for (int U2 = 0; U2 < 2; ++U2) {
                const int ka = 16 * (2 * w + U2) + kk;
                float a2[R / 4];
#pragma unroll
                for (int s2 = 0; s2 < R / 4; ++s2) a2[s2] = ka < DQ_H1 ? sm.h1[0][4 * s2 + lg][ka] : 0.0f;
#pragma unroll
                for (int T = 0; T < 6; ++T)
#pragma unroll
                    for (int s2 = 0; s2 < R / 4; ++s2) dW2acc[U2][T] = DQ_MFMA(a2[s2], b2[T][s2], dW2acc[U2][T]);
            }
        }
        __syncthreads();   // dz1 complete
        TD_MARK(9);          // dh1 + dW2 done
        // ---- db1, dW1: thread k < 120 ----
        if (t < DQ_H1) {
            float gb = 0.0f, gw[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const float d = sm.dz1[r][t];
                gb += d;
#pragma unroll
                for (int c = 0; c < 4; ++c) gw[c] = __builtin_fmaf(d, sm.x[0][r][c], gw[c]);
            }
            gb1 += gb;
#pragma unroll
            for (int c = 0; c < 4; ++c) gw1[c] += gw[c];
        }
        if (nxt_grp < n_groups) { const row_req q_nxt = gather_request(i_nxt); __syncthreads(); gather_commit(nxt_grp, q_nxt); }
    }
    TD_MARK_NOWAIT(10);      // main loop done
    // ---- the workgroup's slab: every gradient element once ----
    if (t < DQ_H2) { part[DQ_W3 + t] = g30; part[DQ_W3 + DQ_H2 + t] = g31; part[DQ_B2 + t] = gb2; }
    else if (t < DQ_H2 + 2) part[DQ_B3 + (t - DQ_H2)] = g30;
    else if (t == DQ_H2 + 2) part[DQ_NP] = g30;
    {
        const int w = mw, kk = mj, lg = mlg;
#pragma unroll
        for (int U2 = 0; U2 < 2; ++U2) {
            const int k0 = 16 * (2 * w + U2) + 4 * lg;
#pragma unroll
            for (int T = 0; T < 6; ++T) {
                const int jj = 16 * T + kk;
                if (jj < DQ_H2 && k0 < DQ_H1) *reinterpret_cast<f32x4_t*>(part + DQ_W2 + DQ_H1 * jj + k0) = dW2acc[U2][T];
            }
        }
    }
    if (t < DQ_H1) {
        part[DQ_B1 + t] = gb1;
        *reinterpret_cast<float4*>(part + DQ_W1 + 4 * t) = make_float4(gw1[0], gw1[1], gw1[2], gw1[3]);
    }
    TD_MARK(11);             // slab stores acknowledged
}

// grads[p] = sum over workgroup slabs in slab order; loss = sum of the slab losses * inv_count.  With `opt.params` set the same launch applies
// optimizer.step() (torch Adam without clipping, dqn.py:68,133) to the element it has just summed: no launch of its own in single-process runs.
// Dueling layout (q_network1.parameters(), dueling_dqn.py:24-40): features as DQN | value W[84] b | advantage W[2][84] b[2] = MI_DUELING_NPARAMS
#define DU_WV 10764
#define DU_BV 10848
#define DU_WA 10849
#define DU_BA 11017
#define DU_NP 11019
// `du_params` set (mi_dueling_td_update): params is the plain-DQN IMAGE of a dueling net (see the dueling block below), m / v / du_params / du_grads are in the dueling
// layout; the element that has summed a plain gradient maps it back (mi_dueling_unpack_grads' expressions), steps the dueling parameters and rewrites the image
// (mi_dueling_pack's expressions): the thread of head element (action 0, unit j) also sums action 1's slabs and owns Wv[j], Wa[0][j], Wa[1][j]; likewise the biases.
struct dqn_opt_t { float* params; float* m; float* v; float w1, b2, w2, step_size, rbc2, eps; float* du_params; float* du_grads; };
__device__ __forceinline__ float dqn_opt_step(const dqn_opt_t& o, float p, float g, int at) {   // Adam on optimizer-state element `at`
    float mi = o.m[at], vi = o.v[at];
    const float r = mi_adam_elem(p, g, mi, vi, o.w1, o.b2, o.w2, o.step_size, o.rbc2, o.eps);
    o.m[at] = mi; o.v[at] = vi;
    return r;
}
// one head column (or the bias triple): plain gradients g0 / g1 of the two actions -> the three dueling parameters stepped, the two image elements rewritten
__device__ __forceinline__ void dueling_head_step(const dqn_opt_t& o, float g0, float g1, int at_v, int at_a0, int at_a1, int img0, int img1) {
    const float gv = g0 + g1;                                  // dWv = sum_a g3[a]
    const float ga0 = g0 - (g0 + g1) / 2.0f, ga1 = g1 - (g0 + g1) / 2.0f;   // dWa[k] = g3[k] - mean_a g3[a]
    o.du_grads[at_v] = gv; o.du_grads[at_a0] = ga0; o.du_grads[at_a1] = ga1;
    const float wv = dqn_opt_step(o, o.du_params[at_v], gv, at_v);
    const float wa0 = dqn_opt_step(o, o.du_params[at_a0], ga0, at_a0);
    const float wa1 = dqn_opt_step(o, o.du_params[at_a1], ga1, at_a1);
    o.du_params[at_v] = wv; o.du_params[at_a0] = wa0; o.du_params[at_a1] = wa1;
    const float mean = (wa0 + wa1) / 2.0f;                     // W3eff[a] = Wv + (Wa[a] - mean_a Wa)
    o.params[img0] = wv + (wa0 - mean);
    o.params[img1] = wv + (wa1 - mean);
}
// PER: one more workgroup behind the summing ones carries per.py:144-145 (per_scatter_role) — |td| and the indices are final at the TD kernel's boundary, and nothing here
// reads priorities.
// WORLD > 0 (sharded run on the P2P carrier without gradient clipping, round 6): the all-reduce of {gradient, loss} (dqn.py:131 -> :133 with the exchange in between)
// happens HERE — the thread that has just summed element p exchanges it as line p (p2p_exchange: rank-ordered sum, the same bits on every rank) and steps it — instead of
// in a launch of its own followed by a clip + Adam launch: the sharded iteration is the single process's two launches plus one exchange latency.  `gate`: the carrier's
// status word (never null when WORLD > 0): a wait that ran out withholds the step.
template <bool PER, int WORLD>
__global__ void __launch_bounds__(256) dqn_reduce_kernel(const float* __restrict__ workspace, int n_slabs, double inv_count,
                                                         float* __restrict__ grads, float* __restrict__ loss, dqn_opt_t opt, per_scatter_t sc, const p2p_args_t x,
                                                         const uint32_t* __restrict__ gate) {
    MI_INSIDE_SCOPE(MI_PROF_DQN_REDUCE);
    if constexpr (PER) {
        if (blockIdx.x == gridDim.x - 1) { __shared__ float wmax[4]; per_scatter_role(wmax, sc); return; }
    }
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p < DQ_NP) {
        // the optimizer state is requested before the slabs, and the slabs 16 at a time with every load in flight at once (the kernel is one memory
        // latency deep instead of one per group of four); the sum keeps its order: accumulator b & 3 takes slab b, then (0 + 1) + (2 + 3)
        float pi = 0.0f, mi = 0.0f, vi = 0.0f;
        if (opt.params && (!opt.du_params || p < DQ_W3)) { pi = opt.params[p]; mi = opt.m[p]; vi = opt.v[p]; }   // (dueling: the image's feature layers are copies of the parameters)
        float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        int b = 0;
        for (; b + 16 <= n_slabs; b += 16) {
            float x[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) x[k] = workspace[(size_t)(b + k) * TD_SLAB + p];
#pragma unroll
            for (int k = 0; k < 16; ++k) acc[k & 3] += x[k];
        }
        for (; b < n_slabs; ++b) acc[b & 3] += workspace[(size_t)b * TD_SLAB + p];
        float g = (acc[0] + acc[1]) + (acc[2] + acc[3]);
        if constexpr (WORLD > 0) {
            g = p2p_exchange<WORLD>(x, p, g);
            grads[p] = g;
            if (opt.params && !mi_gate_closed(gate)) {
                opt.params[p] = mi_adam_elem(pi, g, mi, vi, opt.w1, opt.b2, opt.w2, opt.step_size, opt.rbc2, opt.eps);
                opt.m[p] = mi; opt.v[p] = vi;
            }
            return;
        }
        grads[p] = g;
        if (opt.du_params) {
            if (p < DQ_W3) {                                   // feature layers: the same index in both layouts; the image holds a copy
                opt.du_grads[p] = g;
                const float w = mi_adam_elem(pi, g, mi, vi, opt.w1, opt.b2, opt.w2, opt.step_size, opt.rbc2, opt.eps);
                opt.du_params[p] = w; opt.params[p] = w;
                opt.m[p] = mi; opt.v[p] = vi;
            } else if (p < DQ_W3 + DQ_H2 || p == DQ_B3) {      // action 0's element: sums action 1's slabs too (same order) and steps the column
                const int p1 = p < DQ_B3 ? p + DQ_H2 : p + 1;
                float a1[4] = {0.0f, 0.0f, 0.0f, 0.0f};
                int b1 = 0;
                for (; b1 + 16 <= n_slabs; b1 += 16) {             // (the same 16-at-a-time form as above: every load in flight at once)
                    float y[16];
#pragma unroll
                    for (int k = 0; k < 16; ++k) y[k] = workspace[(size_t)(b1 + k) * TD_SLAB + p1];
#pragma unroll
                    for (int k = 0; k < 16; ++k) a1[k & 3] += y[k];
                }
                for (; b1 < n_slabs; ++b1) a1[b1 & 3] += workspace[(size_t)b1 * TD_SLAB + p1];
                const float g1 = (a1[0] + a1[1]) + (a1[2] + a1[3]);
                if (p < DQ_B3) { const int j = p - DQ_W3; dueling_head_step(opt, g, g1, DU_WV + j, DU_WA + j, DU_WA + DQ_H2 + j, p, p1); }
                else dueling_head_step(opt, g, g1, DU_BV, DU_BA, DU_BA + 1, p, p1);
            }
        } else if (opt.params) {   // the formula of clip_adam_kernel at coef = 1 (max_norm = inf), bit for bit (mi_adam_elem)
            opt.params[p] = mi_adam_elem(pi, g, mi, vi, opt.w1, opt.b2, opt.w2, opt.step_size, opt.rbc2, opt.eps);
            opt.m[p] = mi; opt.v[p] = vi;
        }
    } else if (p == DQ_NP && loss) {
        double l = 0.0;
        for (int b = 0; b < n_slabs; ++b) l += workspace[(size_t)b * TD_SLAB + DQ_NP];
        loss[0] = p2p_exchange<WORLD>(x, DQ_NP, (float)(l * inv_count));   // (WORLD = 0: the value itself)
    }
}

// The same for MANY slabs (scaled batches: 512 slabs at batch 4,096, where the kernel above is 32 dependent rounds of 16 loads per thread: 135 us).  Workgroup = 64
// parameters x 16 slab groups (the shape of PPO's grad_reduce_kernel): group sg sums slabs sg, sg + 16, ... eight loads at a time, the 16 group sums are added in group
// order through LDS.  A different (fixed) summation order than the kernel above: each batch size always takes the same kernel, so every run is reproducible.
#define DR_PARAMS 64
#define DR_GROUPS 16
#define DR_MIN_SLABS 64   // from this many slabs on
template <bool PER, int WORLD>
__global__ void __launch_bounds__(DR_PARAMS * DR_GROUPS) dqn_reduce2_kernel(const float* __restrict__ workspace, int n_slabs, double inv_count,
                                                                         float* __restrict__ grads, float* __restrict__ loss, dqn_opt_t opt, per_scatter_t sc,
                                                                         const p2p_args_t x, const uint32_t* __restrict__ gate) {
    MI_INSIDE_SCOPE(MI_PROF_DQN_REDUCE);
    if constexpr (PER) {
        if (blockIdx.x == gridDim.x - 1) { __shared__ float wmax[DR_GROUPS]; per_scatter_role(wmax, sc); return; }
    }
    const int pl = threadIdx.x & (DR_PARAMS - 1), sg = threadIdx.x >> 6;
    const int pblocks = (DQ_NP + DR_PARAMS - 1) / DR_PARAMS;
    if ((int)blockIdx.x < pblocks) {
        __shared__ float part[DR_GROUPS][DR_PARAMS];
        const int p = blockIdx.x * DR_PARAMS + pl;
        float pi = 0.0f, mi = 0.0f, vi = 0.0f;
        if (sg == 0 && p < DQ_NP && opt.params) { pi = opt.params[p]; mi = opt.m[p]; vi = opt.v[p]; }
        float acc = 0.0f;
        if (p < DQ_NP) {
            const float* src = workspace + p;
            int b = sg;
            for (; b + 7 * DR_GROUPS < n_slabs; b += 8 * DR_GROUPS) {
                float x[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) x[k] = src[(size_t)(b + DR_GROUPS * k) * TD_SLAB];
#pragma unroll
                for (int k = 0; k < 8; ++k) acc += x[k];
            }
            for (; b < n_slabs; b += DR_GROUPS) acc += src[(size_t)b * TD_SLAB];
        }
        part[sg][pl] = acc;
        __syncthreads();
        if (sg == 0 && p < DQ_NP) {
            float g = 0.0f;
#pragma unroll
            for (int k = 0; k < DR_GROUPS; ++k) g += part[k][pl];
            g = p2p_exchange<WORLD>(x, p, g);   // (WORLD = 0: the value itself)
            grads[p] = g;
            if (opt.params && !(WORLD > 0 && mi_gate_closed(gate))) {
                opt.params[p] = mi_adam_elem(pi, g, mi, vi, opt.w1, opt.b2, opt.w2, opt.step_size, opt.rbc2, opt.eps);
                opt.m[p] = mi; opt.v[p] = vi;
            }
        }
    } else if (loss) {   // the slab losses: lane-strided partial sums in f64, fixed butterfly, 16 wave sums added in wave order
        __shared__ double wsum[DR_GROUPS];
        double l = 0.0;
        for (int b = threadIdx.x; b < n_slabs; b += DR_PARAMS * DR_GROUPS) l += workspace[(size_t)b * TD_SLAB + DQ_NP];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) l += __shfl_xor(l, o);
        if (pl == 0) wsum[sg] = l;
        __syncthreads();
        if (threadIdx.x == 0) {
            double t = 0.0;
            for (int k = 0; k < DR_GROUPS; ++k) t += wsum[k];
            loss[0] = p2p_exchange<WORLD>(x, DQ_NP, (float)(t * inv_count));
        }
    }
}

extern "C" size_t mi_dqn_workspace_bytes(int batch) {
    return (size_t)((batch + TD_R - 1) / TD_R) * TD_SLAB * sizeof(float);
}

static dqn_opt_t dqn_no_opt() { dqn_opt_t o; memset(&o, 0, sizeof(o)); return o; }

static int dqn_td_impl(const float* params, const float* target_params, const float* observations, const int64_t* actions,
                       const float* rewards, const uint8_t* terminated, const int64_t* idx, int batch, int n_envs, int64_t slots,
                       float gamma, double inv_count, void* workspace, float* grads, float* loss, const float* weights, float* td_abs, const dqn_opt_t& opt,
                       uint64_t sample_seed, uint64_t sample_update, int64_t sample_upper, void* stream, const per_scatter_t* scatter = nullptr, void* p2p = nullptr) {
    MI_CHECK_ARG(params && target_params && observations && actions && rewards && terminated && idx && workspace && grads, "NULL pointer");
    MI_CHECK_ARG(sample_upper >= 0, "sample_upper must be >= 0");
    MI_CHECK_ARG(batch > 0 && n_envs > 0 && slots >= 2, "batch, n_envs must be positive and slots >= 2");
    hipStream_t s = (hipStream_t)stream;
    // one group of 8 rows per workgroup while that gives at most TD_MAX_BLOCKS workgroups (the reference's batch of 128: 16); larger batches: groups of 16 rows dealt
    // round-robin to TD_MAX_BLOCKS workgroups, one slab each.  A batch size always takes the same form: every run is reproducible.
    const int groups8 = (batch + TD_R - 1) / TD_R, groups16 = (batch + 15) / 16;
    const bool wide = groups8 > TD_MAX_BLOCKS;
    const int blocks = wide ? (groups16 < TD_MAX_BLOCKS ? groups16 : TD_MAX_BLOCKS) : groups8;
    {
        mi_prof_scope prof(MI_PROF_DQN_TD, s);
        if (wide)
            dqn_td_kernel<16><<<blocks, 256, 0, s>>>(params, target_params, observations, actions, rewards, terminated, idx, batch, n_envs,
                                                     (long long)slots, gamma, (float)inv_count, (float*)workspace, weights, td_abs, sample_seed, sample_update,
                                                     (uint64_t)sample_upper, (int64_t*)idx);
        else
            dqn_td_kernel<TD_R><<<blocks, 256, 0, s>>>(params, target_params, observations, actions, rewards, terminated, idx, batch, n_envs,
                                                       (long long)slots, gamma, (float)inv_count, (float*)workspace, weights, td_abs, sample_seed, sample_update,
                                                       (uint64_t)sample_upper, (int64_t*)idx);
    }
    MI_LAUNCH_CHECK();
    {
        mi_prof_scope prof(MI_PROF_DQN_REDUCE, s);
        const int g2 = (DQ_NP + DR_PARAMS - 1) / DR_PARAMS + 1, g1 = (DQ_NP + 1 + 255) / 256;
        const bool many = blocks >= DR_MIN_SLABS;
        per_scatter_t none;
        memset(&none, 0, sizeof(none));
        p2p_args_t x;
        memset(&x, 0, sizeof(x));
        int world = 0;
        if (p2p) {   // the slab sum also all-reduces {grads, loss} over the P2P carrier (exactly one exchange per call, on every rank) and steps the parameters
            if (scatter || opt.du_params || loss != grads + DQ_NP) { mi_set_error("dqn_td_impl: the in-launch exchange needs loss == grads + MI_DQN_NPARAMS and a plain DQN step"); return MI_EINVAL; }
            const int rc = mi_comm_p2p_next(p2p, (size_t)DQ_NP + 1, &x, &world, s);
            if (rc) return rc;
        }
        const uint32_t* gate = mi_comm_gate(p2p);
#define DR_LAUNCH(W) do { if (many) dqn_reduce2_kernel<false, W><<<g2, DR_PARAMS * DR_GROUPS, 0, s>>>((const float*)workspace, blocks, inv_count, grads, loss, opt, none, x, gate); \
                          else dqn_reduce_kernel<false, W><<<g1, 256, 0, s>>>((const float*)workspace, blocks, inv_count, grads, loss, opt, none, x, gate); } while (0)
        if (scatter) {   // PER's one-call update: one more workgroup carries the priority scatter + max_priority
            if (many) dqn_reduce2_kernel<true, 0><<<g2 + 1, DR_PARAMS * DR_GROUPS, 0, s>>>((const float*)workspace, blocks, inv_count, grads, loss, opt, *scatter, x, nullptr);
            else dqn_reduce_kernel<true, 0><<<g1 + 1, 256, 0, s>>>((const float*)workspace, blocks, inv_count, grads, loss, opt, *scatter, x, nullptr);
        } else {
            switch (world) {
                case 0: DR_LAUNCH(0); break;
                case 1: DR_LAUNCH(1); break;
                case 2: DR_LAUNCH(2); break;
                case 3: DR_LAUNCH(3); break;
                case 4: DR_LAUNCH(4); break;
                case 5: DR_LAUNCH(5); break;
                case 6: DR_LAUNCH(6); break;
                case 7: DR_LAUNCH(7); break;
                default: DR_LAUNCH(8); break;
            }
        }
#undef DR_LAUNCH
    }
    MI_LAUNCH_CHECK();
    return MI_OK;
}

extern "C" int mi_dqn_td_grad(const float* params, const float* target_params, const float* observations, const int64_t* actions,
                              const float* rewards, const uint8_t* terminated, const int64_t* idx, int batch, int n_envs, int64_t slots,
                              float gamma, double inv_count, void* workspace, float* grads, float* loss, void* stream) {
    return dqn_td_impl(params, target_params, observations, actions, rewards, terminated, idx, batch, n_envs, slots, gamma, inv_count, workspace, grads, loss,
                       nullptr, nullptr, dqn_no_opt(), 0, 0, 0, stream);
}

// sharded runs, ONE C call (the pattern of mi_ppo_update_sharded): TD gradient share (scaled by 1 / (world * batch)) + slab sum, an in-stream SUM all-reduce
// of gradbuf = {grads [MI_DQN_NPARAMS], loss, pad} and optimizer.step() (mi_clip_adam with `max_norm`) — the same arithmetic as the host-sequenced route
// (mi_dqn_td_grad / mi_per_td_grad, torch.distributed.all_reduce, mi_clip_adam), so the two agree bit for bit; no Python between launches.  sample_upper > 0: the TD launch
// draws the batch itself (dqn.py:116, the keyed uniform draw of mi_dqn_sample) and writes it to idx.
extern "C" int mi_dqn_td_update_sharded(float* params, const float* target_params, const float* observations, const int64_t* actions, const float* rewards,
                                        const uint8_t* terminated, int64_t* idx, int batch, int n_envs, int64_t slots, float gamma, const float* weights,
                                        float* td_abs, void* workspace, float* gradbuf, float* exp_avg, float* exp_avg_sq, int64_t step, double lr, double beta1,
                                        double beta2, double eps, float max_norm, float* grad_norm, uint64_t sample_seed, uint64_t sample_update, int64_t sample_upper,
                                        void* comm, void* stream) {
    MI_CHECK_ARG(gradbuf && exp_avg && exp_avg_sq && step >= 1, "NULL optimizer state / bad step");
    MI_CHECK_ARG(!(sample_upper > 0 && weights), "in-kernel uniform sampling and importance weights exclude each other");
    int world = 1;
    if (comm) {
        if (const int rc = mi_comm_poll_impl(comm)) return rc;   // an earlier wait of the P2P carrier ran out: MI_ESTATE before anything is enqueued
        if (const int rc = mi_comm_info(comm, &world, nullptr, nullptr, nullptr)) return rc;
    }
    const double inv_count = 1.0 / ((double)batch * world);
    // P2P carrier, no gradient clipping (dqn.py / per.py have none): the slab-sum launch exchanges each element it has summed and steps it — the single process's two
    // launches plus one exchange latency (the fused form needs no norm; with a finite max_norm the clip coefficient is a grid-wide dependency: the sequence below)
    if (comm && mi_comm_p2p_fused_ok(comm) && max_norm == __builtin_inff()) {
        dqn_opt_t o;
        const double bc1 = 1.0 - pow(beta1, (double)step), bc2 = 1.0 - pow(beta2, (double)step);
        o.params = params; o.m = exp_avg; o.v = exp_avg_sq; o.w1 = (float)(1.0 - beta1); o.b2 = (float)beta2; o.w2 = (float)(1.0 - beta2);
        o.step_size = (float)(lr / bc1); o.rbc2 = (float)(1.0 / sqrt(bc2)); o.eps = (float)eps; o.du_params = nullptr; o.du_grads = nullptr;
        return dqn_td_impl(params, target_params, observations, actions, rewards, terminated, idx, batch, n_envs, slots, gamma, inv_count, workspace, gradbuf, gradbuf + DQ_NP,
                           weights, td_abs, o, sample_seed, sample_update, sample_upper, stream, nullptr, comm);
    }
    int rc = dqn_td_impl(params, target_params, observations, actions, rewards, terminated, idx, batch, n_envs, slots, gamma, inv_count, workspace,
                         gradbuf, gradbuf + DQ_NP, weights, td_abs, dqn_no_opt(), sample_seed, sample_update, sample_upper, stream);
    if (rc) return rc;
    if (comm) {
        mi_prof_scope prof(MI_PROF_COMM_GRAD, (hipStream_t)stream);
        rc = mi_comm_allreduce_impl(comm, gradbuf, (size_t)DQ_NP + 2, 0, (hipStream_t)stream);
        if (rc) return rc;
    }
    return mi_clip_adam_gated(params, gradbuf, exp_avg, exp_avg_sq, DQ_NP, step, lr, beta1, beta2, eps, max_norm, grad_norm, comm, stream);   // withheld after a timed-out exchange
}

// single-process fusion: TD gradient (optionally importance-weighted, weights / td_abs nullable together) + optimizer.step() in two launches
extern "C" int mi_dqn_td_update(float* params, const float* target_params, const float* observations, const int64_t* actions,
                                const float* rewards, const uint8_t* terminated, int64_t* idx, int batch, int n_envs, int64_t slots,
                                float gamma, const float* weights, float* td_abs, void* workspace, float* grads, float* loss, float* exp_avg, float* exp_avg_sq,
                                int64_t step, double lr, double beta1, double beta2, double eps, uint64_t sample_seed, uint64_t sample_update, int64_t sample_upper,
                                void* stream) {
    MI_CHECK_ARG(exp_avg && exp_avg_sq && step >= 1 && (!weights == !td_abs), "bad optimizer state / weights and td_abs go together");
    MI_CHECK_ARG(!(sample_upper > 0 && weights), "in-kernel uniform sampling and importance weights exclude each other");
    dqn_opt_t o;
    const double bc1 = 1.0 - pow(beta1, (double)step), bc2 = 1.0 - pow(beta2, (double)step);
    o.params = params; o.m = exp_avg; o.v = exp_avg_sq; o.w1 = (float)(1.0 - beta1); o.b2 = (float)beta2; o.w2 = (float)(1.0 - beta2);
    o.step_size = (float)(lr / bc1); o.rbc2 = (float)(1.0 / sqrt(bc2)); o.eps = (float)eps; o.du_params = nullptr; o.du_grads = nullptr;
    return dqn_td_impl(params, target_params, observations, actions, rewards, terminated, idx, batch, n_envs, slots, gamma, 1.0 / batch, workspace, grads, loss,
                       weights, td_abs, o, sample_seed, sample_update, sample_upper, stream);
}

// ---- Dueling head (reference deep_rl/dueling_dqn.py:24-40) as an epilogue on the DQN kernels -----------------
// values + (advantages - mean(advantages)) is LINEAR in the 84 features, so a dueling net equals a plain 4->120->84->2 net whose
// head is W3eff[a] = Wv + (Wa[a] - mean_a Wa), b3eff[a] = bv + (ba[a] - mean_a ba): the acting and TD kernels run unchanged on the
// packed parameters, and the chain rule maps the plain head's gradient back: dWv = sum_a g3[a], dWa[k] = g3[k] - mean_a g3[a].
// Dueling layout (q_network1.parameters()): features as DQN | value W[84] b | advantage W[2][84] b[2] = MI_DUELING_NPARAMS (DU_* above dqn_opt_t).
__global__ void __launch_bounds__(256) dueling_pack_kernel(const float* __restrict__ d, float* __restrict__ q) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < DQ_W3) q[i] = d[i];                                   // feature layers
    else if (i < DQ_W3 + 2 * DQ_H2) {
        const int a = (i - DQ_W3) / DQ_H2, j = (i - DQ_W3) % DQ_H2;
        const float mean = (d[DU_WA + j] + d[DU_WA + DQ_H2 + j]) / 2.0f;
        q[i] = d[DU_WV + j] + (d[DU_WA + a * DQ_H2 + j] - mean);
    } else if (i < DQ_NP) {
        const int a = i - DQ_B3;
        const float mean = (d[DU_BA] + d[DU_BA + 1]) / 2.0f;
        q[i] = d[DU_BV] + (d[DU_BA + a] - mean);
    }
}

__global__ void __launch_bounds__(256) dueling_unpack_kernel(const float* __restrict__ g, float* __restrict__ dg) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < DQ_W3) dg[i] = g[i];
    else if (i < DU_BV) { const int j = i - DU_WV; dg[i] = g[DQ_W3 + j] + g[DQ_W3 + DQ_H2 + j]; }
    else if (i == DU_BV) dg[i] = g[DQ_B3] + g[DQ_B3 + 1];
    else if (i < DU_BA) {
        const int a = (i - DU_WA) / DQ_H2, j = (i - DU_WA) % DQ_H2;
        dg[i] = g[DQ_W3 + a * DQ_H2 + j] - (g[DQ_W3 + j] + g[DQ_W3 + DQ_H2 + j]) / 2.0f;
    } else if (i < DU_NP) {
        const int a = i - DU_BA;
        dg[i] = g[DQ_B3 + a] - (g[DQ_B3] + g[DQ_B3 + 1]) / 2.0f;
    }
}

extern "C" int mi_dueling_pack(const float* dueling_params, float* dqn_params, void* stream) {
    MI_CHECK_ARG(dueling_params && dqn_params, "NULL pointer");
    dueling_pack_kernel<<<(DQ_NP + 255) / 256, 256, 0, (hipStream_t)stream>>>(dueling_params, dqn_params);
    MI_LAUNCH_CHECK();
    return MI_OK;
}

extern "C" int mi_dueling_unpack_grads(const float* dqn_grads, float* dueling_grads, void* stream) {
    MI_CHECK_ARG(dqn_grads && dueling_grads, "NULL pointer");
    dueling_unpack_kernel<<<(DU_NP + 255) / 256, 256, 0, (hipStream_t)stream>>>(dqn_grads, dueling_grads);
    MI_LAUNCH_CHECK();
    return MI_OK;
}

// dueling_dqn.py:109-129 as ONE call (single process, no clipping): TD gradient on the plain-DQN images + slab sum, and — in the slab-sum launch, batches below
// DR_MIN_SLABS row groups; behind it as three launches otherwise — the gradient mapped back, Adam on the dueling parameters, the image rewritten.  Bit-identical to
// mi_dqn_td_grad(images) + mi_dueling_unpack_grads + mi_clip_adam(max_norm = +inf) on the dueling vector + mi_dueling_pack.
extern "C" int mi_dueling_td_update(float* params_img, const float* target_img, const float* observations, const int64_t* actions, const float* rewards,
                                    const uint8_t* terminated, int64_t* idx, int batch, int n_envs, int64_t slots, float gamma, void* workspace, float* grads, float* loss,
                                    float* dueling_params, float* dueling_grads, float* exp_avg, float* exp_avg_sq, int64_t step, double lr, double beta1, double beta2,
                                    double eps, uint64_t sample_seed, uint64_t sample_update, int64_t sample_upper, void* stream) {
    MI_CHECK_ARG(params_img && dueling_params && dueling_grads && exp_avg && exp_avg_sq && step >= 1, "NULL pointer / bad step");
    const int groups8 = (batch + TD_R - 1) / TD_R;
    if (groups8 >= DR_MIN_SLABS) {   // the many-slab sum keeps its own shape: epilogue as launches of its own (still no Python between them)
        int rc = dqn_td_impl(params_img, target_img, observations, actions, rewards, terminated, idx, batch, n_envs, slots, gamma, 1.0 / batch, workspace, grads, loss,
                             nullptr, nullptr, dqn_no_opt(), sample_seed, sample_update, sample_upper, stream);
        if (rc) return rc;
        if ((rc = mi_dueling_unpack_grads(grads, dueling_grads, stream))) return rc;
        if ((rc = mi_clip_adam(dueling_params, dueling_grads, exp_avg, exp_avg_sq, DU_NP, step, lr, beta1, beta2, eps, INFINITY, nullptr, stream))) return rc;
        return mi_dueling_pack(dueling_params, params_img, stream);
    }
    dqn_opt_t o;
    const double bc1 = 1.0 - pow(beta1, (double)step), bc2 = 1.0 - pow(beta2, (double)step);
    o.params = params_img; o.m = exp_avg; o.v = exp_avg_sq; o.w1 = (float)(1.0 - beta1); o.b2 = (float)beta2; o.w2 = (float)(1.0 - beta2);
    o.step_size = (float)(lr / bc1); o.rbc2 = (float)(1.0 / sqrt(bc2)); o.eps = (float)eps; o.du_params = dueling_params; o.du_grads = dueling_grads;
    return dqn_td_impl(params_img, target_img, observations, actions, rewards, terminated, idx, batch, n_envs, slots, gamma, 1.0 / batch, workspace, grads, loss,
                       nullptr, nullptr, o, sample_seed, sample_update, sample_upper, stream);
}

// ---- Prioritized replay (reference deep_rl/per.py) as epilogues on the DQN path ---------------------------------
// priorities: f32 ring [slots][N] beside the replay ring.  per.py samples with torch.multinomial(priorities) — an O(buffer) scan on
// the host generator; here: a keyed three-level prefix-sum descent with a FIXED evaluation order (the contract the oracle implements
// bit for bit): s0 = sums of 64-entry chunks, s1 = sums of 64 s0's, total — all sequential, in double (-ffp-contract=off: no FMA).
// (PER_CHUNK, STREAM_PER, per_ws_t, per_pow, per_n0, per_mark_t: defined in front of the acting kernel, which carries PER's riding roles)
// p^alpha of a priority (per.py:131): 2^(alpha log2 p) on the hardware log2 / exp2 (3 instructions instead of ~150 for powf; relative
// error ~1e-6, it only enters the importance weights, which are compared to 2e-5).  p = 0 -> 0 for EVERY alpha, explicitly: never-written entries and
// the ring's write head must contribute +0 to the incremental sums, and alpha = 0 (uniform PER, legitimate in per.py) would otherwise give 0 * -inf = NaN.
// (torch's 0 ** 0 = 1 would add the count of never-written entries to sum p^alpha at alpha = 0; the weights of per.py:145-146 are normalised by their
// maximum and every sampled entry has p > 0, so they are exactly 1 either way.  the CPU oracle makes the same choice.)
// (per_pow / per_n0: see the top of the file)
static per_ws_t per_ws(void* workspace, int64_t capacity) {
    per_ws_t w;
    const int64_t c0 = per_n0(capacity), c1 = per_n0(c0);
    w.s0 = (double*)workspace; w.a0 = w.s0 + c0; w.s1 = w.a0 + c0; w.a1 = w.s1 + c1; w.totals = w.a1 + c1;
    return w;
}
extern "C" size_t mi_per_workspace_bytes(int64_t capacity) {
    const int64_t c0 = per_n0(capacity), c1 = per_n0(c0);
    return (size_t)(2 * c0 + 2 * c1 + 2) * sizeof(double);
}

// priorities[global_step .. + n_steps) = max_priority (per.py:106; max_priority only changes at an update, i.e. between acting calls);
// the slot after the last written one is the ring's write head (its successor data is not there yet): priority 0, never sampled
__global__ void __launch_bounds__(256) per_mark_kernel(float* __restrict__ prio, int N, long long slots, long long gs, int n_steps, const float* __restrict__ max_prio) {
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= (long long)(n_steps + 1) * N) return;
    const int s = (int)(t / N), e = (int)(t % N);
    prio[((gs + s) % slots) * N + e] = s < n_steps ? max_prio[0] : 0.0f;
}

// level 0: one workgroup per 64 chunks (4,096 priorities).  p^alpha is evaluated with all 256 threads (16 entries each, coalesced), the
// chunk sums are then taken by 64 threads in index order from the LDS image (row stride 65: conflict-free column walks).
__global__ void __launch_bounds__(256) per_sums0_kernel(const float* __restrict__ prio, long long n, float alpha, double* __restrict__ s0, double* __restrict__ a0) {
    __shared__ float pv[PER_CHUNK][PER_CHUNK + 1], pa[PER_CHUNK][PER_CHUNK + 1];
    const long long base = (long long)blockIdx.x * PER_CHUNK * PER_CHUNK;
#pragma unroll
    for (int it = 0; it < 16; ++it) {
        const int e = it * 256 + threadIdx.x;                  // entry within the workgroup's 4,096
        const long long i = base + e;
        const float p = i < n ? prio[i] : 0.0f;
        pv[e >> 6][e & 63] = p; pa[e >> 6][e & 63] = i < n ? per_pow(p, alpha) : 0.0f;
    }
    __syncthreads();
    const long long k = (long long)blockIdx.x * PER_CHUNK + threadIdx.x;
    if (threadIdx.x < PER_CHUNK && k < per_n0(n)) {
        const long long lo = k * PER_CHUNK;
        const int cnt = (int)(lo + PER_CHUNK < n ? PER_CHUNK : n - lo);
        double sum = 0.0, sa = 0.0;
        for (int j = 0; j < cnt; ++j) { sum += (double)pv[threadIdx.x][j]; sa += (double)pa[threadIdx.x][j]; }
        s0[k] = sum; a0[k] = sa;
    }
}

// level 1 + totals: thread m sums its 64 level-0 values in index order (32 loads of each array in flight at a time), thread 0 then the
// level-1 values.  One workgroup; n1 <= PER_MAX_L1 (capacity <= 4M entries).
#define PER_MAX_L1 1024
__global__ void __launch_bounds__(256) per_sums1_kernel(const double* __restrict__ s0, const double* __restrict__ a0, long long n0, double* __restrict__ s1,
                                                        double* __restrict__ a1out, double* __restrict__ totals) {
    __shared__ double l1[PER_MAX_L1], a1[PER_MAX_L1];
    const long long n1 = per_n0(n0);
    for (long long m = threadIdx.x; m < n1; m += 256) {
        const long long lo = m * PER_CHUNK;
        const int cnt = (int)(lo + PER_CHUNK < n0 ? PER_CHUNK : n0 - lo);
        double sum = 0.0, sa = 0.0;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            double v[32], va[32];
#pragma unroll
            for (int j = 0; j < 32; ++j) { v[j] = 32 * h + j < cnt ? s0[lo + 32 * h + j] : 0.0; va[j] = 32 * h + j < cnt ? a0[lo + 32 * h + j] : 0.0; }
#pragma unroll
            for (int j = 0; j < 32; ++j) if (32 * h + j < cnt) { sum += v[j]; sa += va[j]; }
        }
        s1[m] = sum; l1[m] = sum; a1[m] = sa; a1out[m] = sa;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0, ta = 0.0;
#pragma unroll 16
        for (long long m2 = 0; m2 < n1; ++m2) { t += l1[m2]; ta += a1[m2]; }
        totals[0] = t; totals[1] = ta;
    }
}

// Sixteen steps of a prefix-sum walk (the sampler's contract: while the remainder x is >= the next value, subtract it and advance — sequential, in index order, in
// double) for the lanes whose bit is set in `alive`; x, the advance count and `alive` come back updated.  The step is predicated on EXEC instead of selected: v_cmpx
// narrows EXEC to the lanes that go on, the subtraction and the count then simply do not happen in the others — 3 vector instructions per step against the ~14 the
// compiler makes of `go = go & (x >= v); x = go ? x - v : x; m += go` (compare to a scalar mask, mask arithmetic, two selects for the double, a select and an add for the
// count, and scalar-register spills): -DPER_STAMPS measured 73 cycles per step for that form, the whole walk being one wave's instruction issue.  The same subtractions
// of the same values in the same order.  EXEC is restored before the block ends.
__device__ __forceinline__ void per_walk16(double& x, int& cnt, unsigned long long& alive, const double (&v)[16]) {
    unsigned long long save;
    asm volatile(
        "s_mov_b64 %[sv], exec\n\ts_and_b64 exec, exec, %[al]\n\t"
        "v_cmpx_ge_f64_e32 vcc, %[x], %[v0]\n\tv_add_f64 %[x], %[x], -%[v0]\n\tv_add_u32_e32 %[c], 1, %[c]\n\t"
        "v_cmpx_ge_f64_e32 vcc, %[x], %[v1]\n\tv_add_f64 %[x], %[x], -%[v1]\n\tv_add_u32_e32 %[c], 1, %[c]\n\t"
        "v_cmpx_ge_f64_e32 vcc, %[x], %[v2]\n\tv_add_f64 %[x], %[x], -%[v2]\n\tv_add_u32_e32 %[c], 1, %[c]\n\t"
        "v_cmpx_ge_f64_e32 vcc, %[x], %[v3]\n\tv_add_f64 %[x], %[x], -%[v3]\n\tv_add_u32_e32 %[c], 1, %[c]\n\t"
        "v_cmpx_ge_f64_e32 vcc, %[x], %[v4]\n\tv_add_f64 %[x], %[x], -%[v4]\n\tv_add_u32_e32 %[c], 1, %[c]\n\t"
        "v_cmpx_ge_f64_e32 vcc, %[x], %[v5]\n\tv_add_f64 %[x], %[x], -%[v5]\n\tv_add_u32_e32 %[c], 1, %[c]\n\t"
        "v_cmpx_ge_f64_e32 vcc, %[x], %[v6]\n\tv_add_f64 %[x], %[x], -%[v6]\n\tv_add_u32_e32 %[c], 1, %[c]\n\t"
        "v_cmpx_ge_f64_e32 vcc, %[x], %[v7]\n\tv_add_f64 %[x], %[x], -%[v7]\n\tv_add_u32_e32 %[c], 1, %[c]\n\t"
        "v_cmpx_ge_f64_e32 vcc, %[x], %[v8]\n\tv_add_f64 %[x], %[x], -%[v8]\n\tv_add_u32_e32 %[c], 1, %[c]\n\t"
        "v_cmpx_ge_f64_e32 vcc, %[x], %[v9]\n\tv_add_f64 %[x], %[x], -%[v9]\n\tv_add_u32_e32 %[c], 1, %[c]\n\t"
        "v_cmpx_ge_f64_e32 vcc, %[x], %[v10]\n\tv_add_f64 %[x], %[x], -%[v10]\n\tv_add_u32_e32 %[c], 1, %[c]\n\t"
        "v_cmpx_ge_f64_e32 vcc, %[x], %[v11]\n\tv_add_f64 %[x], %[x], -%[v11]\n\tv_add_u32_e32 %[c], 1, %[c]\n\t"
        "v_cmpx_ge_f64_e32 vcc, %[x], %[v12]\n\tv_add_f64 %[x], %[x], -%[v12]\n\tv_add_u32_e32 %[c], 1, %[c]\n\t"
        "v_cmpx_ge_f64_e32 vcc, %[x], %[v13]\n\tv_add_f64 %[x], %[x], -%[v13]\n\tv_add_u32_e32 %[c], 1, %[c]\n\t"
        "v_cmpx_ge_f64_e32 vcc, %[x], %[v14]\n\tv_add_f64 %[x], %[x], -%[v14]\n\tv_add_u32_e32 %[c], 1, %[c]\n\t"
        "v_cmpx_ge_f64_e32 vcc, %[x], %[v15]\n\tv_add_f64 %[x], %[x], -%[v15]\n\tv_add_u32_e32 %[c], 1, %[c]\n\t"
        "s_mov_b64 %[al], exec\n\ts_mov_b64 exec, %[sv]"
        : [x] "+v"(x), [c] "+v"(cnt), [al] "+s"(alive), [sv] "=&s"(save)
        : [v0] "v"(v[0]), [v1] "v"(v[1]), [v2] "v"(v[2]), [v3] "v"(v[3]), [v4] "v"(v[4]), [v5] "v"(v[5]), [v6] "v"(v[6]), [v7] "v"(v[7]), [v8] "v"(v[8]), [v9] "v"(v[9]), [v10] "v"(v[10]), [v11] "v"(v[11]), [v12] "v"(v[12]), [v13] "v"(v[13]), [v14] "v"(v[14]), [v15] "v"(v[15])
        : "vcc");
}
#ifdef PER_STAMPS   // diagnostic build: the phases of per_sample_kernel on the wall clock (s_memrealtime, 100 MHz), thread 0; tools/per_sample_stamps.py
__device__ unsigned long long per_mark_dbg[16];
#define PER_MARK(k) do { __builtin_amdgcn_sched_barrier(0); if (threadIdx.x == 0) { unsigned long long rt_; \
                         asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(rt_) :: "memory"); per_mark_dbg[k] = rt_; } \
                         __builtin_amdgcn_sched_barrier(0); } while (0)
extern "C" int mi_debug_per_sample_marks(unsigned long long* out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(per_mark_dbg), sizeof(unsigned long long) * 16) == hipSuccess ? 0 : -2;
}
#else
#define PER_MARK(k) do {} while (0)
#endif
// indices (sample != 0) by the prefix-sum descent, then the importance weights of per.py:131,145-146, normalised by their maximum
__global__ void __launch_bounds__(256)
per_sample_kernel(uint64_t seed, uint64_t update, const float* __restrict__ prio, long long n, const double* __restrict__ s0, const double* __restrict__ a0,
                  const double* __restrict__ s1, const double* __restrict__ totals, int batch, float count, float alpha, float beta, int sample,
                  int64_t* __restrict__ idx, float* __restrict__ weights) {
    __shared__ float wmax[16];
    __shared__ double l1s[PER_MAX_L1];                      // the level-1 sums as staged
    __shared__ double pre[PER_MAX_L1];                      // ... and their running sums: searched by every draw
    __shared__ double tot[2];
    // wave-private transposition buffer: the 64 values a draw walks at level 0 (and then its 64 priorities) are 512 (256) contiguous bytes, but a lane per draw
    // fetching them itself makes every load instruction touch 64 different lines (~64 cycles of the CU's address unit each, 64 instructions: -DPER_STAMPS showed the
    // "round trip" at 3.5 us).  The wave fetches draw d's values with ONE coalesced instruction (chunk start from lane d by v_readlane), parks them in row d, and every
    // lane then reads its own row (row stride 65: conflict-free for floats, two-way for doubles).
    __shared__ double stg[4][PER_CHUNK][PER_CHUNK + 1];
    const int lane = threadIdx.x & 63, wv = (threadIdx.x >> 6) & 3;
    const long long n0 = per_n0(n), n1 = per_n0(n0);
    PER_MARK(0);
    // Level 1 is staged and its running sums P[j] = s1[0] + ... + s1[j] (sequential, f64: the chain the total always was — per_sums1_kernel's order) go to an array of their
    // own (in place the loop was a load -> add -> store chain through LDS: 3.6 us instead of 2.1 for the prologue);
    // a draw then finds its level-1 group by BINARY SEARCH over P (round 6: the contract's level-1 step, see ref_per_sample) instead of walking up to 256 dependent
    // subtractions: 4.8 of the launch's 13 us.
    {
        __shared__ double a1s[PER_MAX_L1];
        for (long long m = threadIdx.x; m < n1; m += blockDim.x) { l1s[m] = s1[m]; if (a0) a1s[m] = a0[m]; }
        __syncthreads();
        if (threadIdx.x == 0) {
            double t = 0.0;
#pragma unroll 16
            for (long long m2 = 0; m2 < n1; ++m2) { t += l1s[m2]; pre[m2] = t; }
            tot[0] = t;
        } else if (threadIdx.x == 64) {   // sum of p^alpha: the incremental form keeps its level-1 sums current in memory (a0 here), the full-pass form has the total
            double t = 0.0;
            if (a0) {
#pragma unroll 16
                for (long long m2 = 0; m2 < n1; ++m2) t += a1s[m2];
            } else t = totals[1];
            tot[1] = t;
        }
    }
    __syncthreads();
    PER_MARK(1);   // level 1 staged as running sums, totals taken
    const double total = tot[0];
    const float total_alpha = (float)tot[1];
    float mx = 0.0f;
    for (int b = threadIdx.x; b < batch; b += blockDim.x) {
        long long i;
        if (sample) {
            uint32_t r[4];
            mi_philox(seed, update, (uint64_t)b, STREAM_PER, r);
            const double u = ((double)(r[0] >> 5) * 67108864.0 + (double)(r[1] >> 6)) / 9007199254740992.0;
            double x = u * total;
            PER_MARK(2);   // keyed draw
            // the three walks subtract in index order (the contract); each level's 64 values are requested together, so a draw costs
            // three memory round trips instead of up to 384 dependent ones.  32-bit indices (capacity <= 4M entries), no guarded reads and
            // no short-circuit conditions: every guard of this loop nest used to be a branch or a 64-bit scalar compare held in SGPRs
            // (the kernel spilled them to VGPR lanes); now a step is compare + subtract + select on the vector unit.
            const int n_i = (int)n, n0_i = (int)n0, n1_i = (int)n1;
            // level 1: m = the number of running sums P[j], j < n1 - 1, with x >= P[j] (P is non-decreasing: priorities are >= 0), by a fixed-depth binary search
            // — ten dependent LDS reads at most (n1 <= 1024) — and the remainder against P[m - 1]
            int m = 0;
            bool go = true;
            {
                int lo = 0, hi = n1_i - 1;
#pragma unroll
                for (int it = 0; it < 10; ++it) {
                    const int mid = (lo + hi) >> 1;
                    const bool live = lo < hi, up = live && x >= pre[mid];
                    lo = up ? mid + 1 : lo;
                    hi = (live && !up) ? mid : hi;
                }
                m = lo;
                if (m > 0) x = x - pre[m - 1];
            }
            PER_MARK(3);   // level-1 group found
            int k = m * PER_CHUNK;
            {
                const int cnt = k + PER_CHUNK < n0_i ? PER_CHUNK : n0_i - k;
                double v[PER_CHUNK];
                const bool coop = blockDim.x <= 256 && __ballot(true) == ~0ull && __all(cnt == PER_CHUNK);   // (wave-uniform) a full wave of draws on whole chunks
                if (coop) {
                    double (*row)[PER_CHUNK + 1] = stg[wv];
#pragma unroll
                    for (int d = 0; d < PER_CHUNK; ++d) row[d][lane] = s0[__builtin_amdgcn_readlane(k, d) + lane];
#pragma unroll
                    for (int j = 0; j < PER_CHUNK; ++j) v[j] = row[lane][j];
                } else {
#pragma unroll
                    for (int j = 0; j < PER_CHUNK; ++j) v[j] = s0[k + j < n0_i ? k + j : n0_i - 1];
                }
                if (__all(cnt == PER_CHUNK)) {                         // (wave-uniform) whole chunks: the walk may pass 63 of the 64 values
                    v[PER_CHUNK - 1] = __builtin_inf();
                    unsigned long long al = __ballot(true);
#pragma unroll
                    for (int q4 = 0; q4 < PER_CHUNK / 16; ++q4) {
                        const double (&vq)[16] = *reinterpret_cast<const double (*)[16]>(&v[16 * q4]);
                        per_walk16(x, k, al, vq);
                    }
                } else {
                    go = true;
                    double rr = x;
#pragma unroll
                    for (int jj = 0; jj < PER_CHUNK - 1; ++jj) { go = go & (jj + 1 < cnt) & (rr >= v[jj]); rr = rr - v[jj]; x = go ? rr : x; k += go ? 1 : 0; }
                }
            }
            PER_MARK(4);   // level-0 sums: round trip + walk
            int ii = k * PER_CHUNK;
            {
                const int cnt = ii + PER_CHUNK < n_i ? PER_CHUNK : n_i - ii;
                float v[PER_CHUNK];
                const bool coop = blockDim.x <= 256 && __ballot(true) == ~0ull && __all(cnt == PER_CHUNK);
                if (coop) {
                    float (*row)[PER_CHUNK + 1] = reinterpret_cast<float (*)[PER_CHUNK + 1]>(&stg[wv][0][0]);
#pragma unroll
                    for (int d = 0; d < PER_CHUNK; ++d) row[d][lane] = prio[__builtin_amdgcn_readlane(ii, d) + lane];
#pragma unroll
                    for (int j = 0; j < PER_CHUNK; ++j) v[j] = row[lane][j];
                } else {
#pragma unroll
                    for (int j = 0; j < PER_CHUNK; ++j) v[j] = prio[ii + j < n_i ? ii + j : n_i - 1];
                }
                if (__all(cnt == PER_CHUNK)) {
                    unsigned long long al = __ballot(true);
#pragma unroll
                    for (int q4 = 0; q4 < PER_CHUNK / 16; ++q4) {
                        double vq[16];
#pragma unroll
                        for (int j = 0; j < 16; ++j) vq[j] = (double)v[16 * q4 + j];
                        if (q4 == PER_CHUNK / 16 - 1) vq[15] = __builtin_inf();
                        per_walk16(x, ii, al, vq);
                    }
                } else {
                    go = true;
                    double rr = x;
#pragma unroll
                    for (int jj = 0; jj < PER_CHUNK - 1; ++jj) { const double vd = (double)v[jj]; go = go & (jj + 1 < cnt) & (rr >= vd); rr = rr - vd; x = go ? rr : x; ii += go ? 1 : 0; }
                }
            }
            i = ii;
            PER_MARK(5);   // priorities: round trip + walk
            while (i > 0 && prio[i] == 0.0f) --i;
            idx[b] = i;
            PER_MARK(6);   // zero-skip
        } else i = idx[b];
        const float prob = per_pow(prio[i], alpha) / total_alpha;
        const float w = powf(count * prob, -beta);
        weights[b] = w;
        mx = fmaxf(mx, w);
    }
    PER_MARK(7);       // weights
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = mx;
    __syncthreads();
    mx = wmax[0];
    for (int k = 1; k < (int)(blockDim.x >> 6); ++k) mx = fmaxf(mx, wmax[k]);
    for (int b = threadIdx.x; b < batch; b += blockDim.x) weights[b] = weights[b] / mx;
    PER_MARK(8);       // normalised
}

// per_scatter_role as a launch of its own (mi_per_update_priorities)
__global__ void __launch_bounds__(1024) per_scatter_kernel(float* __restrict__ prio, const int64_t* __restrict__ idx, const float* __restrict__ td_abs, int batch,
                                                           int32_t* __restrict__ owner, float* __restrict__ max_prio) {
    __shared__ float wmax[16];
    per_scatter_role(wmax, per_scatter_t{prio, idx, td_abs, batch, owner, max_prio});
}

extern "C" int mi_per_mark(float* priorities, int n_envs, int64_t slots, int64_t global_step, int n_steps, const float* max_priority, void* stream) {
    MI_CHECK_ARG(priorities && max_priority && n_envs > 0 && slots >= 2 && n_steps > 0 && n_steps < slots && global_step >= 0, "bad arguments");
    const long long total = (long long)(n_steps + 1) * n_envs;
    per_mark_kernel<<<(unsigned)((total + 255) / 256), 256, 0, (hipStream_t)stream>>>(priorities, n_envs, (long long)slots, (long long)global_step, n_steps, max_priority);
    MI_LAUNCH_CHECK();
    return MI_OK;
}

// ---- incremental form: the level-0 sums are kept CURRENT by the two calls that change priorities ------------------------------------
// A full pass over 1M priorities per update (per_sums0 + per_sums1 + the sampler: three dependent launches, 65 us) is what the reference's
// O(buffer) torch.multinomial becomes when restated naively.  But between two draws only (a) the rows an acting call has just written and
// the write head and (b) the <= batch entries the last update re-prioritised have changed, and a chunk sum depends on its own 64 entries
// only.  So per_mark_sums_kernel and per_scatter_sums_kernel RECOMPUTE (never adjust) the chunks they touch — same operations in the same
// order as per_sums0_kernel, hence bit-identical sums — and per_sample_kernel takes level 1 and the totals into its own prologue.
// Invariant used: an entry that was never written holds 0 (the priorities ring starts zero-filled) and contributes +0.0 to its chunk, so
// chunk sums over whole chunks equal the contract's sums over prio[:n_valid].
#define PS_ROWS 128   // batch rows whose chunks per_scatter_sums_kernel recomputes per pass (2 x 33 KB of LDS)
// per_mark_role as a launch of its own (mi_per_mark_sums): one workgroup of 1024 threads per LEVEL-1 GROUP (64 chunks = 4,096 entries) that the touched rows reach
__global__ void __launch_bounds__(1024) per_mark_sums_kernel(per_ride_t r) {
    __shared__ per_mark_smem sm;
    per_mark_role<1024>(sm, (int)blockIdx.x, r);
}

// per_scatter_role + per_owed_sums_role as ONE launch of one workgroup (mi_per_update_priorities_sums): the scatter and the sums of every chunk it touched
__global__ void __launch_bounds__(1024) per_scatter_sums_kernel(per_scatter_t sc, long long capacity, float alpha, double* __restrict__ s0, double* __restrict__ a0,
                                                                double* __restrict__ s1, double* __restrict__ a1) {
    __shared__ float wmax[16];
    __shared__ __attribute__((aligned(16))) unsigned char raw[2 * PS_ROWS * (PER_CHUNK + 1) * sizeof(double)];   // phase 1: two float images; phase 2: two double images
    per_scatter_role(wmax, sc);   // (its barriers make every priority store of this workgroup visible to the loads below)
    per_owed_sums_role<1024, PS_ROWS>(raw, nullptr, 0, 1, sc.prio, sc.idx, sc.batch, capacity, alpha, s0, a0, s1, a1, nullptr);
}
// per_owed_sums_role alone (mi_per_settle_sums: the sums a one-call update left owed, when no acting launch came to carry them)
__global__ void __launch_bounds__(1024) per_owed_sums_kernel(const float* __restrict__ prio, const int64_t* __restrict__ idx, int batch, long long capacity, float alpha,
                                                             double* __restrict__ s0, double* __restrict__ a0, double* __restrict__ s1, double* __restrict__ a1) {
    __shared__ __attribute__((aligned(16))) unsigned char raw[2 * PS_ROWS * (PER_CHUNK + 1) * sizeof(double)];
    per_owed_sums_role<1024, PS_ROWS>(raw, nullptr, 0, 1, prio, idx, batch, capacity, alpha, s0, a0, s1, a1, nullptr);
}

static int per_launch_sums(const float* priorities, int64_t n_valid, float alpha, const per_ws_t& w, hipStream_t s) {
    const int64_t n0 = per_n0(n_valid);
    per_sums0_kernel<<<(unsigned)((n0 + PER_CHUNK - 1) / PER_CHUNK), 256, 0, s>>>(priorities, (long long)n_valid, alpha, w.s0, w.a0);
    MI_LAUNCH_CHECK();
    per_sums1_kernel<<<1, 256, 0, s>>>(w.s0, w.a0, (long long)n0, w.s1, w.a1, w.totals);
    MI_LAUNCH_CHECK();
    return MI_OK;
}

extern "C" int mi_per_sample(uint64_t seed, uint64_t update_index, const float* priorities, int64_t n_valid, int64_t capacity, double count, float alpha,
                             float beta, int batch, int sample, void* workspace, int64_t* idx, float* weights, void* stream) {
    MI_CHECK_ARG(priorities && workspace && idx && weights && n_valid > 0 && n_valid <= capacity && batch > 0, "bad arguments");
    MI_CHECK_ARG(capacity <= (int64_t)PER_MAX_L1 * PER_CHUNK * PER_CHUNK, "prioritized sampler: capacity above 4,194,304 entries needs a fourth level");
    const per_ws_t w = per_ws(workspace, capacity);
    mi_prof_scope prof(MI_PROF_PER, (hipStream_t)stream);   // the sums launches and the sampler as one bracket
    const int rc = per_launch_sums(priorities, n_valid, alpha, w, (hipStream_t)stream);
    if (rc) return rc;
    per_sample_kernel<<<1, 256, 0, (hipStream_t)stream>>>(seed, update_index, priorities, (long long)n_valid, w.s0, nullptr, w.s1, w.totals, batch, (float)count, alpha,
                                                          beta, sample, idx, weights);
    MI_LAUNCH_CHECK();
    return MI_OK;
}

// ---- the incremental entry points (see per_mark_sums_kernel) ----
extern "C" int mi_per_sums_refresh(const float* priorities, int64_t capacity, float alpha, void* workspace, void* stream) {
    MI_CHECK_ARG(priorities && workspace && capacity > 0, "bad arguments");
    const per_ws_t w = per_ws(workspace, capacity);
    const int64_t n0 = per_n0(capacity);
    per_sums0_kernel<<<(unsigned)((n0 + PER_CHUNK - 1) / PER_CHUNK), 256, 0, (hipStream_t)stream>>>(priorities, (long long)capacity, alpha, w.s0, w.a0);
    MI_LAUNCH_CHECK();
    per_sums1_kernel<<<1, 256, 0, (hipStream_t)stream>>>(w.s0, w.a0, (long long)n0, w.s1, w.a1, w.totals);
    MI_LAUNCH_CHECK();
    return MI_OK;
}

// the acting call's marks as a per_ride_t; false: the touched rows (nearly) cover the ring — mark, then one full pass (the caller's business)
static bool per_make_ride(per_ride_t* r, float* priorities, int n_envs, int64_t slots, int64_t global_step, int n_steps, const float* max_priority, float alpha,
                          void* workspace) {
    const int64_t capacity = slots * n_envs, touched = (int64_t)(n_steps + 1) * n_envs;
    if (touched + 2 * (int64_t)PER_CHUNK * PER_CHUNK > capacity) return false;
    const per_ws_t w = per_ws(workspace, capacity);
    memset(r, 0, sizeof(*r));
    const int64_t lo = (global_step % slots) * n_envs, hi = lo + touched;
    r->mk.a[0] = lo; r->mk.b[0] = hi < capacity ? hi : capacity;
    r->mk.a[1] = 0; r->mk.b[1] = hi > capacity ? hi - capacity : 0;
    constexpr int64_t G = (int64_t)PER_CHUNK * PER_CHUNK;   // entries per level-1 group = per workgroup
    auto nblocks = [](int64_t a, int64_t b) -> int { return b <= a ? 0 : (int)((b - 1) / G - a / G + 1); };
    r->mk.nb0 = nblocks(r->mk.a[0], r->mk.b[0]);
    r->n_mark = r->mk.nb0 + nblocks(r->mk.a[1], r->mk.b[1]);
    r->prio = priorities; r->max_prio = max_priority; r->s0 = w.s0; r->a0 = w.a0; r->s1 = w.s1; r->a1 = w.a1;
    r->capacity = capacity; r->slots = slots; r->gs = global_step; r->alpha = alpha; r->N = n_envs; r->n_steps = n_steps;
    return true;
}

extern "C" int mi_per_mark_sums(float* priorities, int n_envs, int64_t slots, int64_t global_step, int n_steps, const float* max_priority, float alpha, void* workspace,
                                void* stream) {
    MI_CHECK_ARG(priorities && max_priority && workspace && n_envs > 0 && slots >= 2 && n_steps > 0 && n_steps < slots && global_step >= 0, "bad arguments");
    per_ride_t r;
    if (!per_make_ride(&r, priorities, n_envs, slots, global_step, n_steps, max_priority, alpha, workspace)) {   // the touched rows (nearly) cover the ring: mark, then one full pass
        int rc = mi_per_mark(priorities, n_envs, slots, global_step, n_steps, max_priority, stream);
        if (rc) return rc;
        return mi_per_sums_refresh(priorities, slots * n_envs, alpha, workspace, stream);
    }
    mi_prof_scope prof(MI_PROF_PER, (hipStream_t)stream);
    per_mark_sums_kernel<<<r.n_mark, 1024, 0, (hipStream_t)stream>>>(r);
    MI_LAUNCH_CHECK();
    return MI_OK;
}

extern "C" int mi_per_sample_current(uint64_t seed, uint64_t update_index, const float* priorities, int64_t n_valid, int64_t capacity, double count, float alpha,
                                     float beta, int batch, int sample, void* workspace, int64_t* idx, float* weights, void* stream) {
    MI_CHECK_ARG(priorities && workspace && idx && weights && n_valid > 0 && n_valid <= capacity && batch > 0, "bad arguments");
    MI_CHECK_ARG(capacity <= (int64_t)PER_MAX_L1 * PER_CHUNK * PER_CHUNK, "prioritized sampler: capacity above 4,194,304 entries needs a fourth level");
    const per_ws_t w = per_ws(workspace, capacity);
    mi_prof_scope prof(MI_PROF_PER, (hipStream_t)stream);
    per_sample_kernel<<<1, 256, 0, (hipStream_t)stream>>>(seed, update_index, priorities, (long long)n_valid, w.s0, w.a1, w.s1, nullptr, batch, (float)count, alpha, beta,
                                                          sample, idx, weights);
    MI_LAUNCH_CHECK();
    return MI_OK;
}

extern "C" int mi_per_update_priorities_sums(float* priorities, const int64_t* idx, const float* td_abs, int batch, int32_t* owner, float* max_priority, int64_t capacity,
                                             float alpha, void* workspace, void* stream) {
    MI_CHECK_ARG(priorities && idx && td_abs && owner && max_priority && workspace && batch > 0 && capacity > 0, "bad arguments");
    const per_ws_t w = per_ws(workspace, capacity);
    mi_prof_scope prof(MI_PROF_PER, (hipStream_t)stream);
    per_scatter_sums_kernel<<<1, 1024, 0, (hipStream_t)stream>>>(per_scatter_t{priorities, idx, td_abs, batch, owner, max_priority}, (long long)capacity, alpha, w.s0, w.a0, w.s1,
                                                                 w.a1);
    MI_LAUNCH_CHECK();
    return MI_OK;
}

extern "C" int mi_per_td_grad(const float* params, const float* target_params, const float* observations, const int64_t* actions,
                              const float* rewards, const uint8_t* terminated, const int64_t* idx, int batch, int n_envs, int64_t slots,
                              float gamma, double inv_count, const float* weights, float* td_abs, void* workspace, float* grads, float* loss, void* stream) {
    MI_CHECK_ARG(weights && td_abs, "NULL weights / td_abs");
    return dqn_td_impl(params, target_params, observations, actions, rewards, terminated, idx, batch, n_envs, slots, gamma, inv_count, workspace, grads, loss,
                       weights, td_abs, dqn_no_opt(), 0, 0, 0, stream);
}

extern "C" int mi_per_update_priorities(float* priorities, const int64_t* idx, const float* td_abs, int batch, int32_t* owner, float* max_priority, void* stream) {
    MI_CHECK_ARG(priorities && idx && td_abs && owner && max_priority && batch > 0, "bad arguments");
    per_scatter_kernel<<<1, 1024, 0, (hipStream_t)stream>>>(priorities, idx, td_abs, batch, owner, max_priority);
    MI_LAUNCH_CHECK();
    return MI_OK;
}


// ---- PER, one call per piece of the loop (round 6; the roles are at the top of the file) ----
// mi_dqn_act_steps2 / mi_dqn_act_steps (zero_next NULL) + mi_per_mark_sums in ONE launch, which also carries the chunk sums the last mi_per_td_update left owed
// (owed_idx = that update's batch indices, NULL: nothing owed).
extern "C" int mi_per_act_steps(void* handle, const float* params, int n_steps, int64_t global_step, int64_t slots, int64_t learning_starts, double start_e, double end_e,
                                double exploration_fraction, int64_t total_timesteps, float* obs_cur, float* observations, int64_t* actions, float* rewards,
                                uint8_t* terminated, const int64_t* forced_actions, const double* forced_resets, mi_episode_t* episodes, int32_t* episode_stats, int max_ep,
                                int32_t* zero_next, float* priorities, const float* max_priority, float alpha, void* per_workspace, const int64_t* owed_idx, int owed_batch,
                                void* stream) {
    MI_CHECK_ARG(handle && priorities && max_priority && per_workspace, "NULL pointer");
    MI_CHECK_ARG(n_steps > 0 && n_steps < slots && global_step >= 0 && slots >= 2, "n_steps must be in [1, slots)");
    MI_CHECK_ARG(!owed_idx || owed_batch > 0, "owed_batch must be positive when sums are owed");
    MI_CHECK_ARG(!zero_next || zero_next != episode_stats, "zero_next must be a second statistics buffer");
    const int n_envs = ((mi_env*)handle)->n;
    per_ride_t r;
    const bool inc = per_make_ride(&r, priorities, n_envs, slots, global_step, n_steps, max_priority, alpha, per_workspace);
    if (owed_idx && (!inc || owed_batch > PER_OWED_LIST)) {   // nothing to ride on (or a batch beyond the riding workgroups' row lists): the owed sums first, as a launch of their own
        const int rc0 = inc ? mi_per_settle_sums(priorities, owed_idx, owed_batch, slots * n_envs, alpha, per_workspace, stream) : MI_OK;   // (!inc: the full pass below settles them)
        if (rc0) return rc0;
        owed_idx = nullptr;
    }
    if (inc) { r.owed_idx = owed_idx; r.owed_batch = owed_idx ? owed_batch : 0; }
    int rc = dqn_act_impl(handle, params, n_steps, global_step, slots, learning_starts, start_e, end_e, exploration_fraction, total_timesteps, obs_cur, observations,
                          actions, rewards, terminated, forced_actions, forced_resets, episodes, episode_stats, max_ep, zero_next, zero_next == nullptr, stream, inc ? &r : nullptr);
    if (rc || inc) return rc;
    // the touched rows (nearly) cover the ring: mark, then one full pass over the priorities (which settles whatever was owed)
    rc = mi_per_mark(priorities, n_envs, slots, global_step, n_steps, max_priority, stream);
    if (rc) return rc;
    return mi_per_sums_refresh(priorities, slots * n_envs, alpha, per_workspace, stream);
}

// the chunk sums a mi_per_td_update left owed, as a launch of their own (when the next call is not an acting call: a second update, a checkpoint, a test reading the sums)
extern "C" int mi_per_settle_sums(const float* priorities, const int64_t* idx, int batch, int64_t capacity, float alpha, void* per_workspace, void* stream) {
    MI_CHECK_ARG(priorities && idx && per_workspace && batch > 0 && capacity > 0, "bad arguments");
    const per_ws_t w = per_ws(per_workspace, capacity);
    mi_prof_scope prof(MI_PROF_PER, (hipStream_t)stream);
    per_owed_sums_kernel<<<1, 1024, 0, (hipStream_t)stream>>>(priorities, idx, batch, (long long)capacity, alpha, w.s0, w.a0, w.s1, w.a1);
    MI_LAUNCH_CHECK();
    return MI_OK;
}

// ONE optimisation step of per.py:126-153 in one call, three launches: the sampler (mi_per_sample_current), the weighted TD launch, and the slab sum + Adam launch with
// the priority scatter + max_priority on an extra workgroup (mi_dqn_td_update with weights + the first half of mi_per_update_priorities_sums).  The chunk sums of the
// scattered entries stay OWED: the caller hands `idx` to the next mi_per_act_steps (which carries them) or calls mi_per_settle_sums before anything reads the sums.
extern "C" int mi_per_td_update(float* params, const float* target_params, const float* observations, const int64_t* actions, const float* rewards,
                                const uint8_t* terminated, int64_t* idx, int batch, int n_envs, int64_t slots, float gamma, float* weights, float* td_abs, void* workspace,
                                float* grads, float* loss, float* exp_avg, float* exp_avg_sq, int64_t step, double lr, double beta1, double beta2, double eps, uint64_t seed,
                                uint64_t update_index, float* priorities, int64_t n_valid, double count, float alpha, float beta, int sample, void* per_workspace,
                                int32_t* owner, float* max_priority, void* stream) {
    MI_CHECK_ARG(params && exp_avg && exp_avg_sq && step >= 1 && weights && td_abs && idx && grads, "bad optimizer state / NULL pointer");
    MI_CHECK_ARG(priorities && per_workspace && owner && max_priority && n_valid > 0 && n_valid <= slots * n_envs && batch > 0, "bad PER arguments");
    int rc = mi_per_sample_current(seed, update_index, priorities, n_valid, slots * n_envs, count, alpha, beta, batch, sample, per_workspace, idx, weights, stream);
    if (rc) return rc;
    dqn_opt_t o;
    const double bc1 = 1.0 - pow(beta1, (double)step), bc2 = 1.0 - pow(beta2, (double)step);
    o.params = params; o.m = exp_avg; o.v = exp_avg_sq; o.w1 = (float)(1.0 - beta1); o.b2 = (float)beta2; o.w2 = (float)(1.0 - beta2);
    o.step_size = (float)(lr / bc1); o.rbc2 = (float)(1.0 / sqrt(bc2)); o.eps = (float)eps; o.du_params = nullptr; o.du_grads = nullptr;
    const per_scatter_t sc = {priorities, idx, td_abs, batch, owner, max_priority};
    return dqn_td_impl(params, target_params, observations, actions, rewards, terminated, idx, batch, n_envs, slots, gamma, 1.0 / batch, workspace, grads, loss,
                       weights, td_abs, o, 0, 0, 0, stream, &sc);
}

MI_INSIDE_EXPORT(dqn)
