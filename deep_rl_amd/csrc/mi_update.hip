// mi_update.hip — everything after the rollout in reference ppo.py:144-195:
//   gae_kernel            GAE reverse scan                                   (ppo.py:144-151)
//   perm_kernel           minibatch permutation (replaces np.random.permutation, :155)
//   adv_stats_kernel      per-minibatch sum / sum of squares of advantages   (:169)
//   grad_kernel<ACTOR>    forward + loss + backward of one minibatch on the f32 MFMA pipe (:166-190)
//   grad_reduce_kernel    deterministic reduction of the per-workgroup partial gradients
//   clip_adam_kernel      clip_grad_norm_ + Adam                             (:191-192)
//   explained_var_kernel                                                     (:194-195)
#include "mi_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// =====================================================================================================
// GAE: one lane per env walks t = T-1..0 over (T+1, N) time-major buffers (coalesced over envs at every t).
// Expression order is the reference's, fp32, no contraction (this file is built with -ffp-contract=off).
// =====================================================================================================
__global__ void __launch_bounds__(64) gae_kernel(const float* __restrict__ rewards, const float* __restrict__ dones,
                                                 const float* __restrict__ values, int T, int N, float gamma, float lam,
                                                 float* __restrict__ adv, float* __restrict__ returns) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    float last = 0.0f;
    float vnext = values[(size_t)T * N + i];
    adv[(size_t)T * N + i] = 0.0f;
    returns[(size_t)T * N + i] = 0.0f + vnext;
#pragma unroll 8
    for (int t = T - 1; t >= 0; --t) {
        const size_t c = (size_t)t * N + i, n = c + N;
        const float vcur = values[c];
        const float a = gamma * (1.0f - dones[n]);
        const float b = vnext + lam * last;
        float v = rewards[n] + a * b;
        v = v - vcur;
        adv[c] = v;
        returns[c] = v + vcur;
        last = v;
        vnext = vcur;
    }
}

extern "C" int mi_gae(const float* rewards, const float* dones, const float* values, int T, int N, float gamma, float lam,
                      float* advantages, float* returns, void* stream) {
    MI_CHECK_ARG(rewards && dones && values && advantages && returns, "NULL pointer");
    MI_CHECK_ARG(T > 0 && N > 0, "T and N must be positive");
    mi_prof_scope prof(MI_PROF_GAE, (hipStream_t)stream);
    gae_kernel<<<(N + 63) / 64, 64, 0, (hipStream_t)stream>>>(rewards, dones, values, T, N, gamma, lam, advantages, returns);
    MI_LAUNCH_CHECK();
    return MI_OK;
}

// =====================================================================================================
// Permutation
// =====================================================================================================
__global__ void __launch_bounds__(256) perm_kernel(uint32_t n, uint32_t a, uint32_t b, uint32_t k0, uint32_t k1, int32_t* __restrict__ out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (int32_t)mi_feistel(i, n, a, b, k0, k1);
}

extern "C" int mi_make_perm(uint32_t n, uint64_t key, int32_t* out, void* stream) {
    MI_CHECK_ARG(out != nullptr, "out is NULL");
    MI_CHECK_ARG(n > 0 && n <= (1u << 30), "n out of range");
    uint32_t bits = 1;
    while ((1u << bits) < n) ++bits;
    if (bits < 2) bits = 2;
    const uint32_t a = bits / 2, b = bits - a;
    perm_kernel<<<(n + 255) / 256, 256, 0, (hipStream_t)stream>>>(n, a, b, (uint32_t)key, (uint32_t)(key >> 32), out);
    MI_LAUNCH_CHECK();
    return MI_OK;
}

extern "C" uint64_t mi_perm_key(uint64_t seed, uint64_t update, uint64_t epoch) {
    uint32_t r[4];
    mi_philox(seed, update, epoch, STREAM_PERM, r);
    return ((uint64_t)r[1] << 32) | r[0];
}

// =====================================================================================================
// Advantage statistics: sums[k] = {sum a, sum a^2, count} over idx[k*mb .. (k+1)*mb)
// =====================================================================================================
#define STATS_BLOCKS_PER_MB 32
__global__ void __launch_bounds__(256) adv_stats_kernel(const float* __restrict__ adv, const int32_t* __restrict__ idx, int mb,
                                                        double* __restrict__ sums) {
    const int k = blockIdx.x / STATS_BLOCKS_PER_MB, part = blockIdx.x % STATS_BLOCKS_PER_MB;
    const int32_t* id = idx + (size_t)k * mb;
    double s = 0.0, q = 0.0;
    for (int i = part * 256 + threadIdx.x; i < mb; i += 256 * STATS_BLOCKS_PER_MB) {
        const double a = (double)adv[id[i]];
        s += a;
        q += a * a;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { s += __shfl_xor(s, o); q += __shfl_xor(q, o); }
    __shared__ double ss[4], qq[4];
    if ((threadIdx.x & 63) == 0) { ss[threadIdx.x >> 6] = s; qq[threadIdx.x >> 6] = q; }
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicAdd(&sums[3 * k + 0], (ss[0] + ss[1]) + (ss[2] + ss[3]));
        atomicAdd(&sums[3 * k + 1], (qq[0] + qq[1]) + (qq[2] + qq[3]));
        if (part == 0) sums[3 * k + 2] = (double)mb;
    }
}

// mi_ppo_update's fused form: out[i] = Feistel(i) and, in the same pass, the advantage sums of the minibatch i belongs to
// (mb is a multiple of 256 there, so a workgroup never straddles two minibatches).  sums must be zero on entry.
#define PS_PER_BLOCK 4096  // elements per workgroup: few workgroups per minibatch touch the fp64 atomics (contention; 2048 per workgroup: 26 us instead of 16.7)
#define PS_MAX_EPOCHS 8
struct ps_epochs_t { uint32_t k0[PS_MAX_EPOCHS], k1[PS_MAX_EPOCHS]; int32_t* out[PS_MAX_EPOCHS]; };   // blockIdx.y = epoch: all epochs of an update in one launch
__global__ void __launch_bounds__(256) perm_stats_kernel(uint32_t n, uint32_t a, uint32_t b, ps_epochs_t ep, int mb, int n_mb,
                                                         const float* __restrict__ adv, double* __restrict__ sums_all) {
    MI_INSIDE_SCOPE(MI_PROF_STATS);
    const uint32_t base = blockIdx.x * PS_PER_BLOCK;
    const uint32_t k0 = ep.k0[blockIdx.y], k1 = ep.k1[blockIdx.y];
    int32_t* __restrict__ out = ep.out[blockIdx.y];
    double* __restrict__ sums = sums_all + (size_t)3 * n_mb * blockIdx.y;
    double s = 0.0, q = 0.0;
    if (base + PS_PER_BLOCK <= n) {   // (uniform) a whole block: its 16 gathers per thread in flight together — one memory latency instead of four (same sums in the same order)
        float av[PS_PER_BLOCK / 256];
#pragma unroll
        for (int u = 0; u < PS_PER_BLOCK / 256; ++u) {
            const uint32_t i = base + threadIdx.x + 256 * u;
#ifdef PS_PERM_FROM_MEMORY   // timing-only build (round 6): what the launch would cost if the permutations were generated elsewhere and only read here
            uint32_t p = (uint32_t)out[i];                              // (a slot that still holds 0 — the first update — is computed and stored once)
            if (p == 0u && i != 0u) { p = mi_feistel(i, n, a, b, k0, k1); out[i] = (int32_t)p; }
#else
            const uint32_t p = mi_feistel(i, n, a, b, k0, k1);
            out[i] = (int32_t)p;
#endif
            av[u] = adv[p];
        }
#pragma unroll
        for (int u = 0; u < PS_PER_BLOCK / 256; ++u) { const double v = (double)av[u]; s += v; q += v * v; }
    } else {
#pragma unroll 4
    for (uint32_t i = base + threadIdx.x; i < base + PS_PER_BLOCK && i < n; i += 256) {
        const uint32_t p = mi_feistel(i, n, a, b, k0, k1);
        out[i] = (int32_t)p;
        const double v = (double)adv[p];
        s += v; q += v * v;
    }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { s += __shfl_xor(s, o); q += __shfl_xor(q, o); }
    __shared__ double ss[4], qq[4];
    if ((threadIdx.x & 63) == 0) { ss[threadIdx.x >> 6] = s; qq[threadIdx.x >> 6] = q; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const int k = (int)(base / (uint32_t)mb);
        // (1,024 fp64 atomics on four cache lines per update: 3 us of this launch's 17 by a timing-only build without them, round 4 — left as they are)
        atomicAdd(&sums[3 * k + 0], (ss[0] + ss[1]) + (ss[2] + ss[3]));
        atomicAdd(&sums[3 * k + 1], (qq[0] + qq[1]) + (qq[2] + qq[3]));
        if (base % (uint32_t)mb == 0) sums[3 * k + 2] = (double)mb;
    }
}

extern "C" int mi_adv_stats(const float* advantages, const int32_t* idx, int mb, int n_mb, double* sums, void* stream) {
    MI_CHECK_ARG(advantages && idx && sums, "NULL pointer");
    MI_CHECK_ARG(mb > 0 && n_mb > 0, "mb and n_mb must be positive");
    MI_HIP(hipMemsetAsync(sums, 0, sizeof(double) * 3 * (size_t)n_mb, (hipStream_t)stream));
    mi_prof_scope prof(MI_PROF_STATS, (hipStream_t)stream);
    adv_stats_kernel<<<n_mb * STATS_BLOCKS_PER_MB, 256, 0, (hipStream_t)stream>>>(advantages, idx, mb, sums);
    MI_LAUNCH_CHECK();
    return MI_OK;
}

// =====================================================================================================
// Minibatch gradient on the f32 MFMA pipe.
//
// Work split: a workgroup (4 waves) serves ONE net — the two nets share nothing but the observations (the joint
// grad-norm clip happens later), and each CU hosts one block of each.
// A wave processes tiles of 16 minibatch rows; per tile and net (v_mfma_f32_16x16x4_f32 unless noted):
//     z1^T = W1 x^T            4 MFMA     (rows on the lanes, hidden units in the accumulator registers)
//     z2^T = W2 h1^T          64 MFMA     B operand = h1's accumulator registers as they stand
//     dh1^T = W2^T dz2^T      64 MFMA     B operand = dz2's registers as they stand
//     dW2 += dz2^T h1         64 MFMA     both operands re-read transposed through a wave-private LDS tile
//     dW1 += dz1^T x           16 v_mfma_f32_4x4x1_16b_f32 (16 blocks of 4x4, K = 1 row)
//     dW3 += dout^T h2         16 / 32 FMAs per lane on its own row and units (summed over the rows at exit)
// "As they stand": a 16x16 accumulator tile mt holds, in lane (j = row = lane&15, g = lane>>4), register r, the hidden
// unit 16mt + 4g + r.  An MFMA sums over k in ANY order as long as A and B agree, so k-step s of the next product
// takes B from register (s&3) of tile (s>>2) and A = W[..][16(s>>2) + 4g + (s&3)] — no lane movement, no LDS
// (cdna_hip_programming.md §3 "An accumulator tile as the next MFMA's operand").  The sum order differs from a
// sequential-k fp32 chain; parity is to tolerance, not bitwise.
//
// LDS images (round 2: every read pattern audited against the per-instruction banking table of MI355X_MICROARCH.md §LDS):
//   W2g[(i>>2)][o][i&3]  = c W2[o][i]   layer-2 A fragments: lane (j,g) reads the float4 of k-steps 4c..4c+3 at ((4c+g)*64 + 16mt + j)*4;
//                                       the 16 lanes of a ds_read_b128 group hold 16 different j  -> 64 different banks
//   W2t[(o>>2)][i][o&3]  =   W2[o][i]   the same for dh1 = W2^T dz2 (A[i][k = o]): 16 ds_read_b128 per tile instead of 64 ds_read_b32
//   bufA / bufB [row][68]               wave-private staging images; the dW2 k-step s takes rows s + 4g (NOT 4s + g: lane groups g, g+1
//                                       must sit 16 banks apart — (s+4g)*68 = 4s + 16g mod 32 — which removed the 2-way conflict of r01)
// c = 2 log2(e): the tanh argument scale is folded into W1, b1, W2, b2 when they are staged, so tanh is
// 4 instructions (v_exp, v_add, v_rcp, v_fma) on z' = c z; the backward products use the unscaled W2t.
//
// 16-row tiles keep an activation in 16 registers (a 32-row tile on 32x32x2 spilled 100 VGPRs at 2 waves/SIMD and
// wrote 62 MB of scratch per launch: profiles/r01a_*).  Row inputs are prefetched two tiles ahead (index) / one tile
// ahead (gathered row), so no dependent global load sits on a tile's critical path.
// Everything per tile is wave-private (no workgroup barrier in the loop); the only barriers are around the
// weight staging at entry and the deterministic cross-wave reduction at exit.  Per-block partial gradients go
// to a workspace slab and are summed in a fixed order by grad_reduce_kernel (no float atomics: reproducible).
//
// The optimizer step owed from the PREVIOUS minibatch rides on the weight staging (grad_pending_t): every workgroup
// recomputes the clip coefficient from the L2-resident summed gradient (block_grad_norm: bitwise the same everywhere)
// and steps the parameters it stages on their way into LDS / registers; the first workgroup of each net writes the
// stepped {params, exp_avg, exp_avg_sq} to the OTHER buffer set (ping-pong: nobody reads what this launch writes).
// mi_ppo_update therefore runs 2 launches per optimizer step (gradient, slab sum) instead of 3.
// =====================================================================================================
#define W2S 68  // padded LDS row stride in floats: 68 = 4 (mod 32) keeps b128 row writes and b32 row reads conflict-free
// Fixed design constants.  Each was an A/B switch in rounds 1-2; the alternatives are recorded negatives (profiles/r02a_grad_stamps.txt) and were
// retired in round 3: 4 waves per workgroup / two workgroups per CU (+0 %, twice the slabs), a start-time stagger between co-resident workgroups (no effect),
// alternating s_setprio per tile (no effect), other role bits (+-1 %), unscaled tanh (5 instead of 4 instructions), dW3 on the 4x4x1 MFMA from an LDS image
// (one staging image and a 16-MFMA chain more), LLVM's iglp_opt(0/1) on the tile body (no effect).
#define GRAD_WAVES 8        // ONE 512-thread workgroup per CU: two waves per SIMD share one staged copy of the weights
#define GRAD_WPS 2          // waves per SIMD the kernel is built for
#define GRAD_OCC (GRAD_WPS * 4 / GRAD_WAVES)   // workgroups per CU
// The actor tiles evaluate ONE head, d = l0 - l1 (mi_grad_kernel.inc; round 4).
// GRAD_OLD_SHARE: of every 16 tile rounds, how many go to the first-dispatched ("older") half of the waves (age arbitration: the older wave of a SIMD wins every issue
// conflict).  Measured per launch (profiles/r04_grad_ab.txt): 9: 71.6, 10: 71.0, 11: 70.4, 12: 71.9, 13: 73.3 us.
#define GRAD_OLD_SHARE 11
#define GRAD_ROLE_BIT 3     // which blockIdx bit selects actor / critic (bit 0 would pin one net per XCD)
// Both nets get the same number of workgroups: the minibatch is 8,192 tiles per net = exactly 16 per SIMD at 128 : 128 workgroups, and any other split leaves some SIMDs
// with a 17th tile (+6 % on their CU) — with the one-head actor tile the two tiles cost the same (740 / 725 instructions), and 0 extra actor workgroups per 128 measured
// best (-2: 74.2, -1: 74.2, 0: 71.0, +1: 73.1, +2: 73.1 us; the two-head tile of rounds 1 - 3 ran best at +2).  The A/B switches for both (and for the two-head tile)
// left the source in round 5: docs/LEDGER.md has the numbers and the commit.
#define TROWS 16
#define PART_STRIDE 4624
#define PART_LOSS 4610
#define GRAD_MAX_BLOCKS 1024
#define GRAD_SPARE_SLABS 16                 // the workspace's last slabs hold mi_ppo_update's two spare optimizer-state sets
#define RED_STRIDE 4640                     // per-wave slot of the exit reduction: 4096 dW2 + 544 small values
#define STATE_STRIDE 9216                   // floats between params / exp_avg / exp_avg_sq inside a spare set

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0); }
__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 0, 0, 0); }

// tanh of z given z' = z * 2 log2(e) — the same function as mi_tanhf, with the scale already applied by the weights
__device__ __forceinline__ float tanh_prescaled(float zs) {
    const float e = __builtin_amdgcn_exp2f(zs);
    return __builtin_fmaf(-2.0f, __builtin_amdgcn_rcpf(e + 1.0f), 1.0f);
}
#define GRAD_TANH(x) tanh_prescaled(x)
#define GRAD_PS 2.8853900817779268f

// ---- GRAD_BX (experiment): the two contractions whose B operand is an activation in accumulator registers (layer 2 forward,
// dh1 backward) on v_mfma_f32_16x16x32_bf16 with BOTH operands split into three bf16 parts (x = hi + mid + lo exactly up to 2^-27 |x|) and six products
// per k-half (hi.hi, hi.mid, mid.hi, hi.lo, lo.hi, mid.mid; the dropped ones are < 2^-24 relative): f32-grade results from the bf16 matrix pipe.
// The accumulator-as-B trick carries over: lane (row j, g) supplies k-slot e of k-half s as unit 16 (2s + (e >> 2)) + 4g + (e & 3), i.e. the registers
// of tiles 2s and 2s + 1 as they stand; the A images hold W2 in exactly that slot order, pre-split when the weights are staged.
__device__ __forceinline__ unsigned pk_bf16(float a, float b) {   // v_cvt_pk_bf16_f32: {bf16(a) in the low half, bf16(b) in the high half}, round to nearest even
    typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
    typedef float f2 __attribute__((ext_vector_type(2)));
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f2{a, b}, bf2));
}
// (a, b) -> packed hi / mid / lo parts: hi = bf16(x), mid = bf16(x - hi), lo = bf16(x - hi - mid); both subtractions are exact in f32.
// (Tried: the residuals by v_dot2c_f32_bf16 with a (-1, 0) / (0, -1) selector, one instruction per value instead of unpack + subtract — slower,
// 61.0 vs 59.7 us per launch, and NOT exact: the golden-gradient test fails, the instruction does not keep the f32 addend's low bits.)
__device__ __forceinline__ void split3(float a, float b, unsigned& hi, unsigned& mid, unsigned& lo) {
    hi = pk_bf16(a, b);
    const float ra = a - __builtin_bit_cast(float, hi << 16), rb = b - __builtin_bit_cast(float, hi & 0xffff0000u);
    mid = pk_bf16(ra, rb);
    const float qa = ra - __builtin_bit_cast(float, mid << 16), qb = rb - __builtin_bit_cast(float, mid & 0xffff0000u);
    lo = pk_bf16(qa, qb);
}
struct bx_parts { bf16x8 p[3]; };   // [0] hi, [1] mid, [2] lo of 8 k-slots
__device__ __forceinline__ bx_parts split8(const f32x4& t0, const f32x4& t1) {   // k-slots 0..3 = t0[0..3], 4..7 = t1[0..3]
    unsigned h[4], m[4], l[4];
    split3(t0[0], t0[1], h[0], m[0], l[0]); split3(t0[2], t0[3], h[1], m[1], l[1]);
    split3(t1[0], t1[1], h[2], m[2], l[2]); split3(t1[2], t1[3], h[3], m[3], l[3]);
    bx_parts r;
    r.p[0] = __builtin_bit_cast(bf16x8, u32x4{h[0], h[1], h[2], h[3]});
    r.p[1] = __builtin_bit_cast(bf16x8, u32x4{m[0], m[1], m[2], m[3]});
    r.p[2] = __builtin_bit_cast(bf16x8, u32x4{l[0], l[1], l[2], l[3]});
    return r;
}
// acc += A . B over one k-half (32 k-slots), six products, small terms first
__device__ __forceinline__ f32x4 bx_mac(const bf16x8 (&a)[3], const bx_parts& b, f32x4 acc) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[2], b.p[0], acc, 0, 0, 0);   // lo . hi
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b.p[2], acc, 0, 0, 0);   // hi . lo
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], b.p[1], acc, 0, 0, 0);   // mid . mid
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], b.p[0], acc, 0, 0, 0);   // mid . hi
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b.p[1], acc, 0, 0, 0);   // hi . mid
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b.p[0], acc, 0, 0, 0);   // hi . hi
    return acc;
}

// Transposed bf16 images for dW2, [part][row half][unit][8 rows]: BX_HALF ushorts per row half (1,040 B: the 16 extra bytes put rows 8-15 four banks
// behind rows 0-7, so one ds_write_b16 of the wave — lanes (row j, group g), units 4 apart per g — touches 32 different dwords), read back as one
// ds_read_b128 per lane = the 8 rows of half (lane >> 5) of unit (lane & 31): 16-byte chunks at a 16-byte stride, conflict-free in the b128 lane groups
// (a [unit][16 rows] image reads 32-byte-strided chunks: 2-way conflicts, 31 % of the variant's LDS cycles in its first PMC pass).
#define BX_HALF 520
#define BX_PART (2 * BX_HALF)
#define BX_IMG (3 * BX_PART)
// the lane's 8 split k-slots (units 16 (2 sh) + 4g + {0..3}, 16 (2 sh + 1) + 4g + {0..3} of row j) into the image
__device__ __forceinline__ void bx_store_image(unsigned short* img, const bx_parts& q, int sh, int g, int j) {
#pragma unroll
    for (int pp = 0; pp < 3; ++pp) {
        const u32x4 w = __builtin_bit_cast(u32x4, q.p[pp]);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            unsigned short* at = img + pp * BX_PART + (j >> 3) * BX_HALF + (16 * (2 * sh + (e >> 1)) + 4 * g + 2 * (e & 1)) * 8 + (j & 7);
            at[0] = (unsigned short)(w[e] & 0xffffu);
            at[8] = (unsigned short)(w[e] >> 16);
        }
    }
}
// acc[o][i] += sum over the tile's 16 rows of A[o][row] B[row][i] on v_mfma_f32_32x32x16_bf16, six products, small terms first
__device__ __forceinline__ f32x16 bx_mac32(const bf16x8 (&a)[3], const bf16x8 (&b)[3], f32x16 acc) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[2], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], acc, 0, 0, 0);
    return acc;
}

// Total L2 norm of a flat gradient — ONE expression tree wherever it is evaluated, so every route to the same optimizer step (the clip + Adam launch, the
// owed step on a gradient launch's weight staging, single rank or sharded) gets bitwise the same clip coefficient:
//   s_b   = xor-butterfly (32, 16, 8, 4, 2, 1) over the 64 lanes of g[pos(64 b + lane)]^2, in f64           (block sums)
//   S_t   = s_t + s_(t+256) + ...  sequentially, t < 256
//   W_w   = xor-butterfly over the 64 lanes of S_(64 w + lane), w < 4;   total = (W_0 + W_1) + (W_2 + W_3);   norm = (float) sqrt(total)
// pos() is the identity, except for a PPO parameter vector (n == NPARAMS), where the blocks follow the order in which grad_reduce_kernel produces the elements
// (W2's gradient leaves the slabs in accumulator-fragment order): that kernel holds block b's 64 summed elements in one wave and writes s_b as it goes
// (`parts`), so the 256 workgroups of the next gradient launch each read 144 doubles instead of the whole 36.6 KB gradient (6,100 of the
// prologue's 15,000 cycles).  Without `parts` (mi_clip_adam on a caller's gradient; the sharded route, where the all-reduce changes the gradient after the block sums
// were taken) the same tree is evaluated from the gradient itself: wave w of the workgroup takes blocks w, w + nw, ... (256 % nw == 0, so S_t has one owner).
// Contains __syncthreads (1 with parts, 3 without); every thread of the workgroup must call it; blockDim.x is a multiple of 64 and >= 256.
__device__ __forceinline__ int ppo_slab_to_param(int p) {   // position in [actor slab | critic slab] order -> index in the flat parameter vector
    const int base = p < C_BASE ? 0 : C_BASE, off = p - base;
    if (off >= N_W2 && off < N_W2 + HID * HID) {
        const int s = off - N_W2, r = s & 3, F = s >> 2, ln = F & 63, frag = F >> 6;
        return base + N_W2 + (16 * (frag >> 2) + 4 * (ln >> 4) + r) * HID + 16 * (frag & 3) + (ln & 15);
    }
    return p;
}
__device__ __forceinline__ double wave_butterfly_f64(double s) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    return s;
}
#define NORM_BLOCKS ((NPARAMS + 63) / 64)   // 144
// f64 lane exchanges on the VALU (two 32-bit halves each): the stages of wave_butterfly_f64 without its ds_bpermute round trips
__device__ __forceinline__ double f64_pack(unsigned lo, unsigned hi) { return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo); }
template <int CTRL>
__device__ __forceinline__ double f64_dpp(double v) {
    const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
    return f64_pack((unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)u, CTRL, 0xF, 0xF, true), (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)(u >> 32), CTRL, 0xF, 0xF, true));
}
// lanes 0-31: a[i] + a[i + 32]; lanes 32-63: b[i - 32] + b[i]  — the xor-32 stage of TWO butterflies in one add (v_permlane32_swap: the upper half of the first
// operand trades places with the lower half of the second)
__device__ __forceinline__ double f64_swap32_sum(double a, double b) {
    const unsigned long long ua = __builtin_bit_cast(unsigned long long, a), ub = __builtin_bit_cast(unsigned long long, b);
    auto l = __builtin_amdgcn_permlane32_swap((unsigned)ua, (unsigned)ub, false, false);
    auto h = __builtin_amdgcn_permlane32_swap((unsigned)(ua >> 32), (unsigned)(ub >> 32), false, false);
    return f64_pack((unsigned)l[0], (unsigned)h[0]) + f64_pack((unsigned)l[1], (unsigned)h[1]);
}
// rows (16 lanes) 0 / 2: a[i] + a[i + 16]; rows 1 / 3: b[i - 16] + b[i]  — the xor-16 stage of two butterflies (v_permlane16_swap: odd rows of the first operand trade
// places with even rows of the second)
__device__ __forceinline__ double f64_swap16_sum(double a, double b) {
    const unsigned long long ua = __builtin_bit_cast(unsigned long long, a), ub = __builtin_bit_cast(unsigned long long, b);
    auto l = __builtin_amdgcn_permlane16_swap((unsigned)ua, (unsigned)ub, false, false);
    auto h = __builtin_amdgcn_permlane16_swap((unsigned)(ua >> 32), (unsigned)(ub >> 32), false, false);
    return f64_pack((unsigned)l[0], (unsigned)h[0]) + f64_pack((unsigned)l[1], (unsigned)h[1]);
}
__device__ __forceinline__ float block_grad_norm(const float* __restrict__ grads, const double* __restrict__ parts, int n, double* ws /* shared [4] */,
                                                 double* sparts /* shared [256] */) {
    const int t = threadIdx.x, lane = t & 63, w = t >> 6, nw = (int)blockDim.x >> 6;
    const int nb = (n + 63) >> 6;
    double S = 0.0;
    if (parts) {
        if (t < 256) for (int b = t; b < nb; b += 256) S += parts[b];
    } else if (n == NPARAMS && nw == 8) {
        // The sharded gradient launch (8 waves, the all-reduced PPO gradient, no block sums): the SAME tree, evaluated with a quarter of the instructions.  Wave w owns
        // blocks w + 8k, k < 18 (144 = 18 x 8): all 18 loads in flight before the first add, then the butterflies' xor-32 and xor-16 stages on PAIRS of blocks (one
        // swap + one add serves two blocks: each half / row of the wave keeps one of them), which leaves 5 values for the four in-row stages instead of 18 for all six.
        // Every add has the operands of the butterfly's add in that lane (IEEE addition commutes): bitwise the same s_b.  (this branch used to walk
        // 18 full f64 butterflies on ds_bpermute per wave.)
        static_assert(NORM_BLOCKS == 144, "18 blocks per wave");
        // (Tried, measured slower, removed: the block's share of ppo_slab_to_param on the scalar unit — per-lane constants + per-block scalars, 1 - 3 vector instructions per
        // load instead of ~12.  As `if (block range)` around the index: every load in a basic block of its own, +3.6 us per launch; as selects with unconditional clamped
        // loads: +2.4 us — two scalar branches per block for the scalar selects cost more than the vector arithmetic they replace.  This form: +1.2 us.)
        float x[18];
#pragma unroll
        for (int k = 0; k < 18; ++k) { const int p = 64 * (w + 8 * k) + lane; x[k] = p < n ? grads[ppo_slab_to_param(p)] : 0.0f; }
        if (t >= NORM_BLOCKS && t < 256) sparts[t] = 0.0;
        double r[9], s[5];
#pragma unroll
        for (int k = 0; k < 9; ++k) r[k] = f64_swap32_sum((double)x[k] * (double)x[k], (double)x[k + 9] * (double)x[k + 9]);   // lower half: block k, upper half: block k + 9
#pragma unroll
        for (int q = 0; q < 4; ++q) s[q] = f64_swap16_sum(r[2 * q], r[2 * q + 1]);   // rows 0..3: blocks 2q, 2q + 1, 2q + 9, 2q + 10
        s[4] = f64_swap16_sum(r[8], r[8]);                                            // rows 0, 1: block 8; rows 2, 3: block 17
#pragma unroll
        for (int q = 0; q < 5; ++q) {   // lane ^ 8, ^ 4 (a rotation by 4 of values that repeat every 8 lanes), ^ 2, ^ 1 inside the row
            s[q] += f64_dpp<0x128>(s[q]); s[q] += f64_dpp<0x124>(s[q]); s[q] += f64_dpp<0x4E>(s[q]); s[q] += f64_dpp<0xB1>(s[q]);
        }
        if ((lane & 15) == 0) {
            const int row = lane >> 4, k0 = (row & 1) + 9 * (row >> 1);
#pragma unroll
            for (int q = 0; q < 4; ++q) sparts[w + 8 * (2 * q + k0)] = s[q];
            if ((row & 1) == 0) sparts[w + 8 * (8 + 9 * (row >> 1))] = s[4];
        }
        __syncthreads();
        if (t < 256) S = sparts[t];
    } else {
        const bool perm = n == NPARAMS;
        if (t < 256) sparts[t] = 0.0;
        __syncthreads();
        constexpr int U = 6;   // blocks in flight per wave
        for (int b0 = w; b0 < nb; b0 += nw * U) {
            float x[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int b = b0 + nw * u, p = 64 * b + lane;
                x[u] = (b < nb && p < n) ? grads[perm ? ppo_slab_to_param(p) : p] : 0.0f;
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int b = b0 + nw * u;
                const double sb = wave_butterfly_f64((double)x[u] * (double)x[u]);
                if (lane == 0 && b < nb) sparts[b & 255] = sparts[b & 255] + sb;   // (b & 255) % nw == w: this wave is the only writer, in increasing b
            }
        }
        __syncthreads();
        if (t < 256) S = sparts[t];
    }
    S = wave_butterfly_f64(S);
    if (t < 256 && lane == 0) ws[w] = S;
    __syncthreads();
    return (float)sqrt((ws[0] + ws[1]) + (ws[2] + ws[3]));
}

// The optimizer step owed from the previous minibatch (clip_grad_norm_ + Adam, ppo.py:191-192), applied by the weight staging.
struct grad_pending_t {
    const float* grads;                                   // summed gradient of the previous minibatch [NPARAMS]; nullptr = nothing owed
    const float* p_in; const float* m_in; const float* v_in;   // optimizer state before the owed step
    float* p_out; float* m_out; float* v_out;             // ... after it (never the *_in buffers)
    float* grad_norm;                                     // nullable: pre-clip total norm of `grads`
    const double* norm_parts;                             // nullable: the block sums of squares grad_reduce_kernel wrote for `grads` (single rank only: see block_grad_norm)
    const uint32_t* gate;                                 // status word of the P2P carrier that exchanged `grads` (mi_comm_gate) — non-zero: the step is WITHHELD, out = in; never null when grads is set (ppo_gate)
    float w1, b2, w2, step_size, rbc2, eps, max_norm;
};

// Which net a slab / workgroup serves.  Slab index vb: even = actor, odd = critic.  ri = index within the role (the order grad_reduce_kernel sums in), nr = workgroups of that role.
struct grad_role_t { int role, ri, nr; };
__host__ __device__ inline grad_role_t grad_role(unsigned vb, int n_blocks) {
    return grad_role_t{(int)(vb & 1u), (int)(vb >> 1), n_blocks >> 1};
}
__host__ __device__ inline int grad_slab(int role, int ri) { return 2 * ri + role; }   // inverse: the slab of workgroup ri of a role

struct row_in {
    float x;          // observation component g of the row (lane (j,g))
    int act;
    float f0, f1;     // actor: old log-prob, advantage; critic: return, old value
};

template <bool ACTOR>
__device__ __forceinline__ row_in gather_row(int rid, int g, const float* __restrict__ observations, const int64_t* __restrict__ actions,
                                             const float* __restrict__ log_probs, const float* __restrict__ advantages,
                                             const float* __restrict__ returns, const float* __restrict__ values) {
    row_in r;
    r.x = observations[4 * (size_t)rid + g];
    if constexpr (ACTOR) {
        r.act = reinterpret_cast<const int*>(actions)[2 * (size_t)rid];  // low dword of the int64 (0 or 1)
        r.f0 = log_probs[rid]; r.f1 = advantages[rid];
    } else {
        r.act = 0;
        r.f0 = returns[rid]; r.f1 = values[rid];
    }
    return r;
}

#ifdef GRAD_STAMPS
#define STAMP(k) do { __builtin_amdgcn_sched_barrier(0); { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); \
                      stamp_acc[k] += t_ - stamp_last; stamp_last = t_; } __builtin_amdgcn_sched_barrier(0); } while (0)
#define STAMP_BASE(part) (reinterpret_cast<unsigned long long*>((part) + (size_t)(GRAD_MAX_BLOCKS / 2) * PART_STRIDE))
#else
#define STAMP(k) do {} while (0)
#endif

// ---- the two instantiations of grad_kernel (mi_grad_kernel.inc) ----
#define GRAD_BX 0
#define GV(x) x##_f32
#include "mi_grad_kernel.inc"
#undef GRAD_BX
#undef GV
#define GRAD_BX 1
#define GV(x) x##_bx
#include "mi_grad_kernel.inc"
#undef GRAD_BX
#undef GV

// grads[p] = sum over the partial slabs of p's net in a FIXED order (reproducible); the last block finishes the loss terms.
// 1024 threads = 64 consecutive slab positions x 16 slab groups: every wave reads 256 contiguous bytes per slab, 16 independent
// loads per lane, then the 16 group sums are added in group order through LDS.  Slab positions inside a net's W2 range are in
// accumulator-fragment order (grad_body's exit reduction); the store maps them back to W2[o][i].
// WORLD > 0 (sharded run on the P2P carrier, csrc/mi_comm.hip): the all-reduce of {gradient, loss terms} (ppo.py:189 -> :191 with the exchange in between) happens HERE,
// without a launch of its own — the lane that has just summed element p stores it as a line into slot (parity, rank) of every rank's inbox, polls the WORLD lines of p in
// its own inbox and adds them in RANK ORDER (the same bits on every rank), so `grads`, `loss_terms` and the block sums of squares come out ALL-REDUCED and the next
// gradient launch's owed clip + Adam takes the single-rank branch.  Only wave 0 of a 1,024-thread workgroup waits, for peers' stores that depend on no wait; 145 small
// workgroups never fill the chip, so two ranks time-sharing one device cannot starve each other.  A peer that never arrives: bounded wait, status word, the local share.
#define RED_PARAMS 64
#define RED_GROUPS 16
template <int WORLD>
__global__ void __launch_bounds__(RED_PARAMS * RED_GROUPS)
grad_reduce_kernel(const float* __restrict__ workspace, int n_blocks, float ent_coef, float vf_coef, double inv_count,
                   float* __restrict__ grads, float* __restrict__ loss_terms, double* __restrict__ norm_parts, const p2p_args_t x) {
    MI_INSIDE_SCOPE(MI_PROF_REDUCE);
    const int pblocks = (NPARAMS + RED_PARAMS - 1) / RED_PARAMS;
    if ((int)blockIdx.x < pblocks) {
        __shared__ float part[RED_GROUPS][RED_PARAMS];
        const int pl = threadIdx.x & (RED_PARAMS - 1), sg = threadIdx.x >> 6;
        const int p = blockIdx.x * RED_PARAMS + pl;
        float acc = 0.0f;
        const int role = p < C_BASE ? 0 : 1;
        const int off = p - (role ? C_BASE : 0);   // slab position within the net
        if (p < NPARAMS) {
            const float* src = workspace + off;
            const int nr = n_blocks >> 1;
            if (nr <= 9 * RED_GROUPS) {   // the full grid (130 / 126 slabs per net): every load of the thread in flight at once, summed in workgroup order
                float v[9];
#pragma unroll
                for (int u = 0; u < 9; ++u) { const int k = sg + RED_GROUPS * u; v[u] = k < nr ? src[(size_t)grad_slab(role, k) * PART_STRIDE] : 0.0f; }
#pragma unroll
                for (int u = 0; u < 9; ++u) if (sg + RED_GROUPS * u < nr) acc += v[u];
            } else {
#pragma unroll 4
                for (int k = sg; k < nr; k += RED_GROUPS) acc += src[(size_t)grad_slab(role, k) * PART_STRIDE];
            }
        }
        part[sg][pl] = acc;
        __syncthreads();
        if (sg == 0) {   // wave 0: one element per lane (0 past the end)
            float t = 0.0f;
            if (p < NPARAMS) {
#pragma unroll
                for (int k = 0; k < RED_GROUPS; ++k) t += part[k][pl];
                t = p2p_exchange<WORLD>(x, p, t);   // line p = slab position p on every rank
                grads[ppo_slab_to_param(p)] = t;
            }
            // block sum of squares for the next launch's clip coefficient (block_grad_norm's s_b: same butterfly, same element -> lane map)
            const double sb = wave_butterfly_f64((double)t * (double)t);
            if (pl == 0) norm_parts[blockIdx.x] = sb;
        }
    } else {
        // loss terms: pg / entropy from actor blocks, value loss from critic blocks
        double pg = 0.0, en = 0.0, vl = 0.0;
        for (int b = threadIdx.x; b < n_blocks; b += RED_PARAMS * RED_GROUPS) {
            const float* s = workspace + (size_t)b * PART_STRIDE + PART_LOSS;
            if (grad_role((unsigned)b, n_blocks).role == 0) { pg += s[0]; en += s[1]; } else { vl += s[0]; }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { pg += __shfl_xor(pg, o); en += __shfl_xor(en, o); vl += __shfl_xor(vl, o); }
        __shared__ double t[3][16];
        if ((threadIdx.x & 63) == 0) { t[0][threadIdx.x >> 6] = pg; t[1][threadIdx.x >> 6] = en; t[2][threadIdx.x >> 6] = vl; }
        __syncthreads();
        if constexpr (WORLD > 0) {
            // the four shares travel as lines NPARAMS .. NPARAMS + 3, each summed in rank order like a gradient element (what the one-buffer all-reduce did): lanes 0 .. 3
            // exchange one term each, side by side (one thread doing the four in turn put four memory round trips on this workgroup, the launch's longest)
            __shared__ float lt[4];
            if (threadIdx.x == 0) {
                double PG = 0.0, EN = 0.0, VL = 0.0;
                for (int k = 0; k < 16; ++k) { PG += t[0][k]; EN += t[1][k]; VL += t[2][k]; }
                const float t0 = (float)(PG * inv_count), t1 = (float)(EN * inv_count), t2 = (float)(0.5 * VL * inv_count);
                lt[0] = t0; lt[1] = t1; lt[2] = t2; lt[3] = t0 - ent_coef * t1 + t2 * vf_coef;  // ppo.py:187
            }
            __syncthreads();
            if (threadIdx.x < 4 && loss_terms) loss_terms[threadIdx.x] = p2p_exchange<WORLD>(x, NPARAMS + (int)threadIdx.x, lt[threadIdx.x]);
        } else if (threadIdx.x == 0 && loss_terms) {
            double PG = 0.0, EN = 0.0, VL = 0.0;
            for (int k = 0; k < 16; ++k) { PG += t[0][k]; EN += t[1][k]; VL += t[2][k]; }
            const float t0 = (float)(PG * inv_count), t1 = (float)(EN * inv_count), t2 = (float)(0.5 * VL * inv_count);
            loss_terms[0] = t0; loss_terms[1] = t1; loss_terms[2] = t2;
            loss_terms[3] = t0 - ent_coef * t1 + t2 * vf_coef;  // ppo.py:187
        }
    }
}

static int g_grad_blocks = 0;
static int grad_blocks() {
    if (g_grad_blocks == 0) {
        int dev = 0, cus = 256;
        if (hipGetDevice(&dev) == hipSuccess) {
            hipDeviceProp_t prop;
            if (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) cus = prop.multiProcessorCount;
        }
        int b = GRAD_OCC * cus;  // GRAD_OCC workgroups per CU, alternating actor / critic
        b &= ~1;
        if (b > GRAD_MAX_BLOCKS - GRAD_SPARE_SLABS) b = GRAD_MAX_BLOCKS - GRAD_SPARE_SLABS;  // the last slabs are mi_ppo_update's spare optimizer state
        g_grad_blocks = b;
    }
    return g_grad_blocks;
}
#define NORM_PARTS_OFF (6 * (size_t)STATE_STRIDE)   // floats behind the two spare sets: NORM_BLOCKS doubles written by grad_reduce_kernel
static_assert((size_t)GRAD_SPARE_SLABS * PART_STRIDE >= NORM_PARTS_OFF + 2 * NORM_BLOCKS, "spare optimizer-state sets + norm block sums must fit behind the slabs");
static_assert(NORM_PARTS_OFF % 2 == 0 && ((size_t)(GRAD_MAX_BLOCKS - GRAD_SPARE_SLABS) * PART_STRIDE) % 2 == 0, "norm block sums must be 8-byte aligned");
static double* ws_norm_parts(void* workspace) {
    return reinterpret_cast<double*>(reinterpret_cast<float*>(workspace) + (size_t)(GRAD_MAX_BLOCKS - GRAD_SPARE_SLABS) * PART_STRIDE + NORM_PARTS_OFF);
}
static_assert(STATE_STRIDE >= NPARAMS && STATE_STRIDE % 4 == 0, "STATE_STRIDE");

extern "C" size_t mi_ppo_workspace_bytes(void) { return (size_t)GRAD_MAX_BLOCKS * PART_STRIDE * sizeof(float); }

// Which matrix pipe grad_kernel's three 64 x 64 contractions run on (process-wide; every later gradient launch of this process uses it).
static int g_contraction = MI_CONTRACTION_F32;
extern "C" int mi_ppo_set_contraction(int mode) {
    MI_CHECK_ARG(mode == MI_CONTRACTION_F32 || mode == MI_CONTRACTION_BF16X3, "mode must be MI_CONTRACTION_F32 or MI_CONTRACTION_BF16X3");
    g_contraction = mode;
    return MI_OK;
}
extern "C" int mi_ppo_get_contraction(void) { return g_contraction; }

// the gate of an owed step: the carrier's status word, or (no P2P carrier) a device word that is always zero — the prologue's load of it is unconditional
__device__ uint32_t mi_always_zero_word = 0u;
static const uint32_t* ppo_gate(void* comm) {
    if (const uint32_t* g = mi_comm_gate(comm)) return g;
    static const uint32_t* zero[64] = {nullptr};   // per device (the symbol has one instance per device)
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    if (!zero[dev]) { void* p = nullptr; if (hipGetSymbolAddress(&p, HIP_SYMBOL(mi_always_zero_word)) == hipSuccess) zero[dev] = (const uint32_t*)p; }
    return zero[dev];
}
static grad_pending_t no_pending() { grad_pending_t z; memset(&z, 0, sizeof(z)); return z; }

// TEST HOOK (include/mi_rl.h): mi_ppo_update / mi_ppo_update_sharded behave as at world_size > 1 in everything but the collective — the owed optimizer steps
// recompute the clip coefficient from the gradient itself (norm_parts = nullptr) — so the branch only a multi-GPU run takes can be checked and timed on one GPU.
static int g_assume_sharded = 0;
extern "C" int mi_ppo_test_assume_sharded(int on) { g_assume_sharded = on ? 1 : 0; return MI_OK; }

// gradient launch + slab sum.  `pend.grads != nullptr`: the launch first applies the owed optimizer step (see grad_pending_t).
template <int WORLD>
static void grad_reduce_launch(hipStream_t s, const float* workspace, int blocks, float ent_coef, float vf_coef, double inv_count, float* grads, float* loss_terms,
                               double* norm_parts, const p2p_args_t& x) {
    grad_reduce_kernel<WORLD><<<NORM_BLOCKS + 1, RED_PARAMS * RED_GROUPS, 0, s>>>(workspace, blocks, ent_coef, vf_coef, inv_count, grads, loss_terms, norm_parts, x);
}

// p2p != nullptr: the slab sum also all-reduces {grads, loss_terms} over the P2P carrier (mi_comm_p2p_next is drawn here: exactly one exchange per call, on every rank)
static int ppo_grad_launch(const float* params, const grad_pending_t& pend, const float* observations, const int64_t* actions, const float* log_probs,
                           const float* advantages, const float* returns, const float* values, const int32_t* idx, int mb, const double* adv_sums,
                           float clip_coef, float ent_coef, float vf_coef, double inv_count, void* workspace, float* grads, float* loss_terms,
                           hipStream_t s, void* p2p = nullptr) {
    int blocks = grad_blocks();
    // small minibatches: no point launching blocks that would only write zero slabs
    const int tiles = (mb + TROWS - 1) / TROWS;
    const int need = 2 * ((tiles + GRAD_WAVES - 1) / GRAD_WAVES);
    if (need < blocks) blocks = need;  // (the role swizzle falls back to identity when the grid is not a multiple of 2^(bit+1))
    {
        mi_prof_scope prof(MI_PROF_GRAD, s);
        if (g_contraction == MI_CONTRACTION_BF16X3)
            grad_kernel_bx<<<blocks, 64 * GRAD_WAVES, 0, s>>>(params, pend, observations, actions, log_probs, advantages, returns, values, idx, mb,
                                                              adv_sums, clip_coef, ent_coef, vf_coef, (float)inv_count, (float*)workspace);
        else
            grad_kernel_f32<<<blocks, 64 * GRAD_WAVES, 0, s>>>(params, pend, observations, actions, log_probs, advantages, returns, values, idx, mb,
                                                               adv_sums, clip_coef, ent_coef, vf_coef, (float)inv_count, (float*)workspace);
    }
    MI_LAUNCH_CHECK();
    {
        mi_prof_scope prof(MI_PROF_REDUCE, s);
        static_assert((NPARAMS + RED_PARAMS - 1) / RED_PARAMS == NORM_BLOCKS && RED_PARAMS == 64, "one norm block per reduction workgroup");
        p2p_args_t x;
        memset(&x, 0, sizeof(x));
        int world = 0;
        if (p2p) {
            if (loss_terms != grads + NPARAMS) { mi_set_error("ppo_grad_launch: the P2P exchange needs loss_terms == grads + MI_PPO_NPARAMS"); return MI_EINVAL; }
            int rc = mi_comm_p2p_next(p2p, (size_t)NPARAMS + 4, &x, &world, s);
            if (rc) return rc;
        }
        const float* ws = (const float*)workspace;
        double* np = ws_norm_parts(workspace);
        switch (world) {
            case 0: grad_reduce_launch<0>(s, ws, blocks, ent_coef, vf_coef, inv_count, grads, loss_terms, np, x); break;
            case 1: grad_reduce_launch<1>(s, ws, blocks, ent_coef, vf_coef, inv_count, grads, loss_terms, np, x); break;
            case 2: grad_reduce_launch<2>(s, ws, blocks, ent_coef, vf_coef, inv_count, grads, loss_terms, np, x); break;
            case 3: grad_reduce_launch<3>(s, ws, blocks, ent_coef, vf_coef, inv_count, grads, loss_terms, np, x); break;
            case 4: grad_reduce_launch<4>(s, ws, blocks, ent_coef, vf_coef, inv_count, grads, loss_terms, np, x); break;
            case 5: grad_reduce_launch<5>(s, ws, blocks, ent_coef, vf_coef, inv_count, grads, loss_terms, np, x); break;
            case 6: grad_reduce_launch<6>(s, ws, blocks, ent_coef, vf_coef, inv_count, grads, loss_terms, np, x); break;
            case 7: grad_reduce_launch<7>(s, ws, blocks, ent_coef, vf_coef, inv_count, grads, loss_terms, np, x); break;
            default: grad_reduce_launch<8>(s, ws, blocks, ent_coef, vf_coef, inv_count, grads, loss_terms, np, x); break;
        }
    }
    MI_LAUNCH_CHECK();
    return MI_OK;
}

extern "C" int mi_ppo_minibatch_grad(const float* params, const float* observations, const int64_t* actions,
                                     const float* log_probs, const float* advantages, const float* returns,
                                     const float* values, const int32_t* idx, int mb, const double* adv_sums, float clip_coef,
                                     float ent_coef, float vf_coef, double inv_count, void* workspace, float* grads,
                                     float* loss_terms, void* stream) {
    MI_CHECK_ARG(params && observations && actions && log_probs && advantages && returns && values && idx && adv_sums, "NULL input");
    MI_CHECK_ARG(workspace && grads, "NULL workspace/grads");
    MI_CHECK_ARG(mb > 0, "mb must be positive");
    return ppo_grad_launch(params, no_pending(), observations, actions, log_probs, advantages, returns, values, idx, mb, adv_sums, clip_coef, ent_coef,
                           vf_coef, inv_count, workspace, grads, loss_terms, (hipStream_t)stream);
}

// =====================================================================================================
// clip_grad_norm_ + Adam.  Every 256-thread workgroup recomputes the total norm of the whole (9,155-float, L2-resident)
// gradient in the same fixed order — so all workgroups get bitwise the same clip coefficient without a grid-wide
// hand-off — and then updates its own 256-parameter slice.  Out of place when the *_out pointers differ from *_in
// (mi_ppo_update's last step: from the spare set back into the caller's tensors).
// =====================================================================================================
__global__ void __launch_bounds__(512) clip_adam_kernel(const float* p_in, const float* m_in, const float* v_in, float* p_out, float* m_out, float* v_out,
                                                         const float* __restrict__ grads, int n, float w1, float b2,
                                                         float w2, float step_size, float rbc2, float eps, float max_norm,
                                                         float* __restrict__ grad_norm, const double* __restrict__ norm_parts, const uint32_t* __restrict__ gate) {
    __shared__ double ws[4], sparts[256];
    MI_INSIDE_SCOPE(MI_PROF_CLIP_ADAM);
    const bool hold = mi_gate_closed(gate);   // the exchange that produced `grads` timed out: keep the state (out of place: copy it through)
    const int i = blockIdx.x * blockDim.x + threadIdx.x;   // 256 threads, or 512 for a PPO parameter vector (block_grad_norm's 8-wave branch when there are no block sums)
    const bool live = i < n;
    float pm = live ? m_in[i] : 0.0f, pv = live ? v_in[i] : 0.0f;
    const float pp = live ? p_in[i] : 0.0f, pg = live ? grads[i] : 0.0f;
    const float total = block_grad_norm(grads, norm_parts, n, ws, sparts);
    float coef = max_norm / (total + 1e-6f);
    coef = coef > 1.0f ? 1.0f : coef;
    if (grad_norm && blockIdx.x == 0 && threadIdx.x == 0) *grad_norm = total;
    if (live) {
        float np = pp;
        if (!hold) np = mi_adam_elem(pp, pg * coef, pm, pv, w1, b2, w2, step_size, rbc2, eps);
        m_out[i] = pm; v_out[i] = pv;
        p_out[i] = np;
    }
}

struct adam_consts_t { float w1, b2, w2, step_size, rbc2, eps; };
static adam_consts_t adam_consts(int64_t step, double lr, double beta1, double beta2, double eps) {
    const double bc1 = 1.0 - pow(beta1, (double)step), bc2 = 1.0 - pow(beta2, (double)step);
    adam_consts_t k;
    k.w1 = (float)(1.0 - beta1); k.b2 = (float)beta2; k.w2 = (float)(1.0 - beta2);
    k.step_size = (float)(lr / bc1); k.rbc2 = (float)(1.0 / sqrt(bc2)); k.eps = (float)eps;
    return k;
}

static int clip_adam_launch(const float* p_in, const float* m_in, const float* v_in, float* p_out, float* m_out, float* v_out, const float* grads, int n,
                            const adam_consts_t& k, float max_norm, float* grad_norm, const double* norm_parts, hipStream_t s, const uint32_t* gate = nullptr) {
    mi_prof_scope prof(MI_PROF_CLIP_ADAM, s);
    const int threads = n == NPARAMS ? 512 : 256;   // same tree for any block size; 8 waves take the cheap branch on a PPO vector without block sums (sharded runs, mi_clip_adam)
    clip_adam_kernel<<<(n + threads - 1) / threads, threads, 0, s>>>(p_in, m_in, v_in, p_out, m_out, v_out, grads, n, k.w1, k.b2, k.w2, k.step_size, k.rbc2, k.eps, max_norm,
                                                                    grad_norm, norm_parts, gate);
    MI_LAUNCH_CHECK();
    return MI_OK;
}

// mi_clip_adam behind an exchange on `comm` (mi_dqn_td_update_sharded): the step is withheld when that carrier's status word is set
int mi_clip_adam_gated(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int n, int64_t step, double lr, double beta1, double beta2, double eps,
                       float max_norm, float* grad_norm, void* comm, void* stream) {
    return clip_adam_launch(params, exp_avg, exp_avg_sq, params, exp_avg, exp_avg_sq, grads, n, adam_consts(step, lr, beta1, beta2, eps), max_norm, grad_norm,
                            nullptr, (hipStream_t)stream, mi_comm_gate(comm));
}

extern "C" int mi_clip_adam(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int n, int64_t step, double lr,
                            double beta1, double beta2, double eps, float max_norm, float* grad_norm, void* stream) {
    MI_CHECK_ARG(params && grads && exp_avg && exp_avg_sq, "NULL pointer");
    MI_CHECK_ARG(n > 0 && step >= 1, "n must be positive and step 1-based");
    return clip_adam_launch(params, exp_avg, exp_avg_sq, params, exp_avg, exp_avg_sq, grads, n, adam_consts(step, lr, beta1, beta2, eps), max_norm, grad_norm,
                            nullptr, (hipStream_t)stream);
}

// =====================================================================================================
// explained variance (ppo.py:194-195): 1 - var(values - returns) / var(values), unbiased, NaN if var == 0
// =====================================================================================================
__device__ inline double block_sum_1024(double x, double* sh) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = x;
    __syncthreads();
    double t = 0.0;
    for (int k = 0; k < 16; ++k) t += sh[k];
    return t;
}

__global__ void __launch_bounds__(1024) explained_var_kernel(const float* __restrict__ values, const float* __restrict__ returns, size_t n,
                                                              double* __restrict__ out) {
    __shared__ double sh[16];
    double sv = 0.0, sd = 0.0;
    for (size_t i = threadIdx.x; i < n; i += 1024) { sv += values[i]; sd += (double)values[i] - returns[i]; }
    const double mv = block_sum_1024(sv, sh) / (double)n, md = block_sum_1024(sd, sh) / (double)n;
    double vv = 0.0, vd = 0.0;
    for (size_t i = threadIdx.x; i < n; i += 1024) {
        const double a = values[i] - mv, b = ((double)values[i] - returns[i]) - md;
        vv += a * a; vd += b * b;
    }
    vv = block_sum_1024(vv, sh) / (double)(n - 1);
    vd = block_sum_1024(vd, sh) / (double)(n - 1);
    if (threadIdx.x == 0) *out = vv == 0.0 ? __builtin_nan("") : 1.0 - vd / vv;
}

// The same statistic over the rows of ALL ranks of a sharded run (ppo.py:194-195 is over the whole batch), in two halves around the caller's SUM all-reduces:
// means == nullptr: out = {sum values, sum (values - returns)} of this rank's rows; then, given the GLOBAL means {mean values, mean (values - returns)}:
// out = {sum (values - m0)^2, sum ((values - returns) - m1)^2}.  explained_var = 1 - out1 / out0 of the all-reduced sums (NaN if out0 == 0).
__global__ void __launch_bounds__(1024) explained_var_parts_kernel(const float* __restrict__ values, const float* __restrict__ returns, size_t n,
                                                                    const double* __restrict__ means, double* __restrict__ out) {
    __shared__ double sh[16];
    double a = 0.0, b = 0.0;
    if (means == nullptr) {
        for (size_t i = threadIdx.x; i < n; i += 1024) { a += values[i]; b += (double)values[i] - returns[i]; }
    } else {
        const double mv = means[0], md = means[1];
        for (size_t i = threadIdx.x; i < n; i += 1024) {
            const double x = values[i] - mv, y = ((double)values[i] - returns[i]) - md;
            a += x * x; b += y * y;
        }
    }
    a = block_sum_1024(a, sh); b = block_sum_1024(b, sh);
    if (threadIdx.x == 0) { out[0] = a; out[1] = b; }
}

extern "C" int mi_explained_var_parts(const float* values, const float* returns, size_t n, const double* means, double* out, void* stream) {
    MI_CHECK_ARG(values && returns && out, "NULL pointer");
    MI_CHECK_ARG(n >= 1, "n must be >= 1");
    explained_var_parts_kernel<<<1, 1024, 0, (hipStream_t)stream>>>(values, returns, n, means, out);
    MI_LAUNCH_CHECK();
    return MI_OK;
}

extern "C" int mi_explained_var(const float* values, const float* returns, size_t n, double* out, void* stream) {
    MI_CHECK_ARG(values && returns && out, "NULL pointer");
    MI_CHECK_ARG(n >= 2, "n must be >= 2");
    explained_var_kernel<<<1, 1024, 0, (hipStream_t)stream>>>(values, returns, n, out);
    MI_LAUNCH_CHECK();
    return MI_OK;
}

// =====================================================================================================
// One whole outer update, enqueued back to back (single rank, production RNG)
// =====================================================================================================
// every epoch's permutation (ppo.py:155, keyed) and per-minibatch advantage sums (ppo.py:169) of one update: one launch (+ one memset) when the
// minibatch size is a multiple of 4096, otherwise per epoch.  perm_all: dev i32 [epochs, n_rows]; sums_all: dev f64 [epochs, n_minibatch, 3]
// (LOCAL sums: sharded runs all-reduce them once).
extern "C" int mi_ppo_perms_and_stats(uint64_t seed, int update_index, int epochs, int n_rows, int n_minibatch, const float* advantages, int32_t* perm_all,
                                      double* sums_all, void* stream) {
    MI_CHECK_ARG(advantages && perm_all && sums_all, "NULL pointer");
    MI_CHECK_ARG(epochs > 0 && n_rows > 0 && n_minibatch > 0 && n_rows % n_minibatch == 0, "bad sizes");
    hipStream_t s = (hipStream_t)stream;
    const int mb = n_rows / n_minibatch;
    if (mb % PS_PER_BLOCK == 0 && epochs <= PS_MAX_EPOCHS) {
        MI_HIP(hipMemsetAsync(sums_all, 0, sizeof(double) * 3 * (size_t)n_minibatch * epochs, s));
        ps_epochs_t pe;
        for (int ep = 0; ep < epochs; ++ep) {
            const uint64_t key = mi_perm_key(seed, (uint64_t)update_index, (uint64_t)ep);
            pe.k0[ep] = (uint32_t)key; pe.k1[ep] = (uint32_t)(key >> 32); pe.out[ep] = perm_all + (size_t)ep * n_rows;
        }
        uint32_t bits = 1;
        while ((1u << bits) < (uint32_t)n_rows) ++bits;
        if (bits < 2) bits = 2;
        mi_prof_scope prof(MI_PROF_STATS, s);
        perm_stats_kernel<<<dim3((n_rows + PS_PER_BLOCK - 1) / PS_PER_BLOCK, epochs), 256, 0, s>>>((uint32_t)n_rows, bits / 2, bits - bits / 2, pe, mb, n_minibatch, advantages,
                                                                                                 sums_all);
        MI_LAUNCH_CHECK();
        return MI_OK;
    }
    for (int ep = 0; ep < epochs; ++ep) {
        int rc = mi_make_perm((uint32_t)n_rows, mi_perm_key(seed, (uint64_t)update_index, (uint64_t)ep), perm_all + (size_t)ep * n_rows, stream);
        if (rc) return rc;
        rc = mi_adv_stats(advantages, perm_all + (size_t)ep * n_rows, mb, n_minibatch, sums_all + (size_t)3 * n_minibatch * ep, stream);
        if (rc) return rc;
    }
    return MI_OK;
}

// One optimizer-state set {params, exp_avg, exp_avg_sq}
struct opt_set_t { float* p; float* m; float* v; };

static int ppo_update_impl(void* handle, const mi_ppo_buffers_t* b, const mi_ppo_hparams_t* hp, void* comm, int world, void* stream);

extern "C" int mi_ppo_update(void* handle, const mi_ppo_buffers_t* b, const mi_ppo_hparams_t* hp, void* stream) {
    return ppo_update_impl(handle, b, hp, nullptr, 1, stream);
}

extern "C" int mi_ppo_update_sharded(void* handle, const mi_ppo_buffers_t* b, const mi_ppo_hparams_t* hp, void* comm, void* stream) {
    int world = 1;
    if (comm) {
        int rc = mi_comm_poll_impl(comm);   // an earlier wait of the P2P carrier ran out: MI_ESTATE before anything is enqueued
        if (rc) return rc;
        rc = mi_comm_info(comm, &world, nullptr, nullptr, nullptr);
        if (rc) return rc;
        MI_CHECK_ARG(b && b->loss_terms == b->grads + NPARAMS, "sharded update: loss_terms must be grads + MI_PPO_NPARAMS (one buffer, one all-reduce)");
    }
    return ppo_update_impl(handle, b, hp, comm, world, stream);
}

static int ppo_update_impl(void* handle, const mi_ppo_buffers_t* b, const mi_ppo_hparams_t* hp, void* comm, int world, void* stream) {
    MI_CHECK_ARG(handle && b && hp, "NULL pointer");
    MI_CHECK_ARG(hp->T > 0 && hp->n_minibatch > 0 && hp->update_epochs > 0, "bad hyper-parameters");
    mi_env* e = (mi_env*)handle;
    const int N = e->n, B = hp->T * N;
    MI_CHECK_ARG(B % hp->n_minibatch == 0, "T*N must be divisible by n_minibatch");
    const int mb = B / hp->n_minibatch;
    const bool fused = (mb % PS_PER_BLOCK) == 0;  // perm + stats in one pass; their fp64 accumulators are zeroed by the rollout launch
    // P2P carrier: grad_reduce_kernel itself exchanges {gradient, loss terms} (no launch per all-reduce) and its block sums of squares are those of the ALL-REDUCED
    // gradient, so the owed steps take the single-rank branch; RCCL: an in-stream ncclAllReduce behind every slab sum, the norm recomputed from the gradient
    const bool p2p = comm != nullptr && mi_comm_p2p_fused_ok(comm);   // (a P2P communicator shared by > 2 ranks of one device: the stand-alone launch, as on RCCL)
    const bool sharded_norm = (world > 1 && !p2p) || g_assume_sharded != 0;
    int rc = mi_rollout_gae_internal(handle, b->params, hp->T, b->obs_cur, b->observations, b->values, b->actions, b->log_probs, b->rewards,
                                     b->dones, b->episodes, b->episode_stats, b->max_ep, hp->gamma, hp->gae_lambda, b->advantages, b->returns,
                                     fused ? b->adv_sums : nullptr, 3 * hp->n_minibatch * hp->update_epochs, b->episode_stats_next, stream);
    if (rc) return rc;
    hipStream_t s = (hipStream_t)stream;
    // every epoch's permutation + advantage statistics (they depend on the advantages and the keys only): b->perm is [update_epochs, T*N]
    if (fused && hp->update_epochs <= PS_MAX_EPOCHS) {
        ps_epochs_t pe;
        for (int ep = 0; ep < hp->update_epochs; ++ep) {
            const uint64_t key = mi_perm_key(e->seed, (uint64_t)hp->update_index, (uint64_t)ep);
            pe.k0[ep] = (uint32_t)key; pe.k1[ep] = (uint32_t)(key >> 32); pe.out[ep] = b->perm + (size_t)ep * B;
        }
        uint32_t bits = 1;
        while ((1u << bits) < (uint32_t)B) ++bits;
        if (bits < 2) bits = 2;
        mi_prof_scope prof(MI_PROF_STATS, s);
        perm_stats_kernel<<<dim3((B + PS_PER_BLOCK - 1) / PS_PER_BLOCK, hp->update_epochs), 256, 0, s>>>((uint32_t)B, bits / 2, bits - bits / 2, pe, mb, hp->n_minibatch,
                                                                                                       b->advantages, b->adv_sums);
        MI_LAUNCH_CHECK();
    } else {
        rc = mi_ppo_perms_and_stats(e->seed, hp->update_index, hp->update_epochs, B, hp->n_minibatch, b->advantages, b->perm, b->adv_sums, stream);
        if (rc) return rc;
    }
    if (comm) {   // global advantage mean / std (ppo.py:169 over the union minibatch): one SUM all-reduce of every epoch's local sums
        mi_prof_scope prof(MI_PROF_COMM_STATS, s);
        rc = mi_comm_allreduce_impl(comm, b->adv_sums, (size_t)3 * hp->n_minibatch * hp->update_epochs, 1, s);
        if (rc) return rc;
    }
    // Optimizer steps.  Step k's clip + Adam is applied by the weight staging of gradient launch k+1 (grad_pending_t), reading one
    // state set and writing another: caller's tensors -> spare X -> spare Y -> X ... ; the last step is a launch of its own that
    // lands the state back in the caller's tensors.  2 launches per optimizer step + 1.
    float* spare = reinterpret_cast<float*>(b->workspace) + (size_t)(GRAD_MAX_BLOCKS - GRAD_SPARE_SLABS) * PART_STRIDE;
    const opt_set_t caller = {b->params, b->exp_avg, b->exp_avg_sq};
    const opt_set_t sets[2] = {{spare, spare + STATE_STRIDE, spare + 2 * STATE_STRIDE},
                               {spare + 3 * STATE_STRIDE, spare + 4 * STATE_STRIDE, spare + 5 * STATE_STRIDE}};
    opt_set_t cur = caller;
    int64_t step = hp->opt_step;   // steps applied or owed so far
    bool owed = false;
    int flip = 0;
    for (int ep = 0; ep < hp->update_epochs; ++ep) {
        const double* sums = b->adv_sums + (size_t)3 * hp->n_minibatch * ep;
        const int32_t* perm = b->perm + (size_t)ep * B;
        for (int k = 0; k < hp->n_minibatch; ++k) {
            grad_pending_t pend = no_pending();
            if (owed) {
                const adam_consts_t c = adam_consts(step, hp->lr, hp->beta1, hp->beta2, hp->eps);
                const opt_set_t out = sets[flip];
                flip ^= 1;
                pend.grads = b->grads; pend.p_in = cur.p; pend.m_in = cur.m; pend.v_in = cur.v; pend.p_out = out.p; pend.m_out = out.m; pend.v_out = out.v;
                pend.grad_norm = b->grad_norm; pend.w1 = c.w1; pend.b2 = c.b2; pend.w2 = c.w2; pend.step_size = c.step_size; pend.rbc2 = c.rbc2; pend.eps = c.eps;
                pend.max_norm = hp->max_grad_norm;
                pend.norm_parts = sharded_norm ? nullptr : ws_norm_parts(b->workspace);   // sharded: the all-reduce changed the gradient after the block sums were taken
                pend.gate = ppo_gate(comm);
                if (!pend.gate) { mi_set_error("mi_ppo_update: cannot resolve the always-zero gate word"); return MI_EHIP; }
                cur = out;   // what this launch trains on and what the next owed step starts from
            }
            rc = ppo_grad_launch(owed ? nullptr : cur.p, pend, b->observations, b->actions, b->log_probs, b->advantages, b->returns, b->values,
                                 perm + (size_t)k * mb, mb, sums + 3 * k, hp->clip_coef, hp->ent_coef, hp->vf_coef, 1.0 / ((double)mb * world), b->workspace, b->grads,
                                 b->loss_terms, s, p2p ? comm : nullptr);
            if (rc) return rc;
            if (comm && !p2p) {   // gradient shares (already scaled by 1/(world*mb)) + the 4 loss-term shares: one SUM all-reduce (ppo.py:189 -> :191)
                mi_prof_scope prof(MI_PROF_COMM_GRAD, s);
                rc = mi_comm_allreduce_impl(comm, b->grads, (size_t)NPARAMS + 4, 0, s);
                if (rc) return rc;
            }
            step += 1;
            owed = true;
        }
    }
    return clip_adam_launch(cur.p, cur.m, cur.v, caller.p, caller.m, caller.v, b->grads, NPARAMS, adam_consts(step, hp->lr, hp->beta1, hp->beta2, hp->eps),
                            hp->max_grad_norm, b->grad_norm, sharded_norm ? nullptr : ws_norm_parts(b->workspace), s, mi_comm_gate(comm));
}

// =====================================================================================================
// Hardware self-test of the MFMA fragment layouts grad_kernel relies on (exact small-integer data).
// report[0]: 32x32x2 A/B/D maps   report[1]: 4x4x1 (16 blocks) A/B/D maps   report[2]: accumulator-as-B chain (32x32x2)
// report[3]: accumulator-as-B chain on 16x16x4 (the form grad_kernel uses)   report[4]: 4x4x1 with A broadcast (cbsz / abid; rollout_q4_kernel)
// report[5]: 16x16x32 bf16 maps   report[6]: 32x32x16 bf16 maps (the split-bf16 gradient variant)
// dump (nullable, f32 [3*64*16]): raw accumulators of the three probes for offline diagnosis.
// =====================================================================================================
__global__ void __launch_bounds__(64) selftest_kernel(int32_t* __restrict__ report, float* __restrict__ dump) {
    const int lane = threadIdx.x, li = lane & 31, h = lane >> 5;
    int bad0 = 0, bad1 = 0, bad2 = 0;
    // probe 0: D = A B with A[i][k] = 1 + i + 37k (asymmetric), B[k][j] = 2 + 3j + 101k
    {
        f32x16 d;
#pragma unroll
        for (int r = 0; r < 16; ++r) d[r] = 0.0f;
        const float a = (float)(1 + li + 37 * h), b = (float)(2 + 3 * li + 101 * h);
        d = mfma32(a, b, d);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int i = (r & 3) + 8 * (r >> 2) + 4 * h, j = li;
            const float want = (float)((1 + i) * (2 + 3 * j) + (1 + i + 37) * (2 + 3 * j + 101));
            bad0 += d[r] != want;
            if (dump) dump[(0 * 64 + lane) * 16 + r] = d[r];
        }
    }
    // probe 1: 16 blocks of 4x4, K = 1: D[b][i][j] = A[b][i] B[b][j], A lane = 4b+i, B lane = 4b+j, D reg = i lane = 4b+j
    {
        f32x4 d = {0.0f, 0.0f, 0.0f, 0.0f};
        const float a = (float)(1 + lane), b = (float)(100 + 7 * lane);
        d = mfma4(a, b, d);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int blk = lane >> 2, j = lane & 3;
            const float want = (float)(1 + 4 * blk + i) * (float)(100 + 7 * (4 * blk + j));
            bad1 += d[i] != want;
            if (dump) dump[(1 * 64 + lane) * 16 + i] = d[i];
        }
    }
    // probe 2: Y^T = W X^T with X^T taken from a previous accumulator (permuted k): X[row j][unit u] = 1 + (j % 5) + 2u,
    //          W[o][u] = 1 + ((o + 3u) % 7);  want Y[j][o] = sum_u W[o][u] X[j][u]
    {
        f32x16 x[2];
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int u = (r & 3) + 8 * (r >> 2) + 4 * h + 32 * m;
                x[m][r] = (float)(1 + (li % 5) + 2 * u);
            }
        f32x16 y;
#pragma unroll
        for (int r = 0; r < 16; ++r) y[r] = 0.0f;
#pragma unroll
        for (int s = 0; s < 32; ++s) {
            const int u = ((s & 15) & 3) + 8 * ((s & 15) >> 2) + 4 * h + 32 * (s >> 4);
            const float a = (float)(1 + ((li + 3 * u) % 7));  // W[o = li][u]
            y = mfma32(a, x[s >> 4][s & 15], y);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int o = (r & 3) + 8 * (r >> 2) + 4 * h, j = li;
            int want = 0;
            for (int u = 0; u < 64; ++u) want += (1 + ((o + 3 * u) % 7)) * (1 + (j % 5) + 2 * u);
            bad2 += y[r] != (float)want;
            if (dump) dump[(2 * 64 + lane) * 16 + r] = y[r];
        }
    }
    // probe 3: the 16x16x4 form of probe 2 (what grad_kernel uses): lane (j = lane&15, g = lane>>4), tile mt, register r
    //          holds X[row j][unit 16mt + 4g + r]; k-step s takes B from register (s&3) of tile (s>>2) and
    //          A = W[o = j][16(s>>2) + 4g + (s&3)]; D: lane (j, g), register r = Y[row j][o = 4g + r]
    int bad3 = 0;
    {
        const int j = lane & 15, g = lane >> 4;
        f32x4 x[4];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) x[mt][r] = (float)(1 + (j % 5) + 2 * (16 * mt + 4 * g + r));
        f32x4 y = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const int u = 16 * (s >> 2) + 4 * g + (s & 3);
            y = mfma16((float)(1 + ((j + 3 * u) % 7)), x[s >> 2][s & 3], y);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int o = 4 * g + r;
            int want = 0;
            for (int u = 0; u < 64; ++u) want += (1 + ((o + 3 * u) % 7)) * (1 + (j % 5) + 2 * u);
            bad3 += y[r] != (float)want;
            if (dump) dump[(2 * 64 + lane) * 16 + 8 + r] = y[r];
        }
    }
    // probe 4: the 16-block 4x4x1 with the A operand broadcast from ONE block (cbsz = 4, abid = b'), as rollout_q4_kernel uses it:
    //          D[b][i][j] += A[b'][i] * B[b][j], A value taken from lane 4b' + i, B from lane 4b + j, D register i in lane 4b + j.
    //          Two chained k-steps with different source blocks (5, then 12) and asymmetric integer data.
    int bad4 = 0;
    {
        const float a1 = (float)(1 + lane), a2 = (float)(200 - lane), b1 = (float)(3 + 2 * lane), b2 = (float)(7 + lane % 5);
        f32x4 d = {0.0f, 0.0f, 0.0f, 0.0f};
        d = __builtin_amdgcn_mfma_f32_4x4x1f32(a1, b1, d, 4, 5, 0);
        d = __builtin_amdgcn_mfma_f32_4x4x1f32(a2, b2, d, 4, 12, 0);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float want = (float)(1 + 4 * 5 + i) * (float)(3 + 2 * lane) + (float)(200 - (4 * 12 + i)) * (float)(7 + lane % 5);
            bad4 += d[i] != want;
        }
    }
    // probes 5 / 6: the bf16 forms of the split-bf16 variant (small integers, exact in bf16): A[i][k] = 1 + ((i + 3k) % 7), B[k][j] = 1 + ((2j + k) % 5).
    //   5: 16x16x32 — lane l holds A[l & 15][8 (l >> 4) + e], B[8 (l >> 4) + e][l & 15], e = 0..7; D register r = D[4 (l >> 4) + r][l & 15]
    //   6: 32x32x16 — lane l holds A[l & 31][8 (l >> 5) + e], B[8 (l >> 5) + e][l & 31];        D register q = D[8 (q >> 2) + 4 (l >> 5) + (q & 3)][l & 31]
    int bad5 = 0, bad6 = 0;
    {
        bf16x8 a, b;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int k = 8 * (lane >> 4) + e;
            a[e] = (__bf16)(float)(1 + (((lane & 15) + 3 * k) % 7));
            b[e] = (__bf16)(float)(1 + ((2 * (lane & 15) + k) % 5));
        }
        f32x4 d = {0.0f, 0.0f, 0.0f, 0.0f};
        d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, d, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = 4 * (lane >> 4) + r, jj = lane & 15;
            int want = 0;
            for (int k = 0; k < 32; ++k) want += (1 + ((i + 3 * k) % 7)) * (1 + ((2 * jj + k) % 5));
            bad5 += d[r] != (float)want;
        }
    }
    {
        bf16x8 a, b;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int k = 8 * h + e;
            a[e] = (__bf16)(float)(1 + ((li + 3 * k) % 7));
            b[e] = (__bf16)(float)(1 + ((2 * li + k) % 5));
        }
        f32x16 d;
#pragma unroll
        for (int q = 0; q < 16; ++q) d[q] = 0.0f;
        d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, d, 0, 0, 0);
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int i = 8 * (q >> 2) + 4 * h + (q & 3);
            int want = 0;
            for (int k = 0; k < 16; ++k) want += (1 + ((i + 3 * k) % 7)) * (1 + ((2 * li + k) % 5));
            bad6 += d[q] != (float)want;
        }
    }
    const int t0 = (int)wave_sum((float)bad0), t1 = (int)wave_sum((float)bad1), t2 = (int)wave_sum((float)bad2);
    const int t3 = (int)wave_sum((float)bad3), t4 = (int)wave_sum((float)bad4), t5 = (int)wave_sum((float)bad5), t6 = (int)wave_sum((float)bad6);
    if (lane == 0) {
        report[0] = t0; report[1] = t1; report[2] = t2; report[3] = t3; report[4] = t4; report[5] = t5; report[6] = t6;
        for (int k = 7; k < 16; ++k) report[k] = 0;
    }
}

extern "C" int mi_selftest_mfma(int32_t* report, float* dump, void* stream) {
    MI_CHECK_ARG(report != nullptr, "report is NULL");
    selftest_kernel<<<1, 64, 0, (hipStream_t)stream>>>(report, dump);
    MI_LAUNCH_CHECK();
    return MI_OK;
}

// TEST HOOK (include/mi_rl.h, mi_test_contraction): D[16][16] = A[16][K] . B[K][16] by ONE wave with exactly the building blocks of the gradient kernels' 64 x 64
// contractions — mode F32: v_mfma_f32_16x16x4_f32, K / 4 chained k-steps (grad_kernel_f32); mode BF16X3: both operands through split8 (three bf16 parts), six products
// per 32 k-slots through bx_mac on v_mfma_f32_16x16x32_bf16 (grad_kernel_bx) — so that the variant's error bound can be tested on adversarial operands.
__global__ void __launch_bounds__(64) contraction_kernel(int mode, const float* __restrict__ A, const float* __restrict__ B, int K, float* __restrict__ D) {
    const int lane = threadIdx.x, j = lane & 15, g = lane >> 4;
    f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
    if (mode == MI_CONTRACTION_F32) {
        for (int s = 0; s < K / 4; ++s) acc = mfma16(A[j * K + 4 * s + g], B[(4 * s + g) * 16 + j], acc);   // A lane (i = j, k = 4s + g), B lane (k = 4s + g, col j)
    } else {
        for (int h = 0; h < K / 32; ++h) {   // lane holds A[j][32h + 8g + e], B[32h + 8g + e][j], e = 0..7
            f32x4 a0, a1, b0, b1;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                a0[e] = A[j * K + 32 * h + 8 * g + e]; a1[e] = A[j * K + 32 * h + 8 * g + 4 + e];
                b0[e] = B[(32 * h + 8 * g + e) * 16 + j]; b1[e] = B[(32 * h + 8 * g + 4 + e) * 16 + j];
            }
            const bx_parts ap = split8(a0, a1), bp = split8(b0, b1);
            acc = bx_mac(ap.p, bp, acc);
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) D[(4 * g + r) * 16 + j] = acc[r];   // D register r = D[4g + r][j]
}

extern "C" int mi_test_contraction(int mode, const float* A, const float* B, int K, float* D, void* stream) {
    MI_CHECK_ARG(mode == MI_CONTRACTION_F32 || mode == MI_CONTRACTION_BF16X3, "mode must be MI_CONTRACTION_F32 or MI_CONTRACTION_BF16X3");
    MI_CHECK_ARG(A && B && D, "NULL pointer");
    MI_CHECK_ARG(K >= 32 && K % 32 == 0 && K <= 4096, "K must be a multiple of 32 in [32, 4096]");
    contraction_kernel<<<1, 64, 0, (hipStream_t)stream>>>(mode, A, B, K, D);
    MI_LAUNCH_CHECK();
    return MI_OK;
}

__global__ void __launch_bounds__(256) tanh_kernel(const float* __restrict__ x, float* __restrict__ y, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] = mi_tanhf(x[i]);
}

extern "C" int mi_test_tanh(const float* x, float* y, int n, void* stream) {
    MI_CHECK_ARG(x && y && n > 0, "bad arguments");
    tanh_kernel<<<(n + 255) / 256, 256, 0, (hipStream_t)stream>>>(x, y, n);
    MI_LAUNCH_CHECK();
    return MI_OK;
}

MI_INSIDE_EXPORT(update)
