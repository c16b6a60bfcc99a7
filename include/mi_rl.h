/*
 * mi_rl.h — C ABI of the MI355X-native rollout + policy-update engine (libmirl.so).
 *
 * This is the drop-in boundary for the reference's PPO/CartPole hot path.  The reference
 * (qgallouedec/deep_rl) has no FFI of its own: the path sits behind three de-facto Python protocols
 * (gym-0.21 Env, nn.Module methods, bare torch.Tensor storage).  Each entry point
 * below names the reference lines it replaces; the deep_rl_amd Python modules bind them with ctypes and
 * re-presents the reference's Python surface (INTEGRATION.md shows the stub).
 *
 * Conventions
 *   - plain C: pointers + sizes, no torch / C++ types.  Every pointer marked "dev" is DEVICE memory
 *     owned by the caller (e.g. torch allocations passed by data_ptr()).
 *   - every call returns MI_OK (0) or a negative MI_E* code; mi_last_error() gives the text.
 *     Nothing throws, aborts, allocates or synchronises on the hot path: work is enqueued on the
 *     hipStream_t that is passed (as void*; NULL = the null stream).
 *   - an env handle is bound to the device that was current at mi_env_create and is not
 *     thread-safe (the reference's caller is single-threaded).
 *   - multi-GPU: one process per GPU, each with its own handle and env_id_base = rank * n_envs;
 *     the only exchange is the caller's all-reduce of `grads` (and of the advantage statistics)
 *     between mi_ppo_minibatch_grad and mi_clip_adam.
 *
 * Storage layout (reference ppo.py:93-98 with an env axis after time):
 *   observations f32 [(T+1), N, 4]   values f32 [(T+1), N]    actions i64 [(T+1), N]
 *   log_probs    f32 [(T+1), N]      rewards f32 [(T+1), N]   dones   f32 [(T+1), N]
 *   advantages / returns f32 [(T+1), N];  a flattened row index is t*N + n.
 *
 * Parameter layout: one flat f32 vector of MI_PPO_NPARAMS in the order of agent.parameters()
 * (ppo.py:34-47): actor {W1[64,4] b1[64] W2[64,64] b2[64] W3[2,64] b3[2]} then
 * critic {W1 b1 W2 b2 W3[1,64] b3[1]}, torch Linear row-major [out][in].
 *
 * RNG contract (production mode; in parity mode every random input is supplied explicitly):
 *   philox4x32-10, counter = {env_lo, env_hi, idx_lo, (idx_hi << 4) | stream}, key = {seed_lo, seed_hi}
 *   stream 0: reset noise   idx = episode index of that env; word i -> s_i = -0.05 + 0.1*((w_i+0.5)/2^32) (f64)
 *   stream 1: action draw   for the s-th action an env samples: idx = s >> 2, word w = w_(s & 3) (one Philox block feeds
 *             four consecutive steps); u = (w >> 8) / 2^24 (f32), action = #{j < n_actions-1 : u >= cumsum(probs)[j]}
 *   stream 2: permutation key for (env := update, idx := epoch): key = w_0 | (w_1 << 32); the
 *             permutation itself is the 6-round Feistel bijection with cycle walking of mi_make_perm.
 *   `env` is the GLOBAL env id (env_id_base + local index) so a trajectory does not depend on how
 *   envs are sharded over GPUs.  oracle/cpu_ref.c implements the same contract bit for bit.
 */
#ifndef MI_RL_H
#define MI_RL_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ABI version: bumped whenever a struct layout, a workspace size or a signature changes (101: mi_ppo_buffers_t gained episode_stats_next and the SAC
 * workspace grew in round 2; round 3 adds mi_sac_check / the workspace status words and mi_comm_info's comm_count; 102, round 4: mi_explained_var_parts,
 * mi_ppo_test_assume_sharded; 103: mi_sac_critic_update_deferred / mi_sac_critic_step / mi_sac_act_step_carry; 104: mi_env_episode_stats, episode statistics kept per workgroup; 105, round 5: the P2P carrier of mi_comm — mi_comm_p2p_alloc / _connect / _synthetic, mi_comm_check, mi_comm_carrier, mi_comm_p2p_set_colocated; mi_test_contraction; mi_sac_shadow_*; 106: mi_dueling_td_update; 107, round 6: mi_comm_poll / the P2P carrier's fail-safe, mi_comm_p2p_set_fused, mi_comm_test_set_seq, mi_per_td_update).  Bindings must compare mi_version()
 * with the MI_VERSION they were written against and refuse to run on a mismatch (deep_rl_amd/_native.py does). */
#define MI_VERSION 107
#define MI_PPO_NPARAMS 9155
#define MI_PPO_ACTOR_NPARAMS 4610
#define MI_OBS_DIM 4
#define MI_HIDDEN 64
#define MI_N_ACTIONS 2

enum { MI_OK = 0, MI_EINVAL = -1, MI_EHIP = -2, MI_ENOMEM = -3, MI_ESTATE = -4 };
enum { MI_ENV_CARTPOLE_V1 = 0 };  /* MI_ENV_PENDULUM_V1 = 1: see the SAC section */

typedef struct {
    int32_t env;   /* local env index */
    int32_t t;     /* rollout step at which the episode ended */
    float ret;     /* info["episode"]["r"] (ppo.py:130) */
    int32_t len;   /* info["episode"]["l"] */
} mi_episode_t;

int mi_version(void);
/* 12 hex digits of the sha256 over the CODE (comments and white space stripped: csrc/srcid.py) of the sources the library was built from (csrc/Makefile: ALLSRC, in that
 * order) — "unknown" for a build that bypassed the Makefile.
 * Profiles record it and bench.py prints it beside the static figures it quotes from them (roofline.traffic, kernel_device_ms_per_update). */
const char* mi_source_id(void);
const char* mi_last_error(void); /* thread-local, valid until the next call on this thread */

/* ---- gym-0.21 Env protocol for N envs (replaces TorchWrapper + gym.make + RecordEpisodeStatistics,
 *      ppo.py:10-22,79-84; arithmetic of gym==0.21 cartpole.py / time_limit.py) ------------------ */
int mi_env_create(int kind, int n_envs, uint64_t seed, uint64_t env_id_base, void** handle);
int mi_env_destroy(void* handle);
/* checkpoint / resume (absent in the reference): the complete env state — fp64 dynamics state, TimeLimit and
 * episode-statistics counters, and the RNG counters (episode, step) that key the counter-based draws — as one opaque device blob of
 * mi_env_state_bytes(handle) bytes.  Importing it into a handle created with the same (kind, n_envs, seed, env_id_base) continues the
 * run bit for bit. */
size_t mi_env_state_bytes(void* handle);
int mi_env_export_state(void* handle, void* dst, void* stream);
int mi_env_import_state(void* handle, const void* src, void* stream);
/* Episode statistics of the LAST rollout / acting call on this handle that was given episode_stats == NULL (mi_ppo_rollout*, mi_ppo_update*, mi_dqn_act_steps):
 * out i32 [4] (device memory, or pinned host memory the device can write: a training loop reads the summary without a copy behind the launch) = {finished episodes,
 * sum of their lengths, longest, 0}, one small launch on `stream`.  Such a call keeps the statistics per workgroup inside the
 * handle with plain stores instead of accumulating them with atomics: agent-scope atomics on one address are performed one after the other at the memory side
 * (~8.5 ns each) and a launch is not over before the last one — 26 us per 4096-env rollout, 9 us per DQN acting launch.  With a buffer the calls still fill it before
 * they return (large launches: per-workgroup statistics summed by a small launch right behind; small launches and the episode log: atomics).
 * MI_ESTATE when no such call has run yet. */
int mi_env_episode_stats(void* handle, int32_t* out, void* stream);
/* env.reset() (ppo.py:21,101).  obs: dev f32 [N,4].  forced_state: dev f64 [N,4] or NULL (keyed noise). */
int mi_env_reset(void* handle, float* obs, const double* forced_state, void* stream);
/* env.step(action) followed by ppo.py:128-129's `if done: observation = env.reset()`.
 * actions dev i64 [N]; forced_reset dev f64 [N,4] or NULL; outputs (all dev, all required):
 * obs f32 [N,4] (the reset observation where done), reward f32 [N], done u8 [N] (terminated OR
 * truncated), truncated u8 [N] (info["TimeLimit.truncated"]), fin_ret f32 [N] / fin_len i32 [N]
 * (episode statistics of envs that finished this step, else 0). */
int mi_env_step(void* handle, const int64_t* actions, const double* forced_reset, float* obs, float* reward,
                uint8_t* done, uint8_t* truncated, float* fin_ret, int32_t* fin_len, void* stream);
/* the same for either env kind (actions: dev i64 [N] for CartPole-v1, dev f32 [N] for Pendulum-v1; forced_reset f64 [N,4] / [N,2]) with one more output:
 * raw_obs dev f32 [N,obs_dim] (nullable) = the observation gym's env.step returned BEFORE the script's `if done: observation = env.reset()` — the terminal
 * observation the fused rollout / acting kernels never store.  Used by deep_rl_amd/trace.py to write a run in the golden fixtures' key layout. */
int mi_env_step_ex(void* handle, const void* actions, const double* forced_reset, float* obs, float* reward, uint8_t* done, uint8_t* truncated,
                   float* fin_ret, int32_t* fin_len, float* raw_obs, void* stream);
/* debug/test: copy the float64 state out as dev f64 [N,4] and the TimeLimit counters as dev i32 [N] (nullable). */
int mi_env_get_state(void* handle, double* state, int32_t* elapsed, void* stream);

/* ---- ActorCritic forward on an arbitrary batch (ppo.py:49-54): logits dev f32 [n,2] and/or value dev f32 [n] */
int mi_ppo_forward(const float* params, const float* obs, int n, float* logits, float* value, void* stream);

/* ---- the rollout loop ppo.py:110-141 for N envs, T steps, one launch.
 * obs_cur dev f32 [N,4]: carried-over `observation` (in/out).  Six storage buffers as above.
 * Parity inputs (each NULL in production): forced_actions dev i64 [T,N]; forced_uniforms dev f32 [T,N];
 * forced_resets dev f64 [T,N,4] (state used if env n resets at step t).
 * episodes dev mi_episode_t [max_ep] + episode_stats dev i32 [4] (both nullable).  The call resets episode_stats and
 * the kernel accumulates {[0] number of finished episodes, [1] sum of their lengths, [2] longest, [3] slots handed out in
 * `episodes`} (CartPole: return == length); only the first max_ep episodes are stored individually.  episode_stats == NULL: the statistics stay in the handle,
 * mi_env_episode_stats sums them on request (the fast form for large N). */
int mi_ppo_rollout(void* handle, const float* params, int T, float* obs_cur, float* observations, float* values,
                   int64_t* actions, float* log_probs, float* rewards, float* dones, const int64_t* forced_actions,
                   const float* forced_uniforms, const double* forced_resets, mi_episode_t* episodes,
                   int32_t* episode_stats, int max_ep, void* stream);
/* rollout + GAE in one launch (ppo.py:110-151): each rollout workgroup scans its own envs when its last step is done (T <= 128; beyond
 * that mi_gae runs as a launch of its own).  Production RNG only.  advantages / returns as for mi_gae, bit-identical. */
int mi_ppo_rollout_gae(void* handle, const float* params, int T, float* obs_cur, float* observations, float* values,
                       int64_t* actions, float* log_probs, float* rewards, float* dones, mi_episode_t* episodes,
                       int32_t* episode_stats, int max_ep, float gamma, float gae_lambda, float* advantages, float* returns, void* stream);

/* ---- GAE reverse scan ppo.py:144-151 (expression order preserved, no FMA contraction) */
int mi_gae(const float* rewards, const float* dones, const float* values, int T, int N, float gamma, float lam,
           float* advantages, float* returns, void* stream);

/* ---- minibatch indices (replaces np.random.permutation, ppo.py:155): out dev i32 [n] = Feistel bijection on [0,n) */
int mi_make_perm(uint32_t n, uint64_t key, int32_t* out, void* stream);
uint64_t mi_perm_key(uint64_t seed, uint64_t update, uint64_t epoch); /* host helper, RNG contract stream 2 */

/* ---- advantage statistics of n_mb consecutive minibatches of `mb` indices each (ppo.py:169):
 * sums dev f64 [n_mb,3] = {sum a, sum a^2, count}; the caller may all-reduce(SUM) them across ranks. */
int mi_adv_stats(const float* advantages, const int32_t* idx, int mb, int n_mb, double* sums, void* stream);

/* ---- loss + gradient of one minibatch (ppo.py:159-190).  idx dev i32 [mb] (flattened rows, t < T);
 * adv_sums dev f64 [3] as produced by mi_adv_stats; inv_count = 1 / (rows the means are taken over,
 * i.e. world_size * mb).  Outputs: grads dev f32 [MI_PPO_NPARAMS] = d loss / d params (this rank's
 * share, already scaled by inv_count, so a SUM all-reduce gives the global gradient);
 * loss_terms dev f32 [4] = {pg_loss, entropy, v_loss, loss} shares scaled the same way.
 * workspace: dev, at least mi_ppo_workspace_bytes() bytes.  A count of ONE row (mb == 1 on one rank) gives NaN gradients and loss terms: the reference normalises
 * with torch's unbiased std, NaN for one element (ppo.py:169), and so does this — a result, not an error code. */
size_t mi_ppo_workspace_bytes(void);
int mi_ppo_minibatch_grad(const float* params, const float* observations, const int64_t* actions,
                          const float* log_probs, const float* advantages, const float* returns, const float* values,
                          const int32_t* idx, int mb, const double* adv_sums, float clip_coef, float ent_coef,
                          float vf_coef, double inv_count, void* workspace, float* grads, float* loss_terms,
                          void* stream);

/* ---- EXPERIMENT, off by default: which matrix pipe the gradient kernel's three 64 x 64 contractions (layer 2 forward, its input gradient, its
 * weight gradient; the nn.Linear(64, 64) of ppo.py:56-57 / :62-63 under loss.backward(), ppo.py:190) run on.  Process-wide; applies to every later
 * mi_ppo_minibatch_grad / mi_ppo_update / mi_ppo_update_sharded launch.
 *   MI_CONTRACTION_F32     exact f32 products and sums on v_mfma_f32_16x16x4_f32 (default; what every headline number is measured on)
 *   MI_CONTRACTION_BF16X3  both operands split into three bf16 parts (x = hi + mid + lo to 2^-27 |x|), the six products above 2^-24 accumulated in
 *                          f32 on v_mfma_f32_16x16x32_bf16 / 32x32x16_bf16: f32-grade results (every parity tolerance of the f32 path holds
 *                          unchanged, tests/test_gpu_contraction.py), not bit-identical to the f32 path. */
enum { MI_CONTRACTION_F32 = 0, MI_CONTRACTION_BF16X3 = 1 };
int mi_ppo_set_contraction(int mode);
int mi_ppo_get_contraction(void);
/* SPECIFIED ERROR BOUND of MI_CONTRACTION_BF16X3 (round 5; an opt-in mode, never the headline).  For one contraction y = sum_k a_k b_k over K terms, finite f32
 * operands, no overflow of |a_k b_k| or of the sum:
 *     | y_bf16x3 - y_exact |  <=  MI_BF16X3_REL_BOUND * max(1, K / 64) * sum_k |a_k| |b_k|  +  K * MI_BF16X3_ABS_FLOOR          (K = 64 in the gradient kernels)
 * — a bound relative to the sum of ABSOLUTE products, as for any floating-point dot product (cancellation in the result is not the variant's: the exact f32 MFMA
 * path obeys the same form with 2^-24 (K / 4 + 1) in place of the constant), from: split residual <= 2^-27 |x|, dropped part products (mid.lo, lo.mid, lo.lo)
 * <= 2^-25 |a b|, f32 accumulation of 6 K / 32 matrix instructions.  The absolute floor covers operands and products in the subnormal range, where a bf16 part
 * or an MFMA product may be flushed.  tests/test_gpu_bf16x3_bound.py drives mi_test_contraction — the gradient kernels' own split and multiply-accumulate code on one
 * 16 x K x 16 tile — with operands spanning 2^+-20 per element, cancelling pairs, subnormals and exact integers, and prints the worst observed ratio (round 5, K = 64:
 * 2^-21.2 for bf16x3 against 2^-21.5 for the exact-f32 MFMA path on the same operands). */
#define MI_BF16X3_REL_BOUND 9.5367431640625e-07   /* 2^-20 */
#define MI_BF16X3_ABS_FLOOR 1.1754943508222875e-38 /* 2^-126 */
/* TEST HOOK: D dev f32 [16][16] = A dev f32 [16][K] . B dev f32 [K][16] computed by one wave with the building blocks of the selected mode (K a multiple of 32) */
int mi_test_contraction(int mode, const float* A, const float* B, int K, float* D, void* stream);

/* ---- clip_grad_norm_(max_norm) + Adam step (ppo.py:191-192, torch single-tensor Adam).
 * step is 1-based.  grad_norm dev f32 [1] nullable: receives the pre-clip total norm. */
int mi_clip_adam(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int n, int64_t step, double lr,
                 double beta1, double beta2, double eps, float max_norm, float* grad_norm, void* stream);

/* ---- explained variance ppo.py:194-195 over n = (T+1)*N entries: out dev f64 [1] */
int mi_explained_var(const float* values, const float* returns, size_t n, double* out, void* stream);
/* the same statistic over the rows of ALL ranks of a sharded run (ppo.py:194-195 is over the whole batch), as two launches around the caller's two SUM all-reduces:
 *   means == NULL:          out dev f64 [2] = {sum values, sum (values - returns)} of this rank's n rows          -> all-reduce, divide by the global row count
 *   means dev f64 [2] given (the GLOBAL means): out = {sum (values - m0)^2, sum ((values - returns) - m1)^2}      -> all-reduce
 * explained_var = 1 - out[1] / out[0] of the reduced sums (the unbiased 1 / (n - 1) factors cancel); NaN if out[0] == 0. */
int mi_explained_var_parts(const float* values, const float* returns, size_t n, const double* means, double* out, void* stream);

/* ---- every epoch's keyed permutation (ppo.py:155) and per-minibatch advantage sums (ppo.py:169) of one update at once (they depend on the
 * advantages and the keys only): perm_all dev i32 [epochs, n_rows], sums_all dev f64 [epochs, n_minibatch, 3] (local sums; sharded runs
 * all-reduce them once per update). */
int mi_ppo_perms_and_stats(uint64_t seed, int update_index, int epochs, int n_rows, int n_minibatch, const float* advantages, int32_t* perm_all,
                           double* sums_all, void* stream);

/* ---- one whole outer update (ppo.py:105-192) enqueued back to back on `stream`, production RNG,
 * single rank (no collective).  All pointers dev.  perm: i32 [update_epochs, T*N] (every epoch's permutation is drawn before the first
 * optimizer step); adv_sums: f64 [update_epochs*n_minibatch*3].  Launch sequence: rollout+GAE, permutations+statistics, then per optimizer
 * step {gradient, slab sum} — the clip + Adam of step k rides on the weight staging of gradient launch k+1 (the stepped state ping-pongs
 * between two spare sets kept in the last 16 slabs of `workspace`) — and one clip + Adam launch for the last step, which lands the state
 * back in params / exp_avg / exp_avg_sq.  grads / loss_terms / grad_norm hold the LAST minibatch's values on return. */
typedef struct {
    float* params; float* exp_avg; float* exp_avg_sq; float* grads; float* loss_terms; float* grad_norm;
    float* obs_cur; float* observations; float* values; int64_t* actions; float* log_probs; float* rewards;
    float* dones; float* advantages; float* returns; int32_t* perm; double* adv_sums; void* workspace;
    mi_episode_t* episodes; int32_t* episode_stats; int32_t max_ep;
    int32_t* episode_stats_next;   /* nullable.  Given (a SECOND dev i32 [4]): episode_stats must be ZERO on entry and is not reset by a launch of its own; the
                                    * rollout launch zeroes episode_stats_next for the next update, which passes the two swapped (saves one launch per update) */
} mi_ppo_buffers_t;
typedef struct {
    int32_t T, n_minibatch, update_epochs, update_index; int64_t opt_step; /* optimizer steps done so far */
    float gamma, gae_lambda, clip_coef, ent_coef, vf_coef, max_grad_norm;
    double lr, beta1, beta2, eps;
} mi_ppo_hparams_t;
int mi_ppo_update(void* handle, const mi_ppo_buffers_t* buf, const mi_ppo_hparams_t* hp, void* stream);

/* ---- multi-GPU: one process per GPU, envs sharded (rank r owns global envs [r*N, (r+1)*N): env_id_base = r*N), parameters
 * and Adam state replicated.  The exchange runs straight on RCCL (rccl.h, bound at run time), in-stream:
 *   mi_comm_unique_id   rank 0 draws the 128-byte ncclUniqueId; the caller ships it to the other ranks (torch.distributed / file / MPI);
 *   mi_comm_create      collective: every rank calls it with the same id (ncclCommInitRank on the CURRENT device) -> opaque handle;
 *   mi_comm_allreduce_sum  in-place SUM all-reduce of n f32 (dtype 0) / f64 (dtype 1) elements of a device buffer on `stream`;
 *   mi_ppo_update_sharded  mi_ppo_update with the collectives of the sharded path enqueued between its launches (replaces the host-side
 *       sequence of ppo.py:154-192 with a gradient exchange between backward and clip_grad_norm_):  rollout+GAE, permutations + LOCAL
 *       advantage sums, ONE all-reduce of adv_sums (f64 [update_epochs*n_minibatch*3]), then per optimizer step {gradient (share scaled by
 *       1/(world*mb)), slab sum, ONE all-reduce of the MI_PPO_NPARAMS gradient + 4 loss terms}; clip + Adam of step k rides on gradient
 *       launch k+1 as in mi_ppo_update and runs identically on every rank.  buf->loss_terms MUST be buf->grads + MI_PPO_NPARAMS (one
 *       buffer, one collective).  comm == NULL or world 1: exactly mi_ppo_update. */
#define MI_COMM_ID_BYTES 128
int mi_comm_unique_id(void* id128);
int mi_comm_create(const void* id128, int world_size, int rank, void** comm);
int mi_comm_destroy(void* comm);
/* world_size / rank: as given to mi_comm_create; rccl_version: ncclGetVersion; comm_count: the rank count RCCL itself reports for the communicator
 * (ncclCommCount; -1 if the entry point is missing).  Any out pointer may be NULL. */
int mi_comm_info(void* comm, int* world_size, int* rank, int* rccl_version, int* comm_count);
int mi_comm_allreduce_sum(void* comm, void* buf, size_t n, int dtype, void* stream);
/* The second carrier behind the same handle: a ONE-SHOT peer-to-peer all-reduce over hipIpc-mapped inboxes (round 5; csrc/mi_comm.hip).  Every rank owns an inbox in
 * uncached device memory, mapped into every peer: 256 header bytes (the epoch change's barrier lines) + [2 parities][world] slots of 2 * max_bytes each — a slot holds the
 * message as 8-byte LINES {payload word, sequence number of the all-reduce}, so the footprint is 256 + 2 * world * 2 * max_bytes bytes per rank.  One all-reduce = ONE
 * launch: every 32-bit word of the local share is stored, TOGETHER with the sequence number, as one line into slot (parity, rank) of every rank's inbox; the launch then
 * polls (bounded: MIRL_P2P_TIMEOUT_MS, default 30,000) the world's lines of its own elements in its own inbox until each carries the sequence number — the payload is
 * its own "arrived" flag: no fence, no flag word, no second round trip — and sums them IN RANK ORDER -> bitwise the same result on every rank by construction, for any
 * world size (a ring's grouping depends on the rank); at world 2 also bitwise RCCL's / gloo's a + b.  The parity flips with every all-reduce (a bit of its own); when
 * the 32-bit sequence number is used up, the ranks change the epoch in-stream (own lines cleared, a barrier through the header, numbers restart at 1).  Works with two
 * ranks on ONE device (RCCL refuses that), which is how a one-GPU box runs the one-call *_sharded routes at world_size 2 (tests/test_gpu_p2p.py).  world_size <= 8.
 * FAIL-SAFE.  A wait that runs out (a peer that stalls beyond the budget) sets the rank's status word (plain device memory: only its own launches read it) and a
 * host-pinned mirror of it.  From then on every optimizer step
 * that would consume an exchanged gradient — PPO's owed clip + Adam and its last step, DQN's / PER's clip + Adam, SAC's Adam / polyak / alpha steps on the *_sharded
 * routes — reads that word with its state and is WITHHELD (parameters, moments and targets stay as they were in front of the failed exchange), and every later
 * mi_*_sharded / mi_comm_allreduce_sum call returns MI_ESTATE at its entry (a plain host load of the mirror: no synchronisation).  The communicator is dead after
 * that (destroy it); what is lost is the update, never the parameters.
 *   mi_comm_p2p_alloc    allocates this rank's inbox on the CURRENT device for messages of <= max_bytes and writes its 64-byte hipIpcMemHandle_t; the caller ships the
 *                        handles of all ranks to all ranks (torch.distributed all_gather / file / MPI) ...
 *   mi_comm_p2p_connect  ... and hands them over in rank order (world_size * 64 bytes): maps the peers' inboxes.  The handle then works wherever an RCCL one does
 *                        (mi_comm_allreduce_sum, mi_*_sharded).  All collectives of one communicator must be enqueued on streams ordered with each other, by every rank
 *                        in the same order.  Put a barrier in front of mi_comm_destroy: a peer may still be storing into this rank's inbox.  The inbox itself is
 *                        PARKED by mi_comm_destroy, not freed, and handed to the next communicator of the same size: pages of an uncached, polled inbox that were
 *                        recycled into cached memory returned stale lines to their next owner on this chip (csrc/mi_comm.hip, p2p_park).
 *   mi_comm_p2p_synthetic  ONE process plays world_size ranks into its own inbox (slot 0 = its share, the others zeros: results unchanged): the stores, polls and the
 *                        world-slot sum of a world_size-rank exchange minus the links — for timing on a one-GPU box (bench.py `sharded_route`).
 *                        mi_comm_info reports world_size 1 (the sharded calls scale their shares by 1 / world_size: it IS one rank) and comm_count = the ranks played.
 *   mi_comm_check        host-synchronising.  MI_OK, or MI_ESTATE when a wait ran out (mi_last_error names the ranks that never arrived).  Always MI_OK for RCCL.
 *   mi_comm_poll         the same answer WITHOUT synchronising (the mirror): what the *_sharded calls test at their entry; callers may poll it per update.
 *   mi_comm_carrier      0 = RCCL, 1 = P2P. */
#define MI_COMM_IPC_BYTES 64
int mi_comm_p2p_alloc(int world_size, int rank, size_t max_bytes, void** comm, void* ipc_handle64);
int mi_comm_p2p_connect(void* comm, const void* ipc_handles);
int mi_comm_p2p_synthetic(int world_size, size_t max_bytes, void** comm);
int mi_comm_check(void* comm);
int mi_comm_poll(void* comm);
int mi_comm_carrier(void* comm);
/* The LARGEST number of ranks of the communicator that share one device (default 1 = one rank per GPU) — the SAME value on every rank.  With more than two,
 * mi_ppo_update_sharded takes a stand-alone launch per all-reduce instead of the exchange inside the slab sum: three or more ranks spin-waiting in 145 x 1,024-thread
 * workgroups fill the chip, and the rank they wait for cannot be scheduled (test placements only).  The two forms publish the gradient's lines in different orders (slab
 * order / parameter order) under one sequence number, so ALL ranks must take the same one: deep_rl_amd.dist derives the count from the gathered (host, device uuid) of every
 * rank and, when MIRL_P2P_FUSED is set, checks that every rank has the same setting and fixes it with mi_comm_p2p_set_fused (-1 stand-alone, 1 in-launch, 0 = by count). */
int mi_comm_p2p_set_colocated(void* comm, int max_ranks_on_one_device);
int mi_comm_p2p_set_fused(void* comm, int mode);
/* TEST HOOK: presets the P2P carrier's sequence number (the last all-reduce's number inside the epoch; the same value on every rank, between all-reduces) so that a test
 * crosses the epoch change — at 0xFFFFFFF0 — within a few exchanges. */
int mi_comm_test_set_seq(void* comm, uint32_t seq);
int mi_ppo_update_sharded(void* handle, const mi_ppo_buffers_t* buf, const mi_ppo_hparams_t* hp, void* comm, void* stream);
/* TEST HOOK (process-wide; 0 = off): mi_ppo_update / mi_ppo_update_sharded behave as at world_size > 1 in everything but the collective — the owed optimizer steps
 * recompute the clip coefficient from the (all-reduced) gradient itself instead of reading the slab sum's block sums, the one branch a single-GPU run never takes.
 * Results are bitwise those of the default route (the two evaluations of the norm are one expression tree); lets a one-GPU box test and time the sharded launches
 * (tests/test_gpu_sharded_route.py, bench.py `sharded_route`). */
int mi_ppo_test_assume_sharded(int on);

/* =====================================================================================================================
 * DQN (reference deep_rl/dqn.py; SURVEY.md §8a d1-d8, BASELINE config 3).
 * QNetwork 4 -> 120 -> 84 -> 2 ReLU; flat f32 parameters in q_network.parameters() order: W1[120,4] b1[120] W2[84,120] b2[84]
 * W3[2,84] b3[2] = MI_DQN_NPARAMS.  Replay storage (dqn.py:73-76 with an env axis, ring semantics of iqn.py:174-232):
 *   observations f32 [slots, N, 4], actions i64 [slots, N], rewards f32 [slots, N], terminated u8 [slots, N]
 * time slot g % slots holds obs_g and the action taken from it; slot (g+1) % slots holds the resulting reward / terminated /
 * next observation (the RESET observation after a done).  A flattened index is slot*N + env; its successor is
 * ((slot+1) % slots)*N + env.  slots = total_timesteps + 1 reproduces the reference's linear storage.
 * RNG contract additions: stream 3 = exploration (idx = env step counter; u = (w0 >> 8)/2^24 is compared with epsilon, w1 & 1 is
 * the random action); stream 4 = minibatch sampling (env := update index, idx := row b; index = (w0 | w1 << 32) mod upper).
 * ===================================================================================================================== */
#define MI_DQN_NPARAMS 10934
/* QNetwork.forward (dqn.py:35-36): q dev f32 [n,2] */
int mi_dqn_forward(const float* params, const float* obs, int n, float* q, void* stream);
/* n_steps iterations of the acting half of the loop (dqn.py:84-108) for the handle's N envs, one launch: epsilon-greedy
 * (random while global_step < learning_starts or u < epsilon(global_step), else argmax Q), env.step with auto-reset, ring store.
 * global_step = time steps already taken.  obs_cur dev f32 [N,4] carried in/out.  forced_actions dev i64 [n_steps,N] /
 * forced_resets dev f64 [n_steps,N,4] nullable (parity mode).  episodes / episode_stats as for mi_ppo_rollout (t = step index
 * within the call). */
int mi_dqn_act_steps(void* handle, const float* params, int n_steps, int64_t global_step, int64_t slots, int64_t learning_starts,
                     double start_e, double end_e, double exploration_fraction, int64_t total_timesteps, float* obs_cur,
                     float* observations, int64_t* actions, float* rewards, uint8_t* terminated, const int64_t* forced_actions,
                     const double* forced_resets, mi_episode_t* episodes, int32_t* episode_stats, int max_ep, void* stream);
/* the same with the statistics double-buffered by the caller (saves the tiny launch that resets them): episode_stats must be ZERO on entry;
 * zero_next (dev i32 [4], another buffer) is zeroed by this launch for the next acting call, which passes it as its episode_stats */
int mi_dqn_act_steps2(void* handle, const float* params, int n_steps, int64_t global_step, int64_t slots, int64_t learning_starts,
                      double start_e, double end_e, double exploration_fraction, int64_t total_timesteps, float* obs_cur,
                      float* observations, int64_t* actions, float* rewards, uint8_t* terminated, const int64_t* forced_actions,
                      const double* forced_resets, mi_episode_t* episodes, int32_t* episode_stats, int max_ep, int32_t* zero_next, void* stream);
/* batch_inds = randint(upper, size=batch) (dqn.py:116 / iqn.py:225-226): idx dev i64 [batch], uniform in [0, upper_flat) */
int mi_dqn_sample(uint64_t seed, uint64_t update_index, int64_t upper_flat, int batch, int64_t* idx, void* stream);
/* TD loss + gradient of one batch (dqn.py:118-128): grads dev f32 [MI_DQN_NPARAMS] = d loss/d params scaled by inv_count
 * (1/(world*batch)), loss dev f32 [1] = sum((td-old)^2)*inv_count.  workspace: mi_dqn_workspace_bytes(batch) bytes. */
size_t mi_dqn_workspace_bytes(int batch);
int mi_dqn_td_grad(const float* params, const float* target_params, const float* observations, const int64_t* actions,
                   const float* rewards, const uint8_t* terminated, const int64_t* idx, int batch, int n_envs, int64_t slots,
                   float gamma, double inv_count, void* workspace, float* grads, float* loss, void* stream);
/* single-process fusion: the same two launches, the second of which also applies optimizer.step() (torch Adam, no clipping) to every
 * gradient element it has just summed — bit-identical to mi_dqn_td_grad (or mi_per_td_grad when weights / td_abs are given) followed by
 * mi_clip_adam with max_norm = +inf.  inv_count = 1 / batch.  sample_upper > 0: the TD launch draws idx itself (the mi_dqn_sample contract
 * with (sample_seed, sample_update), bit-identical) and stores it in idx — no sampling launch either; 0: idx is an input. */
int mi_dqn_td_update(float* params, const float* target_params, const float* observations, const int64_t* actions,
                     const float* rewards, const uint8_t* terminated, int64_t* idx, int batch, int n_envs, int64_t slots,
                     float gamma, const float* weights, float* td_abs, void* workspace, float* grads, float* loss, float* exp_avg, float* exp_avg_sq,
                     int64_t step, double lr, double beta1, double beta2, double eps, uint64_t sample_seed, uint64_t sample_update, int64_t sample_upper,
                     void* stream);
/* sharded runs (one process per GPU), ONE call per optimisation step (dqn.py:116-133 with the gradient exchange between backward and step): TD gradient
 * share scaled by 1 / (world * batch), slab sum, in-stream SUM all-reduce of gradbuf = dev f32 [MI_DQN_NPARAMS + 2] {grads, loss, pad}, then mi_clip_adam(max_norm,
 * grad_norm nullable).  weights / td_abs: PER row weights and |td| out (nullable together).  sample_upper > 0 (DQN only): the TD launch draws the batch itself (the keyed
 * uniform draw of mi_dqn_sample(sample_seed, sample_update, sample_upper)) and writes it to idx; 0: idx is given.  The same arithmetic as mi_dqn_td_grad
 * (mi_per_td_grad) + caller all-reduce + mi_clip_adam: bit-identical.  comm NULL or world 1: no collective.
 * On the P2P carrier with max_norm = +inf (dqn.py / per.py clip nothing; round 6) the slab-sum launch carries the exchange AND the step — the thread that has summed
 * gradient element p exchanges it (rank-ordered sum) and applies Adam to it: two launches per step, as in a single process, plus one exchange latency; grad_norm is
 * not written on that route (no norm is computed).  With RCCL, or a finite max_norm (the clip coefficient is a grid-wide dependency): slab sum, all-reduce, clip + Adam. */
int mi_dqn_td_update_sharded(float* params, const float* target_params, const float* observations, const int64_t* actions, const float* rewards,
                             const uint8_t* terminated, int64_t* idx, int batch, int n_envs, int64_t slots, float gamma, const float* weights, float* td_abs,
                             void* workspace, float* gradbuf, float* exp_avg, float* exp_avg_sq, int64_t step, double lr, double beta1, double beta2, double eps,
                             float max_norm, float* grad_norm, uint64_t sample_seed, uint64_t sample_update, int64_t sample_upper, void* comm, void* stream);
/* optimizer.step() (dqn.py:131-133) = mi_clip_adam(..., n = MI_DQN_NPARAMS, eps = 1e-8, max_norm = +inf);
 * target_network.load_state_dict (dqn.py:136-137) = a device-to-device copy of the flat vector by the caller. */

/* ---- Dueling head (reference deep_rl/dueling_dqn.py:24-40: values + (advantages - mean advantages)) as an epilogue on the DQN calls.
 * Dueling flat layout = q_network1.parameters(): feature W1[120,4] b1 W2[84,120] b2 | value W[1,84] b | advantage W[2,84] b[2].
 * The head is linear in the features, so mi_dueling_pack writes the EQUIVALENT plain-DQN parameter vector (W3[a] = Wv + Wa[a] - mean Wa,
 * b3 likewise) that mi_dqn_forward / mi_dqn_act_steps / mi_dqn_td_grad consume (both for the online and the target copy), and
 * mi_dueling_unpack_grads maps the plain gradient back (dWv = sum_a g3[a], dWa[k] = g3[k] - mean_a g3[a]; features unchanged). */
#define MI_DUELING_NPARAMS 11019
int mi_dueling_pack(const float* dueling_params, float* dqn_params, void* stream);
int mi_dueling_unpack_grads(const float* dqn_grads, float* dueling_grads, void* stream);
/* One optimisation step of dueling_dqn.py (:109-129) as ONE call (single process, no gradient clipping): the TD launch on the plain-DQN images `params_img` /
 * `target_img` (mi_dueling_pack's output), and in the launch that sums the gradient slabs the gradient mapped back to the dueling layout (-> dueling_grads), Adam on
 * `dueling_params` (exp_avg / exp_avg_sq: dev f32 [MI_DUELING_NPARAMS]) and `params_img` rewritten — bit-identical to mi_dqn_td_grad(images) + mi_dueling_unpack_grads +
 * mi_clip_adam(max_norm = +inf) + mi_dueling_pack.  grads: dev f32 [MI_DQN_NPARAMS] (the plain gradient).  sample_upper as in mi_dqn_td_update.  Batches of 512 rows
 * and more keep the many-slab sum and run the three epilogue launches behind it, inside this call. */
int mi_dueling_td_update(float* params_img, const float* target_img, const float* observations, const int64_t* actions, const float* rewards,
                         const uint8_t* terminated, int64_t* idx, int batch, int n_envs, int64_t slots, float gamma, void* workspace, float* grads, float* loss,
                         float* dueling_params, float* dueling_grads, float* exp_avg, float* exp_avg_sq, int64_t step, double lr, double beta1, double beta2,
                         double eps, uint64_t sample_seed, uint64_t sample_update, int64_t sample_upper, void* stream);

/* ---- Prioritized replay (reference deep_rl/per.py, run on CartPole-v1) as epilogues on the DQN calls.
 * priorities: dev f32 [slots, N] beside the replay ring; max_priority: dev f32 [1] (per.py:84, initial 1e-2); owner: dev i32 [slots*N],
 * all -1 between calls; workspace: mi_per_workspace_bytes(slots*N).
 * per.py:128 draws batch_inds with torch.multinomial(priorities) (O(buffer), host generator); here a keyed three-level prefix-sum descent
 * with a fixed evaluation order (64-entry chunk sums s0, sums s1 of 64 of those, their running sums P[j] = s1[0] + ... + s1[j] with total = P[last]; all sequential, f64;
 * u = 53-bit Philox uniform of (seed, update_index, b, stream 7), x = u * total).  Level 1: m = the number of j < n1 - 1 with x >= P[j], x -= P[m - 1] (round 6: found by
 * binary search — until round 5 the contract walked s1 by up to n1 - 1 dependent subtractions per draw); level 0 and the priorities: walk while x >= the next value,
 * subtracting it, inside the chosen group / chunk; entries with priority 0 at the end of the walk are skipped downwards.  The oracle implements the same contract bit for bit.  sample = 0 keeps the caller's idx
 * (parity runs) and only computes the importance weights (count * p_i^alpha / sum p^alpha)^-beta / max (per.py:131,145-146; count = the
 * reference's global_step = transitions stored).  mi_per_mark: priorities of the n_steps just written slots = max_priority (per.py:106),
 * the ring's write head = 0.  mi_per_td_grad = mi_dqn_td_grad with loss = mean(weights * td^2) and td_abs[b] = |td_b| out (per.py:139,147).
 * mi_per_update_priorities: priorities[idx] = td_abs (last duplicate wins), max_priority = max(max_priority, those) (per.py:141-142). */
size_t mi_per_workspace_bytes(int64_t capacity);
int mi_per_mark(float* priorities, int n_envs, int64_t slots, int64_t global_step, int n_steps, const float* max_priority, void* stream);
int mi_per_sample(uint64_t seed, uint64_t update_index, const float* priorities, int64_t n_valid, int64_t capacity, double count, float alpha,
                  float beta, int batch, int sample, void* workspace, int64_t* idx, float* weights, void* stream);
int mi_per_td_grad(const float* params, const float* target_params, const float* observations, const int64_t* actions,
                   const float* rewards, const uint8_t* terminated, const int64_t* idx, int batch, int n_envs, int64_t slots,
                   float gamma, double inv_count, const float* weights, float* td_abs, void* workspace, float* grads, float* loss, void* stream);
int mi_per_update_priorities(float* priorities, const int64_t* idx, const float* td_abs, int batch, int32_t* owner, float* max_priority, void* stream);
/* Incremental form of the same sampler (what PERDQNEngine runs): the level-0 chunk sums in `workspace` are kept CURRENT by the calls that
 * change priorities — each RECOMPUTES the chunks it touches with mi_per_sample's own operations, so the sums, the indices drawn and the
 * weights are bit-identical to the calls above — and the draw takes level 1 + totals into its own launch: per update 3 launches that touch
 * O(new rows + batch) priorities instead of 5 that scan the whole ring (per.py:128 is an O(buffer) torch.multinomial).
 *   mi_per_sums_refresh            full level-0 pass (after priorities were written by anything else, e.g. a checkpoint load); a zero-filled
 *                                  workspace is current for a zero-filled ring
 *   mi_per_mark_sums               = mi_per_mark + the sums of the touched chunks
 *   mi_per_sample_current          = mi_per_sample, trusting the level-0 sums
 *   mi_per_update_priorities_sums  = mi_per_update_priorities + the sums of the touched chunks (capacity = slots * N) */
int mi_per_sums_refresh(const float* priorities, int64_t capacity, float alpha, void* workspace, void* stream);
int mi_per_mark_sums(float* priorities, int n_envs, int64_t slots, int64_t global_step, int n_steps, const float* max_priority, float alpha, void* workspace,
                     void* stream);
int mi_per_sample_current(uint64_t seed, uint64_t update_index, const float* priorities, int64_t n_valid, int64_t capacity, double count, float alpha,
                          float beta, int batch, int sample, void* workspace, int64_t* idx, float* weights, void* stream);
int mi_per_update_priorities_sums(float* priorities, const int64_t* idx, const float* td_abs, int batch, int32_t* owner, float* max_priority, int64_t capacity,
                                  float alpha, void* workspace, void* stream);
/* The ONE-CALL form of the PER loop (round 6; per.py:84-153): the bookkeeping rides on launches that exist anyway, 4 launches per iteration instead of 6, with indices,
 * weights, priorities, sums and parameters bit-identical to the calls above.
 *   mi_per_act_steps   = mi_dqn_act_steps (zero_next NULL) / mi_dqn_act_steps2 + mi_per_mark_sums in ONE launch: workgroups behind the acting ones write the new rows'
 *                        priorities (= max_priority, per.py:105 — it only changes at an update) and rebuild the touched groups' sums; eight more workgroups rebuild the chunk
 *                        sums the last mi_per_td_update left OWED (owed_idx = that update's idx buffer, owed_batch its batch; NULL: nothing owed).
 *   mi_per_td_update   one optimisation step (per.py:126-153) = mi_per_sample_current + the weighted TD launch + the slab sum + Adam launch of mi_dqn_td_update, whose
 *                        last workgroup scatters priorities[idx] = |td| (last duplicate wins) and updates max_priority (per.py:144-145).  The scattered entries' chunk
 *                        sums are NOT rebuilt by this call: they are OWED — hand idx to the next mi_per_act_steps, or call mi_per_settle_sums before anything else reads
 *                        the sums (a second update, mi_per_sample_current, a checkpoint).  sample = 0 keeps the caller's idx.  count = transitions stored (per.py:148).
 *   mi_per_settle_sums the owed sums as a launch of their own. */
int mi_per_act_steps(void* handle, const float* params, int n_steps, int64_t global_step, int64_t slots, int64_t learning_starts, double start_e, double end_e,
                     double exploration_fraction, int64_t total_timesteps, float* obs_cur, float* observations, int64_t* actions, float* rewards, uint8_t* terminated,
                     const int64_t* forced_actions, const double* forced_resets, mi_episode_t* episodes, int32_t* episode_stats, int max_ep, int32_t* zero_next,
                     float* priorities, const float* max_priority, float alpha, void* per_workspace, const int64_t* owed_idx, int owed_batch, void* stream);
int mi_per_td_update(float* params, const float* target_params, const float* observations, const int64_t* actions, const float* rewards, const uint8_t* terminated,
                     int64_t* idx, int batch, int n_envs, int64_t slots, float gamma, float* weights, float* td_abs, void* workspace, float* grads, float* loss,
                     float* exp_avg, float* exp_avg_sq, int64_t step, double lr, double beta1, double beta2, double eps, uint64_t seed, uint64_t update_index,
                     float* priorities, int64_t n_valid, double count, float alpha, float beta, int sample, void* per_workspace, int32_t* owner, float* max_priority,
                     void* stream);
int mi_per_settle_sums(const float* priorities, const int64_t* idx, int batch, int64_t capacity, float alpha, void* per_workspace, void* stream);

/* =====================================================================================================================
 * SAC (reference deep_rl/sac.py re-targeted to Pendulum-v1; SURVEY.md §8a s1-s8, BASELINE config 4).
 * Env kind MI_ENV_PENDULUM_V1: obs f32 [N,3] = (cos th, sin th, th_dot), action f32 [N] (one dim, clipped to +-2), reward
 * -(angle_normalize(th)^2 + .1 th_dot^2 + .001 u^2), never terminates, TimeLimit 200 (gym 0.21 pendulum.py, restated).
 * SoftQNetwork (sac.py:29-43) cat(obs, act) 4 -> 256 -> 256 -> 1 ReLU: flat W1[256,4] b1 W2[256,256] b2 W3[1,256] b3 = MI_SAC_Q_NPARAMS;
 * the two critics (and the two targets) are stored back to back: [2 * MI_SAC_Q_NPARAMS].
 * Actor (sac.py:46-78) 3 -> 256 -> 256 ReLU, mean head, tanh-bounded log-std head: flat shared W1[256,3] b1 W2[256,256] b2,
 * mean W[1,256] b, log_std W[1,256] b = MI_SAC_ACTOR_NPARAMS.  action = tanh(mean + std * eps) * 2.
 * Replay ring as for DQN with observations [slots,N,3] and actions f32 [slots,N].  alpha lives on the device (f32 [1]) so that no
 * update needs a host round trip.  eps arguments: standard-normal draws, dev f32 [batch]; NULL = keyed draws (Box-Muller on Philox
 * stream 5, (env := call tag * 2^32 + update index, idx := row)).
 * ===================================================================================================================== */
#define MI_ENV_PENDULUM_V1 1
#define MI_SAC_Q_NPARAMS 67329
#define MI_SAC_ACTOR_NPARAMS 67330
/* env.step for continuous actions (Pendulum): actions dev f32 [N]; forced_reset dev f64 [N,2]; obs dev f32 [N,3] */
int mi_env_step_cont(void* handle, const float* actions, const double* forced_reset, float* obs, float* reward, uint8_t* done,
                     uint8_t* truncated, float* fin_ret, int32_t* fin_len, void* stream);
/* Actor.get_action (sac.py:65-78) on a batch: action / logp dev f32 [n] (logp nullable) */
int mi_sac_actor_sample(const float* actor, const float* obs, const float* eps, int n, float* action, float* logp, void* stream);
/* SoftQNetwork.forward (sac.py:40-43): out dev f32 [n] */
int mi_sac_q_forward(const float* q, const float* obs, const float* act, int n, float* out, void* stream);
/* one iteration of the acting half of the loop (sac.py:138-158) for the handle's N Pendulum envs: uniform random action while
 * global_step < learning_starts (keyed), else actor.get_action; env.step with auto-reset; ring store. */
int mi_sac_act_step(void* handle, const float* actor, int64_t global_step, int64_t slots, int64_t learning_starts, float* obs_cur,
                    float* observations, float* actions, float* rewards, uint8_t* terminated, const float* forced_actions,
                    const float* forced_eps, const double* forced_resets, mi_episode_t* episodes, int32_t* episode_stats, int max_ep,
                    void* stream);
/* scratch for the update calls below; the caller zero-fills it ONCE before first use (it carries a self-resetting ticket word) */
size_t mi_sac_workspace_bytes(int batch);
/* critic update (sac.py:165-185): grads dev f32 [2*MI_SAC_Q_NPARAMS] of qf1_loss + qf2_loss, losses dev f32 [2] */
int mi_sac_critic_grad(const float* q, const float* q_target, const float* actor, const float* observations, const float* actions,
                       const float* rewards, const uint8_t* terminated, const int64_t* idx, int batch, int n_envs, int64_t slots,
                       const float* eps, uint64_t seed, uint64_t update_index, const float* alpha, float gamma, double inv_count,
                       void* workspace, float* grads, float* losses, void* stream);
/* actor update (sac.py:193-197): grads dev f32 [MI_SAC_ACTOR_NPARAMS], out dev f32 [2] = {actor_loss, mean log_prob} */
int mi_sac_actor_grad(const float* actor, const float* q, const float* observations, const int64_t* idx, int batch, const float* eps,
                      uint64_t seed, uint64_t update_index, const float* alpha, double inv_count, void* workspace, float* grads, float* out,
                      void* stream);
/* single-process fusions (no gradient exchange in between): the same launches as the *_grad calls, whose last kernel also applies
 * optimizer.step() (torch Adam, sac.py:185 / :197) to every gradient element it has just assembled and, for the critics, the polyak step of the
 * target copy (sac.py:213-217; tau < 0 skips it).  inv_count = 1 / batch.  grads / losses / out are still written.  sample_upper > 0: the
 * critic launch draws idx itself (the mi_dqn_sample contract with (seed, sample_update), bit-identical) and stores it in idx for the actor /
 * alpha calls that follow; 0: idx is an input. */
int mi_sac_critic_update(float* q, float* q_target, const float* actor, const float* observations, const float* actions, const float* rewards,
                         const uint8_t* terminated, int64_t* idx, int batch, int n_envs, int64_t slots, const float* eps, uint64_t seed,
                         uint64_t update_index, const float* alpha, float gamma, void* workspace, float* grads, float* losses, float* exp_avg,
                         float* exp_avg_sq, int64_t step, double lr, double beta1, double beta2, double adam_eps, float tau, uint64_t sample_update,
                         int64_t sample_upper, void* stream);
int mi_sac_actor_update(float* actor, const float* q, const float* observations, const int64_t* idx, int batch, const float* eps, uint64_t seed,
                        uint64_t update_index, const float* alpha, void* workspace, float* grads, float* out, float* exp_avg, float* exp_avg_sq,
                        int64_t step, double lr, double beta1, double beta2, double adam_eps, void* stream);
/* alpha update (sac.py:203-210): fresh log-probs under the current actor, alpha_loss = mean(-log_alpha (logp + target_entropy)),
 * one Adam step on log_alpha (dev f32 [1], moments dev f32 [1] each), alpha <- exp(log_alpha) (dev f32 [1]); out dev f32 [2] =
 * {alpha_loss, d alpha_loss / d log_alpha} (nullable). */
int mi_sac_alpha_step(const float* actor, const float* observations, const int64_t* idx, int batch, const float* eps, uint64_t seed,
                      uint64_t update_index, float target_entropy, float* log_alpha, float* exp_avg, float* exp_avg_sq, int64_t step,
                      double lr, float* alpha, float* out, void* workspace, void* stream);

/* ---- the alpha step OWED from the last actor update, carried by the next launch instead of a launch of its own.  Its log-prob pass (sac.py:203-204) depends only
 * on the actor and on the batch observations of that actor update (mi_sac_actor_update* stashes them in the workspace).  mi_sac_actor_update_owed /
 * mi_sac_critic_update_owed run it on workgroups of their own launch (accepted while the launch stays within half of the CUs: batch <= 1024 on an MI355X), apply Adam to log_alpha (:205-210) and hand alpha to the
 * launch's own workgroups, which read it only where the reference does (after their forward passes); mi_sac_alpha_step_owed runs it alone (flush).  Identical
 * results to mi_sac_alpha_step called right after the actor update, with eps == NULL (keyed draws).  step: 1-based Adam step of log_alpha, strictly increasing. */
typedef struct {
    float* log_alpha; float* exp_avg; float* exp_avg_sq; float* alpha; float* out /* nullable, [2] = {alpha_loss, d / d log_alpha} */;
    float target_entropy; int64_t step; double lr; uint64_t update_index /* key of the log-prob draw: the actor update's update_index */;
    int32_t epoch;       /* the caller's count of owed steps handed to THIS workspace: strictly increasing from 1 (compared wrap-safely), independent of `step`
                          * — a checkpoint load may rewind the Adam step number, never this counter */
    int32_t stash_slot;  /* 0 / 1: which observation stash the debt reads.  mi_sac_actor_update_owed writes the OTHER slot when it carries a debt and slot 0 when it
                          * carries none, so a debt always reads the observations of the actor update that created it, whatever was sampled since */
} mi_sac_owed_alpha_t;
int mi_sac_critic_update_owed(float* q, float* q_target, const float* actor, const float* observations, const float* actions, const float* rewards,
                              const uint8_t* terminated, int64_t* idx, int batch, int n_envs, int64_t slots, const float* eps, uint64_t seed,
                              uint64_t update_index, const float* alpha, float gamma, void* workspace, float* grads, float* losses, float* exp_avg,
                              float* exp_avg_sq, int64_t step, double lr, double beta1, double beta2, double adam_eps, float tau, uint64_t sample_update,
                              int64_t sample_upper, const mi_sac_owed_alpha_t* owed, void* stream);
int mi_sac_actor_update_owed(float* actor, const float* q, const float* observations, const int64_t* idx, int batch, const float* eps, uint64_t seed,
                             uint64_t update_index, const float* alpha, void* workspace, float* grads, float* out, float* exp_avg, float* exp_avg_sq,
                             int64_t step, double lr, double beta1, double beta2, double adam_eps, const mi_sac_owed_alpha_t* owed, void* stream);
int mi_sac_alpha_step_owed(const float* actor, int batch, uint64_t seed, const mi_sac_owed_alpha_t* owed, void* workspace, void* stream);
int mi_sac_owed_alpha_fits(int batch);   /* 1 when a launch at this batch may carry an owed alpha step on the current device (half of its usable CUs stay free) */

/* ---- the critics' optimizer step DEFERRED to the next launch.  mi_sac_critic_update_owed is two launches: the row-group kernel (forward, loss, backward: H1 / dZ2 / slabs
 * into the workspace) and the step (weight-gradient GEMM + gradient assembly + Adam + polyak: sac.py:183-185,213-217).  The step touches the critics only, and the launch
 * that follows a critic update in the loop of sac.py:137-217 is the NEXT acting step, which reads the actor and the env only — so the step can ride on extra workgroups of
 * that launch instead of being a link of the launch chain:
 *   mi_sac_critic_update_deferred   the first launch only (same arguments, same sampling contract; an alpha step owed from the last actor update rides on it as before);
 *   mi_sac_act_step_carry(.., step) the acting launch + the owed step (step == NULL: plain mi_sac_act_step; padded batches > 512: the step's own two launches, then acting);
 *   mi_sac_critic_step(step)        the step as the launch of its own (what an undeferred call would have made) — for anything else that comes next.
 * The same workgroups do the same arithmetic in every form: bit-identical to mi_sac_critic_update_owed.  Between the deferred call and the step nothing may read or write the
 * critics, their targets, their Adam moments or `grads` / `losses`, or use the workspace for another update.  step: 1-based Adam step; tau < 0: no polyak step. */
typedef struct {
    void* workspace; int32_t batch; float* q; float* q_target; float* exp_avg; float* exp_avg_sq; float* grads /* [2 MI_SAC_Q_NPARAMS] */; float* losses /* [2] */;
    int64_t step; double lr, beta1, beta2, adam_eps; float tau;
} mi_sac_critic_step_t;
int mi_sac_critic_update_deferred(const float* q, const float* q_target, const float* actor, const float* observations, const float* actions, const float* rewards,
                                  const uint8_t* terminated, int64_t* idx, int batch, int n_envs, int64_t slots, const float* eps, uint64_t seed, uint64_t update_index,
                                  const float* alpha, float gamma, void* workspace, uint64_t sample_update, int64_t sample_upper, const mi_sac_owed_alpha_t* owed,
                                  void* stream);
int mi_sac_critic_step(const mi_sac_critic_step_t* step, void* stream);
int mi_sac_act_step_carry(void* handle, const float* actor, int64_t global_step, int64_t slots, int64_t learning_starts, float* obs_cur, float* observations,
                          float* actions, float* rewards, uint8_t* terminated, const float* forced_actions, const float* forced_eps, const double* forced_resets,
                          mi_episode_t* episodes, int32_t* episode_stats, int max_ep, const mi_sac_critic_step_t* step, void* stream);

/* ---- waits between the workgroups of one SAC launch (sibling roles of a row group, the owed alpha step): every waiter only waits for workgroups that precede
 * it in dispatch order, and every spin is bounded (100 ms of wall clock).  A wait that runs out stores a code in a host-pinned status word and in a word on the
 * device, takes NaN as the value (the launch's gradients and losses come out NaN) and lets the kernel finish; from then on every optimizer step of the SAC calls
 * on that device (fused or not: Adam on critics / actor / log_alpha, polyak) is WITHHELD, so parameters, moments and targets stay as they were before the faulted
 * update.  The failure surfaces as MI_ESTATE from the NEXT mi_sac_* update call (a plain host read, no synchronisation) or from mi_sac_check; after
 * mi_sac_clear_error the caller may simply go on (one update lost) or restore a checkpoint first.  One device per status word: the library keeps the word's
 * device pointer per device, so a process may run SAC on several.
 *   mi_sac_check(stream, wait)        wait != 0: synchronise `stream` first.  MI_OK, or MI_ESTATE with mi_last_error() naming what was not published.
 *   mi_sac_clear_error(ws, batch, s)  clears the status word and the current device's fault word and (workspace != NULL) zeroes the workspace's hand-off words,
 *                                     ticket, stash and epoch word (synchronises `stream`).
 *   mi_sac_set_max_cus(n)             how many CUs sibling roles may assume (0 = the device's count cut by HSA_CU_MASK / ROC_GLOBAL_CU_MASK, which
 *                                     hipDeviceProp_t.multiProcessorCount does not see); steers performance only.  mi_sac_usable_cus() reports the figure in use.
 *   mi_sac_test_fault(mode)           TEST HOOK: bit 0 = publishing siblings skip their hand-off words, bit 1 = the owed alpha role does not publish its epoch
 *                                     (later launches then time out as they would if a producer never ran); 0 = off. */
/* ---- transposed copies of the 256 x 256 layer-2 matrices ("shadows"; round 5).  The row-group kernels stream a layer's matrix straight from L2 into MFMA operand
 * registers; for a FORWARD pass on torch's [out][in] layout 16 consecutive lanes then read 16 bytes from each of 16 rows a kilobyte apart, for the backward pass 64
 * contiguous bytes — one warm pass takes 4.84 us against 3.96 (first touch: 7.0 / 5.8).  With a transposed copy W^T[in][out] the forward pass uses the backward pass's
 * access pattern and performs the SAME multiply-adds in the same order: bit-identical results, every forward pass ~1 us shorter.  The caller owns the buffers:
 *   mi_sac_shadow_set(params, is_actor, shadow)   registers (shadow: dev f32 [65536] for the actor's flat vector, [2][65536] for a critics' / targets' vector; invalid
 *                                                 until refreshed) or, with shadow == NULL, drops the entry.  Keyed by the device pointer `params`.
 *   mi_sac_shadow_refresh(params, stream)         one transpose launch; the shadow is valid from here on.
 *   mi_sac_shadow_invalidate(params)              the caller has written the parameters some other way (NULL: every registered vector).
 *   mi_sac_shadow_valid(params)                   1 / 0.
 * While a vector's shadow is valid, mi_sac_act_step*, the critic / actor update and log-prob launches stream its forward passes from the shadow (all the shadows a launch
 * needs must be valid, else it takes the plain form), and the library's FUSED optimizer steps (mi_sac_*_update*, mi_sac_critic_step, the carried step) write every
 * layer-2 element — and its polyak-averaged target — to both copies.  Its other writers (mi_adam, mi_polyak) mark the shadow invalid.  A wrong `valid` can only
 * come from the caller writing parameters behind the library's back without mi_sac_shadow_invalidate / _refresh; deep_rl_amd.SACEngine tracks torch's version counters. */
int mi_sac_shadow_set(const float* params, int is_actor, float* shadow);
int mi_sac_shadow_refresh(const float* params, void* stream);
int mi_sac_shadow_invalidate(const float* params);
int mi_sac_shadow_valid(const float* params);
int mi_sac_check(void* stream, int wait);
int mi_sac_clear_error(void* workspace, int batch, void* stream);
int mi_sac_set_max_cus(int max_cus);
int mi_sac_usable_cus(void);
int mi_sac_test_fault(int mode);
/* the same in two halves for sharded runs (all-reduce *mean_logp between them): mean_logp dev f32 [1] = inv_count * sum of this rank's
 * fresh log-probs; then the Adam step on log_alpha from the global mean. */
int mi_sac_mean_logp(const float* actor, const float* observations, const int64_t* idx, int batch, const float* eps, uint64_t seed,
                     uint64_t update_index, double inv_count, float* mean_logp, void* workspace, void* stream);
int mi_sac_alpha_adam(const float* mean_logp, float target_entropy, float* log_alpha, float* exp_avg, float* exp_avg_sq, int64_t step,
                      double lr, float* alpha, float* out, void* stream);
/* the three of them as ONE call each with the all-reduce in-stream on libmirl's RCCL communicator (the pattern of mi_ppo_update_sharded; sac.py:165-210 with the exchange
 * between backward and optimizer.step()): qbuf = dev f32 [2 MI_SAC_Q_NPARAMS + 2] {grads, qf1_loss, qf2_loss}; abuf = dev f32 [MI_SAC_ACTOR_NPARAMS + 2] {grads, actor_loss,
 * mean log-prob}; mean_logp = dev f32 [1] scratch.  tau < 0 skips the polyak step.  The same launches as the *_grad + caller all-reduce + mi_adam (+ mi_polyak) /
 * mi_sac_mean_logp + all-reduce + mi_sac_alpha_adam sequences: bit-identical.  comm NULL or world 1: no collective.
 * On the P2P carrier (round 6) the exchange rides INSIDE the launch that assembles the gradient: the thread that holds a final gradient element (or loss scalar)
 * exchanges it — line = its index in qbuf / abuf, rank-ordered sum — and applies Adam (+ polyak) behind it; the alpha step's one-wave launch sums this rank's slabs,
 * exchanges the mean and steps log_alpha.  Same arithmetic, same bits; the all-reduce, Adam, polyak and alpha launches of the sequences disappear (14 launches -> 7 per
 * iteration).  A wait that runs out withholds every step behind it (mi_comm_poll). */
int mi_sac_critic_update_sharded(float* q, float* q_target, const float* actor, const float* observations, const float* actions, const float* rewards,
                                 const uint8_t* terminated, const int64_t* idx, int batch, int n_envs, int64_t slots, const float* eps, uint64_t seed, uint64_t update_index,
                                 const float* alpha, float gamma, void* workspace, float* qbuf, float* exp_avg, float* exp_avg_sq, int64_t step, double lr, double beta1,
                                 double beta2, double adam_eps, float tau, void* comm, void* stream);
int mi_sac_actor_update_sharded(float* actor, const float* q, const float* observations, const int64_t* idx, int batch, const float* eps, uint64_t seed,
                                uint64_t update_index, const float* alpha, void* workspace, float* abuf, float* exp_avg, float* exp_avg_sq, int64_t step, double lr,
                                double beta1, double beta2, double adam_eps, void* comm, void* stream);
int mi_sac_alpha_step_sharded(const float* actor, const float* observations, const int64_t* idx, int batch, const float* eps, uint64_t seed, uint64_t update_index,
                              float target_entropy, float* log_alpha, float* exp_avg, float* exp_avg_sq, int64_t step, double lr, float* alpha, float* out,
                              float* mean_logp, void* workspace, void* comm, void* stream);
/* optim.Adam.step without clipping (sac.py:108,117; torch single-tensor formula), any n */
int mi_adam(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int n, int64_t step, double lr, double beta1,
            double beta2, double eps, void* stream);
/* target <- tau * param + (1 - tau) * target (sac.py:213-217) */
int mi_polyak(float* target, const float* param, int n, float tau, void* stream);

/* ---- hardware self-test: probes the MFMA fragment layouts the update kernel relies on with exact
 * integer data; report dev i32 [16] (0 = ok per probe); dump (nullable) dev f32 [3*64*16] receives the raw
 * accumulators of the probes.  Used by tests and smoke(). */
int mi_selftest_mfma(int32_t* report, float* dump, void* stream);

/* ---- test hook: y[i] = the engine's device tanh(x[i]) (x, y dev f32 [n]); lets the tests bound its error */
int mi_test_tanh(const float* x, float* y, int n, void* stream);

/* ---- in-library kernel profiler for bench.py: while enabled, every launch of the tagged kernels is bracketed by a
 * pair of HIP events on ITS stream (so the durations are measured live inside the timed region, on the stream the
 * kernel runs on).  mi_prof_begin(max_launches, tag_mask) arms it for the tags whose bit is set (allocates the event
 * pool, may synchronise);
 * mi_prof_end synchronises, fills total_ms[MI_PROF_NTAGS] / count[MI_PROF_NTAGS] (host arrays) and disarms.
 * mi_prof_pause(1 / 0) between the two stops / resumes the sampling: two events around every launch of a 70 us kernel cost the loop they measure 7.5 % (tools/prof_overhead.py:
 * 1.411 ms per update with all launches bracketed, 1.335 with every 4th update's, 1.312 with none), so bench.py brackets the launches of every 10th update only. */
enum { MI_PROF_ROLLOUT = 0, MI_PROF_GAE = 1, MI_PROF_GRAD = 2, MI_PROF_REDUCE = 3, MI_PROF_CLIP_ADAM = 4, MI_PROF_STATS = 5,
       /* DQN (config 3): acting launch, TD forward+backward, slab sum (+ Adam), PER sampler launches */
       MI_PROF_DQN_ACT = 6, MI_PROF_DQN_TD = 7, MI_PROF_DQN_REDUCE = 8, MI_PROF_PER = 9,
       /* SAC (config 4): acting launch, row-group critic / actor kernels, dW2 GEMM, assembly (+ Adam + polyak), log-prob + alpha.  NOTE: with the critics' optimizer
        * step deferred (mi_sac_act_step_carry; the engines' default) MI_PROF_SAC_ACT also covers the carried dW2 + Adam + polyak workgroups, which then do not
        * appear under MI_PROF_SAC_GEMM; on the P2P carrier the gradient all-reduce runs inside MI_PROF_REDUCE and MI_PROF_COMM_GRAD stays empty */
       MI_PROF_SAC_ACT = 10, MI_PROF_SAC_CRITIC = 11, MI_PROF_SAC_ACTOR = 12, MI_PROF_SAC_GEMM = 13, MI_PROF_SAC_ASSEMBLE = 14, MI_PROF_SAC_LOGP = 15,
       /* the in-stream collectives of mi_ppo_update_sharded: the 9,159-float gradient all-reduce (16 per update), the advantage-statistics all-reduce (1) */
       MI_PROF_COMM_GRAD = 16, MI_PROF_COMM_STATS = 17,
       MI_PROF_NTAGS = 18 };
int mi_prof_begin(int max_launches, uint32_t tag_mask);
int mi_prof_end(float* total_ms, int32_t* count);
int mi_prof_pause(int paused);

/* ---- timing helper for bench.py: HIP events on the given stream (torch.cuda.Event only sees torch's
 * current stream).  mi_timer_* are host-side and may synchronise. */
int mi_timer_create(void** timer);
int mi_timer_destroy(void* timer);
int mi_timer_start(void* timer, void* stream);
int mi_timer_stop(void* timer, void* stream);
int mi_timer_elapsed_ms(void* timer, float* ms); /* synchronises on the stop event */

#ifdef __cplusplus
}
#endif
#endif /* MI_RL_H */
