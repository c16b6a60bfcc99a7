"""PPOEngine — device-resident rollout storage + the per-update launch sequence of reference ppo.py:105-195.

Owns the six ``(T+1, N, …)`` storage tensors with the reference's names (ppo.py:93-98, env axis added after
time), the advantage / return buffers, the gradient workspace and the episode log; every method is one or
two launches through the C ABI (include/mi_rl.h) on torch's current stream.  Nothing here computes: Python
sequences launches and (for world_size > 1) the two tiny all-reduces.

Multi-GPU: envs shard across ranks (rank r owns global envs [r*N, (r+1)*N)); parameters and
Adam state are replicated.  Per update one all-reduce(SUM) of the (epochs, n_minibatch, 3) advantage statistics, per
optimizer step one all-reduce(SUM) of the flat 9,155-float gradient + 4 loss terms (already scaled by 1/(world*mb)):
17 collectives per update; clip + Adam then run identically on every rank.  Minibatch permutations are per-rank-local.
With an NCCL process group the whole sharded update is ONE C call (mi_ppo_update_sharded: RCCL all-reduces enqueued
in-stream between the launches, no Python in between); other backends (gloo) walk the same launches from here.
"""
import ctypes as C
import os

import torch

from . import _native as N
from . import dist as D


# diagnostics: MIRL_PPO_SHARDED_SEQUENCE=1 makes a single process walk the sharded launch sequence (all-reduces are no-ops), to measure its host cost
_FORCE_SHARDED_SEQUENCE = os.environ.get("MIRL_PPO_SHARDED_SEQUENCE", "0") == "1"
# MIRL_PPO_NATIVE_SHARDED=1: take the mi_ppo_update_sharded route whenever an RCCL communicator exists, even at world_size 1 (one-GPU box: RCCL really runs)
_FORCE_NATIVE_SHARDED = os.environ.get("MIRL_PPO_NATIVE_SHARDED", "0") == "1"
# MIRL_PPO_ASSUME_SHARDED=1 (test hook, process-wide: mi_ppo_test_assume_sharded): the one-call updates take the world_size > 1 form of the owed optimizer step (the clip
# coefficient recomputed from the gradient itself) without a second rank, so a one-GPU box can check it bitwise against the default route and time it (bench.py `sharded_route`)
_ASSUME_SHARDED = os.environ.get("MIRL_PPO_ASSUME_SHARDED", "0") == "1"


def set_assume_sharded(on):
    N.check(N.lib().mi_ppo_test_assume_sharded(1 if on else 0), "mi_ppo_test_assume_sharded")


class PPOEngine:
    def __init__(self, env, agent, optimizer, num_steps=128, n_minibatch=4, update_epochs=4, gamma=0.99, gae_lambda=0.95,
                 clip_coef=0.2, ent_coef=0.01, vf_coef=0.5, max_episodes_logged=None, process_group=None):
        self.env, self.agent, self.optimizer = env, agent, optimizer
        self.T, self.N = int(num_steps), env.num_envs
        self.device = env.device
        self.n_minibatch, self.update_epochs = int(n_minibatch), int(update_epochs)
        self.gamma, self.gae_lambda = float(gamma), float(gae_lambda)
        self.clip_coef, self.ent_coef, self.vf_coef = float(clip_coef), float(ent_coef), float(vf_coef)
        self.batch_size = self.T * self.N
        if self.batch_size % self.n_minibatch:
            raise N.MiError("num_steps*num_envs (%d) must be divisible by n_minibatch (%d)" % (self.batch_size, self.n_minibatch))
        self.minibatch_size = self.batch_size // self.n_minibatch
        self.pg = process_group
        self.world_size = D.world_size(process_group)
        T, Nn, dev = self.T, self.N, self.device
        z = lambda *s, dt=torch.float32: torch.zeros(s, dtype=dt, device=dev)  # noqa: E731
        # storage (ppo.py:93-98): num_steps + 1 rows because the terminal values are needed for the advantages
        self.observations = z(T + 1, Nn, 4)
        self.values = z(T + 1, Nn)
        self.actions = z(T + 1, Nn, dt=torch.int64)
        self.log_probs = z(T + 1, Nn)
        self.rewards = z(T + 1, Nn)
        self.dones = z(T + 1, Nn)
        self.advantages = z(T + 1, Nn)
        self.returns = z(T + 1, Nn)
        self._perm_all = z(self.update_epochs, self.batch_size, dt=torch.int32)  # sharded runs keep every epoch's permutation (one stats all-reduce per update)
        self.perm = self._perm_all[0]
        self._adv_sums_all = z(self.update_epochs, self.n_minibatch, 3, dt=torch.float64)  # mi_ppo_update: one slab per epoch
        self.adv_sums = self._adv_sums_all[0]
        # gradient + loss terms share one buffer so that ONE all-reduce carries both when sharded
        self._gradbuf = z(N.NPARAMS + 5)
        self.grads = self._gradbuf[:N.NPARAMS]
        self.loss_terms = self._gradbuf[N.NPARAMS:N.NPARAMS + 4]
        self.explained_var = z(1, dt=torch.float64)
        self.workspace = torch.empty(N.lib().mi_ppo_workspace_bytes(), dtype=torch.uint8, device=dev)
        self.max_ep = int(max_episodes_logged if max_episodes_logged is not None else (1024 if Nn <= 8 else 0))
        self.episodes = torch.zeros((max(self.max_ep, 1), 4), dtype=torch.int32, device=dev)  # mi_episode_t = 4 x 32 bit
        # {finished episodes, sum of lengths, longest, -} of the last rollout, double-buffered: the fused update's rollout launch zeroes the OTHER buffer for the
        # next update instead of spending a launch on the reset (`episode_stats` is the buffer of the last rollout)
        self._stats2 = z(2, 4, dt=torch.int32)
        self._stats_cur = 0
        # without an episode log the statistics stay per workgroup inside the env handle (no atomics in the rollout launch: -26 us at 4096 envs) and
        # `episode_stats` sums them when somebody reads it (mi_env_episode_stats)
        self._lazy_stats = self.max_ep == 0
        self._stats_lazy = z(4, dt=torch.int32)
        self._stats_ready = False
        self._stats_any = False
        self.observation = None  # the carried-over `observation` of the reference loop (ppo.py:101,127-129)
        self.update_index = 0
        self._ev_parts = z(2, 2, dt=torch.float64)   # sharded explained variance: {sums, squared deviations}
        self._check_every = D.replica_check_interval()
        if _ASSUME_SHARDED:
            set_assume_sharded(True)

    # ---- pieces of one outer update ----------------------------------------------------------------
    @property
    def episode_stats(self):
        """{finished episodes, sum of their lengths, longest, slots of the episode log} of the last rollout (device i32 [4]).
        Single-owner, single-stream contract: without an episode log the property hands back ONE persistent tensor that the next read after the next
        rollout overwrites (clone it to keep an update's numbers), and the launch that sums the per-workgroup statistics is ordered behind the rollout through torch's
        current stream only — read it on the stream the rollout ran on."""
        if not self._lazy_stats:
            return self._stats2[self._stats_cur]
        if not self._stats_ready and self._stats_any:
            N.check(N.lib().mi_env_episode_stats(self.env.handle, N.ptr(self._stats_lazy), self._s()), "mi_env_episode_stats")
            self._stats_ready = True
        return self._stats_lazy

    def _stats_arg(self):
        """episode_stats argument of a rollout call: None keeps the statistics in the handle."""
        self._stats_ready, self._stats_any = False, True
        return None if self._lazy_stats else self._stats2[self._stats_cur]

    def _s(self):
        return N.stream_ptr(self.device)

    def reset(self, forced_state=None):
        """observation = env.reset() (ppo.py:101)."""
        self.observation = self.env.reset(forced_state)
        return self.observation

    def rollout(self, forced_actions=None, forced_uniforms=None, forced_resets=None):
        """ppo.py:110-141 in one launch; returns nothing (episode log: drain_episodes())."""
        if self.observation is None:
            self.reset()
        dev = self.device
        fa = None if forced_actions is None else forced_actions.to(dev, torch.int64).contiguous()
        fu = None if forced_uniforms is None else forced_uniforms.to(dev, torch.float32).contiguous()
        fr = None if forced_resets is None else forced_resets.to(dev, torch.float64).contiguous()
        N.check(N.lib().mi_ppo_rollout(self.env.handle, N.ptr(self.agent.flat), self.T, N.ptr(self.observation),
                                       N.ptr(self.observations), N.ptr(self.values), N.ptr(self.actions), N.ptr(self.log_probs),
                                       N.ptr(self.rewards), N.ptr(self.dones), N.ptr(fa), N.ptr(fu), N.ptr(fr),
                                       N.ptr(self.episodes), N.ptr(self._stats_arg()), self.max_ep, self._s()), "mi_ppo_rollout")

    def drain_episodes(self):
        """Host sync.  -> (count, [(env, t, return, length), ...] sorted by (t, env)) of the last rollout."""
        n = int(self.episode_stats[0].item())
        if self.world_size > 1:
            D.poll_native_comm(self.pg)   # behind the sync above: a wait of the P2P carrier that ran out during the last update raises HERE, not one call later
        k = min(n, self.max_ep)
        if k == 0:
            return n, []
        raw = self.episodes[:k].cpu()
        rets = raw[:, 2].contiguous().view(torch.float32)
        eps = sorted((int(raw[i, 1]), int(raw[i, 0]), float(rets[i]), int(raw[i, 3])) for i in range(k))
        return n, [(e, t, r, l) for (t, e, r, l) in eps]

    def episode_summary_async(self, pinned, direct=True):
        """Non-blocking copy of {episodes, sum of lengths, longest, -} of the last rollout into a pinned host tensor
        (read it after the next sync point; CartPole return == length).  Same single-stream contract as `episode_stats`.  The direct route below lets the device
        kernel write into `pinned.data_ptr()`: that holds for tensors torch itself pinned (`.pin_memory()` / hipHostMalloc: mapped at the same address on the device);
        memory pinned some other way (hipHostRegister of a foreign allocation) need not be — pass `direct=False` for such tensors."""
        if self.world_size > 1:
            D.poll_native_comm(self.pg)   # no sync: raises once a wait of the P2P carrier has run out (the optimizer steps behind it were withheld on the device)
        ok = direct and self._lazy_stats and self._stats_any and pinned.is_pinned() and pinned.dtype == torch.int32 and pinned.numel() >= 4 and pinned.is_contiguous()
        if ok:
            # the sum of the per-workgroup statistics is written straight into the pinned host tensor (device-visible at the same address): no copy behind it
            N.check(N.lib().mi_env_episode_stats(self.env.handle, pinned.data_ptr(), self._s()), "mi_env_episode_stats")
            return
        pinned.copy_(self.episode_stats, non_blocking=True)

    def rollout_gae(self):
        """rollout() + compute_gae() in one launch (production RNG): the rollout workgroups scan their own envs (mi_ppo_rollout_gae)."""
        if self.observation is None:
            self.reset()
        N.check(N.lib().mi_ppo_rollout_gae(self.env.handle, N.ptr(self.agent.flat), self.T, N.ptr(self.observation), N.ptr(self.observations), N.ptr(self.values),
                                           N.ptr(self.actions), N.ptr(self.log_probs), N.ptr(self.rewards), N.ptr(self.dones), N.ptr(self.episodes),
                                           N.ptr(self._stats_arg()), self.max_ep, self.gamma, self.gae_lambda, N.ptr(self.advantages), N.ptr(self.returns),
                                           self._s()), "mi_ppo_rollout_gae")

    def compute_gae(self):
        """ppo.py:144-151."""
        N.check(N.lib().mi_gae(N.ptr(self.rewards), N.ptr(self.dones), N.ptr(self.values), self.T, self.N, self.gamma,
                               self.gae_lambda, N.ptr(self.advantages), N.ptr(self.returns), self._s()), "mi_gae")

    def make_perm(self, epoch, key=None):
        """b_inds = permutation(T*N) (ppo.py:155) from the keyed Feistel bijection; or set_perm() explicit indices."""
        if key is None:
            key = N.lib().mi_perm_key(self.env._seed, self.update_index, epoch)
        N.check(N.lib().mi_make_perm(self.batch_size, key, N.ptr(self.perm), self._s()), "mi_make_perm")

    def set_perm(self, indices):
        self.perm.copy_(torch.as_tensor(indices, dtype=torch.int32).reshape(-1).to(self.device))

    def adv_stats(self, mb=None, n_mb=None):
        """Per-minibatch {sum, sum of squares, count} of the advantages (ppo.py:169); all-reduced when sharded."""
        mb = self.minibatch_size if mb is None else mb
        n_mb = self.n_minibatch if n_mb is None else n_mb
        N.check(N.lib().mi_adv_stats(N.ptr(self.advantages), N.ptr(self.perm), mb, n_mb, N.ptr(self.adv_sums), self._s()),
                "mi_adv_stats")
        D.allreduce_sum_(self.adv_sums, self.pg)

    def minibatch_grad(self, k, mb=None):
        """loss + gradient of minibatch k of the current permutation (ppo.py:159-190) -> self.grads, self.loss_terms."""
        mb = self.minibatch_size if mb is None else mb
        idx_ptr = self.perm.data_ptr() + 4 * k * mb
        N.check(N.lib().mi_ppo_minibatch_grad(
            N.ptr(self.agent.flat), N.ptr(self.observations), N.ptr(self.actions), N.ptr(self.log_probs), N.ptr(self.advantages),
            N.ptr(self.returns), N.ptr(self.values), idx_ptr, mb, self.adv_sums.data_ptr() + 24 * k, self.clip_coef, self.ent_coef,
            self.vf_coef, 1.0 / (mb * self.world_size), N.ptr(self.workspace), N.ptr(self.grads), N.ptr(self.loss_terms), self._s()),
            "mi_ppo_minibatch_grad")
        D.allreduce_sum_(self._gradbuf, self.pg)

    def optimizer_step(self):
        """clip_grad_norm_ + optimizer.step() (ppo.py:191-192)."""
        self.optimizer.step(self.grads)

    def compute_explained_var(self):
        """ppo.py:194-195 over the whole batch: all (T+1) * N rows of EVERY rank.  Single process: one launch.  Sharded: the two passes of the same statistic
        (sums -> global means, squared deviations) with a SUM all-reduce of two doubles behind each; only computed when read, never inside update()."""
        n = (self.T + 1) * self.N
        if self.world_size == 1:
            N.check(N.lib().mi_explained_var(N.ptr(self.values), N.ptr(self.returns), n, N.ptr(self.explained_var), self._s()), "mi_explained_var")
            return self.explained_var
        sums, dev2 = self._ev_parts[0], self._ev_parts[1]
        N.check(N.lib().mi_explained_var_parts(N.ptr(self.values), N.ptr(self.returns), n, None, N.ptr(sums), self._s()), "mi_explained_var_parts")
        D.allreduce_sum_(sums, self.pg)
        sums /= float(n * self.world_size)
        N.check(N.lib().mi_explained_var_parts(N.ptr(self.values), N.ptr(self.returns), n, N.ptr(sums), N.ptr(dev2), self._s()), "mi_explained_var_parts")
        D.allreduce_sum_(dev2, self.pg)
        self.explained_var.copy_((1.0 - dev2[1] / dev2[0]).reshape(1))   # 0 / 0 -> NaN as np.var gives it (ppo.py:195)
        return self.explained_var

    # ---- one whole outer update ------------------------------------------------------------------------
    def update(self):
        """Rollout + GAE + update_epochs x n_minibatch optimizer steps (ppo.py:110-192), production RNG.

        Single rank: ONE C call enqueues every launch (mi_ppo_update).  Sharded: the same launches are
        sequenced from here with the two all-reduces in between.
        """
        if self.observation is None:
            self.reset()
        g = self.optimizer.param_groups[0]
        comm = None
        if not _FORCE_SHARDED_SEQUENCE and (self.world_size > 1 or _FORCE_NATIVE_SHARDED):
            comm = D.native_comm(self.pg)   # None unless the process group is NCCL (= RCCL)
        if (self.world_size == 1 and not _FORCE_SHARDED_SEQUENCE) or comm is not None:
            o = self.optimizer
            self._stats_cur ^= 1   # the buffer the previous update's rollout launch zeroed
            buf = N.PPOBuffers(*[N.ptr(t) for t in (
                self.agent.flat, o.exp_avg, o.exp_avg_sq, self.grads, self.loss_terms, o.grad_norm, self.observation,
                self.observations, self.values, self.actions, self.log_probs, self.rewards, self.dones, self.advantages,
                self.returns, self._perm_all, self._adv_sums_all, self.workspace, self.episodes, self._stats_arg())], self.max_ep,
                None if self._lazy_stats else N.ptr(self._stats2[self._stats_cur ^ 1]))
            hp = N.PPOHparams(self.T, self.n_minibatch, self.update_epochs, self.update_index, o.step_count, self.gamma,
                              self.gae_lambda, self.clip_coef, self.ent_coef, self.vf_coef, float(g["max_grad_norm"]),
                              float(g["lr"]), g["betas"][0], g["betas"][1], g["eps"])
            if comm is not None:
                N.check(N.lib().mi_ppo_update_sharded(self.env.handle, C.byref(buf), C.byref(hp), comm, self._s()), "mi_ppo_update_sharded")
            else:
                N.check(N.lib().mi_ppo_update(self.env.handle, C.byref(buf), C.byref(hp), self._s()), "mi_ppo_update")
            o.step_count += self.update_epochs * self.n_minibatch
        else:
            self.rollout_gae()
            # every epoch's permutation and LOCAL advantage statistics first, then ONE all-reduce for the whole update (the statistics
            # depend only on the advantages and the permutation keys, not on the parameters): 17 collectives per update instead of 20
            N.check(N.lib().mi_ppo_perms_and_stats(self.env._seed, self.update_index, self.update_epochs, self.batch_size, self.n_minibatch, N.ptr(self.advantages),
                                                   N.ptr(self._perm_all), N.ptr(self._adv_sums_all), self._s()), "mi_ppo_perms_and_stats")
            D.allreduce_sum_(self._adv_sums_all, self.pg)
            for epoch in range(self.update_epochs):
                self.perm, self.adv_sums = self._perm_all[epoch], self._adv_sums_all[epoch]
                for k in range(self.n_minibatch):
                    self.minibatch_grad(k)
                    self.optimizer_step()
            self.perm, self.adv_sums = self._perm_all[0], self._adv_sums_all[0]
        self.update_index += 1
        if self._check_every and self.world_size > 1 and self.update_index % self._check_every == 0:
            self.check_replicas()

    def check_replicas(self):
        """Raise MiError unless parameters and Adam moments are bitwise equal on every rank (deep_rl_amd.dist.check_replicas; MIRL_CHECK_REPLICAS=K runs it every K updates)."""
        o = self.optimizer
        D.check_replicas([self.agent.flat, o.exp_avg, o.exp_avg_sq], self.pg, "PPO parameters / Adam moments after update %d" % self.update_index)
