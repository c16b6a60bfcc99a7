"""DQNEngine — device-resident replay ring + the launch sequence of reference dqn.py:84-137.

Owns the four replay tensors with the reference's names (dqn.py:73-76) plus an env axis, laid out as a [slots, N]
time-major ring (iqn.py:174-232 semantics; slots = total_timesteps + 1 gives the reference's linear storage).  Methods are
thin launch wrappers over the C ABI (include/mi_rl.h "DQN"); nothing is computed in Python.
"""
import os

import torch

from . import _native as N
from . import dist as D

# diagnostics (tests/_rccl_world1_worker.py): walk the sharded branch even at world_size 1, so that RCCL really runs on a one-GPU box
_FORCE_SHARDED = os.environ.get("MIRL_OFFPOLICY_SHARDED", "0") == "1"
# MIRL_PER_ONE_CALL=0: PERDQNEngine walks the round-5 launch sequence (act, marks, sampler, TD, slab sum + Adam, scatter + sums: six launches) instead of the one-call
# pieces mi_per_act_steps / mi_per_td_update (four launches, bit-identical) — the A/B and the reference of tests/test_gpu_per.py
_PER_ONE_CALL = os.environ.get("MIRL_PER_ONE_CALL", "1") != "0"


class DQNEngine:
    def __init__(self, env, q_network, target_network, optimizer, slots, batch_size=128, gamma=0.99, learning_starts=10_000,
                 start_e=1.0, end_e=0.05, exploration_fraction=0.5, total_timesteps=100_000, max_episodes_logged=None, process_group=None):
        self.env, self.q, self.target, self.optimizer = env, q_network, target_network, optimizer
        self.N, self.device, self.slots = env.num_envs, env.device, int(slots)
        self.batch_size, self.gamma = int(batch_size), float(gamma)
        self.learning_starts, self.total_timesteps = int(learning_starts), int(total_timesteps)
        self.start_e, self.end_e, self.exploration_fraction = float(start_e), float(end_e), float(exploration_fraction)
        self.pg = process_group
        self.world_size = D.world_size(process_group)
        dev, S, Nn = self.device, self.slots, self.N
        self.observations = torch.zeros((S, Nn, 4), dtype=torch.float32, device=dev)   # dqn.py:73
        self.actions = torch.zeros((S, Nn), dtype=torch.int64, device=dev)             # :74
        self.rewards = torch.zeros((S, Nn), dtype=torch.float32, device=dev)           # :75
        self.terminated = torch.zeros((S, Nn), dtype=torch.uint8, device=dev)          # :76 (bool)
        self.batch_inds = torch.zeros(self.batch_size, dtype=torch.int64, device=dev)
        self._gradbuf = torch.zeros(N.DQN_NPARAMS + 2, dtype=torch.float32, device=dev)
        self.grads = self._gradbuf[:N.DQN_NPARAMS]
        self.loss = self._gradbuf[N.DQN_NPARAMS:N.DQN_NPARAMS + 1]
        self.workspace = torch.empty(N.lib().mi_dqn_workspace_bytes(self.batch_size), dtype=torch.uint8, device=dev)
        self.max_ep = int(max_episodes_logged if max_episodes_logged is not None else (1024 if Nn <= 8 else 0))
        self.episodes = torch.zeros((max(self.max_ep, 1), 4), dtype=torch.int32, device=dev)
        self._stats2 = torch.zeros((2, 4), dtype=torch.int32, device=dev)   # double-buffered: an acting launch zeroes the NEXT call's statistics
        self._stats_i = 0
        self._stats_buf = self._stats2[0]
        # without an episode log the statistics stay per workgroup inside the env handle (no atomics in the acting launch: -9 us at 4096 envs) and
        # `episode_stats` sums them when somebody reads it (mi_env_episode_stats)
        self._lazy_stats = self.max_ep == 0
        self._stats_lazy = torch.zeros(4, dtype=torch.int32, device=dev)
        self._stats_ready, self._stats_any = False, False
        self.observation = None
        self.global_step = 0      # time steps taken (each advances every env once)
        self.update_index = 0
        self._check_every = D.replica_check_interval()   # MIRL_CHECK_REPLICAS=K: every K-th train_step of a sharded run checks that the replicas still agree bitwise

    def _s(self):
        return N.stream_ptr(self.device)

    def reset(self, forced_state=None):
        """observation = env.reset(); observations[global_step] = observation (dqn.py:79-81)."""
        self.observation = self.env.reset(forced_state)
        self.observations[self.global_step % self.slots].copy_(self.observation)
        return self.observation

    def act(self, n_steps, forced_actions=None, forced_resets=None):
        """n_steps iterations of dqn.py:84-108 for every env, one launch."""
        dev = self.device
        fa = None if forced_actions is None else forced_actions.to(dev, torch.int64).contiguous()
        fr = None if forced_resets is None else forced_resets.to(dev, torch.float64).contiguous()
        self._act_launch(self.q.flat, n_steps, fa, fr)

    @property
    def episode_stats(self):
        """{finished episodes, sum of their lengths, longest, slots of the episode log} of the last act() call (device i32 [4])."""
        if not self._lazy_stats:
            return self._stats_buf
        if not self._stats_ready and self._stats_any:
            N.check(N.lib().mi_env_episode_stats(self.env.handle, N.ptr(self._stats_lazy), self._s()), "mi_env_episode_stats")
            self._stats_ready = True
        return self._stats_lazy

    def _act_launch(self, flat, n_steps, fa, fr):
        if self._lazy_stats:
            self._stats_ready, self._stats_any = False, True
            N.check(N.lib().mi_dqn_act_steps(
                self.env.handle, N.ptr(flat), int(n_steps), self.global_step, self.slots, self.learning_starts, self.start_e, self.end_e,
                self.exploration_fraction, self.total_timesteps, N.ptr(self.observation), N.ptr(self.observations), N.ptr(self.actions),
                N.ptr(self.rewards), N.ptr(self.terminated), N.ptr(fa), N.ptr(fr), None, None, 0, self._s()), "mi_dqn_act_steps")
            self.global_step += int(n_steps)
            return
        cur, nxt = self._stats2[self._stats_i], self._stats2[self._stats_i ^ 1]
        self._stats_buf = cur
        N.check(N.lib().mi_dqn_act_steps2(
            self.env.handle, N.ptr(flat), int(n_steps), self.global_step, self.slots, self.learning_starts, self.start_e, self.end_e,
            self.exploration_fraction, self.total_timesteps, N.ptr(self.observation), N.ptr(self.observations), N.ptr(self.actions),
            N.ptr(self.rewards), N.ptr(self.terminated), N.ptr(fa), N.ptr(fr), N.ptr(self.episodes), N.ptr(cur), self.max_ep, N.ptr(nxt),
            self._s()), "mi_dqn_act_steps2")
        self._stats_i ^= 1
        self.global_step += int(n_steps)

    def drain_episodes(self):
        """Host sync. -> (count, [(env, step_in_call, return, length)] sorted by (step, env)) of the last act() call."""
        st = self.episode_stats.tolist()
        k = min(st[3], self.max_ep)
        if k == 0:
            return st[0], []
        raw = self.episodes[:k].cpu()
        rets = raw[:, 2].contiguous().view(torch.float32)
        eps = sorted((int(raw[i, 1]), int(raw[i, 0]), float(rets[i]), int(raw[i, 3])) for i in range(k))
        return st[0], [(e, t, r, l) for (t, e, r, l) in eps]

    def sample(self, indices=None):
        """batch_inds = np.random.randint(upper, size=batch_size) (dqn.py:116; iqn.py:225: upper = min(global_step, memory_size))."""
        if indices is not None:
            self.batch_inds.copy_(torch.as_tensor(indices, dtype=torch.int64).reshape(-1).to(self.device))
            return
        upper = min(self.global_step, self.slots) * self.N
        N.check(N.lib().mi_dqn_sample(self.env._seed, self.update_index, upper, self.batch_size, N.ptr(self.batch_inds), self._s()), "mi_dqn_sample")

    def td_grad(self):
        """loss + gradient of the sampled batch (dqn.py:118-128) -> self.grads, self.loss (all-reduced when sharded)."""
        N.check(N.lib().mi_dqn_td_grad(
            N.ptr(self.q.flat), N.ptr(self.target.flat), N.ptr(self.observations), N.ptr(self.actions), N.ptr(self.rewards),
            N.ptr(self.terminated), N.ptr(self.batch_inds), self.batch_size, self.N, self.slots, self.gamma,
            1.0 / (self.batch_size * self.world_size), N.ptr(self.workspace), N.ptr(self.grads), N.ptr(self.loss), self._s()), "mi_dqn_td_grad")
        D.allreduce_sum_(self._gradbuf, self.pg)

    def _row_weights(self):
        """(importance weights, |td| out) of the TD launch — None for plain DQN (PERDQNEngine overrides)."""
        return None, None

    def _after_td(self):
        pass

    def train_step(self, indices=None):
        """One optimisation step (dqn.py:114-133).  Single process without gradient clipping: the launch that sums the gradient slabs also
        applies Adam (mi_dqn_td_update, bit-identical to td_grad() + optimizer.step())."""
        g = self.optimizer.param_groups[0]
        fusable = (self.world_size == 1 and not _FORCE_SHARDED and g["max_grad_norm"] == float("inf")
                   and type(self).td_grad in (DQNEngine.td_grad, PERDQNEngine.td_grad))
        native = not fusable and self._native_sharded()
        # the uniform randint of dqn.py:116 is drawn by the TD launch itself wherever ONE call runs the step (single process, and the sharded one-call route: every rank
        # draws from its own ring with the same keys, as mi_dqn_sample would)
        in_kernel_sampling = (fusable or native) and indices is None and type(self).sample is DQNEngine.sample
        if not in_kernel_sampling:
            self.sample(indices)
        upper = min(self.global_step, self.slots) * self.N if in_kernel_sampling else 0
        if in_kernel_sampling and upper == 0:   # upper == 0 means "read batch_inds" to the launch: an empty ring must not train on stale indices
            raise N.MiError("train_step: the replay ring is empty (global_step == 0); act() before training")
        if fusable:
            o = self.optimizer
            w, td = self._row_weights()
            N.check(N.lib().mi_dqn_td_update(
                N.ptr(self.q.flat), N.ptr(self.target.flat), N.ptr(self.observations), N.ptr(self.actions), N.ptr(self.rewards), N.ptr(self.terminated),
                N.ptr(self.batch_inds), self.batch_size, self.N, self.slots, self.gamma, N.ptr(w), N.ptr(td), N.ptr(self.workspace), N.ptr(self.grads),
                N.ptr(self.loss), N.ptr(o.exp_avg), N.ptr(o.exp_avg_sq), o.step_count + 1, float(g["lr"]), g["betas"][0], g["betas"][1], g["eps"],
                self.env._seed, self.update_index, upper, self._s()), "mi_dqn_td_update")
            o.step_count += 1   # committed only once the call has accepted the step
            self._after_td()
        elif native:
            # sharded, ONE C call — TD share (drawing its batch itself for plain DQN), slab sum, in-stream all-reduce of {grads, loss}, clip + Adam (mi_dqn_td_update_sharded);
            # on the P2P carrier without clipping the slab-sum launch carries the exchange and the step: two launches, as in a single process
            o = self.optimizer
            w, td = self._row_weights()
            N.check(N.lib().mi_dqn_td_update_sharded(
                N.ptr(self.q.flat), N.ptr(self.target.flat), N.ptr(self.observations), N.ptr(self.actions), N.ptr(self.rewards), N.ptr(self.terminated),
                N.ptr(self.batch_inds), self.batch_size, self.N, self.slots, self.gamma, N.ptr(w), N.ptr(td), N.ptr(self.workspace), N.ptr(self._gradbuf),
                N.ptr(o.exp_avg), N.ptr(o.exp_avg_sq), o.step_count + 1, float(g["lr"]), g["betas"][0], g["betas"][1], g["eps"], float(g["max_grad_norm"]),
                N.ptr(o.grad_norm), self.env._seed, self.update_index, upper, D.native_comm(self.pg), self._s()), "mi_dqn_td_update_sharded")
            o.step_count += 1
            self._after_td()
        else:
            self.td_grad()
            self.optimizer.step(self.grads)
        self.update_index += 1
        self._maybe_check_replicas()

    def _maybe_check_replicas(self):
        if self._check_every and self.world_size > 1 and self.update_index % self._check_every == 0:
            self.check_replicas()

    def check_replicas(self):
        """Raise MiError unless the online / target parameters and the Adam moments are bitwise equal on every rank (deep_rl_amd.dist.check_replicas)."""
        o = self.optimizer
        D.check_replicas([self.q.flat, self.target.flat, o.exp_avg, o.exp_avg_sq], self.pg, "%s parameters / Adam moments after update %d" % (type(self).__name__, self.update_index))

    def _native_sharded(self):
        """The one-call RCCL route applies to the plain TD launches (DQN, PER) of a sharded run whose process group is NCCL."""
        return ((self.world_size > 1 or _FORCE_SHARDED) and type(self).td_grad in (DQNEngine.td_grad, PERDQNEngine.td_grad)
                and os.environ.get("MIRL_NATIVE_COMM", "1") != "0" and D.native_comm(self.pg) is not None)

    def sync_target(self):
        """target_network.load_state_dict(q_network.state_dict()) (dqn.py:136-137)."""
        self.target.flat.copy_(self.q.flat)


class DuelingDQNEngine(DQNEngine):
    """The same ring and launches for reference dueling_dqn.py: the dueling head is linear in the features, so acting and the TD
    update run on the networks' plain-DQN images (`.eff`); the gradient is mapped back to the dueling parameters by
    mi_dueling_unpack_grads and the optimizer steps on those (dueling_dqn.py:71-75,109-129)."""

    def __init__(self, env, q_network, target_network, optimizer, slots, **kw):
        super().__init__(env, q_network, target_network, optimizer, slots, **kw)
        dev = self.device
        self._dgradbuf = torch.zeros(N.DUELING_NPARAMS + 2, dtype=torch.float32, device=dev)
        self.dueling_grads = self._dgradbuf[:N.DUELING_NPARAMS]

    def act(self, n_steps, forced_actions=None, forced_resets=None):
        dev = self.device
        fa = None if forced_actions is None else forced_actions.to(dev, torch.int64).contiguous()
        fr = None if forced_resets is None else forced_resets.to(dev, torch.float64).contiguous()
        self._act_launch(self.q.eff, n_steps, fa, fr)

    def td_grad(self):
        N.check(N.lib().mi_dqn_td_grad(
            N.ptr(self.q.eff), N.ptr(self.target.eff), N.ptr(self.observations), N.ptr(self.actions), N.ptr(self.rewards),
            N.ptr(self.terminated), N.ptr(self.batch_inds), self.batch_size, self.N, self.slots, self.gamma,
            1.0 / (self.batch_size * self.world_size), N.ptr(self.workspace), N.ptr(self.grads), N.ptr(self.loss), self._s()), "mi_dqn_td_grad")
        N.check(N.lib().mi_dueling_unpack_grads(N.ptr(self.grads), N.ptr(self.dueling_grads), self._s()), "mi_dueling_unpack_grads")
        if self.world_size > 1:     # gradient share + loss share in one buffer, as the other engines do
            self._dgradbuf[N.DUELING_NPARAMS:N.DUELING_NPARAMS + 1].copy_(self.loss)
            D.allreduce_sum_(self._dgradbuf, self.pg)
            self.loss.copy_(self._dgradbuf[N.DUELING_NPARAMS:N.DUELING_NPARAMS + 1])

    def train_step(self, indices=None):
        """One optimisation step (dueling_dqn.py:109-129).  Single process without gradient clipping: ONE call, two launches — the TD launch draws the batch itself and the
        launch that sums the gradient slabs maps the gradient back, steps the dueling parameters and rewrites the plain-DQN image (mi_dueling_td_update, bit-identical to
        td_grad() + optimizer.step() + repack())."""
        g = self.optimizer.param_groups[0]
        if self.world_size == 1 and not _FORCE_SHARDED and g["max_grad_norm"] == float("inf") and type(self).td_grad is DuelingDQNEngine.td_grad:
            o = self.optimizer
            upper = 0
            if indices is None and type(self).sample is DQNEngine.sample:
                upper = min(self.global_step, self.slots) * self.N
                if upper == 0:
                    raise N.MiError("train_step: the replay ring is empty (global_step == 0); act() before training")
            else:
                self.sample(indices)
            N.check(N.lib().mi_dueling_td_update(
                N.ptr(self.q.eff), N.ptr(self.target.eff), N.ptr(self.observations), N.ptr(self.actions), N.ptr(self.rewards), N.ptr(self.terminated),
                N.ptr(self.batch_inds), self.batch_size, self.N, self.slots, self.gamma, N.ptr(self.workspace), N.ptr(self.grads), N.ptr(self.loss),
                N.ptr(self.q.flat), N.ptr(self.dueling_grads), N.ptr(o.exp_avg), N.ptr(o.exp_avg_sq), o.step_count + 1, float(g["lr"]), g["betas"][0], g["betas"][1],
                g["eps"], self.env._seed, self.update_index, upper, self._s()), "mi_dueling_td_update")
            o.step_count += 1   # committed only once the call has accepted the step
        else:
            self.sample(indices)
            self.td_grad()
            self.optimizer.step(self.dueling_grads)
            self.q.repack()
        self.update_index += 1
        self._maybe_check_replicas()

    def sync_target(self):
        self.target.flat.copy_(self.q.flat)
        self.target.eff.copy_(self.q.eff)


class PERDQNEngine(DQNEngine):
    """The same ring and launches for reference per.py (prioritized replay): a priorities ring beside the replay ring, a keyed
    prefix-sum sampler instead of torch.multinomial's O(buffer) scan, importance-weighted TD loss, priority scatter and the running
    max_priority on the device (per.py:75-84,103-106,124-153)."""

    def __init__(self, env, q_network, target_network, optimizer, slots, alpha=0.6, beta_0=0.4, **kw):
        super().__init__(env, q_network, target_network, optimizer, slots, **kw)
        dev, S, Nn = self.device, self.slots, self.N
        self.alpha, self.beta_0 = float(alpha), float(beta_0)
        self.priorities = torch.zeros((S, Nn), dtype=torch.float32, device=dev)             # per.py:79
        self.max_priority = torch.full((1,), 1e-2, dtype=torch.float32, device=dev)         # :84
        self.weights = torch.zeros(self.batch_size, dtype=torch.float32, device=dev)
        self.td_abs = torch.zeros(self.batch_size, dtype=torch.float32, device=dev)
        self._owner = torch.full((S * Nn,), -1, dtype=torch.int32, device=dev)
        # the sampler's chunk sums, kept current by mark / update_priorities (zero-filled = current for the zero-filled ring)
        self._per_ws = torch.zeros(N.lib().mi_per_workspace_bytes(S * Nn), dtype=torch.uint8, device=dev)
        self._sums_owed = False   # mi_per_td_update has scattered priorities whose chunk sums are still to be rebuilt (they ride on the next acting launch)

    def refresh_sums(self):
        """Rebuild the sampler's chunk sums from `priorities` (after anything but act / train_step wrote them: checkpoint load, tests)."""
        N.check(N.lib().mi_per_sums_refresh(N.ptr(self.priorities), self.slots * self.N, self.alpha, N.ptr(self._per_ws), self._s()), "mi_per_sums_refresh")
        self._sums_owed = False

    def settle(self):
        """The chunk sums the last one-call update left owed (mi_per_td_update scatters the priorities in its slab-sum launch; the sums of the touched chunks normally
        ride on the NEXT acting launch).  Called by everything that reads the sums without an acting call in between; a no-op otherwise."""
        if self._sums_owed:
            N.check(N.lib().mi_per_settle_sums(N.ptr(self.priorities), N.ptr(self.batch_inds), self.batch_size, self.slots * self.N, self.alpha, N.ptr(self._per_ws), self._s()),
                    "mi_per_settle_sums")
            self._sums_owed = False

    def act(self, n_steps, forced_actions=None, forced_resets=None):
        """n_steps iterations of per.py:92-121 for every env.  ONE launch: the acting workgroups, the workgroups that give the new rows the running max_priority
        (per.py:105) and rebuild the touched sums, and the workgroup that rebuilds the sums the last update left owed (mi_per_act_steps)."""
        gs = self.global_step
        if not _PER_ONE_CALL:
            self.settle()
            super().act(n_steps, forced_actions, forced_resets)
            N.check(N.lib().mi_per_mark_sums(N.ptr(self.priorities), self.N, self.slots, gs, int(n_steps), N.ptr(self.max_priority), self.alpha, N.ptr(self._per_ws),
                                             self._s()), "mi_per_mark_sums")   # :105
            return
        dev = self.device
        fa = None if forced_actions is None else forced_actions.to(dev, torch.int64).contiguous()
        fr = None if forced_resets is None else forced_resets.to(dev, torch.float64).contiguous()
        owed = N.ptr(self.batch_inds) if self._sums_owed else None
        if self._lazy_stats:
            self._stats_ready, self._stats_any = False, True
            stats, nxt, eps = None, None, None
        else:
            stats, nxt, eps = self._stats2[self._stats_i], self._stats2[self._stats_i ^ 1], self.episodes
            self._stats_buf = stats
        N.check(N.lib().mi_per_act_steps(
            self.env.handle, N.ptr(self.q.flat), int(n_steps), gs, self.slots, self.learning_starts, self.start_e, self.end_e, self.exploration_fraction,
            self.total_timesteps, N.ptr(self.observation), N.ptr(self.observations), N.ptr(self.actions), N.ptr(self.rewards), N.ptr(self.terminated), N.ptr(fa), N.ptr(fr),
            N.ptr(eps), N.ptr(stats), 0 if self._lazy_stats else self.max_ep, N.ptr(nxt), N.ptr(self.priorities), N.ptr(self.max_priority), self.alpha, N.ptr(self._per_ws),
            owed, self.batch_size, self._s()), "mi_per_act_steps")
        if not self._lazy_stats:
            self._stats_i ^= 1
        self._sums_owed = False
        self.global_step += int(n_steps)

    def beta(self):
        """per.py:126: beta starts at beta_0 and increases linearly to 1."""
        return (1 - self.beta_0) * self.global_step / self.total_timesteps + self.beta_0

    def sample(self, indices=None):
        """batch_inds ~ priorities (per.py:128) and the importance weights (:131,145-146); `indices` keeps the caller's batch."""
        self.settle()
        stored = min(self.global_step, self.slots) * self.N
        if indices is not None:
            self.batch_inds.copy_(torch.as_tensor(indices, dtype=torch.int64).reshape(-1).to(self.device))
        N.check(N.lib().mi_per_sample_current(self.env._seed, self.update_index, N.ptr(self.priorities), stored, self.slots * self.N, float(stored), self.alpha,
                                              self.beta(), self.batch_size, 0 if indices is not None else 1, N.ptr(self._per_ws), N.ptr(self.batch_inds),
                                              N.ptr(self.weights), self._s()), "mi_per_sample_current")

    def train_step(self, indices=None):
        """One optimisation step (per.py:126-153).  Single process without gradient clipping: ONE call, three launches — the sampler, the weighted TD launch, and the
        slab sum + Adam launch whose last workgroup scatters the new priorities and updates max_priority (mi_per_td_update; bit-identical to sample() + td_grad() +
        optimizer.step() and to the round-5 sequence).  The sums of the scattered chunks ride on the next acting launch (`settle()` otherwise)."""
        g = self.optimizer.param_groups[0]
        one_call = (_PER_ONE_CALL and self.world_size == 1 and not _FORCE_SHARDED and g["max_grad_norm"] == float("inf")
                    and type(self).td_grad is PERDQNEngine.td_grad and type(self).sample is PERDQNEngine.sample)
        if not one_call:
            return super().train_step(indices)
        self.settle()   # (two updates in a row: the first one's scatter must be in the sums the second one draws from)
        stored = min(self.global_step, self.slots) * self.N
        if stored == 0:
            raise N.MiError("train_step: the replay ring is empty (global_step == 0); act() before training")
        if indices is not None:
            self.batch_inds.copy_(torch.as_tensor(indices, dtype=torch.int64).reshape(-1).to(self.device))
        o = self.optimizer
        N.check(N.lib().mi_per_td_update(
            N.ptr(self.q.flat), N.ptr(self.target.flat), N.ptr(self.observations), N.ptr(self.actions), N.ptr(self.rewards), N.ptr(self.terminated),
            N.ptr(self.batch_inds), self.batch_size, self.N, self.slots, self.gamma, N.ptr(self.weights), N.ptr(self.td_abs), N.ptr(self.workspace), N.ptr(self.grads),
            N.ptr(self.loss), N.ptr(o.exp_avg), N.ptr(o.exp_avg_sq), o.step_count + 1, float(g["lr"]), g["betas"][0], g["betas"][1], g["eps"], self.env._seed,
            self.update_index, N.ptr(self.priorities), stored, float(stored), self.alpha, self.beta(), 0 if indices is not None else 1, N.ptr(self._per_ws),
            N.ptr(self._owner), N.ptr(self.max_priority), self._s()), "mi_per_td_update")
        o.step_count += 1   # committed only once the call has accepted the step
        self._sums_owed = True
        self.update_index += 1
        self._maybe_check_replicas()

    def td_grad(self):
        """weighted loss + gradient (per.py:133-147), |td| per row; then priorities[batch_inds] = |td| and max_priority (:141-142)."""
        N.check(N.lib().mi_per_td_grad(
            N.ptr(self.q.flat), N.ptr(self.target.flat), N.ptr(self.observations), N.ptr(self.actions), N.ptr(self.rewards),
            N.ptr(self.terminated), N.ptr(self.batch_inds), self.batch_size, self.N, self.slots, self.gamma,
            1.0 / (self.batch_size * self.world_size), N.ptr(self.weights), N.ptr(self.td_abs), N.ptr(self.workspace), N.ptr(self.grads),
            N.ptr(self.loss), self._s()), "mi_per_td_grad")
        self._after_td()
        D.allreduce_sum_(self._gradbuf, self.pg)

    def _row_weights(self):
        return self.weights, self.td_abs

    def _after_td(self):
        N.check(N.lib().mi_per_update_priorities_sums(N.ptr(self.priorities), N.ptr(self.batch_inds), N.ptr(self.td_abs), self.batch_size, N.ptr(self._owner),
                                                      N.ptr(self.max_priority), self.slots * self.N, self.alpha, N.ptr(self._per_ws), self._s()),
                "mi_per_update_priorities_sums")
