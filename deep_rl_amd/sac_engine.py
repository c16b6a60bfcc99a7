"""SACEngine — device-resident replay ring + the launch sequence of reference sac.py:136-217.

Owns the four replay tensors with the reference's names (sac.py:126-129) plus an env axis, laid out as a [slots, N]
time-major ring (slots = total_timesteps + 1 gives the reference's linear storage), the twin critics / targets packed back to
back in one flat buffer each, and the entropy coefficient as DEVICE scalars (log_alpha, alpha): `alpha = log_alpha.exp().item()`
(sac.py:210) is the reference's one host round trip per actor update and it disappears here.  Methods are thin launch
wrappers over the C ABI (include/mi_rl.h "SAC"); nothing is computed in Python.
"""
import ctypes as C
import os

import torch

from . import _native as N
from . import dist as D
from .agent import pack
from .optim import Adam

_OWE_ALPHA = os.environ.get("MIRL_SAC_OWE_ALPHA", "1") != "0"   # 0: every alpha step is a launch of its own (A/B, debugging)
# 0: the critics' optimizer step is always the second launch of update_critic (A/B, debugging); 1: it is deferred and rides on the next acting launch (see update_critic)
_DEFER_CRITIC = os.environ.get("MIRL_SAC_DEFER_CRITIC", "1") != "0"
# diagnostics (tests/_rccl_world1_worker.py): walk the sharded branches even at world_size 1, so that RCCL really runs on a one-GPU box
_FORCE_SHARDED = os.environ.get("MIRL_OFFPOLICY_SHARDED", "0") == "1"
# MIRL_SAC_TRANSPOSED=0: no transposed layer-2 copies (the forward passes stream torch's [out][in] matrices: ~1 us per pass slower, bit-identical) — the A/B switch
_TRANSPOSED = os.environ.get("MIRL_SAC_TRANSPOSED", "1") != "0"


class SACEngine:
    def __init__(self, env, actor, qf1, qf2, qf1_target, qf2_target, slots, batch_size=256, gamma=0.99, tau=0.005, policy_lr=3e-4,
                 q_lr=1e-3, alpha_lr=None, learning_starts=5_000, target_entropy=None, max_episodes_logged=None, process_group=None):
        self.env, self.actor = env, actor
        self.N, self.device, self.slots = env.num_envs, env.device, int(slots)
        self.batch_size, self.gamma, self.tau = int(batch_size), float(gamma), float(tau)
        self.learning_starts = int(learning_starts)
        self.pg = process_group
        self.world_size, self.rank = D.world_size(process_group), D.rank(process_group)
        dev, S, Nn = self.device, self.slots, self.N
        # twin critics share one optimizer (sac.py:117): one flat buffer [2 * MI_SAC_Q_NPARAMS] each for online / target
        # (q_flat, qt_flat, q_grads, q_losses, q_optimizer are properties: reading them settles a critic step that is still owed, see update_critic)
        self._q_flat = pack(qf1, qf2)
        self._qt_flat = pack(qf1_target, qf2_target)
        self.qf1, self.qf2, self.qf1_target, self.qf2_target = qf1, qf2, qf1_target, qf2_target
        self.actor_optimizer = Adam(actor.flat, lr=policy_lr)                                  # sac.py:108
        self._q_optimizer = Adam(self._q_flat, lr=q_lr)                                        # :117
        self._owed_critic = None     # N.SacCriticStep of a deferred critic step (mi_sac_critic_update_deferred), or None
        self.alpha_lr = float(q_lr if alpha_lr is None else alpha_lr)                          # :92
        self.target_entropy = float(-1.0 if target_entropy is None else target_entropy)        # :119 (-prod(action shape))
        # the entropy coefficient's state; an alpha step may be OWED (see update_alpha): every reader goes through the properties below, which settle it first
        self._log_alpha = torch.zeros(1, dtype=torch.float32, device=dev)                      # :120
        self._alpha = torch.ones(1, dtype=torch.float32, device=dev)                           # :121 exp(0)
        self._alpha_m_t = torch.zeros(1, dtype=torch.float32, device=dev)
        self._alpha_v_t = torch.zeros(1, dtype=torch.float32, device=dev)
        self._alpha_steps = 0
        self._owed = None            # (update key of the owed step's log-prob draw, stash slot holding its observations)
        self._owed_epoch = 0         # owed steps handed to this workspace so far: the in-launch hand-off's epoch.  Engine-private and only ever grows — independent of
                                     # alpha_steps, which checkpoint.load() and the alpha_steps setter may rewind (a rewound epoch let consumers skip the wait)
        self._stash_fresh = False
        self._stash_slot = 0         # the stash slot the last fused actor update wrote
        self._owed_fits = bool(N.lib().mi_sac_owed_alpha_fits(self.batch_size))   # the carrying launch must leave half of the device's CUs free
        self.observations = torch.zeros((S, Nn, 3), dtype=torch.float32, device=dev)           # :126
        self.actions = torch.zeros((S, Nn), dtype=torch.float32, device=dev)                   # :127 (one action dim)
        self.rewards = torch.zeros((S, Nn), dtype=torch.float32, device=dev)                   # :128
        self.terminated = torch.zeros((S, Nn), dtype=torch.uint8, device=dev)                  # :129 (bool)
        self.batch_inds = torch.zeros(self.batch_size, dtype=torch.int64, device=dev)
        self._qbuf = torch.zeros(2 * N.SAC_Q_NPARAMS + 2, dtype=torch.float32, device=dev)
        self._q_grads, self._q_losses = self._qbuf[:2 * N.SAC_Q_NPARAMS], self._qbuf[2 * N.SAC_Q_NPARAMS:]
        self._abuf = torch.zeros(N.SAC_ACTOR_NPARAMS + 2, dtype=torch.float32, device=dev)
        self.actor_grads, self.actor_out = self._abuf[:N.SAC_ACTOR_NPARAMS], self._abuf[N.SAC_ACTOR_NPARAMS:]   # out = {actor_loss, mean logp}
        self._alpha_out = torch.zeros(2, dtype=torch.float32, device=dev)                      # {alpha_loss, d/d log_alpha}
        self._mean_logp = torch.zeros(1, dtype=torch.float32, device=dev)
        self.workspace = torch.zeros(N.lib().mi_sac_workspace_bytes(self.batch_size), dtype=torch.uint8, device=dev)   # zero-filled once (ticket word)
        self.max_ep = int(max_episodes_logged if max_episodes_logged is not None else (64 if Nn <= 8 else 0))
        self.episodes = torch.zeros((max(self.max_ep, 1), 4), dtype=torch.int32, device=dev)
        self.episode_stats = torch.zeros(4, dtype=torch.int32, device=dev)
        # transposed copies of the five layer-2 matrices (include/mi_rl.h mi_sac_shadow_*): single-process runs only (the sharded routes step through mi_adam / mi_polyak,
        # which do not maintain them).  The library keeps them in step with its own fused optimizer steps; torch-side writes are caught through the version counters of the
        # flat vectors and of every parameter bound to them (load_state_dict, load_flat, copy_ ...) and answered with a refresh launch.
        self._shadows = []
        if _TRANSPOSED and self._single():
            for flat, mods, is_actor in ((actor.flat, (actor,), 1), (self._q_flat, (qf1, qf2), 0), (self._qt_flat, (qf1_target, qf2_target), 0)):
                sh = torch.zeros((1 if is_actor else 2) * 65536, dtype=torch.float32, device=dev)
                rc = N.lib().mi_sac_shadow_set(N.ptr(flat), is_actor, N.ptr(sh))
                if rc != 0:     # the registry is full: engines that were dropped but not collected yet still hold its slots (their __del__ frees them) — collect, try once more
                    import gc

                    gc.collect()
                    rc = N.lib().mi_sac_shadow_set(N.ptr(flat), is_actor, N.ptr(sh))
                if rc != 0:
                    # still full (live engines): a missing shadow only means the old access pattern, so this engine runs without any — all three or none, the update
                    # launches pick per vector
                    self.close()
                    break
                self._shadows.append([flat, sh, [flat] + [p for m in mods for p in m.parameters()], None])
        self.observation = None
        self.global_step = 0
        self.update_index = 0       # critic updates done
        self.actor_updates = 0
        self._check_every = D.replica_check_interval()   # MIRL_CHECK_REPLICAS=K: every K-th train_step of a sharded run checks that the replicas still agree bitwise

    def _s(self):
        return N.stream_ptr(self.device)

    def _sync_shadows(self):
        """Refresh a transposed copy whose parameters torch has written since the last refresh (or that has never been refreshed)."""
        for ent in self._shadows:
            ver = sum(t._version for t in ent[2])
            if ver != ent[3]:
                N.check(N.lib().mi_sac_shadow_refresh(N.ptr(ent[0]), self._s()), "mi_sac_shadow_refresh")
                ent[3] = ver

    def params_changed(self):
        """Tell the engine that actor / critic / target parameters were written in a way torch's version counters do not see (`.data` in-place ops, foreign kernels)."""
        for ent in self._shadows:
            ent[3] = None

    def close(self):
        """Unregister this engine's transposed copies from the library (idempotent; the engine keeps working on the plain access pattern).  Call it when an engine is
        dropped in a loop that builds many: `__del__` does the same, but only when the object is collected."""
        for ent in getattr(self, "_shadows", []):
            N.lib().mi_sac_shadow_set(N.ptr(ent[0]), 0, None)
        self._shadows = []

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001  (interpreter shutdown)
            pass

    def _single(self):
        return self.world_size == 1 and not _FORCE_SHARDED

    def _comm(self):
        """libmirl's RCCL communicator when the sharded updates can run as ONE C call each (NCCL process group), else None (host-sequenced: gloo)."""
        if os.environ.get("MIRL_NATIVE_COMM", "1") == "0":
            return None
        return D.native_comm(self.pg)

    # ---- the critics' state: reading it settles a deferred optimizer step first ----
    def flush_critic(self):
        """Run a deferred critic step now, as the launch of its own that update_critic would have made (mi_sac_critic_step)."""
        if self._owed_critic is not None:
            self._sync_shadows()
        st = self._owed_critic
        if st is not None:
            N.check(N.lib().mi_sac_critic_step(C.byref(st), self._s()), "mi_sac_critic_step")
            self._owed_critic = None

    def _critic_settled(self, t):
        self.flush_critic()
        return t

    q_flat = property(lambda self: self._critic_settled(self._q_flat))
    qt_flat = property(lambda self: self._critic_settled(self._qt_flat))
    q_grads = property(lambda self: self._critic_settled(self._q_grads))
    q_losses = property(lambda self: self._critic_settled(self._q_losses))
    q_optimizer = property(lambda self: self._critic_settled(self._q_optimizer))

    def flush(self):
        """Settle everything that is owed (critic step, alpha step): the engine's tensors then hold what the reference's variables would."""
        self.flush_critic()
        self.flush_alpha()

    # ---- the entropy coefficient's state: reading it settles an owed alpha step first ----
    def _owed_struct(self):
        """-> (ctypes struct of the owed alpha step | None) for the NEXT counters.  Nothing is committed here: the C call may refuse before it launches anything
        (MI_ESTATE from the status word), and then the debt, the Adam step count and the epoch must stay as they were — `_owed_commit()` after N.check.
        The epoch is a plain wrapping 32-bit counter (the kernel compares `word - epoch` as a signed 32-bit difference, which needs the full mod-2^32 range)."""
        if self._owed is None:
            return None
        key, slot = self._owed
        e = (self._owed_epoch + 1) & 0xFFFFFFFF
        return N.SacOwedAlpha(N.ptr(self._log_alpha), N.ptr(self._alpha_m_t), N.ptr(self._alpha_v_t), N.ptr(self._alpha), N.ptr(self._alpha_out),
                              self.target_entropy, self._alpha_steps + 1, self.alpha_lr, key, e - (1 << 32) if e >= (1 << 31) else e, slot)

    def _owed_commit(self, o):
        """The launch carrying the owed step `o` has been enqueued."""
        if o is not None:
            self._alpha_steps += 1
            self._owed_epoch += 1
            self._owed = None

    def drop_owed(self):
        """Forget a pending alpha debt, the stash and a deferred critic step (the state is about to be replaced wholesale: checkpoint.load)."""
        self._owed = None
        self._stash_fresh = False
        self._owed_critic = None

    def flush_alpha(self):
        """Run an owed alpha step now (a launch of its own)."""
        self._sync_shadows()
        o = self._owed_struct()
        if o is not None:
            self.flush_critic()      # (every launch but the acting one may use workspace regions a deferred critic step still reads)
        if o is not None:
            N.check(N.lib().mi_sac_alpha_step_owed(N.ptr(self.actor.flat), self.batch_size, self.env._seed, C.byref(o), N.ptr(self.workspace), self._s()),
                    "mi_sac_alpha_step_owed")
            self._owed_commit(o)

    def check(self, wait=True):
        """Raise MiError (MI_ESTATE) if a wait between the workgroups of one of this process's SAC launches has timed out (include/mi_rl.h: mi_sac_check);
        wait=True synchronises the stream first."""
        N.check(N.lib().mi_sac_check(self._s(), 1 if wait else 0), "mi_sac_check")

    def clear_error(self):
        """After restoring the state (e.g. checkpoint.load): clear the status word and the workspace's hand-off words."""
        self.drop_owed()
        N.check(N.lib().mi_sac_clear_error(N.ptr(self.workspace), self.batch_size, self._s()), "mi_sac_clear_error")

    def _settled(self, t):
        self.flush_alpha()
        return t

    log_alpha = property(lambda self: self._settled(self._log_alpha))
    alpha = property(lambda self: self._settled(self._alpha))
    alpha_out = property(lambda self: self._settled(self._alpha_out))
    _alpha_m = property(lambda self: self._settled(self._alpha_m_t))
    _alpha_v = property(lambda self: self._settled(self._alpha_v_t))

    @property
    def alpha_steps(self):
        return self._alpha_steps + (1 if self._owed is not None else 0)

    @alpha_steps.setter
    def alpha_steps(self, v):
        self.flush_alpha()
        self._alpha_steps = int(v)

    def _key(self, counter):
        """per-call key of the in-kernel normal draws, distinct per rank"""
        return counter * self.world_size + self.rank

    def reset(self, forced_state=None):
        """observation = env.reset(); observations[global_step] = observation (sac.py:132-134)."""
        self.observation = self.env.reset(forced_state)
        self.observations[self.global_step % self.slots].copy_(self.observation)
        return self.observation

    def act(self, forced_actions=None, forced_eps=None, forced_resets=None):
        """One iteration of sac.py:138-158 for every env, one launch."""
        self._sync_shadows()
        dev = self.device
        fa = None if forced_actions is None else forced_actions.to(dev, torch.float32).reshape(self.N).contiguous()
        fe = None if forced_eps is None else forced_eps.to(dev, torch.float32).reshape(self.N).contiguous()
        fr = None if forced_resets is None else forced_resets.to(dev, torch.float64).contiguous()
        st = self._owed_critic    # a deferred critic step rides on this launch: acting reads the actor and the env only (mi_sac_act_step_carry)
        N.check(N.lib().mi_sac_act_step_carry(
            self.env.handle, N.ptr(self.actor.flat), self.global_step, self.slots, self.learning_starts, N.ptr(self.observation),
            N.ptr(self.observations), N.ptr(self.actions), N.ptr(self.rewards), N.ptr(self.terminated), N.ptr(fa), N.ptr(fe), N.ptr(fr),
            N.ptr(self.episodes), N.ptr(self.episode_stats) if self.max_ep else None, self.max_ep, C.byref(st) if st is not None else None, self._s()),
            "mi_sac_act_step_carry")
        self._owed_critic = None
        self.global_step += 1

    def drain_episodes(self):
        """Host sync. -> [(env, return, length)] finished by the last act() call."""
        if not self.max_ep:
            return []
        st = self.episode_stats.tolist()
        k = min(st[3], self.max_ep)
        if k == 0:
            return []
        raw = self.episodes[:k].cpu()
        rets = raw[:, 2].contiguous().view(torch.float32)
        return sorted((int(raw[i, 0]), float(rets[i]), int(raw[i, 3])) for i in range(k))

    def sample(self, indices=None):
        """batch_inds = np.random.randint(global_step, size=batch_size) (sac.py:162), flat over [slot][env]."""
        if indices is not None:
            self.batch_inds.copy_(torch.as_tensor(indices, dtype=torch.int64).reshape(-1).to(self.device))
            return
        upper = min(self.global_step, self.slots) * self.N
        N.check(N.lib().mi_dqn_sample(self.env._seed, self._key(self.update_index), upper, self.batch_size, N.ptr(self.batch_inds), self._s()),
                "mi_dqn_sample")

    def critic_grad(self, eps=None):
        """sac.py:170-182 + backward -> self.q_grads [2 * MI_SAC_Q_NPARAMS], self.q_losses (all-reduced when sharded)."""
        self._sync_shadows()
        e = None if eps is None else eps.to(self.device, torch.float32).reshape(-1).contiguous()
        N.check(N.lib().mi_sac_critic_grad(
            N.ptr(self.q_flat), N.ptr(self.qt_flat), N.ptr(self.actor.flat), N.ptr(self.observations), N.ptr(self.actions), N.ptr(self.rewards),
            N.ptr(self.terminated), N.ptr(self.batch_inds), self.batch_size, self.N, self.slots, N.ptr(e), self.env._seed,
            self._key(self.update_index), N.ptr(self.alpha), self.gamma, 1.0 / (self.batch_size * self.world_size), N.ptr(self.workspace),
            N.ptr(self.q_grads), N.ptr(self.q_losses), self._s()), "mi_sac_critic_grad")
        D.allreduce_sum_(self._qbuf, self.pg)

    def update_critic(self, eps=None, polyak=False, sample_in_launch=False):
        """sac.py:170-185 (+ the target update of :213-217 when `polyak`).  Single process: ONE fused call — the launch that assembles the
        gradient also applies Adam and the polyak step; sharded: gradient, all-reduce, Adam (, polyak)."""
        self._sync_shadows()
        if self._single():
            e = None if eps is None else eps.to(self.device, torch.float32).reshape(-1).contiguous()
            o = self.q_optimizer
            g = o.param_groups[0]
            if sample_in_launch and min(self.global_step, self.slots) * self.N == 0:   # 0 means "read batch_inds" to the launch
                raise N.MiError("update_critic: the replay ring is empty (global_step == 0); act() before training")
            owed = self._owed_struct()       # an alpha step owed from the last actor update rides on this launch
            upper = min(self.global_step, self.slots) * self.N if sample_in_launch else 0
            if _DEFER_CRITIC:
                # the row-group launch now, the optimizer step (dW2 GEMM + assembly + Adam + polyak) DEFERRED: the next act() carries it on workgroups of its own launch
                # (the step touches the critics only, acting reads the actor and the env only), anything else that comes first settles it alone (flush_critic / the
                # q_* properties).  One launch less on the chain of every iteration that is followed by an acting step; same arithmetic, same bits.
                N.check(N.lib().mi_sac_critic_update_deferred(
                    N.ptr(self.q_flat), N.ptr(self.qt_flat), N.ptr(self.actor.flat), N.ptr(self.observations), N.ptr(self.actions), N.ptr(self.rewards),
                    N.ptr(self.terminated), N.ptr(self.batch_inds), self.batch_size, self.N, self.slots, N.ptr(e), self.env._seed,
                    self._key(self.update_index), N.ptr(self._alpha), self.gamma, N.ptr(self.workspace), self._key(self.update_index), upper,
                    C.byref(owed) if owed is not None else None, self._s()), "mi_sac_critic_update_deferred")
                self._owed_critic = N.SacCriticStep(N.ptr(self.workspace), self.batch_size, N.ptr(self._q_flat), N.ptr(self._qt_flat), N.ptr(o.exp_avg), N.ptr(o.exp_avg_sq),
                                                    N.ptr(self._q_grads), N.ptr(self._q_losses), o.step_count + 1, float(g["lr"]), g["betas"][0], g["betas"][1], g["eps"],
                                                    self.tau if polyak else -1.0)
            else:
                N.check(N.lib().mi_sac_critic_update_owed(
                    N.ptr(self.q_flat), N.ptr(self.qt_flat), N.ptr(self.actor.flat), N.ptr(self.observations), N.ptr(self.actions), N.ptr(self.rewards),
                    N.ptr(self.terminated), N.ptr(self.batch_inds), self.batch_size, self.N, self.slots, N.ptr(e), self.env._seed,
                    self._key(self.update_index), N.ptr(self._alpha), self.gamma, N.ptr(self.workspace), N.ptr(self.q_grads), N.ptr(self.q_losses),
                    N.ptr(o.exp_avg), N.ptr(o.exp_avg_sq), o.step_count + 1, float(g["lr"]), g["betas"][0], g["betas"][1], g["eps"],
                    self.tau if polyak else -1.0, self._key(self.update_index), upper,
                    C.byref(owed) if owed is not None else None, self._s()), "mi_sac_critic_update_owed")
            o.step_count += 1                # counters move only once the call has enqueued its launches (it may refuse with MI_ESTATE before)
            self._owed_commit(owed)
        elif self._comm() is not None:
            # sharded, NCCL process group: ONE C call — gradient share, in-stream RCCL all-reduce of {grads, losses}, Adam, polyak (mi_sac_critic_update_sharded)
            e = None if eps is None else eps.to(self.device, torch.float32).reshape(-1).contiguous()
            o = self.q_optimizer
            g = o.param_groups[0]
            N.check(N.lib().mi_sac_critic_update_sharded(
                N.ptr(self.q_flat), N.ptr(self.qt_flat), N.ptr(self.actor.flat), N.ptr(self.observations), N.ptr(self.actions), N.ptr(self.rewards),
                N.ptr(self.terminated), N.ptr(self.batch_inds), self.batch_size, self.N, self.slots, N.ptr(e), self.env._seed, self._key(self.update_index),
                N.ptr(self.alpha), self.gamma, N.ptr(self.workspace), N.ptr(self._qbuf), N.ptr(o.exp_avg), N.ptr(o.exp_avg_sq), o.step_count + 1, float(g["lr"]),
                g["betas"][0], g["betas"][1], g["eps"], self.tau if polyak else -1.0, self._comm(), self._s()), "mi_sac_critic_update_sharded")
            o.step_count += 1
        else:
            self.critic_grad(eps)
            self.q_optimizer.step(self.q_grads)
            if polyak:
                self.update_targets()
        self.update_index += 1

    def actor_grad(self, eps=None):
        """sac.py:189-193 + backward -> self.actor_grads, self.actor_out."""
        self._sync_shadows()
        e = None if eps is None else eps.to(self.device, torch.float32).reshape(-1).contiguous()
        N.check(N.lib().mi_sac_actor_grad(
            N.ptr(self.actor.flat), N.ptr(self.q_flat), N.ptr(self.observations), N.ptr(self.batch_inds), self.batch_size, N.ptr(e),
            self.env._seed, self._key(self.actor_updates), N.ptr(self.alpha), 1.0 / (self.batch_size * self.world_size), N.ptr(self.workspace),
            N.ptr(self.actor_grads), N.ptr(self.actor_out), self._s()), "mi_sac_actor_grad")
        D.allreduce_sum_(self._abuf, self.pg)

    def update_actor(self, eps=None):
        """sac.py:189-197 (single process: one fused call, as update_critic)."""
        self._sync_shadows()
        if self._single():
            e = None if eps is None else eps.to(self.device, torch.float32).reshape(-1).contiguous()
            o = self.actor_optimizer
            g = o.param_groups[0]
            owed = self._owed_struct()
            N.check(N.lib().mi_sac_actor_update_owed(
                N.ptr(self.actor.flat), N.ptr(self.q_flat), N.ptr(self.observations), N.ptr(self.batch_inds), self.batch_size, N.ptr(e),
                self.env._seed, self._key(self.actor_updates), N.ptr(self._alpha), N.ptr(self.workspace), N.ptr(self.actor_grads),
                N.ptr(self.actor_out), N.ptr(o.exp_avg), N.ptr(o.exp_avg_sq), o.step_count + 1, float(g["lr"]), g["betas"][0], g["betas"][1], g["eps"],
                C.byref(owed) if owed is not None else None, self._s()), "mi_sac_actor_update_owed")
            o.step_count += 1
            self._owed_commit(owed)
            # this launch stashed its batch observations — in the slot a debt it carried did not read — so an alpha step may be owed on them
            self._stash_slot = (owed.stash_slot ^ 1) if owed is not None else 0
            self._stash_fresh = True
        elif self._comm() is not None:
            e = None if eps is None else eps.to(self.device, torch.float32).reshape(-1).contiguous()
            o = self.actor_optimizer
            g = o.param_groups[0]
            N.check(N.lib().mi_sac_actor_update_sharded(
                N.ptr(self.actor.flat), N.ptr(self.q_flat), N.ptr(self.observations), N.ptr(self.batch_inds), self.batch_size, N.ptr(e), self.env._seed,
                self._key(self.actor_updates), N.ptr(self.alpha), N.ptr(self.workspace), N.ptr(self._abuf), N.ptr(o.exp_avg), N.ptr(o.exp_avg_sq), o.step_count + 1,
                float(g["lr"]), g["betas"][0], g["betas"][1], g["eps"], self._comm(), self._s()), "mi_sac_actor_update_sharded")
            o.step_count += 1
        else:
            self.actor_grad(eps)
            self.actor_optimizer.step(self.actor_grads)

    def update_alpha(self, eps=None):
        """sac.py:199-207: fresh log-probs, alpha loss, Adam on log_alpha, alpha = exp(log_alpha) — all on the device."""
        self._sync_shadows()
        self.flush_critic()          # the log-prob launch writes gradient slabs a deferred critic step would still read
        if eps is None and self._single() and self._stash_fresh and self._owed_fits and _OWE_ALPHA:
            # keyed draws, single process, right after a fused actor update: the step is OWED — the next row-group launch (actor or critic update) carries its
            # log-prob pass on workgroups of its own and hands alpha to its consumers in the launch; reading the state (or flush_alpha()) settles it alone
            self.flush_alpha()
            self._owed = (self._key(self.actor_updates), self._stash_slot)
            self._stash_fresh = False
            self.actor_updates += 1
            return
        self.flush_alpha()
        self._stash_fresh = False
        e = None if eps is None else eps.to(self.device, torch.float32).reshape(-1).contiguous()
        step = self._alpha_steps + 1      # committed after the call (it may refuse before launching)
        L = N.lib()
        if self._single():
            N.check(L.mi_sac_alpha_step(
                N.ptr(self.actor.flat), N.ptr(self.observations), N.ptr(self.batch_inds), self.batch_size, N.ptr(e), self.env._seed,
                self._key(self.actor_updates), self.target_entropy, N.ptr(self.log_alpha), N.ptr(self._alpha_m), N.ptr(self._alpha_v),
                step, self.alpha_lr, N.ptr(self.alpha), N.ptr(self.alpha_out), N.ptr(self.workspace), self._s()), "mi_sac_alpha_step")
        elif self._comm() is not None:
            N.check(L.mi_sac_alpha_step_sharded(
                N.ptr(self.actor.flat), N.ptr(self.observations), N.ptr(self.batch_inds), self.batch_size, N.ptr(e), self.env._seed, self._key(self.actor_updates),
                self.target_entropy, N.ptr(self.log_alpha), N.ptr(self._alpha_m), N.ptr(self._alpha_v), step, self.alpha_lr, N.ptr(self.alpha),
                N.ptr(self.alpha_out), N.ptr(self._mean_logp), N.ptr(self.workspace), self._comm(), self._s()), "mi_sac_alpha_step_sharded")
        else:
            N.check(L.mi_sac_mean_logp(
                N.ptr(self.actor.flat), N.ptr(self.observations), N.ptr(self.batch_inds), self.batch_size, N.ptr(e), self.env._seed,
                self._key(self.actor_updates), 1.0 / (self.batch_size * self.world_size), N.ptr(self._mean_logp), N.ptr(self.workspace), self._s()),
                "mi_sac_mean_logp")
            D.allreduce_sum_(self._mean_logp, self.pg)
            N.check(L.mi_sac_alpha_adam(
                N.ptr(self._mean_logp), self.target_entropy, N.ptr(self.log_alpha), N.ptr(self._alpha_m), N.ptr(self._alpha_v), step,
                self.alpha_lr, N.ptr(self.alpha), N.ptr(self.alpha_out), self._s()), "mi_sac_alpha_adam")
        self._alpha_steps = step
        self.actor_updates += 1

    def update_targets(self):
        """sac.py:213-217 for both target critics in one launch."""
        N.check(N.lib().mi_polyak(N.ptr(self.qt_flat), N.ptr(self.q_flat), self.q_flat.numel(), self.tau, self._s()), "mi_polyak")
        if self._shadows:
            self._shadows[2][3] = None   # mi_polyak does not maintain the targets' transposed copy: refreshed before the next launch

    def train_step(self, policy_frequency=2, target_network_frequency=1, indices=None):
        """The optimisation half of one loop iteration (sac.py:161-217) at the current global_step.  The target update uses the critic
        parameters, which the actor / alpha updates do not touch, so it rides on the critic update's last launch."""
        in_launch = indices is None and self._single()            # the critic launch draws the batch indices itself (same contract as sample())
        if not in_launch:
            self.sample(indices)
        self.update_critic(polyak=self.global_step % target_network_frequency == 0, sample_in_launch=in_launch)
        if self.global_step % policy_frequency == 0:
            for _ in range(policy_frequency):
                self.update_actor()
                self.update_alpha()
        if self._check_every and self.world_size > 1 and self.update_index % self._check_every == 0:
            self.check_replicas()

    def check_replicas(self):
        """Raise MiError unless actor, critics, targets, log_alpha and every Adam moment are bitwise equal on every rank (deep_rl_amd.dist.check_replicas)."""
        a, q = self.actor_optimizer, self.q_optimizer
        D.check_replicas([self.actor.flat, self.q_flat, self.qt_flat, self.log_alpha, a.exp_avg, a.exp_avg_sq, q.exp_avg, q.exp_avg_sq, self._alpha_m, self._alpha_v], self.pg,
                         "SAC parameters / Adam moments after update %d" % self.update_index)
