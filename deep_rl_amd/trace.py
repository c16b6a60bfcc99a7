"""Trace dump of a device run in the golden fixtures' key layout ("checkpoint + trace dump in the fixture format").

The reference has no trace facility; the format to match is the one the capture scripts write for the UNMODIFIED reference scripts
(oracle/capture_ppo_trace.py:126-157, capture_dqn_trace.py:83-106, capture_sac_trace.py:124-141): one ``.npz`` whose keys a parity test can
diff against ``tests/golden/*_ref_trace.npz`` one by one.  A recorder wraps an engine; the training loop calls the recorder's ``reset`` /
``update`` (PPO) or ``act`` / ``train_step`` (DQN, SAC) instead of the engine's, and ``save(path)`` writes the file.

What the fused kernels never store — the float64 state before a step, the observation gym's ``env.step`` returned BEFORE the script's
``if done: observation = env.reset()``, terminated vs truncated — comes from a SHADOW env: a second handle with the same keys that is synchronised to
the engine's env (mi_env_export_state / import_state) and re-walks the same actions step by step through ``mi_env_step_ex``.  The walk doubles as a
self-check: the shadow's observations / rewards / flags must equal what the fused launch wrote to the storage BIT FOR BIT, else ``TraceError``.

With an env axis N the per-step keys are flattened in (time, env) order — env-step index ``t * N + e`` inside a rollout / acting call, the order
the drop-in scripts print episodes in — and the ``updK_*`` snapshots drop the env axis when N == 1, so a one-env run has exactly the reference's shapes.
Per-optimizer-step keys need every step's loss terms, so ``PPOTrace.update`` walks the explicit launch sequence {minibatch_grad, clip + Adam} x 16
(bit-identical to the one-call ``engine.update()``: tests/test_gpu_fullsize.py::test_ppo_update_equals_launch_sequence_bitwise).
Tracing synchronises and copies to the host after every launch: a debugging facility, not a training mode.
"""
import time

import numpy as np
import torch

from . import _native as N
from .envs import make


class TraceError(N.MiError):
    pass


def _np(t):
    return t.detach().cpu().numpy()


class _Shadow:
    """A second env handle walking the engine's transitions one launch per step."""

    def __init__(self, env):
        self.src = env
        self.env = make(env.spec.id, num_envs=env.num_envs, device=env.device, seed=env._seed, env_id_base=env.env_id_base)
        self.N, self.dev = env.num_envs, env.device
        self.obs_dim = env.observation_space.shape[0]
        self.state_dim = 2 if env.spec.id == "Pendulum-v1" else 4
        self._blob = torch.empty(N.lib().mi_env_state_bytes(env.handle), dtype=torch.uint8, device=self.dev)

    def sync(self):
        """shadow <- the engine's env, as it stands on the stream."""
        s = N.stream_ptr(self.dev)
        N.check(N.lib().mi_env_export_state(self.src.handle, N.ptr(self._blob), s), "mi_env_export_state")
        N.check(N.lib().mi_env_import_state(self.env.handle, N.ptr(self._blob), s), "mi_env_import_state")

    def state(self):
        st = torch.empty((self.N, self.state_dim), dtype=torch.float64, device=self.dev)
        N.check(N.lib().mi_env_get_state(self.env.handle, N.ptr(st), None, N.stream_ptr(self.dev)), "mi_env_get_state")
        return st

    def walk(self, actions, forced_resets=None):
        """actions [T, N] (i64 CartPole / f32 Pendulum), forced_resets [T, N, state_dim] f64 or None -> dict of host arrays, one row per step."""
        T, n, dev, L, s = actions.shape[0], self.N, self.dev, N.lib(), N.stream_ptr(self.dev)
        actions = actions.contiguous()
        fr = None if forced_resets is None else forced_resets.to(dev, torch.float64).contiguous()
        z = lambda *sh, dt=torch.float32: torch.empty(sh, dtype=dt, device=dev)  # noqa: E731
        out = {"state_before": z(T, n, self.state_dim, dt=torch.float64), "state_after": z(T, n, self.state_dim, dt=torch.float64), "obs": z(T, n, self.obs_dim),
               "raw_obs": z(T, n, self.obs_dim), "reward": z(T, n), "done": z(T, n, dt=torch.uint8), "truncated": z(T, n, dt=torch.uint8), "fin_ret": z(T, n),
               "fin_len": z(T, n, dt=torch.int32)}
        out["state_before"][0].copy_(self.state())
        for t in range(T):
            N.check(L.mi_env_step_ex(self.env.handle, N.ptr(actions[t]), None if fr is None else N.ptr(fr[t]), N.ptr(out["obs"][t]), N.ptr(out["reward"][t]),
                                     N.ptr(out["done"][t]), N.ptr(out["truncated"][t]), N.ptr(out["fin_ret"][t]), N.ptr(out["fin_len"][t]), N.ptr(out["raw_obs"][t]), s),
                    "mi_env_step_ex")
            N.check(L.mi_env_get_state(self.env.handle, N.ptr(out["state_after"][t]), None, s), "mi_env_get_state")
            if t + 1 < T:
                out["state_before"][t + 1].copy_(out["state_after"][t])
        return {k: _np(v) for k, v in out.items()}

    def close(self):
        self.env.close()


class _EnvLog:
    """The env-side keys every fixture shares: reset_states, actions_all, (raw) observations, terminated, after_reset, episodes."""

    def __init__(self):
        self.reset_states, self.actions, self.raw_obs, self.rewards, self.terminated, self.after_reset, self.states = [], [], [], [], [], [], []
        self.ep_step, self.ep_ret = [], []
        self._pending = None   # per env: the next step is the first after a reset

    def on_reset(self, state_f64):
        self.reset_states.append(np.asarray(state_f64, np.float64).reshape(-1, state_f64.shape[-1]))
        self._pending = np.ones(state_f64.shape[0], np.uint8)

    def on_walk(self, w, actions, step0, printed_offset):
        """w: _Shadow.walk output of T steps starting at env-step row step0 (in units of time steps); printed_offset: 0 if the script prints global_step before
        its increment (ppo.py:130), 1 if after (dqn.py:110, sac.py:146)."""
        T, n = w["done"].shape
        for t in range(T):
            self.actions.append(actions[t]); self.raw_obs.append(w["raw_obs"][t]); self.rewards.append(w["reward"][t].astype(np.float64))
            self.terminated.append((w["done"][t].astype(bool) & ~w["truncated"][t].astype(bool)).astype(np.uint8))
            self.after_reset.append(self._pending.copy()); self.states.append(w["state_before"][t])
            d = w["done"][t].astype(bool)
            self._pending = d.astype(np.uint8)
            if d.any():
                self.reset_states.append(w["state_after"][t][d])
                for e in np.flatnonzero(d):
                    self.ep_step.append((step0 + t) * n + e + (printed_offset if n == 1 else printed_offset * n))
                    self.ep_ret.append(w["fin_ret"][t][e])

    def keys(self, act_dtype):
        cat = lambda xs, dt: np.concatenate(xs).astype(dt) if xs else np.zeros(0, dt)  # noqa: E731
        return {"reset_states": np.concatenate(self.reset_states) if self.reset_states else np.zeros((0, 4)),
                "actions_all": np.concatenate(self.actions).astype(act_dtype) if self.actions else np.zeros(0, act_dtype),
                "terminated_all": cat(self.terminated, np.uint8), "after_reset_all": cat(self.after_reset, np.uint8),
                "episode_global_step": np.asarray(self.ep_step, np.int32), "episode_return": np.asarray(self.ep_ret, np.float32)}


def _check_equal(what, a, b):
    if not np.array_equal(a, b):
        bad = np.argwhere(np.asarray(a) != np.asarray(b))
        raise TraceError("trace self-check: the shadow env's %s differ from what the fused launch stored (%d elements, first at %s)" % (what, len(bad), bad[0].tolist()))


# =====================================================================================================================================
class PPOTrace:
    """Records a PPOEngine run under the keys of tests/golden/ppo_ref_trace.npz (oracle/capture_ppo_trace.py:126-157).

    hparams: the script's constants {total_timesteps, num_updates, learning_rate, seed} (the rest is read off the engine)."""

    NAMES = ["observations", "values", "actions", "log_probs", "rewards", "dones", "advantages", "returns"]

    def __init__(self, engine, hparams, full_updates=3, full_opt_steps=16, full_env_steps=512, max_mb_inds=4096):
        self.e, self.hp = engine, dict(hparams)
        self.full_updates, self.full_opt_steps, self.full_env_steps, self.max_mb_inds = full_updates, full_opt_steps, full_env_steps, max_mb_inds
        self.shadow = _Shadow(engine.env)
        self.log = _EnvLog()
        self.init_params = _np(engine.agent.flat).copy()
        self.opt, self.clip_norm, self.mb_inds, self.full_grads, self.full_params, self.update_sums, self.updates = [], [], [], [], [], [], []
        self.time_steps = 0
        self.t0 = time.time()

    def reset(self, forced_state=None):
        """observation = env.reset() (ppo.py:101), recorded."""
        obs = self.e.reset(forced_state)
        self.shadow.sync()
        self.log.on_reset(_np(self.shadow.state()))
        return obs

    def update(self, forced_actions=None, forced_uniforms=None, forced_resets=None, mb_inds=None):
        """One outer update (ppo.py:110-192) through the explicit launch sequence, recorded.  forced_*: parity inputs of engine.rollout; mb_inds: one index array
        per optimizer step (update_epochs * n_minibatch of them) replacing the keyed permutations (ppo.py:155-164)."""
        e, T, n = self.e, self.e.T, self.e.N
        self.shadow.sync()
        params_before = _np(e.agent.flat).copy()
        e.rollout(forced_actions, forced_uniforms, forced_resets)
        e.compute_gae()
        acts = e.actions[:T]
        w = self.shadow.walk(acts, forced_resets)
        _check_equal("observations", w["obs"], _np(e.observations[1:]))
        _check_equal("dones", w["done"].astype(np.float32), _np(e.dones[1:]))
        _check_equal("rewards", w["reward"], _np(e.rewards[1:]))
        self.log.on_walk(w, _np(acts), self.time_steps, 0)
        self.time_steps += T
        snap = {nm: _np(getattr(e, nm)).copy() for nm in self.NAMES}
        self.update_sums.append([float(snap[nm].astype(np.float64).sum()) for nm in self.NAMES])
        if len(self.updates) < self.full_updates:
            if n == 1:
                snap = {nm: v[:, 0] for nm, v in snap.items()}
            snap["params_before"] = params_before
            self.updates.append(snap)
        k_in_update = 0
        for epoch in range(e.update_epochs):
            if mb_inds is None:
                e.perm = e._perm_all[epoch]; e.adv_sums = e._adv_sums_all[epoch]
                e.make_perm(epoch)
                e.adv_stats()
            for k in range(e.n_minibatch):
                if mb_inds is not None:
                    idx = np.asarray(mb_inds[k_in_update]).astype(np.int32)
                    e.perm[:len(idx)].copy_(torch.from_numpy(idx).to(e.device))
                    e.adv_stats(mb=len(idx), n_mb=1)
                    e.minibatch_grad(0, mb=len(idx))
                else:
                    idx = None
                    e.minibatch_grad(k)
                grads = _np(e.grads).copy() if len(self.opt) < self.full_opt_steps else None   # pre-clip, as clip_grad_norm_ sees them (ppo.py:191)
                e.optimizer_step()
                pa = _np(e.agent.flat)
                terms = _np(e.loss_terms)
                self.opt.append([float(terms[0]), float(terms[1]), float(terms[2]), float(terms[3]), float(e.optimizer.param_groups[0]["lr"]),
                                 float(pa.astype(np.float64).sum()), float(np.abs(pa.astype(np.float64)).sum())])
                self.clip_norm.append(float(e.optimizer.grad_norm.item()))
                if idx is None and (e.minibatch_size <= self.max_mb_inds or len(self.mb_inds) < self.full_opt_steps):
                    idx = _np(e.perm[k * e.minibatch_size:(k + 1) * e.minibatch_size])
                if idx is not None:
                    self.mb_inds.append(idx.astype(np.int16 if e.batch_size <= 32767 else np.int32))
                if grads is not None:
                    self.full_grads.append(grads); self.full_params.append(pa.copy())
                k_in_update += 1
        e.perm, e.adv_sums = e._perm_all[0], e._adv_sums_all[0]
        e.update_index += 1

    def state_dict(self):
        e, hp = self.e, self.hp
        g = e.optimizer.param_groups[0]
        out = {"hparams": np.array([hp["total_timesteps"], e.T, hp["num_updates"], e.minibatch_size, e.update_epochs, e.gamma, e.gae_lambda, hp["learning_rate"],
                                    e.clip_coef, e.ent_coef, e.vf_coef, g["max_grad_norm"], hp["seed"]], dtype=np.float64),
               "init_params": self.init_params, "final_params": _np(e.agent.flat).copy()}
        out.update(self.log.keys(np.int8))
        out["obs_all"] = np.concatenate(self.log.raw_obs).astype(np.float32).reshape(-1, 4) if self.log.raw_obs else np.zeros((0, 4), np.float32)
        st = np.concatenate(self.log.states).reshape(-1, 4) if self.log.states else np.zeros((0, 4))
        out["state_first"] = st[:self.full_env_steps]
        out.update({"opt_terms": np.array(self.opt, dtype=np.float64).reshape(-1, 7), "clip_norm": np.array(self.clip_norm, dtype=np.float64),
                    "update_sums": np.array(self.update_sums, dtype=np.float64).reshape(-1, 8),
                    "final_global_step": np.array([self.time_steps * e.N], dtype=np.int64),
                    "final_explained_var": np.array([float(e.compute_explained_var().item())], dtype=np.float64),
                    "ref_wall_seconds": np.array([time.time() - self.t0])})
        if self.mb_inds and all(len(m) == len(self.mb_inds[0]) for m in self.mb_inds):
            out["mb_inds"] = np.stack(self.mb_inds)
        if self.full_grads:
            out["full_grads"] = np.stack(self.full_grads); out["full_params"] = np.stack(self.full_params)
        for i, snap in enumerate(self.updates):
            for nm, v in snap.items():
                out["upd%d_%s" % (i, nm)] = v
        return out

    def save(self, path):
        path = str(path) if str(path).endswith(".npz") else str(path) + ".npz"
        with open(path, "wb") as f:
            np.savez_compressed(f, **self.state_dict())
        return path


# =====================================================================================================================================
class DQNTrace:
    """Records a DQNEngine run under the keys of tests/golden/dqn_ref_trace.npz (oracle/capture_dqn_trace.py:83-106).
    hparams: {train_frequency, learning_rate, target_network_frequency, seed}."""

    def __init__(self, engine, hparams, full_steps=8, full_inds=64, obs_first=12000, checkpoints=()):
        self.e, self.hp = engine, dict(hparams)
        self.full_steps, self.full_inds, self.obs_first, self.checkpoints = full_steps, full_inds, obs_first, set(int(c) for c in checkpoints)
        self.shadow = _Shadow(engine.env)
        self.log = _EnvLog()
        self.init_params = _np(engine.q.flat).copy()
        self.loss, self.psum, self.inds_sum, self.gs, self.inds, self.grads, self.params, self.ck = [], [], [], [], [], [], [], []
        self.t0 = time.time()

    def reset(self, forced_state=None):
        obs = self.e.reset(forced_state)
        self.shadow.sync()
        self.log.on_reset(_np(self.shadow.state()))
        return obs

    def act(self, n_steps, forced_actions=None, forced_resets=None):
        """engine.act(n_steps) (dqn.py:86-108), recorded."""
        e = self.e
        gs, S = e.global_step, e.slots
        if n_steps > S - 1:
            raise TraceError("DQNTrace.act: %d steps overwrite their own ring slots (slots = %d)" % (n_steps, S))
        self.shadow.sync()
        e.act(n_steps, forced_actions, forced_resets)
        slots = [(gs + s) % S for s in range(n_steps)]
        acts = torch.stack([e.actions[sl] for sl in slots])
        w = self.shadow.walk(acts, forced_resets)
        nxt = [(gs + s + 1) % S for s in range(n_steps)]
        _check_equal("observations", w["obs"], _np(torch.stack([e.observations[sl] for sl in nxt])))
        _check_equal("rewards", w["reward"], _np(torch.stack([e.rewards[sl] for sl in nxt])))
        _check_equal("terminated flags", (w["done"].astype(bool) & ~w["truncated"].astype(bool)).astype(np.uint8), _np(torch.stack([e.terminated[sl] for sl in nxt])))
        self.log.on_walk(w, _np(acts), gs, 1)

    def train_step(self, indices=None):
        """engine.train_step() (dqn.py:114-133), recorded."""
        e = self.e
        k = len(self.loss)
        before = _np(e.q.flat).copy() if (k in self.checkpoints) else None
        tgt = _np(e.target.flat).copy() if before is not None else None
        e.train_step(indices)
        pa = _np(e.q.flat)
        inds = _np(e.batch_inds)
        self.loss.append(float(e.loss.item())); self.psum.append(float(pa.astype(np.float64).sum()))
        self.inds_sum.append(int(inds.sum())); self.gs.append(e.global_step)
        if k < self.full_inds:
            self.inds.append(inds.astype(np.int32))
        if k < self.full_steps:
            self.grads.append(_np(e.grads).copy()); self.params.append(pa.copy())
        if before is not None:
            self.ck.append((k, before, tgt, _np(e.grads).copy(), inds.astype(np.int32), self.loss[-1]))

    def state_dict(self):
        e, hp = self.e, self.hp
        out = {"hparams": np.array([e.total_timesteps, e.learning_starts, e.start_e, e.end_e, e.exploration_fraction, hp["train_frequency"], e.batch_size, e.gamma,
                                    hp["learning_rate"], hp["target_network_frequency"], hp["seed"]], dtype=np.float64),
               "init_params": self.init_params, "final_params": _np(e.q.flat).copy(), "final_target_params": _np(e.target.flat).copy()}
        out.update(self.log.keys(np.int8))
        obs = np.concatenate(self.log.raw_obs).astype(np.float32).reshape(-1, 4) if self.log.raw_obs else np.zeros((0, 4), np.float32)
        out["obs_first"] = obs[:self.obs_first]
        nb = len(obs) // 1000
        out["obs_block_sums"] = obs[:nb * 1000].astype(np.float64).reshape(nb, 1000, 4).sum(axis=1)
        out.update({"loss_all": np.array(self.loss, np.float64), "psum_all": np.array(self.psum, np.float64), "inds_sum_all": np.array(self.inds_sum, np.int64),
                    "train_global_step": np.array(self.gs, np.int32),
                    "storage_terminated_sum": np.array([int(e.terminated.sum().item())]), "storage_rewards_sum": np.array([float(e.rewards.double().sum().item())]),
                    "ref_wall_seconds": np.array([time.time() - self.t0])})
        if self.inds:
            out["batch_inds_first"] = np.stack(self.inds)
        if self.grads:
            out["full_grads"] = np.stack(self.grads); out["full_params"] = np.stack(self.params)
        if self.ck:
            out.update({"ck_update": np.array([c[0] for c in self.ck], np.int32), "ck_params": np.stack([c[1] for c in self.ck]), "ck_target": np.stack([c[2] for c in self.ck]),
                        "ck_grads": np.stack([c[3] for c in self.ck]), "ck_inds": np.stack([c[4] for c in self.ck]), "ck_loss": np.array([c[5] for c in self.ck])})
        return out

    save = PPOTrace.save


# =====================================================================================================================================
class SACTrace:
    """Records a SACEngine run under the run-describing keys of tests/golden/sac_ref_trace.npz (oracle/capture_sac_trace.py:124-141).  The fixture's noise_chain /
    chain_* / ck_* keys hold torch's generator draws and gradient projections recorded FROM the reference as parity-test inputs; a device run has no counterpart
    of them (its draws are keyed: include/mi_rl.h, stream 5) and they are not written.
    hparams: {total_timesteps, policy_frequency, target_network_frequency, policy_lr, q_lr, seed}."""

    def __init__(self, engine, hparams, obs_first=8000):
        self.e, self.hp, self.obs_first = engine, dict(hparams), obs_first
        self.shadow = _Shadow(engine.env)
        self.log = _EnvLog()
        self.init = {"init_actor": _np(engine.actor.flat).copy(), "init_q": _np(engine.q_flat).copy(), "init_log_alpha": _np(engine.log_alpha).copy()}
        self.q, self.actor, self.alpha = [], [], []
        self.t0 = time.time()

    def reset(self, forced_state=None):
        obs = self.e.reset(forced_state)
        self.shadow.sync()
        self.log.on_reset(_np(self.shadow.state()))
        return obs

    def act(self, forced_actions=None, forced_eps=None, forced_resets=None):
        """engine.act() (sac.py:138-158), recorded."""
        e = self.e
        gs, S = e.global_step, e.slots
        self.shadow.sync()
        e.act(forced_actions, forced_eps, forced_resets)
        acts = e.actions[gs % S].reshape(1, e.N)
        w = self.shadow.walk(acts, None if forced_resets is None else forced_resets.reshape(1, e.N, 2))
        _check_equal("observations", w["obs"][0], _np(e.observations[(gs + 1) % S]))
        _check_equal("rewards", w["reward"][0], _np(e.rewards[(gs + 1) % S]))
        self.log.on_walk(w, _np(acts), gs, 1)

    def train_step(self, policy_frequency=2, target_network_frequency=1, indices=None):
        """engine.train_step() (sac.py:161-217) as its explicit pieces, recorded after each optimizer step."""
        e = self.e
        gs = e.global_step
        e.sample(indices)
        alpha = float(e.alpha.item())
        e.update_critic(polyak=gs % target_network_frequency == 0)
        ql = _np(e.q_losses)
        self.q.append([float(ql[0]), float(ql[1]), alpha, float(e.q_flat.double().sum().item())])
        if gs % policy_frequency == 0:
            for _ in range(policy_frequency):
                alpha = float(e.alpha.item())
                e.update_actor()
                self.actor.append([gs, float(e.actor_out[0].item()), alpha, float(e.actor.flat.double().sum().item())])
                la = float(e.log_alpha.item())
                e.update_alpha()
                ao = _np(e.alpha_out)     # reading it settles an owed step
                self.alpha.append([gs, float(ao[0]), la, float(ao[1]), float(e.log_alpha.item())])

    def state_dict(self):
        e, hp = self.e, self.hp
        out = {"hparams": np.array([hp["total_timesteps"], e.learning_starts, hp["policy_frequency"], e.batch_size, hp["target_network_frequency"], e.gamma, e.tau,
                                    hp["policy_lr"], hp["q_lr"], e.alpha_lr, hp["seed"], e.target_entropy], dtype=np.float64)}
        out.update(self.init)
        k = self.log.keys(np.float32)
        obs = np.concatenate(self.log.raw_obs).astype(np.float32).reshape(-1, 3) if self.log.raw_obs else np.zeros((0, 3), np.float32)
        out.update({"reset_states": k["reset_states"], "actions_all": k["actions_all"],
                    "rewards_all": np.concatenate(self.log.rewards).astype(np.float64) if self.log.rewards else np.zeros(0), "after_reset_all": k["after_reset_all"],
                    "obs_first": obs[:self.obs_first], "q_losses": np.array(self.q, np.float64).reshape(-1, 4), "actor_losses": np.array(self.actor, np.float64).reshape(-1, 4),
                    "alpha_steps": np.array(self.alpha, np.float64).reshape(-1, 5), "episode_global_step": k["episode_global_step"], "episode_return": k["episode_return"],
                    "final_actor": _np(e.actor.flat).copy(), "final_q": _np(e.q_flat).copy(), "final_log_alpha": _np(e.log_alpha).copy(),
                    "ref_wall_seconds": np.array([time.time() - self.t0])})
        return out

    save = PPOTrace.save
