"""Checkpoint / resume of an engine (SURVEY.md §8f rank 4 — the reference has no torch.save anywhere).

Wire format = the fixtures' format: one ``.npz`` of named arrays.  A checkpoint holds everything the next launch reads: the env blob
(mi_env_export_state: fp64 state, TimeLimit / episode counters and the RNG counters), the carried-over observation, flat parameters,
optimizer moments and step counts, the engines' counters and — for the off-policy engines — the replay ring (optional: 29 B per
transition).  All randomness is counter-based (keys = seed, global env id, counters), so a run resumed from a checkpoint continues
BIT FOR BIT (tests/test_gpu_checkpoint.py).
"""
import numpy as np
import torch

from . import _native as N
from .dqn_engine import DQNEngine, DuelingDQNEngine, PERDQNEngine
from .engine import PPOEngine
from .sac_engine import SACEngine

FORMAT = 1


def _env_blob(env):
    buf = torch.empty(N.lib().mi_env_state_bytes(env.handle), dtype=torch.uint8, device=env.device)
    N.check(N.lib().mi_env_export_state(env.handle, N.ptr(buf), N.stream_ptr(env.device)), "mi_env_export_state")
    return buf.cpu().numpy()


def _env_restore(env, blob):
    buf = torch.from_numpy(np.ascontiguousarray(blob)).to(env.device)
    if buf.numel() != N.lib().mi_env_state_bytes(env.handle):
        raise N.MiError("checkpoint: env blob of %d bytes does not fit this env (%d)" % (buf.numel(), N.lib().mi_env_state_bytes(env.handle)))
    N.check(N.lib().mi_env_import_state(env.handle, N.ptr(buf), N.stream_ptr(env.device)), "mi_env_import_state")
    torch.cuda.current_stream(env.device).synchronize()


def _opt_state(o, prefix):
    return {prefix + "exp_avg": o.exp_avg, prefix + "exp_avg_sq": o.exp_avg_sq, prefix + "step_count": np.int64(o.step_count), prefix + "lr": np.float64(o.param_groups[0]["lr"])}


def _opt_restore(o, z, prefix):
    o.exp_avg.copy_(torch.from_numpy(z[prefix + "exp_avg"])); o.exp_avg_sq.copy_(torch.from_numpy(z[prefix + "exp_avg_sq"]))
    o.step_count = int(z[prefix + "step_count"]); o.param_groups[0]["lr"] = float(z[prefix + "lr"])


def state_dict(engine, include_replay=True):
    """-> {name: array}: everything needed to continue `engine` exactly."""
    env = engine.env
    st = {"format": np.int64(FORMAT), "kind": type(engine).__name__, "num_envs": np.int64(env.num_envs), "seed": np.int64(env._seed),
          "env_id_base": np.int64(env.env_id_base), "env_blob": _env_blob(env), "observation": engine.observation}
    if isinstance(engine, PPOEngine):
        st.update(params=engine.agent.flat, update_index=np.int64(engine.update_index), **_opt_state(engine.optimizer, "opt_"))
    elif isinstance(engine, DQNEngine):
        st.update(params=engine.q.flat, target=engine.target.flat, global_step=np.int64(engine.global_step), update_index=np.int64(engine.update_index),
                  **_opt_state(engine.optimizer, "opt_"))
        if isinstance(engine, PERDQNEngine):
            st.update(max_priority=engine.max_priority)
            if include_replay:
                st.update(priorities=engine.priorities)
        if include_replay:
            st.update(observations=engine.observations, actions=engine.actions, rewards=engine.rewards, terminated=engine.terminated)
    elif isinstance(engine, SACEngine):
        st.update(actor=engine.actor.flat, q=engine.q_flat, q_target=engine.qt_flat, log_alpha=engine.log_alpha, alpha=engine.alpha,
                  alpha_m=engine._alpha_m, alpha_v=engine._alpha_v, alpha_steps=np.int64(engine.alpha_steps), global_step=np.int64(engine.global_step),
                  update_index=np.int64(engine.update_index), actor_updates=np.int64(engine.actor_updates),
                  **_opt_state(engine.actor_optimizer, "aopt_"), **_opt_state(engine.q_optimizer, "qopt_"))
        if include_replay:
            st.update(observations=engine.observations, actions=engine.actions, rewards=engine.rewards, terminated=engine.terminated)
    else:
        raise N.MiError("checkpoint: unknown engine type %r" % type(engine).__name__)
    return {k: (v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in st.items()}


def _npz_path(path):
    """np.savez appends '.npz' to a suffix-less name; load() must open the same file."""
    path = str(path)
    return path if path.endswith(".npz") else path + ".npz"


def save(path, engine, include_replay=True):
    """Write `engine` to ``path`` ('.npz' is appended when missing, as np.savez does).  -> the path written."""
    if engine.observation is None:
        raise N.MiError("checkpoint: the engine was never reset (no carried-over observation to save)")
    torch.cuda.current_stream(engine.device).synchronize()
    path = _npz_path(path)
    with open(path, "wb") as f:
        np.savez(f, **state_dict(engine, include_replay))
    return path


def _check_shape(z, name, want):
    if tuple(z[name].shape) != tuple(want):
        raise N.MiError("checkpoint: %s has shape %s, this engine needs %s" % (name, tuple(z[name].shape), tuple(want)))


def load(path, engine):
    """Restore `engine` (built with the same sizes, seed and env_id_base) from a checkpoint written by save()."""
    with np.load(_npz_path(path), allow_pickle=False) as zf:
        z = {k: zf[k] for k in zf.files}
    env = engine.env
    if "format" not in z or int(z["format"]) != FORMAT:
        raise N.MiError("checkpoint: format %s, this build reads format %d" % (z.get("format"), FORMAT))
    if str(z["kind"]) != type(engine).__name__ or int(z["num_envs"]) != env.num_envs or int(z["seed"]) != env._seed or int(z["env_id_base"]) != env.env_id_base:
        raise N.MiError("checkpoint: written for %s with (num_envs, seed, env_id_base) = (%d, %d, %d); this engine is %s (%d, %d, %d)" % (
            z["kind"], z["num_envs"], z["seed"], z["env_id_base"], type(engine).__name__, env.num_envs, env._seed, env.env_id_base))
    dev = engine.device
    t = lambda name: torch.from_numpy(z[name]).to(dev)   # noqa: E731
    _env_restore(env, z["env_blob"])
    if isinstance(engine, PPOEngine):
        _check_shape(z, "params", engine.agent.flat.shape)
    elif isinstance(engine, DQNEngine):
        _check_shape(z, "params", engine.q.flat.shape); _check_shape(z, "target", engine.target.flat.shape)
    else:
        _check_shape(z, "actor", engine.actor.flat.shape); _check_shape(z, "q", engine.q_flat.shape); _check_shape(z, "q_target", engine.qt_flat.shape)
    for name in ("observations", "actions", "rewards", "terminated", "priorities"):
        if name in z and hasattr(engine, name):
            _check_shape(z, name, getattr(engine, name).shape)
    engine.observation = t("observation")
    if isinstance(engine, PPOEngine):
        engine.agent.flat.copy_(t("params")); engine.update_index = int(z["update_index"]); _opt_restore(engine.optimizer, z, "opt_")
    elif isinstance(engine, DQNEngine):
        engine.q.flat.copy_(t("params")); engine.target.flat.copy_(t("target"))
        if isinstance(engine, DuelingDQNEngine):
            engine.q.repack(); engine.target.repack()
        engine.global_step, engine.update_index = int(z["global_step"]), int(z["update_index"])
        _opt_restore(engine.optimizer, z, "opt_")
        if isinstance(engine, PERDQNEngine):
            engine.max_priority.copy_(t("max_priority"))
            if "priorities" in z:
                engine.priorities.copy_(t("priorities"))
                engine.refresh_sums()   # the sampler's chunk sums follow the priorities
        for name in ("observations", "actions", "rewards", "terminated"):
            if name in z:
                getattr(engine, name).copy_(t(name))
    else:
        engine.drop_owed()   # a pending alpha debt / stash belongs to the state being replaced (the in-launch epoch counter is engine-private and keeps growing)
        engine.actor.flat.copy_(t("actor")); engine.q_flat.copy_(t("q")); engine.qt_flat.copy_(t("q_target"))
        engine.log_alpha.copy_(t("log_alpha")); engine.alpha.copy_(t("alpha")); engine._alpha_m.copy_(t("alpha_m")); engine._alpha_v.copy_(t("alpha_v"))
        engine.alpha_steps, engine.global_step = int(z["alpha_steps"]), int(z["global_step"])
        engine.update_index, engine.actor_updates = int(z["update_index"]), int(z["actor_updates"])
        _opt_restore(engine.actor_optimizer, z, "aopt_"); _opt_restore(engine.q_optimizer, z, "qopt_")
        for name in ("observations", "actions", "rewards", "terminated"):
            if name in z:
                getattr(engine, name).copy_(t(name))
    return engine
