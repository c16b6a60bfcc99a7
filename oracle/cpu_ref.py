"""ctypes binding of oracle/libcpu_ref.so (CPU restatement of the reference PPO path).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  Nothing under deep_rl_amd/ may import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.environ.get("MIRL_ORACLE_SO", os.path.join(_HERE, "libcpu_ref.so"))   # MIRL_ORACLE_SO: the sanitizer build (tests/test_oracle_sanitized.py)
NPARAMS = 9155
T_DEFAULT = 128


def build(force=False):
    src = os.path.join(_HERE, "cpu_ref.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", os.path.basename(_SO)], stdout=subprocess.DEVNULL)
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_SO)
        L = _lib
        L.ref_env_create.restype = C.c_void_p
        L.ref_env_create.argtypes = [C.c_int, C.c_uint64, C.c_uint64]
        L.ref_env_destroy.argtypes = [C.c_void_p]
        L.ref_env_state.restype = C.POINTER(C.c_double)
        L.ref_env_state.argtypes = [C.c_void_p]
        L.ref_env_reset.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.ref_env_step.argtypes = [C.c_void_p] + [C.c_void_p] * 8
        L.ref_cartpole_step.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.ref_actor.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        L.ref_critic.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        L.ref_categorical.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.ref_ppo_rollout.restype = C.c_int
        L.ref_ppo_rollout.argtypes = [C.c_void_p, C.c_void_p, C.c_int] + [C.c_void_p] * 11 + [C.c_int]
        L.ref_gae.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_float, C.c_void_p, C.c_void_p]
        L.ref_adv_stats.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.ref_ppo_minibatch.argtypes = [C.c_void_p] * 8 + [C.c_int, C.c_double, C.c_double, C.c_float, C.c_float, C.c_float,
                                                          C.c_double, C.c_void_p, C.c_void_p]
        L.ref_clip_grad_norm.restype = C.c_float
        L.ref_clip_grad_norm.argtypes = [C.c_void_p, C.c_int, C.c_float]
        L.ref_adam_step.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int64, C.c_double,
                                    C.c_double, C.c_double, C.c_double]
        L.ref_explained_var.restype = C.c_double
        L.ref_explained_var.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        L.ref_philox.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint32, C.c_void_p]
        L.ref_reset_noise.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, C.c_void_p]
        L.ref_action_uniform.restype = C.c_float
        L.ref_action_uniform.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64]
        L.ref_feistel.restype = C.c_uint32
        L.ref_feistel.argtypes = [C.c_uint32, C.c_uint32, C.c_uint64]
        L.ref_make_perm.argtypes = [C.c_uint32, C.c_uint64, C.c_void_p]
        L.ref_perm_key.restype = C.c_uint64
        L.ref_perm_key.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64]
        L.ref_ppo_ws_create.restype = C.c_void_p
        L.ref_ppo_ws_create.argtypes = [C.c_int, C.c_int]
        L.ref_ppo_ws_destroy.argtypes = [C.c_void_p]
        L.ref_ppo_update.restype = C.c_int
        L.ref_ppo_update.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                     C.c_float, C.c_float, C.c_double, C.c_float, C.c_float, C.c_float, C.c_float, C.c_void_p]
        L.ref_dqn_forward.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        L.ref_dqn_td_grads.argtypes = [C.c_void_p] * 7 + [C.c_int, C.c_int, C.c_int64, C.c_float, C.c_double, C.c_void_p, C.c_void_p]
        L.ref_per_sums.argtypes = [C.c_void_p, C.c_int64, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p]
        L.ref_per_sample.argtypes = [C.c_uint64, C.c_uint64, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_double, C.c_int, C.c_void_p]
        L.ref_per_weights.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_float, C.c_double, C.c_double, C.c_void_p]
        L.ref_per_td_grads.argtypes = [C.c_void_p] * 7 + [C.c_int, C.c_int, C.c_int64, C.c_float, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.ref_per_update_priorities.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        L.ref_dueling_forward.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        L.ref_dueling_td_grads.argtypes = [C.c_void_p] * 7 + [C.c_int, C.c_int, C.c_int64, C.c_float, C.c_double, C.c_void_p, C.c_void_p]
        L.ref_dqn_sample.argtypes = [C.c_uint64, C.c_uint64, C.c_int64, C.c_int, C.c_void_p]
        L.ref_dqn_epsilon.restype = C.c_double
        L.ref_dqn_epsilon.argtypes = [C.c_int64, C.c_double, C.c_double, C.c_double, C.c_int64]
        L.ref_dqn_explore_draw.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, C.c_void_p, C.c_void_p]
        L.ref_dqn_act_steps.restype = C.c_int
        L.ref_dqn_act_steps.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int64, C.c_int64, C.c_int64, C.c_double, C.c_double, C.c_double,
                                        C.c_int64] + [C.c_void_p] * 7
        L.ref_pend_create.restype = C.c_void_p
        L.ref_pend_create.argtypes = [C.c_int, C.c_uint64, C.c_uint64]
        L.ref_pend_destroy.argtypes = [C.c_void_p]
        L.ref_pend_state.restype = C.POINTER(C.c_double)
        L.ref_pend_state.argtypes = [C.c_void_p]
        L.ref_pend_reset.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.ref_pend_step.argtypes = [C.c_void_p] * 8
        L.ref_sac_actor_sample.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.ref_sac_q_forward.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        L.ref_sac_critic_grads.argtypes = [C.c_void_p] * 8 + [C.c_int, C.c_int, C.c_int64, C.c_void_p, C.c_float, C.c_float, C.c_double, C.c_void_p, C.c_void_p]
        L.ref_sac_actor_grads.argtypes = [C.c_void_p] * 4 + [C.c_int, C.c_void_p, C.c_float, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p]
        L.ref_sac_mean_logp.restype = C.c_float
        L.ref_sac_mean_logp.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        L.ref_polyak.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_float]
        L.ref_set_sincos_mode.argtypes = [C.c_int]
        L.ref_get_sincos_mode.restype = C.c_int
        L.ref_num_threads.restype = C.c_int
        L.ref_set_num_threads.argtypes = [C.c_int]
    return _lib


def set_sincos_mode(mode):
    """0 = libm (gym-faithful, pinned to the reference trace); 1 = fdlibm kernels (bit-identical to the HIP engine)."""
    lib().ref_set_sincos_mode({"libm": 0, "fdlibm": 1}.get(mode, mode))


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _c(a, dt):
    return None if a is None else np.ascontiguousarray(a, dtype=dt)


class Episode(C.Structure):
    _fields_ = [("env", C.c_int32), ("t", C.c_int32), ("ret", C.c_float), ("len", C.c_int32)]


class VecCartPole:
    """N CartPole-v1 envs with TimeLimit(500), episode statistics and ppo.py's auto-reset."""

    def __init__(self, n, seed=1, env_id_base=0):
        self.n = n
        self.h = lib().ref_env_create(n, seed, env_id_base)

    def __del__(self):
        if getattr(self, "h", None):
            lib().ref_env_destroy(self.h)
            self.h = None

    @property
    def state(self):
        return np.ctypeslib.as_array(lib().ref_env_state(self.h), shape=(self.n, 4))

    def reset(self, forced_state=None):
        obs = np.empty((self.n, 4), np.float32)
        fs = _c(forced_state, np.float64)
        lib().ref_env_reset(self.h, _p(obs), _p(fs))
        return obs

    def step(self, actions, forced_reset=None):
        n = self.n
        a = _c(actions, np.int64)
        fr = _c(forced_reset, np.float64)
        obs = np.empty((n, 4), np.float32); rew = np.empty(n, np.float32)
        done = np.empty(n, np.uint8); trunc = np.empty(n, np.uint8)
        fret = np.empty(n, np.float32); flen = np.empty(n, np.int32)
        lib().ref_env_step(self.h, _p(a), _p(fr), _p(obs), _p(rew), _p(done), _p(trunc), _p(fret), _p(flen))
        return obs, rew, done, trunc, fret, flen


def cartpole_step(state, action):
    s = _c(state, np.float64); out = np.empty(4, np.float64); term = C.c_int(0)
    lib().ref_cartpole_step(_p(s), int(action), _p(out), C.byref(term))
    return out, bool(term.value)


def actor(params, obs):
    p = _c(params, np.float32); o = _c(obs, np.float32).reshape(-1, 4)
    out = np.empty((o.shape[0], 2), np.float32)
    lib().ref_actor(_p(p), _p(o), o.shape[0], _p(out))
    return out


def critic(params, obs):
    p = _c(params, np.float32); o = _c(obs, np.float32).reshape(-1, 4)
    out = np.empty(o.shape[0], np.float32)
    lib().ref_critic(_p(p), _p(o), o.shape[0], _p(out))
    return out


def categorical(logits):
    l = _c(logits, np.float32).reshape(-1, 2); n = l.shape[0]
    nl = np.empty_like(l); p = np.empty_like(l); ent = np.empty(n, np.float32)
    lib().ref_categorical(_p(l), n, _p(nl), _p(p), _p(ent))
    return nl, p, ent


class Storage:
    """The six (T+1, N, ...) rollout tensors of ppo.py:93-98 plus advantages/returns."""

    def __init__(self, T, N):
        self.T, self.N = T, N
        self.observations = np.zeros((T + 1, N, 4), np.float32)
        self.values = np.zeros((T + 1, N), np.float32)
        self.actions = np.zeros((T + 1, N), np.int64)
        self.log_probs = np.zeros((T + 1, N), np.float32)
        self.rewards = np.zeros((T + 1, N), np.float32)
        self.dones = np.zeros((T + 1, N), np.float32)
        self.advantages = np.zeros((T + 1, N), np.float32)
        self.returns = np.zeros((T + 1, N), np.float32)


def rollout(env, params, st, obs_cur, forced_actions=None, forced_uniforms=None, forced_resets=None, max_ep=0):
    p = _c(params, np.float32)
    fa = _c(forced_actions, np.int64); fu = _c(forced_uniforms, np.float32); fr = _c(forced_resets, np.float64)
    eps = (Episode * max(max_ep, 1))()
    n = lib().ref_ppo_rollout(env.h, _p(p), st.T, _p(obs_cur), _p(st.observations), _p(st.values), _p(st.actions),
                              _p(st.log_probs), _p(st.rewards), _p(st.dones), _p(fa), _p(fu), _p(fr),
                              C.cast(eps, C.c_void_p), max_ep)
    return [(e.env, e.t, e.ret, e.len) for e in eps[: min(n, max_ep)]], n


def gae(st, gamma=0.99, lam=0.95):
    lib().ref_gae(_p(st.rewards), _p(st.dones), _p(st.values), st.T, st.N, gamma, lam, _p(st.advantages), _p(st.returns))


def adv_stats(adv_flat, idx):
    idx = _c(idx, np.int32); m = C.c_double(); s = C.c_double()
    lib().ref_adv_stats(_p(adv_flat), _p(idx), len(idx), C.byref(m), C.byref(s))
    return m.value, s.value


def minibatch(params, st, idx, adv_mean=None, adv_std=None, clip_coef=0.2, ent_coef=0.01, vf_coef=0.5, inv_count=None):
    p = _c(params, np.float32); idx = _c(idx, np.int32)
    if adv_mean is None:
        adv_mean, adv_std = adv_stats(st.advantages.reshape(-1), idx)
    if inv_count is None:
        inv_count = 1.0 / len(idx)
    grads = np.empty(NPARAMS, np.float32); terms = np.empty(4, np.float32)
    lib().ref_ppo_minibatch(_p(p), _p(st.observations), _p(st.actions), _p(st.log_probs), _p(st.advantages), _p(st.returns),
                            _p(st.values), _p(idx), len(idx), adv_mean, adv_std, clip_coef, ent_coef, vf_coef, inv_count,
                            _p(grads), _p(terms))
    return grads, terms


def clip_grad_norm(grads, max_norm=0.5):
    return float(lib().ref_clip_grad_norm(_p(grads), grads.size, max_norm))


def adam_step(params, grads, m, v, step, lr, beta1=0.9, beta2=0.999, eps=1e-5):
    lib().ref_adam_step(_p(params), _p(grads), _p(m), _p(v), params.size, step, lr, beta1, beta2, eps)


def explained_var(values, returns):
    v = _c(values, np.float32).reshape(-1); r = _c(returns, np.float32).reshape(-1)
    return lib().ref_explained_var(_p(v), _p(r), v.size)


def philox(seed, env, idx, stream):
    out = np.empty(4, np.uint32)
    lib().ref_philox(seed, env, idx, stream, _p(out))
    return out


def reset_noise(seed, env, episode):
    out = np.empty(4, np.float64)
    lib().ref_reset_noise(seed, env, episode, _p(out))
    return out


def action_uniform(seed, env, step):
    return float(lib().ref_action_uniform(seed, env, step))


def make_perm(n, key):
    out = np.empty(n, np.int32)
    lib().ref_make_perm(n, key, _p(out))
    return out


def perm_key(seed, update, epoch):
    return int(lib().ref_perm_key(seed, update, epoch))


class PPOBaseline:
    """Whole-update CPU loop in production-RNG mode (bench.py cpu_baseline, kind="port")."""

    def __init__(self, params, n_envs, T=128, seed=1, threads=None):
        L = lib()
        if threads:
            L.ref_set_num_threads(threads)
        self.threads = L.ref_num_threads()
        self.T, self.N = T, n_envs
        self.env = VecCartPole(n_envs, seed)
        self.ws = L.ref_ppo_ws_create(T, n_envs)
        self.params = np.array(params, np.float32, copy=True)
        self.obs = self.env.reset()
        self.update = 0
        self.terms = np.zeros(4, np.float32)

    def run_update(self, lr=2.5e-4):
        n = lib().ref_ppo_update(self.env.h, self.ws, _p(self.params), _p(self.obs), self.T, self.update, 4, 4, 0.99, 0.95,
                                 lr, 0.2, 0.01, 0.5, 0.5, _p(self.terms))
        self.update += 1
        return n

    def __del__(self):
        if getattr(self, "ws", None):
            lib().ref_ppo_ws_destroy(self.ws)
            self.ws = None


# ---------------------------------------------------------------------------------------------------------------------
# DQN (reference dqn.py)
DQN_NPARAMS = 10934


class ReplayStorage:
    """dqn.py:73-76 with an env axis: a [slots][N] time-major ring (slots = total_timesteps + 1 reproduces the reference)."""

    def __init__(self, slots, n_envs):
        self.slots, self.N = slots, n_envs
        self.observations = np.zeros((slots, n_envs, 4), np.float32)
        self.actions = np.zeros((slots, n_envs), np.int64)
        self.rewards = np.zeros((slots, n_envs), np.float32)
        self.terminated = np.zeros((slots, n_envs), np.uint8)


def dqn_forward(params, obs):
    p = _c(params, np.float32); o = _c(obs, np.float32).reshape(-1, 4)
    q = np.empty((o.shape[0], 2), np.float32)
    lib().ref_dqn_forward(_p(p), _p(o), o.shape[0], _p(q))
    return q


def dqn_td_grads(params, target_params, st, idx, gamma=0.99, inv_count=None):
    p = _c(params, np.float32); tp = _c(target_params, np.float32); idx = _c(idx, np.int64)
    grads = np.empty(DQN_NPARAMS, np.float32); loss = np.zeros(1, np.float32)
    lib().ref_dqn_td_grads(_p(p), _p(tp), _p(st.observations), _p(st.actions), _p(st.rewards), _p(st.terminated), _p(idx), len(idx),
                           st.N, st.slots, gamma, (1.0 / len(idx)) if inv_count is None else inv_count, _p(grads), _p(loss))
    return grads, float(loss[0])


DUELING_NPARAMS = 11019
PER_CHUNK = 64


def per_sums(prio, n, alpha=0.6):
    """-> (s0, s1, total, total_alpha): the sampler's three-level prefix structure over prio[:n] (flat [slot][env])."""
    p = _c(prio, np.float32).reshape(-1)
    n0 = (n + PER_CHUNK - 1) // PER_CHUNK; n1 = (n0 + PER_CHUNK - 1) // PER_CHUNK
    s0 = np.zeros(n0, np.float64); s1 = np.zeros(n1, np.float64); tot = np.zeros(2, np.float64)
    lib().ref_per_sums(_p(p), n, alpha, _p(s0), _p(s1), _p(tot))
    return s0, s1, float(tot[0]), float(tot[1])


def per_sample(seed, update, prio, n, s0, s1, total, batch):
    p = _c(prio, np.float32).reshape(-1)
    idx = np.empty(batch, np.int64)
    lib().ref_per_sample(seed, update, _p(p), n, _p(s0), _p(s1), total, batch, _p(idx))
    return idx


def per_weights(prio, idx, alpha, beta, total_alpha, count):
    p = _c(prio, np.float32).reshape(-1); idx = _c(idx, np.int64)
    w = np.empty(len(idx), np.float32)
    lib().ref_per_weights(_p(p), _p(idx), len(idx), alpha, beta, total_alpha, float(count), _p(w))
    return w


def per_td_grads(params, target_params, st, idx, weights, gamma=0.99, inv_count=None):
    p = _c(params, np.float32); tp = _c(target_params, np.float32); idx = _c(idx, np.int64); w = _c(weights, np.float32)
    grads = np.empty(DQN_NPARAMS, np.float32); loss = np.zeros(1, np.float32); td = np.empty(len(idx), np.float32)
    lib().ref_per_td_grads(_p(p), _p(tp), _p(st.observations), _p(st.actions), _p(st.rewards), _p(st.terminated), _p(idx), len(idx),
                           st.N, st.slots, gamma, (1.0 / len(idx)) if inv_count is None else inv_count, _p(w), _p(td), _p(grads), _p(loss))
    return grads, float(loss[0]), td


def per_update_priorities(prio, idx, td_abs, max_priority):
    """in place on prio (flat view); returns the new max_priority"""
    idx = _c(idx, np.int64); td = _c(td_abs, np.float32); mp = np.array([max_priority], np.float32)
    flatp = prio.reshape(-1)
    assert flatp.base is not None or flatp is prio
    lib().ref_per_update_priorities(_p(flatp), _p(idx), _p(td), len(idx), _p(mp))
    return float(mp[0])


def dueling_forward(params, obs):
    """QNetwork.forward of dueling_dqn.py:36-40."""
    p = _c(params, np.float32); o = _c(obs, np.float32).reshape(-1, 4)
    q = np.empty((o.shape[0], 2), np.float32)
    lib().ref_dueling_forward(_p(p), _p(o), o.shape[0], _p(q))
    return q


def dueling_td_grads(params, target_params, st, idx, gamma=0.99, inv_count=None):
    p = _c(params, np.float32); tp = _c(target_params, np.float32); idx = _c(idx, np.int64)
    grads = np.empty(DUELING_NPARAMS, np.float32); loss = np.zeros(1, np.float32)
    lib().ref_dueling_td_grads(_p(p), _p(tp), _p(st.observations), _p(st.actions), _p(st.rewards), _p(st.terminated), _p(idx), len(idx),
                               st.N, st.slots, gamma, (1.0 / len(idx)) if inv_count is None else inv_count, _p(grads), _p(loss))
    return grads, float(loss[0])


def dqn_epsilon(gs, start_e=1.0, end_e=0.05, exploration_fraction=0.5, total_timesteps=100_000):
    return lib().ref_dqn_epsilon(gs, start_e, end_e, exploration_fraction, total_timesteps)


def dqn_explore_draw(seed, env, step):
    u = C.c_float(); a = C.c_int()
    lib().ref_dqn_explore_draw(seed, env, step, C.byref(u), C.byref(a))
    return u.value, a.value


def dqn_act_steps(env, params, st, obs_cur, n_steps, global_step, learning_starts=10_000, start_e=1.0, end_e=0.05,
                  exploration_fraction=0.5, total_timesteps=100_000, forced_actions=None, forced_resets=None):
    p = _c(params, np.float32); fa = _c(forced_actions, np.int64); fr = _c(forced_resets, np.float64)
    return lib().ref_dqn_act_steps(env.h, _p(p), n_steps, global_step, st.slots, learning_starts, start_e, end_e, exploration_fraction,
                                   total_timesteps, _p(obs_cur), _p(st.observations), _p(st.actions), _p(st.rewards), _p(st.terminated),
                                   _p(fa), _p(fr))


def dqn_act_steps_log(env, params, st, obs_cur, n_steps, global_step, learning_starts=10_000, start_e=1.0, end_e=0.05,
                      exploration_fraction=0.5, total_timesteps=100_000, forced_actions=None, forced_resets=None, max_ep=0):
    """dqn_act_steps that also returns the finished episodes: -> ([(env, step_in_call, return, length)], count)."""
    p = _c(params, np.float32); fa = _c(forced_actions, np.int64); fr = _c(forced_resets, np.float64)
    eps = (Episode * max(max_ep, 1))()
    f = lib().ref_dqn_act_steps_log
    f.restype = C.c_int
    f.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int64, C.c_int64, C.c_int64, C.c_double, C.c_double, C.c_double, C.c_int64] + [C.c_void_p] * 8 + [C.c_int]
    n = f(env.h, _p(p), n_steps, global_step, st.slots, learning_starts, start_e, end_e, exploration_fraction, total_timesteps, _p(obs_cur), _p(st.observations),
          _p(st.actions), _p(st.rewards), _p(st.terminated), _p(fa), _p(fr), C.cast(eps, C.c_void_p), max_ep)
    return [(e.env, e.t, e.ret, e.len) for e in eps[: min(n, max_ep)]], n


def dqn_sample(seed, update_index, upper_flat, batch):
    idx = np.empty(batch, np.int64)
    lib().ref_dqn_sample(seed, update_index, upper_flat, batch, _p(idx))
    return idx


# ---------------------------------------------------------------------------------------------------------------------
# SAC (reference sac.py on Pendulum-v1)
SQ_NPARAMS = 67329
AC_NPARAMS = 67330


class VecPendulum:
    """N Pendulum-v1 envs with TimeLimit(200), episode statistics and sac.py's auto-reset (sac.py:142-144)."""

    def __init__(self, n, seed=1, env_id_base=0):
        self.n = n
        self.h = lib().ref_pend_create(n, seed, env_id_base)

    def __del__(self):
        if getattr(self, "h", None):
            lib().ref_pend_destroy(self.h)
            self.h = None

    @property
    def state(self):
        return np.ctypeslib.as_array(lib().ref_pend_state(self.h), shape=(self.n, 2))

    def reset(self, forced_state=None):
        obs = np.empty((self.n, 3), np.float32)
        lib().ref_pend_reset(self.h, _p(obs), _p(_c(forced_state, np.float64)))
        return obs

    def step(self, actions, forced_reset=None):
        n = self.n
        a = _c(actions, np.float32).reshape(n)
        obs = np.empty((n, 3), np.float32); rew = np.empty(n, np.float32); done = np.empty(n, np.uint8)
        fret = np.empty(n, np.float32); flen = np.empty(n, np.int32)
        lib().ref_pend_step(self.h, _p(a), _p(_c(forced_reset, np.float64)), _p(obs), _p(rew), _p(done), _p(fret), _p(flen))
        return obs, rew, done, fret, flen


class SacStorage:
    """sac.py:126-129 with an env axis: [slots][N] ring; actions are float32 (one action dim)."""

    def __init__(self, slots, n_envs):
        self.slots, self.N = slots, n_envs
        self.observations = np.zeros((slots, n_envs, 3), np.float32)
        self.actions = np.zeros((slots, n_envs), np.float32)
        self.rewards = np.zeros((slots, n_envs), np.float32)
        self.terminated = np.zeros((slots, n_envs), np.uint8)


def sac_actor_sample(actor, obs, eps):
    o = _c(obs, np.float32).reshape(-1, 3); e = _c(eps, np.float32).reshape(-1); n = o.shape[0]
    a = np.empty(n, np.float32); lp = np.empty(n, np.float32)
    lib().ref_sac_actor_sample(_p(_c(actor, np.float32)), _p(o), _p(e), n, _p(a), _p(lp))
    return a, lp


def sac_q_forward(q, obs, act):
    o = _c(obs, np.float32).reshape(-1, 3); a = _c(act, np.float32).reshape(-1); out = np.empty(o.shape[0], np.float32)
    lib().ref_sac_q_forward(_p(_c(q, np.float32)), _p(o), _p(a), o.shape[0], _p(out))
    return out


def sac_critic_grads(qparams, qtarget, actor, st, idx, eps, alpha, gamma=0.99, inv_count=None):
    idx = _c(idx, np.int64); e = _c(eps, np.float32)
    grads = np.empty(2 * SQ_NPARAMS, np.float32); losses = np.zeros(2, np.float32)
    lib().ref_sac_critic_grads(_p(_c(qparams, np.float32)), _p(_c(qtarget, np.float32)), _p(_c(actor, np.float32)), _p(st.observations),
                               _p(st.actions), _p(st.rewards), _p(st.terminated), _p(idx), len(idx), st.N, st.slots, _p(e), alpha, gamma,
                               (1.0 / len(idx)) if inv_count is None else inv_count, _p(grads), _p(losses))
    return grads, losses


def sac_actor_grads(actor, qparams, st, idx, eps, alpha, inv_count=None):
    idx = _c(idx, np.int64); e = _c(eps, np.float32)
    grads = np.empty(AC_NPARAMS, np.float32); loss = np.zeros(1, np.float32); mlp = np.zeros(1, np.float32)
    lib().ref_sac_actor_grads(_p(_c(actor, np.float32)), _p(_c(qparams, np.float32)), _p(st.observations), _p(idx), len(idx), _p(e), alpha,
                              (1.0 / len(idx)) if inv_count is None else inv_count, _p(grads), _p(loss), _p(mlp))
    return grads, float(loss[0]), float(mlp[0])


def sac_mean_logp(actor, st, idx, eps):
    idx = _c(idx, np.int64)
    return float(lib().ref_sac_mean_logp(_p(_c(actor, np.float32)), _p(st.observations), _p(idx), len(idx), _p(_c(eps, np.float32))))


def polyak(target, param, tau=0.005):
    lib().ref_polyak(_p(target), _p(_c(param, np.float32)), target.size, tau)
