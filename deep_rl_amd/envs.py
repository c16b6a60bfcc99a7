"""gym-0.21-style env surface over the batched HIP stepper (mi_env_* in include/mi_rl.h).

Mirrors what the reference scripts touch (reference ppo.py:10-22,79-84,101,127-130):
``make(id)``, ``env.seed(int)``, ``env.reset() -> obs``, ``env.step(action) -> (obs, reward, done, info)``,
``env.observation_space.shape``, ``env.action_space.n/.shape``, ``env.spec.max_episode_steps``,
``info["episode"]["r"]``, ``info["TimeLimit.truncated"]``, ``env.close()`` — with an env axis N in front and
tensors living on the GPU.  The reference's TorchWrapper (tensor<->numpy hops) disappears: actions and
observations never leave HBM.  As in the reference loop (ppo.py:128-129) a finished env is reset at once and
``step`` returns the RESET observation for it.
"""
import ctypes as C
from types import SimpleNamespace

import torch

from . import _native as N


class Discrete:
    def __init__(self, n):
        self.n = n
        self.shape = ()
        self.dtype = torch.int64
        self._gen = None

    def seed(self, seed=None):
        """env.action_space.seed(seed) (per.py:67).  Host-side sampler only; the engines' random actions are keyed in-kernel."""
        self._gen = torch.Generator().manual_seed(0 if seed is None else int(seed))
        return [seed]

    def sample(self):
        """env.action_space.sample() (per.py:96)."""
        return int(torch.randint(self.n, (), generator=self._gen))


class Box:
    def __init__(self, low, high, shape):
        self.low, self.high, self.shape = low, high, tuple(shape)
        self.dtype = torch.float32
        self._gen = None

    def seed(self, seed=None):
        """env.action_space.seed(seed) (sac.py:104).  Host-side sampler only; the engine's warm-up actions are keyed in-kernel."""
        self._gen = torch.Generator().manual_seed(0 if seed is None else int(seed))
        return [seed]

    def sample(self):
        """env.action_space.sample() (sac.py:139): uniform in [low, high)."""
        import numpy as np
        lo, hi = np.asarray(self.low, np.float32), np.asarray(self.high, np.float32)
        u = torch.rand(self.shape, generator=self._gen).numpy()
        return (lo + (hi - lo) * u).astype(np.float32)


class CartPoleVecEnv:
    """N independent CartPole-v1 envs (TimeLimit 500 + RecordEpisodeStatistics) stepped by one kernel."""

    metadata = {}

    def __init__(self, num_envs=1, device="cuda", seed=0, env_id_base=0):
        self.num_envs = int(num_envs)
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise N.MiError("CartPoleVecEnv runs on an MI355X only (device=%r)" % (device,))
        self.env_id_base = int(env_id_base)
        self.spec = SimpleNamespace(id="CartPole-v1", max_episode_steps=500, reward_threshold=475.0)
        thr = 12 * 2 * 3.141592653589793 / 360
        fmax = torch.finfo(torch.float32).max
        self.observation_space = Box([-4.8, -fmax, -2 * thr, -fmax], [4.8, fmax, 2 * thr, fmax], (4,))
        self.action_space = Discrete(2)
        self.single_observation_space, self.single_action_space = self.observation_space, self.action_space
        self._h = None
        self._seed = int(seed)
        self._create()

    # -- handle management ---------------------------------------------------------------------
    def _create(self):
        self.close()
        h = C.c_void_p()
        with torch.cuda.device(self.device):
            N.check(N.lib().mi_env_create(0, self.num_envs, self._seed, self.env_id_base, C.byref(h)), "mi_env_create")
        self._h = h
        n, dev = self.num_envs, self.device
        self._obs = torch.empty((n, 4), dtype=torch.float32, device=dev)
        self._reward = torch.empty(n, dtype=torch.float32, device=dev)
        self._done = torch.empty(n, dtype=torch.uint8, device=dev)
        self._trunc = torch.empty(n, dtype=torch.uint8, device=dev)
        self._fret = torch.empty(n, dtype=torch.float32, device=dev)
        self._flen = torch.empty(n, dtype=torch.int32, device=dev)

    @property
    def handle(self):
        if self._h is None:
            raise N.MiError("env is closed")
        return self._h

    def seed(self, seed=None):
        """env.seed(seed) (ppo.py:84): re-keys the counter-based RNG; call before reset()."""
        self._seed = 0 if seed is None else int(seed)
        self._create()
        return [self._seed]

    def close(self):
        if getattr(self, "_h", None) is not None:
            N.lib().mi_env_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- gym protocol ---------------------------------------------------------------------------
    def reset(self, forced_state=None):
        """-> obs (N,4) f32 on device.  forced_state (N,4) f64 replaces the keyed reset noise (parity mode)."""
        fs = None if forced_state is None else forced_state.to(self.device, torch.float64).contiguous()
        N.check(N.lib().mi_env_reset(self.handle, N.ptr(self._obs), N.ptr(fs), N.stream_ptr(self.device)), "mi_env_reset")
        return self._obs.clone()

    def step(self, action, forced_reset=None):
        """action (N,) int64 on device -> (obs, reward, done, info); tensors stay on the GPU."""
        a = action.to(self.device, torch.int64).reshape(self.num_envs).contiguous()
        fr = None if forced_reset is None else forced_reset.to(self.device, torch.float64).contiguous()
        N.check(N.lib().mi_env_step(self.handle, N.ptr(a), N.ptr(fr), N.ptr(self._obs), N.ptr(self._reward), N.ptr(self._done),
                                   N.ptr(self._trunc), N.ptr(self._fret), N.ptr(self._flen), N.stream_ptr(self.device)),
                "mi_env_step")
        done = self._done.bool()
        info = {"TimeLimit.truncated": self._trunc.bool(), "episode": {"r": self._fret.clone(), "l": self._flen.clone()},
                "_episode": done}
        return self._obs.clone(), self._reward.clone(), done, info

    def get_state(self):
        """float64 state (N,4) and TimeLimit counters (N,) — test/debug helper."""
        st = torch.empty((self.num_envs, 4), dtype=torch.float64, device=self.device)
        el = torch.empty(self.num_envs, dtype=torch.int32, device=self.device)
        N.check(N.lib().mi_env_get_state(self.handle, N.ptr(st), N.ptr(el), N.stream_ptr(self.device)), "mi_env_get_state")
        return st, el


class PendulumVecEnv(CartPoleVecEnv):
    """N independent Pendulum-v1 envs (TimeLimit 200 + RecordEpisodeStatistics) — the continuous-control env the SAC path of
    reference sac.py:80,96 is re-targeted to (pybullet's Hopper is not reproducible here)."""

    def __init__(self, num_envs=1, device="cuda", seed=0, env_id_base=0):
        import numpy as np
        self.num_envs = int(num_envs)
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise N.MiError("PendulumVecEnv runs on an MI355X only (device=%r)" % (device,))
        self.env_id_base = int(env_id_base)
        self.spec = SimpleNamespace(id="Pendulum-v1", max_episode_steps=200, reward_threshold=None)
        self.observation_space = Box(np.array([-1.0, -1.0, -8.0], np.float32), np.array([1.0, 1.0, 8.0], np.float32), (3,))
        self.action_space = Box(np.array([-2.0], np.float32), np.array([2.0], np.float32), (1,))
        self.single_observation_space, self.single_action_space = self.observation_space, self.action_space
        self._h = None
        self._seed = int(seed)
        self._create()

    def _create(self):
        self.close()
        h = C.c_void_p()
        with torch.cuda.device(self.device):
            N.check(N.lib().mi_env_create(1, self.num_envs, self._seed, self.env_id_base, C.byref(h)), "mi_env_create")
        self._h = h
        n, dev = self.num_envs, self.device
        self._obs = torch.empty((n, 3), dtype=torch.float32, device=dev)
        self._reward = torch.empty(n, dtype=torch.float32, device=dev)
        self._done = torch.empty(n, dtype=torch.uint8, device=dev)
        self._trunc = torch.empty(n, dtype=torch.uint8, device=dev)
        self._fret = torch.empty(n, dtype=torch.float32, device=dev)
        self._flen = torch.empty(n, dtype=torch.int32, device=dev)

    def reset(self, forced_state=None):
        """-> obs (N,3) f32 = (cos th, sin th, th_dot).  forced_state (N,2) f64 = (th, th_dot) replaces the keyed reset noise."""
        return super().reset(forced_state)

    def step(self, action, forced_reset=None):
        """action (N,) or (N,1) f32 on device (clipped to +-2 by the env, pendulum.py) -> (obs, reward, done, info)."""
        a = action.to(self.device, torch.float32).reshape(self.num_envs).contiguous()
        fr = None if forced_reset is None else forced_reset.to(self.device, torch.float64).contiguous()
        N.check(N.lib().mi_env_step_cont(self.handle, N.ptr(a), N.ptr(fr), N.ptr(self._obs), N.ptr(self._reward), N.ptr(self._done),
                                        N.ptr(self._trunc), N.ptr(self._fret), N.ptr(self._flen), N.stream_ptr(self.device)),
                "mi_env_step_cont")
        done = self._done.bool()
        info = {"TimeLimit.truncated": self._trunc.bool(), "episode": {"r": self._fret.clone(), "l": self._flen.clone()},
                "_episode": done}
        return self._obs.clone(), self._reward.clone(), done, info

    def get_state(self):
        st = torch.empty((self.num_envs, 2), dtype=torch.float64, device=self.device)
        el = torch.empty(self.num_envs, dtype=torch.int32, device=self.device)
        N.check(N.lib().mi_env_get_state(self.handle, N.ptr(st), N.ptr(el), N.stream_ptr(self.device)), "mi_env_get_state")
        return st, el


_REGISTRY = {"CartPole-v1": CartPoleVecEnv, "Pendulum-v1": PendulumVecEnv}


def make(env_id, num_envs=1, device="cuda", seed=0, env_id_base=0):
    """gym.make(env_id) + RecordEpisodeStatistics (ppo.py:79) for `num_envs` envs on `device`."""
    if env_id not in _REGISTRY:
        raise KeyError("deep_rl_amd.make: unknown env id %r (have %s)" % (env_id, sorted(_REGISTRY)))
    return _REGISTRY[env_id](num_envs=num_envs, device=device, seed=seed, env_id_base=env_id_base)
