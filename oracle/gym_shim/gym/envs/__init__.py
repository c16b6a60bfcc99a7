"""Env registry + the two classic-control tasks the north-star path uses.

CartPole-v1 / Pendulum-v1 dynamics restate the published gym 0.21 algorithm from memory
(UNVERIFIED: gym is not installable here).  Every transition can be mirrored to a trace sink so
the capture scripts under ``oracle/`` can log (state_f64, action, obs_f32, reward, done, reset noise).
"""
import math

import numpy as np

from gym import spaces
from gym.core import Env
from gym.utils import seeding
from gym.wrappers import TimeLimit

_TRACE_SINK = None


def register_trace_sink(sink):
    """sink(event: str, payload: dict) is called on every reset / step of the raw env."""
    global _TRACE_SINK
    _TRACE_SINK = sink


def _emit(event, **payload):
    if _TRACE_SINK is not None:
        _TRACE_SINK(event, payload)


class EnvSpec:
    def __init__(self, id, max_episode_steps=None, reward_threshold=None):
        self.id = id
        self.max_episode_steps = max_episode_steps
        self.reward_threshold = reward_threshold


class CartPoleEnv(Env):
    """Euler-integrated cart-pole, float64 state, float32 observations."""

    def __init__(self):
        self.gravity = 9.8
        self.masscart = 1.0
        self.masspole = 0.1
        self.total_mass = self.masspole + self.masscart
        self.length = 0.5  # half the pole's length
        self.polemass_length = self.masspole * self.length
        self.force_mag = 10.0
        self.tau = 0.02
        self.kinematics_integrator = "euler"
        self.theta_threshold_radians = 12 * 2 * math.pi / 360
        self.x_threshold = 2.4
        high = np.array(
            [self.x_threshold * 2, np.finfo(np.float32).max, self.theta_threshold_radians * 2, np.finfo(np.float32).max],
            dtype=np.float32,
        )
        self.action_space = spaces.Discrete(2)
        self.observation_space = spaces.Box(-high, high, dtype=np.float32)
        self.seed()
        self.state = None
        self.steps_beyond_done = None

    def seed(self, seed=None):
        self.np_random, seed = seeding.np_random(seed)
        return [seed]

    def step(self, action):
        action = int(action)
        assert action in (0, 1)
        x, x_dot, theta, theta_dot = self.state
        state_before = (float(x), float(x_dot), float(theta), float(theta_dot))
        force = self.force_mag if action == 1 else -self.force_mag
        costheta = math.cos(theta)
        sintheta = math.sin(theta)
        temp = (force + self.polemass_length * theta_dot ** 2 * sintheta) / self.total_mass
        thetaacc = (self.gravity * sintheta - costheta * temp) / (
            self.length * (4.0 / 3.0 - self.masspole * costheta ** 2 / self.total_mass)
        )
        xacc = temp - self.polemass_length * thetaacc * costheta / self.total_mass
        x = x + self.tau * x_dot
        x_dot = x_dot + self.tau * xacc
        theta = theta + self.tau * theta_dot
        theta_dot = theta_dot + self.tau * thetaacc
        self.state = (x, x_dot, theta, theta_dot)
        done = bool(
            x < -self.x_threshold
            or x > self.x_threshold
            or theta < -self.theta_threshold_radians
            or theta > self.theta_threshold_radians
        )
        if not done:
            reward = 1.0
        elif self.steps_beyond_done is None:
            self.steps_beyond_done = 0
            reward = 1.0
        else:
            self.steps_beyond_done += 1
            reward = 0.0
        obs = np.array(self.state, dtype=np.float32)
        _emit("step", state=state_before, action=action, next_state=tuple(float(v) for v in self.state),
              obs=obs.copy(), reward=reward, terminated=done)
        return obs, reward, done, {}

    def reset(self):
        self.state = self.np_random.uniform(low=-0.05, high=0.05, size=(4,))
        self.steps_beyond_done = None
        _emit("reset", state=tuple(float(v) for v in self.state))
        return np.array(self.state, dtype=np.float32)


def _angle_normalize(x):
    return ((x + np.pi) % (2 * np.pi)) - np.pi


class PendulumEnv(Env):
    """Pendulum-v1 (g=10), float64 state, float32 observations [cos, sin, thdot]."""

    def __init__(self, g=10.0):
        self.max_speed = 8
        self.max_torque = 2.0
        self.dt = 0.05
        self.g = g
        self.m = 1.0
        self.l = 1.0
        high = np.array([1.0, 1.0, self.max_speed], dtype=np.float32)
        self.action_space = spaces.Box(low=-self.max_torque, high=self.max_torque, shape=(1,), dtype=np.float32)
        self.observation_space = spaces.Box(low=-high, high=high, dtype=np.float32)
        self.seed()
        self.state = None

    def seed(self, seed=None):
        self.np_random, seed = seeding.np_random(seed)
        return [seed]

    def step(self, u):
        th, thdot = self.state
        g, m, l, dt = self.g, self.m, self.l, self.dt
        u_in = np.asarray(u, dtype=np.float64).reshape(-1)
        u = np.clip(u_in, -self.max_torque, self.max_torque)[0]
        costs = _angle_normalize(th) ** 2 + 0.1 * thdot ** 2 + 0.001 * (u ** 2)
        newthdot = thdot + (3 * g / (2 * l) * np.sin(th) + 3.0 / (m * l ** 2) * u) * dt
        newthdot = np.clip(newthdot, -self.max_speed, self.max_speed)
        newth = th + newthdot * dt
        state_before = (float(th), float(thdot))
        self.state = np.array([newth, newthdot])
        obs = self._get_obs()
        _emit("step", state=state_before, action=float(u_in[0]), next_state=(float(newth), float(newthdot)),
              obs=obs.copy(), reward=float(-costs), terminated=False)
        return obs, -costs, False, {}

    def reset(self):
        high = np.array([np.pi, 1])
        self.state = self.np_random.uniform(low=-high, high=high)
        _emit("reset", state=tuple(float(v) for v in self.state))
        return self._get_obs()

    def _get_obs(self):
        theta, thetadot = self.state
        return np.array([np.cos(theta), np.sin(theta), thetadot], dtype=np.float32)


_REGISTRY = {
    "CartPole-v1": (CartPoleEnv, EnvSpec("CartPole-v1", max_episode_steps=500, reward_threshold=475.0)),
    "CartPole-v0": (CartPoleEnv, EnvSpec("CartPole-v0", max_episode_steps=200, reward_threshold=195.0)),
    "Pendulum-v1": (PendulumEnv, EnvSpec("Pendulum-v1", max_episode_steps=200)),
}


_ALIASES = {}


def alias(id, target):
    """Make gym.make(id) build `target` instead (used to run the unmodified sac.py, whose env id is a Bullet task, on Pendulum)."""
    _ALIASES[id] = target


def make(id, **kwargs):
    id = _ALIASES.get(id, id)
    if id not in _REGISTRY:
        raise KeyError("gym shim: unknown env id {!r} (has {})".format(id, sorted(_REGISTRY)))
    cls, spec = _REGISTRY[id]
    env = cls(**kwargs)
    env.spec = spec
    return TimeLimit(env, max_episode_steps=spec.max_episode_steps)
