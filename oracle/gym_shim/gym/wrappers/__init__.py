"""gym.wrappers.{TimeLimit, RecordEpisodeStatistics} restated from memory of gym 0.21 (UNVERIFIED)."""
import time
from collections import deque

import numpy as np

from gym.core import Wrapper


class TimeLimit(Wrapper):
    def __init__(self, env, max_episode_steps=None):
        super().__init__(env)
        if max_episode_steps is None and self.env.spec is not None:
            max_episode_steps = env.spec.max_episode_steps
        self._max_episode_steps = max_episode_steps
        self._elapsed_steps = None

    def step(self, action):
        assert self._elapsed_steps is not None, "Cannot call env.step() before calling reset()"
        observation, reward, done, info = self.env.step(action)
        self._elapsed_steps += 1
        if self._elapsed_steps >= self._max_episode_steps:
            info["TimeLimit.truncated"] = not done
            done = True
        return observation, reward, done, info

    def reset(self, **kwargs):
        self._elapsed_steps = 0
        return self.env.reset(**kwargs)


class RecordEpisodeStatistics(Wrapper):
    def __init__(self, env, deque_size=100):
        super().__init__(env)
        self.num_envs = getattr(env, "num_envs", 1)
        self.t0 = time.perf_counter()
        self.episode_count = 0
        self.episode_returns = None
        self.episode_lengths = None
        self.return_queue = deque(maxlen=deque_size)
        self.length_queue = deque(maxlen=deque_size)

    def reset(self, **kwargs):
        observations = super().reset(**kwargs)
        self.episode_returns = np.zeros(self.num_envs, dtype=np.float32)
        self.episode_lengths = np.zeros(self.num_envs, dtype=np.int32)
        return observations

    def step(self, action):
        observation, reward, done, info = super().step(action)
        self.episode_returns += reward
        self.episode_lengths += 1
        if done:
            info = info.copy()
            episode_return = self.episode_returns[0]
            episode_length = self.episode_lengths[0]
            info["episode"] = {
                "r": episode_return,
                "l": episode_length,
                "t": round(time.perf_counter() - self.t0, 6),
            }
            self.return_queue.append(episode_return)
            self.length_queue.append(episode_length)
            self.episode_count += 1
            self.episode_returns[0] = 0
            self.episode_lengths[0] = 0
        return observation, reward, done, info
