"""Minimal build-owned stand-in for the `gym==0.21` surface the reference scripts touch.

TEST INFRASTRUCTURE ONLY (part of ``oracle/``).  It exists so that the unmodified reference
scripts under ``/root/reference/deep_rl`` can be executed in this container (real ``gym`` is not
installable here) to generate golden traces; nothing in ``deep_rl_amd`` imports it.

Surface ("gym-0.21 Env" row): ``gym.make``, ``gym.Wrapper``, ``gym.Env``,
``gym.wrappers.RecordEpisodeStatistics``, ``gym.wrappers.TimeLimit``, ``gym.spaces.{Box,Discrete}``,
``gym.utils.seeding.np_random``.

The arithmetic restates the *published* gym 0.21 algorithms (classic_control/cartpole.py,
pendulum.py, wrappers/time_limit.py, wrappers/record_episode_statistics.py, utils/seeding.py)
from memory: gym 0.21 itself is not available in this container, so parity at this boundary is
UNPINNED by any third-party golden vector (see DESIGN.md "Oracle").  Call sites in the reference:
ppo.py:3,10,17,21,79,84; dqn.py:16,20,56,61,64,89; sac.py:21,25,96,101,104,139.
"""
from gym.core import Env, Wrapper  # noqa: F401
from gym import spaces, wrappers, utils  # noqa: F401
from gym.envs import make, register_trace_sink, alias  # noqa: F401

__version__ = "0.21.0+shim"
