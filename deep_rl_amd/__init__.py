"""deep_rl_amd — MI355X-native vectorised-rollout + policy-update engine behind the call surface of
qgallouedec/deep_rl's single-file scripts (ppo.py first).

Host code is Python on PyTorch-ROCm for module / optimizer bookkeeping only; the hot path (batched
CartPole stepper, rollout storage, GAE scan, tiny-MLP forward/backward, clip + Adam) is hand-written HIP
for gfx950 in ``csrc/`` behind the C ABI of ``include/mi_rl.h``.  There is no CPU fallback: importing
``deep_rl_amd._native`` without the built ``libmirl.so`` raises.
"""
from . import _native  # noqa: F401
from ._native import set_contraction, get_contraction  # noqa: F401
from .envs import make, CartPoleVecEnv, PendulumVecEnv  # noqa: F401
from .agent import ActorCritic, QNetwork, DuelingQNetwork, SoftQNetwork, Actor, layer_init, pack  # noqa: F401
from .optim import ClipAdam, Adam  # noqa: F401
from .engine import PPOEngine  # noqa: F401
from .dqn_engine import DQNEngine, DuelingDQNEngine, PERDQNEngine  # noqa: F401
from .sac_engine import SACEngine  # noqa: F401

__all__ = ["make", "CartPoleVecEnv", "ActorCritic", "QNetwork", "layer_init", "ClipAdam", "PPOEngine", "DQNEngine",
           "DuelingQNetwork", "DuelingDQNEngine", "PERDQNEngine", "PendulumVecEnv", "SoftQNetwork", "Actor", "Adam", "SACEngine", "pack", "set_contraction", "get_contraction"]
