"""Prioritized-replay DQN on CartPole-v1 — the drop-in counterpart of the reference single-file script ``deep_rl/per.py``.

The reference's env (LunarLander-v2, per.py:39) needs Box2D, absent here and on the GPU box; the path runs on CartPole-v1, the env
the golden trace of the unmodified reference was captured on.  Same top-level names, hyper-parameters, seeding order, storage index conventions, printed lines and final module globals as the
reference, with an env axis ``num_envs`` (NUM_ENVS, default 1).  One loop iteration = ``train_frequency`` env steps of every env
in one launch (the online network is frozen between two updates, dqn.py:114-115) followed by one TD update.
Env knobs: NUM_ENVS, TOTAL_TIMESTEPS (time steps; default 100_000), MEMORY_SIZE (ring slots; default TOTAL_TIMESTEPS + 1 =
the reference's linear storage), BATCH_SIZE.
"""
import os

import numpy as np
import torch

from deep_rl_amd import ClipAdam, PERDQNEngine, QNetwork, make
from deep_rl_amd.dist import init_from_env

env_id = "CartPole-v1"  # dqn.py:39

num_envs = int(os.environ.get("NUM_ENVS", "1"))
rank, world_size, local_rank = init_from_env("nccl")
device = torch.device("cuda", local_rank)
torch.cuda.set_device(device)

total_timesteps = int(os.environ.get("TOTAL_TIMESTEPS", "100000"))  # :41
learning_starts = int(os.environ.get("LEARNING_STARTS", str(min(10_000, total_timesteps // 10))))  # :42

start_e = 1  # :44
end_e = 0.05
exploration_fraction = 0.5
slope = (end_e - start_e) / (exploration_fraction * total_timesteps)  # :47

alpha = 0.6  # per.py:50-51
beta_0 = 0.4

train_frequency = 10  # :49
batch_size = int(os.environ.get("BATCH_SIZE", "128"))
gamma = 0.99
learning_rate = 2.5e-4
target_network_frequency = 500  # :53
assert target_network_frequency % train_frequency == 0, "target_network_frequency must be a multiple of train_frequency"  # conditions are looked at every train_frequency steps

# Env setup (:56-57)
env = make(env_id, num_envs=num_envs, device=device, env_id_base=rank * num_envs)

# Seeding (:60-64)
seed = int(os.environ.get("SEED", "1"))  # the reference hard-codes 1; SEED re-keys every counter-based stream (tests/test_gpu_learning.py runs seeds 1..10)
env.seed(seed)
np.random.seed(seed)
torch.manual_seed(seed)
env.action_space.seed(seed)  # per.py:67 (the engine's random actions are keyed in-kernel)

# Network setup (:67-70)
q_network = QNetwork(env)
optimizer = ClipAdam(q_network, lr=learning_rate, eps=1e-8)  # optim.Adam defaults, no gradient clipping
target_network = QNetwork(env)
target_network.load_state_dict(q_network.state_dict())

# Storage setup (:73-76) lives in the engine as a [slots, num_envs] ring
memory_size = int(os.environ.get("MEMORY_SIZE", str(total_timesteps + 1)))
print_episodes = int(os.environ.get("PRINT_EPISODES", "1" if num_envs <= 8 else "0"))
# the reference's exploration draw has no learning_starts guard (per.py:95): random actions come from epsilon alone
engine = PERDQNEngine(env, q_network, target_network, optimizer, slots=memory_size, alpha=alpha, beta_0=beta_0, batch_size=batch_size, gamma=gamma,
                   learning_starts=0, start_e=start_e, end_e=end_e, exploration_fraction=exploration_fraction,
                   total_timesteps=total_timesteps, max_episodes_logged=(4 * train_frequency * num_envs if print_episodes else 0))
# At num_envs == 1 the storage globals are views WITHOUT the env axis, i.e. exactly the reference's shapes ("reduces to the reference at N = 1"); the
# engine keeps writing the same memory through its own (T+1, 1, ...) tensors.
_ref = (lambda t: t.squeeze(1)) if num_envs == 1 else (lambda t: t)
observations, actions, rewards, terminated = _ref(engine.observations), _ref(engine.actions), _ref(engine.rewards), _ref(engine.terminated).view(torch.bool)
priorities = _ref(engine.priorities)  # per.py:79

# Initiate the environment and store the initial observation (:79-81)
observation = engine.reset()
observation = observation.squeeze(0) if num_envs == 1 else observation
global_step = 0

# Loop (:84)
while global_step < total_timesteps:
    n = min(train_frequency - global_step % train_frequency, total_timesteps - global_step)
    engine.act(n)  # :86-108 for n time steps
    if print_episodes and rank == 0:
        _, finished = engine.drain_episodes()
        for e, t, r, _l in finished:
            print(f"global_step={(global_step + t + 1)}, episodic_return={r:.2f}")  # :110-111 (printed after the increment)
    global_step += n

    # Optimize the agent (:114-133)
    if global_step >= learning_starts:
        if global_step % train_frequency == 0:
            engine.train_step()
        # Update the target network (:136-137)
        if global_step % target_network_frequency == 0:
            engine.sync_target()

loss = float(engine.loss.item())
max_priority = float(engine.max_priority.item())  # per.py:142
env.close()
