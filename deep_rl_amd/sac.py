"""SAC on Pendulum-v1 — the drop-in counterpart of the reference single-file script ``deep_rl/sac.py``.

Same top-level names, hyper-parameters, seeding order, storage index conventions, printed lines and final module globals as the
reference, with an env axis ``num_envs`` (NUM_ENVS, default 1).  The reference's env (HopperBulletEnv-v0, sac.py:80) needs
pybullet, which neither this image nor the GPU box has; the path is re-targeted to gym's Pendulum-v1, the
continuous-control env the golden trace of the unmodified reference was captured on.
Env knobs: NUM_ENVS, TOTAL_TIMESTEPS (time steps; default 30_000), LEARNING_STARTS, MEMORY_SIZE (ring slots; default
TOTAL_TIMESTEPS + 1 = the reference's linear storage), BATCH_SIZE.
"""
import os

import numpy as np
import torch

from deep_rl_amd import Actor, SACEngine, SoftQNetwork, make
from deep_rl_amd.dist import init_from_env

env_id = "Pendulum-v1"  # sac.py:80 (HopperBulletEnv-v0 in the reference)

num_envs = int(os.environ.get("NUM_ENVS", "1"))
rank, world_size, local_rank = init_from_env("nccl")
device = torch.device("cuda", local_rank)
torch.cuda.set_device(device)

total_timesteps = int(os.environ.get("TOTAL_TIMESTEPS", "30000"))  # :82
learning_starts = int(os.environ.get("LEARNING_STARTS", str(min(5_000, total_timesteps // 6))))  # :83

policy_frequency = 2  # :85
batch_size = int(os.environ.get("BATCH_SIZE", "256"))
target_network_frequency = 1
gamma = 0.99
tau = 0.005
policy_lr = 3e-4
q_lr = 1e-3
alpha_lr = q_lr  # :92

# Env setup (:95-96)
env = make(env_id, num_envs=num_envs, device=device, env_id_base=rank * num_envs)

# Seeding (:99-104)
seed = int(os.environ.get("SEED", "1"))  # the reference hard-codes 1; SEED re-keys every counter-based stream (tests/test_gpu_learning.py runs seeds 1..10)
env.seed(seed)
np.random.seed(seed)
torch.manual_seed(seed)
env.action_space.seed(seed)

# Actor setup (:107-108) — the optimizers live in the engine (one Adam launch per flat buffer)
actor = Actor(env)

# Networks setup (:111-117)
qf1 = SoftQNetwork(env)
qf2 = SoftQNetwork(env)
qf1_target = SoftQNetwork(env)
qf2_target = SoftQNetwork(env)
qf1_target.load_state_dict(qf1.state_dict())
qf2_target.load_state_dict(qf2.state_dict())

target_entropy = -float(np.prod(env.action_space.shape))  # :119

# Storage setup (:126-129) lives in the engine as a [slots, num_envs] ring; log_alpha / alpha (:120-122) as device scalars
memory_size = int(os.environ.get("MEMORY_SIZE", str(total_timesteps + 1)))
print_episodes = int(os.environ.get("PRINT_EPISODES", "1" if num_envs <= 8 else "0"))
engine = SACEngine(env, actor, qf1, qf2, qf1_target, qf2_target, slots=memory_size, batch_size=batch_size, gamma=gamma, tau=tau,
                   policy_lr=policy_lr, q_lr=q_lr, alpha_lr=alpha_lr, learning_starts=learning_starts, target_entropy=target_entropy,
                   max_episodes_logged=(4 * num_envs if print_episodes else 0))
actor_optimizer, q_optimizer = engine.actor_optimizer, engine.q_optimizer
log_alpha = engine.log_alpha
# At num_envs == 1 the storage globals are views WITHOUT the env axis, i.e. exactly the reference's shapes ("reduces to the reference at N = 1"); the
# engine keeps writing the same memory through its own (T+1, 1, ...) tensors.
_ref = (lambda t: t.squeeze(1)) if num_envs == 1 else (lambda t: t)
# actions keep their (slots, act_dim = 1) shape at one env: the env axis and sac.py:127's action axis coincide
observations, actions, rewards, terminated = _ref(engine.observations), engine.actions, _ref(engine.rewards), _ref(engine.terminated).view(torch.bool)

# Initiate the environment and store the initial observation (:132-134)
observation = engine.reset()
observation = observation.squeeze(0) if num_envs == 1 else observation
global_step = 0

# Loop (:137)
while global_step < total_timesteps:
    engine.act()  # :138-158 for every env
    global_step += 1
    if print_episodes and rank == 0 and global_step % 200 == 0:  # Pendulum episodes end every 200 steps (TimeLimit)
        for _e, r, _l in engine.drain_episodes():
            print(f"global_step={global_step}, episodic_return={r:.2f}")  # :160-161

    # Optimize target and agent (:164-217)
    if global_step >= learning_starts:
        engine.train_step(policy_frequency, target_network_frequency)

engine.flush()  # settle what is still owed (the last critic step rides on an acting launch that never comes, the last alpha step on an update launch)
alpha = float(engine.alpha.item())  # :210 — read once here instead of once per actor update
env.close()
