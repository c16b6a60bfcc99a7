"""PPO on CartPole-v1 — the drop-in counterpart of the reference single-file script ``deep_rl/ppo.py``.

Same top-level names, hyper-parameters, seeding order, storage index conventions, printed lines and final
module globals as the reference (file:line comments point into it), with an env axis ``num_envs`` that
reduces to the reference at 1.  The arithmetic runs in hand-written HIP kernels (deep_rl_amd/csrc) on an
MI355X; this file only sequences launches.  Run:  ``python -m deep_rl_amd.ppo``  (one GPU) or
``python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 -m deep_rl_amd.ppo``.

Knobs the reference does not have are read from the environment so the constants below stay the reference's:
NUM_ENVS (default 1), TOTAL_TIMESTEPS (default 20_000 * NUM_ENVS), PRINT_EPISODES (default: 1 if NUM_ENVS <= 8).
"""
import os

import numpy as np
import torch

from deep_rl_amd import ActorCritic, ClipAdam, PPOEngine, make
from deep_rl_amd.dist import init_from_env

env_id = "CartPole-v1"  # ppo.py:62

num_envs = int(os.environ.get("NUM_ENVS", "1"))  # envs per GPU; the reference is implicitly 1
rank, world_size, local_rank = init_from_env("nccl")  # "nccl" is RCCL on ROCm; one process per GPU
device = torch.device("cuda", local_rank)
torch.cuda.set_device(device)

total_timesteps = int(os.environ.get("TOTAL_TIMESTEPS", str(20_000 * num_envs * world_size)))  # ppo.py:64
num_steps = 128  # :65
num_updates = total_timesteps // (num_steps * num_envs * world_size)  # :66, generalised by the env axis
minibatch_size = num_steps * num_envs // 4  # :67
update_epochs = 4  # :68

gamma = 0.99  # :70
gae_lambda = 0.95
learning_rate = 2.5e-4
clip_coef = 0.2
ent_coef = 0.01
vf_coef = 0.5
max_grad_norm = 0.5  # :76

# Env setup (:79-80) — rank r owns global envs [r*num_envs, (r+1)*num_envs)
env = make(env_id, num_envs=num_envs, device=device, env_id_base=rank * num_envs)

# Seeding (:83-86), same order: env, numpy, torch — before the agent is built so the init matches
seed = int(os.environ.get("SEED", "1"))  # the reference hard-codes 1; SEED re-keys every counter-based stream (tests/test_gpu_learning.py runs seeds 1..10)
env.seed(seed)
np.random.seed(seed)
torch.manual_seed(seed)

# Agent setup (:89-90)
agent = ActorCritic(env)
optimizer = ClipAdam(agent, lr=learning_rate, eps=1e-5, max_grad_norm=max_grad_norm)

# Storage setup (:93-98) lives in the engine; expose the reference's names
print_episodes = int(os.environ.get("PRINT_EPISODES", "1" if num_envs <= 8 else "0"))
engine = PPOEngine(env, agent, optimizer, num_steps=num_steps, n_minibatch=4, update_epochs=update_epochs, gamma=gamma,
                   gae_lambda=gae_lambda, clip_coef=clip_coef, ent_coef=ent_coef, vf_coef=vf_coef,
                   max_episodes_logged=(4 * num_steps * num_envs if print_episodes else 0))
# At num_envs == 1 the storage globals are views WITHOUT the env axis, i.e. exactly the reference's shapes ("reduces to the reference at N = 1"); the
# engine keeps writing the same memory through its own (T+1, 1, ...) tensors.
_ref = (lambda t: t.squeeze(1)) if num_envs == 1 else (lambda t: t)
observations, values, actions = _ref(engine.observations), _ref(engine.values), _ref(engine.actions)   # (129, 4), (129,), (129,) at one env (:93-95)
log_probs, rewards, dones = _ref(engine.log_probs), _ref(engine.rewards), _ref(engine.dones)               # (129,) each (:96-98)

# Init the env (:101-102)
observation = engine.reset()
observation = observation.squeeze(0) if num_envs == 1 else observation   # (4,) at one env; a view of the engine's carried-over observation
global_step = 0

# Loop (:105)
for update in range(num_updates):
    # Annealing the rate (:107-108)
    new_lr = (1.0 - update / num_updates) * learning_rate
    optimizer.param_groups[0]["lr"] = new_lr

    # rollout (:110-141), advantages (:144-151) and the 4 x 4 optimizer steps (:154-192): one enqueue
    engine.update()

    # episode log (:130): global_step is the count *before* the step that ended the episode
    n_finished, finished = engine.drain_episodes()
    if print_episodes and rank == 0:
        for e, t, r, _l in finished:
            print(f"global_step={global_step + t * num_envs + e}, episodic_return={r:.2f}")
    elif rank == 0 and n_finished:
        mean_r = int(engine.episode_stats[1].item()) / n_finished  # CartPole: return == length
        print(f"update={update}, global_step={global_step + num_steps * num_envs}, episodes={n_finished}, mean_episodic_return={mean_r:.2f}")
    global_step += num_steps * num_envs

advantages, returns = _ref(engine.advantages), _ref(engine.returns)
pg_loss, entropy_loss, v_loss, loss = (float(x) for x in engine.loss_terms.cpu())
explained_var = float(engine.compute_explained_var().item())  # :194-195

env.close()  # :197
