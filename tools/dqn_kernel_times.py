#!/usr/bin/env python3
"""In-library HIP-event times of the DQN kernels over a few hundred iterations of the config-3 loop (MIRL_SO selects the build, BATCH the batch size: default 128)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import deep_rl_amd as D
from deep_rl_amd import _native as N
dev = torch.device("cuda", 0)
env = D.make("CartPole-v1", num_envs=4096, device=dev, seed=1)
torch.manual_seed(1)
q = D.QNetwork(env); t = D.QNetwork(env); t.load_state_dict(q.state_dict())
eng = D.DQNEngine(env, q, t, D.ClipAdam(q, lr=2.5e-4, eps=1e-8), slots=256, batch_size=int(os.environ.get('BATCH', '128')), learning_starts=100, total_timesteps=10 * 500)
eng.reset()
for _ in range(50):
    eng.act(10); eng.train_step()
torch.cuda.synchronize()
N.prof_begin(200 * 4, tags=["dqn_act", "dqn_td", "dqn_reduce"])
for _ in range(200):
    eng.act(10); eng.train_step()
print('batch', eng.batch_size, {k: round(1e3 * v[0] / max(v[1], 1), 2) for k, v in N.prof_end().items() if v[1]})
