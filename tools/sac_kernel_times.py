#!/usr/bin/env python3
"""In-library HIP-event times of the SAC kernels over a few hundred iterations of the config-4 loop (MIRL_SO selects the build)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import deep_rl_amd as D
from deep_rl_amd import _native as N
dev = torch.device("cuda", 0)
env = D.make("Pendulum-v1", num_envs=2048, device=dev, seed=1)
torch.manual_seed(1)
actor = D.Actor(env); q1 = D.SoftQNetwork(env); q2 = D.SoftQNetwork(env); q1t = D.SoftQNetwork(env); q2t = D.SoftQNetwork(env)
q1t.load_state_dict(q1.state_dict()); q2t.load_state_dict(q2.state_dict())
eng = D.SACEngine(env, actor, q1, q2, q1t, q2t, slots=512, batch_size=256, learning_starts=8)
eng.reset()
for _ in range(60):
    eng.act()
    if eng.global_step > 10: eng.train_step()
torch.cuda.synchronize()
tags = ["sac_act", "sac_critic", "sac_actor", "sac_gemm", "sac_assemble", "sac_logp"]
N.prof_begin(200 * 12, tags=tags)
for _ in range(200):
    eng.act(); eng.train_step()
r = N.prof_end()
print({k: (round(1e3 * v[0] / max(v[1], 1), 2), v[1]) for k, v in r.items() if v[1]})
