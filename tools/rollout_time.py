#!/usr/bin/env python3
"""How does the rollout kernel's duration depend on the policy (episode length)?  Times it at update 0, 5, 20, 50."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import deep_rl_amd as D
from deep_rl_amd import _native as N
dev = torch.device("cuda", 0)
env = D.make("CartPole-v1", num_envs=4096, device=dev, seed=1)
torch.manual_seed(1)
agent = D.ActorCritic(env); opt = D.ClipAdam(agent, lr=2.5e-4, eps=1e-5, max_grad_norm=0.5)
eng = D.PPOEngine(env, agent, opt)
eng.reset()
for u in range(51):
    if u in (0, 1, 5, 10, 20, 50):
        N.prof_begin(256)
        eng.update()
        p = N.prof_end()
        st = eng.episode_stats.tolist()
        print("update %2d: rollout %.3f ms  grad %.1f us  total kernels %.3f ms | episodes %d mean len %.1f" % (
            u, p["rollout"][0], 1e3 * p["grad"][0] / 16, sum(v[0] for v in p.values()), st[0], st[1] / max(st[0], 1)))
    else:
        eng.update()
