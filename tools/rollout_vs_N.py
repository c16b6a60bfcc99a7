#!/usr/bin/env python3
"""Rollout (+ fused GAE) launch time against the env count at T = 128: flat below the count that fills the chip = a latency chain; growing = shared issue slots."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, deep_rl_amd as D
from deep_rl_amd import _native as N
dev = torch.device("cuda", 0)
for n in (256, 1024, 2048, 4096, 8192):
    env = D.make("CartPole-v1", num_envs=n, device=dev, seed=1)
    torch.manual_seed(1)
    agent = D.ActorCritic(env); opt = D.ClipAdam(agent, lr=2.5e-4, eps=1e-5, max_grad_norm=0.5)
    eng = D.PPOEngine(env, agent, opt, num_steps=128)
    eng.reset()
    for _ in range(5): eng.rollout_gae()
    torch.cuda.synchronize()
    N.prof_begin(64, tags=["rollout"])
    for _ in range(20): eng.rollout_gae()
    r = N.prof_end()["rollout"]
    print("N=%5d (%4d workgroups of 2 waves) rollout+GAE %.1f us" % (n, n // 4, 1e3 * r[0] / r[1]))
