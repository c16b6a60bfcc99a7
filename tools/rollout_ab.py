#!/usr/bin/env python3
"""A/B timing of the rollout launch for one build (MIRL_SO=...): mean / min over 40 rollout + GAE launches with the policy FROZEN, (a) at the initial policy (episodes of
~22 steps: the reset path runs every few steps) and (b) after 40 real updates (episodes of 150+ steps).  HIP events on the launch stream (mi_prof)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import deep_rl_amd as D
from deep_rl_amd import _native as N
dev = torch.device("cuda", 0)
env = D.make("CartPole-v1", num_envs=4096, device=dev, seed=1)
torch.manual_seed(1)
agent = D.ActorCritic(env); opt = D.ClipAdam(agent, lr=2.5e-4, eps=1e-5, max_grad_norm=0.5)
eng = D.PPOEngine(env, agent, opt)
eng.reset()


def timed(tag):
    for _ in range(5):
        eng.rollout_gae()
    ts = []
    for _ in range(40):
        N.prof_begin(8, tags=["rollout"])
        eng.rollout_gae()
        p = N.prof_end()
        ts.append(1e3 * p["rollout"][0])
    st = eng.episode_stats.tolist()
    print("%-28s rollout+GAE %.1f us mean, %.1f min, %.1f max | mean episode length %.1f" % (tag, np.mean(ts), np.min(ts), np.max(ts), st[1] / max(st[0], 1)))


timed("initial policy, cold GPU")
for _ in range(400):      # ~80 ms of rollouts: the clocks have ramped by now; the policy is still the initial one
    eng.rollout_gae()
timed("initial policy")
for _ in range(40):
    eng.update()
timed("after 40 updates")
