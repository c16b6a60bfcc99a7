#!/usr/bin/env python3
"""A/B of the one-launch TD update (round 5) against the two-launch form in ONE process, interleaved: config-3 loop (4096 envs, 1M ring, 10 env steps + 1 update per
iteration) at BATCH (default 128); us per iteration by wall clock over 600 iterations, kernel times by the in-library HIP events over 200 more."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import deep_rl_amd as D
from deep_rl_amd import _native as N
dev = torch.device("cuda", 0)
batch = int(os.environ.get("BATCH", "128"))
env = D.make("CartPole-v1", num_envs=4096, device=dev, seed=1)
torch.manual_seed(1)
q = D.QNetwork(env); t = D.QNetwork(env); t.load_state_dict(q.state_dict())
eng = D.DQNEngine(env, q, t, D.ClipAdam(q, lr=2.5e-4, eps=1e-8), slots=256, batch_size=batch, learning_starts=100, total_timesteps=10 * 20000)
eng.reset()
def it():
    eng.act(10); eng.train_step()
for _ in range(300): it()
for rep in range(3):
    for fused in (1, 0):
        N.check(N.lib().mi_dqn_set_fused_step(fused))
        for _ in range(50): it()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(600): it()
        torch.cuda.synchronize(); us = 1e6 * (time.perf_counter() - t0) / 600
        N.prof_begin(200 * 4, tags=["dqn_act", "dqn_td", "dqn_reduce"])
        for _ in range(200): it()
        k = {n: round(1e3 * v[0] / max(v[1], 1), 2) for n, v in N.prof_end().items() if v[1]}
        print("batch %d %s: %.2f us per iteration; kernels (HIP events) %s" % (batch, "one launch " if fused else "two launches", us, k), flush=True)
N.lib().mi_dqn_set_fused_step(1)
