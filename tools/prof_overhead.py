#!/usr/bin/env python3
"""What do the HIP events that bracket grad_kernel inside bench.py's timed region cost?  The same 20 updates with every gradient launch bracketed (bench.py until round 4),
with the launches of every 4th update bracketed, and with none."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import deep_rl_amd as D
from deep_rl_amd import _native as N
dev = torch.device("cuda", 0)
env = D.make("CartPole-v1", num_envs=4096, device=dev, seed=1)
torch.manual_seed(1)
agent = D.ActorCritic(env); opt = D.ClipAdam(agent, lr=2.5e-4, eps=1e-5, max_grad_norm=0.5)
eng = D.PPOEngine(env, agent, opt, num_steps=128, n_minibatch=4, update_epochs=4); eng.reset()
host = torch.zeros(4, dtype=torch.int32).pin_memory()
for _ in range(80): eng.update()
def run(every, n=20):
    torch.cuda.synchronize()
    if every: N.prof_begin(n * 16 + 16, tags=["grad"])
    t0 = time.perf_counter()
    for u in range(n):
        if every > 1: N.lib().mi_prof_pause(0 if u % every == 0 else 1)
        eng.update(); eng.episode_summary_async(host)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    p = N.prof_end() if every else {"grad": (0.0, 0)}
    return 1e3 * dt / n, 1e3 * p["grad"][0] / max(p["grad"][1], 1), p["grad"][1]
for rep in range(3):
    print(" | ".join("%s: %.4f ms/update, grad %.2f us over %d launches" % ((name,) + run(ev)) for name, ev in (("all bracketed", 1), ("every 4th update", 4), ("none", 0))))
