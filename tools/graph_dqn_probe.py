#!/usr/bin/env python3
"""Timing probe (GPU box): what a hipGraph of the config-3 loop could buy at most — the three launches of one iteration (acting, TD, slab sum + Adam) captured with
FROZEN arguments (the replays repeat the same step: timing only) and replayed, against the stream launches; also ten iterations per graph.  Round 5: 37.1 us per iteration
from the stream, 41.3 as a one-iteration graph, 36.6 as a ten-iteration graph — a graph launch costs more than the three boundaries it replaces, and even amortised the
gain is 1.3 %; no graph route was built (the arguments of every launch change every iteration)."""
import os, sys, time
sys.path.insert(0, os.environ.get("MIRL_ROOT", "."))
import torch
import deep_rl_amd as D
dev = torch.device("cuda", 0)
env = D.make("CartPole-v1", num_envs=4096, device=dev, seed=1)
torch.manual_seed(1)
q = D.QNetwork(env); t = D.QNetwork(env); t.load_state_dict(q.state_dict())
eng = D.DQNEngine(env, q, t, D.ClipAdam(q, lr=2.5e-4, eps=1e-8), slots=256, batch_size=128, learning_starts=100, total_timesteps=10 * 2000)
eng.reset()
def it():
    eng.act(10); eng.train_step()
for _ in range(100): it()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(600): it()
torch.cuda.synchronize(); print("stream launches: %.2f us per iteration" % (1e6 * (time.perf_counter() - t0) / 600))
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3): it()
    g = torch.cuda.CUDAGraph()
    torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=s):
        it()
torch.cuda.synchronize()
for _ in range(50): g.replay()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(600): g.replay()
torch.cuda.synchronize(); print("graph replay (frozen arguments): %.2f us per iteration" % (1e6 * (time.perf_counter() - t0) / 600))
# ten iterations per graph
g10 = torch.cuda.CUDAGraph()
with torch.cuda.stream(s):
    with torch.cuda.graph(g10, stream=s):
        for _ in range(10): it()
torch.cuda.synchronize()
for _ in range(10): g10.replay()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(60): g10.replay()
torch.cuda.synchronize(); print("graph of 10 iterations: %.2f us per iteration" % (1e6 * (time.perf_counter() - t0) / 600))
