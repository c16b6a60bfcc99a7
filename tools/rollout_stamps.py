#!/usr/bin/env python3
"""Diagnostic (-DRQ_STAMPS build, MIRL_SO=...): where the actor wave of rollout_q4_kernel spends its cycles per step."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import deep_rl_amd as D
from deep_rl_amd import _native as N
dev = torch.device("cuda", 0)
env = D.make("CartPole-v1", num_envs=4096, device=dev, seed=1)
torch.manual_seed(1)
agent = D.ActorCritic(env); opt = D.ClipAdam(agent, lr=2.5e-4, eps=1e-5, max_grad_norm=0.5)
eng = D.PPOEngine(env, agent, opt); eng.reset()
for _ in range(3): eng.rollout_gae()
torch.cuda.synchronize()
buf = (C.c_ulonglong * 8192)()
f = N.lib().mi_debug_rollout_stamps; f.argtypes = [C.c_void_p, C.c_int]; f.restype = C.c_int
assert f(buf, 8192) == 0
a = np.array(buf, dtype=np.float64).reshape(1024, 8)[:, :4] / 129.0
names = ["scalar section (draw, CartPole step, reset)", "publish + global stores", "hidden layers (L1, tanh, transpose, 64 MFMAs, tanh)", "heads (2 reductions)"]
print("cycles per step, mean over 1024 actor waves (total %.0f):" % a.sum(1).mean())
for k in range(4): print("  %-56s %7.0f  (min %6.0f max %6.0f)" % (names[k], a[:, k].mean(), a[:, k].min(), a[:, k].max()))
