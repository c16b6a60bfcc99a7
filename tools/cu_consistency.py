#!/usr/bin/env python3
"""Diagnostic (GRAD_STAMPS build): are the slow CUs of grad_kernel the same physical CUs in every launch?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import deep_rl_amd as D

dev = torch.device("cuda", 0)
env = D.make("CartPole-v1", num_envs=4096, device=dev, seed=1)
torch.manual_seed(1)
agent = D.ActorCritic(env); opt = D.ClipAdam(agent, lr=2.5e-4, eps=1e-5, max_grad_norm=0.5)
eng = D.PPOEngine(env, agent, opt)
eng.reset(); eng.rollout(); eng.compute_gae(); eng.make_perm(0); eng.adv_stats()
runs = []
for it in range(4):
    eng.minibatch_grad(it % 4)
    torch.cuda.synchronize()
    ws = eng.workspace.view(torch.float32).cpu().numpy()
    per_cu = {}
    for b in range(512):
        base = (512 + b) * 4624
        raw = ws[base:base + 128].view(np.uint64).reshape(4, 16)
        ids = ws[base + 128 + 8:base + 128 + 16].view(np.uint64)
        for w in range(4):
            hid, xcc = int(ids[w]) & 0xffffffff, (int(ids[w]) >> 32) & 0xf
            key = (xcc, (hid >> 13) & 7, (hid >> 8) & 15)
            per_cu.setdefault(key, []).append((int(raw[w, 15]) - int(raw[w, 14])) / 100.0)
    runs.append({k: np.mean(v) for k, v in per_cu.items()})
keys = sorted(runs[0])
M = np.array([[r[k] for k in keys] for r in runs])
print("per-CU loop us: launch means", M.mean(1).round(1), " min", M.min(1).round(1), " max", M.max(1).round(1))
print("correlation of per-CU time between launches:\n", np.corrcoef(M).round(2))
slow = np.argsort(M.mean(0))[-10:]; fast = np.argsort(M.mean(0))[:10]
print("consistently slowest CUs (xcc,se,cu):", [keys[i] for i in slow], M[:, slow].mean(0).round(1))
print("consistently fastest CUs (xcc,se,cu):", [keys[i] for i in fast], M[:, fast].mean(0).round(1))
