#!/usr/bin/env python3
"""Diagnostic: run one full-size minibatch gradient with a GRAD_STAMPS build (MIRL_SO=...) and print where a wave's
cycles go per tile phase (s_memtime stamps, cdna_hip_programming.md §7 'In-kernel stamps').  Never quote its run time."""
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
eng.reset(); eng.rollout(); eng.compute_gae(); eng.make_perm(0); eng.adv_stats()
for _ in range(3):
    eng.minibatch_grad(0)
torch.cuda.synchronize()
ws = eng.workspace.view(torch.float32).cpu().numpy()
names = ["issue gathers", "L1 + tanh", "L2 mfma", "tanh h2 (+L2 drain)", "head + loss", "stage h2 + dW3", "dz2", "dh1 mfma + dz1",
         "stage dz1 + dW1", "stage + dW2", "prefetch wait", "TOTAL loop", "tiles"]
for role, nm in ((0, "actor"), (1, "critic")):
    rows = []
    for b in range(role, 512, 2):
        raw = ws[(512 + b) * 4624:(512 + b) * 4624 + 4 * 32].view(np.uint64).reshape(4, 16)
        rows.append(raw[:, :13].astype(np.float64))
    a = np.concatenate(rows)  # [waves, 13]
    tiles = a[:, 12].mean()
    print("== %s waves: %d, tiles/wave %.1f, loop cycles/wave %.0f (per tile %.0f)" % (nm, len(a), tiles, a[:, 11].mean(), a[:, 11].mean() / tiles))
    for k in range(11):
        print("   %-22s %8.0f cycles/tile  (%4.1f%%)" % (names[k], a[:, k].mean() / tiles, 100 * a[:, k].sum() / a[:, 11].sum()))
