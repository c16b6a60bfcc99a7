#!/usr/bin/env python3
"""Diagnostic: run full-size minibatch gradients with a -DGRAD_STAMPS build (MIRL_SO=...) and print where grad_kernel's time goes:
wall-clock marks per workgroup / role (s_memrealtime), cycles per tile phase and the prologue's sub-phases (s_memtime), and the shader
clock the chip sustains over the tile loop.  STAMP_UPDATE=1: stamps of the LAST gradient launch of a whole update (owed clip + Adam
step on its weight staging).  Never quote its run time (cdna_hip_programming.md §7 'In-kernel stamps')."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import deep_rl_amd as D

WAVES = 8   # GRAD_WAVES (fixed since round 3)
dev = torch.device("cuda", 0)
env = D.make("CartPole-v1", num_envs=4096, device=dev, seed=1)
torch.manual_seed(1)
agent = D.ActorCritic(env); opt = D.ClipAdam(agent, lr=2.5e-4, eps=1e-5, max_grad_norm=0.5)
eng = D.PPOEngine(env, agent, opt)
eng.reset(); eng.rollout(); eng.compute_gae(); eng.make_perm(0); eng.adv_stats()
if os.environ.get("STAMP_UPDATE", "0") == "1":
    for _ in range(3):
        eng.update()
else:
    for _ in range(3):
        eng.minibatch_grad(0)
torch.cuda.synchronize()
ws = eng.workspace.view(torch.float32).cpu().numpy()
nblk = 2048 // WAVES
names = ["issue gathers", "L1 + tanh", "L2 mfma", "tanh h2 (+L2 drain)", "head + loss", "dW3", "dz2", "dh1 mfma + dz1", "stage dz1 + dW1", "stage + dW2", "prefetch wait"]
rows = []
for b in range(nblk):
    raw = ws[(512 + b) * 4624:(512 + b) * 4624 + 2 * 32 * WAVES].view(np.uint64).reshape(WAVES, 32).astype(np.float64)
    for w in range(WAVES):
        rows.append((b & 1, b, w) + tuple(raw[w]))
a = np.array(rows)
role, blk, wv, d = a[:, 0], a[:, 1], a[:, 2], a[:, 3:]
t0 = d[:, 13].min()
entry, staged, loopend, out = [(d[:, k] - t0) / 100.0 for k in (13, 14, 15, 16)]   # us
print("waves %d in %d workgroups; last entry %.2f us, last exit %.2f us (kernel span)" % (len(a), nblk, entry.max(), out.max()))
print("  entry->staged  mean %.2f  max %.2f us | tile loop mean %.2f min %.2f max %.2f us | epilogue mean %.2f max %.2f us" % (
    (staged - entry).mean(), (staged - entry).max(), (loopend - staged).mean(), (loopend - staged).min(), (loopend - staged).max(), (out - loopend).mean(), (out - loopend).max()))
for r, nm in ((0, "actor"), (1, "critic")):
    s = role == r
    old = s & (wv < WAVES // 2 if WAVES == 8 else blk < nblk // 2)
    print("  %-6s loop end mean %.2f max %.2f us; exit mean %.2f max %.2f us; tiles/wave older half %.1f younger %.1f; loop end older %.2f younger %.2f" % (
        nm, loopend[s].mean(), loopend[s].max(), out[s].mean(), out[s].max(), d[old, 12].mean(), d[s & ~old, 12].mean(), loopend[old].mean(), loopend[s & ~old].mean()))
    per_tile = d[s, :11].sum(0) / d[s, 12].sum()
    print("         cycles per tile: " + " | ".join("%s %.0f" % (names[k], per_tile[k]) for k in range(11)) + " | total %.0f" % (d[s, 11].sum() / d[s, 12].sum()))
mhz = d[:, 11] / (loopend - staged)
print("  shader clock over the loop: p10 %.0f p50 %.0f p90 %.0f MHz" % tuple(np.percentile(mhz, [10, 50, 90])))
xcc = (d[:, 17].astype(np.uint64) >> np.uint64(32)).astype(int)
print("  loop end by XCC: " + "  ".join("%d:%.1f" % (x, loopend[xcc == x].mean()) for x in range(8)))
pro = d[:, 20:26]
if pro[:, 0].any():
    dd = np.diff(pro, axis=1)
    print("  prologue cycles (mean over waves): issue loads %.0f | norm (land + reduce + barrier) %.0f | Adam %.0f | LDS stores + first gathers %.0f | barrier %.0f" % tuple(dd.mean(0)))
