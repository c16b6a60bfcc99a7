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
if os.environ.get("STAMP_UPDATE", "0") == "1":   # the stamps of the LAST gradient launch of a whole update (owed clip + Adam step on its weight staging)
    for _ in range(3):
        eng.update()
else:
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

# wall-clock marks (s_memrealtime, 10 ns ticks): entry -> weights staged -> loop end -> exit, over all waves
marks = []
for b in range(512):
    base = (512 + b) * 4624
    raw = ws[base:base + 4 * 32].view(np.uint64).reshape(4, 16)
    outs = ws[base + 128:base + 128 + 8].view(np.uint64)
    for w in range(4):
        if raw[w, 13]:
            marks.append((b, w, int(raw[w, 13]), int(raw[w, 14]), int(raw[w, 15]), int(outs[w])))
m = np.array([x[2:] for x in marks], dtype=np.float64)
t0 = m[:, 0].min()
m = (m - t0) / 100.0  # microseconds
print("waves %d; first entry 0.00 us, last entry %.2f us, last exit %.2f us (kernel span)" % (len(m), m[:, 0].max(), m[:, 3].max()))
print("  entry->staged  mean %.2f  max %.2f us" % ((m[:, 1] - m[:, 0]).mean(), (m[:, 1] - m[:, 0]).max()))
print("  tile loop      mean %.2f  min %.2f max %.2f us" % ((m[:, 2] - m[:, 1]).mean(), (m[:, 2] - m[:, 1]).min(), (m[:, 2] - m[:, 1]).max()))
print("  epilogue       mean %.2f  max %.2f us" % ((m[:, 3] - m[:, 2]).mean(), (m[:, 3] - m[:, 2]).max()))
print("  exit times: p10 %.2f p50 %.2f p90 %.2f max %.2f us" % tuple(np.percentile(m[:, 3], [10, 50, 90, 100])))
roles = np.array([(((x[0] >> 3) & 1)) for x in marks])
for r, nm in ((0, "actor"), (1, "critic")):
    sel = roles == r
    print("  %s: loop mean %.2f us, exit mean %.2f max %.2f us" % (nm, (m[sel, 2] - m[sel, 1]).mean(), m[sel, 3].mean(), m[sel, 3].max()))

# spread of the loop time: by XCD label (block % 8), by wave-in-block, and the slowest / fastest blocks
loop = m[:, 2] - m[:, 1]
blk = np.array([x[0] for x in marks]); wv = np.array([x[1] for x in marks])
print("  loop time by XCD label (b%8): " + "  ".join("%d:%.1f" % (x, loop[blk % 8 == x].mean()) for x in range(8)))
print("  loop time by wave in block  : " + "  ".join("%d:%.1f" % (x, loop[wv == x].mean()) for x in range(4)))
pb = np.array([loop[blk == b].mean() for b in range(512)])
sb = np.array([loop[blk == b].max() - loop[blk == b].min() for b in range(512)])
print("  per-block mean loop: min %.1f p10 %.1f p50 %.1f p90 %.1f max %.1f us; within-block spread mean %.2f us" % (
    pb.min(), *np.percentile(pb, [10, 50, 90]), pb.max(), sb.mean()))
order = np.argsort(pb)
print("  fastest blocks:", [(int(b), round(float(pb[b]), 1)) for b in order[:8]])
print("  slowest blocks:", [(int(b), round(float(pb[b]), 1)) for b in order[-8:]])
# do the two blocks that share a CU differ?  (unknown placement: correlate block b with b+8, b+256 ...)
for d in (8, 16, 256):
    print("  corr(loop[b], loop[b+%d]) = %.2f" % (d, np.corrcoef(pb[:512 - d], pb[d:])[0, 1]))

# is the spread a clock effect (same cycles, different MHz) or extra cycles?
cyc = np.zeros(len(marks))
phase = np.zeros((len(marks), 11))
for n, (b, w, *_rest) in enumerate(marks):
    raw = ws[(512 + b) * 4624:(512 + b) * 4624 + 4 * 32].view(np.uint64).reshape(4, 16)
    cyc[n] = raw[w, 11]; phase[n] = raw[w, :11]
mhz = cyc / loop
print("  shader clock over the loop (cycles/us): min %.0f p50 %.0f max %.0f; corr(loop_us, cycles) = %.3f, corr(loop_us, MHz) = %.3f" % (
    mhz.min(), np.median(mhz), mhz.max(), np.corrcoef(loop, cyc)[0, 1], np.corrcoef(loop, mhz)[0, 1]))
fast = loop < np.percentile(loop, 10); slow = loop > np.percentile(loop, 90)
print("  phase cycles/tile, fastest 10%% vs slowest 10%% of waves (cycles %.0f vs %.0f):" % (cyc[fast].mean(), cyc[slow].mean()))
for k in range(11):
    print("     %-22s %7.0f  %7.0f" % (names[k], phase[fast, k].mean() / 8, phase[slow, k].mean() / 8))

# hardware placement (HW_REG_HW_ID: wave[3:0] simd[5:4] pipe[7:6] cu[11:8] sh[12] se[15:13]; HW_REG_XCC_ID[3:0])
hw = []
for (b, w, *_r) in marks:
    v = int(ws[(512 + b) * 4624 + 128 + 8:(512 + b) * 4624 + 128 + 16].view(np.uint64)[w])
    hid, xcc = v & 0xffffffff, (v >> 32) & 0xf
    hw.append((xcc, (hid >> 13) & 7, (hid >> 12) & 1, (hid >> 8) & 15, (hid >> 4) & 3, hid & 15))
hw = np.array(hw)
cu_key = hw[:, 0] * 1000 + hw[:, 1] * 100 + hw[:, 2] * 50 + hw[:, 3]
uk = np.unique(cu_key)
print("  distinct (xcc,se,sh,cu): %d; waves per CU: %s" % (len(uk), np.unique(np.bincount(np.searchsorted(uk, cu_key)), return_counts=True)))
for nm, col in (("xcc", 0), ("se", 1), ("sh", 2), ("cu", 3), ("simd", 4)):
    vals = np.unique(hw[:, col])
    print("  loop us by %-4s: " % nm + "  ".join("%d:%.1f" % (v, loop[hw[:, col] == v].mean()) for v in vals))
simd_key = cu_key * 4 + hw[:, 4]
cnt = np.bincount(np.searchsorted(np.unique(simd_key), simd_key))
print("  waves per SIMD histogram:", np.unique(cnt, return_counts=True))
# per-CU: loop time vs how its 8 waves are spread over SIMDs
percu = {}
for k, sd, lp in zip(cu_key, hw[:, 4], loop):
    percu.setdefault(k, []).append((sd, lp))
rows = []
for k, lst in percu.items():
    c = np.bincount([x[0] for x in lst], minlength=4)
    rows.append((tuple(sorted(c)), np.mean([x[1] for x in lst])))
import collections
agg = collections.defaultdict(list)
for pat, t in rows:
    agg[pat].append(t)
for pat, ts in sorted(agg.items()):
    print("  SIMD occupancy pattern %s: %d CUs, mean loop %.1f us" % (pat, len(ts), np.mean(ts)))

# prologue sub-phases (s_memtime): entry | loads issued | norm done (loads landed, barrier) | Adam done | LDS stores issued + gathers | barrier
pro = []
for b in range(512):
    base = (512 + b) * 4624
    raw = ws[base + 2 * 72: base + 2 * 72 + 2 * 24].view(np.uint64).reshape(4, 6).astype(np.float64)
    if raw[0, 0]: pro.append(raw)
if pro:
    pro = np.concatenate(pro)
    d = np.diff(pro, axis=1)
    print("  prologue cycles (mean over waves): issue loads %.0f | norm (land + reduce + barrier) %.0f | Adam %.0f | LDS stores + first gathers %.0f | barrier %.0f" % tuple(d.mean(0)))
