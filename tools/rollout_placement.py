#!/usr/bin/env python3
"""Diagnostic (-DRQ_STAMPS build via MIRL_SO): where the hardware puts rollout_q4_kernel's waves — per (XCC, SE, CU, SIMD) how many actor / critic waves, and the
wave-slot ids they got.  HW_ID: wave_id [3:0], simd_id [5:4], pipe [7:6], cu_id [11:8], sh [12], se_id [15:13]."""
import ctypes as C, os, sys, collections
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
buf = (C.c_ulonglong * 9216)()
f = N.lib().mi_debug_rollout_stamps; f.argtypes = [C.c_void_p, C.c_int]; f.restype = C.c_int
assert f(buf, 9216) == 0
a = np.array(buf, dtype=np.uint64)[:8192].reshape(1024, 8)
per_simd = collections.defaultdict(list)
pairs = collections.Counter()
for b in range(1024):
    ids = []
    for w in range(2):
        v = int(a[b, 4 + w]); hw = v & 0xffffffff; xcc = (v >> 32) & 0xf
        wave, simd, cu, sh, se = hw & 15, (hw >> 4) & 3, (hw >> 8) & 15, (hw >> 12) & 1, (hw >> 13) & 7
        per_simd[(xcc, se, sh, cu, simd)].append((w, wave, b))
        ids.append((simd, wave))
    pairs[(ids[0], ids[1])] += 1
hist = collections.Counter()
for k, v in per_simd.items():
    hist[tuple(sorted(r for r, _, _ in v))] += 1
print("SIMDs by resident roles (0 = wave 0 = actor, 1 = critic):", dict(hist))
print("most common (simd, slot) of (wave 0, wave 1):", pairs.most_common(12))
cu0 = sorted((k, v) for k, v in per_simd.items() if k[:4] == (0, 0, 0, 0))
print("one CU:", cu0)
