#!/usr/bin/env python3
"""Diagnostic (-DRQ_STAMPS build via MIRL_SO): wall-clock time line of one workgroup of rollout_q4_kernel — when its actor and critic waves finish each step."""
import ctypes as C, os, sys
sys.path.insert(0, "/root/repo")
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
a = np.array(buf, dtype=np.float64)
t0 = a[8192 + 600]
act = (a[8192:8192 + 129] - t0) / 100.0; cri = (a[8192 + 256:8192 + 256 + 129] - t0) / 100.0
print("actor step-end times (us since kernel entry), steps 0..15:", np.round(act[:16], 2))
print("critic step-end times, steps 0..15:", np.round(cri[:16], 2))
print("actor step durations: first 10 %s ... mean of 20..128 %.3f us" % (np.round(np.diff(act[:11]), 2), np.diff(act[20:129]).mean()))
print("critic step durations: first 10 %s ... mean of 20..128 %.3f us" % (np.round(np.diff(cri[:11]), 2), np.diff(cri[20:129]).mean()))
print("actor end %.1f us, critic end %.1f us" % (act[128], cri[128]))
m = lambda k: (a[8192 + k] - t0) / 100.0  # noqa: E731
print("marks (us since kernel entry): actor W2 in registers %.2f | critic loop done %.1f | last rows flushed %.1f | first GAE block scanned %.1f, its rows stored %.1f | GAE done %.1f" % (
    m(601), m(703), m(700), m(705), m(706), m(702)))
