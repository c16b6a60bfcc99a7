#!/usr/bin/env python3
"""Diagnostic (-DPER_STAMPS build via MIRL_SO): the phases of per_sample_kernel (per.py's sampler: one workgroup, batch draws) on the 100 MHz wall clock, thread 0.
  make -C deep_rl_amd/csrc OBJD=build_perst OUT=../libmirl_perst.so EXTRA=-DPER_STAMPS && MIRL_SO=deep_rl_amd/libmirl_perst.so python tools/per_sample_stamps.py"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import deep_rl_amd as D
from deep_rl_amd import _native as N
dev = torch.device("cuda", 0)
env = D.make("CartPole-v1", num_envs=4096, device=dev, seed=1)
torch.manual_seed(1)
q = D.QNetwork(env); t = D.QNetwork(env); t.load_state_dict(q.state_dict())
eng = D.PERDQNEngine(env, q, t, D.ClipAdam(q, lr=2.5e-4, eps=1e-8), slots=256, batch_size=128, learning_starts=100, total_timesteps=10 * 600)
eng.reset()
f = N.lib().mi_debug_per_sample_marks; f.argtypes = [C.c_void_p]; f.restype = C.c_int
names = ["entry", "level 1 staged + totals", "keyed draw (Philox)", "level-1 walk (<= 256 steps)", "level-0 sums: round trip + 63 steps", "priorities: round trip + 63 steps",
         "zero-skip loop + idx store", "weights (2 pow)", "max + normalise"]
acc = []
for it in range(400):
    eng.act(10); eng.train_step()
    if it >= 300 and it % 10 == 0:
        torch.cuda.synchronize()
        mk = (C.c_ulonglong * 16)()
        assert f(mk) == 0
        acc.append(np.array(mk, dtype=np.float64)[:9] / 100.0)
m = np.stack(acc); rel = m - m[:, :1]
print("per_sample_kernel, batch 128, thread 0: us since entry (mean over %d launches), step since the previous mark" % len(acc))
prev = 0.0
for k, nm in enumerate(names):
    v = rel[:, k].mean(); print("  %-44s %6.2f   +%.2f" % (nm, v, v - prev)); prev = v
