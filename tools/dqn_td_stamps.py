#!/usr/bin/env python3
"""Diagnostic (-DTD_STAMPS build via MIRL_SO): the phases of dqn_td_kernel at the reference batch (128 rows = 16 workgroups of one 8-row group) on the 100 MHz wall clock.
  make -C deep_rl_amd/csrc OBJD=build_tdst OUT=../libmirl_tdst.so EXTRA=-DTD_STAMPS && MIRL_SO=deep_rl_amd/libmirl_tdst.so python tools/dqn_td_stamps.py [batch]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import deep_rl_amd as D
from deep_rl_amd import _native as N
batch = int(sys.argv[1]) if len(sys.argv) > 1 else 128
dev = torch.device("cuda", 0)
env = D.make("CartPole-v1", num_envs=4096, device=dev, seed=1)
torch.manual_seed(1)
q = D.QNetwork(env); t = D.QNetwork(env); t.load_state_dict(q.state_dict())
eng = D.DQNEngine(env, q, t, D.ClipAdam(q, lr=2.5e-4, eps=1e-8), slots=256, batch_size=batch, learning_starts=0, total_timesteps=10 * 500)
eng.reset()
f = N.lib().mi_debug_dqn_td_marks; f.argtypes = [C.c_void_p]; f.restype = C.c_int
names = ["entry", "index derived", "prologue requests issued", "rows in LDS (first barrier)", "layer 1", "layer 2 (MFMA; weight operands landed)", "layer 3 (84-long dots)",
         "TD target + loss", "backward through layer 3", "dh1 + dW2 (MFMA)", "db1 + dW1 = main loop done", "slab stored"]
acc = []
for it in range(300):
    eng.act(10); eng.train_step()
    if it >= 200 and it % 10 == 0:
        torch.cuda.synchronize()
        mk = (C.c_ulonglong * 256)()
        assert f(mk) == 0
        acc.append(np.array(mk, dtype=np.float64).reshape(16, 16)[:, :12] / 100.0)
m = np.stack(acc)                                  # [sample][workgroup][mark] in us
rel = m - m[:, :, :1].min(axis=1, keepdims=True)   # since the first workgroup's entry
print("dqn_td_kernel, batch %d: us since the first workgroup's entry (mean over %d launches; workgroup mean / slowest workgroup), step = mean since the previous mark" % (batch, len(acc)))
prev = None
for k, nm in enumerate(names):
    mean_k, max_k = rel[:, :, k].mean(), rel[:, :, k].max(axis=1).mean()
    print("  %-44s %6.2f / %6.2f   %s" % (nm, mean_k, max_k, "" if prev is None else "+%.2f" % (mean_k - prev)))
    prev = mean_k
