#!/usr/bin/env python3
"""The two SAC update kernels from the inside (build with EXTRA=-DSAC_MARKS, MIRL_SO=that build): wall-clock marks of thread 0 of row group 0's workgroups at the phase
boundaries, us since the launch's first mark, config-4 loop (2048 envs, batch 256).  critic kernel (quad: roles 0 / 1 = target critics, 2 / 3 = critics):
  0 entry  1 rows in LDS  2 actor' forward done  3 target forward done  4 words published | 5 critic role starts  6 critic forward done  7 TD target taken  8 dz2 + thin done  9 dh1 pass done  10 end
actor kernel (role 0 = critic 2's sibling, 1 = main): 0 entry  1 rows in LDS  2 actor forward done  3 critic forward done  4 critic backward done  5 hand-off done  6 thin actor gradients done
  7 actor backward pass done  8 end"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import deep_rl_amd as D
from deep_rl_amd import _native as N
dev = torch.device("cuda", 0)
env = D.make("Pendulum-v1", num_envs=2048, device=dev, seed=1); torch.manual_seed(1)
a = D.Actor(env); qs = [D.SoftQNetwork(env) for _ in range(4)]
eng = D.SACEngine(env, a, *qs, slots=512, batch_size=int(os.environ.get("BATCH", "256")), learning_starts=64, max_episodes_logged=0)
eng.reset()
for _ in range(300):
    eng.act()
    if eng.global_step >= 64: eng.train_step(2, 1)
eng.flush(); torch.cuda.synchronize()
L = C.CDLL(N.SO_PATH)
buf = (C.c_ulonglong * (2 * 4 * 16))()
assert L.mi_debug_sac_marks(buf) == 0
m = np.array(buf, dtype=np.uint64).reshape(2, 4, 16).astype(np.int64)
for k, name in enumerate(("sac_critic_kernel", "sac_actor_kernel")):
    t0 = m[k][m[k] > 0].min()
    print(name)
    for role in range(4):
        row = m[k, role]
        if (row > 0).any():
            print("  role %d: " % role + "  ".join("%d:%.2f" % (i, (v - t0) / 100.0) for i, v in enumerate(row) if v > 0))

fb = (C.c_ulonglong * (4 * 64 * 2))()
if hasattr(L, "mi_debug_sac_fine") and L.mi_debug_sac_fine(fb) == 0:
    f = np.array(fb, dtype=np.uint64).reshape(4, 64, 2).astype(np.int64)
    tags = {1: "layer1", 2: "pass>", 3: "pass<", 4: "bias+head", 5: "combine", 6: "q-bias", 7: "row-scalars"}
    for slot, name in enumerate(("critic kernel, target role 0", "critic kernel, critic role 2", "actor kernel, sibling", "actor kernel, main")):
        rows = [(int(a), int(b)) for a, b in f[slot] if b > 0]
        if rows:
            t0 = rows[0][1]
            print("fine trace, %s (us since its first block end; tag = block that just ended):" % name)
            print("   " + "  ".join("%s:%.2f" % (tags.get(a, str(a)), (b - t0) / 100.0) for a, b in rows))
