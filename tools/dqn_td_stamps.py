#!/usr/bin/env python3
"""Phase marks of the one-launch TD update (build with EXTRA=-DTD_STAMPS, MIRL_SO=that build): per workgroup, us from the earliest entry of the launch."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import deep_rl_amd as D
from deep_rl_amd import _native as N
dev = torch.device("cuda", 0)
env = D.make("CartPole-v1", num_envs=4096, device=dev, seed=1); torch.manual_seed(1)
q = D.QNetwork(env); t = D.QNetwork(env); t.load_state_dict(q.state_dict())
eng = D.DQNEngine(env, q, t, D.ClipAdam(q, lr=2.5e-4, eps=1e-8), slots=256, batch_size=int(os.environ.get("BATCH", "128")), learning_starts=100, total_timesteps=100000)
eng.reset()
for _ in range(100): eng.act(10); eng.train_step()
torch.cuda.synchronize()
L = C.CDLL(N.SO_PATH)
buf = (C.c_ulonglong * (32 * 8))()
assert L.mi_debug_td_stamps(buf) == 0
m = np.array(buf, dtype=np.uint64).reshape(32, 8).astype(np.int64)[: (eng.batch_size + 7) // 8, :5]
m = (m - m[:, 0].min()) / 100.0
print("us since the first entry; columns: entry, main loop done, slab stored (issued), first gather complete, exit")
for b, row in enumerate(m): print("wg %2d " % b + " ".join("%7.2f" % x for x in row))
