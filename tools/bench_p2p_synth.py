#!/usr/bin/env python3
"""The P2P carrier's one-launch all-reduce on ONE GPU with `world` synthetic ranks (mi_comm_p2p_synthetic: stores, polls and the rank-ordered sum of `world` slots, no
link traffic): back-to-back latency per message size, and what 17 of them per update cost mi_ppo_update_sharded (4096 envs) against the plain update."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import deep_rl_amd as D
import deep_rl_amd.dist as DD
import deep_rl_amd.engine as E
from deep_rl_amd import _native as N
dev = torch.device("cuda", 0)
L, s = N.lib(), N.stream_ptr(dev)
def b2b(h, n, dtype=0, reps=300):
    buf = torch.zeros(n, dtype=torch.float64 if dtype else torch.float32, device=dev)
    for _ in range(30): N.check(L.mi_comm_allreduce_sum(h, N.ptr(buf), n, dtype, s))
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): N.check(L.mi_comm_allreduce_sum(h, N.ptr(buf), n, dtype, s))
    torch.cuda.synchronize(); return 1e6 * (time.perf_counter() - t0) / reps
env = D.make("CartPole-v1", num_envs=4096, device=dev, seed=1); torch.manual_seed(1)
agent = D.ActorCritic(env); opt = D.ClipAdam(agent, lr=2.5e-4, eps=1e-5, max_grad_norm=0.5)
eng = D.PPOEngine(env, agent, opt, num_steps=128); eng.reset()
def window(n=30):
    for _ in range(3): eng.update()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): eng.update()
    torch.cuda.synchronize(); return 1e3 * (time.perf_counter() - t0) / n
for _ in range(60): eng.update()
base = window()
E.set_assume_sharded(True); assume = window(); E.set_assume_sharded(False)
print("plain update %.4f ms | assume_sharded %.4f ms (+%.1f us per optimizer step)" % (base, assume, 1e3 * (assume - base) / 16))
for w in (1, 2, 4, 8):
    h = C.c_void_p(); N.check(L.mi_comm_p2p_synthetic(w, 1 << 20, C.byref(h)))
    lat = {n: b2b(h, n) for n in (1, 9159, 10936, 67331, 134660)}; lat64 = b2b(h, 48, 1)
    DD.use_comm(h); E._FORCE_NATIVE_SHARDED = True   # (no assume_sharded: on this carrier the slab sum exchanges the gradient itself and the owed step takes the single-rank branch)
    ms = window()
    E._FORCE_NATIVE_SHARDED = False; DD.use_comm(None)
    N.check(L.mi_comm_check(h)); torch.cuda.synchronize(); L.mi_comm_destroy(h)
    print("world %d: back-to-back us per all-reduce %s, 48 f64 %.2f | update %.4f ms = +%.2f us per optimizer step vs plain, +%.2f vs assume_sharded"
          % (w, " ".join("%d:%.2f" % kv for kv in lat.items()), lat64, ms, 1e3 * (ms - base) / 16, 1e3 * (ms - assume) / 16))
