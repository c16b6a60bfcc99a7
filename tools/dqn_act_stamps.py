#!/usr/bin/env python3
"""Diagnostic (-DDA_STAMPS build via MIRL_SO): where a forward wave and the dynamics wave of one workgroup of dqn_act4_kernel spend a step (s_memtime ticks;
the whole loop of the launch is the last column, so the columns are read as fractions of it).
  make -C deep_rl_amd/csrc OBJD=build_stamps OUT=../libmirl_stamps.so EXTRA=-DDA_STAMPS && MIRL_SO=deep_rl_amd/libmirl_stamps.so python tools/dqn_act_stamps.py"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import deep_rl_amd as D
from deep_rl_amd import _native as N
dev = torch.device("cuda", 0)
env = D.make("CartPole-v1", num_envs=4096, device=dev, seed=1)
torch.manual_seed(1)
q = D.QNetwork(env); t = D.QNetwork(env); t.load_state_dict(q.state_dict())
eng = D.DQNEngine(env, q, t, D.ClipAdam(q, lr=2.5e-4, eps=1e-8), slots=256, batch_size=128, learning_starts=0, total_timesteps=10 * 500)   # as tools/dqn_kernel_times.py
eng.reset()
f = N.lib().mi_debug_dqn_act_stamps; f.argtypes = [C.c_void_p]; f.restype = C.c_int
buf = (C.c_ulonglong * 16)()
fm = N.lib().mi_debug_dqn_act_marks; fm.argtypes = [C.c_void_p]; fm.restype = C.c_int
mk = (C.c_ulonglong * 8192)()
names = ["pre: forward (L1, L2, head, qp)  |  commit of the previous step", "dynamics, both successors + records", "exploration draw of the next step", "barrier",
         "post: action, observation of the chosen record"]
for it in range(60):
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    t0.record(); eng.act(10); t1.record(); torch.cuda.synchronize()
    assert f(buf) == 0
    a = np.array(buf, dtype=np.float64).reshape(2, 8)
    if it < 57: continue
    print("launch of 10 steps: %.1f us by events" % (t0.elapsed_time(t1) * 1e3))
    for wv, nm in ((0, "forward wave 0"), (1, "dynamics wave")):
        tot = a[wv, 5]
        print("  %s: loop = %.0f ticks = %.2f us by the 100 MHz counter (%.0f ticks per us)" % (nm, tot, a[wv, 7] / 100.0, tot / max(a[wv, 7] / 100.0, 1e-9)))
        for k in range(5): print("    %-70s %6.1f %%  (%.0f ticks per step)" % (names[k], 100 * a[wv, k] / tot, a[wv, k] / a[wv, 6]))
# the config-3 loop without a host sync: launch slots (global_step / 10) & 7 keep the marks of the last 8 acting launches
for _ in range(64):
    eng.act(10); eng.train_step()
torch.cuda.synchronize()
mk = (C.c_ulonglong * (8 * 2 * 4 * 256))()
assert fm(mk) == 0
m = np.array(mk, dtype=np.float64).reshape(8, 2, 4, 256) / 100.0   # us
np.save(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "dqn_act_marks.npy"), m)
for L in range(8):
    x = m[L] - m[L][:, 0].min()
    print("launch slot %d: entry by XCD (workgroup %% 8): %s | loop start mean %.2f | last exit %.2f us" % (
        L, " ".join("%.2f" % x[0, 0][k::8].mean() for k in range(8)), x[0, 1].mean(), x[:, 3].max()))
# back-to-back acting launches only: the time between the last exit of one launch and the first entry of the next, by the same 100 MHz counter
for _ in range(64): eng.act(10)
torch.cuda.synchronize()
assert fm(mk) == 0
m = np.array(mk, dtype=np.float64).reshape(8, 2, 4, 256) / 100.0
order = np.argsort(m[:, 0, 0].min(axis=1))
for a, b in zip(order[:-1], order[1:]):
    print("acting launch -> acting launch: span %.2f us, then %.2f us until the first entry of the next one (period %.2f)" % (
        m[a][:, 3].max() - m[a][:, 0].min(), m[b][:, 0].min() - m[a][:, 3].max(), m[b][:, 0].min() - m[a][:, 0].min()))
# the same with a one-wave marker launch between two acting launches: how much of the boundary is the END of the acting launch, how much the START of the next
ft = N.lib().mi_debug_tiny_mark; ft.argtypes = [C.c_int, C.c_void_p]; ft.restype = C.c_int
fr = N.lib().mi_debug_tiny_read; fr.argtypes = [C.c_void_p]; fr.restype = C.c_int
for i in range(64):
    eng.act(10); assert ft(eng.global_step // 10, N.stream_ptr(dev)) == 0
torch.cuda.synchronize()
tb = (C.c_ulonglong * 128)()
assert fm(mk) == 0 and fr(tb) == 0
m = np.array(mk, dtype=np.float64).reshape(8, 2, 4, 256) / 100.0
tm = np.array(tb, dtype=np.float64).reshape(64, 2) / 100.0
gs = eng.global_step // 10
for L in range(gs - 6, gs):          # acting launch L ran in slot (L - 1) & 7 (its global_step / 10 before the call), the marker behind it carries index L
    a, b = m[(L - 1) & 7], m[L & 7]
    print("acting launch (span %.2f us) -> %.2f us -> marker launch (%.2f us) -> %.2f us -> first entry of the next acting launch" % (
        a[:, 3].max() - a[:, 0].min(), tm[L & 63, 0] - a[:, 3].max(), tm[L & 63, 1] - tm[L & 63, 0], b[:, 0].min() - tm[L & 63, 1]))
