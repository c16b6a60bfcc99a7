#!/usr/bin/env python3
"""Every hot launch seen from INSIDE (-DMI_INSIDE build via MIRL_SO): per tagged kernel, from the 100 MHz wall clock its waves store at entry and exit and from one-wave
marker launches in front of and behind it —  marker -> first wave's entry | first entry -> last wave's exit | last exit -> marker behind it.  The third column is time a
launch stays open after its last wave has left (memory-side atomics, write-backs): rocprof, the counters and the stamp builds all charge it to "the kernel".
  make -C deep_rl_amd/csrc OBJD=build_inside OUT=../libmirl_inside.so EXTRA=-DMI_INSIDE && MIRL_SO=deep_rl_amd/libmirl_inside.so python tools/inside_view.py"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import deep_rl_amd as D
from deep_rl_amd import _native as N

dev = torch.device("cuda", 0)
L = N.lib()
TAGS = N.PROF_TAGS if hasattr(N, "PROF_TAGS") else ("rollout", "gae", "grad", "reduce", "clip_adam", "stats", "dqn_act", "dqn_td", "dqn_reduce", "per", "sac_act", "sac_critic",
                                                    "sac_actor", "sac_gemm", "sac_assemble", "sac_logp", "comm_grad", "comm_stats")
SLOTS = 16384


def read(tu):
    f = getattr(L, "mi_debug_inside_" + tu); f.argtypes = [C.c_void_p, C.c_int]; f.restype = C.c_int
    buf = np.zeros((18, 2, SLOTS), dtype=np.uint64)
    assert f(buf.ctypes.data, 0) == 0
    return buf


def clear(tu):
    f = getattr(L, "mi_debug_inside_" + tu); f.argtypes = [C.c_void_p, C.c_int]; f.restype = C.c_int
    assert f(None, 1) == 0


def markers():
    f = L.mi_debug_inside_markers; f.argtypes = [C.c_void_p]; f.restype = C.c_int
    buf = np.zeros((18, 2, 2), dtype=np.uint64)
    assert f(buf.ctypes.data) == 0
    return buf


def report(title, tus, tags):
    torch.cuda.synchronize()
    mk = markers()
    print(title)
    print("  %-12s %6s | %-22s | %-22s | %-22s | %s" % ("launch", "waves", "marker -> first entry", "first entry -> last exit", "last exit -> marker", "marker to marker"))
    for tu in tus:
        m = read(tu)
        for tag in tags:
            t = TAGS.index(tag)
            b1, e0 = int(mk[t, 0, 1]), int(mk[t, 1, 0])
            ent, ext = m[t, 0].astype(np.int64), m[t, 1].astype(np.int64)
            ok = (ent >= b1) & (ext >= ent) & (ext <= e0)      # the waves of the LAST launch of this tag (slots of an earlier, larger launch are older than its marker)
            if not ok.any():
                continue
            first, last = ent[ok].min(), ext[ok].max()
            print("  %-12s %6d | %19.2f us | %19.2f us | %19.2f us | %7.2f us   (entries within %.2f us)" % (
                tag, int(ok.sum()), (first - b1) / 100.0, (last - first) / 100.0, (e0 - last) / 100.0, (e0 - b1) / 100.0, (ent[ok].max() - first) / 100.0))


# ---- PPO headline shape ----
env = D.make("CartPole-v1", num_envs=4096, device=dev, seed=1)
torch.manual_seed(1)
agent = D.ActorCritic(env); opt = D.ClipAdam(agent, lr=2.5e-4, eps=1e-5, max_grad_norm=0.5)
eng = D.PPOEngine(env, agent, opt, num_steps=128, n_minibatch=4, update_epochs=4); eng.reset()
for tu in ("update", "rollout"): clear(tu)
for _ in range(4): eng.update()
report("ppo.py, 4096 envs x 128 steps: the LAST launch of each kind in an update", ("rollout", "update"), ("rollout", "stats", "grad", "reduce", "clip_adam"))
del eng, env

# ---- DQN config 3 ----
env = D.make("CartPole-v1", num_envs=4096, device=dev, seed=1)
torch.manual_seed(1)
q = D.QNetwork(env); t = D.QNetwork(env); t.load_state_dict(q.state_dict())
de = D.DQNEngine(env, q, t, D.ClipAdam(q, lr=2.5e-4, eps=1e-8), slots=256, batch_size=int(os.environ.get("BATCH", "128")), learning_starts=100, total_timesteps=5000)
de.reset(); clear("dqn")
for _ in range(60): de.act(10); de.train_step()
report("dqn.py, 4096 envs, batch %d" % de.batch_size, ("dqn",), ("dqn_act", "dqn_td", "dqn_reduce"))
del de, env

# ---- SAC config 4 ----
env = D.make("Pendulum-v1", num_envs=2048, device=dev, seed=1)
torch.manual_seed(1)
actor = D.Actor(env); qs = [D.SoftQNetwork(env) for _ in range(4)]
qs[2].load_state_dict(qs[0].state_dict()); qs[3].load_state_dict(qs[1].state_dict())
se = D.SACEngine(env, actor, *qs, slots=512, batch_size=int(os.environ.get("SAC_BATCH", "256")), learning_starts=20, max_episodes_logged=0)
se.reset(); clear("sac")
for _ in range(61):
    se.act()
    if se.global_step >= se.learning_starts: se.train_step()
se.flush()
report("sac.py, 2048 envs, batch %d" % se.batch_size, ("sac",), ("sac_act", "sac_critic", "sac_actor", "sac_gemm"))
