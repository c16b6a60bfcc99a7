"""Worker of tests/test_gpu_p2p.py — one rank of a world_size-2 job started by torch.distributed.run, BOTH ranks on cuda:0 (or one per GPU when there are two),
process group gloo (it only carries the IPC handles and the test's own comparisons), MIRL_COMM=p2p.

The raw collective of libmirl's P2P carrier (csrc/mi_comm.hip: mi_comm_p2p_alloc / _connect, mi_comm_allreduce_sum): the exchange that stands between backward and
the optimizer step in the sharded form of reference ppo.py:189-192 / dqn.py:131-133 / sac.py:185-210.
  * f32 and f64 messages of the sizes the engines send (48 doubles, 9,159 / 10,936 / 134,660 floats, 1 float) and ragged / unaligned ones, 40 back-to-back rounds each
    (both parities, no host synchronisation in between): bitwise the gloo SUM all-reduce of the same data and bitwise the other rank's result;
  * a peer that never arrives: the wait runs out (MIRL_P2P_TIMEOUT_MS), the buffer keeps the local share, mi_comm_check says which rank was missing."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import deep_rl_amd.dist as DD  # noqa: E402
from deep_rl_amd import _native as N  # noqa: E402

assert os.environ.get("MIRL_COMM") == "p2p"
rank, world, local_rank = DD.init_from_env("gloo")
assert world == 2
dev = torch.device("cuda", local_rank if torch.cuda.device_count() >= 2 else 0)
torch.cuda.set_device(dev)
comm = DD.native_comm()
assert comm is not None, "the P2P communicator could not be created (hipIpc between two processes on this device?)"
L = N.lib()
assert L.mi_comm_carrier(comm) == 1
ws, rk, ver, cnt = C.c_int(), C.c_int(), C.c_int(), C.c_int()
N.check(L.mi_comm_info(comm, C.byref(ws), C.byref(rk), C.byref(ver), C.byref(cnt)), "mi_comm_info")
assert (ws.value, rk.value, ver.value, cnt.value) == (2, rank, 0, 2)
s = N.stream_ptr(dev)

ROUNDS = 40
gen = torch.Generator(device="cpu").manual_seed(100 + rank)
for dtype, sizes in ((torch.float32, (1, 3, 9159, 10936, 134660, 1024, 4097)), (torch.float64, (48, 1, 7, 2050))):
    for n in sizes:
        for off in (0, 1):   # off 1: a buffer that is not 16-byte aligned (the scalar instantiation)
            host = torch.randn(ROUNDS, n + off, generator=gen, dtype=dtype) * (10.0 ** torch.randint(-3, 4, (ROUNDS, 1), generator=gen).to(dtype))
            mine = host.to(dev)
            want = mine.clone()
            dist.all_reduce(want)                      # gloo: a + b on both ranks
            got = mine.clone()
            for r in range(ROUNDS):                    # back to back on one stream, no host synchronisation
                N.check(L.mi_comm_allreduce_sum(comm, got[r, off:].data_ptr(), n, 0 if dtype == torch.float32 else 1, s), "mi_comm_allreduce_sum")
            torch.cuda.synchronize()
            N.check(L.mi_comm_check(comm), "mi_comm_check")
            assert torch.equal(got[:, off:], want[:, off:]), (dtype, n, off, (got[:, off:] - want[:, off:]).abs().max().item())
            if off:
                assert torch.equal(got[:, 0], mine[:, 0])   # the element in front of the message was not touched
            other = got.clone()
            dist.all_reduce(other)                     # 2 x the result iff both ranks hold the same bits
            assert torch.equal(other[:, off:], got[:, off:] * 2)
torch.cuda.synchronize()
dist.barrier()

# ---- epoch change: both ranks preset the sequence number 3 below the last one of the epoch and run 8 mixed exchanges across it (twice: the second epoch change finds
#      the barrier lines of the first) ----
SEQ_LAST = 0xFFFFFFF0
for rep in range(2):
    N.check(L.mi_comm_test_set_seq(comm, SEQ_LAST - 3), "mi_comm_test_set_seq")
    for k, (dtype, n) in enumerate([(torch.float32, 134660), (torch.float32, 9159), (torch.float64, 48), (torch.float32, 9159), (torch.float32, 134660), (torch.float64, 48),
                                    (torch.float32, 9159), (torch.float32, 1)]):
        mine = (torch.randn(n, generator=gen, dtype=dtype) * 3).to(dev)
        want = mine.clone()
        dist.all_reduce(want)
        got = mine.clone()
        N.check(L.mi_comm_allreduce_sum(comm, got.data_ptr(), n, 0 if dtype == torch.float32 else 1, s), "mi_comm_allreduce_sum")
        torch.cuda.synchronize()
        N.check(L.mi_comm_check(comm), "mi_comm_check")
        N.check(L.mi_comm_poll(comm), "mi_comm_poll")
        assert torch.equal(got, want), ("epoch change", rep, k, dtype, n, (got - want).abs().max().item())
torch.cuda.synchronize()
dist.barrier()

# ---- the form of PPO's gradient exchange is ONE decision per communicator: ranks that disagree on MIRL_P2P_FUSED get no communicator at all (the in-launch
#      form publishes slab order, the stand-alone one parameter order, under the same sequence number: a mix would sum permuted elements with every wait satisfied) ----
os.environ["MIRL_P2P_FUSED"] = str(rank)
h2, ok2, err2 = DD._create_p2p(None)
assert ok2 == 0 and "MIRL_P2P_FUSED differs across the ranks" in err2, (ok2, err2)
dist.barrier()
L.mi_comm_destroy(h2)
os.environ["MIRL_P2P_FUSED"] = "0"          # the same on both ranks: created, and the stand-alone launch is fixed for it
h2, ok2, err2 = DD._create_p2p(None)
assert ok2 == 1, err2
y = torch.full((9159,), float(rank + 1), device=dev)
N.check(L.mi_comm_allreduce_sum(h2, y.data_ptr(), y.numel(), 0, s), "mi_comm_allreduce_sum")
torch.cuda.synchronize()
assert bool((y == 3.0).all()) and L.mi_comm_check(h2) == 0
dist.barrier()
L.mi_comm_destroy(h2)
del os.environ["MIRL_P2P_FUSED"]
dist.barrier()

# a message larger than the slots is refused up front
big = torch.zeros((1 << 20) // 4 + 64, dtype=torch.float32, device=dev)
assert L.mi_comm_allreduce_sum(comm, big.data_ptr(), big.numel(), 0, s) == -1 and b"does not fit" in L.mi_last_error()

# a peer that never arrives: rank 0 enqueues an all-reduce that rank 1 does not
os.environ["MIRL_P2P_TIMEOUT_MS"] = "300"
h, mine = C.c_void_p(), (C.c_char * 64)()
N.check(L.mi_comm_p2p_alloc(2, rank, 4096, C.byref(h), mine), "mi_comm_p2p_alloc")
boxes = [None, None]
dist.all_gather_object(boxes, bytes(mine.raw))
N.check(L.mi_comm_p2p_connect(h, b"".join(boxes)), "mi_comm_p2p_connect")
x = torch.arange(100, dtype=torch.float32, device=dev) + 1
if rank == 0:
    N.check(L.mi_comm_allreduce_sum(h, x.data_ptr(), 100, 0, s), "mi_comm_allreduce_sum")
    torch.cuda.synchronize()
    assert L.mi_comm_check(h) == -4 and b"never arrived: 1" in L.mi_last_error(), L.mi_last_error()
    assert L.mi_comm_poll(h) == -4 and b"WITHHELD" in L.mi_last_error(), L.mi_last_error()   # the same answer without a sync (host-pinned mirror)
    assert torch.equal(x, torch.arange(100, dtype=torch.float32, device=dev) + 1)      # the local share, untouched
    assert L.mi_comm_allreduce_sum(h, x.data_ptr(), 100, 0, s) == -4                    # later calls are refused at their entry
else:
    assert L.mi_comm_check(h) == 0 and L.mi_comm_poll(h) == 0
dist.barrier()
L.mi_comm_destroy(h)


# ---- the fail-safe through the engines: rank 0 runs an update on a communicator whose peer never calls ----
def lonely_comm(nbytes):
    hh, me = C.c_void_p(), (C.c_char * 64)()
    N.check(L.mi_comm_p2p_alloc(2, rank, nbytes, C.byref(hh), me), "mi_comm_p2p_alloc")
    bx = [None, None]
    dist.all_gather_object(bx, bytes(me.raw))
    N.check(L.mi_comm_p2p_connect(hh, b"".join(bx)), "mi_comm_p2p_connect")
    return hh


import deep_rl_amd as D  # noqa: E402
import deep_rl_amd.engine as E  # noqa: E402

assert "MIRL_CHECK_REPLICAS" not in os.environ
h = lonely_comm(1 << 16)
if rank == 0:
    env = D.make("CartPole-v1", num_envs=64, device=dev, seed=5)
    torch.manual_seed(5)
    agent = D.ActorCritic(env)
    opt = D.ClipAdam(agent, lr=2.5e-4, eps=1e-5, max_grad_norm=0.5)
    eng = D.PPOEngine(env, agent, opt, num_steps=128)
    eng.reset()
    before = [t.clone() for t in (agent.flat, opt.exp_avg, opt.exp_avg_sq)]
    DD.use_comm(h)
    try:
        eng.update()                       # every exchange of it times out (300 ms the first, 64 polls the others): all 16 optimizer steps are withheld
        torch.cuda.synchronize()
        for a, b in zip(before, (agent.flat, opt.exp_avg, opt.exp_avg_sq)):
            assert torch.equal(a, b), "a timed-out exchange cost the optimizer state"
        assert not torch.equal(eng.observations[1], eng.observations[0])   # (the rollout itself ran)
        try:
            eng.update()
            raise AssertionError("the update behind a timed-out exchange did not raise")
        except N.MiError as ex:
            assert "never arrived: 1" in str(ex) and "WITHHELD" in str(ex), str(ex)
        try:
            pinned = torch.zeros(4, dtype=torch.int32).pin_memory()
            eng.episode_summary_async(pinned)
            raise AssertionError("episode_summary_async did not poll the carrier")
        except N.MiError:
            pass
        for a, b in zip(before, (agent.flat, opt.exp_avg, opt.exp_avg_sq)):
            assert torch.equal(a, b)
    finally:
        DD.use_comm(None)
dist.barrier()
L.mi_comm_destroy(h)

h = lonely_comm(1 << 16)
if rank == 0:
    env = D.make("CartPole-v1", num_envs=64, device=dev, seed=6)
    torch.manual_seed(6)
    q = D.QNetwork(env); tq = D.QNetwork(env); tq.load_state_dict(q.state_dict())
    opt = D.ClipAdam(q, lr=2.5e-4, eps=1e-8)
    eng = D.DQNEngine(env, q, tq, opt, slots=64, batch_size=128, learning_starts=10, total_timesteps=1000, max_episodes_logged=0)
    eng.reset()
    eng.act(20)
    before = [t.clone() for t in (q.flat, opt.exp_avg, opt.exp_avg_sq)]
    DD.use_comm(h)
    try:
        eng.train_step()
        torch.cuda.synchronize()
        for a, b in zip(before, (q.flat, opt.exp_avg, opt.exp_avg_sq)):
            assert torch.equal(a, b), "DQN: a timed-out exchange cost the optimizer state"
        try:
            eng.train_step()
            raise AssertionError("DQN: the step behind a timed-out exchange did not raise")
        except N.MiError as ex:
            assert "never arrived: 1" in str(ex), str(ex)
    finally:
        DD.use_comm(None)
dist.barrier()
L.mi_comm_destroy(h)

# SAC: the three sharded routes carry their exchange inside the gradient assembly launches (round 6): a peer that never arrives leaves critics, targets, actor, log_alpha
# and every moment as they were, and the next call is refused
h = lonely_comm(1 << 20)
if rank == 0:
    env = D.make("Pendulum-v1", num_envs=16, device=dev, seed=8)
    torch.manual_seed(8)
    a = D.Actor(env)
    qs = [D.SoftQNetwork(env) for _ in range(4)]
    qs[2].load_state_dict(qs[0].state_dict()); qs[3].load_state_dict(qs[1].state_dict())
    sac = D.SACEngine(env, a, *qs, slots=64, batch_size=64, learning_starts=5, max_episodes_logged=0)
    assert sac.world_size == 2
    sac.reset()
    for _ in range(20):
        sac.act()
    DD.use_comm(h)
    try:
        sac.sample()
        tensors = lambda: (sac.q_flat, sac.qt_flat, sac.actor.flat, sac.log_alpha, sac.q_optimizer.exp_avg, sac.q_optimizer.exp_avg_sq, sac.actor_optimizer.exp_avg)  # noqa: E731
        before = [t.clone() for t in tensors()]
        sac.update_critic(polyak=True)       # its exchange times out (300 ms): Adam and polyak are withheld ...
        torch.cuda.synchronize()
        for x, y in zip(before, tensors()):
            assert torch.equal(x, y), "SAC: a timed-out exchange cost the optimizer state"
        for call in (sac.update_actor, sac.update_alpha, lambda: sac.update_critic(polyak=True)):   # ... and every later sharded call is refused at its entry
            try:
                call()
                raise AssertionError("SAC: a call behind a timed-out exchange did not raise")
            except N.MiError as ex:
                assert "never arrived: 1" in str(ex), str(ex)
        for x, y in zip(before, tensors()):
            assert torch.equal(x, y)
    finally:
        DD.use_comm(None)
dist.barrier()
L.mi_comm_destroy(h)

dist.barrier()
DD.destroy_native_comms()
dist.destroy_process_group()
if rank == 0:
    print("P2P_WORKER_OK")
