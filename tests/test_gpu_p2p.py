"""The peer-to-peer carrier of libmirl's communicator (csrc/mi_comm.hip; MIRL_COMM=p2p) — and, through it, the production ONE-CALL sharded routes at world_size 2 on
the one GPU this box has (until round 5 mi_ppo_update_sharded, mi_dqn_td_update_sharded, mi_sac_{critic,actor}_update_sharded and
mi_sac_alpha_step_sharded had only ever run at world_size 1, because RCCL refuses two ranks on one device).  The exchange they carry stands between backward and the
optimizer step of reference ppo.py:189-192, dqn.py:131-133 (per.py:147-153), sac.py:185-210.

  test_synthetic_world_*                 one process plays 1 / 2 / 4 / 8 ranks into its own inbox (slot 0 = its share, the others zeros): every message size the
                                         engines send comes back bit for bit, f32 and f64, aligned or not, 50 back-to-back launches (both parities).
  test_ppo_update_on_synthetic_ranks_*   mi_ppo_update_sharded with grad_reduce_kernel's in-launch exchange at 1 .. 8 synthetic ranks == mi_ppo_update, exactly.
  test_two_ranks_one_gpu_p2p_collective  two PROCESSES on cuda:0, inboxes exchanged with hipIpcGetMemHandle / hipIpcOpenMemHandle: bitwise gloo's a + b, rank == rank,
                                         the bounded wait when a peer never arrives (tests/_p2p_worker.py).
  test_two_ranks_one_gpu_p2p_ppo         PPOEngine.update() on mi_ppo_update_sharded, two ranks on cuda:0, two whole updates: rank == rank bitwise, == the
                                         host-sequenced route over gloo bitwise, == the single process with union minibatches at the tolerances of test_gpu_multigpu.py.
  test_two_ranks_one_gpu_p2p_offpolicy   the same for DQNEngine, PERDQNEngine, SACEngine on their one-call routes (MIRL_CHECK_REPLICAS=2 inside the worker).
  test_two_gpus_p2p_*                    the same two workers with one rank per GPU and an NCCL process group (skip below 2 GPUs).
  test_eight_ranks_one_gpu_p2p_ppo       EIGHT processes on cuda:0 (BASELINE config 5's rank count): the 8-rank exchange between real processes, all ranks bitwise equal."""
import ctypes as C
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from _ports import free_port, run_with_port  # noqa: E402
sys.path.insert(0, os.path.join(ROOT, "tests"))


def _need_gpu():
    import torch

    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    return torch


@pytest.mark.parametrize("world", [1, 2, 4, 8])
def test_synthetic_world_allreduce_is_identity(world):
    torch = _need_gpu()
    from deep_rl_amd import _native as N

    dev = torch.device("cuda", 0)
    L, s = N.lib(), N.stream_ptr(dev)
    h = C.c_void_p()
    N.check(L.mi_comm_p2p_synthetic(world, 1 << 20, C.byref(h)), "mi_comm_p2p_synthetic")
    try:
        assert L.mi_comm_carrier(h) == 1
        ws, rk, ver, cnt = C.c_int(), C.c_int(), C.c_int(), C.c_int()
        N.check(L.mi_comm_info(h, C.byref(ws), C.byref(rk), C.byref(ver), C.byref(cnt)), "mi_comm_info")
        assert (ws.value, rk.value, ver.value, cnt.value) == (1, 0, 0, world)   # ONE rank playing `world` of them
        gen = torch.Generator(device="cpu").manual_seed(world)
        for dtype, sizes in ((torch.float32, (1, 5, 1023, 1024, 1025, 9159, 10936, 67331, 134660, 262144)), (torch.float64, (1, 48, 513, 131072))):
            for n in sizes:
                for off in (0, 1):
                    x = (torch.randn(n + off + 3, generator=gen, dtype=dtype) * 100).to(dev)
                    x[x == 0] = 1.0                   # x + 0 == x bit for bit except for -0
                    y = x.clone()
                    for _ in range(50):
                        N.check(L.mi_comm_allreduce_sum(h, y[off:].data_ptr(), n, 0 if dtype == torch.float32 else 1, s), "mi_comm_allreduce_sum")
                    torch.cuda.synchronize()
                    assert torch.equal(x, y), (dtype, n, off)
        N.check(L.mi_comm_check(h), "mi_comm_check")
        big = torch.zeros((1 << 20) // 4 + 4, dtype=torch.float32, device=dev)
        assert L.mi_comm_allreduce_sum(h, big.data_ptr(), big.numel(), 0, s) == -1
    finally:
        L.mi_comm_destroy(h)


@pytest.mark.parametrize("world", [1, 2, 3, 5, 8])
def test_synthetic_world_epoch_change(world):
    """The 32-bit sequence number runs out (round 5 wrapped 0xFFFFFFFF -> 1, two consecutive all-reduces on one parity, stale lines with a "right"
    number).  Preset it 3 below the epoch's last number and run exchanges of MIXED sizes across the change: a long message first, so that lines of the old epoch sit
    behind the end of the short ones that follow; the same numbers come round again after the change (1, 2, ...) and must not be mistaken for arrivals — the inbox is
    cleared under a barrier.  x + 0 + ... + 0 in rank order == x, bit for bit; both the stand-alone launch and grad_reduce_kernel's in-launch exchange cross it."""
    torch = _need_gpu()
    from deep_rl_amd import _native as N

    dev = torch.device("cuda", 0)
    L, s = N.lib(), N.stream_ptr(dev)
    h = C.c_void_p()
    N.check(L.mi_comm_p2p_synthetic(world, 1 << 20, C.byref(h)), "mi_comm_p2p_synthetic")
    try:
        gen = torch.Generator(device="cpu").manual_seed(7 * world)
        # seed the slots of BOTH parities with sequence numbers 1 .. 6 on long messages: what a fresh epoch will count through again
        for _ in range(6):
            y = torch.ones(262144, dtype=torch.float32, device=dev)
            N.check(L.mi_comm_allreduce_sum(h, y.data_ptr(), y.numel(), 0, s), "mi_comm_allreduce_sum")
        for rep in range(2):
            N.check(L.mi_comm_test_set_seq(h, 0xFFFFFFF0 - 3), "mi_comm_test_set_seq")
            for dtype, n in ((torch.float32, 262144), (torch.float32, 9159), (torch.float64, 48), (torch.float32, 9159), (torch.float32, 134660), (torch.float64, 48),
                             (torch.float32, 262144), (torch.float32, 3)):
                x = (torch.randn(n, generator=gen, dtype=dtype) * 10).to(dev)
                x[x == 0] = 1.0
                y = x.clone()
                N.check(L.mi_comm_allreduce_sum(h, y.data_ptr(), n, 0 if dtype == torch.float32 else 1, s), "mi_comm_allreduce_sum")
                torch.cuda.synchronize()
                assert torch.equal(x, y), (world, rep, dtype, n)
            N.check(L.mi_comm_check(h), "mi_comm_check")
            N.check(L.mi_comm_poll(h), "mi_comm_poll")
    finally:
        L.mi_comm_destroy(h)


def test_ppo_update_on_synthetic_ranks_crosses_the_epoch_change():
    """mi_ppo_update_sharded's in-launch exchange (grad_reduce_kernel draws its sequence numbers through mi_comm_p2p_next like the stand-alone launch) across the epoch
    change: 2 updates x 17 exchanges with the number preset 20 below the last one == mi_ppo_update, exactly."""
    torch = _need_gpu()
    import deep_rl_amd as D
    import deep_rl_amd.dist as DD
    import deep_rl_amd.engine as E
    from deep_rl_amd import _native as N

    dev = torch.device("cuda", 0)
    out = []
    for synthetic in (False, True):
        h = C.c_void_p()
        if synthetic:
            N.check(N.lib().mi_comm_p2p_synthetic(4, 1 << 16, C.byref(h)), "mi_comm_p2p_synthetic")
            N.check(N.lib().mi_comm_test_set_seq(h, 0xFFFFFFF0 - 20), "mi_comm_test_set_seq")
            DD.use_comm(h)
            E._FORCE_NATIVE_SHARDED = True
        try:
            env = D.make("CartPole-v1", num_envs=64, device=dev, seed=22)
            torch.manual_seed(22)
            agent = D.ActorCritic(env)
            opt = D.ClipAdam(agent, lr=2.5e-4, eps=1e-5, max_grad_norm=0.5)
            eng = D.PPOEngine(env, agent, opt, num_steps=128)
            eng.reset()
            for _ in range(2):
                eng.update()
            torch.cuda.synchronize()
            if synthetic:
                N.check(N.lib().mi_comm_check(h), "mi_comm_check")
            out.append([t.clone() for t in (agent.flat, opt.exp_avg, opt.exp_avg_sq, eng.grads, eng.loss_terms, opt.grad_norm)])
        finally:
            E._FORCE_NATIVE_SHARDED = False
            DD.use_comm(None)
            if h.value:
                torch.cuda.synchronize()
                N.lib().mi_comm_destroy(h)
    for a, b in zip(*out):
        assert torch.equal(a, b)


@pytest.mark.parametrize("world", [1, 2, 3, 4, 5, 6, 7, 8])
def test_ppo_update_on_synthetic_ranks_equals_plain_update(world):
    """mi_ppo_update_sharded on the P2P carrier with `world` synthetic ranks (slot 0 = this process's share, the others zeros) against mi_ppo_update: grad_reduce_kernel's
    in-launch exchange — every instantiation 1 .. 8 of it — must return each gradient element and loss term unchanged (x + 0 + ... + 0 in rank order), so parameters, both
    Adam moments, the last gradient, the loss terms and the clip norm agree exactly after 2 updates x 16 optimizer steps, at 64 envs (small grid) and 512 envs (full grid)."""
    torch = _need_gpu()
    import deep_rl_amd as D
    import deep_rl_amd.dist as DD
    import deep_rl_amd.engine as E
    from deep_rl_amd import _native as N

    dev = torch.device("cuda", 0)
    for n_envs in (64, 512):
        out = []
        for synthetic in (False, True):
            h = C.c_void_p()
            if synthetic:
                N.check(N.lib().mi_comm_p2p_synthetic(world, 1 << 16, C.byref(h)), "mi_comm_p2p_synthetic")
                DD.use_comm(h)
                E._FORCE_NATIVE_SHARDED = True
            try:
                env = D.make("CartPole-v1", num_envs=n_envs, device=dev, seed=21)
                torch.manual_seed(21)
                agent = D.ActorCritic(env)
                opt = D.ClipAdam(agent, lr=2.5e-4, eps=1e-5, max_grad_norm=0.5)
                eng = D.PPOEngine(env, agent, opt, num_steps=128)
                eng.reset()
                for _ in range(2):
                    eng.update()
                torch.cuda.synchronize()
                if synthetic:
                    N.check(N.lib().mi_comm_check(h), "mi_comm_check")
                out.append([t.clone() for t in (agent.flat, opt.exp_avg, opt.exp_avg_sq, eng.grads, eng.loss_terms, opt.grad_norm, eng.advantages)])
            finally:
                E._FORCE_NATIVE_SHARDED = False
                DD.use_comm(None)
                if h.value:
                    torch.cuda.synchronize()
                    N.lib().mi_comm_destroy(h)
        for a, b in zip(*out):
            assert torch.equal(a, b), (world, n_envs)
        assert torch.isfinite(out[0][0]).all()


def _launch(worker, env_extra, timeout=600, nproc=2, comm="p2p"):
    env = dict(os.environ, PYTHONPATH=ROOT, HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="1", MIRL_COMM=comm, **env_extra)
    # the ranks are started as children BEFORE anything of theirs touches a GPU (never exec from a process that has initialised HIP); a run that died on EADDRINUSE is
    # repeated on a fresh port (tests/_ports.py)
    return run_with_port(lambda port: ([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr", "127.0.0.1", "--master-port",
                                        str(port), os.path.join(ROOT, "tests", worker)], env), capture_output=True, text=True, timeout=timeout, cwd=ROOT)


def test_two_ranks_one_gpu_p2p_collective():
    _need_gpu()
    out = _launch("_p2p_worker.py", {})
    assert out.returncode == 0, (out.stdout[-2000:], out.stderr[-4000:])
    assert "P2P_WORKER_OK" in out.stdout, out.stdout[-2000:]


def _ppo(backend, comm="p2p", nl=64):
    import tempfile

    import test_gpu_multigpu as M

    with tempfile.TemporaryDirectory() as tmp:
        out = _launch("_sharded_update_worker.py", dict(MIRL_TEST_BACKEND=backend, MIRL_TEST_OUT=tmp, MIRL_TEST_NL=str(nl)), comm=comm)
        assert out.returncode == 0, (out.stdout[-2000:], out.stderr[-4000:])
        assert "SHARDED_WORKER_OK backend=%s native=1 carrier=p2p" % backend in out.stdout, out.stdout[-2000:]
        r0, r1 = dict(np.load(os.path.join(tmp, "rank0.npz"))), dict(np.load(os.path.join(tmp, "rank1.npz")))
    assert int(r0["native"][0]) == 1 and int(r1["native"][0]) == 1     # mi_ppo_update_sharded really was the route
    M._check(r0, r1, M._single_process(r0["params0"], nl), nl)
    # ONE C call with the in-stream peer-to-peer all-reduces == host-sequenced launches with torch.distributed all-reduces in between, bit for bit (a + b on both ranks)
    for rk in (r0, r1):
        for k in ("params", "exp_avg", "exp_avg_sq", "grads", "loss_terms", "grad_norm", "observations", "advantages"):
            assert np.array_equal(rk[k], rk["seq_" + k]), k
        # the per-minibatch advantage sums are fp64 ATOMIC adds of workgroup partials (mi_adv_stats): two runs of the same launch may differ in the last bits of the
        # fp64 sum (never in the f32 mean / std derived from it — the bitwise-equal gradients above)
        assert np.allclose(rk["adv_sums"], rk["seq_adv_sums"], rtol=1e-12, atol=0.0)


def _offpolicy(backend, comm="p2p"):
    import tempfile

    import test_gpu_multigpu as M

    with tempfile.TemporaryDirectory() as tmp:
        out = _launch("_offpolicy_sharded_worker.py", dict(MIRL_TEST_BACKEND=backend, MIRL_TEST_OUT=tmp), timeout=900, comm=comm)
        assert out.returncode == 0, (out.stdout[-2000:], out.stderr[-4000:])
        assert "OFFPOLICY_WORKER_OK backend=%s native=1 carrier=p2p" % backend in out.stdout, out.stdout[-2000:]
        r0, r1 = dict(np.load(os.path.join(tmp, "off_rank0.npz"))), dict(np.load(os.path.join(tmp, "off_rank1.npz")))
    assert int(r0["dqn_native"][0]) == 1 and int(r0["sac_native"][0]) == 1
    M._check_off(r0, r1, M._off_single_process(r0))
    for rk in (r0, r1):
        seq = [k for k in rk if k.startswith("seq_") and not k.endswith("_native")]
        assert len(seq) > 20
        for k in seq:
            assert np.array_equal(rk[k], rk[k[4:]]), k


def test_two_ranks_one_gpu_p2p_ppo():
    _need_gpu()
    _ppo("gloo")


@pytest.mark.parametrize("nl", [5, 301])
def test_two_ranks_one_gpu_p2p_ppo_ragged_env_counts(nl):
    """The same with env counts per rank that are no multiple of a rollout workgroup's four envs or of a gradient tile (round 6; the randomised single-process sweeps of
    tests/test_gpu_fuzz.py cannot start ranks)."""
    _need_gpu()
    _ppo("gloo", nl=nl)


def test_two_ranks_one_gpu_p2p_offpolicy():
    _need_gpu()
    _offpolicy("gloo")


def test_two_ranks_one_gpu_auto_carrier_ppo_and_offpolicy():
    """MIRL_COMM=auto between two real processes (gloo, both on cuda:0): the probe finds RCCL impossible and the P2P carrier sound (known answer), every engine's one-call
    route then runs on it — the same equalities as with MIRL_COMM=p2p."""
    _need_gpu()
    _ppo("gloo", comm="auto")
    _offpolicy("gloo", comm="auto")


def test_two_gpus_p2p_ppo():
    if _need_gpu().cuda.device_count() < 2:
        pytest.skip("needs >= 2 GPUs; the one-GPU variant above runs the same worker with both ranks on cuda:0")
    _ppo("nccl")


def test_two_gpus_p2p_offpolicy():
    if _need_gpu().cuda.device_count() < 2:
        pytest.skip("needs >= 2 GPUs; the one-GPU variant above runs the same worker with both ranks on cuda:0")
    _offpolicy("nccl")


def test_eight_ranks_one_gpu_p2p_ppo():
    """(The placement rule of csrc/mi_comm.hip applies: more than two ranks on a device take the stand-alone all-reduce launch, not the exchange inside grad_reduce_kernel.
    Forcing the in-launch exchange here with MIRL_P2P_FUSED=1 passed once and failed once: seven waiting ranks x 145 workgroups x 16 waves are 16,240 waves on a chip that
    holds 8,192, so whether the eighth rank's launch finds a slot is a matter of timing — the bounded deadlock the rule exists to prevent.  The in-launch exchange at
    eight ranks is covered by test_ppo_update_on_synthetic_ranks_equals_plain_update[8] and, between real processes, at two ranks.)
    BASELINE config 5's rank count on the one GPU this box has: EIGHT processes on cuda:0, every one mapping the other seven inboxes (hipIpc), PPOEngine.update() on
    mi_ppo_update_sharded with grad_reduce_kernel's 8-rank exchange, two whole updates at 32 envs per rank.  All eight ranks end with bitwise the same parameters, moments,
    gradient, loss terms and clip norm (rank-ordered sum: identical by construction, here checked); against the host-sequenced route over gloo — whose 8-rank SUM uses
    another grouping — and against one process that owns all 256 envs with union minibatches, at the tolerances of the two-rank test (ppo.py:189-192)."""
    _need_gpu()
    import tempfile

    import test_gpu_multigpu as M

    W, NL = 8, 32
    with tempfile.TemporaryDirectory() as tmp:
        extra = dict(MIRL_TEST_BACKEND="gloo", MIRL_TEST_OUT=tmp, MIRL_TEST_NL=str(NL), MIRL_TEST_WORLD=str(W))
        out = _launch("_sharded_update_worker.py", extra, timeout=900, nproc=W)
        assert out.returncode == 0, (out.stdout[-2000:], out.stderr[-4000:])
        assert "SHARDED_WORKER_OK backend=gloo native=1 carrier=p2p" in out.stdout, out.stdout[-2000:]
        rk = [dict(np.load(os.path.join(tmp, "rank%d.npz" % r))) for r in range(W)]
    for r in range(1, W):
        for k in ("params0", "params", "exp_avg", "exp_avg_sq", "grads", "loss_terms", "grad_norm"):
            assert np.array_equal(rk[0][k], rk[r][k]), (k, r)
        assert np.allclose(rk[0]["adv_sums"], rk[r]["adv_sums"], rtol=1e-12, atol=0.0)
        assert int(rk[r]["native"][0]) == 1
    assert np.isfinite(rk[0]["params"]).all() and not np.array_equal(rk[0]["params"], rk[0]["params0"])
    # the host-sequenced route (gloo's own summation order over 8 ranks) and the single process with union minibatches: f32 re-association only
    big = M._single_process(rk[0]["params0"], NL, world=W)
    for ref in (rk[0]["seq_params"], big["params"]):
        assert np.abs(rk[0]["params"] - ref).max() < 2e-5, np.abs(rk[0]["params"] - ref).max()
    assert np.allclose(rk[0]["loss_terms"], big["loss_terms"], rtol=5e-3, atol=1e-4)
    assert abs(float(rk[0]["grad_norm"][0]) - float(big["grad_norm"][0])) < 5e-3 * float(big["grad_norm"][0])
    for r in range(W):   # env sharding: the first rollout's trajectories are the matching columns of the big run (the second runs on parameters that differ in the last bits)
        assert rk[r]["observations"].shape == big["observations"][:, r * NL:(r + 1) * NL].shape
