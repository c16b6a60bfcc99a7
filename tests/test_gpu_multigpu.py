"""The sharded PPO update at world_size 2 against one process that owns all the envs (reference ppo.py:189-192 with the gradient
exchange between backward and clip_grad_norm_).

  test_two_gpus_rccl_native_sharded_update   needs >= 2 GPUs (skips otherwise — it runs the day a multi-GPU box appears): one rank per GPU over RCCL,
      PPOEngine.update() on the ONE-CALL route (mi_ppo_update_sharded: 17 in-stream ncclAllReduce per update on libmirl's own communicator), two whole
      updates; checked against the single process with union minibatches, against the host-sequenced route over torch's RCCL (bit for bit) and
      rank against rank (bit for bit).
  test_two_ranks_one_gpu_gloo_whole_updates  the same worker and the same comparison with both ranks on cuda:0 over gloo (host-sequenced route): keeps the
      harness itself green on the one-GPU box.
  test_two_gpus_rccl_offpolicy_one_call_routes / test_two_ranks_one_gpu_gloo_offpolicy   the same pair for DQNEngine, PERDQNEngine and SACEngine (reference dqn.py:131-133,
      per.py:147-153, sac.py:185-210 with the gradient exchange between backward and optimizer.step()): tests/_offpolicy_sharded_worker.py."""
import os
import socket
import subprocess
import sys
import tempfile

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from _ports import free_port, run_with_port  # noqa: E402


def _run(backend, nl):
    with tempfile.TemporaryDirectory() as tmp:
        env = dict(os.environ, PYTHONPATH=ROOT, MIRL_TEST_BACKEND=backend, MIRL_TEST_OUT=tmp, MIRL_TEST_NL=str(nl), HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="1")
        # the ranks are started as children BEFORE anything of theirs touches a GPU (never exec from a process that has initialised HIP)
        out = run_with_port(lambda port: ([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
                                           os.path.join(ROOT, "tests", "_sharded_update_worker.py")], env), capture_output=True, text=True, timeout=300, cwd=ROOT)
        assert out.returncode == 0, (out.stdout[-2000:], out.stderr[-4000:])
        assert "SHARDED_WORKER_OK backend=%s" % backend in out.stdout, out.stdout[-2000:]
        load = lambda n: dict(np.load(os.path.join(tmp, n)))  # noqa: E731
        r0, r1 = load("rank0.npz"), load("rank1.npz")
    return r0, r1, _single_process(r0["params0"], nl)


T, UPDATES, SEED = 128, 2, 11


def _single_process(params0, nl, world=2):
    """One process that owns all world * nl envs: the explicit launch sequence with the UNION minibatches (each rank draws the same keyed permutation of its LOCAL rows;
    local row t * nl + e of rank r is row t * (world nl) + r * nl + e here)."""
    import torch

    import deep_rl_amd as D
    from deep_rl_amd import _native as N

    dev = torch.device("cuda", 0)
    env = D.make("CartPole-v1", num_envs=world * nl, device=dev, seed=SEED)
    torch.manual_seed(SEED)
    agent = D.ActorCritic(env)
    big = D.PPOEngine(env, agent, D.ClipAdam(agent, lr=2.5e-4, eps=1e-5, max_grad_norm=0.5), num_steps=T)
    assert big.world_size == 1
    big.agent.load_flat(params0)
    big.reset()
    scratch = torch.zeros(T * nl, dtype=torch.int32, device=dev)
    mb = T * nl // big.n_minibatch
    for u in range(UPDATES):
        big.rollout(); big.compute_gae()
        for ep in range(big.update_epochs):
            N.check(N.lib().mi_make_perm(T * nl, N.lib().mi_perm_key(SEED, u, ep), N.ptr(scratch), N.stream_ptr(dev)), "mi_make_perm")
            local = scratch.cpu().numpy().astype(np.int64)
            for k in range(big.n_minibatch):
                part = local[k * mb:(k + 1) * mb]
                union = np.concatenate([(part // nl) * (world * nl) + r * nl + part % nl for r in range(world)]).astype(np.int32)
                big.perm[:world * mb].copy_(torch.from_numpy(union).to(dev))
                big.adv_stats(mb=world * mb, n_mb=1)
                big.minibatch_grad(0, mb=world * mb)
                big.optimizer_step()
        big.update_index += 1
    torch.cuda.synchronize()
    o = big.optimizer
    return {"params": big.agent.flat.cpu().numpy(), "exp_avg": o.exp_avg.cpu().numpy(), "loss_terms": big.loss_terms.cpu().numpy(), "grad_norm": o.grad_norm.cpu().numpy(),
            "observations": big.observations.cpu().numpy()}


def _check(r0, r1, big, nl):
    # replicas never diverge: every rank ends with bitwise the same optimizer state, last gradient, loss terms and (all-reduced) advantage statistics
    for k in ("params0", "params", "exp_avg", "exp_avg_sq", "grads", "loss_terms", "grad_norm", "adv_sums"):
        assert np.array_equal(r0[k], r1[k]), k
    assert np.isfinite(r0["params"]).all() and not np.array_equal(r0["params"], r0["params0"])
    # env sharding: each rank's trajectories are the matching columns of the big run (bit-exact while the parameters agree: first rollout exactly;
    # the second rollout runs on parameters that differ in the last bits, so it is compared through the final state below)
    for r, rk in enumerate((r0, r1)):
        sl = slice(r * nl, (r + 1) * nl)
        assert rk["observations"].shape == big["observations"][:, sl].shape
    # 2 ranks x NL envs == one process with 2 NL envs and union minibatches, after 2 x 16 chained optimizer steps: the gradient shares are summed in a
    # different grouping (per rank, then across ranks), which moves last bits of every step (one step alone: <= 2e-7, tests/test_gpu_multirank.py)
    assert np.abs(r0["params"] - big["params"]).max() < 2e-5, np.abs(r0["params"] - big["params"]).max()
    assert np.abs(r0["exp_avg"] - big["exp_avg"]).max() < 1e-4 * max(1e-3, np.abs(big["exp_avg"]).max())
    assert np.allclose(r0["loss_terms"], big["loss_terms"], rtol=5e-3, atol=1e-4), (r0["loss_terms"], big["loss_terms"])
    assert abs(float(r0["grad_norm"][0]) - float(big["grad_norm"][0])) < 5e-3 * float(big["grad_norm"][0])


def test_two_gpus_rccl_native_sharded_update():
    import torch

    if torch.cuda.device_count() < 2:
        pytest.skip("needs >= 2 GPUs (RCCL refuses two ranks on one device); the one-GPU variant below covers the harness")
    r0, r1, big = _run("nccl", 64)
    assert int(r0["native"][0]) == 1 and int(r1["native"][0]) == 1     # mi_ppo_update_sharded really was the route
    _check(r0, r1, big, 64)
    # one C call with in-stream ncclAllReduce == host-sequenced launches with torch.distributed all-reduces over RCCL, bit for bit (a + b on both ranks)
    for rk in (r0, r1):
        for k in ("params", "exp_avg", "exp_avg_sq", "grads", "loss_terms", "grad_norm", "observations", "advantages"):
            assert np.array_equal(rk[k], rk["seq_" + k]), k
        # the per-minibatch advantage sums are fp64 ATOMIC adds of workgroup partials (mi_adv_stats): two runs of the same launch may differ in the last bits of the
        # fp64 sum (never in the f32 mean / std derived from it — the bitwise-equal gradients above)
        assert np.allclose(rk["adv_sums"], rk["seq_adv_sums"], rtol=1e-12, atol=0.0)


def test_two_ranks_one_gpu_gloo_whole_updates():
    import torch

    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    r0, r1, big = _run("gloo", 64)
    assert int(r0["native"][0]) == 0
    _check(r0, r1, big, 64)


# =============================================================== DQN / PER / SAC at world_size 2 ===============================================================
# (the one-call RCCL routes of the off-policy engines were checked at world_size 1 only.)
OFF_NL, OFF_STEPS, OFF_B, OFF_ROUNDS, OFF_SEED = 8, 40, 64, 3, 7   # == tests/_offpolicy_sharded_worker.py


def _run_off(backend):
    with tempfile.TemporaryDirectory() as tmp:
        env = dict(os.environ, PYTHONPATH=ROOT, MIRL_TEST_BACKEND=backend, MIRL_TEST_OUT=tmp, HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="1")
        out = run_with_port(lambda port: ([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
                                           os.path.join(ROOT, "tests", "_offpolicy_sharded_worker.py")], env), capture_output=True, text=True, timeout=600, cwd=ROOT)
        assert out.returncode == 0, (out.stdout[-2000:], out.stderr[-4000:])
        assert "OFFPOLICY_WORKER_OK backend=%s" % backend in out.stdout, out.stdout[-2000:]
        return dict(np.load(os.path.join(tmp, "off_rank0.npz"))), dict(np.load(os.path.join(tmp, "off_rank1.npz")))


def _off_inputs(r):
    rng = np.random.default_rng(1000 + r)
    return [(rng.integers(0, (OFF_STEPS - 1) * OFF_NL, OFF_B), rng.standard_normal((3, OFF_B)).astype(np.float32)) for _ in range(OFF_ROUNDS)]


def _off_single_process(r0):
    """One process that owns all 2 * NL envs, with the union batches (rank r's local index slot * NL + e is slot * 2 NL + r NL + e here) and the concatenated noise,
    on the fused single-process calls."""
    import torch

    import deep_rl_amd as D

    dev = torch.device("cuda", 0)
    nl, n = OFF_NL, 2 * OFF_NL
    ins = [_off_inputs(r) for r in range(2)]
    union = lambda k: np.concatenate([(ins[r][k][0] // nl) * n + r * nl + ins[r][k][0] % nl for r in range(2)])  # noqa: E731
    env = D.make("CartPole-v1", num_envs=n, device=dev, seed=OFF_SEED)
    torch.manual_seed(OFF_SEED)
    q = D.QNetwork(env); t = D.QNetwork(env)
    q.load_flat(r0["dqn_init"]); t.load_flat(r0["dqn_init"])
    dqn = D.DQNEngine(env, q, t, D.ClipAdam(q, lr=2.5e-4, eps=1e-8), slots=OFF_STEPS + 1 + 10 * OFF_ROUNDS, batch_size=2 * OFF_B, learning_starts=10, total_timesteps=400,
                      max_episodes_logged=0)
    dqn.reset(); dqn.act(OFF_STEPS)
    for k in range(OFF_ROUNDS):
        dqn.train_step(union(k))
    env = D.make("Pendulum-v1", num_envs=n, device=dev, seed=OFF_SEED)
    torch.manual_seed(OFF_SEED)
    a = D.Actor(env)
    qs = [D.SoftQNetwork(env) for _ in range(4)]
    sac = D.SACEngine(env, a, *qs, slots=OFF_STEPS + 1, batch_size=2 * OFF_B, learning_starts=10, max_episodes_logged=0)
    a.load_flat(r0["sac_actor_init"])
    sac.q_flat.copy_(torch.from_numpy(r0["sac_q_init"]).to(dev)); sac.qt_flat.copy_(torch.from_numpy(r0["sac_q_init"]).to(dev))
    sac.reset()
    for _ in range(OFF_STEPS):
        sac.act()
    for k in range(OFF_ROUNDS):
        eps = np.concatenate([ins[0][k][1], ins[1][k][1]], axis=1)
        sac.sample(union(k))
        sac.update_critic(torch.from_numpy(eps[0]), polyak=True); sac.update_actor(torch.from_numpy(eps[1])); sac.update_alpha(torch.from_numpy(eps[2]))
    torch.cuda.synchronize()
    return {"dqn_q": dqn.q.flat.cpu().numpy(), "dqn_loss": dqn.loss.cpu().numpy(), "dqn_obs": dqn.observations.cpu().numpy(), "sac_q": sac.q_flat.cpu().numpy(),
            "sac_qt": sac.qt_flat.cpu().numpy(), "sac_actor": sac.actor.flat.cpu().numpy(), "sac_log_alpha": sac.log_alpha.cpu().numpy(), "sac_obs": sac.observations.cpu().numpy(),
            "sac_q_losses": sac.q_losses.cpu().numpy()}


def _check_off(r0, r1, big):
    # replicas stay bitwise identical (the in-worker guard, MIRL_CHECK_REPLICAS=2, has also passed): parameters, moments, all-reduced gradient and losses
    for k in ("dqn_q", "dqn_m", "dqn_v", "dqn_loss", "dqn_grads", "per_q", "per_m", "per_loss", "sac_q", "sac_qt", "sac_actor", "sac_log_alpha", "sac_q_losses",
              "sac_actor_out", "sac_qm", "sac_am"):
        assert np.array_equal(r0[k], r1[k]), k
        assert np.isfinite(r0[k]).all(), k
    assert not np.array_equal(r0["dqn_q"], r0["dqn_init"]) and not np.array_equal(r0["per_q"], r0["dqn_init"]) and not np.array_equal(r0["sac_actor"], r0["sac_actor_init"])
    assert not np.array_equal(r0["per_prio_sum"], r1["per_prio_sum"])      # the priorities are per-rank state (each rank's own ring)
    # env sharding: each rank's ring holds the matching columns of the big run's ring (acting is parameter-independent here: before learning_starts / same actor)
    nl = OFF_NL
    for r, rk in enumerate((r0, r1)):
        assert np.array_equal(rk["dqn_obs"][:OFF_STEPS + 1], big["dqn_obs"][:OFF_STEPS + 1, r * nl:(r + 1) * nl])
        assert np.array_equal(rk["sac_obs"], big["sac_obs"][:, r * nl:(r + 1) * nl])
    # 2 ranks x B rows == one process with the union batch of 2 B rows, after 3 chained steps (shares summed in a different grouping move last bits per step)
    assert np.abs(r0["dqn_q"] - big["dqn_q"]).max() < 2e-5 and abs(float(r0["dqn_loss"][0]) - float(big["dqn_loss"][0])) < 1e-4 * max(1.0, abs(float(big["dqn_loss"][0])))
    assert np.abs(r0["sac_q"] - big["sac_q"]).max() < 5e-5 and np.abs(r0["sac_qt"] - big["sac_qt"]).max() < 2e-6
    assert np.abs(r0["sac_actor"] - big["sac_actor"]).max() < 5e-5 and abs(float(r0["sac_log_alpha"][0]) - float(big["sac_log_alpha"][0])) < 1e-5
    assert np.allclose(r0["sac_q_losses"], big["sac_q_losses"], rtol=1e-3)


def test_two_gpus_rccl_offpolicy_one_call_routes():
    """DQNEngine / PERDQNEngine / SACEngine, one rank per GPU over RCCL: the one-call routes (mi_dqn_td_update_sharded, mi_sac_{critic,actor}_update_sharded,
    mi_sac_alpha_step_sharded) against the host-sequenced route over torch's RCCL (bit for bit), rank against rank (bit for bit) and the single process with union batches."""
    import torch

    if torch.cuda.device_count() < 2:
        pytest.skip("needs >= 2 GPUs (RCCL refuses two ranks on one device); the one-GPU variant below covers the harness")
    r0, r1 = _run_off("nccl")
    assert int(r0["dqn_native"][0]) == 1 and int(r0["sac_native"][0]) == 1
    _check_off(r0, r1, _off_single_process(r0))
    for rk in (r0, r1):
        for k in [k for k in rk if k.startswith("seq_") and not k.endswith("_native")]:
            assert np.array_equal(rk[k], rk[k[4:]]), k


def test_two_ranks_one_gpu_gloo_offpolicy():
    import torch

    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    r0, r1 = _run_off("gloo")
    assert int(r0["dqn_native"][0]) == 0 and int(r0["sac_native"][0]) == 0
    _check_off(r0, r1, _off_single_process(r0))
