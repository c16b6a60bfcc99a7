"""The one-call sharded routes of the OFF-POLICY engines on the P2P carrier with 1 .. 8 SYNTHETIC ranks (mi_comm_p2p_synthetic: this process stores its share into slot 0
and zeros into the other slots, polls them all and sums them in rank order — every store, poll and add of a `world`-rank exchange, minus the links).

x + 0 + ... + 0 is x bit for bit, so whatever the rank count, the training run must be THE SAME RUN: parameters, targets, Adam moments, losses, priorities and the
replay ring after a dozen iterations agree bitwise with the one-rank communicator's.  That pins every instantiation of the in-launch exchange these routes carry since
round 6 — dqn_reduce_kernel<PER, WORLD> for WORLD = 1 .. 8 (templated), SAC's run-time-world exchange at its four assembly sites — of which the two-real-rank tests
(tests/test_gpu_p2p.py) reach WORLD = 2 only and bench.py's `sharded_synthetic` legs only time WORLD = 8.  (PPO's route has had this test since round 5:
test_gpu_p2p.py::test_ppo_update_on_synthetic_ranks_equals_plain_update.)  Reference lines: dqn.py:131-133, per.py:147-153, sac.py:185-210."""
import ctypes as C

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    return torch.device("cuda", 0)


def _on_synthetic_world(world, forced_module, body):
    import deep_rl_amd.dist as DD
    from deep_rl_amd import _native as N

    h = C.c_void_p()
    N.check(N.lib().mi_comm_p2p_synthetic(world, 1 << 20, C.byref(h)), "mi_comm_p2p_synthetic")
    DD.use_comm(h)
    forced_module._FORCE_SHARDED = True
    try:
        out = body()
        torch.cuda.synchronize()
        N.check(N.lib().mi_comm_check(h), "mi_comm_check")       # no wait ran out
        return out
    finally:
        forced_module._FORCE_SHARDED = False
        DD.use_comm(None)
        torch.cuda.synchronize()
        N.lib().mi_comm_destroy(h)


def _dqn_run(dev, kind, n_envs, slots, batch, iters):
    import deep_rl_amd as D
    import deep_rl_amd.dqn_engine as E

    def body():
        env = D.make("CartPole-v1", num_envs=n_envs, device=dev, seed=13)
        torch.manual_seed(13)
        q = D.QNetwork(env); tgt = D.QNetwork(env)
        tgt.load_state_dict(q.state_dict())
        opt = D.ClipAdam(q, lr=2.5e-4, eps=1e-8)
        Eng = D.PERDQNEngine if kind == "per" else D.DQNEngine
        eng = Eng(env, q, tgt, opt, slots=slots, batch_size=batch, learning_starts=0, total_timesteps=100 * slots, max_episodes_logged=0)
        eng.reset()
        assert eng._native_sharded()
        out = []
        for k in range(iters):
            eng.act(10)
            eng.train_step()
            if k % 5 == 4:
                eng.sync_target()
            row = [eng.q.flat.clone(), eng.target.flat.clone(), opt.exp_avg.clone(), opt.exp_avg_sq.clone(), eng._gradbuf.clone(), eng.batch_inds.clone()]
            if kind == "per":
                eng.settle()
                row += [eng.priorities.clone(), eng.max_priority.clone(), eng.weights.clone(), eng.td_abs.clone()]
            out.append(row)
        assert opt.step_count == iters
        return out

    return lambda world: _on_synthetic_world(world, E, body)


@pytest.mark.parametrize("kind,n_envs,slots,batch", [("dqn", 64, 32, 128), ("dqn", 300, 20, 1000), ("per", 64, 40, 128), ("per", 130, 24, 700)])
def test_dqn_and_per_sharded_route_is_the_same_run_on_1_to_8_synthetic_ranks(dev, kind, n_envs, slots, batch):
    run = _dqn_run(dev, kind, n_envs, slots, batch, 12)
    ref = run(1)
    assert torch.isfinite(ref[-1][0]).all() and not torch.equal(ref[-1][0], ref[0][0])
    for world in range(2, 9):
        got = run(world)
        for k, (x, y) in enumerate(zip(ref, got)):
            for j, (u, v) in enumerate(zip(x, y)):
                assert torch.equal(u, v), (kind, world, k, j, (u.double() - v.double()).abs().max().item())


def _sac_run(dev, n_envs, slots, batch, iters):
    import deep_rl_amd as D
    import deep_rl_amd.sac_engine as E

    def body():
        env = D.make("Pendulum-v1", num_envs=n_envs, device=dev, seed=9)
        torch.manual_seed(9)
        actor = D.Actor(env); q1 = D.SoftQNetwork(env); q2 = D.SoftQNetwork(env); q1t = D.SoftQNetwork(env); q2t = D.SoftQNetwork(env)
        q1t.load_state_dict(q1.state_dict()); q2t.load_state_dict(q2.state_dict())
        eng = D.SACEngine(env, actor, q1, q2, q1t, q2t, slots=slots, batch_size=batch, learning_starts=4, max_episodes_logged=0)
        eng.reset()
        assert not eng._single()
        out = []
        for _ in range(iters):
            eng.act()
            if eng.global_step > 6:
                eng.train_step()
                out.append([t.clone() for t in (eng.actor.flat, eng.q_flat, eng.qt_flat, eng.actor_optimizer.exp_avg, eng.actor_optimizer.exp_avg_sq, eng.q_optimizer.exp_avg,
                                                eng.q_optimizer.exp_avg_sq, eng.log_alpha, eng.alpha, eng.q_losses, eng.actor_out)])
        final = [eng.observations.clone(), eng.actions.clone(), eng.rewards.clone()]
        eng.close()
        return out, final

    return lambda world: _on_synthetic_world(world, E, body)


@pytest.mark.parametrize("n_envs,slots,batch", [(48, 40, 256), (20, 30, 40), (64, 24, 600)])
def test_sac_sharded_route_is_the_same_run_on_1_to_8_synthetic_ranks(dev, n_envs, slots, batch):
    run = _sac_run(dev, n_envs, slots, batch, 26)
    ref, ref_ring = run(1)
    assert len(ref) > 15 and torch.isfinite(ref[-1][0]).all() and not torch.equal(ref[-1][0], ref[0][0]) and not torch.equal(ref[-1][7], ref[0][7])
    for world in range(2, 9):
        got, ring = run(world)
        for k, (x, y) in enumerate(zip(ref, got)):
            for j, (u, v) in enumerate(zip(x, y)):
                assert torch.equal(u, v), (world, k, j, (u.double() - v.double()).abs().max().item())
        for u, v in zip(ref_ring, ring):
            assert torch.equal(u, v), world
