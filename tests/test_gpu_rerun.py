"""A second engine in the same process repeats the first one's run bit for bit — for every engine, after other engines of other kinds and shapes have lived and died in
the process (library-global state: the SAC shadow registry, the P2P carrier's parked inboxes, per-device status words, the in-library profiler; allocator reuse: the new
engine's tensors land where a dead engine's were; caches).  Everything keyed (seeds, counters) is per engine, so anything that differs is state that leaked between them.
Round 6 wrote this after tests/test_gpu_synthetic_world.py had found such a leak (a freed inbox's pages)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    return torch.device("cuda", 0)


def _ppo(dev, n, T):
    import deep_rl_amd as D

    env = D.make("CartPole-v1", num_envs=n, device=dev, seed=3)
    torch.manual_seed(3)
    agent = D.ActorCritic(env)
    opt = D.ClipAdam(agent, lr=2.5e-4, eps=1e-5, max_grad_norm=0.5)
    eng = D.PPOEngine(env, agent, opt, num_steps=T)
    eng.reset()
    for _ in range(3):
        eng.update()
    return [t.clone() for t in (agent.flat, opt.exp_avg, opt.exp_avg_sq, eng.loss_terms, eng.observations, eng.advantages, eng.episode_stats)]


def _dqn(dev, kind, n, slots, batch):
    import deep_rl_amd as D

    env = D.make("CartPole-v1", num_envs=n, device=dev, seed=5)
    torch.manual_seed(5)
    Net = D.DuelingQNetwork if kind == "dueling" else D.QNetwork
    q = Net(env); tgt = Net(env)
    tgt.load_state_dict(q.state_dict())
    opt = D.ClipAdam(q, lr=2.5e-4, eps=1e-8)
    Eng = {"dqn": D.DQNEngine, "dueling": D.DuelingDQNEngine, "per": D.PERDQNEngine}[kind]
    eng = Eng(env, q, tgt, opt, slots=slots, batch_size=batch, learning_starts=0, total_timesteps=100 * slots, max_episodes_logged=0)
    eng.reset()
    for k in range(3 * slots // 10 + 4):         # the ring fills and wraps
        eng.act(10)
        eng.train_step()
        if k % 5 == 4:
            eng.sync_target()
    out = [t.clone() for t in (q.flat, tgt.flat, opt.exp_avg, opt.exp_avg_sq, eng.loss, eng.observations, eng.actions, eng.batch_inds)]
    if kind == "per":
        eng.settle()
        out += [eng.priorities.clone(), eng.max_priority.clone()]
    return out


def _sac(dev, n, slots, batch):
    import deep_rl_amd as D

    env = D.make("Pendulum-v1", num_envs=n, device=dev, seed=7)
    torch.manual_seed(7)
    actor = D.Actor(env); q1 = D.SoftQNetwork(env); q2 = D.SoftQNetwork(env); q1t = D.SoftQNetwork(env); q2t = D.SoftQNetwork(env)
    q1t.load_state_dict(q1.state_dict()); q2t.load_state_dict(q2.state_dict())
    eng = D.SACEngine(env, actor, q1, q2, q1t, q2t, slots=slots, batch_size=batch, learning_starts=4, max_episodes_logged=0)
    eng.reset()
    for _ in range(slots + 9):
        eng.act()
        if eng.global_step > 6:
            eng.train_step()
    out = [t.clone() for t in (actor.flat, eng.q_flat, eng.qt_flat, eng.actor_optimizer.exp_avg, eng.q_optimizer.exp_avg_sq, eng.log_alpha, eng.alpha, eng.observations, eng.actions, eng.rewards)]
    eng.close()
    return out


RUNS = [("ppo 64 x 128", lambda d: _ppo(d, 64, 128)), ("dqn batch 128", lambda d: _dqn(d, "dqn", 64, 32, 128)), ("sac batch 256", lambda d: _sac(d, 48, 40, 256)),
        ("per batch 700", lambda d: _dqn(d, "per", 130, 24, 700)), ("ppo 300 x 20", lambda d: _ppo(d, 300, 20)), ("dueling batch 96", lambda d: _dqn(d, "dueling", 64, 32, 96)),
        ("dqn batch 1000", lambda d: _dqn(d, "dqn", 300, 20, 1000)), ("sac batch 600", lambda d: _sac(d, 64, 24, 600)), ("per batch 128", lambda d: _dqn(d, "per", 64, 40, 128))]


def test_every_run_repeats_itself_after_the_others_have_run(dev):
    first = [(name, fn(dev)) for name, fn in RUNS]          # one engine of every kind and shape, one after the other
    for name, fn in reversed(RUNS):                           # again, in the opposite order: each now runs behind different predecessors, on reused memory
        again = fn(dev)
        ref = dict(first)[name]
        for j, (a, b) in enumerate(zip(ref, again)):
            assert torch.equal(a, b), (name, j, (a.double() - b.double()).abs().max().item())
    for name, out in first:
        assert torch.isfinite(out[0]).all(), name
