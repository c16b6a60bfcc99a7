"""Checkpoint / resume: a run restored from deep_rl_amd.checkpoint continues BIT FOR BIT — env blob, carried-over
observation, parameters, optimizer moments, counters, replay ring; all randomness is counter-based."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    return torch.device("cuda", 0)


def _ppo(dev):
    import deep_rl_amd as D

    env = D.make("CartPole-v1", num_envs=96, device=dev, seed=4, env_id_base=32)
    torch.manual_seed(4)
    agent = D.ActorCritic(env)
    return D.PPOEngine(env, agent, D.ClipAdam(agent, lr=2.5e-4, eps=1e-5, max_grad_norm=0.5), num_steps=64)


def _dqn(dev, kind):
    import deep_rl_amd as D

    env = D.make("CartPole-v1", num_envs=48, device=dev, seed=4)
    torch.manual_seed(4)
    Net = D.DuelingQNetwork if kind == "dueling" else D.QNetwork
    Eng = {"dqn": D.DQNEngine, "dueling": D.DuelingDQNEngine, "per": D.PERDQNEngine}[kind]
    q = Net(env); t = Net(env); t.load_state_dict(q.state_dict())
    return Eng(env, q, t, D.ClipAdam(q, lr=2.5e-4, eps=1e-8), slots=64, batch_size=128, learning_starts=20, total_timesteps=400, max_episodes_logged=0)


def _sac(dev):
    import deep_rl_amd as D

    env = D.make("Pendulum-v1", num_envs=48, device=dev, seed=4)
    torch.manual_seed(4)
    a = D.Actor(env); qs = [D.SoftQNetwork(env) for _ in range(4)]
    qs[2].load_state_dict(qs[0].state_dict()); qs[3].load_state_dict(qs[1].state_dict())
    return D.SACEngine(env, a, *qs, slots=64, batch_size=128, learning_starts=10, max_episodes_logged=0)


def _step(eng, k):
    import deep_rl_amd as D

    for _ in range(k):
        if isinstance(eng, D.PPOEngine):
            eng.update()
        elif isinstance(eng, D.SACEngine):
            eng.act()
            if eng.global_step >= eng.learning_starts:
                eng.train_step()
        else:
            eng.act(10)
            if eng.global_step >= eng.learning_starts:
                eng.train_step()
            if eng.global_step % 50 == 0:
                eng.sync_target()


def _fingerprint(eng):
    import deep_rl_amd as D

    if isinstance(eng, D.PPOEngine):
        ts = [eng.agent.flat, eng.optimizer.exp_avg_sq, eng.observations, eng.actions, eng.advantages, eng.observation]
    elif isinstance(eng, D.SACEngine):
        ts = [eng.actor.flat, eng.q_flat, eng.qt_flat, eng.log_alpha, eng.observations, eng.actions, eng.observation, eng.q_optimizer.exp_avg]
    else:
        ts = [eng.q.flat, eng.target.flat, eng.optimizer.exp_avg, eng.observations, eng.actions, eng.terminated, eng.observation]
        if isinstance(eng, D.PERDQNEngine):
            ts += [eng.priorities, eng.max_priority]
    return [t.clone() for t in ts]


@pytest.mark.parametrize("kind", ["ppo", "dqn", "dueling", "per", "sac"])
def test_resume_is_bit_exact(dev, kind, tmp_path):
    from deep_rl_amd import checkpoint

    mk = {"ppo": lambda: _ppo(dev), "sac": lambda: _sac(dev)}.get(kind, lambda: _dqn(dev, kind))
    first, more = (2, 2) if kind == "ppo" else (25, 15)
    a = mk(); a.reset(); _step(a, first)
    path = os.path.join(tmp_path, kind + ".npz")
    checkpoint.save(path, a)
    _step(a, more)
    b = mk()                                   # fresh engine, never reset
    checkpoint.load(path, b)
    _step(b, more)
    for x, y in zip(_fingerprint(a), _fingerprint(b)):
        assert torch.equal(x, y)
    z = np.load(path)
    assert int(z["format"]) == 1 and z["env_blob"].size == 60 * a.env.num_envs
    # a checkpoint is refused by an engine of another shape
    import deep_rl_amd as D
    if kind == "ppo":
        env = D.make("CartPole-v1", num_envs=8, device=dev, seed=4)
        ag = D.ActorCritic(env)
        with pytest.raises(D._native.MiError):
            checkpoint.load(path, D.PPOEngine(env, ag, D.ClipAdam(ag), num_steps=64))
