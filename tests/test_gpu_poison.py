"""Poison-buffer self-checks (SURVEY §5 'race detection / sanitizers': GPU sanitizers are not available on this pool, so the index discipline of the kernels is
checked from outside).  Every storage tensor an engine hands to the C ABI is re-homed into [guard | payload | guard] with a bit pattern no kernel produces;
after the launches (a) both guards are intact — nothing writes outside its buffer —, (b) every element the reference defines is written — no poison left —
and (c) the rows the reference never writes (ppo.py: actions[T], log_probs[T], rewards[0], dones[0]; the ring slots beyond the steps taken) still hold the
poison — nothing writes where it must not."""
import numpy as np
import pytest
import torch

import deep_rl_amd as D

pytestmark = pytest.mark.gpu
GUARD = 4096            # bytes on each side
POISON = 0xA5


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return torch.device("cuda", 0)


def _rehome(obj, names):
    """Move obj.<name> into a poisoned, guarded allocation (same shape / dtype); -> {name: whole byte buffer}."""
    raw = {}
    for n in names:
        t = getattr(obj, n)
        nbytes = t.numel() * t.element_size()
        buf = torch.full((GUARD + nbytes + GUARD,), POISON, dtype=torch.uint8, device=t.device)
        setattr(obj, n, buf[GUARD:GUARD + nbytes].view(t.dtype).view(t.shape))
        raw[n] = buf
    return raw


def _bytes(t):
    return t.contiguous().view(torch.uint8).cpu().numpy().reshape(-1)


def _guards_intact(raw):
    for n, buf in raw.items():
        b = buf.cpu().numpy()
        assert (b[:GUARD] == POISON).all(), "%s: the bytes BEFORE the buffer were written" % n
        assert (b[-GUARD:] == POISON).all(), "%s: the bytes AFTER the buffer were written" % n


def _all_written(t, what):
    """no element still holds the poison pattern in every byte"""
    b = _bytes(t).reshape(-1, t.element_size())
    left = int((b == POISON).all(axis=1).sum())
    assert left == 0, "%s: %d of %d elements were never written" % (what, left, b.shape[0])


def _untouched(t, what):
    assert (_bytes(t) == POISON).all(), "%s was written (the reference never defines it)" % what


@pytest.mark.parametrize("n_envs", [37, 4096])
def test_ppo_update_touches_exactly_its_storage(dev, n_envs):
    T = 128 if n_envs == 4096 else 20
    env = D.make("CartPole-v1", num_envs=n_envs, device=dev, seed=2)
    torch.manual_seed(2)
    agent = D.ActorCritic(env)
    eng = D.PPOEngine(env, agent, D.ClipAdam(agent, lr=2.5e-4, eps=1e-5, max_grad_norm=0.5), num_steps=T, n_minibatch=4 if n_envs == 4096 else 5, update_epochs=4)
    names = ["observations", "values", "actions", "log_probs", "rewards", "dones", "advantages", "returns", "_perm_all", "_adv_sums_all"]
    raw = _rehome(eng, names)
    eng.perm, eng.adv_sums = eng._perm_all[0], eng._adv_sums_all[0]
    eng.reset()
    eng.update()
    torch.cuda.synchronize()
    _guards_intact(raw)
    for n in ("observations", "values", "advantages", "returns", "_perm_all", "_adv_sums_all"):
        _all_written(getattr(eng, n), n)
    _all_written(eng.actions[:T], "actions[:T]"); _untouched(eng.actions[T], "actions[T]")
    _all_written(eng.log_probs[:T], "log_probs[:T]"); _untouched(eng.log_probs[T], "log_probs[T]")
    _all_written(eng.rewards[1:], "rewards[1:]"); _untouched(eng.rewards[0], "rewards[0]")
    _all_written(eng.dones[1:], "dones[1:]"); _untouched(eng.dones[0], "dones[0]")
    # every permutation really is one
    p = eng._perm_all.cpu().numpy()
    for ep in range(p.shape[0]):
        assert np.array_equal(np.sort(p[ep]), np.arange(T * n_envs))
    assert torch.isfinite(agent.flat).all()


@pytest.mark.parametrize("n_envs", [37, 1024])
def test_dqn_ring_touches_exactly_the_steps_taken(dev, n_envs):
    S, k = 64, 25
    env = D.make("CartPole-v1", num_envs=n_envs, device=dev, seed=4)
    torch.manual_seed(4)
    q = D.QNetwork(env); t = D.QNetwork(env); t.load_state_dict(q.state_dict())
    eng = D.DQNEngine(env, q, t, D.ClipAdam(q, lr=2.5e-4, eps=1e-8), slots=S, batch_size=64, learning_starts=10, total_timesteps=400)
    raw = _rehome(eng, ["observations", "actions", "rewards", "terminated", "batch_inds"])
    eng.reset()
    for n in (10, 10, 5):
        eng.act(n)
    eng.train_step()
    torch.cuda.synchronize()
    _guards_intact(raw)
    _all_written(eng.observations[:k + 1], "observations[:k+1]"); _untouched(eng.observations[k + 1:], "observations[k+1:]")
    _all_written(eng.actions[:k], "actions[:k]"); _untouched(eng.actions[k:], "actions[k:]")
    _all_written(eng.rewards[1:k + 1], "rewards[1:k+1]"); _untouched(eng.rewards[0], "rewards[0]"); _untouched(eng.rewards[k + 1:], "rewards[k+1:]")
    tb = _bytes(eng.terminated)
    assert set(np.unique(tb[1 * n_envs:(k + 1) * n_envs]).tolist()) <= {0, 1}, "terminated[1:k+1] holds something else than 0 / 1"
    assert (tb[:n_envs] == POISON).all() and (tb[(k + 1) * n_envs:] == POISON).all(), "terminated written outside slots 1..k"
    _all_written(eng.batch_inds, "batch_inds")
    bi = eng.batch_inds.cpu().numpy()
    assert bi.min() >= 0 and bi.max() < k * n_envs, "batch index outside the filled part of the ring"


def test_sac_ring_touches_exactly_the_steps_taken(dev):
    n_envs, S, k = 50, 32, 12
    env = D.make("Pendulum-v1", num_envs=n_envs, device=dev, seed=6)
    torch.manual_seed(6)
    actor = D.Actor(env); q1 = D.SoftQNetwork(env); q2 = D.SoftQNetwork(env); q1t = D.SoftQNetwork(env); q2t = D.SoftQNetwork(env)
    q1t.load_state_dict(q1.state_dict()); q2t.load_state_dict(q2.state_dict())
    eng = D.SACEngine(env, actor, q1, q2, q1t, q2t, slots=S, batch_size=64, learning_starts=5)
    raw = _rehome(eng, ["observations", "actions", "rewards", "terminated", "batch_inds"])
    eng.reset()
    for _ in range(k):
        eng.act()
    eng.train_step(policy_frequency=1)
    torch.cuda.synchronize()
    _guards_intact(raw)
    g = eng.global_step
    assert g == k
    _all_written(eng.observations[:k + 1], "observations[:k+1]"); _untouched(eng.observations[k + 1:], "observations[k+1:]")
    _all_written(eng.actions[:k], "actions[:k]"); _untouched(eng.actions[k:], "actions[k:]")
    _all_written(eng.rewards[1:k + 1], "rewards[1:k+1]"); _untouched(eng.rewards[0], "rewards[0]"); _untouched(eng.rewards[k + 1:], "rewards[k+1:]")
    bi = eng.batch_inds.cpu().numpy()
    assert bi.min() >= 0 and bi.max() < k * n_envs
