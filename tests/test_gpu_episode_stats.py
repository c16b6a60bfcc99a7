"""Episode statistics without same-address atomics (ABI 104): an acting / rollout call given episode_stats == NULL keeps {episodes, sum of lengths, longest} per workgroup
inside the env handle and mi_env_episode_stats sums them on request; with a buffer, large launches do the same and fill the buffer behind the launch.  All three forms
(atomics, buffer + per-workgroup, handle only) must give the same four integers, which are what ppo.py:130 / dqn.py:110-111 print from."""
import ctypes as C

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    return torch.device("cuda", 0)


def _ppo(dev, n, max_ep, T=64):
    import deep_rl_amd as D

    env = D.make("CartPole-v1", num_envs=n, device=dev, seed=11)
    torch.manual_seed(11)
    agent = D.ActorCritic(env)
    eng = D.PPOEngine(env, agent, D.ClipAdam(agent, lr=2.5e-4, eps=1e-5, max_grad_norm=0.5), num_steps=T, max_episodes_logged=max_ep)
    eng.reset()
    return eng


@pytest.mark.parametrize("n", [4, 100, 128, 1000, 4096])
def test_ppo_rollout_statistics_agree_in_all_three_forms(dev, n):
    from deep_rl_amd import _native as N

    lazy, logged = _ppo(dev, n, 0), _ppo(dev, n, 8)      # handle only | atomics (an episode log forces them)
    assert lazy._lazy_stats and not logged._lazy_stats
    buf = torch.full((4,), 77, dtype=torch.int32, device=dev)
    third = _ppo(dev, n, 0)
    for _ in range(3):
        lazy.rollout_gae(); logged.rollout_gae()
        N.check(N.lib().mi_ppo_rollout_gae(third.env.handle, N.ptr(third.agent.flat), third.T, N.ptr(third.observation), N.ptr(third.observations), N.ptr(third.values),
                                           N.ptr(third.actions), N.ptr(third.log_probs), N.ptr(third.rewards), N.ptr(third.dones), None, N.ptr(buf), 0,
                                           third.gamma, third.gae_lambda, N.ptr(third.advantages), N.ptr(third.returns), N.stream_ptr(dev)), "mi_ppo_rollout_gae")
        a, b, c = lazy.episode_stats.tolist(), logged.episode_stats.tolist(), buf.tolist()
        assert a[:3] == b[:3] == c[:3] and a[0] > 0, (n, a, b, c)
        assert a[3] == 0 and c[3] == 0 and b[3] == b[0]          # slots of the episode log: only the logging form hands any out
        assert torch.equal(lazy.observations, logged.observations) and torch.equal(lazy.returns, third.returns)


def test_fused_update_keeps_the_statistics_in_the_handle(dev):
    lazy, logged = _ppo(dev, 256, 0, T=32), _ppo(dev, 256, 4, T=32)
    for _ in range(3):
        lazy.update(); logged.update()
        assert lazy.episode_stats.tolist()[:3] == logged.episode_stats.tolist()[:3]
        assert torch.equal(lazy.agent.flat, logged.agent.flat)
    host = torch.zeros(4, dtype=torch.int32).pin_memory()
    lazy.episode_summary_async(host); torch.cuda.synchronize()
    assert host.tolist() == lazy.episode_stats.tolist()


@pytest.mark.parametrize("n", [16, 48, 2048, 4096])
def test_dqn_acting_statistics_agree(dev, n):
    import deep_rl_amd as D

    engs = []
    for max_ep in (0, 8):
        env = D.make("CartPole-v1", num_envs=n, device=dev, seed=5)
        torch.manual_seed(5)
        q = D.QNetwork(env); t = D.QNetwork(env); t.load_state_dict(q.state_dict())
        eng = D.DQNEngine(env, q, t, D.ClipAdam(q, lr=2.5e-4, eps=1e-8), slots=64, batch_size=32, learning_starts=0, total_timesteps=1000, max_episodes_logged=max_ep)
        eng.reset(); engs.append(eng)
    lazy, logged = engs
    assert lazy._lazy_stats and not logged._lazy_stats
    tot = 0
    for _ in range(6):
        lazy.act(10); logged.act(10)
        a, b = lazy.episode_stats.tolist(), logged.episode_stats.tolist()
        assert a[:3] == b[:3], (n, a, b)
        tot += a[0]
        assert torch.equal(lazy.observations, logged.observations) and torch.equal(lazy.actions, logged.actions)
    assert tot > 0


def test_statistics_before_any_call_are_an_error_not_garbage(dev):
    import deep_rl_amd as D
    from deep_rl_amd import _native as N

    env = D.make("CartPole-v1", num_envs=64, device=dev, seed=1)
    out = torch.zeros(4, dtype=torch.int32, device=dev)
    rc = N.lib().mi_env_episode_stats(env.handle, N.ptr(out), N.stream_ptr(dev))
    assert rc == -4 and b"episode_stats == NULL" in N.lib().mi_last_error()       # MI_ESTATE
    assert N.lib().mi_env_episode_stats(None, N.ptr(out), None) == -1                  # MI_EINVAL


def test_profiler_brackets_only_what_it_is_asked_to(dev):
    """bench.py's live roofline figure: mi_prof_begin arms the in-library HIP-event profiler for the tagged kernels, mi_prof_pause stops / resumes the sampling in
    between (the timed windows bracket the gradient launches of every 10th update only: a pair of events around every launch costs the window 7.5 %), mi_prof_end
    reports totals and counts of the bracketed launches alone."""
    from deep_rl_amd import _native as N

    eng = _ppo(dev, 256, 0, T=16)
    for _ in range(2):
        eng.update()
    N.prof_begin(64, tags=["grad", "rollout"])
    eng.update()                      # bracketed: 16 gradient launches + 1 rollout
    N.prof_pause(True)
    eng.update(); eng.update()        # not bracketed
    N.prof_pause(False)
    eng.update()                      # bracketed again
    p = N.prof_end()
    assert p["grad"][1] == 32 and p["rollout"][1] == 2 and p["reduce"][1] == 0
    assert 0.0 < p["grad"][0] / 32 < 1.0 and 0.0 < p["rollout"][0]          # ms per launch: sane
    eng.update()                      # disarmed: nothing recorded, nothing breaks
    N.prof_begin(8, tags=["grad"])
    N.prof_pause(True)
    eng.update()
    assert N.prof_end()["grad"][1] == 0
