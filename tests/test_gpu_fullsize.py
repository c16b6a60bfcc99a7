"""Full-size oracle parity: BASELINE.json configs 2 / 3 / 4 at their REAL sizes against the CPU oracle (oracle/cpu_ref.c).

The smaller-size tests (test_gpu_parity / _dqn / _sac) sweep shapes and edge cases; these run each hot kernel once at the size the
benchmark runs it — the 512-workgroup grid with the age-aware tile split of grad_kernel, the 256-workgroup rollout, a wrapped
1,048,576-transition ring — and compare with the oracle on the same inputs: bit-exact for env state / flags / indices, fp32
tolerances (written at each assert) for everything that goes through a network.  The oracle needs 1-3 s per PPO update at this size.
"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
T, N_PPO = 128, 4096


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    return torch.device("cuda", 0)


@pytest.fixture(scope="module")
def R():
    from oracle import cpu_ref

    cpu_ref.lib().ref_set_num_threads(max(1, min(32, os.cpu_count() or 1)))
    return cpu_ref


@pytest.fixture(autouse=True)
def _fdlibm_mode(R):
    R.set_sincos_mode("fdlibm")  # the device-matched sin/cos mode
    yield
    R.set_sincos_mode("libm")


def _ppo_engine(dev, n_envs, seed=1, **kw):
    import deep_rl_amd as D

    env = D.make("CartPole-v1", num_envs=n_envs, device=dev, seed=seed)
    torch.manual_seed(seed)
    agent = D.ActorCritic(env)
    opt = D.ClipAdam(agent, lr=2.5e-4, eps=1e-5, max_grad_norm=0.5)
    kw.setdefault("max_episodes_logged", 65536)
    return D.PPOEngine(env, agent, opt, num_steps=T, **kw)


def _storage_to_oracle(R, eng):
    st = R.Storage(eng.T, eng.N)
    for n in ["observations", "values", "actions", "log_probs", "rewards", "dones", "advantages", "returns"]:
        getattr(st, n)[...] = getattr(eng, n).cpu().numpy()
    return st


# ------------------------------------------------------------------- PPO, config 2 -------------------------------------------------
def test_ppo_fullsize_rollout_vs_oracle(dev, R):
    """4096 envs x 128 steps, keyed RNG, one launch: the device's own actions are replayed on the oracle (a last-bit difference in
    a probability must not fork 524,288 trajectories); observations / rewards / dones / episode log bit-exact, values and log-probs
    within 3e-6, the fused GAE bit-exact given the device's values."""
    eng = _ppo_engine(dev, N_PPO, seed=1)
    rng = np.random.default_rng(3)
    params = (eng.agent.flat.cpu().numpy() + rng.normal(0, 0.2, 9155) * (np.arange(9155) >= 4480) * (np.arange(9155) < 4610)).astype(np.float32)
    eng.agent.load_flat(params)   # a larger actor head: action probabilities away from 1/2
    env = R.VecCartPole(N_PPO, seed=1)
    obs_cur = env.reset()
    assert np.array_equal(eng.reset().cpu().numpy(), obs_cur)
    st = R.Storage(T, N_PPO)
    for upd in range(2):   # the second rollout starts from carried-over env state, counters and RNG indices
        eng.rollout_gae()
        n_ep, eps = eng.drain_episodes()
        acts = eng.actions[:T].cpu().numpy()
        reps, rn = R.rollout(env, params, st, obs_cur, forced_actions=acts, max_ep=65536)
        assert np.array_equal(eng.observations.cpu().numpy(), st.observations), upd
        assert np.array_equal(eng.dones.cpu().numpy(), st.dones) and np.array_equal(eng.rewards.cpu().numpy(), st.rewards)
        assert np.abs(eng.values.cpu().numpy() - st.values).max() < 3e-6
        assert np.abs(eng.log_probs[:T].cpu().numpy() - st.log_probs[:T]).max() < 3e-6
        assert n_ep == rn and sorted(eps) == sorted(reps), (n_ep, rn)
        assert np.array_equal(eng.observation.cpu().numpy(), obs_cur)
        st.values[...] = eng.values.cpu().numpy()   # GAE has no tolerance of its own: same inputs -> same bits
        R.gae(st)
        assert np.array_equal(eng.advantages.cpu().numpy(), st.advantages) and np.array_equal(eng.returns.cpu().numpy(), st.returns)
        assert 0.2 < acts.mean() < 0.8 and n_ep > 1000


def test_ppo_fullsize_minibatch_grad_vs_oracle(dev, R):
    """One 131,072-row mi_ppo_minibatch_grad launch (512 workgroups, age-aware tile split, both nets) against the oracle's gradient of
    the same rows: <= 2e-5 of the largest element, loss terms to 3e-5; bitwise reproducible."""
    eng = _ppo_engine(dev, N_PPO, seed=2)
    eng.reset(); eng.rollout_gae()
    rng = np.random.default_rng(5)
    params = (eng.agent.flat.cpu().numpy() + rng.normal(0, 0.05, 9155)).astype(np.float32)
    eng.agent.load_flat(params)   # != behaviour parameters: ratio != 1, both clip branches taken
    eng.make_perm(0)
    eng.adv_stats()
    st = _storage_to_oracle(R, eng)
    mb = eng.minibatch_size
    assert mb == 131072
    for k in (0, 3):
        eng.minibatch_grad(k)
        idx = eng.perm[k * mb:(k + 1) * mb].cpu().numpy()
        og, ot = R.minibatch(params, st, idx)
        grads = eng.grads.cpu().numpy(); terms = eng.loss_terms.cpu().numpy()
        scale = np.abs(og).max()
        err = np.abs(grads - og)
        assert err.max() <= 2e-5 * scale, (k, err.max() / scale, int(err.argmax()))
        assert np.allclose(terms, ot, rtol=3e-5, atol=1e-5), (k, terms, ot)
        eng.minibatch_grad(k)
        assert np.array_equal(eng.grads.cpu().numpy(), grads)


def test_ppo_fullsize_update_vs_oracle(dev, R):
    """Two whole mi_ppo_update calls at 4096 envs (rollout + GAE, 4 x 4 minibatches of 131,072 rows, the clip + Adam steps riding on the
    gradient launches) against the oracle's ref_ppo_update with the same keys: parameters within 5e-6, loss terms to 1e-4.  (An action
    whose probability ties the uniform to the last bit may differ between the two: the episode counts may differ by a few.)"""
    eng = _ppo_engine(dev, N_PPO, seed=1)
    base = R.PPOBaseline(eng.agent.flat.cpu().numpy(), N_PPO, T=T, seed=1)
    eng.reset()
    assert np.array_equal(eng.observation.cpu().numpy(), base.obs)
    for u in range(2):
        lr = (1.0 - u / 10) * 2.5e-4
        eng.optimizer.param_groups[0]["lr"] = lr
        eng.update()
        n_ep, _ = eng.drain_episodes()
        rn = base.run_update(lr)
        assert abs(n_ep - rn) <= 4, (u, n_ep, rn)
        d = np.abs(eng.agent.flat.cpu().numpy() - base.params).max()
        assert d < 5e-6, (u, d)
        assert np.allclose(eng.loss_terms.cpu().numpy(), base.terms, rtol=1e-4, atol=1e-5), (eng.loss_terms.cpu().numpy(), base.terms)
    assert eng.optimizer.step_count == 32


@pytest.mark.parametrize("n_envs", [8, 64, N_PPO])
def test_ppo_update_equals_launch_sequence_bitwise(dev, n_envs, monkeypatch):
    """mi_ppo_update (owed optimizer steps applied by the next gradient launch's weight staging, state ping-ponging through the
    workspace) == the explicit sequence rollout_gae, perms_and_stats, 16 x {minibatch_grad, clip_adam} the sharded path walks:
    parameters, Adam moments, gradient norm and loss terms bit for bit after two updates."""
    import deep_rl_amd.engine as E

    outs = []
    for forced in (False, True):
        monkeypatch.setattr(E, "_FORCE_SHARDED_SEQUENCE", forced)
        eng = _ppo_engine(dev, n_envs, seed=4)
        eng.reset()
        for u in range(2):
            eng.optimizer.param_groups[0]["lr"] = (1.0 - u / 4) * 2.5e-4
            eng.update()
        o = eng.optimizer
        outs.append([t.clone() for t in (eng.agent.flat, o.exp_avg, o.exp_avg_sq, o.grad_norm, eng.loss_terms, eng.grads, eng.observations, eng.advantages)])
        assert o.step_count == 32
    for a, b in zip(*outs):
        assert torch.equal(a, b)


# ------------------------------------------------------------------- DQN, config 3 -------------------------------------------------
def test_dqn_fullsize_td_grad_vs_oracle(dev, R):
    """4096 envs x 256 slots = 1,048,576 transitions, filled by 300 acting steps (the ring has wrapped): the TD gradient of batches of
    128 (the reference's), 1,024, 4,096 (bench.py's scaled batch: the 16-row form of dqn_td_kernel, 256 workgroups x 1 group) and 4,100 rows (257 groups: a second,
    ragged group in workgroup 0) drawn by the keyed sampler from the whole ring, against the oracle on a copy of the ring."""
    import deep_rl_amd as D

    n, S = 4096, 256
    rng = np.random.default_rng(11)
    for batch in (128, 1024, 4096, 4100):
        env = D.make("CartPole-v1", num_envs=n, device=dev, seed=2)
        torch.manual_seed(2)
        q = D.QNetwork(env); tgt = D.QNetwork(env)
        params = (q.flat.cpu().numpy() + rng.normal(0, 0.05, 10934)).astype(np.float32)
        tparams = (params + rng.normal(0, 0.05, 10934)).astype(np.float32)
        q.load_flat(params); tgt.load_flat(tparams)
        eng = D.DQNEngine(env, q, tgt, D.ClipAdam(q, lr=2.5e-4, eps=1e-8), slots=S, batch_size=batch, learning_starts=100, total_timesteps=600,
                          max_episodes_logged=0)
        eng.reset()
        for _ in range(30):
            eng.act(10)
        assert eng.global_step == 300
        st = R.ReplayStorage(S, n)
        for name in ["observations", "actions", "rewards", "terminated"]:
            getattr(st, name)[...] = getattr(eng, name).cpu().numpy()
        assert st.terminated.sum() > 1000
        eng.sample()
        idx = eng.batch_inds.cpu().numpy()
        assert np.array_equal(idx, R.dqn_sample(2, 0, S * n, batch))
        eng.td_grad()
        og, ol = R.dqn_td_grads(params, tparams, st, idx)
        g = eng.grads.cpu().numpy()
        assert np.abs(g - og).max() <= 1e-5 * np.abs(og).max(), (batch, np.abs(g - og).max() / np.abs(og).max())
        assert abs(float(eng.loss.item()) - ol) <= 2e-5 * ol
        eng.td_grad()
        assert np.array_equal(eng.grads.cpu().numpy(), g)


def test_dqn_fullsize_act_vs_oracle(dev, R):
    """dqn_act4_kernel (the fourth wave computes both CartPole successors) at BASELINE config 3's size — 4096 envs x 256 slots, 300 steps in 30 launches, epsilon
    decaying through the run, the ring wrapping at step 256 — with the device's own actions replayed on the oracle's acting loop (reference dqn.py:86-108):
    observations / actions / rewards / terminated of the whole 1,048,576-transition ring, the carried-over observation and the episode log bit-exact after
    EVERY launch; the decisions themselves checked against the RNG contract and the oracle's Q-values on a sample of envs."""
    import deep_rl_amd as D

    n, S, ls, tt = 4096, 256, 100, 600
    env = D.make("CartPole-v1", num_envs=n, device=dev, seed=2)
    torch.manual_seed(2)
    q = D.QNetwork(env); tgt = D.QNetwork(env)
    rng = np.random.default_rng(11)
    params = (q.flat.cpu().numpy() + rng.normal(0, 0.05, 10934)).astype(np.float32)
    q.load_flat(params); tgt.load_flat(params)
    eng = D.DQNEngine(env, q, tgt, D.ClipAdam(q, lr=2.5e-4, eps=1e-8), slots=S, batch_size=128, learning_starts=ls, total_timesteps=tt, max_episodes_logged=8192)
    renv = R.VecCartPole(n, seed=2)
    st = R.ReplayStorage(S, n)
    obs_cur = renv.reset(); st.observations[0] = obs_cur
    assert np.array_equal(eng.reset().cpu().numpy(), obs_cur)
    gs, n_greedy, n_random, total_ep = 0, 0, 0, 0
    for call in range(30):
        obs_before = obs_cur.copy()
        eng.act(10)
        n_ep, eps = eng.drain_episodes()
        fa = np.stack([eng.actions[(gs + s) % S].cpu().numpy() for s in range(10)])
        reps, rn = R.dqn_act_steps_log(renv, params, st, obs_cur, 10, gs, learning_starts=ls, total_timesteps=tt, forced_actions=fa, max_ep=8192)
        for name in ["observations", "actions", "rewards", "terminated"]:
            assert np.array_equal(getattr(eng, name).cpu().numpy(), getattr(st, name)), (call, name)
        assert np.array_equal(eng.observation.cpu().numpy(), obs_cur), call
        assert n_ep == rn and sorted(eps) == sorted(reps), (call, n_ep, rn)
        total_ep += n_ep
        # the first decision of the launch against the contract (its observation is known: obs_before), 256 envs spread over the workgroups
        e_ = np.float32(R.dqn_epsilon(gs, total_timesteps=tt))
        qv = R.dqn_forward(params, obs_before)
        for e in range(0, n, 16):
            u, ra = R.dqn_explore_draw(2, e, gs)
            if gs < ls or u < e_:
                assert fa[0, e] == ra; n_random += 1
            elif abs(qv[e, 0] - qv[e, 1]) > 1e-4:
                assert fa[0, e] == int(qv[e, 1] > qv[e, 0]); n_greedy += 1
        gs += 10
    assert eng.global_step == 300 and total_ep > 20000 and st.terminated.sum() > 1000
    assert n_greedy > 1000 and n_random > 2000, (n_greedy, n_random)


# ------------------------------------------------------------------- SAC, config 4 -------------------------------------------------
def test_sac_fullsize_act_vs_oracle(dev, R):
    """sac_act_kernel at BASELINE config 4's size — 2048 Pendulum envs x 512 slots, 600 steps (the ring wraps at 512), 40 keyed warm-up steps then the actor with
    supplied normal draws (reference sac.py:138-158).  Per step: the device's actions against the oracle's actor sample on the (bit-identical) observation <= 5e-6
    (tanh-Gaussian through two 256-wide layers), then the oracle env is stepped with the DEVICE's actions: observations / rewards of the ring slot and the
    carried-over observation bit-exact, nothing terminated, episode boundaries (TimeLimit 200) on the same steps."""
    import deep_rl_amd as D

    n, S, ls, steps = 2048, 512, 40, 600
    env = D.make("Pendulum-v1", num_envs=n, device=dev, seed=6)
    torch.manual_seed(6)
    a = D.Actor(env)
    qs = [D.SoftQNetwork(env) for _ in range(4)]
    rng = np.random.default_rng(21)
    a_p = (a.flat.cpu().numpy() + rng.normal(0, 0.03, R.AC_NPARAMS)).astype(np.float32)
    a.load_flat(a_p)
    eng = D.SACEngine(env, a, *qs, slots=S, batch_size=256, learning_starts=ls, max_episodes_logged=0)
    ref = R.VecPendulum(n, seed=6)
    obs = ref.reset()
    assert np.array_equal(eng.reset().cpu().numpy(), obs)
    worst, n_done = 0.0, 0
    for t in range(steps):
        eps = rng.standard_normal(n).astype(np.float32)
        eng.act(forced_eps=torch.from_numpy(eps) if t >= ls else None)
        a_dev = eng.actions[t % S].cpu().numpy()
        if t >= ls:
            ra, _ = R.sac_actor_sample(a_p, obs, eps)
            worst = max(worst, float(np.abs(a_dev - ra).max()))
        else:
            assert a_dev.min() >= -2.0 and a_dev.max() < 2.0
        obs, rew, done, _, _ = ref.step(a_dev)
        n_done += int(done.sum())
        s = (t + 1) % S
        assert np.array_equal(eng.observations[s].cpu().numpy(), obs) and np.array_equal(eng.rewards[s].cpu().numpy(), rew), t
        assert np.array_equal(eng.observation.cpu().numpy(), obs), t
    assert worst <= 5e-6, worst      # measured 3.1e-6 over 1.1 M samples: ~13 float32 ulps of an action near +-2 (tanh of mean + std * eps behind two 256-wide layers)
    assert not eng.terminated.any() and n_done == 3 * n      # TimeLimit 200: every env finished exactly 3 episodes in 600 steps
    assert np.abs(eng.actions.cpu().numpy()).max() <= 2.0


def _rel(a, b):
    return np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64)).max() / max(np.abs(np.asarray(b, np.float64)).max(), 1e-30)


def test_sac_fullsize_grads_vs_oracle(dev, R):
    """2048 Pendulum envs x 512 slots = 1,048,576 transitions written by 600 production acting steps (the ring has wrapped): critic,
    actor and alpha gradients of a batch of 256 ring-resident rows against the oracle on a copy of the ring."""
    import deep_rl_amd as D

    n, S, batch = 2048, 512, 256
    env = D.make("Pendulum-v1", num_envs=n, device=dev, seed=6)
    torch.manual_seed(6)
    a = D.Actor(env)
    qf1, qf2, t1, t2 = D.SoftQNetwork(env), D.SoftQNetwork(env), D.SoftQNetwork(env), D.SoftQNetwork(env)
    t1.load_state_dict(qf1.state_dict()); t2.load_state_dict(qf2.state_dict())
    eng = D.SACEngine(env, a, qf1, qf2, t1, t2, slots=S, batch_size=batch, learning_starts=40, max_episodes_logged=0)
    eng.reset()
    for _ in range(600):
        eng.act()
    assert eng.global_step == 600
    rng = np.random.default_rng(9)
    a_p = (eng.actor.flat.cpu().numpy() + rng.normal(0, 0.02, R.AC_NPARAMS)).astype(np.float32)
    q_p = (eng.q_flat.cpu().numpy() + rng.normal(0, 0.02, 2 * R.SQ_NPARAMS)).astype(np.float32)
    qt_p = (q_p + rng.normal(0, 0.02, 2 * R.SQ_NPARAMS)).astype(np.float32)
    eng.actor.load_flat(a_p)
    eng.q_flat.copy_(torch.from_numpy(q_p).to(dev)); eng.qt_flat.copy_(torch.from_numpy(qt_p).to(dev))
    st = R.SacStorage(S, n)
    for name in ["observations", "actions", "rewards", "terminated"]:
        getattr(st, name)[...] = getattr(eng, name).cpu().numpy()
    eng.sample()   # keyed randint over the whole (full) ring
    idx = eng.batch_inds.cpu().numpy()
    assert idx.min() >= 0 and idx.max() < S * n and np.unique(idx // n).size > 100
    eps = rng.standard_normal((3, batch)).astype(np.float32)
    eng.alpha.fill_(0.37)
    eng.critic_grad(torch.from_numpy(eps[0]))
    g_ref, l_ref = R.sac_critic_grads(q_p, qt_p, a_p, st, idx, eps[0], 0.37)
    g = eng.q_grads.cpu().numpy()
    assert np.allclose(eng.q_losses.cpu().numpy(), l_ref, rtol=2e-5)
    for k in range(2):
        sl = slice(k * R.SQ_NPARAMS, (k + 1) * R.SQ_NPARAMS)
        assert _rel(g[sl], g_ref[sl]) < 2e-5, (k, _rel(g[sl], g_ref[sl]))
    eng.actor_grad(torch.from_numpy(eps[1]))
    ga_ref, loss_ref, mlp_ref = R.sac_actor_grads(a_p, q_p, st, idx, eps[1], 0.37)
    out = eng.actor_out.cpu().numpy()
    assert abs(out[0] - loss_ref) <= 2e-5 * max(1.0, abs(loss_ref)) and abs(out[1] - mlp_ref) <= 2e-5 * max(1.0, abs(mlp_ref))
    assert _rel(eng.actor_grads.cpu().numpy(), ga_ref) < 1e-4
    la = np.array([-0.3], np.float32); m = np.array([0.01], np.float32); v = np.array([0.002], np.float32)
    eng.log_alpha.copy_(torch.from_numpy(la)); eng._alpha_m.copy_(torch.from_numpy(m)); eng._alpha_v.copy_(torch.from_numpy(v)); eng.alpha_steps = 6
    eng.update_alpha(torch.from_numpy(eps[2]))
    mlp = R.sac_mean_logp(a_p, st, idx, eps[2])
    grad = np.array([-(mlp + -1.0)], np.float32)
    ao = eng.alpha_out.cpu().numpy()
    assert abs(ao[1] - grad[0]) <= 2e-5 * max(1.0, abs(grad[0]))
    R.adam_step(la, grad, m, v, 7, 1e-3, eps=1e-8)
    assert abs(float(eng.log_alpha) - la[0]) < 1e-6 and abs(float(eng.alpha) - np.exp(la[0])) < 1e-6
