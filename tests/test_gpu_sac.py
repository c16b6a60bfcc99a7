"""GPU parity tests of the SAC path (deep_rl_amd/csrc/mi_sac.hip through the C ABI) against the CPU oracle and the golden vectors of
the unmodified reference sac.py (tests/golden/sac_ref_trace.npz).  Env state, flags, indices and episode bookkeeping bit-exact;
fp32 tolerances written at each assert."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    return torch.device("cuda", 0)


@pytest.fixture(scope="module")
def R():
    from oracle import cpu_ref

    cpu_ref.lib().ref_set_num_threads(8)
    return cpu_ref


@pytest.fixture(scope="module")
def sac_trace():
    with np.load(os.path.join(ROOT, "tests", "golden", "sac_ref_trace.npz")) as z:
        return {k: z[k] for k in z.files}


@pytest.fixture(autouse=True)
def _fdlibm(R):
    R.set_sincos_mode("fdlibm")
    yield
    R.set_sincos_mode("libm")


def _engine(dev, n_envs, slots, seed=1, base=0, actor=None, q=None, qt=None, **kw):
    import deep_rl_amd as D

    env = D.make("Pendulum-v1", num_envs=n_envs, device=dev, seed=seed, env_id_base=base)
    torch.manual_seed(seed)
    a = D.Actor(env)
    qf1, qf2, t1, t2 = D.SoftQNetwork(env), D.SoftQNetwork(env), D.SoftQNetwork(env), D.SoftQNetwork(env)
    t1.load_state_dict(qf1.state_dict()); t2.load_state_dict(qf2.state_dict())
    kw.setdefault("max_episodes_logged", 4096)
    eng = D.SACEngine(env, a, qf1, qf2, t1, t2, slots=slots, **kw)
    if actor is not None:
        a.load_flat(actor)
    if q is not None:
        eng.q_flat.copy_(torch.from_numpy(np.ascontiguousarray(q, np.float32).reshape(-1)).to(dev))
        eng.qt_flat.copy_(torch.from_numpy(np.ascontiguousarray(q if qt is None else qt, np.float32).reshape(-1)).to(dev))
    return eng


def _upload(eng, st):
    eng.observations.copy_(torch.from_numpy(st.observations)); eng.actions.copy_(torch.from_numpy(st.actions))
    eng.rewards.copy_(torch.from_numpy(st.rewards)); eng.terminated.copy_(torch.from_numpy(st.terminated))


def _random_storage(R, rng, slots, n):
    st = R.SacStorage(slots, n)
    th = rng.uniform(-np.pi, np.pi, (slots, n))
    st.observations[..., 0] = np.cos(th); st.observations[..., 1] = np.sin(th); st.observations[..., 2] = rng.uniform(-8, 8, (slots, n))
    st.actions[:] = rng.uniform(-2, 2, (slots, n)); st.rewards[:] = -rng.uniform(0, 16, (slots, n))
    st.terminated[:] = rng.random((slots, n)) < 0.1   # Pendulum never sets it; the kernel must still honour it
    return st


def _rand_nets(R, rng, scale=1.0):
    """torch-default-like init, optionally scaled up so that heads / ReLU masks are exercised away from the origin"""
    def lin(o, i):
        b = 1.0 / np.sqrt(i)
        return [rng.uniform(-b, b, (o, i)).astype(np.float32) * scale, rng.uniform(-b, b, o).astype(np.float32)]
    qs = []
    for _ in range(2):
        qs.append(np.concatenate([x.reshape(-1) for x in lin(256, 4) + lin(256, 256) + lin(1, 256)]))
    actor = np.concatenate([x.reshape(-1) for x in lin(256, 3) + lin(256, 256) + lin(1, 256) + lin(1, 256)])
    assert qs[0].size == R.SQ_NPARAMS and actor.size == R.AC_NPARAMS
    return actor, np.concatenate(qs)


# ---------------------------------------------------------------- env ---------------------------------------------------------------
def test_pendulum_stepper_bit_exact(dev, R):
    """450 steps of 777 envs (two TimeLimit truncations each) with keyed resets: every observation, reward, flag and episode
    statistic equals the oracle's bit for bit; a second pass covers large angles through forced states."""
    import deep_rl_amd as D

    n = 777
    env = D.make("Pendulum-v1", num_envs=n, device=dev, seed=5, env_id_base=1000)
    assert env.observation_space.shape == (3,) and env.action_space.shape == (1,) and env.spec.max_episode_steps == 200
    ref = R.VecPendulum(n, seed=5, env_id_base=1000)
    o = env.reset(); ro = ref.reset()
    assert np.array_equal(o.cpu().numpy(), ro)
    rng = np.random.default_rng(0)
    n_done = 0
    for t in range(450):
        a = rng.uniform(-2.5, 2.5, n).astype(np.float32)    # beyond +-2: the env clips
        o, r, d, info = env.step(torch.from_numpy(a).to(dev).reshape(n, 1))
        ro, rr, rd, fret, flen = ref.step(a)
        assert np.array_equal(o.cpu().numpy(), ro), t
        assert np.array_equal(r.cpu().numpy(), rr), t
        assert np.array_equal(d.cpu().numpy(), rd.astype(bool)) and np.array_equal(info["TimeLimit.truncated"].cpu().numpy(), rd.astype(bool))
        assert np.array_equal(info["episode"]["r"].cpu().numpy(), fret) and np.array_equal(info["episode"]["l"].cpu().numpy(), flen)
        n_done += int(rd.sum())
    assert n_done == 2 * n
    st, el = env.get_state()
    assert np.array_equal(st.cpu().numpy(), ref.state) and (el.cpu().numpy() == 50).all()
    big = np.stack([rng.uniform(-60, 60, n), rng.uniform(-8, 8, n)], 1)
    o = env.reset(torch.from_numpy(big)); ro = ref.reset(big)
    assert np.array_equal(o.cpu().numpy(), ro)
    for t in range(20):
        a = rng.uniform(-2, 2, n).astype(np.float32)
        o, r, _, _ = env.step(torch.from_numpy(a).to(dev))
        ro, rr, _, _, _ = ref.step(a)
        assert np.array_equal(o.cpu().numpy(), ro) and np.array_equal(r.cpu().numpy(), rr), t


def test_env_kind_errors(dev):
    import deep_rl_amd as D
    from deep_rl_amd import _native as N

    cp = D.make("CartPole-v1", num_envs=4, device=dev)
    pd = D.make("Pendulum-v1", num_envs=4, device=dev)
    with pytest.raises(N.MiError):
        pd.reset(); N.check(N.lib().mi_env_step(pd.handle, N.ptr(torch.zeros(4, dtype=torch.int64, device=dev)), None, N.ptr(pd._obs), N.ptr(pd._reward),
                                                N.ptr(pd._done), N.ptr(pd._trunc), N.ptr(pd._fret), N.ptr(pd._flen), N.stream_ptr(dev)), "mi_env_step")
    with pytest.raises(N.MiError):
        N.check(N.lib().mi_env_step_cont(cp.handle, N.ptr(torch.zeros(4, device=dev)), None, N.ptr(pd._obs), N.ptr(pd._reward), N.ptr(pd._done),
                                         N.ptr(pd._trunc), N.ptr(pd._fret), N.ptr(pd._flen), N.stream_ptr(dev)), "mi_env_step_cont")
    with pytest.raises(N.MiError):
        D.Actor(cp)


# ---------------------------------------------------------------- modules -----------------------------------------------------------
def test_modules_surface_and_forward(dev, R):
    import deep_rl_amd as D

    env = D.make("Pendulum-v1", num_envs=1, device=dev)
    torch.manual_seed(1)
    actor = D.Actor(env); qf = D.SoftQNetwork(env)
    assert [tuple(p.shape) for p in actor.parameters()] == [(256, 3), (256,), (256, 256), (256,), (1, 256), (1,), (1, 256), (1,)]
    assert [tuple(p.shape) for p in qf.parameters()] == [(256, 4), (256,), (256, 256), (256,), (1, 256), (1,)]
    assert float(actor.action_scale) == 2.0 and float(actor.action_bias) == 0.0
    rng = np.random.default_rng(0)
    for scale in (1.0, 3.0):
        a_p, q_p = _rand_nets(R, rng, scale)
        actor.load_flat(a_p); qf.load_flat(q_p[:R.SQ_NPARAMS])
        n = 1003    # not a multiple of the 8-row group
        th = rng.uniform(-np.pi, np.pi, n)
        obs = np.stack([np.cos(th), np.sin(th), rng.uniform(-8, 8, n)], 1).astype(np.float32)
        eps = rng.standard_normal(n).astype(np.float32)
        act, logp = actor.get_action(torch.from_numpy(obs).to(dev), torch.from_numpy(eps))
        assert act.shape == (n, 1) and logp.shape == (n,)
        ra, rl = R.sac_actor_sample(a_p, obs, eps)
        # action = 2 tanh(z): absolute; logp contains log(2 (1 - u^2) + 1e-6) whose conditioning degrades as |u| -> 1
        assert np.abs(act.cpu().numpy()[:, 0] - ra).max() < 2e-5
        cond = 1.0 / np.maximum(1.0 - (ra / 2.0) ** 2, 1e-6)
        assert (np.abs(logp.cpu().numpy() - rl) <= 2e-5 * np.maximum(1.0, np.abs(rl)) + 1e-6 * cond).all()
        a = rng.uniform(-2, 2, n).astype(np.float32)
        qv = qf(torch.from_numpy(obs).to(dev), torch.from_numpy(a).to(dev).reshape(n, 1))
        rq = R.sac_q_forward(q_p[:R.SQ_NPARAMS], obs, a)
        assert qv.shape == (n,) and np.abs(qv.cpu().numpy() - rq).max() < 1e-5 * max(1.0, np.abs(rq).max())
    t = D.SoftQNetwork(env)
    t.load_state_dict(qf.state_dict())
    assert torch.equal(t.flat, qf.flat) and t.flat.data_ptr() != qf.flat.data_ptr()
    joint = D.pack(qf, t)
    assert joint.numel() == 2 * R.SQ_NPARAMS and qf.network[0].weight.data_ptr() == joint.data_ptr()
    assert t.network[0].weight.data_ptr() == joint.data_ptr() + 4 * R.SQ_NPARAMS


# ---------------------------------------------------------------- acting ------------------------------------------------------------
def test_act_step_teacher_forced_vs_reference_trace(dev, R, sac_trace):
    """The reference run's first 1,000 steps (its actions, its reset noise) through mi_sac_act_step: ring storage bit-exact vs the
    oracle, episode returns those the reference printed."""
    from tests.test_oracle_sac_pinned import _fill_storage

    g = sac_trace
    T = 1000
    st, ep = _fill_storage(g, T)
    eng = _engine(dev, 1, 30_001)
    eng.reset(torch.from_numpy(g["reset_states"][:1]))
    ar, resets, acts = g["after_reset_all"], g["reset_states"], g["actions_all"]
    ri = 1
    got_ep = []
    for gs in range(T):
        fr = None
        if ar[gs + 1]:
            fr = torch.from_numpy(resets[ri:ri + 1]); ri += 1
        eng.act(forced_actions=torch.from_numpy(acts[gs:gs + 1]), forced_resets=fr)
        if (gs + 1) % 200 == 0:
            got_ep += [(gs + 1, r) for (_e, r, _l) in eng.drain_episodes()]
    assert np.array_equal(eng.observations[:T + 1].cpu().numpy(), st.observations[:T + 1])
    assert np.array_equal(eng.actions[:T].cpu().numpy(), st.actions[:T])
    assert np.array_equal(eng.rewards[:T + 1].cpu().numpy(), st.rewards[:T + 1])
    assert not eng.terminated.any()
    assert [e[0] for e in got_ep] == [e[0] for e in ep] == list(g["episode_global_step"][:5])
    assert np.array_equal(np.array([e[1] for e in got_ep], np.float32), np.array([e[1] for e in ep], np.float32))
    assert np.abs(np.array([e[1] for e in got_ep]) - g["episode_return"][:5]).max() < 0.02


def test_act_step_policy_with_forced_noise(dev, R):
    """actor.get_action inside the acting kernel (global_step >= learning_starts) with supplied normal draws: actions match the oracle's
    actor to 2e-5; stepping the oracle env with the DEVICE actions reproduces the ring bit for bit (ring wrap included)."""
    n, slots, T = 203, 7, 12
    rng = np.random.default_rng(3)
    a_p, q_p = _rand_nets(R, rng, 2.0)
    eng = _engine(dev, n, slots, seed=9, base=50, actor=a_p, q=q_p, learning_starts=0)
    ref = R.VecPendulum(n, seed=9, env_id_base=50)
    obs = ref.reset()
    assert np.array_equal(eng.reset().cpu().numpy(), obs)
    for t in range(T):
        eps = rng.standard_normal(n).astype(np.float32)
        eng.act(forced_eps=torch.from_numpy(eps))
        a_dev = eng.actions[t % slots].cpu().numpy()
        ra, _ = R.sac_actor_sample(a_p, obs, eps)
        assert np.abs(a_dev - ra).max() < 2e-5, t
        obs, rew, done, _, _ = ref.step(a_dev)
        s = (t + 1) % slots
        assert np.array_equal(eng.observations[s].cpu().numpy(), obs) and np.array_equal(eng.rewards[s].cpu().numpy(), rew), t
        assert np.array_equal(eng.observation.cpu().numpy(), obs)


def test_act_step_keyed_draws(dev, R):
    """Production mode: warm-up actions are keyed uniforms in [-2, 2) (env.action_space.sample(), sac.py:139), policy noise is keyed
    standard normal; both are reproducible functions of (seed, global env id, global_step) — independent of the sharding."""
    n, slots = 4096, 40
    rng = np.random.default_rng(4)
    a_p, q_p = _rand_nets(R, rng, 1.0)
    eng = _engine(dev, n, slots, seed=3, actor=a_p, q=q_p, learning_starts=16)
    eng.reset()
    for _ in range(32):
        eng.act()
    acts = eng.actions[:32].cpu().numpy()
    obs = eng.observations[:33].cpu().numpy()
    u = acts[:16]
    assert u.min() >= -2.0 and u.max() < 2.0 and abs(u.mean()) < 0.02 and abs(u.std() - 4 / np.sqrt(12)) < 0.02
    assert np.unique(u).size > 0.99 * u.size * 0.9
    # recover the normal draws of the policy phase from the oracle's mean / std at the stored observations
    z = []
    for t in range(16, 32):
        a0, _ = R.sac_actor_sample(a_p, obs[t], np.zeros(n, np.float32))
        a1, _ = R.sac_actor_sample(a_p, obs[t], np.ones(n, np.float32))
        m0, m1 = np.arctanh(np.clip(a0 / 2.0, -0.999999, 0.999999)), np.arctanh(np.clip(a1 / 2.0, -0.999999, 0.999999))
        za = np.arctanh(np.clip(acts[t].astype(np.float64) / 2.0, -0.999999, 0.999999))
        ok = (np.abs(acts[t]) < 1.99) & (np.abs(a1) < 1.99)
        z.append(((za - m0) / (m1 - m0))[ok])
    z = np.concatenate(z)
    assert z.size > 0.9 * 16 * n and abs(z.mean()) < 0.02 and abs(z.std() - 1.0) < 0.02 and abs((z ** 3).mean()) < 0.05 and abs((z ** 4).mean() - 3.0) < 0.15
    # two shards of 2048 envs reproduce the same streams
    for half in (0, 1):
        e2 = _engine(dev, n // 2, slots, seed=3, base=half * (n // 2), actor=a_p, q=q_p, learning_starts=16)
        e2.reset()
        for _ in range(20):
            e2.act()
        sl = slice(half * (n // 2), (half + 1) * (n // 2))
        assert np.array_equal(e2.actions[:20].cpu().numpy(), acts[:20, sl]) and np.array_equal(e2.observations[:21].cpu().numpy(), obs[:21, sl])


# ---------------------------------------------------------------- updates -----------------------------------------------------------
def _rel(a, b):
    return np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64)).max() / max(np.abs(np.asarray(b, np.float64)).max(), 1e-30)


# 13 / 256 / 400: the one-launch dW2 + optimizer-step form (sac_dw2_adam_kernel: 1 / 8 / 12-13 k-steps per wave); 1000 / 2100: split-K GEMM + assembly launch
@pytest.mark.parametrize("batch,scale", [(256, 1.0), (13, 2.0), (400, 1.0), (1000, 2.0), (2100, 1.0)])
def test_critic_grad_vs_oracle(dev, R, batch, scale):
    rng = np.random.default_rng(batch)
    n, slots = 5, 64
    a_p, q_p = _rand_nets(R, rng, scale)
    _, qt_p = _rand_nets(R, rng, scale)
    st = _random_storage(R, rng, slots, n)
    eng = _engine(dev, n, slots, actor=a_p, q=q_p, qt=qt_p, batch_size=batch)
    _upload(eng, st)
    idx = rng.integers(0, slots * n, batch)      # includes the last slot: "next" wraps around the ring
    idx[:2] = [slots * n - 1, (slots - 1) * n]
    eps = rng.standard_normal(batch).astype(np.float32)
    eng.alpha.fill_(0.37)
    eng.sample(idx); eng.critic_grad(torch.from_numpy(eps))
    g_ref, l_ref = R.sac_critic_grads(q_p, qt_p, a_p, st, idx, eps, 0.37)
    g = eng.q_grads.cpu().numpy(); l = eng.q_losses.cpu().numpy()
    assert np.allclose(l, l_ref, rtol=2e-5), (l, l_ref)
    for k in range(2):   # per critic: fp32 sums over `batch` rows in a different (fixed) order
        sl = slice(k * R.SQ_NPARAMS, (k + 1) * R.SQ_NPARAMS)
        assert _rel(g[sl], g_ref[sl]) < 2e-5, (k, _rel(g[sl], g_ref[sl]))
    # bitwise reproducible
    eng.critic_grad(torch.from_numpy(eps))
    assert np.array_equal(eng.q_grads.cpu().numpy(), g)


@pytest.mark.parametrize("batch,scale", [(256, 1.0), (13, 2.0), (400, 1.0), (1000, 2.0), (2100, 1.0)])
def test_actor_grad_and_alpha_vs_oracle(dev, R, batch, scale):
    rng = np.random.default_rng(100 + batch)
    n, slots = 3, 50
    a_p, q_p = _rand_nets(R, rng, scale)
    st = _random_storage(R, rng, slots, n)
    eng = _engine(dev, n, slots, actor=a_p, q=q_p, batch_size=batch)
    _upload(eng, st)
    idx = rng.integers(0, slots * n, batch)
    eps = rng.standard_normal(batch).astype(np.float32)
    eng.alpha.fill_(0.21)
    eng.sample(idx); eng.actor_grad(torch.from_numpy(eps))
    g_ref, loss_ref, mlp_ref = R.sac_actor_grads(a_p, q_p, st, idx, eps, 0.21)
    g = eng.actor_grads.cpu().numpy(); out = eng.actor_out.cpu().numpy()
    assert abs(out[0] - loss_ref) <= 2e-5 * max(1.0, abs(loss_ref)) and abs(out[1] - mlp_ref) <= 2e-5 * max(1.0, abs(mlp_ref))
    # min(q1, q2) is a discontinuous selector: a near-tie row may pick the other critic, so compare in the gradient's own scale
    assert _rel(g, g_ref) < 1e-4, _rel(g, g_ref)
    eng.actor_grad(torch.from_numpy(eps))
    assert np.array_equal(eng.actor_grads.cpu().numpy(), g)
    # alpha step: fresh log-probs -> gradient -> Adam on log_alpha -> alpha = exp(log_alpha), all on the device
    eps2 = rng.standard_normal(batch).astype(np.float32)
    la = np.array([-0.3], np.float32); m = np.array([0.01], np.float32); v = np.array([0.002], np.float32)
    eng.log_alpha.copy_(torch.from_numpy(la)); eng._alpha_m.copy_(torch.from_numpy(m)); eng._alpha_v.copy_(torch.from_numpy(v)); eng.alpha_steps = 6
    eng.update_alpha(torch.from_numpy(eps2))
    mlp = R.sac_mean_logp(a_p, st, idx, eps2)
    grad = np.array([-(mlp + -1.0)], np.float32)
    ao = eng.alpha_out.cpu().numpy()
    assert abs(ao[1] - grad[0]) <= 2e-5 * max(1.0, abs(grad[0])) and abs(ao[0] - (-la[0] * (mlp - 1.0))) <= 2e-5 * max(1.0, abs(mlp))
    R.adam_step(la, grad, m, v, 7, 1e-3, eps=1e-8)
    assert abs(float(eng.log_alpha) - la[0]) < 1e-6 and abs(float(eng.alpha) - np.exp(la[0])) < 1e-6
    assert abs(float(eng._alpha_m) - m[0]) < 1e-6 * max(1, abs(m[0])) and abs(float(eng._alpha_v) - v[0]) < 1e-6 * max(1, abs(v[0]))


def test_adam_and_polyak_vs_oracle(dev, R):
    import deep_rl_amd as D

    rng = np.random.default_rng(8)
    n = 2 * R.SQ_NPARAMS
    p = rng.standard_normal(n).astype(np.float32); t = rng.standard_normal(n).astype(np.float32)
    flat = torch.from_numpy(p.copy()).to(dev)
    opt = D.Adam(flat, lr=1e-3)
    m = np.zeros(n, np.float32); v = np.zeros(n, np.float32)
    for step in range(1, 4):
        g = (rng.standard_normal(n) * 10.0 ** rng.uniform(-6, 1, n)).astype(np.float32)
        opt.step(torch.from_numpy(g).to(dev))
        R.adam_step(p, g, m, v, step, 1e-3, eps=1e-8)
        assert np.abs(flat.cpu().numpy() - p).max() <= 1e-9 + 2e-7 * np.abs(p).max()
        assert _rel(opt.exp_avg.cpu().numpy(), m) < 1e-6 and _rel(opt.exp_avg_sq.cpu().numpy(), v) < 1e-6
    from deep_rl_amd import _native as N
    tt = torch.from_numpy(t.copy()).to(dev)
    N.check(N.lib().mi_polyak(N.ptr(tt), N.ptr(flat), n, 0.005, N.stream_ptr(dev)), "mi_polyak")
    want = t.copy(); R.polyak(want, flat.cpu().numpy(), 0.005)
    assert np.array_equal(tt.cpu().numpy(), want)


def test_fused_updates_equal_grad_then_adam(dev, R):
    """mi_sac_critic_update / mi_sac_actor_update / fused alpha step == the *_grad call followed by mi_adam (and mi_polyak), bit for bit."""
    from deep_rl_amd import _native as N

    rng = np.random.default_rng(77)
    n, slots, batch = 4, 40, 300
    a_p, q_p = _rand_nets(R, rng, 1.5)
    _, qt_p = _rand_nets(R, rng, 1.5)
    st = _random_storage(R, rng, slots, n)
    idx = rng.integers(0, slots * n, batch)
    eps = [torch.from_numpy(rng.standard_normal(batch).astype(np.float32)) for _ in range(3)]
    engs = []
    for fused in (True, False):
        eng = _engine(dev, n, slots, actor=a_p, q=q_p, qt=qt_p, batch_size=batch)
        _upload(eng, st)
        eng.alpha.fill_(0.3)
        eng.sample(idx)
        for it in range(2):      # two rounds so that the Adam moments and step counts matter
            if fused:
                eng.update_critic(eps[0], polyak=True); eng.update_actor(eps[1]); eng.update_alpha(eps[2])
            else:
                eng.critic_grad(eps[0]); eng.q_optimizer.step(eng.q_grads); eng.update_targets(); eng.update_index += 1
                eng.actor_grad(eps[1]); eng.actor_optimizer.step(eng.actor_grads)
                eng.alpha_steps += 1
                N.check(N.lib().mi_sac_mean_logp(N.ptr(eng.actor.flat), N.ptr(eng.observations), N.ptr(eng.batch_inds), batch, N.ptr(eps[2].to(dev)), 1, 0, 1.0 / batch,
                                                 N.ptr(eng._mean_logp), N.ptr(eng.workspace), N.stream_ptr(dev)), "mi_sac_mean_logp")
                N.check(N.lib().mi_sac_alpha_adam(N.ptr(eng._mean_logp), eng.target_entropy, N.ptr(eng.log_alpha), N.ptr(eng._alpha_m), N.ptr(eng._alpha_v), eng.alpha_steps,
                                                  eng.alpha_lr, N.ptr(eng.alpha), N.ptr(eng.alpha_out), N.stream_ptr(dev)), "mi_sac_alpha_adam")
                eng.actor_updates += 1
        engs.append(eng)
    f, u = engs
    for name in ("q_flat", "qt_flat", "q_grads", "q_losses", "actor_grads", "actor_out", "log_alpha", "alpha", "alpha_out"):
        assert torch.equal(getattr(f, name), getattr(u, name)), name
    assert torch.equal(f.actor.flat, u.actor.flat)
    assert torch.equal(f.q_optimizer.exp_avg_sq, u.q_optimizer.exp_avg_sq) and torch.equal(f.actor_optimizer.exp_avg, u.actor_optimizer.exp_avg)
    assert not torch.equal(f.qt_flat, torch.from_numpy(qt_p).to(dev))      # the polyak step happened


def test_in_launch_sampling_equals_sample_call(dev, R):
    """train_step() lets the critic launch draw the batch indices (keyed contract); == sample() + the same updates, bit for bit."""
    engs = []
    for in_launch in (True, False):
        eng = _engine(dev, 64, 128, seed=2, batch_size=256, learning_starts=10, max_episodes_logged=0)
        eng.reset()
        for _ in range(60):
            eng.act()
        for it in range(3):
            eng.global_step += 0
            if in_launch:
                eng.train_step()
            else:
                eng.sample(); eng.update_critic(polyak=True)
                if eng.global_step % 2 == 0:
                    for _ in range(2):
                        eng.update_actor(); eng.update_alpha()
            eng.act()
        engs.append(eng)
    a, b = engs
    for name in ("batch_inds", "q_flat", "qt_flat", "q_losses", "log_alpha"):
        assert torch.equal(getattr(a, name), getattr(b, name)), name
    assert torch.equal(a.actor.flat, b.actor.flat)


# ---------------------------------------------------------------- against the reference --------------------------------------------
def test_chained_first_30_steps_on_device(dev, R, sac_trace):
    """The reference run's global steps 5000..5029 on the DEVICE (30 critic, 30 actor, 30 alpha updates, 30 polyak steps chained through the
    engine's own Adam, the reference's recorded noise and indices): every loss, alpha, gradient summary and parameter checksum
    tracks the numbers the reference's autograd produced — the same bars the CPU oracle is pinned with."""
    from tests.test_oracle_sac_pinned import _fill_storage, summarize, close_summary, CHAIN

    R.set_sincos_mode("libm")
    g = sac_trace
    st, _ = _fill_storage(g, 5_125)
    eng = _engine(dev, 1, 30_001, actor=g["init_actor"], q=g["init_q"])
    _upload(eng, st)
    noise, lens = g["noise_chain"], g["noise_chain_lens"]
    off = np.concatenate([[0], np.cumsum(lens)])
    ni = 0

    def draw(n):
        nonlocal ni
        assert lens[ni] == n
        e = noise[off[ni]:off[ni + 1]]; ni += 1
        return torch.from_numpy(np.ascontiguousarray(e, np.float32))

    ka = 0
    ql, al, als = g["q_losses"], g["actor_losses"], g["alpha_steps"]
    for k in range(CHAIN):
        gs = 5000 + k
        if k > 0:
            a, _ = eng.actor.get_action(eng.observations[gs - 1, 0], draw(1))
            assert abs(float(a) - g["actions_all"][gs - 1]) < 2e-5, gs
        eng.global_step = gs
        eng.sample(g["chain_inds"][k])
        eng.critic_grad(draw(256))
        assert np.allclose(eng.q_losses.cpu().numpy(), ql[k, :2], rtol=3e-5, atol=1e-6), (k, eng.q_losses, ql[k, :2])
        assert abs(float(eng.alpha) - ql[k, 2]) < 1e-6
        assert close_summary(summarize(eng.q_grads.cpu().numpy()), g["chain_q_gradsum"][k], 2e-5), k
        eng.q_optimizer.step(eng.q_grads); eng.update_index += 1
        assert abs(eng.q_flat.double().sum().item() - ql[k, 3]) < 1e-4, k
        if gs % 2 == 0:
            for _ in range(2):
                eng.actor_grad(draw(256))
                assert al[ka, 0] == gs and abs(float(eng.actor_out[0]) - al[ka, 1]) <= 3e-5 * max(1.0, abs(al[ka, 1])), (k, eng.actor_out, al[ka])
                assert close_summary(summarize(eng.actor_grads.cpu().numpy()), g["chain_actor_gradsum"][ka], 3e-4), (k, ka)
                eng.actor_optimizer.step(eng.actor_grads)
                assert abs(eng.actor.flat.double().sum().item() - al[ka, 3]) < 2e-3
                assert abs(float(eng.log_alpha) - als[ka, 2]) < 2e-6
                eng.update_alpha(draw(256))
                assert abs(float(eng.alpha_out[1]) - als[ka, 3]) < 3e-5 * max(1.0, abs(als[ka, 3])), (k, eng.alpha_out, als[ka])
                ka += 1
        eng.update_targets()
    assert ka == CHAIN


def test_late_checkpoint_on_device(dev, R, sac_trace):
    """Global step 25,000 of the reference run, un-chained: the reference's parameters, indices and noise -> device losses and gradient
    summaries against the reference's autograd."""
    from tests.test_oracle_sac_pinned import _fill_storage, summarize, close_summary

    R.set_sincos_mode("libm")
    g = sac_trace
    gs = int(g["ck_gs"][0])
    st, _ = _fill_storage(g, gs)
    eng = _engine(dev, 1, 30_001, actor=g["ck_actor_params"][0], q=g["ck_q_params"][0], qt=g["ck_q_target"][0])
    _upload(eng, st)
    eng.sample(g["ck_inds"][0])
    eng.alpha.fill_(float(g["ck_alpha"][0]))
    eng.critic_grad(torch.from_numpy(g["ck_noise_critic"][0]))
    assert np.allclose(eng.q_losses.cpu().numpy(), g["ck_q_losses"][0], rtol=3e-5)
    assert close_summary(summarize(eng.q_grads.cpu().numpy()), g["ck_q_gradsum"][0], 2e-5)
    eng.q_flat.copy_(torch.from_numpy(np.ascontiguousarray(g["ck_actor_qparams"][0]).reshape(-1)).to(dev))
    eng.alpha.fill_(float(g["ck_actor_alpha"][0]))
    eng.actor_grad(torch.from_numpy(g["ck_noise_actor"][0]))
    assert abs(float(eng.actor_out[0]) - g["ck_actor_loss"][0]) <= 3e-5 * abs(g["ck_actor_loss"][0])
    assert close_summary(summarize(eng.actor_grads.cpu().numpy()), g["ck_actor_gradsum"][0], 5e-5)


def test_train_step_learns_pendulum(dev, R):
    """Production path end to end (keyed draws everywhere): 64 envs, 6,000 time steps, one update per time step as in the reference
    loop.  The reference run on this env climbs from about -6.5 reward per step (random) to about -1 within 5,000 updates
    (tests/golden/sac_ref_trace.npz episode returns); here the mean reward of the last 200 steps must beat the warm-up level by 2."""
    T = 6000
    eng = _engine(dev, 64, T + 1, seed=1, batch_size=256, learning_starts=200, max_episodes_logged=0)
    eng.reset()
    for t in range(T):
        eng.act()
        if eng.global_step >= eng.learning_starts:
            eng.train_step()
    torch.cuda.synchronize()
    assert torch.isfinite(eng.q_flat).all() and torch.isfinite(eng.actor.flat).all() and torch.isfinite(eng.alpha).all()
    assert 0.0 < float(eng.alpha) < 1.0          # entropy coefficient annealed from exp(0)
    r_rand = eng.rewards[1:200].mean().item(); r_last = eng.rewards[T - 200:T].mean().item()
    assert r_last > r_rand + 2.0, (r_rand, r_last)


def test_config4_full_size_properties(dev, R):
    """BASELINE config 4 at full size (2048 envs, 512-slot ring, batch 256; the ring wraps): size-independent properties.
    (a) physics: re-stepping a stored observation with the stored action lands on the stored successor (float32 storage of the
        float64 angle: cos / sin / speed within 2e-5) and reproduces the stored reward; nothing is ever `terminated`;
    (b) 200 production iterations keep every parameter finite, alpha falls from 1, fused == unfused updates at this size;
    (c) the critic / actor gradients are additive over the batch (two halves at half weight sum to the whole)."""
    n, S = 2048, 512
    eng = _engine(dev, n, S, seed=6, batch_size=256, learning_starts=40, max_episodes_logged=0)
    eng.reset()
    for _ in range(600):
        eng.act()
        if eng.global_step >= 400:
            eng.train_step()
    torch.cuda.synchronize()
    assert torch.isfinite(eng.q_flat).all() and torch.isfinite(eng.actor.flat).all() and 0.0 < float(eng.alpha) < 1.0
    obs = eng.observations.cpu().numpy(); act = eng.actions.cpu().numpy(); rew = eng.rewards.cpu().numpy()
    assert not eng.terminated.any()
    head = 600 % S
    rng = np.random.default_rng(3)
    checked = 0
    for _ in range(3000):
        s, e = int(rng.integers(0, S)), int(rng.integers(0, n))
        if (s + 1) % S == head or s == head:
            continue
        o = obs[s, e].astype(np.float64)
        th, thd = np.arctan2(o[1], o[0]), o[2]
        u = float(np.clip(act[s, e], -2, 2))
        cost = th ** 2 + 0.1 * thd ** 2 + 0.001 * u ** 2
        nthd = np.clip(thd + (15.0 * np.sin(th) + 3.0 * u) * 0.05, -8, 8)
        nth = th + nthd * 0.05
        want = np.array([np.cos(nth), np.sin(nth), nthd])
        got = obs[(s + 1) % S, e]
        if np.abs(want - got).max() < 2e-5:            # (a reset after TimeLimit puts an unrelated state there: 1 slot in 200)
            assert abs(-cost - rew[(s + 1) % S, e]) < 2e-4 * max(1.0, cost)
            checked += 1
    assert checked > 2700
    # (b) fused == unfused, (c) additivity, on this ring
    idx = rng.integers(0, S * n, 256); idx = idx[(idx // n + 1) % S != head]
    idx = np.concatenate([idx, idx[:256 - idx.size]])
    eps = torch.from_numpy(rng.standard_normal((2, 256)).astype(np.float32))
    a0, q0, t0 = eng.actor.flat.clone(), eng.q_flat.clone(), eng.qt_flat.clone()
    eng.sample(idx)
    eng.critic_grad(eps[0]); gq = eng.q_grads.clone(); lq = eng.q_losses.clone()
    eng.actor_grad(eps[1]); ga = eng.actor_grads.clone()
    half = _engine(dev, n, S, seed=6, batch_size=128, actor=a0.cpu().numpy(), q=q0.cpu().numpy(), qt=t0.cpu().numpy(), max_episodes_logged=0)
    for name in ("observations", "actions", "rewards", "terminated"):
        getattr(half, name).copy_(getattr(eng, name))
    half.alpha.copy_(eng.alpha)
    accq = torch.zeros_like(gq); acca = torch.zeros_like(ga)
    for h in range(2):
        half.sample(idx[128 * h:128 * (h + 1)])
        half.critic_grad(eps[0][128 * h:128 * (h + 1)]); accq += 0.5 * half.q_grads
        half.actor_grad(eps[1][128 * h:128 * (h + 1)]); acca += 0.5 * half.actor_grads
    assert (accq - gq).abs().max().item() <= 1e-5 * gq.abs().max().item()
    assert (acca - ga).abs().max().item() <= 1e-4 * ga.abs().max().item()


@pytest.mark.parametrize("batch", [40, 256, 600])
def test_owed_alpha_step_is_bit_identical_to_a_launch_of_its_own(dev, batch, monkeypatch):
    """The alpha step carried by the next row-group launch (mi_sac_*_update_owed: its log-prob pass on workgroups of the actor / critic update that follows,
    alpha handed to that launch's consumers through the epoch word) against the same training with every alpha step as its own launch: every parameter, the
    optimizer moments, log alpha and its Adam state agree BIT FOR BIT after 30 iterations (policy_frequency 2: both carriers occur), at a batch inside the
    four-workgroup form (40, 256) and one beyond it (600: two-workgroup actor form, single-workgroup critic form)."""
    import deep_rl_amd as D
    import deep_rl_amd.sac_engine as SE

    def run(owe):
        monkeypatch.setattr(SE, "_OWE_ALPHA", owe)
        env = D.make("Pendulum-v1", num_envs=48, device=dev, seed=9)
        torch.manual_seed(9)
        actor = D.Actor(env); q1 = D.SoftQNetwork(env); q2 = D.SoftQNetwork(env); q1t = D.SoftQNetwork(env); q2t = D.SoftQNetwork(env)
        q1t.load_state_dict(q1.state_dict()); q2t.load_state_dict(q2.state_dict())
        eng = D.SACEngine(env, actor, q1, q2, q1t, q2t, slots=40, batch_size=batch, learning_starts=4)
        eng.reset()
        owed_seen = 0
        for _ in range(30):
            eng.act()
            if eng.global_step > 6:
                eng.train_step()
                owed_seen += eng._owed is not None
        return eng, owed_seen

    a, seen_a = run(True)
    b, seen_b = run(False)
    assert seen_a > 5 and seen_b == 0, "the owed path was not exercised"
    assert a.alpha_steps == b.alpha_steps
    for name in ("log_alpha", "alpha", "_alpha_m", "_alpha_v", "alpha_out"):
        assert torch.equal(getattr(a, name), getattr(b, name)), name
    assert torch.equal(a.actor.flat, b.actor.flat) and torch.equal(a.q_flat, b.q_flat) and torch.equal(a.qt_flat, b.qt_flat)
    assert torch.equal(a.actor_optimizer.exp_avg, b.actor_optimizer.exp_avg) and torch.equal(a.q_optimizer.exp_avg_sq, b.q_optimizer.exp_avg_sq)
    assert torch.equal(a.observations, b.observations) and torch.equal(a.actions, b.actions)


@pytest.mark.parametrize("batch", [40, 256, 600])
def test_deferred_critic_step_is_bit_identical_to_a_launch_of_its_own(dev, batch, monkeypatch):
    """The critics' optimizer step (dW2 GEMM + assembly + Adam + polyak, sac.py:183-185,213-217) deferred and carried by the NEXT acting launch (mi_sac_critic_update_deferred
    + mi_sac_act_step_carry) or settled alone when something else comes first (mi_sac_critic_step: the actor update of every second step, a read of the critics) against
    the same training with the step as the second launch of every critic update: every parameter, target, optimizer moment, loss and the whole replay ring agree BIT FOR BIT
    after 31 iterations (policy_frequency 2: both ways of settling occur), below (40, 256) and above (600: padded batch > 512, the step is the split-K pair of launches)
    the fused-step limit."""
    import deep_rl_amd as D
    import deep_rl_amd.sac_engine as SE

    def run(defer):
        monkeypatch.setattr(SE, "_DEFER_CRITIC", defer)
        env = D.make("Pendulum-v1", num_envs=48, device=dev, seed=9)
        torch.manual_seed(9)
        actor = D.Actor(env); q1 = D.SoftQNetwork(env); q2 = D.SoftQNetwork(env); q1t = D.SoftQNetwork(env); q2t = D.SoftQNetwork(env)
        q1t.load_state_dict(q1.state_dict()); q2t.load_state_dict(q2.state_dict())
        eng = D.SACEngine(env, actor, q1, q2, q1t, q2t, slots=40, batch_size=batch, learning_starts=4)
        eng.reset()
        carried = settled = 0
        for _ in range(31):                              # (an odd last step: no actor update behind the last critic update, its step stays owed)
            carried += eng._owed_critic is not None      # a debt at this point rides on the acting launch
            eng.act()
            assert eng._owed_critic is None
            if eng.global_step > 6:
                eng.train_step()
                settled += eng._owed_critic is None      # the actor update of this step settled it
        return eng, carried, settled

    a, carried, settled = run(True)
    b, c0, _ = run(False)
    assert carried > 5 and settled > 5 and c0 == 0, (carried, settled, c0)
    assert a._owed_critic is not None                    # the last critic step is still owed ...
    assert torch.equal(a.q_flat, b.q_flat) and a._owed_critic is None      # ... and reading the critics settles it
    for name in ("qt_flat", "q_grads", "q_losses", "log_alpha", "alpha", "observations", "actions", "rewards"):
        assert torch.equal(getattr(a, name), getattr(b, name)), name
    assert torch.equal(a.actor.flat, b.actor.flat) and torch.equal(a.actor_optimizer.exp_avg, b.actor_optimizer.exp_avg)
    assert torch.equal(a.q_optimizer.exp_avg, b.q_optimizer.exp_avg) and torch.equal(a.q_optimizer.exp_avg_sq, b.q_optimizer.exp_avg_sq)
    assert a.q_optimizer.step_count == b.q_optimizer.step_count and a.update_index == b.update_index


def test_deferred_critic_step_is_settled_before_anything_else_uses_the_workspace(dev):
    """update_critic(); update_alpha() — the log-prob launch writes gradient slabs the deferred step still has to read — and update_critic(); checkpoint: the engine settles
    the debt first, so both orders give the bits of the undeferred engine."""
    import deep_rl_amd.sac_engine as SE

    outs = []
    for defer in (True, False):
        SE._DEFER_CRITIC = defer
        try:
            eng = _engine(dev, 32, 24, seed=4, batch_size=128, learning_starts=2)
            eng.reset()
            for _ in range(8):
                eng.act()
            eng.sample()
            eng.update_critic(polyak=True)
            eng.update_alpha()
            eng.update_critic(polyak=True)
            eng.update_actor()
            eng.update_critic(polyak=False)
            outs.append([t.clone() for t in (eng.q_flat, eng.qt_flat, eng.q_losses, eng.log_alpha, eng.actor.flat, eng.q_optimizer.exp_avg_sq)])
        finally:
            SE._DEFER_CRITIC = True
    for x, y in zip(*outs):
        assert torch.equal(x, y)
