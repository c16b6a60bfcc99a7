"""GPU parity tests of the DQN path (deep_rl_amd/csrc/mi_dqn.hip through the C ABI) against the CPU oracle and the golden
vectors of the unmodified reference dqn.py.  Integer / flag / index work bit-exact; fp32 tolerances written at each assert."""
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
def dqn_trace():
    with np.load(os.path.join(ROOT, "tests", "golden", "dqn_ref_trace.npz")) as z:
        return {k: z[k] for k in z.files}


@pytest.fixture(autouse=True)
def _fdlibm(R):
    R.set_sincos_mode("fdlibm")
    yield
    R.set_sincos_mode("libm")


def _engine(dev, n_envs, slots, params=None, seed=1, base=0, **kw):
    import deep_rl_amd as D

    env = D.make("CartPole-v1", num_envs=n_envs, device=dev, seed=seed, env_id_base=base)
    torch.manual_seed(seed)
    q = D.QNetwork(env); tgt = D.QNetwork(env)
    if params is not None:
        q.load_flat(params)
    tgt.load_state_dict(q.state_dict())
    opt = D.ClipAdam(q, lr=2.5e-4, eps=1e-8)
    kw.setdefault("max_episodes_logged", 4096)
    return D.DQNEngine(env, q, tgt, opt, slots=slots, **kw)


def _upload(eng, st):
    eng.observations.copy_(torch.from_numpy(st.observations)); eng.actions.copy_(torch.from_numpy(st.actions))
    eng.rewards.copy_(torch.from_numpy(st.rewards)); eng.terminated.copy_(torch.from_numpy(st.terminated))


def test_qnetwork_surface_and_forward(dev, R):
    import deep_rl_amd as D

    env = D.make("CartPole-v1", num_envs=1, device=dev)
    torch.manual_seed(1)
    q = D.QNetwork(env)
    assert [tuple(p.shape) for p in q.parameters()] == [(120, 4), (120,), (84, 120), (84,), (2, 84), (2,)]
    t = D.QNetwork(env)
    t.load_state_dict(q.state_dict())
    assert torch.equal(t.flat, q.flat) and t.flat.data_ptr() != q.flat.data_ptr()
    rng = np.random.default_rng(0)
    obs = (rng.normal(0, 1, (5, 300, 4)) * np.array([2.4, 3, 0.2, 3])).astype(np.float32)
    out = q(torch.from_numpy(obs).to(dev))
    assert out.shape == (5, 300, 2)
    ref = R.dqn_forward(q.flat.cpu().numpy(), obs).reshape(5, 300, 2)
    assert np.abs(out.cpu().numpy() - ref).max() < 2e-6 * max(1.0, np.abs(ref).max())   # fp32, 120-term dot products


def test_act_steps_teacher_forced_vs_reference_trace(dev, R, dqn_trace):
    """The reference run's first 3,000 steps (its actions, its reset noise): ring storage bit-exact vs the oracle, terminated /
    episode structure exactly the reference's."""
    g = dqn_trace
    n_steps = 3000
    eng = _engine(dev, 1, slots=100_001, params=g["init_params"])
    env = R.VecCartPole(1); st = R.ReplayStorage(100_001, 1)
    obs_cur = env.reset(g["reset_states"][:1]); st.observations[0, 0] = obs_cur[0]
    eng.reset(torch.from_numpy(g["reset_states"][:1]))
    acts = g["actions_all"].astype(np.int64); ar = g["after_reset_all"]; resets = g["reset_states"]
    ri, gs, n_ep = 1, 0, 0
    while gs < n_steps:
        n = 50
        fr = np.zeros((n, 1, 4))
        for s in range(n):
            if ar[gs + s + 1]:
                fr[s, 0] = resets[ri]; ri += 1
        fa = acts[gs:gs + n].reshape(n, 1)
        R.dqn_act_steps(env, g["init_params"], st, obs_cur, n, gs, forced_actions=fa, forced_resets=fr)
        eng.act(n, forced_actions=torch.from_numpy(fa), forced_resets=torch.from_numpy(fr))
        n_ep += eng.drain_episodes()[0]
        gs += n
    for name in ["actions", "rewards", "terminated"]:
        assert np.array_equal(getattr(eng, name)[:n_steps + 1].cpu().numpy(), getattr(st, name)[:n_steps + 1]), name
    # the oracle runs its device-matched sin/cos here -> bit-exact; the reference (libm) may differ by 1 ulp on a few steps
    assert np.array_equal(eng.observations[:n_steps + 1].cpu().numpy(), st.observations[:n_steps + 1])
    live = ~ar[1:n_steps + 1].astype(bool)
    d = np.abs(eng.observations[1:n_steps + 1, 0].cpu().numpy()[live].astype(np.float64) - g["obs_first"][:n_steps][live])
    assert (d <= np.spacing(np.abs(g["obs_first"][:n_steps][live]))).all() and (d > 0).sum() <= 20
    assert n_ep == int(ar[1:n_steps + 1].sum()) == int((g["episode_global_step"] <= n_steps).sum())
    assert np.array_equal(eng.terminated[1:n_steps + 1, 0].cpu().numpy(), g["terminated_all"][:n_steps])


@pytest.mark.parametrize("n", [96, 37, 3])
def test_act_steps_keyed_rng_ring_wrap(dev, R, n):
    """96 / 37 / 3 envs (the acting kernel owns 16 envs per workgroup: ragged tails), a 16-slot ring (wraps many times), epsilon decaying to
    0.05 within the test: actions follow the RNG contract (random action / greedy argmax decided by the keyed draw), storage bit-exact vs the
    oracle with the device's actions replayed."""
    S, tt, ls = 16, 400, 40
    eng = _engine(dev, n, slots=S, seed=5, base=300, learning_starts=ls, total_timesteps=tt)
    rng = np.random.default_rng(1)
    params = (eng.q.flat.cpu().numpy() + rng.normal(0, 0.05, 10934)).astype(np.float32)
    eng.q.load_flat(params)
    env = R.VecCartPole(n, seed=5, env_id_base=300); st = R.ReplayStorage(S, n)
    obs_cur = env.reset(); st.observations[0] = obs_cur
    assert np.array_equal(eng.reset().cpu().numpy(), obs_cur)
    gs, n_greedy, n_checked = 0, 0, 0
    for call in range(40):
        k = 10
        obs_before = eng.observation.cpu().numpy().copy()
        eng.act(k)
        # recover the device's actions of this call from the ring (slot (gs+s) % S), replay them on the oracle
        fa = np.stack([eng.actions[(gs + s) % S].cpu().numpy() for s in range(k)]) if k <= S else None
        R.dqn_act_steps(env, params, st, obs_cur, k, gs, learning_starts=ls, total_timesteps=tt, forced_actions=fa)
        for name in ["observations", "actions", "rewards", "terminated"]:
            assert np.array_equal(getattr(eng, name).cpu().numpy(), getattr(st, name)), (call, name)
        assert np.array_equal(eng.observation.cpu().numpy(), obs_cur)
        # first step of the call: the decision must follow the contract (obs known = obs_before)
        eps = np.float32(R.dqn_epsilon(gs, total_timesteps=tt))
        qv = R.dqn_forward(params, obs_before)
        for e in range(0, n, 7 if n > 7 else 1):
            u, ra = R.dqn_explore_draw(5, 300 + e, gs)
            if gs < ls or u < eps:
                assert fa[0, e] == ra
            elif abs(qv[e, 0] - qv[e, 1]) > 1e-4:
                assert fa[0, e] == int(qv[e, 1] > qv[e, 0]); n_greedy += 1
            n_checked += 1
        gs += k
    assert n_greedy > (50 if n > 7 else 10) and n_checked > (400 if n == 96 else 100)


def test_sample_bit_exact(dev, R):
    from deep_rl_amd import _native as N

    for upper, batch in [(1, 128), (12345, 128), (1_000_000 * 64, 4096)]:
        idx = torch.empty(batch, dtype=torch.int64, device=dev)
        N.check(N.lib().mi_dqn_sample(9, 77, upper, batch, N.ptr(idx), N.stream_ptr(dev)))
        assert np.array_equal(idx.cpu().numpy(), R.dqn_sample(9, 77, upper, batch))
    assert idx.min() >= 0 and idx.max() < 1_000_000 * 64


def _replay_reference(R, g, upto_steps):
    R.set_sincos_mode("libm")
    env = R.VecCartPole(1); st = R.ReplayStorage(100_001, 1)
    obs_cur = env.reset(g["reset_states"][:1]); st.observations[0, 0] = obs_cur[0]
    acts = g["actions_all"].astype(np.int64); ar = g["after_reset_all"]; resets = g["reset_states"]
    ri, gs = 1, 0
    while gs < upto_steps:
        n = 10
        fr = np.zeros((n, 1, 4))
        for s in range(n):
            if (gs + s + 1 < 100_000 and ar[gs + s + 1]) or (gs + s + 1 == 100_000 and ri < len(resets)):
                fr[s, 0] = resets[ri]; ri += 1
        R.dqn_act_steps(env, g["init_params"], st, obs_cur, n, gs, forced_actions=acts[gs:gs + n].reshape(n, 1), forced_resets=fr)
        gs += n
    R.set_sincos_mode("fdlibm")
    return st


def test_td_grad_vs_reference_checkpoints(dev, R, dqn_trace):
    """Un-chained: the reference's online / target parameters, batch indices and (replayed) storage at updates 1000, 5000, 9000
    -> loss and gradient of the HIP kernel against the reference's autograd and the oracle."""
    g = dqn_trace
    st = _replay_reference(R, g, 100_000)
    eng = _engine(dev, 1, slots=100_001)
    _upload(eng, st)
    for i, k in enumerate(g["ck_update"]):
        eng.q.load_flat(g["ck_params"][i]); eng.target.load_flat(g["ck_target"][i])
        eng.sample(g["ck_inds"][i])
        eng.td_grad()
        grads = eng.grads.cpu().numpy(); loss = float(eng.loss.item())
        og, ol = R.dqn_td_grads(g["ck_params"][i], g["ck_target"][i], st, g["ck_inds"][i])
        scale = np.abs(g["ck_grads"][i]).max()
        assert np.abs(grads - g["ck_grads"][i]).max() <= 5e-6 * scale, (k, np.abs(grads - g["ck_grads"][i]).max() / scale)
        assert np.abs(grads - og).max() <= 5e-6 * scale
        assert abs(loss - g["ck_loss"][i]) <= 1e-5 * g["ck_loss"][i] and abs(loss - ol) <= 1e-5 * ol
        eng.td_grad()
        assert np.array_equal(eng.grads.cpu().numpy(), grads)   # bitwise reproducible


def test_first_300_updates_chained_on_device(dev, R, dqn_trace):
    """The reference's first 300 TD updates chained through the device's own Adam (reference indices, replayed storage, target
    sync every 500 steps): every loss within 2e-5 relative, parameters within 2e-6 of the reference's after the first 8 steps."""
    from tests.test_oracle_dqn_pinned import regenerate_batch_inds

    g = dqn_trace
    inds, _ = regenerate_batch_inds(g)
    st = _replay_reference(R, g, 13_100)
    eng = _engine(dev, 1, slots=100_001, params=g["init_params"])
    _upload(eng, st)
    losses = []
    for k in range(300):
        # storage beyond the current global_step is not sampled: indices < gs by construction
        eng.train_step(inds[k])
        losses.append(eng.loss.clone())
        if k < 8:
            assert np.abs(eng.q.flat.cpu().numpy() - g["full_params"][k]).max() < 2e-6, k
        if (10_000 + 10 * k) % 500 == 0:
            eng.sync_target()
    losses = torch.cat(losses).cpu().numpy()
    rel = np.abs(losses - g["loss_all"][:300]) / np.maximum(np.abs(g["loss_all"][:300]), 1e-3)
    assert rel.max() < 2e-5, rel.max()
    assert abs(eng.q.flat.double().sum().item() - g["psum_all"][299]) < 1e-4


@pytest.mark.parametrize("batch", [128, 1000, 2048, 2056, 4096, 5000])
def test_td_grad_batches_and_ring(dev, R, batch):
    """Bigger / ragged batches on a wrapped 64-slot ring of 32 envs (successor index crosses the ring end).  Up to 2,048 rows dqn_td_kernel runs one 8-row group per
    workgroup; 2,056 is the first batch on the 16-row form (129 workgroups, the last group half empty); 5,000 rows = 313 groups dealt to 256 workgroups (57 of them walk
    two groups and accumulate both in registers, the last group ragged)."""
    n, S = 32, 64
    eng = _engine(dev, n, slots=S, seed=3, batch_size=batch, learning_starts=0, total_timesteps=1000)
    rng = np.random.default_rng(4)
    params = (eng.q.flat.cpu().numpy() + rng.normal(0, 0.05, 10934)).astype(np.float32)
    tparams = (params + rng.normal(0, 0.05, 10934)).astype(np.float32)
    eng.q.load_flat(params); eng.target.load_flat(tparams)
    eng.reset()
    for _ in range(10):
        eng.act(20)   # 200 steps: the ring has wrapped three times
    st = R.ReplayStorage(S, n)
    for name in ["observations", "actions", "rewards", "terminated"]:
        getattr(st, name)[...] = getattr(eng, name).cpu().numpy()
    eng.sample()
    idx = eng.batch_inds.cpu().numpy()
    assert np.array_equal(idx, R.dqn_sample(3, 0, S * n, batch)) and (idx // n == S - 1).any()   # some successors wrap to slot 0
    eng.td_grad()
    og, ol = R.dqn_td_grads(params, tparams, st, idx)
    assert np.abs(eng.grads.cpu().numpy() - og).max() <= 1e-5 * np.abs(og).max()
    assert abs(float(eng.loss.item()) - ol) <= 2e-5 * ol


def test_config3_full_size_properties(dev, R):
    """BASELINE config 3 at full size (4096 envs x 256 slots = 1,048,576 transitions; the ring wraps): size-independent properties.
    (a) shard invariance: two 2048-env engines reproduce the 4096-env ring bit for bit (keyed RNG, MFMA acting kernel);
    (b) physics: re-stepping a stored observation with the stored action lands on the stored successor (float32 storage of a float64
        state: 2e-5), rewards are 1, `terminated` agrees with the thresholds of the successor the env computed;
    (c) the TD gradient is additive over the batch (two halves at half weight sum to the whole) and bitwise reproducible."""
    n, S, steps = 4096, 256, 300
    rng = np.random.default_rng(11)
    params = None
    rings = []
    for base, cnt in ((0, n), (0, n // 2), (n // 2, n // 2)):
        eng = _engine(dev, cnt, slots=S, seed=2, base=base, learning_starts=100, total_timesteps=600, max_episodes_logged=0)
        if params is None:
            params = (eng.q.flat.cpu().numpy() + rng.normal(0, 0.05, 10934)).astype(np.float32)
        eng.q.load_flat(params); eng.target.load_flat(params)
        eng.reset()
        for _ in range(steps // 10):
            eng.act(10)
        rings.append(eng)
    big = rings[0]
    for r, sl in ((rings[1], slice(0, n // 2)), (rings[2], slice(n // 2, n))):
        for name in ("observations", "actions", "terminated"):
            assert torch.equal(getattr(big, name)[:, sl], getattr(r, name)), name
    # (b) on 3,000 random (slot, env) pairs whose successor slot is valid (not the write head) and which did not end an episode
    obs = big.observations.cpu().numpy(); act = big.actions.cpu().numpy(); term = big.terminated.cpu().numpy()
    head = steps % S
    checked = 0
    for _ in range(3000):
        s, e = int(rng.integers(0, S)), int(rng.integers(0, n))
        if (s + 1) % S == head or s == head:
            continue
        nxt, _t = R.cartpole_step(obs[s, e].astype(np.float64), int(act[s, e]))
        ended = abs(nxt[0]) > 2.4 or abs(nxt[2]) > 12 * 2 * np.pi / 360
        if ended:      # the stored successor is the reset state; the flag must say terminated (unless float32 rounding decided the threshold)
            if min(abs(abs(nxt[0]) - 2.4), abs(abs(nxt[2]) - 12 * 2 * np.pi / 360)) > 1e-4:
                assert term[(s + 1) % S, e] == 1
            continue
        if term[(s + 1) % S, e] == 0 and np.abs(obs[(s + 1) % S, e]).max() > 0.06:      # not a reset by truncation (reset states are within +-0.05)
            assert np.abs(nxt - obs[(s + 1) % S, e]).max() < 2e-5, (s, e)
            checked += 1
    assert checked > 1500 and (big.rewards.cpu().numpy()[np.arange(S) != head] == 1.0).all()
    # (c)
    B = 16384
    eng = _engine(dev, n, slots=S, seed=2, batch_size=B, learning_starts=100, total_timesteps=600, max_episodes_logged=0)
    eng.q.load_flat(params); eng.target.load_flat((params + rng.normal(0, 0.05, 10934)).astype(np.float32))
    for name in ("observations", "actions", "rewards", "terminated"):
        getattr(eng, name).copy_(getattr(big, name))
    eng.global_step = steps
    eng.sample()
    idx = eng.batch_inds.cpu().numpy()
    assert idx.min() >= 0 and idx.max() < S * n
    eng.td_grad()
    g_all = eng.grads.clone(); l_all = float(eng.loss)
    eng.td_grad()
    assert torch.equal(eng.grads, g_all)
    half = _engine(dev, n, slots=S, seed=2, batch_size=B // 2, learning_starts=100, total_timesteps=600, max_episodes_logged=0)
    half.q.load_flat(params); half.target.flat.copy_(eng.target.flat)
    for name in ("observations", "actions", "rewards", "terminated"):
        getattr(half, name).copy_(getattr(big, name))
    acc = torch.zeros_like(g_all); lsum = 0.0
    for h in range(2):
        half.sample(idx[h * B // 2:(h + 1) * B // 2]); half.td_grad()
        acc += 0.5 * half.grads; lsum += 0.5 * float(half.loss)
    assert (acc - g_all).abs().max().item() <= 5e-6 * g_all.abs().max().item() and abs(lsum - l_all) <= 1e-5 * l_all


def test_fused_td_update_equals_grad_then_adam(dev, R):
    """mi_dqn_td_update (TD gradient + Adam in the reduction launch) == td_grad() + optimizer.step(), bit for bit, for plain and prioritized DQN."""
    import deep_rl_amd as D

    rng = np.random.default_rng(5)
    for kind in ("dqn", "per"):
        engs = []
        for fused in (True, False):
            env = D.make("CartPole-v1", num_envs=32, device=dev, seed=3)
            torch.manual_seed(3)
            q = D.QNetwork(env); t = D.QNetwork(env); t.load_state_dict(q.state_dict())
            Eng = D.DQNEngine if kind == "dqn" else D.PERDQNEngine
            eng = Eng(env, q, t, D.ClipAdam(q, lr=2.5e-4, eps=1e-8), slots=64, batch_size=200, learning_starts=0, total_timesteps=1000, max_episodes_logged=0)
            eng.reset()
            for _ in range(5):
                eng.act(10)
            for it in range(3):
                if fused:
                    eng.train_step()
                else:
                    eng.sample(); eng.td_grad(); eng.optimizer.step(eng.grads); eng.update_index += 1
            engs.append(eng)
        f, u = engs
        assert f.optimizer.step_count == u.optimizer.step_count == 3
        for name in ("grads", "loss", "batch_inds"):
            assert torch.equal(getattr(f, name), getattr(u, name)), (kind, name)
        assert torch.equal(f.q.flat, u.q.flat) and torch.equal(f.optimizer.exp_avg_sq, u.optimizer.exp_avg_sq)
        if kind == "per":
            assert torch.equal(f.priorities, u.priorities) and torch.equal(f.max_priority, u.max_priority)
