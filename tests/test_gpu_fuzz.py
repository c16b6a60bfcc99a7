"""Randomised shapes through the hot path, checked against the CPU oracle (oracle/cpu_ref.c) — what the fixed parametrisations of test_gpu_{parity,dqn,sac,per}.py
do at hand-picked sizes, at sizes nobody picked: env counts that are no multiple of a workgroup's share, rollouts shorter and LONGER than the 128 steps the rollout
launch keeps in LDS (the ring then wraps and the actor wave waits for the critic wave: T up to 300), minibatches / batches with ragged tails, rings that wrap at odd
slot counts, priorities with holes.  Bars as in those files: bit-exact for env state, flags, indices and the GAE given its inputs; the fp32 tolerances written at each
assert for everything behind a network.

MIRL_FUZZ_CASES (default 4 per family: the suite stays short) and MIRL_FUZZ_SEED (default 1) choose the cases; a failing assert names the case's shape, so it can be
replayed.  Round 6 ran 2 x 1,500 cases per family once (profiles/r06_fuzz.txt: 62,400 cases, one real finding — a one-row minibatch — fixed)."""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
CASES = int(os.environ.get("MIRL_FUZZ_CASES", "4"))
SEED = int(os.environ.get("MIRL_FUZZ_SEED", "1"))


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    return torch.device("cuda", 0)


@pytest.fixture(scope="module")
def R():
    from oracle import cpu_ref

    cpu_ref.lib().ref_set_num_threads(max(1, min(16, os.cpu_count() or 1)))
    return cpu_ref


@pytest.fixture(autouse=True)
def _fdlibm_mode(R):
    R.set_sincos_mode("fdlibm")  # the device-matched sin/cos mode
    yield
    R.set_sincos_mode("libm")


# A ReLU's DERIVATIVE jumps at 0: a pre-activation within rounding of 0 may land on either side on the device and in the oracle, and the gradient of that row then
# differs by a whole term (seen once in ~500 random batches of > 1,000 rows; the fixed-shape tests never met one).  The gradient comparisons below therefore draw
# their rows from those whose hidden pre-activations (computed here in float64) all stay clear of 0 — the tolerances then hold without exception.
NEAR_ZERO = 1e-5


def _mlp_clear_of_zero(x, W1, b1, W2, b2):
    """rows of x whose two hidden layers' pre-activations are all farther than NEAR_ZERO from 0; also returns the second hidden layer"""
    z1 = x.astype(np.float64) @ W1.astype(np.float64).T + b1
    h1 = np.maximum(z1, 0.0)
    z2 = h1 @ W2.astype(np.float64).T + b2
    return (np.abs(z1).min(1) > NEAR_ZERO) & (np.abs(z2).min(1) > NEAR_ZERO), np.maximum(z2, 0.0)


def _dqn_rows_clear(params, obs):
    W1, b1 = params[:480].reshape(120, 4), params[480:600]
    W2, b2 = params[600:10680].reshape(84, 120), params[10680:10764]
    return _mlp_clear_of_zero(obs, W1, b1, W2, b2)[0]


def _sacq(q_p, c, R):
    p = q_p[c * R.SQ_NPARAMS:(c + 1) * R.SQ_NPARAMS]
    return p[:1024].reshape(256, 4), p[1024:1280], p[1280:1280 + 65536].reshape(256, 256), p[66816:67072], p[67072:67328], p[67328]


def _sac_q_rows_clear(q_p, x, R):
    """rows of x = [obs, action] clear of 0 in BOTH critics; and the two Q-values"""
    ok, qs = np.ones(len(x), bool), []
    for c in range(2):
        W1, b1, W2, b2, w3, b3 = _sacq(q_p, c, R)
        o, h2 = _mlp_clear_of_zero(x, W1, b1, W2, b2)
        ok &= o
        qs.append(h2 @ w3.astype(np.float64) + b3)
    return ok, qs[0], qs[1]


def _sac_actor_rows_clear(a_p, obs):
    W1, b1 = a_p[:768].reshape(256, 3), a_p[768:1024]
    W2, b2 = a_p[1024:1024 + 65536].reshape(256, 256), a_p[66560:66816]
    return _mlp_clear_of_zero(obs, W1, b1, W2, b2)[0]


def _ppo_rows_clear(params, obs, actions, old_logp, old_values, returns, clip=0.2):
    """rows of the storage away from the kinks of ppo.py:172-186: the ratio's clip boundaries 1 +- clip (the derivative of max(-A r, -A clamp(r)) jumps there), the value
    clip's boundary |v - v_old| = clip and the tie of the two value losses — a row within rounding of one lands on either side on the device and in the oracle"""
    p = params.astype(np.float64)
    def mlp(o, out):
        W1, b1 = p[o:o + 256].reshape(64, 4), p[o + 256:o + 320]
        W2, b2 = p[o + 320:o + 4416].reshape(64, 64), p[o + 4416:o + 4480]
        W3, b3 = p[o + 4480:o + 4480 + 64 * out].reshape(out, 64), p[o + 4480 + 64 * out:o + 4480 + 65 * out]
        h = np.tanh(np.tanh(obs.astype(np.float64) @ W1.T + b1) @ W2.T + b2)
        return h @ W3.T + b3
    logits = mlp(0, 2)                                  # actor first (4,610 parameters), then the critic
    v = mlp(4610, 1)[:, 0]
    lse = np.log(np.exp(logits - logits.max(1, keepdims=True)).sum(1)) + logits.max(1)
    logp = logits[np.arange(len(obs)), actions] - lse
    ratio = np.exp(logp - old_logp)
    ok = (np.abs(ratio - (1 - clip)) > 1e-4) & (np.abs(ratio - (1 + clip)) > 1e-4)
    dv = v - old_values
    vc = old_values + np.clip(dv, -clip, clip)
    ok &= np.abs(np.abs(dv) - clip) > 1e-5
    ok &= (np.abs(dv) <= clip) | (np.abs((v - returns) ** 2 - (vc - returns) ** 2) > 1e-6)
    return ok


def _log(rec):
    path = os.path.join(ROOT, "gpurun_out", "fuzz_cases.txt")
    os.makedirs(os.path.dirname(path), exist_ok=True)
    with open(path, "a") as f:
        f.write(rec + "\n")


@pytest.mark.parametrize("case", range(CASES))
def test_ppo_rollout_gae_grad_any_shape(dev, R, case):
    """ppo.py:110-151 + :160-189 at a random (envs, steps, minibatch): two rollouts (the second from carried-over env state, counters and RNG indices) with the device's
    own actions replayed on the oracle, GAE bit-exact given the device's values, then one minibatch gradient of a random size on parameters that are not the
    behaviour parameters."""
    import test_gpu_parity as P

    rng = np.random.default_rng([SEED, 1, case])
    n = int(rng.choice([rng.integers(1, 9), rng.integers(9, 70), rng.integers(70, 260)]))
    T = int(rng.choice([rng.integers(1, 20), rng.integers(20, 129), rng.integers(129, 301)]))
    seed, base = int(rng.integers(1, 1000)), int(rng.integers(0, 5000))
    mb = int(rng.integers(1, min(T * n, 6000) + 1))
    shape = "ppo case %d: envs %d, T %d, seed %d, env_id_base %d, mb %d" % (case, n, T, seed, base, mb)
    eng = P._engine(dev, n, seed=seed, env_id_base=base, T_=T, max_episodes_logged=T * n + 16, n_minibatch=1)   # (any T x envs: no divisibility by 4 asked of the shape)
    params = (eng.agent.flat.cpu().numpy() + rng.normal(0, 0.15, 9155) * (np.arange(9155) >= 4480) * (np.arange(9155) < 4610)).astype(np.float32)
    eng.agent.load_flat(params)
    env = R.VecCartPole(n, seed=seed, env_id_base=base)
    obs_cur = env.reset()
    assert np.array_equal(eng.reset().cpu().numpy(), obs_cur), shape
    st = R.Storage(T, n)
    for upd in range(2):
        eng.rollout_gae()
        n_ep, eps = eng.drain_episodes()
        acts = eng.actions[:T].cpu().numpy()
        reps, rn = R.rollout(env, params, st, obs_cur, forced_actions=acts, max_ep=T * n + 16)
        assert np.array_equal(eng.observations.cpu().numpy(), st.observations), (shape, upd)
        assert np.array_equal(eng.dones.cpu().numpy(), st.dones) and np.array_equal(eng.rewards.cpu().numpy(), st.rewards), (shape, upd)
        assert np.abs(eng.values.cpu().numpy() - st.values).max() < 3e-6, (shape, upd)
        assert np.abs(eng.log_probs[:T].cpu().numpy() - st.log_probs[:T]).max() < 3e-6, (shape, upd)
        assert n_ep == rn and sorted(eps) == sorted(reps), (shape, upd, n_ep, rn)
        assert np.array_equal(eng.observation.cpu().numpy(), obs_cur), (shape, upd)
        st.values[...] = eng.values.cpu().numpy()
        R.gae(st)
        assert np.array_equal(eng.advantages.cpu().numpy(), st.advantages) and np.array_equal(eng.returns.cpu().numpy(), st.returns), (shape, upd)
    p2 = (params + rng.normal(0, 0.05, 9155)).astype(np.float32)
    eng.agent.load_flat(p2)
    eng.make_perm(0)
    assert np.array_equal(np.sort(eng.perm.cpu().numpy()), np.arange(T * n)), shape
    so = P._storage_to_oracle(R, eng)
    clear = np.flatnonzero(_ppo_rows_clear(p2, so.observations[:T].reshape(T * n, 4), so.actions[:T].reshape(T * n).astype(np.int64), so.log_probs[:T].reshape(T * n).astype(np.float64),
                                           so.values[:T].reshape(T * n).astype(np.float64), so.returns[:T].reshape(T * n).astype(np.float64)))
    if clear.size >= mb:      # the minibatch: the keyed permutation's first mb rows that are clear of the loss's kinks (the permutation itself was checked above)
        perm = eng.perm.cpu().numpy()
        keep = perm[np.isin(perm, clear)][:mb]
        eng.perm[:mb] = torch.from_numpy(keep).to(eng.perm.device, eng.perm.dtype)
    eng.adv_stats(mb=mb, n_mb=1)
    eng.minibatch_grad(0, mb=mb)
    idx = eng.perm[:mb].cpu().numpy()
    og, ot = R.minibatch(p2, so, idx)
    grads = eng.grads.cpu().numpy(); terms = eng.loss_terms.cpu().numpy()
    scale = max(np.abs(og).max(), 1e-12)
    # (a one-row minibatch has advantage std 0 / nan in the reference's normalisation: the comparison is of finite cases only, like ppo.py can run them)
    if np.isfinite(og).all():
        assert np.abs(grads - og).max() <= 3e-5 * scale, (shape, np.abs(grads - og).max() / scale)
        assert np.allclose(terms, ot, rtol=1e-4, atol=2e-5), (shape, terms, ot)
    else:
        assert not np.isfinite(grads).all(), shape
    eng.minibatch_grad(0, mb=mb)
    assert np.array_equal(eng.grads.cpu().numpy(), grads, equal_nan=True), shape
    _log(shape + " ok")


@pytest.mark.parametrize("case", range(CASES))
def test_dqn_act_and_td_any_shape(dev, R, case):
    """dqn.py:86-108 + :113-133 at a random (envs, slots, steps per launch, batch): the ring after every acting launch bit-exact against the oracle with the device's
    actions replayed, the keyed batch indices bit-exact, the TD gradient within 1e-5 of the oracle's largest element."""
    import test_gpu_dqn as Q

    rng = np.random.default_rng([SEED, 2, case])
    n = int(rng.choice([rng.integers(1, 17), rng.integers(17, 100), rng.integers(100, 400)]))
    S = int(rng.integers(2, 41))
    k = int(rng.integers(1, min(S, 12) + 1))
    calls = int(rng.integers(max(2, (S + k - 1) // k + 1), max(3, 3 * S // k + 3)))
    ls, tt = int(rng.integers(0, 40)), int(rng.integers(50, 600))
    batch = int(rng.choice([rng.integers(1, 130), rng.integers(130, 2049), rng.integers(2049, 4500)]))
    seed, base = int(rng.integers(1, 1000)), int(rng.integers(0, 5000))
    shape = "dqn case %d: envs %d, slots %d, %d calls x %d steps, learning_starts %d, total %d, batch %d, seed %d, base %d, episode log %d" % (
        case, n, S, calls, k, ls, tt, batch, seed, base, case & 1)
    eplog = bool(case & 1)               # every second case keeps the per-episode log (another instantiation of the acting kernel)
    max_ep = (calls * k * n) // 8 + n + 16
    eng = Q._engine(dev, n, slots=S, seed=seed, base=base, batch_size=batch, learning_starts=ls, total_timesteps=tt, max_episodes_logged=max_ep if eplog else 0)
    params = (eng.q.flat.cpu().numpy() + rng.normal(0, 0.05, 10934)).astype(np.float32)
    tparams = (params + rng.normal(0, 0.05, 10934)).astype(np.float32)
    eng.q.load_flat(params); eng.target.load_flat(tparams)
    env = R.VecCartPole(n, seed=seed, env_id_base=base); st = R.ReplayStorage(S, n)
    obs_cur = env.reset(); st.observations[0] = obs_cur
    assert np.array_equal(eng.reset().cpu().numpy(), obs_cur), shape
    gs = 0
    for call in range(calls):
        eng.act(k)
        fa = np.stack([eng.actions[(gs + s) % S].cpu().numpy() for s in range(k)])
        if eplog:
            n_ep, eps_ = eng.drain_episodes()
            reps, rn = R.dqn_act_steps_log(env, params, st, obs_cur, k, gs, learning_starts=ls, total_timesteps=tt, forced_actions=fa, max_ep=max_ep)
            assert n_ep == rn and sorted(eps_) == sorted(reps), (shape, call, n_ep, rn)
        else:
            R.dqn_act_steps(env, params, st, obs_cur, k, gs, learning_starts=ls, total_timesteps=tt, forced_actions=fa)
        for name in ["observations", "actions", "rewards", "terminated"]:
            assert np.array_equal(getattr(eng, name).cpu().numpy(), getattr(st, name)), (shape, call, name)
        assert np.array_equal(eng.observation.cpu().numpy(), obs_cur), (shape, call)
        gs += k
    assert gs >= S, shape      # the ring is full: the sampler's range is the whole ring
    eng.sample()
    idx = eng.batch_inds.cpu().numpy()
    assert np.array_equal(idx, R.dqn_sample(seed, 0, S * n, batch)), shape
    clear = np.flatnonzero(_dqn_rows_clear(params, st.observations.reshape(S * n, 4)))
    assert clear.size > 0.5 * S * n, shape
    idx = rng.choice(clear, batch)
    last = clear[clear // n == S - 1]
    if last.size:
        idx[0] = last[0]               # a successor that wraps to slot 0
    eng.sample(idx)
    eng.td_grad()
    og, ol = R.dqn_td_grads(params, tparams, st, idx)
    g = eng.grads.cpu().numpy()
    assert np.abs(g - og).max() <= 1e-5 * np.abs(og).max(), (shape, np.abs(g - og).max() / np.abs(og).max())
    assert abs(float(eng.loss.item()) - ol) <= 2e-5 * max(ol, 1e-6), (shape, float(eng.loss.item()), ol)
    eng.td_grad()
    assert np.array_equal(eng.grads.cpu().numpy(), g), shape
    _log(shape + " ok")


@pytest.mark.parametrize("case", range(CASES))
def test_sac_grads_any_shape(dev, R, case):
    """sac.py:165-211 at a random (envs, slots, batch, weight scale) on a random ring: critic, actor and alpha gradients against the oracle."""
    import test_gpu_sac as S_

    rng = np.random.default_rng([SEED, 3, case])
    n, slots = int(rng.integers(1, 40)), int(rng.integers(2, 80))
    batch = int(rng.choice([rng.integers(1, 17), rng.integers(17, 600), rng.integers(600, 2300)]))
    scale = float(rng.choice([1.0, 1.5, 2.0]))
    alpha = float(rng.uniform(0.05, 1.0))
    shape = "sac case %d: envs %d, slots %d, batch %d, scale %.1f, alpha %.3f" % (case, n, slots, batch, scale, alpha)
    a_p, q_p = S_._rand_nets(R, rng, scale)
    _, qt_p = S_._rand_nets(R, rng, scale)
    st = S_._random_storage(R, rng, slots, n)
    eng = S_._engine(dev, n, slots, actor=a_p, q=q_p, qt=qt_p, batch_size=batch)
    S_._upload(eng, st)
    flat_obs = st.observations.reshape(slots * n, 3)
    x_all = np.concatenate([flat_obs, st.actions.reshape(slots * n, 1)], 1)
    clear = np.flatnonzero(_sac_q_rows_clear(q_p, x_all, R)[0] & _sac_actor_rows_clear(a_p, flat_obs))
    assert clear.size > 0.3 * slots * n, (shape, clear.size)
    idx = rng.choice(clear, batch)
    last = clear[clear // n == slots - 1]
    if last.size:
        idx[0] = last[0]                      # the last slot: "next" wraps around the ring
    eps = rng.standard_normal((2, batch)).astype(np.float32)
    for _ in range(50):                       # the actor update evaluates the critics at the actor's OWN action: redraw the noise of rows that land near a ReLU's 0, a min(q1, q2) tie or a saturated tanh
        a_new, _ = R.sac_actor_sample(a_p, flat_obs[idx], eps[1])
        ok, q1, q2 = _sac_q_rows_clear(q_p, np.concatenate([flat_obs[idx], np.asarray(a_new, np.float32).reshape(-1, 1)], 1), R)
        y = np.asarray(a_new, np.float64).reshape(-1) / 2.0
        # (sac.py:75 takes log(action_scale (1 - tanh(u)^2) + 1e-6) in fp32: where tanh has saturated, 1 - y^2 is a difference of last bits and its derivative changes by
        # ~10 % with one ulp of y — ill-conditioned in the reference itself, so such rows say nothing about the kernels)
        bad = np.flatnonzero(~ok | (np.abs(q1 - q2) < 1e-4 * np.maximum(1.0, np.abs(q1))) | (1.0 - y * y < 1e-3))
        if bad.size == 0:
            break
        eps[1, bad] = rng.standard_normal(bad.size).astype(np.float32)
        idx[bad] = rng.choice(clear, bad.size)   # (and the row: a saturated tanh gives the same action whatever the noise)
    else:
        raise AssertionError((shape, "no clear noise found"))
    eng.alpha.fill_(alpha)
    a32 = float(np.float32(alpha))
    eng.sample(idx); eng.critic_grad(torch.from_numpy(eps[0]))
    g_ref, l_ref = R.sac_critic_grads(q_p, qt_p, a_p, st, idx, eps[0], a32)
    g = eng.q_grads.cpu().numpy()
    assert np.allclose(eng.q_losses.cpu().numpy(), l_ref, rtol=1e-4), (shape, eng.q_losses.cpu().numpy(), l_ref)   # (scale 2 nets: |q| of tens, a mean of their squared differences)
    for c in range(2):
        sl = slice(c * R.SQ_NPARAMS, (c + 1) * R.SQ_NPARAMS)
        err = S_._rel(g[sl], g_ref[sl])
        assert err < 3e-5, (shape, "critic", c, err)
    eng.critic_grad(torch.from_numpy(eps[0]))
    assert np.array_equal(eng.q_grads.cpu().numpy(), g), shape
    eng.actor_grad(torch.from_numpy(eps[1]))
    ga_ref, loss_ref, mlp_ref = R.sac_actor_grads(a_p, q_p, st, idx, eps[1], a32)
    out = eng.actor_out.cpu().numpy()
    assert abs(out[0] - loss_ref) <= 1e-4 * max(1.0, abs(loss_ref)) and abs(out[1] - mlp_ref) <= 1e-4 * max(1.0, abs(mlp_ref)), (shape, out, loss_ref, mlp_ref)   # (scale 2: saturated tanh, log(1 - a^2 + 1e-6) of rows at the clamp)
    err = S_._rel(eng.actor_grads.cpu().numpy(), ga_ref)
    assert err < 1e-4, (shape, "actor", err)
    _log(shape + " ok")


@pytest.mark.parametrize("case", range(CASES))
def test_per_sampler_any_shape(dev, R, case):
    """per.py:126-128,145-147 at a random (envs, slots, fill, batch, alpha): chunk sums, sampled indices (bit-exact) and importance weights against the oracle on
    priorities with holes."""
    import test_gpu_per as E

    rng = np.random.default_rng([SEED, 4, case])
    n = int(rng.choice([rng.integers(1, 17), rng.integers(17, 300), rng.integers(300, 1500)]))
    slots = int(rng.integers(2, max(3, min(3000, 600000 // n))))
    stored = int(rng.integers(1, slots + 1))
    batch = int(rng.integers(1, 2500))
    alpha = float(rng.choice([0.0, 0.3, 0.6, 1.0]))
    holes = float(rng.uniform(0.0, 0.6))
    seed, upd = int(rng.integers(1, 1000)), int(rng.integers(0, 100000))
    shape = "per case %d: envs %d, slots %d, stored %d, batch %d, alpha %.1f, holes %.2f, seed %d, update %d" % (case, n, slots, stored, batch, alpha, holes, seed, upd)
    eng = E._engine(dev, n, slots, seed=seed, batch_size=batch, total_timesteps=10 * slots, alpha=alpha)
    cap = slots * n
    prio = rng.gamma(0.5, 1.0, cap).astype(np.float32)
    prio[rng.random(cap) < holes] = 0.0
    prio[stored * n:] = 0.0
    if not (prio[:stored * n] > 0).any():
        prio[0] = 1.0
    eng.priorities.copy_(torch.from_numpy(prio.reshape(slots, n)))
    eng.refresh_sums()
    eng.global_step = stored
    eng.update_index = upd
    eng.sample()
    m = stored * n
    s0, s1, total, total_alpha = R.per_sums(prio, m, alpha)
    idx = R.per_sample(seed, upd, prio, m, s0, s1, total, batch)
    got = eng.batch_inds.cpu().numpy()
    assert np.array_equal(got, idx) and (prio[got] > 0).all(), (shape, int((got != idx).sum()))
    w = R.per_weights(prio, idx, alpha, np.float32(eng.beta()), total_alpha, m)
    assert np.allclose(eng.weights.cpu().numpy(), w, rtol=3e-5) and abs(float(eng.weights.max()) - 1.0) < 1e-6, shape
    _log(shape + " ok")


FORMS = ("ppo", "per", "dueling", "sac_owed_alpha", "sac_deferred_critic", "sac_shadows", "per_incremental", "ppo_synthetic_world", "shards")


@pytest.mark.parametrize("case", range(max(CASES, len(FORMS))))
def test_one_call_forms_equal_their_launch_sequences_any_shape(dev, R, case, monkeypatch):
    """The fused / owed / deferred / riding forms of the update paths against the explicit launch sequences they replace, BIT FOR BIT, at random shapes (the fixed-shape
    tests these bodies come from: test_gpu_fullsize.py:136, test_gpu_per.py:289, test_gpu_dueling.py:155, test_gpu_sac.py:564,600, test_gpu_sac_shadow.py:62).  No oracle
    and no tolerance here: the two forms run the same arithmetic in the same order by construction, so any shape at which they differ is a bug of one of them."""
    rng = np.random.default_rng([SEED, 5, case])
    form = FORMS[case % len(FORMS)]
    if form == "ppo":
        import deep_rl_amd.engine as E
        import test_gpu_parity as P

        n = int(rng.choice([rng.integers(1, 9), rng.integers(9, 70), rng.integers(70, 400)]))
        T = int(rng.choice([rng.integers(1, 20), rng.integers(20, 129), rng.integers(129, 260)]))
        T += (-T * n) % 4 if n % 4 else 0          # ppo.py:95-96: the batch splits into 4 minibatches
        while (T * n) % 4:
            T += 1
        seed = int(rng.integers(1, 1000))
        shape = "forms case %d: ppo update, envs %d, T %d, seed %d" % (case, n, T, seed)
        outs = []
        for forced in (False, True):
            monkeypatch.setattr(E, "_FORCE_SHARDED_SEQUENCE", forced)
            eng = P._engine(dev, n, seed=seed, T_=T, max_episodes_logged=0)
            eng.reset()
            for u in range(2):
                eng.optimizer.param_groups[0]["lr"] = (1.0 - u / 4) * 2.5e-4
                eng.update()
            o = eng.optimizer
            outs.append([t.clone() for t in (eng.agent.flat, o.exp_avg, o.exp_avg_sq, o.grad_norm, eng.loss_terms, eng.grads, eng.observations, eng.advantages)])
            assert o.step_count == 32, shape
        for k, (a, b) in enumerate(zip(*outs)):      # (a one-row minibatch — 4 rows in all — is NaN in both forms, as in ppo.py: NaN compares equal to NaN here)
            assert torch.equal(torch.nan_to_num(a, nan=12345.0), torch.nan_to_num(b, nan=12345.0)), (shape, k)
    elif form == "ppo_synthetic_world":      # mi_ppo_update_sharded on the P2P carrier with 1 .. 8 synthetic ranks (x + 0 + ... + 0) against mi_ppo_update, bit for bit
        import ctypes as C

        import deep_rl_amd.dist as DD
        import deep_rl_amd.engine as E
        import test_gpu_parity as P
        from deep_rl_amd import _native as N

        n = int(rng.choice([rng.integers(1, 9), rng.integers(9, 70), rng.integers(70, 700)]))
        T = int(rng.choice([rng.integers(1, 20), rng.integers(20, 129), rng.integers(129, 200)]))
        while (T * n) % 4:
            T += 1
        world, seed = int(rng.integers(1, 9)), int(rng.integers(1, 1000))
        shape = "forms case %d: ppo on %d synthetic ranks, envs %d, T %d, seed %d" % (case, world, n, T, seed)
        outs = []
        for synthetic in (False, True):
            h = C.c_void_p()
            if synthetic:
                N.check(N.lib().mi_comm_p2p_synthetic(world, 1 << 16, C.byref(h)), "mi_comm_p2p_synthetic")
                DD.use_comm(h)
                monkeypatch.setattr(E, "_FORCE_NATIVE_SHARDED", True)
            try:
                eng = P._engine(dev, n, seed=seed, T_=T, max_episodes_logged=0)
                eng.reset()
                for _ in range(2):
                    eng.update()
                torch.cuda.synchronize()
                if synthetic:
                    N.check(N.lib().mi_comm_check(h), "mi_comm_check")
                o = eng.optimizer
                outs.append([t.clone() for t in (eng.agent.flat, o.exp_avg, o.exp_avg_sq, eng.grads, eng.loss_terms, o.grad_norm, eng.advantages)])
            finally:
                monkeypatch.setattr(E, "_FORCE_NATIVE_SHARDED", False)
                DD.use_comm(None)
                if h.value:
                    torch.cuda.synchronize()
                    N.lib().mi_comm_destroy(h)
        for k, (a, b) in enumerate(zip(*outs)):
            assert torch.equal(torch.nan_to_num(a, nan=12345.0), torch.nan_to_num(b, nan=12345.0)), (shape, k)
    elif form == "shards":      # env-sharding invariance of the acting / rollout launches: an engine of n envs == two engines of n1 + n2 envs with env_id_base 0 / n1 (keyed RNG)
        import deep_rl_amd as D
        import test_gpu_dqn as Q
        import test_gpu_parity as P

        n = int(rng.integers(2, 600))
        n1 = int(rng.integers(1, n))
        seed = int(rng.integers(1, 1000))
        T = int(rng.choice([rng.integers(1, 129), rng.integers(129, 200)]))
        shape = "forms case %d: shards, envs %d = %d + %d, T %d, seed %d" % (case, n, n1, n - n1, T, seed)
        whole = None
        for base, cnt in ((0, n), (0, n1), (n1, n - n1)):
            eng = P._engine(dev, cnt, seed=seed, env_id_base=base, T_=T, max_episodes_logged=0, n_minibatch=1)
            pr = (torch.arange(9155, device=dev, dtype=torch.float32) % 7 - 3.0) * 0.01     # the same (non-initial) parameters for every shard
            eng.agent.load_flat(eng.agent.flat * 0 + pr)
            eng.reset(); eng.rollout_gae(); eng.rollout_gae()
            got = [eng.observations, eng.actions, eng.rewards, eng.dones, eng.values, eng.log_probs, eng.advantages]
            if whole is None:
                whole = [t.clone() for t in got]
            else:
                for k, (w, g) in enumerate(zip(whole, got)):
                    assert torch.equal(w[:, base:base + cnt], g), (shape, "ppo", base, k)
        S, steps = int(rng.integers(4, 40)), int(rng.integers(1, 12))
        whole = None
        for base, cnt in ((0, n), (0, n1), (n1, n - n1)):
            eng = Q._engine(dev, cnt, slots=S, seed=seed, base=base, batch_size=8, learning_starts=int(S // 2), total_timesteps=20 * S, max_episodes_logged=0)
            eng.reset()
            for _ in range(2 * S // steps + 2):
                eng.act(steps)
            got = [eng.observations, eng.actions, eng.rewards, eng.terminated]
            if whole is None:
                whole = [t.clone() for t in got]
            else:
                for k, (w, g) in enumerate(zip(whole, got)):
                    assert torch.equal(w[:, base:base + cnt], g), (shape, "dqn", base, k)
    elif form == "per":
        import test_gpu_per as E_

        n, slots, batch = int(rng.choice([rng.integers(1, 17), rng.integers(17, 700)])), int(rng.integers(12, 90)), int(rng.choice([rng.integers(1, 300), rng.integers(300, 3000)]))
        shape = "forms case %d: per pieces, envs %d, slots %d, batch %d" % (case, n, slots, batch)
        E_.test_one_call_pieces_are_bitwise_the_launch_sequence(dev, n, slots, batch)
    elif form == "per_incremental":      # production acting + training on a ring that fills and wraps: incremental chunk sums == a full pass, indices == mi_per_sample's == the oracle's
        import test_gpu_per as E_

        n, slots = int(rng.choice([rng.integers(1, 17), rng.integers(17, 700)])), int(rng.integers(12, 120))
        shape = "forms case %d: per incremental sums, envs %d, slots %d" % (case, n, slots)
        E_.test_incremental_sums_equal_full_pass_and_oracle(dev, R, n, slots)
    elif form == "dueling":
        import test_gpu_dueling as U

        batch = int(rng.choice([rng.integers(1, 300), rng.integers(300, 3000)]))
        shape = "forms case %d: dueling update, batch %d" % (case, batch)
        U.test_one_call_update_is_bitwise_the_launch_sequence(dev, batch)
    else:
        import test_gpu_sac as S_
        import test_gpu_sac_shadow as W

        # (the alpha step is owed to the next launch while that launch leaves half of the device's CUs free — mi_sac_owed_alpha_fits: batches up to ~1,000 —; beyond it is a launch of its own in both forms)
        batch = int(rng.choice([rng.integers(1, 64), rng.integers(64, 520), rng.integers(520, 1001 if form == "sac_owed_alpha" else 1400)]))
        shape = "forms case %d: %s, batch %d" % (case, form, batch)
        if form == "sac_owed_alpha":
            S_.test_owed_alpha_step_is_bit_identical_to_a_launch_of_its_own(dev, batch, monkeypatch)
        elif form == "sac_deferred_critic":
            S_.test_deferred_critic_step_is_bit_identical_to_a_launch_of_its_own(dev, batch, monkeypatch)
        else:
            W.test_shadows_change_no_bit(dev, batch)
    _log(shape + " ok")


@pytest.mark.parametrize("case", range(CASES))
def test_sac_acting_any_shape(dev, R, case):
    """sac.py:138-158 at a random (envs, slots, warm-up length): keyed warm-up actions, then the actor with supplied normal draws; per step the device's actions against the
    oracle's actor sample on the (bit-identical) observation, then the oracle env stepped with the DEVICE's actions: ring slot and carried-over observation bit-exact."""
    import deep_rl_amd as D

    rng = np.random.default_rng([SEED, 6, case])
    n = int(rng.choice([rng.integers(1, 17), rng.integers(17, 100), rng.integers(100, 500)]))
    S = int(rng.integers(4, 64))
    ls = int(rng.integers(0, 12))
    steps = int(rng.integers(S + 2, 2 * S + 20))
    seed, base = int(rng.integers(1, 1000)), int(rng.integers(0, 5000))
    shape = "sac acting case %d: envs %d, slots %d, learning_starts %d, %d steps, seed %d, base %d" % (case, n, S, ls, steps, seed, base)
    env = D.make("Pendulum-v1", num_envs=n, device=dev, seed=seed, env_id_base=base)
    torch.manual_seed(seed)
    a = D.Actor(env)
    qs = [D.SoftQNetwork(env) for _ in range(4)]
    a_p = (a.flat.cpu().numpy() + rng.normal(0, 0.03, R.AC_NPARAMS)).astype(np.float32)
    a.load_flat(a_p)
    eng = D.SACEngine(env, a, *qs, slots=S, batch_size=16, learning_starts=ls, max_episodes_logged=0)
    ref = R.VecPendulum(n, seed=seed, env_id_base=base)
    obs = ref.reset()
    assert np.array_equal(eng.reset().cpu().numpy(), obs), shape
    worst = 0.0
    for t in range(steps):
        eps = rng.standard_normal(n).astype(np.float32)
        eng.act(forced_eps=torch.from_numpy(eps) if t >= ls else None)
        a_dev = eng.actions[t % S].cpu().numpy()
        if t >= ls:
            ra, _ = R.sac_actor_sample(a_p, obs, eps)
            worst = max(worst, float(np.abs(a_dev - ra).max()))
        else:
            assert a_dev.min() >= -2.0 and a_dev.max() < 2.0, shape
        obs, rew, done, _, _ = ref.step(a_dev)
        s_ = (t + 1) % S
        assert np.array_equal(eng.observations[s_].cpu().numpy(), obs) and np.array_equal(eng.rewards[s_].cpu().numpy(), rew), (shape, t)
        assert np.array_equal(eng.observation.cpu().numpy(), obs), (shape, t)
    # (the fixed-size test's bar is 5e-6 over 1.1 M samples, 3.1e-6 measured; 22 M random samples reach 6.7e-6: u = mean + exp(log_std) eps with log_std = -1.5 + 3.5 head
    # (sac.py:69, no squashing), so one ulp of the head moves u by 3.5 |eps| std ulps — |eps| up to 5 and std up to e^2 in this many draws)
    assert worst <= 1.5e-5, (shape, worst)
    assert not eng.terminated.any(), shape
    eng.close()
    _log(shape + " ok")


@pytest.mark.parametrize("case", range(CASES))
def test_per_and_dueling_td_any_shape(dev, R, case):
    """per.py:131-147 (importance weights, weighted TD gradient, |td| -> priorities with the last duplicate winning, max_priority) and dueling_dqn.py:36-40,118-128 (the
    dueling head's gradient mapped back from the plain-DQN image) at a random (envs, slots, batch) on a ring filled by production acting, rows clear of ReLU kinks."""
    import test_gpu_dueling as U
    import test_gpu_per as E_

    rng = np.random.default_rng([SEED, 7, case])
    n = int(rng.choice([rng.integers(1, 17), rng.integers(17, 300)]))
    S = int(rng.integers(12, 60))
    batch = int(rng.choice([rng.integers(1, 130), rng.integers(130, 2049), rng.integers(2049, 4200)]))
    seed = int(rng.integers(1, 1000))
    # ---- PER
    shape = "per td case %d: envs %d, slots %d, batch %d, seed %d" % (case, n, S, batch, seed)
    eng = E_._engine(dev, n, S, seed=seed, batch_size=batch, learning_starts=0, total_timesteps=100 * S)
    params = (eng.q.flat.cpu().numpy() + rng.normal(0, 0.05, 10934)).astype(np.float32)
    tparams = (params + rng.normal(0, 0.05, 10934)).astype(np.float32)
    eng.q.load_flat(params); eng.target.load_flat(tparams)
    eng.reset()
    for _ in range(S // 10 + 2):
        eng.act(10)                                     # the ring is full and has wrapped; every row carries the running max_priority
    cap = S * n
    prio = rng.gamma(0.5, 1.0, cap).astype(np.float32)
    prio[rng.random(cap) < 0.2] = 0.0
    head = eng.global_step % S                          # the write head keeps priority 0 (never sampled: per.py:105 marks it at the next step)
    prio.reshape(S, n)[head] = 0.0
    eng.priorities.copy_(torch.from_numpy(prio.reshape(S, n)))
    eng.refresh_sums()
    mp0 = float(rng.uniform(0.01, 3.0))
    eng.max_priority.fill_(mp0)
    st = R.ReplayStorage(S, n)
    for name in ["observations", "actions", "rewards", "terminated"]:
        getattr(st, name)[...] = getattr(eng, name).cpu().numpy()
    clear = np.flatnonzero(_dqn_rows_clear(params, st.observations.reshape(cap, 4)) & (prio > 0))
    assert clear.size > 0.2 * cap, shape
    idx = rng.choice(clear, batch)                      # duplicates included: the scatter's "last duplicate wins"
    eng.sample(idx)
    s0, s1, total, total_alpha = R.per_sums(prio, cap, E_.ALPHA)
    w = R.per_weights(prio, idx, E_.ALPHA, np.float32(eng.beta()), total_alpha, cap)
    wd = eng.weights.cpu().numpy()
    assert np.allclose(wd, w, rtol=3e-5), shape
    eng.td_grad()
    og, ol, otd = R.per_td_grads(params, tparams, st, idx, wd)
    g = eng.grads.cpu().numpy()
    assert np.abs(g - og).max() <= 1e-5 * np.abs(og).max(), (shape, np.abs(g - og).max() / np.abs(og).max())
    assert abs(float(eng.loss.item()) - ol) <= 3e-5 * max(ol, 1e-6), (shape, float(eng.loss.item()), ol)
    td = eng.td_abs.cpu().numpy()
    assert np.allclose(td, np.abs(otd), rtol=2e-5, atol=2e-5), shape
    want = prio.copy()
    mp = R.per_update_priorities(want, idx, td, mp0)    # the scatter itself is index work: bit-exact given the device's |td|
    assert np.array_equal(eng.priorities.cpu().numpy().reshape(-1), want) and float(eng.max_priority) == mp, shape
    _log(shape + " ok")
    # ---- dueling
    shape = "dueling td case %d: envs %d, slots %d, batch %d, seed %d" % (case, n, S, batch, seed)
    eng = U._engine(dev, n, slots=S, seed=seed, batch_size=batch, learning_starts=0, total_timesteps=100 * S)
    dp = (eng.q.flat.cpu().numpy() + rng.normal(0, 0.05, eng.q.flat.numel())).astype(np.float32)
    dt = (dp + rng.normal(0, 0.05, dp.size)).astype(np.float32)
    eng.q.load_flat(dp); eng.target.load_flat(dt)
    eng.reset()
    for _ in range(S // 10 + 2):
        eng.act(10)
    st = R.ReplayStorage(S, n)
    for name in ["observations", "actions", "rewards", "terminated"]:
        getattr(st, name)[...] = getattr(eng, name).cpu().numpy()
    clear = np.flatnonzero(_dqn_rows_clear(eng.q.eff.cpu().numpy(), st.observations.reshape(cap, 4)))     # the image's feature layers are the dueling net's
    idx = rng.choice(clear, batch)
    eng.sample(idx)
    eng.td_grad()
    og, ol = R.dueling_td_grads(dp, dt, st, idx)
    g = eng.dueling_grads.cpu().numpy()
    assert np.abs(g - og).max() <= 1e-5 * np.abs(og).max(), (shape, np.abs(g - og).max() / np.abs(og).max())
    assert abs(float(eng.loss.item()) - ol) <= 3e-5 * max(ol, 1e-6), (shape, float(eng.loss.item()), ol)
    _log(shape + " ok")
