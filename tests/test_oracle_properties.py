"""Property tests (hypothesis) of the contracts both sides are built on — the CPU oracle's versions, which the GPU tests pin the kernels to bit for bit:
the keyed Feistel permutation is a bijection for every size and key, the keyed draws stay in range and depend on every coordinate, GAE equals the textbook
recursion on random blocks (ppo.py:144-151), the advantage statistics equal numpy's, the prioritized sampler only ever returns rows with a non-zero
priority inside the filled range.  CPU only, bounded example counts."""
import numpy as np
from hypothesis import given, settings, strategies as st

from oracle import cpu_ref as R

U64 = st.integers(min_value=0, max_value=2**64 - 1)
SET = dict(max_examples=40, deadline=None)


@settings(**SET)
@given(n=st.integers(min_value=1, max_value=6000), key=U64)
def test_feistel_permutation_is_a_bijection(n, key):
    p = R.make_perm(n, key)
    assert p.min() == 0 and p.max() == n - 1 and np.unique(p).size == n


@settings(**SET)
@given(seed=U64, upd=st.integers(0, 2**31 - 1), ep=st.integers(0, 7))
def test_perm_key_separates_updates_and_epochs(seed, upd, ep):
    k = R.perm_key(seed, upd, ep)
    assert k != R.perm_key(seed, upd + 1, ep) and k != R.perm_key(seed, upd, ep + 1)


@settings(**SET)
@given(seed=U64, env=st.integers(0, 2**40), step=st.integers(0, 2**40))
def test_keyed_draws_in_range_and_keyed_by_every_coordinate(seed, env, step):
    u = R.action_uniform(seed, env, step)
    assert 0.0 <= u < 1.0
    r = R.reset_noise(seed, env, step)
    assert (r >= -0.05).all() and (r < 0.05).all()
    a = R.philox(seed, env, step, 1)
    assert not np.array_equal(a, R.philox(seed, env, step, 2)) and not np.array_equal(a, R.philox(seed, env + 1, step, 1))
    assert not np.array_equal(a, R.philox(seed, env, step + 1, 1)) and not np.array_equal(a, R.philox(seed ^ 1, env, step, 1))


@settings(**SET)
@given(T=st.integers(1, 40), N=st.integers(1, 9), seed=st.integers(0, 2**31 - 1), gamma=st.sampled_from([0.99, 0.9, 1.0]), lam=st.sampled_from([0.95, 0.0, 1.0]))
def test_gae_equals_the_reference_recursion(T, N, seed, gamma, lam):
    rng = np.random.default_rng(seed)
    s = R.Storage(T, N)
    s.rewards[:] = rng.normal(size=(T + 1, N)).astype(np.float32)
    s.values[:] = rng.normal(size=(T + 1, N)).astype(np.float32)
    s.dones[:] = (rng.random((T + 1, N)) < 0.15).astype(np.float32)
    R.gae(s, gamma, lam)
    adv = np.zeros((T + 1, N), np.float32)
    g32, l32 = np.float32(gamma), np.float32(lam)
    for t in range(T - 1, -1, -1):   # ppo.py:146-150, float32 throughout as torch does
        nonterminal = np.float32(1.0) - s.dones[t + 1]
        delta = s.rewards[t + 1] + g32 * s.values[t + 1] * nonterminal - s.values[t]
        adv[t] = delta + g32 * l32 * nonterminal * adv[t + 1]
    assert np.abs(s.advantages - adv).max() <= 2e-6 * (1.0 + np.abs(adv).max())
    assert np.abs(s.returns - (s.advantages + s.values)).max() <= 1e-6 * (1.0 + np.abs(s.values).max() + np.abs(adv).max())


@settings(**SET)
@given(n=st.integers(8, 3000), m=st.integers(2, 400), seed=st.integers(0, 2**31 - 1))
def test_advantage_statistics_equal_numpy(n, m, seed):
    rng = np.random.default_rng(seed)
    adv = rng.normal(1.0, 3.0, n).astype(np.float32)
    idx = rng.integers(0, n, min(m, n)).astype(np.int32)
    mean, std = R.adv_stats(adv, idx)
    x = adv[idx].astype(np.float64)
    assert abs(mean - x.mean()) <= 1e-9 * (1 + abs(x.mean())) and abs(std - x.std(ddof=1)) <= 1e-9 * (1 + x.std(ddof=1))


@settings(max_examples=25, deadline=None)
@given(n=st.integers(1, 20000), seed=st.integers(0, 2**31 - 1), upd=st.integers(0, 10**6), frac_zero=st.sampled_from([0.0, 0.5, 0.95]))
def test_prioritized_sampler_returns_filled_rows_with_mass(n, seed, upd, frac_zero):
    rng = np.random.default_rng(seed)
    cap = n + int(rng.integers(0, 500))
    prio = rng.random(cap).astype(np.float32) + np.float32(1e-3)
    prio[rng.random(cap) < frac_zero] = 0.0
    prio[int(rng.integers(0, n))] = 1.0   # at least one row with mass inside the filled range
    s0, s1, total, _ = R.per_sums(prio, n)
    idx = R.per_sample(seed, upd, prio, n, s0, s1, total, 64)
    assert idx.min() >= 0 and idx.max() < n
    assert (prio[idx] > 0).all(), "a zero-priority row was drawn"
    assert abs(total - prio[:n].astype(np.float64).sum()) <= 1e-9 * total
