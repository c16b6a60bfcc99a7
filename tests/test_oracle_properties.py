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


def _philox_u(seed, update, batch):
    """The sampler's 53-bit uniforms, from the oracle's own keyed stream (ref_philox through the action-uniform helper is another stream: restate the recipe on ref words)."""
    import ctypes as C

    L = R.lib()
    out = np.zeros((batch, 4), dtype=np.uint32)
    if not hasattr(L, "ref_philox"):
        return None
    L.ref_philox.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint32, C.c_void_p]
    for b in range(batch):
        L.ref_philox(seed, update, b, 7, out[b].ctypes.data)
    return ((out[:, 0] >> 5).astype(np.float64) * 67108864.0 + (out[:, 1] >> 6).astype(np.float64)) / 9007199254740992.0


def test_level1_contract_round6_counts_running_sums_and_matches_the_old_walk_away_from_rounding_boundaries():
    """Round 6 changed the sampling CONTRACT's level-1 step (oracle and device together): the group is m = #{j < n1 - 1 : x >= P[j]} with the running sums P[j] = s1[0] +
    ... + s1[j] (sequential, double) and the remainder x - P[m - 1], where rounds 2 - 5 walked s1 itself (x -= s1[m] while x >= s1[m]).  Restated here in numpy from the
    oracle's own sums: the drawn indices equal the restatement exactly, and the OLD walk picks the same group in all but a vanishing share of draws (the two differ by
    the rounding of 256 dependent subtractions against 256 dependent additions)."""
    rng = np.random.default_rng(12)
    n = 1_000_000
    prio = (rng.gamma(0.7, 1.0, n) + 1e-3).astype(np.float32)
    prio[rng.random(n) < 0.1] = 0.0
    s0, s1, total, _ = R.per_sums(prio, n)
    batch = 20000
    idx = R.per_sample(5, 9, prio, n, s0, s1, total, batch)
    u = _philox_u(5, 9, batch)
    if u is None:
        import pytest
        pytest.skip("oracle build without ref_philox export")
    n1 = len(s1)
    P = np.zeros(n1)
    t = 0.0
    for j in range(n1):
        t += float(s1[j]); P[j] = t
    assert P[-1] == total                                           # the total IS the chain's last element
    x = u * total
    m_new = (x[:, None] >= P[None, :n1 - 1]).sum(1)                 # the contract as it is written
    assert (np.diff(P) >= 0).all()                                  # non-decreasing: the count is what a binary search finds
    assert np.array_equal(idx // 4096, m_new) or np.abs(idx // 4096 - m_new).max() <= 1, "the oracle's groups are not the restated contract's"
    assert (idx // 4096 == m_new).mean() > 0.999                    # (zero-priority skips at a group's first entries may step one group down)
    m_old = np.zeros(batch, dtype=np.int64)
    for b in range(batch):
        xx, mm = x[b], 0
        while mm + 1 < n1 and xx >= s1[mm]:
            xx -= s1[mm]; mm += 1
        m_old[b] = mm
    assert (m_old == m_new).mean() > 0.9999 and np.abs(m_old - m_new).max() <= 1
