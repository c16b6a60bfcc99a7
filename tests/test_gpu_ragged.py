"""Ragged and degenerate shapes of the small PPO launches, through the C ABI, against the oracle (edge cases beside the reference's one shape):
GAE at T from 1 to 300 and N from 1 to 333 with every done pattern (none, all, random), the Feistel permutation at sizes around powers of two, the advantage
statistics with minibatches that are not a multiple of anything, gamma / lambda at their limits."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    import torch

    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    return torch.device("cuda", 0)


@pytest.fixture(scope="module")
def R():
    from oracle import cpu_ref

    return cpu_ref


@pytest.mark.parametrize("T,n", [(1, 1), (1, 333), (2, 7), (5, 64), (127, 65), (128, 3), (129, 130), (300, 17)])
@pytest.mark.parametrize("dones", ["none", "all", "random"])
@pytest.mark.parametrize("gamma,lam", [(0.99, 0.95), (1.0, 1.0), (0.0, 0.5)])
def test_gae_any_shape_bit_exact(dev, R, T, n, dones, gamma, lam):
    import torch

    from deep_rl_amd import _native as N

    rng = np.random.default_rng(1000 * T + n)
    st = R.Storage(T, n)
    st.rewards[:] = rng.normal(1, 2, (T + 1, n)).astype(np.float32)
    st.values[:] = rng.normal(0, 10, (T + 1, n)).astype(np.float32)
    st.dones[:] = {"none": 0.0, "all": 1.0}.get(dones, (rng.random((T + 1, n)) < 0.2).astype(np.float32))
    d = {k: torch.from_numpy(getattr(st, k)).to(dev) for k in ("rewards", "dones", "values")}
    adv, ret = torch.full((T + 1, n), 7.0, device=dev), torch.full((T + 1, n), 7.0, device=dev)
    N.check(N.lib().mi_gae(N.ptr(d["rewards"]), N.ptr(d["dones"]), N.ptr(d["values"]), T, n, gamma, lam, N.ptr(adv), N.ptr(ret), N.stream_ptr(dev)), "mi_gae")
    R.gae(st, gamma, lam)
    assert np.array_equal(adv.cpu().numpy(), st.advantages) and np.array_equal(ret.cpu().numpy(), st.returns)
    assert np.all(adv[T].cpu().numpy() == 0.0)                  # ppo.py:144-151 never writes row T of advantages; returns[T] = values[T]
    assert np.array_equal(ret[T].cpu().numpy(), st.values[T])


@pytest.mark.parametrize("n", [1, 2, 3, 4, 5, 15, 16, 17, 255, 257, 1023, 1025, 65535, 65537, 99991, 262145])
def test_perm_sizes_around_powers_of_two(dev, R, n):
    import torch

    from deep_rl_amd import _native as N

    for key in (N.lib().mi_perm_key(1, 0, 0), N.lib().mi_perm_key(2 ** 50 + 3, 777, 3)):
        out = torch.full((n + 2,), -5, dtype=torch.int32, device=dev)
        N.check(N.lib().mi_make_perm(n, key, N.ptr(out), N.stream_ptr(dev)), "mi_make_perm")
        o = out.cpu().numpy()
        assert o[n] == -5 and o[n + 1] == -5                    # nothing written past the end
        assert np.array_equal(o[:n], R.make_perm(n, key)) and np.array_equal(np.sort(o[:n]), np.arange(n))


@pytest.mark.parametrize("rows,mb,n_mb", [(7, 7, 1), (100, 33, 3), (1000, 250, 4), (4099, 1366, 3), (70000, 17500, 4)])
def test_adv_stats_ragged_minibatches(dev, R, rows, mb, n_mb):
    """sums[k] = {sum, sum of squares, count} over idx[k mb .. (k + 1) mb): any mb, idx need not cover the rows (mb n_mb <= rows)."""
    import torch

    from deep_rl_amd import _native as N

    rng = np.random.default_rng(rows)
    adv = rng.normal(-2, 5, rows).astype(np.float32)
    idx = rng.permutation(rows)[: mb * n_mb].astype(np.int32)
    sums = torch.full((n_mb, 3), 99.0, dtype=torch.float64, device=dev)
    adv_d, idx_d = torch.from_numpy(adv).to(dev), torch.from_numpy(idx).to(dev)    # (named: a temporary would be freed, and its block reused, before the launch reads it)
    N.check(N.lib().mi_adv_stats(N.ptr(adv_d), N.ptr(idx_d), mb, n_mb, N.ptr(sums), N.stream_ptr(dev)), "mi_adv_stats")
    s = sums.cpu().numpy()
    for k in range(n_mb):
        a = adv[idx[k * mb:(k + 1) * mb]].astype(np.float64)
        assert s[k, 2] == mb
        assert abs(s[k, 0] - a.sum()) <= 1e-12 * np.abs(a).sum() + 1e-12 and abs(s[k, 1] - (a * a).sum()) <= 1e-12 * (a * a).sum() + 1e-12
        mean, std = R.adv_stats(adv, idx[k * mb:(k + 1) * mb])  # the oracle's (mean, unbiased std) from the same sums
        var = max((s[k, 1] - s[k, 0] * (s[k, 0] / mb)) / (mb - 1.0), 0.0) if mb > 1 else float("nan")
        assert abs(s[k, 0] / mb - mean) <= 1e-9 * max(1.0, abs(mean))
        if mb > 1:
            assert abs(var ** 0.5 - std) <= 1e-6 * max(1.0, std)
