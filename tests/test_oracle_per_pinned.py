"""Pins the prioritized-replay part of the CPU oracle against golden vectors from the UNMODIFIED reference deep_rl/per.py, executed on
CartPole-v1 through the gym shim's id alias (tests/golden/per_ref_trace.npz, oracle/capture_per_trace.py).  CPU-only.
torch.multinomial's draws cannot be reproduced, so the reference's batch indices are replayed; the keyed sampler that replaces it in
production is tested against its own contract (proportionality, zero-priority exclusion, determinism)."""
import os

import numpy as np
import pytest

from oracle import cpu_ref as R

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TT, LS, ALPHA, BETA0 = 100_000, 10_000, 0.6, 0.4


@pytest.fixture(scope="module")
def per_trace():
    with np.load(os.path.join(ROOT, "tests", "golden", "per_ref_trace.npz")) as z:
        return {k: z[k] for k in z.files}


def replay_per(g, upto_update, on_update):
    """Teacher-forced replay of the reference run (its actions, its reset noise) with the priority bookkeeping of per.py:103-106."""
    R.lib().ref_set_num_threads(8)
    env = R.VecCartPole(1)
    st = R.ReplayStorage(TT + 1, 1)
    prio = np.zeros(TT + 1, np.float32)
    obs_cur = env.reset(g["reset_states"][:1])
    st.observations[0, 0] = obs_cur[0]
    acts = g["actions_all"].astype(np.int64); ar = g["after_reset_all"]; resets = g["reset_states"]
    ri, k, gs = 1, 0, 0
    state = {"max_prio": np.float32(1e-2)}
    while gs < TT and k < upto_update:
        n = 10
        fr = np.zeros((n, 1, 4))
        for s in range(n):
            if (gs + s + 1 < TT and ar[gs + s + 1]) or (gs + s + 1 == TT and ri < len(resets)):
                fr[s, 0] = resets[ri]; ri += 1
        R.dqn_act_steps(env, g["init_params"], st, obs_cur, n, gs, forced_actions=acts[gs:gs + n].reshape(n, 1), forced_resets=fr)
        prio[gs:gs + n] = state["max_prio"]          # per.py:106 (max_priority only changes at an update, i.e. between these blocks)
        gs += n
        if gs >= LS:
            on_update(k, gs, st, prio, state)
            k += 1
    return st, prio, gs


def beta_of(gs):
    return np.float32((1 - BETA0) * gs / TT + BETA0)


def test_hparams(per_trace):
    assert np.allclose(per_trace["hparams"], [TT, LS, 1, 0.05, 0.5, ALPHA, BETA0, 10, 128, 0.99, 2.5e-4, 500, 1])
    assert per_trace["init_params"].size == R.DQN_NPARAMS and per_trace["batch_inds_chain"].shape == (400, 128)


def test_reference_per_run_first_400_updates_chained(per_trace):
    """The first 400 updates chained through the oracle's own priorities, importance weights, weighted TD gradient and Adam, with the
    reference's batch indices: every loss, weight sum, priority sum, max_priority and parameter checksum tracks the reference."""
    g = per_trace
    n = R.DQN_NPARAMS
    net = {"p": g["init_params"].copy(), "t": g["init_params"].copy(), "m": np.zeros(n, np.float32), "v": np.zeros(n, np.float32)}

    def on_update(k, gs, st, prio, state):
        assert gs == g["train_global_step"][k]
        idx = g["batch_inds_chain"][k].astype(np.int64)
        _, _, _, total_alpha = R.per_sums(prio, gs, ALPHA)
        w = R.per_weights(prio, idx, ALPHA, beta_of(gs), total_alpha, gs)
        # measured: <= 3e-7 everywhere except single updates where the batch's smallest priority (a small |td| = a float32 difference of two
        # O(10) numbers, relative error ~1e-5) sets the normalising maximum weight: 3.9e-5 at update 269
        assert abs(w.astype(np.float64).sum() - g["wsum_all"][k]) <= 2e-4 * g["wsum_all"][k], (k, w.sum(), g["wsum_all"][k])
        grads, loss, td = R.per_td_grads(net["p"], net["t"], st, idx, w)
        if k < len(g["full_grads"]):
            assert np.abs(grads - g["full_grads"][k]).max() <= 5e-6 * np.abs(g["full_grads"][k]).max(), k
        assert abs(loss - g["loss_all"][k]) <= 2e-4 * max(abs(g["loss_all"][k]), 1e-3), (k, loss, g["loss_all"][k])
        state["max_prio"] = np.float32(R.per_update_priorities(prio, idx, td, state["max_prio"]))
        assert abs(prio.astype(np.float64).sum() - g["prio_sum_all"][k]) <= 2e-6 * g["prio_sum_all"][k], (k, prio.astype(np.float64).sum(), g["prio_sum_all"][k])
        assert abs(state["max_prio"] - g["max_prio_all"][k]) <= 2e-6 * g["max_prio_all"][k], (k, state["max_prio"], g["max_prio_all"][k])
        R.adam_step(net["p"], grads, net["m"], net["v"], k + 1, 2.5e-4, eps=1e-8)
        if k < len(g["full_params"]):
            assert np.abs(net["p"] - g["full_params"][k]).max() < 1e-7, k
        assert abs(net["p"].astype(np.float64).sum() - g["psum_all"][k]) < 2e-4, k
        if gs % 500 == 0:
            net["t"] = net["p"].copy()

    replay_per(g, 400, on_update)


def test_reference_per_late_checkpoints_unchained(per_trace):
    """Updates 1000 / 5000 / 9000 on their own: the reference's pre-update priorities, parameters, target and indices -> probabilities,
    importance weights, TD errors, loss, gradient and the new max_priority against the reference's values."""
    g = per_trace
    cks = {int(u): i for i, u in enumerate(g["ck_update"])}
    seen = []

    def on_update(k, gs, st, prio, state):
        if k not in cks:
            return
        i = cks[k]
        pre = g["ck_pre_%d" % k]
        assert gs == g["ck_gs"][i] and pre.size == gs + 1 and pre[gs] == 0.0
        idx = g["ck_inds"][i].astype(np.int64)
        _, _, total, total_alpha = R.per_sums(pre, gs, ALPHA)
        bprob = np.power(pre[idx], np.float32(ALPHA)) / np.float32(total_alpha)
        assert np.allclose(bprob, g["ck_bprob"][i], rtol=2e-5)
        w = R.per_weights(pre, idx, ALPHA, beta_of(gs), total_alpha, gs)
        assert np.allclose(w, g["ck_weights"][i], rtol=3e-5), k
        grads, loss, td = R.per_td_grads(g["ck_params"][i], g["ck_target"][i], st, idx, w)
        assert np.allclose(td, np.abs(g["ck_td"][i]), rtol=2e-5, atol=2e-4)    # td = difference of two O(100) Q-values: a few float32 ulps of those
        assert abs(loss - g["ck_loss"][i]) <= 2e-5 * abs(g["ck_loss"][i]), (k, loss, g["ck_loss"][i])
        assert np.abs(grads - g["ck_grads"][i]).max() <= 1e-5 * np.abs(g["ck_grads"][i]).max(), k
        post = pre.copy()
        mp = R.per_update_priorities(post, idx, td, float(pre.max()))
        assert abs(mp - g["ck_max_prio"][i]) <= 2e-5 * g["ck_max_prio"][i]
        seen.append(k)

    replay_per(g, 9001, on_update)
    assert seen == [1000, 5000, 9000]


def test_sampler_contract():
    """The keyed prefix-sum sampler: proportional to the priorities, never a zero-priority entry, reproducible, range-limited."""
    rng = np.random.default_rng(0)
    n = 70_000
    prio = rng.gamma(0.5, 1.0, n + 500).astype(np.float32)
    prio[rng.random(n + 500) < 0.2] = 0.0
    prio[n:] = 7.0                                   # beyond the valid range: must never be sampled
    s0, s1, total, total_alpha = R.per_sums(prio, n, ALPHA)
    assert abs(total - prio[:n].astype(np.float64).sum()) < 1e-9 * total
    assert abs(total_alpha - np.power(prio[:n], np.float32(ALPHA)).astype(np.float64).sum()) < 1e-5 * total_alpha
    idx = R.per_sample(5, 3, prio, n, s0, s1, total, 400_000)
    assert idx.min() >= 0 and idx.max() < n and (prio[idx] > 0).all()
    assert np.array_equal(idx, R.per_sample(5, 3, prio, n, s0, s1, total, 400_000))
    assert not np.array_equal(idx[:128], R.per_sample(5, 4, prio, n, s0, s1, total, 128))
    # proportionality: mass of 70 coarse bins, 400,000 draws => 3-sigma binomial bands with room to spare
    bins = np.arange(n) // 1000
    want = np.bincount(bins, weights=prio[:n].astype(np.float64)) / total
    got = np.bincount(bins[idx], minlength=70) / idx.size
    assert np.abs(got - want).max() < 5 * np.sqrt(want.max() / idx.size)
    # the single heaviest entries are hit in proportion too
    top = np.argsort(prio[:n])[-5:]
    for t in top:
        assert abs((idx == t).mean() - prio[t] / total) < 6 * np.sqrt(prio[t] / total / idx.size)


def test_alpha_zero_convention():
    """0^alpha := 0 for every alpha (ref_per_pow): at alpha = 0 the sum counts the written entries only and every weight is exactly 1 — what per.py:145-146
    gives under torch's 0 ** 0 = 1 as well, because the weights are normalised by their maximum."""
    rng = np.random.default_rng(2)
    n = 5000
    prio = rng.gamma(0.5, 1.0, n).astype(np.float32)
    prio[rng.random(n) < 0.3] = 0.0
    s0, s1, total, total_alpha = R.per_sums(prio, n, 0.0)
    assert total_alpha == float((prio > 0).sum()) and np.isfinite(total)
    idx = R.per_sample(1, 0, prio, n, s0, s1, total, 256)
    w = R.per_weights(prio, idx, 0.0, np.float32(0.4), total_alpha, n)
    assert (w == 1.0).all()
    # torch's convention on the same data: a different (larger) sum, identical weights
    torch_sum = float(n)
    w_torch = (np.float32(n) * (np.float32(1.0) / np.float32(torch_sum))) ** np.float32(-0.4)
    assert w_torch / w_torch == 1.0
