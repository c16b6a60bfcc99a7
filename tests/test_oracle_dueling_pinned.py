"""Pins the dueling-head part of the CPU oracle against golden vectors from the UNMODIFIED reference deep_rl/dueling_dqn.py
(tests/golden/dueling_ref_trace.npz, oracle/capture_dqn_trace.py --script dueling_dqn).  CPU-only."""
import os

import numpy as np
import pytest

from oracle import cpu_ref as R
from tests.test_oracle_dqn_pinned import _replay

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def du_trace():
    with np.load(os.path.join(ROOT, "tests", "golden", "dueling_ref_trace.npz")) as z:
        return {k: z[k] for k in z.files}


def regenerate_batch_inds_dueling():
    """dueling_dqn.py:67,93,120: np.random.seed(1); one random() on EVERY step (no learning_starts guard on the exploration draw),
    randint(global_step, size=128) whenever the incremented global_step >= learning_starts is a multiple of 10."""
    rng = np.random.RandomState(1)
    tt, ls, tf, bs = 100_000, 10_000, 10, 128
    inds = []
    for gs in range(tt):
        rng.random_sample()
        g1 = gs + 1
        if g1 >= ls and g1 % tf == 0:
            inds.append(rng.randint(g1, size=bs))
    return np.array(inds)


def test_hparams_and_indices(du_trace):
    g = du_trace
    assert np.allclose(g["hparams"], [100_000, 10_000, 1, 0.05, 0.5, 10, 128, 0.99, 2.5e-4, 500, 1])
    assert g["init_params"].size == R.DUELING_NPARAMS
    inds = regenerate_batch_inds_dueling()
    assert inds.shape == (9001, 128) and np.array_equal(inds[:64], g["batch_inds_first"]) and np.array_equal(inds.sum(axis=1), g["inds_sum_all"])


def test_dueling_forward_is_the_linear_head(du_trace):
    """values + (advantages - mean) == a plain head with W3[a] = Wv + Wa[a] - mean Wa (the identity the device path uses)."""
    p = du_trace["full_params"][3]
    rng = np.random.default_rng(0)
    obs = (rng.normal(0, 1, (500, 4)) * np.array([2.4, 3, 0.2, 3])).astype(np.float32)
    q = R.dueling_forward(p, obs)
    eff = np.empty(R.DQN_NPARAMS, np.float32)
    eff[:10764] = p[:10764]
    wv, bv, wa, ba = p[10764:10848], p[10848], p[10849:11017].reshape(2, 84), p[11017:11019]
    eff[10764:10932] = (wv[None] + (wa - wa.mean(0, keepdims=True))).reshape(-1)
    eff[10932:] = bv + (ba - ba.mean())
    assert np.abs(R.dqn_forward(eff, obs) - q).max() < 2e-6 * max(1.0, np.abs(q).max())


def test_reference_dueling_run_first_500_updates_chained(du_trace):
    """The first 500 TD updates CHAINED through the oracle's own Adam (full gradients / parameters for the first 8, loss and parameter
    checksum for all), then the late checkpoints un-chained — same structure and reasons as the plain-DQN pin."""
    g = du_trace
    inds = regenerate_batch_inds_dueling()
    n = R.DUELING_NPARAMS
    state = {"p": g["init_params"].copy(), "t": g["init_params"].copy(), "m": np.zeros(n, np.float32), "v": np.zeros(n, np.float32)}

    def on_update(k, gs, st):
        grads, loss = R.dueling_td_grads(state["p"], state["t"], st, inds[k])
        if k < len(g["full_grads"]):
            assert np.abs(grads - g["full_grads"][k]).max() <= 3e-6 * np.abs(g["full_grads"][k]).max(), k
        assert abs(loss - g["loss_all"][k]) <= 1e-5 * max(abs(g["loss_all"][k]), 1e-3), (k, loss, g["loss_all"][k])
        R.adam_step(state["p"], grads, state["m"], state["v"], k + 1, 2.5e-4, eps=1e-8)
        if k < len(g["full_params"]):
            assert np.abs(state["p"] - g["full_params"][k]).max() < 1e-7, k
        assert abs(state["p"].astype(np.float64).sum() - g["psum_all"][k]) < 1e-4, k
        if gs % 500 == 0:
            state["t"] = state["p"].copy()

    _replay(g, inds, 500, on_update)


def test_reference_dueling_late_checkpoints_unchained(du_trace):
    g = du_trace
    inds = regenerate_batch_inds_dueling()
    cks = {int(u): i for i, u in enumerate(g["ck_update"])}
    seen = []

    def on_update(k, gs, st):
        if k in cks:
            i = cks[k]
            assert np.array_equal(inds[k], g["ck_inds"][i])
            grads, loss = R.dueling_td_grads(g["ck_params"][i], g["ck_target"][i], st, inds[k])
            assert abs(loss - g["ck_loss"][i]) <= 1e-5 * abs(g["ck_loss"][i]), (k, loss, g["ck_loss"][i])
            assert np.abs(grads - g["ck_grads"][i]).max() <= 1e-5 * np.abs(g["ck_grads"][i]).max(), k
            seen.append(k)

    st, gs, ri = _replay(g, inds, 9001, on_update)
    assert seen == [1000, 2500, 5000, 7500, 9000] and gs == 100_000 and ri == len(g["reset_states"])
    assert int(st.terminated.sum()) == int(g["storage_terminated_sum"][0])
