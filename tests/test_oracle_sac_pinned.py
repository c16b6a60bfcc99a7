"""Pins the SAC part of the CPU oracle against golden vectors from the UNMODIFIED reference deep_rl/sac.py, executed on
Pendulum-v1 through the gym shim's id alias (tests/golden/sac_ref_trace.npz, oracle/capture_sac_trace.py).  CPU-only."""
import os

import numpy as np
import pytest

from oracle import cpu_ref as R

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N_PROJ = 32


@pytest.fixture(scope="module")
def sac_trace():
    with np.load(os.path.join(ROOT, "tests", "golden", "sac_ref_trace.npz")) as z:
        return {k: z[k] for k in z.files}


def summarize(gvec):
    gvec = np.asarray(gvec, np.float32)
    P = np.random.default_rng(20260101 + gvec.size).standard_normal((N_PROJ, gvec.size)).astype(np.float32)
    return np.concatenate([P.astype(np.float64) @ gvec.astype(np.float64), [np.linalg.norm(gvec.astype(np.float64))]])


def close_summary(mine, ref, rtol):
    """32 fixed random projections + the norm of a gradient: a wrong gradient cannot pass."""
    scale = ref[-1]   # the gradient norm; a projection has std ~ norm
    return np.abs(mine - ref).max() <= rtol * scale


def _fill_storage(g, upto):
    """Replay the env (reference actions + reset noise forced) into a linear storage [upto+1]."""
    env = R.VecPendulum(1)
    st = R.SacStorage(30_001, 1)
    obs = env.reset(g["reset_states"][:1])
    st.observations[0, 0] = obs[0]
    ar, resets, acts = g["after_reset_all"], g["reset_states"], g["actions_all"]
    ri = 1
    ep = []
    for gs in range(upto):
        st.actions[gs, 0] = acts[gs]
        fr = None
        if (gs + 1 < len(ar) and ar[gs + 1]) or (gs + 1 == len(ar) and ri < len(resets)):
            fr = resets[ri:ri + 1]; ri += 1
        obs, rew, done, fret, flen = env.step(acts[gs:gs + 1], fr)
        st.observations[gs + 1, 0] = obs[0]; st.rewards[gs + 1, 0] = rew[0]; st.terminated[gs + 1, 0] = 0
        if done[0]:
            ep.append((gs + 1, fret[0]))
    return st, ep


def test_hparams(sac_trace):
    assert np.allclose(sac_trace["hparams"], [30_000, 5_000, 2, 256, 1, 0.99, 0.005, 3e-4, 1e-3, 1e-3, 1, -1.0])
    assert sac_trace["init_actor"].size == R.AC_NPARAMS and sac_trace["init_q"].size == 2 * R.SQ_NPARAMS


def test_pendulum_against_trace(sac_trace):
    """30,000 steps of the reference run (its actions, its reset noise).  numpy's float64 sin/cos (what gym's pendulum.py calls) is a
    SIMD routine that may differ from libm in the last bit, so observations are compared to 2 float32 ulps and rewards to 1e-6."""
    g = sac_trace
    st, ep = _fill_storage(g, 30_000)
    want = g["obs_first"]; ar = g["after_reset_all"].astype(bool)
    live = ~ar[1:len(want) + 1]
    got = st.observations[1:len(want) + 1, 0]
    assert (np.abs(got[live].astype(np.float64) - want[live]) <= 2 * np.spacing(np.maximum(np.abs(want[live]), 1e-3))).all()
    assert np.abs(st.rewards[1:, 0] - g["rewards_all"].astype(np.float32)).max() < 2e-5
    assert [e[0] for e in ep] == list(g["episode_global_step"])
    assert np.abs(np.array([e[1] for e in ep]) - g["episode_return"]).max() < 0.02   # printed with 2 decimals


CHAIN = 30


def test_chained_first_30_steps(sac_trace):
    """Global steps 5000..5029 (30 critic, 30 actor, 30 alpha updates, 30 polyak steps) chained through the oracle's own Adam with the
    reference's recorded noise stream: every loss, alpha, gradient summary and parameter checksum tracks the reference.  (Measured: the
    chain stays within 2e-6 / 3e-5 for ~30 steps, then SAC's lr-1e-3 Adam amplifies float32 rounding chaotically — by step 100 the
    gradient summaries differ by 35 %; no re-implementation can follow the reference further in a chained replay.  Late behaviour is
    pinned un-chained below.)"""
    g = sac_trace
    R.lib().ref_set_num_threads(8)
    st, _ = _fill_storage(g, 5_125)
    noise, lens = g["noise_chain"], g["noise_chain_lens"]
    off = np.concatenate([[0], np.cumsum(lens)])
    ni = 0

    def draw(n):
        nonlocal ni
        assert lens[ni] == n, (ni, lens[ni], n)
        e = noise[off[ni]:off[ni + 1]]; ni += 1
        return e

    actor = g["init_actor"].copy(); q = g["init_q"].copy(); qt = q.copy()
    la = g["init_log_alpha"].copy().astype(np.float32)
    am, av = np.zeros_like(actor), np.zeros_like(actor)
    qm, qv = np.zeros_like(q), np.zeros_like(q)
    lm, lv = np.zeros(1, np.float32), np.zeros(1, np.float32)
    alpha = float(np.exp(la[0]))
    ka = kal = 0
    ql, al, als = g["q_losses"], g["actor_losses"], g["alpha_steps"]
    for k in range(CHAIN):
        gs = 5000 + k
        if k > 0:   # the acting draw of the step that led to this global_step (sac.py:138-140): action must be the logged one
            e = draw(1)
            a, _ = R.sac_actor_sample(actor, st.observations[gs - 1, 0], e)
            assert abs(a[0] - g["actions_all"][gs - 1]) < 2e-5, (gs, a[0], g["actions_all"][gs - 1])
        idx = g["chain_inds"][k]
        grads, losses = R.sac_critic_grads(q, qt, actor, st, idx, draw(256), alpha)
        assert np.allclose(losses, ql[k, :2], rtol=3e-5, atol=1e-6), (k, losses, ql[k, :2])
        assert abs(alpha - ql[k, 2]) < 1e-6
        assert close_summary(summarize(grads), g["chain_q_gradsum"][k], 2e-5), k
        R.adam_step(q, grads, qm, qv, k + 1, 1e-3, eps=1e-8)
        assert abs(q.astype(np.float64).sum() - ql[k, 3]) < 1e-4, (k, q.astype(np.float64).sum(), ql[k, 3])
        if gs % 2 == 0:
            for _ in range(2):
                ag, aloss, _ = R.sac_actor_grads(actor, q, st, idx, draw(256), alpha)
                assert al[ka, 0] == gs and abs(aloss - al[ka, 1]) <= 3e-5 * max(1.0, abs(al[ka, 1])), (k, aloss, al[ka])
                assert close_summary(summarize(ag), g["chain_actor_gradsum"][ka], 3e-4), (k, ka)   # min(q1, q2) selection is discontinuous
                R.adam_step(actor, ag, am, av, ka + 1, 3e-4, eps=1e-8)
                assert abs(actor.astype(np.float64).sum() - al[ka, 3]) < 2e-3
                ka += 1
                mlp = R.sac_mean_logp(actor, st, idx, draw(256))
                agrad = np.array([-(mlp + -1.0)], np.float32)   # d/d log_alpha of mean(-log_alpha * (logp + target_entropy)), target_entropy = -1
                assert abs(la[0] - als[kal, 2]) < 2e-6 and abs(agrad[0] - als[kal, 3]) < 3e-5 * max(1.0, abs(als[kal, 3])), (k, agrad, als[kal])
                R.adam_step(la, agrad, lm, lv, kal + 1, 1e-3, eps=1e-8)
                alpha = float(np.exp(la[0]))
                kal += 1
        R.polyak(qt, q, 0.005)
    assert ka == CHAIN and kal == CHAIN


def test_late_checkpoint_unchained(sac_trace):
    """Global step 25,000 on its own: the reference's actor / critic / target parameters, indices and noise at that step ->
    the oracle's critic and actor losses and gradient summaries against the reference's autograd."""
    g = sac_trace
    R.lib().ref_set_num_threads(8)
    gs = int(g["ck_gs"][0])
    st, _ = _fill_storage(g, gs)
    grads, losses = R.sac_critic_grads(g["ck_q_params"][0], g["ck_q_target"][0], g["ck_actor_params"][0], st, g["ck_inds"][0],
                                       g["ck_noise_critic"][0], float(g["ck_alpha"][0]))
    assert np.allclose(losses, g["ck_q_losses"][0], rtol=3e-5), (losses, g["ck_q_losses"][0])
    assert close_summary(summarize(grads), g["ck_q_gradsum"][0], 2e-5)
    ag, aloss, _ = R.sac_actor_grads(g["ck_actor_params"][0], g["ck_actor_qparams"][0], st, g["ck_inds"][0], g["ck_noise_actor"][0],
                                     float(g["ck_actor_alpha"][0]))
    assert abs(aloss - g["ck_actor_loss"][0]) <= 3e-5 * abs(g["ck_actor_loss"][0]), (aloss, g["ck_actor_loss"][0])
    assert close_summary(summarize(ag), g["ck_actor_gradsum"][0], 5e-5)
