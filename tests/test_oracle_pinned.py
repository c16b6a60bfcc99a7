"""Pins the CPU oracle (oracle/cpu_ref.c) against golden vectors produced by the UNMODIFIED
reference /root/reference/deep_rl/ppo.py (tests/golden/ppo_ref_trace.npz) and the SURVEY §8a KAT.

CPU-only.  If these fail the oracle is wrong and no HIP parity claim means anything.
"""
import numpy as np
import pytest

from oracle import cpu_ref as R

T = 128


def test_hparams_match_reference(ref_trace):
    # ppo.py:62-76,83
    hp = ref_trace["hparams"]
    assert list(hp[:5]) == [20000, 128, 156, 32, 4]
    assert np.allclose(hp[5:], [0.99, 0.95, 2.5e-4, 0.2, 0.01, 0.5, 0.5, 1])


def test_cartpole_entire_run_bit_exact(ref_trace):
    """Teacher-forced actions + logged reset noise: obs_f32, termination, truncation, episode stats bit-exact
    over all 19,968 steps / 201 episodes of the reference run."""
    g = ref_trace
    resets, acts, obs_all, term = g["reset_states"], g["actions_all"], g["obs_all"], g["terminated_all"]
    env = R.VecCartPole(1)
    obs = env.reset(resets[:1])
    assert np.array_equal(obs[0], resets[0].astype(np.float32))
    assert np.array_equal(env.state[0], g["state_first"][0])
    ri, eps = 1, []
    for s in range(len(acts)):
        if s < len(g["state_first"]):
            assert np.array_equal(env.state[0], g["state_first"][s]), s
        fr = resets[ri:ri + 1] if ri < len(resets) else np.zeros((1, 4))
        o, r, d, tr, fret, flen = env.step([acts[s]], fr)
        assert r[0] == 1.0
        if d[0]:
            assert bool(term[s]) != bool(tr[0])  # done = terminated xor truncated
            assert np.array_equal(o[0], resets[ri].astype(np.float32))  # ppo.py:128-129: obs is the reset obs
            eps.append((s, fret[0], flen[0]))
            ri += 1
        else:
            assert term[s] == 0
            assert np.array_equal(o[0], obs_all[s]), s
    assert ri == len(resets)
    # ppo.py:130 prints global_step *before* the increment
    assert np.array_equal(np.array([e[0] for e in eps]), g["episode_global_step"])
    assert np.array_equal(np.array([e[1] for e in eps], np.float32), g["episode_return"])
    assert all(e[1] == e[2] for e in eps)  # CartPole: return == length
    assert max(e[2] for e in eps) <= 500


def _forced_resets(g, u, ri):
    fr = np.zeros((T, 1, 4))
    ar = g["after_reset_all"]
    for t in range(T):
        s = u * T + t
        if s + 1 < len(ar) and ar[s + 1]:
            fr[t, 0] = g["reset_states"][ri]
            ri += 1
    return fr, ri


def test_whole_reference_run_teacher_forced(ref_trace):
    """Replays all 156 updates / 2,496 optimizer steps with the reference's actions, reset noise and
    minibatch indices; every loss term, grad-norm and the final parameters must track the reference."""
    g = ref_trace
    acts = g["actions_all"].astype(np.int64)
    env, st = R.VecCartPole(1), R.Storage(T, 1)
    obs_cur = env.reset(g["reset_states"][:1])
    ri, k = 1, 0
    params = g["init_params"].copy()
    m, v = np.zeros_like(params), np.zeros_like(params)
    opt = g["opt_terms"]
    for u in range(156):
        fr, ri = _forced_resets(g, u, ri)
        if u < 3:
            assert np.abs(params - g["upd%d_params_before" % u]).max() < 1e-7
        R.rollout(env, params, st, obs_cur, forced_actions=acts[u * T:(u + 1) * T].reshape(T, 1), forced_resets=fr)
        R.gae(st)
        if u < 3:
            for nm in ["observations", "actions", "rewards", "dones"]:
                assert np.array_equal(getattr(st, nm)[:, 0], g["upd%d_%s" % (u, nm)]), (u, nm)
            for nm, tol in [("values", 5e-7), ("log_probs", 5e-7), ("advantages", 5e-6), ("returns", 5e-6)]:
                assert np.abs(getattr(st, nm)[:, 0] - g["upd%d_%s" % (u, nm)]).max() < tol, (u, nm)
        names = ["observations", "values", "actions", "log_probs", "rewards", "dones", "advantages", "returns"]
        sums = np.array([getattr(st, n).astype(np.float64).sum() for n in names])
        assert np.allclose(sums, g["update_sums"][u], rtol=2e-5, atol=2e-4), u
        lr = (1.0 - u / 156) * 2.5e-4  # ppo.py:107
        for _ in range(16):
            assert abs(lr - opt[k, 4]) < 1e-15
            grads, terms = R.minibatch(params, st, g["mb_inds"][k].astype(np.int32))
            if k < len(g["full_grads"]):
                rg = g["full_grads"][k]
                assert np.abs(grads - rg).max() <= 2e-6 * np.abs(rg).max(), k
            assert np.allclose(terms, opt[k, :4], rtol=2e-5, atol=5e-6), (k, terms, opt[k, :4])
            nrm = R.clip_grad_norm(grads)
            assert abs(nrm - g["clip_norm"][k]) <= 1e-5 * g["clip_norm"][k], k
            R.adam_step(params, grads, m, v, k + 1, lr)
            if k < len(g["full_params"]):
                assert np.abs(params - g["full_params"][k]).max() < 1e-7, k
            assert abs(params.astype(np.float64).sum() - opt[k, 5]) < 5e-5, k
            k += 1
    assert k == 2496 and ri == len(g["reset_states"])
    assert np.abs(params - g["final_params"]).max() < 2e-6
    ev = R.explained_var(st.values, st.returns)
    assert abs(ev - g["final_explained_var"][0]) < 1e-3 * abs(ev)


def test_survey_kat_gae_and_loss():
    """SURVEY.md §8a known-answer test (computed with the reference's exact torch expressions)."""
    st = R.Storage(4, 1)
    st.rewards[:, 0] = [0, 1, 1, 1, 1]
    st.dones[:, 0] = [0, 0, 1, 0, 0]
    st.values[:, 0] = [.5, .6, .7, .8, .9]
    R.gae(st, 0.99, 0.95)
    assert np.allclose(st.advantages[:, 0], [1.470200062, 0.399999976, 2.118085623, 1.091000080, 0], rtol=0, atol=1e-7)
    assert np.allclose(st.returns[:, 0], [1.970200062, 1.0, 2.818085670, 1.891000032, 0.899999976], rtol=0, atol=2e-7)
    nl, p, ent = R.categorical(np.array([[.1, -.2], [.3, 0], [-.5, .5], [0, 0]], np.float32))
    assert abs(ent.mean() - 0.659848809) < 1e-7
    mean, std = R.adv_stats(st.advantages.reshape(-1), np.arange(4))
    a = (st.advantages[:4, 0] - np.float32(mean)) / (np.float32(std) + np.float32(1e-8))
    assert np.allclose(a, [0.278925806, -1.210785866, 1.180778384, -0.248918146], atol=2e-7)
    acts = np.array([0, 1, 1, 0])
    ratio = np.exp(nl[np.arange(4), acts] - np.array([-.6, -.8, -.4, -.7], np.float32))
    pg = np.maximum(-a * ratio, -a * np.clip(ratio, 0.8, 1.2)).mean()
    assert abs(pg - -0.045590729) < 1e-7


def test_philox_known_answer():
    """Random123 philox4x32-10 KAT: counter=0,key=0 and the pi-digits vector."""
    L = R.lib()
    out = R.philox(0, 0, 0, 0)
    assert [hex(x) for x in out] == ["0x6627e8d5", "0xe169c58d", "0xbc57ac4c", "0x9b00dbd8"]


def test_feistel_is_a_permutation():
    for n in [1, 2, 3, 128, 1000, 1024, 4096 * 128 // 64]:
        p = R.make_perm(n, R.perm_key(1, 3, 2))
        assert np.array_equal(np.sort(p), np.arange(n))
    a, b = R.make_perm(4096, R.perm_key(1, 0, 0)), R.make_perm(4096, R.perm_key(1, 0, 1))
    assert (a != b).mean() > 0.99 and abs(np.corrcoef(a, np.arange(4096))[0, 1]) < 0.05


def test_reset_noise_and_uniform_ranges():
    s = np.array([R.reset_noise(1, e, k) for e in range(64) for k in range(4)])
    assert s.min() >= -0.05 and s.max() < 0.05 and abs(s.mean()) < 5e-3
    u = np.array([R.action_uniform(1, e, t) for e in range(32) for t in range(64)])
    assert u.min() >= 0 and u.max() < 1 and abs(u.mean() - 0.5) < 0.03


def test_sincos_modes_bridge(ref_trace):
    """The oracle's device-matched sin/cos mode ("fdlibm": the polynomial kernels the HIP engine evaluates, bit-identical
    on CPU and GPU) against the gym-faithful libm mode on the whole reference run: identical done flags / episode
    structure, observations equal except for a handful of steps that differ by one float32 ulp."""
    g = ref_trace
    resets, acts, obs_all, ar = g["reset_states"], g["actions_all"], g["obs_all"], g["after_reset_all"]
    R.set_sincos_mode("fdlibm")
    try:
        env = R.VecCartPole(1)
        env.reset(resets[:1])
        ri, n_mis = 1, 0
        for s in range(len(acts) - 1):
            o, r, d, tr, fret, flen = env.step([acts[s]], resets[ri:ri + 1] if ri < len(resets) else np.zeros((1, 4)))
            assert bool(d[0]) == bool(ar[s + 1]), s
            if d[0]:
                ri += 1
            else:
                diff = np.abs(o[0].astype(np.float64) - obs_all[s])
                assert (diff <= np.spacing(np.abs(obs_all[s]))).all(), s
                n_mis += int((diff > 0).any())
        assert n_mis <= 20, n_mis
    finally:
        R.set_sincos_mode("libm")
