"""GPU parity tests of the prioritized-replay epilogues (mi_per_* around the DQN kernels) against the CPU oracle and the golden vectors
of the unmodified reference per.py (run on CartPole-v1).  Sampler indices, scatter and max_priority bit-exact; fp32 tolerances at each assert."""
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ALPHA, BETA0 = 0.6, 0.4


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
def per_trace():
    with np.load(os.path.join(ROOT, "tests", "golden", "per_ref_trace.npz")) as z:
        return {k: z[k] for k in z.files}


def _engine(dev, n_envs, slots, params=None, seed=1, **kw):
    import deep_rl_amd as D

    env = D.make("CartPole-v1", num_envs=n_envs, device=dev, seed=seed)
    torch.manual_seed(seed)
    q = D.QNetwork(env); tgt = D.QNetwork(env)
    if params is not None:
        q.load_flat(params)
    tgt.load_state_dict(q.state_dict())
    opt = D.ClipAdam(q, lr=2.5e-4, eps=1e-8)
    kw.setdefault("max_episodes_logged", 0)
    return D.PERDQNEngine(env, q, tgt, opt, slots=slots, **kw)


@pytest.mark.parametrize("n_envs,slots,frac", [(7, 1000, 0.63), (64, 4096, 1.0), (1024, 1024, 1.0)])
def test_sampler_and_weights_bit_exact_indices(dev, R, n_envs, slots, frac):
    """Random priorities with zeros: the device's prefix sums, sampled indices (bit-exact) and importance weights vs the oracle."""
    eng = _engine(dev, n_envs, slots, seed=9, batch_size=1000, total_timesteps=10 * slots)
    rng = np.random.default_rng(slots)
    cap = slots * n_envs
    stored_steps = int(slots * frac)
    prio = rng.gamma(0.5, 1.0, cap).astype(np.float32)
    prio[rng.random(cap) < 0.15] = 0.0
    prio[stored_steps * n_envs:] = 0.0
    eng.priorities.copy_(torch.from_numpy(prio.reshape(slots, n_envs)))
    eng.refresh_sums()   # priorities written behind the engine's back: rebuild the sampler's chunk sums
    eng.global_step = stored_steps
    eng.update_index = 17
    eng.sample()
    n = stored_steps * n_envs
    s0, s1, total, total_alpha = R.per_sums(prio, n, ALPHA)
    idx = R.per_sample(9, 17, prio, n, s0, s1, total, 1000)
    got = eng.batch_inds.cpu().numpy()
    assert np.array_equal(got, idx) and (prio[got] > 0).all()
    w = R.per_weights(prio, idx, ALPHA, np.float32(eng.beta()), total_alpha, n)
    assert np.allclose(eng.weights.cpu().numpy(), w, rtol=2e-5) and abs(float(eng.weights.max()) - 1.0) < 1e-6
    # caller-supplied indices: weights only
    mine = rng.choice(np.flatnonzero(prio[:n] > 0), 1000)
    eng.sample(mine)
    assert np.array_equal(eng.batch_inds.cpu().numpy(), mine)
    assert np.allclose(eng.weights.cpu().numpy(), R.per_weights(prio, mine, ALPHA, np.float32(eng.beta()), total_alpha, n), rtol=2e-5)


def test_alpha_zero_is_uniform_per_not_nan(dev, R):
    """alpha = 0 (uniform PER, a legitimate per.py setting): 0^0 := 0, so never-written entries and the write head contribute +0 and
    nothing becomes NaN; every importance weight is exactly 1 (per.py:145-146 normalises by the maximum); indices bit-exact vs the oracle; the
    incremental sums of a ring that fills under acting + training equal a full pass."""
    n_envs, slots = 64, 64
    eng = _engine(dev, n_envs, slots, seed=9, batch_size=256, total_timesteps=10 * slots, alpha=0.0)
    rng = np.random.default_rng(3)
    cap = slots * n_envs
    prio = rng.gamma(0.5, 1.0, cap).astype(np.float32)
    prio[rng.random(cap) < 0.2] = 0.0
    prio[40 * n_envs:] = 0.0
    eng.priorities.copy_(torch.from_numpy(prio.reshape(slots, n_envs)))
    eng.refresh_sums()
    eng.global_step = 40
    eng.update_index = 5
    eng.sample()
    n = 40 * n_envs
    s0, s1, total, total_alpha = R.per_sums(prio, n, 0.0)
    assert total_alpha == float((prio[:n] > 0).sum())          # p^0 = 1 for written entries, 0 for the rest
    ws = eng._per_ws.view(torch.float64)
    assert torch.isfinite(ws[:2 * ((cap + 63) // 64)]).all()
    got = eng.batch_inds.cpu().numpy()
    assert np.array_equal(got, R.per_sample(9, 5, prio, n, s0, s1, total, 256)) and (prio[got] > 0).all()
    w = eng.weights.cpu().numpy()
    assert (w == 1.0).all()
    assert np.array_equal(w, R.per_weights(prio, got, 0.0, np.float32(eng.beta()), total_alpha, n))
    # production acting + training at alpha = 0: finite, incremental == full pass
    eng2 = _engine(dev, 64, 37, seed=4, batch_size=128, learning_starts=0, total_timesteps=3700, alpha=0.0)
    eng2.reset()
    for it in range(8):
        eng2.act(10); eng2.train_step()
    eng2.settle()   # (round 6: the one-call update leaves the scattered chunks' sums to the next acting launch)
    inc = eng2._per_ws.clone()
    eng2.refresh_sums()
    n0 = (37 * 64 + 63) // 64; n1 = (n0 + 63) // 64
    assert torch.equal(inc.view(torch.float64)[:2 * n0 + 2 * n1], eng2._per_ws.view(torch.float64)[:2 * n0 + 2 * n1])
    assert bool(torch.isfinite(eng2.weights).all()) and bool(torch.isfinite(eng2.q.flat).all()) and float(eng2.weights.max()) == 1.0


def test_scatter_last_duplicate_wins_and_max_priority(dev, R):
    eng = _engine(dev, 5, 100, batch_size=512)
    rng = np.random.default_rng(1)
    prio = rng.random(500).astype(np.float32)
    eng.priorities.copy_(torch.from_numpy(prio.reshape(100, 5)))
    idx = rng.integers(0, 500, 512); idx[100:140] = idx[60:100]          # plenty of duplicates
    td = rng.gamma(1.0, 2.0, 512).astype(np.float32)
    for start in (np.float32(0.01), np.float32(50.0)):
        eng.priorities.copy_(torch.from_numpy(prio.reshape(100, 5))); eng.max_priority.fill_(float(start))
        eng.batch_inds.copy_(torch.from_numpy(idx).to(dev)); eng.td_abs.copy_(torch.from_numpy(td).to(dev))
        from deep_rl_amd import _native as N
        N.check(N.lib().mi_per_update_priorities(N.ptr(eng.priorities), N.ptr(eng.batch_inds), N.ptr(eng.td_abs), 512, N.ptr(eng._owner), N.ptr(eng.max_priority),
                                                 N.stream_ptr(dev)), "mi_per_update_priorities")
        want = prio.copy()
        mp = R.per_update_priorities(want, idx, td, float(start))
        assert np.array_equal(eng.priorities.cpu().numpy().reshape(-1), want) and float(eng.max_priority) == mp
        assert (eng._owner == -1).all()


def test_mark_priorities_and_ring_head(dev, R):
    """per.py:106 for every env of every step of an acting call; the ring's write head gets priority 0 (never sampled)."""
    eng = _engine(dev, 6, 16, learning_starts=0, total_timesteps=1000)
    eng.reset()
    eng.max_priority.fill_(0.75)
    eng.act(10)
    p = eng.priorities.cpu().numpy()
    assert (p[:10] == 0.75).all() and (p[10:] == 0).all()
    eng.max_priority.fill_(1.5)
    eng.act(10)                                   # wraps: slots 10..15, 0..3 written, head = slot 4
    p = eng.priorities.cpu().numpy()
    assert (p[10:] == 1.5).all() and (p[:4] == 1.5).all() and (p[4] == 0).all() and (p[5:10] == 0.75).all()


@pytest.mark.parametrize("n_envs,slots", [(64, 37), (4096, 256)])
def test_incremental_sums_equal_full_pass_and_oracle(dev, R, n_envs, slots):
    """What PERDQNEngine runs (mi_per_mark_sums / mi_per_sample_current / mi_per_update_priorities_sums: only the touched chunks are
    recomputed) against the full-pass form and the oracle, on a ring that fills and wraps under production acting and training:
    the chunk sums are bit-identical to a full level-0 pass, the indices drawn bit-identical to mi_per_sample's and the oracle's."""
    from deep_rl_amd import _native as N

    eng = _engine(dev, n_envs, slots, seed=4, batch_size=128, learning_starts=0, total_timesteps=100 * slots)
    eng.reset()
    cap = slots * n_envs
    n0 = (cap + 63) // 64
    n1 = (n0 + 63) // 64
    for it in range(slots // 10 + 12):          # fills the ring and wraps it
        eng.act(10)
        eng.train_step()
        if it % 7 == 3 or it == slots // 10 + 11:
            eng.settle()                            # (round 6: the one-call update leaves the scattered chunks' sums to the next acting launch; nothing has carried them yet)
            inc = eng._per_ws.clone()
            eng.refresh_sums()
            assert torch.equal(inc.view(torch.float64)[:2 * n0 + 2 * n1], eng._per_ws.view(torch.float64)[:2 * n0 + 2 * n1]), it   # level-0 and level-1 sums of p and p^alpha
            prio = eng.priorities.cpu().numpy().reshape(-1)
            stored = min(eng.global_step, slots) * n_envs
            eng.sample()
            got = eng.batch_inds.cpu().numpy().copy()
            s0, s1, total, total_alpha = R.per_sums(prio, stored, ALPHA)
            assert np.array_equal(got, R.per_sample(4, eng.update_index, prio, stored, s0, s1, total, 128)), it
            full_idx = torch.zeros_like(eng.batch_inds); full_w = torch.zeros_like(eng.weights)
            ws2 = torch.empty_like(eng._per_ws)
            N.check(N.lib().mi_per_sample(4, eng.update_index, N.ptr(eng.priorities), stored, cap, float(stored), ALPHA, eng.beta(), 128, 1, N.ptr(ws2),
                                          N.ptr(full_idx), N.ptr(full_w), N.stream_ptr(dev)), "mi_per_sample")
            assert torch.equal(full_idx, eng.batch_inds) and torch.equal(full_w, eng.weights), it
    assert (eng.priorities > 0).sum().item() > 0.9 * cap


def _replay_storage(R, g, upto_steps):
    from tests.test_oracle_per_pinned import replay_per

    st, prio, gs = replay_per(g, 10**9 if upto_steps >= 100_000 else max(0, (upto_steps - 10_000) // 10 + 1), lambda *a: None)
    return st


def test_weighted_td_grad_vs_reference_checkpoints(dev, R, per_trace):
    """Un-chained: the reference's pre-update priorities, parameters, target and indices at updates 1000 / 5000 / 9000 -> device
    importance weights, |td|, loss, gradient and max_priority against the reference's own values (and the oracle's)."""
    g = per_trace
    st = _replay_storage(R, g, 100_000)
    eng = _engine(dev, 1, slots=100_001, total_timesteps=100_000)
    from tests.test_gpu_dqn import _upload
    _upload(eng, st)
    for i, k in enumerate(g["ck_update"]):
        gs = int(g["ck_gs"][i]); pre = g["ck_pre_%d" % k]
        eng.priorities.zero_(); eng.priorities[:gs + 1, 0].copy_(torch.from_numpy(pre))
        eng.refresh_sums()
        eng.max_priority.fill_(float(pre.max()))
        eng.global_step = gs
        eng.q.load_flat(g["ck_params"][i]); eng.target.load_flat(g["ck_target"][i])
        eng.sample(g["ck_inds"][i])
        assert np.allclose(eng.weights.cpu().numpy(), g["ck_weights"][i], rtol=5e-5), k
        eng.td_grad()
        assert np.allclose(eng.td_abs.cpu().numpy(), np.abs(g["ck_td"][i]), rtol=2e-5, atol=2e-4)
        scale = np.abs(g["ck_grads"][i]).max()
        assert np.abs(eng.grads.cpu().numpy() - g["ck_grads"][i]).max() <= 2e-5 * scale, k
        assert abs(float(eng.loss) - g["ck_loss"][i]) <= 3e-5 * g["ck_loss"][i]
        assert abs(float(eng.max_priority) - g["ck_max_prio"][i]) <= 2e-5 * g["ck_max_prio"][i]


def test_first_200_updates_chained_on_device(dev, R, per_trace):
    """The reference's first 200 updates chained on the device: its batch indices, the device's own priorities / weights / weighted TD
    gradient / Adam / scatter: loss, priority sum and max_priority track the reference."""
    g = per_trace
    st = _replay_storage(R, g, 12_100)
    eng = _engine(dev, 1, slots=100_001, params=g["init_params"], total_timesteps=100_000)
    from tests.test_gpu_dqn import _upload
    _upload(eng, st)
    eng.priorities[:10_000].fill_(1e-2)
    for k in range(200):
        gs = 10_000 + 10 * k
        eng.global_step = gs
        eng.refresh_sums()   # (this loop writes the new rows' priorities itself instead of acting)
        eng.train_step(g["batch_inds_chain"][k])
        assert abs(float(eng.loss) - g["loss_all"][k]) <= 3e-4 * max(abs(g["loss_all"][k]), 1e-3), k
        assert abs(eng.priorities.double().sum().item() - g["prio_sum_all"][k]) <= 5e-6 * g["prio_sum_all"][k], k
        assert abs(float(eng.max_priority) - g["max_prio_all"][k]) <= 5e-6 * g["max_prio_all"][k], k
        if k < 8:
            assert np.abs(eng.q.flat.cpu().numpy() - g["full_params"][k]).max() < 2e-6, k
        if gs % 500 == 0:
            eng.sync_target()
        eng.priorities[gs:gs + 10].copy_(eng.max_priority.expand(10, 1))      # per.py:106 for the next 10 steps


def test_script_reference_shape_n1():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    env = dict(os.environ, PYTHONPATH=ROOT, NUM_ENVS="1", TOTAL_TIMESTEPS="6000")
    code = ("import runpy, json; g = runpy.run_module('deep_rl_amd.per', run_name='__main__');"
            "print('GLOBALS', json.dumps({k: g[k] for k in ['env_id','total_timesteps','learning_starts','alpha','beta_0','train_frequency','batch_size','global_step','memory_size']}));"
            "print('PRIO', tuple(g['priorities'].shape), int((g['priorities'] > 0).sum()), g['max_priority'] >= 0.01, g['optimizer'].step_count); print('LOSS', g['loss'])")
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("global_step=")]
    assert len(lines) > 100 and all(re.fullmatch(r"global_step=\d+, episodic_return=\d+\.\d\d", ln) for ln in lines)
    assert '"alpha": 0.6' in out.stdout and '"beta_0": 0.4' in out.stdout and '"global_step": 6000' in out.stdout
    assert "PRIO (6001,) 6000 True 541" in out.stdout     # per.py:79: no env axis at one env
    assert np.isfinite(float(out.stdout.split("LOSS")[1].split()[0]))


def test_sampler_random_shapes_bit_exact(dev, R):
    """The sampler's fast paths (round 5: EXEC-predicated 16-step walks on full groups / whole chunks, coalesced wave loads of the per-draw chunks when a wave is full of
    draws) and their fall-backs meet at every boundary: ring sizes that leave a partial last level-1 group, exactly 64 level-1 entries, a partial last chunk, batches of
    1 / 63 / 65 / 128 / 300 draws (partially active waves, several passes of the draw loop) — indices bit-exact against the oracle in every case."""
    rng = np.random.default_rng(2025)
    shapes = [(1, 64 * 64 * 3 + 100, 1.0), (1, 64 * 64, 1.0), (1, 64 * 64 + 1, 1.0), (3, 700, 0.77), (17, 1000, 0.3), (64, 512, 1.0), (5, 13109, 0.9), (2, 2, 1.0)]
    for k, (n_envs, slots, frac) in enumerate(shapes):
        for batch in (1, 63, 65, 128, 300):
            eng = _engine(dev, n_envs, slots, seed=3 + k, batch_size=batch, total_timesteps=10 * slots)
            cap = slots * n_envs
            stored_steps = max(int(slots * frac), 1)
            prio = rng.gamma(0.5, 1.0, cap).astype(np.float32)
            prio[rng.random(cap) < 0.2] = 0.0
            prio[stored_steps * n_envs:] = 0.0
            if not (prio[:stored_steps * n_envs] > 0).any():
                prio[0] = 1.0
            eng.priorities.copy_(torch.from_numpy(prio.reshape(slots, n_envs)))
            eng.refresh_sums()
            eng.global_step = stored_steps
            eng.update_index = 100 + batch
            eng.sample()
            n = stored_steps * n_envs
            s0, s1, total, total_alpha = R.per_sums(prio, n, ALPHA)
            want = R.per_sample(3 + k, 100 + batch, prio, n, s0, s1, total, batch)
            got = eng.batch_inds.cpu().numpy()
            assert np.array_equal(got, want), (n_envs, slots, frac, batch, np.flatnonzero(got != want)[:5])
            assert (prio[got] > 0).all()


@pytest.mark.parametrize("n_envs,slots,batch", [(64, 150, 128), (4096, 32, 128), (512, 64, 1000), (512, 64, 2500)])
def test_one_call_pieces_are_bitwise_the_launch_sequence(dev, n_envs, slots, batch):
    """Round 6: one PER iteration = mi_per_act_steps (the acting launch also marks the new rows and rebuilds their sums, and carries the sums the last
    update left owed) + mi_per_td_update (sampler, weighted TD launch, slab sum + Adam with the priority scatter + max_priority on its last workgroup) — four launches —
    against the round-5 sequence of six (act, mi_per_mark_sums, mi_per_sample_current, TD, slab sum + Adam, mi_per_update_priorities_sums; MIRL_PER_ONE_CALL=0).  Over
    chained iterations on a ring that fills and WRAPS (so scattered entries fall into groups the next acting call re-marks: the one-writer rule of per_owed_sums_role),
    with two updates behind one acting call now and then (the settle path) and target syncs (batch 2500: beyond the riding workgroups' row lists — the acting call settles
    the owed sums in a launch of their own first): drawn indices, weights, |td|, priorities, max_priority, every level-0 /
    level-1 sum, parameters, both moments, gradient and loss bit for bit (per.py:92-153).  batch 1000: the many-slab sum launch carries the scatter."""
    import deep_rl_amd.dqn_engine as E

    cap = slots * n_envs
    n0 = (cap + 63) // 64
    n1 = (n0 + 63) // 64
    iters = 2 * slots // 10 + 7

    def run(one_call):
        E._PER_ONE_CALL = one_call
        try:
            eng = _engine(dev, n_envs, slots, seed=9, batch_size=batch, learning_starts=0, total_timesteps=100 * slots)
            eng.reset()
            out = []
            for k in range(iters):
                eng.act(10)
                eng.train_step()
                if k % 4 == 1:
                    eng.train_step()            # a second update without an acting call in between: the first one's owed sums are settled by a launch of their own
                if k % 5 == 4:
                    eng.sync_target()
                o = eng.optimizer
                row = [t.clone() for t in (eng.batch_inds, eng.weights, eng.td_abs, eng.priorities, eng.max_priority, eng.q.flat, o.exp_avg, o.exp_avg_sq, eng.grads, eng.loss)]
                if k % 3 == 0 or k == iters - 1:
                    eng.settle()
                    row.append(eng._per_ws.view(torch.float64)[:2 * n0 + 2 * n1].clone())
                out.append(row)
            return out
        finally:
            E._PER_ONE_CALL = True

    a, b = run(True), run(False)
    names = ("indices", "weights", "|td|", "priorities", "max_priority", "params", "exp_avg", "exp_avg_sq", "grads", "loss", "sums")
    for k, (x, y) in enumerate(zip(a, b)):
        assert len(x) == len(y)
        for name, u, v in zip(names, x, y):
            assert torch.equal(u, v), (k, name, (u.double() - v.double()).abs().max().item())
    assert torch.isfinite(a[-1][5]).all() and not torch.equal(a[-1][5], a[0][5]) and a[-1][4].item() > 1e-2
