"""GPU parity tests of the dueling-DQN epilogue (mi_dueling_pack / mi_dueling_unpack_grads around the DQN kernels) against the CPU
oracle's explicit `values + (advantages - mean advantages)` head and the golden vectors of the unmodified reference dueling_dqn.py."""
import os
import re
import subprocess
import sys

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
def du_trace():
    with np.load(os.path.join(ROOT, "tests", "golden", "dueling_ref_trace.npz")) as z:
        return {k: z[k] for k in z.files}


def _engine(dev, n_envs, slots, params=None, seed=1, **kw):
    import deep_rl_amd as D

    env = D.make("CartPole-v1", num_envs=n_envs, device=dev, seed=seed)
    torch.manual_seed(seed)
    q = D.DuelingQNetwork(env); tgt = D.DuelingQNetwork(env)
    if params is not None:
        q.load_flat(params)
    tgt.load_state_dict(q.state_dict())
    opt = D.ClipAdam(q, lr=2.5e-4, eps=1e-8)
    kw.setdefault("max_episodes_logged", 0)
    return D.DuelingDQNEngine(env, q, tgt, opt, slots=slots, **kw)


def test_module_surface_and_forward(dev, R):
    import deep_rl_amd as D

    env = D.make("CartPole-v1", num_envs=1, device=dev)
    torch.manual_seed(1)
    q = D.DuelingQNetwork(env)
    assert [n for n, _ in q.named_parameters()] == ["feauture_layer.0.weight", "feauture_layer.0.bias", "feauture_layer.2.weight", "feauture_layer.2.bias",
                                                    "value_stream.weight", "value_stream.bias", "advantage_stream.weight", "advantage_stream.bias"]
    assert q.flat.numel() == 11019
    rng = np.random.default_rng(0)
    p = (q.flat.cpu().numpy() + rng.normal(0, 0.1, 11019)).astype(np.float32)
    q.load_flat(p)
    obs = (rng.normal(0, 1, (7, 100, 4)) * np.array([2.4, 3, 0.2, 3])).astype(np.float32)
    out = q(torch.from_numpy(obs).to(dev))
    ref = R.dueling_forward(p, obs).reshape(7, 100, 2)
    assert out.shape == (7, 100, 2) and np.abs(out.cpu().numpy() - ref).max() < 3e-6 * max(1.0, np.abs(ref).max())
    t = D.DuelingQNetwork(env)
    t.load_state_dict(q.state_dict())
    assert torch.equal(t.flat, q.flat) and torch.equal(t.eff, q.eff)


def test_td_grad_vs_reference_checkpoints(dev, R, du_trace):
    """Un-chained: the reference's online / target dueling parameters, batch indices and (replayed) storage at updates
    1000 ... 9000 -> loss and dueling gradient of the device path against the reference's autograd and the oracle."""
    from tests.test_gpu_dqn import _replay_reference, _upload

    g = du_trace
    st = _replay_reference(R, g, 100_000)
    eng = _engine(dev, 1, slots=100_001)
    _upload(eng, st)
    for i, k in enumerate(g["ck_update"]):
        eng.q.load_flat(g["ck_params"][i]); eng.target.load_flat(g["ck_target"][i])
        eng.sample(g["ck_inds"][i])
        eng.td_grad()
        grads = eng.dueling_grads.cpu().numpy(); loss = float(eng.loss.item())
        og, ol = R.dueling_td_grads(g["ck_params"][i], g["ck_target"][i], st, g["ck_inds"][i])
        scale = np.abs(g["ck_grads"][i]).max()
        assert np.abs(grads - g["ck_grads"][i]).max() <= 1e-5 * scale, (k, np.abs(grads - g["ck_grads"][i]).max() / scale)
        assert np.abs(grads - og).max() <= 1e-5 * scale
        assert abs(loss - g["ck_loss"][i]) <= 2e-5 * g["ck_loss"][i] and abs(loss - ol) <= 2e-5 * ol


def test_first_200_updates_chained_on_device(dev, R, du_trace):
    """The reference's first 200 TD updates chained through the device's own Adam on the dueling parameters (reference indices,
    replayed storage): every loss within 3e-5 relative, parameters within 2e-6 of the reference's after the first 8 steps."""
    from tests.test_gpu_dqn import _replay_reference, _upload
    from tests.test_oracle_dueling_pinned import regenerate_batch_inds_dueling

    g = du_trace
    inds = regenerate_batch_inds_dueling()
    st = _replay_reference(R, g, 12_100)
    eng = _engine(dev, 1, slots=100_001, params=g["init_params"])
    _upload(eng, st)
    losses = []
    for k in range(200):
        eng.train_step(inds[k])
        losses.append(eng.loss.clone())
        if k < 8:
            assert np.abs(eng.q.flat.cpu().numpy() - g["full_params"][k]).max() < 2e-6, k
        if (10_000 + 10 * k) % 500 == 0:
            eng.sync_target()
    losses = torch.cat(losses).cpu().numpy()
    rel = np.abs(losses - g["loss_all"][:200]) / np.maximum(np.abs(g["loss_all"][:200]), 1e-3)
    assert rel.max() < 3e-5, rel.max()
    assert abs(eng.q.flat.double().sum().item() - g["psum_all"][199]) < 2e-4


def test_acting_follows_the_dueling_q_values(dev, R):
    """Greedy decisions of the acting kernel (run on the packed parameters) agree with the oracle's explicit dueling head."""
    n, S = 96, 16
    eng = _engine(dev, n, slots=S, seed=5, learning_starts=0, start_e=0.0, end_e=0.0, total_timesteps=100)
    rng = np.random.default_rng(2)
    params = (eng.q.flat.cpu().numpy() + rng.normal(0, 0.1, 11019)).astype(np.float32)
    eng.q.load_flat(params)
    obs = eng.reset().cpu().numpy()
    checked = 0
    for step in range(12):
        eng.act(1)
        a = eng.actions[step % S].cpu().numpy()
        qv = R.dueling_forward(params, obs)
        sure = np.abs(qv[:, 0] - qv[:, 1]) > 1e-4
        assert np.array_equal(a[sure], (qv[sure, 1] > qv[sure, 0]).astype(np.int64)), step
        checked += int(sure.sum())
        obs = eng.observation.cpu().numpy()
    assert checked > 1000


def test_script_reference_shape_n1():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    env = dict(os.environ, PYTHONPATH=ROOT, NUM_ENVS="1", TOTAL_TIMESTEPS="6000")
    code = ("import runpy, json; g = runpy.run_module('deep_rl_amd.dueling_dqn', run_name='__main__');"
            "print('GLOBALS', json.dumps({k: g[k] for k in ['env_id','total_timesteps','learning_starts','train_frequency','batch_size','global_step','memory_size']}));"
            "print('NETS', type(g['q_network1']).__name__, g['q_network1'].flat.numel(), g['q_network2'].flat.numel(), g['optimizer'].step_count); print('LOSS', g['loss'])")
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("global_step=")]
    assert len(lines) > 100 and all(re.fullmatch(r"global_step=\d+, episodic_return=\d+\.\d\d", ln) for ln in lines)
    assert '"learning_starts": 600' in out.stdout and '"global_step": 6000' in out.stdout
    assert "NETS DuelingQNetwork 11019 11019 541" in out.stdout     # updates at steps 600, 610, ..., 6000
    assert np.isfinite(float(out.stdout.split("LOSS")[1].split()[0]))


@pytest.mark.parametrize("batch", [128, 96, 1000])
def test_one_call_update_is_bitwise_the_launch_sequence(dev, batch):
    """mi_dueling_td_update (round 5: TD launch + ONE launch that sums the slabs, maps the gradient back, steps the dueling parameters and rewrites the plain-DQN image;
    batch 1000: the many-slab sum + three epilogue launches inside the call) against td_grad() + optimizer.step() + repack() — parameters, image, both moments, both
    gradients, loss and the drawn indices bit for bit over 12 chained steps with target syncs in between (dueling_dqn.py:109-137)."""
    import deep_rl_amd.dqn_engine as E

    def run(fused):
        E._FORCE_SHARDED = not fused          # the launch sequence: sample, TD + slab sum, unpack, clip + Adam, pack
        try:
            eng = _engine(dev, 64, slots=32, batch_size=batch, learning_starts=0, total_timesteps=10_000)
            eng.reset()
            out = []
            for k in range(12):
                eng.act(10)
                eng.train_step()
                if k % 5 == 4:
                    eng.sync_target()
                o = eng.optimizer
                out.append([t.clone() for t in (eng.q.flat, eng.q.eff, o.exp_avg, o.exp_avg_sq, eng.grads, eng.dueling_grads, eng.loss, eng.batch_inds)])
            assert o.step_count == 12
            return out
        finally:
            E._FORCE_SHARDED = False

    a, b = run(True), run(False)
    for k, (x, y) in enumerate(zip(a, b)):
        for name, u, v in zip(("params", "image", "exp_avg", "exp_avg_sq", "plain grads", "dueling grads", "loss", "indices"), x, y):
            assert torch.equal(u, v), (k, name, (u.float() - v.float()).abs().max().item())
    assert torch.isfinite(a[-1][0]).all() and not torch.equal(a[-1][0], a[0][0])
