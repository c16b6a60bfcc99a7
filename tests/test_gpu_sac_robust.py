"""Robustness of the in-launch hand-offs of the SAC update kernels: every wait is bounded and surfaces as a return code,
roles are ordered so that a waiter only waits for workgroups dispatched before it (no co-residency of the grid needed: CU masks, other tenants), the
owed alpha step's epoch is engine-private (a checkpoint loaded into a trained engine resumes bit for bit) and its observation stash is double-buffered
(re-sampling between update_alpha() and the next update_actor() cannot clobber it).  Reference lines: sac.py:165-210."""
import json
import os
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


@pytest.fixture(autouse=True)
def _clean_status(dev):
    from deep_rl_amd import _native as N

    N.check(N.lib().mi_sac_test_fault(0), "mi_sac_test_fault")
    N.check(N.lib().mi_sac_clear_error(None, 0, N.stream_ptr(dev)), "mi_sac_clear_error")
    yield
    N.check(N.lib().mi_sac_test_fault(0), "mi_sac_test_fault")
    N.check(N.lib().mi_sac_clear_error(None, 0, N.stream_ptr(dev)), "mi_sac_clear_error")


def _engine(dev, batch=256, n_envs=64, slots=64, seed=3):
    import deep_rl_amd as D

    env = D.make("Pendulum-v1", num_envs=n_envs, device=dev, seed=seed)
    torch.manual_seed(seed)
    actor = D.Actor(env)
    qs = [D.SoftQNetwork(env) for _ in range(4)]
    qs[2].load_state_dict(qs[0].state_dict()); qs[3].load_state_dict(qs[1].state_dict())
    eng = D.SACEngine(env, actor, *qs, slots=slots, batch_size=batch, learning_starts=4)
    eng.reset()
    return eng


def _steps(eng, n):
    for _ in range(n):
        eng.act()
        if eng.global_step > 6:
            eng.train_step()


def _state(eng):
    return [t.clone() for t in (eng.actor.flat, eng.q_flat, eng.qt_flat, eng.log_alpha, eng.alpha, eng._alpha_m, eng._alpha_v, eng.actor_optimizer.exp_avg,
                                eng.actor_optimizer.exp_avg_sq, eng.q_optimizer.exp_avg, eng.q_optimizer.exp_avg_sq, eng.observations, eng.actions)]


@pytest.mark.parametrize("mode,batch", [(1, 256), (1, 600), (2, 256)])
def test_lost_producer_is_an_error_code_never_a_hang(dev, tmp_path, mode, batch):
    """A producer that never publishes (injected: mi_sac_test_fault) makes the waiting workgroups run out of their 100 ms budget: the launches END, their losses are
    NaN, NO optimizer step is applied from then on (parameters, Adam moments, targets, log_alpha stay finite and frozen: ADVICE r03 — a spurious timeout must not destroy
    the state), mi_sac_check and the next update call return MI_ESTATE with text — and after restoring a checkpoint + clear_error the run continues bit for bit."""
    from deep_rl_amd import _native as N
    from deep_rl_amd import checkpoint as CK

    eng = _engine(dev, batch)
    _steps(eng, 12)
    eng.check()
    ck = CK.save(str(tmp_path / "good"), eng)
    _steps(eng, 4)
    want = _state(eng)
    CK.load(ck, eng)
    try:
        N.check(N.lib().mi_sac_test_fault(mode), "mi_sac_test_fault")
        try:
            for _ in range(2):               # quad / split forms lose their hand-off words (mode 1) or the owed alpha step's epoch (mode 2: the debt rides on the next launch)
                eng.act(); eng.train_step()
        except N.MiError as e:               # a later call of the sequence may already see the status word: that IS the contract (MI_ESTATE, no hang)
            assert "rc=-4" in str(e)
        torch.cuda.synchronize()             # returns: nothing hangs
    finally:
        N.check(N.lib().mi_sac_test_fault(0), "mi_sac_test_fault")
    with pytest.raises(N.MiError, match="timed out waiting for a sibling workgroup"):
        eng.check()
    with pytest.raises(N.MiError, match="rc=-4"):     # MI_ESTATE, sticky: every later update call refuses
        eng.update_critic()
    # the poisoned launch's losses are NaN (the critics' when a hand-off word was lost, the actor's — whose loss needs alpha first — when the epoch was) ...
    assert not bool(torch.isfinite(eng.q_losses).all() and torch.isfinite(eng.actor_out).all())
    # ... but every optimizer step behind the timeout was withheld: nothing holds a NaN, and further launches leave the state where it is
    frozen = [t.clone() for t in (eng.actor.flat, eng.q_flat, eng.qt_flat, eng._log_alpha, eng._alpha_m_t, eng.actor_optimizer.exp_avg, eng.actor_optimizer.exp_avg_sq,
                                  eng.q_optimizer.exp_avg, eng.q_optimizer.exp_avg_sq)]
    assert all(bool(torch.isfinite(t).all()) for t in frozen)
    N.check(N.lib().mi_polyak(N.ptr(eng.qt_flat), N.ptr(eng.q_flat), eng.q_flat.numel(), 0.5, N.stream_ptr(dev)), "mi_polyak")   # an unfused step: withheld too
    torch.cuda.synchronize()
    assert torch.equal(frozen[2], eng.qt_flat)
    CK.load(ck, eng)
    eng.clear_error()
    eng.check()
    _steps(eng, 4)
    eng.check()
    for a, b in zip(want, _state(eng)):
        assert torch.equal(a, b)


def test_checkpoint_loaded_into_a_trained_engine_resumes_bit_for_bit(dev, tmp_path):
    """the epoch of the in-launch alpha hand-off used to be alpha_steps, which load() rewinds — consumers then skipped the wait.  Now engine-private."""
    from deep_rl_amd import checkpoint as CK

    a = _engine(dev)
    _steps(a, 12)
    ck = CK.save(str(tmp_path / "a12"), a)
    _steps(a, 9)
    want = _state(a)
    b = _engine(dev)
    _steps(b, 30)                       # trained further than the checkpoint: its epoch word is far ahead of the checkpoint's alpha_steps, and a debt is pending
    assert b._owed is not None and b.alpha_steps > 12
    CK.load(ck, b)
    assert b._owed is None and not b._stash_fresh
    _steps(b, 9)
    b.check()
    for x, y in zip(want, _state(b)):
        assert torch.equal(x, y)


def test_owed_alpha_reads_the_stash_of_its_own_actor_update(dev, monkeypatch):
    """update_actor(); update_alpha(); sample(); update_actor() — the second actor update carries the first one's alpha debt and stashes a NEW batch in the
    same launch.  The debt must see the first batch's observations (double-buffered stash): identical to paying every alpha step with a launch of its own."""
    import deep_rl_amd.sac_engine as SE

    def run(owe):
        monkeypatch.setattr(SE, "_OWE_ALPHA", owe)
        eng = _engine(dev)
        _steps(eng, 10)
        rng = np.random.default_rng(5)
        seen = 0
        for k in range(6):
            eng.sample(rng.integers(0, eng.global_step * eng.N, eng.batch_size))
            eng.update_actor(); eng.update_alpha()
            seen += eng._owed is not None
        return eng, seen

    a, seen_a = run(True)
    b, seen_b = run(False)
    assert seen_a == 6 and seen_b == 0
    for x, y in zip(_state(a), _state(b)):
        assert torch.equal(x, y)


def _child(extra_env, batch=256):
    env = dict(os.environ, PYTHONPATH=ROOT, MIRL_TEST_BATCH=str(batch))
    env.update(extra_env)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "_sac_masked_worker.py")], env=env, capture_output=True, text=True, timeout=60, cwd=ROOT)
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-3000:])
    line = [ln for ln in out.stdout.splitlines() if ln.startswith("SAC_WORKER ")][-1]
    return json.loads(line[len("SAC_WORKER "):])


def test_cu_masked_process_runs_the_fallback_or_the_ordered_roles_never_hangs(dev):
    """A fresh child process whose environment holds a 16-CU mask before it initialises the GPU: the library sees the mask (hipDeviceProp does not), runs the
    single-workgroup forms, and — with the mask deliberately ignored, so that 80 sibling workgroups are launched onto 16 CUs — still finishes, because waiters only wait
    for workgroups dispatched before them.  60 s subprocess timeout; all digests equal the unmasked run's (every form is bit-identical)."""
    full = _child({})
    assert full["finite"] and full["usable_cus"] == full["device_cus"] and full["owed_fits"] and full["owed_seen"] > 3
    masked = _child({"HSA_CU_MASK": "0:0-15"})
    assert masked["usable_cus"] == 16 and not masked["owed_fits"] and masked["owed_seen"] == 0
    forced = _child({"HSA_CU_MASK": "0:0-15", "MIRL_SAC_IGNORE_CU_MASK": "1"})
    assert forced["usable_cus"] == forced["device_cus"] and forced["owed_seen"] > 3
    glob = _child({"ROC_GLOBAL_CU_MASK": "0xffffffff"})
    assert glob["usable_cus"] == 32
    assert full["digest"] == masked["digest"] == forced["digest"] == glob["digest"]
