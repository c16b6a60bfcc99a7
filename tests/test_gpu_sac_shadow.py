"""Transposed copies of SAC's 256 x 256 layer-2 matrices (include/mi_rl.h mi_sac_shadow_*, round 5): the forward passes of the acting / update launches stream them with the
backward pass's access pattern — the same multiply-adds in the same order, ~1 us per pass faster.  What has to hold (sac.py:56-77 forward, :165-217 updates):
  * the engine with shadows == the engine without (MIRL_SAC_TRANSPOSED=0), BIT FOR BIT, over a run that acts, updates critics / actor / alpha / targets at two batch sizes;
  * the library keeps the shadows in step with its own fused optimizer steps (the shadow equals the transpose of the live matrix after the run);
  * parameters written through torch (load_flat, load_state_dict, copy_ on a parameter) are noticed through the version counters and answered with a refresh;
    `.data` in-place writes are not — params_changed() is the documented way;
  * the library's other writers (mi_adam, mi_polyak) mark a shadow invalid, and an invalid shadow only means the plain access pattern."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    return torch.device("cuda", 0)


def _engine(dev, transposed, batch, seed=3, n_envs=64):
    import deep_rl_amd as D
    import deep_rl_amd.sac_engine as SE

    old = SE._TRANSPOSED
    SE._TRANSPOSED = transposed
    try:
        env = D.make("Pendulum-v1", num_envs=n_envs, device=dev, seed=seed)
        torch.manual_seed(seed)
        a = D.Actor(env)
        qs = [D.SoftQNetwork(env) for _ in range(4)]
        qs[2].load_state_dict(qs[0].state_dict()); qs[3].load_state_dict(qs[1].state_dict())
        eng = D.SACEngine(env, a, *qs, slots=64, batch_size=batch, learning_starts=8, max_episodes_logged=0)
    finally:
        SE._TRANSPOSED = old
    assert bool(eng._shadows) == transposed
    return eng


def _run(eng, steps):
    eng.reset()
    for _ in range(steps):
        eng.act()
        if eng.global_step >= 8:
            eng.train_step(2, 1)
    eng.flush()
    torch.cuda.synchronize()
    return [t.clone() for t in (eng.actor.flat, eng.q_flat, eng.qt_flat, eng.log_alpha, eng.q_losses, eng.actor_out, eng.observations, eng.actions, eng.rewards)]


def _transposed_of(flat, is_actor):
    from deep_rl_amd import _native as N

    off = 1024 if is_actor else 1280      # AC_W2 = 768 + 256, SQ_W2 = 1024 + 256: W1 | b1 in front of the layer-2 matrix
    nets = 1 if is_actor else 2
    per = N.SAC_ACTOR_NPARAMS if is_actor else N.SAC_Q_NPARAMS
    return torch.cat([flat[n * per + off:n * per + off + 65536].view(256, 256).t().contiguous().reshape(-1) for n in range(nets)])


@pytest.mark.parametrize("batch", [256, 100, 1024])
def test_shadows_change_no_bit(dev, batch):
    a = _run(_engine(dev, True, batch), 40)
    b = _run(_engine(dev, False, batch), 40)
    for x, y in zip(a, b):
        assert torch.equal(x, y)
    assert torch.isfinite(a[0]).all() and torch.isfinite(a[1]).all()


def test_library_keeps_shadows_in_step_with_its_fused_steps(dev):
    from deep_rl_amd import _native as N

    eng = _engine(dev, True, 256)
    _run(eng, 40)
    for flat, sh, _, _ in eng._shadows:
        assert N.lib().mi_sac_shadow_valid(N.ptr(flat)) == 1
    (af, ash, _, _), (qf, qsh, _, _), (tf, tsh, _, _) = eng._shadows
    assert torch.equal(ash, _transposed_of(af, True)) and torch.equal(qsh, _transposed_of(qf, False)) and torch.equal(tsh, _transposed_of(tf, False))


def test_torch_side_writes_are_noticed(dev):
    """After each kind of torch-side write, the engine with shadows must behave like the engine without: same action for the same observation and noise."""
    import deep_rl_amd as D

    e1, e0 = _engine(dev, True, 256), _engine(dev, False, 256)
    for e in (e1, e0):
        _run(e, 12)
    gen = torch.Generator().manual_seed(5)

    def same():
        for e in (e1, e0):
            e.act(); e.train_step(2, 1)
        e1.flush(); e0.flush(); torch.cuda.synchronize()
        return torch.equal(e1.actions, e0.actions) and torch.equal(e1.actor.flat, e0.actor.flat) and torch.equal(e1.q_flat, e0.q_flat) and torch.equal(e1.qt_flat, e0.qt_flat)

    assert same()
    new_actor = (torch.randn(e1.actor.flat.numel(), generator=gen) * 0.05).to(dev)
    for e in (e1, e0):
        e.actor.load_flat(new_actor)                                   # flat.copy_
    assert same()
    sd = {k: v + 0.01 for k, v in e0.qf1.state_dict().items()}
    for e in (e1, e0):
        e.qf1.load_state_dict(sd)                                      # parameter.copy_ under no_grad: the parameters' own version counters
    assert same()
    for e in (e1, e0):
        with torch.no_grad():
            list(e.qf2_target.parameters())[2].mul_(1.01)              # the target critic's layer-2 matrix itself
    assert same()
    for e in (e1, e0):
        list(e.actor.parameters())[2].data.mul_(0.99)                  # `.data`: invisible to version counters ...
    e1.params_changed()                                                # ... the documented way to say so
    assert same()


def test_other_writers_invalidate_and_invalid_means_plain_pattern(dev):
    from deep_rl_amd import _native as N

    e1, e0 = _engine(dev, True, 256), _engine(dev, False, 256)
    for e in (e1, e0):
        _run(e, 12)
    L = N.lib()
    qt = e1._shadows[2][0]
    assert L.mi_sac_shadow_valid(N.ptr(qt)) == 1
    for e in (e1, e0):
        e.update_targets()                                             # mi_polyak: does not maintain the copy
    assert L.mi_sac_shadow_valid(N.ptr(qt)) == 0
    e1._shadows[2][3] = sum(t._version for t in e1._shadows[2][2])     # (undo the engine's own note that update_targets() leaves: here the LIBRARY's flag is under test)
    for e in (e1, e0):                                                 # the critic launch now takes the plain form (not all of its shadows are valid) — same bits
        e.act(); e.train_step(2, 1)
    e1.flush(); e0.flush(); torch.cuda.synchronize()
    assert torch.equal(e1.q_flat, e0.q_flat) and torch.equal(e1.qt_flat, e0.qt_flat) and torch.equal(e1.actor.flat, e0.actor.flat)
    assert L.mi_sac_shadow_valid(N.ptr(qt)) == 0                       # (a fused step does not resurrect an invalid copy)
    e1.params_changed()
    e1._sync_shadows()
    assert L.mi_sac_shadow_valid(N.ptr(qt)) == 1
    assert torch.equal(e1._shadows[2][1], _transposed_of(qt, False))
