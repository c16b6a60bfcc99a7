"""The branch of the owed optimizer step that only a multi-GPU run takes, executed on ONE GPU (VERDICT r03 missing #2).

With world_size > 1 the all-reduce changes the gradient after grad_reduce_kernel took its block sums of squares, so the next gradient launch's prologue (and the last
clip + Adam launch) recompute the clip coefficient of reference ppo.py:191 (`clip_grad_norm_`) from all 9,155 all-reduced gradients (`norm_parts = nullptr`,
csrc/mi_update.hip ppo_update_impl).  `mi_ppo_test_assume_sharded(1)` forces exactly that on mi_ppo_update and mi_ppo_update_sharded without a second rank.  The two
evaluations of the norm are ONE expression tree (block_grad_norm), so everything must agree BITWISE with the default route: parameters, both Adam moments, the
pre-clip gradient norm, the loss terms, the last gradient — at 8 / 64 / 4096 envs, over three chained updates (48 optimizer steps).
The world_size-1 RCCL route with the hook on is checked in tests/_rccl_world1_worker.py (mode "native+assume")."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _run(n_envs, assume, updates=3):
    import deep_rl_amd as D
    from deep_rl_amd import engine as E

    dev = torch.device("cuda", 0)
    E.set_assume_sharded(assume)
    try:
        env = D.make("CartPole-v1", num_envs=n_envs, device=dev, seed=5)
        torch.manual_seed(5)
        agent = D.ActorCritic(env)
        opt = D.ClipAdam(agent, lr=2.5e-4, eps=1e-5, max_grad_norm=0.5)
        eng = D.PPOEngine(env, agent, opt, num_steps=128)
        eng.reset()
        norms = []
        for u in range(updates):
            opt.param_groups[0]["lr"] = (1.0 - u / 4) * 2.5e-4
            eng.update()
            norms.append(opt.grad_norm.clone())
        torch.cuda.synchronize()
        assert opt.step_count == 16 * updates
        return [t.clone() for t in (agent.flat, opt.exp_avg, opt.exp_avg_sq, eng.loss_terms, eng.grads, eng.advantages, eng.observations)] + norms
    finally:
        E.set_assume_sharded(False)


@pytest.mark.parametrize("n_envs", [8, 64, 4096])
def test_assume_sharded_route_is_bitwise_the_default_route(n_envs):
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    ref = _run(n_envs, False)
    got = _run(n_envs, True)
    for i, (a, b) in enumerate(zip(ref, got)):
        assert torch.equal(a, b), (n_envs, i, float((a.double() - b.double()).abs().max()))
    assert torch.isfinite(ref[0]).all() and float(ref[-1]) > 0.0


def test_explicit_sequence_equals_assume_sharded_route(monkeypatch):
    """The explicit launch sequence {mi_ppo_minibatch_grad, mi_clip_adam} x 16 (MIRL_PPO_SHARDED_SEQUENCE: what gloo runs walk) — every step a launch of its own,
    the norm recomputed from the gradient inside clip_adam_kernel — equals the one-call update whose owed steps recompute it on the gradient launches' weight staging:
    bit for bit, like the default route that reads grad_reduce_kernel's block sums."""
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    from deep_rl_amd import engine as E

    got = _run(64, True, updates=2)
    monkeypatch.setattr(E, "_FORCE_SHARDED_SEQUENCE", True)
    seq = _run(64, False, updates=2)
    for i, (a, b) in enumerate(zip(seq, got)):
        assert torch.equal(a, b), i
