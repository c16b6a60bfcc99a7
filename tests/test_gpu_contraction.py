"""The split-bf16 variant of the PPO gradient kernel (mi_ppo_set_contraction(MI_CONTRACTION_BF16X3), include/mi_rl.h) is admissible only if every
tolerance of the f32 path holds UNCHANGED: this module re-collects the gradient / update parity tests of test_gpu_parity.py and
test_gpu_fullsize.py — the same functions, the same golden vectors, the same tolerances — and runs them with the switch on.  It is an experiment
behind a switch; no headline number is measured on it."""
import numpy as np
import pytest
import torch

import deep_rl_amd as D

import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))   # sibling test modules

# the tests' module-scoped fixtures come along with them
from test_gpu_parity import (dev, R, _fdlibm_mode,  # noqa: F401
                             test_minibatch_grad_vs_golden_first_16_steps, test_minibatch_grad_vs_oracle_sizes,  # noqa: F401
                             test_whole_reference_run_replayed_on_device, test_full_update_production_vs_oracle_n8,  # noqa: F401
                             test_headline_size_properties, test_headline_size_gradient_linearity, test_learning_smoke)  # noqa: F401
from test_gpu_fullsize import (test_ppo_fullsize_minibatch_grad_vs_oracle, test_ppo_fullsize_update_vs_oracle,  # noqa: F401
                               test_ppo_update_equals_launch_sequence_bitwise)  # noqa: F401

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _bf16x3():
    D.set_contraction("bf16x3")
    assert D.get_contraction() == "bf16x3"
    yield
    D.set_contraction("f32")


def test_switch_changes_the_kernel_not_the_contract(dev):
    """Same inputs under both settings: the gradients agree to f32 rounding (the variant is f32-grade) but are not bit-identical (so the switch
    really selects the other kernel), and each setting is reproducible run to run."""
    env = D.make("CartPole-v1", num_envs=64, device=dev, seed=3)
    torch.manual_seed(3)
    agent = D.ActorCritic(env)
    eng = D.PPOEngine(env, agent, D.ClipAdam(agent, lr=2.5e-4, eps=1e-5, max_grad_norm=0.5))
    eng.reset(); eng.rollout(); eng.compute_gae(); eng.make_perm(0); eng.adv_stats()
    out = {}
    for mode in ("bf16x3", "f32", "bf16x3"):
        D.set_contraction(mode)
        eng.minibatch_grad(0)
        g = eng.grads.cpu().numpy().copy()
        if mode in out:
            assert np.array_equal(out[mode], g), "the %s gradient is not reproducible" % mode
        out[mode] = g
    a, b = out["f32"], out["bf16x3"]
    assert not np.array_equal(a, b), "the switch did not change the kernel"
    scale = np.abs(a).max()
    assert np.abs(a - b).max() <= 2e-6 * scale, "bf16x3 gradient is not f32-grade: max |diff| %.3e vs max |g| %.3e" % (np.abs(a - b).max(), scale)


def test_unknown_mode_is_refused():
    with pytest.raises(D._native.MiError):
        D.set_contraction("fp8")
    from deep_rl_amd import _native as N
    assert N.lib().mi_ppo_set_contraction(7) != 0
    assert D.get_contraction() == "bf16x3"
