"""CPU checks of the learning-equivalence machinery (tests/test_gpu_learning.py runs it on the MI355X): the fixture tests/golden/learning_stats.npz is self-consistent and
large enough for an equivalence claim, and the TOST used there is the pair of one-sided Welch tests scipy computes."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))


def test_fixture_holds_enough_reference_runs_and_its_statistic_is_recomputable():
    import test_gpu_learning as L

    g = np.load(os.path.join(ROOT, "tests", "golden", "learning_stats.npz"))
    for script, n_min in (("ppo", 50), ("dqn", 50), ("dueling_dqn", 50), ("per", 50), ("sac", 30)):
        seeds = g[script + "_seeds"].tolist()
        assert seeds == list(range(1, len(seeds) + 1)) and len(seeds) >= n_min, (script, len(seeds))
        off, rets = g[script + "_offsets"], g[script + "_episode_return"]
        assert len(off) == len(seeds) + 1 and off[-1] == len(rets) == len(g[script + "_episode_global_step"])
        stat = np.array([L.last_tenth(rets[off[i]:off[i + 1]]) for i in range(len(seeds))])
        assert np.allclose(stat, g[script + "_last_tenth_mean"])
        # the margin of the equivalence test is a fraction of this spread: it must be a real spread, and the standard error of a 2 x n comparison well inside the margin
        sd = stat.std(ddof=1)
        assert sd > 0 and np.sqrt(2.0 / len(seeds)) < 0.5 * L.MARGIN, (script, sd)


def test_tost_is_the_two_one_sided_welch_tests():
    from scipy.stats import ttest_ind

    import test_gpu_learning as L

    rng = np.random.default_rng(7)
    for shift, scale in ((0.0, 1.0), (0.3, 2.0), (-0.9, 0.5)):
        a, b = rng.normal(shift, scale, 50), rng.normal(0.0, 1.0, 40)
        m = 0.75
        p, dof = L.tost_welch(a, b, m)
        p_low = ttest_ind(a + m, b, equal_var=False, alternative="greater").pvalue
        p_high = ttest_ind(a - m, b, equal_var=False, alternative="less").pvalue
        assert abs(p - max(p_low, p_high)) < 1e-12 and 38 < dof < 90
    a = rng.normal(0, 1, 50)
    assert L.tost_welch(a, a.copy(), 0.75)[0] < 1e-3            # same sample: equivalent
    assert L.tost_welch(a + 0.75, a.copy(), 0.75)[0] >= 0.49    # a shift of exactly the margin: never "equivalent"
    assert L.tost_welch(a, a.copy(), 0.05)[0] > 0.05            # a margin far below the standard error cannot be shown with 50 runs
