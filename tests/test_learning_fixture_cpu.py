"""tests/golden/learning_stats.npz (oracle/capture_learning_stats.py: the UNMODIFIED reference scripts over seeds 1..50, sac.py 1..30 or more; 1..10 until round 6) is what tests/test_gpu_learning.py compares the
drop-in scripts' learning behaviour with.  Here: the fixture is complete, holds numbers only, its statistic is reproducible from its own episode lists, and its seed-1
runs ARE the runs the trace fixtures hold (same `global_step=…, episodic_return=…` lines: ppo.py:130, dqn.py:110-111) — so the seed remapping done from outside the
reference (env.seed / np.random.seed / torch.manual_seed / action_space.seed shifted by seed - 1) is the identity at seed 1."""
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TRACE = {"ppo": "ppo_ref_trace.npz", "dqn": "dqn_ref_trace.npz", "dueling_dqn": "dueling_ref_trace.npz", "per": "per_ref_trace.npz", "sac": "sac_ref_trace.npz"}


@pytest.fixture(scope="module")
def stats():
    with np.load(os.path.join(ROOT, "tests", "golden", "learning_stats.npz")) as z:
        return {k: z[k] for k in z.files}


@pytest.mark.parametrize("script", ["ppo", "dqn", "dueling_dqn", "per"])
def test_fixture_is_complete_and_pinned_to_the_trace_run(stats, script):
    seeds, off = stats[script + "_seeds"], stats[script + "_offsets"]
    rets, steps, lt = stats[script + "_episode_return"], stats[script + "_episode_global_step"], stats[script + "_last_tenth_mean"]
    n = len(seeds)
    assert n >= 50 and seeds.tolist() == list(range(1, n + 1)) and len(off) == n + 1 and off[0] == 0 and off[-1] == len(rets) == len(steps) and len(lt) == n
    for k in (seeds, off, rets, steps, lt):
        assert k.dtype.kind in "iuf"                                   # numbers only: no source text travels
    for i in range(n):
        r = rets[off[i]:off[i + 1]].astype(np.float64)
        assert len(r) > 50 and abs(float(np.mean(r[-max(len(r) // 10, 1):])) - lt[i]) < 1e-4
        assert np.all(np.diff(steps[off[i]:off[i + 1]]) > 0)
    with np.load(os.path.join(ROOT, "tests", "golden", TRACE[script])) as g:   # seed 1 == the reference run the parity fixtures were captured from
        assert np.array_equal(g["episode_global_step"], steps[off[0]:off[1]]) and np.allclose(g["episode_return"], rets[off[0]:off[1]])
    assert lt.std() > 5.0                                               # different runs, not one run many times


def test_sac_seeds(stats):
    """sac.py on Pendulum-v1: thirty seeds or more since round 6 (7.5 CPU-minutes each in the build container); seed 1 is the run the trace fixture holds."""
    off, rets = stats["sac_offsets"], stats["sac_episode_return"]
    n = len(stats["sac_seeds"])
    assert n >= 30 and stats["sac_seeds"].tolist() == list(range(1, n + 1)) and len(off) == n + 1 and np.diff(off).tolist() == [150] * n
    assert np.isfinite(rets).all() and len(stats["sac_last_tenth_mean"]) == n and stats["sac_last_tenth_mean"].std() > 5.0
    with np.load(os.path.join(ROOT, "tests", "golden", TRACE["sac"])) as g:
        assert np.allclose(g["episode_return"], rets[off[0]:off[1]])
