"""Does the PRODUCTION random path learn like the reference?  (VERDICT r03 missing #3.)

Teacher-forced parity pins the arithmetic; it says nothing about keyed Philox action draws, Feistel permutations and in-launch sampling against the reference's torch /
numpy / gym generators (ppo.py:83-86, dqn.py:60-64).  Bit-level agreement is impossible there by construction, so the claim is distributional: over seeds 1..10 the mean
episodic return of the last tenth of a run's episodes (the `global_step=…, episodic_return=…` lines of ppo.py:130 / dqn.py:110-111) has the same distribution for
  * the UNMODIFIED reference scripts run on the CPU under oracle/gym_shim (tests/golden/learning_stats.npz, written by oracle/capture_learning_stats.py), and
  * the drop-in scripts `python -m deep_rl_amd.<script>` at the reference's own shape (NUM_ENVS=1, default budgets) on the MI355X, SEED=1..10 (sac.py on
    Pendulum-v1: sac.py:100-104,160-161).
Asserted per script: two-sided Mann-Whitney U p > 0.01 AND |difference of means| <= 2 pooled standard errors.  Both samples are written to
gpurun_out/learning_stats_gpu.json (copied to profiles/ when a round's numbers are recorded).  Deterministic: fixed seeds, counter-based streams."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_CODE = r"""
import contextlib, io, json, os, runpy, sys
name, seeds = sys.argv[1], [int(s) for s in sys.argv[2].split(',')]
out = {}
for s in seeds:
    os.environ.update(SEED=str(s), NUM_ENVS='1')
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        runpy.run_module('deep_rl_amd.' + name, run_name='__main__')
    out[s] = [float(ln.split('episodic_return=')[1]) for ln in buf.getvalue().splitlines() if ln.startswith('global_step=')]
print('LEARNING_JSON ' + json.dumps(out))
"""


def last_tenth(rets):
    return float(np.mean(rets[-max(len(rets) // 10, 1):]))


def _ours(script, seeds, extra_env=None):
    env = dict(os.environ, PYTHONPATH=ROOT)
    for k in ("SEED", "NUM_ENVS", "TOTAL_TIMESTEPS", "MEMORY_SIZE", "BATCH_SIZE", "LEARNING_STARTS", "PRINT_EPISODES", "MIRL_PPO_CONTRACTION"):
        env.pop(k, None)
    env.update(extra_env or {})
    out = subprocess.run([sys.executable, "-c", _CODE, script, ",".join(map(str, seeds))], env=env, capture_output=True, text=True, timeout=1500, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    line = [ln for ln in out.stdout.splitlines() if ln.startswith("LEARNING_JSON ")][0]
    return {int(k): v for k, v in json.loads(line[len("LEARNING_JSON "):]).items()}


def _compare(script, extra_env=None, key=None):
    from scipy.stats import mannwhitneyu

    g = np.load(os.path.join(ROOT, "tests", "golden", "learning_stats.npz"))
    SEEDS = g[script + "_seeds"].tolist()          # 1..10 for all five scripts (sac.py: ten seeds since round 5; the reference needs 7.5 CPU-minutes per seed)
    assert SEEDS == list(range(1, len(SEEDS) + 1)) and len(SEEDS) >= 6
    ref = g[script + "_last_tenth_mean"].astype(np.float64)
    off, rets = g[script + "_offsets"], g[script + "_episode_return"]
    assert np.allclose([last_tenth(rets[off[i]:off[i + 1]]) for i in range(len(SEEDS))], ref)   # the fixture's statistic is the one computed here
    runs = _ours(script, SEEDS, extra_env)
    ours = np.array([last_tenth(runs[s]) for s in SEEDS])
    p = float(mannwhitneyu(ours, ref, alternative="two-sided").pvalue)
    se = float(np.sqrt(ours.var(ddof=1) / len(ours) + ref.var(ddof=1) / len(ref)))
    rec = {"script": script, "env": extra_env or {}, "seeds": SEEDS, "statistic": "mean episodic return of the last tenth of the episodes of a run",
           "ours_gpu": [round(x, 2) for x in ours.tolist()], "reference_cpu": [round(x, 2) for x in ref.tolist()],
           "ours_mean": round(float(ours.mean()), 2), "reference_mean": round(float(ref.mean()), 2), "pooled_se": round(se, 2),
           "mean_difference_in_se": round(float(ours.mean() - ref.mean()) / se, 3), "mannwhitney_p": round(p, 4),
           "episodes_ours": [len(runs[s]) for s in SEEDS], "episodes_reference": np.diff(off).tolist()}
    path = os.path.join(ROOT, "gpurun_out", "learning_stats_gpu.json")
    os.makedirs(os.path.dirname(path), exist_ok=True)
    allrec = json.load(open(path)) if os.path.exists(path) else {}
    allrec[key or script] = rec
    json.dump(allrec, open(path, "w"), indent=1)
    print(json.dumps(rec))
    assert p > 0.01, rec
    assert abs(ours.mean() - ref.mean()) <= 2.0 * se, rec


@pytest.mark.parametrize("script", ["ppo", "dqn", "dueling_dqn", "per", "sac"])
def test_production_rng_path_learns_like_the_reference(script):
    import torch

    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    _compare(script)


def test_bf16x3_contraction_learns_like_the_reference():
    """The opt-in split-bf16 contraction mode (MIRL_PPO_CONTRACTION=bf16x3; include/mi_rl.h: an experiment with a specified error bound, never the headline) under the
    same ten-seed criterion as the default: ppo.py at the reference's shape, seeds 1..10."""
    import torch

    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    _compare("ppo", {"MIRL_PPO_CONTRACTION": "bf16x3"}, key="ppo_bf16x3")
