#!/usr/bin/env python3
"""One-off study behind tests/test_gpu_learning.py (round 6): the same comparison with 200 seeds per script instead of 50, to tell a small real difference between the
drop-ins' and the reference's learning behaviour from seed noise (with 50 + 50 runs all five differences came out on the same side, +0.02 ... +0.30 reference standard
deviations; 200 + 200 runs put the standard error at 0.1).  Runs on the GPU box:
    python tools/learning_study.py [first_seed] [last_seed] [scripts]      ->  gpurun_out/r06_learning_study.json
The reference side (the UNMODIFIED scripts, oracle/capture_learning_stats.py --first-seed 51 --seeds 200 in the build container, 150 x 18 - 40 s per script) travels as
numbers: profiles/r06_learning_study_reference.json = {script: [last-tenth mean return of seed 1, 2, ...]}."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from scipy.stats import ks_2samp, mannwhitneyu
import test_gpu_learning as L

first = int(sys.argv[1]) if len(sys.argv) > 1 else 1
last = int(sys.argv[2]) if len(sys.argv) > 2 else 200
scripts = sys.argv[3].split(",") if len(sys.argv) > 3 else ["ppo", "dqn", "dueling_dqn", "per"]
L.WORKERS = int(os.environ.get("STUDY_WORKERS", "8"))
ref_all = json.load(open(os.path.join(ROOT, "profiles", "r06_learning_study_reference.json")))
out = {}
for script in scripts:
    seeds = list(range(first, min(last, len(ref_all[script])) + 1))
    ref = np.array([ref_all[script][s - 1] for s in seeds])
    runs = L._ours(script, seeds)
    ours = np.array([L.last_tenth(runs[s]) for s in seeds])
    s_ref = float(ref.std(ddof=1))
    se = float(np.sqrt(ours.var(ddof=1) / len(ours) + ref.var(ddof=1) / len(ref)))
    d = float(ours.mean() - ref.mean())
    out[script] = {"seeds": [seeds[0], seeds[-1]], "n": len(seeds), "ours_mean": round(float(ours.mean()), 2), "reference_mean": round(float(ref.mean()), 2),
                   "ours_sd": round(float(ours.std(ddof=1)), 2), "reference_sd": round(s_ref, 2), "difference": round(d, 2), "se_of_difference": round(se, 2),
                   "difference_in_reference_sd": round(d / s_ref, 3), "z": round(d / se, 2),
                   "mannwhitney_p": round(float(mannwhitneyu(ours, ref, alternative="two-sided").pvalue), 4), "ks_p": round(float(ks_2samp(ours, ref).pvalue), 4),
                   "tost_p_margin_0.75": float("%.3g" % L.tost_welch(ours, ref, 0.75 * s_ref)[0]), "tost_p_margin_0.5": float("%.3g" % L.tost_welch(ours, ref, 0.5 * s_ref)[0]),
                   "tost_p_margin_0.35": float("%.3g" % L.tost_welch(ours, ref, 0.35 * s_ref)[0]),
                   "ours": [round(x, 2) for x in ours.tolist()]}
    print(script, {k: v for k, v in out[script].items() if k != "ours"}, flush=True)
zs = [out[s]["difference_in_reference_sd"] for s in out]
ses = [out[s]["se_of_difference"] / out[s]["reference_sd"] for s in out]
comb = float(np.mean(zs) / (np.sqrt(np.sum(np.square(ses))) / len(ses)))
out["combined"] = {"mean_difference_in_reference_sd": round(float(np.mean(zs)), 3), "z": round(comb, 2), "scripts": list(scripts)}
print("combined", out["combined"])
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "r06_learning_study.json"), "w"), indent=1)
