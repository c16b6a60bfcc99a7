#!/usr/bin/env python3
"""Runs the five drop-in scripts at the reference's own run shape (one env, default hyper-parameters and step budgets) on the GPU and puts the
episodic returns they print beside the ones the UNMODIFIED reference printed (tests/golden/*_ref_trace.npz, CPU, seed 1).  The random
streams differ by construction (counter-based keys vs the host generators), so this is a behavioural comparison, not a parity test."""
import contextlib, io, json, os, runpy, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

RUNS = [("ppo", "ppo_ref_trace.npz", {}), ("dqn", "dqn_ref_trace.npz", {}), ("dueling_dqn", "dueling_ref_trace.npz", {}), ("per", "per_ref_trace.npz", {}),
        ("sac", "sac_ref_trace.npz", {})]


def summarize(steps, rets):
    steps, rets = np.asarray(steps), np.asarray(rets, np.float64)
    k = max(len(rets) // 10, 1)
    return {"episodes": int(len(rets)), "first_tenth_mean": round(float(rets[:k].mean()), 2), "last_tenth_mean": round(float(rets[-k:].mean()), 2),
            "best_20_episode_mean": round(float(max(rets[i:i + 20].mean() for i in range(max(len(rets) - 19, 1)))), 2), "last_global_step": int(steps[-1])}


out = {}
for name, fixture, env_over in RUNS:
    os.environ.update({"NUM_ENVS": "1", **env_over})
    buf = io.StringIO(); t0 = time.time()
    with contextlib.redirect_stdout(buf):
        runpy.run_module("deep_rl_amd." + name, run_name="__main__")
    wall = time.time() - t0
    lines = [ln for ln in buf.getvalue().splitlines() if ln.startswith("global_step=")]
    steps = [int(ln.split(",")[0].split("=")[1]) for ln in lines]; rets = [float(ln.split("episodic_return=")[1]) for ln in lines]
    g = np.load(os.path.join(ROOT, "tests", "golden", fixture))
    out[name] = {"ours_gpu": dict(summarize(steps, rets), wall_seconds=round(wall, 1)),
                 "reference_cpu": dict(summarize(g["episode_global_step"], g["episode_return"]), wall_seconds=round(float(g["ref_wall_seconds"][0]), 1) if "ref_wall_seconds" in g.files else None)}
    print(name, json.dumps(out[name]), flush=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "learning_curves.json"), "w"), indent=1)
