#!/usr/bin/env python3
"""Episodic-return statistics of the UNMODIFIED reference scripts over several seeds (TEST INFRASTRUCTURE ONLY).

Runs in the build container (needs ``/root/reference``); the output ``tests/golden/learning_stats.npz`` holds numbers only
and is the only thing that travels.  Each run is ``runpy.run_path('/root/reference/deep_rl/<script>.py')`` under
``oracle/gym_shim`` in its own process.  The scripts hard-code ``seed = 1`` (ppo.py:83, dqn.py:60, per.py:63,
dueling_dqn.py:64, sac.py:100); to get seed s the four seeding entry points they call are wrapped FROM OUTSIDE to add
``s - 1`` to a non-None argument:  ``env.seed`` (the shim's raw env), ``np.random.seed``, ``torch.manual_seed`` and
``env.action_space.seed`` (ppo.py:84-86, dqn.py:61-64).  With s = 1 the run is the one the trace fixtures hold (checked below
against ``tests/golden/*_ref_trace.npz`` when present).

Stored per (script, seed): every ``global_step=…, episodic_return=…`` line (ppo.py:130, dqn.py:110-111) as two ragged arrays
(offsets + concatenated values) and the mean return of the last tenth of the episodes — the statistic
``tests/test_gpu_learning.py`` compares the drop-in scripts with.

Usage:  python oracle/capture_learning_stats.py [--scripts ppo,dqn,dueling_dqn,per] [--seeds 10] [--jobs 4] [--sac-seeds 0] [--first-seed 1]
Seeds first-seed..seeds are run; seeds the output file already holds for a script are kept (round 6 extended the sample from 10 to 50 / 30 seeds this way, and a
re-run of seeds 1..10 reproduced the round-3 numbers exactly: the runs are deterministic).
"""
import argparse, contextlib, io, multiprocessing as mp, os, runpy, sys, time
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "..", "tests", "golden")
FIX = {"ppo": "ppo_ref_trace.npz", "dqn": "dqn_ref_trace.npz", "dueling_dqn": "dueling_ref_trace.npz", "per": "per_ref_trace.npz", "sac": "sac_ref_trace.npz"}


def run_one(job):
    script, seed = job
    sys.path.insert(0, os.path.join(HERE, "gym_shim"))
    import gym, gym.envs, gym.spaces, torch
    torch.set_num_threads(1)
    off = seed - 1

    def shifted(fn):
        def w(*a, **kw):
            if a and a[-1] is not None and isinstance(a[-1], int):
                a = a[:-1] + (a[-1] + off,)
            elif kw.get("seed") is not None:
                kw["seed"] = kw["seed"] + off
            return fn(*a, **kw)
        return w

    np.random.seed = shifted(np.random.seed)
    torch.manual_seed = shifted(torch.manual_seed)
    gym.spaces.Space.seed = shifted(gym.spaces.Space.seed)
    for cls in (gym.envs.CartPoleEnv, gym.envs.PendulumEnv):
        cls.seed = shifted(cls.seed)
    if script == "sac":  # sac.py:96 names a Bullet task; config 4 re-targets it to Pendulum-v1, as capture_sac_trace.py does
        gym.alias("HopperBulletEnv-v0", "Pendulum-v1")
    if script == "per":  # per.py:39 names LunarLander-v2 (needs Box2D); run on CartPole-v1 as capture_per_trace.py does
        gym.alias("LunarLander-v2", "CartPole-v1")
    buf = io.StringIO(); t0 = time.time()
    with contextlib.redirect_stdout(buf):
        runpy.run_path("/root/reference/deep_rl/%s.py" % script, run_name="__ref_%s__" % script)
    lines = [ln for ln in buf.getvalue().splitlines() if ln.startswith("global_step=")]
    steps = np.array([int(ln.split(",")[0].split("=")[1]) for ln in lines], np.int64)
    rets = np.array([float(ln.split("episodic_return=")[1]) for ln in lines], np.float64)
    return script, seed, steps, rets, time.time() - t0


def last_tenth(rets):
    k = max(len(rets) // 10, 1)
    return float(np.mean(rets[-k:]))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scripts", default="ppo,dqn,dueling_dqn,per")
    ap.add_argument("--seeds", type=int, default=10)
    ap.add_argument("--sac-seeds", type=int, default=0, help="sac.py takes ~8 CPU-minutes per seed")
    ap.add_argument("--jobs", type=int, default=4)
    ap.add_argument("--first-seed", type=int, default=1)
    ap.add_argument("--out", default=os.path.join(GOLD, "learning_stats.npz"))
    args = ap.parse_args()
    jobs = [(s, k) for s in args.scripts.split(",") if s for k in range(args.first_seed, args.seeds + 1)]
    jobs += [("sac", k) for k in range(args.first_seed, args.sac_seeds + 1)]
    out = dict(np.load(args.out)) if os.path.exists(args.out) else {}
    with mp.get_context("spawn").Pool(args.jobs, maxtasksperchild=1) as pool:
        res = {}
        for script, seed, steps, rets, wall in pool.imap_unordered(run_one, jobs):
            res.setdefault(script, {})[seed] = (steps, rets)
            print("%-12s seed %2d: %4d episodes, last-tenth mean %8.2f  (%.0f s)" % (script, seed, len(rets), last_tenth(rets), wall), flush=True)
    for script, by_seed in res.items():
        if script + "_seeds" in out:                                             # keep the seeds the file already holds
            off = out[script + "_offsets"]
            for i, s0 in enumerate(out[script + "_seeds"].tolist()):
                by_seed.setdefault(s0, (out[script + "_episode_global_step"][off[i]:off[i + 1]].astype(np.int64), out[script + "_episode_return"][off[i]:off[i + 1]].astype(np.float64)))
        seeds = sorted(by_seed)
        if 1 in by_seed and os.path.exists(os.path.join(GOLD, FIX[script])):  # seed 1 must be the run the trace fixture holds
            g = np.load(os.path.join(GOLD, FIX[script]))
            assert np.array_equal(g["episode_global_step"], by_seed[1][0]) and np.allclose(g["episode_return"], by_seed[1][1]), script
        out[script + "_seeds"] = np.array(seeds, np.int32)
        out[script + "_offsets"] = np.cumsum([0] + [len(by_seed[s][1]) for s in seeds]).astype(np.int64)
        out[script + "_episode_global_step"] = np.concatenate([by_seed[s][0] for s in seeds]).astype(np.int32)
        out[script + "_episode_return"] = np.concatenate([by_seed[s][1] for s in seeds]).astype(np.float32)
        out[script + "_last_tenth_mean"] = np.array([last_tenth(by_seed[s][1]) for s in seeds], np.float64)
        print(script, "last-tenth means:", np.round(out[script + "_last_tenth_mean"], 1))
    np.savez_compressed(args.out, **out)
    print("->", args.out, "%.0f KB" % (os.path.getsize(args.out) / 1024))


if __name__ == "__main__":
    main()
