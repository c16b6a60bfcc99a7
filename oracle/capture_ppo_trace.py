#!/usr/bin/env python3
"""Generate golden vectors by executing the UNMODIFIED reference ``deep_rl/ppo.py``.

TEST INFRASTRUCTURE ONLY.  Runs in the build container (needs ``/root/reference``); the output
fixture ``tests/golden/ppo_ref_trace.npz`` is committed and is the only thing that travels.

How: ``runpy.run_path('/root/reference/deep_rl/ppo.py')`` with ``oracle/gym_shim`` on ``sys.path``
(real gym 0.21 is not installable here, see oracle/gym_shim/gym/__init__.py).  Instrumentation is
entirely outside the reference source:
  * the shim's raw env mirrors every reset/step to a trace sink           (ppo.py:101,127,129)
  * ``torch.optim.Adam.__init__`` is wrapped to snapshot the initial params (ppo.py:89-90)
  * ``torch.nn.utils.clip_grad_norm_`` is wrapped to record pre-clip grads  (ppo.py:191)
  * ``torch.optim.Adam.step`` is wrapped to record loss terms (read from the script's module
    globals), minibatch indices, lr and post-step params                   (ppo.py:155-192)
  * stdout is captured for the ``global_step=…, episodic_return=…`` lines   (ppo.py:130)

Usage:  python oracle/capture_ppo_trace.py [--out tests/golden/ppo_ref_trace.npz]
"""
import argparse
import contextlib
import io
import os
import runpy
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/deep_rl/ppo.py"
FULL_UPDATES = 3        # rollouts stored in full
FULL_OPT_STEPS = 16     # optimizer steps with full grad / param vectors
FULL_ENV_STEPS = 512    # env steps with float64 state stored


def flat(params, grad=False):
    import torch

    with torch.no_grad():
        return torch.cat([(p.grad if grad else p).detach().reshape(-1) for p in params]).numpy().copy()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(HERE, "..", "tests", "golden", "ppo_ref_trace.npz"))
    args = ap.parse_args()

    sys.path.insert(0, os.path.join(HERE, "gym_shim"))
    import gym
    import torch

    torch.set_num_threads(1)

    env_log = {"reset_states": [], "state": [], "action": [], "obs": [], "terminated": [], "step_is_after_reset": []}
    pending_reset = [False]

    def sink(event, p):
        if event == "reset":
            env_log["reset_states"].append(p["state"])
            pending_reset[0] = True
        else:
            env_log["state"].append(p["state"])
            env_log["action"].append(p["action"])
            env_log["obs"].append(p["obs"])
            env_log["terminated"].append(p["terminated"])
            env_log["step_is_after_reset"].append(pending_reset[0])
            pending_reset[0] = False

    gym.register_trace_sink(sink)

    rec = {
        "init_params": None, "opt": [], "full_grads": [], "full_params": [], "clip_norm": [],
        "mb_inds": [], "updates": [], "update_sums": [],
    }
    state = {"params": None, "preclip": None}

    orig_init = torch.optim.Adam.__init__
    orig_step = torch.optim.Adam.step
    orig_clip = torch.nn.utils.clip_grad_norm_

    def patched_init(self, params, *a, **kw):
        params = list(params)
        state["params"] = params
        rec["init_params"] = flat(params)
        rec["adam_kwargs"] = dict(kw)
        return orig_init(self, params, *a, **kw)

    def patched_clip(parameters, max_norm, *a, **kw):
        parameters = list(parameters)
        state["preclip"] = flat(parameters, grad=True)
        total = orig_clip(parameters, max_norm, *a, **kw)
        rec["clip_norm"].append(float(total))
        return total

    def script_globals():
        f = sys._getframe(2)
        while f is not None:
            if "pg_loss" in f.f_globals and "mb_inds" in f.f_globals:
                return f.f_globals
            f = f.f_back
        raise RuntimeError("reference module frame not found")

    def patched_step(self, *a, **kw):
        g = script_globals()
        k = len(rec["opt"])
        if k % 16 == 0:  # first optimizer step of an outer update: snapshot the rollout + GAE outputs
            names = ["observations", "values", "actions", "log_probs", "rewards", "dones", "advantages", "returns"]
            snap = {n: g[n].detach().numpy().copy() for n in names}
            if len(rec["updates"]) < FULL_UPDATES:
                snap["params_before"] = flat(state["params"])
                rec["updates"].append(snap)
            rec["update_sums"].append([float(snap[n].astype(np.float64).sum()) for n in names])
        out = orig_step(self, *a, **kw)
        pa = flat(state["params"])
        rec["opt"].append([
            float(g["pg_loss"]), float(g["entropy_loss"]), float(g["v_loss"]), float(g["loss"]),
            float(self.param_groups[0]["lr"]), float(pa.astype(np.float64).sum()),
            float(np.abs(pa.astype(np.float64)).sum()),
        ])
        rec["mb_inds"].append(np.asarray(g["mb_inds"]).astype(np.int16))
        if k < FULL_OPT_STEPS:
            rec["full_grads"].append(state["preclip"])
            rec["full_params"].append(pa)
        return out

    torch.optim.Adam.__init__ = patched_init
    torch.optim.Adam.step = patched_step
    torch.nn.utils.clip_grad_norm_ = patched_clip

    buf = io.StringIO()
    t0 = time.time()
    with contextlib.redirect_stdout(buf):
        g = runpy.run_path(REF, run_name="__ref_ppo__")
    wall = time.time() - t0
    torch.optim.Adam.__init__ = orig_init
    torch.optim.Adam.step = orig_step
    torch.nn.utils.clip_grad_norm_ = orig_clip

    lines = [ln for ln in buf.getvalue().splitlines() if ln.startswith("global_step=")]
    ep_step = np.array([int(ln.split(",")[0].split("=")[1]) for ln in lines], dtype=np.int32)
    ep_ret = np.array([float(ln.split("episodic_return=")[1]) for ln in lines], dtype=np.float32)

    n_steps = len(env_log["action"])
    opt = np.array(rec["opt"], dtype=np.float64)
    out = {
        # hyper-parameters as the script defined them (ppo.py:62-76)
        "hparams": np.array([g[k] for k in ("total_timesteps", "num_steps", "num_updates", "minibatch_size",
                                              "update_epochs", "gamma", "gae_lambda", "learning_rate", "clip_coef",
                                              "ent_coef", "vf_coef", "max_grad_norm", "seed")], dtype=np.float64),
        "init_params": rec["init_params"],
        "final_params": flat(state["params"]),
        "reset_states": np.array(env_log["reset_states"], dtype=np.float64),
        "actions_all": np.array(env_log["action"], dtype=np.int8),
        "obs_all": np.array(env_log["obs"], dtype=np.float32),
        "terminated_all": np.array(env_log["terminated"], dtype=np.uint8),
        "after_reset_all": np.array(env_log["step_is_after_reset"], dtype=np.uint8),
        "state_first": np.array(env_log["state"][:FULL_ENV_STEPS], dtype=np.float64),
        "opt_terms": opt,  # [n_opt, 7]: pg_loss, entropy, v_loss, loss, lr, sum(params), sum|params| after the step
        "clip_norm": np.array(rec["clip_norm"], dtype=np.float64),
        "mb_inds": np.stack(rec["mb_inds"]),
        "full_grads": np.stack(rec["full_grads"]),
        "full_params": np.stack(rec["full_params"]),
        "update_sums": np.array(rec["update_sums"], dtype=np.float64),
        "episode_global_step": ep_step,
        "episode_return": ep_ret,
        "final_global_step": np.array([g["global_step"]], dtype=np.int64),
        "final_explained_var": np.array([float(g["explained_var"])], dtype=np.float64),
        "ref_wall_seconds": np.array([wall]),
    }
    for i, snap in enumerate(rec["updates"]):
        for n, v in snap.items():
            out["upd%d_%s" % (i, n)] = v
    os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
    np.savez_compressed(args.out, **out)
    print("reference ppo.py: %d env steps, %d optimizer steps, %d episodes, %.1fs wall -> %s (%.0f KB)" % (
        n_steps, len(opt), len(lines), wall, args.out, os.path.getsize(args.out) / 1024))
    print("last loss terms:", opt[-1, :4], " explained_var:", float(g["explained_var"]))


if __name__ == "__main__":
    main()
