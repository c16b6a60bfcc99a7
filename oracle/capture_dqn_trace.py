#!/usr/bin/env python3
"""Generate golden vectors by executing the UNMODIFIED reference ``deep_rl/dqn.py`` (TEST INFRASTRUCTURE ONLY).

Same method as capture_ppo_trace.py: runpy + oracle/gym_shim + instrumentation outside the reference source:
  * the shim's raw env mirrors every reset / step                                   (dqn.py:78,98-100)
  * ``torch.optim.Adam.__init__`` snapshots the initial q_network parameters        (dqn.py:67-68)
  * ``torch.optim.Adam.step`` records loss, batch indices (module globals), gradients, parameters (dqn.py:116-133)
Output: tests/golden/dqn_ref_trace.npz (``--script dueling_dqn``: tests/golden/dueling_ref_trace.npz from deep_rl/dueling_dqn.py).  The batch indices of all 9,001 updates are NOT stored: they come from numpy's
legacy global generator (np.random.seed(1); rand() once per step after learning_starts; randint(gs, size=128) per update,
dqn.py:63,88,116), which numpy guarantees stable — tests regenerate them with RandomState(1) and check the stored sums.
"""
import argparse, contextlib, io, os, runpy, sys, time
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/deep_rl/dqn.py"
FULL_STEPS = 8
FULL_INDS = 64
CHECKPOINTS = (1000, 2500, 5000, 7500, 9000)  # un-chained single-step pins late in the run
OBS_FIRST = 12000


def flat(params, grad=False):
    import torch
    with torch.no_grad():
        return torch.cat([(p.grad if grad else p).detach().reshape(-1) for p in params]).numpy().copy()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--script", default="dqn", choices=["dqn", "dueling_dqn"],
                    help="dueling_dqn: the same instrumentation on deep_rl/dueling_dqn.py (its optimizer owns q_network1, dueling_dqn.py:71-73)")
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    ref = "/root/reference/deep_rl/%s.py" % args.script
    if args.out is None:
        args.out = os.path.join(HERE, "..", "tests", "golden", "dqn_ref_trace.npz" if args.script == "dqn" else "dueling_ref_trace.npz")
    sys.path.insert(0, os.path.join(HERE, "gym_shim"))
    import gym, torch
    torch.set_num_threads(1)
    log = {"reset_states": [], "action": [], "obs": [], "terminated": [], "after_reset": []}
    pend = [False]

    def sink(event, p):
        if event == "reset":
            log["reset_states"].append(p["state"]); pend[0] = True
        else:
            log["action"].append(p["action"]); log["obs"].append(p["obs"]); log["terminated"].append(p["terminated"])
            log["after_reset"].append(pend[0]); pend[0] = False

    gym.register_trace_sink(sink)
    rec = {"init": None, "loss": [], "psum": [], "inds_sum": [], "inds": [], "grads": [], "params": [], "gs": []}
    st = {"params": None}
    orig_init, orig_step = torch.optim.Adam.__init__, torch.optim.Adam.step

    def p_init(self, params, *a, **kw):
        params = list(params)
        st["params"] = params
        rec["init"] = flat(params)
        return orig_init(self, params, *a, **kw)

    def p_step(self, *a, **kw):
        f = sys._getframe(1)
        while f is not None and not ("batch_inds" in f.f_globals and "td_target" in f.f_globals):
            f = f.f_back
        g = f.f_globals
        k = len(rec["loss"])
        if k < FULL_STEPS:
            rec["grads"].append(flat(st["params"], grad=True))
        if k in CHECKPOINTS:
            rec.setdefault("ck", []).append((k, flat(st["params"]), flat(list(g["target_network"].parameters())), flat(st["params"], grad=True),
                                             np.asarray(g["batch_inds"]).astype(np.int32), float(g["loss"])))
        out = orig_step(self, *a, **kw)
        pa = flat(st["params"])
        rec["loss"].append(float(g["loss"])); rec["psum"].append(float(pa.astype(np.float64).sum()))
        rec["inds_sum"].append(int(np.asarray(g["batch_inds"]).sum())); rec["gs"].append(int(g["global_step"]))
        if k < FULL_INDS:
            rec["inds"].append(np.asarray(g["batch_inds"]).astype(np.int32))
        if k < FULL_STEPS:
            rec["params"].append(pa)
        return out

    torch.optim.Adam.__init__, torch.optim.Adam.step = p_init, p_step
    buf = io.StringIO(); t0 = time.time()
    with contextlib.redirect_stdout(buf):
        g = runpy.run_path(ref, run_name="__ref_dqn__")
    wall = time.time() - t0
    torch.optim.Adam.__init__, torch.optim.Adam.step = orig_init, orig_step
    lines = [ln for ln in buf.getvalue().splitlines() if ln.startswith("global_step=")]
    obs = np.array(log["obs"], dtype=np.float32)
    out = {
        "hparams": np.array([g[k] for k in ("total_timesteps", "learning_starts", "start_e", "end_e", "exploration_fraction",
                                              "train_frequency", "batch_size", "gamma", "learning_rate", "target_network_frequency", "seed")], dtype=np.float64),
        "init_params": rec["init"], "final_params": flat(st["params"]),
        "final_target_params": flat(list(g["target_network"].parameters())),
        "reset_states": np.array(log["reset_states"], dtype=np.float64),
        "actions_all": np.array(log["action"], dtype=np.int8),
        "terminated_all": np.array(log["terminated"], dtype=np.uint8),
        "after_reset_all": np.array(log["after_reset"], dtype=np.uint8),
        "obs_first": obs[:OBS_FIRST],
        "obs_block_sums": obs.astype(np.float64).reshape(-1, 1000, 4).sum(axis=1),
        "loss_all": np.array(rec["loss"], dtype=np.float64), "psum_all": np.array(rec["psum"], dtype=np.float64),
        "inds_sum_all": np.array(rec["inds_sum"], dtype=np.int64), "train_global_step": np.array(rec["gs"], dtype=np.int32),
        "batch_inds_first": np.stack(rec["inds"]), "full_grads": np.stack(rec["grads"]), "full_params": np.stack(rec["params"]),
        "episode_global_step": np.array([int(ln.split(",")[0].split("=")[1]) for ln in lines], dtype=np.int32),
        "episode_return": np.array([float(ln.split("episodic_return=")[1]) for ln in lines], dtype=np.float32),
        "storage_terminated_sum": np.array([int(g["terminated"].sum())]), "storage_rewards_sum": np.array([float(g["rewards"].sum())]),
        "ref_wall_seconds": np.array([wall]),
        "ck_update": np.array([c[0] for c in rec["ck"]], dtype=np.int32), "ck_params": np.stack([c[1] for c in rec["ck"]]),
        "ck_target": np.stack([c[2] for c in rec["ck"]]), "ck_grads": np.stack([c[3] for c in rec["ck"]]),
        "ck_inds": np.stack([c[4] for c in rec["ck"]]), "ck_loss": np.array([c[5] for c in rec["ck"]]),
    }
    np.savez_compressed(args.out, **out)
    print("reference " + args.script + ".py: %d env steps, %d updates, %d episodes, %.1fs -> %s (%.0f KB); last loss %.5f" % (
        len(log["action"]), len(rec["loss"]), len(lines), wall, args.out, os.path.getsize(args.out) / 1024, rec["loss"][-1]))


if __name__ == "__main__":
    main()
