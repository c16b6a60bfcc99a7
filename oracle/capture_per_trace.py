#!/usr/bin/env python3
"""Generate golden vectors by executing the UNMODIFIED reference ``deep_rl/per.py`` (TEST INFRASTRUCTURE ONLY).

The reference's env (LunarLander-v2, per.py:39) needs Box2D, absent here; the gym shim aliases the id to CartPole-v1, so the script
itself runs unmodified with a 4 -> 120 -> 84 -> 2 QNetwork (its constructor reads the env's shapes).  Instrumentation as in
capture_dqn_trace.py, plus ``torch.multinomial`` is wrapped to snapshot the PRE-update priorities at the checkpoints (per.py:128).
Output: tests/golden/per_ref_trace.npz.
"""
import argparse, contextlib, io, os, runpy, sys, time
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/deep_rl/per.py"
CHAIN = 400            # updates with full batch indices (chained pin)
FULL_STEPS = 8         # updates with full gradients / parameters
CHECKPOINTS = (1000, 5000, 9000)


def flat(params, grad=False):
    import torch
    with torch.no_grad():
        return torch.cat([(p.grad if grad else p).detach().reshape(-1) for p in params]).numpy().copy()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(HERE, "..", "tests", "golden", "per_ref_trace.npz"))
    args = ap.parse_args()
    sys.path.insert(0, os.path.join(HERE, "gym_shim"))
    import gym, torch
    torch.set_num_threads(1)
    gym.alias("LunarLander-v2", "CartPole-v1")
    log = {"reset_states": [], "action": [], "terminated": [], "after_reset": []}
    pend = [False]

    def sink(event, p):
        if event == "reset":
            log["reset_states"].append(p["state"]); pend[0] = True
        else:
            log["action"].append(p["action"]); log["terminated"].append(p["terminated"]); log["after_reset"].append(pend[0]); pend[0] = False

    gym.register_trace_sink(sink)
    rec = {"init": None, "loss": [], "psum": [], "prio_sum": [], "max_prio": [], "inds": [], "grads": [], "params": [], "gs": [], "wsum": [], "ck": [], "ck_pre": {}}
    st = {"params": None, "n_multi": 0}
    orig_init, orig_step, orig_multi = torch.optim.Adam.__init__, torch.optim.Adam.step, torch.multinomial

    def p_init(self, params, *a, **kw):
        params = list(params)
        st["params"] = params
        rec["init"] = flat(params)
        return orig_init(self, params, *a, **kw)

    def p_multi(inp, *a, **kw):
        k = st["n_multi"]; st["n_multi"] += 1
        if k in CHECKPOINTS:
            rec["ck_pre"][k] = inp.detach().numpy().copy()
        return orig_multi(inp, *a, **kw)

    def p_step(self, *a, **kw):
        f = sys._getframe(1)
        while f is not None and not ("batch_inds" in f.f_globals and "td_errors" in f.f_globals):
            f = f.f_back
        g = f.f_globals
        k = len(rec["loss"])
        gs = int(g["global_step"])
        if k < FULL_STEPS:
            rec["grads"].append(flat(st["params"], grad=True))
        if k in CHECKPOINTS:
            rec["ck"].append(dict(k=k, gs=gs, params=flat(st["params"]), target=flat(list(g["target_network"].parameters())), grads=flat(st["params"], grad=True),
                                  inds=g["batch_inds"].numpy().astype(np.int32).copy(), loss=float(g["loss"]), weights=g["weights"].detach().numpy().copy(),
                                  td=g["td_errors"].detach().numpy().copy(), bprob=g["b_probabilities"].detach().numpy().copy(),
                                  pre=rec["ck_pre"][k][:gs + 1].copy(), max_prio=float(g["max_priority"])))
        out = orig_step(self, *a, **kw)
        pa = flat(st["params"])
        rec["loss"].append(float(g["loss"])); rec["psum"].append(float(pa.astype(np.float64).sum())); rec["gs"].append(gs)
        rec["prio_sum"].append(float(g["priorities"].double().sum())); rec["max_prio"].append(float(g["max_priority"]))
        rec["wsum"].append(float(g["weights"].double().sum()))
        if k < CHAIN:
            rec["inds"].append(g["batch_inds"].numpy().astype(np.int32).copy())
        if k < FULL_STEPS:
            rec["params"].append(pa)
        return out

    torch.optim.Adam.__init__, torch.optim.Adam.step, torch.multinomial = p_init, p_step, p_multi
    buf = io.StringIO(); t0 = time.time()
    with contextlib.redirect_stdout(buf):
        g = runpy.run_path(REF, run_name="__ref_per__")
    wall = time.time() - t0
    torch.optim.Adam.__init__, torch.optim.Adam.step, torch.multinomial = orig_init, orig_step, orig_multi
    lines = [ln for ln in buf.getvalue().splitlines() if ln.startswith("global_step=")]
    ck = rec["ck"]
    out = {
        "hparams": np.array([g[k] for k in ("total_timesteps", "learning_starts", "start_e", "end_e", "exploration_fraction", "alpha", "beta_0",
                                              "train_frequency", "batch_size", "gamma", "learning_rate", "target_network_frequency", "seed")], dtype=np.float64),
        "init_params": rec["init"], "final_params": flat(st["params"]),
        "reset_states": np.array(log["reset_states"], dtype=np.float64),
        "actions_all": np.array(log["action"], dtype=np.int8), "terminated_all": np.array(log["terminated"], dtype=np.uint8),
        "after_reset_all": np.array(log["after_reset"], dtype=np.uint8),
        "loss_all": np.array(rec["loss"]), "psum_all": np.array(rec["psum"]), "prio_sum_all": np.array(rec["prio_sum"]), "max_prio_all": np.array(rec["max_prio"]),
        "wsum_all": np.array(rec["wsum"]), "train_global_step": np.array(rec["gs"], dtype=np.int32),
        "batch_inds_chain": np.stack(rec["inds"]), "full_grads": np.stack(rec["grads"]), "full_params": np.stack(rec["params"]),
        "episode_global_step": np.array([int(ln.split(",")[0].split("=")[1]) for ln in lines], dtype=np.int32),
        "episode_return": np.array([float(ln.split("episodic_return=")[1]) for ln in lines], dtype=np.float32),
        "final_priorities_sum": np.array([float(g["priorities"].double().sum())]),
        "ck_update": np.array([c["k"] for c in ck], dtype=np.int32), "ck_gs": np.array([c["gs"] for c in ck], dtype=np.int32),
        "ck_params": np.stack([c["params"] for c in ck]), "ck_target": np.stack([c["target"] for c in ck]), "ck_grads": np.stack([c["grads"] for c in ck]),
        "ck_inds": np.stack([c["inds"] for c in ck]), "ck_loss": np.array([c["loss"] for c in ck]), "ck_weights": np.stack([c["weights"] for c in ck]),
        "ck_td": np.stack([c["td"] for c in ck]), "ck_bprob": np.stack([c["bprob"] for c in ck]), "ck_max_prio": np.array([c["max_prio"] for c in ck]),
        "ref_wall_seconds": np.array([wall]),
    }
    for c in ck:
        out["ck_pre_%d" % c["k"]] = c["pre"].astype(np.float32)
    np.savez_compressed(args.out, **out)
    print("reference per.py (on CartPole-v1): %d env steps, %d updates, %d episodes, %.1fs -> %s (%.0f KB); last loss %.5f" % (
        len(log["action"]), len(rec["loss"]), len(lines), wall, args.out, os.path.getsize(args.out) / 1024, rec["loss"][-1]))


if __name__ == "__main__":
    main()
