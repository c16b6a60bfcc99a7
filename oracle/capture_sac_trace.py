#!/usr/bin/env python3
"""Generate golden vectors by executing the UNMODIFIED reference ``deep_rl/sac.py`` (TEST INFRASTRUCTURE ONLY).

The script's env (HopperBulletEnv-v0 via pybullet_envs, sac.py:5,81) cannot exist here; BASELINE config 4 / SURVEY §8a s8
re-target the SAC path to Pendulum-v1, so the gym shim ALIASES the id to its Pendulum and an empty ``pybullet_envs`` stub
satisfies the import — the script text itself runs untouched.  Instrumentation (all outside the reference source):
  * env trace sink (reset / step)                                                      (sac.py:134,142-144)
  * ``Normal.rsample`` wrapped to draw its standard-normal noise through torch.randn and RECORD it (the reference draws it
    from torch's global generator inside rsample, sac.py:71); same formula loc + eps * scale
  * the three ``optim.Adam`` instances are told apart by construction order (actor, q, alpha: sac.py:108,117,122); their
    ``step`` records loss terms from the module globals, gradients and parameters
Output: tests/golden/sac_ref_trace.npz
"""
import argparse, contextlib, io, os, runpy, sys, time
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/deep_rl/sac.py"
CHAIN_UPDATES = 120            # global steps (after learning_starts) with every noise draw / loss stored
CHECKPOINTS = (20000,)           # one un-chained late pin (even step: has an actor update too)
N_PROJ = 32                      # gradients are stored as N_PROJ fixed random projections + their norm, not in full


def flat(params, grad=False):
    import torch
    with torch.no_grad():
        return torch.cat([(p.grad if grad else p).detach().reshape(-1) for p in params]).numpy().copy()


def proj_matrix(n):
    """Fixed pseudo-random projection vectors (regenerated identically by the tests)."""
    return np.random.default_rng(20260101 + n).standard_normal((N_PROJ, n)).astype(np.float32)


def summarize(gvec):
    gvec = np.asarray(gvec, np.float32)
    return np.concatenate([proj_matrix(gvec.size).astype(np.float64) @ gvec.astype(np.float64), [np.linalg.norm(gvec.astype(np.float64))]])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(HERE, "..", "tests", "golden", "sac_ref_trace.npz"))
    args = ap.parse_args()
    sys.path.insert(0, os.path.join(HERE, "gym_shim"))
    import gym, torch
    from torch.distributions import Normal
    gym.alias("HopperBulletEnv-v0", "Pendulum-v1")
    torch.set_num_threads(8)
    log = {"reset_states": [], "action": [], "obs": [], "reward": [], "after_reset": []}
    pend = [False]

    def sink(event, p):
        if event == "reset":
            log["reset_states"].append(p["state"]); pend[0] = True
        else:
            log["action"].append(p["action"]); log["obs"].append(p["obs"]); log["reward"].append(p["reward"])
            log["after_reset"].append(pend[0]); pend[0] = False

    gym.register_trace_sink(sink)
    noise = []          # every rsample draw, in call order: (global_step, tag, eps array)

    def rsample(self, sample_shape=torch.Size()):
        shape = self._extended_shape(sample_shape)
        eps = torch.randn(shape, dtype=self.loc.dtype, device=self.loc.device)
        noise.append(eps.detach().numpy().copy().reshape(-1))
        return self.loc + eps * self.scale

    orig_rsample = Normal.rsample
    Normal.rsample = rsample
    opts = []
    rec = {"q": [], "actor": [], "alpha": [], "ck": {}, "init": {}}
    orig_init, orig_step = torch.optim.Adam.__init__, torch.optim.Adam.step

    def p_init(self, params, *a, **kw):
        params = list(params)
        self._mi_params = params
        self._mi_tag = ("actor", "q", "alpha")[len(opts)]
        opts.append(self)
        rec["init"][self._mi_tag] = flat(params)
        return orig_init(self, params, *a, **kw)

    def gl():
        f = sys._getframe(2)
        while f is not None and "qf_loss" not in f.f_globals:
            f = f.f_back
        return f.f_globals

    def p_step(self, *a, **kw):
        g = gl()
        tag, gs = self._mi_tag, int(g["global_step"])
        k = gs - 5000       # update index since learning_starts (0-based)
        full = k < CHAIN_UPDATES or k in CHECKPOINTS
        if tag == "q":
            e = {"gs": gs, "qf1_loss": float(g["qf1_loss"]), "qf2_loss": float(g["qf2_loss"]), "alpha": float(g["alpha"]), "noise_mark": len(noise)}
            if full:
                e["inds"] = np.asarray(g["batch_inds"]).astype(np.int32)
                e["grads"] = flat(self._mi_params, grad=True)
                e["params_before"] = flat(self._mi_params)
                e["target_before"] = np.concatenate([flat(list(g["qf1_target"].parameters())), flat(list(g["qf2_target"].parameters()))])
                e["actor_params"] = flat(opts[0]._mi_params)
        elif tag == "actor":
            e = {"gs": gs, "actor_loss": float(g["actor_loss"]), "alpha": float(g["alpha"]), "noise_mark": len(noise)}
            if full:
                e["grads"] = flat(self._mi_params, grad=True); e["params_before"] = flat(self._mi_params)
                e["q_params"] = flat(opts[1]._mi_params)
        else:
            e = {"gs": gs, "alpha_loss": float(g["alpha_loss"]), "log_alpha_before": float(g["log_alpha"]), "noise_mark": len(noise)}
            e["grad"] = float(g["log_alpha"].grad)
        out = orig_step(self, *a, **kw)
        e["psum_after"] = float(flat(self._mi_params).astype(np.float64).sum())
        rec[tag].append(e)
        return out

    torch.optim.Adam.__init__, torch.optim.Adam.step = p_init, p_step
    buf = io.StringIO(); t0 = time.time()
    with contextlib.redirect_stdout(buf):
        g = runpy.run_path(REF, run_name="__ref_sac__")
    wall = time.time() - t0
    torch.optim.Adam.__init__, torch.optim.Adam.step = orig_init, orig_step
    Normal.rsample = orig_rsample
    lines = [ln for ln in buf.getvalue().splitlines() if ln.startswith("global_step=")]

    # keep the noise of the chained window and of the checkpoints only
    def noise_between(a, b):
        return np.concatenate(noise[a:b]) if b > a else np.zeros(0, np.float32)

    out = {
        "hparams": np.array([g[k] for k in ("total_timesteps", "learning_starts", "policy_frequency", "batch_size", "target_network_frequency",
                                              "gamma", "tau", "policy_lr", "q_lr", "alpha_lr", "seed", "target_entropy")], dtype=np.float64),
        "init_actor": rec["init"]["actor"], "init_q": rec["init"]["q"], "init_log_alpha": rec["init"]["alpha"],
        "reset_states": np.array(log["reset_states"], dtype=np.float64),
        "actions_all": np.array(log["action"], dtype=np.float32), "rewards_all": np.array(log["reward"], dtype=np.float64),
        "after_reset_all": np.array(log["after_reset"], dtype=np.uint8), "obs_first": np.array(log["obs"][:8000], dtype=np.float32),
        "q_losses": np.array([[e["qf1_loss"], e["qf2_loss"], e["alpha"], e["psum_after"]] for e in rec["q"]]),
        "actor_losses": np.array([[e["gs"], e["actor_loss"], e["alpha"], e["psum_after"]] for e in rec["actor"]]),
        "alpha_steps": np.array([[e["gs"], e["alpha_loss"], e["log_alpha_before"], e["grad"], e["psum_after"]] for e in rec["alpha"]]),
        "episode_global_step": np.array([int(ln.split(",")[0].split("=")[1]) for ln in lines], dtype=np.int32),
        "episode_return": np.array([float(ln.split("episodic_return=")[1]) for ln in lines], dtype=np.float32),
        "final_actor": flat(opts[0]._mi_params), "final_q": flat(opts[1]._mi_params), "final_log_alpha": flat(opts[2]._mi_params),
        "ref_wall_seconds": np.array([wall]),
    }
    # ---- chained window: every noise draw of the first CHAIN_UPDATES global steps after learning_starts, in call order
    chain_end_mark = rec["q"][CHAIN_UPDATES]["noise_mark"] - 2 if len(rec["q"]) > CHAIN_UPDATES else len(noise)
    win = noise[:chain_end_mark]
    out["noise_chain"] = np.concatenate(win)
    out["noise_chain_lens"] = np.array([len(x) for x in win], dtype=np.int32)
    q_full = [e for e in rec["q"] if "inds" in e]
    a_full = [e for e in rec["actor"] if "grads" in e]
    chain_q = [e for e in q_full if e["gs"] - 5000 < CHAIN_UPDATES]
    chain_a = [e for e in a_full if e["gs"] - 5000 < CHAIN_UPDATES]
    out["chain_inds"] = np.stack([e["inds"] for e in chain_q])
    out["chain_q_gradsum"] = np.stack([summarize(e["grads"]) for e in chain_q])       # [CHAIN, N_PROJ + 1]
    out["chain_actor_gs"] = np.array([e["gs"] for e in chain_a], dtype=np.int32)
    out["chain_actor_gradsum"] = np.stack([summarize(e["grads"]) for e in chain_a])
    # ---- un-chained checkpoint(s): full INPUTS (params), summarized gradients
    ck_q = [e for e in q_full if e["gs"] - 5000 in CHECKPOINTS]
    out["ck_gs"] = np.array([e["gs"] for e in ck_q], dtype=np.int32)
    out["ck_q_params"] = np.stack([e["params_before"] for e in ck_q]); out["ck_q_target"] = np.stack([e["target_before"] for e in ck_q])
    out["ck_actor_params"] = np.stack([e["actor_params"] for e in ck_q]); out["ck_inds"] = np.stack([e["inds"] for e in ck_q])
    out["ck_q_gradsum"] = np.stack([summarize(e["grads"]) for e in ck_q]); out["ck_alpha"] = np.array([e["alpha"] for e in ck_q])
    out["ck_q_losses"] = np.array([[e["qf1_loss"], e["qf2_loss"]] for e in ck_q])
    out["ck_noise_critic"] = np.stack([noise[e["noise_mark"] - 1] for e in ck_q])
    ck_a = [[x for x in a_full if x["gs"] == e["gs"]][0] for e in ck_q]
    out["ck_actor_gradsum"] = np.stack([summarize(x["grads"]) for x in ck_a])
    out["ck_actor_qparams"] = np.stack([x["q_params"] for x in ck_a])
    out["ck_actor_loss"] = np.array([x["actor_loss"] for x in ck_a]); out["ck_actor_alpha"] = np.array([x["alpha"] for x in ck_a])
    out["ck_noise_actor"] = np.stack([noise[x["noise_mark"] - 1] for x in ck_a])
    os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
    np.savez_compressed(args.out, **out)
    print("reference sac.py on Pendulum-v1: %d env steps, %d critic / %d actor / %d alpha updates, %d episodes, %.1fs -> %s (%.0f KB)" % (
        len(log["action"]), len(rec["q"]), len(rec["actor"]), len(rec["alpha"]), len(lines), wall, args.out, os.path.getsize(args.out) / 1024))
    print("last: qf1 %.4f qf2 %.4f actor %.4f alpha %.4f; last returns %s" % (rec["q"][-1]["qf1_loss"], rec["q"][-1]["qf2_loss"], rec["actor"][-1]["actor_loss"],
                                                                              rec["q"][-1]["alpha"], out["episode_return"][-5:]))


if __name__ == "__main__":
    main()
