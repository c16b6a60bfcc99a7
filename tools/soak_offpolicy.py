#!/usr/bin/env python3
"""Soak of the off-policy loops at the BASELINE config 3 / 4 shapes (the 1M-transition rings wrap thousands of times in a real run, the tests wrap
them once or twice): dqn.py (4096 envs, 256-slot ring, batch 128, 10 env steps per update) or sac.py on Pendulum-v1 (2048 envs, 512-slot ring, batch 256, one update per
step) for ENV_STEPS env steps (default 1e9), production RNG, the loops of deep_rl_amd/dqn.py / sac.py.  Prints a progress line every ~5 % and a JSON summary: ring wraps,
updates, finiteness of every parameter / moment tensor, loss, episodic returns at the end.
    python tools/soak_offpolicy.py dqn|dueling|per|sac [env_steps]     (dueling / per: dueling_dqn.py / per.py on the same ring and loop as dqn)"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import deep_rl_amd as D

kind = sys.argv[1]
target = float(sys.argv[2]) if len(sys.argv) > 2 else 1e9
dev = torch.device("cuda", 0)
t0 = time.time()
if kind in ("dqn", "dueling", "per"):
    N_ENVS, SLOTS = 4096, 256
    steps = int(target // N_ENVS) // 10 * 10
    env = D.make("CartPole-v1", num_envs=N_ENVS, device=dev, seed=1); torch.manual_seed(1)
    Net = D.DuelingQNetwork if kind == "dueling" else D.QNetwork
    Eng = {"dqn": D.DQNEngine, "dueling": D.DuelingDQNEngine, "per": D.PERDQNEngine}[kind]
    q = Net(env); t = Net(env); t.load_state_dict(q.state_dict())
    eng = Eng(env, q, t, D.ClipAdam(q, lr=2.5e-4, eps=1e-8), slots=SLOTS, batch_size=128, learning_starts=1000, total_timesteps=steps, max_episodes_logged=0)
    eng.reset()
    gs, mark = 0, max(steps // 20, 10)
    while gs < steps:
        eng.act(10); gs += 10
        if gs >= 1000:
            eng.train_step()
            if gs % 500 == 0: eng.sync_target()
        if gs % mark < 10:
            st = eng.episode_stats.tolist()
            print("time_step=%d env_steps=%.3e ring_wraps=%.1f updates=%d loss=%.4f episodes_in_last_launch=%d mean_return=%.1f longest=%d"
                  % (gs, gs * N_ENVS, gs / SLOTS, eng.update_index, float(eng.loss), st[0], st[1] / max(st[0], 1), st[2]), flush=True)
    torch.cuda.synchronize()
    st = eng.episode_stats.tolist()
    tensors = {"q": eng.q.flat, "target": eng.target.flat, "exp_avg": eng.optimizer.exp_avg, "exp_avg_sq": eng.optimizer.exp_avg_sq, "observations": eng.observations, "rewards": eng.rewards}
    if kind == "per":
        tensors["priorities"] = eng.priorities; tensors["weights"] = eng.weights
    out = {"script": {"dqn": "dqn.py", "dueling": "dueling_dqn.py", "per": "per.py"}[kind], "envs": N_ENVS, "ring_slots": SLOTS, "ring_transitions": N_ENVS * SLOTS, "batch": 128, "time_steps": gs, "env_steps": gs * N_ENVS, "ring_wraps": round(gs / SLOTS, 1),
           "updates": eng.update_index, "final_loss": float(eng.loss), "mean_return_last_launch": round(st[1] / max(st[0], 1), 2), "episodes_last_launch": st[0], "longest_last_launch": st[2]}
else:
    N_ENVS, SLOTS = 2048, 512
    steps = int(target // N_ENVS) // 200 * 200
    env = D.make("Pendulum-v1", num_envs=N_ENVS, device=dev, seed=1); torch.manual_seed(1)
    a = D.Actor(env); qs = [D.SoftQNetwork(env) for _ in range(4)]
    qs[2].load_state_dict(qs[0].state_dict()); qs[3].load_state_dict(qs[1].state_dict())
    eng = D.SACEngine(env, a, *qs, slots=SLOTS, batch_size=256, learning_starts=5000 // 8, max_episodes_logged=4 * N_ENVS)
    eng.reset()
    gs, mark, rets = 0, max(steps // 20 // 200 * 200, 200), []
    while gs < steps:
        eng.act(); gs += 1
        if gs % mark == 0:   # every env's episode ends on a multiple of 200 (TimeLimit): the acting launch has just logged all of them
            ep = eng.drain_episodes()
            rets = [r for _e, r, _l in ep]
            print("time_step=%d env_steps=%.3e ring_wraps=%.1f updates=%d alpha=%.4f episodes=%d mean_return=%.1f min=%.1f max=%.1f"
                  % (gs, gs * N_ENVS, gs / SLOTS, eng.update_index, float(eng.alpha), len(rets), sum(rets) / max(len(rets), 1), min(rets or [0]), max(rets or [0])), flush=True)
        if gs >= 5000 // 8:
            eng.train_step(2, 1)
    eng.flush(); torch.cuda.synchronize()
    tensors = {"actor": eng.actor.flat, "q": eng.q_flat, "q_target": eng.qt_flat, "log_alpha": eng.log_alpha, "actor_exp_avg_sq": eng.actor_optimizer.exp_avg_sq,
               "q_exp_avg_sq": eng.q_optimizer.exp_avg_sq, "observations": eng.observations, "rewards": eng.rewards}
    out = {"script": "sac.py (Pendulum-v1)", "envs": N_ENVS, "ring_slots": SLOTS, "ring_transitions": N_ENVS * SLOTS, "batch": 256, "time_steps": gs, "env_steps": gs * N_ENVS,
           "ring_wraps": round(gs / SLOTS, 1), "updates": eng.update_index, "alpha": float(eng.alpha), "q_losses": [float(x) for x in eng.q_losses.tolist()],
           "mean_return_last_episodes": round(sum(rets) / max(len(rets), 1), 2), "episodes_in_that_mean": len(rets)}
out["finite"] = {k: bool(torch.isfinite(v.float()).all().item()) for k, v in tensors.items()}
out["all_finite"] = all(out["finite"].values())
out["wall_seconds"] = round(time.time() - t0, 1)
print("SOAK_JSON " + json.dumps(out))
