#!/usr/bin/env python3
"""PPO at the headline shape (4096 envs x 128 steps) trained twice from the same seed, once per contraction setting (f32 / split-bf16): mean episodic return of
the episodes finished in each update's rollout.  Evidence that the experiment behind mi_ppo_set_contraction trains like the default; not a parity test
(trajectories diverge after the first differing rounding, as they would between any two f32 summation orders)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import deep_rl_amd as D

U = int(os.environ.get("UPDATES", "60"))
dev = torch.device("cuda", 0)
out = {}
for mode in ("f32", "bf16x3"):
    D.set_contraction(mode)
    env = D.make("CartPole-v1", num_envs=4096, device=dev, seed=1)
    torch.manual_seed(1)
    agent = D.ActorCritic(env)
    opt = D.ClipAdam(agent, lr=2.5e-4, eps=1e-5, max_grad_norm=0.5)
    eng = D.PPOEngine(env, agent, opt, num_steps=128, n_minibatch=4, update_epochs=4)
    eng.reset()
    curve = []
    for u in range(U):
        opt.param_groups[0]["lr"] = (1.0 - u / U) * 2.5e-4
        eng.update()
        st = eng.episode_stats.tolist()
        curve.append(round(st[1] / max(st[0], 1), 2))
    out[mode] = {"mean_return_per_update": curve, "final_params_finite": bool(torch.isfinite(agent.flat).all().item())}
D.set_contraction("f32")
a, b = out["f32"]["mean_return_per_update"], out["bf16x3"]["mean_return_per_update"]
out["summary"] = {"updates": U, "env_steps": U * 128 * 4096, "f32_last5": a[-5:], "bf16x3_last5": b[-5:],
                  "max_abs_diff_first_10_updates": max(abs(x - y) for x, y in zip(a[:10], b[:10]))}
print(json.dumps(out))
