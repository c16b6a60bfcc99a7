#!/usr/bin/env python3
"""BASELINE config 4 measurement (not the headline bench): sac.py Pendulum-v1, 2048 envs on one MI355X, twin-Q + reparameterised
actor kernels.  One iteration = one env step of every env + one critic update (+ 2 actor / alpha updates every 2nd step) + polyak,
i.e. exactly one pass of the reference loop body (sac.py:137-217) with an env axis."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import deep_rl_amd as D

ap = argparse.ArgumentParser()
ap.add_argument("--envs", type=int, default=2048); ap.add_argument("--slots", type=int, default=512)
ap.add_argument("--batch", type=int, default=256); ap.add_argument("--iters", type=int, default=400)
a = ap.parse_args()
dev = torch.device("cuda", 0)
env = D.make("Pendulum-v1", num_envs=a.envs, device=dev, seed=1)
torch.manual_seed(1)
actor = D.Actor(env)
qs = [D.SoftQNetwork(env) for _ in range(4)]
qs[2].load_state_dict(qs[0].state_dict()); qs[3].load_state_dict(qs[1].state_dict())
eng = D.SACEngine(env, actor, *qs, slots=a.slots, batch_size=a.batch, learning_starts=20, max_episodes_logged=0)
eng.reset()
def it():
    eng.act()
    if eng.global_step >= eng.learning_starts: eng.train_step()
for _ in range(60): it()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(a.iters): it()
t_enq = time.perf_counter() - t0   # host time to ENQUEUE the loop: if it is the whole of dt the loop is host-bound, not GPU-bound
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(json.dumps({"workload": "sac.py Pendulum-v1, %d envs, %d-slot ring (%d transitions), batch %d, 1 critic + 1 actor + 1 alpha update per time step" % (a.envs, a.slots, a.envs * a.slots, a.batch),
                  "env_steps_per_s": round(a.iters * a.envs / dt, 1), "updates_per_s": round(a.iters / dt, 1), "us_per_iteration": round(1e6 * dt / a.iters, 1), "host_enqueue_us_per_iteration": round(1e6 * t_enq / a.iters, 1),
                  "q_losses": eng.q_losses.tolist(), "alpha": float(eng.alpha)}))
