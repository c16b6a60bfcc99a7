#!/usr/bin/env python3
"""BASELINE config 3 measurement (not the headline bench): dqn.py CartPole-v1 on one MI355X with a 1M-transition on-HBM replay
ring (256 slots x 4096 envs) and a batched Q-target.  One iteration = 10 env steps of every env (one launch) + one TD update."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import deep_rl_amd as D

ap = argparse.ArgumentParser()
ap.add_argument("--envs", type=int, default=4096); ap.add_argument("--slots", type=int, default=256)
ap.add_argument("--batch", type=int, default=128); ap.add_argument("--iters", type=int, default=300)
ap.add_argument("--variant", default="dqn", choices=["dqn", "dueling", "per"], help="dueling_dqn.py / per.py epilogues on the same kernels")
a = ap.parse_args()
dev = torch.device("cuda", 0)
env = D.make("CartPole-v1", num_envs=a.envs, device=dev, seed=1)
torch.manual_seed(1)
Net = D.DuelingQNetwork if a.variant == "dueling" else D.QNetwork
Eng = {"dqn": D.DQNEngine, "dueling": D.DuelingDQNEngine, "per": D.PERDQNEngine}[a.variant]
q = Net(env); t = Net(env); t.load_state_dict(q.state_dict())
opt = D.ClipAdam(q, lr=2.5e-4, eps=1e-8)
eng = Eng(env, q, t, opt, slots=a.slots, batch_size=a.batch, learning_starts=100, total_timesteps=10 * (a.iters + 60))
eng.reset()
def it():
    eng.act(10); eng.train_step()
    if eng.global_step % 500 == 0: eng.sync_target()
for _ in range(50): it()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(a.iters): it()
t_enq = time.perf_counter() - t0   # host time to ENQUEUE the loop: if it is the whole of dt the loop is host-bound, not GPU-bound
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(json.dumps({"variant": a.variant, "workload": "dqn.py CartPole-v1, %d envs, %d-slot ring (%d transitions), batch %d, train every 10 steps" % (a.envs, a.slots, a.envs * a.slots, a.batch),
                  "env_steps_per_s": round(a.iters * 10 * a.envs / dt, 1), "updates_per_s": round(a.iters / dt, 1), "us_per_iteration": round(1e6 * dt / a.iters, 1), "host_enqueue_us_per_iteration": round(1e6 * t_enq / a.iters, 1),
                  "loss": float(eng.loss.item())}))
