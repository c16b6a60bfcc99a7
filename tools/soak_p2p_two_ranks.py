#!/usr/bin/env python3
"""Soak of the ONE-CALL sharded PPO route on the P2P carrier with two ranks on ONE GPU (both processes on cuda:0; process group gloo, which only carries the IPC handles
and the replica checks): UPDATES outer updates of ENVS envs per rank = 17 exchanges each, the replica-divergence guard every CHECK updates (bitwise equality of parameters
and Adam moments across the ranks, and no timed-out wait on either).  Evidence for the slot-reuse argument of csrc/mi_comm.hip (two parities, sequence numbers) under
tens of thousands of back-to-back exchanges between two processes that time-share the device.
    MIRL_COMM=p2p python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29531 tools/soak_p2p_two_ranks.py [updates] [envs] [check] [epoch_every]
epoch_every > 0 (round 6): every that many updates both ranks preset the carrier's sequence number 40 below the end of its epoch (mi_comm_test_set_seq), so that the
next three updates' exchanges run across an EPOCH CHANGE (own lines cleared, barrier through the header, numbers restart at 1) in the middle of real training."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import deep_rl_amd as D
import deep_rl_amd.dist as DD

updates = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
envs = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
check = int(sys.argv[3]) if len(sys.argv) > 3 else 50
epoch_every = int(sys.argv[4]) if len(sys.argv) > 4 else 0
assert os.environ.get("MIRL_COMM") == "p2p"
rank, world, _ = DD.init_from_env("gloo")
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
env = D.make("CartPole-v1", num_envs=envs, device=dev, seed=1, env_id_base=rank * envs)
torch.manual_seed(1)
agent = D.ActorCritic(env)
opt = D.ClipAdam(agent, lr=2.5e-4, eps=1e-5, max_grad_norm=0.5)
eng = D.PPOEngine(env, agent, opt, num_steps=128)
eng.reset()
assert eng.world_size == 2 and DD.native_comm(eng.pg) is not None
t0 = time.time()
from deep_rl_amd import _native as N
epochs = 0
for u in range(updates):
    opt.param_groups[0]["lr"] = (1.0 - u / updates) * 2.5e-4
    if epoch_every and u % epoch_every == epoch_every - 1:
        N.check(N.lib().mi_comm_test_set_seq(DD.native_comm(eng.pg), 0xFFFFFFF0 - 40), "mi_comm_test_set_seq")   # (host-side counter: both ranks, same update)
        epochs += 1
    eng.update()
    if (u + 1) % check == 0:
        eng.check_replicas()          # raises on every rank on divergence or on a timed-out wait
torch.cuda.synchronize()
eng.check_replicas()
st = eng.episode_stats.tolist()
if rank == 0:
    print("SOAK_P2P_JSON " + json.dumps({"ranks": 2, "placement": "both on cuda:0", "envs_per_rank": envs, "updates": updates, "exchanges": 17 * updates, "replica_checks": updates // check + 1, "epoch_changes_crossed": epochs,
                                          "replicas_bitwise_identical": True, "params_finite": bool(torch.isfinite(agent.flat).all().item()),
                                          "mean_return_last_rollout_rank0": round(st[1] / max(st[0], 1), 1), "wall_seconds": round(time.time() - t0, 1)}))
torch.distributed.barrier()
DD.destroy_native_comms()
torch.distributed.destroy_process_group()
