#!/usr/bin/env python3
"""What the multi-GPU exchange costs, measured on ONE GPU with a real one-rank RCCL communicator:
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 tools/bench_rccl_world1.py   (MIRL_FORCE_PG=1)
  * us per in-stream ncclAllReduce of the 9,159-float gradient buffer (36.6 KB) and of the 48-double statistics buffer, libmirl's communicator;
  * ms per outer update (4096 envs) for: the single-process fusion, mi_ppo_update_sharded (ONE C call, 17 RCCL all-reduces in-stream),
    and the host-sequenced route with 17 torch.distributed all-reduces.
At world_size 1 a collective moves no bytes between GPUs: these are the FIXED costs (enqueue + kernel) every rank pays per collective."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import deep_rl_amd as D
import deep_rl_amd.dist as DD
import deep_rl_amd.engine as E
from deep_rl_amd import _native as N

rank, world, local_rank = DD.init_from_env("nccl")
dev = torch.device("cuda", local_rank); torch.cuda.set_device(dev)
comm = DD.native_comm()
assert comm is not None, "run under torch.distributed.run with MIRL_FORCE_PG=1"
out = {}
for name, t, dt in (("grad_9159_f32", torch.zeros(9159, device=dev), 0), ("stats_48_f64", torch.zeros(48, dtype=torch.float64, device=dev), 1)):
    for _ in range(20):
        N.check(N.lib().mi_comm_allreduce_sum(comm, N.ptr(t), t.numel(), dt, N.stream_ptr(dev)))
    torch.cuda.synchronize(); tm = N.Timer(); tm.start(N.stream_ptr(dev))
    for _ in range(500):
        N.check(N.lib().mi_comm_allreduce_sum(comm, N.ptr(t), t.numel(), dt, N.stream_ptr(dev)))
    tm.stop(N.stream_ptr(dev)); out["us_per_allreduce_" + name] = round(1e3 * tm.elapsed_ms() / 500, 2)
    t0 = time.perf_counter()
    for _ in range(500):
        torch.distributed.all_reduce(t)
    torch.cuda.synchronize(); out["us_per_torch_allreduce_" + name] = round(1e6 * (time.perf_counter() - t0) / 500, 2)


def run(mode, steps=30, warm=5):
    E._FORCE_NATIVE_SHARDED = mode == "native"; E._FORCE_SHARDED_SEQUENCE = mode == "torch"; DD._FORCE_COLLECTIVES = mode == "torch"
    env = D.make("CartPole-v1", num_envs=4096, device=dev, seed=1)
    torch.manual_seed(1)
    agent = D.ActorCritic(env); opt = D.ClipAdam(agent, lr=2.5e-4, eps=1e-5, max_grad_norm=0.5)
    eng = D.PPOEngine(env, agent, opt, num_steps=128); eng.reset()
    for _ in range(warm): eng.update()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps): eng.update()
    torch.cuda.synchronize()
    return round(1e3 * (time.perf_counter() - t0) / steps, 4)

for rep in range(2):
    for mode in ("fused", "native", "torch"):
        out.setdefault("ms_per_update_" + mode, []).append(run(mode))
out["collectives_per_update"] = 17
out["us_per_collective_in_update_native"] = round(1e3 * (min(out["ms_per_update_native"]) - min(out["ms_per_update_fused"])) / 17, 2)
out["us_per_collective_in_update_torch"] = round(1e3 * (min(out["ms_per_update_torch"]) - min(out["ms_per_update_fused"])) / 17, 2)
print(json.dumps(out))
DD.destroy_native_comms(); torch.distributed.barrier(); torch.distributed.destroy_process_group()
