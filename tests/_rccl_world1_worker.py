"""Worker of tests/test_gpu_rccl.py — launched by torch.distributed.run with ONE rank and MIRL_FORCE_PG=1, so that a real NCCL (= RCCL)
process group and libmirl's own RCCL communicator exist on a one-GPU box.  Checks, bit for bit, that the three update routes agree:
  (a) mi_ppo_update                              single-process fusion, no collective
  (b) mi_ppo_update_sharded over RCCL            ONE C call, 17 in-stream ncclAllReduce per update (world_size 1: the data is unchanged)
  (c) the host-sequenced launches with torch.distributed all-reduces over RCCL between them
and that mi_comm_allreduce_sum really runs on the stream it is given."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

import deep_rl_amd as D  # noqa: E402
import deep_rl_amd.dist as DD  # noqa: E402
import deep_rl_amd.engine as E  # noqa: E402
from deep_rl_amd import _native as N  # noqa: E402

rank, world, local_rank = DD.init_from_env("nccl")
assert world == 1 and torch.distributed.is_initialized() and torch.distributed.get_backend() == "nccl"
dev = torch.device("cuda", local_rank)
torch.cuda.set_device(dev)
comm = DD.native_comm()
assert comm is not None
ws, rk, ver, cnt = N.C.c_int(), N.C.c_int(), N.C.c_int(), N.C.c_int()
N.check(N.lib().mi_comm_info(comm, N.C.byref(ws), N.C.byref(rk), N.C.byref(ver), N.C.byref(cnt)), "mi_comm_info")
assert ws.value == 1 and rk.value == 0 and ver.value > 0 and cnt.value == 1

# the collective itself: in-stream, in place, f32 and f64
x = torch.arange(9159, dtype=torch.float32, device=dev) * 0.25
y = x.clone()
N.check(N.lib().mi_comm_allreduce_sum(comm, N.ptr(y), y.numel(), 0, N.stream_ptr(dev)), "mi_comm_allreduce_sum")
z = torch.linspace(-1, 1, 48, dtype=torch.float64, device=dev)
z2 = z.clone()
N.check(N.lib().mi_comm_allreduce_sum(comm, N.ptr(z2), z2.numel(), 1, N.stream_ptr(dev)), "mi_comm_allreduce_sum")
assert torch.equal(x, y) and torch.equal(z, z2)


def run(mode, n_envs):
    E._FORCE_NATIVE_SHARDED = mode == "native"
    E._FORCE_SHARDED_SEQUENCE = mode == "torch"
    DD._FORCE_COLLECTIVES = mode == "torch"
    env = D.make("CartPole-v1", num_envs=n_envs, device=dev, seed=5)
    torch.manual_seed(5)
    agent = D.ActorCritic(env)
    opt = D.ClipAdam(agent, lr=2.5e-4, eps=1e-5, max_grad_norm=0.5)
    eng = D.PPOEngine(env, agent, opt, num_steps=128)
    eng.reset()
    for u in range(2):
        opt.param_groups[0]["lr"] = (1.0 - u / 4) * 2.5e-4
        eng.update()
    torch.cuda.synchronize()
    assert opt.step_count == 32
    return [t.clone() for t in (agent.flat, opt.exp_avg, opt.exp_avg_sq, opt.grad_norm, eng.loss_terms, eng.advantages)]


for n_envs in (8, 4096):
    ref = run("fused", n_envs)
    for mode in ("native", "torch"):
        got = run(mode, n_envs)
        for a, b in zip(ref, got):
            assert torch.equal(a, b), (n_envs, mode)
    assert torch.isfinite(ref[0]).all()
DD.destroy_native_comms()
torch.distributed.barrier()
torch.distributed.destroy_process_group()
print("RCCL_WORLD1_OK rccl_version=%d" % ver.value)
