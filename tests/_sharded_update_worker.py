"""Worker of tests/test_gpu_multigpu.py — one rank of a world_size-2 job started by torch.distributed.run.

  backend nccl (needs >= 2 GPUs, one per rank): PPOEngine.update() takes the ONE-CALL route, mi_ppo_update_sharded with 17 in-stream ncclAllReduce
      per update on libmirl's own RCCL communicator (reference ppo.py:189-192: backward -> [gradient exchange] -> clip_grad_norm_ -> step);
  backend gloo (both ranks on cuda:0): the host-sequenced route over the same launches — run on the one-GPU box so that the comparison
      harness below is itself exercised every round;
  MIRL_COMM=p2p with either backend: the ONE-CALL route on libmirl's peer-to-peer carrier (hipIpc-mapped inboxes, rank-ordered sum) — with gloo and both ranks on
      cuda:0 this is how the one-GPU box runs mi_ppo_update_sharded at world_size 2.

Each rank owns NL envs (global ids [rank*NL, (rank+1)*NL)), runs UPDATES whole updates and dumps its final state to OUT_DIR/rank<r>.npz; the TEST process
(no process group: an engine built inside a rank would join the ranks' collectives) then plays the single process that owns all 2*NL envs — the explicit
launch sequence with the UNION minibatches, each rank's keyed local permutation mapped into the big run's row numbering — and compares.  With nccl, a second pair of engines repeats the updates on the
host-sequenced route (torch.distributed all-reduces over RCCL between the launches): at world_size 2 a SUM all-reduce is a + b on every rank
whatever the algorithm, so the two routes must agree bit for bit."""
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

T, NL, UPDATES, SEED = 128, int(os.environ.get("MIRL_TEST_NL", "64")), 2, 11
backend = os.environ["MIRL_TEST_BACKEND"]
out_dir = os.environ["MIRL_TEST_OUT"]
rank, world, local_rank = DD.init_from_env(backend)
assert world == int(os.environ.get("MIRL_TEST_WORLD", "2"))
dev = torch.device("cuda", local_rank if backend == "nccl" else 0)
torch.cuda.set_device(dev)


def mk(n, base):
    env = D.make("CartPole-v1", num_envs=n, device=dev, seed=SEED, env_id_base=base)
    torch.manual_seed(SEED)
    agent = D.ActorCritic(env)
    opt = D.ClipAdam(agent, lr=2.5e-4, eps=1e-5, max_grad_norm=0.5)
    return D.PPOEngine(env, agent, opt, num_steps=T)


def run_rank(params0=None):
    eng = mk(NL, rank * NL)
    if params0 is not None:
        eng.agent.load_flat(params0)
    p0 = eng.agent.flat.cpu().numpy().copy()
    eng.reset()
    for _ in range(UPDATES):
        eng.update()
    torch.cuda.synchronize()
    return eng, p0


def state(eng):
    o = eng.optimizer
    return {"params": eng.agent.flat.cpu().numpy(), "exp_avg": o.exp_avg.cpu().numpy(), "exp_avg_sq": o.exp_avg_sq.cpu().numpy(), "grads": eng.grads.cpu().numpy(),
            "loss_terms": eng.loss_terms.cpu().numpy(), "grad_norm": o.grad_norm.cpu().numpy(), "observations": eng.observations.cpu().numpy(),
            "advantages": eng.advantages.cpu().numpy(), "adv_sums": eng._adv_sums_all.cpu().numpy()}


eng, params0 = run_rank()
native = DD.native_comm(eng.pg) is not None
carrier_name = DD.resolved_carrier()   # (MIRL_COMM=auto: what the probe chose; taken before destroy_native_comms forgets it)
p2p = carrier_name == "p2p"    # MIRL_COMM=p2p: the one-call route on the peer-to-peer carrier, whatever the process group (gloo with both ranks on cuda:0 included)
assert native == (backend == "nccl" or p2p), "backend %s, carrier %s: one-call route %s" % (backend, DD.resolved_carrier(), native)
if native:
    ws, rk, ver, cnt = N.C.c_int(), N.C.c_int(), N.C.c_int(), N.C.c_int()
    N.check(N.lib().mi_comm_info(DD.native_comm(eng.pg), N.C.byref(ws), N.C.byref(rk), N.C.byref(ver), N.C.byref(cnt)), "mi_comm_info")
    assert ws.value == world and rk.value == rank and cnt.value == world and (ver.value == 0 if p2p else ver.value > 0)
    assert N.lib().mi_comm_carrier(DD.native_comm(eng.pg)) == int(p2p)
    DD.check_native_comm(eng.pg)   # no wait of the P2P carrier ran out
    eng.check_replicas()
st = state(eng)
st["params0"] = params0
st["native"] = np.array([int(native)])
if native:   # the host-sequenced route over torch.distributed all-reduces (RCCL, or gloo under the P2P carrier), from the same initial parameters: bit-identical at world_size 2
    E._FORCE_SHARDED_SEQUENCE = True
    eng2, _ = run_rank(params0)
    E._FORCE_SHARDED_SEQUENCE = False
    for k, v in state(eng2).items():
        st["seq_" + k] = v
np.savez(os.path.join(out_dir, "rank%d.npz" % rank), **st)
torch.distributed.barrier()
DD.destroy_native_comms()
torch.distributed.destroy_process_group()
if rank == 0:
    print("SHARDED_WORKER_OK backend=%s native=%d carrier=%s" % (backend, int(native), carrier_name))
