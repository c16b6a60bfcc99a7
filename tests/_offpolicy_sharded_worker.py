"""Worker of tests/test_gpu_multigpu.py (off-policy half) — one rank of a world_size-2 job started by torch.distributed.run.

DQNEngine, PERDQNEngine and SACEngine with world_size 2: every rank owns NL envs (global ids [rank*NL, (rank+1)*NL)) and its own replay ring, contributes its gradient
share scaled by 1 / (world * batch), and steps the replicated parameters identically (reference dqn.py:118-133, per.py:124-153, sac.py:165-217 with the gradient
exchange between backward and optimizer.step()).
  backend nccl (needs >= 2 GPUs, one per rank): the ONE-CALL routes — mi_dqn_td_update_sharded, mi_sac_critic_update_sharded, mi_sac_actor_update_sharded,
      mi_sac_alpha_step_sharded with the in-stream ncclAllReduce on libmirl's own RCCL communicator; then the same work again on the host-sequenced route
      (MIRL_NATIVE_COMM=0: *_grad launch, torch.distributed all-reduce over RCCL, Adam launch) — at world_size 2 a SUM all-reduce is a + b on both ranks whatever the
      algorithm, so the two routes must agree bit for bit;
  backend gloo (both ranks on cuda:0): the host-sequenced route, so that the harness and the comparison below run every round on the one-GPU box;
  MIRL_COMM=p2p with either backend: the ONE-CALL routes on libmirl's peer-to-peer carrier — with gloo and both ranks on cuda:0 this is how the one-GPU box runs them
      at world_size 2.
Each rank dumps its final state to OUT_DIR/off_rank<r>.npz; the TEST process compares rank against rank (bitwise), route against route (bitwise) and, for DQN and SAC
with caller-supplied batches and noise, against a single process that owns all 2*NL envs and takes the union batch."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

import deep_rl_amd as D  # noqa: E402
import deep_rl_amd.dist as DD  # noqa: E402

NL, STEPS, B, ROUNDS, SEED = 8, 40, 64, 3, 7
backend = os.environ["MIRL_TEST_BACKEND"]
out_dir = os.environ["MIRL_TEST_OUT"]
os.environ["MIRL_CHECK_REPLICAS"] = "2"      # the divergence guard runs inside every second train_step of every engine below
rank, world, local_rank = DD.init_from_env(backend)
assert world == 2
dev = torch.device("cuda", local_rank if backend == "nccl" else 0)
torch.cuda.set_device(dev)


def inputs(r):
    """Per-rank batches (flat local ring indices of filled slots) and standard-normal draws: ROUNDS x {idx [B], eps [3, B]} — the test process rebuilds the union."""
    rng = np.random.default_rng(1000 + r)
    return [(rng.integers(0, (STEPS - 1) * NL, B), rng.standard_normal((3, B)).astype(np.float32)) for _ in range(ROUNDS)]


def mk_dqn(per):
    env = D.make("CartPole-v1", num_envs=NL, device=dev, seed=SEED, env_id_base=rank * NL)
    torch.manual_seed(SEED)
    q = D.QNetwork(env); t = D.QNetwork(env); t.load_state_dict(q.state_dict())
    Eng = D.PERDQNEngine if per else D.DQNEngine
    return Eng(env, q, t, D.ClipAdam(q, lr=2.5e-4, eps=1e-8), slots=STEPS + 1 + 10 * ROUNDS, batch_size=B, learning_starts=10, total_timesteps=400, max_episodes_logged=0)


def mk_sac():
    env = D.make("Pendulum-v1", num_envs=NL, device=dev, seed=SEED, env_id_base=rank * NL)
    torch.manual_seed(SEED)
    a = D.Actor(env)
    qs = [D.SoftQNetwork(env) for _ in range(4)]
    qs[2].load_state_dict(qs[0].state_dict()); qs[3].load_state_dict(qs[1].state_dict())
    return D.SACEngine(env, a, *qs, slots=STEPS + 1, batch_size=B, learning_starts=10, max_episodes_logged=0)


def run_all():
    out = {}
    ins = inputs(rank)
    # DQN: caller-supplied batches (dqn.py:116-133)
    dqn = mk_dqn(False)
    assert dqn.world_size == 2
    out["dqn_init"] = dqn.q.flat.cpu().numpy().copy()
    dqn.reset(); dqn.act(STEPS)
    for idx, _ in ins:
        dqn.train_step(idx)
    out.update(dqn_q=dqn.q.flat, dqn_m=dqn.optimizer.exp_avg, dqn_v=dqn.optimizer.exp_avg_sq, dqn_loss=dqn.loss, dqn_grads=dqn.grads, dqn_obs=dqn.observations)
    out["dqn_native"] = np.array([int(dqn._native_sharded())])
    # PER: keyed prioritized sampling from the rank's own ring (per.py:124-153), acting in between
    per = mk_dqn(True)
    per.reset(); per.act(STEPS)
    for _ in range(ROUNDS):
        per.act(10); per.train_step()
    out.update(per_q=per.q.flat, per_m=per.optimizer.exp_avg, per_loss=per.loss, per_prio_sum=per.priorities.double().sum().reshape(1), per_maxp=per.max_priority)
    # SAC: caller-supplied batches and noise (sac.py:165-217)
    sac = mk_sac()
    assert sac.world_size == 2 and sac.rank == rank
    out["sac_actor_init"], out["sac_q_init"] = sac.actor.flat.cpu().numpy().copy(), sac.q_flat.cpu().numpy().copy()
    sac.reset()
    for _ in range(STEPS):
        sac.act()
    for idx, eps in ins:
        sac.sample(idx)
        sac.update_critic(torch.from_numpy(eps[0]), polyak=True); sac.update_actor(torch.from_numpy(eps[1])); sac.update_alpha(torch.from_numpy(eps[2]))
        if sac.update_index % 2 == 0:
            sac.check_replicas()
    out.update(sac_q=sac.q_flat, sac_qt=sac.qt_flat, sac_actor=sac.actor.flat, sac_log_alpha=sac.log_alpha, sac_q_losses=sac.q_losses, sac_actor_out=sac.actor_out,
               sac_obs=sac.observations, sac_qm=sac.q_optimizer.exp_avg, sac_am=sac.actor_optimizer.exp_avg_sq)
    out["sac_native"] = np.array([int(sac._comm() is not None)])
    torch.cuda.synchronize()
    return {k: (v.detach().cpu().numpy() if torch.is_tensor(v) else v) for k, v in out.items()}


st = run_all()
native = bool(st["dqn_native"][0]) and bool(st["sac_native"][0])
carrier_name = DD.resolved_carrier()   # (MIRL_COMM=auto: what the probe chose; taken before destroy_native_comms forgets it)
p2p = carrier_name == "p2p"    # MIRL_COMM=p2p: the one-call routes on the peer-to-peer carrier, whatever the process group
assert native == (backend == "nccl" or p2p), "backend %s, carrier %s: one-call routes %s" % (backend, DD.resolved_carrier(), native)
if native:
    DD.check_native_comm()
if native:   # host-sequenced route over torch.distributed all-reduces (RCCL, or gloo under the P2P carrier): bit-identical at world_size 2
    os.environ["MIRL_NATIVE_COMM"] = "0"
    st2 = run_all()
    os.environ["MIRL_NATIVE_COMM"] = "1"
    assert not bool(st2["dqn_native"][0]) and not bool(st2["sac_native"][0])
    for k, v in st2.items():
        st["seq_" + k] = v
np.savez(os.path.join(out_dir, "off_rank%d.npz" % rank), **st)
torch.distributed.barrier()
DD.destroy_native_comms()
torch.distributed.destroy_process_group()
if rank == 0:
    print("OFFPOLICY_WORKER_OK backend=%s native=%d carrier=%s" % (backend, int(native), carrier_name))
