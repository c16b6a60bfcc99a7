"""Worker of tests/test_gpu_rccl.py — launched by torch.distributed.run with ONE rank and MIRL_FORCE_PG=1, so that a real NCCL (= RCCL)
process group and libmirl's own RCCL communicator exist on a one-GPU box.  Checks, bit for bit, that the three update routes agree:
  (a) mi_ppo_update                              single-process fusion, no collective
  (b) mi_ppo_update_sharded over RCCL            ONE C call, 17 in-stream ncclAllReduce per update (world_size 1: the data is unchanged)
  (b') the same with mi_ppo_test_assume_sharded(1): the owed optimizer steps recompute the clip coefficient from the gradient (what world_size > 1 does)
  (c) the host-sequenced launches with torch.distributed all-reduces over RCCL between them
and that mi_comm_allreduce_sum really runs on the stream it is given; the same three-way identity for the DQN, PER and SAC engines
(mi_dqn_td_update_sharded, mi_sac_critic / actor_update_sharded, mi_sac_alpha_step_sharded)."""
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
    E._FORCE_NATIVE_SHARDED = mode.startswith("native")
    E.set_assume_sharded(mode == "native+assume")   # the world_size > 1 form of the owed optimizer step (norm_parts = nullptr) on the real RCCL route
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
    for mode in ("native", "native+assume", "torch"):
        got = run(mode, n_envs)
        for a, b in zip(ref, got):
            assert torch.equal(a, b), (n_envs, mode)
    assert torch.isfinite(ref[0]).all()
E.set_assume_sharded(False)

# ---- DQN / PER / SAC: fused single-process calls == one-call RCCL route (mi_dqn_td_update_sharded, mi_sac_*_sharded) == host-sequenced route with torch all-reduces ----
import deep_rl_amd.dqn_engine as DE  # noqa: E402
import deep_rl_amd.sac_engine as SE  # noqa: E402


def set_mode(mode):
    DE._FORCE_SHARDED = SE._FORCE_SHARDED = mode != "fused"
    DD._FORCE_COLLECTIVES = mode == "torch"
    os.environ["MIRL_NATIVE_COMM"] = "0" if mode == "torch" else "1"


def run_dqn(mode, per):
    set_mode(mode)
    env = D.make("CartPole-v1", num_envs=64, device=dev, seed=5)
    torch.manual_seed(5)
    q = D.QNetwork(env); t = D.QNetwork(env); t.load_state_dict(q.state_dict())
    Eng = D.PERDQNEngine if per else D.DQNEngine
    eng = Eng(env, q, t, D.ClipAdam(q, lr=2.5e-4, eps=1e-8), slots=64, batch_size=128, learning_starts=10, total_timesteps=2000, max_episodes_logged=0)
    eng.reset()
    rng = np.random.default_rng(3)
    for k in range(12):
        eng.act(10)
        eng.train_step(rng.integers(0, min(eng.global_step, 64) * 64 - 64, 128) if not per else None)
    torch.cuda.synchronize()
    return [x.clone() for x in (q.flat, eng.optimizer.exp_avg, eng.optimizer.exp_avg_sq, eng.loss, eng.grads)]


def run_sac(mode):
    set_mode(mode)
    env = D.make("Pendulum-v1", num_envs=48, device=dev, seed=9)
    torch.manual_seed(9)
    actor = D.Actor(env)
    qs = [D.SoftQNetwork(env) for _ in range(4)]
    qs[2].load_state_dict(qs[0].state_dict()); qs[3].load_state_dict(qs[1].state_dict())
    eng = D.SACEngine(env, actor, *qs, slots=40, batch_size=256, learning_starts=4)
    eng.reset()
    rng = np.random.default_rng(4)
    for _ in range(20):
        eng.act()
        if eng.global_step > 6:
            eng.train_step(indices=rng.integers(0, min(eng.global_step, 40) * 48 - 48, 256))
    torch.cuda.synchronize()
    return [x.clone() for x in (actor.flat, eng.q_flat, eng.qt_flat, eng.log_alpha, eng.alpha, eng._alpha_m, eng.actor_optimizer.exp_avg, eng.q_optimizer.exp_avg_sq, eng.q_losses)]


for name, fn in (("dqn", lambda m: run_dqn(m, False)), ("per", lambda m: run_dqn(m, True)), ("sac", run_sac)):
    ref = fn("fused")
    for mode in ("native", "torch"):
        got = fn(mode)
        for a, b in zip(ref, got):
            assert torch.equal(a, b), (name, mode)
    assert all(torch.isfinite(x).all() for x in ref), name
set_mode("fused")
DD.destroy_native_comms()
torch.distributed.barrier()
torch.distributed.destroy_process_group()
print("RCCL_WORLD1_OK rccl_version=%d" % ver.value)
