"""The N>1 code path on ONE GPU: two ranks (both on cuda:0, gloo backend carrying CUDA tensors) run the real HIP kernels and
the engine's two all-reduces; a single process that owns all the envs and takes the union minibatch must agree
(SURVEY.md §4 tier 4: "1 GPU with 2N envs == 2 GPUs with N envs").  RCCL itself is exercised by the driver's 8-GPU run."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
T, NL = 128, 64  # envs per rank


def _mk(dev, n, base, pg_ready):
    import deep_rl_amd as D

    env = D.make("CartPole-v1", num_envs=n, device=dev, seed=11, env_id_base=base)
    torch.manual_seed(11)
    agent = D.ActorCritic(env)
    opt = D.ClipAdam(agent, lr=2.5e-4, eps=1e-5, max_grad_norm=0.5)
    return D.PPOEngine(env, agent, opt, num_steps=T)


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from deep_rl_amd import dist as D

    D.init_from_env("gloo")
    dev = torch.device("cuda", 0)
    eng = _mk(dev, NL, rank * NL, True)
    assert eng.world_size == world
    params0 = eng.agent.flat.cpu().numpy().copy()
    eng.reset(); eng.rollout(); eng.compute_gae()
    out = {"params0": params0, "storage": {n: getattr(eng, n).cpu().numpy() for n in ["observations", "actions", "dones", "advantages"]}}
    eng.make_perm(0)                       # same key on every rank, applied to the rank's local rows
    out["perm"] = eng.perm.cpu().numpy()
    eng.adv_stats()                        # all-reduce #1
    out["sums"] = eng.adv_sums.cpu().numpy().copy()
    eng.minibatch_grad(0)                  # all-reduce #2
    out["grads"] = eng.grads.cpu().numpy().copy(); out["terms"] = eng.loss_terms.cpu().numpy().copy()
    eng.optimizer_step()
    out["params1"] = eng.agent.flat.cpu().numpy().copy()
    # and two whole sharded updates through engine.update()'s multi-rank branch
    eng.update(); eng.update()
    out["params3"] = eng.agent.flat.cpu().numpy().copy()
    q.put((rank, out))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def test_two_ranks_on_one_gpu_match_single_process():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    dev = torch.device("cuda", 0)
    eng = _mk(dev, 2 * NL, 0, False)
    # torch's CPU orthogonal init (QR) may differ in the last bit between a fresh process and this one (thread count / code path
    # chosen after other tests ran); the comparison is about the device path, so start from the ranks' own initial parameters
    assert np.array_equal(res[0]["params0"], res[1]["params0"])
    assert np.abs(eng.agent.flat.cpu().numpy() - res[0]["params0"]).max() < 1e-6
    eng.agent.load_flat(res[0]["params0"])
    eng.reset(); eng.rollout(); eng.compute_gae()
    # N-invariance: each rank's trajectories are the matching columns of the big run
    for r in range(2):
        sl = slice(r * NL, (r + 1) * NL)
        for n in ["observations", "actions", "dones", "advantages"]:
            big, small = getattr(eng, n)[:, sl].cpu().numpy(), res[r]["storage"][n]
            bad = np.argwhere(big != small)
            assert bad.shape[0] == 0, (r, n, bad.shape[0], bad[:8].tolist(), [(float(big[tuple(b)]), float(small[tuple(b)])) for b in bad[:4]])
    # union minibatch in the big run's row numbering: local row t*NL + e  ->  t*(2 NL) + r*NL + e
    mb = T * NL // 4
    gidx = np.concatenate([(res[r]["perm"][:mb] // NL) * (2 * NL) + r * NL + res[r]["perm"][:mb] % NL for r in range(2)]).astype(np.int32)
    eng.perm[:2 * mb].copy_(torch.from_numpy(gidx).to(dev))
    eng.adv_stats(mb=2 * mb, n_mb=1)
    sums = eng.adv_sums[0].cpu().numpy()
    for r in range(2):
        assert np.allclose(res[r]["sums"][0], sums, rtol=1e-12, atol=1e-9), (res[r]["sums"][0], sums)
    eng.minibatch_grad(0, mb=2 * mb)
    g = eng.grads.cpu().numpy(); t = eng.loss_terms.cpu().numpy()
    for r in range(2):
        assert np.abs(res[r]["grads"] - g).max() <= 5e-6 * np.abs(g).max(), np.abs(res[r]["grads"] - g).max() / np.abs(g).max()
        assert np.allclose(res[r]["terms"], t, rtol=2e-5, atol=1e-6)
    eng.optimizer_step()
    p1 = eng.agent.flat.cpu().numpy()
    assert np.abs(res[0]["params1"] - p1).max() < 2e-7
    # replicas never diverge: bitwise identical parameters on both ranks after every step
    assert np.array_equal(res[0]["params1"], res[1]["params1"]) and np.array_equal(res[0]["params3"], res[1]["params3"])
    assert np.isfinite(res[0]["params3"]).all() and not np.array_equal(res[0]["params1"], res[0]["params3"])
