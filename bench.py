#!/usr/bin/env python3
"""bench.py — headline benchmark: PPO / CartPole-v1 at 4096 envs per GPU (BASELINE.json configs[1] / [4]).

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...)

One "step" = one outer update of reference ppo.py:105-192 over synthetic CartPole data: a 128-step rollout of
4096 envs (on-device env.step, keyed RNG), the GAE scan, and 4 epochs x 4 minibatches of 131,072 rows
(forward + backward + grad-clip + Adam), with the policy really learning.  W untimed updates, then exactly K timed
updates bracketed by barrier + torch.cuda.synchronize(); MAX over ranks; rank 0 prints ONE JSON line.
  value      = env-steps/s of the whole job (T * envs_per_gpu * n_gpus * K / seconds); updates/s is reported beside it.
  roofline   = the dominant kernel (grad_kernel: f32 MFMA bound): algorithmic FLOPs per launch / its average launch
               duration, measured live over the timed region with HIP events on the launch stream (mi_prof_*).
  cpu_baseline = the CPU oracle (a C port of the reference loop, OpenMP over envs / rows) on this box's host cores,
               rank 0 and N = 1 only, on a bounded sample of the SAME workload.
Inputs are resident in HBM when the timed region starts (storage, parameters and env state never leave the device).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ENVS_PER_GPU = 4096
T = 128
# algorithmic work per env-step (restated in DESIGN.md)
MACS_FWD_BOTH_NETS = 8896                       # actor 4480 + critic 4416
FLOPS_PER_ROW_UPDATE = 3 * 2 * MACS_FWD_BOTH_NETS  # forward + ~2x backward, per row and epoch  = 53,376
BYTES_PER_ENV_STEP = 292                         # rollout 40 + env state 72 + GAE 20 + 4 epochs x 40 gathered
PEAK_F32_MFMA_TFLOPS = 157.3                     # MI355X_MICROARCH.md: f32-input MFMA == f32 vector peak
PEAK_HBM_GBS = 8000.0


def usable_cpus():
    """CPUs this process may really use: min(logical CPUs, affinity mask, cgroup v2 cpu.max quota)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except AttributeError:
        pass
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(quota) // int(period)))
    except (OSError, ValueError):
        pass
    return n


def cpu_baseline(params, min_seconds=10.0):
    """The oracle's whole-update loop (oracle/cpu_ref.c ref_ppo_update) on the usable host cores, same 4096-env workload."""
    from oracle import cpu_ref as R

    cores = usable_cpus()
    base = R.PPOBaseline(params, ENVS_PER_GPU, T=T, seed=1, threads=cores)
    t0 = time.perf_counter()
    k = 0
    while True:
        base.run_update()
        k += 1
        dt = time.perf_counter() - t0
        if dt >= min_seconds or k >= 16:
            break
    return {"value": round(k * T * ENVS_PER_GPU / dt, 1), "unit": "env-steps/s", "updates_per_s": round(k / dt, 4),
            "cores": base.threads, "kind": "port",
            "sample": "%d outer update(s) of the same 4096-env x 128-step workload (%.1f s) in the C oracle, OpenMP over envs/rows" % (k, dt)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch

    import deep_rl_amd as D
    from deep_rl_amd import _native as N
    from deep_rl_amd.dist import init_from_env

    # production: RCCL ("nccl"), one rank per GPU.  MIRL_BENCH_BACKEND=gloo MIRL_BENCH_ONE_GPU=1 lets the N > 1 code path of this script be
    # exercised on a single-GPU box (both ranks on cuda:0; tests/test_gpu_script.py) — RCCL itself refuses two ranks on one device.
    rank, world, local_rank = init_from_env(os.environ.get("MIRL_BENCH_BACKEND", "nccl"))
    if os.environ.get("MIRL_BENCH_ONE_GPU", "0") == "1":
        local_rank = 0
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d (launch N>1 with torch.distributed.run)" % (args.gpus, world))
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    num_updates = args.warmup + args.steps
    env = D.make("CartPole-v1", num_envs=ENVS_PER_GPU, device=dev, seed=1, env_id_base=rank * ENVS_PER_GPU)
    torch.manual_seed(1)
    agent = D.ActorCritic(env)
    params0 = agent.flat.cpu().numpy().copy()
    opt = D.ClipAdam(agent, lr=2.5e-4, eps=1e-5, max_grad_norm=0.5)
    eng = D.PPOEngine(env, agent, opt, num_steps=T, n_minibatch=4, update_epochs=4)
    eng.reset()
    stats_host = torch.zeros(4, dtype=torch.int32).pin_memory()

    def one_update(u):
        opt.param_groups[0]["lr"] = (1.0 - u / num_updates) * 2.5e-4  # ppo.py:107-108
        eng.update()
        eng.episode_summary_async(stats_host)  # what a training loop reads per update; no host sync

    for u in range(args.warmup):
        one_update(u)
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    N.prof_begin(args.steps * 16 + 16, tags=["grad"])  # only the dominant kernel is bracketed inside the timed region
    t0 = time.perf_counter()
    for u in range(args.warmup, num_updates):
        one_update(u)
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    prof = N.prof_end()
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
        dt = float(tt.item())

    # untimed: per-kernel breakdown of 3 more updates (all tags)
    N.prof_begin(3 * 64 + 64)
    for u in range(3):
        eng.update()
    breakdown = N.prof_end()

    finite = bool(torch.isfinite(agent.flat).all().item())
    ep = stats_host.tolist()
    if rank == 0:
        env_steps = args.steps * T * ENVS_PER_GPU * world
        mb = T * ENVS_PER_GPU // 4
        g_ms, g_n = prof["grad"]
        flops_per_launch = FLOPS_PER_ROW_UPDATE * mb
        ach = flops_per_launch / (g_ms / max(g_n, 1) * 1e-3) / 1e12 if g_n else 0.0
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "latest_pmc.json")
        if os.path.exists(pmc):
            try:
                traffic = json.load(open(pmc)).get("grad_kernel", {}).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        steps_per_s = env_steps / dt
        out = {
            "metric": "env-steps/sec + updates/sec, PPO CartPole-v1 4096 envs/GPU",
            "value": round(steps_per_s, 1), "unit": "env-steps/s",
            "updates_per_s": round(args.steps / dt, 3), "optimizer_steps_per_s": round(16 * args.steps / dt, 2),
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * dt / args.steps, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "ppo.py CartPole-v1, %d envs/GPU x %d steps per update, 4 epochs x 4 minibatches of %d rows, "
                                   "2x64-tanh actor+critic (9155 params), on-device env.step + GAE + fwd/bwd + clip + Adam" % (ENVS_PER_GPU, T, mb),
                       "envs_per_gpu": ENVS_PER_GPU, "num_steps": T, "minibatch_rows": mb, "parallelism": "env-sharded x%d, grad all-reduce" % world},
            "roofline": {"bound": "mfma", "kernel": "grad_kernel", "achieved": round(ach, 3), "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(ach / PEAK_F32_MFMA_TFLOPS, 4), "traffic": traffic,
                         "flops_per_launch": flops_per_launch, "avg_launch_us": round(1e3 * g_ms / max(g_n, 1), 2), "launches": g_n},
            "hbm_roofline": {"algorithmic_bytes_per_env_step": BYTES_PER_ENV_STEP,
                             "achieved_GBps_per_gpu": round(BYTES_PER_ENV_STEP * steps_per_s / world / 1e9, 2), "peak_GBps": PEAK_HBM_GBS,
                             "frac": round(BYTES_PER_ENV_STEP * steps_per_s / world / 1e9 / PEAK_HBM_GBS, 6)},
            "kernel_ms_per_update": {k: round(v[0] / 3, 4) for k, v in breakdown.items()},
            "last_rollout": {"episodes": ep[0], "mean_return": round(ep[1] / max(ep[0], 1), 2), "max_return": ep[2]},
            "params_finite": finite,
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(params0, float(os.environ.get("MIRL_CPU_BASELINE_SECONDS", "10")))  # bounded sample (default 10 s)
        print(json.dumps(out), flush=True)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
