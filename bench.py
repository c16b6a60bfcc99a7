#!/usr/bin/env python3
"""bench.py — headline benchmark: PPO / CartPole-v1 at 4096 envs per GPU (BASELINE.json configs[1] / [4]).

    python bench.py --gpus N --steps K --warmup W
    N > 1 without WORLD_SIZE in the environment: this process — before it imports torch or touches a GPU — starts
        python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port <free> bench.py --gpus N ...
    as a CHILD process (never exec), relays rank 0's JSON line and exits with the child's return code.  Launched by the driver through
    torch.distributed.run itself (WORLD_SIZE set), it is a rank.

One "step" = one outer update of reference ppo.py:105-192 over synthetic CartPole data: a 128-step rollout of
4096 envs (on-device env.step, keyed RNG), the GAE scan, and 4 epochs x 4 minibatches of 131,072 rows
(forward + backward + grad-clip + Adam), with the policy really learning.  W untimed updates, then exactly K timed
updates bracketed by barrier + torch.cuda.synchronize(); MAX over ranks; rank 0 prints ONE JSON line.
  value      = env-steps/s of the whole job (T * envs_per_gpu * n_gpus * K / seconds); updates/s is reported beside it.
  roofline   = the dominant kernel (grad_kernel: f32 MFMA bound): algorithmic FLOPs per launch / its average launch
               duration, measured live over the timed region with HIP events on the launch stream (mi_prof_*).
  cpu_baseline = the CPU oracle (a C port of the reference loop, OpenMP over envs / rows) on this box's host cores,
               rank 0 and N = 1 only, on a bounded sample of the SAME workload; cpu_baseline_n1 = the same oracle at the
               reference's own shape (1 env, 1 thread), the like-for-like stand-in for the unmodified ppo.py.
Inputs are resident in HBM when the timed region starts (storage, parameters and env state never leave the device).
At N = 1 the same JSON line also carries `config3_dqn` and `config4_sac`: BASELINE.json configs[2] / [3] run AFTER (outside)
the PPO timed region, each with its own ms_per_step, dominant-kernel roofline (HIP events) and oracle CPU baseline.
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
PEAK_BF16_MFMA_TFLOPS = 2516.8                   # dense bf16 MFMA = 16 x the f32 MFMA rate (MI355X_MICROARCH.md: ~2.5 PF)
PEAK_HBM_GBS = 8000.0
REFERENCE_PY_STEPS_PER_S = 886.0                 # the unmodified reference ppo.py, torch CPU, 1 thread, build container
PREWARM_UPDATES = 60   # ~90 ms of throwaway updates on a scratch engine before the W warm-up steps: the GPU's clocks have ramped by then (see main)
PROF_EVERY = 10        # the timed windows bracket grad_kernel's launches with HIP events in every 10th update only (see timed_updates)


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


def cpu_baseline_n1(params, min_seconds=3.0):
    """The oracle at the REFERENCE's shape: 1 env, 1 thread, 128-step rollouts, 4 x 4 minibatches of 32 rows (ppo.py:62-76)."""
    from oracle import cpu_ref as R

    base = R.PPOBaseline(params, 1, T=T, seed=1, threads=1)
    t0 = time.perf_counter()
    k = 0
    while True:
        base.run_update()
        k += 1
        dt = time.perf_counter() - t0
        if dt >= min_seconds:
            break
    R.lib().ref_set_num_threads(usable_cpus())
    v = k * T / dt
    return {"value": round(v, 1), "unit": "env-steps/s", "updates_per_s": round(k / dt, 2), "cores": 1, "kind": "port",
            "sample": "%d outer updates of ppo.py's own shape (1 env x 128 steps, minibatches of 32) in %.1f s, C oracle, 1 thread" % (k, dt),
            "reference_python_env_steps_per_s": REFERENCE_PY_STEPS_PER_S,
            "ratio_to_reference_python": round(v / REFERENCE_PY_STEPS_PER_S, 1),
            "note": "the unmodified reference ppo.py (torch CPU, 1 thread) ran 886 env-steps/s / 6.9 updates/s in the build container "
                    "(SURVEY.md §6); it cannot travel to the GPU box, the oracle at its shape stands in for it"}


# ---- BASELINE.json configs[2] and [3]: secondary workloads, reported as extra keys of the same line (N = 1 only) ----------------------
DQN_MACS_FWD = 4 * 120 + 120 * 84 + 84 * 2          # 10,728 per Q-network forward
SAC_MACS_NET = 4 * 256 + 256 * 256 + 256            # 66,816: critic (3+1 -> 256 -> 256 -> 1); the actor (3 -> 256 -> 256 -> 2 heads) has the same count


# ---- chain floors of the latency-bound configs (derivations: DESIGN.md sections 8 / 9) --------------------------------------------------------
# These loops are chains of DEPENDENT launches on a few workgroups: a fraction of the MFMA peak says nothing a builder can act on.  The floor they are measured
# against instead = (matrix cycles on the longest wave's dependent chain, per SIMD, at the clock the chip sustains under the PPO gradient launch: 2.07 GHz)
#                 + (kernel boundaries on the chain x 1.3 us: tools/ubench/boundary_gap.hip) + (dependent global-memory round trips on the chain x 0.45 us, a cold load).
# A v_mfma_f32_16x16x4_f32 holds a SIMD's matrix pipe for 32 cycles (2,048 FLOP at 64 FLOP / clk / SIMD).
SUSTAINED_GHZ = 2.07
BOUNDARY_US = 1.3
ROUND_TRIP_US = 0.45
MFMA16_CYCLES = 32


def chain_floor(name, mfma_terms, boundaries, round_trips, extra_cycles=None):
    """-> {chain_floor_us, terms}: mfma_terms = [(what, MFMAs on the chain)], extra_cycles = [(what, cycles)] for dependent non-matrix chains (PER's prefix walks)."""
    cyc = sum(n for _, n in mfma_terms) * MFMA16_CYCLES + sum(c for _, c in (extra_cycles or []))
    us = cyc / (SUSTAINED_GHZ * 1e3) + boundaries * BOUNDARY_US + round_trips * ROUND_TRIP_US
    return {"chain_floor_us": round(us, 2),
            "chain_floor_terms": {"matrix_chain": {k: "%d MFMA x %d cycles" % (n, MFMA16_CYCLES) for k, n in mfma_terms},
                                  "other_dependent_chains_cycles": dict(extra_cycles or []), "chain_cycles": cyc, "clock_GHz": SUSTAINED_GHZ,
                                  "kernel_boundaries": "%d x %.1f us" % (boundaries, BOUNDARY_US), "memory_round_trips": "%d x %.2f us" % (round_trips, ROUND_TRIP_US),
                                  "what": name}}


# dqn.py iteration at batch 128: acting = 10 dependent steps x 72 MFMAs on a forward wave (8 layer-1 + 64 layer-2); TD = one 8-row group per workgroup:
# 96 (layer 2, three unit tiles per wave) + 42 (dh1) + 24 (dW2) MFMAs on a wave; the slab sum has no matrix work.  3 launches; round trips: acting prologue, TD index ->
# rows, slab sum's loads, Adam state.
def dqn_chain_floor(batch):
    groups = 1 if batch <= 8 * 256 else -(-((batch + 15) // 16) // 256)   # row groups per workgroup (R = 8 up to 2,048 rows, then 16-row groups dealt to 256 workgroups)
    td = (96 + 42 + (24 if batch <= 8 * 256 else 48)) * groups
    return chain_floor("act (10 dependent steps) -> TD -> slab sum + Adam", [("act: 10 steps x (8 + 64)", 720), ("td: layer 2 + dh1 + dW2, %d group(s)" % groups, td)], 3, 4)


# per.py adds the sampler launch in front of the TD launch: behind a 256-step f64 chain (the level-1 running sums) and three dependent fetches a draw does a ten-step
# binary search over LDS (~100 cycles per dependent read) and walks <= 63 + 63 chunk sums / priorities (3 instructions of ~8 issue cycles each per step; a wave of 64 draws
# ends near the far end of both walks)
def per_chain_floor(batch):
    f = dqn_chain_floor(batch)
    g = chain_floor("act -> sampler (prefix-sum descent) -> TD -> slab sum + Adam + scatter", [("act: 10 steps x (8 + 64)", 720), ("td", 162)], 4, 7,
                    [("sampler: 256-step f64 running sums", 256 * 8), ("sampler: 10-step binary search", 10 * 100), ("sampler: ~126 walk steps x 3 instructions", 126 * 24)])
    return g if batch <= 2048 else f


# sac.py iteration at batch 256: a 256 x 256 pass over a 16-row group = 1,024 MFMAs on 4 waves = 256 per wave.  On the chain: acting 1 (actor layer 2;
# 8 waves, two tiles each: 128 MFMAs per wave but two waves per SIMD -> the same 8,192 cycles per SIMD), critic launch 2 (actor' forward, target forward; the critics'
# own forward and unit-weight backward run inside the wait), actor launch 4 (actor forward, critic forward, critic backward, actor backward), the actor's dW2 GEMM
# launch (256 x 256 x batch MACs over sac_dw2_blocks workgroups: ~1/4 pass per SIMD at batch 256).  4 launches with the critics' step carried by the acting launch.
def sac_chain_floor(batch):
    rg_per_wg = max(1, -(-((batch + 15) // 16) // 256))
    passes = 7 * rg_per_wg
    return chain_floor("act (+ carried critic step) -> critic -> actor -> actor's dW2 + Adam", [("%d dependent 256 x 256 passes x 256 MFMAs per wave" % passes, 256 * passes),
                                                                                                ("dW2 GEMM, per SIMD", 64 * rg_per_wg)], 4, 6)


def with_floor(out, floor):
    out.update(floor)
    out["frac_of_chain_floor"] = round(floor["chain_floor_us"] / (1e3 * out["ms_per_step"]), 4)
    return out


def sharded_synthetic(run_iters, set_forced, world=8):
    """The one-call sharded route of an off-policy engine on the P2P carrier with `world` SYNTHETIC ranks (mi_comm_p2p_synthetic: ONE process stores, polls and sums what
    a `world`-rank exchange does, minus the links — the arithmetic stays the single rank's): what the exchange and the launches only a sharded run takes (the optimizer
    steps are launches of their own there) cost per iteration, and the weak-scaling efficiency that implies before a byte crosses xGMI.  run_iters(n) -> seconds."""
    import ctypes as C

    import torch

    import deep_rl_amd.dist as DD
    from deep_rl_amd import _native as N

    h = C.c_void_p()
    try:
        N.check(N.lib().mi_comm_p2p_synthetic(world, 1 << 20, C.byref(h)), "mi_comm_p2p_synthetic")
        DD.use_comm(h)
        set_forced(True)
        run_iters(30)
        n = 200
        dt, enq = run_iters(n)
        N.check(N.lib().mi_comm_check(h), "mi_comm_check")
        return {"world": world, "ms_per_step": round(1e3 * dt / n, 5), "host_enqueue_ms_per_step": round(1e3 * enq / n, 5)}
    except Exception as ex:  # noqa: BLE001
        return {"world": world, "error": "%s: %s" % (type(ex).__name__, ex)}
    finally:
        set_forced(False)
        DD.use_comm(None)
        if h.value:
            torch.cuda.synchronize()
            N.lib().mi_comm_destroy(h)


def same_binary(profiled_id):
    """'source id X = this run's library' / '... DIFFERS from this run's library (Y)': whether a static figure quoted from profiles/ describes the binary being timed
    (mi_source_id: sha256 of the kernel sources the library was built from)."""
    try:
        from deep_rl_amd import _native as N
        mine = N.lib().mi_source_id().decode()
    except Exception:  # noqa: BLE001
        mine = "unknown"
    if not profiled_id:
        return "source id of the profiled build not recorded; this run's library: %s" % mine
    return ("source id %s = this run's library" % profiled_id) if profiled_id == mine else ("source id %s DIFFERS from this run's library (%s)" % (profiled_id, mine))


def pmc_traffic(key):
    """HBM bytes per launch of a kernel from the committed PMC passes (profiles/latest_pmc.json, written by tools/profile_round.sh: separate rocprofv3 --pmc FETCH_SIZE /
    WRITE_SIZE passes of the same workloads) -> (bytes | None, source text | None).  Static: not measured in this run."""
    pmc = os.path.join(ROOT, "profiles", "latest_pmc.json")
    try:
        j = json.load(open(pmc))
        e = j.get(key)
        if not e or e.get("hbm_bytes_per_launch") is None:
            return None, None
        return e["hbm_bytes_per_launch"], "profiles/latest_pmc.json[%s] (static: separate rocprofv3 --pmc passes, build %s, %s; not measured in this run)" % (
            key, j.get("build", "?"), same_binary(j.get("source_id")))
    except Exception:
        return None, None


def bench_dqn(dev, iters=300, cpu_seconds=3.0, batch=128, variant="dqn"):
    """dqn.py CartPole-v1, 4096 envs, 256-slot ring (1,048,576 transitions on HBM), batch 128, train every 10 steps (dqn.py:84-137).
    One step = one loop iteration: 10 env steps of every env (one launch) + sample + TD update (+ target sync every 500 steps).
    batch != 128: the SCALED-batch line SURVEY.md §8d asks for beside the reference's batch (labelled as such by the caller)."""
    import torch

    import deep_rl_amd as D
    from deep_rl_amd import _native as N

    envs, slots = 4096, 256
    env = D.make("CartPole-v1", num_envs=envs, device=dev, seed=1)
    torch.manual_seed(1)
    Net = D.DuelingQNetwork if variant == "dueling" else D.QNetwork
    Eng = {"dqn": D.DQNEngine, "dueling": D.DuelingDQNEngine, "per": D.PERDQNEngine}[variant]
    q = Net(env); t = Net(env); t.load_state_dict(q.state_dict())
    params0 = q.flat.cpu().numpy().copy()
    eng = Eng(env, q, t, D.ClipAdam(q, lr=2.5e-4, eps=1e-8), slots=slots, batch_size=batch, learning_starts=100, total_timesteps=10 * (iters + 600),
              max_episodes_logged=0)
    eng.reset()

    def it():
        eng.act(10); eng.train_step()
        if eng.global_step % 500 == 0:
            eng.sync_target()
    for _ in range(50):
        it()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        it()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if variant != "dqn":   # dueling_dqn.py / per.py on the same ring and loop: the iteration time against its chain floor, nothing else
        script = {"dueling": "dueling_dqn.py", "per": "per.py"}[variant]
        out = {"workload": "%s CartPole-v1, %d envs, %d-slot ring, batch %d, train every 10 steps" % (script, envs, slots, batch), "value": round(iters * 10 * envs / dt, 1),
               "unit": "env-steps/s", "updates_per_s": round(iters / dt, 1), "ms_per_step": round(1e3 * dt / iters, 5), "dtype": "f32", "loss": float(eng.loss.item()),
               "step": "10 env steps of every env + 1 update" + (" (one call per piece: mi_per_act_steps + mi_per_td_update, 4 launches)" if variant == "per" else
                                                                 " (mi_dueling_td_update: 2 launches)")}
        return with_floor(out, per_chain_floor(batch) if variant == "per" else dqn_chain_floor(batch))
    # kernel durations from a separate, shorter pass: the event pairs around every launch would otherwise sit inside the timed loop
    N.prof_begin(4 * 100 + 8, tags=["dqn_act", "dqn_td", "dqn_reduce"])
    for _ in range(100):
        it()
    prof = N.prof_end()
    us = {k: 1e3 * v[0] / max(v[1], 1) for k, v in prof.items() if v[1]}
    # algorithmic FLOPs: acting = one forward per env-step (the greedy branch; exploring rows skip it, so this is an upper bound of the
    # work and the fraction below an upper bound too); TD = target forward + online forward + ~2x backward per batch row
    act_flops = 2 * DQN_MACS_FWD * envs * 10
    td_flops = 2 * DQN_MACS_FWD * 4 * batch
    act_traffic, act_src = pmc_traffic("dqn_act4_kernel")
    td_traffic, td_src = pmc_traffic("dqn_td_kernel@%d" % batch)
    out = {"workload": "dqn.py CartPole-v1, %d envs, %d-slot ring (%d transitions on HBM), batch %d, train every 10 steps" % (envs, slots, envs * slots, batch),
           "value": round(iters * 10 * envs / dt, 1), "unit": "env-steps/s", "updates_per_s": round(iters / dt, 1), "ms_per_step": round(1e3 * dt / iters, 5),
           "step": "10 env steps of every env + 1 TD update", "dtype": "f32", "kernel_us": {k: round(v, 2) for k, v in us.items()},
           "roofline": {"bound": "mfma", "kernel": "dqn_act4_kernel", "achieved": round(act_flops / (us["dqn_act"] * 1e-6) / 1e12, 3), "peak": PEAK_F32_MFMA_TFLOPS,
                        "unit": "TFLOP/s", "frac": round(act_flops / (us["dqn_act"] * 1e-6) / 1e12 / PEAK_F32_MFMA_TFLOPS, 4), "traffic": act_traffic, "traffic_source": act_src,
                        "flops_per_launch": act_flops, "avg_launch_us": round(us["dqn_act"], 2),
                        "note": "10 dependent env steps per launch: 16 envs per workgroup (3 forward waves + 1 wave computing both CartPole successors), latency-bound chain (forward, barrier, argmax)"},
           "roofline_td": {"bound": "latency (f32 VALU peak quoted)", "kernel": "dqn_td_kernel", "achieved": round(td_flops / (us["dqn_td"] * 1e-6) / 1e12, 4),
                           "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": round(td_flops / (us["dqn_td"] * 1e-6) / 1e12 / PEAK_F32_MFMA_TFLOPS, 5),
                           "traffic": td_traffic, "traffic_source": td_src, "flops_per_launch": td_flops, "avg_launch_us": round(us["dqn_td"], 2),
                           "note": ("the reference's batch of 128 rows is 16 workgroups on a 256-CU chip: the launch is a fixed ~12 us latency, not a throughput" if batch == 128 else
                                    "%d rows = %d workgroups of 8 rows" % (batch, batch // 8))},
           "loss": float(eng.loss.item())}
    if batch == 128:   # (the scaled batch is a throughput, not a chain: its MFMA fraction is the informative figure)
        with_floor(out, dqn_chain_floor(batch))
        out["roofline"]["note"] += "; a fraction of the MFMA peak is uninformative for a dependent launch chain: see chain_floor_us / frac_of_chain_floor"
    if batch == 128:   # what the sharded form of this loop costs on one GPU (the exchange + clip / Adam as a launch of its own), 8 synthetic ranks on the P2P carrier
        import deep_rl_amd.dqn_engine as DE

        def run_iters(n):
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(n):
                it()
            enq = time.perf_counter() - t1   # (host time to enqueue the loop: when it is the whole of dt the sharded loop is host-bound, not GPU-bound)
            torch.cuda.synchronize()
            return time.perf_counter() - t1, enq

        sh = sharded_synthetic(run_iters, lambda on: setattr(DE, "_FORCE_SHARDED", on))
        if "ms_per_step" in sh:
            sh["efficiency_model"] = round(out["ms_per_step"] / sh["ms_per_step"], 3)
            sh["note"] = ("mi_dqn_td_update_sharded on the P2P carrier, ONE process playing 8 ranks (every store, poll and rank-ordered add of an 8-rank exchange of the "
                          "10,935-float {gradient, loss} message, minus the links): the TD launch draws its batch itself and the slab-sum launch carries the exchange AND "
                          "Adam (round 6: two launches, as in a single process; until then five: 68 us, efficiency 0.55); efficiency_model = single-rank iteration / this "
                          "(weak scaling; real xGMI adds one link latency, ~1 - 2 us, per iteration)")
        out["sharded_synthetic"] = sh
    out["cpu_baseline"] = cpu_baseline_dqn(params0, envs, slots, batch, cpu_seconds)
    return out


def cpu_baseline_dqn(params0, envs, slots, batch, min_seconds):
    """The same loop iteration from the oracle's primitives (ref_dqn_act_steps, ref_dqn_sample, ref_dqn_td_grads, Adam), OpenMP over envs / rows."""
    import numpy as np

    from oracle import cpu_ref as R

    cores = usable_cpus()
    R.lib().ref_set_num_threads(cores)
    env = R.VecCartPole(envs, seed=1)
    st = R.ReplayStorage(slots, envs)
    obs = env.reset()
    p = params0.copy(); tp = params0.copy(); m = np.zeros_like(p); v = np.zeros_like(p)
    gs, k = 0, 0
    t0 = time.perf_counter()
    while True:
        R.dqn_act_steps(env, p, st, obs, 10, gs, learning_starts=100, total_timesteps=10 ** 6)
        gs += 10
        idx = R.dqn_sample(1, k, min(gs, slots) * envs, batch)
        g, _ = R.dqn_td_grads(p, tp, st, idx)
        k += 1
        R.adam_step(p, g, m, v, k, 2.5e-4, eps=1e-8)
        if gs % 500 == 0:
            tp[...] = p
        dt = time.perf_counter() - t0
        if dt >= min_seconds:
            break
    return {"value": round(gs * envs / dt, 1), "unit": "env-steps/s", "updates_per_s": round(k / dt, 2), "cores": cores, "kind": "port",
            "sample": "%d loop iterations (10 steps x %d envs + 1 update each) in %.1f s, C oracle, OpenMP over envs / rows" % (k, envs, dt)}


def bench_sac(dev, iters=400, cpu_seconds=3.0, batch=256):
    """sac.py on Pendulum-v1, 2048 envs, 512-slot ring, batch 256 (sac.py:137-217): one step = one env step of every env + one critic
    update (+ polyak) + one actor + one alpha update (the reference does two of each every 2nd step).
    batch != 256: the SCALED-batch line SURVEY.md §8d asks for beside the reference's batch (labelled as such by the caller)."""
    import torch

    import deep_rl_amd as D
    from deep_rl_amd import _native as N

    envs, slots = 2048, 512
    env = D.make("Pendulum-v1", num_envs=envs, device=dev, seed=1)
    torch.manual_seed(1)
    actor = D.Actor(env)
    qs = [D.SoftQNetwork(env) for _ in range(4)]
    qs[2].load_state_dict(qs[0].state_dict()); qs[3].load_state_dict(qs[1].state_dict())
    eng = D.SACEngine(env, actor, *qs, slots=slots, batch_size=batch, learning_starts=20, max_episodes_logged=0)
    a0, q0 = actor.flat.cpu().numpy().copy(), eng.q_flat.cpu().numpy().copy()
    eng.reset()

    def it():
        eng.act()
        if eng.global_step >= eng.learning_starts:
            eng.train_step()
    for _ in range(60):
        it()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        it()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    tags = ["sac_act", "sac_critic", "sac_actor", "sac_gemm", "sac_assemble", "sac_logp"]
    N.prof_begin(12 * 100 + 16, tags=tags)   # separate pass: see bench_dqn
    for _ in range(100):
        it()
    prof = N.prof_end()
    us = {k: 1e3 * v[0] / max(v[1], 1) for k, v in prof.items() if v[1]}
    per_it = {k: round(1e3 * v[0] / 100, 2) for k, v in prof.items() if v[1]}
    # sac_critic_kernel per batch row: actor forward on the next observation, two target-Q forwards, two critic forwards + ~2x backward
    critic_flops = 2 * SAC_MACS_NET * (1 + 2 + 2 * 3) * batch
    cr_traffic, cr_src = pmc_traffic("sac_critic_kernel@%d" % batch)
    nrg = (batch + 15) // 16
    shape_note = ("batch 256 = 16 row groups x 4 workgroups (two target roles, two critic roles) on a 256-CU chip: 3 dependent 256x256 passes + one hand-off on the critical path, "
                  "latency-bound (DESIGN.md §7c)") if batch == 256 else "%d row groups of 16 rows, one workgroup each (7 dependent 256x256 passes per workgroup)" % nrg
    out = {"workload": "sac.py Pendulum-v1, %d envs, %d-slot ring (%d transitions on HBM), batch %d, 1 critic + 1 actor + 1 alpha update per time step" % (
               envs, slots, envs * slots, batch),
           "value": round(iters * envs / dt, 1), "unit": "env-steps/s", "updates_per_s": round(iters / dt, 1), "ms_per_step": round(1e3 * dt / iters, 5),
           "step": "1 env step of every env + critic update + polyak + actor update + alpha update", "dtype": "f32",
           "kernel_us": {k: round(v, 2) for k, v in us.items()}, "kernel_us_per_step": per_it,
           "kernel_us_note": "sac_act includes the critics' deferred optimizer step (dW2 GEMM + Adam + polyak on extra workgroups of the acting launch: MIRL_SAC_DEFER_CRITIC=1, "
                             "the default), which therefore does not appear under sac_gemm",
           "roofline": {"bound": "mfma", "kernel": "sac_critic_kernel", "achieved": round(critic_flops / (us["sac_critic"] * 1e-6) / 1e12, 3),
                        "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": round(critic_flops / (us["sac_critic"] * 1e-6) / 1e12 / PEAK_F32_MFMA_TFLOPS, 4),
                        "traffic": cr_traffic, "traffic_source": cr_src, "flops_per_launch": critic_flops, "avg_launch_us": round(us["sac_critic"], 2),
                        "note": shape_note},
           "alpha": float(eng.alpha), "q_losses": [round(float(x), 5) for x in eng.q_losses.tolist()]}
    if batch == 256:
        with_floor(out, sac_chain_floor(batch))
        out["roofline"]["note"] += "; a fraction of the MFMA peak is uninformative for a dependent launch chain: see chain_floor_us / frac_of_chain_floor"
    if batch == 256:
        import deep_rl_amd.sac_engine as SE

        def run_iters(n):
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(n):
                it()
            enq = time.perf_counter() - t1
            eng.flush()
            torch.cuda.synchronize()
            return time.perf_counter() - t1, enq

        eng.flush()
        sh = sharded_synthetic(run_iters, lambda on: setattr(SE, "_FORCE_SHARDED", on))
        if "ms_per_step" in sh:
            sh["efficiency_model"] = round(out["ms_per_step"] / sh["ms_per_step"], 3)
            sh["note"] = ("mi_sac_{critic,actor}_update_sharded + mi_sac_alpha_step_sharded on the P2P carrier, ONE process playing 8 ranks: three exchanges per iteration "
                          "(134,660 / 67,332 floats and one scalar: up to 539 KB), each INSIDE the launch that assembles the gradient and steps it (round 6: Adam / polyak / "
                          "alpha ride behind the exchange; until then 14 launches: 123 us, efficiency 0.60); no deferred or owed steps on this route, and its seven calls "
                          "per iteration make the loop nearly host-bound (host_enqueue_ms_per_step); efficiency_model = single-rank iteration / this (weak scaling; real "
                          "xGMI adds three link latencies per iteration)")
        out["sharded_synthetic"] = sh
    out["cpu_baseline"] = cpu_baseline_sac(a0, q0, envs, slots, batch, cpu_seconds)
    return out


def cpu_baseline_sac(actor0, q0, envs, slots, batch, min_seconds):
    """The same loop iteration from the oracle's primitives (Pendulum step, actor sample, critic / actor / alpha gradients, Adam, polyak)."""
    import numpy as np

    from oracle import cpu_ref as R

    cores = usable_cpus()
    R.lib().ref_set_num_threads(cores)
    rng = np.random.default_rng(1)
    env = R.VecPendulum(envs, seed=1)
    st = R.SacStorage(slots, envs)
    obs = env.reset()
    a_p = actor0.copy(); q_p = q0.copy(); qt_p = q0.copy()
    am, av = np.zeros_like(a_p), np.zeros_like(a_p)
    qm, qv = np.zeros_like(q_p), np.zeros_like(q_p)
    la = np.zeros(1, np.float32); lm = np.zeros(1, np.float32); lv = np.zeros(1, np.float32)
    gs, k = 0, 0
    st.observations[0] = obs
    t0 = time.perf_counter()
    while True:
        act, _ = R.sac_actor_sample(a_p, obs, rng.standard_normal(envs).astype(np.float32))
        obs, rew, done, _, _ = env.step(act)
        st.actions[gs % slots] = act
        gs += 1
        st.observations[gs % slots] = obs; st.rewards[gs % slots] = rew; st.terminated[gs % slots] = 0
        if gs >= 20:
            alpha = float(np.exp(la[0]))
            idx = R.dqn_sample(1, k, min(gs, slots) * envs, batch)
            g, _ = R.sac_critic_grads(q_p, qt_p, a_p, st, idx, rng.standard_normal(batch).astype(np.float32), alpha)
            k += 1
            R.adam_step(q_p, g, qm, qv, k, 1e-3, eps=1e-8)
            ga, _, _ = R.sac_actor_grads(a_p, q_p, st, idx, rng.standard_normal(batch).astype(np.float32), alpha)
            R.adam_step(a_p, ga, am, av, k, 3e-4, eps=1e-8)
            mlp = R.sac_mean_logp(a_p, st, idx, rng.standard_normal(batch).astype(np.float32))
            R.adam_step(la, np.array([-(mlp - 1.0)], np.float32), lm, lv, k, 1e-3, eps=1e-8)
            R.polyak(qt_p, q_p)
        dt = time.perf_counter() - t0
        if dt >= min_seconds:
            break
    return {"value": round(gs * envs / dt, 1), "unit": "env-steps/s", "updates_per_s": round(k / dt, 2), "cores": cores, "kind": "port",
            "sample": "%d loop iterations (1 step x %d envs + critic / actor / alpha update each) in %.1f s, C oracle, OpenMP over envs / rows" % (gs, envs, dt)}


def kernel_device_durations(window_ms):
    """Per-kernel DEVICE time per update from the committed rocprofv3 kernel trace of the headline command (profiles/latest_kernel_durations.json, written by
    tools/make_latest_durations.py): sum of End - Start of each kernel's launches between two rollouts, the update's span and launch_gaps = span - sum, all from ONE
    profiled run so that they add up.  Static (not measured by this run) and labelled so; this run's own window stands beside it."""
    try:
        d = json.load(open(os.path.join(ROOT, "profiles", "latest_kernel_durations.json")))
        return {"kernels": {n: v["ms_per_update"] for n, v in d["ppo_update"].items()}, "sum_ms": d["sum_ms"], "span_ms_profiled": d["span_ms"], "launch_gaps_ms": d["launch_gaps_ms"],
                "launches_per_update": d.get("launches_per_update"), "this_run_window_ms": round(window_ms, 4),
                "source": "%s, build %s (%s), %s updates averaged — static: device durations, span and gaps of the SAME profiled run (they add up; HIP-event brackets cannot)"
                          % (d.get("source", "?"), d.get("build", "?"), same_binary(d.get("source_id")), d.get("updates_averaged", "?"))}
    except Exception as ex:  # noqa: BLE001
        return {"error": "profiles/latest_kernel_durations.json: %s" % ex}


def collectives_label(world, is_p2p, native, carriers, policy):
    """config.collectives: which carrier the HEADLINE windows ran on, and — when that is not the RCCL configuration BASELINE.json names — why not."""
    if world == 1 and not carriers:
        return "none (single process)"
    why_not_rccl = carriers.get("rccl", (None, {}))[1].get("why")
    if native and not is_p2p:
        return "RCCL direct (mi_ppo_update_sharded: one C call per update, 17 in-stream ncclAllReduce over xGMI); the P2P carrier is reported beside it (value_p2p)"
    tail = ("RCCL could not carry this run: %s" % why_not_rccl) if why_not_rccl else ("policy: %s" % policy)
    if native:
        return ("P2P over hipIpc inboxes (mi_ppo_update_sharded: one C call per update, the gradient exchange inside the slab-sum launch — a launch of its own when more "
                "than two ranks share a device) — NOT the RCCL configuration; " + tail)
    return "torch.distributed (host-sequenced, 17 all-reduces per update) — NOT the RCCL configuration; " + tail


def rendezvous_port():
    """A free TCP port on 127.0.0.1 OUTSIDE the kernel's ephemeral range (probed by a bind): a number drawn with bind(0) comes from the range every outgoing connection
    takes its source port from, and can be gone again by the time the job's store listens on it (EADDRINUSE: seen once in the GPU suite, round 6)."""
    import random
    import socket

    try:
        low = int(open("/proc/sys/net/ipv4/ip_local_port_range").read().split()[0])
    except (OSError, ValueError, IndexError):
        low = 32768
    lo = max(10000, low - 12000)
    for _ in range(200):
        port = random.randrange(lo, low) if low > lo else random.randrange(10000, 30000)
        with socket.socket() as sk:
            try:
                sk.bind(("127.0.0.1", port))
            except OSError:
                continue
            return port
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def self_launch(n, argv):
    """`python bench.py --gpus N` with N > 1: start the N ranks as a child process and relay rank 0's line.  The parent makes no GPU call
    and never replaces itself (a process that has initialised HIP must not exec; this one has not even imported torch)."""
    import socket
    import subprocess

    port = rendezvous_port()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # the pool's driver only supports dmabuf IPC (RCCL / CUDA-tensor sharing)
    env.setdefault("OMP_NUM_THREADS", "1")              # what torch.distributed.run would set itself (with a warning)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.abspath(__file__)] + list(argv)
    child = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True, cwd=ROOT)
    lines = 0
    for line in child.stdout:               # rank 0 prints exactly one JSON line; anything else a rank writes to stdout goes to our stderr
        if line.startswith('{"metric"'):
            sys.stdout.write(line); sys.stdout.flush()
            lines += 1
        else:
            sys.stderr.write(line)
    rc = child.wait()
    if rc == 0 and lines != 1:
        sys.stderr.write("bench.py: expected one JSON line from rank 0, saw %d\n" % lines)
        return 1
    return rc


def collective_diagnostics(eng, comm, world, dev, breakdown, n_break):
    """What the sharded update's 17 collectives cost, so that a scaling run carries its own diagnosis:
      in_update   the in-stream ncclAllReduce calls bracketed by HIP events INSIDE the update (mi_prof tags comm_grad / comm_stats; native route only):
                  includes waiting for the slowest rank's gradient launch, i.e. the exposed cost on the critical path
      back_to_back  100 all-reduces of the same 9,159-float buffer on an otherwise idle stream through the carrier in use: the collective's own latency"""
    import ctypes as C

    import torch

    from deep_rl_amd import _native as N
    from deep_rl_amd import dist as DD

    out = {"per_update": 17, "grad_allreduce": {"count_per_update": 16, "elements": N.NPARAMS + 4, "bytes": 4 * (N.NPARAMS + 4), "dtype": "f32"},
           "stats_allreduce": {"count_per_update": 1, "elements": eng._adv_sums_all.numel(), "bytes": 8 * eng._adv_sums_all.numel(), "dtype": "f64"}}
    # Everything that can fail on ONE rank (communicator queries, allocations) happens before the first collective of this function, then the ranks agree on going
    # on: a rank that raised part-way through ~220 collectives would leave the others blocked in ncclAllReduce until the driver's timeout.
    ok, err = 1, None
    try:
        if comm is not None:
            ws, rk, ver, cnt = C.c_int(), C.c_int(), C.c_int(), C.c_int()
            N.check(N.lib().mi_comm_info(comm, C.byref(ws), C.byref(rk), C.byref(ver), C.byref(cnt)), "mi_comm_info")
            out["carrier"] = ("P2P (libmirl mi_comm over hipIpc-mapped inboxes: one launch per all-reduce, rank-ordered sum, inside ONE C call per update)"
                              if N.lib().mi_comm_carrier(comm) == 1 else "RCCL direct (rccl.h via libmirl mi_comm, in-stream ncclAllReduce inside ONE C call per update)")
            out["rccl_version"] = ver.value
            out["rccl_comm_count"] = cnt.value
            g, st = breakdown.get("comm_grad", (0.0, 0)), breakdown.get("comm_stats", (0.0, 0))
            out["in_update"] = {"us_per_allreduce_grad": round(1e3 * g[0] / max(g[1], 1), 2), "us_per_allreduce_stats": round(1e3 * st[0] / max(st[1], 1), 2),
                                "ms_per_update": round((g[0] + st[0]) / max(n_break, 1), 4), "samples": [g[1], st[1]],
                                "note": "HIP events around each in-stream ncclAllReduce inside the update: includes the wait for the slowest rank (exposed cost)"}
        else:
            out["carrier"] = "torch.distributed %s (host-sequenced: 17 all_reduce calls between the launches)" % torch.distributed.get_backend()
        buf, sums = torch.zeros(N.NPARAMS + 4, device=dev), torch.zeros_like(eng._adv_sums_all)
    except Exception as ex:  # noqa: BLE001
        ok, err = 0, "%s: %s" % (type(ex).__name__, ex)
    flag = torch.tensor([ok], dtype=torch.int32, device=dev)
    torch.distributed.all_reduce(flag, op=torch.distributed.ReduceOp.MIN)
    if int(flag.item()) == 0:
        out["error"] = err or "another rank failed before the back-to-back measurement; skipped on every rank"
        out["world_size"] = world
        return out

    def once(t, dtype):
        if comm is not None:
            N.check(N.lib().mi_comm_allreduce_sum(comm, N.ptr(t), t.numel(), dtype, N.stream_ptr(dev)), "mi_comm_allreduce_sum")
        else:
            DD.allreduce_sum_(t, eng.pg)
    b2b = {}
    for name, t, dtype in (("us_per_allreduce_grad", buf, 0), ("us_per_allreduce_stats", sums, 1)):
        for _ in range(10):
            once(t, dtype)
        torch.cuda.synchronize(); torch.distributed.barrier(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(100):
            once(t, dtype)
        torch.cuda.synchronize()
        b2b[name] = round(1e6 * (time.perf_counter() - t0) / 100, 2)
    b2b["note"] = "100 back-to-back all-reduces on an idle stream, wall clock / 100, this rank"
    out["back_to_back"] = b2b
    out["world_size"] = world
    return out


def carrier_legs(eng, world, dev, one_update, timed_updates, u0, steps):
    """The same K updates on EACH carrier of libmirl's communicator: RCCL (in-stream ncclAllReduce) and P2P (one launch per all-reduce over
    hipIpc-mapped inboxes, rank-ordered sum).  Per carrier: a plain timed window, a short window with HIP events around every in-stream collective (in_update: includes
    the wait for the slowest rank), 200 back-to-back all-reduces on an idle stream, and whether the replicas are still bitwise identical afterwards.  Runs AFTER the
    headline measurement; every step is collective and agreed on, a failing carrier costs its own entry only."""
    import ctypes as C

    import torch

    import deep_rl_amd.dist as DD
    import deep_rl_amd.engine as E
    from deep_rl_amd import _native as N

    res = {}
    for which in ("rccl", "p2p"):
        entry = {}
        try:
            comm = DD.native_comm(eng.pg, which=which)
        except Exception as ex:  # noqa: BLE001
            comm, entry = None, {"error": "%s: %s" % (type(ex).__name__, ex)}
        if comm is None:
            entry.setdefault("error", "no %s communicator (process group %s)" % (which, torch.distributed.get_backend()))
            res[which] = entry
            continue
        try:
            DD.use_comm(comm)
            E._FORCE_NATIVE_SHARDED = True
            for _ in range(2):
                one_update(u0)
            dt, _ = timed_updates(u0, steps)
            n_ev = min(steps, 10)
            _, prof = timed_updates(u0, n_ev, prof_every=1, tags=("comm_grad", "comm_stats", "reduce"))
            g, st, rd = prof["comm_grad"], prof["comm_stats"], prof["reduce"]
            entry = {"ms_per_step": round(1e3 * dt / steps, 4),
                     "in_update": {"us_per_allreduce_grad": round(1e3 * g[0] / max(g[1], 1), 2) if g[1] else None, "us_per_allreduce_stats": round(1e3 * st[0] / max(st[1], 1), 2),
                                   "us_per_slab_sum": round(1e3 * rd[0] / max(rd[1], 1), 2), "ms_per_update": round((g[0] + st[0] + rd[0]) / n_ev, 4), "samples": [g[1], st[1], rd[1]],
                                   "note": "p2p: the gradient all-reduce happens INSIDE the slab-sum launch (us_per_slab_sum includes the exchange and the wait for the slowest "
                                           "rank; no all-reduce launch); rccl: slab sum, then an in-stream ncclAllReduce"}}
            buf = torch.zeros(N.NPARAMS + 4, device=dev)
            for _ in range(20):
                N.check(N.lib().mi_comm_allreduce_sum(comm, N.ptr(buf), buf.numel(), 0, N.stream_ptr(dev)), "mi_comm_allreduce_sum")
            torch.cuda.synchronize(); torch.distributed.barrier(); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(200):
                N.check(N.lib().mi_comm_allreduce_sum(comm, N.ptr(buf), buf.numel(), 0, N.stream_ptr(dev)), "mi_comm_allreduce_sum")
            torch.cuda.synchronize()
            entry["back_to_back_us_per_allreduce_grad"] = round(1e6 * (time.perf_counter() - t0) / 200, 2)
            try:
                if world > 1:
                    eng.check_replicas()
                else:
                    DD.check_native_comm(eng.pg)
                    N.check(N.lib().mi_comm_check(comm), "mi_comm_check")
                entry["replicas_identical"] = True
            except Exception as ex:  # noqa: BLE001
                entry["replicas_identical"] = False
                entry["error"] = "%s: %s" % (type(ex).__name__, ex)
        except Exception as ex:  # noqa: BLE001
            entry["error"] = "%s: %s" % (type(ex).__name__, ex)
        finally:
            E._FORCE_NATIVE_SHARDED = False
            DD.use_comm(None)
        res[which] = entry
    res["note"] = ("in_update: HIP events around each in-stream all-reduce inside the update (includes the wait for the slowest rank = the exposed cost); back_to_back: 200 "
                   "all-reduces of the 9,159-float gradient buffer on an idle stream, wall clock / 200, this rank; the headline windows run on RCCL whenever it passed its probe (carrier_choice.policy), the P2P carrier's "
                   "own windows stand beside them (value_p2p)")
    return res


def tune_carrier(eng, dev, n=15):
    """MIRL_COMM=auto set by the caller, second stage (runs on the THROWAWAY prewarm engine): when both carriers passed the library's known-answer probe,
    the same window of `n` real sharded updates is timed on each (MAX over ranks) and the faster one becomes the choice — the probe compares stand-alone all-reduces,
    but on the P2P carrier PPO's sixteen gradient exchanges per update ride INSIDE the slab-sum launches.  Every step is collective; a carrier whose window leaves a
    timed-out wait or diverged replicas behind is not eligible."""
    import torch

    import deep_rl_amd.dist as DD
    import deep_rl_amd.engine as E

    DD.native_comm(eng.pg)                      # first use: the probe (collective)
    rep = DD.carrier_report(eng.pg)
    if rep is None:
        return None
    cands = [w for w in ("p2p", "rccl") if rep.get(w, {}).get("ok")]
    if len(cands) < 2:
        return None
    multi = torch.distributed.get_world_size() > 1
    res, rounds = {w: float("inf") for w in cands}, {w: [] for w in cands}
    dead = set()
    for rnd in range(2):                        # alternating, best of two windows per carrier: a window's place in the sequence must not decide
        for w in cands:
            if w in dead:
                continue
            comm = DD.native_comm(eng.pg, which=w)
            ok, ms = 1, float("inf")
            try:
                DD.use_comm(comm)
                E._FORCE_NATIVE_SHARDED = True
                for _ in range(3):
                    eng.update()
                torch.cuda.synchronize(); torch.distributed.barrier(); torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(n):
                    eng.update()
                torch.cuda.synchronize()
                ms = 1e3 * (time.perf_counter() - t0) / n
                if multi:
                    eng.check_replicas()
                else:
                    DD.check_native_comm(eng.pg)
            except Exception as ex:  # noqa: BLE001
                ok = 0
                sys.stderr.write("bench.py: carrier %s left the tuning window with %s: %s\n" % (w, type(ex).__name__, ex))
            finally:
                E._FORCE_NATIVE_SHARDED = False
                DD.use_comm(None)
            t = torch.tensor([ms if ok else float("inf")], dtype=torch.float64, device=dev)
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
            v = float(t.item())
            if v == float("inf"):
                dead.add(w)
                res[w] = v
            else:
                rounds[w].append(round(v, 4))
                res[w] = min(res[w], v)
    best = min(res, key=res.get)
    if res[best] == float("inf"):
        return {"error": "no carrier finished its tuning window", "ms_per_update": {k: None for k in res}}
    how = {"what": "windows of %d real sharded updates of the throwaway prewarm engine, two per carrier, alternating; wall clock, MAX over ranks, best window" % n,
           "ms_per_update": {k: (round(v, 4) if v != float("inf") else None) for k, v in res.items()}, "windows": rounds}
    DD.set_auto_choice(eng.pg, best, how)
    return how


def sharded_route_leg(eng, one_update, timed_updates, u0, steps, dev, base_ms, base_grad_us):
    """What the launches that ONLY a multi-GPU run takes cost, measured on this one GPU (VERDICT r03 missing #2): the same K updates
      (a) with mi_ppo_test_assume_sharded(1): the owed optimizer steps recompute the clip coefficient from the 9,155 gradients (norm_parts = nullptr) — the prologue every
          rank of a sharded run executes;
      (b) the same on mi_ppo_update_sharded over a REAL one-rank RCCL communicator: 17 in-stream ncclAllReduce per update on top.
    Deltas are against a plain window timed right in front of the legs; per optimizer step = / 16.  Untimed for the headline: runs after it."""
    import socket

    import torch

    import deep_rl_amd.dist as DD
    import deep_rl_amd.engine as E

    out = {"note": "single GPU; (a) the world_size > 1 form of the owed clip + Adam step without a collective, (b) the same through mi_ppo_update_sharded on a one-rank RCCL "
                   "communicator (17 in-stream all-reduces per update); deltas vs a plain window timed right in front of the legs, per optimizer step = / 16"}
    # the baseline of the deltas: one more plain window right here (same learning state, same clocks as the two legs behind it — the headline's first window is up to
    # 10 us per update away from later ones, which is the size of the effect being measured)
    dt0, prof0 = timed_updates(u0, steps)
    base_ms, base_grad_us = 1e3 * dt0 / steps, 1e3 * prof0["grad"][0] / max(prof0["grad"][1], 1)
    out["baseline"] = {"ms_per_step": round(base_ms, 4), "grad_kernel_avg_launch_us": round(base_grad_us, 2), "what": "a plain window of the same K updates right in front of the legs"}
    E.set_assume_sharded(True)
    try:
        for _ in range(2):
            one_update(u0)
        dt, prof = timed_updates(u0, steps)
        g_us = 1e3 * prof["grad"][0] / max(prof["grad"][1], 1)
        out["assume_sharded"] = {"ms_per_step": round(1e3 * dt / steps, 4), "grad_kernel_avg_launch_us": round(g_us, 2),
                                 "delta_us_per_update": round(1e3 * (1e3 * dt / steps - base_ms), 1), "delta_us_per_optimizer_step": round(1e3 * (1e3 * dt / steps - base_ms) / 16, 2),
                                 "grad_kernel_delta_us": round(g_us - base_grad_us, 2)}
        # (c) the P2P carrier with world = 2 / 4 / 8 SYNTHETIC ranks (mi_comm_p2p_synthetic: this process stores its share into slot 0 and zeros into the other world - 1
        # slots of its own inbox, publishes and polls world flags, sums world slots — results unchanged): what the one-launch exchange costs per optimizer step before a byte
        # crosses xGMI, and how it grows with the number of slots summed
        import ctypes as C

        from deep_rl_amd import _native as N

        p2p = {"note": "mi_ppo_update_sharded on the P2P carrier, ONE process playing `world` ranks into its own inbox: grad_reduce_kernel stores each summed gradient element "
                       "as a line into `world` slots, polls `world` lines and adds them in rank order (no launch per all-reduce, no link traffic here); the one stats all-reduce per "
                       "update is a launch of its own; back_to_back = 200 stand-alone all-reduces of the 9,159-float buffer on an idle stream, wall clock / 200"}
        for w in (2, 4, 8):
            h = C.c_void_p()
            try:
                N.check(N.lib().mi_comm_p2p_synthetic(w, 1 << 20, C.byref(h)), "mi_comm_p2p_synthetic")
                DD.use_comm(h)
                E.set_assume_sharded(False)   # on this carrier grad_reduce_kernel exchanges the gradient itself: the owed steps take the single-rank branch by design
                E._FORCE_NATIVE_SHARDED = True
                for _ in range(2):
                    one_update(u0)
                dt, prof = timed_updates(u0, steps)
                E._FORCE_NATIVE_SHARDED = False
                buf = torch.zeros(N.NPARAMS + 4, device=dev)
                for _ in range(20):
                    N.check(N.lib().mi_comm_allreduce_sum(h, N.ptr(buf), buf.numel(), 0, N.stream_ptr(dev)), "mi_comm_allreduce_sum")
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(200):
                    N.check(N.lib().mi_comm_allreduce_sum(h, N.ptr(buf), buf.numel(), 0, N.stream_ptr(dev)), "mi_comm_allreduce_sum")
                torch.cuda.synchronize()
                b2b = 1e6 * (time.perf_counter() - t0) / 200
                N.check(N.lib().mi_comm_check(h), "mi_comm_check")
                p2p["world%d" % w] = {"ms_per_step": round(1e3 * dt / steps, 4), "delta_us_per_update": round(1e3 * (1e3 * dt / steps - base_ms), 1),
                                      "delta_us_per_optimizer_step": round(1e3 * (1e3 * dt / steps - base_ms) / 16, 2), "back_to_back_us_per_allreduce": round(b2b, 2)}
            except Exception as ex:  # noqa: BLE001
                p2p["world%d" % w] = {"error": "%s: %s" % (type(ex).__name__, ex)}
            finally:
                E._FORCE_NATIVE_SHARDED = False
                E.set_assume_sharded(True)
                DD.use_comm(None)
                if h.value:
                    torch.cuda.synchronize()
                    N.lib().mi_comm_destroy(h)
        out["p2p_synthetic"] = p2p
        made_pg = False
        if not (torch.distributed.is_available() and torch.distributed.is_initialized()):
            torch.distributed.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % rendezvous_port(), rank=0, world_size=1, device_id=dev)
            made_pg = True
        try:
            comm = DD.native_comm(eng.pg)
            if comm is None:
                out["rccl_world1"] = {"error": "no native RCCL communicator"}
            else:
                E._FORCE_NATIVE_SHARDED = True
                for _ in range(2):
                    one_update(u0)
                dt, prof = timed_updates(u0, steps)
                E._FORCE_NATIVE_SHARDED = False
                g_us = 1e3 * prof["grad"][0] / max(prof["grad"][1], 1)
                out["rccl_world1"] = {"ms_per_step": round(1e3 * dt / steps, 4), "grad_kernel_avg_launch_us": round(g_us, 2),
                                      "delta_us_per_update": round(1e3 * (1e3 * dt / steps - base_ms), 1),
                                      "delta_us_per_optimizer_step": round(1e3 * (1e3 * dt / steps - base_ms) / 16, 2)}
        finally:
            E._FORCE_NATIVE_SHARDED = False
            if made_pg:
                DD.destroy_native_comms()
                torch.distributed.destroy_process_group()
    finally:
        E.set_assume_sharded(False)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--headline-only", action="store_true", help="skip the config-3 (DQN) / config-4 (SAC) extra keys, the bf16x3 variant and the sharded-route leg")
    ap.add_argument("--single-window", action="store_true", help="only the one timed window of K steps (no repeat windows)")
    ap.add_argument("--no-prewarm", action="store_true", help="skip the clock ramp (PREWARM_UPDATES throwaway updates on a scratch engine before the W warm-up steps)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args.gpus, sys.argv[1:]))

    # stdout carries exactly ONE line, the JSON line: anything a library prints there (RCCL's version banner at communicator creation goes to C stdout) is sent to stderr
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    import torch

    import deep_rl_amd as D
    from deep_rl_amd import _native as N
    from deep_rl_amd.dist import init_from_env

    # production: RCCL ("nccl"), one rank per GPU.  MIRL_BENCH_BACKEND=gloo MIRL_BENCH_ONE_GPU=1 lets the N > 1 code path of this script be
    # exercised on a single-GPU box (both ranks on cuda:0; tests/test_gpu_script.py) — RCCL itself refuses two ranks on one device.
    # N > 1, HEADLINE POLICY: `value` / `ms_per_step` / `config.collectives` are measured on RCCL — BASELINE.json configs[4] and north_star say
    # "RCCL grad all-reduce over xGMI" — whenever an RCCL communicator can be created and passes its known-answer probe; libmirl's P2P carrier is measured BESIDE it
    # (value_p2p, ms_per_step_p2p, timed_windows_p2p, replicas_identical_p2p, best_carrier / value_best_carrier: its own three windows), never instead of it.  Only when
    # RCCL cannot carry the run (a gloo group on a one-GPU box, a failed probe) does `value` fall to the carrier that ran, and config.collectives says so with the
    # reason.  MIRL_COMM=p2p / auto set by the CALLER overrides the policy (and the line says which carrier it measured).
    # (an N-rank run on a node with fewer than N GPUs is refused here, before any rendezvous, with the reason: ranks that time-share a device would print a scaling
    # curve that says nothing about xGMI.  device_count() does not initialise the GPU.)
    if os.environ.get("MIRL_BENCH_ONE_GPU", "0") != "1" and int(os.environ.get("LOCAL_RANK", "0")) >= torch.cuda.device_count():
        raise SystemExit("bench.py: LOCAL_RANK=%s but this node has %d GPU(s): --gpus N needs N GPUs (one rank per GPU).  To exercise the N > 1 code path on one GPU: "
                         "MIRL_BENCH_ONE_GPU=1 MIRL_BENCH_BACKEND=gloo (a diagnostic, labelled as such in the line)" % (os.environ.get("LOCAL_RANK", "0"), torch.cuda.device_count()))
    rank, world, local_rank = init_from_env(os.environ.get("MIRL_BENCH_BACKEND", "nccl"))
    if os.environ.get("MIRL_BENCH_ONE_GPU", "0") == "1":
        local_rank = 0
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d (launch N>1 with torch.distributed.run)" % (args.gpus, world))
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)
    if world > 1:
        # a multi-rank run that blocks (a rank lost in communicator set-up, a collective one rank never joins) must not sit on an 8-GPU node until the driver's own
        # limit: after MIRL_BENCH_WATCHDOG_S seconds (default 900; the whole run takes ~1 minute) every rank says where it is and leaves with code 124
        import faulthandler
        import threading

        def _give_up():
            sys.stderr.write("bench.py: rank %d of %d still running after the watchdog's limit: giving up (stacks follow)\n" % (rank, world))
            faulthandler.dump_traceback(file=sys.stderr, all_threads=True)
            sys.stderr.flush()
            os._exit(124)

        _wd = threading.Timer(float(os.environ.get("MIRL_BENCH_WATCHDOG_S", "900")), _give_up)
        _wd.daemon = True
        _wd.start()

    # Clock ramp, reported as `prewarm` in the line: a fresh process runs its first ~80 ms of GPU work at ramping clocks (tools/rollout_ab.py: 202 us per rollout for the
    # first 40 launches, 188 us afterwards, whatever the build), which is longer than W = 5 warm-up updates (7 ms) and put the first timed window of a short run 1.2 - 1.8 %
    # above the windows behind it (timed_windows, profiles/r04_grad_ab.txt).  A THROWAWAY engine of the same shape runs PREWARM_UPDATES updates first; the engine that is
    # measured is built afterwards and starts from its own fresh state (same seed, same parameters, same W warm-up and K timed updates as without the ramp).
    from deep_rl_amd import dist as DD
    import deep_rl_amd.engine as E
    carriers, headline_carrier, policy = {}, None, None
    # (world == 1 with a process group: MIRL_FORCE_PG=1, the one-GPU diagnostic that makes a single process join a world_size-1 RCCL group — the same policy, and the
    # engine is told to take the one-call sharded route although it is alone, so that the branch a multi-GPU run takes is the one measured: tests/test_gpu_script.py)
    policy_on = world > 1 or (torch.distributed.is_available() and torch.distributed.is_initialized())
    if policy_on:
        forced = os.environ.get("MIRL_COMM", "").lower() or None
        for w in ("rccl", "p2p"):
            try:
                carriers[w] = DD.probed_comm(None, w)
            except Exception as ex:  # noqa: BLE001  (every step in there is agreed over the ranks; an exception here is local set-up)
                carriers[w] = (None, {"ok": False, "why": "%s: %s" % (type(ex).__name__, ex)})
        if forced in ("rccl", "p2p"):
            headline_carrier, policy = (forced if carriers[forced][0] is not None else None), "MIRL_COMM=%s set by the caller" % forced
        elif forced == "auto":
            headline_carrier, policy = "auto", "MIRL_COMM=auto set by the caller: the faster carrier that passed its probe (collectives.carrier_choice)"
        else:
            headline_carrier = "rccl" if carriers["rccl"][0] is not None else ("p2p" if carriers["p2p"][0] is not None else None)
            policy = "RCCL when it can be created and passes its probe (the configuration BASELINE.json names), else P2P, else host-sequenced torch.distributed"
        if headline_carrier in ("rccl", "p2p"):
            DD.use_comm(carriers[headline_carrier][0])   # every engine of this process takes its one-call route on this communicator from here on
        if world == 1:
            E._FORCE_NATIVE_SHARDED = True

    # The engine that is MEASURED is built first and the throwaway engine of the clock ramp runs behind it, so that the W warm-up updates follow the ramp's last launch
    # in the same stream without the GPU going idle in between (until round 6 the measured engine's construction — allocations, fills, a host-side orthogonal init — sat
    # BETWEEN the ramp and the warm-up).  Neither engine touches the other's state.  Why: some runs on this pool's boxes show a first window 1 - 25 % above the two behind
    # it (1.3702 / 1.3174 / 1.3162, 1.6657 / 1.3303 / 1.3117, the round-5 tree on the same boxes likewise).  A same-box comparison of the two orders (12 runs each,
    # profiles/r06i_first_window_ab.txt) did NOT separate them — first / later windows <= 1.004 in 23 of 24 runs, one 1.013 in the new order: the outliers are the box's,
    # sporadic, and `timed_windows` shows them for what they are — so this order stays for the smaller idle gap, not as a cure.
    num_updates = args.warmup + args.steps
    env = D.make("CartPole-v1", num_envs=ENVS_PER_GPU, device=dev, seed=1, env_id_base=rank * ENVS_PER_GPU)
    torch.manual_seed(1)
    agent = D.ActorCritic(env)
    params0 = agent.flat.cpu().numpy().copy()
    opt = D.ClipAdam(agent, lr=2.5e-4, eps=1e-5, max_grad_norm=0.5)
    eng = D.PPOEngine(env, agent, opt, num_steps=T, n_minibatch=4, update_epochs=4)
    eng.reset()
    stats_host = torch.zeros(4, dtype=torch.int32).pin_memory()

    prewarm_updates = 0 if args.no_prewarm else PREWARM_UPDATES
    p_keep = None
    if prewarm_updates:
        p_env = D.make("CartPole-v1", num_envs=ENVS_PER_GPU, device=dev, seed=12345, env_id_base=rank * ENVS_PER_GPU)
        torch.manual_seed(12345)
        p_agent = D.ActorCritic(p_env)
        p_eng = D.PPOEngine(p_env, p_agent, D.ClipAdam(p_agent, lr=2.5e-4, eps=1e-5, max_grad_norm=0.5), num_steps=T, n_minibatch=4, update_epochs=4)
        p_eng.reset()
        for _ in range(prewarm_updates):
            p_eng.update()
        if headline_carrier == "auto":
            torch.cuda.synchronize()
            tune_carrier(p_eng, dev)    # recorded in dist.carrier_report -> collectives.carrier_choice
            headline_carrier = DD.resolved_carrier(None)
            for _ in range(prewarm_updates // 2):   # (the tuning ends in synchronisations: ramp again)
                p_eng.update()
        p_keep = (p_eng, p_agent, p_env)   # freed behind the measurement (a hipFree in front of it would synchronise the device)

    def one_update(u):
        opt.param_groups[0]["lr"] = (1.0 - u / num_updates) * 2.5e-4  # ppo.py:107-108
        eng.update()
        eng.episode_summary_async(stats_host)  # what a training loop reads per update; no host sync

    def timed_updates(u0, n, prof_every=PROF_EVERY, tags=("grad",)):
        """n updates bracketed by barrier + synchronize on both sides; -> (seconds, MAX over ranks; the in-library HIP-event profile of `tags` in every prof_every-th update)"""
        # (the profiler's events are created while the GPU is still busy with what came before: nothing but the barrier stands between the synchronisation and the window)
        N.prof_begin((n // prof_every + 1) * 20 * len(tags) + 16, tags=list(tags))
        N.prof_pause(True)
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()
        # Only the dominant kernel is bracketed inside the timed region, and only in every PROF_EVERY-th update (its 16 launches): a pair of HIP events around EVERY
        # launch of a 70 us kernel costs the loop it measures 0.098 ms per update = 7.5 % (tools/prof_overhead.py: 1.411 ms with all 320 launches of 20 updates
        # bracketed, 1.312 ms with none, the same 70.2 - 70.5 us per launch either way) - until round 4 the headline carried that.
        # (the bracketed updates sit inside the window — the 6th, 16th, ... — not right behind the synchronisation that opens it)
        phase = min(prof_every // 2, n - 1)
        t0 = time.perf_counter()
        for k, u in enumerate(range(u0, u0 + n)):
            N.prof_pause(k % prof_every != phase)
            one_update(min(u, num_updates - 1))
        N.prof_pause(False)
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
        return dt, prof

    for u in range(args.warmup):
        one_update(u)
    dt, prof = timed_updates(args.warmup, args.steps)   # THE measurement: f32 contractions
    # two more windows of the same K updates right behind it: the spread says whether a margin of a fraction of a per cent is signal
    windows = [dt]
    if not args.single_window:
        for _ in range(2):
            windows.append(timed_updates(num_updates, args.steps)[0])

    p_keep = None   # the clock ramp's throwaway engine

    # a sharded headline is only worth its number if the exchange really happened: right behind the timed windows, no wait of the carrier may have run out on any rank
    # and the replicas must still be bitwise identical (one MAX all-reduce of a checksum; host-synchronising, outside the timed region)
    headline_exchange = None
    if world > 1:
        try:
            eng.check_replicas()
            headline_exchange = {"replicas_identical": True}
        except Exception as ex:  # noqa: BLE001
            headline_exchange = {"replicas_identical": False, "error": "%s: %s" % (type(ex).__name__, ex)}

    # BESIDE the headline, never instead of it: the same three windows on libmirl's P2P carrier (rank-ordered sum over hipIpc-mapped inboxes, the gradient exchange
    # inside the slab-sum launch), the engine simply going on learning; then back to the headline's carrier for everything behind
    p2p_beside = None
    if policy_on and headline_carrier == "rccl" and carriers.get("p2p", (None,))[0] is not None and os.environ.get("MIRL_BENCH_P2P_BESIDE", "1") != "0":
        p2p_beside = {}
        try:
            DD.use_comm(carriers["p2p"][0])
            for _ in range(2):
                one_update(num_updates - 1)
            p2p_beside["windows"] = [timed_updates(num_updates, args.steps)[0] for _ in range(1 if args.single_window else 3)]
            try:
                eng.check_replicas()                                                    # (a no-op for a single process ...)
                N.check(N.lib().mi_comm_check(carriers["p2p"][0]), "mi_comm_check")     # ... so the carrier's own status word is asked too
                p2p_beside["replicas_identical"] = True
            except Exception as ex:  # noqa: BLE001
                p2p_beside["replicas_identical"] = False
                p2p_beside["error"] = "%s: %s" % (type(ex).__name__, ex)
        except Exception as ex:  # noqa: BLE001  (a failing carrier costs its own keys only; its optimizer steps are withheld on the device: mi_comm_poll)
            p2p_beside["error"] = "%s: %s" % (type(ex).__name__, ex)
        finally:
            DD.use_comm(carriers["rccl"][0])
    if policy_on and world == 1:
        E._FORCE_NATIVE_SHARDED = False   # the legs below set the route themselves

    # the round-1..3 methodology beside it: one more window with a pair of HIP events around EVERY gradient launch of EVERY update
    dt_all = None
    if not args.single_window:
        dt_all = timed_updates(num_updates, args.steps, prof_every=1)[0]

    # beside it, never instead of it: the same K updates with the split-bf16 experiment switched on (include/mi_rl.h, mi_ppo_set_contraction)
    variant = None
    if not args.headline_only:
        D.set_contraction("bf16x3")
        for u in range(2):
            one_update(num_updates - 1)
        v_dt, v_prof = timed_updates(num_updates, args.steps)
        D.set_contraction("f32")
        variant = (v_dt, v_prof)

    # untimed: per-kernel breakdown of 3 more updates (all tags)
    n_break = 3
    N.prof_begin(n_break * 64 + 64)
    for u in range(n_break):
        eng.update()
    breakdown = N.prof_end()

    sharded = None
    leg = os.environ.get("MIRL_BENCH_SHARDED_LEG", "")   # "1": also with --headline-only (A/B builds); "0": never
    if world == 1 and leg != "0" and (not args.headline_only or leg == "1"):
        try:
            sharded = sharded_route_leg(eng, one_update, timed_updates, num_updates - 1, args.steps, dev, 1e3 * dt / args.steps,
                                        1e3 * prof["grad"][0] / max(prof["grad"][1], 1))
        except Exception as ex:  # noqa: BLE001  (a diagnostic leg must not cost the measurement its line)
            sharded = {"error": "%s: %s" % (type(ex).__name__, ex)}

    from deep_rl_amd import dist as _dist
    # (MIRL_FORCE_PG=1 makes a single process join a world_size-1 RCCL group: the diagnostics' native-communicator branch then runs on a one-GPU box, tests/test_gpu_script.py)
    dist_on = world > 1 or (torch.distributed.is_available() and torch.distributed.is_initialized())
    comm = _dist.native_comm(eng.pg) if dist_on else None
    eng_native = comm is not None
    collectives = None
    if dist_on:   # every rank takes part in the collectives; a failure here must not cost the measurement above its line
        try:
            collectives = collective_diagnostics(eng, comm, world, dev, breakdown, n_break)
        except Exception as ex:  # noqa: BLE001
            collectives = {"error": "%s: %s" % (type(ex).__name__, ex)}
        if collectives is not None and "error" not in collectives and os.environ.get("MIRL_BENCH_CARRIER_LEGS", "1") != "0":
            try:
                collectives["carriers"] = carrier_legs(eng, world, dev, one_update, timed_updates, num_updates - 1, args.steps)
            except Exception as ex:  # noqa: BLE001
                collectives["carriers"] = {"error": "%s: %s" % (type(ex).__name__, ex)}
            collectives["rccl_env"] = _dist.apply_rccl_env() or None
        if collectives is not None:
            which_ran = (("p2p" if N.lib().mi_comm_carrier(comm) == 1 else "rccl") if eng_native else None)
            collectives["carrier_choice"] = {"MIRL_COMM": os.environ.get("MIRL_COMM"), "policy": policy, "headline_carrier": which_ran,
                                             "probes": {w: c[1] for w, c in carriers.items()}, "measured": _dist.carrier_report(eng.pg)}
            # whatever carried the headline: the top-level RCCL facts come from the RCCL communicator whenever one exists ("did RCCL see N ranks?")
            if carriers.get("rccl", (None,))[0] is not None:
                import ctypes as _C
                ws_, rk_, ver_, cnt_ = _C.c_int(), _C.c_int(), _C.c_int(), _C.c_int()
                if N.lib().mi_comm_info(carriers["rccl"][0], _C.byref(ws_), _C.byref(rk_), _C.byref(ver_), _C.byref(cnt_)) == 0:
                    collectives["rccl_version"], collectives["rccl_comm_count"] = ver_.value, cnt_.value
    finite = bool(torch.isfinite(agent.flat).all().item())
    ep = stats_host.tolist()
    if rank == 0:
        env_steps = args.steps * T * ENVS_PER_GPU * world
        mb = T * ENVS_PER_GPU // 4
        g_ms, g_n = prof["grad"]
        flops_per_launch = FLOPS_PER_ROW_UPDATE * mb
        ach = flops_per_launch / (g_ms / max(g_n, 1) * 1e-3) / 1e12 if g_n else 0.0
        traffic, traffic_source = pmc_traffic("grad_kernel")
        steps_per_s = env_steps / dt
        out = {
            "metric": "env-steps/sec + updates/sec, PPO CartPole-v1 4096 envs/GPU",
            "value": round(steps_per_s, 1), "unit": "env-steps/s",
            "updates_per_s": round(args.steps / dt, 3), "optimizer_steps_per_s": round(16 * args.steps / dt, 2),
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * dt / args.steps, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "ppo.py CartPole-v1, %d envs/GPU x %d steps per update, 4 epochs x 4 minibatches of %d rows, "
                                   "2x64-tanh actor+critic (9155 params), on-device env.step + GAE + fwd/bwd + clip + Adam" % (ENVS_PER_GPU, T, mb),
                       "envs_per_gpu": ENVS_PER_GPU, "num_steps": T, "minibatch_rows": mb, "parallelism": "env-sharded x%d, grad all-reduce" % world,
                       "collectives": collectives_label(world, eng_native and N.lib().mi_comm_carrier(comm) == 1, eng_native, carriers, policy)},
            "roofline": {"bound": "mfma", "kernel": "grad_kernel_f32", "achieved": round(ach, 3), "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(ach / PEAK_F32_MFMA_TFLOPS, 4), "traffic": traffic, "traffic_source": traffic_source,
                         "flops_per_launch": flops_per_launch, "avg_launch_us": round(1e3 * g_ms / max(g_n, 1), 2), "launches": g_n,
                         "sampling": "HIP events around the 16 launches of every %d-th update of the timed window (a pair around every launch costs the window 7.5 %%: "
                                     "tools/prof_overhead.py); launches = the bracketed ones" % PROF_EVERY},
            "hbm_roofline": {"algorithmic_bytes_per_env_step": BYTES_PER_ENV_STEP,
                             "achieved_GBps_per_gpu": round(BYTES_PER_ENV_STEP * steps_per_s / world / 1e9, 2), "peak_GBps": PEAK_HBM_GBS,
                             "frac": round(BYTES_PER_ENV_STEP * steps_per_s / world / 1e9 / PEAK_HBM_GBS, 6)},
            "kernel_ms_per_update_bracketed": dict({k: round(v[0] / n_break, 4) for k, v in breakdown.items() if v[1]},
                                                   note="HIP events around EVERY tagged launch in an untimed pass of %d updates: each figure carries its own event overhead "
                                                        "(~3 us per launch); the sum exceeds the timed window by construction — use kernel_device_ms_per_update for a sum" % n_break),
            "kernel_device_ms_per_update": kernel_device_durations(1e3 * dt / args.steps),
            "methodology": {"prewarm_updates": prewarm_updates, "events": "HIP events around grad_kernel's 16 launches of every %d-th update of the timed windows" % PROF_EVERY,
                            "protocol_changed_in": "r04 (rounds 1-3: no prewarm, events around every launch of every update; BASELINE.md section 3 names neither)",
                            "ms_per_step_all_launches_bracketed": None if dt_all is None else round(1e3 * dt_all / args.steps, 4),
                            "value_all_launches_bracketed": None if dt_all is None else round(env_steps / dt_all, 1),
                            "note": "the r01-r03 style figure (a pair of events around every gradient launch) from one more window of the same K updates right behind the "
                                    "repeat windows; compare BENCH_r01..r03 with this key, BENCH_r04.. with `value`"},
            "last_rollout": {"episodes": ep[0], "mean_return": round(ep[1] / max(ep[0], 1), 2), "max_return": ep[2]},
            "params_finite": finite,
            "build": {"source_id": N.lib().mi_source_id().decode(), "abi": N.ABI_VERSION,
                      "what": "sha256 (12 hex digits) of the kernel sources libmirl.so was built from (csrc/Makefile); the static figures quoted from profiles/ carry theirs"},
            "prewarm": {"updates": prewarm_updates, "what": "throwaway engine of the same shape (its own envs, parameters and storage), run right in front of the W warm-up updates "
                                                             "of the engine that is measured (clock ramp of a fresh process: ~80 ms; the measured engine is built first so that the GPU does "
                                                             "not idle between the ramp and the warm-up); the W warm-up and K timed updates are the measured engine's first W + K updates; "
                                                             "--no-prewarm turns it off"},
        }
        wms = sorted(1e3 * w / args.steps for w in windows)
        out["timed_windows"] = {"count": len(wms), "steps_each": args.steps, "ms_per_step": [round(1e3 * w / args.steps, 4) for w in windows],
                                "min": round(wms[0], 4), "median": round(wms[len(wms) // 2], 4), "max": round(wms[-1], 4),
                                "note": "window 1 is `value` / `ms_per_step`; the others repeat it back to back (same barriers), learning continuing (episodes get longer: "
                                        "fewer resets per rollout)"}
        if p2p_beside is not None:
            if p2p_beside.get("windows"):
                pw = [1e3 * w / args.steps for w in p2p_beside["windows"]]
                out["value_p2p"] = round(env_steps / p2p_beside["windows"][0], 1)
                out["ms_per_step_p2p"] = round(pw[0], 4)
                out["timed_windows_p2p"] = {"count": len(pw), "steps_each": args.steps, "ms_per_step": [round(x, 4) for x in pw],
                                            "note": "the same K updates on libmirl's P2P carrier, right behind the headline's windows (learning continuing); window 1 is value_p2p"}
                out["replicas_identical_p2p"] = p2p_beside.get("replicas_identical")
                best = "p2p" if p2p_beside.get("replicas_identical") and p2p_beside["windows"][0] < dt else "rccl"
                out["best_carrier"], out["value_best_carrier"] = best, (out["value_p2p"] if best == "p2p" else round(steps_per_s, 1))
            if "error" in p2p_beside:
                out["p2p_error"] = p2p_beside["error"]
        if sharded is not None:
            out["sharded_route"] = sharded
        if collectives is not None:
            if headline_exchange is not None:
                collectives["headline_exchange"] = headline_exchange
            out["collectives"] = collectives
        if variant is not None:
            v_dt, v_prof = variant
            vg_ms, vg_n = v_prof["grad"]
            out["variant_bf16x3"] = {
                "dtype": "f32 operands split into 3 bf16 parts, 6 products, f32 accumulate (bf16 MFMA)", "switch": "mi_ppo_set_contraction(MI_CONTRACTION_BF16X3) / MIRL_PPO_CONTRACTION=bf16x3",
                "status": "experiment; every f32 parity tolerance holds unchanged (tests/test_gpu_contraction.py); not the headline",
                "value": round(env_steps / v_dt, 1), "unit": "env-steps/s", "ms_per_step": round(1e3 * v_dt / args.steps, 4), "steps": args.steps,
                "grad_kernel_avg_launch_us": round(1e3 * vg_ms / max(vg_n, 1), 2),
                "error_bound": "per contraction |y - y_exact| <= 2^-20 sum|a||b| + K 2^-126 (include/mi_rl.h MI_BF16X3_REL_BOUND; tests/test_gpu_bf16x3_bound.py: worst observed "
                               "2^-21.2 at K = 64 against 2^-21.5 for the exact-f32 MFMA path); ten-seed learning equivalence: tests/test_gpu_learning.py"}
            # its OWN roofline, against the bf16 matrix peak: the three 64 x 64 contractions per net (49,152 of the 53,376 algorithmic FLOP per row) run as SIX bf16 part
            # products each, so the matrix pipe executes 6 x their FLOP as bf16 — the honest fraction of the bf16 peak is small, and the algorithmic one smaller still
            v_us = 1e3 * vg_ms / max(vg_n, 1)
            contraction_flops = 3 * (2 * 64 * 64) * 2 * mb            # layer 2 forward, dh1, dW2: 2 x 64 x 64 FLOP each per row, both nets = 49,152 of the 53,376 per row
            executed_bf16 = 6 * contraction_flops
            out["variant_bf16x3"]["roofline"] = {
                "bound": "mfma", "kernel": "grad_kernel_bx", "peak": PEAK_BF16_MFMA_TFLOPS, "unit": "TFLOP/s", "peak_what": "dense bf16 MFMA (16 x the f32 MFMA peak)",
                "achieved": round(executed_bf16 / (v_us * 1e-6) / 1e12, 2) if vg_n else 0.0, "frac": round(executed_bf16 / (v_us * 1e-6) / 1e12 / PEAK_BF16_MFMA_TFLOPS, 4) if vg_n else 0.0,
                "flops_per_launch": executed_bf16, "flops_what": "bf16 FLOP the matrix pipe EXECUTES: 6 part products per f32 product of the three 64 x 64 contractions",
                "achieved_algorithmic": round(flops_per_launch / (v_us * 1e-6) / 1e12, 2) if vg_n else 0.0,
                "frac_algorithmic": round(flops_per_launch / (v_us * 1e-6) / 1e12 / PEAK_BF16_MFMA_TFLOPS, 4) if vg_n else 0.0,
                "note": "VALU-issue- and dependency-bound (176 of ~550 vector instructions per tile are the operand splits), not matrix-bound: docs/LEDGER.md"}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(params0, float(os.environ.get("MIRL_CPU_BASELINE_SECONDS", "10")))  # bounded sample (default 10 s)
            out["cpu_baseline_n1"] = cpu_baseline_n1(params0)
        if world == 1 and not args.headline_only:
            cs = 0.0 if args.no_cpu_baseline else 3.0
            del eng, env, agent, opt
            out["config3_dqn"] = bench_dqn(dev, cpu_seconds=cs)
            out["config3_dueling"] = bench_dqn(dev, iters=200, variant="dueling")
            out["config3_per"] = bench_dqn(dev, iters=200, variant="per")
            out["config4_sac"] = bench_sac(dev, cpu_seconds=cs)
            # "(also a scaled batch, stated)" — NOT the reference's batch: the same loops with batch 4,096, where the update kernels are throughputs, not fixed latencies
            out["config3_dqn_scaled"] = dict(bench_dqn(dev, iters=150, cpu_seconds=min(cs, 2.0), batch=4096), scaled="batch 4096 instead of the reference's 128 (dqn.py:46)")
            out["config4_sac_scaled"] = dict(bench_sac(dev, iters=150, cpu_seconds=min(cs, 2.0), batch=4096), scaled="batch 4096 instead of the reference's 256 (sac.py:85)")
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if dist_on:
        torch.distributed.barrier()
        from deep_rl_amd.dist import destroy_native_comms
        destroy_native_comms()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
