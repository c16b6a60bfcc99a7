"""bench.py's N > 1 self-launch without a GPU: the parent starts the ranks as a child process (torch.distributed.run), relays nothing but rank 0's
JSON line and exits with the child's return code.  On this CPU-only container the ranks cannot get a device, so the observable contract is the FAILURE path: a non-zero
return code and an empty stdout — never a half-written line, never a hang.  (The success path runs on the GPU box: tests/test_gpu_script.py.)"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_self_launch_propagates_rank_failure_and_prints_no_line():
    import torch

    if torch.cuda.is_available():
        import pytest
        pytest.skip("CPU-only contract (with a GPU the ranks would succeed)")
    env = dict(os.environ, PYTHONPATH=ROOT)
    env.pop("WORLD_SIZE", None); env.pop("RANK", None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], env=env, capture_output=True, text=True,
                         timeout=240, cwd=ROOT)
    assert out.returncode != 0
    assert out.stdout == "", out.stdout[:500]
    assert "torch.distributed" in out.stderr or "Traceback" in out.stderr or "exitcode" in out.stderr


def test_parent_makes_no_gpu_call_before_launching():
    """The parent must not import torch (let alone touch HIP) before it starts the ranks: a process that has initialised the GPU must never fork-exec a launcher."""
    src = open(os.path.join(ROOT, "bench.py")).read()
    head = src[:src.index("def self_launch")]
    top_level_imports = [l for l in head.splitlines() if l.startswith(("import ", "from "))]
    assert not any("torch" in l or "deep_rl_amd" in l for l in top_level_imports), top_level_imports


def _bench():
    sys.path.insert(0, ROOT)
    import bench
    return bench


def test_chain_floors_follow_their_definition():
    """The floors the latency-bound configs are measured against: matrix cycles on the longest wave's dependent chain at 2.07 GHz + kernel boundaries x
    1.3 us + dependent memory round trips x 0.45 us — the numbers DESIGN.md sections 8 / 9 derive, recomputed here from the pass counts."""
    b = _bench()
    f = b.dqn_chain_floor(128)
    assert f["chain_floor_terms"]["chain_cycles"] == (10 * 72 + 96 + 42 + 24) * 32
    assert abs(f["chain_floor_us"] - ((720 + 162) * 32 / 2070.0 + 3 * 1.3 + 4 * 0.45)) < 0.01 and 19.0 < f["chain_floor_us"] < 19.7
    s = b.sac_chain_floor(256)
    assert s["chain_floor_terms"]["chain_cycles"] == (7 * 256 + 64) * 32 and abs(s["chain_floor_us"] - (s["chain_floor_terms"]["chain_cycles"] / 2070.0 + 4 * 1.3 + 6 * 0.45)) < 0.01
    p = b.per_chain_floor(128)
    assert p["chain_floor_us"] > f["chain_floor_us"] + 1.3          # the sampler launch: one more boundary + its own dependent chains
    out = b.with_floor({"ms_per_step": 0.0372}, f)
    assert abs(out["frac_of_chain_floor"] - f["chain_floor_us"] / 37.2) < 1e-3 and 0 < out["frac_of_chain_floor"] < 1


def test_collectives_label_names_the_carrier_that_ran_and_why_not_rccl():
    """config.collectives at N > 1: RCCL when RCCL carried the headline; otherwise the carrier that ran, "NOT the RCCL configuration", and the reason."""
    b = _bench()
    assert b.collectives_label(1, False, False, {}, None) == "none (single process)"
    ok = {"rccl": (object(), {"ok": True}), "p2p": (object(), {"ok": True})}
    lab = b.collectives_label(8, False, True, ok, "policy")
    assert lab.startswith("RCCL direct") and "value_p2p" in lab
    no = {"rccl": (None, {"ok": False, "why": "process group is gloo (RCCL needs nccl: one device per rank)"}), "p2p": (object(), {"ok": True})}
    lab = b.collectives_label(2, True, True, no, "policy")
    assert lab.startswith("P2P over hipIpc inboxes") and "NOT the RCCL configuration" in lab and "process group is gloo" in lab
    lab = b.collectives_label(2, False, False, {"rccl": (None, {"ok": False, "why": "x"}), "p2p": (None, {"ok": False, "why": "y"})}, "policy")
    assert lab.startswith("torch.distributed") and "NOT the RCCL configuration" in lab


def test_a_rank_without_a_gpu_of_its_own_is_refused_before_any_rendezvous():
    """bench.py started as rank 1 of 2 on a node with fewer GPUs than ranks leaves at once with the reason (no rendezvous, no hang, nothing on stdout): ranks that
    time-share a device would print a scaling curve that says nothing about xGMI.  (One GPU shared on purpose: MIRL_BENCH_ONE_GPU=1, tests/test_gpu_script.py.)"""
    import torch

    if torch.cuda.device_count() >= 2:
        import pytest
        pytest.skip("needs a node with fewer than two GPUs")
    env = dict(os.environ, PYTHONPATH=ROOT, RANK="1", LOCAL_RANK="1", WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT="1")
    env.pop("MIRL_BENCH_ONE_GPU", None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], env=env, capture_output=True, text=True, timeout=240, cwd=ROOT)
    assert out.returncode != 0 and out.stdout == ""
    assert "LOCAL_RANK=1 but this node has" in out.stderr and "MIRL_BENCH_ONE_GPU" in out.stderr
