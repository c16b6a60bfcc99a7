"""CPU-only checks of the host logic around the two carriers of libmirl's communicator (deep_rl_amd/dist.py; no GPU, no compute calls): carrier selection, the
MIRL_RCCL_ENV pass-through, the communicator override bench.py's legs use, the P2P entry points' argument checking through the C ABI, and the tool that condenses a
rocprofv3 kernel trace into per-kernel device time per update (bench.py's `kernel_device_ms_per_update`)."""
import ctypes as C
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from _ports import free_port, run_with_port  # noqa: E402


def test_carrier_selection_and_rccl_env(monkeypatch):
    import deep_rl_amd.dist as D
    from deep_rl_amd._native import MiError

    monkeypatch.delenv("MIRL_COMM", raising=False)
    assert D.carrier() == "rccl"                                   # the default stays RCCL (BASELINE.json north_star names it)
    monkeypatch.setenv("MIRL_COMM", "P2P")
    assert D.carrier() == "p2p"
    monkeypatch.setenv("MIRL_COMM", "auto")
    assert D.carrier() == "auto"
    assert D.resolved_carrier() is None and D.carrier_report() is None and D.native_comm() is None   # no process group: nothing to choose, single process
    monkeypatch.setenv("MIRL_COMM", "p2p")
    assert D.resolved_carrier() == "p2p"                            # a fixed carrier resolves to itself
    monkeypatch.setenv("MIRL_COMM", "mpi")
    with pytest.raises(MiError):
        D.carrier()
    monkeypatch.setenv("MIRL_RCCL_ENV", "NCCL_PROTO=LL, NCCL_MAX_NCHANNELS=1")
    monkeypatch.delenv("NCCL_PROTO", raising=False)
    monkeypatch.delenv("NCCL_MAX_NCHANNELS", raising=False)
    assert D.apply_rccl_env() == {"NCCL_PROTO": "LL", "NCCL_MAX_NCHANNELS": "1"} and os.environ["NCCL_PROTO"] == "LL"
    monkeypatch.setenv("MIRL_RCCL_ENV", "LD_PRELOAD=evil.so")      # only NCCL_* / RCCL_* names pass
    with pytest.raises(MiError):
        D.apply_rccl_env()
    monkeypatch.setenv("MIRL_RCCL_ENV", "")
    D._rccl_env.clear()


def test_native_comm_without_process_group_and_override(monkeypatch):
    import deep_rl_amd.dist as D

    monkeypatch.delenv("MIRL_COMM", raising=False)
    monkeypatch.delenv("MIRL_NATIVE_COMM", raising=False)
    assert D.native_comm() is None                                 # no process group: single process, no communicator
    token = C.c_void_p(0x1234)
    D.use_comm(token)
    try:
        assert D.native_comm() is token                            # bench.py's carrier legs hand every engine a communicator of their choice
        monkeypatch.setenv("MIRL_NATIVE_COMM", "0")
        assert D.native_comm() is None                             # ... unless the host-sequenced route is forced
    finally:
        D.use_comm(None)
    D.check_native_comm()                                          # nothing created: nothing to check
    D.destroy_native_comms()


_AUTO_WORKER = r"""
import os, sys
import torch, torch.distributed as dist
sys.path.insert(0, os.environ["MIRL_ROOT"])
import deep_rl_amd.dist as D
dist.init_process_group("gloo", rank=int(os.environ["RANK"]), world_size=2)
assert D.carrier() == "auto"
assert D.native_comm() is None            # no GPU, gloo: neither carrier exists -> the host-sequenced torch.distributed route, agreed by both ranks
r = D.carrier_report()
assert r["chosen"] is None and r["p2p"]["ok"] is False and r["rccl"]["ok"] is False and "gloo" in r["rccl"]["why"], r
assert D.resolved_carrier() is None
assert D.native_comm() is None            # decided once
t = torch.full((5,), float(dist.get_rank() + 1))
D.allreduce_sum_(t)
assert (t == 3.0).all()
D.destroy_native_comms()
assert D.carrier_report() is None
dist.destroy_process_group()
print("AUTO_WORKER_OK")
"""


def test_auto_carrier_on_cpu_gloo_falls_back_collectively(tmp_path):
    """MIRL_COMM=auto at world_size 2 without a GPU: both ranks agree that no carrier of libmirl's communicator can run and keep the torch.distributed route."""
    import socket

    w = tmp_path / "w.py"
    w.write_text(_AUTO_WORKER)
    port = free_port()
    procs = []
    for r in range(2):
        env = dict(os.environ, MIRL_ROOT=ROOT, MIRL_COMM="auto", RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, str(w)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    for pr in procs:
        out, err = pr.communicate(timeout=300)
        assert pr.returncode == 0 and "AUTO_WORKER_OK" in out, err[-3000:]


def test_p2p_entry_points_check_their_arguments():
    from deep_rl_amd import _native as N

    L = N.lib()
    h, buf = C.c_void_p(), (C.c_char * 64)()
    assert L.mi_comm_p2p_alloc(9, 0, 4096, C.byref(h), buf) == -1 and b"world_size" in L.mi_last_error()      # one node: at most 8 ranks
    assert L.mi_comm_p2p_alloc(2, 2, 4096, C.byref(h), buf) == -1
    assert L.mi_comm_p2p_alloc(2, 0, 0, C.byref(h), buf) == -1 and b"max_bytes" in L.mi_last_error()
    assert L.mi_comm_p2p_synthetic(0, 4096, C.byref(h)) == -1
    assert L.mi_comm_p2p_connect(None, None) == -1 and L.mi_comm_check(None) == -1 and L.mi_comm_carrier(None) == -1
    assert L.mi_test_contraction(0, None, None, 64, None, None) == -1 and L.mi_test_contraction(2, None, None, 64, None, None) == -1


def test_kernel_trace_condenser_adds_up(tmp_path):
    """tools/make_latest_durations.py on a synthetic trace: 14 updates of the fixed 35-launch sequence, 1.5 us between launches."""
    names = ["void rollout_q4_kernel<false, false>(mi_env)", "perm_stats_kernel(unsigned int)"] + ["grad_kernel_f32(float const*)", "void grad_reduce_kernel<0>(float const*)"] * 16 + ["clip_adam_kernel(float const*)"]
    dur = {"rollout_q4_kernel": 158000, "perm_stats_kernel": 15000, "grad_kernel_f32": 66000, "grad_reduce_kernel": 4600, "clip_adam_kernel": 4800}
    rows, t = ["Kernel_Name,Start_Timestamp,End_Timestamp"], 10 ** 9
    for _ in range(14):
        for n in names:
            d = next(v for k, v in dur.items() if k in n)
            rows.append('"%s",%d,%d' % (n, t, t + d))
            t += d + 1500
    p = tmp_path / "trace.csv"
    p.write_text("\n".join(rows) + "\n")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "make_latest_durations.py"), "t", str(p)], capture_output=True, text=True, timeout=60)
    assert out.returncode == 0, out.stderr
    d = json.loads(out.stdout)
    assert d["updates_averaged"] == 2 and d["launches_per_update"] == 35          # 13 complete updates, the first 10 and the last one dropped
    k = d["ppo_update"]
    assert abs(k["grad_kernel_f32"]["ms_per_update"] - 16 * 0.066) < 1e-6 and k["grad_kernel_f32"]["avg_launch_us"] == 66.0
    assert abs(d["sum_ms"] + d["launch_gaps_ms"] - d["span_ms"]) < 1e-4 and abs(d["launch_gaps_ms"] - 35 * 0.0015) < 1e-4
