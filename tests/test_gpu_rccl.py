"""RCCL really executes ("RCCL has never executed in this project"): a one-rank NCCL process group on the one-GPU box, libmirl's
own communicator (csrc/mi_comm.hip, direct rccl.h), and the sharded update as ONE C call (mi_ppo_update_sharded) — bit-identical to the
single-process fusion and to the host-sequenced route with torch.distributed all-reduces (tests/_rccl_world1_worker.py).  The multi-rank
arithmetic is covered by tests/test_dist_gloo.py (CPU, world_size 2) and tests/test_gpu_multirank.py (two ranks on one GPU over gloo);
RCCL on 2..8 GPUs is the driver's scaling run."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from _ports import free_port, run_with_port  # noqa: E402


def test_rccl_world_size_1_native_sharded_update_is_bit_identical():
    import torch

    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    env = dict(os.environ, PYTHONPATH=ROOT, MIRL_FORCE_PG="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    # torch.distributed.run is started BEFORE anything touches the GPU in the child (never re-exec a process that has initialised HIP)
    out = run_with_port(lambda port: ([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1", "--master-port", str(port),
                                       os.path.join(ROOT, "tests", "_rccl_world1_worker.py")], env), capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert out.returncode == 0, (out.stdout[-2000:], out.stderr[-4000:])
    assert "RCCL_WORLD1_OK" in out.stdout, out.stdout[-2000:]
