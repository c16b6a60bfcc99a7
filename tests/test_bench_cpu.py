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
