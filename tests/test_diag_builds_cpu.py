"""The diagnostic builds of the HIP sources (-D*_STAMPS / *_MARKS / MI_INSIDE: wall-clock marks inside the kernels, tools/*_stamps.py, tools/sac_marks.py,
tools/inside_view.py) must keep compiling for gfx950 — they are how DESIGN.md's inside views were measured and are not part of the default build (hipcc
cross-compiles without a GPU; device side only: the marks live in device code)."""
import os
import subprocess
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "deep_rl_amd", "csrc")
FLAGS = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-ffp-contract=off", "-fno-slp-vectorize", "-mllvm", "-amdgpu-mfma-vgpr-form=1", "--cuda-device-only", "-c"]


@pytest.mark.parametrize("src,defs", [
    ("mi_dqn.hip", ["-DTD_STAMPS", "-DDA_STAMPS", "-DPER_STAMPS"]),
    ("mi_sac.hip", ["-DSAC_MARKS"]),
    ("mi_sac.hip", ["-DSAC_STAMPS"]),
    ("mi_update.hip", ["-DGRAD_STAMPS"]),
    ("mi_rollout.hip", ["-DRQ_STAMPS"]),
    ("mi_env.hip", ["-DMI_INSIDE"]),
])
def test_diagnostic_build_compiles(src, defs):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    with tempfile.TemporaryDirectory() as d:
        out = subprocess.run([hipcc] + FLAGS + defs + [os.path.join(CSRC, src), "-o", os.path.join(d, "x.o")], capture_output=True, text=True, timeout=900)
        assert out.returncode == 0, out.stderr[-3000:]
