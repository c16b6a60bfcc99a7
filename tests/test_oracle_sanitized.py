"""The CPU oracle under AddressSanitizer + UndefinedBehaviorSanitizer (SURVEY §5 'race detection / sanitizers'; GPU sanitizers are not available on this
pool, so the sanitizers run on the CPU build only): the C restatement is what every parity claim rests on, so an out-of-bounds read in it would silently
poison the expected values.  The pinned-vector tests of the three oracles (17 tests, ~40 s) run in a child process against oracle/libcpu_ref_asan.so."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _asan_runtime():
    p = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    return p if p and os.path.isabs(p) and os.path.exists(p) else None


@pytest.mark.skipif(_asan_runtime() is None, reason="gcc's libasan.so is not installed")
def test_oracle_pinned_vectors_under_asan_ubsan():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "libcpu_ref_asan.so"], stdout=subprocess.DEVNULL)
    env = dict(os.environ)
    env.update({"MIRL_ORACLE_SO": os.path.join(ROOT, "oracle", "libcpu_ref_asan.so"), "LD_PRELOAD": _asan_runtime(),
                "ASAN_OPTIONS": "detect_leaks=0:abort_on_error=0:exitcode=23", "UBSAN_OPTIONS": "print_stacktrace=1:halt_on_error=1", "OMP_NUM_THREADS": "2"})
    out = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-p", "no:cacheprovider",
                          os.path.join(ROOT, "tests", "test_oracle_pinned.py"), os.path.join(ROOT, "tests", "test_oracle_dqn_pinned.py"),
                          os.path.join(ROOT, "tests", "test_oracle_sac_pinned.py")], capture_output=True, text=True, env=env, cwd=ROOT, timeout=900)
    text = out.stdout + out.stderr
    assert "AddressSanitizer" not in text and "runtime error" not in text, text[-4000:]
    assert out.returncode == 0, text[-4000:]
    assert "17 passed" in out.stdout, out.stdout[-2000:]
