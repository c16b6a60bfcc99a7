"""CPU-only checks of the drop-in boundary: libmirl.so loads without a GPU, exports every symbol that
include/mi_rl.h declares, reports errors through return codes, and its host helpers agree with the oracle."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def N():
    from deep_rl_amd import _native

    if not os.path.exists(_native.SO_PATH):
        import __graft_entry__

        __graft_entry__.build()
    return _native


def test_header_symbols_all_exported_and_bound(N):
    hdr = open(os.path.join(ROOT, "include", "mi_rl.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(mi_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 20
    L = C.CDLL(N.SO_PATH)
    for name in declared:
        assert hasattr(L, name), "libmirl.so does not export %s" % name
    assert declared == set(N.SIGNATURES), declared ^ set(N.SIGNATURES)
    assert N.lib().mi_version() == N.ABI_VERSION == int(re.search(r"#define MI_VERSION (\d+)", hdr).group(1))


def test_struct_layouts_match_header(N):
    assert C.sizeof(N.Episode) == 16
    assert C.sizeof(N.PPOBuffers) == 20 * 8 + 8 + 8  # 20 pointers + int32 (padded) + episode_stats_next
    assert C.sizeof(N.PPOHparams) == 4 * 4 + 8 + 6 * 4 + 4 * 8
    assert C.sizeof(N.SacCriticStep) == 8 + 8 + 6 * 8 + 8 + 4 * 8 + 8   # pointer, int32 (padded), 6 pointers, int64, 4 doubles, float (padded)
    assert N.lib().mi_ppo_workspace_bytes() >= 512 * 4624 * 4


def test_errors_are_return_codes_not_crashes(N):
    L = N.lib()
    h = C.c_void_p()
    assert L.mi_env_create(99, 8, 1, 0, C.byref(h)) == -1 and b"kind" in L.mi_last_error()
    assert L.mi_env_create(0, 0, 1, 0, C.byref(h)) == -1 and b"n_envs" in L.mi_last_error()
    assert L.mi_env_create(0, 8, 1, 0, None) == -1
    assert L.mi_gae(None, None, None, 128, 8, 0.99, 0.95, None, None, None) == -1
    assert L.mi_clip_adam(None, None, None, None, 9155, 1, 1e-3, 0.9, 0.999, 1e-5, 0.5, None, None) == -1
    assert L.mi_env_destroy(None) == 0
    with pytest.raises(N.MiError):
        N.check(L.mi_make_perm(0, 1, None, None), "mi_make_perm")


def test_host_perm_key_matches_oracle(N):
    from oracle import cpu_ref as R

    for seed, u, e in [(1, 0, 0), (1, 155, 3), (2 ** 40 + 7, 12, 1)]:
        assert N.lib().mi_perm_key(seed, u, e) == R.perm_key(seed, u, e)


def test_product_never_imports_the_oracle():
    """oracle/ is test infrastructure: nothing under deep_rl_amd/ may reference it."""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "deep_rl_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "cpu_ref" not in txt.replace("oracle/cpu_ref.c implements", "") or f in ("mi_common.h",), os.path.join(dirpath, f)
                assert "import oracle" not in txt and "from oracle" not in txt, os.path.join(dirpath, f)


_NULL_PROBE = r"""
import ctypes as C, json, sys
sys.path.insert(0, %r)
from deep_rl_amd import _native as N
L, out = N.lib(), {}
for name, (res, args) in sorted(N.SIGNATURES.items()):
    vals = []
    for a in args:
        if a in (C.c_void_p, C.c_char_p) or (hasattr(a, "_type_") and not isinstance(a._type_, str)):
            vals.append(None)
        elif a in (C.c_float, C.c_double):
            vals.append(0.0)
        else:
            vals.append(0)
    r = getattr(L, name)(*vals)
    out[name] = r if isinstance(r, int) else None
    print("DONE", name, flush=True)
print("RESULT", json.dumps(out))
"""


def test_every_entry_point_survives_null_and_zero_arguments(N):
    """include/mi_rl.h: "every failure is a negative return code".  Each exported function is called with NULL for every pointer and 0 for every number (in a child
    process: a crash would be a finding, not the end of the test run): none may fault, and every call that cannot possibly succeed must say so in its return code."""
    import json
    import subprocess
    import sys

    p = subprocess.run([sys.executable, "-c", _NULL_PROBE % ROOT], capture_output=True, text=True, timeout=240)
    done = [ln.split()[1] for ln in p.stdout.splitlines() if ln.startswith("DONE")]
    assert p.returncode == 0, "crashed after %s: %s" % (done[-1] if done else "nothing", p.stderr[-800:])
    res = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("RESULT")][0][7:])
    assert set(res) == set(N.SIGNATURES)
    harmless = {"mi_comm_destroy", "mi_env_destroy", "mi_timer_destroy",                    # destroying nothing is fine
                "mi_version", "mi_last_error", "mi_source_id", "mi_perm_key", "mi_ppo_workspace_bytes", "mi_dqn_workspace_bytes", "mi_sac_workspace_bytes", "mi_per_workspace_bytes",
                "mi_env_state_bytes", "mi_ppo_get_contraction", "mi_ppo_set_contraction",   # (mode 0 = f32 is valid)
                "mi_sac_set_max_cus", "mi_sac_usable_cus", "mi_sac_test_fault", "mi_sac_owed_alpha_fits", "mi_ppo_test_assume_sharded", "mi_prof_pause",   # (0 = off is valid)
                "mi_sac_shadow_invalidate", "mi_sac_shadow_valid"}   # (NULL = every registered vector / "not registered": 0)
    # valid with all-zero arguments wherever HIP works: "is the status word clear" / "clear it, no workspace" — MI_OK on a GPU box; on a CPU-only box
    # their hipHostMalloc fails and they say so.  Either way never a positive code or a crash.
    host_dependent = {"mi_sac_check", "mi_sac_clear_error"}
    for name, r in res.items():
        if name in harmless:
            continue
        if name in host_dependent:
            assert isinstance(r, int) and r <= 0, (name, r)
            continue
        assert isinstance(r, int) and r < 0, (name, r)


def test_committed_profiles_describe_the_committed_kernels(N):
    """the static figures bench.py quotes (roofline.traffic from profiles/latest_pmc.json, kernel_device_ms_per_update from
    profiles/latest_kernel_durations.json) must come from the binary that is timed.  The library's mi_source_id() hashes the CODE of its sources (comments stripped:
    csrc/srcid.py); both files record the id of the library they were measured on.  A kernel change without a new profiling round (tools/profile_round.sh,
    tools/profile_headline.sh) fails here — and bench.py would print "DIFFERS from this run's library" beside the figures."""
    import json

    mine = N.lib().mi_source_id().decode()
    assert len(mine) == 12 and mine != "unknown"
    for name in ("latest_pmc.json", "latest_kernel_durations.json"):
        rec = json.load(open(os.path.join(ROOT, "profiles", name)))
        assert rec.get("source_id") == mine, (name, rec.get("source_id"), mine)
