"""tests/_ports.py: the rendezvous ports of the multi-process tests come from OUTSIDE the kernel's ephemeral range (a bind(0) number can be taken again — as some
connection's source port — before the job's store listens on it: one EADDRINUSE in the GPU suite, round 6), and a launch that died on EADDRINUSE is repeated."""
import os
import socket
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from _ports import _ephemeral_low, free_port, run_with_port  # noqa: E402


def test_free_port_is_bindable_and_not_ephemeral():
    low = _ephemeral_low()
    seen = set()
    for _ in range(20):
        p = free_port()
        assert 10000 <= p < max(low, 30001)
        with socket.socket() as s:
            s.bind(("127.0.0.1", p))      # still free
        seen.add(p)
    assert len(seen) > 10                 # drawn at random, not one fixed number


def test_run_with_port_repeats_a_launch_that_died_on_eaddrinuse(tmp_path):
    marker = tmp_path / "tries"
    script = ("import sys, pathlib; p = pathlib.Path(%r); n = int(p.read_text()) if p.exists() else 0; p.write_text(str(n + 1));\n"
              "sys.stderr.write('EADDRINUSE: address already in use\\n' if n == 0 else 'fine\\n'); sys.exit(1 if n == 0 else 0)" % str(marker))
    ports = []
    out = run_with_port(lambda port: (ports.append(port) or [sys.executable, "-c", script], dict(os.environ)), capture_output=True, text=True, timeout=60)
    assert out.returncode == 0 and marker.read_text() == "2" and len(ports) == 2
    # any other failure is NOT repeated
    out = run_with_port(lambda port: ([sys.executable, "-c", "import sys; sys.stderr.write('boom'); sys.exit(3)"], dict(os.environ)), capture_output=True, text=True, timeout=60)
    assert out.returncode == 3
