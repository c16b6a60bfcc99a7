"""Rendezvous ports for the multi-process tests.

`bind(("127.0.0.1", 0))`, read the number, close, hand it to torch.distributed — the usual recipe — draws from the kernel's EPHEMERAL range, the very range every outgoing
connection of every other process takes its source port from: between the close and the job's TCPStore listening the number can be taken (a lingering client socket of the
previous test's store is enough), and the job dies with EADDRINUSE.  Seen once in 400 tests on the GPU box (round 6) — once is a red `pytest -x`.  So: ports are drawn
OUTSIDE the ephemeral range (below its lower bound, probed by a bind), and the launchers that can be re-run are re-run on a fresh port when the job says EADDRINUSE."""
import os
import random
import socket
import subprocess


def _ephemeral_low():
    try:
        return int(open("/proc/sys/net/ipv4/ip_local_port_range").read().split()[0])
    except (OSError, ValueError, IndexError):
        return 32768


def free_port():
    """A TCP port on 127.0.0.1 that is free now and is not an ephemeral source port: random in [max(10000, low - 12000), low), probed by a bind."""
    low = _ephemeral_low()
    lo = max(10000, low - 12000)
    rng = random.Random(os.getpid() * 1000003 + int.from_bytes(os.urandom(4), "little"))
    for _ in range(200):
        port = rng.randrange(lo, low) if low > lo else rng.randrange(10000, 30000)
        with socket.socket() as s:
            try:
                s.bind(("127.0.0.1", port))
            except OSError:
                continue
            return port
    with socket.socket() as s:   # (everything probed was taken: the usual recipe after all)
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def run_with_port(make, attempts=3, **kw):
    """subprocess.run(*make(port), **kw) with a port from free_port(); re-run on a fresh port (at most `attempts` runs) when the job died on EADDRINUSE.
    `make(port)` -> (argv, env).  Returns the last CompletedProcess."""
    out = None
    for k in range(attempts):
        argv, env = make(free_port())
        out = subprocess.run(argv, env=env, **kw)
        err = (out.stderr or "") if isinstance(out.stderr, str) else ""
        if out.returncode != 0 and ("EADDRINUSE" in err or "address already in use" in err.lower()) and k + 1 < attempts:
            continue
        break
    return out
