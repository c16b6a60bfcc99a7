"""The only cross-rank exchange of the path: SUM all-reduces of two tiny buffers.

One process per GPU, ``torch.distributed`` backend "nccl" (= RCCL over xGMI on ROCm); "gloo" on CPU for the
world_size-2 tests.  Envs shard embarrassingly (rank r owns global envs [r*N, (r+1)*N)), so there is no
data-path collective besides these:
  * advantage statistics {sum, sum of squares, count} per minibatch of every epoch, once per update  -> global mean / unbiased std (ppo.py:169)
  * the flat gradient (+ 4 loss terms), each rank's share already scaled by 1/(world*mb) (ppo.py:189-192)
Messages are <= 36.6 KB: latency-bound, one fused buffer per collective, in-stream, no bucketing.

Carriers: (1) libmirl's own communicator (`native_comm`, csrc/mi_comm.hip) — the production path: the whole sharded update is ONE C call
(mi_ppo_update_sharded) with the collectives enqueued between its launches — either RCCL (direct rccl.h; default) or, with MIRL_COMM=p2p, the
one-shot peer-to-peer exchange over hipIpc-mapped inboxes (rank-ordered sum, works with two ranks on one device); (2) `torch.distributed`
(`allreduce_sum_`) — the host-sequenced path, kept for gloo (CPU tests, two ranks on one GPU) and as the A/B reference.
Diagnostics on a one-GPU box: MIRL_FORCE_PG=1 makes a single process join a (world_size 1) process group so that RCCL really runs;
MIRL_FORCE_COLLECTIVES=1 makes `allreduce_sum_` issue its collective even at world_size 1.
"""
import os

# This pool's driver only supports dmabuf IPC: without this, RCCL's set-up and hipIpcGetMemHandle (the P2P carrier's inboxes, CUDA-tensor sharing) fail with
# "invalid argument".  Set here — at import, before anything of this process can have initialised HIP (importing torch does not) — so that bench.py started directly as a
# rank by torch.distributed.run and the drop-in scripts' documented 8-GPU command get it too, not only the launchers that remembered to export it.
_IPC_MODE_WAS_SET = "HSA_ENABLE_IPC_MODE_LEGACY" in os.environ
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import ctypes as C  # noqa: E402

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from ._native import MiError  # noqa: E402

if not _IPC_MODE_WAS_SET and torch.cuda.is_initialized():   # too late for this process: the HIP runtime read its environment when it was initialised
    import sys as _sys
    print("deep_rl_amd.dist: HSA_ENABLE_IPC_MODE_LEGACY was not exported and HIP is already initialised in this process — import deep_rl_amd (or export "
          "HSA_ENABLE_IPC_MODE_LEGACY=0) BEFORE the first CUDA / HIP call, or RCCL set-up and hipIpc (the P2P carrier) may fail with 'invalid argument' on this driver",
          file=_sys.stderr)
_FORCE_PG = os.environ.get("MIRL_FORCE_PG", "0") == "1"
_FORCE_COLLECTIVES = os.environ.get("MIRL_FORCE_COLLECTIVES", "0") == "1"


_rccl_env = {}


def apply_rccl_env():
    """MIRL_RCCL_ENV="NCCL_PROTO=LL,NCCL_MIN_NCHANNELS=1,NCCL_MAX_NCHANNELS=1": RCCL's small-message knobs for the path's 36.6 KB all-reduces, set in this process'
    environment before any communicator exists (only NCCL_* / RCCL_* names are accepted).  Returns what was set (bench.py records it in its line)."""
    spec = os.environ.get("MIRL_RCCL_ENV", "")
    for item in filter(None, (x.strip() for x in spec.split(","))):
        k, sep, v = item.partition("=")
        if not sep or not (k.startswith("NCCL_") or k.startswith("RCCL_")):
            raise MiError("MIRL_RCCL_ENV: %r is not NCCL_*=value / RCCL_*=value" % item)
        os.environ[k] = v
        _rccl_env[k] = v
    return dict(_rccl_env)


def init_from_env(backend=None):
    """Join the job described by RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* (torch.distributed.run).
    Returns (rank, world_size, local_rank); a single process needs no process group."""
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    apply_rccl_env()
    if (world > 1 or _FORCE_PG) and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        kw = {}
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            kw["device_id"] = torch.device("cuda", local_rank)
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return rank, world, local_rank


def world_size(group=None):
    return dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1


def rank(group=None):
    return dist.get_rank(group) if dist.is_available() and dist.is_initialized() else 0


def allreduce_sum_(t, group=None):
    """In-place SUM all-reduce on the tensor's device/stream; no-op for a single process."""
    if world_size(group) > 1 or (_FORCE_COLLECTIVES and dist.is_available() and dist.is_initialized()):
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t


def replica_check_interval():
    """MIRL_CHECK_REPLICAS=K: every K-th update of a sharded engine ends with check_replicas() (0 / unset: never)."""
    try:
        return max(int(os.environ.get("MIRL_CHECK_REPLICAS", "0")), 0)
    except ValueError:
        return 0


def state_checksum(tensors):
    """Order-sensitive 62-bit checksum of the raw 32-bit words of `tensors` (sum of (word_i + 1) * (2 i + 1), wrapping, i running over the concatenation) as a
    0-d int64 tensor on the tensors' device.  Plumbing, not arithmetic of the path: torch integer ops, the same on CPU (gloo tests) and GPU."""
    acc, base = None, 0
    for t in tensors:
        w = t.detach().reshape(-1).view(torch.int32).to(torch.int64)
        k = torch.arange(base, base + w.numel(), dtype=torch.int64, device=w.device) * 2 + 1
        c = ((w + 1) * k).sum()
        acc = c if acc is None else acc + c
        base += w.numel()
    return acc & 0x3FFFFFFFFFFFFFFF


def check_replicas(tensors, group=None, what="replicated state"):
    """Replica-divergence guard: the replicated state of a sharded run — parameters and optimizer moments, stepped identically on every rank from
    the same all-reduced gradient (reference ppo.py:189-192 / dqn.py:131-133 / sac.py:185-210 with the exchange in between) — must be BITWISE equal on all ranks.
    One MAX all-reduce of {c, -c} of its checksum gives max and min; a mismatch raises MiError on every rank.  Host-synchronising; no-op for a single process."""
    if world_size(group) == 1:
        return
    c = state_checksum(tensors)
    bad = 0
    try:   # a wait of the P2P carrier that ran out on ANY rank (its buffers then hold local shares) is reported on every rank through the same exchange
        check_native_comm(group)
    except MiError as e:
        bad, why = 1, str(e)
    pair = torch.stack([c, -c, torch.full_like(c, bad)])
    dist.all_reduce(pair, op=dist.ReduceOp.MAX, group=group)
    if int(pair[2].item()):
        raise MiError("a collective of libmirl's P2P carrier timed out on %s" % (("this rank (%d): %s" % (rank(group), why)) if bad else "another rank"))
    hi, lo = int(pair[0].item()), -int(pair[1].item())
    if hi != lo:
        raise MiError("replica divergence: %s differs across ranks (checksum of rank %d: %016x; min %016x, max %016x over %d ranks) — the replicas no longer "
                      "hold the same parameters" % (what, rank(group), int(c.item()), lo, hi, world_size(group)))


_native_comms = {}


def carrier():
    """MIRL_COMM: "rccl" (default — libmirl's own RCCL communicator; needs an NCCL process group, one device per rank), "p2p" (the one-shot exchange over
    hipIpc-mapped inboxes, csrc/mi_comm.hip; any process group — the handles travel through it — and any placement, two ranks on one device included) or "auto"
    (both are created, checked against a known answer and timed on the path's own message when the first engine asks for a communicator; the ranks agree on the
    faster one that passed: `resolved_carrier`, `carrier_report`)."""
    c = os.environ.get("MIRL_COMM", "rccl").lower()
    if c not in ("rccl", "p2p", "auto"):
        raise MiError("MIRL_COMM=%r: known carriers are rccl, p2p and auto" % c)
    return c


def _agree(ok, group):
    """MIN over the ranks of a 0 / 1 flag (every rank takes the one-call route or none does)."""
    dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu")
    flag = torch.tensor([ok], dtype=torch.int32, device=dev)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
    return int(flag.item())


def _create_rccl(group):
    from . import _native as N

    ident = (C.c_char * 128)()
    ok = 1
    if dist.get_rank(group) == 0:
        ok = 1 if N.lib().mi_comm_unique_id(ident) == 0 else 0
    box = [bytes(ident.raw), ok]
    dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
    ident.raw, ok = box[0], box[1]
    h = C.c_void_p()
    err = ""
    if ok and N.lib().mi_comm_create(ident, dist.get_world_size(group), dist.get_rank(group), C.byref(h)) != 0:
        ok, err = 0, N.lib().mi_last_error().decode()
    return h, ok, err


def _create_p2p(group):
    """Every rank allocates its inbox, the 64-byte hipIpcMemHandle_t of all ranks travel through the process group, every rank maps its peers' inboxes."""
    from . import _native as N

    world, rk = dist.get_world_size(group), dist.get_rank(group)
    max_bytes = int(os.environ.get("MIRL_P2P_MAX_BYTES", str(1 << 20)))   # SAC's twin-critic gradient is 539 KB, PPO's 36.6 KB
    h, mine = C.c_void_p(), (C.c_char * 64)()
    ok, err = 1, ""
    if N.lib().mi_comm_p2p_alloc(world, rk, max_bytes, C.byref(h), mine) != 0:
        ok, err = 0, N.lib().mi_last_error().decode()
    import socket

    dev = torch.cuda.current_device()
    try:
        where = (socket.gethostname(), str(torch.cuda.get_device_properties(dev).uuid))
    except Exception:  # noqa: BLE001  (no uuid on this torch: the index — right on one node, where CUDA_VISIBLE_DEVICES-style remapping is the launcher's business)
        where = (socket.gethostname(), str(dev))
    boxes = [None] * world
    dist.all_gather_object(boxes, (bytes(mine.raw), ok, where, os.environ.get("MIRL_P2P_FUSED")), group=group)
    if ok and all(b[1] for b in boxes):
        if N.lib().mi_comm_p2p_connect(h, b"".join(b[0] for b in boxes)) != 0:
            ok, err = 0, N.lib().mi_last_error().decode()
        else:
            # Which form PPO's gradient exchange takes (inside the slab sum / a launch of its own) must be ONE decision for the whole communicator: the two forms publish
            # the gradient's lines in different orders under the same sequence number, and a mix sums permuted elements without any wait failing.  Every
            # rank derives it from the SAME gathered list: the LARGEST number of ranks on one device (test placements: > 2 take the stand-alone launch, csrc/mi_comm.hip
            # mi_comm_p2p_fused_ok) and the MIRL_P2P_FUSED settings, which must agree.
            per_device = {}
            for b in boxes:
                per_device[b[2]] = per_device.get(b[2], 0) + 1
            settings = {b[3] for b in boxes}
            if len(settings) > 1:
                ok, err = 0, "MIRL_P2P_FUSED differs across the ranks (%s): the ranks of a communicator must take the same form of the exchange" % sorted(map(str, settings))
            else:
                N.check(N.lib().mi_comm_p2p_set_colocated(h, max(per_device.values())), "mi_comm_p2p_set_colocated")
                forced = settings.pop()
                if forced is not None:
                    try:
                        mode = 1 if int(forced) != 0 else -1
                    except ValueError:
                        ok, err, mode = 0, "MIRL_P2P_FUSED=%r is not an integer" % forced, 0
                    if ok:
                        N.check(N.lib().mi_comm_p2p_set_fused(h, mode), "mi_comm_p2p_set_fused")
    elif ok:
        ok, err = 0, "a peer could not allocate its inbox"
    return h, ok, err


_override = None


def use_comm(handle):
    """Make `handle` (a libmirl communicator the caller created: mi_comm_p2p_synthetic, or another carrier's from native_comm(group, which)) the one every engine's
    one-call route takes from here on; None restores the default.  bench.py's carrier legs."""
    global _override
    _override = handle


def native_comm(group=None, which=None):
    """libmirl's communicator for `group` (created collectively on first use) or None when there is no process group, MIRL_NATIVE_COMM=0, the carrier is RCCL and
    the backend is not nccl (gloo runs keep the host-sequenced path), or the creation failed on ANY rank (the ranks agree on that through the process group, so either
    all of them take the one-call path or all of them fall back).  RCCL: rank 0 draws the ncclUniqueId, the group broadcasts it.  P2P (MIRL_COMM=p2p): see _create_p2p."""
    if os.environ.get("MIRL_NATIVE_COMM", "1") == "0":
        return None
    if _override is not None and which is None:
        return _override
    if not (dist.is_available() and dist.is_initialized()):
        return None
    which = which or carrier()
    if which == "auto":
        which = _auto_choice(group)
        if which is None:
            return None
    if which == "rccl" and dist.get_backend(group) != "nccl":
        return None
    key = (id(group) if group is not None else 0, which)
    if key not in _native_comms:
        import sys

        from . import _native as N

        h, ok, err = _create_p2p(group) if which == "p2p" else _create_rccl(group)
        if not ok:
            print("deep_rl_amd: creating the %s communicator failed on rank %d (%s): falling back to torch.distributed collectives" % (which, dist.get_rank(group), err),
                  file=sys.stderr)
        if _agree(ok, group) == 0:
            if h.value:
                N.lib().mi_comm_destroy(h)
            _native_comms[key] = None
        else:
            _native_comms[key] = h
    return _native_comms[key]


_auto = {}


def _probe(h, group, n_words, rounds):
    """Known-answer check + timing of ONE carrier on the path's own message (collective; every step runs on every rank, then the ranks agree).
    -> (ok agreed over the ranks, us per all-reduce: MAX over the ranks, why not)."""
    import time

    from . import _native as N

    dev = torch.device("cuda", torch.cuda.current_device())
    world, rk = dist.get_world_size(group), dist.get_rank(group)
    ok, why, us = 1, "", 0.0
    try:
        # integers: every summation order gives the same bits, so the answer is known whatever the carrier's grouping (ring, tree, rank order)
        buf = torch.empty(n_words, dtype=torch.float32, device=dev)
        want = float(world * (world + 1) // 2)
        for k in range(3):
            buf.fill_(float(rk + 1))
            N.check(N.lib().mi_comm_allreduce_sum(h, N.ptr(buf), buf.numel(), 0, N.stream_ptr(dev)), "mi_comm_allreduce_sum")
            torch.cuda.synchronize()
            N.check(N.lib().mi_comm_check(h), "mi_comm_check")
            bad = int((buf != want).sum().item())
            if bad:
                raise MiError("known-answer all-reduce %d: %d of %d elements differ from %g" % (k, bad, n_words, want))
        buf.zero_()
        for _ in range(10):
            N.check(N.lib().mi_comm_allreduce_sum(h, N.ptr(buf), buf.numel(), 0, N.stream_ptr(dev)), "mi_comm_allreduce_sum")
        torch.cuda.synchronize()
    except Exception as ex:  # noqa: BLE001  (a carrier that cannot run here is an answer, not an error)
        ok, why = 0, "%s: %s" % (type(ex).__name__, ex)
    if _agree(ok, group) == 0:
        return 0, 0.0, why or "failed on another rank"
    try:
        dist.barrier(group=group)
        t0 = time.perf_counter()
        for _ in range(rounds):
            N.check(N.lib().mi_comm_allreduce_sum(h, N.ptr(buf), buf.numel(), 0, N.stream_ptr(dev)), "mi_comm_allreduce_sum")
        torch.cuda.synchronize()
        us = 1e6 * (time.perf_counter() - t0) / rounds
        N.check(N.lib().mi_comm_check(h), "mi_comm_check")
    except Exception as ex:  # noqa: BLE001
        ok, why = 0, "%s: %s" % (type(ex).__name__, ex)
    t = torch.tensor([us if ok else float("inf")], dtype=torch.float64, device=dev if dist.get_backend(group) == "nccl" else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    if not torch.isfinite(t).all():
        return 0, 0.0, why or "failed on another rank"
    return 1, float(t.item()), ""


def _auto_choice(group):
    """MIRL_COMM=auto: "p2p", "rccl" or None (neither carrier passed: host-sequenced torch.distributed collectives).  Decided once per process group, collectively:
    each carrier that can be created sums a known-answer buffer of the path's largest PPO message (9,159 floats; mi_comm_check behind it) and is then timed on 100
    back-to-back all-reduces of it; the time of a carrier is the MAX over the ranks, the choice the smaller time — the same on every rank by construction.  The
    message is latency-bound (36.6 KB, sixteen dependent ones per update: DESIGN.md §6), which is what the probe measures."""
    gkey = id(group) if group is not None else 0
    if gkey in _auto:
        return _auto[gkey]["chosen"]
    from . import _native as N

    report = {"probe": "known-answer SUM of %d floats x 3, then 100 back-to-back all-reduces of it on an idle stream; us = MAX over ranks" % (N.NPARAMS + 4)}
    best, best_us = None, None
    for which in ("p2p", "rccl"):
        if which == "rccl" and dist.get_backend(group) != "nccl":
            report[which] = {"ok": False, "why": "process group is %s (RCCL needs nccl)" % dist.get_backend(group)}
            continue
        if which == "p2p" and not torch.cuda.is_available():
            report[which] = {"ok": False, "why": "no GPU in this process"}
            continue
        h = native_comm(group, which=which)
        if h is None:
            report[which] = {"ok": False, "why": "communicator could not be created (stderr has the reason)"}
            continue
        ok, us, why = _probe(h, group, N.NPARAMS + 4, 100)
        report[which] = {"ok": bool(ok), "us_per_allreduce": round(us, 2)} if ok else {"ok": False, "why": why}
        if not ok:   # a carrier that failed its probe is not kept: it may hold a timed-out wait
            key = (gkey, which)
            if _native_comms.get(key) is not None:
                N.lib().mi_comm_destroy(_native_comms[key])
            _native_comms[key] = None
        elif best is None or us < best_us:
            best, best_us = which, us
    report["chosen"] = best
    _auto[gkey] = report
    return best


def probed_comm(group, which, rounds=50):
    """libmirl's communicator of carrier `which` ("rccl" / "p2p") for `group`, created AND checked: the known-answer all-reduce of the path's own message (9,159 floats,
    three times, mi_comm_check behind it) and a short timing — the same probe MIRL_COMM=auto runs — whatever MIRL_COMM says.  Collective.  -> (handle | None, report);
    a carrier that cannot be created or fails its probe on ANY rank is destroyed on every rank (it may hold a timed-out wait) and reported with the reason.
    bench.py's N > 1 policy stands on this: the headline runs on RCCL (the configuration BASELINE.json names), the P2P carrier beside it."""
    from . import _native as N

    gkey = id(group) if group is not None else 0
    if not (dist.is_available() and dist.is_initialized()):
        return None, {"ok": False, "why": "no process group"}
    if which == "rccl" and dist.get_backend(group) != "nccl":
        return None, {"ok": False, "why": "process group is %s (RCCL needs nccl: one device per rank)" % dist.get_backend(group)}
    if which == "p2p" and not torch.cuda.is_available():
        return None, {"ok": False, "why": "no GPU in this process"}
    h = native_comm(group, which=which)
    if h is None:
        return None, {"ok": False, "why": "communicator could not be created (stderr has the reason)"}
    ok, us, why = _probe(h, group, N.NPARAMS + 4, rounds)
    if not ok:
        key = (gkey, which)
        if _native_comms.get(key) is not None:
            N.lib().mi_comm_destroy(_native_comms[key])
        _native_comms[key] = None
        return None, {"ok": False, "why": why}
    return h, {"ok": True, "us_per_allreduce": round(us, 2), "probe": "known-answer SUM of %d floats x 3, then %d back-to-back all-reduces; us = MAX over ranks" % (N.NPARAMS + 4, rounds)}


def set_auto_choice(group, which, how):
    """Replace the probe's choice for `group` by a measurement of the caller's (bench.py: the same short window of real sharded updates on every carrier that passed
    the probe — the probe times stand-alone all-reduces, while on the P2P carrier PPO's gradient exchange rides inside the slab-sum launch).  Every rank must call it
    with the same `which`; only a carrier that passed its known-answer probe can be chosen."""
    gkey = id(group) if group is not None else 0
    rep = _auto.get(gkey)
    if rep is None or not rep.get(which, {}).get("ok"):
        raise MiError("set_auto_choice(%r): not a carrier that passed the MIRL_COMM=auto probe of this process group" % (which,))
    rep["chosen_by_probe"], rep["chosen"], rep["chosen_by"] = rep["chosen"], which, how


def resolved_carrier(group=None):
    """The carrier the one-call routes of `group` take: MIRL_COMM, with "auto" resolved (collective on first use) — "rccl", "p2p" or None (host-sequenced)."""
    c = carrier()
    if c != "auto":
        return c
    if not (dist.is_available() and dist.is_initialized()):
        return None
    return _auto_choice(group)


def carrier_report(group=None):
    """What MIRL_COMM=auto measured for `group` (None before the first communicator was asked for, or with a fixed carrier)."""
    return _auto.get(id(group) if group is not None else 0)


def check_native_comm(group=None):
    """Host-synchronising: raise MiError when a wait of the P2P carrier ran out on this rank (mi_comm_check); no-op without a communicator or on RCCL."""
    from . import _native as N

    for (g, _), h in _native_comms.items():
        if h is not None and g == (id(group) if group is not None else 0):
            N.check(N.lib().mi_comm_check(h), "mi_comm_check")


def poll_native_comm(group=None):
    """The same question WITHOUT a synchronisation (mi_comm_poll: the host-pinned mirror of the carrier's status word, written by the wait that ran out): raises
    MiError once a wait of the P2P carrier has timed out on this rank.  Every mi_*_sharded call asks it at its entry anyway; loops that want to stop at the update
    that failed rather than at the next call poll it after their per-update host touch."""
    from . import _native as N

    if _override is not None:
        N.check(N.lib().mi_comm_poll(_override), "mi_comm_poll")
    for (g, _), h in _native_comms.items():
        if h is not None and g == (id(group) if group is not None else 0):
            N.check(N.lib().mi_comm_poll(h), "mi_comm_poll")


def destroy_native_comms():
    """Collective when a P2P communicator exists: a barrier in front (a peer may still be storing into this rank's inbox)."""
    from . import _native as N

    live = [h for h in _native_comms.values() if h is not None]
    if live and dist.is_available() and dist.is_initialized() and any(N.lib().mi_comm_carrier(h) == 1 for h in live):
        torch.cuda.synchronize()
        dist.barrier()
    for h in live:
        N.lib().mi_comm_destroy(h)
    _native_comms.clear()
    _auto.clear()


def global_adv_mean_std(sums):
    """{sum, sum sq, count} (already all-reduced) -> (mean, unbiased std) exactly as the gradient kernel derives them."""
    s1, s2, n = (float(x) for x in sums)
    mean = s1 / n
    var = max((s2 - s1 * mean) / (n - 1.0), 0.0)
    return mean, var ** 0.5
