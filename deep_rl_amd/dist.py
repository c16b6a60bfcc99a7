"""The only cross-rank exchange of the path: SUM all-reduces of two tiny buffers.

One process per GPU, ``torch.distributed`` backend "nccl" (= RCCL over xGMI on ROCm); "gloo" on CPU for the
world_size-2 tests.  Envs shard embarrassingly (rank r owns global envs [r*N, (r+1)*N)), so there is no
data-path collective besides these:
  * advantage statistics {sum, sum of squares, count} per minibatch of every epoch, once per update  -> global mean / unbiased std (ppo.py:169)
  * the flat gradient (+ 4 loss terms), each rank's share already scaled by 1/(world*mb) (ppo.py:189-192)
Messages are <= 36.6 KB: latency-bound, one fused buffer per collective, in-stream, no bucketing.

Two carriers: (1) libmirl's own RCCL communicator (`native_comm`, csrc/mi_comm.hip, direct rccl.h) — the production path: the whole
sharded update is ONE C call (mi_ppo_update_sharded) with the collectives enqueued between its launches; (2) `torch.distributed`
(`allreduce_sum_`) — the host-sequenced path, kept for gloo (CPU tests, two ranks on one GPU) and as the A/B reference.
Diagnostics on a one-GPU box: MIRL_FORCE_PG=1 makes a single process join a (world_size 1) process group so that RCCL really runs;
MIRL_FORCE_COLLECTIVES=1 makes `allreduce_sum_` issue its collective even at world_size 1.
"""
import os

import ctypes as C

import torch
import torch.distributed as dist

from ._native import MiError

_FORCE_PG = os.environ.get("MIRL_FORCE_PG", "0") == "1"
_FORCE_COLLECTIVES = os.environ.get("MIRL_FORCE_COLLECTIVES", "0") == "1"


def init_from_env(backend=None):
    """Join the job described by RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* (torch.distributed.run).
    Returns (rank, world_size, local_rank); a single process needs no process group."""
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
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
    pair = torch.stack([c, -c])
    dist.all_reduce(pair, op=dist.ReduceOp.MAX, group=group)
    hi, lo = int(pair[0].item()), -int(pair[1].item())
    if hi != lo:
        raise MiError("replica divergence: %s differs across ranks (checksum of rank %d: %016x; min %016x, max %016x over %d ranks) — the replicas no longer "
                      "hold the same parameters" % (what, rank(group), int(c.item()), lo, hi, world_size(group)))


_native_comms = {}


def native_comm(group=None):
    """libmirl's RCCL communicator for `group` (created collectively on first use) or None when there is no process group, its backend
    is not nccl (gloo runs keep the host-sequenced path), MIRL_NATIVE_COMM=0, or the creation failed on ANY rank (the ranks agree on
    that through the process group, so either all of them take the one-call path or all of them fall back).
    Rank 0 draws the ncclUniqueId, the group broadcasts it."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_backend(group) != "nccl" or os.environ.get("MIRL_NATIVE_COMM", "1") == "0":
        return None
    key = id(group) if group is not None else 0
    if key not in _native_comms:
        import sys

        from . import _native as N

        ident = (C.c_char * 128)()
        ok = 1
        if dist.get_rank(group) == 0:
            ok = 1 if N.lib().mi_comm_unique_id(ident) == 0 else 0
        box = [bytes(ident.raw), ok]
        dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        ident.raw, ok = box[0], box[1]
        h = C.c_void_p()
        if ok and N.lib().mi_comm_create(ident, dist.get_world_size(group), dist.get_rank(group), C.byref(h)) != 0:
            ok, err = 0, N.lib().mi_last_error().decode()
            print("deep_rl_amd: mi_comm_create failed on rank %d (%s): falling back to torch.distributed collectives" % (dist.get_rank(group), err), file=sys.stderr)
        import torch as _t
        flag = _t.tensor([ok], dtype=_t.int32, device=_t.device("cuda", _t.cuda.current_device()))
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
        if int(flag.item()) == 0:
            if h.value:
                N.lib().mi_comm_destroy(h)
            _native_comms[key] = None
        else:
            _native_comms[key] = h
    return _native_comms[key]


def destroy_native_comms():
    from . import _native as N

    for h in _native_comms.values():
        if h is not None:
            N.lib().mi_comm_destroy(h)
    _native_comms.clear()


def global_adv_mean_std(sums):
    """{sum, sum sq, count} (already all-reduced) -> (mean, unbiased std) exactly as the gradient kernel derives them."""
    s1, s2, n = (float(x) for x in sums)
    mean = s1 / n
    var = max((s2 - s1 * mean) / (n - 1.0), 0.0)
    return mean, var ** 0.5
