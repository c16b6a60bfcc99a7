"""The only cross-rank exchange of the path: SUM all-reduces of two tiny buffers.

One process per GPU, ``torch.distributed`` backend "nccl" (= RCCL over xGMI on ROCm); "gloo" on CPU for the
world_size-2 tests.  Envs shard embarrassingly (rank r owns global envs [r*N, (r+1)*N)), so there is no
data-path collective besides these:
  * advantage statistics {sum, sum of squares, count} per minibatch of every epoch, once per update  -> global mean / unbiased std (ppo.py:169)
  * the flat gradient (+ 4 loss terms), each rank's share already scaled by 1/(world*mb) (ppo.py:189-192)
Messages are <= 36.6 KB: latency-bound, one fused buffer per collective, in-stream, no bucketing.
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Join the job described by RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* (torch.distributed.run).
    Returns (rank, world_size, local_rank); a single process needs no process group."""
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
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
    if world_size(group) > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t


def global_adv_mean_std(sums):
    """{sum, sum sq, count} (already all-reduced) -> (mean, unbiased std) exactly as the gradient kernel derives them."""
    s1, s2, n = (float(x) for x in sums)
    mean = s1 / n
    var = max((s2 - s1 * mean) / (n - 1.0), 0.0)
    return mean, var ** 0.5
