"""Data parallelism for the training loop: one process per GPU, `torch.distributed` (backend "nccl"
== RCCL over xGMI on ROCm; "gloo" for the CPU tests), clouds sharded by rank, ONE all-reduce of the
flat fp32 gradient bucket per step (5.9-11.1 MB for the in-scope models).  The
reference has no distributed code at all; this is an addition.
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Initialise from RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* (torchrun).  Returns (rank, world, local)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def shard_range(total, rank, world):
    """rank r owns clouds [r*B/R, (r+1)*B/R)"""
    per = total // world
    return rank * per, (rank + 1) * per


def allreduce_mean_(flat_grad, world=None):
    """In-place mean over ranks of the flat gradient bucket (sum all-reduce, then 1/R)."""
    if not dist.is_initialized():
        return flat_grad
    world = world or dist.get_world_size()
    if world == 1:
        return flat_grad
    dist.all_reduce(flat_grad, op=dist.ReduceOp.SUM)
    flat_grad.mul_(1.0 / world)
    return flat_grad


def broadcast_(tensor, src=0):
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.broadcast(tensor, src=src)
    return tensor


def barrier():
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()


def max_over_ranks(value, device):
    if not (dist.is_initialized() and dist.get_world_size() > 1):
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
