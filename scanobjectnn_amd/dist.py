"""Data parallelism for the training loop: one process per GPU, `torch.distributed` (backend "nccl"
== RCCL over xGMI on ROCm; "gloo" for the CPU tests), clouds sharded by rank, ONE all-reduce of the
flat fp32 gradient bucket per step (5.9-11.1 MB for the in-scope models).  The
reference has no distributed code at all; this is an addition.
"""
import os

import torch
import torch.distributed as dist

# Optional SyncBN: when True and a process group with more than one rank exists, every BatchNorm
# layer all-reduces its per-channel batch statistics (forward: sum, sum of squares; backward: the two BN-backward
# sums) so that training normalises with the statistics of the GLOBAL batch, i.e. reproduces single-device batch
# statistics at the global batch size.  Off: statistics are per rank (plain data parallelism) and the moving
# averages differ between ranks until `broadcast_buffers_` is called (the trainers do so before every checkpoint).
SYNC_BN = False
# Test hook: a SINGLE process takes the kernel forms of the SyncBN mode (statistics finalised from the exchanged sums, no
# compacted rows, a stored first layer) with the exchange itself left out -- the reference a multi-rank SyncBN run is
# compared with row for row (tests/test_dist_gpu.py): what then differs is the exchange and nothing else.
SYNC_FORMS_LOCAL = False


def _multi_rank():
    return dist.is_initialized() and dist.get_world_size() > 1


def sync_bn_active():
    return SYNC_BN and (_multi_rank() or SYNC_FORMS_LOCAL)


_STAT_GROUP = None
# PCOPS_STAT_GROUP=1: the SyncBN statistics travel on their OWN communicator (so that the small, latency-critical
# all-reduces inside forward / backward do not queue behind the multi-MB gradient ranges `FlatParams.enable_overlap`
# issues from autograd hooks).  Off by default: two RCCL communicators in flight at once are only safe if
# every rank's GPU schedules their kernels in the same order, which nothing guarantees when the gradient ranges run
# asynchronously -- on ONE communicator the issue order (identical on every rank: same graph, same hook order) IS the
# execution order.  When requested, the group is created EAGERLY in init_from_env(), where every rank is known to
# arrive (new_group() is a world collective; a lazily created group hangs the job if some rank's first step never
# reaches a SyncBN layer), and each statistics exchange first waits for the gradient ranges already under way
# (`pending_work`), which orders the two communicators explicitly.
STAT_GROUP_SEPARATE = os.environ.get("PCOPS_STAT_GROUP", "0") == "1"
pending_work = []        # async gradient all-reduces in flight on the default group (FlatParams registers / clears them)


def stat_group():
    """The communicator of the SyncBN statistics: None = the default group, or the dedicated one init_from_env() made"""
    return _STAT_GROUP


def _before_stat_exchange():
    if _STAT_GROUP is not None:
        for w in pending_work:
            w.wait()


def allreduce_stat_partials(part, rows, pivot=None):
    """(P, 2, C) fp32 partial column sums of this rank -> ((2, 2, C) global sums, global row count).
    The P partials are added in float64 (as the finalisation kernels do), all-reduced in float64, and handed back as
    TWO fp32 partial rows -- the total rounded to fp32 and what the rounding left over -- which the finalisation kernels
    add in float64 again: the global sums reach them with ~48 bits, like a single rank's.
    pivot (C,): the partials are SHIFTED moments, sum (y - pivot) and sum (y - pivot)^2 (pcops.h).  Each rank's pivot is
    its own moving mean, and nothing guarantees those are bit-identical across ranks (per-rank restore, buffers broadcast
    only at epoch ends), so the shift is taken out in float64 BEFORE the all-reduce -- the caller then finalises the
    returned sums with no pivot."""
    tot = part.double().sum(dim=0, keepdim=True)
    if pivot is not None:
        c = tot.shape[-1]
        p = pivot.detach().double()[:c]
        tot[0, 1] += 2.0 * p * tot[0, 0] + float(rows) * p * p
        tot[0, 0] += float(rows) * p
    if _multi_rank():
        _before_stat_exchange()
        dist.all_reduce(tot, op=dist.ReduceOp.SUM, group=stat_group())
    hi = tot.float()
    lo = (tot - hi.double()).float()
    return torch.cat([hi, lo], dim=0).contiguous(), rows * (dist.get_world_size() if _multi_rank() else 1)


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
    global _STAT_GROUP
    if world > 1 and STAT_GROUP_SEPARATE and _STAT_GROUP is None:
        _STAT_GROUP = dist.new_group()          # a world collective: here, where every rank arrives
    return rank, world, local


def shard_range(total, rank, world):
    """rank r owns clouds [r*B/R, (r+1)*B/R)"""
    per = total // world
    return rank * per, (rank + 1) * per


def allreduce_mean_(flat_grad, world=None, force=False):
    """In-place mean over ranks of the flat gradient bucket (sum all-reduce, then 1/R).  force: issue the collective
    even for a single rank (the RCCL smoke test)."""
    if not dist.is_initialized():
        return flat_grad
    world = world or dist.get_world_size()
    if world == 1 and not force:
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


def sync_batch_stats(flat):
    """(rows, C) -> per-channel (mean, biased variance, global row count) over the rows of ALL ranks, differentiable:
    the all-reduce of (sum, sum of squares) is autograd-aware (its backward is the all-reduce of the incoming
    gradients), which yields exactly the cross-rank terms of the SyncBN backward.  For the layers that do not go
    through the fused kernels (FC head, small stacks)."""
    import torch.distributed.nn.functional as dfn
    n = flat.shape[0]
    s = torch.stack([flat.sum(dim=0), (flat * flat).sum(dim=0)])
    if _multi_rank():
        _before_stat_exchange()
        s = dfn.all_reduce(s, op=dist.ReduceOp.SUM, group=stat_group())
    total = n * (dist.get_world_size() if _multi_rank() else 1)
    mean = s[0] / total
    var = (s[1] / total - mean * mean).clamp_min(0.0)
    return mean, var, total


def gather_floats(value, device):
    """[value on rank 0, value on rank 1, ...] on every rank"""
    if not (dist.is_initialized() and dist.get_world_size() > 1):
        return [float(value)]
    t = torch.tensor([value], dtype=torch.float64, device=device)
    out = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return [float(o.item()) for o in out]


def broadcast_buffers_(module, src=0):
    """Make the non-trainable state (BN moving mean / variance) of every replica that of rank `src`: without SyncBN
    each rank tracks the statistics of its own shard, and a checkpoint must not depend on which rank wrote it."""
    if not (dist.is_initialized() and dist.get_world_size() > 1):
        return
    bufs = [b for b in module.buffers() if b.is_floating_point()]
    if not bufs:
        return
    flat = torch.cat([b.reshape(-1) for b in bufs])
    dist.broadcast(flat, src=src)
    off = 0
    for b in bufs:
        k = b.numel()
        b.copy_(flat[off:off + k].view_as(b))
        off += k
