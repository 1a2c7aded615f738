"""Data-parallel path without a cluster: 2 processes, `gloo` backend, 127.0.0.1 rendezvous.  R ranks each take
their shard of the clouds; after ONE all-reduce of the flat gradient bucket every rank holds the gradient of the
whole batch and the TF-Adam step keeps the replicas bit-identical."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from scanobjectnn_amd import dist as D
from scanobjectnn_amd import train_util as TU


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _make_model():
    torch.manual_seed(0)
    return torch.nn.Sequential(torch.nn.Linear(6, 16), torch.nn.ReLU(), torch.nn.Linear(16, 3))


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    r, w, _ = D.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    torch.manual_seed(123)
    x, y = torch.randn(8, 6), torch.randint(0, 3, (8,))
    net = _make_model()
    if rank == 1:
        with torch.no_grad():
            for p in net.parameters():
                p.add_(1.0)                     # replicas start different: the broadcast must fix it
    fp = TU.FlatParams(net)
    D.broadcast_(fp.flat)
    opt = TU.TFAdam(fp)
    lo, hi = D.shard_range(8, rank, world)
    for step in range(3):
        fp.begin_step()                          # exactly bench.py's / train.py's step: fresh grads, one cat, one all-reduce
        # per-rank mean loss; mean over ranks of equal shards == mean over the global batch
        torch.nn.functional.cross_entropy(net(x[lo:hi]), y[lo:hi]).backward()
        D.allreduce_mean_(fp.collect(), world)
        opt.step(TU.get_learning_rate(step, 8))
    D.barrier()
    t = D.max_over_ranks(float(rank), torch.device("cpu"))
    # plain Python lists: a torch tensor travels as a shared-memory handle that dies with this process
    q.put((rank, fp.flat.tolist(), fp.grad.tolist(), t))
    dist.destroy_process_group()


def test_two_rank_gloo_equals_single_process():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in procs], key=lambda t: t[0])
    res = [(r, torch.tensor(a), torch.tensor(b), t) for r, a, b, t in res]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # single-process reference on the whole batch
    torch.manual_seed(123)
    x, y = torch.randn(8, 6), torch.randint(0, 3, (8,))
    net = _make_model()
    fp = TU.FlatParams(net)
    opt = TU.TFAdam(fp)
    for step in range(3):
        fp.zero_grad()
        torch.nn.functional.cross_entropy(net(x), y).backward()
        opt.step(TU.get_learning_rate(step, 8))
    assert torch.equal(res[0][1], res[1][1])                       # replicas identical
    assert torch.allclose(res[0][1], fp.flat, atol=1e-6)           # == single-process training
    assert torch.allclose(res[0][2], fp.grad, atol=1e-6)
    assert res[0][3] == 1.0 and res[1][3] == 1.0                   # max over ranks


def test_shard_range_partitions_the_batch():
    spans = [D.shard_range(1024, r, 8) for r in range(8)]
    assert spans[0] == (0, 128) and spans[-1] == (896, 1024)
    assert all(spans[i][1] == spans[i + 1][0] for i in range(7))


# ---------------------------------------------------------------------------------------------------------------
# The PRODUCT model object (graph.Model + FlatParams + TFAdam) under data parallelism: BASELINE config 1 (PointNet,
# pure dense layers, runs on the host) on 2 and 4 gloo ranks.
def _pointnet_setup(batch, npts):
    import numpy as np
    from scanobjectnn_amd.graph import Model
    from scanobjectnn_amd.pointnet import pointnet_cls
    from scanobjectnn_amd.synth import synth_clouds, synth_labels
    pointnet_cls.tf_util.dropout = lambda x, *a, **k: x            # dropout draws per-rank random masks: off
    x = torch.from_numpy(synth_clouds(batch, npts, seed=11))
    y = torch.from_numpy(synth_labels(batch, seed=11))
    net = Model(pointnet_cls.get_model, seed=0).build(x[:2])
    return pointnet_cls, net, x, y


def _model_worker(rank, world, port, q, sync_bn, batch, npts, steps, overlap=0, own_stat_group=False):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    torch.set_num_threads(1)
    D.STAT_GROUP_SEPARATE = own_stat_group          # PCOPS_STAT_GROUP=1: created eagerly by init_from_env on every rank
    D.init_from_env(backend="gloo")
    assert (D.stat_group() is not None) == own_stat_group
    D.SYNC_BN = sync_bn
    mod, net, x, y = _pointnet_setup(batch, npts)
    fp = TU.FlatParams(net)
    if overlap:
        fp.enable_overlap(world, nbuckets=overlap)
        assert len(fp._buckets) >= 2 and fp._buckets[0][2] == 0 and fp._buckets[-1][3] == fp.numel
    D.broadcast_(fp.flat)
    opt = TU.TFAdam(fp)
    lo, hi = D.shard_range(batch, rank, world)
    for step in range(steps):
        fp.begin_step()
        logits, ep = net(x[lo:hi], is_training=True, bn_decay=0.5)
        mod.get_loss(logits, y[lo:hi], ep, reg_weight=0.0).backward()
        if overlap:
            assert sum(1 for pend in fp._pending if pend == -1) >= len(fp._buckets) - 1   # issued DURING backward
        fp.collect_mean(world)
        if step == 0:
            grad0 = fp.grad.tolist()
        opt.step(1e-3)
    bufs_before = torch.cat([b.reshape(-1) for b in net.buffers()]).tolist()
    D.broadcast_buffers_(net)                                      # what the trainers do before a checkpoint
    bufs_after = torch.cat([b.reshape(-1) for b in net.buffers()]).tolist()
    q.put((rank, fp.flat.tolist(), bufs_before, bufs_after, grad0))
    dist.destroy_process_group()


def _run_model_dp(world, sync_bn, batch=8, npts=32, steps=2, overlap=0, own_stat_group=False):
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_model_worker, args=(r, world, port, q, sync_bn, batch, npts, steps, overlap, own_stat_group))
             for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return [(torch.tensor(a), torch.tensor(b), torch.tensor(c), torch.tensor(d)) for _, a, b, c, d in res]


def _single_process(batch=8, npts=32, steps=2):
    mod, net, x, y = _pointnet_setup(batch, npts)
    fp = TU.FlatParams(net)
    opt = TU.TFAdam(fp)
    for step in range(steps):
        fp.begin_step()
        logits, ep = net(x, is_training=True, bn_decay=0.5)
        mod.get_loss(logits, y, ep, reg_weight=0.0).backward()
        fp.collect()
        if step == 0:
            grad0 = fp.grad.clone()
        opt.step(1e-3)
    return fp.flat.clone(), torch.cat([b.reshape(-1) for b in net.buffers()]), grad0


def test_model_data_parallel_sync_bn_equals_single_process():
    """graph.Model + FlatParams + TFAdam on 2 and 4 ranks with SyncBN == one process on the whole batch: parameters
    after two Adam steps and the BN moving statistics (global-batch statistics on every rank)."""
    want_flat, want_bufs, want_grad0 = _single_process()
    for world in (2, 4):
        res = _run_model_dp(world, sync_bn=True)
        for flat, bufs, _, _ in res:
            assert torch.equal(flat, res[0][0])                              # replicas identical
            assert torch.allclose(bufs, res[0][1], atol=1e-6)                # and so are their moving statistics
        # the all-reduced gradient of the first step IS the gradient of the whole batch
        assert (res[0][3] - want_grad0).abs().max().item() <= 1e-5 * max(1.0, want_grad0.abs().max().item())
        # parameters after two Adam steps: Adam moves every parameter by ~lr whatever the gradient magnitude, so a
        # parameter whose gradient sits at rounding level may differ by O(lr); the bulk must agree
        assert (res[0][0] - want_flat).abs().mean().item() < 2e-5
        assert torch.allclose(res[0][1], want_bufs, atol=1e-3, rtol=1e-3)     # statistics of the global batch


def test_model_data_parallel_local_bn_and_buffer_broadcast():
    """without SyncBN: one all-reduce keeps the PARAMETERS of the replicas identical, the BN moving statistics are
    per rank (each saw its own shard) until broadcast_buffers_ makes them rank 0's"""
    res = _run_model_dp(2, sync_bn=False)
    assert torch.equal(res[0][0], res[1][0])
    assert not torch.allclose(res[0][1], res[1][1], atol=1e-6)               # shards differ -> statistics differ
    assert torch.equal(res[0][2], res[1][2]) and torch.equal(res[0][2], res[0][1])


def test_overlapped_bucket_allreduce_equals_the_single_collective():
    """`FlatParams.enable_overlap`: the flat bucket leaves in ranges, each all-reduced asynchronously as soon as the
    backward pass has produced it (checked: all but at most one range are under way when backward() returns) -- the
    same averaged gradient, bit for bit, and the same parameters as the one-collective step, with and without SyncBN
    (whose own collectives run inside the backward pass, interleaved with the ranges)."""
    for sync_bn in (False, True):
        plain = _run_model_dp(2, sync_bn=sync_bn)
        over = _run_model_dp(2, sync_bn=sync_bn, overlap=3)
        for r in range(2):
            assert torch.equal(over[r][3], plain[r][3])                      # first step's averaged gradient
            assert torch.equal(over[r][0], plain[r][0])                      # parameters after two Adam steps
        assert torch.equal(over[0][0], over[1][0])


def test_sync_bn_on_its_own_communicator_with_overlapped_ranges():
    """PCOPS_STAT_GROUP=1: the SyncBN statistics on a dedicated process group -- created eagerly by
    init_from_env() on every rank (a lazily created group hangs a job whose ranks do not all reach a SyncBN layer), and
    every exchange ordered behind the gradient ranges already in flight on the default group -- gives the same gradient
    and parameters as the one-communicator run, with the overlapped ranges enabled"""
    plain = _run_model_dp(2, sync_bn=True, overlap=3)
    own = _run_model_dp(2, sync_bn=True, overlap=3, own_stat_group=True)
    for r in range(2):
        assert torch.equal(own[r][3], plain[r][3]) and torch.equal(own[r][0], plain[r][0])
    assert torch.equal(own[0][0], own[1][0])


def test_overlap_with_a_parameter_the_loss_does_not_reach():
    """a range whose parameter never gets a gradient is issued by collect_mean() with zeros -- no hang, no stale data"""
    net = _make_model()
    extra = torch.nn.Linear(4, 4)                       # registered, never used
    both = torch.nn.ModuleList([net, extra])
    fp = TU.FlatParams(both)
    assert fp.enable_overlap(1) is fp and fp._buckets is None          # single rank: nothing to overlap
    fp.begin_step()
    x = torch.randn(5, 6)
    net(x).sum().backward()
    g = fp.collect_mean(1)
    assert g[-20:].abs().max().item() == 0.0 and g[:10].abs().max().item() > 0.0


def _stat_worker(rank, world, port, q):
    """each rank: shifted-moment partials of ITS shard about ITS OWN pivot -> dist.allreduce_stat_partials(part, rows, pivot)"""
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    D.init_from_env(backend="gloo")
    g = torch.Generator().manual_seed(0)
    y = 3.0 + 0.5 * torch.randn(64, 5, generator=g)                  # the whole batch; |mean| = 6 sigma
    rows = 64 // world
    mine = y[rank * rows:(rank + 1) * rows]
    pivot = torch.full((5,), 2.5 + 0.4 * rank)                       # a DIFFERENT pivot on every rank
    d = mine - pivot
    # P = 4 partial rows of (sum (y - pivot), sum (y - pivot)^2), fp32 like the kernels'
    part = torch.stack([torch.stack([c.sum(0), (c * c).sum(0)]) for c in d.chunk(4)]).float()
    glob, total = D.allreduce_stat_partials(part, rows, pivot)
    s = glob.double().sum(0)                                          # (2, C): plain sums of the GLOBAL batch, no pivot left
    q.put((rank, total, s.tolist()))
    D.dist.barrier()
    D.dist.destroy_process_group()


def test_sync_bn_statistics_with_a_different_pivot_on_every_rank():
    """the SyncBN exchange must not assume that all ranks hold the same pivot (moving mean).  Every rank takes its
    own pivot out of its shifted moments in float64 before the all-reduce (which runs on the statistics' own process
    group): what comes back are the plain sums of the global batch -- mean and variance right to 1e-6 although the ranks'
    pivots differ by 0.4 and the data sit 6 sigma from the origin."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_stat_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=60) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    g = torch.Generator().manual_seed(0)
    y = (3.0 + 0.5 * torch.randn(64, 5, generator=g)).double()
    for _, total, s in res:
        assert total == 64
        s = torch.tensor(s, dtype=torch.float64)
        mean = s[0] / total
        var = s[1] / total - mean * mean
        assert torch.allclose(mean, y.mean(0), rtol=0, atol=1e-6)
        assert torch.allclose(var, y.var(0, unbiased=False), rtol=1e-5, atol=1e-7)


def test_sync_forms_without_a_process_group(monkeypatch):
    """dist.SYNC_FORMS_LOCAL (the single-process reference of tests/test_dist_gpu.py): SyncBN's code paths are active, the
    exchange is the identity -- the shifted partials come back as un-shifted sums of this process's rows, split into an
    fp32 head and the float64 remainder, and the row count is the local one"""
    monkeypatch.setattr(D, "SYNC_BN", True)
    assert not D.sync_bn_active()
    monkeypatch.setattr(D, "SYNC_FORMS_LOCAL", True)
    assert D.sync_bn_active()
    g = torch.Generator().manual_seed(1)
    rows, c = 1000, 12
    y = torch.randn(rows, c, generator=g, dtype=torch.float64) * 3.0 + 5.0
    piv = torch.randn(c, generator=g).float() + 5.0
    chunks = y.view(10, 100, c)
    part = torch.stack([torch.stack([(ch - piv.double()).sum(0), ((ch - piv.double()) ** 2).sum(0)]) for ch in chunks]).float()
    out, n = D.allreduce_stat_partials(part, rows, piv)
    assert n == rows and out.shape == (2, 2, c)
    tot = out.double().sum(0)
    assert torch.allclose(tot[0], y.sum(0), rtol=1e-6) and torch.allclose(tot[1], (y * y).sum(0), rtol=1e-6)
    mean, var, total = D.sync_batch_stats(y.float())
    assert total == rows and torch.allclose(mean.double(), y.mean(0), atol=1e-5)
    assert torch.allclose(var.double(), y.var(0, unbiased=False), rtol=1e-4)
