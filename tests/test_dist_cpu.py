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
