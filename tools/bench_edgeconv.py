"""Micro-benchmark of the DGCNN gather / scatter entry points at the cfg3 shapes (runs on the GPU box): the graph is a
REAL kNN graph of synthetic clouds (pcops_knn_graph), so the inverse lists have the real length distribution.
usage: bench_edgeconv.py [reps] [C ...]      under rocprofv3 --kernel-trace --stats for the per-kernel split"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scanobjectnn_amd import _lib
from scanobjectnn_amd.dgcnn import tf_util
from scanobjectnn_amd.synth import synth_clouds

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
widths = [int(a) for a in sys.argv[2:]] or [64, 128]
dev = "cuda:0"
lib = _lib.load()
b, n, k = 256, 2048, 20
x = torch.from_numpy(synth_clouds(b, n, seed=1234)).to(dev)
idx = tf_util.knn_graph(x, k=k)
m, S = n, k
G = b * m
P = lambda t: t.data_ptr() if t is not None else None


def timed(name, fn, nbytes):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print("%-34s %8.1f us  %6.0f GB/s algorithmic (%.0f MB)  hbm_frac %.2f" % (name, ms * 1e3, nbytes / ms / 1e6, nbytes / 1e6,
                                                                           nbytes / ms / 1e6 / 8000.0), flush=True)


for C in widths:
    g = torch.Generator(device=dev).manual_seed(C)
    Q = torch.randn(b, n, C, device=dev, generator=g)
    Ctr = torch.randn(b, m, C, device=dev, generator=g)
    gamma = torch.rand(C, device=dev, generator=g) + 0.5
    mm = torch.zeros(C, device=dev)
    SQ, qsel, ysel = (torch.empty(G, C, device=dev) for _ in range(3))
    arg = torch.empty(G, C, dtype=torch.uint8, device=dev)
    part = torch.empty(lib.pcops_edge_pool_fwd_stats_rows(b, n, m, S, C), 2, C, device=dev)
    f = 4 * G * C
    timed("edge_pool_fwd C=%d" % C,
          lambda: _lib.call("pcops_edge_pool_fwd", b, n, m, S, C, P(Q), P(Ctr), P(idx), P(gamma), P(SQ), P(qsel), P(arg),
                            P(part), P(mm)), 4 * f + G * C + 4 * G * S + 4 * b * n * C // m * m)
    sc, sh, p, q, t = (torch.randn(C, device=dev, generator=g) for _ in range(5))
    out = torch.empty(G, C, device=dev)
    _lib.call("pcops_edge_pool_out", G, C, P(qsel), P(Ctr), P(sc), P(sh), P(out), P(ysel))
    gpool = torch.randn(G, C, device=dev, generator=g)
    dQ, dCtr = torch.empty(b, n, C, device=dev), torch.empty(b, m, C, device=dev)
    wsp = torch.empty(int(lib.pcops_sa_scatter_workspace_bytes(b, n, m, S)) // 4, dtype=torch.int32, device=dev)
    timed("edge_pool_bwd C=%d" % C,
          lambda: _lib.call("pcops_edge_pool_bwd", b, n, m, S, C, P(Q), P(Ctr), P(idx), P(gpool), P(ysel), P(SQ), P(arg),
                            P(sc), P(sh), P(p), P(q), P(t), P(dQ), P(dCtr), P(wsp)), 7 * f + G * C + 4 * G * S)
    if C != 64:
        continue
    R = G * S
    Y = torch.empty(R, C, device=dev)
    rows = lib.pcops_sa_gather_fwd_stats_rows(b, n, m, S, C, 1, 1, 0, 0)
    part2 = torch.empty(rows, 2, C, device=dev)
    timed("sa_gather_fwd (Q+Ctr) C=%d" % C,
          lambda: _lib.call("pcops_sa_gather_fwd", b, n, m, S, C, P(Q), P(Ctr), None, None, None, None, P(idx), P(Y), None,
                            P(part2), P(mm), None), 4 * R * C + 2 * f + 4 * G * S)
    Gm = torch.randn(R, C, device=dev, generator=g)
    timed("sa_scatter_bwd (Q+Ctr) C=%d" % C,
          lambda: _lib.call("pcops_sa_scatter_bwd", b, n, m, S, C, P(Gm), P(Y), P(p), P(q), P(t), None, None, None, None,
                            P(idx), None, None, P(dQ), P(dCtr), None, None, None, P(Q), P(Ctr), None, None, P(wsp)),
          2 * 4 * R * C + 4 * f + 4 * G * S)
    del Y, Gm
