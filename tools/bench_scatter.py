"""Micro-benchmark of pcops_sa_scatter_bwd on the SSG / DGCNN shapes (runs on the GPU box).
usage: bench_scatter.py [reps]"""
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scanobjectnn_amd import _lib

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
dev = "cuda:0"
lib = _lib.load()
CASES = [  # b, n, m, S, C, has_dq, has_xyz, pooled
    (256, 512, 128, 64, 128, True, True, False),    # SSG SA2
    (256, 512, 128, 64, 128, False, True, False),   # ... streaming part only
    (256, 512, 128, 64, 128, True, False, False),   # ... no coordinate term
    (256, 2048, 512, 32, 64, False, True, False),   # SSG SA1 (no features)
    (256, 2048, 512, 32, 64, True, True, False),    # SA1 with features
    (64, 2048, 2048, 20, 64, True, False, True),    # EdgeConv, single pooled layer
]
for (b, n, m, S, C, has_dq, has_xyz, pooled) in CASES:
    g = torch.Generator(device=dev).manual_seed(1)
    R = b * m * S
    Y = torch.randn(R, C, device=dev, generator=g)
    G = torch.randn(R, C, device=dev, generator=g)
    idx = torch.randint(0, n, (b, m, S), device=dev, generator=g, dtype=torch.int32)
    xyz = torch.rand(b, n, 3, device=dev, generator=g)
    new_xyz = torch.rand(b, m, 3, device=dev, generator=g)
    p, q, t = (torch.randn(C, device=dev, generator=g) for _ in range(3))
    gpool = torch.randn(b * m, C, device=dev, generator=g)
    argmax = torch.randint(0, S, (b * m, C), device=dev, generator=g, dtype=torch.int32).to(torch.uint8)
    dQ = torch.empty(b, n, C, device=dev) if has_dq else None
    dCtr = torch.empty(b, m, C, device=dev) if not has_xyz else None
    wpart = torch.empty(lib.pcops_sa_scatter_rows(b, m) * 4 * C, device=dev)
    dW, db = torch.empty(3, C, device=dev), torch.empty(C, device=dev)
    wsp = torch.empty(int(lib.pcops_sa_scatter_workspace_bytes(b, n, m, S)) // 4, dtype=torch.int32, device=dev)
    P = lambda x: x.data_ptr() if x is not None else None

    def run():
        _lib.call("pcops_sa_scatter_bwd", b, n, m, S, C, None if pooled else G.data_ptr(), Y.data_ptr(), p.data_ptr(),
                  q.data_ptr(), t.data_ptr(), gpool.data_ptr() if pooled else None, argmax.data_ptr() if pooled else None,
                  p.data_ptr() if pooled else None, q.data_ptr() if pooled else None, idx.data_ptr(),
                  P(xyz) if has_xyz else None, P(new_xyz) if has_xyz else None, P(dQ), P(dCtr), wpart.data_ptr(),
                  dW.data_ptr() if has_xyz else None, db.data_ptr(), None, None, None, None, wsp.data_ptr())
    run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        run()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    by = 4 * R * C * (1 if pooled else 2)
    print("b=%d n=%d m=%d S=%d C=%d dq=%d xyz=%d pooled=%d : %8.1f us  %6.0f GB/s (Y,G reads)" % (
        b, n, m, S, C, has_dq, has_xyz, pooled, ms * 1e3, by / ms / 1e6))
