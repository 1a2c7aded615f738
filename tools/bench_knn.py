#!/usr/bin/env python3
"""Runs ON THE GPU BOX: pcops_knn_graph at the DGCNN config (256 x 2048, k = 20) -- 64-channel feature graphs and
coordinate graphs -- timed with events, checked bit-exact against the oracle on a few clouds, and (diagnostics build:
PCOPS_LIB=.../libpcops_knnstats.so) the fp16 filter's counters.  PCOPS_KNN_F16=0 selects the fp32-MFMA kernel."""
import ctypes
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from oracle import oracle as O  # noqa: E402
from scanobjectnn_amd import _lib  # noqa: E402
from scanobjectnn_amd.dgcnn import tf_util as td  # noqa: E402
from scanobjectnn_amd.synth import synth_clouds  # noqa: E402

B = int(os.environ.get("B", 256))
N = int(os.environ.get("N", 2048))
g = torch.Generator().manual_seed(3)
cases = {"feat64_relu": torch.relu(torch.randn(B, N, 64, generator=g)),
         "feat64_smooth": (torch.randn(B, N, 8, generator=g) @ torch.randn(8, 64, generator=g)).tanh(),   # low intrinsic dimension
         "xyz": torch.from_numpy(synth_clouds(B, N, seed=1234))}
lib = _lib.load()
has_stats = hasattr(lib, "pcops_knn_debug_stats")
for name, x in cases.items():
    xd = x.cuda().contiguous()
    nn = td.knn_graph(xd, k=20)
    torch.cuda.synchronize()
    if has_stats:
        out = (ctypes.c_ulonglong * 4)()
        lib.pcops_knn_debug_stats.argtypes = [ctypes.c_void_p]
        lib.pcops_knn_debug_stats(out)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 10
    e0.record()
    for _ in range(reps):
        td.knn_graph(xd, k=20)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / reps * 1e3
    pick = [0, B // 2, B - 1]
    ok = np.array_equal(nn[pick].cpu().numpy(), O.knn_graph(x[pick].numpy(), 20))
    c = x.shape[2]
    flops = 2.0 * B * N * N * c
    # priced against the pipe the distances run on: the 64-channel graphs' filter is fp16 MFMA
    # (2 500 TF/s dense), the coordinate graphs' distances fp32 MFMA (157.3 TF/s) -- and the kernel is bound by neither: the
    # selection is vector-instruction issue, so the pair rate against 9 lane-operations per pair test is printed too
    path = int(lib.pcops_knn_graph_path(B, N, c, 20, xd.data_ptr()))          # bits 0-3: 3 = fp16 pre-filter kernel
    peak, pipe = (2500.0, "fp16 MFMA") if (path & 15) == 3 else (157.3, "fp32 MFMA")
    pairs = float(B) * N * N
    line = "%-14s c=%-3d %8.1f us  %6.1f TF/s = %.4f of the %s peak (%.0f TF/s); %.2f T pair tests/s = %.3f of the fp32 VALU rate at 9 lane-ops per pair;  bit-exact vs oracle: %s" % (
        name, c, us, flops / us / 1e6, flops / us / 1e6 / peak, pipe, peak, pairs / us / 1e6, pairs * 9 / (us * 1e-6) / 78.6e12, ok)
    if has_stats and out[0]:
        line += "   pairs %d  survivors/query %.1f  accepted/query %.1f  rounds/wave %.1f" % (
            out[0], out[1] / (B * N), out[2] / (B * N), out[3] / (B * N / 32.0))
    print(line, flush=True)
