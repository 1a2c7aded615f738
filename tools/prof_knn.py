#!/usr/bin/env python3
"""Runs ON THE GPU BOX: the DGCNN graph kernels alone (kNN on 3 and 64 channels, B=256, N=2048, k=20) for rocprofv3."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from scanobjectnn_amd.dgcnn import tf_util as T  # noqa: E402
from scanobjectnn_amd.synth import synth_clouds  # noqa: E402

B = int(os.environ.get("B", 256))
x3 = torch.from_numpy(synth_clouds(B, 2048, seed=1234)).cuda()
x64 = torch.randn(B, 2048, 64, device="cuda")
for name, x in (("knn_c3", x3), ("knn_c64", x64)):
    T.knn_graph(x, k=20)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        T.knn_graph(x, k=20)
    torch.cuda.synchronize()
    print("%s %.1f us" % (name, (time.perf_counter() - t0) / 3 * 1e6))
