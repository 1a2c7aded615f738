#!/usr/bin/env python3
"""Runs ON THE GPU BOX: the geometry kernels (FPS, ball query) of the SSG config alone, for rocprofv3 counter passes.
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU ... -- python tools/prof_geometry.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from scanobjectnn_amd.pointnet2 import tf_grouping, tf_sampling  # noqa: E402
from scanobjectnn_amd.synth import synth_clouds  # noqa: E402

B = int(os.environ.get("B", 256))
x = torch.from_numpy(synth_clouds(B, 2048, seed=1234)).cuda()
q = tf_sampling.gather_point(x, tf_sampling.farthest_point_sample(512, x))
q2 = tf_sampling.gather_point(q, tf_sampling.farthest_point_sample(128, q))
reps = int(os.environ.get("REPS", 5))
for name, fn in (("qbp_sa1", lambda: tf_grouping.query_ball_point(0.2, 32, x, q)),
                 ("qbp_sa2", lambda: tf_grouping.query_ball_point(0.4, 64, q, q2)),
                 ("fps_sa1", lambda: tf_sampling.farthest_point_sample(512, x)),
                 ("fps_sa2", lambda: tf_sampling.farthest_point_sample(128, q))):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    print("%s %.1f us" % (name, (time.perf_counter() - t0) / reps * 1e6))
