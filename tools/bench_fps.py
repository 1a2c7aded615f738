"""FPS timing on the SSG SA1 shape (256 clouds x 2048 points -> 512 samples); runs on the GPU box."""
import sys, os, torch
sys.path.insert(0, os.getcwd())
from scanobjectnn_amd.pointnet2.tf_sampling import farthest_point_sample
from scanobjectnn_amd.synth import synth_clouds
x = torch.from_numpy(synth_clouds(256, 2048, seed=1)).cuda()
farthest_point_sample(512, x); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): idx = farthest_point_sample(512, x)
e1.record(); torch.cuda.synchronize()
print("%.1f us" % (e0.elapsed_time(e1) / 10 * 1e3), int(idx.sum()))
