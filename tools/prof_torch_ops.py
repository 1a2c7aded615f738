"""Which Python lines launch the generic torch kernels of a training step (runs on the GPU box):
torch.profiler with stacks over ONE step, aten ops grouped by the innermost frame inside the repo.
usage: prof_torch_ops.py [model]"""
import collections
import importlib
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import ProfilerActivity, profile

import bench
from scanobjectnn_amd import train_util as TU
from scanobjectnn_amd.graph import Model
from scanobjectnn_amd.synth import synth_clouds, synth_labels, synth_masks

name = sys.argv[1] if len(sys.argv) > 1 else "dgcnn"
modpath, has_mask, B, N = bench.MODELS[name]
mod = importlib.import_module(modpath)
dev = "cuda:0"
x = torch.from_numpy(synth_clouds(B, N, seed=77)).to(dev)
y = torch.from_numpy(synth_labels(B, seed=77)).to(dev)
mask = torch.from_numpy(synth_masks(B, N, seed=77)).to(dev) if has_mask else None
net = Model(mod.get_model, device=dev, seed=0).build(x[:2].contiguous())
fp = TU.FlatParams(net)
opt = TU.TFAdam(fp)


def step(i):
    fp.begin_step()
    out = net(x, is_training=True, bn_decay=TU.get_bn_decay(i, B))
    loss = mod.get_loss(out[0], out[1], y, mask)[0] if has_mask else mod.get_loss(out[0], y, out[1])
    loss.backward()
    fp.collect()
    opt.step(TU.get_learning_rate(i, B))


for i in range(3):
    step(i)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step(3)
    torch.cuda.synchronize()
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
agg = collections.defaultdict(lambda: [0, 0.0])
for ev in prof.key_averages(group_by_stack_n=12):
    t = 0.0
    for attr in ("self_device_time_total", "self_cuda_time_total"):
        t = max(t, float(getattr(ev, attr, 0.0) or 0.0))
    if not ev.key.startswith("aten::") or t <= 0:
        continue
    where = "<autograd / unknown>"
    for fr in ev.stack:
        if root in fr and "tools/prof_torch_ops" not in fr:
            where = fr.replace(root + "/", "")
            break
    k = (ev.key, where[:120])
    agg[k][0] += ev.count
    agg[k][1] += t
tot = sum(v[1] for v in agg.values())
print("%s: %d generic ops with device time, %.1f us of device time in one step" % (name, sum(v[0] for v in agg.values()), tot))
for (op, where), (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:80]:
    print("%4d x %-26s %8.1f us  %s" % (n, op, t, where))
