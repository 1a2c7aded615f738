"""Which Python lines issue the generic aten kernels of a training step (runs on the GPU box): a TorchDispatchMode around
ONE step (autograd single-threaded so that backward runs under it), ops grouped by the innermost frame inside the repo.
usage: trace_aten_ops.py [model]"""
import collections
import importlib
import os
import sys
import traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.utils._python_dispatch import TorchDispatchMode

import bench
from scanobjectnn_amd import train_util as TU
from scanobjectnn_amd.graph import Model
from scanobjectnn_amd.synth import synth_clouds, synth_labels, synth_masks

name = sys.argv[1] if len(sys.argv) > 1 else "pointnet2_cls_ssg"
modpath, has_mask, B, N = bench.MODELS[name]
mod = importlib.import_module(modpath)
dev = "cuda:0"
x = torch.from_numpy(synth_clouds(B, N, seed=77)).to(dev)
y = torch.from_numpy(synth_labels(B, seed=77)).to(dev)
mask = torch.from_numpy(synth_masks(B, N, seed=77)).to(dev) if has_mask else None
net = Model(mod.get_model, device=dev, seed=0).build(x[:2].contiguous())
fp = TU.FlatParams(net)
opt = TU.TFAdam(fp)
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
VIEWS = {"view", "reshape", "detach", "alias", "expand", "slice", "select", "t", "transpose", "unsqueeze", "squeeze",
         "as_strided", "empty", "empty_like", "empty_strided", "_unsafe_view", "permute", "split", "unbind", "narrow",
         "set_", "lift_fresh", "sym_size", "sym_stride", "sym_numel", "sym_storage_offset", "is_same_size", "new_empty"}


class Log(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.rows = collections.Counter()

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        nm = str(func)
        if nm.split(".")[1] not in VIEWS:
            where = "<autograd engine>"
            for fr in reversed(traceback.extract_stack()):
                if fr.filename.startswith(root) and "trace_aten_ops" not in fr.filename:
                    where = "%s:%d" % (fr.filename[len(root) + 1:], fr.lineno)
                    break
            shp = [tuple(a.shape) for a in args if isinstance(a, torch.Tensor)][:3]
            self.rows[(nm, where, str(shp))] += 1
        return out


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
with torch.autograd.set_multithreading_enabled(False):
    log = Log()
    with log:
        step(3)
torch.cuda.synchronize()
print("%s: %d aten calls that are not views in one step" % (name, sum(log.rows.values())))
for (nm, where, shp), c in sorted(log.rows.items(), key=lambda kv: (kv[0][1], kv[0][0])):
    print("%3d x %-34s %-52s %s" % (c, nm, where, shp))
