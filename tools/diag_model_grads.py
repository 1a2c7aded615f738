"""diagnostic (GPU): per-parameter gradient error of pointnet2_cls_ssg, fused path vs layer-wise torch path,
each against the float64 CPU restatement."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from oracle import ref_models as R
from scanobjectnn_amd.graph import Model
from scanobjectnn_amd.pointnet2 import pointnet2_cls_ssg as m, tf_util
from scanobjectnn_amd.synth import synth_clouds, synth_labels
from test_models_parity_gpu import _randomise
DEV = "cuda:0"
B, N = int(sys.argv[1]) if len(sys.argv) > 1 else 16, int(sys.argv[2]) if len(sys.argv) > 2 else 512
tf_util.dropout = lambda inputs, is_training, scope, keep_prob=0.5, noise_shape=None: inputs
c = synth_clouds(B, N, seed=7); y = synth_labels(B)
x = torch.from_numpy(c).to(DEV)
net = Model(m.get_model, device=DEV, seed=3).build(x)
_randomise(net, 8)
sd = {k: v.clone() for k, v in net.state_dict().items()}
P = {k: v.requires_grad_(v.is_floating_point()) for k, v in R.params_from_state_dict(sd, dtype=torch.float64).items()}
want = R.pointnet2_cls_ssg(torch.from_numpy(c).double(), P, True)
torch.nn.functional.cross_entropy(want, torch.from_numpy(y).long()).backward()
res = {}
for fused in (True, False):
    tf_util.FUSED_MLP = fused
    net.load_state_dict(sd); net.zero_grad()
    logits, _ = net(x, is_training=True, bn_decay=0.9)
    m.get_loss(logits, torch.from_numpy(y).to(DEV)).backward()
    res[fused] = ({n: p.grad.cpu().double() for n, p in net.named_parameters()}, logits.detach().cpu().double())
print("logits err fused %.2e layerwise %.2e" % ((res[True][1] - want.detach()).abs().max(), (res[False][1] - want.detach()).abs().max()))
for n in res[True][0]:
    ref = P[n[len("graph."):]].grad
    sc = ref.abs().max().item() + 1e-30
    print("%-34s scale %.2e  fused %.2e  layerwise %.2e" % (n[6:], sc, (res[True][0][n] - ref).abs().max().item() / sc, (res[False][0][n] - ref).abs().max().item() / sc))
