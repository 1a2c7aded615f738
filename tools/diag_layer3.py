import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from scanobjectnn_amd import fused_mlp
from scanobjectnn_amd.graph import Model
from scanobjectnn_amd.pointnet2 import pointnet2_cls_ssg as m, tf_util
from scanobjectnn_amd.synth import synth_clouds, synth_labels
from test_models_parity_gpu import _randomise
from test_fused_mlp_gpu import reference, EPS
DEV = "cuda:0"
tf_util.dropout = lambda inputs, is_training, scope, keep_prob=0.5, noise_shape=None: inputs
c = synth_clouds(16, 512, seed=7); y = synth_labels(16)
x = torch.from_numpy(c).to(DEV)
net = Model(m.get_model, device=DEV, seed=3).build(x)
_randomise(net, 8)
rec = {}
orig = fused_mlp.mlp_stack
def spy(xx, S, pool, training, decay, eps, unbiased, layers):
    out = orig(xx, S, pool, training, decay, eps, unbiased, layers)
    if S == 128 and training:
        rec["x"] = xx.detach().clone(); rec["layers"] = [[t.detach().clone() for t in l] for l in layers]; rec["S"] = S
        out.register_hook(lambda g: rec.__setitem__("go", g.detach().clone()))
    return out
fused_mlp.mlp_stack = spy
logits, _ = net(x, is_training=True, bn_decay=0.9)
m.get_loss(logits, torch.from_numpy(y).to(DEV)).backward()
fused_mlp.mlp_stack = orig
xx, layers, go = rec["x"].reshape(-1, rec["x"].shape[-1]), rec["layers"], rec["go"]
print("x", xx.shape, "go", go.shape, "go absmax %.3e" % go.abs().max().item(), "frac zero x %.3f" % (xx == 0).float().mean().item())
res = {}
for mode in ("fused", "fp32", "fp64"):
    dt = torch.float64 if mode == "fp64" else torch.float32
    xi = xx.detach().to(dt).requires_grad_(True)
    ls = [[t.detach().to(dt).requires_grad_(True) for t in l[:4]] + [l[4].clone(), l[5].clone()] for l in layers]
    out = orig(xi, 128, True, True, 0.9, EPS, True, [tuple(l) for l in ls]) if mode == "fused" else reference(xi, ls, 128, True, True, dt)
    out.backward(go.reshape(out.shape).to(dt))
    res[mode] = [out.detach().double(), xi.grad.double()] + [t.grad.double() for l in ls for t in l[:4]]
names = ["out", "dx"] + ["L%d.%s" % (i, n) for i in range(3) for n in ("dW", "db", "dgamma", "dbeta")]
for i, n in enumerate(names):
    t = res["fp64"][i]; sc = t.abs().max().item() + 1e-30
    print("  %-10s scale %.3e  fused %.2e  torch-fp32 %.2e" % (n, sc, (res["fused"][i] - t).abs().max().item() / sc, (res["fp32"][i] - t).abs().max().item() / sc))
# per-channel statistics of layer outputs: |mean|/std
a = xx.double()
for li, l in enumerate(layers):
    yv = a @ l[0].double().reshape(-1, l[0].shape[-1]) + l[1].double()
    var, mean = torch.var_mean(yv, dim=0, unbiased=False)
    print("layer", li, "max |mean|/std %.2f  min var %.3e  max var %.3e" % ((mean.abs() / var.sqrt()).max().item(), var.min().item(), var.max().item()))
    a = torch.relu((yv - mean) * torch.rsqrt(var + EPS) * l[2].double() + l[3].double())

# ---- step-by-step: replay the backward of the captured stack with explicit checks of the L2 -> L1 dgrad
print("---- dgrad check on captured data")
from scanobjectnn_amd import _lib
lib = _lib.load()
class Ctx: pass
ctx = Ctx(); ctx.needs_input_grad = [True] + [False] * 40
flat = []
for l in layers: flat.extend([l[0].reshape(-1, l[0].shape[-1]), l[1], l[2], l[3], l[4].clone(), l[5].clone()])
out = fused_mlp.FusedMLPStack.forward(ctx, xx.contiguous(), None, None, None, None, None, None, 128, True, True, 0.9, EPS, True, 3, *flat)
a0, _, _, _, _, _, _, Ys, means, rstds, scales, shifts, Ws, gammas, argmax = ctx.saved
R = xx.shape[0]; G = R // 128
gpool = go.reshape(G, -1).contiguous()
# reference quantities in fp64
Y2 = Ys[2].double(); sc2, sh2 = scales[2][:1024].double(), shifts[2][:1024].double()
act = torch.relu(Y2 * sc2 + sh2).view(G, 128, -1)
am_ref = act.argmax(dim=1)
print("argmax mismatches vs torch.argmax:", (am_ref != argmax.long()).sum().item(), "of", am_ref.numel())
vals = act.gather(1, argmax.long().unsqueeze(1)).squeeze(1)
print("max-value mismatch:", (vals - act.amax(dim=1)).abs().max().item())
onehot = torch.zeros_like(act); onehot.scatter_(1, argmax.long().unsqueeze(1), 1.0)
gm = (onehot * gpool.double().unsqueeze(1) * (act > 0)).view(R, -1)
mu, rs, gam = means[2][:1024].double(), rstds[2][:1024].double(), layers[2][2].double()
dbeta = gm.sum(0); dgamma = ((gm * (Y2 - mu)).sum(0)) * rs
p = gam * rs; q = -p * rs * dgamma / R; t = -p * dbeta / R - q * mu
dY2 = p * gm + q * Y2 + t
Gprev_ref = dY2 @ Ws[2].double().t()
mask1 = (Ys[1].double() * scales[1][:512].double() + shifts[1][:512].double()) > 0
Gm1_ref = Gprev_ref * mask1
# device: coefficients then dgrad
P = lib.pcops_mlp_bwd_pool_stats_rows(G); part = torch.empty(P, 2, 1024, device=DEV)
_lib.call("pcops_mlp_pool_bwd_stats", G, 128, 1024, gpool.data_ptr(), argmax.data_ptr(), Ys[2].data_ptr(), scales[2].data_ptr(), shifts[2].data_ptr(), part.data_ptr())
ws = torch.empty(int(lib.pcops_mlp_reduce_workspace_bytes(1024)) // 8, dtype=torch.float64, device=DEV)
dg, db = torch.empty(1024, device=DEV), torch.empty(1024, device=DEV)
pv, qv, tv = torch.zeros(1024, device=DEV), torch.zeros(1024, device=DEV), torch.zeros(1024, device=DEV)
_lib.call("pcops_mlp_bn_bwd_coeffs", P, 1024, R, part.data_ptr(), ws.data_ptr(), layers[2][2].data_ptr(), means[2].data_ptr(), rstds[2].data_ptr(), dg.data_ptr(), db.data_ptr(), pv.data_ptr(), qv.data_ptr(), tv.data_ptr())
for nm, a_, b_ in (("dgamma", dg, dgamma), ("dbeta", db, dbeta), ("p", pv, p), ("q", qv, q), ("t", tv, t)):
    print("  coeff %-7s scale %.3e relerr %.2e" % (nm, b_.abs().max().item(), (a_.double() - b_).abs().max().item() / (b_.abs().max().item() + 1e-30)))
Wt = torch.empty(1024, 512, device=DEV); _lib.call("pcops_mlp_transpose", 512, 1024, Ws[2].data_ptr(), Wt.data_ptr())
Gprev = torch.empty(R, 512, device=DEV); P2 = lib.pcops_mlp_stats_rows(R); part2 = torch.full((P2, 2, 512), float("nan"), device=DEV)
_lib.call("pcops_mlp_gemm_dgrad", R, 1024, 512, None, Ys[2].data_ptr(), pv.data_ptr(), qv.data_ptr(), tv.data_ptr(), gpool.data_ptr(), argmax.data_ptr(), 128,
          scales[2].data_ptr(), shifts[2].data_ptr(), Wt.data_ptr(), Ys[1].data_ptr(), scales[1].data_ptr(), shifts[1].data_ptr(), Gprev.data_ptr(), part2.data_ptr())
err = (Gprev.double() - Gm1_ref).abs()
print("  Gm1 scale %.3e maxerr %.3e  rows with err>1e-3*scale: %d" % (Gm1_ref.abs().max().item(), err.max().item(), (err.max(1).values > 1e-3 * Gm1_ref.abs().max().item()).sum().item()))
print("  stats: sumG relerr %.2e  (ref scale %.3e)   nan in part2: %s" % ((part2[:, 0].double().sum(0) - Gm1_ref.sum(0)).abs().max().item() / Gm1_ref.sum(0).abs().max().item(), Gm1_ref.sum(0).abs().max().item(), torch.isnan(part2).any().item()))
print("  |sum G| / sum|G| =", (Gm1_ref.sum(0).abs().max() / Gm1_ref.abs().sum(0).max()).item())
# ---- layer 1 coefficients and wgrad
Y1 = Ys[1].double(); mu1, rs1, gam1 = means[1][:512].double(), rstds[1][:512].double(), layers[1][2].double()
dbeta1 = Gm1_ref.sum(0); dgamma1 = (Gm1_ref * (Y1 - mu1)).sum(0) * rs1
p1 = gam1 * rs1; q1 = -p1 * rs1 * dgamma1 / R; t1 = -p1 * dbeta1 / R - q1 * mu1
dY1 = p1 * Gm1_ref + q1 * Y1 + t1
A0 = torch.relu(Ys[0].double() * scales[0][:256].double() + shifts[0][:256].double())
dW1_ref = A0.t() @ dY1
dg1, db1 = torch.empty(512, device=DEV), torch.empty(512, device=DEV)
pv1, qv1, tv1 = torch.zeros(512, device=DEV), torch.zeros(512, device=DEV), torch.zeros(512, device=DEV)
_lib.call("pcops_mlp_bn_bwd_coeffs", P2, 512, R, part2.data_ptr(), ws.data_ptr(), layers[1][2].data_ptr(), means[1].data_ptr(), rstds[1].data_ptr(), dg1.data_ptr(), db1.data_ptr(), pv1.data_ptr(), qv1.data_ptr(), tv1.data_ptr())
for nm, a_, b_ in (("dgamma1", dg1, dgamma1), ("dbeta1", db1, dbeta1), ("p1", pv1, p1), ("q1", qv1, q1), ("t1", tv1, t1)):
    print("  coeff %-7s scale %.3e relerr %.2e" % (nm, b_.abs().max().item(), (a_.double() - b_).abs().max().item() / (b_.abs().max().item() + 1e-30)))
splits = lib.pcops_mlp_wgrad_splits(R, 256, 512); scratch = torch.empty(splits * (256 * 512 + 512), device=DEV)
dW1, dbb = torch.empty(256, 512, device=DEV), torch.empty(512, device=DEV)
_lib.call("pcops_mlp_wgrad", R, 256, 512, Ys[0].data_ptr(), 256, scales[0].data_ptr(), shifts[0].data_ptr(), Gprev.data_ptr(), Ys[1].data_ptr(),
          pv1.data_ptr(), qv1.data_ptr(), tv1.data_ptr(), None, None, 128, None, None, scratch.data_ptr(), dW1.data_ptr(), dbb.data_ptr())
print("  dW1 scale %.3e relerr %.2e" % (dW1_ref.abs().max().item(), (dW1.double() - dW1_ref).abs().max().item() / dW1_ref.abs().max().item()))
print("  res L1.dW vs this ref: %.2e ; fp64-autograd vs this ref: %.2e" % ((res["fused"][6] - dW1_ref).abs().max().item() / dW1_ref.abs().max().item(), (res["fp64"][6] - dW1_ref).abs().max().item() / dW1_ref.abs().max().item()))
mx = act.amax(dim=1, keepdim=True)
ties = ((act == mx) & (mx > 0)).sum(dim=1)
print("ties at positive max: groups x channels with >1 maximiser:", (ties > 1).sum().item(), "max multiplicity", ties.max().item())
xr = xx.double()
d = torch.cdist(xr.view(G, 128, -1), xr.view(G, 128, -1))
d = d + torch.eye(128, device=DEV, dtype=torch.float64) * 1e9
print("identical input rows within a cloud:", (d.min(dim=2).values == 0).sum().item(), "of", G * 128)
print("identical xyz rows:", ((xr.view(G,128,-1)[:, :, None, :3] - xr.view(G,128,-1)[:, None, :, :3]).abs().sum(-1) + torch.eye(128, device=DEV, dtype=torch.float64) * 1e9).min(dim=2).values.eq(0).sum().item())
# fp64 forward from x: where does its arg-max differ from the device one?
a = xx.double()
for li, l in enumerate(layers):
    yv = a @ l[0].double().reshape(-1, l[0].shape[-1]) + l[1].double()
    var, mean = torch.var_mean(yv, dim=0, unbiased=False)
    a = torch.relu((yv - mean) * torch.rsqrt(var + EPS) * l[2].double() + l[3].double())
a64 = a.view(G, 128, -1)
am64 = a64.argmax(dim=1)
diff = (am64 != argmax.long()) & (a64.amax(dim=1) > 0)
print("arg-max differs from the float64 forward in", diff.sum().item(), "of", diff.numel(), "(group, channel) pairs")
top2 = a64.topk(2, dim=1).values
gap = (top2[:, 0] - top2[:, 1])[diff]
print("top-2 gaps there:", gap[:8].tolist(), " |go| there:", gpool.double()[diff][:8].abs().tolist())
# ---- autograd fp64 with retained intermediates vs the hand formula
xi = xx.double().requires_grad_(True)
inter = {}
a = xi
for li, l in enumerate(layers):
    W = l[0].double().reshape(-1, l[0].shape[-1]).requires_grad_(True)
    yv = a @ W + l[1].double(); yv.retain_grad(); inter["Y%d" % li] = yv; inter["W%d" % li] = W
    var, mean = torch.var_mean(yv, dim=0, unbiased=False)
    a = torch.relu((yv - mean) * torch.rsqrt(var + EPS) * l[2].double() + l[3].double()); a.retain_grad(); inter["a%d" % li] = a
pooled = a.view(G, 128, -1).amax(dim=1)
pooled.backward(go.reshape(pooled.shape).double())
def rel(a_, b_): return (a_ - b_).abs().max().item() / (b_.abs().max().item() + 1e-30)
print("autograd vs formula:  d a2 (=gm before relu mask)", rel((onehot * gpool.double().unsqueeze(1)).view(R, -1), inter["a2"].grad))
print("                      d Y2", rel(dY2, inter["Y2"].grad), " d a1", rel(Gprev_ref, inter["a1"].grad), " d Y1", rel(dY1, inter["Y1"].grad), " dW1", rel(dW1_ref, inter["W1"].grad))
print("                      Y1 device vs fp64", rel(Ys[1].double(), inter["Y1"].detach()), " Y2", rel(Ys[2].double(), inter["Y2"].detach()), " mask1 flips", ((inter["a1"].detach() > 0) != mask1).sum().item(), " mask2 flips", ((inter["a2"].detach() > 0) != (act.view(R, -1) > 0)).sum().item())
