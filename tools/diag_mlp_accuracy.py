"""diagnostic (GPU): error of the fused MLP stack vs a float64 reference, next to a plain torch fp32 autograd
implementation of the same chain, per gradient tensor."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scanobjectnn_amd import fused_mlp
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from test_fused_mlp_gpu import make_layers, reference, EPS, DEV

def run(R, S, K0, widths, pool, bias_scale, xoff=0.0):
    g = torch.Generator().manual_seed(R + 7)
    x0 = (torch.randn(R, K0, generator=g) * 0.1 + xoff).to(DEV)
    layers = make_layers(K0, widths, seed=K0 + 1)
    for l in layers:
        l[1].mul_(bias_scale)
    go = None
    res = {}
    for mode in ("fused", "fp32", "fp64"):
        dt = torch.float64 if mode == "fp64" else torch.float32
        x = x0.detach().to(dt).requires_grad_(True)
        ls = [[t.detach().to(dt).requires_grad_(True) for t in l[:4]] + [l[4].clone(), l[5].clone()] for l in layers]
        if mode == "fused":
            out = fused_mlp.mlp_stack(x, S, pool, True, 0.9, EPS, True, [tuple(l) for l in ls])
        else:
            out = reference(x, ls, S, pool, True, dt)
        if go is None:
            go = torch.randn(out.shape, generator=g).to(DEV)
        out.backward(go.to(dt))
        res[mode] = [out.detach().double(), x.grad.double()] + [t.grad.double() for l in ls for t in l[:4]]
    names = ["out", "dx"] + ["L%d.%s" % (i, n) for i in range(len(layers)) for n in ("dW", "db", "dgamma", "dbeta")]
    print("case R=%d S=%d K0=%d widths=%s pool=%s bias_scale=%g" % (R, S, K0, widths, pool, bias_scale))
    for i, n in enumerate(names):
        t = res["fp64"][i]
        sc = t.abs().max().item() + 1e-30
        print("  %-10s scale %.3e  fused %.2e  torch-fp32 %.2e" % (
            n, sc, (res["fused"][i] - t).abs().max().item() / sc, (res["fp32"][i] - t).abs().max().item() / sc))

if len(sys.argv) == 1:
  run(512 * 32 * 4, 32, 3, [64, 64, 128], True, 1.0)
  run(512 * 32 * 4, 32, 3, [64, 64, 128], True, 20.0)
  run(128 * 64 * 4, 64, 131, [128, 128, 256], True, 1.0)
  run(4096, 1, 384, [256, 128], False, 1.0)
if len(sys.argv) > 1 and sys.argv[1] == "ragged":
    run(512 * 64 + 37, 1, 128, [128, 256], False, 1.0)
    run(512 * 64, 1, 128, [128, 256], False, 1.0)
    run(512 * 64 + 37, 1, 128, [128], False, 1.0)
if len(sys.argv) > 1 and sys.argv[1] == "sa3":
    run(16 * 128, 128, 259, [256, 512, 1024], True, 1.0)
    run(16 * 128, 128, 260, [256, 512, 1024], True, 1.0)
    run(16 * 128, 128, 256, [512, 1024], True, 1.0)
