"""Kernel-level A/B of the statistics pivot: relative Frobenius error of every gradient of the fused stacks (dense
CASES and gather / EdgeConv cases of tests/test_fused_mlp_gpu.py) against float64 with the imposed activation pattern,
with the pivot on and off."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import mlp_ref as MR  # noqa: E402
import test_fused_mlp_gpu as T  # noqa: E402
from scanobjectnn_amd import fused_mlp  # noqa: E402

DEV = "cuda:0"


def dense_case(R, S, K0, widths, pool):
    g = torch.Generator().manual_seed(R + 7)
    x0 = torch.randn(R, K0, generator=g).to(DEV)
    res = {}
    for piv in (True, False):
        fused_mlp.STAT_PIVOT = piv
        x = x0.clone().requires_grad_(True)
        layers = T.make_layers(K0, widths, seed=K0 + 1)
        for l in layers:
            for t in l[:4]:
                t.requires_grad_(True)
        mov = [(l[4].clone(), l[5].clone()) for l in layers]
        out = fused_mlp.mlp_stack(x, S, pool, True, 0.9, T.EPS, True, [tuple(l) for l in layers])
        pattern = MR.fused_pattern(out)
        go = torch.randn(out.shape, generator=torch.Generator().manual_seed(3)).to(DEV)
        out.backward(go)
        got = [x.grad.double()] + [t.grad.double() for l in layers for t in l[:4]]
        xr = x0.double().requires_grad_(True)
        lr = [[t.detach().double().requires_grad_(True) for t in l[:4]] + list(mb) for l, mb in zip(layers, mov)]
        o = MR.run_stack(None, xr, lr, S, pool, True, torch.float64, pattern)
        o.backward(go.double())
        want = [xr.grad] + [t.grad for l in lr for t in l[:4]]
        num = sum(((a - b).norm() ** 2).item() for a, b in zip(got, want))
        den = sum((b.norm() ** 2).item() for b in want)
        worst = max(((a - b).norm() / b.norm().clamp_min(1e-30)).item() for a, b in zip(got, want))
        res[piv] = ((num / den) ** 0.5, worst, (out.double() - o.detach()).abs().max().item())
    fused_mlp.STAT_PIVOT = True
    return res


def gather_case(B, N, M, S, widths, pool, form):
    g = torch.Generator().manual_seed(B * 1000 + N)
    C1 = widths[0]
    src0 = {
        "Q": torch.randn(B, N, C1, generator=g).to(DEV) if form != "xyz_bias" else None,
        "Ctr": torch.randn(B, M, C1, generator=g).to(DEV) if form == "q_ctr" else None,
        "xyz": torch.rand(B, N, 3, generator=g).to(DEV) if form != "q_ctr" else None,
        "new_xyz": torch.rand(B, M, 3, generator=g).to(DEV) if form != "q_ctr" else None,
        "wxyz": torch.randn(3, C1, generator=g).to(DEV) if form != "q_ctr" else None,
        "bias": torch.randn(C1, generator=g).to(DEV) if form == "xyz_bias" else None,
    }
    idx = torch.randint(0, N, (B, M, S), generator=g, dtype=torch.int32).to(DEV)
    diff = ("Q", "Ctr", "wxyz", "bias")
    res = {}
    for piv in (True, False):
        fused_mlp.STAT_PIVOT = piv
        layers = T.make_layers(C1, widths, seed=N)

        def leaves(dt):
            s = {k: (v.detach().to(dt).requires_grad_(k in diff) if v is not None else None) for k, v in src0.items()}
            ls = [[t.detach().to(dt).requires_grad_(True) for t in l[:4]] + [l[4].clone(), l[5].clone()] for l in layers]
            return s, ls
        s, ls = leaves(torch.float32)
        out = fused_mlp.gather_mlp_stack(idx, pool, True, 0.9, T.EPS, True, [tuple(l) for l in ls], Q=s["Q"], Ctr=s["Ctr"],
                                         xyz=s["xyz"], new_xyz=s["new_xyz"], wxyz=s["wxyz"], bias=s["bias"])
        pattern = MR.fused_pattern(out)
        go = torch.randn(out.shape, generator=torch.Generator().manual_seed(5)).to(DEV)
        out.backward(go)
        got = [s[k].grad.double() for k in diff if s[k] is not None] + \
              [t.grad.double() for li, l in enumerate(ls) for ti, t in enumerate(l[:4]) if not (li == 0 and ti < 2)]
        s64, l64 = leaves(torch.float64)
        y1 = MR.gather_first_layer(s64["Q"], s64["Ctr"], s64["xyz"], s64["new_xyz"], s64["wxyz"], s64["bias"], idx, torch.float64)
        o = MR.run_stack(y1, None, l64, S, pool, True, torch.float64, pattern)
        o.backward(go.double())
        want = [s64[k].grad for k in diff if s64[k] is not None] + \
               [t.grad for li, l in enumerate(l64) for ti, t in enumerate(l[:4]) if not (li == 0 and ti < 2)]
        num = sum(((a - b).norm() ** 2).item() for a, b in zip(got, want))
        den = sum((b.norm() ** 2).item() for b in want)
        worst = max(((a - b).norm() / b.norm().clamp_min(1e-30)).item() for a, b in zip(got, want))
        res[piv] = ((num / den) ** 0.5, worst, (out.double() - o.detach()).abs().max().item())
    fused_mlp.STAT_PIVOT = True
    return res


if __name__ == "__main__":
    for c in T.CASES:
        r = dense_case(*c)
        print("dense %-42s pivot %.2e (worst %.2e, fwd %.1e) | off %.2e (worst %.2e, fwd %.1e)" %
              ((str(c),) + r[True] + r[False]), flush=True)
    for c in T.GATHER_CASES:
        for form in ("q_ctr", "xyz_bias", "q_xyz"):
            r = gather_case(*c, form)
            print("gather %-36s %-8s pivot %.2e (worst %.2e, fwd %.1e) | off %.2e (worst %.2e, fwd %.1e)" %
                  ((str(c), form) + r[True] + r[False]), flush=True)
