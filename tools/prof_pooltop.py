"""Per-call HIP-event timings of the algebraic top-layer backward (fused_mlp._pool_top_backward) against the plain
dgrad + wgrad it replaces, at the benchmark shapes.  python tools/prof_pooltop.py"""
import collections
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scanobjectnn_amd import _lib, fused_mlp  # noqa: E402

DEV = "cuda:0"
SHAPES = [("SSG SA3", 256 * 128, 128, 256, [256, 512, 1024]), ("DGCNN agg", 256 * 2048, 256, 320, [1024]),
          ("DGCNN tconv3", 256 * 2048, 256, 128, [1024])]


def layers(k0, widths):
    g = torch.Generator().manual_seed(k0)
    out, k = [], k0
    for n in widths:
        out.append([(torch.randn(k, n, generator=g) / k ** 0.5).to(DEV).requires_grad_(True),
                    (torch.randn(n, generator=g) * 0.1).to(DEV).requires_grad_(True),
                    (1 + 0.1 * torch.randn(n, generator=g)).to(DEV).requires_grad_(True),
                    (0.1 * torch.randn(n, generator=g)).to(DEV).requires_grad_(True),
                    torch.zeros(n, device=DEV), torch.ones(n, device=DEV)])
        k = n
    return out


class Timer:
    def __init__(self):
        self.t = collections.OrderedDict()
        self.open = None

    def __call__(self, name, phase, args):
        ev = torch.cuda.Event(enable_timing=True)
        ev.record()
        if phase == "pre":
            self.open = ev
        else:
            self.t.setdefault(name, []).append((self.open, ev))

    def report(self):
        torch.cuda.synchronize()
        tot = 0.0
        for k, v in self.t.items():
            ms = sum(a.elapsed_time(b) for a, b in v) / len(v)
            tot += ms * 1e3
            print("    %-34s x%d %8.1f us" % (k, len(v), ms * 1e3))
        print("    sum %.1f us" % tot)


for name, R, S, k0, widths in SHAPES:
    x = torch.randn(R, k0, device=DEV).requires_grad_(True)
    ls = layers(k0, widths)
    for mode in (True, False):
        fused_mlp.POOL_TOP = mode
        for it in range(3):
            out = fused_mlp.mlp_stack(x, S, True, True, 0.9, 1e-3, True, [tuple(l) for l in ls])
            go = torch.randn_like(out)
            tm = Timer()
            if it == 2:
                _lib._hooks.append(tm)
            out.backward(go)
            if it == 2:
                _lib._hooks.remove(tm)
                print("%s rows %d S %d %d -> %s  %s" % (name, R, S, k0, widths, "algebraic" if mode else "plain"))
                tm.report()
