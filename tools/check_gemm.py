"""quick numerical check of pcops_mlp_gemm_fwd (plain / BN+ReLU prologue, with statistics) against torch float64"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scanobjectnn_amd import _lib
lib = _lib.load()
dev = "cuda:0"
g = torch.Generator().manual_seed(0)
for (M, K, N) in [(16384, 64, 64), (16384, 64, 128), (16389, 128, 256), (40000, 256, 64), (33000, 320, 1024), (9000, 72, 96)]:
    for pro in (False, True):
        X = torch.randn(M, K, generator=g).to(dev)
        W = (torch.randn(K, N, generator=g) / K ** 0.5).to(dev)
        b = torch.randn(N, generator=g).to(dev)
        sc = (torch.rand(K, generator=g) + 0.5).to(dev)
        sh = (0.3 * torch.randn(K, generator=g)).to(dev)
        piv = (0.2 * torch.randn(N, generator=g)).to(dev)
        Y = torch.full((M, N), float("nan"), device=dev)
        P = lib.pcops_mlp_stats_rows(M)
        part = torch.zeros(P, 2, N, device=dev)
        _lib.call("pcops_mlp_gemm_fwd", M, K, N, X.data_ptr(), K, sc.data_ptr() if pro else None, sh.data_ptr() if pro else None,
                  W.data_ptr(), b.data_ptr(), Y.data_ptr(), part.data_ptr(), piv.data_ptr())
        torch.cuda.synchronize()
        A = torch.relu(X.double() * sc.double() + sh.double()) if pro else X.double()
        want = A @ W.double() + b.double()
        err = (Y.double() - want).abs().max().item()
        s = part.double().sum(0)
        d = want - piv.double()
        e1 = (s[0] - d.sum(0)).abs().max().item() / M
        e2 = (s[1] - (d * d).sum(0)).abs().max().item() / M
        print("M=%6d K=%4d N=%4d pro=%d  max|Y-want| %.2e  stats %.1e %.1e" % (M, K, N, pro, err, e1, e2), flush=True)
        if err > 1e-3:
            bad = (Y.double() - want).abs() > 1e-3
            rows = bad.any(1).nonzero().flatten()
            cols = bad.any(0).nonzero().flatten()
            print("   bad elements %d of %d; rows %d (first %s) cols %d (first %s)" % (int(bad.sum()), bad.numel(), rows.numel(), rows[:12].tolist(), cols.numel(), cols[:12].tolist()))
            r0 = int(rows[0])
            print("   row", r0, "got", Y[r0, :6].tolist(), "want", want[r0, :6].tolist())
            # is the bad row some other row of want?
            dist = (want[:64, :].float() - Y[r0].float()).abs().max(1).values
            print("   closest row among first 64:", int(dist.argmin()), float(dist.min()))
