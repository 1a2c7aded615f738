import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scanobjectnn_amd import _lib
lib = _lib.load()
dev = "cuda:0"
torch.manual_seed(0)
def vec(n): return torch.randn((n + 3) // 4 * 4, device=dev)
for (M, K, N) in [(32805, 128, 128), (32805, 256, 128), (32805, 128, 256), (32768 + 32, 128, 128), (32805, 64, 64)]:
    G = torch.randn(M, K, device=dev); Y = torch.randn(M, K, device=dev)
    p, q, t = vec(K), vec(K), vec(K)
    Wt = torch.randn(K, N, device=dev) / K ** 0.5
    Yprev = torch.randn(M, N, device=dev); sc, sh = vec(N), vec(N)
    dY = p[:K] * G + q[:K] * Y + t[:K]
    want = (dY.double() @ Wt.double())
    # plain
    out = torch.empty(M, N, device=dev)
    _lib.call("pcops_mlp_gemm_dgrad", M, K, N, G.data_ptr(), Y.data_ptr(), p.data_ptr(), q.data_ptr(), t.data_ptr(),
              None, None, 1, None, None, Wt.data_ptr(), None, None, None, out.data_ptr(), None)
    err = (out.double() - want).abs()
    bad = (err.max(dim=1).values > 1e-3).nonzero().flatten()
    print("plain", M, K, N, "maxerr %.3e" % err.max().item(), "bad rows:", bad[:10].tolist(), len(bad))
    # masked + stats
    P = lib.pcops_mlp_stats_rows(M)
    part = torch.full((P, 2, N), float("nan"), device=dev)
    _lib.call("pcops_mlp_gemm_dgrad", M, K, N, G.data_ptr(), Y.data_ptr(), p.data_ptr(), q.data_ptr(), t.data_ptr(),
              None, None, 1, None, None, Wt.data_ptr(), Yprev.data_ptr(), sc.data_ptr(), sh.data_ptr(), out.data_ptr(), part.data_ptr())
    mask = (Yprev * sc[:N] + sh[:N]) > 0
    wantm = want * mask
    err = (out.double() - wantm).abs()
    bad = (err.max(dim=1).values > 1e-3).nonzero().flatten()
    s1 = part[:, 0].double().sum(0); s2 = part[:, 1].double().sum(0)
    print("mask ", M, K, N, "maxerr %.3e" % err.max().item(), "bad rows:", bad[:10].tolist(), len(bad),
          "stats err %.3e %.3e" % ((s1 - wantm.sum(0)).abs().max().item() / wantm.sum(0).abs().max().item(),
                                   (s2 - (wantm * Yprev.double()).sum(0)).abs().max().item() / (wantm * Yprev.double()).sum(0).abs().max().item()))
