"""accuracy of pcops_mlp_gemm_fwd against float64 on random inputs (relative RMS, worst rows, column sums):
   python tools/check_gemm_accuracy.py      (PCOPS_GEMM_BF3=0 for the fp32 pipe)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scanobjectnn_amd import _lib
lib = _lib.load(); dev = "cuda:0"
torch.manual_seed(0)
for (M, K, N) in [(131072, 64, 64), (131072, 64, 128), (65536, 128, 128), (65536, 128, 256), (131072 + 77, 64, 128), (262144, 64, 128)]:
    X = torch.randn(M, K, device=dev); W = torch.randn(K, N, device=dev) / K ** 0.5; b = torch.randn(N, device=dev)
    sc = torch.rand(K, device=dev) + 0.5; sh = torch.randn(K, device=dev) * 0.3
    Y = torch.empty(M, N, device=dev)
    P = lib.pcops_mlp_stats_rows(M); part = torch.zeros(P, 2, N, device=dev)
    _lib.call("pcops_mlp_gemm_fwd", M, K, N, X.data_ptr(), K, sc.data_ptr(), sh.data_ptr(), W.data_ptr(), b.data_ptr(), Y.data_ptr(), part.data_ptr(), None)
    A = torch.relu(X.double() * sc.double() + sh.double())
    ref = A @ W.double() + b.double()
    err = (Y.double() - ref).abs()
    rowmax = err.max(dim=1).values
    rms = err.pow(2).mean().sqrt().item() / ref.pow(2).mean().sqrt().item()
    s1 = part.double().sum(0)
    print("M=%d K=%d N=%d  rel rms %.2e  max %.2e  worst rows %s   stats err sum %.2e sumsq %.2e" % (
        M, K, N, rms, err.max().item(), rowmax.topk(3).indices.tolist(),
        ((s1[0] - ref.sum(0)).abs().max() / ref.sum(0).abs().max()).item(),
        ((s1[1] - (ref * ref).sum(0)).abs().max() / (ref * ref).sum(0).abs().max()).item()))
