"""micro-benchmark of the MLP GEMM kernels at the BASELINE config-2 layer shapes (B=256):
python tools/bench_gemm.py [fwd|dgrad|wgrad|all] [reps]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scanobjectnn_amd import _lib
lib = _lib.load()
dev = "cuda:0"
which = sys.argv[1] if len(sys.argv) > 1 else "all"
POOLED = "--pooled" in sys.argv   # wgrad: dY rebuilt from (gpool, argmax, Y) with S=64
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
def vec(n): return (torch.randn((n + 3) // 4 * 4, device=dev) * 0.1 + 1.0)
if "--shape" in sys.argv:
    i = sys.argv.index("--shape")
    ONLY = [tuple(int(v) for v in sys.argv[i + 1:i + 4])]
else:
    ONLY = None
SHAPES = ONLY or [(4194304, 64, 64), (4194304, 64, 128), (2097152, 128, 128), (2097152, 128, 256), (32768, 256, 512), (32768, 512, 1024)]
def timeit(fn):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps
for (M, K, N) in SHAPES:
    X = torch.randn(M, K, device=dev); W = torch.randn(K, N, device=dev) / K ** 0.5; b = torch.randn(N, device=dev)
    sc, sh = vec(K), vec(K)
    Y = torch.empty(M, N, device=dev)
    P = lib.pcops_mlp_stats_rows(M); part = torch.empty(P, 2, N, device=dev)
    if which in ("fwd", "all"):
        ms = timeit(lambda: _lib.call("pcops_mlp_gemm_fwd", M, K, N, X.data_ptr(), K, sc.data_ptr(), sh.data_ptr(), W.data_ptr(), b.data_ptr(), Y.data_ptr(), part.data_ptr(), None))
        gb = (M * K + M * N) * 4 / 1e9; gf = 2.0 * M * K * N / 1e9
        print("fwd   M=%8d K=%4d N=%4d  %8.3f ms  %7.1f GB/s  %6.1f TF/s" % (M, K, N, ms, gb / ms * 1e3, gf / ms))
    if which in ("dgrad", "all"):
        # dgrad of the layer K->N: dY is (M,N), output (M,K)
        G = torch.randn(M, N, device=dev); Yl = torch.randn(M, N, device=dev); Yp = torch.randn(M, K, device=dev)
        p, q, t = vec(N), vec(N), vec(N); Wt = torch.randn(N, K, device=dev) / N ** 0.5
        out = torch.empty(M, K, device=dev); part2 = torch.empty(P, 2, K, device=dev)
        ms = timeit(lambda: _lib.call("pcops_mlp_gemm_dgrad", M, N, K, G.data_ptr(), Yl.data_ptr(), p.data_ptr(), q.data_ptr(), t.data_ptr(),
                                      None, None, 1, None, None, Wt.data_ptr(), Yp.data_ptr(), sc.data_ptr(), sh.data_ptr(), out.data_ptr(), part2.data_ptr()))
        gb = (2 * M * N + 2 * M * K) * 4 / 1e9; gf = 2.0 * M * K * N / 1e9
        print("dgrad M=%8d K=%4d N=%4d  %8.3f ms  %7.1f GB/s  %6.1f TF/s" % (M, N, K, ms, gb / ms * 1e3, gf / ms))
        del G, Yl, Yp, out
    if which in ("wgrad", "all"):
        G = torch.randn(M, N, device=dev); Yl = torch.randn(M, N, device=dev)
        p, q, t = vec(N), vec(N), vec(N)
        splits = lib.pcops_mlp_wgrad_splits(M, K, N); scratch = torch.empty(splits * (K * N + N), device=dev)
        dW = torch.empty(K, N, device=dev); db = torch.empty(N, device=dev)
        if POOLED:
            S = 64
            gp = torch.randn(M // S, N, device=dev); am = torch.randint(0, S, (M // S, N), device=dev, dtype=torch.int32).to(torch.uint8)
            ms = timeit(lambda: _lib.call("pcops_mlp_wgrad", M, K, N, X.data_ptr(), K, sc.data_ptr(), sh.data_ptr(), None, Yl.data_ptr(),
                                          p.data_ptr(), q.data_ptr(), t.data_ptr(), gp.data_ptr(), am.data_ptr(), S, p.data_ptr(), q.data_ptr(),
                                          scratch.data_ptr(), dW.data_ptr(), db.data_ptr()))
        else:
            ms = timeit(lambda: _lib.call("pcops_mlp_wgrad", M, K, N, X.data_ptr(), K, sc.data_ptr(), sh.data_ptr(), G.data_ptr(), Yl.data_ptr(),
                                      p.data_ptr(), q.data_ptr(), t.data_ptr(), None, None, 1, None, None, scratch.data_ptr(), dW.data_ptr(), db.data_ptr()))
        gb = (M * K + 2 * M * N) * 4 / 1e9; gf = 2.0 * M * K * N / 1e9
        print("wgrad M=%8d K=%4d N=%4d  %8.3f ms  %7.1f GB/s  %6.1f TF/s" % (M, K, N, ms, gb / ms * 1e3, gf / ms))
    del X, Y
    torch.cuda.empty_cache()
