"""micro-benchmark of the one-pass backward (pcops_mlp_bwd_fused) against its Gram-form weight gradient
(pcops_mlp_bwd_fused_gw) at SA1's pooled top layer (4.19 M rows, 64 -> 128, groups of 32) and the T-Net's (10.5 M, groups of 20):
python tools/bench_bwd_fused.py [reps]   -- sustained loops of `reps` launches each, alternating"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scanobjectnn_amd import _lib
lib = _lib.load()
_lib.set_option(_lib.OPT_BWD_FUSED_GRAM_WGRAD, 1)     # (opt-in: DESIGN.md section 4.16)
dev = "cuda:0"
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30


def vec(n):
    return torch.randn(n, device=dev) * 0.1 + 1.0


def timeit(fn):
    global reps
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3


for (M, K, N, S) in [(4194304, 64, 128, 32), (10485760, 64, 128, 20)]:
    G = M // S
    Yp = torch.randn(M, K, device=dev)
    W = torch.randn(K, N, device=dev) / K ** 0.5; b = torch.randn(N, device=dev) * 0.1
    asc, ash, p, q, t = vec(K), vec(K) * 0.1, vec(N), vec(N) * 0.01, vec(N) * 0.01
    Y = torch.empty(M, N, device=dev)
    for r0 in range(0, M, 1 << 20):     # the layer's own forward: Y = relu(bn(Yprev)) W + b
        Y[r0:r0 + (1 << 20)] = torch.addmm(b, torch.relu(Yp[r0:r0 + (1 << 20)] * asc + ash), W)
    gp = torch.randn(G, N, device=dev); am = torch.randint(0, S, (G, N), device=dev, dtype=torch.int32).to(torch.uint8)
    groups = lib.pcops_mlp_bwd_fused_groups(M, K, N, S, 1)
    scratch = torch.empty(groups * (K * N + N + K * K + K), device=dev)
    dW, db, Gprev = torch.empty(K, N, device=dev), torch.empty(N, device=dev), torch.empty(M, K, device=dev)
    part = torch.empty(groups, 2, K, device=dev)
    f0 = lambda: _lib.call("pcops_mlp_bwd_fused", M, K, N, Yp.data_ptr(), asc.data_ptr(), ash.data_ptr(), None, Y.data_ptr(),
                           p.data_ptr(), q.data_ptr(), t.data_ptr(), gp.data_ptr(), am.data_ptr(), S, W.data_ptr(),
                           scratch.data_ptr(), dW.data_ptr(), db.data_ptr(), Gprev.data_ptr(), part.data_ptr())
    f1 = lambda: _lib.call("pcops_mlp_bwd_fused_gw", M, K, N, Yp.data_ptr(), asc.data_ptr(), ash.data_ptr(), Y.data_ptr(),
                           p.data_ptr(), q.data_ptr(), t.data_ptr(), gp.data_ptr(), am.data_ptr(), S, W.data_ptr(), b.data_ptr(),
                           scratch.data_ptr(), dW.data_ptr(), db.data_ptr(), Gprev.data_ptr(), part.data_ptr())
    f0(); d0 = dW.clone(); f1(); d1 = dW.clone()
    print("M=%d S=%d: max |dW_gw - dW| / max|dW| = %.3e" % (M, S, float((d1 - d0).abs().max() / d0.abs().max())))
    for rnd in range(3):
        print("   plain %8.1f us   gram form %8.1f us" % (timeit(f0), timeit(f1)), flush=True)
    # sustained (0.5 s each) with the shader clock and socket power the chip holds meanwhile (bench.py's hwmon poller)
    import bench
    keep = reps
    for label, fn in (("plain", f0), ("gram form", f1), ("plain", f0), ("gram form", f1)):
        reps = max(30, int(0.5e6 / max(timeit(fn), 1.0)))
        with bench.ClockPoller() as poller:
            us = timeit(fn)
        c = poller.summary() or {}
        print("   sustained %-9s %8.1f us   sclk %6.0f MHz   %6.0f W   (%d launches)" % (
            label, us, c.get("sclk_mhz", float("nan")), c.get("power_w", float("nan")), reps), flush=True)
    reps = keep
    del Yp, Y, Gprev
    torch.cuda.empty_cache()
