"""Where does a wave of gemm_ws_kernel spend its cycles?  Needs a -DPCOPS_PHASE_PROF build of the library:
    hipcc ... -DPCOPS_PHASE_PROF -c mlp.hip -o /tmp/mlp_prof.o ; link as scanobjectnn_amd/libpcops_prof.so
    PCOPS_LIB=scanobjectnn_amd/libpcops_prof.so python tools/phase_prof.py
Per launch: shader cycles per 32-row tile and wave in the three phases (operand staging, MFMA loop, epilogue), next to
the cycles the tile's MFMAs occupy the matrix pipe (64 per v_mfma_f32_32x32x2_f32) -- with two waves per SIMD a wave's
MFMA phase lasting ~2x its pipe time means the partner was in ITS MFMA phase at the same time."""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scanobjectnn_amd import _lib
lib = _lib.load()
lib.pcops_debug_phase_prof.argtypes = [C.POINTER(C.c_ulonglong)]
dev = "cuda:0"


def prof(label, fn, mfma_per_tile, reps=3):
    buf = (C.c_ulonglong * 8)()
    fn()
    lib.pcops_debug_phase_prof(buf)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    lib.pcops_debug_phase_prof(buf)
    us = s.elapsed_time(e) / reps * 1e3
    st, mf, ep, tot, tiles, waves = (buf[i] / reps for i in range(6))
    print("%-34s %7.0f us | per tile: stage %6.0f  mfma %6.0f (pipe %5d)  epilogue %6.0f | wave total %9.0f cyc = %5.2f GHz-equiv, tiles/wave %.1f, accounted %.2f"
          % (label, us, st / tiles, mf / tiles, mfma_per_tile * 64, ep / tiles, tot / waves, tot / waves / us / 1e3, tiles / waves,
             (st + mf + ep) / tot), flush=True)


def vec(n):
    return torch.randn((n + 3) // 4 * 4, device=dev) * 0.1 + 1.0


for (M, K, N) in [(2097152, 128, 256), (2097152, 128, 128), (4194304, 64, 128), (4194304, 64, 64)]:
    X = torch.randn(M, K, device=dev)
    W = torch.randn(K, N, device=dev) / K ** 0.5
    b = torch.randn(N, device=dev)
    sc, sh = vec(K), vec(K)
    Y = torch.empty(M, N, device=dev)
    P = lib.pcops_mlp_stats_rows(M)
    part = torch.empty(P, 2, N, device=dev)
    bn = 128 if N > 64 else 64
    prof("fwd %d %d->%d" % (M, K, N),
         lambda: _lib.call("pcops_mlp_gemm_fwd", M, K, N, X.data_ptr(), K, sc.data_ptr(), sh.data_ptr(), W.data_ptr(), b.data_ptr(),
                           Y.data_ptr(), part.data_ptr(), None), (K // 2) * (bn // 32))
    gam = vec(N)
    ys = torch.empty(M // 32, N, device=dev)
    ps = torch.empty(M // 32, N, dtype=torch.uint8, device=dev)
    if lib.pcops_mlp_gemm_fwd_pool_supported(M, K, N, 32):
        prof("fwd+pool(32) %d %d->%d" % (M, K, N),
             lambda: _lib.call("pcops_mlp_gemm_fwd_pool", M, K, N, 32, X.data_ptr(), K, sc.data_ptr(), sh.data_ptr(), W.data_ptr(),
                               b.data_ptr(), gam.data_ptr(), Y.data_ptr(), part.data_ptr(), None, ys.data_ptr(), ps.data_ptr()),
             (K // 2) * (bn // 32))
    G = torch.randn(M, N, device=dev)
    Yl = torch.randn(M, N, device=dev)
    Yp = torch.randn(M, K, device=dev)
    p, q, t = vec(N), vec(N), vec(N)
    Wt = torch.randn(N, K, device=dev) / N ** 0.5
    out = torch.empty(M, K, device=dev)
    part2 = torch.empty(P, 2, K, device=dev)
    bk = 128 if K > 64 else 64
    prof("dgrad %d %d->%d" % (M, N, K),
         lambda: _lib.call("pcops_mlp_gemm_dgrad", M, N, K, G.data_ptr(), Yl.data_ptr(), p.data_ptr(), q.data_ptr(), t.data_ptr(),
                           None, None, 1, None, None, Wt.data_ptr(), Yp.data_ptr(), sc.data_ptr(), sh.data_ptr(), out.data_ptr(),
                           part2.data_ptr()), (N // 2) * (bk // 32))
    del X, Y, G, Yl, Yp, out
    torch.cuda.empty_cache()
