"""Which part of the one-pass backward sets its time?  Needs the -DPCOPS_BF_DEBUG build of mlp.hip's parts 0 and 5
(scanobjectnn_amd/libpcops_bfdbg.so): PCOPS_BF_DEBUG = bit mask  1: no dX matrix instructions, 2: no epilogue (mask, column sums,
Gprev stores), 4: no dW matrix instructions, 8: producers do not stage, 16: producers do not load.  SA1's pooled layer, plain form
(default) or, with `xyz`, SA1's 64 -> 64 layer above the arithmetic first layer (dense upstream gradient).
    PCOPS_LIB=$PWD/scanobjectnn_amd/libpcops_bfdbg.so python tools/ablate_bwd_fused.py [xyz]"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
XYZ = "xyz" in sys.argv[1:]
if "child" in sys.argv[1:]:          # child: one setting
    sys.path.insert(0, ROOT)
    import torch
    from scanobjectnn_amd import _lib
    lib = _lib.load()
    dev = "cuda:0"
    M, K, N, S = 4194304, 64, 128, 32
    G = M // S
    g = torch.Generator().manual_seed(1)
    Yp = torch.randn(M, K, device=dev); Y = torch.randn(M, N, device=dev)
    W = torch.randn(K, N, device=dev) / 8
    v = lambda n: torch.randn(n, device=dev) * 0.1 + 1.0
    asc, ash, p, q, t = v(K), v(K) * 0.1, v(N), v(N) * 0.01, v(N) * 0.01
    gp = torch.randn(G, N, device=dev); am = torch.randint(0, S, (G, N), device=dev, dtype=torch.int32).to(torch.uint8)
    groups = lib.pcops_mlp_bwd_fused_groups(M, K, N, S, 1)
    scratch = torch.empty(groups * (K * N + N), device=dev)
    dW, db, Gprev = torch.empty(K, N, device=dev), torch.empty(N, device=dev), torch.empty(M, K, device=dev)
    part = torch.empty(groups, 2, K, device=dev)
    f = lambda: _lib.call("pcops_mlp_bwd_fused", M, K, N, Yp.data_ptr(), asc.data_ptr(), ash.data_ptr(), None, Y.data_ptr(),
                          p.data_ptr(), q.data_ptr(), t.data_ptr(), gp.data_ptr(), am.data_ptr(), S, W.data_ptr(),
                          scratch.data_ptr(), dW.data_ptr(), db.data_ptr(), Gprev.data_ptr(), part.data_ptr())
    if XYZ:
        N = 64
        off4 = torch.randn(M, 4, device=dev) * 0.1; xyzw = torch.randn(4, K, device=dev)
        Gd = torch.randn(M, N, device=dev); Y = torch.randn(M, N, device=dev); W = torch.randn(K, N, device=dev) / 8
        p, q, t = v(N), v(N) * 0.01, v(N) * 0.01
        groups = lib.pcops_mlp_bwd_fused_groups(M, K, N, 0, 0)
        scratch = torch.empty(groups * (K * N + N), device=dev)
        dW, db = torch.empty(K, N, device=dev), torch.empty(N, device=dev)
        xst = torch.empty(groups, 3, K, device=dev)
        f = lambda: _lib.call("pcops_mlp_bwd_fused_xyz_rows", M, K, N, off4.data_ptr(), xyzw.data_ptr(), asc.data_ptr(), ash.data_ptr(),
                              Gd.data_ptr(), Y.data_ptr(), p.data_ptr(), q.data_ptr(), t.data_ptr(), None, None, 1, W.data_ptr(),
                              scratch.data_ptr(), dW.data_ptr(), db.data_ptr(), part.data_ptr(), xst.data_ptr(), None)
    f(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(100):
        f()
    e.record(); torch.cuda.synchronize()
    print("PCOPS_BF_DEBUG=%-3s %8.1f us" % (os.environ.get("PCOPS_BF_DEBUG", "0"), s.elapsed_time(e) * 10))
else:
    for flags, what in ((0, "everything"), (4, "no dW matrix instructions"), (1, "no dX matrix instructions"), (5, "no matrix instructions"),
                        (2, "no epilogue / Gprev stores"), (7, "consumers idle"), (8, "producers load, do not stage"),
                        (24, "producers idle"), (31, "nothing but the barriers")):
        out = subprocess.run([sys.executable, os.path.abspath(__file__), "child"] + (["xyz"] if XYZ else []), env=dict(os.environ, PCOPS_BF_DEBUG=str(flags)),
                             capture_output=True, text=True).stdout.strip().splitlines()
        print("%-34s %s" % (what, out[-1] if out else "?"), flush=True)
