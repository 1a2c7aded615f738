"""Runs ON THE GPU BOX with PCOPS_LIB=.../libpcops_ecprof.so (tools/build_variant.sh ecprof "-DPCOPS_EC_PROF=1"): phase split
of ec_bwd_lds_kernel -- cycle-counter ticks of lane 0 per workgroup"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scanobjectnn_amd import _lib
from scanobjectnn_amd.dgcnn import tf_util
from scanobjectnn_amd.synth import synth_clouds
lib = _lib.load()
dev = "cuda:0"
b, n, k, C = 256, 2048, 20, 64
x = torch.from_numpy(synth_clouds(b, n, seed=1234)).to(dev)
idx = tf_util.knn_graph(x, k=k)
G = b * n
g = torch.Generator(device=dev).manual_seed(1)
Q, Ctr = torch.randn(b, n, C, device=dev, generator=g), torch.randn(b, n, C, device=dev, generator=g)
gpool, ysel, SQ = (torch.randn(G, C, device=dev, generator=g) for _ in range(3))
arg = torch.randint(0, k, (G, C), device=dev, generator=g, dtype=torch.int32).to(torch.uint8)
sc, sh, p, q, t = (torch.randn(C, device=dev, generator=g) for _ in range(5))
dQ, dCtr = torch.empty(b, n, C, device=dev), torch.empty(b, n, C, device=dev)
wsp = torch.empty(int(lib.pcops_sa_scatter_workspace_bytes(b, n, n, k)) // 4, dtype=torch.int32, device=dev)
P = lambda v: v.data_ptr()
out = (ctypes.c_ulonglong * 8)()
lib.pcops_ec_debug_prof.argtypes = [ctypes.c_void_p]
deg = torch.bincount(idx[0].flatten().long(), minlength=n)
print("in-degree of cloud 0: mean %.1f max %d, 64 longest %s" % (deg.float().mean().item(), deg.max().item(),
                                                                  sorted(deg.tolist())[-64::8]))
for rep in range(3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    _lib.call("pcops_edge_pool_bwd", b, n, n, k, C, P(Q), P(Ctr), P(idx), P(gpool), P(ysel), P(SQ), P(arg), P(sc), P(sh),
              P(p), P(q), P(t), P(dQ), P(dCtr), P(wsp))
    e1.record()
    lib.pcops_ec_debug_prof(out)
    nb = max(1, out[7])
    names = ["staging", "fill0", "walk0", "fill1", "walk1", "tail"]
    print("%.0f us  blocks %d  " % (e0.elapsed_time(e1) * 1e3, out[7]) +
          "  ".join("%s %.0f" % (nm, out[i] / nb) for i, nm in enumerate(names)) + "   (ticks per workgroup)", flush=True)
