"""Does the memory-side cache (256 MB) serve a re-read of what a kernel just wrote?  The T-Net's second layer writes a
(10.5 M, 128) fp32 activation (5.4 GB) that bn_relu_maxpool reads back at once.  Timed here: GEMM + max-pool over the whole
tensor vs the same two kernels alternating over chunks of whole clouds (chunk activation 21-336 MB).  Runs on the GPU box."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scanobjectnn_amd import _lib

dev = "cuda:0"
lib = _lib.load()
B, N, S, K, C = 256, 2048, 20, 64, 128
R = B * N * S
g = torch.Generator(device=dev).manual_seed(0)
X = torch.randn(R, K, device=dev, generator=g)
W = torch.randn(K, C, device=dev, generator=g) / 8
bias = torch.zeros(C, device=dev)
sc, sh = torch.ones(K, device=dev), torch.zeros(K, device=dev)
sc2, sh2 = torch.ones(C, device=dev), torch.zeros(C, device=dev)
Y = torch.empty(R, C, device=dev)
out = torch.empty(B * N, C, device=dev)
arg = torch.empty(B * N, C, dtype=torch.uint8, device=dev)
ysel = torch.empty(B * N, C, device=dev)
part = torch.empty(lib.pcops_mlp_stats_rows(R) + 4096, 2, C, device=dev)


def run(chunks):
    rc = R // chunks
    gc = B * N // chunks
    for i in range(chunks):
        _lib.call("pcops_mlp_gemm_fwd", rc, K, C, X.data_ptr() + i * rc * K * 4, K, sc.data_ptr(), sh.data_ptr(), W.data_ptr(),
                  bias.data_ptr(), Y.data_ptr() + i * rc * C * 4, part.data_ptr(), None)
        _lib.call("pcops_mlp_bn_relu_maxpool", gc, S, C, Y.data_ptr() + i * rc * C * 4, sc2.data_ptr(), sh2.data_ptr(),
                  out.data_ptr() + i * gc * C * 4, arg.data_ptr() + i * gc * C, ysel.data_ptr() + i * gc * C * 4)


for chunks in (1, 2, 4, 8, 16, 32, 64, 128):
    run(chunks)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        run(chunks)
    e1.record()
    torch.cuda.synchronize()
    print("chunks %3d (activation chunk %6.0f MB): %.3f ms for GEMM + max-pool over the whole tensor" % (
        chunks, R // chunks * C * 4 / 1e6, e0.elapsed_time(e1) / 5), flush=True)
