"""Sampling ops -- mirror of `pointnet2/tf_ops/sampling/tf_sampling.py` on libpcops.

Same names, positional order (attrs first, like the Python wrappers of the reference:
tf_sampling.py:49-57), dtypes (f32 / i32) and differentiability: gather_point has a
gradient w.r.t. `inp` only (tf_sampling.py:44-48); farthest_point_sample is
non-differentiable (`ops.NoGradient`, :58).  `prob_sample` is out of scope (no model in
scope calls it).
"""
import torch

from .. import _lib


def _check_xyz(t, name):
    t = _lib.check(t, torch.float32, name, 3)
    if t.shape[2] != 3:
        # tf_sampling.cpp:105 "FarthestPointSample expects (batch_size,num_points,3) inp shape"
        raise ValueError("%s expects (batch_size,num_points,3) shape, got %s" % (name, tuple(t.shape)))
    return t


def farthest_point_sample(npoint, inp):
    """inp (B,N,3) f32 -> (B,npoint) i32; first index 0, reference tie rule (Appendix A3)."""
    npoint = int(npoint)
    if npoint <= 0:
        raise ValueError("FarthestPointSample expects positive npoint")  # tf_sampling.cpp:99
    inp = _check_xyz(inp.detach(), "inp")
    b, n, _ = inp.shape
    out = torch.empty((b, npoint), dtype=torch.int32, device=inp.device)
    # scratch the launcher asks for (0 bytes while the cloud fits the register-resident kernel, i.e. n <= 16384;
    # the reference always allocates its (32, n) `temp`, tf_sampling.cpp:115)
    nbytes = int(_lib.load().pcops_farthest_point_sample_workspace_bytes(b, n))
    temp = torch.empty(nbytes // 4, dtype=torch.float32, device=inp.device) if nbytes else None
    _lib.call("pcops_farthest_point_sample", b, n, npoint, _lib.ptr(inp), _lib.ptr(temp), _lib.ptr(out))
    return out


class _GatherPoint(torch.autograd.Function):
    @staticmethod
    def forward(ctx, inp, idx):
        b, n, _ = inp.shape
        m = idx.shape[1]
        out = torch.empty((b, m, 3), dtype=torch.float32, device=inp.device)
        _lib.call("pcops_gather_point", b, n, m, _lib.ptr(inp), _lib.ptr(idx), _lib.ptr(out))
        ctx.save_for_backward(idx)
        ctx.n = n
        return out

    @staticmethod
    def backward(ctx, out_g):
        (idx,) = ctx.saved_tensors
        out_g = out_g.contiguous()
        b, m, _ = out_g.shape
        if _lib.deterministic():     # ordered owner walk instead of float atomics
            return _lib.scatter_rows_sorted(idx, out_g, ctx.n), None
        inp_g = torch.empty((b, ctx.n, 3), dtype=torch.float32, device=out_g.device)
        _lib.call("pcops_gather_point_grad", b, ctx.n, m, _lib.ptr(out_g), _lib.ptr(idx), _lib.ptr(inp_g))
        return inp_g, None


def gather_point(inp, idx):
    """inp (B,N,3) f32, idx (B,M) i32 -> (B,M,3) f32"""
    inp = _check_xyz(inp, "inp")
    idx = _lib.check(idx, torch.int32, "idx", 2)
    if idx.shape[0] != inp.shape[0]:
        raise ValueError("GatherPoint expects (batch_size,num_result) idx shape")  # tf_sampling.cpp:135
    return _GatherPoint.apply(inp, idx)
