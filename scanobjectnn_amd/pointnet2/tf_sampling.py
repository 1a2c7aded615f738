"""Sampling ops -- mirror of `pointnet2/tf_ops/sampling/tf_sampling.py` on libpcops.

Same names, positional order (attrs first, like the Python wrappers of the reference:
tf_sampling.py:49-57), dtypes (f32 / i32) and differentiability: gather_point has a
gradient w.r.t. `inp` only (tf_sampling.py:44-48); farthest_point_sample is
non-differentiable (`ops.NoGradient`, :58), and so is prob_sample (:24).
"""
import torch

from .. import _lib


def _check_xyz(t, name):
    t = _lib.check(t, torch.float32, name, 3)
    if t.shape[2] != 3:
        # tf_sampling.cpp:105 "FarthestPointSample expects (batch_size,num_points,3) inp shape"
        raise ValueError("%s expects (batch_size,num_points,3) shape, got %s" % (name, tuple(t.shape)))
    return t


def prob_sample(inp, inpr, return_cumsum=False):
    """inp (B,ncategory) f32 weights, inpr (B,npoints) f32 numbers in [0,1] -> (B,npoints) i32: the category whose
    cumulative-weight interval holds inpr * total (tf_sampling.py:14-23; ProbSampleGpuOp tf_sampling.cpp:65-92).
    return_cumsum also hands back the op's scratch tensor, the row cumsum in the reference's association."""
    if not isinstance(inp, torch.Tensor) or inp.dim() != 2:
        raise ValueError("ProbSample expects (batch_size,num_choices) inp shape")      # tf_sampling.cpp:76
    inp = _lib.check(inp.detach(), torch.float32, "inp", 2)
    if not isinstance(inpr, torch.Tensor) or inpr.dim() != 2 or inpr.shape[0] != inp.shape[0]:
        raise ValueError("ProbSample expects (batch_size,num_points) inpr shape")      # tf_sampling.cpp:79
    inpr = _lib.check(inpr.detach(), torch.float32, "inpr", 2)
    b, n = inp.shape
    m = inpr.shape[1]
    if b > 0 and n < 1:
        raise ValueError("ProbSample expects (batch_size,num_choices) inp shape with num_choices >= 1")
    out = torch.empty((b, m), dtype=torch.int32, device=inp.device)
    temp = torch.empty((b, n), dtype=torch.float32, device=inp.device)   # the op's allocate_temp, tf_sampling.cpp:86
    _lib.call("pcops_prob_sample", b, n, m, _lib.ptr(inp), _lib.ptr(inpr), _lib.ptr(temp), _lib.ptr(out))
    return (out, temp) if return_cumsum else out


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
