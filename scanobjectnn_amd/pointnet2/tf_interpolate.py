"""3-NN interpolation ops -- mirror of `pointnet2/tf_ops/3d_interpolation/tf_interpolate.py`
on libpcops (device kernels; the reference only has CPU kernels for these).

three_nn is non-differentiable (`ops.NoGradient`, tf_interpolate.py:19); three_interpolate
has a gradient w.r.t. `points` only (:30-35).
"""
import os

import torch

from .. import _lib


def three_nn(xyz1, xyz2):
    """xyz1 (B,n,3) unknown, xyz2 (B,m,3) known -> dist (B,n,3) f32 SQUARED distances
    ascending, idx (B,n,3) i32.  m<3: missing neighbours are (+inf, 0)."""
    xyz1 = _lib.check(xyz1.detach(), torch.float32, "xyz1", 3)
    xyz2 = _lib.check(xyz2.detach(), torch.float32, "xyz2", 3)
    if xyz1.shape[2] != 3:  # tf_interpolate.cpp:163
        raise ValueError("ThreeNN expects (b,n,3) xyz1 shape")
    if xyz2.shape[2] != 3 or xyz2.shape[0] != xyz1.shape[0]:  # tf_interpolate.cpp:168
        raise ValueError("ThreeNN expects (b,m,3) xyz2 shape")
    b, n, _ = xyz1.shape
    m = xyz2.shape[1]
    dist = torch.empty((b, n, 3), dtype=torch.float32, device=xyz1.device)
    idx = torch.empty((b, n, 3), dtype=torch.int32, device=xyz1.device)
    _lib.call("pcops_three_nn", b, n, m, _lib.ptr(xyz1), _lib.ptr(xyz2), _lib.ptr(dist), _lib.ptr(idx))
    return dist, idx


SORTED_GRAD = os.environ.get("PCOPS_INTERP_SORTED", "1") != "0"

class _ThreeInterpolate(torch.autograd.Function):
    @staticmethod
    def forward(ctx, points, idx, weight):
        b, m, c = points.shape
        n = idx.shape[1]
        out = torch.empty((b, n, c), dtype=torch.float32, device=points.device)
        _lib.call("pcops_three_interpolate", b, m, c, n, _lib.ptr(points), _lib.ptr(idx),
                  _lib.ptr(weight), _lib.ptr(out))
        ctx.save_for_backward(idx, weight)
        ctx.shape = (b, m, c)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        idx, weight = ctx.saved_tensors
        b, m, c = ctx.shape
        n = idx.shape[1]
        grad_out = grad_out.contiguous()
        # ordered owner walk instead of float atomics: always in deterministic mode, and by default where it is also the
        # faster form -- enough destination points to spread the lists (measured at the BGA config: 2048 -> 512 points,
        # 128 channels: 244 us against 321 us with atomics; a single destination point would serialise one list)
        if _lib.deterministic() or (SORTED_GRAD and m >= 64 and _lib.scatter_rows_sorted_supported(3 * n, m)):
            return _lib.scatter_rows_sorted(idx.view(b, 3 * n), grad_out, m, div=3, w=weight), None, None
        grad_points = torch.empty((b, m, c), dtype=torch.float32, device=grad_out.device)
        _lib.call("pcops_three_interpolate_grad", b, n, c, m, _lib.ptr(grad_out), _lib.ptr(idx),
                  _lib.ptr(weight), _lib.ptr(grad_points))
        return grad_points, None, None


def three_interpolate(points, idx, weight):
    """points (B,m,C) known features, idx/weight (B,n,3) -> (B,n,C)"""
    points = _lib.check(points, torch.float32, "points", 3)
    idx = _lib.check(idx, torch.int32, "idx", 3)
    weight = _lib.check(weight.detach(), torch.float32, "weight", 3)
    if idx.shape[0] != points.shape[0] or idx.shape[2] != 3:  # tf_interpolate.cpp:203
        raise ValueError("ThreeInterpolate expects (b,n,3) idx shape")
    if weight.shape != idx.shape:  # tf_interpolate.cpp:206
        raise ValueError("ThreeInterpolate expects (b,n,3) weight shape")
    return _ThreeInterpolate.apply(points, idx, weight)
