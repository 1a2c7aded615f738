"""Grouping ops -- mirror of `pointnet2/tf_ops/grouping/tf_grouping.py` on libpcops.

query_ball_point / select_top_k / knn_point are non-differentiable (`ops.NoGradient`,
tf_grouping.py:22,33); group_point has a gradient w.r.t. `points` only (:43-47).
"""
import ctypes as C

import torch

from .. import _lib


def _check_pair(xyz1, xyz2):
    xyz1 = _lib.check(xyz1.detach(), torch.float32, "xyz1", 3)
    xyz2 = _lib.check(xyz2.detach(), torch.float32, "xyz2", 3)
    if xyz1.shape[2] != 3:  # tf_grouping.cpp:79
        raise ValueError("QueryBallPoint expects (batch_size, ndataset, 3) xyz1 shape.")
    if xyz2.shape[2] != 3 or xyz2.shape[0] != xyz1.shape[0]:  # tf_grouping.cpp:84
        raise ValueError("QueryBallPoint expects (batch_size, npoint, 3) xyz2 shape.")
    return xyz1, xyz2


def query_ball_point(radius, nsample, xyz1, xyz2):
    """xyz1 (B,N,3) dataset, xyz2 (B,M,3) queries -> idx (B,M,nsample) i32, pts_cnt (B,M) i32.
    First `nsample` in-ball dataset indices in ascending order, padded with the first hit."""
    radius, nsample = float(radius), int(nsample)
    if not radius > 0:
        raise ValueError("QueryBallPoint expects positive radius")  # tf_grouping.cpp:71
    if nsample <= 0:
        raise ValueError("QueryBallPoint expects positive nsample")  # tf_grouping.cpp:74
    xyz1, xyz2 = _check_pair(xyz1, xyz2)
    b, n, _ = xyz1.shape
    m = xyz2.shape[1]
    idx = torch.empty((b, m, nsample), dtype=torch.int32, device=xyz1.device)
    cnt = torch.empty((b, m), dtype=torch.int32, device=xyz1.device)
    _lib.call("pcops_query_ball_point", b, n, m, radius, nsample, _lib.ptr(xyz1), _lib.ptr(xyz2),
              _lib.ptr(idx), _lib.ptr(cnt))
    return idx, cnt


def query_ball_point_multi(radius_list, nsample_list, xyz1, xyz2):
    """All radii of an MSG layer in one pass over the dataset (fast path beside -- not instead
    of -- query_ball_point).  Returns [(idx, pts_cnt), ...] identical to per-radius calls."""
    ns = len(radius_list)
    if ns != len(nsample_list) or not 1 <= ns <= 4:
        raise ValueError("query_ball_point_multi takes 1..4 (radius, nsample) pairs")
    for r, s in zip(radius_list, nsample_list):
        if not float(r) > 0 or int(s) <= 0:
            raise ValueError("QueryBallPoint expects positive radius and nsample")
    xyz1, xyz2 = _check_pair(xyz1, xyz2)
    b, n, _ = xyz1.shape
    m = xyz2.shape[1]
    outs = [(torch.empty((b, m, int(s)), dtype=torch.int32, device=xyz1.device),
             torch.empty((b, m), dtype=torch.int32, device=xyz1.device)) for s in nsample_list]
    radii = (C.c_float * ns)(*[float(r) for r in radius_list])
    nsamp = (C.c_int * ns)(*[int(s) for s in nsample_list])
    idxp = (C.c_void_p * ns)(*[o[0].data_ptr() for o in outs])
    cntp = (C.c_void_p * ns)(*[o[1].data_ptr() for o in outs])
    _lib.call("pcops_query_ball_point_multi", b, n, m, ns, C.cast(radii, C.c_void_p),
              C.cast(nsamp, C.c_void_p), _lib.ptr(xyz1), _lib.ptr(xyz2),
              C.cast(idxp, C.c_void_p), C.cast(cntp, C.c_void_p))
    return outs


def select_top_k(k, dist):
    """dist (B,M,N) -> outi (B,M,N) i32, out (B,M,N) f32; first k columns = the k smallest
    (literal unstable selection sort of the reference, tf_grouping_g.cu:83-123)."""
    k = int(k)
    if k <= 0:
        raise ValueError("SelectionSort expects positive k")  # tf_grouping.cpp:113
    dist = _lib.check(dist.detach(), torch.float32, "dist", 3)
    b, m, n = dist.shape
    outi = torch.empty((b, m, n), dtype=torch.int32, device=dist.device)
    out = torch.empty((b, m, n), dtype=torch.float32, device=dist.device)
    _lib.call("pcops_selection_sort", b, n, m, k, _lib.ptr(dist), _lib.ptr(outi), _lib.ptr(out))
    return outi, out


class _GroupPoint(torch.autograd.Function):
    @staticmethod
    def forward(ctx, points, idx):
        b, n, c = points.shape
        _, m, s = idx.shape
        out = torch.empty((b, m, s, c), dtype=torch.float32, device=points.device)
        _lib.call("pcops_group_point", b, n, c, m, s, _lib.ptr(points), _lib.ptr(idx), _lib.ptr(out))
        ctx.save_for_backward(idx)
        ctx.shape = (b, n, c)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        (idx,) = ctx.saved_tensors
        b, n, c = ctx.shape
        _, m, s = idx.shape
        grad_out = grad_out.contiguous()
        if _lib.deterministic():     # ordered owner walk instead of float atomics
            return _lib.scatter_rows_sorted(idx.view(b, m * s), grad_out.view(b, m * s, c), n), None
        grad_points = torch.empty((b, n, c), dtype=torch.float32, device=grad_out.device)
        _lib.call("pcops_group_point_grad", b, n, c, m, s, _lib.ptr(grad_out), _lib.ptr(idx),
                  _lib.ptr(grad_points))
        return grad_points, None


def group_point(points, idx):
    """points (B,N,C) f32, idx (B,M,S) i32 -> (B,M,S,C) f32"""
    points = _lib.check(points, torch.float32, "points", 3)
    idx = _lib.check(idx, torch.int32, "idx", 3)
    if idx.shape[0] != points.shape[0]:  # tf_grouping.cpp:155
        raise ValueError("GroupPoint expects (batch_size, npoints, nsample) idx shape")
    return _GroupPoint.apply(points, idx)


def knn_point(k, xyz1, xyz2):
    """xyz1 (B,N,C) dataset, xyz2 (B,M,C) queries -> val (B,M,k) f32, idx (B,M,k) i32
    (tf_grouping.py:49-74: squared distances, selection sort, first k columns)."""
    xyz1 = _lib.check(xyz1.detach(), torch.float32, "xyz1", 3)
    xyz2 = _lib.check(xyz2.detach(), torch.float32, "xyz2", 3)
    k = int(k)
    if k <= 0:
        raise ValueError("SelectionSort expects positive k")  # tf_grouping.cpp:113
    if xyz2.shape[0] != xyz1.shape[0] or xyz2.shape[2] != xyz1.shape[2]:
        raise ValueError("knn_point expects (b,n,c) xyz1 and (b,m,c) xyz2")
    b, n, c = xyz1.shape
    m = xyz2.shape[1]
    if k > n:
        raise ValueError("knn_point: k = %d neighbours of %d points" % (k, n))
    # dist[b,j,t] = sum_c (xyz1[b,t,c] - xyz2[b,j,c])^2, c ascending, uncontracted -- in the kernel
    if _lib.load().pcops_knn_point_supported(n):
        val = torch.empty((b, m, k), dtype=torch.float32, device=xyz1.device)
        idx = torch.empty((b, m, k), dtype=torch.int32, device=xyz1.device)
        _lib.call("pcops_knn_point", b, n, c, m, k, _lib.ptr(xyz1), _lib.ptr(xyz2), _lib.ptr(val), _lib.ptr(idx))
        return val, idx
    dist = torch.empty((b, m, n), dtype=torch.float32, device=xyz1.device)     # clouds beyond the fused kernel's LDS row
    _lib.call("pcops_knn_point_dist", b, n, c, m, _lib.ptr(xyz1), _lib.ptr(xyz2), _lib.ptr(dist))
    outi, out = select_top_k(k, dist)
    return out[:, :, :k].contiguous(), outi[:, :, :k].contiguous()
