"""PointNet++ layers -- mirror of `pointnet2/utils/pointnet_util.py` on the libpcops ops.

Same function names, positional order, keyword defaults and return tuples as the reference:
sample_and_group (:22-56), sample_and_group_all (:59-84), pointnet_sa_module (:87-154),
pointnet_sa_module_msg (:156-196), pointnet_fp_module (:199-229).  Channel order of every concat
follows SURVEY.md Appendix A9.
"""
import torch

from . import tf_util
from .. import _lib, fused_mlp
from ..graph import variable_scope
from .tf_grouping import group_point, knn_point, query_ball_point, query_ball_point_multi
from .tf_interpolate import three_interpolate, three_nn
from .tf_sampling import farthest_point_sample, gather_point


def sample_and_group(npoint, radius, nsample, xyz, points, knn=False, use_xyz=True):
    """xyz (B,N,3), points (B,N,C)|None ->
    new_xyz (B,npoint,3), new_points (B,npoint,nsample,3+C), idx (B,npoint,nsample),
    grouped_xyz (B,npoint,nsample,3) (centred on the sampled point)."""
    new_xyz = gather_point(xyz, farthest_point_sample(npoint, xyz))
    if knn:
        _, idx = knn_point(nsample, xyz, new_xyz)
    else:
        idx, _pts_cnt = query_ball_point(radius, nsample, xyz, new_xyz)
    grouped_xyz = group_point(xyz, idx) - new_xyz.unsqueeze(2)  # translation normalisation (:46)
    if points is None:
        new_points = grouped_xyz
    else:
        grouped_points = group_point(points, idx)
        new_points = torch.cat([grouped_xyz, grouped_points], dim=-1) if use_xyz else grouped_points
    return new_xyz, new_points, idx, grouped_xyz


def sample_and_group_all(xyz, points, use_xyz=True):
    """One group holding the whole cloud, centroid (0,0,0) (:59-84)."""
    b, n, _ = xyz.shape
    new_xyz = torch.zeros((b, 1, 3), dtype=torch.float32, device=xyz.device)
    idx = torch.arange(n, dtype=torch.int32, device=xyz.device).view(1, 1, n).expand(b, 1, n)
    grouped_xyz = xyz.view(b, 1, n, 3)
    if points is None:
        new_points = grouped_xyz
    else:
        new_points = (torch.cat([xyz, points], dim=2) if use_xyz else points).unsqueeze(1)
    return new_xyz, new_points, idx, grouped_xyz


def _mlp_stack(x, widths, scope_fmt, bn, is_training, bn_decay, data_format, pool_max=False):
    """the reference's `for i, num_out_channel in enumerate(mlp): conv2d(...)` loops (+ the max over the
    neighbourhood axis when pool_max), fused on the device when the stack is the standard BN+ReLU one"""
    if tf_util.FUSED_MLP and fused_mlp.fused_supported(x, widths, bn, True):
        if pool_max and x.shape[2] > 256:   # arg-max is kept in 8 bits: pool outside the kernel
            return tf_util.conv2d_stack(x, widths, scope_fmt, is_training, bn_decay).amax(dim=2, keepdim=True)
        return tf_util.conv2d_stack(x, widths, scope_fmt, is_training, bn_decay, pool_max=pool_max)
    x = _mlp_stack_unfused(x, widths, scope_fmt, bn, is_training, bn_decay, data_format)
    return x.amax(dim=2, keepdim=True) if pool_max else x


def _mlp_stack_unfused(x, widths, scope_fmt, bn, is_training, bn_decay, data_format):
    for i, width in enumerate(widths):
        x = tf_util.conv2d(x, width, [1, 1], padding='VALID', stride=[1, 1], bn=bn,
                           is_training=is_training, scope=scope_fmt % i, bn_decay=bn_decay,
                           data_format=data_format)
    return x


def _grouped_mlp_fused(xyz, points, new_xyz, idx, widths, scope_fmt, is_training, bn_decay, use_xyz,
                       xyz_first, pool_max, identity_idx=False, pts_cnt=None):
    """Grouped shared MLP without ever building the grouped input.  The first 1x1 conv is linear, so
         concat(xyz[idx] - new_xyz, points[idx]) W + b  =  (points W_f + b)[idx] + (xyz[idx] - new_xyz) W_xyz :
    the feature part runs once per SOURCE point (B*N rows through a library GEMM instead of B*M*S) and the
    three coordinate channels are evaluated inside the gather kernel on the centred offsets (csrc/gather.hip).
    `xyz_first`: channel order of the reference's concat -- [xyz | feats] in sample_and_group (:50),
    [feats | xyz] in the MSG module (:184).  Returns (B,M,1,C) if pool_max else (B,M,S,C)."""
    b, m, s = idx.shape
    n = xyz.shape[1]
    cin = (points.shape[-1] if points is not None else 0) + (3 if (use_xyz or points is None) else 0)
    layers = tf_util._stack_variables(cin, widths, scope_fmt, 1e-3, None, True, ('moving_mean', 'moving_variance'))
    w1, b1 = layers[0][0], layers[0][1]
    c1 = w1.shape[1]
    kw = {}
    if points is None:
        kw = dict(xyz=xyz, new_xyz=new_xyz, wxyz=w1, bias=b1)
    else:
        cf = points.shape[-1]
        pts2d = points.reshape(b * n, cf)
        lin = fused_mlp.rows_linear if cf % 4 == 0 else (lambda x, w, bias: torch.addmm(bias, x, w))
        if use_xyz:
            if xyz_first:
                w_xyz, w_f = fused_mlp.split_rows(w1, 3)
            else:
                w_f, w_xyz = fused_mlp.split_rows(w1, cf)
            kw = dict(Q=lin(pts2d, w_f, b1).view(b, n, c1), xyz=xyz, new_xyz=new_xyz, wxyz=w_xyz)
        else:
            kw = dict(Q=lin(pts2d, w1, b1).view(b, n, c1))
    decay = bn_decay if bn_decay is not None else 0.9
    # pts_cnt (the ball query's hit counts): rows beyond it are copies of the group's first member -- the stack may
    # leave them out and weigh that member instead (fused_mlp "compacted rows"); None: every row is computed
    out = fused_mlp.gather_mlp_stack(idx, pool_max, is_training, decay, tf_util.BN_EPS, True, layers,
                                     identity_idx=identity_idx, pts_cnt=pts_cnt, **kw)
    return out.view(b, m, 1 if pool_max else s, widths[-1])


_WHOLE_CLOUD = {}


def _whole_cloud_group(b, n, device):
    """(origin (b, 1, 3), idx (b, 1, n) = 0 .. n-1) of sample_and_group_all (:59-84) -- constants of the shape: built once per
    (b, n, device) instead of three launches per step (nothing downstream writes into them)"""
    key = (int(b), int(n), str(device))
    if not fused_mlp.TAIL_FOLD or key not in _WHOLE_CLOUD:
        new_xyz = torch.zeros((b, 1, 3), dtype=torch.float32, device=device)
        idx = torch.arange(n, dtype=torch.int32, device=device).view(1, 1, n).expand(b, 1, n).contiguous()
        if not fused_mlp.TAIL_FOLD:
            return new_xyz, idx
        if len(_WHOLE_CLOUD) > 8:
            _WHOLE_CLOUD.clear()
        _WHOLE_CLOUD[key] = (new_xyz, idx)
    return _WHOLE_CLOUD[key]


def _gather_fusable(points, widths, bn, nsample, pool_max, xyz=None):
    c1 = widths[0]
    if xyz is not None and torch.is_grad_enabled() and xyz.requires_grad:
        # coordinates carry gradient (a T-Net in front, input saliency, adversarial perturbation): only the unfused
        # path differentiates w.r.t. xyz -- group_point / gather_point / the centring subtraction, as in the
        # reference (tf_grouping.py:43-47, tf_sampling.py:44-48)
        return False
    return (tf_util.FUSED_MLP and bn and all(w % 32 == 0 for w in widths) and 256 % (c1 // 4) == 0
            and (c1 >= 256 or 256 % c1 == 0) and c1 <= 1024 and (not pool_max or nsample <= 256))


def pointnet_sa_module(xyz, points, npoint, radius, nsample, mlp, mlp2, group_all, is_training,
                       bn_decay, scope, bn=True, pooling='max', knn=False, use_xyz=True,
                       use_nchw=False):
    """Set abstraction.  Returns new_xyz (B,npoint,3), new_points (B,npoint,mlp[-1] or mlp2[-1]),
    idx (B,npoint,nsample)."""
    # NCHW is a TF conv-layout hint; here every 1x1 conv is the same channel-last contraction, so
    # the flag changes nothing numerically (the reference transposes in and out, :116-123).
    with variable_scope(scope):
        pool_max = pooling == 'max'
        if group_all:
            nsample = xyz.shape[1]
        if (pooling in ('max', 'avg', 'max_and_avg') and xyz.is_cuda and (points is not None or not group_all)
                and _gather_fusable(points, mlp, bn, nsample, pool_max, xyz)):
            # fast path: sample -> query -> [first conv before grouping] -> gather+add -> fused stack
            if group_all:
                # one group holding the whole cloud around the origin (:59-84): idx = 0..n-1, so the "gather" is
                # the identity and the K = 3 + C first layer becomes an aligned K = C contraction + 3 inline terms
                b, n, _ = xyz.shape
                new_xyz, idx = _whole_cloud_group(b, n, xyz.device)
            else:
                new_xyz = gather_point(xyz, farthest_point_sample(npoint, xyz))
                _pts_cnt = None
                if knn:
                    _, idx = knn_point(nsample, xyz, new_xyz)
                else:
                    idx, _pts_cnt = query_ball_point(radius, nsample, xyz, new_xyz)
            new_points = _grouped_mlp_fused(xyz, points, new_xyz, idx, mlp, 'conv%d', is_training, bn_decay,
                                            use_xyz, True, pool_max, identity_idx=group_all,
                                            pts_cnt=None if group_all else _pts_cnt)
            grouped_xyz = None
        else:
            if group_all:
                new_xyz, new_points, idx, grouped_xyz = sample_and_group_all(xyz, points, use_xyz)
            else:
                new_xyz, new_points, idx, grouped_xyz = sample_and_group(npoint, radius, nsample, xyz,
                                                                         points, knn, use_xyz)
            new_points = _mlp_stack(new_points, mlp, 'conv%d', bn, is_training, bn_decay, 'NHWC',
                                    pool_max=pool_max)

        if pooling == 'max':
            pass  # fused into the stack above
        elif pooling == 'avg':
            new_points = new_points.mean(dim=2, keepdim=True)
        elif pooling == 'weighted_avg':
            dists = grouped_xyz.norm(dim=-1, p=2, keepdim=True)
            exp_dists = torch.exp(-dists * 5)
            weights = exp_dists / exp_dists.sum(dim=2, keepdim=True)
            new_points = (new_points * weights).sum(dim=2, keepdim=True)
        elif pooling == 'max_and_avg':
            new_points = torch.cat([new_points.mean(dim=2, keepdim=True),
                                    new_points.amax(dim=2, keepdim=True)], dim=-1)
        else:
            raise ValueError("unknown pooling %r" % (pooling,))

        if mlp2 is not None:
            new_points = _mlp_stack(new_points, mlp2, 'conv_post_%d', bn, is_training, bn_decay, 'NHWC')
        return new_xyz, new_points.squeeze(2), idx


def pointnet_sa_module_msg(xyz, points, npoint, radius_list, nsample_list, mlp_list, is_training,
                           bn_decay, scope, bn=True, use_xyz=True, use_nchw=False):
    """Multi-scale grouping: one FPS, then per radius ball query -> group -> [feats | xyz] -> MLP ->
    max; scales concatenated.  Returns new_xyz (B,npoint,3), new_points (B,npoint,sum mlp[k][-1])."""
    with variable_scope(scope):
        new_xyz = gather_point(xyz, farthest_point_sample(npoint, xyz))
        scales = query_ball_point_multi(radius_list, nsample_list, xyz, new_xyz)  # one dataset pass
        outs = []
        for i, (idx, _cnt) in enumerate(scales):
            fmt = 'conv%d_' % i + '%d'
            if xyz.is_cuda and _gather_fusable(points, mlp_list[i], bn, idx.shape[2], True, xyz):
                grouped = _grouped_mlp_fused(xyz, points, new_xyz, idx, mlp_list[i], fmt, is_training, bn_decay,
                                             use_xyz, False, True, pts_cnt=_cnt)   # [feats | xyz] order (:184)
            else:
                grouped_xyz = group_point(xyz, idx) - new_xyz.unsqueeze(2)
                if points is None:
                    grouped = grouped_xyz
                else:
                    grouped = group_point(points, idx)
                    if use_xyz:
                        grouped = torch.cat([grouped, grouped_xyz], dim=-1)  # note: feats first (:184)
                grouped = _mlp_stack(grouped, mlp_list[i], fmt, bn, is_training, bn_decay, 'NHWC', pool_max=True)
            outs.append(grouped.squeeze(2))
        return new_xyz, torch.cat(outs, dim=-1)


def pointnet_fp_module(xyz1, xyz2, points1, points2, mlp, is_training, bn_decay, scope, bn=True):
    """Feature propagation: xyz1 (B,n1,3) dense, xyz2 (B,n2,3) sparse, points1 (B,n1,C1)|None,
    points2 (B,n2,C2) -> (B,n1,mlp[-1])."""
    with variable_scope(scope):
        dist, idx = three_nn(xyz1, xyz2)
        if fused_mlp.TAIL_FOLD and dist.is_cuda:            # (:212-215) in one launch, 1/inf = 0 for m<3
            weight = torch.empty_like(dist)
            _lib.call("pcops_three_nn_weights", dist.shape[0], dist.shape[1], dist.data_ptr(), weight.data_ptr())
        else:
            inv = 1.0 / torch.clamp_min(dist, 1e-10)        # (:212-215), 1/inf = 0 for m<3
            weight = inv / inv.sum(dim=2, keepdim=True)
        interpolated = three_interpolate(points2, idx, weight)
        new_points1 = interpolated if points1 is None else torch.cat([interpolated, points1], dim=2)
        new_points1 = _mlp_stack(new_points1.unsqueeze(2), mlp, 'conv_%d', bn, is_training, bn_decay,
                                 'NHWC')
        return new_points1.squeeze(2)
