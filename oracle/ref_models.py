"""CPU restatement of the model-level callers: PointNet++ SA / SA-MSG
/ FP modules, the SSG / BGA / MSG classifiers, the DGCNN EdgeConv stack and its BGA variant.

TEST INFRASTRUCTURE ONLY (tests/, smoke(), bench.py cpu_baseline).  dtype- and device-generic: fp32 tensors give the
CPU baseline, fp64 tensors the high-precision truth used by the parity tests (the bench-size and multi-seed tests
place the float64 algebra on the GPU -- plain torch ops, the geometry still comes from the C oracle on the host).  Independent of the product's host
code: geometry comes from the C oracle (oracle/pcops_oracle.c), dense algebra from torch-CPU fp32 ops,
variables are looked up by the TF scope names of the reference in a plain dict
(`layer1/conv0/weights`, `layer1/conv0/bn/{beta,gamma,moving_mean,moving_variance}`, `fc1/weights` ...).
"parity unpinned": TensorFlow is absent, so conv/BN/top_k semantics follow TF 1.10 documentation
(DESIGN.md lists every assumption).  Dropout is the identity here (tests disable it on both sides).

Each function cites the reference lines it restates (paths relative to the reference root).
"""
import numpy as np
import torch

from . import oracle as O

EPS = 1e-3


# ------------------------------------------------------------------------------- discrete decisions
class Decisions:
    """The discrete decisions of one evaluation of a network, keyed by the TF scope of the layer:
         relu[scope] : bool (rows, C)   -- which pre-activations passed the ReLU (rows in (b, point, sample) order)
         pool[scope] : (arg long (groups, C), active bool (groups, C)) -- for a layer whose output is max-pooled:
                       the member the maximum was taken from (index along the pooled axis) and whether that member
                       passed the ReLU.  (The ReLU of the other members of a pooled layer decides nothing: the
                       maximum of relu(z) is relu of the maximum, and only the arg-max member receives gradient.)
    A network is a smooth function of its parameters ONCE these are fixed.  An fp32 evaluation whose pre-activation
    sits within rounding of 0 (or whose two best pool members tie within rounding) can decide differently from a
    float64 evaluation; that is not an arithmetic error of either, but it moves gradient elements by O(1).  The
    parity tests therefore evaluate the float64 truth a second time WITH THE DECISIONS THE PATH UNDER TEST TOOK
    (`imposing(...)`), count and bound the decisions that differ (`report`), and compare against that."""

    def __init__(self):
        self.relu = {}
        self.pool = {}


_FORCE = None      # Decisions imposed on the run
_RECORD = None     # Decisions that receives the run's own
_REPORT = None     # {scope: {...}} how the imposed decisions differ from the run's own


class imposing:
    """with imposing(D=None, record=None, report=None): ref_fn(...)"""

    def __init__(self, decisions=None, record=None, report=None):
        self.new = (decisions, record, report)

    def __enter__(self):
        global _FORCE, _RECORD, _REPORT
        self.old = (_FORCE, _RECORD, _REPORT)
        _FORCE, _RECORD, _REPORT = self.new
        return self

    def __exit__(self, *a):
        global _FORCE, _RECORD, _REPORT
        _FORCE, _RECORD, _REPORT = self.old


def _act(z, scope):
    """ReLU of the pre-activation z of layer `scope`, with the run's own mask or an imposed one"""
    if _FORCE is None and _RECORD is None:
        return torch.relu(z)
    c = z.shape[-1]
    own = z.detach() > 0
    if _RECORD is not None:
        _RECORD.relu[scope] = own.reshape(-1, c)
    m = _FORCE.relu.get(scope) if _FORCE is not None else None
    if m is None:
        return torch.relu(z)
    m = m.to(z.device).reshape(z.shape)
    if _REPORT is not None:
        dis = m != own
        n = int(dis.sum())
        _REPORT[scope] = {"kind": "relu", "elements": own.numel(), "flips": n,
                          "worst_abs_z": float(z.detach()[dis].abs().max()) if n else 0.0}
    return z * m.to(z.dtype)


def _act_pool(z, scope, dim):
    """max over axis `dim` of relu(z) for the pooled layer `scope` (z = its pre-activation): the run's own arg-max /
    activity, or imposed ones"""
    if _FORCE is None and _RECORD is None:
        return torch.relu(z).amax(dim=dim)
    zd = z.detach()
    own_max, own_arg = zd.max(dim=dim)
    c = z.shape[-1]
    if _RECORD is not None:
        _RECORD.pool[scope] = (own_arg.reshape(-1, c), (own_max > 0).reshape(-1, c))
    f = _FORCE.pool.get(scope) if _FORCE is not None else None
    if f is None:
        return torch.relu(z).amax(dim=dim)
    arg, active = f[0].to(z.device).reshape(own_arg.shape), f[1].to(z.device).reshape(own_arg.shape)
    picked = torch.gather(z, dim, arg.unsqueeze(dim)).squeeze(dim)
    if _REPORT is not None:
        gap = (own_max - picked.detach())                      # >= 0; 0 for the same member or an exact copy of it
        other = (gap > 0) & active                             # an ACTIVE imposed member that is not this run's maximum
        aflip = active != (own_max > 0)
        # an activity flip is a tie when the deciding value sits at 0 in this run: the imposed member's value if the imposed
        # decision is "active"; this run's own maximum if it is "inactive" (the imposed member is then arbitrary -- every
        # member ties at relu(.) = 0 -- and its gap to the maximum means nothing)
        za = torch.where(active, picked.detach(), own_max).abs()
        _REPORT[scope] = {"kind": "pool", "elements": own_arg.numel(), "flips": int(other.sum()),
                          "worst_gap": float(gap[other].max()) if other.any() else 0.0,
                          "active_flips": int(aflip.sum()),
                          "worst_abs_z": float(za[aflip].max()) if aflip.any() else 0.0}
    return picked * active.to(z.dtype)


def _np(t):
    return t.detach().cpu().numpy()


def _idx(a):
    return torch.from_numpy(np.ascontiguousarray(a)).long()


def batch_gather(x, idx):
    """x (B,N,C), idx (B,...) long -> (B,...,C)"""
    b = x.shape[0]
    flat = idx.reshape(b, -1).to(x.device)
    out = torch.gather(x, 1, flat.unsqueeze(-1).expand(-1, -1, x.shape[2]))
    return out.reshape(*idx.shape, x.shape[2])


def bn(x, P, scope, training, flavour="contrib"):
    """pointnet2/utils/tf_util.py:512-531 ("contrib") / dgcnn/utils/tf_util.py:462-535 ("moments")"""
    c = x.shape[-1]
    flat = x.reshape(-1, c)
    if training:
        var, mean = torch.var_mean(flat, dim=0, unbiased=False)
    else:
        if flavour == "dist":
            mean, var = P[scope + "/pop_mean"], P[scope + "/pop_var"]
        else:
            mean, var = P[scope + "/moving_mean"], P[scope + "/moving_variance"]
    out = (flat - mean) * torch.rsqrt(var + EPS) * P[scope + "/gamma"] + P[scope + "/beta"]
    return out.reshape(x.shape)


def dense(x, P, scope, training, use_bn=True, act=True, flavour="contrib", pool_dim=None):
    """1x1 conv / conv1d(1) / fully_connected: X·W + b -> BN -> ReLU (tf_util.py:120-185,327-363);
    pool_dim: followed by the max over that axis (pointnet_util.py:127, dgcnn.py:47,84)"""
    w = P[scope + "/weights"]
    w = w.reshape(-1, w.shape[-1])
    out = x.reshape(-1, x.shape[-1]) @ w + P[scope + "/biases"]
    out = out.reshape(*x.shape[:-1], w.shape[-1])
    if use_bn:
        out = bn(out, P, scope + "/bn", training, flavour)
    if pool_dim is not None:
        return _act_pool(out, scope, pool_dim)
    return _act(out, scope) if act else out


def sample_and_group(npoint, radius, nsample, xyz, points, use_xyz=True):
    """pointnet2/utils/pointnet_util.py:22-56"""
    fps = O.farthest_point_sample(npoint, _np(xyz))
    new_xyz = batch_gather(xyz, _idx(fps))
    idx, _ = O.query_ball_point(radius, nsample, _np(xyz), _np(new_xyz))
    idx = _idx(idx)
    grouped_xyz = batch_gather(xyz, idx) - new_xyz.unsqueeze(2)
    if points is None:
        return new_xyz, grouped_xyz, idx
    grouped_points = batch_gather(points, idx)
    return new_xyz, (torch.cat([grouped_xyz, grouped_points], -1) if use_xyz else grouped_points), idx


def sa_module(xyz, points, npoint, radius, nsample, mlp, P, scope, training, group_all=False):
    """pointnet2/utils/pointnet_util.py:87-154 (pooling='max', mlp2=None)"""
    if group_all:
        b = xyz.shape[0]
        new_xyz = torch.zeros((b, 1, 3), dtype=xyz.dtype, device=xyz.device)
        new_points = (xyz if points is None else torch.cat([xyz, points], 2)).unsqueeze(1)
    else:
        new_xyz, new_points, _ = sample_and_group(npoint, radius, nsample, xyz, points)
    for i in range(len(mlp)):
        new_points = dense(new_points, P, "%s/conv%d" % (scope, i), training, pool_dim=2 if i == len(mlp) - 1 else None)
    return new_xyz, new_points


def sa_module_msg(xyz, points, npoint, radius_list, nsample_list, mlp_list, P, scope, training):
    """pointnet2/utils/pointnet_util.py:156-196; concat order [feats | xyz] (:184)"""
    fps = O.farthest_point_sample(npoint, _np(xyz))
    new_xyz = batch_gather(xyz, _idx(fps))
    outs = []
    for i, (r, s) in enumerate(zip(radius_list, nsample_list)):
        idx = _idx(O.query_ball_point(r, s, _np(xyz), _np(new_xyz))[0])
        g = batch_gather(xyz, idx) - new_xyz.unsqueeze(2)
        if points is not None:
            g = torch.cat([batch_gather(points, idx), g], -1)
        for j in range(len(mlp_list[i])):
            g = dense(g, P, "%s/conv%d_%d" % (scope, i, j), training, pool_dim=2 if j == len(mlp_list[i]) - 1 else None)
        outs.append(g)
    return new_xyz, torch.cat(outs, -1)


def fp_module(xyz1, xyz2, points1, points2, mlp, P, scope, training):
    """pointnet2/utils/pointnet_util.py:199-229"""
    dist, idx = O.three_nn(_np(xyz1), _np(xyz2))
    dist = torch.clamp_min(torch.from_numpy(dist).to(points2), 1e-10)
    inv = 1.0 / dist
    w = inv / inv.sum(dim=2, keepdim=True)
    nb = batch_gather(points2, _idx(idx))                     # (B,n,3,C)
    interp = nb[:, :, 0] * w[:, :, 0:1] + nb[:, :, 1] * w[:, :, 1:2] + nb[:, :, 2] * w[:, :, 2:3]
    x = interp if points1 is None else torch.cat([interp, points1], 2)
    for i in range(len(mlp)):
        x = dense(x, P, "%s/conv_%d" % (scope, i), training)
    return x


def _cls_head(feat, P, training, num_class_scope="fc3"):
    net = dense(feat, P, "fc1", training)
    net = dense(net, P, "fc2", training)
    return net, dense(net, P, num_class_scope, training, use_bn=False, act=False)


def pointnet2_cls_ssg(point_cloud, P, training):
    """pointnet2/models/pointnet2_cls_ssg.py:23-47"""
    l1_xyz, l1 = sa_module(point_cloud, None, 512, 0.2, 32, [64, 64, 128], P, "layer1", training)
    l2_xyz, l2 = sa_module(l1_xyz, l1, 128, 0.4, 64, [128, 128, 256], P, "layer2", training)
    _, l3 = sa_module(l2_xyz, l2, None, None, None, [256, 512, 1024], P, "layer3", training, group_all=True)
    return _cls_head(l3.reshape(point_cloud.shape[0], -1), P, training)[1]


def pointnet2_cls_msg(point_cloud, P, training):
    """layer: pointnet_util.py:156-196; hyper-parameters: upstream PointNet++ (not in the reference)"""
    l1_xyz, l1 = sa_module_msg(point_cloud, None, 512, [0.1, 0.2, 0.4], [16, 32, 128],
                               [[32, 32, 64], [64, 64, 128], [64, 96, 128]], P, "layer1", training)
    l2_xyz, l2 = sa_module_msg(l1_xyz, l1, 128, [0.2, 0.4, 0.8], [32, 64, 128],
                               [[64, 64, 128], [128, 128, 256], [128, 128, 256]], P, "layer2", training)
    _, l3 = sa_module(l2_xyz, l2, None, None, None, [256, 512, 1024], P, "layer3", training, group_all=True)
    return _cls_head(l3.reshape(point_cloud.shape[0], -1), P, training)[1]


def pointnet2_cls_bga(point_cloud, P, training):
    """pointnet2/models/pointnet2_cls_bga.py:21-75 -> (class_pred, seg_pred)"""
    l0_xyz = point_cloud[:, :, :3]
    l1_xyz, l1 = sa_module(l0_xyz, None, 512, 0.2, 64, [64, 64, 128], P, "layer1", training)
    l2_xyz, l2 = sa_module(l1_xyz, l1, 128, 0.4, 64, [128, 128, 256], P, "layer2", training)
    l3_xyz, l3 = sa_module(l2_xyz, l2, None, None, None, [256, 512, 1024], P, "layer3", training, group_all=True)
    fc2, class_pred = _cls_head(l3.reshape(point_cloud.shape[0], -1), P, training)
    class_vector = fc2.unsqueeze(1)
    l2p = fp_module(l2_xyz, l3_xyz, l2, class_vector, [256, 256], P, "fa_layer1", training)
    l1p = fp_module(l1_xyz, l2_xyz, l1, l2p, [256, 128], P, "fa_layer2", training)
    l0p = fp_module(l0_xyz, l1_xyz, None, l1p, [128, 128, 128], P, "fa_layer3", training)
    net = dense(l0p, P, "seg_fc1", training)
    return class_pred, dense(net, P, "seg_fc2", training, use_bn=False, act=False)


def pointnet2_cls_partseg(point_cloud, P, training):
    """pointnet2/models/pointnet2_cls_partseg.py:20-45 -> seg_pred (B,N,6)"""
    l0_xyz = point_cloud[:, :, :3]
    l1_xyz, l1 = sa_module(l0_xyz, None, 512, 0.2, 64, [64, 64, 128], P, "layer1", training)
    l2_xyz, l2 = sa_module(l1_xyz, l1, 128, 0.4, 64, [128, 128, 256], P, "layer2", training)
    l3_xyz, l3 = sa_module(l2_xyz, l2, None, None, None, [256, 512, 1024], P, "layer3", training, group_all=True)
    l2p = fp_module(l2_xyz, l3_xyz, l2, l3, [256, 256], P, "fa_layer1", training)
    l1p = fp_module(l1_xyz, l2_xyz, l1, l2p, [256, 128], P, "fa_layer2", training)
    l0p = fp_module(l0_xyz, l1_xyz, None, l1p, [128, 128, 128], P, "fa_layer3", training)
    net = dense(l0p, P, "seg_fc1", training)
    return dense(net, P, "seg_fc2", training, use_bn=False, act=False)


# ------------------------------------------------------------------------------- DGCNN
def _edge_features(x, k, nn=None):
    """dgcnn/utils/tf_util.py:638-706 via the fused oracle graph.  `nn` overrides the graph (tests feed
    the product's own indices so that a 1e-6 feature difference cannot flip a near-tie neighbour)."""
    nn = _idx(O.knn_graph(_np(x), k) if nn is None else nn)
    nb = batch_gather(x, nn)                                   # (B,N,k,C)
    ctr = x.unsqueeze(2).expand_as(nb)
    return torch.cat([ctr, nb - ctr], -1)


def _dgcnn_backbone(point_cloud, P, training, k=20, nn_list=None):
    """dgcnn/models/dgcnn.py:31-86, transform_nets.py:10-55"""
    b, n, _ = point_cloud.shape
    nn_list = list(nn_list) if nn_list is not None else [None] * 5
    ef = _edge_features(point_cloud, k, nn_list[0])
    t = dense(ef, P, "transform_net1/tconv1", training, flavour="moments")
    t = dense(t, P, "transform_net1/tconv2", training, flavour="moments", pool_dim=2).unsqueeze(2)
    t = dense(t, P, "transform_net1/tconv3", training, flavour="moments", pool_dim=1).reshape(b, -1)
    t = dense(t, P, "transform_net1/tfc1", training, flavour="moments")
    t = dense(t, P, "transform_net1/tfc2", training, flavour="moments")
    tr = t @ P["transform_net1/transform_XYZ/weights"] + (P["transform_net1/transform_XYZ/biases"]
                                                          + torch.eye(3, dtype=t.dtype, device=t.device).flatten())
    x = point_cloud @ tr.reshape(b, 3, 3)
    nets = []
    for li, scope in enumerate(("dgcnn1", "dgcnn2", "dgcnn3", "dgcnn4")):
        x = dense(_edge_features(x, k, nn_list[li + 1]), P, scope, training, flavour="moments", pool_dim=2)
        nets.append(x)
    out_max = dense(torch.cat(nets, -1), P, "agg", training, flavour="moments", pool_dim=1)   # max over the points of (B,N,1024)
    return nets, out_max


def dgcnn(point_cloud, P, training, nn_list=None):
    """dgcnn/models/dgcnn.py:24-102"""
    _, out_max = _dgcnn_backbone(point_cloud, P, training, nn_list=nn_list)
    net = dense(out_max, P, "fc1", training, flavour="moments")
    net = dense(net, P, "fc2", training, flavour="moments")
    return dense(net, P, "fc3", training, use_bn=False, act=False)


def dgcnn_bga(point_cloud, P, training, nn_list=None):
    """dgcnn/models/dgcnn_bga.py:27-134 -> (class_pred, seg_pred)"""
    b, n, _ = point_cloud.shape
    nets, out_max = _dgcnn_backbone(point_cloud, P, training, nn_list=nn_list)
    net = dense(out_max, P, "fc1", training, flavour="moments")
    fc2 = dense(net, P, "fc2", training, flavour="moments")
    class_pred = dense(fc2, P, "fc3", training, use_bn=False, act=False)
    cat = torch.cat([fc2.unsqueeze(1).expand(b, n, 256), out_max.unsqueeze(1).expand(b, n, 1024)] + nets, -1)
    s = dense(cat, P, "seg/conv1", training, flavour="dist")
    s = dense(s, P, "seg/conv2", training, flavour="dist")
    return class_pred, dense(s, P, "seg/conv3", training, use_bn=False, act=False)


def spidercnn_cls_xyz(point_cloud, P, training):
    """SpiderCNN/models/spidercnn_cls_xyz.py:20-71 with SpiderCNN/utils/tf_util.py:127-236,363-377,407-429"""
    xyz = point_cloud[:, :, :3]
    b, n, _ = xyz.shape
    k, G, T = 20, 16, 5
    _, idx = O.knn_point(k, _np(xyz), _np(xyz))
    idx = _idx(idx)
    delta = batch_gather(xyz, idx) - xyz.unsqueeze(2)
    X, Y, Z = delta[..., 0:1], delta[..., 1:2], delta[..., 2:3]
    feats, feat = [], xyz
    for li, width in enumerate((32, 64, 128, 256)):
        s = "fanConv%d/taylor/" % (li + 1)
        w = lambda name: P[s + "weight_" + name]          # noqa: E731
        g_d = (w("x") * X + w("y") * Y + w("z") * Z + w("xyz") * X * Y * Z) \
            + (w("xy") * X * Y + w("yz") * Y * Z + w("xz") * X * Z + P[s + "biases"]) \
            + (w("xx") * X * X + w("yy") * Y * Y + w("zz") * Z * Z) \
            + (w("xxy") * X * X * Y + w("xyy") * X * Y * Y + w("xxz") * X * X * Z) \
            + (w("xzz") * X * Z * Z + w("yyz") * Y * Y * Z + w("yzz") * Y * Z * Z) \
            + (w("xxx") * X * X * X + w("yyy") * Y * Y * Y + w("zzz") * Z * Z * Z)
        grouped = batch_gather(feat, idx)                                    # (B,N,k,C)
        c = grouped.shape[-1]
        x = (grouped.unsqueeze(-1) * g_d.unsqueeze(3)).reshape(b * n, k * c * T)
        out = x @ P[s + "conv/weights"].reshape(k * c * T, width) + P[s + "conv/biases"]
        og = out.reshape(b, n, min(G, width), width // min(G, width)).permute(0, 2, 3, 1)   # (B,G,C/G,N)
        var, mean = torch.var_mean(og, dim=(2, 3), unbiased=False, keepdim=True)
        og = ((og - mean) / torch.sqrt(var + 1e-6)).permute(0, 3, 1, 2).reshape(b, n, width)
        feat = torch.relu(og * P[s + "conv/gn/gamma"] + P[s + "conv/gn/beta"])
        feats.append(feat)
    net = torch.topk(torch.cat(feats, 2).permute(0, 2, 1), 2, dim=2).values.reshape(b, -1)
    net = dense(net, P, "fc1", training)
    net = dense(net, P, "fc2", training)
    return dense(net, P, "fc3", training, use_bn=False, act=False)


def params_from_state_dict(sd, prefix="graph.", dtype=torch.float32, device="cpu"):
    """product Model.state_dict() -> {tf_scope_name: cpu tensor}.  dtype=torch.float64 gives the
    high-precision "truth" the fp32 paths are judged against (every function here follows its inputs'
    dtype; the geometry oracle always sees the fp32 coordinates, which doubles hold exactly)."""
    return {k[len(prefix):] if k.startswith(prefix) else k: v.detach().to(device=device, dtype=dtype).clone()
            for k, v in sd.items()}


# ------------------------------------------------------------------------------- PointNet (config 1)
def _tnet(x, P, scope, training, head, k_out):
    """pointnet/models/transform_nets.py:10-95; x (B,N,C)"""
    t = dense(x, P, scope + "/tconv1", training)
    t = dense(t, P, scope + "/tconv2", training)
    t = dense(t, P, scope + "/tconv3", training).amax(dim=1)
    t = dense(t, P, scope + "/tfc1", training)
    t = dense(t, P, scope + "/tfc2", training)
    out = t @ P["%s/%s/weights" % (scope, head)] + (P["%s/%s/biases" % (scope, head)] + torch.eye(k_out, dtype=t.dtype, device=t.device).flatten())
    return out.reshape(x.shape[0], k_out, k_out)


def pointnet_cls(point_cloud, P, training):
    """pointnet/models/pointnet_cls.py:21-75 -> (logits, feature transform)"""
    t1 = _tnet(point_cloud, P, "transform_net1", training, "transform_XYZ", 3)
    net = dense(point_cloud @ t1, P, "conv1", training)
    net = dense(net, P, "conv2", training)
    t2 = _tnet(net, P, "transform_net2", training, "transform_feat", 64)
    net = net @ t2
    for s in ("conv3", "conv4", "conv5"):
        net = dense(net, P, s, training)
    net = dense(net.amax(dim=1), P, "fc1", training)
    net = dense(net, P, "fc2", training)
    return dense(net, P, "fc3", training, use_bn=False, act=False), t2
