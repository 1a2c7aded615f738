"""SpiderCNN layers -- mirror of `SpiderCNN/utils/tf_util.py`: conv2d with group norm (:59-125), spiderConv (:127-236),
topk_pool (:363-377), group_norm_for_conv (:407-429).  The neighbourhood front end (kNN + grouping) runs on the
libpcops ops; the Taylor-kernel algebra is a handful of dense torch ops (no new kernels)."""
import torch

from ..graph import constant_initializer, get_variable, variable_scope, xavier_initializer
from ..pointnet2.tf_grouping import group_point
from ..pointnet2.tf_util import dropout, fully_connected  # noqa: F401  (tf.contrib batch norm flavour, :493-512)

relu = torch.relu
_TAYLOR = ('x', 'y', 'z', 'xyz', 'xy', 'yz', 'xz', 'xx', 'yy', 'zz', 'xxy', 'xyy', 'xxz', 'xzz', 'yyz', 'yzz',
           'xxx', 'yyy', 'zzz')


def group_norm_for_conv(x, G=32, esp=1e-6, scope='gn'):
    """x (B,H,W,C): statistics over (C/G, H, W) per (sample, group), biased variance, per-channel gamma / beta"""
    with variable_scope(scope):
        n, h, w, c = x.shape
        g = min(G, c)
        xg = x.permute(0, 3, 1, 2).reshape(n, g, c // g, h, w)
        var, mean = torch.var_mean(xg, dim=(2, 3, 4), unbiased=False, keepdim=True)
        xg = (xg - mean) / torch.sqrt(var + esp)
        gamma = get_variable('gamma', [c], constant_initializer(1.0))
        beta = get_variable('beta', [c], constant_initializer(0.0))
        out = xg.reshape(n, c, h, w) * gamma.view(1, c, 1, 1) + beta.view(1, c, 1, 1)
        return out.permute(0, 2, 3, 1)


def conv2d(inputs, num_output_channels, kernel_size, scope, stride=(1, 1), padding='SAME', activation_fn=relu,
           gn=False, G=32):
    """only the form spiderConv uses: a [1, K] VALID kernel over the neighbour axis = one contraction over (k, cin)"""
    with variable_scope(scope):
        kh, kw = kernel_size
        b, h, w, cin = inputs.shape
        if not (kh == 1 and kw == w and padding == 'VALID' and tuple(stride) == (1, 1)):
            raise NotImplementedError("spidercnn conv2d: only the [1, K] VALID kernel of spiderConv is mirrored")
        kernel = get_variable('weights', [kh, kw, cin, num_output_channels], xavier_initializer())
        biases = get_variable('biases', [num_output_channels], constant_initializer(0.0))
        out = torch.addmm(biases, inputs.reshape(b * h, kw * cin), kernel.reshape(kw * cin, num_output_channels))
        out = out.view(b, h, 1, num_output_channels)
        if gn:
            out = group_norm_for_conv(out, G=G, scope='gn')
        return activation_fn(out) if activation_fn is not None else out


def spiderConv(feat, idx, delta, num_conv, taylor_channel, gn=False, G=32, activation_fn=relu, scope='taylor'):
    """feat (B,N,C), idx (B,N,k) int32, delta (B,N,k,3) -> (B,N,num_conv): neighbour features weighted by
    taylor_channel order-3 polynomials of the offset, then a [1,k] conv (:168-236)"""
    with variable_scope(scope):
        grouped = group_point(feat.contiguous(), idx)                       # (B,N,k,C)
        b, n, k, c = grouped.shape
        X, Y, Z = delta[..., 0:1], delta[..., 1:2], delta[..., 2:3]
        w = {}
        for name in _TAYLOR[:4]:
            w[name] = get_variable('weight_' + name, [1, 1, 1, taylor_channel], xavier_initializer())
        for name in _TAYLOR[4:7]:
            w[name] = get_variable('weight_' + name, [1, 1, 1, taylor_channel], xavier_initializer())
        biases = get_variable('biases', [1, 1, 1, taylor_channel], constant_initializer(0.0))
        for name in _TAYLOR[7:]:
            w[name] = get_variable('weight_' + name, [1, 1, 1, taylor_channel], xavier_initializer())
        g1 = w['x'] * X + w['y'] * Y + w['z'] * Z + w['xyz'] * X * Y * Z
        g2 = w['xy'] * X * Y + w['yz'] * Y * Z + w['xz'] * X * Z + biases
        g3 = w['xx'] * X * X + w['yy'] * Y * Y + w['zz'] * Z * Z
        g4 = w['xxy'] * X * X * Y + w['xyy'] * X * Y * Y + w['xxz'] * X * X * Z
        g5 = w['xzz'] * X * Z * Z + w['yyz'] * Y * Y * Z + w['yzz'] * Y * Z * Z
        g6 = w['xxx'] * X * X * X + w['yyy'] * Y * Y * Y + w['zzz'] * Z * Z * Z
        g_d = g1 + g2 + g3 + g4 + g5 + g6                                   # (B,N,k,T)
        x = (grouped.unsqueeze(-1) * g_d.unsqueeze(3)).reshape(b, n, k, c * taylor_channel)
        out = conv2d(x, num_conv, [1, k], padding='VALID', stride=[1, 1], scope='conv', gn=gn, G=G,
                     activation_fn=activation_fn)
        return out.squeeze(2)


def topk_pool(inputs, scope, k=2):
    """inputs (B,N,C) -> the k largest values per channel over the points, (B,C,k), descending (:363-377)"""
    with variable_scope(scope):
        return torch.topk(inputs.permute(0, 2, 1), k, dim=2).values
