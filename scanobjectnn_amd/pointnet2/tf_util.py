"""Layer wrappers -- mirror of `pointnet2/utils/tf_util.py` (conv1d :52, conv2d :120,
fully_connected :327, max_pool2d :366, avg_pool2d :391, batch_norm_template :512,
dropout :594) with the same names, argument order and defaults, on torch tensors.

Semantics restated from the reference + the TF 1.10 behaviour it relies on:
  * conv kernels are `weights` [kh,kw,Cin,Cout] (xavier-uniform) + `biases` [Cout] (zeros); only
    the shapes the in-scope models use are implemented: 1x1 convs, and [1,K] VALID convs whose
    input width is K (PointNet's first layer) -- both are dense contractions X(P,Cin*K)·W;
  * BN = tf.contrib.layers.batch_norm(center, scale, decay=bn_decay or 0.9, eps=1e-3,
    updates_collections=None): training uses batch mean / biased variance, the moving variance is
    fed the unbiased batch variance (TF fused path; an assumption), moving stats
    m <- decay*m + (1-decay)*batch;
  * dropout = tf.nn.dropout(keep_prob): kept values scaled by 1/keep_prob, identity in eval.
`is_training` is a Python bool here (the reference feeds a bool placeholder).
"""
import os

import torch
import torch.nn.functional as F

from .. import dist as _dist
from .. import fused_mlp
from ..graph import (constant_initializer, get_variable, truncated_normal_initializer,
                     variable_scope, xavier_initializer, get_default_graph)

relu = F.relu
BN_EPS = 1e-3  # tf.contrib.layers.batch_norm default epsilon


def _variable_with_weight_decay(name, shape, stddev, wd, use_xavier=True):
    """tf_util.py:24-49"""
    init = xavier_initializer() if use_xavier else truncated_normal_initializer(stddev)
    var = get_variable(name, shape, init)
    if wd is not None:
        g = get_default_graph()
        g.end_points.setdefault("losses", []).append(0.5 * wd * (var * var).sum())
    return var


def batch_norm_template(inputs, is_training, scope, moments_dims_unused, bn_decay, data_format='NHWC'):
    """tf_util.py:512-531.  Normalises over every axis but the channel axis."""
    bn_decay = bn_decay if bn_decay is not None else 0.9
    ch_axis = 1 if (data_format == 'NCHW' and inputs.dim() > 2) else inputs.dim() - 1
    c = inputs.shape[ch_axis]
    with variable_scope(scope):
        beta = get_variable('beta', [c], constant_initializer(0.0))
        gamma = get_variable('gamma', [c], constant_initializer(1.0))
        moving_mean = get_variable('moving_mean', [c], constant_initializer(0.0), trainable=False)
        moving_var = get_variable('moving_variance', [c], constant_initializer(1.0), trainable=False)
    if is_training and _dist.sync_bn_active():
        # SyncBN: batch statistics of the global batch (all-reduced sums), same moving-average rule as the local path
        x = inputs.movedim(ch_axis, -1) if ch_axis != inputs.dim() - 1 else inputs
        flat = x.reshape(-1, c)
        mean, var, total = _dist.sync_batch_stats(flat)
        with torch.no_grad():
            d = float(bn_decay)
            moving_mean.mul_(d).add_(mean.detach(), alpha=1.0 - d)
            moving_var.mul_(d).add_(var.detach() * (total / max(total - 1, 1)), alpha=1.0 - d)
        scale = gamma * torch.rsqrt(var + BN_EPS)
        out = (flat * scale + (beta - mean * scale)).reshape(x.shape)
        return out.movedim(-1, ch_axis) if ch_axis != inputs.dim() - 1 else out
    if ch_axis == 1:
        return F.batch_norm(inputs, moving_mean, moving_var, gamma, beta, bool(is_training),
                            1.0 - float(bn_decay), BN_EPS)
    flat = inputs.reshape(-1, c)
    out = F.batch_norm(flat, moving_mean, moving_var, gamma, beta, bool(is_training),
                       1.0 - float(bn_decay), BN_EPS)
    return out.reshape(inputs.shape)


def batch_norm_for_fc(inputs, is_training, bn_decay, scope):
    return batch_norm_template(inputs, is_training, scope, [0, ], bn_decay)


def batch_norm_for_conv1d(inputs, is_training, bn_decay, scope, data_format):
    return batch_norm_template(inputs, is_training, scope, [0, 1], bn_decay, data_format)


def batch_norm_for_conv2d(inputs, is_training, bn_decay, scope, data_format):
    return batch_norm_template(inputs, is_training, scope, [0, 1, 2], bn_decay, data_format)


DOUBLE_BACKWARD = False      # set True to build graphs for gradient-of-gradient uses (penalties, saliency of gradients):
#                              the libpcops linear layers have no second derivative and are then bypassed


def _double_backward_requested():
    return DOUBLE_BACKWARD


def _dense(x2d, kernel2d, biases):
    """X W + b.  Many rows into a few output columns (the per-point heads: 262 144 x 128 -> 2) is a shape the
    library GEMM runs on a handful of CUs, forward and weight gradient alike: those go through the libpcops GEMMs
    with the output padded to 4 columns."""
    rows, k = x2d.shape
    n = kernel2d.shape[1]
    # the classifier / T-Net heads (B x 1024 -> 512 -> 256 -> classes): the regime the small-GEMM kernel was measured
    # in (round 2) -- a few hundred rows, K and N of a head.  Anything else keeps the library GEMM.
    if (x2d.is_cuda and FC_PCOPS and rows <= 4096 and k <= 2048 and n <= 1024 and x2d.dtype == torch.float32
            and not (torch.is_grad_enabled() and _double_backward_requested())):
        return fused_mlp.small_linear(x2d, kernel2d, biases)
    if x2d.is_cuda and k % 8 == 0 and rows >= 32768 and n <= 64:
        pad = (-n) % 4
        w = F.pad(kernel2d, (0, pad)) if pad else kernel2d
        b = F.pad(biases, (0, pad)) if pad else biases
        out = fused_mlp.rows_linear(x2d, w, b)
        return out[:, :n] if pad else out
    return torch.addmm(biases, x2d, kernel2d)


def conv2d(inputs, num_output_channels, kernel_size, scope, stride=[1, 1], padding='SAME',
           data_format='NHWC', use_xavier=True, stddev=1e-3, weight_decay=None,
           activation_fn=relu, bn=False, bn_decay=None, is_training=None):
    """tf_util.py:120-185.  inputs BxHxWxC (NHWC) or BxCxHxW (NCHW)."""
    assert data_format in ('NHWC', 'NCHW')
    kernel_h, kernel_w = kernel_size
    if list(stride) != [1, 1]:
        raise NotImplementedError("conv2d: only stride [1,1] is used by the in-scope models")
    with variable_scope(scope):
        cin = inputs.shape[-1] if data_format == 'NHWC' else inputs.shape[1]
        kernel = _variable_with_weight_decay('weights', [kernel_h, kernel_w, cin, num_output_channels],
                                             stddev=stddev, wd=weight_decay, use_xavier=use_xavier)
        biases = get_variable('biases', [num_output_channels], constant_initializer(0.0))
        if data_format == 'NCHW':
            x = inputs.permute(0, 2, 3, 1)
        else:
            x = inputs
        b, h, w, _ = x.shape
        if kernel_h == 1 and kernel_w == 1:
            out = _dense(x.reshape(-1, cin), kernel.view(cin, num_output_channels), biases)
            out = out.view(b, h, w, num_output_channels)
        elif kernel_h == 1 and kernel_w == w and padding == 'VALID':
            out = _dense(x.reshape(b * h, w * cin), kernel.view(w * cin, num_output_channels), biases)
            out = out.view(b, h, 1, num_output_channels)
        else:
            raise NotImplementedError("conv2d: kernel %s / padding %s not used by the in-scope models"
                                      % (kernel_size, padding))
        if data_format == 'NCHW':
            out = out.permute(0, 3, 1, 2)
        if bn:
            out = batch_norm_for_conv2d(out, is_training, bn_decay=bn_decay, scope='bn',
                                        data_format=data_format)
        if activation_fn is not None:
            out = activation_fn(out)
        return out


def _stack_variables(cin, widths, scope_fmt, stddev, weight_decay, use_xavier, mov_names):
    """creates / fetches exactly the variables a chain of conv2d(..., bn=True) calls would.
    scope_fmt: a '%d' format, or a list/tuple with one scope name per layer."""
    layers = []
    for i, width in enumerate(widths):
        with variable_scope(scope_fmt[i] if isinstance(scope_fmt, (list, tuple)) else scope_fmt % i):
            kernel = _variable_with_weight_decay('weights', [1, 1, cin, width], stddev=stddev,
                                                 wd=weight_decay, use_xavier=use_xavier)
            biases = get_variable('biases', [width], constant_initializer(0.0))
            with variable_scope('bn'):
                beta = get_variable('beta', [width], constant_initializer(0.0))
                gamma = get_variable('gamma', [width], constant_initializer(1.0))
                mm = get_variable(mov_names[0], [width], constant_initializer(0.0), trainable=False)
                mv = get_variable(mov_names[1], [width], constant_initializer(1.0), trainable=False)
        layers.append((kernel.view(cin, width), biases, gamma, beta, mm, mv))
        cin = width
    return layers


FUSED_MLP = os.environ.get("PCOPS_FUSED_MLP", "1") != "0"
# fully connected head (a few hundred rows into 512 / 256 / class columns) on pcops_small_gemm_ex instead of the
# library GEMM.  (Round 2 first tried the big-row GEMM kernels for this: 14.0 ms/step against 13.5 -- they are built
# for millions of rows.  The 32 x 32-tile kernel with a K split is the right shape: see the A/B in DESIGN section 9.)
FC_PCOPS = os.environ.get("PCOPS_FC", "1") != "0"


def conv2d_stack(inputs, widths, scope_fmt, is_training, bn_decay, pool_max=False, use_xavier=True,
                 stddev=1e-3, weight_decay=None, unbiased_moving_var=True,
                 mov_names=('moving_mean', 'moving_variance')):
    """`len(widths)` x conv2d([1,1], VALID, bn=True, relu) in sequence (variables named scope_fmt % i, exactly
    as the reference's loops create them: pointnet_util.py:117-122,186-189,223-227), optionally followed by
    the max over axis 2 (pointnet_util.py:127) -- executed as ONE fused fp32-MFMA pipeline (fused_mlp.py).
    inputs (B,H,W,C) channel-last.  Returns (B,H,W,widths[-1]) or, with pool_max, (B,H,1,widths[-1])."""
    b, h, w, cin = inputs.shape
    layers = _stack_variables(cin, widths, scope_fmt, stddev, weight_decay, use_xavier, mov_names)
    decay = bn_decay if bn_decay is not None else 0.9
    out = fused_mlp.mlp_stack(inputs, w, pool_max, is_training, decay, BN_EPS, unbiased_moving_var, layers)
    return out.view(b, h, 1 if pool_max else w, widths[-1])


def conv1d(inputs, num_output_channels, kernel_size, scope, stride=1, padding='SAME',
           data_format='NHWC', use_xavier=True, stddev=1e-3, weight_decay=None,
           activation_fn=relu, bn=False, bn_decay=None, is_training=None):
    """tf_util.py:52-115.  inputs BxLxC; only kernel_size 1 / stride 1 is used in scope."""
    if kernel_size != 1 or stride != 1:
        raise NotImplementedError("conv1d: only kernel_size=1, stride=1 is used by the in-scope models")
    assert data_format in ('NHWC', 'NCHW')
    with variable_scope(scope):
        x = inputs if data_format == 'NHWC' else inputs.permute(0, 2, 1)
        cin = x.shape[-1]
        kernel = _variable_with_weight_decay('weights', [kernel_size, cin, num_output_channels],
                                             stddev=stddev, wd=weight_decay, use_xavier=use_xavier)
        biases = get_variable('biases', [num_output_channels], constant_initializer(0.0))
        if (bn and activation_fn is relu and FUSED_MLP and data_format == 'NHWC'
                and fused_mlp.fused_supported(x, [num_output_channels], True, True)):
            # conv + bias + BN + ReLU over B*L rows: one single-layer fused stack (same variables as below)
            with variable_scope('bn'):
                beta = get_variable('beta', [num_output_channels], constant_initializer(0.0))
                gamma = get_variable('gamma', [num_output_channels], constant_initializer(1.0))
                mm = get_variable('moving_mean', [num_output_channels], constant_initializer(0.0), trainable=False)
                mv = get_variable('moving_variance', [num_output_channels], constant_initializer(1.0), trainable=False)
            decay = bn_decay if bn_decay is not None else 0.9
            out = fused_mlp.mlp_stack(x, 1, False, is_training, decay, BN_EPS, True,
                                      [(kernel.view(cin, num_output_channels), biases, gamma, beta, mm, mv)])
            return out.view(x.shape[0], x.shape[1], num_output_channels)
        out = _dense(x.reshape(-1, cin), kernel.view(cin, num_output_channels), biases)
        out = out.view(x.shape[0], x.shape[1], num_output_channels)
        if data_format == 'NCHW':
            out = out.permute(0, 2, 1)
        if bn:
            out = batch_norm_for_conv1d(out, is_training, bn_decay=bn_decay, scope='bn',
                                        data_format=data_format)
        if activation_fn is not None:
            out = activation_fn(out)
        return out


def fully_connected(inputs, num_outputs, scope, use_xavier=True, stddev=1e-3, weight_decay=None,
                    activation_fn=relu, bn=False, bn_decay=None, is_training=None):
    """tf_util.py:327-363.  inputs BxN."""
    with variable_scope(scope):
        nin = inputs.shape[-1]
        weights = _variable_with_weight_decay('weights', [nin, num_outputs], stddev=stddev,
                                              wd=weight_decay, use_xavier=use_xavier)
        biases = get_variable('biases', [num_outputs], constant_initializer(0.0))
        out = _dense(inputs, weights, biases)
        if bn and activation_fn in (relu, None) and fused_mlp.fc_batch_norm_supported(out) and not (
                torch.is_grad_enabled() and _double_backward_requested()):
            # BN (+ ReLU) of the head as one launch per direction (csrc/head.hip); the variables of batch_norm_template
            with variable_scope('bn'):
                beta = get_variable('beta', [num_outputs], constant_initializer(0.0))
                gamma = get_variable('gamma', [num_outputs], constant_initializer(1.0))
                moving_mean = get_variable('moving_mean', [num_outputs], constant_initializer(0.0), trainable=False)
                moving_var = get_variable('moving_variance', [num_outputs], constant_initializer(1.0), trainable=False)
            decay = float(bn_decay) if bn_decay is not None else 0.9
            return fused_mlp.fc_batch_norm(out, gamma, beta, moving_mean, moving_var, is_training, decay, BN_EPS, True,
                                           activation_fn is relu)
        if bn:
            out = batch_norm_for_fc(out, is_training, bn_decay, 'bn')
        if activation_fn is not None:
            out = activation_fn(out)
        return out


def max_pool2d(inputs, kernel_size, scope, stride=[2, 2], padding='VALID'):
    """tf_util.py:366-388 (NHWC).  Used as a global pool ([num_point,1]) by the T-Nets."""
    kh, kw = kernel_size
    b, h, w, c = inputs.shape
    if kh == h and kw == w:
        if h == 1 and w == 1:           # a window of one element (the T-Net behind a stack that already pooled): the input itself
            return inputs
        return inputs.amax(dim=(1, 2), keepdim=True)
    x = F.max_pool2d(inputs.permute(0, 3, 1, 2), (kh, kw), stride=tuple(stride))
    return x.permute(0, 2, 3, 1)


def avg_pool2d(inputs, kernel_size, scope, stride=[2, 2], padding='VALID'):
    """tf_util.py:391-414 (NHWC)."""
    kh, kw = kernel_size
    x = F.avg_pool2d(inputs.permute(0, 3, 1, 2), (kh, kw), stride=tuple(stride))
    return x.permute(0, 2, 3, 1)


def dropout(inputs, is_training, scope, keep_prob=0.5, noise_shape=None):
    """tf_util.py:594-615"""
    if noise_shape is not None:
        raise NotImplementedError("dropout: noise_shape is not used by the in-scope models")
    return F.dropout(inputs, p=1.0 - keep_prob, training=bool(is_training))
