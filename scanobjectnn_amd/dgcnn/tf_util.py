"""DGCNN layer wrappers + dynamic-graph ops -- mirror of `dgcnn/utils/tf_util.py`
(conv2d :115, fully_connected :317, max_pool2d :357, batch_norm_template :462 (EMA flavour),
batch_norm_dist_template :502 ("dist" flavour, used with is_dist=True), dropout :614,
pairwise_distance :638, knn :660, get_edge_feature :674) on libpcops.

BN here is the explicit `tf.nn.moments` flavour: batch mean / BIASED variance in training, eps 1e-3,
moving stats m <- decay*m + (1-decay)*batch with decay = bn_decay or 0.9, biased variance in the moving
stats too (unlike the pointnet2 flavour).  The EMA flavour's zero-debiasing of TF's
ExponentialMovingAverage on tensors is not reproduced (eval-only detail).

The graph ops keep the reference's three-function API; `knn_graph` is the fused fast path that never
materialises the (B,N,N) adjacency and yields the same indices.
"""
import os

import torch
import torch.nn.functional as F

from .. import _lib, fused_mlp
from .. import dist as _dist
from ..graph import constant_initializer, get_variable, variable_scope
from ..pointnet2 import tf_util as _pn2
from ..pointnet2.tf_util import (_dense, _variable_with_weight_decay, avg_pool2d, dropout,  # noqa: F401
                                 max_pool2d, relu)

BN_EPS = 1e-3


def _batch_norm(inputs, is_training, scope, bn_decay, names):
    c = inputs.shape[-1]
    decay = float(bn_decay) if bn_decay is not None else 0.9
    with variable_scope(scope):
        beta = get_variable('beta', [c], constant_initializer(0.0))
        gamma = get_variable('gamma', [c], constant_initializer(1.0))
        mov_mean = get_variable(names[0], [c], constant_initializer(0.0), trainable=False)
        mov_var = get_variable(names[1], [c], constant_initializer(1.0), trainable=False)
    flat = inputs.reshape(-1, c)
    if is_training:
        if _dist.sync_bn_active():
            mean, var, _ = _dist.sync_batch_stats(flat)
        else:
            var, mean = torch.var_mean(flat, dim=0, unbiased=False)
        with torch.no_grad():
            mov_mean.mul_(decay).add_(mean.detach(), alpha=1.0 - decay)
            mov_var.mul_(decay).add_(var.detach(), alpha=1.0 - decay)
    else:
        mean, var = mov_mean, mov_var
    scale = gamma * torch.rsqrt(var + BN_EPS)
    out = flat * scale + (beta - mean * scale)
    return out.reshape(inputs.shape)


def batch_norm_template(inputs, is_training, scope, moments_dims, bn_decay):
    """tf_util.py:462-499"""
    return _batch_norm(inputs, is_training, scope, bn_decay, ('moving_mean', 'moving_variance'))


def batch_norm_dist_template(inputs, is_training, scope, moments_dims, bn_decay):
    """tf_util.py:502-535"""
    return _batch_norm(inputs, is_training, scope, bn_decay, ('pop_mean', 'pop_var'))


def _bn(inputs, is_training, bn_decay, scope, is_dist):
    fn = batch_norm_dist_template if is_dist else batch_norm_template
    return fn(inputs, is_training, scope, None, bn_decay)


def conv2d(inputs, num_output_channels, kernel_size, scope, stride=[1, 1], padding='SAME',
           use_xavier=True, stddev=1e-3, weight_decay=0.0, activation_fn=relu, bn=False,
           bn_decay=None, is_training=None, is_dist=False):
    """tf_util.py:115-173 (NHWC only).  1x1 kernels (and [1,K] VALID over width K)."""
    kernel_h, kernel_w = kernel_size
    if list(stride) != [1, 1]:
        raise NotImplementedError("conv2d: only stride [1,1] is used by the in-scope models")
    with variable_scope(scope):
        b, h, w, cin = inputs.shape
        kernel = _variable_with_weight_decay('weights', [kernel_h, kernel_w, cin, num_output_channels],
                                             stddev=stddev, wd=weight_decay or None, use_xavier=use_xavier)
        biases = get_variable('biases', [num_output_channels], constant_initializer(0.0))
        if kernel_h == 1 and kernel_w == 1:
            out = _dense(inputs.reshape(-1, cin), kernel.view(cin, num_output_channels), biases)
            out = out.view(b, h, w, num_output_channels)
        elif kernel_h == 1 and kernel_w == w and padding == 'VALID':
            out = _dense(inputs.reshape(b * h, w * cin), kernel.view(w * cin, num_output_channels), biases)
            out = out.view(b, h, 1, num_output_channels)
        else:
            raise NotImplementedError("conv2d: kernel %s / padding %s not used in scope" % (kernel_size, padding))
        if bn:
            out = _bn(out, is_training, bn_decay, 'bn', is_dist)
        if activation_fn is not None:
            out = activation_fn(out)
        return out


def fully_connected(inputs, num_outputs, scope, use_xavier=True, stddev=1e-3, weight_decay=0.0,
                    activation_fn=relu, bn=False, bn_decay=None, is_training=None, is_dist=False):
    """tf_util.py:317-354"""
    with variable_scope(scope):
        nin = inputs.shape[-1]
        weights = _variable_with_weight_decay('weights', [nin, num_outputs], stddev=stddev,
                                              wd=weight_decay or None, use_xavier=use_xavier)
        biases = get_variable('biases', [num_outputs], constant_initializer(0.0))
        out = _dense(inputs, weights, biases)
        if bn and activation_fn in (relu, None) and fused_mlp.fc_batch_norm_supported(out) and not (
                torch.is_grad_enabled() and _pn2._double_backward_requested()):
            # BN (+ ReLU) of the head as one launch per direction (csrc/head.hip); same variables as _batch_norm creates
            names = ('pop_mean', 'pop_var') if is_dist else ('moving_mean', 'moving_variance')
            with variable_scope('bn'):
                beta = get_variable('beta', [num_outputs], constant_initializer(0.0))
                gamma = get_variable('gamma', [num_outputs], constant_initializer(1.0))
                mov_mean = get_variable(names[0], [num_outputs], constant_initializer(0.0), trainable=False)
                mov_var = get_variable(names[1], [num_outputs], constant_initializer(1.0), trainable=False)
            decay = float(bn_decay) if bn_decay is not None else 0.9
            return fused_mlp.fc_batch_norm(out, gamma, beta, mov_mean, mov_var, is_training, decay, BN_EPS, False,
                                           activation_fn is relu)
        if bn:
            out = _bn(out, is_training, bn_decay, 'bn', is_dist)
        if activation_fn is not None:
            out = activation_fn(out)
        return out


# ---------------------------------------------------------------------------- graph ops
def _squeeze_cloud(point_cloud):
    """tf_util.py:647-650 / :685-688: tf.squeeze, re-expanding the batch axis when it was 1."""
    og = point_cloud.shape[0]
    x = point_cloud.squeeze()
    if og == 1:
        x = x.unsqueeze(0)
    if x.dim() != 3:
        raise ValueError("expected (B,N,C) or (B,N,1,C) with N,C > 1, got %s" % (tuple(point_cloud.shape),))
    return x


def pairwise_distance(point_cloud):
    """(B,N,C) or (B,N,1,C) -> (B,N,N) f32, D_ij = (s_i + (-2<x_i,x_j>)) + s_j"""
    x = _lib.check(_squeeze_cloud(point_cloud).detach(), torch.float32, "point_cloud", 3)
    b, n, c = x.shape
    adj = torch.empty((b, n, n), dtype=torch.float32, device=x.device)
    _lib.call("pcops_pairwise_distance", b, n, c, _lib.ptr(x), _lib.ptr(adj))
    return adj


def knn(adj_matrix, k=20):
    """(B,N,N) -> (B,N,k) i32: k smallest per row, ties -> lower index (tf.nn.top_k(-adj))"""
    adj = _lib.check(adj_matrix.detach(), torch.float32, "adj_matrix", 3)
    b, n, n2 = adj.shape
    if not 0 < k <= n2:
        raise ValueError("input must have at least k columns")  # tf.nn.top_k
    out = torch.empty((b, n, k), dtype=torch.int32, device=adj.device)
    _lib.call("pcops_knn_topk", b * n, n2, k, _lib.ptr(adj), _lib.ptr(out))
    return out


KNN_SEED = os.environ.get("PCOPS_KNN_SEED", "1") != "0"
KNN_SEED_MAX_C = int(os.environ.get("PCOPS_KNN_SEED_MAX_C", "64"))
KNN_SEED_FORCE = False       # tests: take the hint whatever the shape (every seeded kernel variant is then exercised)


def knn_graph(point_cloud, k=20, seed=None):
    """fused pairwise_distance + knn: (B,N,C)|(B,N,1,C) -> (B,N,k) i32, identical indices.
    seed: (B,N,k) i32 neighbour lists of an EARLIER graph of the same points (each row k distinct indices): a hint that
    lets the kernel reject most candidates with one compare (pcops_knn_graph_seeded) -- same result."""
    x = _lib.check(_squeeze_cloud(point_cloud).detach(), torch.float32, "point_cloud", 3)
    b, n, c = x.shape
    if not 0 < k <= n:
        raise ValueError("input must have at least k columns")
    out = torch.empty((b, n, k), dtype=torch.int32, device=x.device)
    # measured at the DGCNN config (B = 256, N = 2048, k = 20, MI355X): coordinate graphs 905 -> 613 us with the previous
    # graph as the hint.  64-channel graphs: on the fp32-MFMA kernel the hint LOST (round 3: 2445 -> 2750 us, the k seed
    # rows cost more than the queue entries they saved); on round 4's fp16-filter kernel every survivor costs an exact
    # 64-channel distance plus an insertion and the hint WINS (1720 -> 1555 us, profiles/r04_knn_seed_ab.txt), so it is
    # taken for narrow inputs (<= 16 channels) and for exactly the shape that kernel takes (64 channels, k <= 20); anything
    # else would run the fp32-MFMA kernel, where it does not pay
    # (which kernel the call takes, and whether it honours a seed at all, is the LIBRARY's answer: pcops_knn_graph_path)
    path = int(_lib.load().pcops_knn_graph_path(b, n, c, k, _lib.ptr(x)))
    pays = bool(path & 16) and (KNN_SEED_FORCE or c <= 16 or (path & 15) == 3)
    if (seed is not None and KNN_SEED and c <= KNN_SEED_MAX_C and pays and tuple(seed.shape) == (b, n, k)
            and seed.dtype == torch.int32):
        _lib.call("pcops_knn_graph_seeded", b, n, c, k, _lib.ptr(x), _lib.ptr(seed.contiguous()), _lib.ptr(out))
    else:
        _lib.call("pcops_knn_graph", b, n, c, k, _lib.ptr(x), _lib.ptr(out))
    return out


class _EdgeFeature(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, nn_idx):
        b, n, c = x.shape
        k = nn_idx.shape[2]
        out = torch.empty((b, n, k, 2 * c), dtype=torch.float32, device=x.device)
        _lib.call("pcops_edge_feature", b, n, c, k, _lib.ptr(x), _lib.ptr(nn_idx), _lib.ptr(out))
        ctx.save_for_backward(nn_idx)
        ctx.shape = (b, n, c)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        (nn_idx,) = ctx.saved_tensors
        b, n, c = ctx.shape
        k = nn_idx.shape[2]
        grad_out = grad_out.contiguous()
        grad_x = torch.empty((b, n, c), dtype=torch.float32, device=grad_out.device)
        if _lib.deterministic():     # x_i half with plain stores, neighbour half by the ordered owner walk
            _lib.call("pcops_edge_feature_grad_central", b, n, c, k, _lib.ptr(grad_out), _lib.ptr(grad_x))
            _lib.scatter_rows_sorted(nn_idx.view(b, n * k), None, n, out=grad_x, c=c, ld=2 * c,
                                     src_ptr=grad_out.data_ptr() + 4 * c)
            return grad_x, None
        _lib.call("pcops_edge_feature_grad", b, n, c, k, _lib.ptr(grad_out), _lib.ptr(nn_idx),
                  _lib.ptr(grad_x))
        return grad_x, None


def get_edge_feature(point_cloud, nn_idx, k=20):
    """(B,N,C)|(B,N,1,C), nn_idx (B,N,k) -> (B,N,k,2C) = [x_i | x_j - x_i]"""
    x = _lib.check(_squeeze_cloud(point_cloud), torch.float32, "point_cloud", 3)
    nn_idx = _lib.check(nn_idx, torch.int32, "nn_idx", 3)
    if nn_idx.shape[:2] != x.shape[:2] or nn_idx.shape[2] != k:
        raise ValueError("nn_idx must be (B,N,k)")
    return _EdgeFeature.apply(x, nn_idx)


# ---------------------------------------------------------------------------- fused EdgeConv / conv stacks
class _Transform3(torch.autograd.Function):
    """point_cloud (B, N, 3) x transform (B, 3, 3) -- tf.matmul(point_cloud, transform) of dgcnn.py:37 -- on
    pcops_transform3_fwd / _bwd (one launch per direction; a library GEMM ran two Tensile kernels of 30-40 us for it)"""

    @staticmethod
    def forward(ctx, pc, T):
        pc, T = pc.contiguous(), T.contiguous()
        out = torch.empty_like(pc)
        _lib.call("pcops_transform3_fwd", pc.shape[0], pc.shape[1], pc.data_ptr(), T.data_ptr(), out.data_ptr())
        ctx.save_for_backward(pc, T)
        return out

    @staticmethod
    def backward(ctx, g):
        pc, T = ctx.saved_tensors
        g = g.contiguous()
        dT = torch.empty_like(T)
        dx = torch.empty_like(pc) if ctx.needs_input_grad[0] else None
        _lib.call("pcops_transform3_bwd", pc.shape[0], pc.shape[1], pc.data_ptr(), T.data_ptr(), g.data_ptr(), dT.data_ptr(),
                  None if dx is None else dx.data_ptr())
        return dx, dT


def apply_transform(point_cloud, transform):
    """tf.matmul(point_cloud, transform) for a (B, N, 3) cloud and a (B, 3, 3) transform"""
    if (point_cloud.is_cuda and point_cloud.dtype == torch.float32 and point_cloud.dim() == 3 and point_cloud.shape[-1] == 3
            and tuple(transform.shape) == (point_cloud.shape[0], 3, 3)):
        return _Transform3.apply(point_cloud, transform)
    return torch.matmul(point_cloud, transform)


def fused_ok(x, widths):
    return (_pn2.FUSED_MLP and x.is_cuda and x.dtype == torch.float32 and all(w % 32 == 0 for w in widths)
            and 256 % (widths[0] // 4) == 0 and (widths[0] >= 256 or 256 % widths[0] == 0) and widths[0] <= 1024)


def edge_conv_stack(point_cloud, nn_idx, widths, scopes, is_training, bn_decay, is_dist=False, cat_slot=None):
    """get_edge_feature + len(widths) x conv2d([1,1], bn, relu) + max over the k neighbours, without ever
    building the (B,N,k,2C) edge tensor.  The first conv is linear in the edge feature [x_i | x_j - x_i]:
        [x_i | x_j - x_i] W + b = x_i (W_a - W_b) + x_j W_b + b        (W = [W_a ; W_b], rows 0..C-1 / C..2C-1)
    so it is evaluated once per POINT (two small library GEMMs) and the (B,N,k,C') activation is the gather + add
    of csrc/gather.hip (`Q[b, nn_idx] + Ctr[b, i]`); the rest is the fused MLP stack.  Variables are exactly the
    ones `conv2d(..., scope=scopes[i], bn=True)` creates (dgcnn/models/dgcnn.py:39-48, transform_nets.py:18-27).
    point_cloud (B,N,C)|(B,N,1,C), nn_idx (B,N,k) -> (B,N,1,widths[-1])
    cat_slot = (fused_mlp.CatBuffer over (B,N,C_total), column): a single-layer stack on the one-GEMM path ALSO stores its
    output as that column block of the buffer and returns (out, block alias) -- the caller assembles the concatenation of
    several layers' outputs with fused_mlp.cat_assemble instead of torch.cat (no copy pass)."""
    x = _squeeze_cloud(point_cloud)
    b, n, c = x.shape
    names = ('pop_mean', 'pop_var') if is_dist else ('moving_mean', 'moving_variance')
    layers = _pn2._stack_variables(2 * c, widths, list(scopes), 1e-3, None, True, names)
    w1, b1 = layers[0][0], layers[0][1]
    x2d = x.reshape(b * n, c)
    decay = bn_decay if bn_decay is not None else 0.9
    k = nn_idx.shape[2]
    if b * n >= 8192 and widths[0] % 4 == 0 and fused_mlp.edge_qc_supported(b, n, k, widths[0]):
        # ONE per-point GEMM: X [W_b | W_a - W_b] + [0 | b1] = [Q | Ctr] (pcops.h "[Q | Ctr] forms"), one weight gradient,
        # one data gradient -- nothing for autograd to slice, pad or add up
        kp = c if c % 8 == 0 else (c + 7) // 8 * 8
        xp = x2d if kp == c else F.pad(x2d, (0, kp - c))
        if fused_mlp.edge_direct_supported(b, n, k, c, widths[0], len(widths), x):
            # the input needs no gradient (the T-Net on the raw cloud): [Q | Ctr] outside autograd, the layer's weight
            # gradient straight from the masked gradient of its output (no scatter to per-point gradients, no GEMM backward)
            with torch.no_grad():
                wcat, bcat = fused_mlp.edge_weights(w1, b1, kp)
                qc = fused_mlp.rows_linear(xp, wcat, bcat).view(b, n, 2 * widths[0])
            out = fused_mlp.gather_mlp_stack(nn_idx, True, is_training, decay, BN_EPS, False, layers, QC=qc,
                                             direct=(x.detach(), w1, b1))
            return (out.view(b, n, 1, widths[-1]), None) if cat_slot is not None else out.view(b, n, 1, widths[-1])
        wcat, bcat = fused_mlp.edge_weights(w1, b1, kp)
        qc = fused_mlp.rows_linear(xp, wcat, bcat).view(b, n, 2 * widths[0])
        if cat_slot is not None and len(widths) == 1:
            out, block = fused_mlp.gather_mlp_stack(nn_idx, True, is_training, decay, BN_EPS, False, layers, QC=qc,
                                                    cat_slot=cat_slot)
            return out.view(b, n, 1, widths[-1]), block
        out = fused_mlp.gather_mlp_stack(nn_idx, True, is_training, decay, BN_EPS, False, layers, QC=qc)
        return (out.view(b, n, 1, widths[-1]), None) if cat_slot is not None else out.view(b, n, 1, widths[-1])
    w_a, w_b = w1[:c], w1[c:]
    if c % 8 == 0 and b * n >= 8192 and widths[0] % 4 == 0:
        # B*N rows into a C x C' weight: the libpcops GEMMs (the library picks a few-CU kernel for these weight gradients)
        q = fused_mlp.rows_linear(x2d, w_b).view(b, n, widths[0])
        ctr = fused_mlp.rows_linear(x2d, w_a - w_b, b1).view(b, n, widths[0])
    elif b * n >= 8192 and widths[0] % 4 == 0:
        # 3 coordinate channels: zero-padded to 4 so that the same kernels take it (the library GEMM the K = 3 product
        # otherwise lands on needs 1.2 ms for these 0.2 GFLOP, four times per DGCNN step)
        padc = (-c) % 8
        xp = F.pad(x2d, (0, padc))
        q = fused_mlp.rows_linear(xp, F.pad(w_b, (0, 0, 0, padc))).view(b, n, widths[0])
        ctr = fused_mlp.rows_linear(xp, F.pad(w_a - w_b, (0, 0, 0, padc)), b1).view(b, n, widths[0])
    else:
        q = (x2d @ w_b).view(b, n, widths[0])                       # neighbour term, gathered by nn_idx
        ctr = torch.addmm(b1, x2d, w_a - w_b).view(b, n, widths[0])  # centre term + bias
    decay = bn_decay if bn_decay is not None else 0.9
    out = fused_mlp.gather_mlp_stack(nn_idx, True, is_training, decay, BN_EPS, False, layers, Q=q, Ctr=ctr)
    return (out.view(b, n, 1, widths[-1]), None) if cat_slot is not None else out.view(b, n, 1, widths[-1])


def cloud_point_ok(per_cloud, per_point, widths):
    b, n = per_point.shape[0], per_point.shape[1]
    return (fused_ok(per_point, widths) and b * n >= 8192 and n <= 16384 and per_point.shape[-1] % 4 == 0
            and fused_mlp.CLOUD_POINT)


def conv2d_stack_cloud_point(per_cloud, per_point, widths, scopes, is_training, bn_decay, is_dist=False):
    """len(widths) x conv2d([1,1], bn, relu) on concat([per_cloud broadcast over the points | per_point], -1) WITHOUT the
    concatenation.  The first conv is linear:  [c_b | x_bn] W + bias = x_bn W_x + (c_b W_c + bias)  -- a (B N, Cx) product per
    point and a (B, Cc) product per cloud, and Y1 = Q + Ctr is the gather form of csrc/gather.hip with the identity index (one
    group per cloud).  The reference builds the (B, N, 1, 1600) tensor of dgcnn_bga's segmentation head, 1280 of whose channels
    are per-cloud constants (dgcnn_bga.py:118-128): four fifths of its first conv's work, forward and backward, multiplied the
    same two vectors 2048 times.  Variables are the ones conv2d(..., scope=scopes[i]) creates on the concatenated width.
    per_cloud (B, Cc), per_point (B, N, 1, Cx) | (B, N, Cx) -> (B, N, 1, widths[-1])"""
    b, n = per_point.shape[0], per_point.shape[1]
    cc, cx = per_cloud.shape[-1], per_point.shape[-1]
    names = ('pop_mean', 'pop_var') if is_dist else ('moving_mean', 'moving_variance')
    layers = _pn2._stack_variables(cc + cx, widths, list(scopes), 1e-3, None, True, names)
    w1, b1 = layers[0][0], layers[0][1]
    w_c, w_x = fused_mlp.split_rows(w1, cc)                  # (one concatenation as their gradient)
    q = fused_mlp.rows_linear(per_point.reshape(b * n, cx), w_x).view(b, n, widths[0])
    if fused_mlp.TAIL_FOLD and per_cloud.is_cuda:           # the per-cloud rows on the small-GEMM kernel, not a library GEMM
        ctr = fused_mlp.small_linear(per_cloud.reshape(b, cc), w_c, b1).view(b, 1, widths[0])
        from ..pointnet2.pointnet_util import _whole_cloud_group
        _, idx = _whole_cloud_group(b, n, q.device)
    else:
        ctr = torch.addmm(b1, per_cloud.reshape(b, cc), w_c).view(b, 1, widths[0])
        idx = torch.arange(n, dtype=torch.int32, device=q.device).view(1, 1, n).expand(b, 1, n).contiguous()
    decay = bn_decay if bn_decay is not None else 0.9
    out = fused_mlp.gather_mlp_stack(idx, False, is_training, decay, BN_EPS, False, layers, Q=q, Ctr=ctr, identity_idx=True)
    return out.view(b, n, 1, widths[-1])


def conv2d_stack(inputs, widths, scopes, is_training, bn_decay, is_dist=False, pool_max=False):
    """len(widths) x conv2d([1,1], bn, relu) on a channel-last (B,H,W,C) tensor through the fused MLP stack
    (this module's BN flavour: biased variance in the moving statistics); pool_max: + max over axis 2."""
    names = ('pop_mean', 'pop_var') if is_dist else ('moving_mean', 'moving_variance')
    return _pn2.conv2d_stack(inputs, widths, list(scopes), is_training, bn_decay, pool_max=pool_max,
                             unbiased_moving_var=False, mov_names=names)


class _FirstMax(torch.autograd.Function):
    """amax over one dimension whose gradient goes to the FIRST maximal member (the contract of every pooling
    kernel here; torch's own amax backward splits the gradient evenly over exact fp32 ties)"""

    @staticmethod
    def forward(ctx, x, dim):
        out = x.amax(dim=dim, keepdim=True)              # (amax: the decision reader of the parity tests hooks it, tests/decisions.py)
        ctx.save_for_backward(x.argmax(dim=dim, keepdim=True))
        ctx.dim, ctx.shape = dim, x.shape
        return out

    @staticmethod
    def backward(ctx, g):
        (arg,) = ctx.saved_tensors
        return torch.zeros(ctx.shape, dtype=g.dtype, device=g.device).scatter_(ctx.dim, arg, g), None


def conv2d_stack_global_max(inputs, widths, scopes, is_training, bn_decay, is_dist=False):
    """conv stack on (B,N,1,C) followed by the max over ALL points (the reference's `agg` conv + max_pool2d over
    [num_point,1], dgcnn.py:79-84): the max is associative, so it is taken inside the fused stack over chunks of 256
    points (8-bit arg-max) and finished over the N/256 chunk maxima -- the (B,N,1,C') activation and its dense
    gradient never exist.  Returns (B,1,1,widths[-1])."""
    b, n, _, c = inputs.shape
    chunk = 256 if n % 256 == 0 else (n if n <= 256 else 0)
    if chunk == 0:
        return conv2d_stack(inputs, widths, scopes, is_training, bn_decay, is_dist).amax(dim=1, keepdim=True)
    part = conv2d_stack(inputs.reshape(b, n // chunk, chunk, c), widths, scopes, is_training, bn_decay, is_dist,
                        pool_max=True)                               # (B, N/chunk, 1, C')
    return _FirstMax.apply(part, 1)
