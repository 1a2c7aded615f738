"""DGCNN classifier -- mirror of `dgcnn/models/dgcnn.py` (placeholder_inputs :17-21,
get_model :24-102, get_loss :105-111).  BASELINE config 3.  The five graph rebuilds use the fused
`knn_graph` (same indices as pairwise_distance+knn, no (B,N,N) tensor)."""
import torch

from .. import fused_mlp
from . import tf_util
from ..graph import variable_scope
from .transform_nets import input_transform_net

NUM_CLASSES = 15


def placeholder_inputs(batch_size, num_point, device=None):
    pointclouds_pl = torch.zeros((batch_size, num_point, 3), dtype=torch.float32, device=device)
    labels_pl = torch.zeros((batch_size,), dtype=torch.int32, device=device)
    return pointclouds_pl, labels_pl


def _edge_conv(x, width, scope, k, is_training, bn_decay, seed=None, cat_slot=None):
    """-> (features (B,N,1,width), this layer's neighbour graph[, the alias of the layer's block in cat_slot's buffer]).  seed: the previous layer's graph -- the reference
    rebuilds the kNN graph on every layer's features (dgcnn.py:31-71); consecutive graphs of the same points mostly
    agree, which the kernel uses as a starting threshold (same result, tf_util.knn_graph)."""
    nn_idx = tf_util.knn_graph(x, k=k, seed=seed)
    if tf_util.fused_ok(x, [width]):
        # EdgeConv without the (B,N,k,2C) edge tensor: first conv per point, gather + add, fused BN/ReLU/max
        if cat_slot is not None:
            net, block = tf_util.edge_conv_stack(x, nn_idx, [width], [scope], is_training, bn_decay, cat_slot=cat_slot)
            return net, nn_idx, block
        return tf_util.edge_conv_stack(x, nn_idx, [width], [scope], is_training, bn_decay), nn_idx
    edge_feature = tf_util.get_edge_feature(x, nn_idx=nn_idx, k=k)
    net = tf_util.conv2d(edge_feature, width, [1, 1], padding='VALID', stride=[1, 1], bn=True,
                         is_training=is_training, scope=scope, bn_decay=bn_decay)
    if cat_slot is not None:
        return net.amax(dim=-2, keepdim=True), nn_idx, None
    return net.amax(dim=-2, keepdim=True), nn_idx               # (B,N,1,width)


def backbone(point_cloud, is_training, bn_decay, k=20):
    """shared by dgcnn / dgcnn_bga: returns (net1..net4, their concatenation, out_max) with out_max (B,1,1,1024) = the max over the points
    of the 1024-wide `agg` layer (both models only ever use that max, dgcnn.py:79-84 / dgcnn_bga.py)"""
    nn_idx = tf_util.knn_graph(point_cloud, k=k)
    with variable_scope('transform_net1'):
        if tf_util.fused_ok(point_cloud, [64, 128]):
            transform = input_transform_net(None, is_training, bn_decay, K=3, point_cloud=point_cloud, nn_idx=nn_idx)
        else:
            edge_feature = tf_util.get_edge_feature(point_cloud, nn_idx=nn_idx, k=k)
            transform = input_transform_net(edge_feature, is_training, bn_decay, K=3)
    point_cloud_transformed = tf_util.apply_transform(point_cloud, transform)
    cat = None
    if point_cloud.is_cuda and tf_util.fused_ok(point_cloud_transformed, [64]):
        # the four EdgeConv layers store their outputs straight into their column blocks of the (B, N, 1, 320) tensor the
        # reference builds with tf.concat (dgcnn.py:83): no concatenation pass (round 5)
        from .. import fused_mlp
        b, n = point_cloud.shape[0], point_cloud.shape[1]
        buf = fused_mlp.CatBuffer((b, n, 1, 320), point_cloud.device)
        net1, g1, s1 = _edge_conv(point_cloud_transformed, 64, 'dgcnn1', k, is_training, bn_decay, seed=nn_idx, cat_slot=(buf, 0))
        net2, g2, s2 = _edge_conv(net1, 64, 'dgcnn2', k, is_training, bn_decay, seed=g1, cat_slot=(buf, 64))
        net3, g3, s3 = _edge_conv(net2, 64, 'dgcnn3', k, is_training, bn_decay, seed=g2, cat_slot=(buf, 128))
        net4, _, s4 = _edge_conv(net3, 128, 'dgcnn4', k, is_training, bn_decay, seed=g3, cat_slot=(buf, 192))
        if all(s is not None for s in (s1, s2, s3, s4)):
            cat = fused_mlp.cat_assemble(buf, [s1, s2, s3, s4])
    else:
        net1, g1 = _edge_conv(point_cloud_transformed, 64, 'dgcnn1', k, is_training, bn_decay, seed=nn_idx)
        net2, g2 = _edge_conv(net1, 64, 'dgcnn2', k, is_training, bn_decay, seed=g1)
        net3, g3 = _edge_conv(net2, 64, 'dgcnn3', k, is_training, bn_decay, seed=g2)
        net4, _ = _edge_conv(net3, 128, 'dgcnn4', k, is_training, bn_decay, seed=g3)
    if cat is None:
        cat = torch.cat([net1, net2, net3, net4], dim=-1)
    if tf_util.fused_ok(cat, [1024]):
        out_max = tf_util.conv2d_stack_global_max(cat, [1024], ['agg'], is_training, bn_decay)
    else:
        agg = tf_util.conv2d(cat, 1024, [1, 1], padding='VALID', stride=[1, 1], bn=True, is_training=is_training,
                             scope='agg', bn_decay=bn_decay)
        out_max = tf_util.max_pool2d(agg, [agg.shape[1], 1], padding='VALID', scope='maxpool')
    return net1, net2, net3, net4, cat, out_max    # cat: the (B,N,1,320) concatenation (dgcnn_bga's segmentation head reuses it)


def get_model(point_cloud, is_training, bn_decay=None, num_class=NUM_CLASSES):
    """point_cloud (B,N,3) -> logits (B,num_class), end_points"""
    batch_size = point_cloud.shape[0]
    end_points = {}
    *_, out_max = backbone(point_cloud, is_training, bn_decay)
    net = out_max.reshape(batch_size, -1)
    net = tf_util.fully_connected(net, 512, bn=True, is_training=is_training, scope='fc1', bn_decay=bn_decay)
    net = tf_util.dropout(net, keep_prob=0.5, is_training=is_training, scope='dp1')
    net = tf_util.fully_connected(net, 256, bn=True, is_training=is_training, scope='fc2', bn_decay=bn_decay)
    net = tf_util.dropout(net, keep_prob=0.5, is_training=is_training, scope='dp2')
    net = tf_util.fully_connected(net, num_class, activation_fn=None, scope='fc3')
    return net, end_points


def get_loss(pred, label, end_points=None, num_class=NUM_CLASSES):
    """softmax CE with label_smoothing 0.2 (tf.losses.softmax_cross_entropy: onehot*(1-s) + s/C)"""
    return fused_mlp.softmax_cross_entropy(pred, label, label_smoothing=0.2)
