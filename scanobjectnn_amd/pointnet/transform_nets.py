"""PointNet T-Nets -- mirror of `pointnet/models/transform_nets.py` (input_transform_net :10-50,
feature_transform_net :53-95).  Pure dense layers (no custom kernel), runs on any device."""
import torch

from ..graph import constant_initializer, get_variable, variable_scope
from ..dgcnn import tf_util      # the `tf.nn.moments` + EMA batch-norm flavour of `pointnet/utils/tf_util.py:455-490` IS the one
#                                   `dgcnn/utils/tf_util.py:462-499` copies (biased variance in the moving average too): one layer module


def _trunk(net, num_point, is_training, bn_decay, first_kernel):
    net = tf_util.conv2d(net, 64, first_kernel, padding='VALID', stride=[1, 1], bn=True,
                         is_training=is_training, scope='tconv1', bn_decay=bn_decay)
    net = tf_util.conv2d(net, 128, [1, 1], padding='VALID', stride=[1, 1], bn=True,
                         is_training=is_training, scope='tconv2', bn_decay=bn_decay)
    net = tf_util.conv2d(net, 1024, [1, 1], padding='VALID', stride=[1, 1], bn=True,
                         is_training=is_training, scope='tconv3', bn_decay=bn_decay)
    net = tf_util.max_pool2d(net, [num_point, 1], padding='VALID', scope='tmaxpool')
    net = net.reshape(net.shape[0], -1)
    net = tf_util.fully_connected(net, 512, bn=True, is_training=is_training, scope='tfc1', bn_decay=bn_decay)
    return tf_util.fully_connected(net, 256, bn=True, is_training=is_training, scope='tfc2', bn_decay=bn_decay)


def _head(net, scope, k_out):
    with variable_scope(scope):
        weights = get_variable('weights', [256, k_out * k_out], constant_initializer(0.0))
        biases = get_variable('biases', [k_out * k_out], constant_initializer(0.0))
        eye = torch.eye(k_out, dtype=torch.float32, device=net.device).flatten()
        return torch.addmm(biases + eye, net, weights).view(net.shape[0], k_out, k_out)


def input_transform_net(point_cloud, is_training, bn_decay=None, K=3):
    """point_cloud (B,N,3) -> (B,3,3)"""
    assert K == 3
    b, n, _ = point_cloud.shape
    net = _trunk(point_cloud.unsqueeze(-1), n, is_training, bn_decay, [1, 3])   # (B,N,3,1), [1,3] conv
    return _head(net, 'transform_XYZ', 3)


def feature_transform_net(inputs, is_training, bn_decay=None, K=64):
    """inputs (B,N,1,K) -> (B,K,K)"""
    net = _trunk(inputs, inputs.shape[1], is_training, bn_decay, [1, 1])
    return _head(net, 'transform_feat', K)
