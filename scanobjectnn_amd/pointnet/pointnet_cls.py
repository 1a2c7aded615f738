"""Vanilla PointNet classifier -- mirror of `pointnet/models/pointnet_cls.py` (placeholder_inputs :13-18,
get_model :21-75, get_loss :78-93).  BASELINE config 1 ("plumbing"): MLP + max only, no custom kernel, so it
also runs on the host CPU through the plain `tf_util` layers."""
import torch
import torch.nn.functional as F

from ..graph import variable_scope
from ..dgcnn import tf_util      # the `tf.nn.moments` + EMA batch-norm flavour of `pointnet/utils/tf_util.py:455-490` IS the one
#                                   `dgcnn/utils/tf_util.py:462-499` copies (biased variance in the moving average too): one layer module
from .transform_nets import feature_transform_net, input_transform_net

NUM_CLASSES = 15


def placeholder_inputs(batch_size, num_point, device=None):
    pointclouds_pl = torch.zeros((batch_size, num_point, 3), dtype=torch.float32, device=device)
    labels_pl = torch.zeros((batch_size,), dtype=torch.int32, device=device)
    return pointclouds_pl, labels_pl


def get_model(point_cloud, is_training, bn_decay=None, num_class=NUM_CLASSES):
    """point_cloud (B,N,3) -> logits (B,num_class), end_points {'transform': (B,64,64)}"""
    batch_size, num_point = point_cloud.shape[0], point_cloud.shape[1]
    end_points = {}
    with variable_scope('transform_net1'):
        transform = input_transform_net(point_cloud, is_training, bn_decay, K=3)
    input_image = torch.matmul(point_cloud, transform).unsqueeze(-1)            # (B,N,3,1)
    net = tf_util.conv2d(input_image, 64, [1, 3], padding='VALID', stride=[1, 1], bn=True,
                         is_training=is_training, scope='conv1', bn_decay=bn_decay)
    net = tf_util.conv2d(net, 64, [1, 1], padding='VALID', stride=[1, 1], bn=True,
                         is_training=is_training, scope='conv2', bn_decay=bn_decay)
    with variable_scope('transform_net2'):
        transform = feature_transform_net(net, is_training, bn_decay, K=64)
    end_points['transform'] = transform
    net = torch.matmul(net.squeeze(2), transform).unsqueeze(2)
    for i, width in zip((3, 4, 5), (64, 128, 1024)):
        net = tf_util.conv2d(net, width, [1, 1], padding='VALID', stride=[1, 1], bn=True,
                             is_training=is_training, scope='conv%d' % i, bn_decay=bn_decay)
    net = tf_util.max_pool2d(net, [num_point, 1], padding='VALID', scope='maxpool')
    net = net.reshape(batch_size, -1)
    net = tf_util.fully_connected(net, 512, bn=True, is_training=is_training, scope='fc1', bn_decay=bn_decay)
    net = tf_util.dropout(net, keep_prob=0.7, is_training=is_training, scope='dp1')
    net = tf_util.fully_connected(net, 256, bn=True, is_training=is_training, scope='fc2', bn_decay=bn_decay)
    net = tf_util.dropout(net, keep_prob=0.7, is_training=is_training, scope='dp2')
    net = tf_util.fully_connected(net, num_class, activation_fn=None, scope='fc3')
    return net, end_points


def get_loss(pred, label, end_points, reg_weight=0.001):
    """CE + reg_weight * l2_loss(T T^T - I)  (tf.nn.l2_loss = sum(x^2)/2)"""
    classify_loss = F.cross_entropy(pred, label.long())
    transform = end_points['transform']
    k = transform.shape[1]
    mat_diff = torch.matmul(transform, transform.transpose(1, 2)) - torch.eye(k, device=transform.device)
    return classify_loss + 0.5 * (mat_diff * mat_diff).sum() * reg_weight
