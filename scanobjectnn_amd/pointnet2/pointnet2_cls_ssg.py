"""PointNet++ SSG classifier -- mirror of `pointnet2/models/pointnet2_cls_ssg.py`
(placeholder_inputs :18-21, get_model :23-47, get_loss :50-57).  BASELINE config 2."""
import torch

from .. import fused_mlp
from . import tf_util
from .pointnet_util import pointnet_sa_module

NUM_CLASSES = 15


def placeholder_inputs(batch_size, num_point, device=None):
    pointclouds_pl = torch.zeros((batch_size, num_point, 3), dtype=torch.float32, device=device)
    labels_pl = torch.zeros((batch_size,), dtype=torch.int32, device=device)
    return pointclouds_pl, labels_pl


def get_model(point_cloud, is_training, bn_decay=None, num_class=NUM_CLASSES):
    """point_cloud (B,N,3) -> logits (B,num_class), end_points"""
    batch_size = point_cloud.shape[0]
    end_points = {'l0_xyz': point_cloud}
    l0_xyz, l0_points = point_cloud, None

    l1_xyz, l1_points, _ = pointnet_sa_module(l0_xyz, l0_points, npoint=512, radius=0.2, nsample=32,
                                              mlp=[64, 64, 128], mlp2=None, group_all=False,
                                              is_training=is_training, bn_decay=bn_decay,
                                              scope='layer1', use_nchw=True)
    l2_xyz, l2_points, _ = pointnet_sa_module(l1_xyz, l1_points, npoint=128, radius=0.4, nsample=64,
                                              mlp=[128, 128, 256], mlp2=None, group_all=False,
                                              is_training=is_training, bn_decay=bn_decay,
                                              scope='layer2')
    _, l3_points, _ = pointnet_sa_module(l2_xyz, l2_points, npoint=None, radius=None, nsample=None,
                                         mlp=[256, 512, 1024], mlp2=None, group_all=True,
                                         is_training=is_training, bn_decay=bn_decay, scope='layer3')

    net = l3_points.reshape(batch_size, -1)
    net = tf_util.fully_connected(net, 512, bn=True, is_training=is_training, scope='fc1', bn_decay=bn_decay)
    net = tf_util.dropout(net, keep_prob=0.5, is_training=is_training, scope='dp1')
    net = tf_util.fully_connected(net, 256, bn=True, is_training=is_training, scope='fc2', bn_decay=bn_decay)
    net = tf_util.dropout(net, keep_prob=0.5, is_training=is_training, scope='dp2')
    net = tf_util.fully_connected(net, num_class, activation_fn=None, scope='fc3')
    return net, end_points


def get_loss(pred, label, end_points=None):
    """mean sparse softmax cross-entropy; pred (B,C), label (B,)"""
    return fused_mlp.softmax_cross_entropy(pred, label)
