"""PointNet++ MSG classifier for BASELINE config 5 (multi-radius ball-query stress).

The reference ships the layer (`pointnet_sa_module_msg`, utils/pointnet_util.py:156-196) but NO
model file for it.  The hyper-parameters below are upstream PointNet++
`pointnet2_cls_msg` values -- external knowledge, not from the reference -- with the reference's
15-class head."""
from .. import fused_mlp
from . import tf_util
from .pointnet_util import pointnet_sa_module, pointnet_sa_module_msg
from .pointnet2_cls_ssg import placeholder_inputs  # noqa: F401  (same inputs)

NUM_CLASSES = 15


def get_model(point_cloud, is_training, bn_decay=None, num_class=NUM_CLASSES):
    batch_size = point_cloud.shape[0]
    end_points = {}
    l0_xyz, l0_points = point_cloud, None
    l1_xyz, l1_points = pointnet_sa_module_msg(l0_xyz, l0_points, 512, [0.1, 0.2, 0.4], [16, 32, 128],
                                               [[32, 32, 64], [64, 64, 128], [64, 96, 128]],
                                               is_training, bn_decay, scope='layer1')
    l2_xyz, l2_points = pointnet_sa_module_msg(l1_xyz, l1_points, 128, [0.2, 0.4, 0.8], [32, 64, 128],
                                               [[64, 64, 128], [128, 128, 256], [128, 128, 256]],
                                               is_training, bn_decay, scope='layer2')
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
    return fused_mlp.softmax_cross_entropy(pred, label)
