"""DGCNN BGA (classification + background mask head) -- mirror of `dgcnn/models/dgcnn_bga.py`
(get_model :27-134, get_loss :137-153).  The reference file raises NameError at import (NUM_CLASSES
commented out, :13-16,27); 15 classes as everywhere else in the repo."""
import torch
import torch.nn.functional as F

from .. import fused_mlp
from . import tf_util
from .dgcnn import backbone

NUM_CLASSES = 15


def placeholder_inputs(batch_size, num_point, device=None):
    pointclouds_pl = torch.zeros((batch_size, num_point, 3), dtype=torch.float32, device=device)
    labels_pl = torch.zeros((batch_size,), dtype=torch.int32, device=device)
    mask_pl = torch.zeros((batch_size, num_point), dtype=torch.int32, device=device)
    return pointclouds_pl, labels_pl, mask_pl


def get_model(point_cloud, is_training, bn_decay=None, num_class=NUM_CLASSES):
    """point_cloud (B,N,3) -> class_pred (B,num_class), seg_pred (B,N,2)"""
    batch_size, num_point = point_cloud.shape[0], point_cloud.shape[1]
    net1, net2, net3, net4, local, out_max = backbone(point_cloud, is_training, bn_decay)  # out_max (B,1,1,1024)
    expand = out_max.expand(batch_size, num_point, 1, 1024)

    net = out_max.reshape(batch_size, -1)
    net = tf_util.fully_connected(net, 512, bn=True, is_training=is_training, scope='fc1', bn_decay=bn_decay)
    net = tf_util.dropout(net, keep_prob=0.5, is_training=is_training, scope='dp1')
    net = tf_util.fully_connected(net, 256, bn=True, is_training=is_training, scope='fc2', bn_decay=bn_decay)
    class_vector = net.view(batch_size, 1, 1, 256)
    net = tf_util.dropout(net, keep_prob=0.5, is_training=is_training, scope='dp2')
    class_pred = tf_util.fully_connected(net, num_class, activation_fn=None, scope='fc3')

    per_cloud = torch.cat([class_vector.view(batch_size, 256), out_max.view(batch_size, 1024)], dim=-1)
    if tf_util.cloud_point_ok(per_cloud, local, [512, 256]):
        # the first 1280 of the reference's 1600 concatenated channels are per-cloud constants: their part of seg/conv1 is
        # ONE (B, 1280) product, not 2048 of them per cloud (tf_util.conv2d_stack_cloud_point); decay as below
        net = tf_util.conv2d_stack_cloud_point(per_cloud, local, [512, 256], ['seg/conv1', 'seg/conv2'], is_training, None,
                                               is_dist=True)
        net = tf_util.dropout(net, keep_prob=0.7, is_training=is_training, scope='dp1')
        net = tf_util.conv2d(net, 2, [1, 1], padding='VALID', stride=[1, 1], activation_fn=None,
                             scope='seg/conv3', is_dist=True)
        return class_pred, net.squeeze(2)
    concat = torch.cat([class_vector.expand(batch_size, num_point, 1, 256), expand, net1, net2, net3, net4],
                       dim=-1)                                                             # 1600 ch
    if tf_util.fused_ok(concat, [512, 256]):
        # note: the reference passes no bn_decay here (dgcnn_bga.py:125-128) -> decay 0.9
        net = tf_util.conv2d_stack(concat, [512, 256], ['seg/conv1', 'seg/conv2'], is_training, None, is_dist=True)
    else:
        net = tf_util.conv2d(concat, 512, [1, 1], padding='VALID', stride=[1, 1], bn=True,
                             is_training=is_training, scope='seg/conv1', is_dist=True)
        net = tf_util.conv2d(net, 256, [1, 1], padding='VALID', stride=[1, 1], bn=True,
                             is_training=is_training, scope='seg/conv2', is_dist=True)
    net = tf_util.dropout(net, keep_prob=0.7, is_training=is_training, scope='dp1')
    net = tf_util.conv2d(net, 2, [1, 1], padding='VALID', stride=[1, 1], activation_fn=None,
                         scope='seg/conv3', is_dist=True)
    return class_pred, net.squeeze(2)


def get_loss(class_pred, seg_pred, gt_label, gt_mask, seg_weight=0.5):
    classify_loss = fused_mlp.softmax_cross_entropy(class_pred, gt_label)
    b, n, c = seg_pred.shape
    if fused_mlp.TAIL_FOLD and seg_pred.is_cuda:        # every cloud has n points: the mean of the clouds' means = the mean of all rows
        seg_loss = fused_mlp.softmax_cross_entropy(seg_pred.reshape(b * n, c), gt_mask.reshape(b * n))
    else:
        per_point = F.cross_entropy(seg_pred.reshape(b * n, c), gt_mask.reshape(b * n).long(),
                                    reduction='none').view(b, n)
        seg_loss = per_point.mean(dim=1).mean()
    return (1 - seg_weight) * classify_loss + seg_weight * seg_loss, classify_loss, seg_loss
