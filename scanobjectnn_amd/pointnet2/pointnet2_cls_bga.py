"""PointNet++ BGA (classification + background mask head) -- mirror of
`pointnet2/models/pointnet2_cls_bga.py` (placeholder_inputs :14-18, get_model :21-75,
get_loss :78-93).  BASELINE config 4."""
import torch
import torch.nn.functional as F

from .. import fused_mlp
from . import tf_util
from .pointnet_util import pointnet_fp_module, pointnet_sa_module

NUM_CLASSES = 15
BACKGROUND_CLASS = -1


def placeholder_inputs(batch_size, num_point, device=None):
    pointclouds_pl = torch.zeros((batch_size, num_point, 3), dtype=torch.float32, device=device)
    labels_pl = torch.zeros((batch_size,), dtype=torch.int32, device=device)
    mask_pl = torch.zeros((batch_size, num_point), dtype=torch.int32, device=device)
    return pointclouds_pl, labels_pl, mask_pl


def get_model(point_cloud, is_training, bn_decay=None, num_class=NUM_CLASSES):
    """point_cloud (B,N,>=3) -> class_pred (B,num_class), seg_pred (B,N,2)"""
    batch_size = point_cloud.shape[0]
    l0_xyz = point_cloud[:, :, :3].contiguous()
    l0_points = None

    l1_xyz, l1_points, _ = pointnet_sa_module(l0_xyz, l0_points, npoint=512, radius=0.2, nsample=64,
                                              mlp=[64, 64, 128], mlp2=None, group_all=False,
                                              is_training=is_training, bn_decay=bn_decay, scope='layer1')
    l2_xyz, l2_points, _ = pointnet_sa_module(l1_xyz, l1_points, npoint=128, radius=0.4, nsample=64,
                                              mlp=[128, 128, 256], mlp2=None, group_all=False,
                                              is_training=is_training, bn_decay=bn_decay, scope='layer2')
    l3_xyz, l3_points, _ = pointnet_sa_module(l2_xyz, l2_points, npoint=None, radius=None, nsample=None,
                                              mlp=[256, 512, 1024], mlp2=None, group_all=True,
                                              is_training=is_training, bn_decay=bn_decay, scope='layer3')

    # classification branch
    net = l3_points.reshape(batch_size, -1)
    net = tf_util.fully_connected(net, 512, bn=True, is_training=is_training, scope='fc1', bn_decay=bn_decay)
    net = tf_util.dropout(net, keep_prob=0.5, is_training=is_training, scope='dp1')
    net = tf_util.fully_connected(net, 256, bn=True, is_training=is_training, scope='fc2', bn_decay=bn_decay)
    class_vector = net.unsqueeze(1)                                   # (B,1,256)
    net = tf_util.dropout(net, keep_prob=0.5, is_training=is_training, scope='dp2')
    class_pred = tf_util.fully_connected(net, num_class, activation_fn=None, scope='fc3')

    # segmentation branch: the class vector is propagated from the single l3 point (m=1 -> w=(1,0,0))
    l2_points = pointnet_fp_module(l2_xyz, l3_xyz, l2_points, class_vector, [256, 256], is_training,
                                   bn_decay, scope='fa_layer1')
    l1_points = pointnet_fp_module(l1_xyz, l2_xyz, l1_points, l2_points, [256, 128], is_training,
                                   bn_decay, scope='fa_layer2')
    l0_points = pointnet_fp_module(l0_xyz, l1_xyz, l0_points, l1_points, [128, 128, 128], is_training,
                                   bn_decay, scope='fa_layer3')

    net = tf_util.conv1d(l0_points, 128, 1, padding='VALID', bn=True, is_training=is_training,
                         scope='seg_fc1', bn_decay=bn_decay)
    net = tf_util.dropout(net, keep_prob=0.5, is_training=is_training, scope='seg_dp1')
    seg_pred = tf_util.conv1d(net, 2, 1, padding='VALID', activation_fn=None, scope='seg_fc2')
    return class_pred, seg_pred


def get_loss(class_pred, seg_pred, gt_label, gt_mask, seg_weight=0.5):
    """total = (1-w)*CE_cls + w*mean_b(mean_n CE_seg); returns (total, classify, seg)"""
    classify_loss = fused_mlp.softmax_cross_entropy(class_pred, gt_label)
    b, n, c = seg_pred.shape
    if fused_mlp.TAIL_FOLD and seg_pred.is_cuda:        # every cloud has n points: the mean of the clouds' means = the mean of all rows
        seg_loss = fused_mlp.softmax_cross_entropy(seg_pred.reshape(b * n, c), gt_mask.reshape(b * n))
    else:
        per_point = F.cross_entropy(seg_pred.reshape(b * n, c), gt_mask.reshape(b * n).long(),
                                    reduction='none').view(b, n)
        seg_loss = per_point.mean(dim=1).mean()
    total_loss = (1 - seg_weight) * classify_loss + seg_weight * seg_loss
    return total_loss, classify_loss, seg_loss
