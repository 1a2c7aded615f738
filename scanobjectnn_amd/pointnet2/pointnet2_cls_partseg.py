"""PointNet++ part segmentation (6 part classes) -- mirror of `pointnet2/models/pointnet2_cls_partseg.py`
(placeholder_inputs :13-17, get_model :20-45, get_loss :54-87).  Same set-abstraction + feature-propagation
kernels as the BGA model; SURVEY §8f-3.  Like the reference, get_model returns seg_pred only."""
import torch
import torch.nn.functional as F

from . import tf_util
from .pointnet_util import pointnet_fp_module, pointnet_sa_module

NUM_CLASSES = 6


def placeholder_inputs(batch_size, num_point, device=None):
    pointclouds_pl = torch.zeros((batch_size, num_point, 3), dtype=torch.float32, device=device)
    labels_pl = torch.zeros((batch_size,), dtype=torch.int32, device=device)
    mask_pl = torch.zeros((batch_size, num_point), dtype=torch.int32, device=device)
    return pointclouds_pl, labels_pl, mask_pl


def get_model(point_cloud, is_training, bn_decay=None, num_class=NUM_CLASSES):
    """point_cloud (B,N,>=3) -> seg_pred (B,N,num_class)"""
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

    # feature propagation: the global feature of the single l3 point goes to every l2 point (m=1 -> w=(1,0,0))
    l2_points = pointnet_fp_module(l2_xyz, l3_xyz, l2_points, l3_points, [256, 256], is_training,
                                   bn_decay, scope='fa_layer1')
    l1_points = pointnet_fp_module(l1_xyz, l2_xyz, l1_points, l2_points, [256, 128], is_training,
                                   bn_decay, scope='fa_layer2')
    l0_points = pointnet_fp_module(l0_xyz, l1_xyz, l0_points, l1_points, [128, 128, 128], is_training,
                                   bn_decay, scope='fa_layer3')

    net = tf_util.conv1d(l0_points, 128, 1, padding='VALID', bn=True, is_training=is_training,
                         scope='seg_fc1', bn_decay=bn_decay)
    net = tf_util.dropout(net, keep_prob=0.5, is_training=is_training, scope='seg_dp1')
    seg_pred = tf_util.conv1d(net, num_class, 1, padding='VALID', activation_fn=None, scope='seg_fc2')
    return seg_pred


def get_loss(seg_pred, gt_seg):
    """mean over clouds of the mean per-point cross entropy (:84-87)"""
    b, n, c = seg_pred.shape
    per_point = F.cross_entropy(seg_pred.reshape(b * n, c), gt_seg.reshape(b * n).long(),
                                reduction='none').view(b, n)
    return per_point.mean(dim=1).mean()
