"""SpiderCNN classifier on xyz -- mirror of `SpiderCNN/models/spidercnn_cls_xyz.py` (get_model :20-71, get_loss
:74-82).  Its front end is the kNN branch of the op library: `knn_point` (pairwise squared distances +
the literal selection sort) on the cloud against itself, k = 20, then `group_point`."""
import torch
import torch.nn.functional as F

from ..graph import variable_scope
from ..pointnet2.tf_grouping import group_point, knn_point
from . import tf_util

NUM_CLASSES = 15


def front_end(xyz, nsample=20):
    """xyz (B,N,3) -> idx (B,N,nsample) int32 (self first), delta (B,N,nsample,3) = neighbour - centre (:28-35)"""
    _, idx = knn_point(nsample, xyz, xyz)
    grouped_xyz = group_point(xyz, idx)
    return idx, grouped_xyz - xyz.unsqueeze(2)


def get_model(xyz, is_training, bn_decay=None, num_class=NUM_CLASSES):
    """xyz (B,N,3) -> logits (B,num_class)"""
    xyz = xyz[:, :, :3].contiguous()
    batch_size = xyz.shape[0]
    nsample, G, taylor_channel = 20, 16, 5
    with variable_scope('delta'):
        idx, delta = front_end(xyz, nsample)
    feats, feat = [], xyz
    for i, width in enumerate((32, 64, 128, 256)):
        with variable_scope('fanConv%d' % (i + 1)):
            feat = tf_util.spiderConv(feat, idx, delta, width, taylor_channel=taylor_channel, gn=True, G=G)
        feats.append(feat)
    feat = torch.cat(feats, dim=2)
    net = tf_util.topk_pool(feat, k=2, scope='topk_pool').reshape(batch_size, -1)
    net = tf_util.fully_connected(net, 1024, bn=True, is_training=is_training, scope='fc1', bn_decay=bn_decay)
    net = tf_util.dropout(net, keep_prob=0.3, is_training=is_training, scope='dp1')
    net = tf_util.fully_connected(net, 512, bn=True, is_training=is_training, scope='fc2', bn_decay=bn_decay)
    net = tf_util.dropout(net, keep_prob=0.3, is_training=is_training, scope='dp2')
    return tf_util.fully_connected(net, num_class, activation_fn=None, scope='fc3')


def get_loss(pred, label):
    return F.cross_entropy(pred, label.long())
