"""Input T-Net on edge features -- mirror of `dgcnn/models/transform_nets.py:10-55`."""
import torch

from . import tf_util
from ..pointnet2.tf_util import _dense
from ..graph import constant_initializer, get_variable, variable_scope


def input_transform_net(edge_feature, is_training, bn_decay=None, K=3, is_dist=False, point_cloud=None,
                        nn_idx=None):
    """edge_feature (B,N,k,2C) -> transform (B,K,K).  Fast path: pass `point_cloud` (B,N,C) + `nn_idx` (B,N,k)
    instead of the edge tensor (edge_feature=None) and tconv1/tconv2/max run as one fused gather stack."""
    if edge_feature is None:
        batch_size, num_point = point_cloud.shape[0], point_cloud.shape[1]
        net = tf_util.edge_conv_stack(point_cloud, nn_idx, [64, 128], ['tconv1', 'tconv2'], is_training, bn_decay,
                                      is_dist=is_dist)                                  # (B,N,1,128)
        # tconv3 + the max over all points (tmaxpool) inside the fused stack
        net = tf_util.conv2d_stack_global_max(net, [1024], ['tconv3'], is_training, bn_decay, is_dist=is_dist)
        num_point = 1
    else:
        batch_size, num_point = edge_feature.shape[0], edge_feature.shape[1]
        net = tf_util.conv2d(edge_feature, 64, [1, 1], padding='VALID', stride=[1, 1], bn=True,
                             is_training=is_training, scope='tconv1', bn_decay=bn_decay, is_dist=is_dist)
        net = tf_util.conv2d(net, 128, [1, 1], padding='VALID', stride=[1, 1], bn=True,
                             is_training=is_training, scope='tconv2', bn_decay=bn_decay, is_dist=is_dist)
        net = net.amax(dim=-2, keepdim=True)
        net = tf_util.conv2d(net, 1024, [1, 1], padding='VALID', stride=[1, 1], bn=True,
                             is_training=is_training, scope='tconv3', bn_decay=bn_decay, is_dist=is_dist)
    net = tf_util.max_pool2d(net, [num_point, 1], padding='VALID', scope='tmaxpool')
    net = net.reshape(batch_size, -1)
    net = tf_util.fully_connected(net, 512, bn=True, is_training=is_training, scope='tfc1',
                                  bn_decay=bn_decay, is_dist=is_dist)
    net = tf_util.fully_connected(net, 256, bn=True, is_training=is_training, scope='tfc2',
                                  bn_decay=bn_decay, is_dist=is_dist)
    with variable_scope('transform_XYZ'):
        weights = get_variable('weights', [256, K * K], constant_initializer(0.0))
        biases = get_variable('biases', [K * K], constant_initializer(0.0))
        eye = torch.eye(K, dtype=torch.float32, device=net.device).flatten()
        transform = _dense(net, weights, biases + eye)        # (B, 256) -> K*K: the small-GEMM kernel on the GPU (PCOPS_FC)
    return transform.view(batch_size, K, K)
