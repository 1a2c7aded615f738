"""Layer wrappers of vanilla PointNet -- mirror of `pointnet/utils/tf_util.py` (conv2d :115-173,
fully_connected :311-348, max_pool2d :351-374, batch_norm_template :455-490, dropout :548-569).

The BatchNorm of this directory is NOT the `tf.contrib.layers.batch_norm` of pointnet2/: it is the explicit
`tf.nn.moments` + `tf.train.ExponentialMovingAverage` template (:455-490) -- batch mean / BIASED variance, and
the BIASED variance is also what the moving average tracks (the pointnet2 flavour feeds the unbiased one).  That
is the flavour `dgcnn/utils/tf_util.py:462-499` copies, so the layer functions are shared with
`scanobjectnn_amd.dgcnn.tf_util` (same variable names: `bn/beta`, `bn/gamma`, `bn/moving_mean`,
`bn/moving_variance`); only the argument lists differ (no `is_dist` here).
"""
from ..dgcnn import tf_util as _dg
from ..pointnet2.tf_util import avg_pool2d, dropout, max_pool2d, relu  # noqa: F401


def conv2d(inputs, num_output_channels, kernel_size, scope, stride=[1, 1], padding='SAME', use_xavier=True,
           stddev=1e-3, weight_decay=0.0, activation_fn=relu, bn=False, bn_decay=None, is_training=None):
    """tf_util.py:115-173"""
    return _dg.conv2d(inputs, num_output_channels, kernel_size, scope, stride=stride, padding=padding,
                      use_xavier=use_xavier, stddev=stddev, weight_decay=weight_decay, activation_fn=activation_fn,
                      bn=bn, bn_decay=bn_decay, is_training=is_training, is_dist=False)


def fully_connected(inputs, num_outputs, scope, use_xavier=True, stddev=1e-3, weight_decay=0.0,
                    activation_fn=relu, bn=False, bn_decay=None, is_training=None):
    """tf_util.py:311-348"""
    return _dg.fully_connected(inputs, num_outputs, scope, use_xavier=use_xavier, stddev=stddev,
                               weight_decay=weight_decay, activation_fn=activation_fn, bn=bn, bn_decay=bn_decay,
                               is_training=is_training, is_dist=False)
