"""BatchNorm (+ ReLU) of the classifier / T-Net heads as one launch per direction (csrc/head.hip, pcops_fc_bn_fwd / _bwd)
against float64 autograd of the reference's formulas: batch mean / biased variance in training, eps 1e-3 inside the root,
moving <- decay moving + (1 - decay) batch with the unbiased (pointnet2 flavour, tf.contrib.layers.batch_norm:
pointnet2/utils/tf_util.py:512-531) or biased (DGCNN flavour, tf.nn.moments: dgcnn/utils/tf_util.py:462-499) variance."""
import pytest
import torch

from scanobjectnn_amd import fused_mlp

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
EPS = 1e-3


def reference(x, gamma, beta, mm, mv, training, decay, unbiased, relu):
    R = x.shape[0]
    if training:
        mean = x.mean(0)
        var = ((x - mean) ** 2).mean(0)
        mm_new = decay * mm + (1 - decay) * mean.detach()
        mv_new = decay * mv + (1 - decay) * (var.detach() * (R / max(R - 1, 1)) if unbiased else var.detach())
    else:
        mean, var, mm_new, mv_new = mm, mv, mm, mv
    y = (x - mean) * torch.rsqrt(var + EPS) * gamma + beta
    return (torch.relu(y) if relu else y), mm_new, mv_new


@pytest.mark.parametrize("R,C", [(256, 512), (256, 256), (16, 512), (1, 64), (3, 20), (4096, 40), (300, 1024)])
@pytest.mark.parametrize("training", [True, False])
@pytest.mark.parametrize("unbiased,relu", [(True, True), (False, True), (False, False)])
def test_fc_batch_norm(R, C, training, unbiased, relu):
    g = torch.Generator().manual_seed(R * 7 + C)
    x = (torch.randn(R, C, generator=g) * 2 + torch.randn(C, generator=g) * 3).to(DEV)
    gamma = (torch.rand(C, generator=g) + 0.5).to(DEV)
    gamma[::5] *= -1                                        # either sign
    beta = torch.randn(C, generator=g).to(DEV)
    mm0 = torch.randn(C, generator=g).to(DEV)
    mv0 = (torch.rand(C, generator=g) + 0.5).to(DEV)
    go = torch.randn(R, C, generator=g).to(DEV)

    xs, gs, bs = (t.clone().requires_grad_(True) for t in (x, gamma, beta))
    mm, mv = mm0.clone(), mv0.clone()
    y = fused_mlp.fc_batch_norm(xs, gs, bs, mm, mv, training, 0.9, EPS, unbiased, relu)
    y.backward(go)

    xd, gd, bd = (t.double().clone().requires_grad_(True) for t in (x, gamma, beta))
    yr, mmr, mvr = reference(xd, gd, bd, mm0.double(), mv0.double(), training, 0.9, unbiased, relu)
    yr.backward(go.double())
    scale = max(1.0, yr.abs().max().item())
    assert (y.double() - yr).abs().max().item() <= 2e-5 * scale
    assert torch.allclose(mm.double(), mmr, atol=1e-5, rtol=1e-5)
    assert torch.allclose(mv.double(), mvr, atol=1e-5, rtol=1e-5)
    if R == 1 and training:
        return                                              # one row: xhat = 0 / rsqrt(eps), gradients ill-conditioned
    # the ReLU decisions of the two evaluations can differ only where the pre-activation is within rounding of zero
    for got, want in ((xs.grad, xd.grad), (gs.grad, gd.grad), (bs.grad, bd.grad)):
        err = (got.double() - want).abs().max().item()
        assert err <= 1e-4 * max(1.0, want.abs().max().item()), err


def test_fc_layers_use_the_head_kernel():
    """fully_connected(bn=True) of both tf_util mirrors routes through it (same variables, same outputs as the torch path)"""
    from scanobjectnn_amd.graph import Model
    from scanobjectnn_amd.dgcnn import tf_util as d_util
    from scanobjectnn_amd.pointnet2 import tf_util as p_util

    for util in (p_util, d_util):
        def net(x, is_training, bn_decay=None):
            return util.fully_connected(x, 96, bn=True, is_training=is_training, scope='fc', bn_decay=bn_decay), {}
        x = torch.randn(64, 40, device=DEV)
        outs = []
        for flag in (True, False):
            fused_mlp.FC_BN = flag
            try:
                m = Model(net, device=DEV, seed=3).build(x)
                y, _ = m(x, is_training=True, bn_decay=0.7)
                outs.append((y, {k: v.clone() for k, v in m.state_dict().items()}))
            finally:
                fused_mlp.FC_BN = True
        assert sorted(outs[0][1]) == sorted(outs[1][1])
        assert (outs[0][0] - outs[1][0]).abs().max().item() < 1e-5
        for k in outs[0][1]:
            assert torch.allclose(outs[0][1][k], outs[1][1][k], atol=1e-5, rtol=1e-5), k


@pytest.mark.parametrize("B,N,need_dx", [(5, 257, True), (256, 2048, False)])
def test_input_transform_applied_by_the_head_kernel(B, N, need_dx):
    """pcops_transform3_fwd / _bwd = tf.matmul(point_cloud, transform) (dgcnn/models/dgcnn.py:37) and its gradients, against
    float64; dgcnn/tf_util.apply_transform routes (B, N, 3) x (B, 3, 3) through it."""
    from scanobjectnn_amd.dgcnn import tf_util as td
    g = torch.Generator().manual_seed(B + N)
    pc = torch.randn(B, N, 3, generator=g).to(DEV).requires_grad_(need_dx)
    T = (torch.eye(3) + 0.3 * torch.randn(B, 3, 3, generator=g)).to(DEV).requires_grad_(True)
    w = torch.randn(B, N, 3, generator=g).to(DEV)
    out = td.apply_transform(pc, T)
    (out * w).sum().backward()
    pc64 = pc.detach().double().requires_grad_(need_dx)
    T64 = T.detach().double().requires_grad_(True)
    want = torch.matmul(pc64, T64)
    (want * w.double()).sum().backward()
    assert (out.double() - want).abs().max().item() <= 1e-6 * want.abs().max().item()
    assert (T.grad.double() - T64.grad).abs().max().item() <= 1e-5 * T64.grad.abs().max().item()
    if need_dx:
        assert (pc.grad.double() - pc64.grad).abs().max().item() <= 1e-6 * pc64.grad.abs().max().item()
    else:
        assert pc.grad is None
