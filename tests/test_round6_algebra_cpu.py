"""CPU statements of round 6's algebra (no GPU, no library call): the Gram form of a pooled layer's weight gradient
(csrc/mlp.hip bwd_fused_kernel<..., GW> + bwd_fused_gw_finish_kernel) and the first-maximiser rule of
the whole-cloud pool's host-side step (dgcnn/tf_util._FirstMax)."""
import pytest
import torch


@pytest.mark.parametrize("rows,K,N,S,bias", [(640, 64, 128, 32, True), (600, 48, 96, 20, False), (36 * 12, 64, 64, 12, True)])
def test_gram_form_of_the_pooled_weight_gradient(rows, K, N, S, bias):
    """dY = p.G + q.Y + t with ONE non-zero row of G per (group, channel) and Y = X W + b:
       X^T dY = X^T (p.G) + (X^T X) W diag(q) + (X^T 1)(q.b + t)^T   -- what the kernel accumulates (arg rows as vector work, a
       K x K Gram matrix on the matrix pipe, column sums of X) and what the finishing kernel combines."""
    g = torch.Generator().manual_seed(rows + N)
    X = torch.relu(torch.randn(rows, K, generator=g, dtype=torch.float64))
    W = torch.randn(K, N, generator=g, dtype=torch.float64) / K ** 0.5
    b = torch.randn(N, generator=g, dtype=torch.float64) if bias else torch.zeros(N, dtype=torch.float64)
    p, q, t = (torch.randn(N, generator=g, dtype=torch.float64) for _ in range(3))
    G = rows // S
    gpool = torch.randn(G, N, generator=g, dtype=torch.float64)
    arg = torch.randint(0, S, (G, N), generator=g)
    Gfull = torch.zeros(G, S, N, dtype=torch.float64).scatter_(1, arg.unsqueeze(1), gpool.unsqueeze(1)).view(rows, N)
    Y = X @ W + b
    direct = X.t() @ (p * Gfull + q * Y + t)
    # the three pieces the kernel's workgroups leave
    sparse = torch.zeros(K, N, dtype=torch.float64)
    for grp in range(G):
        for c in range(N):
            sparse[:, c] += p[c] * gpool[grp, c] * X[grp * S + arg[grp, c]]
    gram, xsum = X.t() @ X, X.sum(0)
    combined = sparse + (gram @ W) * q + torch.outer(xsum, q * b + t)
    assert (combined - direct).abs().max() <= 1e-10 * direct.abs().max()


def test_first_maximiser_takes_the_gradient_of_the_chunk_maxima():
    """two chunks of a cloud whose maxima agree to the last bit: torch's amax backward halves the gradient between them, the
    contract and every pooling kernel give it to the first -- dgcnn/tf_util._FirstMax"""
    from scanobjectnn_amd.dgcnn.tf_util import _FirstMax
    part = torch.tensor([[[[1.0, 5.0]], [[3.0, 5.0]], [[3.0, 2.0]]]], requires_grad=True)     # (B=1, chunks=3, 1, C=2)
    out = _FirstMax.apply(part, 1)
    assert out.shape == (1, 1, 1, 2) and out.flatten().tolist() == [3.0, 5.0]
    out.backward(torch.tensor([[[[10.0, 20.0]]]]))
    assert part.grad.flatten().tolist() == [0.0, 20.0, 10.0, 0.0, 0.0, 0.0]
    ref = part.detach().clone().requires_grad_(True)
    ref.amax(dim=1, keepdim=True).backward(torch.tensor([[[[10.0, 20.0]]]]))
    assert ref.grad.flatten().tolist() == [0.0, 10.0, 5.0, 10.0, 5.0, 0.0]                  # (what the product used to do)


def test_split_rows_is_the_slices_with_one_concatenated_gradient():
    """fused_mlp.split_rows (the first-layer weight of a set-abstraction stack split into its xyz and feature rows,
    pointnet_util.py:50 concat order): the same views as w[:k] / w[k:], the same gradient as autograd's slices"""
    from scanobjectnn_amd import fused_mlp
    assert fused_mlp.TAIL_FOLD
    g = torch.Generator().manual_seed(1)
    w = torch.randn(131, 16, generator=g, requires_grad=True)
    a, b = fused_mlp.split_rows(w, 3)
    assert a.shape == (3, 16) and b.shape == (128, 16) and a.data_ptr() == w.data_ptr() and b.is_contiguous()
    ga, gb = torch.randn(3, 16, generator=g), torch.randn(128, 16, generator=g)
    ((a * ga).sum() + (b * gb).sum()).backward()
    w2 = w.detach().clone().requires_grad_(True)
    ((w2[:3] * ga).sum() + (w2[3:] * gb).sum()).backward()
    assert torch.equal(w.grad, w2.grad)
    w.grad = None
    a, b = fused_mlp.split_rows(w, 3)
    (b * gb).sum().backward()                                # only one half reached by the loss
    assert torch.equal(w.grad[3:], gb) and not w.grad[:3].any()
    assert fused_mlp.split_rows(w.detach(), 3)[1].data_ptr() == w.data_ptr() + 3 * 16 * 4      # no gradient: plain views


def test_softmax_cross_entropy_off_the_device_is_torchs():
    import torch.nn.functional as F
    from scanobjectnn_amd import fused_mlp
    g = torch.Generator().manual_seed(2)
    x = torch.randn(32, 15, generator=g, requires_grad=True)
    y = torch.randint(0, 15, (32,), generator=g, dtype=torch.int32)
    for s in (0.0, 0.2):
        assert torch.equal(fused_mlp.softmax_cross_entropy(x, y, label_smoothing=s), F.cross_entropy(x, y.long(), label_smoothing=s))


def test_whole_cloud_group_is_built_once_per_shape():
    """sample_and_group_all's origin and index (pointnet_util.py:59-84) are constants of (b, n): the cached pair is the one
    three launches built"""
    from scanobjectnn_amd.pointnet2 import pointnet_util
    z1, i1 = pointnet_util._whole_cloud_group(4, 7, torch.device("cpu"))
    z2, i2 = pointnet_util._whole_cloud_group(4, 7, torch.device("cpu"))
    assert z1 is z2 and i1 is i2
    assert z1.shape == (4, 1, 3) and not z1.any()
    assert i1.dtype == torch.int32 and i1.is_contiguous() and torch.equal(i1, torch.arange(7, dtype=torch.int32).expand(4, 1, 7))
    z3, i3 = pointnet_util._whole_cloud_group(5, 7, torch.device("cpu"))
    assert i3.shape == (5, 1, 7) and i3 is not i1


def test_max_pool_over_one_element_returns_its_input():
    from scanobjectnn_amd.pointnet2 import tf_util
    x = torch.randn(3, 1, 1, 8)
    assert tf_util.max_pool2d(x, [1, 1], "p") is x
    z = torch.randn(3, 5, 1, 8)
    assert torch.equal(tf_util.max_pool2d(z, [5, 1], "p"), z.amax(dim=(1, 2), keepdim=True))
