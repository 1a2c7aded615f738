"""Deterministic backward switch (reference hazard: float atomics in groupPointGrad /
scatteraddpoint / threeinterpolate_grad, tf_grouping_g.cu:61-78, tf_sampling_g.cu:183-192).

With `_lib.set_deterministic(True)` every scatter-add of the library is taken by one owner per destination point
in ascending row order, so two runs of the same training step must give BIT-IDENTICAL gradients -- asserted with
torch.equal, no tolerance.  The ordered sums are also checked against the float64 sum and against the default
(atomic) path."""
import numpy as np
import pytest
import torch

from scanobjectnn_amd import _lib
from scanobjectnn_amd.graph import Model
from scanobjectnn_amd.synth import synth_clouds, synth_labels, synth_masks

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture
def deterministic():
    _lib.set_deterministic(True)
    yield
    _lib.set_deterministic(False)


def _scatter_truth(idx, src, ndst, div=1, w=None):
    b, rows = idx.shape
    c = src.shape[-1]
    out = torch.zeros((b, ndst, c), dtype=torch.float64, device=idx.device)
    rows_src = torch.arange(rows, device=idx.device) // div
    val = src.double()[:, rows_src, :]
    if w is not None:
        val = val * w.double().view(b, rows, 1)
    out.scatter_add_(1, idx.long().view(b, rows, 1).expand(b, rows, c), val)
    return out


@pytest.mark.parametrize("b,rows,ndst,c,div", [(3, 512, 64, 3, 1), (4, 4096, 300, 64, 1), (2, 3 * 700, 128, 96, 3),
                                               (2, 2048 * 20, 2048, 64, 1), (1, 5, 7, 1, 1)])
def test_scatter_rows_sorted(b, rows, ndst, c, div):
    """pcops_scatter_rows_sorted == the float64 scatter-add, twice the same bits, on skewed destination histograms
    (half of the rows hit 3 destinations: long lists)."""
    g = torch.Generator().manual_seed(rows + c)
    idx = torch.randint(0, ndst, (b, rows), generator=g, dtype=torch.int32)
    idx[:, ::2] = idx[:, ::2] % min(3, ndst)
    src = torch.randn((b, rows // div, c), generator=g)
    w = torch.rand((b, rows), generator=g) if div > 1 else None
    idx, src = idx.to(DEV), src.to(DEV)
    w = w.to(DEV) if w is not None else None
    want = _scatter_truth(idx, src, ndst, div, w)
    got = _lib.scatter_rows_sorted(idx, src, ndst, div=div, w=w)
    again = _lib.scatter_rows_sorted(idx, src, ndst, div=div, w=w)
    assert torch.equal(got, again)
    scale = want.abs().max().item() + 1e-30
    assert (got.double() - want).abs().max().item() <= 2e-6 * scale * max(1.0, (rows / ndst) ** 0.5)
    # accumulate form: out += scatter
    base = torch.randn_like(got)
    acc = _lib.scatter_rows_sorted(idx, src, ndst, div=div, w=w, out=base.clone())
    assert torch.allclose(acc, base + got, rtol=0, atol=1e-5 * scale)


def test_ordered_sum_is_the_sequential_sum():
    """the owner adds in ascending row order: equal to a float32 running sum over the rows, bit for bit"""
    g = torch.Generator().manual_seed(5)
    b, rows, ndst, c = 2, 777, 9, 8
    idx = torch.randint(0, ndst, (b, rows), generator=g, dtype=torch.int32)
    src = torch.randn((b, rows, c), generator=g)
    got = _lib.scatter_rows_sorted(idx.to(DEV), src.to(DEV), ndst).cpu().numpy()
    want = np.zeros((b, ndst, c), np.float32)
    s, ix = src.numpy(), idx.numpy()
    for bi in range(b):
        for r in range(rows):
            want[bi, ix[bi, r]] = want[bi, ix[bi, r]] + s[bi, r]
    assert np.array_equal(got, want)


def test_unfused_gradient_ops(deterministic):
    """group_point / gather_point / three_interpolate / get_edge_feature backward in deterministic mode: same
    values as the atomic path (to rounding), identical bits run to run; the atomic launchers refuse"""
    from scanobjectnn_amd.dgcnn import tf_util as td
    from scanobjectnn_amd.pointnet2.tf_grouping import group_point
    from scanobjectnn_amd.pointnet2.tf_interpolate import three_interpolate
    from scanobjectnn_amd.pointnet2.tf_sampling import gather_point
    g = torch.Generator().manual_seed(3)
    B, N, M, S, C, K = 4, 512, 128, 32, 64, 20
    pts = torch.randn((B, N, C), generator=g).to(DEV)
    xyz = torch.randn((B, N, 3), generator=g).to(DEV)
    idx = torch.randint(0, N, (B, M, S), generator=g, dtype=torch.int32).to(DEV)
    idx[:, :, S // 2:] = idx[:, :, :1]                       # ball-query style padding with the first index
    fidx = torch.randint(0, N, (B, M), generator=g, dtype=torch.int32).to(DEV)
    idx3 = torch.randint(0, M, (B, N, 3), generator=g, dtype=torch.int32).to(DEV)
    w3 = torch.rand((B, N, 3), generator=g).to(DEV)
    sparse = torch.randn((B, M, C), generator=g).to(DEV)
    nn = torch.randint(0, N, (B, N, K), generator=g, dtype=torch.int32).to(DEV)

    def grads():
        out = []
        p = pts.clone().requires_grad_(True)
        (group_point(p, idx) ** 2).sum().backward()
        out.append(p.grad)
        x = xyz.clone().requires_grad_(True)
        (gather_point(x, fidx) ** 2).sum().backward()
        out.append(x.grad)
        s = sparse.clone().requires_grad_(True)
        (three_interpolate(s, idx3, w3) ** 2).sum().backward()
        out.append(s.grad)
        e = pts.clone().requires_grad_(True)
        (td.get_edge_feature(e, nn, k=K) ** 2).sum().backward()
        out.append(e.grad)
        return out

    a, bb = grads(), grads()
    for u, v in zip(a, bb):
        assert torch.equal(u, v)
    with pytest.raises(_lib.PcopsError):
        _lib.call("pcops_group_point_grad", B, N, C, M, S, pts.data_ptr(), idx.data_ptr(), pts.data_ptr())
    _lib.set_deterministic(False)
    ref = grads()
    for u, v in zip(a, ref):
        assert torch.allclose(u, v, rtol=1e-4, atol=1e-4 * v.abs().max().item())


MODELS = ["ssg", "msg", "bga", "partseg", "dgcnn", "dgcnn_bga"]


@pytest.mark.parametrize("name", MODELS)
def test_training_step_is_bit_reproducible(name, deterministic):
    """forward + backward of a whole model twice from the same state: every gradient, the loss and the BN moving
    statistics identical to the last bit"""
    from scanobjectnn_amd.dgcnn import dgcnn, dgcnn_bga
    from scanobjectnn_amd.pointnet2 import (pointnet2_cls_bga, pointnet2_cls_msg, pointnet2_cls_partseg,
                                            pointnet2_cls_ssg)
    mod, n_pts, kind = {
        "ssg": (pointnet2_cls_ssg, 1024, "cls"), "msg": (pointnet2_cls_msg, 1024, "cls"),
        "bga": (pointnet2_cls_bga, 1024, "mask"), "partseg": (pointnet2_cls_partseg, 1024, "seg"),
        "dgcnn": (dgcnn, 512, "cls"), "dgcnn_bga": (dgcnn_bga, 512, "mask")}[name]
    B = 8
    x = torch.from_numpy(synth_clouds(B, n_pts, seed=4)).to(DEV)
    y = torch.from_numpy(synth_labels(B, seed=4)).to(DEV)
    mask = torch.from_numpy(synth_masks(B, n_pts, seed=4)).to(DEV)
    net = Model(mod.get_model, device=DEV, seed=2).build(x)
    sd = {k: v.clone() for k, v in net.state_dict().items()}

    def step():
        net.load_state_dict(sd)
        net.zero_grad(set_to_none=True)
        torch.manual_seed(11)                  # dropout masks
        out = net(x, is_training=True, bn_decay=0.9)
        if kind == "mask":
            loss = mod.get_loss(out[0], out[1], y, mask)[0]
        elif kind == "seg":
            loss = mod.get_loss(out[0] if isinstance(out, (tuple, list)) else out, mask)
        else:
            loss = mod.get_loss(out[0], y)
        loss.backward()
        grads = {k: p.grad.clone() for k, p in net.named_parameters() if p.grad is not None}
        bufs = {k: v.clone() for k, v in net.state_dict().items()}
        return loss.detach().clone(), grads, bufs

    l1, g1, b1 = step()
    l2, g2, b2 = step()
    assert len(g1) > 10
    assert torch.equal(l1, l2)
    diff = [k for k in g1 if not torch.equal(g1[k], g2[k])]
    assert not diff, "gradients differ between two deterministic runs: %s" % diff[:8]
    assert all(torch.equal(b1[k], b2[k]) for k in b1)
    # and the ordered sums are the same numbers as the default path, to rounding
    _lib.set_deterministic(False)
    _, g3, _ = step()
    num = sum(float(((g1[k] - g3[k]).double() ** 2).sum()) for k in g1)
    den = sum(float((g3[k].double() ** 2).sum()) for k in g1)
    assert (num / den) ** 0.5 <= 1e-3


def test_large_destination_sets_fall_back_to_atomics_or_say_why():
    """the ordered scatter keeps its counting sort in LDS (<= 19 968 destination points per cloud).  Beyond
    that the DEFAULT three_interpolate backward must still work (atomic form), and deterministic mode must refuse with
    a message that names the limit instead of a bare status code."""
    from scanobjectnn_amd.pointnet2 import tf_interpolate
    b, n, m, c = 1, 4096, 20480, 8
    assert not _lib.scatter_rows_sorted_supported(3 * n, m) and _lib.scatter_rows_sorted_supported(3 * n, 19968)
    g = torch.Generator().manual_seed(0)
    pts = torch.randn((b, m, c), generator=g).to(DEV).requires_grad_(True)
    idx = torch.randint(0, m, (b, n, 3), generator=g, dtype=torch.int32).to(DEV)
    w = torch.rand((b, n, 3), generator=g).to(DEV)
    up = torch.randn((b, n, c), generator=g).to(DEV)
    tf_interpolate.three_interpolate(pts, idx, w).backward(up)
    want = _scatter_truth(idx.view(b, 3 * n), up, m, div=3, w=w.view(b, 3 * n))
    assert (pts.grad.double() - want).abs().max().item() <= 1e-5
    _lib.set_deterministic(True)
    try:
        pts.grad = None
        with pytest.raises(_lib.PcopsError, match="19968"):
            tf_interpolate.three_interpolate(pts, idx, w).backward(up)
    finally:
        _lib.set_deterministic(False)


def test_edge_family_backward_survives_the_switch_between_forward_and_backward():
    """fused_mlp picks the [Q | Ctr] EdgeConv family at forward time; its backward entry points used to re-check
    the process-wide deterministic switch and return UNSUPPORTED if it had been thrown in between.  They check the shape only
    now: the step completes (correct, not bit-reproducible), and its gradients agree with an undisturbed step to rounding."""
    from scanobjectnn_amd.dgcnn import dgcnn
    _lib.set_deterministic(False)
    B, n_pts = 8, 2048                     # (the LDS-resident [Q | Ctr] family: whole 64-group chunks)
    x = torch.from_numpy(synth_clouds(B, n_pts, seed=4)).to(DEV)
    y = torch.from_numpy(synth_labels(B, seed=4)).to(DEV)
    assert _lib.load().pcops_edge_ld_supported(B, n_pts, n_pts, 20, 64) == 1
    net = Model(dgcnn.get_model, device=DEV, seed=2).build(x)
    sd = {k: v.clone() for k, v in net.state_dict().items()}

    def step(flip):
        net.load_state_dict(sd)
        net.zero_grad(set_to_none=True)
        torch.manual_seed(11)
        out = net(x, is_training=True, bn_decay=0.9)
        loss = dgcnn.get_loss(out[0], y)
        try:
            if flip:
                _lib.set_deterministic(True)
                assert _lib.load().pcops_edge_ld_supported(B, n_pts, n_pts, 20, 64) == 0      # the QUERY says no now
            loss.backward()
        finally:
            _lib.set_deterministic(False)
        return {k: p.grad.clone() for k, p in net.named_parameters() if p.grad is not None}

    g0, g1 = step(False), step(True)
    num = sum(float(((g1[k] - g0[k]).double() ** 2).sum()) for k in g0)
    den = sum(float((g0[k].double() ** 2).sum()) for k in g0)
    assert (num / den) ** 0.5 <= 1e-3
