"""The step's short generic launches folded into their neighbours (fused_mlp.TAIL_FOLD, round 6), each against float64 of the
formula it replaces and the whole step against the unfolded torch forms:
  * pcops_small_gemm_colsum     db = 1^T dY out of the dW = X^T dY launch of fully_connected (pointnet2/utils/tf_util.py:327-363)
  * pcops_mlp_pool_top_prep / _finish   the algebraic top layer's operands and closing sums (include/pcops.h)
  * pcops_softmax_ce            the classification losses (pointnet2_cls_ssg.py:47-53; dgcnn.py:99-105 with label smoothing 0.2)
  * fused_mlp.split_rows        one concatenation as the gradient of a split first-layer weight (pointnet_util.py:50 concat order)"""
import importlib

import pytest
import torch
import torch.nn.functional as F

from scanobjectnn_amd import _lib, fused_mlp
from scanobjectnn_amd.graph import Model
from scanobjectnn_amd.synth import synth_clouds, synth_labels, synth_masks

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("R,K,N", [(256, 1024, 512), (256, 512, 256), (256, 256, 15), (200, 70, 9), (1, 5, 3), (1000, 33, 130)])
def test_small_gemm_colsum(R, K, N):
    g = torch.Generator().manual_seed(R + K + N)
    x = torch.randn(R, K, generator=g).to(DEV)
    gy = (torch.randn(R, N, generator=g) + 0.3).to(DEV)
    dw, db = torch.empty(K, N, device=DEV), torch.full((N,), float("nan"), device=DEV)
    _lib.call("pcops_small_gemm_colsum", K, R, N, x.data_ptr(), K, 1, gy.data_ptr(), N, 0, None, dw.data_ptr(), N, db.data_ptr())
    want_w = x.double().t() @ gy.double()
    want_b = gy.double().sum(0)
    assert (dw.double() - want_w).abs().max().item() <= 1e-5 * max(1.0, want_w.abs().max().item())
    assert (db.double() - want_b).abs().max().item() <= 2e-6 * max(1.0, gy.abs().sum(0).max().item())
    # the same launch without the sums, and the transposed-B form (B stored [N][K]) with them
    dw2 = torch.empty(K, N, device=DEV)
    _lib.call("pcops_small_gemm_colsum", K, R, N, x.data_ptr(), K, 1, gy.data_ptr(), N, 0, None, dw2.data_ptr(), N, None)
    assert torch.equal(dw, dw2)
    gyt = gy.t().contiguous()
    dw3, db3 = torch.empty(K, N, device=DEV), torch.empty(N, device=DEV)
    _lib.call("pcops_small_gemm_colsum", K, R, N, x.data_ptr(), K, 1, gyt.data_ptr(), R, 1, None, dw3.data_ptr(), N, db3.data_ptr())
    assert torch.equal(dw3, dw) and torch.equal(db3, db)
    # run to run the same bits
    db4 = torch.empty(N, device=DEV)
    _lib.call("pcops_small_gemm_colsum", K, R, N, x.data_ptr(), K, 1, gy.data_ptr(), N, 0, None, dw2.data_ptr(), N, db4.data_ptr())
    assert torch.equal(db4, db)


def test_small_linear_bias_gradient():
    g = torch.Generator().manual_seed(3)
    x = torch.randn(256, 512, generator=g).to(DEV)
    w = (torch.randn(512, 40, generator=g) * 0.05).to(DEV)
    b = torch.randn(40, generator=g).to(DEV)
    go = torch.randn(256, 40, generator=g).to(DEV)
    xs, ws, bs = (t.clone().requires_grad_(True) for t in (x, w, b))
    fused_mlp.small_linear(xs, ws, bs).backward(go)
    xd, wd, bd = (t.double().clone().requires_grad_(True) for t in (x, w, b))
    (xd @ wd + bd).backward(go.double())
    for got, want in ((xs.grad, xd.grad), (ws.grad, wd.grad), (bs.grad, bd.grad)):
        assert (got.double() - want).abs().max().item() <= 1e-5 * max(1.0, want.abs().max().item())


@pytest.mark.parametrize("K,N,R", [(512, 1024, 32768), (320, 1024, 524288), (128, 1024, 524288), (33, 70, 1000)])
def test_pool_top_prep_and_finish(K, N, R):
    g = torch.Generator().manual_seed(K + N)
    W = torch.randn(K, N, generator=g).to(DEV)
    b, q, t = (torch.randn(N, generator=g).to(DEV) for _ in range(3))
    Wt, Wq, u = torch.empty(N, K, device=DEV), torch.empty(K, N, device=DEV), torch.empty(N, device=DEV)
    v = torch.full((K,), float("nan"), device=DEV)
    _lib.call("pcops_mlp_pool_top_prep", K, N, W.data_ptr(), b.data_ptr(), q.data_ptr(), t.data_ptr(), Wt.data_ptr(),
              Wq.data_ptr(), u.data_ptr(), v.data_ptr())
    assert torch.equal(Wt, W.t().contiguous())
    assert torch.equal(Wq, W * q)
    ud = q.double() * b.double() + t.double()
    assert (u.double() - ud).abs().max().item() <= 1e-6 * 10
    want_v = W.double() @ ud
    assert (v.double() - want_v).abs().max().item() <= 1e-5 * (W.double().abs() @ ud.abs()).max().item()
    Wt2, Wq2, u2 = torch.empty_like(Wt), torch.empty_like(Wq), torch.empty_like(u)      # v is optional
    _lib.call("pcops_mlp_pool_top_prep", K, N, W.data_ptr(), b.data_ptr(), q.data_ptr(), t.data_ptr(), Wt2.data_ptr(),
              Wq2.data_ptr(), u2.data_ptr(), None)
    assert torch.equal(Wt2, Wt) and torch.equal(Wq2, Wq) and torch.equal(u2, u)

    dW0 = torch.randn(K, N, generator=g).to(DEV) * 50
    Ssp = torch.randn(K, N, generator=g).to(DEV)
    xsum = torch.randn(K, generator=g).to(DEV) * 30
    cfsum = torch.randn(N, generator=g).to(DEV)
    dW, db = dW0.clone(), torch.empty(N, device=DEV)
    _lib.call("pcops_mlp_pool_top_finish", K, N, R, dW.data_ptr(), Ssp.data_ptr(), xsum.data_ptr(), u.data_ptr(),
              cfsum.data_ptr(), q.data_ptr(), W.data_ptr(), b.data_ptr(), t.data_ptr(), db.data_ptr())
    want_w = dW0.double() + Ssp.double() + torch.outer(xsum.double(), u.double())
    want_b = cfsum.double() + q.double() * (xsum.double() @ W.double() + R * b.double()) + R * t.double()
    xw = xsum @ W
    assert (dW.double() - want_w).abs().max().item() <= 1e-6 * want_w.abs().max().item()
    assert (db.double() - want_b).abs().max().item() <= 1e-6 * want_b.abs().max().item()
    # the torch form it replaces, to rounding
    old_w = torch.addr(dW0.clone().add_(Ssp), xsum, u)
    old_b = cfsum + q * (xw + float(R) * b) + float(R) * t
    assert (dW - old_w).abs().max().item() <= 2e-6 * old_w.abs().max().item()
    assert (db - old_b).abs().max().item() <= 2e-6 * old_b.abs().max().item()


@pytest.mark.parametrize("R,C", [(256, 15), (128, 15), (300, 40), (1, 2), (4096, 7)])
@pytest.mark.parametrize("smoothing", [0.0, 0.2])
def test_softmax_cross_entropy(R, C, smoothing):
    g = torch.Generator().manual_seed(R * 31 + C)
    x = (torch.randn(R, C, generator=g) * 4).to(DEV)
    x[0, 0] = 60.0                                           # a saturated row
    y = torch.randint(0, C, (R,), generator=g, dtype=torch.int32).to(DEV)
    xs = x.clone().requires_grad_(True)
    loss = fused_mlp.softmax_cross_entropy(xs, y, label_smoothing=smoothing)
    (loss * 1.7).backward()
    xd = x.double().clone().requires_grad_(True)
    want = F.cross_entropy(xd, y.long(), label_smoothing=smoothing)
    (want * 1.7).backward()
    assert loss.shape == () and abs(loss.item() - want.item()) <= 2e-6 * max(1.0, abs(want.item()))
    assert (xs.grad.double() - xd.grad).abs().max().item() <= 1e-6 * max(1.0 / R, xd.grad.abs().max().item())
    # int64 labels, and the unfolded switch, give the same number
    l2 = fused_mlp.softmax_cross_entropy(x, y.long(), label_smoothing=smoothing)
    assert torch.equal(l2, loss.detach())


@pytest.mark.parametrize("R,C", [(8192, 2), (262144, 2), (4097, 5)])
def test_softmax_cross_entropy_of_many_rows(R, C):
    """beyond 4096 rows several workgroups leave their share of the mean (the per-point mask loss of the BGA models)"""
    assert _lib.load().pcops_softmax_ce_blocks(R) > 1 and _lib.load().pcops_softmax_ce_blocks(4096) == 1
    g = torch.Generator().manual_seed(R)
    x0 = (torch.randn(R, C, generator=g) * 3).to(DEV)
    y = torch.randint(0, C, (R,), generator=g, dtype=torch.int32).to(DEV)
    x = x0.clone().requires_grad_(True)
    loss = fused_mlp.softmax_cross_entropy(x, y)
    loss.backward()
    xd = x0.double().requires_grad_(True)
    want = F.cross_entropy(xd, y.long())
    want.backward()
    assert abs(loss.item() - want.item()) <= 2e-6 * max(1.0, abs(want.item()))
    assert (x.grad.double() - xd.grad).abs().max().item() <= 1e-6 / R * 2


@pytest.mark.parametrize("b,n", [(128, 2048), (3, 5), (1, 1)])
def test_three_nn_weights(b, n):
    g = torch.Generator().manual_seed(b + n)
    dist = (torch.rand(b, n, 3, generator=g) * 0.1).to(DEV)
    dist[0, 0, 0] = 0.0                                      # a query on top of a known point: clamped at 1e-10
    dist[-1, -1, 2] = float("inf")                           # fewer than three known points
    w = torch.empty_like(dist)
    _lib.call("pcops_three_nn_weights", b, n, dist.data_ptr(), w.data_ptr())
    inv = 1.0 / torch.clamp_min(dist, 1e-10)
    want = inv / inv.sum(dim=2, keepdim=True)
    assert torch.allclose(w, want, rtol=1e-6, atol=0), float(((w - want).abs() / want.abs().clamp_min(1e-30)).max())
    assert w[-1, -1, 2].item() == 0.0 and abs(w[0, 0].sum().item() - 1.0) <= 1e-6


def test_split_rows_gradient_is_one_concatenation():
    w = torch.randn(259, 256, device=DEV, requires_grad=True)
    a, b = fused_mlp.split_rows(w, 3)
    assert a.data_ptr() == w.data_ptr() and b.data_ptr() == w.data_ptr() + 3 * 256 * 4 and b.is_contiguous()
    ga, gb = torch.randn(3, 256, device=DEV), torch.randn(256, 256, device=DEV)
    (a * ga).sum().backward(retain_graph=True)
    assert torch.equal(w.grad[:3], ga) and not w.grad[3:].any()        # one half without gradient: zeros for it
    w.grad = None
    ((a * ga).sum() + (b * gb).sum()).backward()
    assert torch.equal(w.grad, torch.cat([ga, gb]))


def _step_grads(modpath, has_mask, B, N, fold, monkeypatch):
    monkeypatch.setattr(fused_mlp, "TAIL_FOLD", fold)
    mod = importlib.import_module(modpath)
    x = torch.from_numpy(synth_clouds(B, N, seed=5)).to(DEV)
    y = torch.from_numpy(synth_labels(B, seed=5)).to(DEV)
    mask = torch.from_numpy(synth_masks(B, N, seed=5)).to(DEV) if has_mask else None
    net = Model(mod.get_model, device=DEV, seed=0).build(x[:2].contiguous())
    torch.manual_seed(11)                                    # the dropout masks
    out = net(x, is_training=True, bn_decay=0.5)
    loss = mod.get_loss(out[0], out[1], y, mask)[0] if has_mask else mod.get_loss(out[0], y, out[1])
    loss.backward()
    names = [n for n, _ in net.named_parameters()] if hasattr(net, "named_parameters") else None
    return loss.detach(), [p.grad.clone() for p in net.parameters()], names


@pytest.mark.parametrize("modpath,has_mask,B,N", [
    ("scanobjectnn_amd.pointnet2.pointnet2_cls_ssg", False, 16, 1024),
    ("scanobjectnn_amd.pointnet2.pointnet2_cls_msg", False, 8, 1024),
    ("scanobjectnn_amd.pointnet2.pointnet2_cls_bga", True, 8, 1024),
    ("scanobjectnn_amd.dgcnn.dgcnn", False, 8, 1024),
    ("scanobjectnn_amd.dgcnn.dgcnn_bga", True, 8, 1024),
])
def test_folded_step_equals_the_torch_forms(modpath, has_mask, B, N, monkeypatch):
    """same decisions (the forward is untouched up to the loss), so every gradient agrees to summation order"""
    l1, g1, _ = _step_grads(modpath, has_mask, B, N, True, monkeypatch)
    l0, g0, _ = _step_grads(modpath, has_mask, B, N, False, monkeypatch)
    assert abs(l1.item() - l0.item()) <= 2e-6 * max(1.0, abs(l0.item()))
    whole = max(float(g.abs().max()) for g in g0)
    for a, b in zip(g1, g0):
        assert a.shape == b.shape
        # (the biases in front of a BatchNorm have a zero gradient: both runs hold rounding noise of the whole gradient's scale there)
        tol = 2e-4 * float(b.abs().max()) + 1e-5 * whole
        assert float((a - b).abs().max()) <= tol, (a.shape, float((a - b).abs().max()), tol)


def test_first_max_on_the_device_gives_ties_to_the_first_member():
    """dgcnn/tf_util._FirstMax on the device: of equal maxima the first one takes the gradient"""
    from scanobjectnn_amd.dgcnn.tf_util import _FirstMax
    x = torch.zeros(64, 8, 1, 1024, device=DEV)
    x[:, 2] = 1.0
    x[:, 5] = 1.0                                            # exact ties in every (cloud, channel)
    x[:, 7, :, ::2] = 1.0
    x.requires_grad_(True)
    out = _FirstMax.apply(x, 1)
    assert out.shape == (64, 1, 1, 1024) and bool((out == 1.0).all())
    g = torch.randn(64, 1, 1, 1024, device=DEV)
    out.backward(g)
    assert torch.equal(x.grad[:, 2:3], g) and not x.grad[:, 5].any() and not x.grad[:, 7].any()


def test_max_pool_over_a_window_of_one_is_the_input():
    from scanobjectnn_amd.pointnet2 import tf_util
    x = torch.randn(4, 1, 1, 64, device=DEV, requires_grad=True)
    y = tf_util.max_pool2d(x, [1, 1], "p")
    assert y.shape == x.shape and torch.equal(y, x)
    y.sum().backward()
    assert bool((x.grad == 1).all())
    z = torch.randn(4, 7, 1, 64, device=DEV)
    assert torch.equal(tf_util.max_pool2d(z, [7, 1], "p"), z.amax(dim=(1, 2), keepdim=True))


@pytest.mark.parametrize("M,N", [(32768, 256), (1000, 64), (1, 4), (70001, 12)])
def test_dy_apply(M, N):
    g = torch.Generator().manual_seed(M + N)
    G, Y = torch.randn(M, N, generator=g).to(DEV), torch.randn(M, N, generator=g).to(DEV)
    p, q, t = (torch.randn(N, generator=g).to(DEV) for _ in range(3))
    out = torch.empty(M, N, device=DEV)
    _lib.call("pcops_mlp_dy_apply", M, N, G.data_ptr(), Y.data_ptr(), p.data_ptr(), q.data_ptr(), t.data_ptr(), out.data_ptr())
    want = t.double() + G.double() * p.double() + Y.double() * q.double()
    assert (out.double() - want).abs().max().item() <= 1e-6 * max(1.0, want.abs().max().item())
    old = torch.addcmul(t, G, p).addcmul_(Y, q)             # the torch form it replaces: the same association
    assert (out - old).abs().max().item() <= 1e-6 * max(1.0, want.abs().max().item())
    with pytest.raises(_lib.PcopsError):                    # N % 4 != 0 is refused
        _lib.call("pcops_mlp_dy_apply", M, 6, G.data_ptr(), Y.data_ptr(), p.data_ptr(), q.data_ptr(), t.data_ptr(), out.data_ptr())


@pytest.mark.parametrize("G,C,two,ld", [(524288, 64, True, 320), (524288, 128, False, 320), (1000, 32, True, 32), (37, 256, False, 260)])
def test_pool_bwd_stats_sum(G, C, two, ld):
    """the pooled gradient of an EdgeConv layer arriving as two pieces and / or as a column block of the concatenation's gradient
    (dgcnn.py:39-81): statistics and the contiguous sum out of one pass -- the same bits as autograd's sum + pcops_mlp_pool_bwd_stats"""
    lib = _lib.load()
    g = torch.Generator().manual_seed(G + C)
    wide = torch.randn(G, ld, generator=g).to(DEV)
    off = 0 if ld == C else 4 * ((ld - C) // 8)
    gb = wide[:, off:off + C]                                # a strided column block
    ga = torch.randn(G, C, generator=g).to(DEV) if two else None
    ysel = torch.randn(G, C, generator=g).to(DEV)
    scale, shift = (torch.rand(C, generator=g) + 0.5).to(DEV), (torch.randn(C, generator=g) * 0.3).to(DEV)
    P = lib.pcops_mlp_bwd_pool_stats_rows(G)
    part, gsum = torch.empty(P, 2, C, device=DEV), torch.empty(G, C, device=DEV)
    if two:
        assert fused_mlp._row_stride(ga) == C and fused_mlp._row_stride(gb) == ld
        _lib.call("pcops_mlp_pool_bwd_stats_sum", G, C, ga.data_ptr(), C, gb.data_ptr(), ld, ysel.data_ptr(), scale.data_ptr(),
                  shift.data_ptr(), part.data_ptr(), gsum.data_ptr())
        want = ga + gb
    else:
        _lib.call("pcops_mlp_pool_bwd_stats_sum", G, C, gb.data_ptr(), ld, None, 0, ysel.data_ptr(), scale.data_ptr(),
                  shift.data_ptr(), part.data_ptr(), gsum.data_ptr())
        want = gb.contiguous()
    assert torch.equal(gsum, want)
    part0 = torch.empty(P, 2, C, device=DEV)
    _lib.call("pcops_mlp_pool_bwd_stats", G, C, want.data_ptr(), ysel.data_ptr(), scale.data_ptr(), shift.data_ptr(),
              part0.data_ptr(), None)
    assert torch.equal(part, part0)
    mask = (ysel.double() * scale.double() + shift.double()) > 0
    gm = want.double() * mask
    tot = part.double().sum(0)
    assert (tot[0] - gm.sum(0)).abs().max().item() <= 1e-5 * gm.abs().sum(0).max().item()
    assert (tot[1] - (gm * ysel.double()).sum(0)).abs().max().item() <= 1e-5 * (gm * ysel.double()).abs().sum(0).max().item()


def test_small_gemm_pair_equals_two_launches():
    """pcops_small_gemm_pair: the two products of a fully connected layer's backward (dX = dY W^T, dW = X^T dY + db) and the
    algebraic top layer's two K-sized products in ONE launch -- bit for bit what two launches give"""
    g = torch.Generator().manual_seed(9)
    for (R, K, N) in [(256, 1024, 512), (256, 256, 15), (200, 70, 9), (1, 33, 5)]:
        x, w, gy = (torch.randn(R, K, generator=g).to(DEV), torch.randn(K, N, generator=g).to(DEV),
                    torch.randn(R, N, generator=g).to(DEV))
        dx0, dw0, db0 = torch.empty(R, K, device=DEV), torch.empty(K, N, device=DEV), torch.empty(N, device=DEV)
        _lib.call("pcops_small_gemm_ex", R, N, K, gy.data_ptr(), N, 0, w.data_ptr(), N, 1, None, dx0.data_ptr(), K)
        _lib.call("pcops_small_gemm_colsum", K, R, N, x.data_ptr(), K, 1, gy.data_ptr(), N, 0, None, dw0.data_ptr(), N, db0.data_ptr())
        dx, dw, db = torch.empty_like(dx0), torch.empty_like(dw0), torch.empty_like(db0)
        _lib.small_gemm_pair((R, N, K, gy.data_ptr(), N, 0, w.data_ptr(), N, 1, None, dx.data_ptr(), K, None),
                             (K, R, N, x.data_ptr(), K, 1, gy.data_ptr(), N, 0, None, dw.data_ptr(), N, db.data_ptr()))
        assert torch.equal(dx, dx0) and torch.equal(dw, dw0) and torch.equal(db, db0)
        assert (dx.double() - gy.double() @ w.double().t()).abs().max().item() <= 1e-5 * max(1.0, (gy.double() @ w.double().t()).abs().max().item())
    K, N = 320, 1024
    Wq, Wt, gram = (torch.randn(K, N, generator=g).to(DEV), torch.randn(N, K, generator=g).to(DEV), torch.randn(K, K, generator=g).to(DEV))
    bias = torch.randn(K, generator=g).to(DEV)
    Mq0, dW0 = torch.empty(K, K, device=DEV), torch.empty(K, N, device=DEV)
    _lib.call("pcops_small_gemm_ex", K, N, K, Wq.data_ptr(), N, 0, Wt.data_ptr(), K, 0, bias.data_ptr(), Mq0.data_ptr(), K)
    _lib.call("pcops_small_gemm", K, K, N, gram.data_ptr(), K, Wq.data_ptr(), N, dW0.data_ptr(), N)
    Mq, dW = torch.empty_like(Mq0), torch.empty_like(dW0)
    _lib.small_gemm_pair((K, N, K, Wq.data_ptr(), N, 0, Wt.data_ptr(), K, 0, bias.data_ptr(), Mq.data_ptr(), K, None),
                         (K, K, N, gram.data_ptr(), K, 0, Wq.data_ptr(), N, 0, None, dW.data_ptr(), N, None))
    assert torch.equal(Mq, Mq0) and torch.equal(dW, dW0)
    with pytest.raises(_lib.PcopsError):
        _lib.small_gemm_pair((0, N, K, Wq.data_ptr(), N, 0, Wt.data_ptr(), K, 0, None, Mq.data_ptr(), K, None),
                             (K, K, N, gram.data_ptr(), K, 0, Wq.data_ptr(), N, 0, None, dW.data_ptr(), N, None))
