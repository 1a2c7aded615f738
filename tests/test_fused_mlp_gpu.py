"""The fused shared-MLP stack (csrc/mlp.hip via fused_mlp.py) against a plain PyTorch fp32/fp64 reference of
the same op chain: L x [X·W + b -> BatchNorm (batch stats, biased var, eps 1e-3) -> ReLU] (-> max over S).
Forward within 1e-4 (abs, activations are O(1)); parameter / input gradients within 1e-3 relative to the
gradient's max; moving statistics updated like TF (decay, unbiased variance)."""
import pytest
import torch

import mlp_ref as MR
from scanobjectnn_amd import fused_mlp

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
EPS = 1e-3


def make_layers(k0, widths, seed, dtype=torch.float32):
    g = torch.Generator().manual_seed(seed)
    layers, cin = [], k0
    for w in widths:
        W = (torch.randn(cin, w, generator=g) / cin ** 0.5).to(DEV)
        b = (0.1 * torch.randn(w, generator=g)).to(DEV)
        # every third channel gets a negative gamma: the fused max-pool then has to pick the group MINIMUM
        gamma = ((0.5 + torch.rand(w, generator=g)) * (1.0 - 2.0 * (torch.arange(w) % 3 == 2))).to(DEV)
        beta = (0.2 * torch.randn(w, generator=g)).to(DEV)
        mm = (0.1 * torch.randn(w, generator=g)).to(DEV)
        mv = (0.5 + torch.rand(w, generator=g)).to(DEV)
        layers.append([W, b, gamma, beta, mm, mv])
        cin = w
    return layers


def reference(x, layers, S, pool, training, dtype):
    a = x.to(dtype)
    for (W, b, gamma, beta, mm, mv) in layers:
        y = a @ W.to(dtype) + b.to(dtype)
        if training:
            var, mean = torch.var_mean(y, dim=0, unbiased=False)
        else:
            mean, var = mm.to(dtype), mv.to(dtype)
        a = torch.relu((y - mean) * torch.rsqrt(var + EPS) * gamma.to(dtype) + beta.to(dtype))
    if pool:
        a = a.view(-1, S, a.shape[1]).amax(dim=1)
    return a


CASES = [  # (R, S, K0, widths, pool)
    (512 * 32, 32, 3, [64, 64, 128], True),        # SA1 shape (small batch)
    (128 * 64, 64, 131, [128, 128, 256], True),    # SA2: K0 not a multiple of 4
    (4 * 128, 128, 259, [256, 512, 1024], True),   # SA3 group_all
    (1000, 1, 384, [256, 128], False),             # FP stack, ragged row count
    (777, 1, 128, [128], False),                   # single layer
    (130 * 20, 20, 6, [64], True),                 # EdgeConv-like
    # >= 32768 rows: the wave-stream GEMM variant (weights resident in LDS) takes the K%8==0 layers
    (1024 * 32 * 2, 32, 3, [64, 64, 128], True),   # SA1, KC=64 stripes, N 64 / 128
    (512 * 64 + 37, 1, 128, [128, 256], False),    # KC=128, N=256 (two column blocks), ragged tail rows
    (128 * 64 * 4, 64, 132, [128, 128, 256], True),  # pooled backward through the wave-stream dgrad (K=256 in 2 chunks)
    (40000, 1, 256, [64], False),                  # K=256 chunked, N=64
    (400 * 96, 96, 64, [64, 128], True),           # pooling fused into the GEMM epilogue, 3 tiles per group
    (130 * 256, 256, 32, [128, 64], True),         # ... 8 tiles per group (largest 8-bit arg index)
    (1100 * 32, 32, 16, [32, 32], True),           # ... N=32: half-empty column block
    # K > 256: weights streamed through LDS in 64-row chunks (one workgroup barrier per chunk)
    (256 * 128, 128, 260, [256, 512, 1024], True),  # SA3 of the SSG config at B=256 (pooled 1024-wide last layer)
    (33000, 1, 320, [1024, 64], False),            # DGCNN aggregation shapes, ragged tail, N=64 behind K=1024
    # one-layer stacks pooled over 256-row chunks (DGCNN agg / T-Net tconv3): pooled forward on the raw input without
    # storing Y, algebraic backward with the plain-input variants (A_PLAIN operand, E_PLAINA epilogue, plain Gram)
    (64 * 256, 256, 320, [1024], True),
    (48 * 256, 256, 128, [1024], True),
    # pooling groups that are not multiples of the 32-row tile on the wave-stream kernels: a tile meets up to four groups
    # and adds their arg rows after the dense staging (DGCNN's T-Net: k = 20; MSG scale 0: nsample 16)
    (4096 * 20, 20, 64, [64, 128], True),
    (2048 * 16, 16, 32, [64, 64], True),
    (1024 * 48, 48, 32, [64, 128], True),
    (3000 * 11, 11, 64, [128], True),
    # 96-wide layers (MSG's 64 -> 96 -> 128 scale): three 32-column blocks in the forward / data-gradient kernels, the
    # 96 x 32 consumer layout of the weight gradient, the one-pass backward without the zero columns
    (1024 * 32 * 2 + 64, 32, 32, [64, 96, 128], True),
    (70000 + 19, 1, 64, [96, 128], False),         # ... ragged tail, dense upstream gradient
    (2048 * 16, 16, 64, [96, 128], True),          # ... groups that are not tile multiples
    # a pooled top layer WIDER than 64 on both sides behind groups that are not stripe multiples: the per-row group
    # arithmetic of the split-operand weight gradient (four consecutive rows per producer lane)
    (2048 * 20, 20, 32, [128, 128], True),
    (1100 * 48, 48, 16, [96, 128], True),
]


@pytest.mark.parametrize("R,S,K0,widths,pool", CASES)
def test_forward_train_and_eval(R, S, K0, widths, pool):
    g = torch.Generator().manual_seed(R + K0)
    x = torch.randn(R, K0, generator=g).to(DEV)
    for training in (True, False):
        layers = make_layers(K0, widths, seed=K0)
        mov_before = [(l[4].clone(), l[5].clone()) for l in layers]
        out = fused_mlp.mlp_stack(x, S, pool, training, 0.9, EPS, True, [tuple(l) for l in layers])
        want = reference(x, layers if not training else [l[:4] + list(mb) for l, mb in zip(layers, mov_before)],
                         S, pool, training, torch.float64)
        assert out.shape == want.shape
        assert (out.double() - want).abs().max().item() < 1e-4
        if training:   # TF moving-average update with the unbiased batch variance
            a = x.double()
            for l, (mm0, mv0) in zip(layers, mov_before):
                y = a @ l[0].double() + l[1].double()
                var, mean = torch.var_mean(y, dim=0, unbiased=False)
                n = y.shape[0]
                assert torch.allclose(l[4].double(), 0.9 * mm0.double() + 0.1 * mean, atol=1e-5)
                assert torch.allclose(l[5].double(), 0.9 * mv0.double() + 0.1 * var * n / (n - 1), atol=1e-5)
                a = torch.relu((y - mean) * torch.rsqrt(var + EPS) * l[2].double() + l[3].double())
        else:
            for l, (mm0, mv0) in zip(layers, mov_before):
                assert torch.equal(l[4], mm0) and torch.equal(l[5], mv0)


@pytest.mark.parametrize("R,S,K0,widths,pool", CASES)
def test_backward(R, S, K0, widths, pool):
    """every gradient of the stack against float64 autograd of the same chain evaluated with the activation pattern
    the kernels used (tests/mlp_ref.py): 1e-3 of the gradient's max, bounded below only by the error plain fp32
    autograd makes on the same computation"""
    g = torch.Generator().manual_seed(R + 7)
    x = torch.randn(R, K0, generator=g).to(DEV).requires_grad_(True)
    layers = make_layers(K0, widths, seed=K0 + 1)
    for l in layers:
        for t in l[:4]:
            t.requires_grad_(True)
    mov = [(l[4].clone(), l[5].clone()) for l in layers]
    out = fused_mlp.mlp_stack(x, S, pool, True, 0.9, EPS, True, [tuple(l) for l in layers])
    pattern = MR.fused_pattern(out)
    go = torch.randn(out.shape, generator=g).to(DEV)
    out.backward(go)
    got = [x.grad.clone()] + [t.grad.clone() for l in layers for t in l[:4]]

    def run_ref(dtype, report=None):
        xr = x.detach().to(dtype).requires_grad_(True)
        lr = [[t.detach().to(dtype).requires_grad_(True) for t in l[:4]] + list(mb) for l, mb in zip(layers, mov)]
        o = MR.run_stack(None, xr, lr, S, pool, True, dtype, pattern, report)
        o.backward(go.to(dtype))
        return o.detach(), [xr.grad.double()] + [t.grad.double() for l in lr for t in l[:4]]

    rep = {}
    fwd64, want = run_ref(torch.float64, rep)
    MR.check_pattern(rep, R * sum(widths))
    assert (out.detach().double() - fwd64).abs().max().item() < 1e-4
    _, plain = run_ref(torch.float32)
    names = ["dx"] + ["L%d.%s" % (i, n) for i in range(len(layers)) for n in ("dW", "db", "dgamma", "dbeta")]
    MR.assert_grads_close(names, got, want, plain, floor_scale=go.abs().max().item())


def test_fused_model_as_accurate_as_layerwise(monkeypatch):
    """pointnet2_cls_ssg through the fused stacks vs the same model through tf_util.conv2d layer by layer
    (library GEMM + torch BN), both judged against the float64 restatement: same logits (1e-4), and the fused
    gradients are at least as close to the truth as the layer-wise ones (up to 2x + a small floor)."""
    from oracle import ref_models as R
    from scanobjectnn_amd.graph import Model
    from scanobjectnn_amd.pointnet2 import pointnet2_cls_ssg as m
    from scanobjectnn_amd.pointnet2 import tf_util
    from scanobjectnn_amd.synth import synth_clouds
    c = synth_clouds(8, 1024, seed=1)
    x = torch.from_numpy(c).to(DEV)
    monkeypatch.setattr(tf_util, "dropout", lambda inputs, is_training, scope, keep_prob=0.5, noise_shape=None: inputs)
    net = Model(m.get_model, device=DEV, seed=0).build(x)
    sd = {k: v.clone() for k, v in net.state_dict().items()}
    P = {k: v.requires_grad_(v.is_floating_point())
         for k, v in R.params_from_state_dict(sd, dtype=torch.float64).items()}
    truth = R.pointnet2_cls_ssg(torch.from_numpy(c).double(), P, True)
    truth.square().mean().backward()
    outs = []
    for fused in (True, False):
        monkeypatch.setattr(tf_util, "FUSED_MLP", fused)
        net.load_state_dict(sd)
        logits, _ = net(x, is_training=True, bn_decay=0.9)
        net.zero_grad()
        logits.square().mean().backward()
        outs.append((logits.detach().cpu().double(), {n: p.grad.cpu().double() for n, p in net.named_parameters()}))
    assert (outs[0][0] - truth.detach()).abs().max().item() < 1e-4
    # Frobenius norms: a single ReLU landing on the other side of 0 than in float64 moves one gradient element
    # by O(1) in either fp32 path (see test_models_parity_gpu.py), which a max-norm comparison cannot absorb
    tot = {"f": 0.0, "l": 0.0, "r": 0.0}
    for n in outs[0][1]:
        if n.endswith("biases") and n[:-len("biases")] + "bn/gamma" in outs[0][1]:
            continue   # analytically zero gradient (bias in front of a batch norm): rounding noise only
        ref = P[n[len("graph."):]].grad
        tot["f"] += (outs[0][1][n] - ref).norm().item() ** 2
        tot["l"] += (outs[1][1][n] - ref).norm().item() ** 2
        tot["r"] += ref.norm().item() ** 2
    e_fused, e_layer = (tot["f"] / tot["r"]) ** 0.5, (tot["l"] / tot["r"]) ** 0.5
    assert e_fused <= max(4.0 * e_layer, 2e-2), (e_fused, e_layer)


# ---------------------------------------------------------------------------- gather-first stacks
@pytest.mark.parametrize("R,S,K0,widths", [(64 * 256, 256, 320, [1024]), (256 * 128, 128, 260, [256, 512, 1024]),
                                           (128 * 64 * 4, 64, 132, [128, 128, 256])])
def test_pool_top_backward_is_taken_and_equals_the_plain_form(R, S, K0, widths, monkeypatch):
    """the algebraic backward of a pooled top layer (pcops.h; fused_mlp._pool_top_backward) runs for these shapes --
    its entry points are seen on the call hook, the pooled forward stores no Y -- and gives the gradients of the plain
    dgrad / wgrad kernels to fp32 rounding (the two forms differ in summation order only)"""
    from scanobjectnn_amd import _lib
    g = torch.Generator().manual_seed(R)
    x0 = torch.randn(R, K0, generator=g).to(DEV)
    go = None
    res = {}
    for mode in (True, False):
        monkeypatch.setattr(fused_mlp, "POOL_TOP", mode)
        x = x0.clone().requires_grad_(True)
        layers = make_layers(K0, widths, seed=5)
        for l in layers:
            for t in l[:4]:
                t.requires_grad_(True)
        names = []

        def hook(name, phase, args):
            if phase == "pre":
                names.append((name, args))
        _lib._hooks.append(hook)
        try:
            out = fused_mlp.mlp_stack(x, S, True, True, 0.9, EPS, True, [tuple(l) for l in layers])
            if go is None:
                go = torch.randn(out.shape, generator=g).to(DEV)
            out.backward(go)
        finally:
            _lib._hooks.remove(hook)
        called = [n for n, _ in names]
        if mode:
            for want in ("pcops_mlp_gemm_dgrad_top", "pcops_mlp_gram", "pcops_mlp_pool_top_addend",
                         "pcops_mlp_pool_top_wsparse"):
                assert want in called, called
            pooled = [a for n, a in names if n == "pcops_mlp_gemm_fwd_pool"]
            assert pooled and pooled[-1][11] is None          # Y: not stored
        else:
            assert "pcops_mlp_gemm_dgrad_top" not in called
        res[mode] = [out.detach().clone(), x.grad.clone()] + [t.grad.clone() for l in layers for t in l[:4]]
    assert torch.equal(res[True][0], res[False][0])           # same forward kernel, with and without the Y store
    names = ["dx"] + ["L%d.%s" % (i, n) for i in range(len(widths)) for n in ("dW", "db", "dgamma", "dbeta")]
    for name, a, b in zip(names, res[True][1:], res[False][1:]):
        # the gradient of a bias in front of a batch norm is zero in exact arithmetic: both forms return rounding
        # noise there, judged on the scale of the upstream gradient instead of its own
        scale = go.abs().max().item() if name.endswith(".db") else b.abs().max().item()
        assert (a - b).abs().max().item() <= 2e-4 * scale + 1e-6, (name, (a - b).abs().max().item(), scale)


@pytest.mark.parametrize("R,K,N", [(256, 1024, 512), (256, 512, 256), (256, 256, 15), (7, 33, 5), (1, 64, 40),
                                   (300, 100, 130)])
def test_small_linear_forward_and_backward(R, K, N):
    """the classifier head's dense layers on pcops_small_gemm_ex (plain, transposed-B and transposed-A products) against
    float64"""
    g = torch.Generator().manual_seed(R * K + N)
    x = torch.randn(R, K, generator=g).to(DEV).requires_grad_(True)
    w = (torch.randn(K, N, generator=g) / K ** 0.5).to(DEV).requires_grad_(True)
    b = torch.randn(N, generator=g).to(DEV).requires_grad_(True)
    go = torch.randn(R, N, generator=g).to(DEV)
    y = fused_mlp.small_linear(x, w, b)
    y.backward(go)
    xd, wd, bd = (t.detach().double().requires_grad_(True) for t in (x, w, b))
    yd = xd @ wd + bd
    yd.backward(go.double())
    for got, want in ((y, yd), (x.grad, xd.grad), (w.grad, wd.grad), (b.grad, bd.grad)):
        scale = want.abs().max().item()
        assert (got.double() - want.detach()).abs().max().item() <= 3e-6 * scale * max(1.0, (max(R, K) / 64) ** 0.5)


GATHER_CASES = [  # (B, N, M, S, widths, pool)
    (4, 256, 64, 32, [64, 64, 128], True),      # SA1-like
    (3, 100, 37, 16, [128, 128, 256], True),    # SA2-like, ragged group count
    (2, 128, 128, 20, [64], True),              # EdgeConv: single pooled layer
    (2, 90, 30, 8, [32, 32, 64], True),         # MSG scale 0 widths
    (2, 64, 16, 4, [64, 128], False),           # un-pooled output
    # >= 32768 grouped rows: wave-stream kernels; in the xyz_bias form the first layer is never stored (rebuilt from
    # the row offsets in the next layer's forward, weight gradient and data-gradient mask)
    (8, 512, 256, 32, [64, 64, 128], True),
    (4, 512, 128, 64, [64, 128], False),
    (3, 700, 130, 96, [32, 64, 64], True),      # ragged tail tile, 3 tiles per pooling group
    (2, 1024, 168, 128, [64, 96, 128], True),   # MSG scale 2: arithmetic first layer into a 96-wide one, compacted rows
]


def gather_reference(Q, Ctr, xyz, new_xyz, wxyz, bias, idx, layers, pool, training, dtype):
    B, M, S = idx.shape
    ii = idx.long().reshape(B, M * S, 1)
    y = 0
    if Q is not None:
        y = y + torch.gather(Q.to(dtype), 1, ii.expand(-1, -1, Q.shape[2])).view(B, M, S, -1)
    if Ctr is not None:
        y = y + Ctr.to(dtype).unsqueeze(2)
    if wxyz is not None:
        g = torch.gather(xyz.to(dtype), 1, ii.expand(-1, -1, 3)).view(B, M, S, 3) - new_xyz.to(dtype).unsqueeze(2)
        y = y + g @ wxyz.to(dtype)
    if bias is not None:
        y = y + bias.to(dtype)
    a = None
    y = y.reshape(B * M * S, -1)
    for li, (W, b, gamma, beta, mm, mv) in enumerate(layers):
        if li > 0:
            y = a @ W.to(dtype) + b.to(dtype)
        if training:
            var, mean = torch.var_mean(y, dim=0, unbiased=False)
        else:
            mean, var = mm.to(dtype), mv.to(dtype)
        a = torch.relu((y - mean) * torch.rsqrt(var + EPS) * gamma.to(dtype) + beta.to(dtype))
    if pool:
        a = a.view(-1, S, a.shape[1]).amax(dim=1)
    return a


@pytest.mark.parametrize("form", ["q_ctr", "xyz_bias", "q_xyz"])
@pytest.mark.parametrize("B,N,M,S,widths,pool", GATHER_CASES)
def test_gather_stack_forward_backward(B, N, M, S, widths, pool, form):
    g = torch.Generator().manual_seed(B * 1000 + N)
    C1 = widths[0]
    src = {
        "Q": torch.randn(B, N, C1, generator=g).to(DEV) if form != "xyz_bias" else None,
        "Ctr": torch.randn(B, M, C1, generator=g).to(DEV) if form == "q_ctr" else None,
        "xyz": torch.rand(B, N, 3, generator=g).to(DEV) if form != "q_ctr" else None,
        "new_xyz": torch.rand(B, M, 3, generator=g).to(DEV) if form != "q_ctr" else None,
        "wxyz": torch.randn(3, C1, generator=g).to(DEV) if form != "q_ctr" else None,
        "bias": torch.randn(C1, generator=g).to(DEV) if form == "xyz_bias" else None,
    }
    idx = torch.randint(0, N, (B, M, S), generator=g, dtype=torch.int32).to(DEV)
    idx[:, :, S // 2:] = idx[:, :, :1]                     # ball-query style padding: duplicate rows
    layers = make_layers(C1, widths, seed=N)                # layers[0]'s W/b are unused by the gather form
    diff = ("Q", "Ctr", "wxyz", "bias")

    def call(mode, training, s):
        dt = torch.float64 if mode == "fp64" else torch.float32
        ls = [[t.detach().to(dt).requires_grad_(training) for t in l[:4]] + [l[4].clone(), l[5].clone()] for l in layers]
        if mode == "fused":
            out = fused_mlp.gather_mlp_stack(idx, pool, training, 0.9, EPS, True, [tuple(l) for l in ls],
                                             Q=s["Q"], Ctr=s["Ctr"], xyz=s["xyz"], new_xyz=s["new_xyz"],
                                             wxyz=s["wxyz"], bias=s["bias"])
        else:
            out = gather_reference(s["Q"], s["Ctr"], s["xyz"], s["new_xyz"], s["wxyz"], s["bias"], idx, ls, pool,
                                   training, dt)
        return out, ls

    for training in (True, False):
        out, _ = call("fused", training, src)
        want, _ = call("fp64", training, src)
        assert (out.double() - want).abs().max().item() < 1e-4

    _gather_backward_check(src, idx, layers, pool)


def _gather_backward_check(src, idx, layers, pool, go_seed=5, pts_cnt=None):
    """gradients of a gather-first stack w.r.t. Q / Ctr / wxyz / bias and every layer variable, against float64
    autograd evaluated with the activation pattern the kernels used"""
    diff = ("Q", "Ctr", "wxyz", "bias")
    B, M, S = idx.shape

    def leaves(dt):
        s = {k: (v.detach().to(dt).requires_grad_(k in diff) if v is not None else None) for k, v in src.items()}
        ls = [[t.detach().to(dt).requires_grad_(True) for t in l[:4]] + [l[4].clone(), l[5].clone()] for l in layers]
        return s, ls

    def grads(s, ls):
        res = [s[k].grad.double() for k in diff if s[k] is not None]
        for li, l in enumerate(ls):
            for ti, t in enumerate(l[:4]):
                if li == 0 and ti < 2:
                    continue                                  # layer-1 W/b live outside the gather form
                res.append(t.grad.double())
        return res

    s, ls = leaves(torch.float32)
    out = fused_mlp.gather_mlp_stack(idx, pool, True, 0.9, EPS, True, [tuple(l) for l in ls], Q=s["Q"], Ctr=s["Ctr"],
                                     xyz=s["xyz"], new_xyz=s["new_xyz"], wxyz=s["wxyz"], bias=s["bias"],
                                     pts_cnt=pts_cnt)
    if pts_cnt is not None:
        assert out.grad_fn.rows is not None, "the stack was expected to run on compacted rows"
    pattern = MR.fused_pattern(out)
    torch.manual_seed(go_seed)
    go = torch.randn(out.shape, device=DEV)
    out.backward(go)
    got = grads(s, ls)

    def run_ref(dt, report=None):
        s, ls = leaves(dt)
        y1 = MR.gather_first_layer(s["Q"], s["Ctr"], s["xyz"], s["new_xyz"], s["wxyz"], s["bias"], idx, dt)
        o = MR.run_stack(y1, None, ls, S, pool, True, dt, pattern, report)
        o.backward(go.to(dt))
        return o.detach(), grads(s, ls)

    rep = {}
    fwd64, want = run_ref(torch.float64, rep)
    MR.check_pattern(rep, B * M * S * sum(l[2].shape[0] for l in layers))
    assert (out.detach().double() - fwd64).abs().max().item() < 1e-4
    _, plain = run_ref(torch.float32)
    MR.assert_grads_close(["g%d" % i for i in range(len(got))], got, want, plain)
    del fwd64, want, plain
    torch.cuda.empty_cache()


# ---------------------------------------------------------------------------- compacted rows (ball-query padding left out)
COMPACT_CASES = [  # (B, N, M, S, radius, widths, form)
    (8, 512, 128, 64, 0.4, [128, 128, 256], "q_xyz"),       # SA2 of the SSG config
    (4, 1024, 256, 64, 0.2, [64, 64, 128], "xyz_bias"),     # SA1 of the BGA config (nsample 64): arithmetic first layer
    (8, 600, 100, 48, 0.3, [64, 128], "q_xyz"),             # 3 blocks per group, two layers
    (2, 2048, 128, 128, 0.25, [32, 64, 128], "xyz_bias"),   # MSG-sized groups
    (8, 512, 256, 64, 0.3, [64, 64, 128], "q_xyz"),         # narrow layers: one-pass data + weight gradient on compacted rows
    (4, 1024, 128, 128, 0.25, [64, 96, 128], "xyz_bias"),   # MSG's 96-wide scale (three-block tiles, weighted statistics)
    (4, 1024, 128, 64, 0.22, [64, 96], "q_xyz"),            # ... a pooled 96-wide top layer behind a stored first layer
    (8, 512, 128, 64, 2.5, [128, 128], "q_xyz"),            # every ball full: nothing to leave out, all weights 1
    (8, 512, 128, 64, 0.02, [128, 128], "q_xyz"),           # (almost) every ball holds the query alone: weights 49
]


@pytest.mark.parametrize("B,N,M,S,radius,widths,form", COMPACT_CASES)
def test_gather_stack_on_compacted_rows(B, N, M, S, radius, widths, form):
    """the grouped stack on the compacted row set (pcops.h "compacted rows": per group 16 * ceil(pts_cnt / 16) rows,
    the first row weighted for the copies left out) against the float64 reference of the FULL padded (b, m, S) tensor,
    forward and every gradient; geometry from the real FPS + ball query, so the padding is the real thing"""
    from scanobjectnn_amd.pointnet2 import tf_grouping, tf_sampling
    from scanobjectnn_amd.synth import synth_clouds
    g = torch.Generator().manual_seed(B * 100 + S)
    xyz = torch.from_numpy(synth_clouds(B, N, seed=B + S)).to(DEV)
    new_xyz = tf_sampling.gather_point(xyz, tf_sampling.farthest_point_sample(M, xyz))
    idx, cnt = tf_grouping.query_ball_point(radius, S, xyz, new_xyz)
    C1 = widths[0]
    src = {"Q": (0.5 * torch.randn(B, N, C1, generator=g)).to(DEV) if form == "q_xyz" else None, "Ctr": None,
           "xyz": xyz, "new_xyz": new_xyz, "wxyz": torch.randn(3, C1, generator=g).to(DEV),
           "bias": (0.1 * torch.randn(C1, generator=g)).to(DEV) if form == "xyz_bias" else None}
    layers = make_layers(C1, widths, seed=S)
    for training in (True, False):
        outs = []
        for pc in (cnt, None):
            ls = [[t.clone() for t in l] for l in layers]
            outs.append(fused_mlp.gather_mlp_stack(idx, True, training, 0.9, EPS, True, [tuple(l) for l in ls],
                                                   Q=src["Q"], xyz=xyz, new_xyz=new_xyz, wxyz=src["wxyz"],
                                                   bias=src["bias"], pts_cnt=pc))
            mov = [(l[4], l[5]) for l in ls]
            if pc is cnt:
                mov_c = mov
        y1 = MR.gather_first_layer(src["Q"], None, xyz, new_xyz, src["wxyz"], src["bias"], idx, torch.float64)
        want = MR.run_stack(y1, None, layers, S, True, training, torch.float64)
        assert (outs[0].double() - want).abs().max().item() < 1e-4
        assert (outs[0] - outs[1]).abs().max().item() < 2e-5             # compacted == uncompacted kernels
        if training:                                                        # ... and so are the moving statistics
            for (mm_c, mv_c), (mm_u, mv_u) in zip(mov_c, mov):
                assert torch.allclose(mm_c, mm_u, atol=1e-6) and torch.allclose(mv_c, mv_u, atol=1e-6, rtol=1e-5)
    _gather_backward_check(src, idx, layers, True, pts_cnt=cnt)


@pytest.mark.parametrize("R,K,N,bias", [(256 * 512, 128, 128, True), (3000, 64, 96, False), (70000, 32, 64, True)])
def test_rows_linear(R, K, N, bias):
    """Y = X W + b through libpcops (the per-source-point contraction of a grouped first layer) against torch in
    float64, forward and all three gradients"""
    g = torch.Generator().manual_seed(R + K)
    x = torch.randn(R, K, generator=g).to(DEV).requires_grad_(True)
    w = (torch.randn(K, N, generator=g) / K ** 0.5).to(DEV).requires_grad_(True)
    b = torch.randn(N, generator=g).to(DEV).requires_grad_(True) if bias else None
    y = fused_mlp.rows_linear(x, w, b)
    go = torch.randn(R, N, generator=g).to(DEV)
    y.backward(go)
    xd, wd = x.detach().double().requires_grad_(True), w.detach().double().requires_grad_(True)
    bd = b.detach().double().requires_grad_(True) if bias else None
    yd = xd @ wd + (bd if bias else 0.0)
    yd.backward(go.double())
    assert (y.double() - yd).abs().max().item() < 1e-4
    assert (x.grad.double() - xd.grad).abs().max().item() < 1e-4 * max(1.0, xd.grad.abs().max().item())
    assert (w.grad.double() - wd.grad).abs().max().item() < 1e-4 * wd.grad.abs().max().item()
    if bias:
        assert (b.grad.double() - bd.grad).abs().max().item() < 1e-4 * bd.grad.abs().max().item()


@pytest.mark.parametrize("M,K,N", [(131072 + 77, 64, 128), (65536, 128, 128), (40000, 128, 256), (65536, 320, 1024)])
def test_split_operand_forward_is_more_accurate_than_an_fp32_chain(M, K, N):
    """the forward product on the 16-bit matrix pipe (three bf16 pieces per operand, six products, the
    h.h products alone in the tile's accumulators) -- against float64: relative RMS error at most HALF of what a K-long
    fp32 fmaf chain makes on the same operands (measured: 0.35 ... 0.4 of it), no row off, and the statistics the epilogue
    emits are the sums of what it stored."""
    from scanobjectnn_amd import _lib
    lib = _lib.load()
    if _lib.get_option(_lib.OPT_GEMM_SPLIT_BF16) == 0:
        pytest.skip("split operands switched off")
    g = torch.Generator().manual_seed(M + K)
    X = torch.randn(M, K, generator=g).to(DEV)
    W = (torch.randn(K, N, generator=g) / K ** 0.5).to(DEV)
    b = torch.randn(N, generator=g).to(DEV)
    sc = (torch.rand(K, generator=g) + 0.5).to(DEV)
    sh = (0.3 * torch.randn(K, generator=g)).to(DEV)
    Y = torch.empty(M, N, device=DEV)
    P = lib.pcops_mlp_stats_rows(M)
    part = torch.zeros(P, 2, N, device=DEV)
    _lib.call("pcops_mlp_gemm_fwd", M, K, N, X.data_ptr(), K, sc.data_ptr(), sh.data_ptr(), W.data_ptr(), b.data_ptr(),
              Y.data_ptr(), part.data_ptr(), None)
    A32 = torch.relu(torch.addcmul(sh, X, sc))                 # the operand as the kernel forms it (one fma, then max)
    ref = A32.double() @ W.double() + b.double()
    scale = ref.pow(2).mean().sqrt().item()
    rms = ((Y.double() - ref).pow(2).mean().sqrt() / scale).item()
    # a K-long fp32 chain on the same operands (sequential accumulation, what the fp32 matrix pipe does)
    rows = slice(0, 4096)
    chain = b.unsqueeze(0).expand(4096, N).clone()
    for k in range(K):
        chain = torch.addcmul(chain, A32[rows, k:k + 1], W[k:k + 1, :])
    rms_chain = ((chain.double() - ref[rows]).pow(2).mean().sqrt() / scale).item()
    assert rms <= 0.5 * rms_chain, (rms, rms_chain)
    assert (Y.double() - ref).abs().max().item() <= 2e-5 * max(1.0, scale)
    s = part.double().sum(0)
    assert torch.allclose(s[0], Y.double().sum(0), rtol=0, atol=1e-6 * M ** 0.5 * scale + 1e-7 * Y.double().sum(0).abs().max().item())


@pytest.mark.parametrize("M,K,N", [(262144, 128, 128), (262144 + 45, 128, 256), (524288, 96, 128)])
def test_split_operand_weight_gradient_accuracy(M, K, N):
    """dW = X^T dY on the bf16 matrix pipe with split operands (wgrad_bf3_kernel; the producers hand
    the stripe over transposed and pre-split) against float64 -- measured 1.5e-7 ... 2.2e-7 relative RMS (the fp32
    producer / consumer kernel: 2.1e-7 ... 4.0e-7; torch's fp32 matmul of the same operands: 3.5e-6 ... 4.9e-6), and the
    bias gradient beside it."""
    from scanobjectnn_amd import _lib
    lib = _lib.load()
    g = torch.Generator().manual_seed(M + K)
    X = torch.relu(torch.randn(M, K, generator=g)).to(DEV)
    G = torch.randn(M, N, generator=g).to(DEV)
    Y = torch.randn(M, N, generator=g).to(DEV)
    one, zero = torch.ones(N, device=DEV), torch.zeros(N, device=DEV)        # dY = 1 . G + 0 . Y + 0
    splits = lib.pcops_mlp_wgrad_splits(M, K, N)
    scratch = torch.empty(splits * (K * N + N), device=DEV)
    dW, db = torch.empty(K, N, device=DEV), torch.empty(N, device=DEV)
    _lib.call("pcops_mlp_wgrad", M, K, N, X.data_ptr(), K, None, None, G.data_ptr(), Y.data_ptr(), one.data_ptr(),
              zero.data_ptr(), zero.data_ptr(), None, None, 1, None, None, scratch.data_ptr(), dW.data_ptr(), db.data_ptr())
    ref = X.double().t() @ G.double()
    scale = ref.pow(2).mean().sqrt().item()
    err = ((dW.double() - ref).pow(2).mean().sqrt() / scale).item()
    err_torch = (((X.t() @ G).double() - ref).pow(2).mean().sqrt() / scale).item()
    assert err <= 5e-7 and err <= 0.2 * err_torch, (err, err_torch)
    sums = G.double().sum(0)
    assert ((db.double() - sums).abs().max() / sums.abs().max()).item() <= 2e-6


# ---------------------------------------------------------------------------- [Q | Ctr] forms (round 5)
@pytest.mark.parametrize("cin,widths", [(3, [64]), (64, [64]), (64, [128]), (3, [64, 128])])
def test_edge_conv_stack_one_gemm_matches_two(cin, widths):
    """edge_conv_stack through ONE per-point product X [W_b | W_a - W_b] (pcops.h "[Q | Ctr] forms": strided Q / Ctr / dQ /
    dCtr in the edgeconv.hip kernels, concatenated weight by pcops_edge_weights_*) against the two-GEMM path it replaces:
    same variables, same output, same gradients w.r.t. the input and every variable, same moving statistics"""
    from scanobjectnn_amd.dgcnn import tf_util as D
    from scanobjectnn_amd.graph import Model
    B, N, k = 8, 1024, 20
    g = torch.Generator().manual_seed(cin * 10 + len(widths))
    x0 = torch.randn(B, N, cin, generator=g).to(DEV)
    nn_idx = D.knn_graph(x0, k=k)
    scopes = ['l%d' % i for i in range(len(widths))]

    def net(x, is_training, bn_decay=None):
        return D.edge_conv_stack(x, nn_idx, widths, scopes, is_training, bn_decay), {}

    assert fused_mlp.edge_qc_supported(B, N, k, widths[0])
    res = []
    for flag in (True, False):
        fused_mlp.EDGE_QC = flag
        try:
            m = Model(net, device=DEV, seed=5).build(x0)
            x = x0.clone().requires_grad_(True)
            y, _ = m(x, is_training=True, bn_decay=0.8)
            torch.manual_seed(1)
            go = torch.randn(y.shape, device=DEV)
            y.backward(go)
            grads = {n: p.grad.clone() for n, p in m.named_parameters()}
            res.append((y.detach(), x.grad.clone(), grads, {k_: v.clone() for k_, v in m.state_dict().items()}))
        finally:
            fused_mlp.EDGE_QC = True
    (y1, dx1, g1, s1), (y2, dx2, g2, s2) = res
    assert sorted(g1) == sorted(g2) and sorted(s1) == sorted(s2)
    assert (y1 - y2).abs().max().item() <= 2e-5 * max(1.0, y2.abs().max().item())
    assert (dx1 - dx2).abs().max().item() <= 1e-4 * max(1.0, dx2.abs().max().item())
    for n in g1:
        # a bias in front of BatchNorm has the exact gradient 0 (BN removes the mean): what either path returns for it is
        # the rounding residue of a sum over all rows, 1e-4-sized and different between ANY two summation orders
        tol = 1e-3 if n.endswith("biases") else 2e-4 * max(1.0, g2[n].abs().max().item())
        assert (g1[n] - g2[n]).abs().max().item() <= tol, n
    for n in s1:
        assert torch.allclose(s1[n], s2[n], atol=1e-5, rtol=1e-4), n


@pytest.mark.parametrize("B,N,k,widths", [(8, 1024, 20, [64, 128]), (4, 2048, 20, [64, 128]), (8, 1024, 16, [128, 64, 64])])
def test_edge_conv_first_layer_gradient_without_a_scatter(B, N, k, widths):
    """an EdgeConv stack on an input that needs NO gradient (DGCNN's T-Net on the raw cloud): the first layer's weight / bias
    gradient comes from ONE pass over the masked gradient of its output and the 27 edge moments (pcops.h pcops_edge_first_*)
    instead of the scatter to per-point gradients and the GEMM backward -- same output, same gradients of every variable,
    same moving statistics as the scatter path"""
    from scanobjectnn_amd.dgcnn import tf_util as D
    from scanobjectnn_amd.graph import Model
    g = torch.Generator().manual_seed(N + k)
    x0 = torch.randn(B, N, 3, generator=g).to(DEV)
    nn_idx = D.knn_graph(x0, k=k)
    scopes = ['l%d' % i for i in range(len(widths))]

    def net(x, is_training, bn_decay=None):
        return D.edge_conv_stack(x, nn_idx, widths, scopes, is_training, bn_decay), {}

    assert fused_mlp.edge_direct_supported(B, N, k, 3, widths[0], len(widths), x0)
    res = []
    for flag in (True, False):
        fused_mlp.EDGE_DIRECT = flag
        try:
            fused_mlp.TRACE = []
            m = Model(net, device=DEV, seed=5).build(x0)
            y, _ = m(x0, is_training=True, bn_decay=0.8)
            took = [bool(getattr(e, "direct", False)) for e in fused_mlp.TRACE if hasattr(e, "saved")]
            torch.manual_seed(1)
            go = torch.randn(y.shape, device=DEV)
            y.backward(go)
            res.append((y.detach(), {n: p.grad.clone() for n, p in m.named_parameters()},
                        {k_: v.clone() for k_, v in m.state_dict().items()}, took))
        finally:
            fused_mlp.EDGE_DIRECT = True
            fused_mlp.TRACE = None
    (y1, g1, s1, t1), (y2, g2, s2, t2) = res
    assert any(t1) and not any(t2)                                   # the two runs really took the two paths
    assert sorted(g1) == sorted(g2)
    assert torch.equal(y1, y2)                                       # the forward is the same computation
    for n in g1:
        # (a bias in front of BatchNorm has the exact gradient 0: both paths return rounding residue for it)
        tol = 1e-3 if n.endswith("biases") else 2e-4 * max(1.0, g2[n].abs().max().item())
        assert (g1[n] - g2[n]).abs().max().item() <= tol, (n, (g1[n] - g2[n]).abs().max().item(), g2[n].abs().max().item())
    for n in s1:
        assert torch.allclose(s1[n], s2[n], atol=1e-6, rtol=1e-5), n


def test_arithmetic_options_are_per_call_state_of_the_library():
    """pcops_set_option: the split-operand forward product can be switched per call -- the launcher reads
    the table at every call, reports the pipe it took, and both formulations agree to fp32 rounding"""
    from scanobjectnn_amd import _lib
    lib = _lib.load()
    M, K, N = 65536, 128, 128
    g = torch.Generator().manual_seed(3)
    X = torch.randn(M, K, generator=g).to(DEV)
    W = (torch.randn(K, N, generator=g) / K ** 0.5).to(DEV)
    outs, pipes = [], []
    prev = _lib.get_option(_lib.OPT_GEMM_SPLIT_BF16)
    try:
        for mode in (1, 0, 1):
            assert _lib.set_option(_lib.OPT_GEMM_SPLIT_BF16, mode) in (0, 1, 2)
            Y = torch.empty(M, N, device=DEV)
            _lib.call("pcops_mlp_gemm_fwd", M, K, N, X.data_ptr(), K, None, None, W.data_ptr(), None, Y.data_ptr(), None, None)
            pipes.append(int(lib.pcops_last_launch_pipe()))
            outs.append(Y)
    finally:
        _lib.set_option(_lib.OPT_GEMM_SPLIT_BF16, prev)
    assert pipes == [1, 0, 1]
    assert torch.equal(outs[0], outs[2])
    ref = X.double() @ W.double()
    for Y in outs:
        assert (Y.double() - ref).abs().max().item() < 2e-5
    assert lib.pcops_set_option(99, 1) < 0 and lib.pcops_set_option(_lib.OPT_KNN_F16_PREFILTER, 2) < 0
