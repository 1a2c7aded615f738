"""Batch-norm statistics as SHIFTED moments (reference semantics: `tf.nn.moments` /
fused batch norm compute the variance in two passes, `pointnet2/utils/tf_util.py:512-531`).

Every producer of forward statistics sums (y - pivot) and (y - pivot)^2 with pivot = the layer's moving mean and
`pcops_mlp_bn_finalize` undoes the shift in fp64 (include/pcops.h, pcops_mlp_gemm_fwd).  The regime that broke the
one-pass form  E[y^2] - mean^2  on fp32 partial sums is |mean| >> std; here every pre-BN channel has |mean| = 30 std.

  * with a warm pivot (moving mean within a standard deviation of the batch mean -- any trained or training network
    after its first steps) the fused stacks hold the same 1e-4 against float64 as in the benign regime, on every
    producer: tiled GEMM, wave-stream GEMM (+ pooled epilogue, + streamed weights), gather first layer (all three
    forms), compacted rows, EdgeConv pooled layer;
  * the shift is algebraically neutral: ANY pivot gives the same statistics up to rounding -- checked with a cold pivot
    (zeros) and a deliberately wrong one, at the looser tolerance the unshifted sums always had;
  * the batch mean / variance written to the moving buffers are right to 1e-6 relative in the warm case.
"""
import pytest
import torch

import mlp_ref as MR
from scanobjectnn_amd import fused_mlp

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
EPS = 1e-3
BIG = 30.0


def _layers(k0, widths, seed):
    g = torch.Generator().manual_seed(seed)
    layers, cin = [], k0
    for w in widths:
        W = (torch.randn(cin, w, generator=g) / cin ** 0.5).to(DEV)
        sign = 1.0 - 2.0 * (torch.arange(w) % 2)
        b = (BIG * sign * (1.0 + 0.1 * torch.rand(w, generator=g))).to(DEV)     # |mean| ~ 30 std (std of x W is ~1)
        gamma = ((0.5 + torch.rand(w, generator=g)) * (1.0 - 2.0 * (torch.arange(w) % 3 == 2))).to(DEV)
        beta = (0.2 * torch.randn(w, generator=g)).to(DEV)
        layers.append([W, b, gamma, beta, torch.zeros(w, device=DEV), torch.ones(w, device=DEV)])
        cin = w
    return layers


def _truth(y1, x, layers, S, pool):
    """float64 chain; returns (output, [(mean, biased var, rows)] per layer)"""
    a, stats = None, []
    for li, (W, b, gamma, beta, _mm, _mv) in enumerate(layers):
        y = y1.double() if (li == 0 and y1 is not None) else (x.double() if a is None else a) @ W.double() + b.double()
        var, mean = torch.var_mean(y, dim=0, unbiased=False)
        stats.append((mean, var, y.shape[0]))
        a = torch.relu((y - mean) * torch.rsqrt(var + EPS) * gamma.double() + beta.double())
    if pool:
        a = a.view(-1, S, a.shape[1]).amax(dim=1)
    return a, stats


def _set_pivots(layers, stats, kind, seed=0):
    g = torch.Generator().manual_seed(seed)
    for l, (mean, var, _n) in zip(layers, stats):
        if kind == "warm":       # within a standard deviation of the batch mean, like a moving mean a few steps in
            l[4] = (mean + 0.7 * var.sqrt() * torch.randn(mean.shape, generator=g).to(DEV).double()).float()
        elif kind == "cold":     # a freshly initialised network
            l[4] = torch.zeros_like(mean).float()
        else:                    # "wrong": far on the other side
            l[4] = (-0.5 * mean).float()
        l[5] = torch.ones_like(mean).float()


DENSE = [  # (R, S, K0, widths, pool)
    (777, 1, 128, [128, 64], False),                 # tiled kernel
    (4 * 128, 128, 64, [256, 512], True),            # tiled kernel, pooled
    (512 * 64 + 37, 1, 128, [128, 256], False),      # wave stream, ragged tail
    (1024 * 32 * 2, 32, 32, [64, 128], True),        # wave stream, pooling fused into the epilogue
    (256 * 128, 128, 260, [256, 512, 1024], True),   # streamed weights, pooled top layer without a stored Y
]


@pytest.mark.parametrize("R,S,K0,widths,pool", DENSE)
def test_dense_stack_with_large_channel_means(R, S, K0, widths, pool):
    g = torch.Generator().manual_seed(R + K0)
    x = torch.randn(R, K0, generator=g).to(DEV)
    layers = _layers(K0, widths, seed=K0)
    want, stats = _truth(None, x, layers, S, pool)
    errs = {}
    for kind in ("warm", "cold", "wrong"):
        _set_pivots(layers, stats, kind)
        piv = [l[4].clone() for l in layers]
        out = fused_mlp.mlp_stack(x, S, pool, True, 0.5, EPS, False, [tuple(l) for l in layers])
        errs[kind] = (out.double() - want).abs().max().item()
        if kind == "warm":
            for l, p0, (mean, var, _n) in zip(layers, piv, stats):      # moving <- 0.5 moving + 0.5 batch
                bm = 2.0 * l[4].double() - p0.double()
                bv = 2.0 * l[5].double() - 1.0
                # (the buffers are fp32 and this undoes a 0.5 / 0.5 blend: 4 ulp of |mean| ~ 30 on top)
                assert ((bm - mean).abs() - 1.5e-6 * mean.abs()).max().item() <= 1e-5 * var.sqrt().min().item()
                assert ((bv - var).abs() / var).max().item() <= 2e-5
    assert errs["warm"] <= 1e-4, errs
    assert errs["cold"] <= 2e-2 and errs["wrong"] <= 2e-2, errs        # neutral shift: still the same statistics


GATHER = [  # (B, N, M, S, widths, pool, form)
    (4, 256, 64, 32, [64, 64, 128], True, "q_ctr"),
    (3, 100, 37, 16, [128, 128], True, "q_xyz"),
    (8, 512, 256, 32, [64, 64, 128], True, "xyz_bias"),      # arithmetic first layer (never stored) + wave stream
    (4, 512, 128, 64, [64, 128], False, "q_ctr"),
    (2, 128, 128, 20, [64], True, "q_ctr"),                  # EdgeConv pooled layer (Q and Ctr both far from zero)
    (4, 300, 300, 20, [128], True, "q_ctr"),
]


@pytest.mark.parametrize("B,N,M,S,widths,pool,form", GATHER)
def test_gather_and_edgeconv_with_large_channel_means(B, N, M, S, widths, pool, form):
    g = torch.Generator().manual_seed(B * 1000 + N)
    C1 = widths[0]
    sign = (1.0 - 2.0 * (torch.arange(C1) % 2))
    Q = Ctr = xyz = new_xyz = wxyz = bias = None
    if form != "xyz_bias":
        # the neighbour term and the centre term are BOTH large and partly cancel: y = q + ctr has |mean| = 30 std
        Q = (torch.randn(B, N, C1, generator=g) + 80.0 * sign).to(DEV)
    if form == "q_ctr":
        Ctr = (0.5 * torch.randn(B, M, C1, generator=g) - 50.0 * sign).to(DEV)
    if form != "q_ctr":
        xyz = torch.rand(B, N, 3, generator=g).to(DEV)
        new_xyz = torch.rand(B, M, 3, generator=g).to(DEV)
        wxyz = torch.randn(3, C1, generator=g).to(DEV)
    if form == "xyz_bias":
        bias = (BIG * sign).to(DEV)
    idx = torch.randint(0, N, (B, M, S), generator=g, dtype=torch.int32).to(DEV)
    layers = _layers(C1, widths, seed=N)
    y1 = MR.gather_first_layer(Q, Ctr, xyz, new_xyz, wxyz, bias, idx, torch.float64)
    want, stats = _truth(y1, None, layers, S, pool)
    assert (stats[0][0].abs() / stats[0][1].sqrt()).min().item() > 10.0      # the regime under test
    errs = {}
    for kind in ("warm", "cold"):
        _set_pivots(layers, stats, kind)
        out = fused_mlp.gather_mlp_stack(idx, pool, True, 0.5, EPS, False, [tuple(l) for l in layers],
                                         Q=Q, Ctr=Ctr, xyz=xyz, new_xyz=new_xyz, wxyz=wxyz, bias=bias)
        errs[kind] = (out.double() - want).abs().max().item()
    assert errs["warm"] <= 1e-4, errs
    assert errs["cold"] <= 2e-2, errs


def test_compacted_rows_with_large_channel_means():
    from scanobjectnn_amd.pointnet2 import tf_grouping, tf_sampling
    from scanobjectnn_amd.synth import synth_clouds
    B, N, M, S, widths = 8, 512, 128, 64, [128, 128, 256]
    g = torch.Generator().manual_seed(5)
    xyz = torch.from_numpy(synth_clouds(B, N, seed=3)).to(DEV)
    new_xyz = tf_sampling.gather_point(xyz, tf_sampling.farthest_point_sample(M, xyz))
    idx, cnt = tf_grouping.query_ball_point(0.4, S, xyz, new_xyz)
    C1 = widths[0]
    sign = (1.0 - 2.0 * (torch.arange(C1) % 2))
    Q = (0.5 * torch.randn(B, N, C1, generator=g) + BIG * sign).to(DEV)
    wxyz = torch.randn(3, C1, generator=g).to(DEV)
    layers = _layers(C1, widths, seed=S)
    y1 = MR.gather_first_layer(Q, None, xyz, new_xyz, wxyz, None, idx, torch.float64)
    want, stats = _truth(y1, None, layers, S, True)
    _set_pivots(layers, stats, "warm")
    out = fused_mlp.gather_mlp_stack(idx, True, True, 0.5, EPS, False, [tuple(l) for l in layers], Q=Q, xyz=xyz,
                                     new_xyz=new_xyz, wxyz=wxyz, pts_cnt=cnt)
    assert (out.double() - want).abs().max().item() <= 1e-4
