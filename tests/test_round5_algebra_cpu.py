"""CPU statements of the algebra behind round 5's kernels (no GPU, no library compute calls):
  * the 64-bit list keys of the kNN kernels (csrc/knn.hip TopKey): as IEEE doubles they order candidates exactly by
    (distance, then lower index) -- for negative, zero and fp32-denormal distances too -- and never collide;
  * the scatter-free weight gradient of a first EdgeConv layer whose input needs no gradient (pcops.h pcops_edge_first_*,
    DESIGN.md section 4.14):  dW = p E^T Gm + q (E^T E W + E^T 1 b) + t E^T 1  against autograd of the layer itself;
  * the per-cloud / per-point split of a conv on a broadcast concatenation;
  * which pooling group sizes the GEMM epilogue takes (host logic of pcops_mlp_gemm_fwd_pool_supported)."""
import numpy as np
import pytest
import torch

IDX_MASK, BIAS_HI, SENT_HI = 0x1FFFFFFF, 0x20000000, 0x7FE00000


def make_key(d, j):
    """TopKey::make for a finite fp32 distance d and an index j < 2^29, as the float64 the kernel compares"""
    bits = np.float64(np.float32(d)).view(np.int64)
    hi, lo = np.int64(bits >> 32), np.int64(bits & 0xFFFFFFFF)
    t = np.int64(-1) if hi < 0 else np.int64(0)
    lo = lo | ((np.int64(j) ^ t) & IDX_MASK)
    hi = hi + BIAS_HI
    return np.array([(int(hi) << 32) | int(lo & 0xFFFFFFFF)], dtype=np.int64).view(np.float64)[0]


def key_value_index(key):
    bits = np.array([key], dtype=np.float64).view(np.int64)[0]
    hi, lo = int(bits >> 32), int(bits & 0xFFFFFFFF)
    d = np.array([((hi - BIAS_HI) << 32) | (lo & ~IDX_MASK & 0xFFFFFFFF)], dtype=np.int64).view(np.float64)[0]
    t = -1 if hi < 0 else 0
    return np.float32(d), (lo ^ t) & IDX_MASK


def test_knn_list_keys_order_like_distance_then_index():
    rng = np.random.default_rng(0)
    specials = [0.0, 1e-45, -1e-45, 1.5e-39, -3e-40, 1.0, -1.0, np.float32(3.4e38), -np.float32(1e-3), 2.0 ** -126]
    ds = np.concatenate([np.array(specials, dtype=np.float32), rng.standard_normal(300).astype(np.float32) * 1e-3,
                         np.abs(rng.standard_normal(300)).astype(np.float32) * 50]).astype(np.float32)
    ds = np.concatenate([ds, ds[:100]])                               # repeated distances: the index decides
    js = rng.permutation(len(ds)).astype(np.int64) * 1000 + 7         # distinct indices, up to ~7e5
    keys = np.array([make_key(d, j) for d, j in zip(ds, js)])
    assert np.isfinite(keys).all() and len(np.unique(keys)) == len(keys)
    assert (np.abs(keys) >= np.finfo(np.float64).tiny).all()          # no fp64 denormal among them (d = 0 included)
    want = sorted(range(len(ds)), key=lambda i: (float(ds[i]), int(js[i])))
    got = list(np.argsort(keys, kind="stable"))
    assert got == want
    sentinel = np.array([SENT_HI << 32], dtype=np.int64).view(np.float64)[0]
    assert (keys < sentinel).all()
    for d, j, k in zip(ds[:40], js[:40], keys[:40]):                  # the round trip the kernel's value() / index() make
        dv, jv = key_value_index(k)
        assert dv == d and jv == j
    # the sorted insertion  L[s] <- min(L[s], max(L[s-1], x))  keeps the k smallest keys in order
    K = 20
    L = np.full(K, sentinel)
    for x in keys:
        new = L.copy()
        for s in range(K - 1, 0, -1):
            new[s] = min(L[s], max(L[s - 1], x))
        new[0] = min(L[0], x)
        L = new
    assert list(L) == list(np.sort(keys)[:K])


def test_first_edge_layer_weight_gradient_without_a_scatter():
    torch.manual_seed(0)
    B, N, k, C = 3, 40, 5, 8
    x = torch.randn(B, N, 3, dtype=torch.float64)
    idx = torch.randint(0, N, (B, N, k))
    W = torch.randn(6, C, dtype=torch.float64, requires_grad=True)
    b = torch.randn(C, dtype=torch.float64, requires_grad=True)
    gamma, beta = torch.rand(C, dtype=torch.float64) + 0.5, torch.randn(C, dtype=torch.float64)
    xg = x.unsqueeze(2).expand(B, N, k, 3)
    xj = torch.gather(x.unsqueeze(1).expand(B, N, N, 3), 2, idx.unsqueeze(-1).expand(B, N, k, 3))
    E = torch.cat([xg, xj - xg], dim=-1).reshape(-1, 6)               # the six edge channels of every row
    Y = E @ W + b
    mean, var = Y.mean(0), Y.var(0, unbiased=False)
    rstd = torch.rsqrt(var + 1e-3)
    A = torch.relu((Y - mean) * rstd * gamma + beta)
    G_up = torch.randn_like(A)
    (A * G_up).sum().backward()
    # the kernel's inputs: the masked upstream gradient and the BN backward's coefficients dY = p Gm + q Y + t
    R = Y.shape[0]
    Gm = (G_up * (A > 0)).detach()
    Yd = Y.detach()
    yhat = (Yd - mean.detach()) * rstd.detach()
    s1, s2 = Gm.sum(0), (Gm * yhat).sum(0)
    p = gamma * rstd.detach()
    q = -p * s2 / R * rstd.detach()
    t = -p * s1 / R - q * mean.detach()
    dY = p * Gm + q * Yd + t
    M, S = E.T @ E, E.sum(0)
    dW = p * (E.T @ Gm) + q * (M @ W.detach() + torch.outer(S, b.detach())) + t * S.unsqueeze(1)
    db = p * s1 + q * Yd.sum(0) + t * R
    assert torch.allclose(dY.sum(0), db)
    assert torch.allclose(dW, W.grad, rtol=1e-9, atol=1e-10)
    assert torch.allclose(db, b.grad, rtol=1e-9, atol=1e-9)


def test_conv_on_a_broadcast_concatenation_splits_into_per_cloud_and_per_point_products():
    torch.manual_seed(1)
    B, N, Cc, Cx, Co = 4, 16, 10, 6, 5
    c, xp = torch.randn(B, Cc, dtype=torch.float64), torch.randn(B, N, Cx, dtype=torch.float64)
    W, bias = torch.randn(Cc + Cx, Co, dtype=torch.float64), torch.randn(Co, dtype=torch.float64)
    concat = torch.cat([c.unsqueeze(1).expand(B, N, Cc), xp], dim=-1)
    want = concat @ W + bias
    got = xp @ W[Cc:] + (c @ W[:Cc] + bias).unsqueeze(1)
    assert torch.allclose(got, want, rtol=1e-12, atol=1e-12)


def test_pooled_epilogue_group_sizes():
    """host logic only (no launch): groups of whole 32-row tiles as before, and -- round 5 -- groups that are multiples of four rows
    whose least common multiple with 32 is at most 256 rows (the T-Net's 20 neighbours, MSG's nsample 16)"""
    from scanobjectnn_amd import _lib
    lib = _lib.load()
    ok = lambda M, S: bool(lib.pcops_mlp_gemm_fwd_pool_supported(M, 64, 128, S))       # noqa: E731
    assert ok(160 * 1024, 32) and ok(160 * 1024, 64)
    assert ok(160 * 1024, 20) and ok(32 * 4096, 16) and ok(96 * 2048, 48) and ok(96 * 2048, 12)
    assert not ok(160 * 1024 + 32, 20)                               # not whole walks of lcm(20, 32) = 160 rows
    assert not ok(22 * 8192, 22) and not ok(36 * 8192, 36)           # 22: not a multiple of four; 36: lcm 288 > 256
    assert not ok(4 * 65536, 4)                                      # below eight rows per group
