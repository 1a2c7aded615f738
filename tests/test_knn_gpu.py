"""DGCNN graph ops (HIP, through the C ABI) against the CPU oracle: bit-exact indices, exact
distances (same fmaf chains), exact edge features; gradient of get_edge_feature to 1e-4."""
import numpy as np
import pytest
import torch

from oracle import oracle as O
from scanobjectnn_amd.dgcnn import tf_util as dg
from scanobjectnn_amd.synth import synth_clouds

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def N(t):
    return t.detach().cpu().numpy()


@pytest.mark.parametrize("b,n,c,k", [(2, 2048, 3, 20), (2, 512, 64, 20), (1, 300, 64, 20), (3, 100, 7, 5),
                                     (2, 64, 128, 20), (1, 21, 3, 20), (2, 257, 16, 33)])
def test_knn_graph_vs_oracle(b, n, c, k):
    rng = np.random.default_rng(n + c)
    x = synth_clouds(b, n, seed=n) if c == 3 else rng.standard_normal((b, n, c)).astype(np.float32)
    if n > 30:
        x[0, 17] = x[0, 3]                                     # exact duplicate -> distance tie at 0
    nn = dg.knn_graph(T(x), k=k)
    np.testing.assert_array_equal(N(nn), O.knn_graph(x, k))


@pytest.mark.parametrize("b,n,c,k", [(2, 2048, 3, 20), (2, 1024, 64, 20), (1, 300, 64, 20), (2, 200, 128, 20), (2, 257, 16, 33)])
@pytest.mark.parametrize("kind", ["previous", "random", "far", "lattice", "duplicates", "out_of_range"])
def test_seeded_knn_graph_is_the_same_graph(b, n, c, k, kind, monkeypatch):
    """pcops_knn_graph_seeded: a hint (k distinct points per query) that starts every query from an upper bound of its
    k-th distance.  Whatever the hint -- the graph of slightly different features (DGCNN: the previous layer), random
    points, the k FARTHEST points, a tie-ridden lattice -- the result is bit for bit the unseeded graph / the oracle's."""
    rng = np.random.default_rng(n * 7 + c)
    x = rng.standard_normal((b, n, c)).astype(np.float32)
    if kind == "lattice":
        x = (rng.integers(0, 3, (b, n, c)) * 0.5).astype(np.float32)            # exact distance ties everywhere
    want = O.knn_graph(x, k)
    if kind == "previous":
        seed = O.knn_graph((x + 0.05 * rng.standard_normal(x.shape)).astype(np.float32), k)
    elif kind == "far":
        d = ((x[:, :, None, :].astype(np.float64) - x[:, None, :, :]) ** 2).sum(-1) if n <= 1024 else None
        seed = (np.argsort(-d, axis=2)[:, :, :k] if d is not None
                else np.stack([np.stack([rng.permutation(n)[:k] for _ in range(n)]) for _ in range(b)])).astype(np.int32)
    else:
        seed = np.stack([np.stack([rng.permutation(n)[:k] for _ in range(n)]) for _ in range(b)]).astype(np.int32)
    if kind == "duplicates":        # a row that repeats an index names < k distinct points -- its bound is void and
        seed[:, ::2, 1] = seed[:, ::2, 0]                        # must be IGNORED, not trusted (every other query, nearest twice)
        seed[:, ::2, :] = O.knn_graph(x, k)[:, ::2, :1]          # ... and the whole row = the nearest point, k times
    if kind == "out_of_range":
        seed[:, ::3, k // 2] = n + 5
        seed[:, 1::3, 0] = -1
    monkeypatch.setattr(dg, "KNN_SEED_MAX_C", 1 << 20)          # the wrapper only takes the hint where it pays ...
    monkeypatch.setattr(dg, "KNN_SEED_FORCE", True)             # ... here every seeded kernel variant is exercised
    nn = dg.knn_graph(T(x), k=k, seed=T(seed.astype(np.int32)))
    np.testing.assert_array_equal(N(nn), want)
    assert torch.equal(nn, dg.knn_graph(T(x), k=k))


@pytest.mark.parametrize("case", ["huge", "nan_free_mixed", "tiny", "ties", "subnormal_mix"])
def test_knn_graph_fp16_filter_edge_inputs(case):
    """the 64-channel graph kernel pre-filters in fp16 (csrc/knn.hip knn_f16_kernel): features beyond the fp16 range, far
    below its normal range, wildly mixed magnitudes and exact ties must all give the oracle's indices -- the filter only
    ever REJECTS on a rigorous bound and switches itself off where the bound does not hold"""
    rng = np.random.default_rng(17)
    b, n, c, k = 2, 512, 64, 20
    x = rng.standard_normal((b, n, c)).astype(np.float32)
    if case == "huge":
        x *= 3.0e5                                               # |x| > 65504: fp16 overflows -> the filter turns itself off
    elif case == "nan_free_mixed":
        x[:, ::7] *= 1.0e5                                       # a few far outliers among O(1) points
        x[:, 1::7] *= 1.0e-6
    elif case == "tiny":
        x *= 1.0e-6                                              # fp16 subnormals / flush to zero
    elif case == "subnormal_mix":
        # the filter's error bound assumes GRADUAL underflow in the float -> half conversion and in the fp16 MFMA.
        # Channels are either fp16-subnormal (3e-5 .. 6e-5) or small (2e-3 .. 1e-2), all positive: were subnormals flushed,
        # a subnormal channel of one point against a small channel of another would move a distance by ~3e-7 per channel
        # with one sign -- ~1e-5 over the row, five times the bound A (s_i + s_j) + B -- and the graph would lose neighbours
        n = 2048
        tiny = rng.uniform(3.0e-5, 6.0e-5, (b, n, c))
        small = rng.uniform(2.0e-3, 1.0e-2, (b, n, c))
        x = np.where(rng.random((b, n, c)) < 0.5, tiny, small).astype(np.float32)
    else:
        x = (rng.integers(0, 2, (b, n, c)) * 0.5).astype(np.float32)   # many exactly equal distances
    nn = dg.knn_graph(T(x), k=k)
    np.testing.assert_array_equal(N(nn), O.knn_graph(x, k))


@pytest.mark.parametrize("c", [3, 64, 16])
def test_knn_graph_zero_and_negative_distances(c):
    """round 5: the sorted lists are 64-bit keys ordered as doubles (csrc/knn.hip TopKey).  Exercised here: distance exactly
    +0 (every point to itself, and exact duplicates -> the index decides), distances that come out NEGATIVE from
    (s_i - 2 <x_i, x_j>) + s_j (near-duplicates at a large offset: the index order reverses inside a negative double unless it
    is complemented), fp32-denormal distances, and a cloud with fewer distinct points than k"""
    rng = np.random.default_rng(c)
    b, n, k = 2, 640, 20
    base = rng.standard_normal((b, n // 4, c)).astype(np.float32) + np.float32(50.0)
    x = np.concatenate([base, base,                                                  # exact duplicates
                        np.nextafter(base, np.float32(np.inf)),                      # one ulp away: rounding decides the sign
                        np.nextafter(base, np.float32(-np.inf))], axis=1).astype(np.float32)
    x = x[:, rng.permutation(n)]
    ref = O.knn_graph(x, k)
    d = O.pairwise_distance(x[:1])[0]
    assert (d < 0).any() and (d == 0).any()                                          # the case is what it claims to be
    np.testing.assert_array_equal(N(dg.knn_graph(T(x), k=k)), ref)
    tiny = (rng.standard_normal((b, 512, c)) * 1e-21).astype(np.float32)             # squared distances ~1e-42: fp32 denormals
    np.testing.assert_array_equal(N(dg.knn_graph(T(tiny), k=k)), O.knn_graph(tiny, k))
    few = np.repeat(rng.standard_normal((b, 4, c)).astype(np.float32), 64, axis=1)   # 4 distinct points, 64 copies each
    np.testing.assert_array_equal(N(dg.knn_graph(T(few), k=k)), O.knn_graph(few, k))


def test_knn_graph_4d_input_and_lattice_ties():
    rng = np.random.default_rng(0)
    x = (rng.integers(0, 4, (2, 200, 1, 3)) * 0.25).astype(np.float32)     # heavy ties
    nn = dg.knn_graph(T(x), k=20)
    np.testing.assert_array_equal(N(nn), O.knn_graph(x[:, :, 0, :], 20))


@pytest.mark.parametrize("b,n,c", [(2, 300, 3), (1, 130, 64), (2, 64, 40)])
def test_materialised_path_vs_oracle_and_fused(b, n, c):
    rng = np.random.default_rng(c)
    x = rng.standard_normal((b, n, c)).astype(np.float32)
    adj = dg.pairwise_distance(T(x))
    np.testing.assert_array_equal(N(adj), O.pairwise_distance(x))
    nn = dg.knn(adj, k=20)
    np.testing.assert_array_equal(N(nn), O.knn(N(adj), 20))
    assert torch.equal(nn, dg.knn_graph(T(x), k=20))


def test_edge_feature_and_grad():
    rng = np.random.default_rng(1)
    for (b, n, c, k) in [(2, 128, 3, 20), (2, 100, 64, 20), (1, 50, 5, 7)]:
        x = rng.standard_normal((b, n, c)).astype(np.float32)
        nn = rng.integers(0, n, (b, n, k)).astype(np.int32)
        xt = T(x).requires_grad_(True)
        ef = dg.get_edge_feature(xt, T(nn), k=k)
        np.testing.assert_array_equal(N(ef), O.get_edge_feature(x, nn, k))
        go = rng.standard_normal(tuple(ef.shape)).astype(np.float32)
        ef.backward(T(go))
        # dense fp64 reference of the gradient
        gx = np.zeros((b, n, c))
        ga, gb = go[..., :c].astype(np.float64), go[..., c:].astype(np.float64)
        gx += (ga - gb).sum(axis=2)
        for bi in range(b):
            np.add.at(gx[bi], nn[bi].reshape(-1), gb[bi].reshape(-1, c))
        np.testing.assert_allclose(N(xt.grad), gx, rtol=1e-5, atol=1e-4)


def test_knn_argument_errors():
    x = T(np.zeros((1, 10, 3), np.float32))
    with pytest.raises(ValueError):
        dg.knn_graph(x, k=11)
    with pytest.raises(ValueError):
        dg.knn_graph(x, k=0)
