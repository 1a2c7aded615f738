"""Parity of the HIP path (through the C ABI of libpcops.so) against the CPU oracle on seeded
inputs and against the committed golden vectors.  Integer outputs bit-exact; copies ==;
atomically-reduced gradients within 1e-4 (summation order is free); float outputs produced in
the oracle's own order ==.
"""
import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle import oracle as O
from scanobjectnn_amd import _lib
from scanobjectnn_amd.pointnet2 import tf_grouping, tf_interpolate, tf_sampling
from scanobjectnn_amd.synth import synth_clouds

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def N(t):
    return t.detach().cpu().numpy()


def lattice(b, n, seed, side=5, step=0.125):
    rng = np.random.default_rng(seed)
    return (rng.integers(0, side, (b, n, 3)) * step - 0.25).astype(np.float32)


# ------------------------------------------------------------------ ball query
@pytest.mark.parametrize("case", sorted(load_golden("query_ball_point")))
def test_query_ball_point_golden(case):
    g = load_golden("query_ball_point")[case]
    idx, cnt = tf_grouping.query_ball_point(float(g["radius"]), int(g["nsample"]), T(g["xyz1"]), T(g["xyz2"]))
    np.testing.assert_array_equal(N(idx), g["idx"])
    _, ocnt = O.query_ball_point(float(g["radius"]), int(g["nsample"]), g["xyz1"], g["xyz2"])
    np.testing.assert_array_equal(N(cnt), ocnt)


QBP_SHAPES = [  # (B, N, M, radius, nsample, kind)  -- every tuple of SURVEY Appendix B + odd sizes
    (4, 2048, 512, 0.2, 32, "surface"), (4, 2048, 512, 0.2, 64, "ball"), (4, 512, 128, 0.4, 64, "surface"),
    (2, 4096, 512, 0.1, 16, "surface"), (2, 4096, 512, 0.4, 128, "ball"), (2, 512, 128, 0.8, 128, "surface"),
    (3, 1000, 77, 0.3, 20, "ball"), (1, 5000, 300, 0.15, 48, "surface"), (2, 100, 1, 0.5, 7, "ball"),
    (1, 1, 1, 0.1, 3, "ball"), (2, 63, 65, 0.3, 1, "ball"),
]


@pytest.mark.parametrize("B,Nn,M,r,S,kind", QBP_SHAPES)
def test_query_ball_point_vs_oracle(B, Nn, M, r, S, kind):
    c = synth_clouds(B, Nn, seed=B * 1000 + Nn, kind=kind)
    rng = np.random.default_rng(Nn + M)
    q = np.stack([c[b, rng.permutation(Nn)[:M] if M <= Nn else rng.integers(0, Nn, M)] for b in range(B)])
    idx, cnt = tf_grouping.query_ball_point(r, S, T(c), T(q))
    oidx, ocnt = O.query_ball_point(r, S, c, q)
    np.testing.assert_array_equal(N(idx), oidx)
    np.testing.assert_array_equal(N(cnt), ocnt)


def test_query_ball_point_lattice_ties_and_foreign_queries():
    c = lattice(3, 700, 11)
    q = lattice(3, 90, 12) + np.float32(0.0625)
    for r in (0.125, 0.25, 0.2165, 0.0625, 1e-21, 1e30):
        idx, cnt = tf_grouping.query_ball_point(r, 24, T(c), T(q))
        oidx, ocnt = O.query_ball_point(r, 24, c, q)
        np.testing.assert_array_equal(N(idx), oidx)
        np.testing.assert_array_equal(N(cnt), ocnt)


def test_query_ball_point_radius_boundary_is_sqrt_exact():
    """d == r must be OUTSIDE (sqrtf(d2) < r), including where fl(r*r) would let it in."""
    rng = np.random.default_rng(3)
    for r in (0.1, 0.2, 0.3, 0.4, 0.8):
        r32 = np.float32(r)
        # points at distances straddling r along x: t such that sqrt(t) crosses r
        t = np.float32(r32 * r32)
        cands = [np.nextafter(t, np.float32(0)), t, np.nextafter(t, np.float32(1))]
        xs = np.array([np.sqrt(np.float64(v)) for v in cands], dtype=np.float32)
        xs = np.concatenate([xs, np.float32(r) + rng.normal(0, 1e-7, 61).astype(np.float32)])
        pts = np.zeros((1, xs.size, 3), np.float32)
        pts[0, :, 0] = xs
        q = np.zeros((1, 1, 3), np.float32)
        idx, cnt = tf_grouping.query_ball_point(r, 64, T(pts), T(q))
        oidx, ocnt = O.query_ball_point(r, 64, pts, q)
        np.testing.assert_array_equal(N(idx), oidx)
        np.testing.assert_array_equal(N(cnt), ocnt)


def test_query_ball_point_multi_equals_single():
    c = synth_clouds(3, 4096, seed=5, kind="surface")
    fps = tf_sampling.farthest_point_sample(512, T(c))
    q = tf_sampling.gather_point(T(c), fps)
    radii, ns = [0.1, 0.2, 0.4], [16, 32, 128]
    multi = tf_grouping.query_ball_point_multi(radii, ns, T(c), q)
    for (idx, cnt), r, s in zip(multi, radii, ns):
        i1, c1 = tf_grouping.query_ball_point(r, s, T(c), q)
        assert torch.equal(idx, i1) and torch.equal(cnt, c1)
        oidx, ocnt = O.query_ball_point(r, s, c, N(q))
        np.testing.assert_array_equal(N(idx), oidx)
        np.testing.assert_array_equal(N(cnt), ocnt)


def test_query_ball_point_argument_errors():
    c = T(synth_clouds(1, 64, 0))
    with pytest.raises(ValueError):
        tf_grouping.query_ball_point(0.0, 4, c, c)
    with pytest.raises(ValueError):
        tf_grouping.query_ball_point(0.1, 0, c, c)
    with pytest.raises(ValueError):
        tf_grouping.query_ball_point(0.1, 4, c[:, :, :2], c)
    with pytest.raises(Exception):
        tf_grouping.query_ball_point(0.1, 4, c.cpu(), c.cpu())   # no CPU fallback


# ------------------------------------------------------------------ FPS / gather
@pytest.mark.parametrize("B,Nn,M,kind", [(4, 2048, 512, "surface"), (4, 512, 128, "ball"), (2, 4096, 512, "surface"),
                                         (3, 700, 64, "ball"), (2, 100, 100, "ball"), (1, 1, 1, "ball"),
                                         (2, 64, 70, "ball"), (1, 8192, 1024, "surface"), (1, 9000, 40, "ball"),
                                         (2, 1024, 33, "surface"), (5, 130, 17, "ball")])
def test_fps_vs_oracle(B, Nn, M, kind):
    c = synth_clouds(B, Nn, seed=Nn + M, kind=kind)
    idx = tf_sampling.farthest_point_sample(M, T(c))
    np.testing.assert_array_equal(N(idx), O.farthest_point_sample(M, c))


def test_fps_beyond_the_register_resident_size():
    """n > 16384: the streamed kernel (running min-distance in the caller's scratch), same indices as the oracle
    incl. the (k mod 512, k) tie rule on duplicated points; the reference handles any n (tf_sampling_g.cu:133-141)"""
    c = synth_clouds(2, 20000, seed=3, kind="ball")
    c[1, 15000:15010] = c[1, 7]                                  # exact duplicates -> distance ties
    idx = tf_sampling.farthest_point_sample(96, T(c))
    np.testing.assert_array_equal(N(idx), O.farthest_point_sample(96, c))
    assert _lib.load().pcops_farthest_point_sample_workspace_bytes(2, 20000) == 4 * 2 * 20000
    assert _lib.load().pcops_farthest_point_sample_workspace_bytes(2, 16384) == 0


def test_fps_ties_follow_reference_rule():
    pts = np.zeros((2, 1030, 3), np.float32)
    pts[0, 600] = (1, 0, 0); pts[0, 40] = (-1, 0, 0); pts[0, 1029] = (0, 1, 0)
    pts[1] = lattice(1, 1030, 4)[0]                     # massive ties + duplicates
    idx = tf_sampling.farthest_point_sample(200, T(pts))
    np.testing.assert_array_equal(N(idx), O.farthest_point_sample(200, pts))
    assert N(idx)[0, 1] == 1029


def test_gather_point_and_grad():
    c = synth_clouds(3, 500, seed=2)
    rng = np.random.default_rng(0)
    idx = rng.integers(0, 500, (3, 77)).astype(np.int32)       # with repeats
    x = T(c).requires_grad_(True)
    out = tf_sampling.gather_point(x, T(idx))
    np.testing.assert_array_equal(N(out), O.gather_point(c, idx))
    go = rng.standard_normal((3, 77, 3)).astype(np.float32)
    out.backward(T(go))
    np.testing.assert_allclose(N(x.grad), O.gather_point_grad(c.shape, idx, go), rtol=0, atol=1e-5)


# ------------------------------------------------------------------ prob_sample
@pytest.mark.parametrize("B,n,m", [(1, 1, 5), (3, 2, 9), (4, 15, 64), (5, 100, 1000), (3, 1000, 300), (2, 1021, 77),
                                   (7, 4096, 4096), (2, 8191, 513), (2, 8192, 100), (3, 8193, 500), (2, 16389, 64),
                                   (2, 20000, 3000), (300, 40, 2048)])
def test_prob_sample_vs_oracle(B, n, m):
    """cumsum (the op's scratch) == and indices bit-exact, incl. zero-probability runs, r = 0, r = 1, r -> 1, rows that
    are not 16-byte aligned (odd n), more than one 8192-element chunk, rows beyond the LDS-staged search (n > 16384)"""
    rng = np.random.default_rng(B * 7 + n + m)
    p = rng.random((B, n)).astype(np.float32)
    p[0, : n // 2] = 0
    p[-1, rng.integers(0, n, n // 3 + 1)] = 0
    r = rng.random((B, m)).astype(np.float32)
    r[:, 0] = 0.0
    r[:, -1] = 1.0
    r[:, 1 % m] = np.nextafter(np.float32(1), np.float32(0))
    out, temp = tf_sampling.prob_sample(T(p), T(r), return_cumsum=True)
    oout, otemp = O.prob_sample(p, r, return_temp=True)
    np.testing.assert_array_equal(N(temp), otemp)
    np.testing.assert_array_equal(N(out), oout)
    assert out.dtype == torch.int32 and tuple(out.shape) == (B, m)


def test_prob_sample_boundaries_and_errors():
    p = np.array([[1, 1, 2, 4, 0, 0, 8]], np.float32)
    r = np.array([[0, 1 / 16, 1 / 8, 0.126, 1 / 4, 1 / 2, 0.51, 1.0]], np.float32)
    np.testing.assert_array_equal(N(tf_sampling.prob_sample(T(p), T(r))), [[0, 0, 1, 2, 2, 3, 6, 6]])
    assert tuple(tf_sampling.prob_sample(T(p), T(r[:, :0])).shape) == (1, 0)
    with pytest.raises(ValueError):
        tf_sampling.prob_sample(T(p)[0], T(r))                     # rank (tf_sampling.cpp:76)
    with pytest.raises(ValueError):
        tf_sampling.prob_sample(T(p), T(np.zeros((2, 4), np.float32)))   # batch mismatch (:79)
    with pytest.raises(Exception):
        tf_sampling.prob_sample(T(p).cpu(), T(r).cpu())            # no CPU fallback


# ------------------------------------------------------------------ group_point
@pytest.mark.parametrize("case", sorted(load_golden("group_point")))
def test_group_point_golden(case):
    g = load_golden("group_point")[case]
    p = T(g["points"]).requires_grad_(True)
    out = tf_grouping.group_point(p, T(g["idx"]))
    np.testing.assert_array_equal(N(out), g["out"])
    out.backward(T(g["grad_out"]))
    np.testing.assert_allclose(N(p.grad), g["grad_points"], rtol=0, atol=1e-4)


@pytest.mark.parametrize("b,n,c,m,s", [(4, 2048, 3, 512, 32), (2, 512, 128, 128, 64), (2, 512, 131, 128, 64),
                                       (1, 128, 259, 1, 128), (3, 77, 5, 9, 3)])
def test_group_point_vs_oracle(b, n, c, m, s):
    rng = np.random.default_rng(b + n + c)
    pts = rng.standard_normal((b, n, c)).astype(np.float32)
    idx = rng.integers(0, n, (b, m, s)).astype(np.int32)
    p = T(pts).requires_grad_(True)
    out = tf_grouping.group_point(p, T(idx))
    np.testing.assert_array_equal(N(out), O.group_point(pts, idx))
    go = rng.standard_normal(out.shape).astype(np.float32)
    out.backward(T(go))
    np.testing.assert_allclose(N(p.grad), O.group_point_grad(pts.shape, idx, go), rtol=1e-5, atol=1e-4)


def test_group_point_gradient_check_like_reference():
    """tf_grouping_op_test.py:9-25: gradient error of group_point < 1e-4 at (1,128,16)->(1,8,32,16),
    radius 0.3, nsample 32 -- here against the dense one-hot formulation in fp64."""
    rng = np.random.default_rng(100)
    pts = rng.random((1, 128, 16)).astype(np.float32)
    xyz1 = rng.random((1, 128, 3)).astype(np.float32)
    xyz2 = rng.random((1, 8, 3)).astype(np.float32)
    idx, _ = tf_grouping.query_ball_point(0.3, 32, T(xyz1), T(xyz2))
    p = T(pts).requires_grad_(True)
    out = tf_grouping.group_point(p, idx)
    go = rng.standard_normal(tuple(out.shape))
    out.backward(T(go.astype(np.float32)))
    onehot = np.zeros((8 * 32, 128))
    onehot[np.arange(8 * 32), N(idx).reshape(-1)] = 1
    dense = onehot.T @ go.reshape(8 * 32, 16)
    assert np.abs(N(p.grad)[0] - dense).max() < 1e-4


# ------------------------------------------------------------------ selection sort / knn_point
@pytest.mark.parametrize("case", sorted(load_golden("selection_sort")))
def test_selection_sort_golden(case):
    g = load_golden("selection_sort")[case]
    outi, out = tf_grouping.select_top_k(int(g["k"]), T(g["dist"]))
    np.testing.assert_array_equal(N(outi), g["outi"])
    np.testing.assert_array_equal(N(out), g["out"])


@pytest.mark.parametrize("n,k", [(70, 5), (1000, 32), (3000, 20), (8192, 8), (9000, 4), (33, 64)])
def test_selection_sort_rows_of_every_size_vs_oracle(n, k):
    """the wave-per-row kernel (row in LDS, n <= 8192) and the lane-per-row fallback against the literal restatement of
    the reference's scalar loop, on rows with many exact ties (quantised values), a NaN inside a row and a NaN heading
    one: the unstable swap order, `x < mv` picking the FIRST minimum, NaNs never winning -- all bit for bit"""
    rng = np.random.default_rng(n)
    d = (rng.integers(0, 50, size=(2, 5, n)) * 0.25).astype(np.float32)
    d[0, 1, n // 2] = np.nan
    d[1, 2, 0] = np.nan
    d[1, 3, :] = 1.0                                   # a constant row: nothing may move
    outi, out = tf_grouping.select_top_k(k, T(d))
    oi, ov = O.select_top_k(k, d)
    np.testing.assert_array_equal(N(outi), oi)
    np.testing.assert_array_equal(N(out), ov)


@pytest.mark.parametrize("b,n,m,c,k", [(2, 150, 40, 3, 12), (3, 1024, 1024, 3, 20), (1, 8192, 7, 3, 5), (2, 333, 50, 8, 16),
                                       (1, 9000, 6, 3, 4), (2, 64, 64, 1, 64)])
def test_knn_point_vs_oracle(b, n, m, c, k):
    """knn_point (tf_grouping.py:49-74) with the distances computed INSIDE the selection-sort kernel (n <= 8192: no
    (b,m,n,c) / (b,m,n) tensor), and through pcops_knn_point_dist + pcops_selection_sort beyond; values and indices bit
    for bit against the oracle (c ascending, uncontracted), incl. lattice points whose distances tie exactly"""
    rng = np.random.default_rng(n + k)
    x1 = rng.standard_normal((b, n, c)).astype(np.float32)
    x1[:, : n // 2] = np.round(x1[:, : n // 2] * 4) / 4          # a lattice half: exact ties between candidates
    x2 = x1[:, :m].copy() if m <= n else rng.standard_normal((b, m, c)).astype(np.float32)
    val, idx = tf_grouping.knn_point(k, T(x1), T(x2))
    oval, oidx = O.knn_point(k, x1, x2)
    assert val.shape == (b, m, k) and idx.dtype == torch.int32
    np.testing.assert_array_equal(N(idx), oidx)
    np.testing.assert_array_equal(N(val), oval)


def test_knn_point_dist_matrix_vs_oracle_order():
    """the stand-alone distance kernel == the fused kernel's rows == sum over c ascending of the squared differences"""
    from scanobjectnn_amd import _lib
    rng = np.random.default_rng(3)
    x1 = rng.standard_normal((2, 500, 3)).astype(np.float32)
    x2 = rng.standard_normal((2, 33, 3)).astype(np.float32)
    d = torch.empty((2, 33, 500), dtype=torch.float32, device=DEV)
    a, q = T(x1), T(x2)
    _lib.call("pcops_knn_point_dist", 2, 500, 3, 33, _lib.ptr(a), _lib.ptr(q), _lib.ptr(d))
    want = np.zeros((2, 33, 500), np.float32)
    for l in range(3):
        df = x1[:, None, :, l] - x2[:, :, None, l]
        want = want + df * df
    np.testing.assert_array_equal(N(d), want)
    with pytest.raises(ValueError):
        tf_grouping.knn_point(0, a, q)


# ------------------------------------------------------------------ three_nn / interpolate
@pytest.mark.parametrize("case", sorted(load_golden("three_interp")))
def test_three_nn_interp_golden(case):
    g = load_golden("three_interp")[case]
    dist, idx = tf_interpolate.three_nn(T(g["xyz1"]), T(g["xyz2"]))
    np.testing.assert_array_equal(N(idx), g["idx"])
    np.testing.assert_array_equal(N(dist), g["dist"])
    p = T(g["points"]).requires_grad_(True)
    out = tf_interpolate.three_interpolate(p, idx, T(g["weight"]))
    np.testing.assert_allclose(N(out), g["out"], rtol=0, atol=1e-6)
    out.backward(T(g["grad_out"]))
    np.testing.assert_allclose(N(p.grad), g["grad_points"], rtol=1e-5, atol=1e-4)


@pytest.mark.parametrize("b,n,m,c", [(4, 2048, 512, 128), (4, 512, 128, 256), (4, 128, 1, 256), (2, 333, 2, 7),
                                     (1, 5000, 4500, 4)])
def test_three_nn_interp_vs_oracle(b, n, m, c):
    x1 = synth_clouds(b, n, seed=n + m)
    x2 = x1[:, :: max(1, n // m)][:, :m].copy() if m > 2 else np.zeros((b, m, 3), np.float32)
    dist, idx = tf_interpolate.three_nn(T(x1), T(x2))
    od, oi = O.three_nn(x1, x2)
    np.testing.assert_array_equal(N(idx), oi)
    np.testing.assert_array_equal(N(dist), od)
    rng = np.random.default_rng(c)
    pts = rng.standard_normal((b, m, c)).astype(np.float32)
    w = rng.random((b, n, 3)).astype(np.float32)
    out = tf_interpolate.three_interpolate(T(pts), idx, T(w))
    np.testing.assert_array_equal(N(out), O.three_interpolate(pts, oi, w))


def test_three_interpolate_gradient_check_like_reference():
    """tf_interpolate_op_test.py:9-21: (1,8,16) -> (1,128,16), weights 1/3, error < 1e-4."""
    rng = np.random.default_rng(100)
    pts = rng.random((1, 8, 16)).astype(np.float32)
    x1 = rng.random((1, 128, 3)).astype(np.float32)
    x2 = rng.random((1, 8, 3)).astype(np.float32)
    _, idx = tf_interpolate.three_nn(T(x1), T(x2))
    w = np.full((1, 128, 3), 1.0 / 3.0, np.float32)
    p = T(pts).requires_grad_(True)
    out = tf_interpolate.three_interpolate(p, idx, T(w))
    go = rng.standard_normal(tuple(out.shape))
    out.backward(T(go.astype(np.float32)))
    dense = np.zeros((8, 16))
    ii = N(idx)[0]
    for j in range(128):
        for t in range(3):
            dense[ii[j, t]] += go[0, j] / 3.0
    assert np.abs(N(p.grad)[0] - dense).max() < 1e-4


# ------------------------------------------------------------------ full-size properties
def test_full_size_sa1_properties():
    """BASELINE config 2 geometry stage at B=256: size-independent properties (the oracle is too slow
    for the whole batch): rows ascending then padded, every listed index inside the ball, clouds
    independent (batch of 256 == the same clouds run 8 at a time)."""
    c = synth_clouds(256, 2048, seed=1234, kind="surface")
    x = T(c)
    fps = tf_sampling.farthest_point_sample(512, x)
    q = tf_sampling.gather_point(x, fps)
    idx, cnt = tf_grouping.query_ball_point(0.2, 32, x, q)
    assert int(fps[:, 0].abs().max()) == 0
    srt = torch.sort(fps, dim=1).values
    assert bool((srt[:, 1:] != srt[:, :-1]).all())                       # no repeats
    ar = torch.arange(32, device=DEV).view(1, 1, 32)
    live = ar < cnt.unsqueeze(-1)
    d = idx[:, :, 1:] - idx[:, :, :-1]
    assert bool(((d > 0) | ~live[:, :, 1:]).all())                         # ascending while live
    assert bool(((idx == idx[:, :, :1]) | live).all())                     # padded with first hit
    g = tf_grouping.group_point(x, idx) - q.unsqueeze(2)
    dist = torch.sqrt((g[..., 0] * g[..., 0] + g[..., 1] * g[..., 1]) + g[..., 2] * g[..., 2])
    assert bool((dist < 0.2).all())
    assert int(cnt.min()) >= 1                                            # queries are dataset points
    sub = slice(40, 48)
    i2, c2 = tf_grouping.query_ball_point(0.2, 32, x[sub].contiguous(), q[sub].contiguous())
    assert torch.equal(i2, idx[sub]) and torch.equal(c2, cnt[sub])
    oidx, ocnt = O.query_ball_point(0.2, 32, c[sub], N(q[sub]))
    np.testing.assert_array_equal(N(i2), oidx)
    np.testing.assert_array_equal(N(fps[sub]), O.farthest_point_sample(512, c[sub]))


def test_adam_step_is_the_reference_update():
    """`pcops_adam_step` (one launch) against the elementwise form of tf.train.AdamOptimizer the host code used before
    (`train_util.TFAdam`, reference trainers `pointnet2/train.py:165-168`), three steps on a ragged flat bucket"""
    from scanobjectnn_amd import train_util as TU
    torch.manual_seed(3)
    net_a = torch.nn.Sequential(torch.nn.Linear(37, 53), torch.nn.Linear(53, 7)).to(DEV)
    net_b = torch.nn.Sequential(torch.nn.Linear(37, 53), torch.nn.Linear(53, 7)).to(DEV)
    net_b.load_state_dict(net_a.state_dict())
    fa, fb = TU.FlatParams(net_a), TU.FlatParams(net_b)
    oa, ob = TU.TFAdam(fa), TU.TFAdam(fb)
    assert fa.numel % 4 == 0
    for step in range(3):
        g = torch.randn(fa.numel, device=DEV) * (10.0 ** (step - 1))
        fa.grad.copy_(g)
        fb.grad.copy_(g)
        oa.step(1e-3)                      # device launch
        TU.FUSED_ADAM, keep = False, TU.FUSED_ADAM
        try:
            ob.step(1e-3)                  # seven elementwise launches
        finally:
            TU.FUSED_ADAM = keep
        assert torch.allclose(oa.m, ob.m, rtol=1e-6, atol=1e-12) and torch.allclose(oa.v, ob.v, rtol=1e-6, atol=1e-20)
        # lr-sized updates computed a few ulp apart, added to parameters of magnitude <= 0.2: one ulp of those is 1.5e-8
        assert (fa.flat - fb.flat).abs().max().item() <= 3.1e-8


# ------------------------------------------------------------------ EdgeConv backward, both terms in one owner walk (round 5)
@pytest.mark.parametrize("b,n,k,C,ld", [(3, 2048, 20, 64, False), (2, 2048, 20, 128, True), (2, 1024, 20, 64, True),
                                        (2, 1088, 16, 64, False), (1, 1984, 20, 256, False)])
def test_edge_pool_bwd_against_float64(b, n, k, C, ld):
    """pcops_edge_pool_bwd / _ld (csrc/edgeconv.hip ec_bwd_lds_kernel: inverse index, then ONE walk that adds the Ctr rows
    AND the arg-row values of every list entry) against the definition in float64:
      a[g,c] = p gpool [scale ysel + shift > 0],  dCtr[g] = q (SQ + k Ctr) + k t + a,
      dQ[i] = q (cnt_i Q[i] + sum_{(g,s) -> i} Ctr[g]) + cnt_i t + sum_{(g,s) -> i, arg[g,c] == s} a[g,c]"""
    from scanobjectnn_amd.dgcnn import tf_util
    lib = _lib.load()
    gen = torch.Generator(device=DEV).manual_seed(n + C)
    x = torch.from_numpy(synth_clouds(b, n, seed=C)).to(DEV)
    idx = tf_util.knn_graph(x, k=k)                                   # a real graph: list lengths 0 .. ~100
    m, G = n, b * n
    QC = torch.randn(b, n, 2 * C, device=DEV, generator=gen)
    Q, Ctr = (QC[..., :C], QC[..., C:]) if ld else (QC[..., :C].contiguous(), QC[..., C:].contiguous())
    gpool, ysel, SQ = (torch.randn(G, C, device=DEV, generator=gen) for _ in range(3))
    arg = torch.randint(0, k, (G, C), device=DEV, generator=gen, dtype=torch.int32).to(torch.uint8)
    sc, sh, p, q, t = (torch.randn(C, device=DEV, generator=gen) for _ in range(5))
    wsp = torch.empty(int(lib.pcops_sa_scatter_workspace_bytes(b, n, m, k)) // 4, dtype=torch.int32, device=DEV)
    P = lambda v: v.data_ptr()
    if ld:
        dQC = torch.full((b, n, 2 * C), float("nan"), device=DEV)
        _lib.call("pcops_edge_pool_bwd_ld", b, n, m, k, C, P(QC), 2 * C, P(QC) + 4 * C, 2 * C, P(idx), P(gpool), P(ysel), P(SQ),
                  P(arg), P(sc), P(sh), P(p), P(q), P(t), P(dQC), 2 * C, P(dQC) + 4 * C, 2 * C, P(wsp))
        dQ, dCtr = dQC[..., :C], dQC[..., C:]
    else:
        dQ, dCtr = torch.full((b, n, C), float("nan"), device=DEV), torch.full((b, m, C), float("nan"), device=DEV)
        _lib.call("pcops_edge_pool_bwd", b, n, m, k, C, P(Q), P(Ctr), P(idx), P(gpool), P(ysel), P(SQ), P(arg), P(sc), P(sh),
                  P(p), P(q), P(t), P(dQ), P(dCtr), P(wsp))
    torch.cuda.synchronize()
    d = torch.float64
    a = torch.where(ysel * sc + sh > 0, (p * gpool).to(d), torch.zeros((), dtype=d, device=DEV)).view(b, m, C)
    Ctr64 = Ctr.to(d)
    want_ctr = q.to(d) * (SQ.to(d).view(b, m, C) + k * Ctr64) + k * t.to(d) + a
    ii = idx.long()                                                    # (b, m, k)
    cnt = torch.zeros(b, n, dtype=d, device=DEV).scatter_add_(1, ii.view(b, -1), torch.ones(b, m * k, dtype=d, device=DEV))
    sumc = torch.zeros(b, n, C, dtype=d, device=DEV).index_put_(
        (torch.arange(b, device=DEV).view(b, 1, 1).expand(b, m, k), ii), Ctr64.unsqueeze(2).expand(b, m, k, C), accumulate=True)
    hit = arg.view(b, m, 1, C).long() == torch.arange(k, device=DEV).view(1, 1, k, 1)          # (b, m, k, C)
    suma = torch.zeros(b, n, C, dtype=d, device=DEV).index_put_(
        (torch.arange(b, device=DEV).view(b, 1, 1).expand(b, m, k), ii), a.unsqueeze(2) * hit.to(d), accumulate=True)
    want_q = q.to(d) * (cnt.unsqueeze(-1) * Q.to(d) + sumc) + cnt.unsqueeze(-1) * t.to(d) + suma
    assert (dCtr.to(d) - want_ctr).abs().max().item() <= 2e-5 * want_ctr.abs().max().item()
    assert (dQ.to(d) - want_q).abs().max().item() <= 2e-5 * want_q.abs().max().item()
