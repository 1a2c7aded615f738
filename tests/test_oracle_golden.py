"""The CPU oracle (oracle/pcops_oracle.c) against the golden vectors produced by the
reference's own compiled CPU functions (tests/golden/make_golden.py), and -- when
oracle/_ref is present -- against the compiled reference directly on fresh inputs.
Integer outputs bit-exact; copies ==; float sums to 1e-6 (same order, so in fact ==)."""
import numpy as np
import pytest

from conftest import load_golden
from oracle import oracle as O


@pytest.mark.parametrize("case", sorted(load_golden("query_ball_point")))
def test_query_ball_point_golden(case):
    g = load_golden("query_ball_point")[case]
    idx, cnt = O.query_ball_point(float(g["radius"]), int(g["nsample"]), g["xyz1"], g["xyz2"])
    np.testing.assert_array_equal(idx, g["idx"])
    # pts_cnt is not produced by the CPU twin: check it against the row structure
    s = int(g["nsample"])
    for b in range(idx.shape[0]):
        for j in range(idx.shape[1]):
            c, row = cnt[b, j], idx[b, j]
            assert 0 <= c <= s
            assert np.all(np.diff(row[:c]) > 0)
            if c:
                assert np.all(row[c:] == row[0])
            else:
                assert np.all(row == 0)


def test_query_ball_point_edge_semantics():
    g = load_golden("query_ball_point")
    idx, cnt = O.query_ball_point(1e-21, 4, g["tiny_radius"]["xyz1"], g["tiny_radius"]["xyz2"])
    assert cnt.sum() == 0 and idx.sum() == 0          # max(d,1e-20) < r is never true
    z = g["ball_foreign_r0.02_s4"]
    idx, cnt = O.query_ball_point(float(z["radius"]), 4, z["xyz1"], z["xyz2"])
    assert (cnt == 0).any()


@pytest.mark.parametrize("case", sorted(load_golden("group_point")))
def test_group_point_golden(case):
    g = load_golden("group_point")[case]
    np.testing.assert_array_equal(O.group_point(g["points"], g["idx"]), g["out"])
    np.testing.assert_array_equal(
        O.group_point_grad(g["points"].shape, g["idx"], g["grad_out"]), g["grad_points"])


@pytest.mark.parametrize("case", sorted(load_golden("selection_sort")))
def test_selection_sort_golden(case):
    g = load_golden("selection_sort")[case]
    k = int(g["k"])
    outi, out = O.select_top_k(k, g["dist"])
    np.testing.assert_array_equal(outi, g["outi"])
    np.testing.assert_array_equal(out, g["out"])


def test_selection_sort_known_answer_print():
    """selection_sort.cpp:68-78 run unmodified prints idx `3 2 1 0` per row."""
    g = load_golden("selection_sort")["known_answer"]
    outi, out = O.select_top_k(3, g["dist"])
    assert (outi == np.array([3, 2, 1, 0], np.int32)).all()
    np.testing.assert_array_equal(out[0, 0], [7, 8, 9, 10])
    np.testing.assert_array_equal(out[1, 1], [-5, -4, -3, -2])


@pytest.mark.parametrize("case", sorted(load_golden("three_interp")))
def test_three_nn_interp_golden(case):
    g = load_golden("three_interp")[case]
    dist, idx = O.three_nn(g["xyz1"], g["xyz2"])
    np.testing.assert_array_equal(idx, g["idx"])
    np.testing.assert_array_equal(dist, g["dist"])       # incl. +inf for m<3
    np.testing.assert_array_equal(O.three_interpolate(g["points"], idx, g["weight"]), g["out"])
    np.testing.assert_allclose(
        O.three_interpolate_grad(g["points"].shape, idx, g["weight"], g["grad_out"]),
        g["grad_points"], rtol=0, atol=1e-6)


def test_three_nn_m1_matches_bga_assumption():
    """pointnet2_cls_bga.py:56 relies on m=1 -> dist=(d,inf,inf), idx=(0,0,0)."""
    g = load_golden("three_interp")["m1"]
    assert np.isinf(g["dist"][..., 1:]).all() and (g["idx"] == 0).all()


@pytest.mark.skipif(not O.have_ref(), reason="oracle/_ref not built (no reference checkout)")
def test_oracle_vs_compiled_reference_fresh_inputs():
    from scanobjectnn_amd.synth import synth_clouds
    rng = np.random.default_rng(123)
    for seed, (n, m, r, s) in enumerate([(512, 128, 0.2, 32), (300, 77, 0.4, 64), (128, 128, 0.1, 16)]):
        c = synth_clouds(3, n, seed=100 + seed, kind="ball" if seed % 2 else "surface")
        q = c[:, rng.permutation(n)[:m]].copy()
        np.testing.assert_array_equal(O.query_ball_point(r, s, c, q)[0],
                                      O.ref_query_ball_point(r, s, c, q))
        d0, i0 = O.three_nn(c, q)
        d1, i1 = O.ref_three_nn(c, q)
        np.testing.assert_array_equal(i0, i1)
        np.testing.assert_array_equal(d0, d1)
    dist = rng.integers(0, 9, (2, 7, 33)).astype(np.float32)
    a, b = O.select_top_k(12, dist), O.ref_select_top_k(12, dist)
    np.testing.assert_array_equal(a[0], b[0])
    np.testing.assert_array_equal(a[1], b[1])


# ---- restatement-only functions: property pins (no CPU code in the reference) ----
def test_fps_properties():
    from scanobjectnn_amd.synth import synth_clouds
    c = synth_clouds(2, 700, seed=9, kind="ball")       # n > 512 exercises the (k mod 512) rule
    idx = O.farthest_point_sample(64, c)
    assert (idx[:, 0] == 0).all()
    for b in range(2):
        assert len(set(idx[b])) == 64
        mind = np.full(700, 1e38, np.float32)
        for j in range(1, 64):
            p = c[b, idx[b, j - 1]]
            d = ((c[b] - p) ** 2)
            d = (d[:, 0] + d[:, 1]) + d[:, 2]
            mind = np.minimum(mind, d.astype(np.float32))
            assert mind[idx[b, j]] == mind.max()        # greedy optimality


def test_fps_tie_rule_mod512():
    """ties -> smaller (k mod 512) then smaller k (tf_sampling_g.cu:130-165)."""
    pts = np.zeros((1, 1030, 3), np.float32)
    pts[0, 600] = (1, 0, 0)     # 600 mod 512 = 88
    pts[0, 40] = (-1, 0, 0)     # 40
    pts[0, 1029] = (0, 1, 0)    # 1029 mod 512 = 5  -> wins the tie at distance 1
    idx = O.farthest_point_sample(2, pts)
    assert idx[0, 1] == 1029
    pts2 = np.zeros((1, 300, 3), np.float32)
    pts2[0, 250] = (1, 0, 0)
    pts2[0, 17] = (0, 0, 1)
    assert O.farthest_point_sample(2, pts2)[0, 1] == 17   # n <= 512: lowest k


def test_knn_graph_matches_materialised_path():
    rng = np.random.default_rng(5)
    x = rng.standard_normal((2, 90, 7)).astype(np.float32)
    x[0, 13] = x[0, 2]                                   # exact duplicate -> tie at distance 0
    adj = O.pairwise_distance(x)
    np.testing.assert_array_equal(O.knn(adj, 9), O.knn_graph(x, 9))
    nn = O.knn_graph(x, 9)
    assert nn[0, 13, 0] == 2 and nn[0, 13, 1] == 13      # lower index first on ties
    ef = O.get_edge_feature(x, nn, 9)
    np.testing.assert_array_equal(ef[..., :7], np.broadcast_to(x[:, :, None, :], ef[..., :7].shape))
    np.testing.assert_array_equal(ef[1, 4, 3, 7:], x[1, nn[1, 4, 3]] - x[1, 4])
