"""The CPU checker is called from many threads by bench.py's cpu_baseline leg (one persistent worker per host
core).  Round 3's driver record lost `cpu_baseline.ops` to a ctypes race: `argtypes` were
re-assigned on every call while 32 threads used the same function object.  Signatures are now attached once per
symbol under a lock (oracle/oracle.py: lib(), _reffn); this file calls EVERY oracle_* restatement and EVERY
compiled reference twin from 32 threads that start together on a cold module and checks each result against a
single-threaded call."""
import threading
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

THREADS = 32


def _cases(O, use_ref):
    rng = np.random.default_rng(11)
    b, n, m, s, c, k = 2, 96, 24, 8, 8, 5
    xyz1 = rng.random((b, n, 3), dtype=np.float32)
    xyz2 = np.ascontiguousarray(xyz1[:, ::4][:, :m])
    pts = rng.random((b, n, c), dtype=np.float32)
    idx = O.query_ball_point(0.3, s, xyz1, xyz2)[0]
    gout = rng.random((b, m, s, c), dtype=np.float32)
    w = rng.random((b, n, 3), dtype=np.float32)
    i3 = rng.integers(0, m, (b, n, 3)).astype(np.int32)
    p2 = rng.random((b, m, c), dtype=np.float32)
    g2 = rng.random((b, n, c), dtype=np.float32)
    dist = rng.random((b, m, n), dtype=np.float32)
    fidx = rng.integers(0, n, (b, m)).astype(np.int32)
    g3 = rng.random((b, m, 3), dtype=np.float32)
    feat = rng.random((b, n, c), dtype=np.float32)
    adj = O.pairwise_distance(feat)
    nn = O.knn(adj, k)
    cases = {
        "query_ball_point": lambda: O.query_ball_point(0.3, s, xyz1, xyz2),
        "group_point": lambda: O.group_point(pts, idx),
        "group_point_grad": lambda: O.group_point_grad(pts.shape, idx, gout),
        "select_top_k": lambda: O.select_top_k(k, dist),
        "knn_point": lambda: O.knn_point(k, xyz1, xyz2),
        "farthest_point_sample": lambda: O.farthest_point_sample(m, xyz1),
        "gather_point": lambda: O.gather_point(xyz1, fidx),
        "gather_point_grad": lambda: O.gather_point_grad(xyz1.shape, fidx, g3),
        "three_nn": lambda: O.three_nn(xyz1, xyz2),
        "three_interpolate": lambda: O.three_interpolate(p2, i3, w),
        "three_interpolate_grad": lambda: O.three_interpolate_grad(p2.shape, i3, w, g2),
        "pairwise_distance": lambda: O.pairwise_distance(feat),
        "knn": lambda: O.knn(adj, k),
        "knn_graph": lambda: O.knn_graph(feat, k),
        "get_edge_feature": lambda: O.get_edge_feature(feat, nn, k),
    }
    if use_ref:
        cases.update({
            "ref_query_ball_point": lambda: O.ref_query_ball_point(0.3, s, xyz1, xyz2),
            "ref_group_point": lambda: O.ref_group_point(pts, idx),
            "ref_group_point_grad": lambda: O.ref_group_point_grad(pts.shape, idx, gout),
            "ref_select_top_k": lambda: O.ref_select_top_k(k, dist),
            "ref_three_nn": lambda: O.ref_three_nn(xyz1, xyz2),
            "ref_three_interpolate": lambda: O.ref_three_interpolate(p2, i3, w),
            "ref_three_interpolate_grad": lambda: O.ref_three_interpolate_grad(p2.shape, i3, w, g2),
        })
    return cases


def _same(a, b):
    if isinstance(a, tuple):
        return all(_same(x, y) for x, y in zip(a, b))
    return np.array_equal(a, b, equal_nan=True)


def test_every_oracle_function_from_32_threads():
    from oracle import oracle as O
    use_ref = O.have_ref()
    cases = _cases(O, use_ref)
    want = {name: fn() for name, fn in cases.items()}
    # COLD handles: the library handles and signatures are re-created by whichever of the 32 threads gets there
    # first (the closures look the functions up at call time)
    O._lib = None
    O._ref.clear()
    O._ref_fns.clear()
    start = threading.Barrier(THREADS)
    errors = []

    def worker(t):
        start.wait()
        names = list(cases)
        for rep in range(6):
            for j in range(len(names)):
                name = names[(j + t) % len(names)]          # every thread starts on a different symbol
                try:
                    got = cases[name]()
                    if not _same(got, want[name]):
                        errors.append("%s: result differs under threads" % name)
                except Exception as ex:                      # ctypes.ArgumentError was round 3's failure
                    errors.append("%s: %r" % (name, ex))

    with ThreadPoolExecutor(max_workers=THREADS) as ex:
        list(ex.map(worker, range(THREADS)))
    assert not errors, errors[:5]
    assert len(cases) == (22 if use_ref else 15)


def test_bench_cpu_ops_leg_runs_and_reports_every_op():
    """bench.py's op-level CPU leg itself (tiny budget): six ops, no error entry, kind = reference when oracle/_ref
    is built (it is in this container and ships prebuilt to the GPU box)."""
    import bench
    from oracle import oracle as O
    out = bench.cpu_ops_baseline(seconds_budget=0.6)
    assert out["kind"] == ("reference" if O.have_ref() else "port")
    assert sorted(out["ops"]) == sorted(["query_ball_point", "group_point", "group_point_grad", "three_nn",
                                         "three_interpolate", "three_interpolate_grad"])
    for name, d in out["ops"].items():
        assert d["clouds_per_s_1core"] > 0 and d["clouds_per_s_sharded"] > 0, name
