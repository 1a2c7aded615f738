"""ctypes/numpy front end of the CPU oracle (oracle/pcops_oracle.c) and of the
compiled reference CPU twins (oracle/_ref/*.so, built by oracle/Makefile from the
reference sources where they lie).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  Nothing under scanobjectnn_amd/ may import this.
"""
import ctypes as C
import os
import subprocess
import threading

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
_i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")


def build(force=False):
    """Compile liboracle.so (and oracle/_ref when the reference checkout exists)."""
    so = os.path.join(_HERE, "liboracle.so")
    src = os.path.join(_HERE, "pcops_oracle.c")
    stale = (not os.path.exists(so)) or os.path.getmtime(so) < os.path.getmtime(src)
    if force or stale or (os.path.isdir("/root/reference") and not have_ref()):
        subprocess.check_call(["make", "-s", "-C", _HERE, "all"])
    return so


_lib = None
_load_lock = threading.Lock()


def lib():
    """The restatement library.  Signatures are attached ONCE, under a lock, before the handle is
    published: bench.py's CPU leg calls these from `cores` threads at once, and a thread that saw
    the handle before its argtypes (or re-assigned them mid-call, the round-3 race) failed with a
    ctypes ArgumentError."""
    global _lib
    if _lib is not None:
        return _lib
    with _load_lock:
        if _lib is not None:
            return _lib
        h = C.CDLL(build())
        I, F = C.c_int, C.c_float
        sig = {
            "oracle_query_ball_point": [I, I, I, F, I, _f32p, _f32p, _i32p, _i32p],
            "oracle_group_point": [I, I, I, I, I, _f32p, _i32p, _f32p],
            "oracle_group_point_grad": [I, I, I, I, I, _f32p, _i32p, _f32p],
            "oracle_selection_sort": [I, I, I, I, _f32p, _i32p, _f32p],
            "oracle_knn_point": [I, I, I, I, I, _f32p, _f32p, _f32p, _i32p],
            "oracle_farthest_point_sample": [I, I, I, _f32p, _i32p],
            "oracle_cumsum": [I, I, _f32p, _f32p],
            "oracle_prob_sample": [I, I, I, _f32p, _f32p, _f32p, _i32p],
            "oracle_gather_point": [I, I, I, _f32p, _i32p, _f32p],
            "oracle_gather_point_grad": [I, I, I, _f32p, _i32p, _f32p],
            "oracle_three_nn": [I, I, I, _f32p, _f32p, _f32p, _i32p],
            "oracle_three_interpolate": [I, I, I, I, _f32p, _i32p, _f32p, _f32p],
            "oracle_three_interpolate_grad": [I, I, I, I, _f32p, _i32p, _f32p, _f32p],
            "oracle_pairwise_distance": [I, I, I, _f32p, _f32p],
            "oracle_knn_topk": [I, I, I, _f32p, _i32p],
            "oracle_knn_graph": [I, I, I, I, _f32p, _i32p],
            "oracle_edge_feature": [I, I, I, I, _f32p, _i32p, _f32p],
        }
        for name, argtypes in sig.items():
            fn = getattr(h, name)
            fn.argtypes = argtypes
            fn.restype = None
        _lib = h
    return _lib


def _f(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _i(a):
    return np.ascontiguousarray(a, dtype=np.int32)


# --------------------------------------------------------------------------
# restatement (liboracle.so)
# --------------------------------------------------------------------------
def query_ball_point(radius, nsample, xyz1, xyz2):
    xyz1, xyz2 = _f(xyz1), _f(xyz2)
    b, n, _ = xyz1.shape
    m = xyz2.shape[1]
    idx = np.zeros((b, m, nsample), np.int32)
    cnt = np.zeros((b, m), np.int32)
    lib().oracle_query_ball_point(b, n, m, float(radius), nsample, xyz1, xyz2, idx, cnt)
    return idx, cnt


def group_point(points, idx, out=None):
    points, idx = _f(points), _i(idx)
    b, n, c = points.shape
    _, m, s = idx.shape
    out = np.empty((b, m, s, c), np.float32) if out is None else out   # out: caller-owned buffer (bench.py's sharded CPU leg)
    lib().oracle_group_point(b, n, c, m, s, points, idx, out)
    return out


def group_point_grad(points_shape, idx, grad_out):
    idx, grad_out = _i(idx), _f(grad_out)
    b, n, c = points_shape
    _, m, s = idx.shape
    g = np.empty((b, n, c), np.float32)
    lib().oracle_group_point_grad(b, n, c, m, s, grad_out, idx, g)
    return g


def select_top_k(k, dist):
    dist = _f(dist)
    b, m, n = dist.shape
    outi = np.empty((b, m, n), np.int32)
    out = np.empty((b, m, n), np.float32)
    lib().oracle_selection_sort(b, n, m, k, dist, outi, out)
    return outi, out


def knn_point(k, xyz1, xyz2):
    xyz1, xyz2 = _f(xyz1), _f(xyz2)
    b, n, c = xyz1.shape
    m = xyz2.shape[1]
    val = np.empty((b, m, k), np.float32)
    idx = np.empty((b, m, k), np.int32)
    lib().oracle_knn_point(b, n, c, m, k, xyz1, xyz2, val, idx)
    return val, idx


def farthest_point_sample(npoint, inp):
    inp = _f(inp)
    b, n, _ = inp.shape
    out = np.zeros((b, npoint), np.int32)
    lib().oracle_farthest_point_sample(b, n, npoint, inp, out)
    return out


def cumsum(inp):
    """(b,n) f32 -> (b,n) f32, the association of tf_sampling_g.cu:7-81."""
    inp = _f(inp)
    b, n = inp.shape
    out = np.zeros((b, n), np.float32)
    lib().oracle_cumsum(b, n, inp, out)
    return out


def prob_sample(inp, inpr, return_temp=False):
    """inp (b,ncategory) weights, inpr (b,npoints) uniforms -> (b,npoints) i32 (tf_sampling.py:14-23)."""
    inp, inpr = _f(inp), _f(inpr)
    b, n = inp.shape
    m = inpr.shape[1]
    temp = np.zeros((b, n), np.float32)
    out = np.zeros((b, m), np.int32)
    lib().oracle_prob_sample(b, n, m, inp, inpr, temp, out)
    return (out, temp) if return_temp else out


def gather_point(inp, idx):
    inp, idx = _f(inp), _i(idx)
    b, n, _ = inp.shape
    m = idx.shape[1]
    out = np.empty((b, m, 3), np.float32)
    lib().oracle_gather_point(b, n, m, inp, idx, out)
    return out


def gather_point_grad(inp_shape, idx, out_g):
    idx, out_g = _i(idx), _f(out_g)
    b, n, _ = inp_shape
    m = idx.shape[1]
    g = np.empty((b, n, 3), np.float32)
    lib().oracle_gather_point_grad(b, n, m, out_g, idx, g)
    return g


def three_nn(xyz1, xyz2):
    xyz1, xyz2 = _f(xyz1), _f(xyz2)
    b, n, _ = xyz1.shape
    m = xyz2.shape[1]
    dist = np.empty((b, n, 3), np.float32)
    idx = np.empty((b, n, 3), np.int32)
    lib().oracle_three_nn(b, n, m, xyz1, xyz2, dist, idx)
    return dist, idx


def three_interpolate(points, idx, weight):
    points, idx, weight = _f(points), _i(idx), _f(weight)
    b, m, c = points.shape
    n = idx.shape[1]
    out = np.empty((b, n, c), np.float32)
    lib().oracle_three_interpolate(b, m, c, n, points, idx, weight, out)
    return out


def three_interpolate_grad(points_shape, idx, weight, grad_out):
    idx, weight, grad_out = _i(idx), _f(weight), _f(grad_out)
    b, m, c = points_shape
    n = idx.shape[1]
    g = np.empty((b, m, c), np.float32)
    lib().oracle_three_interpolate_grad(b, n, c, m, grad_out, idx, weight, g)
    return g


def pairwise_distance(x):
    x = _f(x)
    b, n, c = x.shape
    adj = np.empty((b, n, n), np.float32)
    lib().oracle_pairwise_distance(b, n, c, x, adj)
    return adj


def knn(adj, k=20):
    adj = _f(adj)
    b, n, n2 = adj.shape
    out = np.empty((b, n, k), np.int32)
    lib().oracle_knn_topk(b * n, n2, k, adj, out)
    return out


def knn_graph(x, k=20):
    x = _f(x)
    b, n, c = x.shape
    out = np.empty((b, n, k), np.int32)
    lib().oracle_knn_graph(b, n, c, k, x, out)
    return out


def get_edge_feature(x, nn_idx, k=20):
    x, nn_idx = _f(x), _i(nn_idx)
    b, n, c = x.shape
    out = np.empty((b, n, k, 2 * c), np.float32)
    lib().oracle_edge_feature(b, n, c, k, x, nn_idx, out)
    return out


# --------------------------------------------------------------------------
# compiled reference CPU twins (oracle/_ref) -- validation + "reference" baseline
# --------------------------------------------------------------------------
_REF = os.path.join(_HERE, "_ref")
_REF_LIBS = ("libref_grouping.so", "libref_selsort.so", "libref_interp.so")
_ref = {}
_ref_fns = {}


def have_ref():
    return all(os.path.exists(os.path.join(_REF, n)) for n in _REF_LIBS)


def _reffn(libname, mangled, argtypes):
    """One function object per symbol, its signature attached once under the load lock (every caller
    of a symbol passes the same argtypes; re-assigning them per call raced across threads)."""
    fn = _ref_fns.get(mangled)
    if fn is not None:
        return fn
    with _load_lock:
        fn = _ref_fns.get(mangled)
        if fn is None:
            if libname not in _ref:
                _ref[libname] = C.CDLL(os.path.join(_REF, libname))
            fn = getattr(_ref[libname], mangled)
            fn.argtypes = argtypes
            fn.restype = None
            _ref_fns[mangled] = fn
    return fn


class _silence_stdout:
    """selection_sort_cpu printf()s from inside the function (selection_sort.cpp:35-48)."""

    _lock = threading.Lock()     # fd 1 is process-wide: one redirection at a time

    def __enter__(self):
        self._lock.acquire()
        C.CDLL(None).fflush(None)
        self._saved = os.dup(1)
        self._null = os.open(os.devnull, os.O_WRONLY)
        os.dup2(self._null, 1)

    def __exit__(self, *a):
        C.CDLL(None).fflush(None)
        os.dup2(self._saved, 1)
        os.close(self._null)
        os.close(self._saved)
        self._lock.release()


def ref_query_ball_point(radius, nsample, xyz1, xyz2):
    """query_ball_point_cpu (grouping/test/query_ball_point.cpp:19-47). No pts_cnt.
    idx is zero-filled first: the reference leaves zero-hit rows unwritten."""
    I, F = C.c_int, C.c_float
    fn = _reffn("libref_grouping.so", "_Z20query_ball_point_cpuiiifiPKfS0_Pi",
                [I, I, I, F, I, _f32p, _f32p, _i32p])
    xyz1, xyz2 = _f(xyz1), _f(xyz2)
    b, n, _ = xyz1.shape
    m = xyz2.shape[1]
    idx = np.zeros((b, m, nsample), np.int32)
    fn(b, n, m, float(radius), nsample, xyz1, xyz2, idx)
    return idx


def ref_group_point(points, idx, out=None):
    I = C.c_int
    fn = _reffn("libref_grouping.so", "_Z15group_point_cpuiiiiiPKfPKiPf",
                [I, I, I, I, I, _f32p, _i32p, _f32p])
    points, idx = _f(points), _i(idx)
    b, n, c = points.shape
    _, m, s = idx.shape
    out = np.empty((b, m, s, c), np.float32) if out is None else out   # out: caller-owned buffer (bench.py's sharded CPU leg)
    fn(b, n, c, m, s, points, idx, out)
    return out


def ref_group_point_grad(points_shape, idx, grad_out):
    I = C.c_int
    fn = _reffn("libref_grouping.so", "_Z20group_point_grad_cpuiiiiiPKfPKiPf",
                [I, I, I, I, I, _f32p, _i32p, _f32p])
    idx, grad_out = _i(idx), _f(grad_out)
    b, n, c = points_shape
    _, m, s = idx.shape
    g = np.zeros((b, n, c), np.float32)
    fn(b, n, c, m, s, grad_out, idx, g)
    return g


def ref_select_top_k(k, dist):
    I = C.c_int
    fn = _reffn("libref_selsort.so", "_Z18selection_sort_cpuiiiiPKfPiPf",
                [I, I, I, I, _f32p, _i32p, _f32p])
    dist = _f(dist)
    b, m, n = dist.shape
    outi = np.zeros((b, m, n), np.int32)
    out = np.zeros((b, m, n), np.float32)
    with _silence_stdout():
        fn(b, n, m, k, dist, outi, out)
    return outi, out


def ref_three_nn(xyz1, xyz2):
    I = C.c_int
    fn = _reffn("libref_interp.so", "_Z11threenn_cpuiiiPKfS0_PfPi",
                [I, I, I, _f32p, _f32p, _f32p, _i32p])
    xyz1, xyz2 = _f(xyz1), _f(xyz2)
    b, n, _ = xyz1.shape
    m = xyz2.shape[1]
    dist = np.empty((b, n, 3), np.float32)
    idx = np.empty((b, n, 3), np.int32)
    fn(b, n, m, xyz1, xyz2, dist, idx)
    return dist, idx


def ref_three_interpolate(points, idx, weight):
    I = C.c_int
    fn = _reffn("libref_interp.so", "_Z20threeinterpolate_cpuiiiiPKfPKiS0_Pf",
                [I, I, I, I, _f32p, _i32p, _f32p, _f32p])
    points, idx, weight = _f(points), _i(idx), _f(weight)
    b, m, c = points.shape
    n = idx.shape[1]
    out = np.empty((b, n, c), np.float32)
    fn(b, m, c, n, points, idx, weight, out)
    return out


def ref_three_interpolate_grad(points_shape, idx, weight, grad_out):
    I = C.c_int
    fn = _reffn("libref_interp.so", "_Z25threeinterpolate_grad_cpuiiiiPKfPKiS0_Pf",
                [I, I, I, I, _f32p, _i32p, _f32p, _f32p])
    idx, weight, grad_out = _i(idx), _f(weight), _f(grad_out)
    b, m, c = points_shape
    n = idx.shape[1]
    g = np.zeros((b, m, c), np.float32)
    fn(b, n, c, m, grad_out, idx, weight, g)
    return g
