"""ctypes binding of libpcops.so (the C ABI of include/pcops.h).

The product path is HIP-only: there is NO CPU fallback.  A missing library, a CPU
tensor or a non-zero status raises immediately (the CPU restatement under oracle/ is
test infrastructure and is never imported from here).
"""
import ctypes as C
import os
import weakref

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PCOPS_LIB") or os.path.join(_HERE, "libpcops.so")      # PCOPS_LIB: A/B runs of two builds

_I, _F, _P, _U64, _LL = C.c_int, C.c_float, C.c_void_p, C.c_ulonglong, C.c_longlong
ABI_VERSION = 4      # pcops_abi_version() of the library this binding matches (include/pcops.h), checked in load()

# name -> (argtypes without the trailing stream, has_stream)
SIGNATURES = {
    "pcops_farthest_point_sample": ([_I, _I, _I, _P, _P, _P], True),
    "pcops_prob_sample": ([_I, _I, _I, _P, _P, _P, _P], True),
    "pcops_gather_point": ([_I, _I, _I, _P, _P, _P], True),
    "pcops_gather_point_grad": ([_I, _I, _I, _P, _P, _P], True),
    "pcops_query_ball_point": ([_I, _I, _I, _F, _I, _P, _P, _P, _P], True),
    "pcops_query_ball_point_multi": ([_I, _I, _I, _I, _P, _P, _P, _P, _P, _P], True),
    "pcops_group_point": ([_I, _I, _I, _I, _I, _P, _P, _P], True),
    "pcops_group_point_grad": ([_I, _I, _I, _I, _I, _P, _P, _P], True),
    "pcops_selection_sort": ([_I, _I, _I, _I, _P, _P, _P], True),
    "pcops_three_nn": ([_I, _I, _I, _P, _P, _P, _P], True),
    "pcops_three_interpolate": ([_I, _I, _I, _I, _P, _P, _P, _P], True),
    "pcops_three_interpolate_grad": ([_I, _I, _I, _I, _P, _P, _P, _P], True),
    "pcops_pairwise_distance": ([_I, _I, _I, _P, _P], True),
    "pcops_knn_topk": ([_I, _I, _I, _P, _P], True),
    "pcops_knn_graph": ([_I, _I, _I, _I, _P, _P], True),
    "pcops_knn_graph_seeded": ([_I, _I, _I, _I, _P, _P, _P], True),
    "pcops_edge_feature": ([_I, _I, _I, _I, _P, _P, _P], True),
    "pcops_edge_feature_grad": ([_I, _I, _I, _I, _P, _P, _P], True),
    "pcops_mlp_gemm_fwd": ([_I, _I, _I, _P, _I, _P, _P, _P, _P, _P, _P, _P], True),
    "pcops_mlp_bn_finalize": ([_I, _I, _LL, _P, _P, _P, _P, _P, _F, _F, _I, _P, _P, _P, _P, _P, _P], True),
    "pcops_mlp_bn_eval_coeffs": ([_I, _P, _P, _P, _P, _F, _P, _P], True),
    "pcops_mlp_bn_relu_maxpool": ([_LL, _I, _I, _P, _P, _P, _P, _P, _P], True),
    "pcops_mlp_gemm_fwd_pool": ([_I, _I, _I, _I, _P, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P], True),
    "pcops_mlp_pool_select": ([_LL, _I, _P, _P, _P, _P], True),
    "pcops_mlp_bn_relu_apply": ([_LL, _I, _P, _P, _P, _P], True),
    "pcops_mlp_relu_mask_stats": ([_LL, _I, _P, _P, _P, _P, _P, _P], True),
    "pcops_mlp_pool_bwd_stats": ([_LL, _I, _P, _P, _P, _P, _P, _P], True),
    "pcops_mlp_pool_bwd_stats_sum": ([_LL, _I, _P, _LL, _P, _LL, _P, _P, _P, _P, _P], True),
    "pcops_mlp_bn_bwd_coeffs": ([_I, _I, _LL, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P], True),
    "pcops_mlp_gemm_dgrad": ([_I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _I, _P, _P, _P, _P, _P, _P, _P, _P], True),
    "pcops_mlp_wgrad": ([_LL, _I, _I, _P, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _P, _P, _P, _P, _P], True),
    "pcops_mlp_transpose": ([_I, _I, _P, _P], True),
    "pcops_small_gemm": ([_I, _I, _I, _P, _I, _P, _I, _P, _I], True),
    "pcops_small_gemm_ex": ([_I, _I, _I, _P, _I, _I, _P, _I, _I, _P, _P, _I], True),
    "pcops_mlp_dy_apply": ([_LL, _I] + [_P] * 6, True),
    "pcops_small_gemm_colsum": ([_I, _I, _I, _P, _I, _I, _P, _I, _I, _P, _P, _I, _P], True),
    "pcops_small_gemm_pair": ([_P], True),
    "pcops_mlp_pool_top_prep": ([_I, _I] + [_P] * 8, True),
    "pcops_mlp_pool_top_finish": ([_I, _I, _LL] + [_P] * 10, True),
    "pcops_mlp_pool_top_addend": ([_I, _I, _I, _I] + [_P] * 9, True),
    "pcops_mlp_gemm_dgrad_top": ([_I, _I] + [_P] * 6 + [_LL] + [_P] * 3, True),
    "pcops_mlp_gram": ([_LL, _I, _P, _I, _P, _P, _P, _P, _P], True),
    "pcops_mlp_pool_top_wsparse": ([_I, _I, _I, _I] + [_P] * 11, True),
    "pcops_sa_gather_fwd": ([_I, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P], True),
    "pcops_mlp_gemm_fwd_xyz": ([_I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P], True),
    "pcops_mlp_gemm_dgrad_xyz": ([_I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P], True),
    "pcops_mlp_wgrad_xyz": ([_LL, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _P, _P, _P, _P, _P], True),
    "pcops_edge_pool_fwd": ([_I, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P], True),
    "pcops_edge_pool_out": ([_LL, _I, _P, _P, _P, _P, _P, _P], True),
    "pcops_edge_pool_bwd": ([_I, _I, _I, _I, _I] + [_P] * 15, True),
    "pcops_xyz_first_layer_grads": ([_I, _P, _I, _P, _I, _P, _P, _P, _P, _P, _P, _P, _LL, _P, _P], True),
    "pcops_cloud_bias_fwd": ([_LL, _I, _I, _P, _P, _P, _P, _P], True),
    "pcops_cloud_bias_bwd": ([_LL, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P], True),
    "pcops_edge_first_moments": ([_I, _I, _I, _I, _P, _P, _P, _P], True),
    "pcops_mlp_bwd_fused_edge": ([_LL, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _P, _P, _P, _P, _P, _P, _P], True),
    "pcops_edge_first_wgrad": ([_I, _I, _I, _I, _I, _P, _P, _P, _P], True),
    "pcops_edge_first_layer_grads": ([_I, _P, _I, _P, _I, _P, _P, _P, _P, _P, _P, _P, _LL, _P, _P], True),
    "pcops_sa_scatter_bwd": ([_I, _I, _I, _I, _I] + [_P] * 22, True),
    # ---- compacted rows (pcops.h "compacted rows"): the suffix-less signature + a pcops_rows_t* before the stream
    "pcops_rows_plan": ([_I, _I, _I, _P, _P, _P, _P], True),
    "pcops_mlp_gemm_fwd_rows": ([_I, _I, _I, _P, _I, _P, _P, _P, _P, _P, _P, _P, _P], True),
    "pcops_mlp_gemm_fwd_xyz_rows": ([_I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P], True),
    "pcops_mlp_gemm_dgrad_rows": ([_I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P], True),
    "pcops_mlp_gemm_dgrad_xyz_rows": ([_I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P], True),
    "pcops_mlp_bwd_fused": ([_LL, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _P, _P, _P, _P, _P, _P], True),
    "pcops_mlp_bwd_fused_rows": ([_LL, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _P, _P, _P, _P, _P, _P, _P], True),
    "pcops_adam_step": ([_LL, _P, _P, _P, _P, _F, _F, _F, _F], True),
    # round 6: the one-pass backward of a pooled layer with the weight gradient in its Gram form (+ bias)
    "pcops_mlp_bwd_fused_gw": ([_LL, _I, _I] + [_P] * 9 + [_I] + [_P] * 7, True),
    "pcops_mlp_bwd_fused_edge_gw": ([_LL, _I, _I] + [_P] * 9 + [_I] + [_P] * 8, True),
    "pcops_mlp_bwd_fused_xyz_rows": ([_LL, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _P, _P, _P, _P, _P, _P, _P], True),
    "pcops_mlp_wgrad_rows": ([_LL, _I, _I, _P, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _P, _P, _P, _P, _P, _P], True),
    "pcops_mlp_wgrad_xyz_rows": ([_LL, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _P, _P, _P, _P, _P, _P], True),
    "pcops_mlp_bn_relu_maxpool_rows": ([_LL, _I, _P, _P, _P, _P, _P, _P, _P], True),
    "pcops_mlp_gemm_fwd_pool_rows": ([_I, _I, _I, _P, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P], True),
    "pcops_mlp_pool_combine_rows": ([_LL, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P], True),
    "pcops_sa_gather_fwd_rows": ([_I, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P], True),
    "pcops_sa_scatter_bwd_rows": ([_I, _I, _I, _I, _I] + [_P] * 23, True),
    "pcops_knn_point": ([_I, _I, _I, _I, _I, _P, _P, _P, _P], True),
    "pcops_knn_point_dist": ([_I, _I, _I, _I, _P, _P, _P], True),
    "pcops_scatter_rows_sorted": ([_I, _I, _I, _I, _I, _I, _P, _P, _P, _P, _I, _P], True),
    "pcops_edge_feature_grad_central": ([_I, _I, _I, _I, _P, _P], True),
    "pcops_edge_pool_fwd_ld": ([_I, _I, _I, _I, _I, _P, _I, _P, _I, _P, _P, _P, _P, _P, _P, _P], True),
    "pcops_edge_pool_out_ld": ([_LL, _I, _P, _P, _I, _P, _P, _P, _P], True),
    "pcops_edge_pool_out_ld2": ([_LL, _I, _P, _P, _I, _P, _P, _P, _P, _P, _I], True),
    "pcops_edge_pool_bwd_ld": ([_I, _I, _I, _I, _I, _P, _I, _P, _I] + [_P] * 10 + [_P, _I, _P, _I, _P], True),
    "pcops_sa_gather_fwd_ld": ([_I, _I, _I, _I, _I, _P, _I, _P, _I, _P, _P, _P, _P], True),
    "pcops_sa_scatter_bwd_ld": ([_I, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _I, _P, _I, _P, _I, _P, _I, _P], True),
    "pcops_edge_weights_fwd": ([_I, _I, _I, _P, _P, _P, _P], True),
    "pcops_edge_weights_bwd": ([_I, _I, _P, _P, _P, _P], True),
    "pcops_transform3_fwd": ([_I, _I, _P, _P, _P], True),
    "pcops_transform3_bwd": ([_I, _I, _P, _P, _P, _P, _P], True),
    "pcops_fc_bn_fwd": ([_I, _I, _P, _P, _P, _P, _P, _I, _F, _F, _I, _I, _P, _P, _P], True),
    "pcops_fc_bn_bwd": ([_I, _I, _P, _P, _P, _P, _P, _P, _I, _I, _P, _P, _P], True),
    "pcops_softmax_ce": ([_I, _I, _P, _P, _F, _P, _P], True),
    "pcops_three_nn_weights": ([_I, _I, _P, _P], True),
}
PLAIN = {
    "pcops_strerror": ([_I], C.c_char_p),
    "pcops_abi_version": ([], _I),
    "pcops_softmax_ce_blocks": ([_I], _I),
    "pcops_farthest_point_sample_workspace_bytes": ([_I, _I], _U64),
    "pcops_mlp_stats_rows": ([_I], _I),
    "pcops_mlp_reduce_workspace_bytes": ([_I], _U64),
    "pcops_mlp_gemm_fwd_pool_supported": ([_I, _I, _I, _I], _I),
    "pcops_mlp_xyz_supported": ([_I, _I, _I], _I),
    "pcops_mlp_bwd_stats_rows": ([_LL], _I),
    "pcops_mlp_bwd_pool_stats_rows": ([_LL], _I),
    "pcops_mlp_wgrad_splits": ([_LL, _I, _I], _I),
    "pcops_mlp_bwd_fused_groups": ([_LL, _I, _I, _I, _I], _I),
    "pcops_mlp_bwd_fused_gw_groups": ([_LL, _I, _I, _I], _I),
    "pcops_gather_stack_rows_supported": ([_I, _I, _I, _I, _I, _I, _P], _I),
    "pcops_sa_scatter_rows_supported": ([_I, _I, _I, _I], _I),
    "pcops_sa_gather_stats_rows": ([_LL], _I),
    "pcops_sa_gather_fwd_stats_rows": ([_I] * 9, _I),
    "pcops_sa_scatter_rows": ([_I, _I], _I),
    "pcops_edge_pool_stats_rows": ([_LL], _I),
    "pcops_edge_pool_fwd_stats_rows": ([_I] * 5, _I),
    "pcops_edge_ld_supported": ([_I] * 5, _I),
    "pcops_edge_first_rows": ([], _I),
    "pcops_cloud_bias_supported": ([_LL, _I, _I], _I),
    "pcops_cloud_bias_rows": ([_LL], _I),
    "pcops_edge_first_supported": ([_I] * 5, _I),
    "pcops_sa_scatter_workspace_bytes": ([_I, _I, _I, _I], _U64),
    "pcops_rows_max_blocks": ([_I, _I, _I], _U64),
    "pcops_mlp_gemm_fwd_pool_rows_supported": ([_I, _I, _I], _I),
    "pcops_scatter_rows_workspace_bytes": ([_I, _I, _I], _U64),
    "pcops_knn_point_supported": ([_I], _I),
    "pcops_scatter_rows_sorted_max_ndst": ([], _I),
    "pcops_scatter_rows_sorted_supported": ([_I, _I], _I),
    "pcops_mlp_pool_top_supported": ([_I, _I, _I, _I], _I),
    "pcops_knn_graph_path": ([_I, _I, _I, _I, _P], _I),
    "pcops_last_launch_pipe": ([], _I),
    "pcops_set_option": ([_I, _I], _I),
    "pcops_get_option": ([_I], _I),
    "pcops_set_deterministic": ([_I], None),
    "pcops_get_deterministic": ([], _I),
}


class GemmProblem(C.Structure):
    """pcops_gemm_problem_t: one product C = op(A) op(B) + bias (+ column sums of op(B)) of pcops_small_gemm_pair"""
    _fields_ = [("M", C.c_int), ("K", C.c_int), ("N", C.c_int), ("A", C.c_void_p), ("lda", C.c_int), ("transA", C.c_int),
                ("B", C.c_void_p), ("ldb", C.c_int), ("transB", C.c_int), ("bias", C.c_void_p), ("C", C.c_void_p),
                ("ldc", C.c_int), ("colsum", C.c_void_p)]


def small_gemm_pair(p0, p1):
    """two independent small products in one launch; p = (M, K, N, A, lda, transA, B, ldb, transB, bias, C, ldc, colsum) with device
    pointers as integers or None"""
    arr = (GemmProblem * 2)(GemmProblem(*p0), GemmProblem(*p1))
    lib = load()
    stream = torch.cuda.current_stream().cuda_stream
    shape = tuple(int(v) for v in p0[:3]) + tuple(int(v) for v in p1[:3])     # what a profiling hook sees of the launch
    for h in _hooks:
        h("pcops_small_gemm_pair", "pre", shape)
    status = lib.pcops_small_gemm_pair(arr, stream)
    for h in _hooks:
        h("pcops_small_gemm_pair", "post", shape)
    if status != 0:
        raise PcopsError("pcops_small_gemm_pair failed: %s (status %d)" % (strerror(status), status))


class RowsT(C.Structure):
    """pcops_rows_t: device pointers of a compacted row set (blocks, block_start, rows)"""
    _fields_ = [("blocks", C.c_void_p), ("block_start", C.c_void_p), ("rows", C.c_void_p)]


class Rows:
    """A compacted row set of one grouped stack (pcops.h "compacted rows"): owns the three device buffers and the
    host-side struct the *_rows entry points take."""
    by_struct = weakref.WeakValueDictionary()

    def __init__(self, pts_cnt, nsample):
        b, m = pts_cnt.shape
        lib = load()
        nb = int(lib.pcops_rows_max_blocks(b, m, int(nsample)))
        dev = pts_cnt.device
        self.blocks = torch.empty((nb, 4), dtype=torch.int32, device=dev)
        self.block_start = torch.empty(b * m + 1, dtype=torch.int32, device=dev)
        self.rows = torch.empty(1, dtype=torch.int32, device=dev)
        call("pcops_rows_plan", b, m, int(nsample), pts_cnt.data_ptr(), self.blocks.data_ptr(),
             self.block_start.data_ptr(), self.rows.data_ptr())
        self.struct = RowsT(self.blocks.data_ptr(), self.block_start.data_ptr(), self.rows.data_ptr())
        self.full_rows = b * m * int(nsample)
        Rows.by_struct[id(self.struct)] = self          # lets a profiling hook find the owner of a byref() argument

    def num_rows(self):
        """rows actually computed (device -> host copy: profiling / tests only)"""
        return int(self.rows.item())

    @property
    def ref(self):
        return C.byref(self.struct)


_lib = None


class PcopsError(RuntimeError):
    pass


def load():
    """dlopen libpcops.so; raises if it has not been built (python __graft_entry__.py)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise PcopsError(
            "libpcops.so is missing (%s). Build it with `python -c 'import __graft_entry__ as g; "
            "g.build()'` or `make -C scanobjectnn_amd/csrc`. There is no CPU fallback." % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, (argtypes, _) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.argtypes = list(argtypes) + [_P]
        fn.restype = _I
    for name, (argtypes, restype) in PLAIN.items():
        fn = getattr(lib, name)
        fn.argtypes = argtypes
        fn.restype = restype
    got = int(lib.pcops_abi_version())
    if got != ABI_VERSION:
        # the signatures above are positional: a stale or newer library would take the wrong arguments SILENTLY (round 3
        # inserted stat_pivot mid-signature in five entry points) -- refuse it instead
        raise PcopsError("%s reports ABI version %d, this binding is written for %d: rebuild it "
                         "(make -C scanobjectnn_amd/csrc)" % (LIB_PATH, got, ABI_VERSION))
    _lib = lib
    return lib


def strerror(status):
    return load().pcops_strerror(int(status)).decode()


def ptr(t):
    """device pointer of a tensor that already passed check()"""
    return t.data_ptr() if t is not None else None


def check(t, dtype, name, ndim=None):
    if not isinstance(t, torch.Tensor):
        raise TypeError("%s must be a torch.Tensor" % name)
    if not t.is_cuda:
        raise PcopsError("%s is on %s: pcops ops run on the MI355X only (no CPU fallback)"
                         % (name, t.device))
    if t.dtype != dtype:
        raise TypeError("%s must be %s, got %s" % (name, dtype, t.dtype))
    if ndim is not None and t.dim() != ndim:
        raise ValueError("%s must have rank %d, got shape %s" % (name, ndim, tuple(t.shape)))
    return t.contiguous()


OPT_GEMM_SPLIT_BF16, OPT_WGRAD_SPLIT_BF16, OPT_BWD_FUSED_DX_SPLIT_BF16, OPT_KNN_F16_PREFILTER = 1, 2, 3, 4
OPT_DGRAD_SPLIT_BF16, OPT_BWD_FUSED_GRAM_WGRAD = 5, 6            # round 6


def set_option(option, value):
    """pcops_set_option (pcops.h "Arithmetic options"); returns the previous value"""
    prev = int(load().pcops_set_option(int(option), int(value)))
    if prev < 0:
        raise PcopsError("pcops_set_option(%d, %d): %s" % (option, value, strerror(prev)))
    return prev


def get_option(option):
    return int(load().pcops_get_option(int(option)))


def set_deterministic(on=True):
    """Bit-reproducible backward passes (pcops.h "deterministic backward passes"): every scatter-add is taken by one
    owner in ascending row order.  Process-wide; PCOPS_DETERMINISTIC=1 in the environment sets it at load time."""
    load().pcops_set_deterministic(1 if on else 0)


def deterministic():
    return bool(load().pcops_get_deterministic())


def scatter_rows_sorted_supported(rows, ndst):
    """the launcher's own predicate (LDS-resident counting sort: ndst <= 19 968, rows < 2^30): callers that may use
    the atomic form instead ask before choosing the ordered one"""
    return bool(load().pcops_scatter_rows_sorted_supported(int(rows), int(ndst)))


def scatter_rows_sorted(idx, src, ndst, div=1, w=None, out=None, c=None, ld=None, src_ptr=None):
    """out (B, ndst, C) = ordered scatter-add of the rows of src (pcops_scatter_rows_sorted).  idx (B, rows) int32;
    src (B, rows / div, C) unless c / ld / src_ptr describe a strided view; out given -> accumulate."""
    b, rows = idx.shape[0], idx[0].numel()
    if not scatter_rows_sorted_supported(rows, ndst):
        raise PcopsError("ordered scatter-add (deterministic backward): %d destination points per cloud, the limit is "
                         "%d (the per-cloud counting sort lives in LDS); rows %d must stay below 2^30.  Switch "
                         "deterministic mode off for clouds this large." % (ndst, load().pcops_scatter_rows_sorted_max_ndst(), rows))
    if c is None:
        c, ld = src.shape[-1], src.shape[-1]
    acc = out is not None
    if out is None:
        out = torch.empty((b, ndst, c), dtype=torch.float32, device=idx.device)
    ws = torch.empty(int(load().pcops_scatter_rows_workspace_bytes(b, rows, ndst)) // 8 + 1, dtype=torch.int64,
                     device=idx.device)
    call("pcops_scatter_rows_sorted", b, rows, ndst, c, div, ld, idx.data_ptr(), ptr(w),
         src_ptr if src_ptr is not None else src.data_ptr(), out.data_ptr(), 1 if acc else 0, ws.data_ptr())
    return out


_SYNC_EVERY_CALL = os.environ.get("PCOPS_SYNC", "0") == "1"
_hooks = []  # profiling hooks: callables (name, phase, args) with phase in {"pre", "post"}


def call(name, *args):
    """Invoke a stream-taking entry point on torch's current HIP stream."""
    lib = load()
    stream = torch.cuda.current_stream().cuda_stream
    for h in _hooks:
        h(name, "pre", args)
    status = getattr(lib, name)(*args, stream)
    for h in _hooks:
        h(name, "post", args)
    if _SYNC_EVERY_CALL and status == 0:      # debugging aid (PCOPS_SYNC=1): a device fault is reported at its launch
        try:
            torch.cuda.synchronize()
        except RuntimeError as e:
            raise PcopsError("%s: device fault (%s)" % (name, e))
    if status != 0:
        raise PcopsError("%s failed: %s (status %d)" % (name, strerror(status), status))
