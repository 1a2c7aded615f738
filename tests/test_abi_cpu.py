"""CPU-side checks of the drop-in boundary: libpcops.so loads without a GPU and exports EVERY symbol that
include/pcops.h declares; the Python binding table matches the header; the product path refuses CPU tensors
(no fallback) and never imports the oracle."""
import ctypes
import os
import re
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "pcops.h")


def declared_symbols():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(pcops_[a-z0-9_]+)\s*\(", text)) - {"pcops_status"})


def test_library_exports_every_declared_symbol():
    from scanobjectnn_amd import _lib
    lib = _lib.load()
    syms = declared_symbols()
    assert len(syms) >= 35
    for s in syms:
        assert hasattr(lib, s), "libpcops.so does not export %s (declared in include/pcops.h)" % s
    assert lib.pcops_abi_version() >= 4
    assert _lib.strerror(0) == "ok" and "null" in _lib.strerror(-1)


def test_binding_table_covers_header():
    from scanobjectnn_amd import _lib
    bound = set(_lib.SIGNATURES) | set(_lib.PLAIN)
    assert set(declared_symbols()) == bound


def test_argument_validation_without_gpu():
    """status codes come back before any launch: NULL pointers / bad attributes (OP_REQUIRES of the reference)"""
    from scanobjectnn_amd import _lib
    lib = _lib.load()
    assert lib.pcops_query_ball_point(1, 8, 4, ctypes.c_float(-1.0), 4, None, None, None, None, None) == -3
    assert lib.pcops_query_ball_point(1, 8, 4, ctypes.c_float(0.1), 0, None, None, None, None, None) == -3
    assert lib.pcops_query_ball_point(1, 8, 4, ctypes.c_float(0.1), 4, None, None, None, None, None) == -1
    assert lib.pcops_farthest_point_sample(1, 8, 0, None, None, None, None) == -3
    assert lib.pcops_knn_graph(1, 8, 3, 9, None, None, None) == -3          # k > n
    assert lib.pcops_prob_sample(2, 0, 4, None, None, None, None, None) == -2     # the reference reads cumsum[n-1]
    assert lib.pcops_prob_sample(2, 8, 4, None, None, None, None, None) == -1
    assert lib.pcops_prob_sample(0, 8, 4, None, None, None, None, None) == 0
    assert lib.pcops_farthest_point_sample_workspace_bytes(32, 2048) == 0
    assert lib.pcops_mlp_stats_rows(4194304) == 512 and lib.pcops_mlp_stats_rows(100) == 1
    assert lib.pcops_adam_step(ctypes.c_longlong(6), None, None, None, None, ctypes.c_float(0.9), ctypes.c_float(0.999),
                               ctypes.c_float(1e-3), ctypes.c_float(1e-8), None) == -2        # n % 4 != 0
    assert lib.pcops_adam_step(ctypes.c_longlong(8), None, None, None, None, ctypes.c_float(0.9), ctypes.c_float(0.999),
                               ctypes.c_float(1e-3), ctypes.c_float(1e-8), None) == -1
    assert lib.pcops_mlp_bwd_fused_groups(ctypes.c_longlong(1 << 22), 64, 128, 32, 1) in (0, 256)   # (0: PCOPS_BWD_FUSED=0)
    assert lib.pcops_mlp_bwd_fused_groups(ctypes.c_longlong(1 << 22), 128, 128, 32, 1) == 0
    # the ordered scatter-add says what it supports (LDS-resident counting sort), callers ask before choosing it
    lim = lib.pcops_scatter_rows_sorted_max_ndst()
    assert lim == 19968
    assert lib.pcops_scatter_rows_sorted_supported(3 * 4096, lim) == 1
    assert lib.pcops_scatter_rows_sorted_supported(3 * 4096, lim + 1) == 0
    assert lib.pcops_scatter_rows_sorted_supported(1 << 30, 64) == 0


def test_compacted_stack_support_is_one_library_answer():
    """`fused_mlp._compactable` asks the library whether EVERY launch of a stack on compacted rows has a kernel (the
    *_rows entry points have no tiled fallback) instead of restating the kernels' shape conditions in Python"""
    from scanobjectnn_amd import _lib
    lib = _lib.load()

    def q(b, n, m, s, has_q, widths):
        arr = (ctypes.c_int * len(widths))(*widths)
        return lib.pcops_gather_stack_rows_supported(b, n, m, s, has_q, len(widths), arr)

    assert q(256, 512, 128, 64, 1, [128, 128, 256]) == 1          # SA2 of the SSG config
    assert q(128, 2048, 512, 64, 0, [64, 64, 128]) == 1           # BGA's SA1: coordinate-only first layer
    assert q(256, 512, 128, 60, 1, [128, 128, 256]) == 0          # groups are whole 16-row blocks
    assert q(256, 512, 128, 512, 1, [128, 128, 256]) == 0         # 8-bit arg index
    assert q(2, 100, 10, 64, 1, [128, 128]) == 0                  # too few rows for the wave-stream kernels
    assert q(256, 512, 128, 64, 1, [100, 128, 256]) == 0          # the inverse-index walk wants c1 in {32, 64, 128, 256 k}
    assert q(256, 20000, 128, 64, 1, [128, 128, 256]) == 0        # ... and the cloud's histogram in LDS
    assert q(256, 512, 128, 64, 1, [128, 130, 256]) == 0          # K % 8 of a middle layer
    assert lib.pcops_sa_scatter_rows_supported(512, 128, 64, 128) == 1
    code = ("import os, ctypes; os.environ['PCOPS_GEMM_WS'] = '0'; from scanobjectnn_amd import _lib; lib = _lib.load(); "
            "arr = (ctypes.c_int * 3)(128, 128, 256); print(lib.pcops_gather_stack_rows_supported(256, 512, 128, 64, 1, 3, arr))")
    out = subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True)
    assert out.stdout.strip() == "0", out.stderr                  # wave-stream kernels switched off: do not compact


def test_no_cpu_fallback():
    from scanobjectnn_amd import _lib
    from scanobjectnn_amd.pointnet2 import tf_grouping, tf_sampling
    x = torch.zeros((1, 16, 3))
    with pytest.raises(_lib.PcopsError):
        tf_sampling.farthest_point_sample(4, x)
    with pytest.raises(_lib.PcopsError):
        tf_grouping.query_ball_point(0.1, 4, x, x)


def test_product_never_imports_the_oracle():
    """only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may touch oracle/"""
    pkg = os.path.join(ROOT, "scanobjectnn_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                text = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", text, flags=re.M), os.path.join(dirpath, f)
                assert "liboracle" not in text, os.path.join(dirpath, f)
    code = "import sys; import scanobjectnn_amd.pointnet2.pointnet2_cls_ssg, scanobjectnn_amd.dgcnn.dgcnn; " \
           "print(any(m == 'oracle' or m.startswith('oracle.') for m in sys.modules))"
    out = subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True)
    assert out.stdout.strip() == "False", out.stderr
