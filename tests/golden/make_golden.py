#!/usr/bin/env python3
"""Generate the golden vectors in tests/golden/*.npz.

Runs ONLY in the build container: it calls the reference's own CPU functions,
compiled in place from /root/reference by oracle/Makefile into oracle/_ref/
(query_ball_point_cpu, group_point_cpu, group_point_grad_cpu -- grouping/test/
query_ball_point.cpp:19-84; selection_sort_cpu -- grouping/test/selection_sort.cpp:
20-63; threenn_cpu, threeinterpolate_cpu, threeinterpolate_grad_cpu --
3d_interpolation/tf_interpolate.cpp:60-153).  The .npz files hold inputs and the
reference's outputs only (data, no source).  Re-run:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402
from scanobjectnn_amd.synth import synth_clouds  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def lattice(b, n, seed, side=5, step=0.125):
    """integer-lattice points (exactly representable) -> many exact distance ties
    and duplicate points."""
    rng = np.random.default_rng(seed)
    return (rng.integers(0, side, (b, n, 3)) * step - 0.25).astype(np.float32)


def qbp_cases():
    cases = {}
    # (name, xyz1, xyz2, radius, nsample)
    c = synth_clouds(2, 256, seed=0, kind="surface")
    cases["surface_subset_r0.2_s32"] = (c, c[:, ::4].copy(), 0.2, 32)       # queries in dataset
    cases["surface_subset_r0.4_s8"] = (c, c[:, ::4].copy(), 0.4, 8)         # cnt >= S early stop
    c = synth_clouds(2, 200, seed=1, kind="ball")                          # non multiple of 64
    q = synth_clouds(2, 37, seed=2, kind="ball")
    cases["ball_foreign_r0.15_s16"] = (c, q, 0.15, 16)                     # cnt < S padding
    cases["ball_foreign_r0.02_s4"] = (c, q * 3.0, 0.02, 4)                 # zero-hit rows
    l = lattice(2, 192, seed=3)
    cases["lattice_r0.25_s12"] = (l, l[:, :48].copy(), 0.25, 12)           # d == r ties, duplicates
    cases["lattice_r0.125_s64"] = (l, l[:, 5:69].copy(), 0.125, 64)
    c = synth_clouds(1, 64, seed=4, kind="ball")
    cases["single_query_s1"] = (c, c[:, 7:8].copy(), 0.3, 1)               # M=1, S=1
    cases["tiny_radius"] = (c, c[:, :16].copy(), 1e-21, 4)                 # radius <= 1e-20 -> no hits
    return cases


def main():
    assert O.have_ref(), "run `make -C oracle` in the build container first"
    rng = np.random.default_rng(0)

    # ---- query_ball_point (idx only: the CPU twin has no pts_cnt) -------------
    d = {}
    for name, (x1, x2, r, s) in qbp_cases().items():
        d[name + "/xyz1"], d[name + "/xyz2"] = x1, x2
        d[name + "/radius"], d[name + "/nsample"] = np.float32(r), np.int32(s)
        d[name + "/idx"] = O.ref_query_ball_point(r, s, x1, x2)
    np.savez_compressed(os.path.join(OUT, "query_ball_point.npz"), **d)

    # ---- group_point / group_point_grad -----------------------------------------
    d = {}
    for name, (b, n, c, m, s) in {"c3": (2, 128, 3, 32, 8), "c16": (1, 128, 16, 8, 32),
                                  "c67": (2, 50, 67, 9, 5)}.items():
        pts = rng.standard_normal((b, n, c)).astype(np.float32)
        idx = rng.integers(0, n, (b, m, s)).astype(np.int32)
        go = rng.standard_normal((b, m, s, c)).astype(np.float32)
        d[name + "/points"], d[name + "/idx"], d[name + "/grad_out"] = pts, idx, go
        d[name + "/out"] = O.ref_group_point(pts, idx)
        d[name + "/grad_points"] = O.ref_group_point_grad(pts.shape, idx, go)
    np.savez_compressed(os.path.join(OUT, "group_point.npz"), **d)

    # ---- selection_sort: the reference's own known-answer case + random + ties --
    d = {}
    ka = (10 - np.arange(16)).astype(np.float32).reshape(2, 2, 4)   # selection_sort.cpp:68-78
    d["known_answer/dist"], d["known_answer/k"] = ka, np.int32(3)
    d["known_answer/outi"], d["known_answer/out"] = O.ref_select_top_k(3, ka)
    rd = rng.random((2, 5, 40)).astype(np.float32)
    d["random/dist"], d["random/k"] = rd, np.int32(7)
    d["random/outi"], d["random/out"] = O.ref_select_top_k(7, rd)
    td = rng.integers(0, 4, (2, 6, 24)).astype(np.float32)           # heavy ties -> unstable order
    d["ties/dist"], d["ties/k"] = td, np.int32(10)
    d["ties/outi"], d["ties/out"] = O.ref_select_top_k(10, td)
    np.savez_compressed(os.path.join(OUT, "selection_sort.npz"), **d)

    # ---- three_nn / three_interpolate(+grad) -------------------------------------
    d = {}
    c = synth_clouds(2, 160, seed=5, kind="surface")
    cases = {
        "subset": (c, c[:, ::5].copy()),
        "m1": (c[:, :40].copy(), np.zeros((2, 1, 3), np.float32)),     # BGA fa_layer1 (m=1)
        "m2": (c[:, :40].copy(), c[:, 3:5].copy()),
        "lattice": (lattice(2, 96, 6), lattice(2, 24, 7)),              # exact ties
    }
    for name, (x1, x2) in cases.items():
        dist, idx = O.ref_three_nn(x1, x2)
        d[name + "/xyz1"], d[name + "/xyz2"] = x1, x2
        d[name + "/dist"], d[name + "/idx"] = dist, idx
        m, ch = x2.shape[1], 19
        pts = rng.standard_normal((x2.shape[0], m, ch)).astype(np.float32)
        w = rng.random(dist.shape).astype(np.float32)
        w /= w.sum(axis=2, keepdims=True)
        go = rng.standard_normal((x1.shape[0], x1.shape[1], ch)).astype(np.float32)
        d[name + "/points"], d[name + "/weight"], d[name + "/grad_out"] = pts, w, go
        d[name + "/out"] = O.ref_three_interpolate(pts, idx, w)
        d[name + "/grad_points"] = O.ref_three_interpolate_grad(pts.shape, idx, w, go)
    np.savez_compressed(os.path.join(OUT, "three_interp.npz"), **d)

    for f in sorted(os.listdir(OUT)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(OUT, f)), "bytes")


if __name__ == "__main__":
    main()
