#!/usr/bin/env python3
"""Algorithmic vs counted HBM bytes per launch of the cfg3 (DGCNN) and cfg5 (MSG) steps' largest kernels.
Counted = FETCH_SIZE x 2 + WRITE_SIZE per launch from tools/collect_traffic.py --model ... (separate rocprofv3 --pmc passes,
MI355X_MICROARCH.md corrections); algorithmic = the tensors the kernel has to touch once.
  python tools/traffic_table.py profiles/r05_pmc_traffic_detail_dgcnn.json dgcnn > profiles/r05_pmc_traffic_dgcnn.json"""
import json
import sys

B, N, K = 256, 2048, 20
G = B * N
R = G * K
F = 4


def dgcnn_rows():
    ec = lambda c: F * G * c          # one (B, N, c) fp32 tensor   # noqa: E731
    idx = F * G * K
    return [  # (kernel-name fragment, grid threads or None, what, algorithmic bytes)
        ("ec_fwd_lds_kernel", 256 * 4 * 1024, "EdgeConv forward, 64 wide: Q, Ctr in; SQ, qsel, arg out; idx", 4 * ec(64) + G * 64 + idx),
        ("ec_fwd_lds_kernel", 256 * 8 * 1024, "EdgeConv forward, 128 wide", 4 * ec(128) + G * 128 + idx),
        ("edge_pool_out_ld_kernel", None, "EdgeConv out (+ the layer's block of the concatenation): averaged over 3 x 64 + 1 x 128 wide", int(5 * (3 * ec(64) + ec(128)) / 4)),
        ("ec_sparse_kernel", 256 * 4 * 1024, "EdgeConv backward, arg-row term + dCtr, 64 wide", 6 * ec(64) + G * 64 + idx),
        ("ec_sparse_kernel", 256 * 8 * 1024, "... 128 wide", 6 * ec(128) + G * 128 + idx),
        ("ec_walk_lds_kernel", 256 * 8 * 1024, "EdgeConv backward, owner walk, 64 wide: Q, Ctr, dQ in; dQ out; 16-bit list", 4 * ec(64) + R * 2 + 2 * F * G),
        ("ec_walk_lds_kernel", 256 * 16 * 1024, "... 128 wide", 4 * ec(128) + R * 2 + 2 * F * G),
        ("ec_csr_build_kernel", None, "inverse index: idx in; order (u32), codes (u16), start, perm out", idx + R * 4 + R * 2 + 2 * F * G),
        ("ec_fwd_kernel<1>", None, "T-Net first layer Y = Q[idx] + Ctr stored (10.5 M x 64)", F * R * 64 + 2 * ec(64) + idx),
        ("ec_tnet_ctr_kernel", None, "T-Net scatter, per-group output: G streamed, Q gathered", F * R * 64 + 3 * ec(64) + idx),
        ("ec_walk_kernel<true>", None, "T-Net scatter, owner walk: G rows gathered once", F * R * 64 + 3 * ec(64) + R * 4),
        ("bwd_fused_kernel<2, 3, false, false, true, true>", None, "T-Net one-pass backward 64 -> 128 (10.5 M rows), E^T Gprev reduced inside: Yprev, Y, 32 B of edge channels per row in; no Gprev", F * R * (64 + 128 + 8)),
        ("bwd_fused_kernel<2, 3, false, false, true>", None, "T-Net one-pass backward 64 -> 128 (10.5 M rows): Yprev, Y in; Gprev out (first half of the round)", F * R * (64 + 128 + 64)),
        ("gemm_ws_kernel<4, 1, 0, 32, 8, 4, 13>", 163840, "T-Net forward 64 -> 128 (10.5 M rows) with the k = 20 max-pool in its epilogue: Y1 in; Y2, extrema + arg out", F * R * (64 + 128) + 5 * G * 128),
        ("gemm_ws_kernel<4, 1, 0, 32, 8, 4, 4>", 163840, "T-Net forward 64 -> 128 (10.5 M rows), separate max-pool pass (first half of the round)", F * R * (64 + 128)),
        ("bn_relu_maxpool_kernel", 8388608, "T-Net max over k of the 128-wide layer (first half of the round; gone)", F * R * 128 + 2 * ec(128) + G * 128),
        ("edge_moments_kernel", None, "T-Net edge rows (32 B each) + their 27 moments: idx in, rows out", idx + 32 * R),
        ("gemm_ws_kernel<4, 0, 0, 32, 8, 4, 7>", None, "aggregation forward 320 -> 1024 with the pooled epilogue (Y not stored)", F * G * 320),
        ("gemm_ws_kernel<4, 0, 5, 64, 8, 2, 2>", None, "aggregation data gradient (algebraic form): X in, G out", 2 * F * G * 320),
        ("gram_full_kernel<10", None, "Gram matrix of the 320-wide input", F * G * 320),
        ("knn_f16_kernel", 1048576, "64-channel kNN graph (seeded): x, seed in; nn_idx out", F * G * 64 + 2 * idx),
        ("knn_mfma_kernel<4, 20>", 1048576, "coordinate kNN graph", F * G * 3 + idx),
        ("CatArrayBatchedCopy", None, "torch.cat of the four EdgeConv outputs (round 4; gone in round 5)", 2 * F * G * 320),
        ("edge_pool_fwd_kernel", None, "round-4 EdgeConv forward (averaged over 3 x 64 + 1 x 128 wide)", int((3 * (4 * ec(64) + G * 64 + idx) + 4 * ec(128) + G * 128 + idx) / 4)),
        ("edge_pool_bwd_dense_kernel<16", None, "round-4 dense walk, 64 wide", 4 * ec(64) + R * 8),
        ("edge_pool_bwd_dense_kernel<32", None, "round-4 dense walk, 128 wide", 4 * ec(128) + R * 8),
        ("sa_scatter_csr_kernel<16", None, "round-4 T-Net scatter (gather form): G and Y rows", 2 * F * R * 64 + ec(64) + R * 8),
        ("sa_scatter_lds_kernel", None, "round-4 T-Net scatter (streaming part): G and Y", 2 * F * R * 64 + ec(64)),
        ("sa_gather_fwd_kernel", 16777216, "round-4 T-Net first layer", F * R * 64 + 2 * ec(64) + idx),
    ]


def msg_rows():
    M1, M2 = 256 * 512 * 128, 256 * 128 * 128          # padded rows of the two widest scales (compaction leaves fewer)
    return [
        ("gemm_ws_kernel<3, 6, 1, 32, 8, 3, 4>", None, "SA1 scale 2 data gradient 128 -> 96, split operands, three column blocks (round 6; <= 16.8 M rows, compacted)", F * M1 * (128 + 96 + 96)),
        ("gemm_ws_kernel<3, 6, 1, 64, 8, 3, 0>", None, "SA1 scale 2 data gradient 128 -> 96 on the fp32 pipe (rounds 4-5)", F * M1 * (128 + 96 + 96)),
        ("gemm_ws_kernel<2, 6, 1, 32, 8, 2, 4>", None, "SA2 data gradients 256 -> 128, split operands, 64-column passes (round 6; several scales share the instantiation)", None),
        ("bwd_fused_kernel<2, 7", None, "SA1 scale 2 one-pass backward 64 -> 96 (arithmetic first layer: offsets, G and Y in)", F * M1 * (96 + 96 + 4)),
        ("wgrad_bf3_kernel<1, 6>", None, "weight gradients of the 128-wide layers (several shapes share the instantiation)", None),
        ("gemm_ws_kernel<4, 1, 0, 32, 8, 4, 5>", 262144, "forward products with the pooled epilogue (several shapes)", None),
        ("sa_scatter_csr_kernel<32", None, "SA2 scatter (gather form), 128 wide, S = 128 / 64", 2 * F * M2 * 128),
        ("sa_gather_fwd_kernel", 1048576, "SA2 first layers stored", F * M2 * 128),
    ]


def bga_rows():
    """cfg4 (pointnet2_cls_bga at its per-GPU batch of 128 clouds): the feature-propagation kernels (VERDICT r5 missing #4)
    and the largest MLP launches"""
    Bb = 128
    R1 = Bb * 512 * 64                 # SA1 rows before compaction (nsample 64)
    return [
        ("scatter_rows_sorted_kernel", None, "three_interpolate gradient (ordered owner walk): averaged over the three FP levels", None),
        ("three_nn_kernel", None, "three_nn: unknown + known clouds in, 3 distances + 3 indices out (largest level n 2048, m 512)", Bb * (12 * 2048 + 12 * 512 + 24 * 2048)),
        ("three_interpolate_kernel", None, "three_interpolate (largest level: C 128, n 2048, m 512)", Bb * (4 * 512 * 128 + 24 * 2048 + 4 * 2048 * 128)),
        ("bwd_fused_kernel<2, 6", None, "SA1 one-pass backward 64 -> 128 on compacted rows (<= 4.19 M): Yprev, Y in; Gprev out", F * R1 * (64 + 128 + 64)),
        ("bwd_fused_kernel<1, 7", None, "SA1 one-pass backward 64 -> 64 above the arithmetic first layer (compacted)", F * R1 * (64 + 64 + 4)),
        ("gemm_ws_kernel<2, 6, 1, 32, 8, 2, 4>", None, "SA2 data gradient 256 -> 128, split operands, 64-column passes (<= 1.05 M rows)", F * Bb * 128 * 64 * (256 + 128 + 128)),
        ("gemm_ws_kernel<4, 1, 0, 32, 8, 4, 5>", None, "forward products with the pooled epilogue (several shapes share the instantiation)", None),
        ("sa_scatter_csr_kernel<32", None, "SA2 scatter (gather form), 128 wide", 2 * F * Bb * 128 * 64 * 128),
    ]


def main():
    detail = json.load(open(sys.argv[1]))
    rows = dgcnn_rows() if sys.argv[2] == "dgcnn" else (bga_rows() if sys.argv[2] == "bga" else msg_rows())
    out = []
    for frag, grid, what, alg in rows:
        for k, v in detail.items():
            name, g = k.rsplit("|grid=", 1)
            if frag in name and (grid is None or int(g) == grid):
                cnt = v["bytes_per_launch"]
                out.append({"kernel": name, "grid_threads": int(g), "what": what, "counted_bytes_per_launch": cnt,
                            "algorithmic_bytes_per_launch": alg, "counted_over_algorithmic": (round(cnt / alg, 3) if alg else None),
                            "launches_counted": v["launches"]})
                break
    json.dump({"source": sys.argv[1], "method": "rocprofv3 --pmc FETCH_SIZE (x2, KiB) and WRITE_SIZE (KiB) in separate passes of "
               "`bench.py --model %s --steps 3 --warmup 1` (tools/collect_traffic.py)" % sys.argv[2], "kernels": out}, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
