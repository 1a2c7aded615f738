#!/usr/bin/env python3
"""Runs ON THE GPU BOX (via gpurun): HBM traffic per launch of the kernels of the REAL bench step, from the PMC
counters collected the way MI355X_MICROARCH.md prescribes -- FETCH_SIZE and WRITE_SIZE in SEPARATE rocprofv3 --pmc
passes (they do not fit one pass), no tracing flags next to --pmc, FETCH_SIZE doubled (gfx950 tallies 128-byte
requests at 64 bytes for wide coalesced reads), both counters in KiB.

  python tools/collect_traffic.py            -> gpurun_out/pmc_traffic.json         {C-ABI key: bytes per launch}
                                                gpurun_out/pmc_traffic_detail.json  per (kernel, grid) averages

A dispatch is identified by (kernel name, grid size); KEYS maps the (name fragment, grid) pairs of the SSG bench's
biggest kernels to the C-ABI call + leading shape arguments bench.py prints in its kernel table."""
import collections
import csv
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "gpurun_out")
# (kernel-name fragment -- a SUBSTRING, trailing template arguments may be left open --, total grid threads or None) -> bench.py key.  Template arguments:
# gemm_ws_kernel<NT, AM, EM, KC, WAVES, EH, VAR>, wgrad_pc_kernel<TK, TN, AMODE, DMODE>
KEYS = [
    # round 2: SA2 runs on compacted rows (bench.py tags those shapes 'compacted'; the grid is the launch's upper bound)
    # round 6: the data gradients of SA2 on split operands, 64-column passes (PCOPS_DGRAD_BF3=0 keeps the fp32 names below)
    ("gemm_ws_kernel<2, 6, 1, 32, 8, 2, 4>", None, "pcops_mlp_gemm_dgrad(2097152, 256, 128, 'compacted')"),
    ("gemm_ws_kernel<2, 2, 1, 32, 8, 2, 4>", None, "pcops_mlp_gemm_dgrad(2097152, 128, 128, 'compacted')"),
    ("gemm_ws_kernel<4, 6, 1, 64, 8, 2, 2>", None, "pcops_mlp_gemm_dgrad(2097152, 256, 128, 'compacted')"),
    # round 4: SA2's two weight gradients run as ONE template instantiation of the split-operand kernel with equal grid
    # sizes (128 x 1 x 2 and 256 x 1 x 1 workgroups): the counters cannot tell them apart, the entry is their average
    ("wgrad_bf3_kernel<1, 6>", None, "pcops_mlp_wgrad(2097152, 128, 128|256, 'compacted') [two launches averaged]"),
    ("wgrad_pc_kernel<2, 4, 1, 6, false>", None, "pcops_mlp_wgrad(2097152, 128, 256, 'compacted')"),
    # round 4: the forward products run as the split-operand variants (KC = 32, EH = 4, VAR + 4); the fp32 names are
    # kept for PCOPS_GEMM_BF3=0 runs
    ("gemm_ws_kernel<4, 1, 0, 32, 8, 4, 5>", 1024 * 512, "pcops_mlp_gemm_fwd(2097152, 128, 256, 'compacted')"),   # + per-block pooling
    ("gemm_ws_kernel<4, 1, 0, 32, 8, 4, 4>", 512 * 512, "pcops_mlp_gemm_fwd(2097152, 128, 128, 'compacted')"),
    ("gemm_ws_kernel<4, 1, 0, 32, 8, 4, 5>", 512 * 512, "pcops_mlp_gemm_fwd_pool(4194304, 64, 128, 32)"),
    ("gemm_ws_kernel<2, 5, 0, 32, 8, 2, 4>", None, "pcops_mlp_gemm_fwd_xyz(4194304, 64, 64)"),
    ("gemm_ws_kernel<4, 1, 0, 64, 8, 2, 1>", 1024 * 512, "pcops_mlp_gemm_fwd(2097152, 128, 256, 'compacted')"),
    ("gemm_ws_kernel<4, 1, 0, 64, 8, 2, 0>", 512 * 512, "pcops_mlp_gemm_fwd(2097152, 128, 128, 'compacted')"),
    ("gemm_ws_kernel<4, 2, 1, 64, 8, 2, 0>", None, "pcops_mlp_gemm_dgrad(2097152, 128, 128, 'compacted')"),
    ("wgrad_pc_kernel<2, 2, 1, 7, false>", None, "pcops_mlp_wgrad(2097152, 128, 128, 'compacted')"),
    ("gemm_ws_kernel<4, 1, 4, 64, 8, 2, 2>", None, "pcops_mlp_gemm_dgrad_top(32768, 512)"),
    ("wgrad_pc_kernel<2, 4, 1, 8, false>", None, "pcops_mlp_gram(32768, 512)"),
    ("sa_scatter_csr_kernel<32, false, 64>", None, "pcops_sa_scatter_bwd(256, 512, 128, 64, 128, 'compacted')"),
    ("gemm_ws_kernel<4, 1, 0, 64, 8, 2, 1>", 512 * 512, "pcops_mlp_gemm_fwd_pool(4194304, 64, 128, 32)"),
    ("bwd_fused_kernel<2, 4, false, false", None, "pcops_mlp_bwd_fused(4194304, 64, 128)"),            # round 3: one pass for ...
    ("bwd_fused_kernel<1, 2, true, false", None, "pcops_mlp_bwd_fused_xyz(4194304, 64, 64)"),
    ("wgrad_pc_kernel<1, 2, 1, 4, false>", None, "pcops_mlp_wgrad(4194304, 64, 128)"),                   # ... these four (PCOPS_BWD_FUSED=0)
    ("gemm_ws_kernel<2, 4, 1, 64, 8, 1, 0>", None, "pcops_mlp_gemm_dgrad(4194304, 128, 64)"),
    ("gemm_ws_kernel<2, 2, 3, 64, 8, 1, 0>", None, "pcops_mlp_gemm_dgrad_xyz(4194304, 64, 64)"),
    ("gemm_ws_kernel<2, 5, 0, 64, 8, 1, 0>", None, "pcops_mlp_gemm_fwd_xyz(4194304, 64, 64)"),
    ("wgrad_pc_kernel<1, 1, 5, 2, false>", None, "pcops_mlp_wgrad_xyz(4194304, 64, 64)"),
    ("qbp_kernel<1, 32>", 512 * 1024, "pcops_query_ball_point(256, 2048, 512, 0.2, 32)"),
    ("fps_kernel<256, 8, true>", 256 * 256, "pcops_farthest_point_sample(256, 2048, 512)"),
]


MODEL = sys.argv[sys.argv.index("--model") + 1] if "--model" in sys.argv else None     # another bench workload (cfg3 / cfg5)
MODEL_ARGS = ["--model", MODEL] if MODEL else []
SUFFIX = "_" + MODEL if MODEL else ""


def one_pass(counter):
    d = os.path.join(OUT, "pmc_%s%s" % (counter, SUFFIX))
    subprocess.run(["rm", "-rf", d])
    cmd = ["rocprofv3", "--pmc", counter, "-d", d, "-o", "p", "--output-format", "csv", "--",
           sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-extras"] + MODEL_ARGS
    subprocess.run(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=False, cwd="/tmp",
                   env=dict(os.environ, TMPDIR="/tmp"))
    agg = collections.defaultdict(list)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                agg[(r["Kernel_Name"], int(r["Grid_Size"]))].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in agg.items()}, {k: len(v) for k, v in agg.items()}


def mfma_pass():
    """third pass: matrix-pipe busy cycles and active GPU cycles per dispatch -> MFMA utilisation and effective clock.
    GRBM_GUI_ACTIVE is summed over the 8 XCDs; SQ_VALU_MFMA_BUSY_CYCLES counts per-SIMD busy cycles (1024 SIMDs)."""
    d = os.path.join(OUT, "pmc_MFMA")
    subprocess.run(["rm", "-rf", d])
    cmd = ["rocprofv3", "--pmc", "SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE", "-d", d, "-o", "p", "--output-format", "csv",
           "--", sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-extras"]
    subprocess.run(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=False, cwd="/tmp",
                   env=dict(os.environ, TMPDIR="/tmp"))
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            agg[(r["Kernel_Name"], int(r["Grid_Size"]))][r["Counter_Name"]].append(float(r["Counter_Value"]))
    out = {}
    # largest launches first: a template instantiation shared by several shapes is keyed by its biggest user
    for (name, grid), c in sorted(agg.items(), key=lambda kv: -sum(kv[1].get("GRBM_GUI_ACTIVE", [0])) /
                                  max(1, len(kv[1].get("GRBM_GUI_ACTIVE", [0])))):
        busy = sum(c.get("SQ_VALU_MFMA_BUSY_CYCLES", [0])) / max(1, len(c.get("SQ_VALU_MFMA_BUSY_CYCLES", [0])))
        act = sum(c.get("GRBM_GUI_ACTIVE", [0])) / max(1, len(c.get("GRBM_GUI_ACTIVE", [0])))
        if busy <= 0 or act <= 0:
            continue
        for frag, g, key in KEYS:
            if frag in name and (g is None or g == grid) and key not in out:
                out[key] = {"mfma_busy_cycles_per_simd": busy / 1024.0, "gpu_active_cycles": act / 8.0,
                            "mfma_utilisation": busy / 1024.0 / (act / 8.0)}
                print("%-52s MFMA pipe busy %.1f %% of the active cycles" % (key, 100 * out[key]["mfma_utilisation"]),
                      flush=True)
    json.dump(out, open(os.path.join(OUT, "pmc_mfma.json"), "w"), indent=1)


def main():
    os.makedirs(OUT, exist_ok=True)
    if "--mfma" in sys.argv:
        return mfma_pass()
    if "--from-detail" in sys.argv:         # re-key an existing per-(kernel, grid) table with the current KEYS (no GPU)
        detail = json.load(open(sys.argv[sys.argv.index("--from-detail") + 1]))
        table = {}
        for k, v in sorted(detail.items(), key=lambda kv: -kv[1]["bytes_per_launch"]):
            name, grid = k.rsplit("|grid=", 1)
            for frag, g, key in KEYS:
                if frag in name and (g is None or g == int(grid)) and key not in table:
                    table[key] = v["bytes_per_launch"]
        json.dump(table, sys.stdout, indent=1)
        return
    fetch, nf = one_pass("FETCH_SIZE")
    write, _ = one_pass("WRITE_SIZE")
    table, detail = {}, {}
    for (name, grid), f in sorted(fetch.items(), key=lambda kv: -kv[1]):
        w = write.get((name, grid), 0.0)
        total = int((2.0 * f + w) * 1024)
        short = name.replace("(anonymous namespace)::", "").replace("pcops_mlp::", "").split("(")[0].replace("void ", "")
        detail["%s|grid=%d" % (short, grid)] = {"FETCH_SIZE_KiB_raw": f, "WRITE_SIZE_KiB": w, "fetch_correction": 2.0,
                                                "bytes_per_launch": total, "launches": nf[(name, grid)]}
        for frag, g, key in KEYS:
            if frag in name and (g is None or g == grid) and key not in table:
                table[key] = total
                print("%-52s %7.3f GB per launch (fetch raw %.3f GiB x2, write %.3f GiB)" % (
                    key, total / 1e9, f / 2 ** 20, w / 2 ** 20), flush=True)
    json.dump(table, open(os.path.join(OUT, "pmc_traffic%s.json" % SUFFIX), "w"), indent=1)
    json.dump(detail, open(os.path.join(OUT, "pmc_traffic_detail%s.json" % SUFFIX), "w"), indent=1)
    for d in ("pmc_FETCH_SIZE", "pmc_WRITE_SIZE"):          # the raw per-dispatch CSVs are large: keep the tables only
        subprocess.run(["rm", "-rf", os.path.join(OUT, d + SUFFIX)])


if __name__ == "__main__":
    main()
