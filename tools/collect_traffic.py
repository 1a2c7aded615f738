#!/usr/bin/env python3
"""Runs ON THE GPU BOX (via gpurun): HBM traffic of the big MLP kernels from the PMC counters, collected the way
MI355X_MICROARCH.md prescribes -- FETCH_SIZE and WRITE_SIZE in SEPARATE rocprofv3 --pmc passes (they do not fit
one pass), no tracing flags next to --pmc, FETCH_SIZE doubled (gfx950 tallies 128-byte requests at 64 bytes for
wide coalesced reads), both counters in KiB.  Writes gpurun_out/pmc_traffic.json:
    {"pcops_mlp_gemm_dgrad(2097152, 256, 128)": bytes_per_launch, ...}
keys are the C-ABI name + the leading shape arguments, as bench.py's kernel table prints them."""
import csv, glob, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "gpurun_out")
CASES = [  # (kind, M, K, N) in tools/bench_gemm.py's convention (dgrad of the K->N layer: dY is M x N)
    ("fwd", 2097152, 128, 256), ("dgrad", 2097152, 128, 256), ("wgrad", 2097152, 128, 256),
    ("fwd", 4194304, 64, 128), ("dgrad", 4194304, 64, 128), ("wgrad", 4194304, 64, 128),
    ("fwd", 2097152, 128, 128), ("dgrad", 2097152, 128, 128), ("wgrad", 2097152, 128, 128),
    ("fwd", 4194304, 64, 64), ("dgrad", 4194304, 64, 64), ("wgrad", 4194304, 64, 64),
]
KERNEL_OF = {"fwd": ("gemm_ws_kernel", "gemm_rt_kernel"), "dgrad": ("gemm_ws_kernel", "gemm_rt_kernel"),
             "wgrad": ("wgrad_ws_kernel", "wgrad_kernel")}
ABI = {"fwd": "pcops_mlp_gemm_fwd", "dgrad": "pcops_mlp_gemm_dgrad", "wgrad": "pcops_mlp_wgrad"}


def one_pass(counter, kind, M, K, N):
    d = os.path.join(OUT, "pmc_%s" % counter)
    subprocess.run(["rm", "-rf", d])
    cmd = ["rocprofv3", "--pmc", counter, "-d", d, "-o", "p", "--output-format", "csv", "--",
           sys.executable, os.path.join(ROOT, "tools", "bench_gemm.py"), kind, "3", "--shape", str(M), str(K), str(N)]
    subprocess.run(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=False, cwd="/tmp")
    vals = []
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter and any(k in r["Kernel_Name"] for k in KERNEL_OF[kind]):
                vals.append(float(r["Counter_Value"]))
    return sum(vals) / len(vals) if vals else None


def main():
    os.makedirs(OUT, exist_ok=True)
    table, detail = {}, {}
    for kind, M, K, N in CASES:
        f, w = one_pass("FETCH_SIZE", kind, M, K, N), one_pass("WRITE_SIZE", kind, M, K, N)
        if f is None or w is None:
            continue
        shape = (M, N, K) if kind == "dgrad" else (M, K, N)     # C-ABI order of pcops_mlp_gemm_dgrad: (M, K=N_l, Nout)
        key = "%s%s" % (ABI[kind], shape)
        table[key] = int((2.0 * f + w) * 1024)
        detail[key] = {"FETCH_SIZE_KiB_raw": f, "WRITE_SIZE_KiB": w, "fetch_correction": 2.0}
        print(key, "traffic %.3f GB (fetch raw %.3f GiB x2, write %.3f GiB)" % (table[key] / 1e9, f / 2**20, w / 2**20), flush=True)
    json.dump(table, open(os.path.join(OUT, "pmc_traffic.json"), "w"), indent=1)
    json.dump(detail, open(os.path.join(OUT, "pmc_traffic_detail.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
