#!/bin/bash
# Runs ON THE GPU BOX: FETCH / WRITE counters of ONE dense data-gradient launch shape through tools/bench_gemm.py
#   bash tools/pmc_gemm_shape.sh M K N      (layer K -> N: dY (M, N) in, Gprev (M, K) out)
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE "SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_sum"; do
  D=gpurun_out/pmc_shape; rm -rf $D
  ( cd /tmp && rocprofv3 --pmc $c -d $OLDPWD/$D -o p --output-format csv -- python $OLDPWD/tools/bench_gemm.py dgrad 2 --shape $1 $2 $3 > /dev/null 2>&1 )
  python - <<'PY'
import collections, csv, glob
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pmc_shape/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "gemm_ws_kernel" in r["Kernel_Name"]:
            agg[r["Kernel_Name"].split("(")[0].replace("void ", "").replace("pcops_mlp::", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in agg.items():
    for n, v in c.items():
        print("%-48s %-24s %14.0f (n=%d)" % (k, n, sum(v) / len(v), len(v)))
PY
done
