#!/bin/bash
# usage (on the GPU box): tools/pmc_gemm.sh <fwd|dgrad|wgrad> M K N   -> per-kernel PMC summary (SQ + GRBM, one pass)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_$1_$3_$4
rm -rf $OUT
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_BUSY_CYCLES \
  -d $OUT -o p --output-format csv -- python $R/tools/bench_gemm.py $1 3 --shape $2 $3 $4 > /dev/null 2>&1
rocprofv3 --kernel-trace -d $OUT/t -o t --output-format csv -- python $R/tools/bench_gemm.py $1 3 --shape $2 $3 $4 > /dev/null 2>&1
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
dur = collections.defaultdict(list)
for f in glob.glob("$OUT/t/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        dur[r["Kernel_Name"][:70]].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
for k, c in agg.items():
    if "gemm" not in k and "wgrad" not in k: continue
    m = {n: sum(v) / len(v) for n, v in c.items()}
    d = sum(dur[k]) / max(1, len(dur[k]))
    print(k)
    print("  dur(us, unprofiled trace) %.1f  GUI_ACTIVE %.3g -> clock %.2f GHz" % (d / 1e3, m.get("GRBM_GUI_ACTIVE", 0), m.get("GRBM_GUI_ACTIVE", 0) / max(d, 1)))
    for n in sorted(m): print("  %-28s %.4g" % (n, m[n]))
    if m.get("GRBM_GUI_ACTIVE"):
        print("  MFMA busy / (GUI_ACTIVE * 1024 SIMD) = %.3f" % (m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (m["GRBM_GUI_ACTIVE"] * 1024)))
PY
