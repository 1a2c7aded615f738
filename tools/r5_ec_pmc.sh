#!/bin/bash
# SQ counters of the edgeconv micro-benchmark kernels (two passes of 8 SQ counters); args: tag [bench args]
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
T=${1:-x}; shift
O=$PWD/gpurun_out/ecpmc_$T; rm -rf $O; mkdir -p $O
P1="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"
P2="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"
( cd /tmp && rocprofv3 --pmc $P1 -d $O/p1 -o p --output-format csv -- python $OLDPWD/tools/bench_edgeconv.py 2 "$@" > $O/p1.log 2>&1 )
( cd /tmp && rocprofv3 --pmc $P2 -d $O/p2 -o p --output-format csv -- python $OLDPWD/tools/bench_edgeconv.py 2 "$@" > $O/p2.log 2>&1 )
python - "$O" <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = (r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("pcops_mlp::", "").replace("void ", "").split("(")[0][:40], int(r["Grid_Size"]))
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(sys.argv[1] + "/counters.txt", "w") as o:
    for k, c in sorted(agg.items(), key=lambda kv: -sum(kv[1].get("GRBM_GUI_ACTIVE", [0]))):
        if not (k[0].startswith("ec_") or k[0].startswith("edge_")):
            continue
        m = {n: sum(v) / len(v) for n, v in c.items()}
        o.write("%s grid=%d\n" % k)
        for n in sorted(m):
            o.write("    %-24s %14.0f   /1024 = %10.0f\n" % (n, m[n], m[n] / 1024))
PY
rm -rf $O/p1 $O/p2
cat $O/counters.txt | head -150; tail -3 $O/p2.log
