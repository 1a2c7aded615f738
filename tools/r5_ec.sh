#!/bin/bash
# per-kernel split (rocprofv3 kernel trace) + PMC traffic of the edgeconv micro-benchmark; args: tag
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
T=${1:-x}
O=$PWD/gpurun_out/ec_$T; rm -rf $O; mkdir -p $O
python tools/bench_edgeconv.py 10 > $O/plain.txt 2>&1
( cd /tmp && rocprofv3 --kernel-trace --stats -d $O/kt -o p --output-format csv -- python $OLDPWD/tools/bench_edgeconv.py 5 > /dev/null 2>&1 )
python - "$O" <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/kt/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        agg[(r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("pcops_mlp::", "").replace("void ", "").split("(")[0][:60], int(r.get("Grid_Size") or 0))].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
with open(sys.argv[1] + "/kernels.txt", "w") as o:
    for (n, g), v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        o.write("%-62s grid %9d n=%3d avg %9.1f us\n" % (n, g, len(v), sum(v) / len(v) / 1e3))
PY
for c in FETCH_SIZE WRITE_SIZE; do
( cd /tmp && rocprofv3 --pmc $c -d $O/pmc_$c -o p --output-format csv -- python $OLDPWD/tools/bench_edgeconv.py 2 > /dev/null 2>&1 )
done
python - "$O" <<'PY'
import csv, glob, sys, collections
tab = collections.defaultdict(lambda: collections.defaultdict(list))
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(sys.argv[1] + "/pmc_%s/**/*counter_collection.csv" % c, recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == c:
                tab[(r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("pcops_mlp::", "").replace("void ", "").split("(")[0][:60], int(r["Grid_Size"]))][c].append(float(r["Counter_Value"]))
with open(sys.argv[1] + "/traffic.txt", "w") as o:
    for k, v in sorted(tab.items(), key=lambda kv: -sum(kv[1].get("FETCH_SIZE", [0]))):
        f = sum(v.get("FETCH_SIZE", [0])) / max(1, len(v.get("FETCH_SIZE", [0])))
        w = sum(v.get("WRITE_SIZE", [0])) / max(1, len(v.get("WRITE_SIZE", [0])))
        o.write("%-62s grid %9d fetch(x2) %8.1f MB write %8.1f MB\n" % (k[0], k[1], 2 * f * 1024 / 1e6, w * 1024 / 1e6))
PY
rm -rf $O/kt $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE
cat $O/plain.txt; head -30 $O/kernels.txt; head -24 $O/traffic.txt
