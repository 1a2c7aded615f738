#!/bin/bash
# launches per step and the generic (torch / rocclr) share of a bench step; args: model [tag]
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
M=${1:-dgcnn}; T=${2:-x}
O=$PWD/gpurun_out/launch_${M}_$T; rm -rf $O; mkdir -p $O
( cd /tmp && rocprofv3 --kernel-trace --stats -d $O/kt -o p --output-format csv -- python $OLDPWD/bench.py --model $M --steps 10 --warmup 3 --no-cpu-baseline --no-extras > $O/bench.json 2>/dev/null )
python - "$O" <<'PY'
import csv, glob, sys, collections, json
O = sys.argv[1]
steps = 13 + 10        # warmup + timed + the bracketed kernel pass of bench.py (10 steps)
rows = []
for f in glob.glob(O + "/kt/**/*kernel_stats.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
tot_calls = sum(int(r["Calls"]) for r in rows)
adam = [int(r["Calls"]) for r in rows if "adam_step_kernel" in r["Name"]]     # exactly one launch per training step
gen = [r for r in rows if ("at::native" in r["Name"] or "rocclr" in r["Name"] or r["Name"].startswith("Cijk"))]
with open(O + "/summary.txt", "w") as o:
    try:
        d = json.loads(open(O + "/bench.json").read().strip().splitlines()[-1])
        o.write("%s: %.0f clouds/s, %.3f ms/step\n" % (d["config"]["workload"][:40], d["value"], d["ms_per_step"]))
        steps = d["steps"] + d["warmup"] + d.get("kernels_steps", 0)
    except Exception as e:
        o.write("bench line: %s\n" % e)
    if adam:
        steps = adam[0]          # (the bench runs a few steps more than steps + warmup + its bracketed pass: builds, probes)
    o.write("steps profiled %d: launches/step %.1f, generic launches/step %.1f, generic ms/step %.3f, all kernels ms/step %.3f\n" % (
        steps, tot_calls / steps, sum(int(r["Calls"]) for r in gen) / steps, sum(float(r["TotalDurationNs"]) for r in gen) / steps / 1e6,
        sum(float(r["TotalDurationNs"]) for r in rows) / steps / 1e6))
    for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:70]:
        o.write("%7.1f calls/step %9.1f us/step  avg %8.1f us  %s\n" % (int(r["Calls"]) / steps, float(r["TotalDurationNs"]) / steps / 1e3,
                                                                 float(r["AverageNs"]) / 1e3, r["Name"].replace("(anonymous namespace)::", "").replace("pcops_mlp::", "")[:110]))
PY
cp $O/kt/*/*kernel_stats.csv $O/kernel_stats.csv 2>/dev/null || find $O/kt -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
rm -rf $O/kt
head -75 $O/summary.txt
