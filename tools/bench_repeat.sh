#!/bin/bash
# usage (GPU box): tools/bench_repeat.sh N  -> N bench runs, step time + the kernels whose time moved most between runs
N=${1:-3}
for i in $(seq 1 $N); do
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/rep_$i.json 2>/dev/null
done
python - <<PY
import json, glob
runs = []
for f in sorted(glob.glob("gpurun_out/rep_*.json")):
    d = json.loads([l for l in open(f) if l.startswith("{")][-1])
    runs.append(d)
    print(f, "%.1f clouds/s %.2f ms/step, listed kernels %.2f ms" % (d["value"], d["ms_per_step"],
          sum(k["avg_us"] * k["launches"] for k in d["kernels"]) / d["steps"] / 1e3))
keys = {}
for d in runs:
    for k in d["kernels"]:
        keys.setdefault((k["kernel"], tuple(k["shape"])), []).append(k["avg_us"])
rows = sorted(keys.items(), key=lambda kv: -(max(kv[1]) - min(kv[1])))[:8]
for (n, s), v in rows:
    print("%-28s %-32s %s" % (n.replace("pcops_", ""), s, " ".join("%8.1f" % x for x in v)))
PY
