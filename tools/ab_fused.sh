#!/bin/bash
# Runs ON THE GPU BOX: the SSG step with the one-pass backward of the narrow layers (default) and with the two-kernel
# path it replaced (PCOPS_BWD_FUSED=0), interleaved on one box -> gpurun_out/fused_ab.txt
cd "$(dirname "$0")/.."
out=gpurun_out/fused_ab.txt; : > $out
for rep in 1 2 3; do
for v in 0 1; do
  PCOPS_BWD_FUSED=$v python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-extras 2>/dev/null > /tmp/ab_line.json
  python - "$v" "$rep" >> $out <<'PY'
import json, sys
v, rep = sys.argv[1], int(sys.argv[2])
d = json.loads([l for l in open("/tmp/ab_line.json") if l.startswith("{")][-1])
print("PCOPS_BWD_FUSED=%s run %d: %8.1f clouds/s  %7.3f ms/step   dominant %s %s  %.1f us  frac %.3f" % (
    v, rep, d["value"], d["ms_per_step"], d["roofline"]["kernel"], d["roofline"]["shape"], d["roofline"]["avg_launch_us"], d["roofline"]["frac"]))
if rep == 1:
    for k in d["kernels"][:14]:
        print("      %-28s %-40s %8.1f us  %s" % (k["kernel"], k["shape"], k["avg_us"], k.get("bound_frac") and round(k["bound_frac"], 3)))
PY
done
done
cat $out
