#!/bin/bash
# Runs ON THE GPU BOX: A/B of libpcops builds on one box, interleaved.
#   tools/ab_libs.sh OUT "lib1 lib2 ..." "model1 model2 ..." [reps]
# libN = "default" or the NAME of scanobjectnn_amd/libpcops_NAME.so (tools/build_variant.sh); per (lib, model, rep) one
# line with clouds/s, ms/step and the dominant kernel; rep 1 also lists the kernel table.
cd "$(dirname "$0")/.."
out=gpurun_out/$1; : > $out
libs=$2; models=${3:-pointnet2_cls_ssg}; reps=${4:-2}
for rep in $(seq 1 $reps); do
for m in $models; do
for l in $libs; do
  if [ "$l" = default ]; then unset PCOPS_LIB; else export PCOPS_LIB=$PWD/scanobjectnn_amd/libpcops_$l.so; fi
  python bench.py --model $m --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null > /tmp/ab_line.json
  python - "$l" "$rep" "$m" >> $out <<'PY'
import json, sys
l, rep, m = sys.argv[1], int(sys.argv[2]), sys.argv[3]
try:
    d = json.loads([x for x in open("/tmp/ab_line.json") if x.startswith("{")][-1])
except Exception as e:
    print("%-8s %-20s run %d: FAILED %r" % (l, m, rep, e)); sys.exit(0)
r = d["roofline"]
print("%-8s %-20s run %d: %8.1f clouds/s  %7.3f ms/step   dominant %s %s  %.1f us  frac %.3f" % (
    l, m, rep, d["value"], d["ms_per_step"], r["kernel"], r["shape"], r["avg_launch_us"], r["frac"]))
if rep == 1:
    for k in d["kernels"][:16]:
        print("      %-30s %-44s x%-3d %8.1f us  %s" % (k["kernel"], k["shape"], k["launches"], k["avg_us"], round(k["bound_frac"], 3)))
PY
done
done
done
cat $out
