#!/bin/bash
# Runs ON THE GPU BOX: alternating A/B of one environment switch on the training step of a model.
#   tools/ab_env.sh VAR A B [model] [reps] [kernel-substring]
# prints clouds/s and ms/step of every run (A B A B ...) and, with a kernel substring, that kernel's rows of the table.
cd "$(dirname "$0")/.."
VAR=$1; A=$2; B=$3; MODEL=${4:-pointnet2_cls_ssg}; REPS=${5:-2}; KSUB=${6:-}
mkdir -p gpurun_out
for i in $(seq 1 $REPS); do
for v in $A $B; do
  env $VAR=$v python bench.py --model $MODEL --no-cpu-baseline --no-extras --steps 20 --warmup 5 2>/dev/null | KSUB="$KSUB" TAG="$VAR=$v" python -c "
import sys, json, os
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('%-28s %9.1f clouds/s %7.3f ms/step' % (os.environ['TAG'], d['value'], d['ms_per_step']))
ks = os.environ.get('KSUB')
if ks:
    for k in d['kernels']:
        if ks in k['kernel']:
            print('      %-26s %-40s %8.1f us' % (k['kernel'].replace('pcops_', ''), k['shape'], k['avg_us']))
"
done
done
