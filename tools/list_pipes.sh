#!/bin/bash
# Runs ON THE GPU BOX: per model, the kernels of the training step by matrix pipe (the library's own label), largest first.
cd "$(dirname "$0")/.."
for m in ${@:-pointnet2_cls_ssg pointnet2_cls_msg dgcnn pointnet2_cls_bga dgcnn_bga}; do
  python bench.py --model $m --no-cpu-baseline --no-extras --steps 10 --warmup 3 2>/dev/null | M=$m python -c "
import sys, json, os
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('== %s  %.1f clouds/s  %.3f ms/step' % (os.environ['M'], d['value'], d['ms_per_step']))
for k in d['kernels']:
    if k.get('work_unit') == 'flop' or 'mfma' in str(k.get('pipe')):
        print('   %-28s %-44s x%-3d %8.1f us  %-30s %s' % (k['kernel'][6:], str(k['shape'])[:44], k['launches'] // 10, k['avg_us'], k.get('pipe'), round(k.get('bound_frac') or 0, 3)))
"
done
