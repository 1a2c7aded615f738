#!/bin/bash
# Runs ON THE GPU BOX: the pooled epilogue for groups that are not whole tiles (PCOPS_POOL_S4, mlp.hip) -- tests, then A/B
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_fused_mlp_gpu.py -x -q -k "forward_train or backward" 2>&1 | tail -4
{
for v in 1 0 1; do
  echo "== PCOPS_POOL_S4=$v dgcnn"; PCOPS_POOL_S4=$v python bench.py --model dgcnn --no-cpu-baseline --no-extras --steps 10 --warmup 3 | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step']); [print('   ', k['kernel'], k['shape'], round(k['avg_us'],1)) for k in d.get('kernels',[]) if k['shape'][:1]==[10485760] or 'maxpool' in k['kernel']]"
done
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/s4_ab.txt
