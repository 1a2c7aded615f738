#!/bin/bash
# Runs ON THE GPU BOX: the pooled epilogue for groups that are not whole tiles (PCOPS_POOL_S4, mlp.hip) -- tests, then A/B
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_fused_mlp_gpu.py -x -q 2>&1 | tail -8
{
for v in 1 0 1; do
  echo "== PCOPS_POOL_S4=$v dgcnn"; PCOPS_POOL_S4=$v python bench.py --model dgcnn --no-cpu-baseline --no-extras --steps 10 --warmup 3 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
done
echo "== ssg"; python bench.py --no-cpu-baseline --no-extras --steps 20 --warmup 5 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
echo "== msg S4=1"; python bench.py --model pointnet2_cls_msg --no-cpu-baseline --no-extras --steps 10 --warmup 3 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
echo "== msg S4=0"; PCOPS_POOL_S4=0 python bench.py --model pointnet2_cls_msg --no-cpu-baseline --no-extras --steps 10 --warmup 3 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/s4_ab.txt
