#!/bin/bash
# Runs ON THE GPU BOX: non-temporal stores for output streams of 256 MB and more (PCOPS_NT_STORE) -- step A/B of every model
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
{
for m in pointnet2_cls_ssg dgcnn pointnet2_cls_msg; do
for v in 1 0 1 0; do
  echo "== $m PCOPS_NT_STORE=$v"; PCOPS_NT_STORE=$v python bench.py --model $m --no-cpu-baseline --no-extras --steps 20 --warmup 5 | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
done
done
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/nt_ab.txt
timeout 900 python -m pytest tests/test_fused_mlp_gpu.py -x -q 2>&1 | tail -3
