#!/bin/bash
# Runs ON THE GPU BOX: the T-Net's first-layer weight gradient without a scatter (PCOPS_EDGE_DIRECT; PCOPS_EDGE_DIRECT_FUSED:
# E^T Gm inside the one-pass backward of the layer above) -- tests, step A/B
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_fused_mlp_gpu.py -x -q -k "without_a_scatter or one_gemm" 2>&1 | tail -12
{
for m in dgcnn dgcnn_bga; do
for v in 1 0 1 0; do
  echo "== $m PCOPS_EDGE_DIRECT_FUSED=$v"; PCOPS_EDGE_DIRECT_FUSED=$v python bench.py --model $m --no-cpu-baseline --no-extras --steps 10 --warmup 3 | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
done
done
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/direct_fused_ab.txt
timeout 1500 python -m pytest tests/test_models_parity_gpu.py tests/test_models_gpu.py -x -q -k "dgcnn" 2>&1 | tail -6
