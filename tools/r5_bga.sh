#!/bin/bash
# Runs ON THE GPU BOX: dgcnn_bga's segmentation head as per-cloud + per-point products (PCOPS_CLOUD_POINT) and its first layer's
# Y = Q + Ctr[cloud] / backward as streaming passes (PCOPS_CLOUD_BIAS) -- tests, step A/B
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_models_parity_gpu.py tests/test_models_gpu.py tests/test_deterministic_gpu.py tests/test_checkpoint_eval_gpu.py -x -q -k "dgcnn" 2>&1 | tail -8
{
for v in "1 1" "1 0" "0 0" "1 1" "1 0" "0 0"; do
  set -- $v
  echo "== dgcnn_bga PCOPS_CLOUD_POINT=$1 PCOPS_CLOUD_BIAS=$2"; PCOPS_CLOUD_POINT=$1 PCOPS_CLOUD_BIAS=$2 python bench.py --model dgcnn_bga --no-cpu-baseline --no-extras --steps 10 --warmup 3 | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
done
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/cloud_point_ab.txt
