#!/bin/bash
# Runs ON THE GPU BOX: the one-walk EdgeConv backward (PCOPS_EDGECONV_BWD_FUSED) -- test, micro-benchmark A/B, DGCNN step A/B
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -k "edge_pool_bwd" 2>&1 | tail -5
{
echo "== fused"; python tools/bench_edgeconv.py 10 | grep "bwd\|fwd C"
echo "== two kernels"; PCOPS_EDGECONV_BWD_FUSED=0 python tools/bench_edgeconv.py 10 | grep "bwd\|fwd C"
if [ "${1:-}" = "step" ]; then
for v in 1 0 1; do
  echo "== PCOPS_EDGECONV_BWD_FUSED=$v dgcnn"; PCOPS_EDGECONV_BWD_FUSED=$v python bench.py --model dgcnn --no-cpu-baseline --no-extras --steps 10 --warmup 3 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
done
fi
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/ecb_ab.txt
if [ "${1:-}" = "step" ]; then
timeout 1200 python -m pytest tests/test_fused_mlp_gpu.py tests/test_models_parity_gpu.py -x -q -k "edge_conv or dgcnn" 2>&1 | tail -5
fi
