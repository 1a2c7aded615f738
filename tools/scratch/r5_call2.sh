#!/bin/bash
# round 5, GPU call 2: parity of the new edgeconv kernels + DGCNN bench A/B
set -u
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
O=gpurun_out/r5c2; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_fused_mlp_gpu.py tests/test_knn_gpu.py tests/test_bn_shifted_moments_gpu.py -x -q > $O/pytest_a.log 2>&1; echo "rc=$?" >> $O/pytest_a.log
timeout 900 python -m pytest tests/test_models_gpu.py tests/test_models_parity_gpu.py -x -q -k "dgcnn" > $O/pytest_b.log 2>&1; echo "rc=$?" >> $O/pytest_b.log
timeout 900 python -m pytest tests/test_bench_size_gpu.py -x -q -k "tnet or knn or dgcnn" > $O/pytest_c.log 2>&1; echo "rc=$?" >> $O/pytest_c.log
python bench.py --model dgcnn --no-cpu-baseline --steps 10 --warmup 3 > $O/bench_dgcnn_new.json 2> $O/bench_dgcnn_new.err
PCOPS_EDGECONV_R5=0 python bench.py --model dgcnn --no-cpu-baseline --steps 10 --warmup 3 > $O/bench_dgcnn_old.json 2> $O/bench_dgcnn_old.err
tail -2 $O/pytest_a.log $O/pytest_b.log $O/pytest_c.log
