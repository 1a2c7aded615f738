#!/bin/bash
set -u
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
O=gpurun_out/r5c4; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_fused_mlp_gpu.py tests/test_bn_shifted_moments_gpu.py tests/test_head_gpu.py -x -q > $O/pytest_a.log 2>&1; echo "rc=$?" >> $O/pytest_a.log
timeout 1200 python -m pytest tests/test_models_gpu.py tests/test_models_parity_gpu.py -x -q > $O/pytest_b.log 2>&1; echo "rc=$?" >> $O/pytest_b.log
for f in $O/pytest_a.log $O/pytest_b.log; do tail -n 4 $f; done
bash tools/r5_ec.sh lds > $O/ec.log 2>&1
head -40 $O/ec.log
