#!/bin/bash
set -u
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
O=gpurun_out/r5c5; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_fused_mlp_gpu.py tests/test_bn_shifted_moments_gpu.py -x -q > $O/pytest_a.log 2>&1; echo "rc=$?" >> $O/pytest_a.log
timeout 1200 python -m pytest tests/test_models_gpu.py tests/test_models_parity_gpu.py -x -q -k dgcnn > $O/pytest_b.log 2>&1; echo "rc=$?" >> $O/pytest_b.log
for f in $O/pytest_a.log $O/pytest_b.log; do tail -n 4 $f; done
python tools/bench_edgeconv.py 10 > $O/plain.txt 2>&1; cat $O/plain.txt
bash tools/r5_ec_pmc.sh lds2 64 > $O/pmc.log 2>&1
grep -A18 "ec_fwd_lds_kernel\|ec_walk_lds_kernel" gpurun_out/ecpmc_lds2/counters.txt | head -60
