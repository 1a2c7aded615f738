#!/bin/bash
set -u
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
O=gpurun_out/r5c10; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_fused_mlp_gpu.py -x -q -k "one_gemm or gather or edge" > $O/pytest_a.log 2>&1; echo "rc=$?" >> $O/pytest_a.log
timeout 1200 python -m pytest tests/test_models_parity_gpu.py tests/test_models_gpu.py -x -q -k dgcnn > $O/pytest_b.log 2>&1; echo "rc=$?" >> $O/pytest_b.log
for f in $O/pytest_a.log $O/pytest_b.log; do tail -n 3 $f; done
python tools/bench_edgeconv.py 10 > $O/plain.txt 2>&1; cat $O/plain.txt
bash tools/r5_ec_pmc.sh sp2 64 > /dev/null 2>&1; grep -A2 "ec_sparse_kernel\|ec_argi" gpurun_out/ecpmc_sp2/counters.txt | head -12
for i in 1 2; do python bench.py --model dgcnn --no-cpu-baseline --no-extras --steps 10 --warmup 3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"; done
