#!/bin/bash
set -u
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
O=gpurun_out/r5c6; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_fused_mlp_gpu.py -x -q -k "gather or edge" > $O/pytest_a.log 2>&1; echo "rc=$?" >> $O/pytest_a.log
timeout 1200 python -m pytest tests/test_models_parity_gpu.py -x -q -k dgcnn > $O/pytest_b.log 2>&1; echo "rc=$?" >> $O/pytest_b.log
for f in $O/pytest_a.log $O/pytest_b.log; do tail -n 3 $f; done
python tools/bench_edgeconv.py 10 > $O/plain.txt 2>&1; cat $O/plain.txt
bash tools/r5_ec_pmc.sh lds3 64 > $O/pmc.log 2>&1
grep -A18 "ec_walk_lds_kernel" gpurun_out/ecpmc_lds3/counters.txt | head -20
python bench.py --model dgcnn --no-cpu-baseline --steps 10 --warmup 3 > $O/bench_dgcnn.json 2> $O/bench_dgcnn.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r5c6/bench_dgcnn.json").read().strip().splitlines()[-1])
print("dgcnn", d["value"], d["ms_per_step"])
PY
