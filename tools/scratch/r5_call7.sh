#!/bin/bash
set -u
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
O=gpurun_out/r5c7; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_fused_mlp_gpu.py -x -q -k "one_gemm or gather or edge" > $O/pytest_a.log 2>&1; echo "rc=$?" >> $O/pytest_a.log
timeout 1200 python -m pytest tests/test_models_gpu.py tests/test_models_parity_gpu.py tests/test_bench_size_gpu.py -x -q -k "dgcnn or tnet" > $O/pytest_b.log 2>&1; echo "rc=$?" >> $O/pytest_b.log
for f in $O/pytest_a.log $O/pytest_b.log; do tail -n 12 $f; done
python bench.py --model dgcnn --no-cpu-baseline --steps 10 --warmup 3 > $O/bench_dgcnn.json 2> $O/bench_dgcnn.err
PCOPS_EDGE_QC=0 python bench.py --model dgcnn --no-cpu-baseline --steps 10 --warmup 3 > $O/bench_dgcnn_noqc.json 2> $O/bench_dgcnn_noqc.err
python - <<'PY'
import json
for f in ("bench_dgcnn", "bench_dgcnn_noqc"):
    try:
        d = json.loads(open("gpurun_out/r5c7/%s.json" % f).read().strip().splitlines()[-1])
        print(f, d["value"], d["ms_per_step"])
    except Exception as e:
        print(f, "ERR", e, open("gpurun_out/r5c7/%s.err" % f).read()[-2000:])
PY
