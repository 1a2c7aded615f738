#!/bin/bash
set -u
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
O=gpurun_out/r5c3; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_head_gpu.py tests/test_knn_gpu.py tests/test_ops_gpu.py -x -q > $O/pytest_a.log 2>&1; echo "rc=$?" >> $O/pytest_a.log
timeout 1200 python -m pytest tests/test_models_gpu.py tests/test_models_parity_gpu.py tests/test_spidercnn_gpu.py -x -q > $O/pytest_b.log 2>&1; echo "rc=$?" >> $O/pytest_b.log
python bench.py --model dgcnn --no-cpu-baseline --steps 10 --warmup 3 > $O/bench_dgcnn.json 2> $O/bench_dgcnn.err
python bench.py --no-cpu-baseline --steps 20 --warmup 5 > $O/bench_ssg.json 2> $O/bench_ssg.err
for f in $O/pytest_a.log $O/pytest_b.log; do tail -n 4 $f; done
python - <<'PY'
import json
for f in ("dgcnn", "ssg"):
    try:
        d = json.loads(open("gpurun_out/r5c3/bench_%s.json" % f).read().strip().splitlines()[-1])
        print(f, d["value"], d["ms_per_step"])
        for k in d["kernels"][:28]:
            print("   %-26s %-40s n=%3d %8.1f us" % (k["kernel"][6:], str(k["shape"]), k["launches"], k["avg_us"]))
    except Exception as e:
        print(f, "ERR", e)
PY
