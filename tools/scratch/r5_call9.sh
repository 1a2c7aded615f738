#!/bin/bash
set -u
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
O=gpurun_out/r5c9; rm -rf $O; mkdir -p $O
timeout 1200 python -m pytest tests/test_models_gpu.py tests/test_models_parity_gpu.py tests/test_deterministic_gpu.py tests/test_checkpoint_eval_gpu.py -x -q -k "dgcnn" > $O/pytest_b.log 2>&1; echo "rc=$?" >> $O/pytest_b.log
tail -n 6 $O/pytest_b.log
timeout 600 python -m pytest tests/test_bench_gpu.py -x -q > $O/pytest_c.log 2>&1; echo "rc=$?" >> $O/pytest_c.log; tail -n 4 $O/pytest_c.log
python bench.py --model dgcnn --no-cpu-baseline --no-extras --steps 10 --warmup 3 > $O/bench_dgcnn.json 2> $O/bench_dgcnn.err
python bench.py --model dgcnn_bga --no-cpu-baseline --no-extras --steps 10 --warmup 3 > $O/bench_dgcnn_bga.json 2> $O/bench_dgcnn_bga.err
python - <<'PY'
import json
for f in ("bench_dgcnn", "bench_dgcnn_bga"):
    try:
        d = json.loads(open("gpurun_out/r5c9/%s.json" % f).read().strip().splitlines()[-1])
        print(f, d["value"], d["ms_per_step"])
    except Exception as e:
        print(f, "ERR", e, open("gpurun_out/r5c9/%s.err" % f).read()[-1500:])
PY
