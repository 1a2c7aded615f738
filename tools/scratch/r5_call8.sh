#!/bin/bash
set -u
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
O=gpurun_out/r5c8; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_fused_mlp_gpu.py tests/test_bwd_fused_gpu.py -x -q > $O/pytest_a.log 2>&1; echo "rc=$?" >> $O/pytest_a.log
timeout 1200 python -m pytest tests/test_bench_size_gpu.py tests/test_models_parity_gpu.py -x -q -k "sa_stack or ssg or msg or bga" > $O/pytest_b.log 2>&1; echo "rc=$?" >> $O/pytest_b.log
for f in $O/pytest_a.log $O/pytest_b.log; do tail -n 5 $f; done
for v in 1 0; do
PCOPS_SCATTER_REBUILD_Y=$v python bench.py --no-cpu-baseline --no-extras --steps 20 --warmup 5 > $O/bench_ssg_$v.json 2> $O/bench_ssg_$v.err
PCOPS_SCATTER_REBUILD_Y=$v python bench.py --model pointnet2_cls_msg --no-cpu-baseline --no-extras --steps 10 --warmup 3 > $O/bench_msg_$v.json 2> $O/bench_msg_$v.err
done
python - <<'PY'
import json
for f in ("ssg_1", "ssg_0", "msg_1", "msg_0"):
    try:
        d = json.loads(open("gpurun_out/r5c8/bench_%s.json" % f).read().strip().splitlines()[-1])
        print(f, d["value"], d["ms_per_step"])
        for k in d["kernels"]:
            if "scatter" in k["kernel"]:
                print("    ", k["kernel"], k["shape"], k["avg_us"])
    except Exception as e:
        print(f, "ERR", e)
PY
