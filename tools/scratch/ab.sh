#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/ab
timeout 900 python -m pytest tests/test_fused_mlp_gpu.py tests/test_bench_size_gpu.py -x -q -m gpu > gpurun_out/ab/tests.log 2>&1
tail -3 gpurun_out/ab/tests.log
for rep in 1 2 3; do
for v in base ""; do
  lib=scanobjectnn_amd/libpcops${v:+_$v}.so
  PCOPS_LIB=$PWD/$lib timeout 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-extras > gpurun_out/ab/bench_${v:-new}_$rep.json 2> gpurun_out/ab/bench_${v:-new}_$rep.err
  python - <<PY
import json
l=[x for x in open("gpurun_out/ab/bench_${v:-new}_$rep.json") if x.startswith("{")]
d=json.loads(l[-1])
print("${v:-new}", $rep, round(d["value"],1), round(d["ms_per_step"],4), d["roofline"].get("kernel",""), round(d["roofline"]["frac"],4))
PY
done
done
for v in base ""; do
  lib=scanobjectnn_amd/libpcops${v:+_$v}.so
  echo "== ${v:-new}"; PCOPS_LIB=$PWD/$lib python tools/bench_gemm.py dgrad 10 2>&1 | grep dgrad
done
