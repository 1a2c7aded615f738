python -m pytest tests/test_fused_mlp_gpu.py tests/test_bench_size_gpu.py tests/test_deterministic_gpu.py tests/test_models_parity_gpu.py -q 2>&1 | tail -8
python bench.py --model dgcnn --steps 10 --warmup 3 --no-cpu-baseline --no-extras > gpurun_out/r3_dg.json 2>/dev/null
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > gpurun_out/r3_bench_full.json 2> gpurun_out/r3_bench_full.err
python - <<'PY'
import json
for f in ("r3_dg","r3_bench_full"):
    d=json.loads(open("gpurun_out/%s.json"%f).read().strip().splitlines()[-1])
    print(f, d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["kernel"], d["roofline"]["shape"], d["roofline"]["avg_launch_us"])
    for k in d["kernels"][:12]: print("   ", k["kernel"], k["shape"], round(k["avg_us"],1), round(k["bound_frac"],3))
PY
