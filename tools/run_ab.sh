python -m pytest tests/test_fused_mlp_gpu.py tests/test_bn_shifted_moments_gpu.py tests/test_bench_size_gpu.py -q -x 2>&1 | tail -15 > gpurun_out/r3_t3.log
tail -3 gpurun_out/r3_t3.log
PCOPS_LIB=scanobjectnn_amd/libpcops_prof.so python tools/phase_prof.py 2>&1 | grep -v amdgpu.ids
python tools/bench_gemm.py fwd 10 2>&1 | grep -v amdgpu.ids
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > gpurun_out/r3_bench_full.json 2> gpurun_out/r3_bench_full.err
python - <<'PY'
import json
for f in ("full",):
    try:
        d=json.loads(open("gpurun_out/r3_bench_%s.json"%f).read().strip().splitlines()[-1])
        print(f, d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["kernel"], d["roofline"]["shape"], d["roofline"]["avg_launch_us"])
        for k in d["kernels"][:24]: print("   ", k["kernel"], k["shape"], round(k["avg_us"],1), round(k["bound_frac"],3))
    except Exception as e: print(f, "ERR", e)
PY
