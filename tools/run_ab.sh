python -m pytest tests/test_fused_mlp_gpu.py tests/test_bn_shifted_moments_gpu.py -q -k "gather or edgeconv or Edge or large or compact" 2>&1 | tail -3
python bench.py --model dgcnn --steps 10 --warmup 3 --no-cpu-baseline --no-extras > gpurun_out/r3_dg.json 2>/dev/null
python - <<'PY'
import json
for f in ("r3_dg",):
    d=json.loads(open("gpurun_out/%s.json"%f).read().strip().splitlines()[-1])
    print(f, d["value"], d["ms_per_step"])
    for k in d["kernels"][:24]:
        print("   ", k["kernel"], k["shape"], round(k["avg_us"],1), round(k["bound_frac"],3))
PY
