python -m pytest tests/test_knn_gpu.py tests/test_models_parity_gpu.py tests/test_models_gpu.py -q -k "knn or dgcnn" 2>&1 | tail -6
python bench.py --model dgcnn --steps 10 --warmup 3 --no-cpu-baseline --no-extras > gpurun_out/r3_dg.json 2>/dev/null
PCOPS_KNN_SEED=0 python bench.py --model dgcnn --steps 10 --warmup 3 --no-cpu-baseline --no-extras > gpurun_out/r3_dg0.json 2>/dev/null
python - <<'PY'
import json
for f in ("r3_dg","r3_dg0"):
    d=json.loads(open("gpurun_out/%s.json"%f).read().strip().splitlines()[-1])
    print(f, d["value"], d["ms_per_step"])
    for k in d["kernels"][:24]:
        if "knn" in k["kernel"]: print("   ", k["kernel"], k["shape"], round(k["avg_us"],1), round(k["bound_frac"],3))
PY
