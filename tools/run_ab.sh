python -m pytest tests/test_fused_mlp_gpu.py tests/test_bn_shifted_moments_gpu.py tests/test_bench_size_gpu.py tests/test_deterministic_gpu.py tests/test_models_parity_gpu.py -q 2>&1 | tail -8
