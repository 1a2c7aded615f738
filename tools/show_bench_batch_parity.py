"""The bench-batch parity record (tests/test_bench_size_gpu.py writes gpurun_out/parity_bench_batch.json; the committed copy is
profiles/rNN_parity_bench_batch.json): per model the whole-gradient figures and the six worst judged variables.
usage: show_bench_batch_parity.py [record.json]"""
import json
import sys

d = json.load(open(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/parity_bench_batch.json"))
for k, v in d.items():
    print(k, {a: v[a] for a in ("batch", "loss_fused", "loss_ref", "masked_gradient_error", "unmasked_gradient_error", "worst_variable")},
          v["flips"]["relu_flips"], v["flips"]["relu_elements"])
    pv = sorted(v["relative_error_per_variable"].items(), key=lambda kv: -kv[1])[:6]
    print("   ", pv)
    if "zero_gradient_variables" in v:
        print("    exact-zero gradients (not judged per variable):", len(v["zero_gradient_variables"]))
