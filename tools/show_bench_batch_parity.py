import json
d=json.load(open("gpurun_out/parity_bench_batch.json"))
for k,v in d.items():
    print(k, {a:v[a] for a in ("batch","loss_fused","loss_ref","masked_gradient_error","unmasked_gradient_error","worst_variable")}, v["flips"]["relu_flips"], v["flips"]["relu_elements"])
    pv=sorted(v["relative_error_per_variable"].items(), key=lambda kv:-kv[1])[:6]
    print("   ", pv)
