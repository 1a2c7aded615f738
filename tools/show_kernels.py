"""Pretty-print the `kernels` table of a bench.py JSON line (stdin or file)."""
import json
import sys

line = [l for l in (open(sys.argv[1]) if len(sys.argv) > 1 else sys.stdin) if l.startswith("{")][-1]
d = json.loads(line)
print("%.1f %s  %.2f ms/step" % (d["value"], d["unit"], d["ms_per_step"]))
tot = 0.0
for k in d["kernels"]:
    ms = k["avg_us"] * k["launches"] / d.get("kernels_steps", d["steps"]) / 1e3
    tot += ms
    print("%-28s %-34s n/step %4.1f avg %8.1f us  %7.0f GB/s %9.0f G%s/s  %.3f ms/step" % (
        k["kernel"].replace("pcops_", ""), k["shape"], k["launches"] / d.get("kernels_steps", d["steps"]), k["avg_us"], k["gbs"],
        k["gwork_s"], k["work_unit"] or "-", ms))
print("listed total %.2f ms/step" % tot)
