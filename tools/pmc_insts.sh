#!/bin/bash
# Runs ON THE GPU BOX: instruction counts per kernel of the SSG bench step (rocprofv3 --pmc, no tracing flags).
#   bash tools/pmc_insts.sh [bench args]   -> gpurun_out/pmc_insts.txt   (per kernel x grid: VALU / SALU / LDS / MFMA
#   wave-instructions, matrix-pipe busy cycles, per-SIMD instruction-issue estimate)
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
D=gpurun_out/pmc_insts; rm -rf $D
( cd /tmp && rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d $OLDPWD/$D -o p --output-format csv -- python $OLDPWD/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras "$@" > /dev/null 2>&1 )
python - <<'PY'
import collections, csv, glob
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pmc_insts/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        agg[(r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("pcops_mlp::", "").split("(")[0].replace("void ", ""), int(r["Grid_Size"]))][r["Counter_Name"]].append(float(r["Counter_Value"]))
rows = []
for (k, g), c in agg.items():
    m = {n: sum(v) / len(v) for n, v in c.items()}
    act = m.get("GRBM_GUI_ACTIVE", 0) / 8.0
    rows.append((act, k, g, m))
out = open("gpurun_out/pmc_insts.txt", "w")
for act, k, g, m in sorted(rows, reverse=True)[:40]:
    valu, salu, lds, mf = m.get("SQ_INSTS_VALU", 0), m.get("SQ_INSTS_SALU", 0), m.get("SQ_INSTS_LDS", 0), m.get("SQ_INSTS_MFMA", 0)
    busy = m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / 1024.0
    line = "%-50s grid %8d  active %9.0f clk | VALU %6.1fM (MFMA %5.1fM) SALU %6.1fM LDS %6.1fM | per SIMD: insts %8.0f  x4.5 = %8.0f clk, mfma busy %8.0f clk" % (
        k[:50], g, act, valu / 1e6, mf / 1e6, salu / 1e6, lds / 1e6, (valu + salu + lds) / 1024, (valu + salu + lds - mf) / 1024 * 4.5, busy)
    print(line); out.write(line + "\n")
PY
