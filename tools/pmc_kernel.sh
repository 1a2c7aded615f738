#!/bin/bash
# Runs ON THE GPU BOX: hardware counters of the kernels of one bench step whose name contains a substring.
#   bash tools/pmc_kernel.sh SUBSTRING [bench args]   -> gpurun_out/pmc_kernel.txt
# One rocprofv3 --pmc pass per counter group (no tracing flags next to --pmc); FETCH_SIZE / WRITE_SIZE in KiB, FETCH doubled
# as MI355X_MICROARCH.md prescribes for gfx950.
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
SUB=$1; shift
D=gpurun_out/pmc_kernel; rm -rf $D; mkdir -p $D
GROUPS_=("SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"
         "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"
         "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM"
         "SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_LDS_UNALIGNED_STALL"
         "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum")
i=0
for g in "${GROUPS_[@]}"; do
  ( cd /tmp && rocprofv3 --pmc $g -d $OLDPWD/$D/g$i -o p --output-format csv -- python $OLDPWD/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras "$@" > /dev/null 2> $OLDPWD/$D/g$i.err )
  i=$((i+1))
done
SUB="$SUB" python - <<'PY'
import collections, csv, glob, os
sub = os.environ["SUB"]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pmc_kernel/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("pcops_mlp::", "").split("(")[0].replace("void ", "")
        if sub in name:
            agg[(name, int(r["Grid_Size"]))][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open("gpurun_out/pmc_kernel.txt", "w") as out:
    for (k, g), c in sorted(agg.items()):
        line = "%s grid %d" % (k, g)
        print(line); out.write(line + "\n")
        for n, v in sorted(c.items()):
            line = "    %-28s %16.0f  (n=%d)" % (n, sum(v) / len(v), len(v))
            print(line); out.write(line + "\n")
PY
