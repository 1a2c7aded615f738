"""Summarise hipcc's -Rpass-analysis=kernel-resource-usage remarks: one line per kernel (registers, spills, LDS,
occupancy).  Usage:
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Rpass-analysis=kernel-resource-usage \\
          -c scanobjectnn_amd/csrc/mlp.hip -o /tmp/x.o 2> /tmp/res.txt
    python tools/kernel_resources.py /tmp/res.txt [filter]
"""
import re
import subprocess
import sys


def main():
    text = open(sys.argv[1]).read()
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    rows = []
    cur = None
    for line in text.splitlines():
        m = re.search(r"remark:\s+(Function Name|TotalSGPRs|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|"
                      r"SGPRs Spill|VGPRs Spill|LDS Size \[bytes/block\]): (\S+)", line)
        if not m:
            continue
        k, v = m.group(1), m.group(2)
        if k == "Function Name":
            cur = {"name": v}
            rows.append(cur)
        elif cur is not None:
            cur[k.split(" ")[0] + ("Spill" if "Spill" in k else "")] = v
    names = subprocess.run(["c++filt"], input="\n".join(r["name"] for r in rows),
                           capture_output=True, text=True).stdout.splitlines()
    print("%-6s %-6s %-6s %-8s %-5s %-7s %s" % ("VGPR", "AGPR", "SGPR", "scratch", "occ", "vspill", "kernel"))
    for r, n in zip(rows, names):
        n = n.replace("(anonymous namespace)::", "").replace("pcops_mlp::", "").split("(")[0].replace("void ", "")
        if flt and flt not in n:
            continue
        print("%-6s %-6s %-6s %-8s %-5s %-7s %s" % (r.get("VGPRs"), r.get("AGPRs"), r.get("TotalSGPRs"), r.get("ScratchSize"),
                                                   r.get("Occupancy"), r.get("VGPRsSpill"), n))


if __name__ == "__main__":
    main()
