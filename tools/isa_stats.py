"""Static instruction mix of one kernel in a hipcc -S listing, per basic block (label to label), with the loop structure
visible through the branch targets: how many MFMA / VALU / SALU / LDS / VMEM instructions a block issues.
    hipcc --offload-arch=gfx950 -O3 ... -S --cuda-device-only -o /tmp/mlp.s scanobjectnn_amd/csrc/mlp.hip
    python tools/isa_stats.py /tmp/mlp.s 'gemm_ws_kernelILi4ELi1ELi0ELi64ELi8ELi2ELi1E'
"""
import re
import sys


def classify(op):
    if op.startswith("v_mfma"):
        return "mfma"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("buffer_", "global_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith("s_waitcnt") or op.startswith("s_nop"):
        return "wait"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("v_"):
        return "valu"
    return "other"


def main():
    path, pat = sys.argv[1], sys.argv[2]
    lines = open(path).read().splitlines()
    start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\S*%s\S*:" % re.escape(pat), l))
    end = next(i for i in range(start, len(lines)) if lines[i].startswith("\t.section") or ".Lfunc_end" in lines[i])
    blocks, cur = [], {"label": "entry", "n": {}, "br": []}
    for l in lines[start + 1:end]:
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            blocks.append(cur)
            cur = {"label": m.group(1), "n": {}, "br": []}
            continue
        l = l.strip()
        if not l or l.startswith((";", ".")):
            continue
        op = l.split()[0]
        k = classify(op)
        cur["n"][k] = cur["n"].get(k, 0) + 1
        if op.startswith(("s_cbranch", "s_branch")):
            cur["br"].append(l.split()[-1])
    blocks.append(cur)
    tot = {}
    print("%-12s %5s %5s %5s %5s %5s %5s  %s" % ("block", "mfma", "valu", "salu", "lds", "vmem", "wait", "branches"))
    for b in blocks:
        n = b["n"]
        if sum(n.values()) == 0:
            continue
        for k, v in n.items():
            tot[k] = tot.get(k, 0) + v
        print("%-12s %5d %5d %5d %5d %5d %5d  %s" % (b["label"], n.get("mfma", 0), n.get("valu", 0), n.get("salu", 0),
                                                    n.get("lds", 0), n.get("vmem", 0), n.get("wait", 0), " ".join(b["br"])))
    print("%-12s %5d %5d %5d %5d %5d %5d" % ("static total", tot.get("mfma", 0), tot.get("valu", 0), tot.get("salu", 0),
                                              tot.get("lds", 0), tot.get("vmem", 0), tot.get("wait", 0)))


if __name__ == "__main__":
    main()
