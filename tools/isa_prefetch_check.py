"""reads gfx950 assembly of mlp.hip parts (hipcc -S --cuda-device-only) and reports, per gemm_ws / bwd_fused / wgrad kernel:
   * scratch loads / stores that sit inside a loop (a scratch reload is a vector-memory load: the first use of the reloaded
     register waits on vmcnt, and behind a prefetch that means waiting for the prefetch),
   * `s_waitcnt vmcnt(0)` between a group of prefetch loads (buffer_load_dwordx4) and the next matrix instruction.
   python tools/isa_prefetch_check.py part1.s [part2.s ...]"""
import re, subprocess, sys

for path in sys.argv[1:]:
    lines = open(path).read().split("\n")
    cur, funcs = None, {}
    for l in lines:
        m = re.match(r"^(_Z\w+):", l)
        if m:
            cur = m.group(1); funcs[cur] = []
        elif l.startswith(".Lfunc_end"):
            cur = None
        elif cur:
            funcs[cur].append(l)
    for name, ls in funcs.items():
        if not any("v_mfma" in x for x in ls):
            continue
        inloop, sc_loop, sc_all = False, 0, 0
        waits, last_load = [], None
        for i, l in enumerate(ls):
            if l.startswith(".LBB"):
                inloop = "in Loop" in l or "Loop Header" in l
            if i + 1 < len(ls) and ls[i].startswith(".LBB") and "Loop" in ls[i + 1]:
                inloop = True
            if "scratch_" in l:
                sc_all += 1
                sc_loop += inloop
            if "buffer_load_dwordx4" in l:
                last_load = i
            if "v_mfma" in l and last_load is not None:
                w = [x.strip() for x in ls[last_load:i] if "s_waitcnt" in x and "vmcnt(0)" in x]
                if w and i - last_load < 150:      # (further away it is a later phase's own wait, e.g. the streamed weights' store)
                    waits.append(i - last_load)
                last_load = None
        if sc_all or waits:
            d = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
            m = re.search(r"(\w+<.*>)\(", d)
            print("%-50s scratch ops %3d (in loops %3d)   vmcnt(0) behind a prefetch: %s" % (
                (m.group(1) if m else d)[:50], sc_all, sc_loop, waits or "-"))
