#!/bin/bash
# poll socket power and shader clock while a command runs:  tools/power_poll.sh OUT -- cmd ...
OUT=$1; shift; shift
( while true; do rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power \(W\)|sclk" | tr '\n' ' '; echo; sleep 0.05; done ) > $OUT &
P=$!
"$@"
kill $P
wait $P 2>/dev/null
python3 - "$OUT" <<'PY'
import re, sys
pw, ck = [], []
for l in open(sys.argv[1]):
    m = re.search(r"Power \(W\): ([\d.]+)", l); c = re.search(r"sclk clock level: \d+: \((\d+)Mhz\)", l)
    if m and c: pw.append(float(m.group(1))); ck.append(int(c.group(1)))
busy = [(p, c) for p, c in zip(pw, ck) if p > 600]
if busy:
    print("samples %d busy %d  power mean %.0f W max %.0f W   sclk mean %.0f MHz min %d max %d" % (
        len(pw), len(busy), sum(p for p, _ in busy) / len(busy), max(p for p, _ in busy),
        sum(c for _, c in busy) / len(busy), min(c for _, c in busy), max(c for _, c in busy)))
else:
    print("no busy samples", len(pw), pw[:5], ck[:5])
PY
