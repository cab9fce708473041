#!/bin/bash
# tools/gaps.sh [workload]: the stream's timeline around the persistent launches (rocprofv3 --kernel-trace of bench.py): every dispatch with its
# duration and the idle gap in front of it, for the last 45 dispatches; and the totals (busy, idle) over the timed steps
w=${1:-c2}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt && timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/kt -- python $GRAFT_REPO_ROOT/bench.py --workload $w --steps 5 --warmup 1 --no-cpu-baseline --no-unfused > /dev/null 2>&1
python - $(find /tmp/kt -name "*kernel_trace.csv" | head -1) $(find /tmp/kt -name "*memory_copy_trace.csv" | head -1) <<'PY'
import csv, sys
ev = []
for r in csv.DictReader(open(sys.argv[1])):
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("(anonymous namespace)::", "")[:60]))
if len(sys.argv) > 2 and sys.argv[2]:
    try:
        for r in csv.DictReader(open(sys.argv[2])):
            ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "memcpy " + r.get("Direction", "")))
    except Exception as e:
        print("# no copy trace:", e)
ev.sort()
last = None
rows = []
for s, e, n in ev:
    gap = (s - last) / 1e3 if last is not None else 0.0
    rows.append((gap, (e - s) / 1e3, n))
    last = max(last or e, e)
for gap, dur, n in rows[-45:]:
    print("gap %8.1f us | %9.1f us  %s" % (gap, dur, n))
tail = rows[len(rows) // 3:]
print("# last two thirds of the run: busy %.1f us, idle gaps %.1f us (gaps > 1 ms not counted: %d)" % (
    sum(d for g, d, n in tail), sum(g for g, d, n in tail if g < 1000), sum(1 for g, d, n in tail if g >= 1000)))
PY
