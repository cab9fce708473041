#!/bin/bash
# kernel-trace statistics of bench.py's default command (top kernels)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-unfused "$@" > /dev/null 2>&1
python - <<'PY'
import csv, glob
f = glob.glob("/tmp/kt/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:8]:
    print("%-70s calls %6s avg %9.2f us  %5s %%" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
