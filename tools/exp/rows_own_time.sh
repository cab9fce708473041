#!/bin/bash
cd /tmp && export TMPDIR=/tmp
for cfg in "8 32768" "4 16384" "2 8192"; do
rm -rf /tmp/kt && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -- python $GRAFT_REPO_ROOT/tools/exp/rows_own_time.py $cfg > /tmp/o.txt 2>&1
tail -1 /tmp/o.txt
python - <<'PY'
import csv, glob
f = glob.glob("/tmp/kt/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:5]:
    print("  %-72s calls %6s avg %9.2f us" % (r["Name"][:72], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done
