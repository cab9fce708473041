#!/bin/bash
# k_exch_plan's and the chain kernel's average duration per library build: tools/exp/plan_ab.sh libA.so libB.so ...
cd /tmp && export TMPDIR=/tmp
for lib in "$@"; do
  rm -rf /tmp/kt && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -- python $GRAFT_REPO_ROOT/tools/exp/bench_ab.py --child $lib > /tmp/out.txt 2>&1
  python - "$lib" <<'PY'
import csv, glob, sys, json
f = glob.glob("/tmp/kt/**/*kernel_stats.csv", recursive=True)
if not f: print(sys.argv[1], "no stats"); sys.exit()
r = {x["Name"]: x for x in csv.DictReader(open(f[0]))}
def avg(p):
    for n, x in r.items():
        if p in n: return float(x["AverageNs"]) / 1e3, int(x["Calls"])
    return (0, 0)
try: v = json.loads(open("/tmp/out.txt").read().strip().splitlines()[-1])["value"] / 1e6
except Exception: v = float("nan")
print("%-16s plan %8.1f us x%d   chain<2,true> %6.2f us   bench %.1f M/s" % (sys.argv[1], *avg("k_exch_plan("), avg("k_chain_iter_norm<2, true>")[0], v))
PY
done
