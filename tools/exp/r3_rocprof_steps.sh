#!/bin/bash
# does the rocprofv3 average of the chain kernel depend on how long the profiled run is?
cd /tmp && export TMPDIR=/tmp
for st in 2 5 20 60; do
  rm -rf /tmp/kts && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kts -- python $GRAFT_REPO_ROOT/bench.py --steps $st --warmup 1 --no-cpu-baseline --no-unfused > /tmp/b.json 2>/dev/null
  python - $st <<'P'
import csv,glob,sys,json
f=glob.glob("/tmp/kts/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "k_chain_iter_norm<2, true>" in r["Name"]:
        print("steps %3s: rocprofv3 calls %5s avg %.2f us min %.2f max %.2f" % (sys.argv[1], r["Calls"], float(r["AverageNs"])/1e3, float(r["MinNs"])/1e3, float(r["MaxNs"])/1e3), end="   ")
d=json.loads([l for l in open("/tmp/b.json") if l.startswith("{")][-1]); print("bench under the profiler: %.1f M/s, events %.2f us" % (d["value"]/1e6, d["roofline"]["avg_kernel_us"]))
P
done
