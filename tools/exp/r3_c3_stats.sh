#!/bin/bash
# kernel-trace statistics of the C3 workload on one GPU: single shard, and one rank of the generic p2p form (what a rank of 8 GPUs launches)
out=$GRAFT_REPO_ROOT/gpurun_out/c3_stats
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
top() { python - "$1" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:10]:
    print("%-70s calls %6s avg %9.2f us  %5s %%" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
}
for mode in single p2p; do
  rm -rf /tmp/kt
  if [ $mode = p2p ]; then export SMM_BENCH_FORCE_SHARDED=1; X="--protocol p2p"; else unset SMM_BENCH_FORCE_SHARDED; X=""; fi
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -- python $GRAFT_REPO_ROOT/bench.py --workload c3 $X --steps 2 --warmup 1 --no-cpu-baseline --no-unfused > $out/run_$mode.txt 2>&1
  f=$(find /tmp/kt -name "*kernel_stats.csv" | head -1)
  cp $f $out/kernel_stats_$mode.csv
  echo "== $mode"; top $f
  python $GRAFT_REPO_ROOT/bench.py --workload c3 $X --no-cpu-baseline --no-unfused 2>/dev/null | grep "^{" | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('value %.4g  us/iter %.2f' % (d['value'], d['ms_per_step']*1e3/200))"
done
