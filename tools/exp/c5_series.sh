#!/bin/bash
# per-launch durations of the C5 chain kernel over a run (kernel trace), to see where the spread comes from
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt5; mkdir -p /tmp/kt5
rocprofv3 --kernel-trace --output-format csv -d /tmp/kt5 -- python $GRAFT_REPO_ROOT/tools/run_objective.py c5 400 > $GRAFT_REPO_ROOT/gpurun_out/c5_series_run.txt 2>&1
f=$(find /tmp/kt5 -name '*kernel_trace.csv' | head -1)
python - "$f" <<'PY' > $GRAFT_REPO_ROOT/gpurun_out/c5_series.txt
import csv, sys, numpy as np
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
d = np.array([int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows if "k_chain_iter<2" in r["Kernel_Name"]]) / 1e3
s = np.array([int(r["Start_Timestamp"]) for r in rows if "k_chain_iter<2" in r["Kernel_Name"]]) / 1e3
print("n", len(d), "mean %.1f min %.1f p10 %.1f p50 %.1f p90 %.1f p99 %.1f max %.1f" % (d.mean(), d.min(), *np.percentile(d, [10, 50, 90, 99]), d.max()))
print("series (every launch, first 120):", np.round(d[:120], 1).tolist())
print("series 300..420:", np.round(d[300:420], 1).tolist())
print("start-to-start (us) 300..360:", np.round(np.diff(s)[300:360], 1).tolist())
PY
