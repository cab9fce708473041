#!/bin/bash
# tools/profile_objectives.sh <tag>: rocprofv3 evidence for BASELINE configs C3 (32768 chains on one GPU), C4 (banana, 8192 chains) and C5 (dense, FP64 MFMA).
# The profiled commands are bench.py's own (C5: its default --steps 8 = the steady state past 1600 iterations), so that every figure of
# profiles/<tag>_bench_c5.json can be recomputed from <tag>_c5_kernel_stats.csv and <tag>_c5_pmc_summary.txt; the per-launch
# distribution of the chain kernel (p50 / p99 / max over the run) goes to <tag>_c5_launches.txt.
tag=${1:-rXX}
out=$GRAFT_REPO_ROOT/gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for cfg in c3 c4 c5; do
  B="python $GRAFT_REPO_ROOT/bench.py --workload $cfg --reps 1 --no-cpu-baseline --no-unfused"
  rm -rf /tmp/kt && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -- $B > $out/${cfg}_run.txt 2>&1
  cp $(find /tmp/kt -name "*kernel_stats.csv" | head -1) $out/${cfg}_kernel_stats.csv
  python - $(find /tmp/kt -name "*kernel_trace.csv" | head -1) > $out/${cfg}_launches.txt <<'PY'
import csv, sys, collections
d = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    d[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print("# per-launch durations (us) from rocprofv3 --kernel-trace of `bench.py --workload ...` (its default steps): launches, mean, p50, p90, p99, max; and by")
print("# quarter of the run (the redraw tail of mysample grows as sigma adapts)")
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1]))[:6]:
    s = sorted(v); n = len(s)
    q = [sum(v[i * n // 4:(i + 1) * n // 4]) / max(1, len(v[i * n // 4:(i + 1) * n // 4])) for i in range(4)]
    print("%-70s n=%6d mean %8.2f p50 %8.2f p90 %8.2f p99 %8.2f max %8.2f | quarters %s" % (k[:70], n, sum(v) / n, s[n // 2], s[int(n * 0.9)], s[min(n - 1, int(n * 0.99))], s[-1], " ".join("%.1f" % x for x in q)))
PY
  : > $out/${cfg}_pmc_summary.txt
  # (C4's persistent chain kernel covers many iterations per launch: per-iteration figures = totals over the chain kernels / the iterations of the
  #  command; bench.py runs warmup + steps + one profiled step of 200 iterations each: 1 + 5 + 1 in the kernel trace, 1 + 2 + 1 in the counter passes)
  [ $cfg = c4 ] && echo "# iterations_kernel_trace=1400 iterations_pmc=800" >> $out/${cfg}_pmc_summary.txt
  # (C5's persistent tile kernel alike; its default is --steps 8: 1 + 8 + 1 steps in the kernel trace)
  [ $cfg = c5 ] && echo "# iterations_kernel_trace=2000 iterations_pmc=800" >> $out/${cfg}_pmc_summary.txt
  for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_INSTS_LDS"; do
    rm -rf /tmp/pm && timeout 600 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d /tmp/pm -- $B --steps 2 > /dev/null 2>&1
    python $GRAFT_REPO_ROOT/tools/summarise_pmc.py $(find /tmp/pm -name "*counter_collection.csv" | head -1) | grep -a "k_chain_iter\|k_chain_persist\|k_cone_\|k_exch_resolve\|k_exch_plan\|k_pregen\|^#" >> $out/${cfg}_pmc_summary.txt
  done
  rm -rf /tmp/pm && timeout 600 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d /tmp/pm -- $B --steps 2 > /dev/null 2>&1
  python $GRAFT_REPO_ROOT/tools/clock_from_pmc.py /tmp/pm | grep -a "k_chain_iter\|k_chain_persist" >> $out/${cfg}_pmc_summary.txt
  grep -a "^{" $out/${cfg}_run.txt > $out/bench_${cfg}.json
  head -4 $out/${cfg}_kernel_stats.csv | cut -c1-200
  cat $out/${cfg}_launches.txt | cut -c1-220
done
