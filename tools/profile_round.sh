#!/bin/bash
# tools/profile_round.sh <tag>: the evidence bundle of profiles/ (run on the GPU box through gpurun)
#   kernel-trace stats, three PMC passes, the bench line; everything lands in gpurun_out/prof_<tag>/
tag=${1:-rXX}
out=$GRAFT_REPO_ROOT/gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
B="python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --reps 1 --no-cpu-baseline --no-unfused"
# (the kernel trace of bench.py's DEFAULT command: --steps 5 --warmup 1; the short form below is for the counter passes only.  The average
#  does not depend on the run's length beyond that: 15.9 us at 2 steps, 15.5 at 5, 20 and 60 - tools/exp/r3_rocprof_steps.sh)
rm -rf /tmp/kt && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 1 --reps 1 --no-cpu-baseline --no-unfused > /dev/null 2>&1
cp $(find /tmp/kt -name "*kernel_stats.csv" | head -1) $out/kernel_stats.csv
echo "# kernel_source_sha16=$(cd $GRAFT_REPO_ROOT && python -c 'import bench; print(bench.kernel_source_hash())')" > $out/pmc_summary.txt
# (the persistent chain kernel covers many iterations per launch: per-iteration figures = totals over the chain kernels / iterations of the command;
#  bench.py runs warmup + steps + one profiled step of 200 iterations each)
echo "# iterations_kernel_trace=1400 iterations_pmc=800" >> $out/pmc_summary.txt
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F64 SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_INSTS_LDS" "SQC_ICACHE_REQ SQC_ICACHE_MISSES TCC_HIT_sum TCC_MISS_sum"; do
  rm -rf /tmp/pm && timeout 600 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d /tmp/pm -- $B > /dev/null 2>&1
  python $GRAFT_REPO_ROOT/tools/summarise_pmc.py $(find /tmp/pm -name "*counter_collection.csv" | head -1) >> $out/pmc_summary.txt
done
# the effective clock under the kernels (VERDICT r5 "Next #3a"): GRBM_GUI_ACTIVE / dispatch duration, its own pass
rm -rf /tmp/pm && timeout 600 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d /tmp/pm -- $B > /dev/null 2>&1
python $GRAFT_REPO_ROOT/tools/clock_from_pmc.py /tmp/pm >> $out/pmc_summary.txt
cd $GRAFT_REPO_ROOT && python bench.py > $out/bench_line.json 2> $out/bench_stderr.txt
tail -c 3000 $out/bench_line.json
