#!/bin/bash
# tools/profile_objectives.sh <tag>: rocprofv3 evidence for BASELINE configs C4 (banana, 8192 chains) and C5 (dense, FP64 MFMA)
tag=${1:-rXX}
out=$GRAFT_REPO_ROOT/gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for cfg in c4 c5; do
  B="python $GRAFT_REPO_ROOT/tools/run_objective.py $cfg 400"
  rm -rf /tmp/kt && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -- $B > $out/${cfg}_run.txt 2>&1
  cp $(find /tmp/kt -name "*kernel_stats.csv" | head -1) $out/${cfg}_kernel_stats.csv
  : > $out/${cfg}_pmc_summary.txt
  for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_INSTS_LDS"; do
    rm -rf /tmp/pm && timeout 600 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d /tmp/pm -- $B > /dev/null 2>&1
    python $GRAFT_REPO_ROOT/tools/summarise_pmc.py $(find /tmp/pm -name "*counter_collection.csv" | head -1) | grep -a "k_chain_iter\|k_exch_resolve\|^#" >> $out/${cfg}_pmc_summary.txt
  done
  grep -a "chains x" $out/${cfg}_run.txt
  head -4 $out/${cfg}_kernel_stats.csv | cut -c1-200
done
