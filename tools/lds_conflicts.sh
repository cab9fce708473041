#!/bin/bash
# tools/lds_conflicts.sh: LDS bank-conflict cycles of the chain kernels of every bench workload (one rocprofv3 --pmc pass each, kernel-trace only):
# SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE (cycles the LDS spent on conflicts / cycles it was busy), per kernel
cd /tmp && export TMPDIR=/tmp
for w in c2 c4 c5 c3; do
  rm -rf /tmp/ldsc && mkdir -p /tmp/ldsc
  timeout 600 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS --output-format csv -d /tmp/ldsc -- python $GRAFT_REPO_ROOT/bench.py --workload $w --steps 2 --warmup 1 --no-cpu-baseline --no-unfused > /dev/null 2>&1
  echo "# workload $w"
  python $GRAFT_REPO_ROOT/tools/summarise_pmc.py $(find /tmp/ldsc -name "*counter_collection.csv" | head -1) | grep -a "k_chain\|k_exch_plan\|k_cone" | cut -c1-200
done
