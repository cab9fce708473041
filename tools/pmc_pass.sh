#!/bin/bash
# tools/pmc_pass.sh <outdir> <counter> [<counter> ...] : one rocprofv3 --pmc pass of the bench (kernel-trace only)
out=$1; shift
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc_$$ && mkdir -p /tmp/pmc_$$
timeout 600 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d /tmp/pmc_$$ -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
f=$(find /tmp/pmc_$$ -name "*counter_collection.csv" | head -1)
python $GRAFT_REPO_ROOT/tools/summarise_pmc.py $f | grep -a "k_chain_iter\|k_exch_resolve" > $out
