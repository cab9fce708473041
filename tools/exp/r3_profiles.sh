#!/bin/bash
# round-3 evidence bundle (run on the GPU box): C2 kernel stats + PMC, the sharded forms on one GPU, multi-process p2p, C4 / C5
cd $GRAFT_REPO_ROOT
tools/profile_round.sh r03 > /dev/null 2>&1
out=$GRAFT_REPO_ROOT/gpurun_out/prof_r03
B="python bench.py --steps 5 --warmup 1 --no-cpu-baseline"
{
echo "# the sharded iteration on ONE MI355X (gpurun lease), objfunc_norm 2p/2m, ns = 10000: per-iteration time of the whole step (ms_per_step / 200),"
echo "# the chain kernel's own events, the stand-alone exchange kernel where there is one"
for cfg in "single shard (no sharding)|$B" \
           "1 rank, p2p windows (SMM_BENCH_FORCE_SHARDED=1)|SMM_BENCH_FORCE_SHARDED=1 $B --protocol p2p" \
           "1 rank, RCCL all-gather of records|SMM_BENCH_FORCE_SHARDED=1 $B --protocol records" \
           "1 rank, RCCL values + all-to-all|SMM_BENCH_FORCE_SHARDED=1 $B --protocol values" \
           "2 PROCESSES on the one GPU x 2048 chains, p2p over HIP IPC|$B --gpus 2 --same-device" \
           "4 PROCESSES on the one GPU x 1024 chains, p2p over HIP IPC|$B --gpus 4 --same-device" \
           "8 PROCESSES on the one GPU x 512 chains, p2p over HIP IPC|$B --gpus 8 --same-device" \
           "2 PROCESSES on the one GPU x 8192 chains (rows form), p2p over HIP IPC|$B --gpus 2 --same-device --chains 8192" \
           "4 PROCESSES on the one GPU x 4096 chains (rows form), p2p over HIP IPC|$B --gpus 4 --same-device --chains 4096" \
           "C3 as 8 PROCESSES on the one GPU x 4096 chains (rows form; time-sliced)|$B --gpus 8 --same-device --workload c3" \
           "16384 chains, single shard (what 4 GPUs hold)|$B --chains 16384" \
           "16384 chains, 1 rank, p2p rows form (SMM_BENCH_FORCE_SHARDED=1)|SMM_BENCH_FORCE_SHARDED=1 $B --chains 16384 --protocol p2p" \
           "C3 32768 chains, single shard|$B --workload c3" \
           "C3 32768 chains, 1 rank, p2p rows form (SMM_BENCH_FORCE_SHARDED=1)|SMM_BENCH_FORCE_SHARDED=1 $B --workload c3 --protocol p2p" \
           "C3 32768 chains, 1 rank, RCCL all-gather of records|SMM_BENCH_FORCE_SHARDED=1 $B --workload c3 --protocol records"; do
  name=${cfg%%|*}; cmd=${cfg#*|}
  eval "timeout 200 env $cmd" 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline'] or {}
        print('%-64s %6.1f M chain-evals/s  %6.2f us/iteration  chain kernel %s us  exchange kernel %s us' % ('$name', d['value']/1e6, d['ms_per_step']*5, ('%.2f' % r['avg_kernel_us']) if r.get('avg_kernel_us') else '-', ('%.2f' % r['avg_exchange_us']) if r.get('avg_exchange_us') else '-'))
"
done
} > $out/sharded.txt
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt2 && SMM_BENCH_FORCE_SHARDED=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt2 -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --protocol p2p > /dev/null 2>&1
cp $(find /tmp/kt2 -name "*kernel_stats.csv" | head -1) $out/p2p_kernel_stats.csv
cd $GRAFT_REPO_ROOT
bash tools/exp/r3_c3_stats.sh > $out/c3_kernel_stats.txt 2>&1
{ echo "# tools/exp/rows_own_time.sh: what each rank of 8 (32768 chains), of 4 (16384) and of 2 (8192: the inline form, one launch) launches per iteration, measured by running the shards as contexts of one"
  echo "# process on ONE GPU, one context at a time (rocprofv3 --kernel-trace --stats; objfunc_norm 2p/2m, ns = 10000, 40 iterations)"
  bash tools/exp/rows_own_time.sh 2>&1 | grep -v "^W2026"; } > $out/c3_per_rank.txt
cd $GRAFT_REPO_ROOT
python tools/exch_time.py > $out/exch_time.txt 2>&1
python tools/dbg_ts.py 2>&1 | grep -v "^\[W\|amdgpu" > $out/phase_stamps.txt
python tools/exp/ts_p2p.py 2>&1 | grep -v "^\[W\|amdgpu" >> $out/phase_stamps.txt
for w in c3 c4 c5; do timeout 300 $B --workload $w 2>/dev/null | grep "^{" > $out/bench_$w.json; done
tools/profile_objectives.sh r03 > /dev/null 2>&1
cat $out/sharded.txt; head -5 $out/kernel_stats.csv | cut -c1-160; tail -c 600 $out/bench_line.json
