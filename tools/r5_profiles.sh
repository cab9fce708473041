#!/bin/bash
# round-5 evidence bundle (run on the GPU box): C2 kernel stats + PMC + bench line (tools/profile_round.sh), in-kernel phase times of the
# persistent kernels (single shard, thresholds, shards as processes on the one GPU), the plan kernel's, the other BASELINE configurations'
# bench lines and rocprofv3 bundles, the user objective's table
cd $GRAFT_REPO_ROOT
tools/profile_round.sh r05 > /dev/null 2>&1
out=$GRAFT_REPO_ROOT/gpurun_out/prof_r05
{ echo "# tools/persist_time.py 5: the persistent chain kernel (k_chain_persist_loc) against the one-launch-per-iteration kernel (C2: 4096 chains, ns = 10000), in-kernel phase times"
  echo "# of the control wave (SMMHIP_TS=1: wall-clock sums of every tile / iterations of the launch)"
  python tools/persist_time.py 5 2>&1 | grep -v "^\[W\|amdgpu"
  echo; echo "# PT_MIN_IMPROVE=0.05 tools/persist_time.py 5: the same with the threshold of the reference's own test (the 16-byte slots' form)"
  PT_MIN_IMPROVE=0.05 python tools/persist_time.py 5 2>&1 | grep -v "^\[W\|amdgpu"
  echo; echo "# tools/persist_gen_time.py 5 [8192 | 2048]: the persistent chain kernel of objectives without a simulation (banana, 10 parameters)"
  python tools/persist_gen_time.py 5 2>&1 | grep -v "^\[W\|amdgpu"
  python tools/persist_gen_time.py 5 2048 2>&1 | grep -v "^\[W\|amdgpu"
  echo; echo "# tools/persist_tile_time.py c5 / norm6 / norm18 3: the persistent TILE kernel (k_chain_persist_tile: the dense objective of C5; objfunc_norm with 6 and 18"
  echo "# parameters, the reference's own larger examples, ns = 10000) against the one-launch-per-iteration kernels, 4096 chains"
  for w in c5 norm6 norm18; do python tools/persist_tile_time.py $w 3 2>&1 | grep -v "^\[W\|amdgpu"; done
  echo; echo "# tools/dense_bench: the dense objective's two products in isolation (per evaluation of 256 tiles)"
  tools/dense_bench 2>&1
  echo; echo "# tools/c5_tail.py 10: C5 (dense, 50 parameters; the instance of round 5) over 2000 iterations, us per iteration by block of 200"
  python tools/c5_tail.py 10 2 2>&1 | grep -v "^\[W\|amdgpu"
  echo; echo "# tools/ts_objective.py c5: a tile's timeline"
  python tools/ts_objective.py c5 2>&1 | grep -v "^\[W\|amdgpu" | head -12
  echo; echo "# tools/plan_ts.py: k_exch_plan (one workgroup per iteration of a look-ahead window of 256; with the tiles' cones), workgroup 0"
  python tools/plan_ts.py 2>&1 | grep -v "^\[W\|amdgpu"
  echo; echo "# tools/bench_objectives.py: user objectives in the persistent loop / three launches per iteration"
  python tools/bench_objectives.py 2>&1 | grep -v "^\[W\|amdgpu"
} > $out/phase_stamps.txt
{ python tools/exp/sharded_persist_time.py 2>&1 | grep -v "^\[W\|amdgpu"; echo; python tools/exp/shard_plan_time.py 2>&1 | grep "^rank"; } > $out/sharded_persist.txt
B="python bench.py --no-cpu-baseline"
for g in 2 4; do timeout 600 $B --gpus $g --same-device 2>/dev/null | grep "^{" > $out/bench_same_device_$g.json; done
python tools/exch_time.py > $out/exch_time.txt 2>&1
tools/profile_objectives.sh r05 > /dev/null 2>&1
for w in c3 c4 c5; do timeout 300 $B --workload $w 2>/dev/null | grep "^{" > $out/bench_$w.json; done
tools/lds_conflicts.sh > $out/lds_conflicts.txt 2>&1
{ timeout 900 python tools/fuzz_tile.py 90 31 2>&1 | grep -v "^\[W\|amdgpu"; timeout 900 python tools/fuzz_r5.py 40 31 2>&1 | grep -v "^\[W\|amdgpu"; } > $out/fuzz.txt
tail -c 400 $out/bench_line.json; echo; head -30 $out/phase_stamps.txt
