#!/bin/bash
# round-4 evidence bundle (run on the GPU box): C2 kernel stats + PMC + bench line (tools/profile_round.sh), the persistent kernel's
# in-kernel phase times and the plan kernel's, the other BASELINE configurations' bench lines, C4 / C5 rocprofv3 bundles
cd $GRAFT_REPO_ROOT
tools/profile_round.sh r04 > /dev/null 2>&1
out=$GRAFT_REPO_ROOT/gpurun_out/prof_r04
{ echo "# tools/persist_time.py 5: the persistent chain kernel against the one-launch-per-iteration kernel (C2: 4096 chains, ns = 10000), in-kernel phase times"
  echo "# of the control wave (SMMHIP_TS=1: wall-clock sums of every tile / iterations of the launch)"
  python tools/persist_time.py 5 2>&1 | grep -v "^\[W\|amdgpu"
  echo; echo "# tools/persist_gen_time.py 5: the persistent chain kernel of objectives without a simulation (C4: banana, 10 parameters, 8192 chains; its own problem"
  echo "# instance — bench.py's is another: see bench_c4.json); in-kernel phase times of wave 0 of every workgroup"
  python tools/persist_gen_time.py 5 2>&1 | grep -v "^\[W\|amdgpu"
  echo; echo "# tools/c5_tail.py 10 1000000 2: C5 (dense, 50 parameters) over 2000 iterations, us per iteration by block of 200: the late tries of mysample in rounds of one"
  echo "# try per lane segment only (round 3) / scouted by groups of 16 lanes after two such rounds (test build: SMMHIP_SCOUT_AFTER)"
  python tools/c5_tail.py 10 1000000 2 2>&1 | grep -v "^\[W\|amdgpu"
  echo; echo "# tools/plan_ts.py: k_exch_plan (one workgroup per iteration of a look-ahead window of 256; with the tiles' cones), workgroup 0"
  python tools/plan_ts.py 2>&1 | grep -v "^\[W\|amdgpu"
  echo; echo "# tools/persist_proto (the kill-criterion prototype of VERDICT r3 #1), 1000 iterations: mode 0 (cones), 1 (no exchange), 2 (every tile gathers all slots)"
  (cd tools && hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o persist_proto persist_proto.hip 2>/dev/null; for m in 0 1 2; do ./persist_proto 1000 $m | tail -2; done)
} > $out/phase_stamps.txt
B="python bench.py --no-cpu-baseline"
for w in c3 c4 c5; do timeout 300 $B --workload $w 2>/dev/null | grep "^{" > $out/bench_$w.json; done
python tools/exch_time.py > $out/exch_time.txt 2>&1
tools/profile_objectives.sh r04 > /dev/null 2>&1
for w in c4 c5; do timeout 300 $B --workload $w 2>/dev/null | grep "^{" > $out/bench_$w.json; done   # (un-profiled lines, now that the bundles they cite exist ... on the NEXT run: see profiles/README.md)
tail -c 400 $out/bench_line.json; echo; cat $out/phase_stamps.txt | head -30
