#!/bin/bash
# round-6 evidence bundle (run on the GPU box): C2 kernel stats + PMC + clock + bench line (tools/profile_round.sh), the other BASELINE configurations' bench lines and
# rocprofv3 bundles with C5 = BASELINE config 5 AS WORDED (SMM_OBJ_DENSE2: the 256 x 256 stage) and the instance without it (c5v1), the in-kernel phase times,
# the dense tile function in isolation (tools/dense2_bench), the shards as processes on the one GPU, user objectives in the persistent loops, the
# bit-exactness check at every BASELINE shape, the enumerated error sequences, the register-walk prototype, the fuzz sweeps
cd $GRAFT_REPO_ROOT
tools/profile_round.sh r06 > /dev/null 2>&1
out=$GRAFT_REPO_ROOT/gpurun_out/prof_r06
{ echo "# tools/persist_time.py 5: the persistent chain kernel (k_chain_persist_loc) against the one-launch-per-iteration kernel (C2: 4096 chains, ns = 10000), in-kernel phase times"
  python tools/persist_time.py 5 2>&1 | grep -v "^\[W\|amdgpu"
  echo; echo "# PT_MIN_IMPROVE=0.5 tools/persist_time.py 5: the same with the reference's DEFAULT threshold (AlgoBGP.jl:522; the 16-byte slots' form)"
  PT_MIN_IMPROVE=0.5 python tools/persist_time.py 5 2>&1 | grep -v "^\[W\|amdgpu"
  echo; echo "# tools/persist_gen_time.py 5: the persistent chain kernel of objectives without a simulation (banana, 10 parameters, 8192 chains)"
  python tools/persist_gen_time.py 5 2>&1 | grep -v "^\[W\|amdgpu"
  echo; echo "# PG_INSTANCE=bench tools/persist_gen_time.py 5: the same kernel on the BENCH's C4 instance (smm.jl_amd/workloads.py; EXPERIMENTS R6.4)"
  PG_INSTANCE=bench python tools/persist_gen_time.py 5 2>&1 | grep -v "^\[W\|amdgpu"
  echo; echo "# tools/persist_tile_time.py c5 / c5v1 / norm6 3: the persistent TILE kernel — BASELINE config 5 as worded (SMM_OBJ_DENSE2), the instance without the 256 x 256 stage,"
  echo "# objfunc_norm with 6 parameters (ns = 10000) — against the one-launch-per-iteration kernels, 4096 chains"
  for w in c5 c5v1 norm6; do python tools/persist_tile_time.py $w 3 2>&1 | grep -v "^\[W\|amdgpu"; done
  echo; echo "# tools/persist_tile_time.py c4user 3 4096: C4's banana as a map-reduce user objective in the tile kernel (16-chain workgroups: what such a cone costs to walk)"
  python tools/persist_tile_time.py c4user 3 4096 2>&1 | grep -v "^\[W\|amdgpu"
  echo; echo "# tools/plan_ts.py 4096: k_exch_plan's phases (workgroup 0 of the last launch; EXPERIMENTS R6.5)"
  python tools/plan_ts.py 4096 2>&1 | grep -v "^\[W\|amdgpu" | tail -10
  echo; echo "# tools/dense2_bench: the FP64 matrix pipe alone, and the dense objective's tile function (spec v1 / v2) in isolation (per evaluation of 256 tiles)"
  tools/dense2_bench 2>&1
  echo; echo "# tools/regwalk_proto: the exchange walk in registers against the level-parallel walk on LDS slots"
  tools/regwalk_proto 2>&1
  echo; echo "# tools/bench_objectives.py: user objectives in the persistent loops (one thread per evaluation: gen_user; map-reduce: tile_user) / three launches per iteration"
  python tools/bench_objectives.py 2>&1 | grep -v "^\[W\|amdgpu"
} > $out/phase_stamps.txt
{ python tools/exp/sharded_persist_time.py 2>&1 | grep -v "^\[W\|amdgpu"; echo; python tools/exp/shard_plan_time.py 2>&1 | grep "^rank"; } > $out/sharded_persist.txt
B="python bench.py --no-cpu-baseline"
for g in 2 4; do timeout 600 $B --gpus $g --same-device 2>/dev/null | grep "^{" > $out/bench_same_device_$g.json; done
tools/profile_objectives.sh r06 > /dev/null 2>&1
timeout 300 $B --workload c5v1 2>/dev/null | grep "^{" > $out/bench_c5v1.json
python tools/exact_check.py > $out/exact_check.txt 2>&1
{ python -m pytest tests/test_gpu_error_enumeration.py -m gpu -q 2>&1 | tail -3; } > $out/error_enumeration_run.txt
{ timeout 900 python tools/fuzz_tile.py 60 61 2>&1 | grep -v "^\[W\|amdgpu" | tail -4; timeout 900 python tools/fuzz_r5.py 30 61 2>&1 | grep -v "^\[W\|amdgpu" | tail -4;
  timeout 600 python tools/fuzz_errors.py 40 61 2>&1 | grep -v "^\[W\|amdgpu" | tail -3; timeout 600 python tools/fuzz_parity.py 40 61 2>&1 | grep -v "^\[W\|amdgpu" | tail -3;
  timeout 600 python tools/fuzz_cones.py 60 61 2>&1 | grep -v "^\[W\|amdgpu" | tail -3; } > $out/fuzz.txt
tail -c 600 $out/bench_line.json; echo; cat $out/exact_check.txt | tail -12; cat $out/fuzz.txt
