#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "dist_fun" 2>&1 | tail -15 > gpurun_out/dist_tests.txt
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -6 >> gpurun_out/dist_tests.txt
python bench.py --no-cpu-baseline > gpurun_out/dist_bench.json 2> gpurun_out/dist_bench.err
( python tools/exch_time.py; EXCH_MIN_IMPROVE=0.05 python tools/exch_time.py 2048 4096 7400; SMMHIP_KEY_WALK=0 python tools/exch_time.py 4096 8192 ) > gpurun_out/dist_exch.txt 2>&1
TS_MIN_IMPROVE=0.05 SMMHIP_KEY_WALK=0 python tools/dbg_ts.py 2>/dev/null | sed -n 8,10p > gpurun_out/dist_ts_old.txt
