#!/bin/bash
cd $GRAFT_REPO_ROOT
TS_MIN_IMPROVE=0.05 python tools/dbg_ts.py 2>/dev/null | sed -n 1,22p > gpurun_out/wide_ts.txt
( EXCH_MIN_IMPROVE=0.05 python tools/exch_time.py 2048 4096 6000 7400 8192; echo old; SMMHIP_KEY_WALK=0 EXCH_MIN_IMPROVE=0.05 python tools/exch_time.py 2048 4096 6000 7400 8192 ) > gpurun_out/wide_exch.txt 2>&1
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "wide_walk" 2>&1 | tail -5 > gpurun_out/wide_tests.txt
