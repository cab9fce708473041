#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "wide_walk or key_walk or lean_walk" 2>&1 | tail -15 > gpurun_out/wide_tests.txt
TS_MIN_IMPROVE=0.05 python tools/dbg_ts.py 2>/dev/null | sed -n 1,22p > gpurun_out/wide_ts.txt
SMMHIP_KEY_WALK=0 TS_MIN_IMPROVE=0.05 python tools/dbg_ts.py 2>/dev/null | sed -n 1,22p > gpurun_out/wide_ts_old.txt
python tools/dbg_ts.py 2>/dev/null | sed -n 1,22p > gpurun_out/wide_ts_keys.txt
