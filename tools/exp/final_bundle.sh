#!/bin/bash
cd $GRAFT_REPO_ROOT
bash tools/profile_round.sh r02 > gpurun_out/fb_round.txt 2>&1
bash tools/profile_objectives.sh r02 > gpurun_out/fb_obj.txt 2>&1
python tools/bench_objectives.py > gpurun_out/fb_objectives.txt 2>&1
python tools/dbg_ts.py > gpurun_out/fb_ts.txt 2>/dev/null
python tools/ts_objective.py c5 > gpurun_out/fb_c5ts.txt 2>&1
python tools/exch_time.py > gpurun_out/fb_exch.txt 2>&1
EXCH_MIN_IMPROVE=0.05 python tools/exch_time.py 2048 4096 7400 >> gpurun_out/fb_exch.txt 2>&1
