#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -8 > gpurun_out/tries_tests.txt
timeout 900 python tools/fuzz_parity.py 120 7 > gpurun_out/tries_fuzz.txt 2>&1
python tools/run_objective.py c5 400 > gpurun_out/tries_c5.txt 2>&1
python tools/run_objective.py c5 400 >> gpurun_out/tries_c5.txt 2>&1
