#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -6 > gpurun_out/wide_tests.txt
timeout 900 python tools/fuzz_parity.py 150 11 > gpurun_out/wide_fuzz.txt 2>&1
python bench.py --no-cpu-baseline > gpurun_out/wide_bench.json 2> gpurun_out/wide_bench.err
