#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "many_tries or dense or cholesky or no_draw or hard_error or general" 2>&1 | tail -15 > gpurun_out/tries_tests.txt
bash tools/exp/c5_series.sh
