#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 1500 python tools/fuzz_parity.py 220 21 > gpurun_out/fz_small.txt 2>&1
timeout 1500 python tools/fuzz_parity.py 36 22 big > gpurun_out/fz_big.txt 2>&1
