#!/bin/bash
cd $GRAFT_REPO_ROOT
for r in 1 2 3; do for lib in libsmmhip.so libsmmhip_plain.so; do
SMM_BENCH_FORCE_SHARDED=1 python - $lib <<'P' 2>/dev/null
import sys, os, json, subprocess
sys.path.insert(0, ".")
import smm_jl_amd
smm_jl_amd._abi.LIB_PATH = os.path.join("smm.jl_amd/csrc", sys.argv[1])
sys.argv = ["bench.py", "--steps", "5", "--warmup", "1", "--no-cpu-baseline", "--protocol", "p2p"]
import runpy, io, contextlib
buf = io.StringIO()
with contextlib.redirect_stdout(buf):
    runpy.run_path("bench.py", run_name="__main__")
d = json.loads([l for l in buf.getvalue().splitlines() if l.startswith("{")][-1])
print("%-22s %.1f M/s  iter %.2f us  kernel %.2f us" % (smm_jl_amd._abi.LIB_PATH.split("/")[-1], d["value"]/1e6, d["ms_per_step"]*5, d["roofline"]["avg_kernel_us"]))
P
done; done
