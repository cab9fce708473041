#!/bin/bash
o=$GRAFT_REPO_ROOT/gpurun_out/r3q; mkdir -p $o; cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_p2p.py -m gpu -x -q 2>&1 | tail -5
B="python bench.py --steps 5 --warmup 1 --no-cpu-baseline"
SMM_BENCH_FORCE_SHARDED=1 $B --protocol p2p > $o/sh_p2p.json 2> $o/sh_p2p.err
timeout 300 $B --gpus 2 --same-device > $o/sd2.json 2> $o/sd2.err
timeout 300 $B --gpus 4 --same-device > $o/sd4.json 2> $o/sd4.err

for f in sh_p2p sd2 sd4; do python - $o/$f.json $f <<'P'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1]); r=d.get("roofline") or {}
    print("%-10s %8.1f M/s  iter %.2f us  kernel %s exch %s other %s frac %s  proto %s" % (sys.argv[2], d["value"]/1e6, d["ms_per_step"]*1e3/200, r.get("avg_kernel_us"), r.get("avg_exchange_us"), r.get("profiled_other_us"), r.get("frac"), d["config"].get("protocol")))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
P
done | tee $o/summary.txt
