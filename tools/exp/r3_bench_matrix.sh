#!/bin/bash
# round-3 bench matrix on a 1-GPU lease: single shard, the sharded forms on one GPU, multi-process p2p on the one device, other workloads
o=$GRAFT_REPO_ROOT/gpurun_out/r3m; mkdir -p $o; cd $GRAFT_REPO_ROOT
B="python bench.py --steps 5 --warmup 1"
$B > $o/n1.json 2> $o/n1.err
SMM_BENCH_FORCE_SHARDED=1 $B --no-cpu-baseline --protocol p2p > $o/sh_p2p.json 2> $o/sh_p2p.err
SMM_BENCH_FORCE_SHARDED=1 $B --no-cpu-baseline --protocol records > $o/sh_records.json 2> $o/sh_records.err
SMM_BENCH_FORCE_SHARDED=1 $B --no-cpu-baseline --protocol values > $o/sh_values.json 2> $o/sh_values.err
timeout 300 $B --no-cpu-baseline --gpus 2 --same-device > $o/sd2.json 2> $o/sd2.err
timeout 300 $B --no-cpu-baseline --gpus 4 --same-device > $o/sd4.json 2> $o/sd4.err
for w in c3 c4 c5; do timeout 300 $B --no-cpu-baseline --workload $w > $o/$w.json 2> $o/$w.err; done
for f in n1 sh_p2p sh_records sh_values sd2 sd4 c3 c4 c5; do python - $o/$f.json $f <<'P'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1]); r=d.get("roofline") or {}
    print("%-10s %8.1f M/s  iter %.2f us  kernel %s exch %s other %s frac %s  proto %s" % (sys.argv[2], d["value"]/1e6, d["ms_per_step"]*1e3/200, r.get("avg_kernel_us"), r.get("avg_exchange_us"), r.get("profiled_other_us"), r.get("frac"), d["config"].get("protocol")))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
P
done | tee $o/summary.txt
tail -3 $o/*.err | tail -40
