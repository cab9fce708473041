#!/bin/bash
cd $GRAFT_REPO_ROOT
B="python bench.py --steps 5 --warmup 1 --no-cpu-baseline"
timeout 300 python -m pytest tests/test_gpu_p2p.py -m gpu -x -q 2>&1 | tail -2
for cfg in "SMM_BENCH_FORCE_SHARDED=1 $B --protocol p2p" "$B --gpus 2 --same-device" "$B --gpus 4 --same-device" "$B --gpus 4 --same-device --chains 2048" "$B --gpus 2 --same-device --chains 4096"; do
  echo "== $cfg"
  eval "timeout 120 env $cfg" 2>/tmp/err.txt | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline'] or {}
        print('   n=%d chains/rank %d  %.1f M/s  iter %.2f us  kernel %.2f' % (d['n_gpus'], d['config']['chains_per_gpu'], d['value']/1e6, d['ms_per_step']*5, r.get('avg_kernel_us') or 0))
"
  grep -i "error\|Traceback" /tmp/err.txt | grep -v "^\[W" | head -3
done
