#!/bin/bash
cd $GRAFT_REPO_ROOT
B="python bench.py --steps 5 --warmup 1 --no-cpu-baseline"
for m in 2 0 1; do
  export SMMHIP_P2P_MEM=$m
  echo "== mem mode $m"
  timeout 300 python -m pytest tests/test_gpu_p2p.py -m gpu -x -q 2>&1 | tail -2
  for cfg in "SMM_BENCH_FORCE_SHARDED=1 $B --protocol p2p" "$B --gpus 2 --same-device" "$B --gpus 4 --same-device"; do
    eval "timeout 300 env $cfg" 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline'] or {}
        print('   n=%d  %.1f M/s  iter %.2f us  kernel %.2f' % (d['n_gpus'], d['value']/1e6, d['ms_per_step']*5, r.get('avg_kernel_us') or 0))
"
  done
done
