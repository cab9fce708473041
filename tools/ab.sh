#!/bin/bash
# A/B helper: tools/ab.sh "ENV1=.. ENV2=.." runs bench twice with and without the given env, prints value / kernel / exchange us
for rep in 1 2 3; do
  for envs in "" "$1"; do
    env $envs python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('%-24s value %.1fM  chain %.2f us  exch %.2f us' % ('$envs' or 'default', d['value']/1e6, r['avg_kernel_us'], r['avg_exchange_us']))"
  done
done
