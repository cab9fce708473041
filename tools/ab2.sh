#!/bin/bash
# tools/ab2.sh "ENVA" "ENVB" ... : bench under each env setting, 3 rounds
for rep in 1 2 3; do
  for envs in "$@"; do
    env $envs python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('%-32s value %.1fM  chain %.2f us  exch %.2f us' % ('$envs', d['value']/1e6, r['avg_kernel_us'], r['avg_exchange_us']))"
  done
done
