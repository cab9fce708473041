for rep in 1 2 3; do for v in old new; do cp tools/exp/libsmmhip_$v.so smm.jl_amd/csrc/libsmmhip.so; for w in ${AB_W:-c2 c4 c5}; do python bench.py --workload $w --no-cpu-baseline --no-unfused 2>/dev/null | grep "^{" | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('$v $w value %.1fM  kernel %.2f us' % (d['value']/1e6, r['avg_kernel_us']))"; done; done; done
