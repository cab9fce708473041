cd $GRAFT_REPO_ROOT
R=${1:-3}; shift
for i in $(seq $R); do
  for L in "$@"; do
    SMMHIP_LIB=$PWD/smm.jl_amd/csrc/$L python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-unfused 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$L', round(d['value']/1e6,1), 'M/s  kernel', round(d['roofline']['avg_kernel_us'],2), 'us')"
  done
done
