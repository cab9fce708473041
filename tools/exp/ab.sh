# A/B of two builds of libsmmhip.so in ONE session on the same box: tools/exp/ab.sh <libA> <libB> [rounds]
cd $GRAFT_REPO_ROOT
A=$1; B=$2; R=${3:-3}
for i in $(seq $R); do
  for L in $A $B; do
    SMMHIP_LIB=$PWD/$L python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-unfused 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$L', round(d['value']/1e6,1), 'M/s  kernel', round(d['roofline']['avg_kernel_us'],2), 'us')"
  done
done
