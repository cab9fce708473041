set -x
cd $GRAFT_REPO_ROOT
python tools/dbg_ts.py > gpurun_out/e1_ts.txt 2>&1
python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-unfused > gpurun_out/e1_bench_default.json 2> gpurun_out/e1_bench_default.err
HIP_FORCE_DEV_KERNARG=1 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-unfused > gpurun_out/e1_bench_devkernarg.json 2> gpurun_out/e1_bench_devkernarg.err
HIP_FORCE_DEV_KERNARG=0 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-unfused > gpurun_out/e1_bench_hostkernarg.json 2> gpurun_out/e1_bench_hostkernarg.err
cat gpurun_out/e1_ts.txt
python - <<'PY'
import json
for n in ("default","devkernarg","hostkernarg"):
    try:
        d=json.loads(open("gpurun_out/e1_bench_%s.json"%n).read().strip().splitlines()[-1])
        print(n, d["value"], d["ms_per_step"], d["roofline"]["avg_kernel_us"], d["roofline"]["net_kernel_us"])
    except Exception as e: print(n, "ERR", e)
PY
