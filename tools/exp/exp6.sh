cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -x -q -m gpu -k "eval_batch or c1_serial or c2_full or norm_kernel or general_dims or dense or user" > gpurun_out/e6_tests.txt 2>&1
tail -5 gpurun_out/e6_tests.txt
python tools/dbg_ts.py > gpurun_out/e6_ts.txt 2>&1; sed -n 1,7p gpurun_out/e6_ts.txt; grep -A5 "per-WG phase" gpurun_out/e6_ts.txt
python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-unfused > gpurun_out/e6_bench.json 2> gpurun_out/e6_bench.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/e6_bench.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["avg_kernel_us"], d["roofline"]["net_kernel_us"], d["roofline"]["frac"])
PY
