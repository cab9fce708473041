cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "norm_kernel or hard_error or three_phase or c1_serial or c2_full or injected_tables or chunked or failbox or error_ or state_roundtrip or fused_sharded or sharded_equals or inline_walk or window" > gpurun_out/e4_tests.txt 2>&1
tail -15 gpurun_out/e4_tests.txt
python tools/dbg_ts.py > gpurun_out/e4_ts.txt 2>&1; head -30 gpurun_out/e4_ts.txt
python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-unfused > gpurun_out/e4_bench.json 2> gpurun_out/e4_bench.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/e4_bench.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["avg_kernel_us"], d["roofline"]["net_kernel_us"], d["roofline"]["frac"])
PY
