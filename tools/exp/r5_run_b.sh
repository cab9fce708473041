python -m pytest tests/test_gpu_persist_loc.py tests/test_gpu_persist_gen.py tests/test_bench_launcher.py -q -x 2>&1 | tail -8
python -m pytest tests/test_gpu_parity.py -q -x -k "c5 or dense" 2>&1 | tail -4
python bench.py --workload c5 --no-cpu-baseline > gpurun_out/r5_bench_c5.json 2> gpurun_out/r5_bench_c5.err
python bench.py --workload c4 --no-cpu-baseline > gpurun_out/r5_bench_c4.json 2> gpurun_out/r5_bench_c4.err
python bench.py --workload c3 --no-cpu-baseline > gpurun_out/r5_bench_c3.json 2> gpurun_out/r5_bench_c3.err
python tools/c5_tail.py 10 > gpurun_out/r5_c5_tail.txt 2>&1
for f in c5 c4 c3; do tail -1 gpurun_out/r5_bench_$f.json | cut -c1-330; echo; done
tail -4 gpurun_out/r5_c5_tail.txt
