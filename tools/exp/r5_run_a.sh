python -m pytest tests -m gpu -q -x 2>&1 | tail -12
python bench.py --steps 20 --warmup 2 --no-unfused > gpurun_out/r5_bench_c2.json 2> gpurun_out/r5_bench_c2.err
python bench.py --gpus 2 --same-device --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/r5_bench_2x2048.json 2> gpurun_out/r5_bench_2x2048.err
python bench.py --gpus 4 --same-device --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/r5_bench_4x1024.json 2> gpurun_out/r5_bench_4x1024.err
bash tools/profile_round.sh r05a > gpurun_out/r5_profile_round.log 2>&1
for f in gpurun_out/r5_bench_c2.json gpurun_out/r5_bench_2x2048.json gpurun_out/r5_bench_4x1024.json; do tail -1 $f | cut -c1-420; echo; done
tail -3 gpurun_out/r5_bench_2x2048.err
