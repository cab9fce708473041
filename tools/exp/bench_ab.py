"""A/B of library builds on one box: python tools/exp/bench_ab.py <rounds> libA.so libB.so ...  (names relative to smm.jl_amd/csrc).
Each run is a fresh process of bench.py with the loader's path redirected (an experiment hook that lives here, not in the product)."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    sys.path.insert(0, ROOT)
    import smm_jl_amd
    smm_jl_amd._abi.LIB_PATH = os.path.join(ROOT, "smm.jl_amd", "csrc", sys.argv[2])
    sys.argv = ["bench.py", "--steps", "5", "--warmup", "1", "--no-cpu-baseline", "--no-unfused"] + os.environ.get("BENCH_AB_ARGS", "").split()
    import runpy
    runpy.run_path(os.path.join(ROOT, "bench.py"), run_name="__main__")
    sys.exit(0)
rounds = int(sys.argv[1])
for _ in range(rounds):
    for lib in sys.argv[2:]:
        out = subprocess.run([sys.executable, __file__, "--child", lib], capture_output=True, text=True).stdout.strip().splitlines()
        try:
            d = json.loads(out[-1])
            print("%-22s %6.1f M/s   kernel %6.2f us" % (lib, d["value"] / 1e6, d["roofline"]["avg_kernel_us"]), flush=True)
        except Exception as e:
            print(lib, "failed", e, out[-3:])
