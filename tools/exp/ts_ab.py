"""phase stamps (tools/dbg_ts.py) of several library builds on one box: python tools/exp/ts_ab.py libA.so libB.so ..."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    sys.path.insert(0, ROOT)
    import smm_jl_amd
    smm_jl_amd._abi.LIB_PATH = os.path.join(ROOT, "smm.jl_amd", "csrc", sys.argv[2])
    sys.argv = ["dbg_ts.py"]
    import runpy
    os.chdir(ROOT)
    runpy.run_path(os.path.join(ROOT, "tools", "dbg_ts.py"), run_name="__main__")
    sys.exit(0)
for lib in sys.argv[1:]:
    out = subprocess.run([sys.executable, __file__, "--child", lib], capture_output=True, text=True).stdout
    print("====", lib)
    print("\n".join(l for l in out.splitlines() if any(k in l for k in ("staged", "walk", "record", "sim  ", "objective", "stores", "kernel span"))))
