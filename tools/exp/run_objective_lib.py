"""tools/run_objective.py with another build of the library: python tools/exp/run_objective_lib.py <lib.so> c4|c5 [iters]"""
import os, sys, runpy
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import smm_jl_amd
smm_jl_amd._abi.LIB_PATH = os.path.join(ROOT, "smm.jl_amd", "csrc", sys.argv[1])
sys.argv = ["run_objective.py"] + sys.argv[2:]
runpy.run_path(os.path.join(ROOT, "tools", "run_objective.py"), run_name="__main__")
