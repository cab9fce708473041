"""register / spill / LDS usage of the kernels of smmhip.hip as the compiler reports it:
python tools/kernel_resources.py [pattern] [src_dir]   (cross-compiles for gfx950, no GPU needed)"""
import os, re, subprocess, sys
pat = sys.argv[1] if len(sys.argv) > 1 else "k_chain_iter_norm"
src = sys.argv[2] if len(sys.argv) > 2 else os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "smm.jl_amd", "csrc")
out = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
                      "--cuda-device-only", "-c", "smmhip.hip", "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"],
                     cwd=src, capture_output=True, text=True).stderr
cur, rows = None, {}
for l in out.splitlines():
    m = re.search(r"remark: Function Name: (\S+)", l)
    if m:
        cur = m.group(1); rows[cur] = {}
        continue
    m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\S+) \[-Rpass", l)
    if m and cur:
        rows[cur][m.group(1)] = m.group(2)
for n, r in rows.items():
    d = subprocess.run(["c++filt", n], capture_output=True, text=True).stdout.strip()
    m = re.search(r"(k_[a-z_0-9]+(?:<[^>]*>)?)", d)
    name = m.group(1) if m else d
    if pat in name:
        print("%-40s" % name[:40], " ".join("%s=%s" % (k.replace(" ", ""), v) for k, v in r.items() if k not in ("Dynamic Stack", "AGPRs")))
