#!/usr/bin/env python3
"""Register / scratch / LDS use of the dispersion kernels, one line per instantiation (hipcc -Rpass-analysis=kernel-resource-usage).
    python tools/kernel_resources.py [file.hip ...]   (default: swd_group_kernel.hip swd_kernel.hip)"""
import os, re, subprocess, sys
CS = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bayhunter_amd", "csrc")
FLAGS = "--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-pass-failed".split()
EXTRA = {"swd_group_kernel.hip": ["-mllvm", "-disable-machine-licm"], "rf_kernel.hip": ["-ffp-contract=fast", "-mllvm", "-disable-machine-licm"]}
for f in (sys.argv[1:] or ["swd_group_kernel.hip", "swd_kernel.hip"]):
    out = subprocess.run(["/opt/rocm/bin/hipcc"] + FLAGS + EXTRA.get(f, []) + ["-Rpass-analysis=kernel-resource-usage", "-c", f, "-o", "/dev/null"],
                         cwd=CS, capture_output=True, text=True).stderr
    cur = {}
    for line in out.splitlines():
        m = re.search(r"remark: (?:\[[^\]]*\] )?\s*(Function Name|VGPRs|AGPRs|SGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (\S+)", line)
        if not m:
            continue
        k, v = m.group(1), m.group(2)
        if k == "Function Name":
            cur = {"name": subprocess.run(["c++filt", v], capture_output=True, text=True).stdout.strip().replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]}
        else:
            cur[k.split(" ")[0]] = v
            if k.startswith("LDS"):
                print("%-62s VGPR %3s SGPR %3s scratch %4s occ %s" % (cur["name"].replace(" ", "")[:62], cur.get("VGPRs"), cur.get("SGPRs"), cur.get("ScratchSize"), cur.get("Occupancy")))
