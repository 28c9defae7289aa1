#!/usr/bin/env python3
"""Registers, spills, scratch and LDS of every kernel: parses `hipcc ... -Rpass-analysis=kernel-resource-usage` remarks.
  python3 tools/resource_usage.py [extra hipcc flags...]      (compiles pathtracer_amd/csrc/pt_kernels.hip for the device only)"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "pathtracer_amd", "csrc")
cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-I" + os.path.join(ROOT, "include"), "-I" + src,
       "-fno-slp-vectorize", "--offload-device-only", "-Rpass-analysis=kernel-resource-usage", "-c", os.path.join(src, "pt_kernels.hip"), "-o", "/tmp/pt_resource_usage.o"] + sys.argv[1:]
text = subprocess.run(cmd, capture_output=True, text=True).stderr
SCRATCH, OCC, LDS = r"ScratchSize \[bytes/lane\]", r"Occupancy \[waves/SIMD\]", r"LDS Size \[bytes/block\]"
for block in text.split("Function Name: ")[1:]:
    name = block.split()[0]
    demangled = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    short = re.sub(r"\(.*", "", demangled.replace("(anonymous namespace)::", "").replace("void ", ""))
    g = lambda key: (re.search(key + r": (\d+)", block) or [0, "?"])[1]
    print("%-40s VGPR %3s AGPR %3s SGPR %3s  spilled V %3s S %3s  scratch %4s B/lane  waves/SIMD %s  LDS %6s" % (
        short[:40], g("VGPRs"), g("AGPRs"), g("SGPRs"), g("VGPRs Spill"), g("SGPRs Spill"), g(SCRATCH), g(OCC), g(LDS)))
