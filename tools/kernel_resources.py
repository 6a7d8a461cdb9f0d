#!/usr/bin/env python3
"""Per-kernel register / LDS / spill report of one .hip source as hipcc compiles it for gfx950 (no GPU needed).
    python tools/kernel_resources.py pair_reproject [extra hipcc flags]"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1]
cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "--offload-arch=gfx950", *sys.argv[2:],
       "-Rpass-analysis=kernel-resource-usage", "-c", src + ".hip", "-o", "/tmp/kr_%d.o" % os.getpid()]
txt = subprocess.run(cmd, cwd=os.path.join(ROOT, "multi-spatialmllm_amd", "csrc"), capture_output=True, text=True).stderr
os.path.exists("/tmp/kr_%d.o" % os.getpid()) and os.remove("/tmp/kr_%d.o" % os.getpid())
for b in re.split(r"remark: [^\n]*Function Name: ", txt)[1:]:
    name = b.split("\n")[0].strip().split(" ")[0]
    name = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip() or name
    name = re.sub(r"^void mspa::", "", name)
    name = re.sub(r"\(unsigned short const\*.*", "", name)

    def g(k):
        m = re.search(k + r": (\d+)", b)
        return m.group(1) if m else "?"
    print("%-100s sgpr %3s vgpr %3s spill s/v %s/%s lds %6s occ %s" % (name[:100], g("SGPRs"), g("VGPRs"), g("SGPRs Spill"), g("VGPRs Spill"),
                                                                     g(r"LDS Size \[bytes/block\]"), g(r"Occupancy \[waves/SIMD\]")))
