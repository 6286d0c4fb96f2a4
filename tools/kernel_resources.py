#!/usr/bin/env python3
"""Print VGPR / SGPR / scratch / LDS / occupancy of every kernel in dronechase_amd/csrc/te_env.hip
(hipcc -Rpass-analysis=kernel-resource-usage), one line per kernel."""
import re
import subprocess
import sys

src = sys.argv[1] if len(sys.argv) > 1 else "dronechase_amd/csrc/te_env.hip"
out = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-gpu-rdc", "-c", src,
                      "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage", *sys.argv[2:]],
                     capture_output=True, text=True).stderr
cur = {}
for line in out.splitlines():
    m = re.search(r"remark: [^:]+:\d+:\d+:\s+(.*?) \[-Rpass", line) or re.search(r"remark:\s+(.*?) \[-Rpass", line)
    if not m:
        continue
    t = m.group(1).strip()
    if t.startswith("Function Name:"):
        if cur:
            print(cur)
        name = t.split(":", 1)[1].strip()
        cur = {"kernel": subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()[:70]}
    else:
        k, _, v = t.partition(":")
        if k.strip() in ("VGPRs", "AGPRs", "TotalSGPRs", "ScratchSize [bytes/lane]", "Occupancy [waves/SIMD]", "LDS Size [bytes/block]"):
            cur[k.strip().split(" ")[0]] = v.strip()
if cur:
    print(cur)
