#!/usr/bin/env python3
"""Registers / spills / LDS of every kernel of one translation unit (hipcc -Rpass-analysis=kernel-resource-usage), one line
per kernel.   usage: python tools/lab/resusage.py gx_swav.hip [name filter]"""
import os
import re
import subprocess
import sys

root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
src = os.path.join(root, "ganecdotes_amd", "csrc", sys.argv[1])
flt = sys.argv[2] if len(sys.argv) > 2 else ""
cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-fno-slp-vectorize", *os.environ.get("GX_EXTRA_FLAGS", "").split(),
       "-I", os.path.dirname(src), "-I", os.path.join(root, "include"), "-c", src, "-o", "/dev/null", "--cuda-device-only",
       "-Rpass-analysis=kernel-resource-usage"]
out = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = {}
rows = []
for line in out.splitlines():
    m = re.search(r"remark:\s+(Function Name|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|VGPRs Spill|SGPRs Spill|LDS Size \[bytes/block\]|Occupancy \[waves/SIMD\]): (\S+)", line)
    if "error" in line:
        print(line)
    if not m:
        continue
    k, v = m.groups()
    if k == "Function Name":
        cur = {"name": subprocess.run(["c++filt", v], capture_output=True, text=True).stdout.strip().split("(")[0]}
        rows.append(cur)
    else:
        cur[k] = v
for r in rows:
    if flt in r["name"]:
        print(f"{r['name'][:70]:70s} vgpr {r.get('VGPRs', '?'):>4s} agpr {r.get('AGPRs', '?'):>3s} spill {r.get('VGPRs Spill', '?'):>3s} "
              f"scratch {r.get('ScratchSize [bytes/lane]', '?'):>4s} lds {r.get('LDS Size [bytes/block]', '?'):>6s} occ {r.get('Occupancy [waves/SIMD]', '?')}")
