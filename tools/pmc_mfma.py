#!/usr/bin/env python3
"""MFMA-pipe utilisation and sustained clock per kernel from one rocprofv3 pass:
    rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d DIR -o X -- python3 bench.py ...
GRBM_GUI_ACTIVE is summed over the 8 XCDs (cycles each XCD was busy); SQ_VALU_MFMA_BUSY_CYCLES over the
1024 SIMDs.  clock = GUI_ACTIVE / 8 / duration;  busy = MFMA_BUSY / (GUI_ACTIVE / 8 * 1024).
usage: pmc_mfma.py <dir> [out.json]"""
import collections
import csv
import glob
import json
import sys

d = sys.argv[1]
cc = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
dur = {}
for r in csv.DictReader(open(cc)):
    key = (r["Kernel_Name"].split("(")[0], r["Dispatch_Id"])
    acc[key][r["Counter_Name"]] += float(r["Counter_Value"])
    if "Start_Timestamp" in r and r["Start_Timestamp"]:
        dur[key] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9
kt = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)
if kt and not dur:
    for r in csv.DictReader(open(kt[0])):
        dur[(r["Kernel_Name"].split("(")[0], r["Dispatch_Id"])] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9
out = collections.defaultdict(lambda: dict(launches=0, seconds=0.0, gui=0.0, mfma=0.0))
for key, c in acc.items():
    if key not in dur or "GRBM_GUI_ACTIVE" not in c:
        continue
    o = out[key[0]]
    o["launches"] += 1
    o["seconds"] += dur[key]
    o["gui"] += c["GRBM_GUI_ACTIVE"]
    o["mfma"] += c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
res = {}
for k, o in sorted(out.items(), key=lambda kv: -kv[1]["seconds"]):
    if o["mfma"] <= 0:
        continue
    cyc = o["gui"] / 8.0
    res[k] = dict(launches=o["launches"], total_ms=1e3 * o["seconds"], clock_ghz=cyc / o["seconds"] / 1e9,
                  mfma_busy=o["mfma"] / (cyc * 1024.0))
    res[k]["busy_x_clock_ghz"] = res[k]["mfma_busy"] * res[k]["clock_ghz"]
    print(f"{k[:64]:64s} n={o['launches']:4d} {res[k]['total_ms']:9.2f} ms  clock {res[k]['clock_ghz']:.2f} GHz  "
          f"MFMA busy {100 * res[k]['mfma_busy']:5.1f} %  busy x clock {res[k]['busy_x_clock_ghz']:.2f}")
if len(sys.argv) > 2:
    json.dump(res, open(sys.argv[2], "w"), indent=1)
