#!/usr/bin/env python3
"""HBM traffic per launch from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE), per kernel.

MI355X_MICROARCH.md §HBM: the counters are in KiB... (x1024); on gfx950 FETCH_SIZE reports exactly
half of the bytes of a wide coalesced streaming read, so the read side is doubled; WRITE_SIZE
is exact for 16-B-per-lane stores.  usage: pmc_traffic.py <fetch_dir> <write_dir> [out.json]"""
import collections
import csv
import glob
import json
import sys


def per_kernel(d, counter):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    tot, n = collections.Counter(), collections.Counter()
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == counter:
            k = r["Kernel_Name"].split("(")[0]
            tot[k] += float(r["Counter_Value"])
            n[k] += 1
    return {k: (tot[k] / n[k], n[k]) for k in tot}


fetch = per_kernel(sys.argv[1], "FETCH_SIZE")
write = per_kernel(sys.argv[2], "WRITE_SIZE")
out = {}
for k in sorted(fetch, key=lambda k: -fetch[k][0] * fetch[k][1]):
    fr, n = fetch[k]
    wr = write.get(k, (0.0, 0))[0]
    out[k] = dict(launches=n, fetch_size_kib=fr, write_size_kib=wr,
                  hbm_bytes_per_launch=(2.0 * fr + wr) * 1024.0)
for k, v in list(out.items())[:14]:
    print(f"{k[:60]:60s} n={v['launches']:5d} FETCH={v['fetch_size_kib']/1024:9.1f} MiB  WRITE={v['write_size_kib']/1024:9.1f} MiB "
          f"-> {v['hbm_bytes_per_launch']/1e6:9.1f} MB/launch")
if len(sys.argv) > 3:
    json.dump(out, open(sys.argv[3], "w"), indent=1)
