#!/usr/bin/env python3
"""Average of every collected PMC counter per kernel from a rocprofv3 --pmc run directory.
usage: pmc_summary.py <dir> [kernel-substring]"""
import collections
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
want = sys.argv[2] if len(sys.argv) > 2 else ""
tot, n = collections.defaultdict(collections.Counter), collections.defaultdict(collections.Counter)
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].split("(")[0]
    if want in k:
        tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
        n[k][r["Counter_Name"]] += 1
for k in tot:
    print(k[:100])
    for c in sorted(tot[k]):
        print(f"   {c:32s} {tot[k][c] / n[k][c]:16.1f}  (n={n[k][c]})")
