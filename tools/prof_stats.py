#!/usr/bin/env python3
"""Print the top rows of a rocprofv3 `*_kernel_stats.csv`."""
import csv
import glob
import sys

path = sys.argv[1]
files = glob.glob(path + "/**/*kernel_stats.csv", recursive=True) if not path.endswith(".csv") else [path]
rows = list(csv.DictReader(open(files[0])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
for r in rows[:n]:
    print(f"{r['Name'][:84]:84s} calls={r['Calls']:>6s} total_ms={float(r['TotalDurationNs']) / 1e6:9.2f} "
          f"avg_us={float(r['AverageNs']) / 1e3:10.1f} pct={float(r['Percentage']):6.2f}")
