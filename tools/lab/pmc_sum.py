"""sum rocprofv3 counter_collection CSVs per kernel: pmc_sum.py <dir> <kernel substring>"""
import collections, csv, glob, sys
d, pat = sys.argv[1], sys.argv[2]
tot = collections.defaultdict(float); n = collections.defaultdict(set)
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if pat in r["Kernel_Name"]:
            tot[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]].add(r["Dispatch_Id"])
for k in sorted(tot):
    print(f"{k:36s} {tot[k] / max(1, len(n[k])):18.0f} per launch ({len(n[k])} launches)")
