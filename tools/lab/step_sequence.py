"""The kernel sequence of the last bench step of a rocprofv3 kernel trace (see launch_gaps.py): name, duration, gap before it."""
import csv
import glob
import re
import sys

path = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(path)))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
# the last step starts at the last rotflip / view_latents launch group: take the last third
ev = ev[-int(len(ev) / 5) - 5:]
prev = ev[0][0]
for s, e, n in ev:
    n = re.sub(r"^void ", "", n)
    n = re.sub(r"\(.*", "", n)[:60]
    print(f"{(s - prev) / 1e3:8.1f} us gap  {(e - s) / 1e3:9.1f} us  {n}")
    prev = e
