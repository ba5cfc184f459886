"""Idle time between kernels of the bench step, from a rocprofv3 kernel trace:
    cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --output-format csv -d <dir> -o kt -- python3 bench.py --steps 3 --warmup 2 \
        --no-cpu-baseline --reference-faithful off
    python tools/lab/launch_gaps.py <dir>
Takes the last 40 % of the trace (the timed steps), merges overlapping dispatches and reports span, busy time and the gap histogram."""
import csv
import glob
import sys

path = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(path)))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
ev = ev[int(len(ev) * 0.6):]
span = ev[-1][1] - ev[0][0]
busy, gaps, cur_end = 0, [], ev[0][0]
for s, e, _ in ev:
    if s > cur_end:
        gaps.append(s - cur_end)
        busy += e - s
    else:
        busy += max(0, e - cur_end)
    cur_end = max(cur_end, e)
gaps.sort()
n = len(ev)
print(f"{n} dispatches, span {span / 1e6:.2f} ms, busy {busy / 1e6:.2f} ms, idle {100 * (span - busy) / span:.1f} % "
      f"({(span - busy) / 1e3 / n:.2f} us per dispatch)")
if gaps:
    for q in (0.5, 0.9, 0.99):
        print(f"  gap p{int(q * 100)}: {gaps[int(q * (len(gaps) - 1))] / 1e3:.2f} us")
    print(f"  gaps > 20 us: {sum(g > 20000 for g in gaps)} totalling {sum(g for g in gaps if g > 20000) / 1e6:.3f} ms")
