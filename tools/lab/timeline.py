"""Timeline of a two-stream bench run from a rocprofv3 --kernel-trace CSV: per queue busy time, the time both queues have a
kernel in flight, idle time — over the dispatches of the timed steps.   usage: python tools/lab/timeline.py <kernel_trace.csv>"""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
ks = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"], r["Kernel_Name"]) for r in rows]
ks.sort()
# the timed region: between the first and the last modconv_xp launch of the run's second half
t_lo, t_hi = ks[len(ks) // 3][0], ks[-len(ks) // 6][1]
ks = [k for k in ks if k[0] >= t_lo and k[1] <= t_hi]
span = (t_hi - t_lo) / 1e6
byq = defaultdict(list)
for s, e, q, n in ks:
    byq[q].append((s, e))


def union(iv):
    iv = sorted(iv)
    tot, cs, ce = 0, None, None
    for s, e in iv:
        if cs is None:
            cs, ce = s, e
        elif s <= ce:
            ce = max(ce, e)
        else:
            tot += ce - cs
            cs, ce = s, e
    return tot + (ce - cs if cs is not None else 0)


allbusy = union([(s, e) for s, e, _, _ in ks]) / 1e6
print(f"window {span:.2f} ms, {len(ks)} dispatches; some kernel in flight {allbusy:.2f} ms ({100 * allbusy / span:.1f} %)")
tot = 0.0
for q, iv in sorted(byq.items(), key=lambda kv: -len(kv[1])):
    b = union(iv) / 1e6
    tot += b
    print(f"  queue {q}: {len(iv)} dispatches, busy {b:.2f} ms ({100 * b / span:.1f} %)")
print(f"  sum of queue-busy {tot:.2f} ms -> two queues in flight together for >= {tot - allbusy:.2f} ms ({100 * (tot - allbusy) / span:.1f} %)")
dur = defaultdict(float)
for s, e, q, n in ks:
    dur[n.split("(")[0][:60]] += (e - s) / 1e6
for n, d in sorted(dur.items(), key=lambda kv: -kv[1])[:12]:
    print(f"    {d:8.2f} ms  {n}")
