#!/usr/bin/env python3
"""Idle time BETWEEN kernels in a rocprofv3 kernel trace: trace_gaps.py kernel_trace.csv [last_fraction]
(sorted by start time, the last `last_fraction` of the dispatches = the timed steps of bench.py).  Prints the span, the summed
kernel time, the summed gaps (next start - previous end, overlaps count as 0), the gap histogram and the kernels in front of
the largest gaps."""
import csv, re, sys
from collections import Counter
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r['Start_Timestamp']))
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
rows = rows[int(len(rows) * (1 - frac)):]
def short(n):
    m = re.search(r'(conv3d_\w+<[^>]*>|\w+_kernel(<[^>]*>)?|at::native::\w+)', n)
    return (m.group(1) if m else n)[:60]
t0, t1 = int(rows[0]['Start_Timestamp']), max(int(r['End_Timestamp']) for r in rows)
busy = sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in rows)
gaps, end = [], int(rows[0]['End_Timestamp'])
for prev, r in zip(rows, rows[1:]):
    g = int(r['Start_Timestamp']) - end
    gaps.append((max(g, 0), short(prev['Kernel_Name']), short(r['Kernel_Name'])))
    end = max(end, int(r['End_Timestamp']))
tot = sum(g for g, _, _ in gaps)
print(f"{len(rows)} dispatches, span {(t1 - t0) / 1e6:.2f} ms, kernel time {busy / 1e6:.2f} ms, gaps {tot / 1e6:.3f} ms "
      f"({100.0 * tot / (t1 - t0):.1f} % of the span), mean gap {tot / len(gaps) / 1e3:.2f} us")
h = Counter(min(int(g / 1000), 20) for g, _, _ in gaps)
print("gap histogram (us: count):", " ".join(f"{k}{'+' if k == 20 else ''}:{v}" for k, v in sorted(h.items())))
by = Counter()
for g, a, b in gaps:
    by[(a, b)] += g
for (a, b), g in by.most_common(8):
    print(f"  {g / 1e3:9.1f} us total between  {a}  ->  {b}")
