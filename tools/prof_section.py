#!/usr/bin/env python3
"""Per-kernel totals of the tail of a rocprofv3 kernel trace: prof_section.py trace.csv START_SUBSTRING [fraction]
(rows from the first kernel whose name contains START_SUBSTRING; `fraction` keeps the last part of that section)."""
import csv, re, sys
from collections import defaultdict
rows = list(csv.DictReader(open(sys.argv[1])))
def short(n):
    m = re.search(r'(conv3d_\w+<[^>]*>|\w+_kernel(<[^>]*>)?|at::native::\w+)', n)
    return m.group(1)[:80] if m else n[:60]
names = [short(r['Kernel_Name']) for r in rows]
start = min(i for i, n in enumerate(names) if sys.argv[2] in n)
frac = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
sect = list(zip(rows, names))[start:]
sect = sect[int(len(sect) * (1 - frac)):]
agg = defaultdict(lambda: [0, 0.0])
t0 = min(int(r['Start_Timestamp']) for r, _ in sect); t1 = max(int(r['End_Timestamp']) for r, _ in sect)
for r, n in sect:
    agg[n][0] += 1; agg[n][1] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
tot = sum(v[1] for v in agg.values())
print(f"span {(t1 - t0) / 1e6:.1f} ms, kernel time {tot / 1e3:.1f} ms")
for n, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:16]:
    print(f"{v[1] / 1e3:8.2f} ms {v[1] / tot * 100:5.1f}%  x{v[0]:5d}  avg {v[1] / v[0]:8.1f} us  {n}")
