#!/usr/bin/env python3
"""GPU idle gaps in a rocprofv3 --kernel-trace CSV: python tools/gap_report.py trace.csv [min_gap_us] [skip_first_n_kernels]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
thr = float(sys.argv[2]) if len(sys.argv) > 2 else 20.0
skip = int(sys.argv[3]) if len(sys.argv) > 3 else 0
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:60]) for r in rows))[skip:]
busy_end = ev[0][1]
t_begin = ev[0][0]
gaps, idle = [], 0
for s, e, n in ev[1:]:
    if s > busy_end:
        g = (s - busy_end) / 1e3
        idle += s - busy_end
        if g >= thr:
            gaps.append((g, n))
    busy_end = max(busy_end, e)
total = (busy_end - t_begin) / 1e3
print(f"span {total / 1e3:.2f} ms, idle {idle / 1e6:.2f} ms ({100 * idle / 1e3 / total:.1f} %), gaps >= {thr} us: {len(gaps)}")
import collections
agg = collections.OrderedDict()
for g, n in gaps:
    a = agg.setdefault(n, [0, 0.0]); a[0] += 1; a[1] += g
for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:12]:
    print(f"  {t / 1e3:7.2f} ms in {c:4d} gaps (avg {t / c:7.1f} us) before {n}")
