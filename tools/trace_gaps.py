#!/usr/bin/env python3
"""Idle time between consecutive kernels of a rocprofv3 --kernel-trace CSV.

usage: trace_gaps.py kernel_trace.csv [skip_first_fraction]
Prints, over the kernels after the skipped head: span, busy, idle, and the idle time by (previous kernel -> next kernel) pair.
Diagnostic tool, not product code."""
import csv
import sys
from collections import defaultdict


def short(name):
    name = name.replace("void ", "")
    return name.split("(")[0][:40]


def main():
    rows = []
    with open(sys.argv[1]) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])))
    rows.sort()
    skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
    rows = rows[int(len(rows) * skip):]
    span = rows[-1][1] - rows[0][0]
    busy = sum(e - s for s, e, _ in rows)
    gaps = defaultdict(lambda: [0, 0])
    for (s0, e0, n0), (s1, e1, n1) in zip(rows, rows[1:]):
        g = gaps[(n0, n1)]
        g[0] += 1
        g[1] += s1 - e0
    print(f"kernels {len(rows)}  span {span / 1e3:.1f} us  busy {busy / 1e3:.1f} us  idle {(span - busy) / 1e3:.1f} us ({100 * (span - busy) / span:.1f} %)")
    for (n0, n1), (cnt, tot) in sorted(gaps.items(), key=lambda kv: -kv[1][1])[:12]:
        print(f"  {n0:40s} -> {n1:40s} x{cnt:6d}  avg gap {tot / cnt / 1e3:8.2f} us  total {tot / 1e3:10.1f} us")


if __name__ == "__main__":
    main()
