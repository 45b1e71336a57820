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
    glist = defaultdict(list)  # (gap, duration of the kernel before, duration of the kernel after)
    for (s0, e0, n0), (s1, e1, n1) in zip(rows, rows[1:]):
        g = gaps[(n0, n1)]
        g[0] += 1
        g[1] += s1 - e0
        glist[(n0, n1)].append((s1 - e0, e0 - s0, e1 - s1))
    print(f"kernels {len(rows)}  span {span / 1e3:.1f} us  busy {busy / 1e3:.1f} us  idle {(span - busy) / 1e3:.1f} us ({100 * (span - busy) / span:.1f} %)")
    for (n0, n1), (cnt, tot) in sorted(gaps.items(), key=lambda kv: -kv[1][1])[:12]:
        print(f"  {n0:40s} -> {n1:40s} x{cnt:6d}  avg gap {tot / cnt / 1e3:8.2f} us  total {tot / 1e3:10.1f} us")
    print("gap quantiles (us) of the four pairs with the most idle time: p10 p50 p90; mean gap by the half of the "
          "runs whose PREVIOUS kernel was short / long, whose NEXT kernel was short / long")
    for (n0, n1), (cnt, tot) in sorted(gaps.items(), key=lambda kv: -kv[1][1])[:4]:
        gl = glist[(n0, n1)]
        gs = sorted(g for g, _, _ in gl)
        q = lambda f: gs[min(len(gs) - 1, int(f * len(gs)))] / 1e3
        by_prev = sorted(gl, key=lambda t: t[1])
        by_next = sorted(gl, key=lambda t: t[2])
        h = len(gl) // 2 or 1
        mean = lambda xs: sum(x[0] for x in xs) / max(len(xs), 1) / 1e3
        print(f"  {n0:28s} -> {n1:28s} {q(.1):7.2f} {q(.5):7.2f} {q(.9):7.2f} | prev short {mean(by_prev[:h]):6.2f} long {mean(by_prev[h:]):6.2f}"
              f" | next short {mean(by_next[:h]):6.2f} long {mean(by_next[h:]):6.2f}")
    durs = defaultdict(list)
    for s0, e0, n0 in rows:
        durs[n0].append(e0 - s0)
    print("durations (us): kernel, calls, mean, p10, p25, p50, p75, p90")
    for n0, v in sorted(durs.items(), key=lambda kv: -sum(kv[1]))[:10]:
        v.sort()
        q = lambda f: v[min(len(v) - 1, int(f * len(v)))] / 1e3
        print(f"  {n0:40s} {len(v):7d} {sum(v) / len(v) / 1e3:8.2f} {q(.1):8.2f} {q(.25):8.2f} {q(.5):8.2f} {q(.75):8.2f} {q(.9):8.2f}")


if __name__ == "__main__":
    main()
