"""Per-kernel time of the pivots between two profiled runs of tools/run_pivots.py:
  python3 tools/regime_stats.py gpurun_out/<A>/kernel_stats.csv gpurun_out/<B>/kernel_stats.csv <pivots B - A>
"""
import csv
import sys

a = {r["Name"]: r for r in csv.DictReader(open(sys.argv[1]))}
b = {r["Name"]: r for r in csv.DictReader(open(sys.argv[2]))}
d = int(sys.argv[3])
rows = []
for name, rb in b.items():
    ra = a.get(name, {"Calls": "0", "TotalDurationNs": "0"})
    calls = int(rb["Calls"]) - int(ra["Calls"])
    ns = float(rb["TotalDurationNs"]) - float(ra["TotalDurationNs"])
    if calls > 0:
        rows.append((ns, name, calls))
rows.sort(reverse=True)
total = sum(r[0] for r in rows)
print(f"{d} pivots: {total / d / 1e3:.2f} us of kernel time per pivot")
for ns, name, calls in rows[:14]:
    print(f"{name[:60]:60s} calls {calls:6d}  avg {ns / calls / 1e3:9.2f} us  per pivot {ns / d / 1e3:8.2f} us")
