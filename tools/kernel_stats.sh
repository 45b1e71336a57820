#!/bin/bash
# rocprofv3 --kernel-trace --stats of one bench.py invocation; prints the per-kernel summary.
#   tools/kernel_stats.sh <output name under gpurun_out/> <bench.py arguments...>
# (bench.py detects the profiler and keeps to the primary workload in this one process)
set -e
name=$1; shift
out=$PWD/gpurun_out/$name
mkdir -p "$out"
root=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/raw" -- python3 "$root/bench.py" "$@" > "$out/bench.json" 2> "$out/bench.err" || { tail -5 "$out/bench.err"; exit 1; }
f=$(find "$out/raw" -name '*kernel_stats.csv' | head -1)
cp "$f" "$out/kernel_stats.csv"
rm -rf "$out/raw"
python3 - "$out/kernel_stats.csv" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:16]:
    print(f'{r["Name"][:70]:70s} calls {int(r["Calls"]):7d}  avg {float(r["AverageNs"])/1e3:9.2f} us  {float(r["Percentage"]):6.2f} %')
PY
