#!/bin/bash
# rocprofv3 --kernel-trace --stats of an arbitrary python tool; prints the per-kernel summary.
#   tools/kernel_stats_cmd.sh <output name under gpurun_out/> <script.py> <arguments...>
set -e
name=$1; shift
out=$PWD/gpurun_out/$name
mkdir -p "$out"
root=$PWD
script=$root/$1; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/raw" -- python3 "$script" "$@" > "$out/stdout.txt" 2> "$out/stderr.txt" || { tail -5 "$out/stderr.txt"; exit 1; }
f=$(find "$out/raw" -name '*kernel_stats.csv' | head -1)
cp "$f" "$out/kernel_stats.csv"
rm -rf "$out/raw"
python3 - "$out/kernel_stats.csv" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:18]:
    print(f'{r["Name"][:64]:64s} calls {int(r["Calls"]):7d}  avg {float(r["AverageNs"])/1e3:10.2f} us  total {float(r["TotalDurationNs"])/1e6:9.2f} ms  {float(r["Percentage"]):6.2f} %')
PY
