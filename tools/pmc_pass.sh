#!/bin/bash
# One rocprofv3 --pmc pass (counters only: no --kernel-trace / --stats with it) over a bench.py
# invocation; prints per-kernel averages of the requested counters (summed over XCDs per dispatch).
#   tools/pmc_pass.sh <name under gpurun_out/> "<COUNTER1 COUNTER2 ...>" <kernel name substring> <bench.py arguments...>
#   tools/pmc_pass.sh <name> "<COUNTERS>" <kernel substring> tools/some_tool.py <its arguments...>   (any python tool)
set -e
name=$1; counters=$2; kernel=$3; shift 3
out=$PWD/gpurun_out/$name
mkdir -p "$out"
root=$PWD
prog=$root/bench.py
case "$1" in *.py) prog=$root/$1; shift;; esac
cd /tmp && export TMPDIR=/tmp
echo "pmc pass $name: $counters" && timeout -k 10 150 rocprofv3 --pmc $counters --output-format csv -d "$out/raw" -- python3 "$prog" "$@" > "$out/bench.json" 2> "$out/bench.err" || { tail -5 "$out/bench.err"; exit 1; }
python3 - "$out/raw" "$kernel" <<'PY' | tee "$out/summary.txt"
import csv, glob, os, sys
tot, cnt = {}, {}
for path in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(path)):
        if sys.argv[2] in row["Kernel_Name"]:
            key = (row["Kernel_Name"].split("(")[0], row["Counter_Name"])
            tot[key] = tot.get(key, 0.0) + float(row["Counter_Value"])
            cnt[key] = cnt.get(key, 0) + 1
for key in sorted(tot):
    print(f"{key[0]:40s} {key[1]:32s} avg per dispatch {tot[key] / cnt[key]:16.1f}   ({cnt[key]} dispatches)")
PY
rm -rf "$out/raw"
