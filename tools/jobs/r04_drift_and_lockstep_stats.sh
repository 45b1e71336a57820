set -x
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "live_lists or rows_alone or row_wise_pricing_alone or state_drift or sparse_four" > gpurun_out/t8.log 2>&1; echo "tests rc=$?"; tail -4 gpurun_out/t8.log
root=$PWD
( cd /tmp && export TMPDIR=/tmp && out=$root/gpurun_out/r04_lockstep_rs_stats && mkdir -p $out && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/raw -- python3 $root/tools/lockstep_profile.py 8 200 32768 65536 1005 0 16384 1 > $out/run.txt 2>&1; f=$(find $out/raw -name '*kernel_stats.csv' | head -1); cp "$f" $out/kernel_stats.csv; rm -rf $out/raw; tail -1 $out/run.txt; python3 - $out/kernel_stats.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:18]:
    print(f'{r["Name"][:64]:64s} calls {int(r["Calls"]):7d}  avg {float(r["AverageNs"])/1e3:9.2f} us  {float(r["Percentage"]):6.2f} %')
PY
)
timeout -k 10 800 python tools/drift_vs_strict.py 12000 2048 4096 2002 4000 > gpurun_out/r04_drift_vs_strict_2048x4096.txt 2>&1; echo "drift rc=$?"; tail -8 gpurun_out/r04_drift_vs_strict_2048x4096.txt
