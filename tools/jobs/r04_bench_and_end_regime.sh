# round 4: new tests, the default bench line, and the kernel picture of the regime that owns the solve
set -x
mkdir -p gpurun_out
timeout -k 10 400 python -m pytest tests/test_gpu_parity.py tests/test_sharded.py -m gpu -x -q -k "state_drift or processes_on_one_gpu or price_pass or row_sharded_whole" > gpurun_out/t4.log 2>&1; echo "tests rc=$?"; tail -5 gpurun_out/t4.log
timeout -k 10 560 python bench.py > gpurun_out/r04_bench_default.json 2> gpurun_out/r04_bench_default.err; echo "bench rc=$?"; tail -c 600 gpurun_out/r04_bench_default.err
root=$PWD
cd /tmp && export TMPDIR=/tmp
for k in 7700 6000; do
  out=$root/gpurun_out/r04_end_k$k
  mkdir -p $out
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/raw -- python3 $root/tools/run_pivots.py 2000 8192 16384 1003 0 $k > $out/run.txt 2>&1
  f=$(find $out/raw -name '*kernel_stats.csv' | head -1); cp "$f" $out/kernel_stats.csv; rm -rf $out/raw
  cat $out/run.txt | tail -2
  python3 - $out/kernel_stats.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:12]:
    print(f'{r["Name"][:70]:70s} calls {int(r["Calls"]):7d}  avg {float(r["AverageNs"])/1e3:9.2f} us  {float(r["Percentage"]):6.2f} %')
PY
done
