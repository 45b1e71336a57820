# round 4: per-kernel averages of config 4's sparse-basis path at the end of its solve (k = 43 828, resumed from the carried state)
mkdir -p gpurun_out
root=$PWD
cd /tmp && export TMPDIR=/tmp
out=$root/gpurun_out/r04_c4_end
mkdir -p $out
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $out/raw -- python3 $root/tools/full_solve_sparse.py 50000 100000 50 1004 25 20000 --state $root/carry/config4_state_r04.npz --state-out $out/state_unused.npz > $out/run.txt 2>&1
f=$(find $out/raw -name '*kernel_stats.csv' | head -1); cp "$f" $out/kernel_stats.csv; rm -rf $out/raw $out/state_unused.npz
grep -v rocprof $out/run.txt | tail -4
python3 - $out/kernel_stats.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:14]:
    print(f'{r["Name"][:60]:60s} calls {int(r["Calls"]):7d}  avg {float(r["AverageNs"])/1e3:10.2f} us  {float(r["Percentage"]):6.2f} %')
PY
