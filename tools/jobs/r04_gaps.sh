# round 4: idle time between the kernels of an early pivot (k = 100..), and the early-regime rate
set -x
mkdir -p gpurun_out
root=$PWD
timeout -k 10 200 python bench.py --no-late --no-cpu-baseline --no-pmc-traffic --no-secondary > gpurun_out/r04_nts_c3.json 2>/dev/null
python3 -c "import json;d=json.load(open('gpurun_out/r04_nts_c3.json'));print('config3 value',d['value'],'ms',d['ms_per_step'],'price us',d['roofline']['avg_launch_us'])"
cd /tmp && export TMPDIR=/tmp
for k in 100; do
  out=$root/gpurun_out/r04_gaps_$k
  mkdir -p $out
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $out/raw -- python3 $root/tools/run_pivots.py 3000 8192 16384 1003 0 $k > $out/run.txt 2>&1
  f=$(find $out/raw -name '*kernel_trace.csv' | head -1)
  python3 $root/tools/trace_gaps.py "$f" 0.5 > $out/gaps.txt
  rm -rf $out/raw
  cat $out/gaps.txt
done
