# round 4: per-kernel durations of the LAST 1 500 pivots of `bench.py --warmup 20000 --steps 1500` (the deep block's regime, k = 4 049..4 080,
# in the real solve) from a rocprofv3 --kernel-trace of that command
mkdir -p gpurun_out
root=$PWD
cd /tmp && export TMPDIR=/tmp
out=$root/gpurun_out/r04_late_trace
mkdir -p $out
timeout -k 10 500 rocprofv3 --kernel-trace --output-format csv -d $out/raw -- python3 $root/bench.py --warmup 20000 --steps 1500 > $out/bench.json 2> $out/bench.err
f=$(find $out/raw -name '*kernel_trace.csv' | head -1)
python3 $root/tools/trace_gaps.py "$f" 0.931 > $out/tail.txt
rm -rf $out/raw
cat $out/tail.txt | head -30
python3 -c "import json;d=json.load(open('$out/bench.json'));print('value',d['value'],d['config']['k_at_start'],d['config']['k_at_end'],d['roofline']['avg_launch_us'],d['roofline']['frac'],d['roofline']['algorithmic_bytes_per_launch'])"
