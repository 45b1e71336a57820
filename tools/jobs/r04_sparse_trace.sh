# round 4: kernel durations (quantiles: primal and dual steps differ) and gaps of the sparse-basis iteration, config 4 at k ~ 2 200
# usage: r04_sparse_trace.sh [grid of k_price_csc_rl ...]
mkdir -p gpurun_out
root=$PWD
cd /tmp && export TMPDIR=/tmp
for g in ${@:-2048}; do
out=$root/gpurun_out/r04_sparse_trace_g$g
mkdir -p $out
DZG_RL_GRID=$g timeout -k 10 400 python3 $root/bench.py --rows 50000 --cols 100000 --seed 1004 --sparse-per-col 50 --warmup 20000 --steps 3000 --no-late --no-cpu-baseline --no-pmc-traffic --no-secondary > $out/plain.json 2> $out/run.err
python3 -c "import json;d=json.load(open('$out/plain.json'));print('grid $g without the profiler: value',d['value'])"
DZG_RL_GRID=$g timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $out/raw -- python3 $root/bench.py --rows 50000 --cols 100000 --seed 1004 --sparse-per-col 50 --warmup 20000 --steps 3000 --no-late --no-cpu-baseline --no-pmc-traffic --no-secondary > $out/run.json 2> $out/run.err
f=$(find $out/raw -name '*kernel_trace.csv' | head -1)
python3 $root/tools/trace_gaps.py "$f" 0.87 > $out/gaps.txt
rm -rf $out/raw
grep -v copyBuffer $out/gaps.txt | head -24
python3 -c "import json;d=json.load(open('$out/run.json'));print('grid $g under the profiler: value',d['value'],d['config'].get('k_at_start'),d['config'].get('k_at_end'))"
done
