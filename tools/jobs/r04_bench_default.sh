# round 4: the driver's own command (default flags), timed
mkdir -p gpurun_out
t0=$(date +%s)
timeout -k 10 900 python bench.py > gpurun_out/r04_bench_default_v3.json 2> gpurun_out/r04_bench_default_v3.err
echo "rc=$? wall=$(( $(date +%s) - t0 )) s"
python3 -c "
import json;d=json.load(open('gpurun_out/r04_bench_default_v3.json'))
print('value',d['value'],'whole',d.get('whole_solve'))
print(json.dumps(d['cpu_baseline'],indent=1))"
