# round 4: the driver's own command (default flags), timed
mkdir -p gpurun_out
t0=$(date +%s)
timeout -k 10 900 python bench.py > gpurun_out/r04_bench_default_v4.json 2> gpurun_out/r04_bench_default_v4.err
echo "rc=$? wall=$(( $(date +%s) - t0 )) s"
python3 -c "
import json;d=json.load(open('gpurun_out/r04_bench_default_v4.json'))
print('value',d['value'],'whole',d.get('whole_solve'))
print(json.dumps(d['cpu_baseline'],indent=1))"
python3 - <<'PY'
import json
d=json.load(open('gpurun_out/r04_bench_default_v4.json'))
for k in ('late','deep','end'):
    print(k,d[k]['value'],d[k]['roofline']['frac'],d[k]['roofline']['avg_launch_us'],d[k]['kernel_us_per_pivot'])
print('whole',d['whole_solve'], d['whole_solve_remainder']['near_ties'], d['whole_solve_remainder']['refactors'], d['whole_solve_remainder'].get('state_drift'))
print('secondary',d['secondary']['value'],d['secondary']['deep']['value'])
print('roofline',d['roofline']['frac'],d['roofline']['avg_launch_us'],d['roofline'].get('traffic'))
PY
