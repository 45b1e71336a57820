# round 4: the whole GPU suite, then the default bench line
set -x
mkdir -p gpurun_out
timeout -k 10 700 python -m pytest tests -m gpu -q > gpurun_out/t6_full.log 2>&1; echo "suite rc=$?"; tail -4 gpurun_out/t6_full.log
timeout -k 10 400 python bench.py > gpurun_out/r04_bench_default_v2.json 2> gpurun_out/r04_bench_default_v2.err; echo "bench rc=$?"
python3 - <<'PY'
import json
d=json.load(open('gpurun_out/r04_bench_default_v2.json'))
print('value',d['value'])
for k in ('late','deep','end'):
    print(k,d[k]['value'],d[k]['kernel_us_per_pivot'])
print('whole',d['whole_solve'], d['whole_solve_remainder']['near_ties'], d['whole_solve_remainder']['refactors'], d['whole_solve_remainder'].get('state_drift'))
print('secondary',d['secondary']['value'],d['secondary']['deep']['value'])
PY
