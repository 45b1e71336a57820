mkdir -p gpurun_out
t0=$(date +%s)
timeout -k 10 900 python bench.py --no-cpu-baseline --no-secondary --no-mfma --no-whole-solve > gpurun_out/r04_regime_traffic.json 2> gpurun_out/r04_regime_traffic.err; echo "rc=$? wall=$(( $(date +%s) - t0 )) s"; tail -3 gpurun_out/r04_regime_traffic.err
python3 - <<'PY'
import json
d=json.load(open('gpurun_out/r04_regime_traffic.json'))
print('timed', d['roofline']['traffic'], d['roofline']['algorithmic_bytes_per_launch'])
for k in ('deep','end'):
    r=d[k]['roofline']; print(k, r.get('traffic'), r['algorithmic_bytes_per_launch'], (r.get('traffic_detail') or {}).get('algorithmic_bytes_per_launch_same_pivots'), (r.get('traffic_detail') or {}).get('write_bytes'))
PY
