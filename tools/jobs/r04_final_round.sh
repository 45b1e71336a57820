# round 4: the whole GPU suite as the driver runs it, smoke(), the default bench line, kernel stats of the driver-style command
mkdir -p gpurun_out
t0=$(date +%s)
timeout -k 10 880 python -m pytest tests -m gpu -x -q --durations=12 > gpurun_out/t17_full.log 2>&1; echo "suite rc=$? wall=$(( $(date +%s) - t0 )) s"; tail -22 gpurun_out/t17_full.log
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
t0=$(date +%s)
timeout -k 10 900 python bench.py > gpurun_out/r04_bench_default_v8.json 2> gpurun_out/r04_bench_default_v8.err; echo "bench rc=$? wall=$(( $(date +%s) - t0 )) s"
python3 - <<'PY'
import json
d=json.load(open('gpurun_out/r04_bench_default_v8.json'))
print('value',d['value'],d['ms_per_step'],d['roofline']['kernel'],d['roofline']['frac'],d['roofline']['avg_launch_us'],d['roofline'].get('traffic'),d['roofline']['algorithmic_bytes_per_launch'])
for k in ('late','deep','end'):
    print(k,d[k]['value'],d[k]['roofline']['frac'],d[k]['roofline']['avg_launch_us'],d[k]['kernel_us_per_pivot'])
print('whole',d['whole_solve'], d['whole_solve_remainder']['near_ties'], d['whole_solve_remainder']['refactors'], d['whole_solve_remainder'].get('state_drift'), d['whole_solve_remainder']['objective'])
print('secondary',d['secondary']['value'],d['secondary']['deep']['value'])
print('cpu',d['cpu_baseline']['value'],d['cpu_baseline']['at_benchmark_size'])
print('mfma',d['mfma']['seconds'],d['mfma']['full_basis']['seconds'],d['mfma']['MfmaUtil'])
PY
bash tools/kernel_stats.sh r04_final_kernel_stats --steps 1500 --warmup 100 > gpurun_out/r04_final_kernel_stats.txt 2>&1; echo "kernel stats rc=$?"; head -6 gpurun_out/r04_final_kernel_stats.txt
timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --force-sharded > gpurun_out/r04_sharded_1rank.json 2> gpurun_out/r04_sharded_1rank.err; echo "sharded rc=$?"
python3 -c "
import json;d=json.load(open('gpurun_out/r04_sharded_1rank.json'));print('sharded value',d['value'],'deep',d.get('deep',{}).get('value'),'cpu',(d.get('cpu_baseline') or {}).get('value'))"
