set -x
mkdir -p gpurun_out
timeout -k 10 500 python -m pytest tests/test_gpu_parity.py tests/test_sharded.py -m gpu -x -q -k "chain or seven or row_wise or replicated or bit_identical or oracle_pivot_log" > gpurun_out/t11.log 2>&1; echo "tests rc=$?"; tail -4 gpurun_out/t11.log
for i in 1 2; do
timeout -k 10 200 python bench.py --no-late --no-cpu-baseline --no-pmc-traffic --no-secondary > gpurun_out/r04_pf_c3_$i.json 2>/dev/null
python3 -c "import json;d=json.load(open('gpurun_out/r04_pf_c3_$i.json'));print('config3 value',d['value'],'ms',d['ms_per_step'])"
done
timeout -k 10 200 python bench.py --rows 1024 --cols 2048 --seed 1002 --steps 5000 --warmup 500 --no-late --no-cpu-baseline --no-pmc-traffic --no-secondary > gpurun_out/r04_pf_c2.json 2>/dev/null
python3 -c "import json;d=json.load(open('gpurun_out/r04_pf_c2.json'));print('config2 value',d['value'],'ms',d['ms_per_step'])"
DZG_CHAIN_DEBUG=1 python tools/run_pivots.py 1600 2>&1 | tail -5
