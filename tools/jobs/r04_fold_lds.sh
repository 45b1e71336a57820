set -x
mkdir -p gpurun_out
timeout -k 10 500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -x -q -k "chain or seven or row_wise or fast_run_invariants or first_pivots_are_the_cpu_oracles" > gpurun_out/t10.log 2>&1; echo "tests rc=$?"; tail -4 gpurun_out/t10.log
for fk in 2000000000 256; do
DZG_CHAIN_FOLD_K=$fk timeout -k 10 400 python bench.py --no-cpu-baseline --no-pmc-traffic --no-secondary --no-mfma > gpurun_out/r04_fold_lds_$fk.json 2> /dev/null; echo "bench rc=$?"
python3 - <<PY
import json
d=json.load(open('gpurun_out/r04_fold_lds_$fk.json'))
print('FOLD_K=$fk value',d['value'])
for k in ('late','deep','end'):
    print(k,d[k]['value'],d[k]['kernel_us_per_pivot'])
print('whole',d['whole_solve'])
PY
done
