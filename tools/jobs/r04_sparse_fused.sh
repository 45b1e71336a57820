set -x
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "sparse_four or degenerate_shapes or live_lists or rows_alone or row_wise_pricing_alone" > gpurun_out/t7.log 2>&1; echo "tests rc=$?"; tail -8 gpurun_out/t7.log
DZG_SP_FUSED=0 timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "live_lists_survive" > gpurun_out/t7b.log 2>&1; echo "unfused live_lists rc=$?"; tail -3 gpurun_out/t7b.log
for f in 1 0; do
  DZG_SP_FUSED=$f timeout -k 10 300 python bench.py --rows 50000 --cols 100000 --sparse-per-col 50 --seed 1004 --steps 3000 --warmup 1000 --late-pivots 100000 --no-cpu-baseline --no-pmc-traffic --no-secondary > gpurun_out/r04_config4_fused$f.json 2>/dev/null
  python3 -c "
import json;d=json.load(open('gpurun_out/r04_config4_fused$f.json'));print('FUSED=$f value',d['value'],'ms',d['ms_per_step'],'late',d['late']['value'],d['late']['k_at_start'],d['late'].get('kernel_us_per_pivot'))"
done
