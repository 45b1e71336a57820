set -e
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "row_wise or chain_and_seven or fast_matches_strict" > gpurun_out/s7_tests.log 2>&1
python bench.py --no-secondary --no-cpu-baseline > gpurun_out/s7_bench.json 2> gpurun_out/s7_bench.err
bash tools/kernel_stats.sh s7_stats --no-secondary --no-cpu-baseline > gpurun_out/s7_stats.txt 2>&1
