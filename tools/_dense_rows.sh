set -e
python -m pytest tests/test_gpu_parity.py tests/test_sharded.py -x -q -m gpu -k "not sparse and not csc and not live and not strict and not lu_" > gpurun_out/s5_dense_tests.log 2>&1
python bench.py --no-secondary --no-cpu-baseline --no-mfma > gpurun_out/s5_bench_rows.json 2> gpurun_out/s5_bench_rows.err
