set -e
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "sparse or csc or live" > gpurun_out/s4_sparse_tests.log 2>&1
python bench.py --rows 50000 --cols 100000 --seed 1004 --sparse-per-col 50 --steps 3000 --warmup 1000 --late-pivots 100000 --no-secondary --no-cpu-baseline > gpurun_out/s4_cfg4_rl.json 2> gpurun_out/s4_cfg4_rl.err
bash tools/kernel_stats.sh s4_cfg4_stats --rows 50000 --cols 100000 --seed 1004 --sparse-per-col 50 --steps 3000 --warmup 1000 --no-late --no-secondary --no-cpu-baseline --no-pmc-traffic > gpurun_out/s4_cfg4_stats.txt 2>&1
