set -e
python -m pytest tests -m gpu -x -q > gpurun_out/z1_gputests.log 2>&1
python bench.py > gpurun_out/z1_bench.json 2> gpurun_out/z1_bench.err
python bench.py --steps 20 --warmup 5 --no-whole-solve > gpurun_out/z1_bench_driver_like.json 2> gpurun_out/z1_bench_driver_like.err
bash tools/kernel_stats.sh z1_stats --no-secondary --no-cpu-baseline > gpurun_out/z1_stats.txt 2>&1
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/z1_smoke.log 2>&1
