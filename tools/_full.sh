set -e
python -m pytest tests -m gpu -x -q --durations=8 > gpurun_out/s9_gputests.log 2>&1
python bench.py --no-secondary --no-cpu-baseline > gpurun_out/s9_bench.json 2> gpurun_out/s9_bench.err
bash tools/kernel_stats.sh s9_stats --no-secondary --no-cpu-baseline > gpurun_out/s9_stats.txt 2>&1
