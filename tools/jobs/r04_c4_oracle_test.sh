mkdir -p gpurun_out
t0=$(date +%s)
timeout -k 10 800 python -m pytest tests/test_gpu_fullsize.py -m gpu -x -q -k "config4_first_pivots" > gpurun_out/t9.log 2>&1; echo "rc=$? wall=$(( $(date +%s) - t0 )) s"; tail -15 gpurun_out/t9.log
