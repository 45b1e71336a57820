mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_fullsize.py -m gpu -x -q -k "deep_inside" 2>&1 | tail -3
timeout -k 10 900 python3 tools/strict_windows.py 50000 100000 1004 4 200000 1000000 --sparse-per-col 50 > gpurun_out/r04_strict_windows_c4.txt 2>&1; echo "rc=$?"; cat gpurun_out/r04_strict_windows_c4.txt
