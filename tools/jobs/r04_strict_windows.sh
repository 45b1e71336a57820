mkdir -p gpurun_out
timeout -k 10 1150 python3 tools/strict_windows.py 8192 16384 1003 150 50000 100000 200000 300000 450000 510000 > gpurun_out/r04_strict_windows_150.txt 2>&1; echo "rc=$?"; cat gpurun_out/r04_strict_windows_150.txt
