# round 4: the benchmark LP solved with the final code and certified optimal by LAPACK on the host; STRICT windows inside config 5's solve
mkdir -p gpurun_out
timeout -k 10 500 python3 tools/full_solve.py 8192 16384 1003 > gpurun_out/r04_full_solve_8192x16384.txt 2>&1; echo "full solve rc=$?"; tail -8 gpurun_out/r04_full_solve_8192x16384.txt
timeout -k 10 600 python3 tools/strict_windows.py 32768 65536 1005 4 20000 100000 > gpurun_out/r04_strict_windows_c5.txt 2>&1; echo "windows rc=$?"; cat gpurun_out/r04_strict_windows_c5.txt
