# round 4: more fuzz on fresh seeds -- small dense, small CSC, mid-size dense, integer / 0-1 of 200-400 rows
mkdir -p gpurun_out
timeout -k 10 400 python3 tools/fuzz_parity.py 1500 50000 > gpurun_out/r04_fuzz_1500_small_lps.txt 2>&1; echo "small rc=$?"; tail -3 gpurun_out/r04_fuzz_1500_small_lps.txt | cut -c1-400
timeout -k 10 400 python3 tools/fuzz_parity.py 1000 52000 70 20000 1 012 csc > gpurun_out/r04_fuzz_1000_csc_lps.txt 2>&1; echo "csc rc=$?"; tail -3 gpurun_out/r04_fuzz_1000_csc_lps.txt | cut -c1-400
timeout -k 10 500 python3 tools/fuzz_parity.py 240 53000 320 20000 71 > gpurun_out/r04_fuzz_240_midsize_lps.txt 2>&1; echo "mid rc=$?"; tail -3 gpurun_out/r04_fuzz_240_midsize_lps.txt | cut -c1-400
timeout -k 10 500 python3 tools/fuzz_parity.py 120 54000 400 120 200 12 > gpurun_out/r04_fuzz_120_integer_lps_200_400_rows.txt 2>&1; echo "int rc=$?"; tail -3 gpurun_out/r04_fuzz_120_integer_lps_200_400_rows.txt | cut -c1-400
