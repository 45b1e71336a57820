mkdir -p gpurun_out
timeout -k 10 300 python3 tools/fuzz_parity.py 1500 60000 > gpurun_out/r04_fuzz3_small.txt 2>&1; echo "small rc=$?"; tail -3 gpurun_out/r04_fuzz3_small.txt | cut -c1-300
timeout -k 10 300 python3 tools/fuzz_parity.py 800 62000 70 20000 1 012 csc > gpurun_out/r04_fuzz3_csc.txt 2>&1; echo "csc rc=$?"; tail -3 gpurun_out/r04_fuzz3_csc.txt | cut -c1-300
