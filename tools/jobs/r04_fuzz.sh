# round 4: the fuzz against the CPU oracle with the final code -- dense and CSC input, fresh seeds
mkdir -p gpurun_out
timeout -k 10 520 python3 tools/fuzz_parity.py 600 40000 > gpurun_out/r04_fuzz_600_small_lps.txt 2>&1; echo "dense rc=$?"; tail -3 gpurun_out/r04_fuzz_600_small_lps.txt
timeout -k 10 520 python3 tools/fuzz_parity.py 510 41000 70 20000 1 012 csc > gpurun_out/r04_fuzz_510_csc_lps.txt 2>&1; echo "csc rc=$?"; tail -3 gpurun_out/r04_fuzz_510_csc_lps.txt
