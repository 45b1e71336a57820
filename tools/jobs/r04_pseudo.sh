# round 4: the pseudo-candidate fix -- parity suites, then the fuzz (both inputs)
mkdir -p gpurun_out
timeout -k 10 800 python -m pytest tests/test_gpu_parity.py tests/test_sharded.py tests/test_surface.py -m gpu -x -q > gpurun_out/t16.log 2>&1; echo "tests rc=$?"; tail -5 gpurun_out/t16.log
timeout -k 10 300 python3 tools/fuzz_parity.py 600 40000 > gpurun_out/r04_fuzz_600_small_lps.txt 2>&1; echo "dense rc=$?"; tail -3 gpurun_out/r04_fuzz_600_small_lps.txt
timeout -k 10 300 python3 tools/fuzz_parity.py 510 41000 70 20000 1 012 csc > gpurun_out/r04_fuzz_510_csc_lps.txt 2>&1; echo "csc rc=$?"; tail -3 gpurun_out/r04_fuzz_510_csc_lps.txt
