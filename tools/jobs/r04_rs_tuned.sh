set -x
mkdir -p gpurun_out
timeout -k 10 500 python -m pytest tests/test_sharded.py -m gpu -x -q -k "row_sharded or warm_start" > gpurun_out/t9.log 2>&1; echo "tests rc=$?"; tail -4 gpurun_out/t9.log
timeout -k 10 200 python -m pytest tests/test_gpu_fullsize.py -m gpu -x -q -k "config5_row_sharded or config5_first_pivots_are_the_cpu" > gpurun_out/t9b.log 2>&1; echo "config5 rc=$?"; tail -3 gpurun_out/t9b.log
L=gpurun_out/r04_lockstep_rs_tuned.txt; : > $L
for args in "8 200 32768 65536 1005 0 16384 1" "8 200 32768 65536 1005 1 16384 1" "8 300 32768 65536 1005 0 0 1" "8 400 8192 16384 1003 0 4096 1" "8 600 8192 16384 1003 0 0 1"; do
  timeout -k 10 300 python tools/lockstep_profile.py $args >> $L 2>&1 || echo "FAILED: $args" >> $L
done
cat $L
