set -x
timeout -k 10 500 python -m pytest tests/test_sharded.py -m gpu -x -q > gpurun_out/t3_sharded.log 2>&1; echo "sharded rc=$?"; tail -3 gpurun_out/t3_sharded.log
timeout -k 10 500 python -m pytest tests/test_gpu_fullsize.py -m gpu -x -q -k "config5" > gpurun_out/t3_c5.log 2>&1; echo "config5 rc=$?"; tail -3 gpurun_out/t3_c5.log
L=gpurun_out/r04_lockstep_per_rank_compute.txt
: > $L
for args in "1 300 32768 65536 1005 0 16384 0" "8 200 32768 65536 1005 0 16384 0" "8 200 32768 65536 1005 0 16384 1" "1 600 32768 65536 1005 0 0 0" "8 300 32768 65536 1005 0 0 0" "8 300 32768 65536 1005 0 0 1"; do
  timeout -k 10 300 python tools/lockstep_profile.py $args >> $L 2>&1 || echo "FAILED: $args" >> $L
done
for args in "1 600 8192 16384 1003 0 0 0" "8 600 8192 16384 1003 0 0 0" "8 600 8192 16384 1003 0 0 1" "1 600 8192 16384 1003 0 4096 0" "8 400 8192 16384 1003 0 4096 0" "8 400 8192 16384 1003 0 4096 1"; do
  timeout -k 10 200 python tools/lockstep_profile.py $args >> $L 2>&1 || echo "FAILED: $args" >> $L
done
cat $L
