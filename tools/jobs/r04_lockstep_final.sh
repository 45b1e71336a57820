# round 4: per-rank compute of config 5 at k = 16 384 on the FINAL code (row stride of the inverse, LDS flush)
mkdir -p gpurun_out
L=gpurun_out/r04_lockstep_final.txt; : > $L
for args in "1 300 32768 65536 1005 0 16384 0" "8 200 32768 65536 1005 0 16384 1" "8 200 32768 65536 1005 0 16384 0"; do
  timeout -k 10 300 python tools/lockstep_profile.py $args >> $L 2>&1 || echo "FAILED: $args" >> $L
done
cat $L
