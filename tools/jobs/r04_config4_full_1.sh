# round 4: config 4 solved to optimality with the final code, first call (the six state arrays carry it over)
mkdir -p gpurun_out
timeout -k 10 1150 python3 tools/full_solve_sparse.py 50000 100000 50 1004 1000 200000 --state carry/config4_state_r04.npz --state-out gpurun_out/config4_state_r04.npz > gpurun_out/r04_config4_full_1.txt 2>&1
echo "rc=$?"; tail -5 gpurun_out/r04_config4_full_1.txt
