# round 4: config 4 against the oracle's two pivots, then the second call of the whole solve (resumes from the carried state)
mkdir -p gpurun_out
t0=$(date +%s)
timeout -k 10 600 python -m pytest tests/test_gpu_fullsize.py -m gpu -x -q -k "config4_first_pivots" > gpurun_out/t9.log 2>&1; echo "oracle test rc=$? wall=$(( $(date +%s) - t0 )) s"; tail -3 gpurun_out/t9.log
timeout -k 10 700 python3 tools/full_solve_sparse.py 50000 100000 50 1004 480 100000 --state carry/config4_state_r04.npz --state-out gpurun_out/config4_state2_r04.npz > gpurun_out/r04_config4_full_2.txt 2>&1
echo "rc=$?"; tail -6 gpurun_out/r04_config4_full_2.txt
