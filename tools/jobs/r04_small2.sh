mkdir -p gpurun_out
timeout -k 10 700 python -m pytest tests/test_gpu_parity.py tests/test_sharded.py -m gpu -x -q -k "chain or seven or hold or co_tenant or recovers or sparse_four" > gpurun_out/t12.log 2>&1; echo "tests rc=$?"; tail -5 gpurun_out/t12.log
