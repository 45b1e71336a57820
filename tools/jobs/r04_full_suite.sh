# round 4: the whole GPU suite as the driver runs it (-x), then smoke()
mkdir -p gpurun_out
t0=$(date +%s)
timeout -k 10 850 python -m pytest tests -m gpu -x -q > gpurun_out/t7_full.log 2>&1; echo "suite rc=$? wall=$(( $(date +%s) - t0 )) s"; tail -4 gpurun_out/t7_full.log
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3
