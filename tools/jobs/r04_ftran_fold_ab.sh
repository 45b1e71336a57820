# round 4: A/B of the FTRAN row loads (DZG_FTRAN_VARIANT) in the end regime, and of the folded finishing launch
set -x
mkdir -p gpurun_out
timeout -k 10 500 python -m pytest tests/test_gpu_parity.py tests/test_sharded.py -m gpu -x -q -k "chain or seven or processes_on_one_gpu or row_wise" > gpurun_out/t5.log 2>&1; echo "tests rc=$?"; tail -5 gpurun_out/t5.log
for nf in 0 1; do
  DZG_CHAIN_NO_FOLD=$nf timeout -k 10 200 python bench.py --no-late --no-cpu-baseline --no-pmc-traffic --no-secondary > gpurun_out/r04_fold_nofold$nf.json 2>/dev/null
  python3 -c "import json;d=json.load(open('gpurun_out/r04_fold_nofold$nf.json'));print('NO_FOLD=$nf value',d['value'],'ms',d['ms_per_step'],'price us',d['roofline']['avg_launch_us'])"
  DZG_CHAIN_NO_FOLD=$nf timeout -k 10 200 python bench.py --rows 1024 --cols 2048 --seed 1002 --steps 5000 --warmup 500 --no-late --no-cpu-baseline --no-pmc-traffic --no-secondary > gpurun_out/r04_fold_c2_nofold$nf.json 2>/dev/null
  python3 -c "import json;d=json.load(open('gpurun_out/r04_fold_c2_nofold$nf.json'));print('config2 NO_FOLD=$nf value',d['value'],'ms',d['ms_per_step'])"
done
root=$PWD
cd /tmp && export TMPDIR=/tmp
for v in 0 1 2; do
  out=$root/gpurun_out/r04_ftran_v$v
  mkdir -p $out
  DZG_FTRAN_VARIANT=$v timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/raw -- python3 $root/tools/run_pivots.py 2000 8192 16384 1003 0 7700 > $out/run.txt 2>&1
  f=$(find $out/raw -name '*kernel_stats.csv' | head -1); cp "$f" $out/kernel_stats.csv; rm -rf $out/raw
  tail -1 $out/run.txt
  grep -E "k_chain_pre|k_chain_post|k_price_tree" $out/kernel_stats.csv | cut -c1-200
done
