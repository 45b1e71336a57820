# round 4: rocprofv3 --kernel-trace --stats of 2 000 pivots warm-started at k = 1 050 and k = 4 050 (the regimes of the bench's late and deep blocks), final code
mkdir -p gpurun_out
root=$PWD
cd /tmp && export TMPDIR=/tmp
for k in 1050 4050; do
  out=$root/gpurun_out/r04_regime_k$k
  mkdir -p $out
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/raw -- python3 $root/tools/run_pivots.py 2000 8192 16384 1003 0 $k > $out/run.txt 2>&1
  f=$(find $out/raw -name '*kernel_stats.csv' | head -1); cp "$f" $out/kernel_stats.csv; rm -rf $out/raw
  grep -v rocprof $out/run.txt | tail -1
  grep -E "k_chain_pre|k_chain_post|k_price" $out/kernel_stats.csv | cut -d'"' -f2- | awk -F'",' '{print substr($1,1,40), $2}' | cut -c1-100
done
