# round 4: does a share of Binv0 kept in the Infinity Cache (plain loads on the leading columns, nontemporal on the rest) speed FTRAN up?
set -x
mkdir -p gpurun_out
root=$PWD
cd /tmp && export TMPDIR=/tmp
for mb in 0 64 128 192 256; do
  out=$root/gpurun_out/r04_mall_$mb
  mkdir -p $out
  DZG_FTRAN_VARIANT=3 DZG_FTRAN_MALL_MB=$mb timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/raw -- python3 $root/tools/run_pivots.py 2000 8192 16384 1003 0 7700 > $out/run.txt 2>&1
  f=$(find $out/raw -name '*kernel_stats.csv' | head -1); cp "$f" $out/kernel_stats.csv; rm -rf $out/raw
  tail -1 $out/run.txt
  grep -E "k_chain_pre|k_chain_post|k_price_tree" $out/kernel_stats.csv | cut -d, -f2-4 
done
