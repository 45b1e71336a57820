# round 4: row stride of the compact inverse (FTRAN's stream) at k = 7 700 and k = 4 050
mkdir -p gpurun_out
root=$PWD
cd /tmp && export TMPDIR=/tmp
for pad in 32 64 96 160 288 544 1056; do
for k in 7700 6000; do
  out=$root/gpurun_out/r04_ldb_$pad_$k
  mkdir -p $out
  DZG_LDB_PAD=$pad timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/raw -- python3 $root/tools/run_pivots.py 2000 8192 16384 1003 0 $k > $out/run.txt 2>&1
  f=$(find $out/raw -name '*kernel_stats.csv' | head -1)
  echo "PAD=$pad k=$k $(grep -E 'k_chain_pre' $f | cut -d'"' -f3 | cut -d, -f2-4) flush $(grep -E 'k_fast_flush_mfma' $f | cut -d'"' -f3 | cut -d, -f2-4)"
  rm -rf $out/raw
done
done
