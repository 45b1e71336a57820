# round 4: column stride of the matrix (k_price_tree's streams) at k = 7 700 and 6 000
mkdir -p gpurun_out
root=$PWD
cd /tmp && export TMPDIR=/tmp
for pad in 272 0 16 160 288 528 544 1040; do
for k in 7700 6000; do
  out=$root/gpurun_out/r04_lda
  mkdir -p $out
  DZG_LDA_PAD=$pad timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/raw -- python3 $root/tools/run_pivots.py 2000 8192 16384 1003 0 $k > $out/run.txt 2>&1
  f=$(find $out/raw -name '*kernel_stats.csv' | head -1)
  echo "LDA PAD=$pad k=$k tree $(grep -E 'k_price_tree' $f | sed 's/.*)",//' | cut -d, -f1-3) pre $(grep -E 'k_chain_pre' $f | sed 's/.*)",//' | cut -d, -f3)"
  rm -rf $out/raw
done
done
