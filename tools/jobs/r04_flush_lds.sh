# round 4: the dense eta flush with the Wc tile staged through LDS -- same bits, kernel time at k = 7 700 / 4 050 / 1 050
mkdir -p gpurun_out
for no in "" 1; do
  if [ -n "$no" ]; then export DZG_FLUSH_NO_LDS=1; else unset DZG_FLUSH_NO_LDS; fi
  python3 - <<'PY'
import hashlib, os, numpy as np
from dantzig_amd import core
out = []
for (m, ns, seed, piv, wk) in ((1024, 2048, 1002, 3000, 0), (2100, 4200, 77, 700, 1500)):
    a, b, c = core.gen_dense_lp(seed=seed, m=m, n_struct=ns)
    lp = core.CoreLP.from_inequality_form(a, b, c)
    if wk: lp = core.warm_started(lp, wk)
    r = core.solve(lp, numerics=core.FAST, max_iter=piv, poll_interval=50)
    h = hashlib.sha256(np.concatenate([r.x, r.xbar, r.z, r.zbar]).tobytes() + str([p[:3] for p in r.pivots]).encode()).hexdigest()[:16]
    out.append((m, r.iterations, r.dense_columns, h, f"{r.max_pivot_error:.3e}"))
print("DZG_FLUSH_NO_LDS=%s" % os.environ.get("DZG_FLUSH_NO_LDS", ""), out, flush=True)
PY
done
root=$PWD
cd /tmp && export TMPDIR=/tmp
for no in "" 1; do
  if [ -n "$no" ]; then export DZG_FLUSH_NO_LDS=1; else unset DZG_FLUSH_NO_LDS; fi
for k in 7700 4050 1050; do
  out=$root/gpurun_out/r04_flush
  mkdir -p $out
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/raw -- python3 $root/tools/run_pivots.py 2000 8192 16384 1003 0 $k > $out/run.txt 2>&1
  f=$(find $out/raw -name '*kernel_stats.csv' | head -1)
  echo "NO_LDS=$no k=$k $(grep -E 'k_fast_flush_mfma' $f | sed 's/(.*)",/,/' | cut -d, -f1-4)"
  rm -rf $out/raw
done
done
