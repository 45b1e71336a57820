# round 4: the look-ahead of the factorisation -- does a one-workgroup chain overlap a chip-filling grid, the refactor tests in both forms, timings
mkdir -p gpurun_out
./tools/bin/stream_overlap_bench > gpurun_out/r04_stream_overlap.txt 2>&1; cat gpurun_out/r04_stream_overlap.txt
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_sharded.py -m gpu -x -q -k "refactor" > gpurun_out/t10.log 2>&1; echo "tests rc=$?"; tail -5 gpurun_out/t10.log
for off in 0 1; do
  if [ $off = 1 ]; then export DZG_REF_NO_LOOKAHEAD=1; else unset DZG_REF_NO_LOOKAHEAD; fi
  python3 - <<'PY'
import os, time, numpy as np
from dantzig_amd import core
m, ns = 8192, 16384
a, b, c = core.gen_dense_lp(seed=1003, m=m, n_struct=ns)
for k in (4084, 8192):
    lp = core.warm_started(core.CoreLP.from_inequality_form(a, b, c), k)
    with core.Solver(lp, numerics=core.FAST, refactor_interval=-1, poll_interval=16) as s:
        s.run(1)
        best = 1e9
        for rep in range(3):
            t0 = time.perf_counter(); s.refactor(); best = min(best, time.perf_counter() - t0)
        r = s.result(log=False)
    print(f"DZG_REF_NO_LOOKAHEAD={os.environ.get('DZG_REF_NO_LOOKAHEAD','')} k={k}: refactor {1e3*best:.1f} ms, max_pivot_error {r.max_pivot_error:.2e}", flush=True)
PY
done
