import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dantzig_amd import core
m, ns, seed = 8192, 16384, 1003
a, b, c = core.gen_dense_lp(seed=seed, m=m, n_struct=ns)
lp = core.CoreLP.from_inequality_form(a, b, c)
with core.Solver(lp, numerics=core.FAST, refactor_interval=-1, poll_interval=50) as s:
    done = 0
    for chunk in (2000, 4000, 6000):
        t = time.perf_counter(); s.run(chunk); dt = time.perf_counter() - t
        done += chunk
        r = s.result(log=False)
        k = int((r.basis < ns).sum())
        print(f"iters {r.iterations} k={k} rate={chunk/dt:.0f} it/s max_pivot_err={r.max_pivot_error:.2e}", flush=True)
        t = time.perf_counter(); s.refactor(); dt = time.perf_counter() - t
        print(f"   refactor at k={k}: {dt*1e3:.1f} ms", flush=True)
        t = time.perf_counter(); s.run(200); dt = time.perf_counter() - t
        r = s.result(log=False)
        print(f"   after refactor: 200 its at {200/dt:.0f} it/s, max_pivot_err={r.max_pivot_error:.2e} status={r.status}", flush=True)
