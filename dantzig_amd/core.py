"""Host-side Python view of the Level-1 C ABI (include/dantzig_amd.h).

`CoreLP` carries the state Simplex::new leaves behind (src/simplex.rs:209-223) and `Solver`
replaces Simplex::solve (src/simplex.rs:332-343) with the HIP engine.  Everything here is
plumbing over ctypes; the arithmetic lives in dantzig_amd/csrc/*.hip.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field

import numpy as np

from . import _ffi
from ._ffi import (AUTO, FAST, STRICT, PRICE_AUTO, PRICE_SEQ, PRICE_WAVE, PRICE_TREE,  # noqa: F401
                   STEP_DUAL,
                   STEP_PRIMAL, NEAR_TIE_COUNT, NEAR_TIE_STOP, DantzigAmdError, f64, i64, ptr)

STATUS_NAMES = {0: "optimal", 1: "unbounded", 2: "infeasible", 3: "iter_limit", 4: "singular",
                5: "panic", 6: "running", 7: "near_tie"}


@dataclass
class CoreLP:
    """maximise c.x + constant  s.t.  [A | slacks] x = rhs, x >= 0, in Simplex::new's layout."""
    a: np.ndarray | None          # (m, n_struct) any layout; uploaded column-major.  None: CSC
    c: np.ndarray                 # n
    basis: np.ndarray             # m
    nonbasis: np.ndarray          # n - m
    x: np.ndarray                 # m
    z: np.ndarray                 # n - m
    var_col: np.ndarray | None = None
    constant: float = 0.0
    # sparse structural block (used when a is None): CSC, rows ascending inside a column
    col_ptr: np.ndarray | None = None
    row_idx: np.ndarray | None = None
    val: np.ndarray | None = None
    # column sharding: `a` holds only the structural columns [block[0], block[1]) of an LP with
    # n_struct_full structural columns (opts.a_is_block); everything else is complete
    block: tuple | None = None
    n_struct_full: int | None = None
    # perturbation vectors (Simplex.x_bar / z_bar); None = ones (Simplex::new).  With a non-slack
    # basis, its x and z: resume a solve from a CoreResult (CoreLP.resumed_from)
    xbar: np.ndarray | None = None
    zbar: np.ndarray | None = None

    @property
    def m(self) -> int:
        return int(self.a.shape[0]) if self.a is not None else len(self.basis)

    @property
    def n_struct(self) -> int:
        if self.n_struct_full is not None:
            return int(self.n_struct_full)
        return int(self.a.shape[1]) if self.a is not None else len(self.col_ptr) - 1

    @property
    def n(self) -> int:
        return len(self.c)

    @classmethod
    def from_inequality_form(cls, a, b, c, constant: float = 0.0) -> "CoreLP":
        """max c.x st A x <= b, x >= 0 in the benchmark convention of SURVEY 8(d): variables
        0..ns-1 structural, ns..ns+m-1 slacks, slack basis, x = b, z = -c."""
        a = np.asarray(a, dtype=np.float64)
        m, ns = a.shape
        return cls(a=a, c=np.concatenate([f64(c), np.zeros(m)]),
                   basis=np.arange(ns, ns + m, dtype=np.int64),
                   nonbasis=np.arange(ns, dtype=np.int64), x=f64(b).copy(), z=-f64(c),
                   var_col=None, constant=constant)

    @classmethod
    def from_inequality_block(cls, a_block, b, c, begin: int, end: int,
                              constant: float = 0.0) -> "CoreLP":
        """One rank's view of from_inequality_form: `a_block` = columns [begin, end) of A."""
        a_block = np.asarray(a_block, dtype=np.float64)
        m, ns = a_block.shape[0], len(c)
        if a_block.shape[1] != end - begin or not 0 <= begin <= end <= ns:
            raise ValueError("a_block does not match [begin, end)")
        return cls(a=a_block, c=np.concatenate([f64(c), np.zeros(m)]),
                   basis=np.arange(ns, ns + m, dtype=np.int64),
                   nonbasis=np.arange(ns, dtype=np.int64), x=f64(b).copy(), z=-f64(c),
                   var_col=None, constant=constant, block=(int(begin), int(end)),
                   n_struct_full=ns)

    @classmethod
    def from_csc(cls, m, col_ptr, row_idx, val, b, c, constant: float = 0.0) -> "CoreLP":
        """Same convention with the structural block in CSC (kept sparse on the device)."""
        ns = len(col_ptr) - 1
        return cls(a=None, c=np.concatenate([f64(c), np.zeros(m)]),
                   basis=np.arange(ns, ns + m, dtype=np.int64),
                   nonbasis=np.arange(ns, dtype=np.int64), x=f64(b).copy(), z=-f64(c),
                   constant=constant, col_ptr=i64(col_ptr),
                   row_idx=np.ascontiguousarray(row_idx, dtype=np.int32), val=f64(val))


def warm_started(lp: CoreLP, k: int) -> CoreLP:
    """The same LP started from the basis "first k structural columns in the place of the first k
    slacks" (x = 1, z = -1: any state will do for a benchmark -- the engine factorises the basis and
    pivots from there).  How the deep regimes of a solve -- a compact inverse k columns wide -- are
    reached without running the hundreds of thousands of pivots that lead there (bench.py,
    tests/test_gpu_fullsize.py); works on whole and on column-block LPs of the benchmark convention."""
    import dataclasses

    m, ns = lp.m, lp.n_struct
    if lp.var_col is not None or not 0 <= k <= min(m, ns):
        raise ValueError("warm_started: benchmark-convention LP and 0 <= k <= min(m, n_struct)")
    basis = np.concatenate([np.arange(k), ns + np.arange(k, m)]).astype(np.int64)
    nonbasis = np.concatenate([np.arange(k, ns), ns + np.arange(k)]).astype(np.int64)
    return dataclasses.replace(lp, basis=basis, nonbasis=nonbasis, x=np.ones(m), z=-np.ones(ns),
                               xbar=None, zbar=None)


def resumed_from(lp: CoreLP, r) -> CoreLP:
    """The same LP in the state a CoreResult (or anything with basis, nonbasis, x, xbar, z, zbar)
    left it in: a solver created on it factorises that basis and carries the solve on."""
    import dataclasses

    def get(name):  # a CoreResult, or a mapping such as an .npz archive of one
        return r[name] if hasattr(r, "keys") else getattr(r, name)
    return dataclasses.replace(lp, basis=i64(get("basis")).copy(), nonbasis=i64(get("nonbasis")).copy(),
                               x=f64(get("x")).copy(), z=f64(get("z")).copy(),
                               xbar=f64(get("xbar")).copy(), zbar=f64(get("zbar")).copy())


@dataclass
class CoreResult:
    status: str
    status_code: int
    numerics: str
    iterations: int
    objective: float
    basis: np.ndarray
    nonbasis: np.ndarray
    x: np.ndarray
    xbar: np.ndarray
    z: np.ndarray
    zbar: np.ndarray
    pivots: list = field(default_factory=list)   # (kind, entering, leaving, mu)
    kernel_ms: dict = field(default_factory=dict)
    kernel_launches: dict = field(default_factory=dict)
    price_bytes: float = 0.0
    solve_ms: float = 0.0
    max_pivot_error: float = 0.0
    # FAST near-tie arbitration: the pivot log is certified to be the reference's up to (not
    # including) pivot first_near_tie; near_ties == 0: all of it
    near_ties: int = 0
    first_near_tie: int = -1
    min_margin: float = float("inf")
    margins: np.ndarray | None = None   # per-pivot smallest decision margin (log=True)
    dense_columns: int = 0              # k: structural basics = dense columns of the inverse
    refactors: int = 0
    chain_fallbacks: int = 0            # failed device-wide barriers recovered from (k_chain.hip)
    price_pass_used: int = 0            # bit mask: 1 row-wise pricing pass ran, 2 column-wise
    price_rows_copy: int = 0            # 1: the row-major copy of the matrix is resident
    state_drift: float = 0.0            # carried x_B / z_N vs the fresh inverse at the last refactorisation


def _counters(r) -> dict:
    """The scalar tail of a dzg_result that every entry point reports alike."""
    return dict(max_pivot_error=float(r.max_pivot_error), near_ties=int(r.near_ties),
                first_near_tie=int(r.first_near_tie), min_margin=float(r.min_margin),
                dense_columns=int(r.dense_columns), refactors=int(r.refactors),
                chain_fallbacks=int(r.chain_fallbacks), price_pass_used=int(r.price_pass_used),
                price_rows_copy=int(r.price_rows_copy), state_drift=float(r.state_drift))


class Solver:
    """One LP resident on one GPU.  create -> run (repeatable, budgeted) -> result."""

    def __init__(self, lp: CoreLP, **opts):
        self._prepare(lp, opts)
        rc = _ffi.lib().dzg_solver_create(C.byref(self._c_lp), C.byref(self._opts),
                                          C.byref(self._h))
        _ffi.check(rc, "dzg_solver_create")

    def _prepare(self, lp: CoreLP, opts: dict) -> "Solver":
        """Marshals the LP and the options into their C structs (no device work)."""
        _ffi.require_gpu()
        self._lp = lp
        m, ns, n = lp.m, lp.n_struct, lp.n
        # column-major m x ns == C-contiguous (ns, m)
        a_cm = None if lp.a is None else np.ascontiguousarray(np.asarray(lp.a, dtype=np.float64).T)
        self._keep = dict(a=a_cm, c=f64(lp.c), basis=i64(lp.basis), nonbasis=i64(lp.nonbasis),
                          x=f64(lp.x), z=f64(lp.z),
                          var_col=None if lp.var_col is None else i64(lp.var_col),
                          col_ptr=None if lp.col_ptr is None else i64(lp.col_ptr),
                          row_idx=None if lp.row_idx is None
                          else np.ascontiguousarray(lp.row_idx, dtype=np.int32),
                          val=None if lp.val is None else f64(lp.val),
                          xbar=None if lp.xbar is None else f64(lp.xbar),
                          zbar=None if lp.zbar is None else f64(lp.zbar))
        k = self._keep
        if len(k["basis"]) != m or len(k["x"]) != m or len(k["nonbasis"]) != n - m \
                or len(k["z"]) != n - m \
                or (k["xbar"] is not None and len(k["xbar"]) != m) \
                or (k["zbar"] is not None and len(k["zbar"]) != n - m):
            raise ValueError("CoreLP vectors do not match (m, n)")
        self._c_lp = _ffi.Lp(m, n, ns, ptr(a_cm), max(m, 1), ptr(k["var_col"]), ptr(k["c"]),
                             float(lp.constant), ptr(k["basis"]), ptr(k["nonbasis"]),
                             ptr(k["x"]), ptr(k["z"]), ptr(k["col_ptr"]), ptr(k["row_idx"]),
                             ptr(k["val"]), ptr(k["xbar"]), ptr(k["zbar"]))
        if lp.block is not None:
            if (opts.get("col_begin"), opts.get("col_end")) != tuple(lp.block):
                raise ValueError("a column-block LP needs a sharded solver on exactly that block")
            opts["a_is_block"] = 1
        self._opts = _ffi.default_opts(**opts)
        self._h = C.c_void_p(None)
        return self

    def run(self, max_new_iters: int = 0) -> str:
        rc = _ffi.lib().dzg_solver_run(self._h, int(max_new_iters))
        _ffi.check(rc, "dzg_solver_run")
        return STATUS_NAMES[rc]

    def result(self, log: bool = True, log_cap: int | None = None) -> CoreResult:
        m, q = self._lp.m, self._lp.n - self._lp.m
        basis, nonbasis = np.zeros(max(m, 1), np.int64), np.zeros(max(q, 1), np.int64)
        x, xbar = np.zeros(max(m, 1)), np.zeros(max(m, 1))
        z, zbar = np.zeros(max(q, 1)), np.zeros(max(q, 1))
        cap = int(log_cap if log_cap is not None else (1 << 22)) if log else 0
        # first learn the iteration count so the log buffer is not oversized
        r = _ffi.Result()
        r.basis, r.nonbasis, r.x, r.xbar = ptr(basis), ptr(nonbasis), ptr(x), ptr(xbar)
        r.z, r.zbar = ptr(z), ptr(zbar)
        r.log, r.log_cap = None, 0
        _ffi.check(_ffi.lib().dzg_solver_result(self._h, C.byref(r)), "dzg_solver_result")
        pivots, margins = [], None
        if log and r.iterations > 0:
            cnt = int(min(r.iterations, cap))
            buf = (_ffi.Pivot * cnt)()
            margins = np.full(cnt, np.inf)
            r.log, r.log_cap, r.margins = C.cast(buf, C.c_void_p), cnt, ptr(margins)
            _ffi.check(_ffi.lib().dzg_solver_result(self._h, C.byref(r)), "dzg_solver_result")
            arr = np.ctypeslib.as_array(buf)
            pivots = list(zip(arr["kind"].tolist(), arr["entering"].tolist(),
                              arr["leaving"].tolist(), arr["mu"].tolist()))
        return CoreResult(
            status=STATUS_NAMES.get(r.status, str(r.status)), status_code=r.status,
            numerics="strict" if r.numerics_used == STRICT else "fast",
            iterations=int(r.iterations), objective=float(r.objective),
            basis=basis[:m].copy(), nonbasis=nonbasis[:q].copy(), x=x[:m].copy(),
            xbar=xbar[:m].copy(), z=z[:q].copy(), zbar=zbar[:q].copy(), pivots=pivots,
            kernel_ms={k: r.kernel_ms[i] for i, k in enumerate(_ffi.KERNEL_CLASSES)},
            kernel_launches={k: r.kernel_launches[i] for i, k in enumerate(_ffi.KERNEL_CLASSES)},
            price_bytes=float(r.price_bytes), solve_ms=float(r.solve_ms), margins=margins,
            **_counters(r))

    def set_profile(self, mask: int) -> None:
        """Which kernel classes (bits 1 << _ffi.K_*) the following runs time with HIP events."""
        _ffi.check(_ffi.lib().dzg_solver_set_profile(self._h, int(mask)), "dzg_solver_set_profile")

    def refactor(self) -> None:
        """FAST: rebuild the basis inverse from scratch now (blocked LU + MFMA GEMMs)."""
        _ffi.check(_ffi.lib().dzg_solver_refactor(self._h), "dzg_solver_refactor")

    def close(self) -> None:
        if self._h:
            _ffi.lib().dzg_solver_destroy(self._h)
            self._h = C.c_void_p(None)

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def solve(lp: CoreLP, log: bool = True, **opts) -> CoreResult:
    """Simplex::solve on the GPU: create + run to termination + result."""
    with Solver(lp, **opts) as s:
        s.run(0)
        return s.result(log=log)


def core_solve(lp: CoreLP, log_cap: int = 1 << 20, **opts) -> CoreResult:
    """dzg_core_solve: one call, with the AUTO policy of the C ABI (FAST that stops at the first
    near tie and is then re-solved in STRICT, up to DZG_AUTO_STRICT_RESTART_ROWS rows)."""
    with Solver.__new__(Solver)._prepare(lp, opts) as s:
        m, q = lp.m, lp.n - lp.m
        basis, nonbasis = np.zeros(max(m, 1), np.int64), np.zeros(max(q, 1), np.int64)
        x, xbar = np.zeros(max(m, 1)), np.zeros(max(m, 1))
        z, zbar = np.zeros(max(q, 1)), np.zeros(max(q, 1))
        buf = (_ffi.Pivot * max(log_cap, 1))()
        margins = np.full(max(log_cap, 1), np.inf)
        r = _ffi.Result()
        r.basis, r.nonbasis, r.x, r.xbar = ptr(basis), ptr(nonbasis), ptr(x), ptr(xbar)
        r.z, r.zbar = ptr(z), ptr(zbar)
        r.log, r.log_cap, r.margins = C.cast(buf, C.c_void_p), log_cap, ptr(margins)
        rc = _ffi.lib().dzg_core_solve(C.byref(s._c_lp), C.byref(s._opts), C.byref(r))
        _ffi.check(rc, "dzg_core_solve")
        cnt = int(min(r.iterations, log_cap))
        arr = np.ctypeslib.as_array(buf)[:cnt]
        pivots = list(zip(arr["kind"].tolist(), arr["entering"].tolist(),
                          arr["leaving"].tolist(), arr["mu"].tolist()))
        return CoreResult(
            status=STATUS_NAMES.get(r.status, str(r.status)), status_code=r.status,
            numerics="strict" if r.numerics_used == STRICT else "fast",
            iterations=int(r.iterations), objective=float(r.objective),
            basis=basis[:m].copy(), nonbasis=nonbasis[:q].copy(), x=x[:m].copy(),
            xbar=xbar[:m].copy(), z=z[:q].copy(), zbar=zbar[:q].copy(), pivots=pivots,
            margins=margins[:cnt].copy(), **_counters(r))


def core_solve_full_csc(m: int, n: int, col_ptr, row_idx, val, c, constant, basis, nonbasis, x, z,
                        log_cap: int = 1 << 16, **opts) -> CoreResult:
    """dzg_core_solve_full_csc: Level 1 on the reference's own `Simplex` fields -- ONE CSC over all n
    columns, slack columns included, 64-bit indices (src/simplex.rs:84-112, src/linalg.rs:161-168).
    The in/out arrays of the C call are copies here; the final state comes back in the result."""
    _ffi.require_gpu()
    q = n - m
    col_ptr, row_idx, val, c = i64(col_ptr), i64(row_idx), f64(val), f64(c)
    basis, nonbasis = i64(basis).copy(), i64(nonbasis).copy()
    x, z = f64(x).copy(), f64(z).copy()
    if len(row_idx) == 0:
        row_idx, val = np.zeros(1, np.int64), np.zeros(1)
    xbar, zbar = np.zeros(max(m, 1)), np.zeros(max(q, 1))
    buf = (_ffi.Pivot * max(log_cap, 1))()
    margins = np.full(max(log_cap, 1), np.inf)
    r = _ffi.Result()
    r.xbar, r.zbar = ptr(xbar), ptr(zbar)
    r.log, r.log_cap, r.margins = C.cast(buf, C.c_void_p), log_cap, ptr(margins)
    o = _ffi.default_opts(**opts)
    rc = _ffi.lib().dzg_core_solve_full_csc(
        int(m), int(n), ptr(col_ptr), ptr(row_idx), ptr(val), ptr(c), float(constant),
        ptr(basis if m else np.zeros(1, np.int64)), ptr(nonbasis if q else np.zeros(1, np.int64)),
        ptr(x if m else np.zeros(1)), ptr(z if q else np.zeros(1)), C.byref(o), C.byref(r))
    _ffi.check(rc, "dzg_core_solve_full_csc")
    cnt = int(min(r.iterations, log_cap))
    arr = np.ctypeslib.as_array(buf)[:cnt]
    pivots = list(zip(arr["kind"].tolist(), arr["entering"].tolist(), arr["leaving"].tolist(),
                      arr["mu"].tolist()))
    return CoreResult(
        status=STATUS_NAMES.get(r.status, str(r.status)), status_code=r.status,
        numerics="strict" if r.numerics_used == STRICT else "fast",
        iterations=int(r.iterations), objective=float(r.objective),
        basis=basis[:m], nonbasis=nonbasis[:q], x=x[:m], xbar=xbar[:m].copy(), z=z[:q],
        zbar=zbar[:q].copy(), pivots=pivots, margins=margins[:cnt].copy(), **_counters(r))


# ------------------------------------------------------------------ synthetic LPs
def gen_dense_lp(seed: int, m: int, n_struct: int):
    """Generator G1.  Returns (A as an (m, n_struct) Fortran-ordered array, b, c)."""
    a = np.empty((n_struct, m), dtype=np.float64)  # C-contiguous (ns, m) == column-major m x ns
    b = np.empty(m)
    c = np.empty(n_struct)
    rc = _ffi.lib().dzg_gen_dense_lp(C.c_uint64(seed), m, n_struct, ptr(a), m, ptr(b), ptr(c))
    _ffi.check(rc, "dzg_gen_dense_lp")
    return a.T, b, c


def gen_dense_lp_block(seed: int, m: int, n_struct: int, begin: int, end: int):
    """Generator G1, columns [begin, end) of A only (bit-identical to that slice of
    gen_dense_lp); b and c are complete.  Returns (A block (m, end-begin), b, c)."""
    a = np.empty((max(end - begin, 1), m), dtype=np.float64)
    b = np.empty(m)
    c = np.empty(n_struct)
    rc = _ffi.lib().dzg_gen_dense_lp_block(C.c_uint64(seed), m, n_struct, begin, end, ptr(a), m,
                                           ptr(b), ptr(c))
    _ffi.check(rc, "dzg_gen_dense_lp_block")
    return a[:end - begin].T, b, c


def gen_sparse_lp(seed: int, m: int, n_struct: int, per_col: int):
    """Generator G2.  Returns (col_ptr, row_idx, val, b, c) with `per_col` nonzeros per column."""
    col_ptr = np.zeros(n_struct + 1, dtype=np.int64)
    row_idx = np.zeros(n_struct * per_col, dtype=np.int32)
    val = np.zeros(n_struct * per_col)
    b, c = np.empty(m), np.empty(n_struct)
    rc = _ffi.lib().dzg_gen_sparse_lp(C.c_uint64(seed), C.c_int64(m), C.c_int64(n_struct),
                                      C.c_int64(per_col), ptr(col_ptr), ptr(row_idx), ptr(val),
                                      ptr(b), ptr(c))
    _ffi.check(rc, "dzg_gen_sparse_lp")
    return col_ptr, row_idx, val, b, c


# ------------------------------------------------------------------ single reference functions
def lu_solve(a: np.ndarray, b: np.ndarray, device: int = 0):
    """lu_solve (src/linalg.rs:8-10) on the GPU.  Returns (x, packed LU, pivots)."""
    _ffi.require_gpu()
    a, b = f64(a), f64(b)
    n = a.shape[0]
    x, lu, p = np.empty(n), np.empty((n, n)), np.zeros(max(n - 1, 1), np.int64)
    rc = _ffi.lib().dzg_kernel_lu_solve(C.c_int64(n), ptr(a), ptr(b), ptr(x), ptr(lu), ptr(p),
                                        C.c_int32(device))
    _ffi.check(rc, "dzg_kernel_lu_solve")
    return x, lu, p[: n - 1]


def neg_t_dot(a: np.ndarray, cols, v, kernel: int = PRICE_SEQ, device: int = 0) -> np.ndarray:
    """collect_columns(cols).neg_t_dot(v) (src/linalg.rs:188-207) on the GPU, a is (m, ns)."""
    _ffi.require_gpu()
    a = np.asarray(a, dtype=np.float64)
    m, ns = a.shape
    a_cm = np.ascontiguousarray(a.T)
    cols, v = i64(cols), f64(v)
    out = np.empty(max(len(cols), 1))
    rc = _ffi.lib().dzg_kernel_neg_t_dot(C.c_int64(m), C.c_int64(ns), ptr(a_cm), C.c_int64(m),
                                         ptr(cols), C.c_int64(len(cols)), ptr(v), ptr(out),
                                         C.c_int32(kernel), C.c_int32(device))
    _ffi.check(rc, "dzg_kernel_neg_t_dot")
    return out[: len(cols)]


def neg_t_dot_csc(m: int, col_ptr, row_idx, val, cols, v, device: int = 0) -> np.ndarray:
    """neg_t_dot over a CSC matrix kept sparse on the device (k_price_csc)."""
    _ffi.require_gpu()
    col_ptr, cols, v, val = i64(col_ptr), i64(cols), f64(v), f64(val)
    row_idx = np.ascontiguousarray(row_idx, dtype=np.int32)
    out = np.empty(max(len(cols), 1))
    rc = _ffi.lib().dzg_kernel_neg_t_dot_csc(
        C.c_int64(m), C.c_int64(len(col_ptr) - 1), ptr(col_ptr), ptr(row_idx), ptr(val), ptr(cols),
        C.c_int64(len(cols)), ptr(v), ptr(out), C.c_int32(device))
    _ffi.check(rc, "dzg_kernel_neg_t_dot_csc")
    return out[: len(cols)]


def first_pivot(y, ybar, device: int = 0) -> int:
    _ffi.require_gpu()
    y, ybar = f64(y), f64(ybar)
    out = C.c_int64(-1)
    rc = _ffi.lib().dzg_kernel_first_pivot(C.c_int64(len(y)), ptr(y), ptr(ybar), C.byref(out),
                                           C.c_int32(device))
    _ffi.check(rc, "dzg_kernel_first_pivot")
    return int(out.value)


def second_pivot(mu, y, ybar, dy, device: int = 0) -> int:
    _ffi.require_gpu()
    y, ybar, dy = f64(y), f64(ybar), f64(dy)
    out = C.c_int64(-1)
    rc = _ffi.lib().dzg_kernel_second_pivot(C.c_int64(len(y)), C.c_double(mu), ptr(y), ptr(ybar),
                                            ptr(dy), C.byref(out), C.c_int32(device))
    _ffi.check(rc, "dzg_kernel_second_pivot")
    return int(out.value)


def merge_candidates(cands) -> int:
    """cands: iterable of (ratio, pos, y, ybar, dy); pos < 0 means empty."""
    arr = (_ffi.Candidate * max(len(cands), 1))()
    for i, (r, p, y, yb, dy) in enumerate(cands):
        arr[i] = _ffi.Candidate(r, p if p >= 0 else (1 << 63) - 1, y, yb, dy)
    return int(_ffi.lib().dzg_merge_candidates(arr, C.c_int64(len(cands))))
