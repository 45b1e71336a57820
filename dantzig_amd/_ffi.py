"""ctypes binding of include/dantzig_amd.h (the C ABI of the HIP engine).

There is no CPU fallback: if the shared library is missing or no MI355X is visible the
calls raise, loudly.  Build the library with `python -m dantzig_amd._build` (or
`__graft_entry__.build()`); it is kept in-tree next to this file.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libdantzig_amd.so")

# dzg_status
OPTIMAL, UNBOUNDED, INFEASIBLE, ITER_LIMIT, SINGULAR, PANIC, RUNNING, NEAR_TIE = range(8)
E_DEVICE, E_ARG, E_NOMEM = -1, -2, -3
STRICT, FAST, AUTO = 0, 1, 2
PRICE_AUTO, PRICE_SEQ, PRICE_WAVE, PRICE_TREE = 0, 1, 2, 3
STEP_PRIMAL, STEP_DUAL = 0, 1
NEAR_TIE_COUNT, NEAR_TIE_STOP = 0, 1
(K_STATUS, K_FTRAN, K_RATIO, K_BTRAN, K_PRICE, K_UPDATE, K_BASIS_UPDATE, K_LU, K_XCHG1, K_XCHG2,
 K_COUNT) = range(11)
KERNEL_CLASSES = ["status", "ftran", "ratio", "btran", "price", "update", "basis_update", "lu",
                  "exchange1", "exchange2"]

EXPORTS = [
    "dzg_abi_version", "dzg_status_str", "dzg_last_error", "dzg_device_count",
    "dzg_opts_default", "dzg_solver_create", "dzg_solver_run", "dzg_solver_result",
    "dzg_solver_destroy", "dzg_core_solve", "dzg_model_solve", "dzg_build_standard_form",
    "dzg_kernel_lu_solve", "dzg_kernel_neg_t_dot", "dzg_kernel_first_pivot",
    "dzg_kernel_second_pivot", "dzg_gen_dense_lp", "dzg_gen_sparse_lp", "dzg_merge_candidates",
    "dzg_shard_record_doubles", "dzg_shard_phase1", "dzg_shard_phase2", "dzg_shard_phase3",
    "dzg_solver_poll", "dzg_solver_set_budget", "dzg_comm_unique_id", "dzg_shard_comm_init",
    "dzg_shard_run", "dzg_shard_run_lockstep", "dzg_solver_stream", "dzg_solver_refactor",
    "dzg_gen_dense_lp_block", "dzg_solver_set_profile", "dzg_kernel_neg_t_dot_csc",
    "dzg_shard_comm_size", "dzg_solver_upload_columns", "dzg_debug_hold_cus", "dzg_debug_hold_wait",
    "dzg_core_solve_full_csc", "dzg_debug_live_lists", "dzg_debug_rl_listed",
]


class Lp(C.Structure):
    _fields_ = [
        ("m", C.c_int64), ("n", C.c_int64), ("n_struct", C.c_int64),
        ("a", C.c_void_p), ("lda", C.c_int64), ("var_col", C.c_void_p),
        ("c", C.c_void_p), ("constant", C.c_double),
        ("basis", C.c_void_p), ("nonbasis", C.c_void_p), ("x", C.c_void_p), ("z", C.c_void_p),
        ("col_ptr", C.c_void_p), ("row_idx", C.c_void_p), ("val", C.c_void_p),
        ("xbar", C.c_void_p), ("zbar", C.c_void_p),
    ]


class Opts(C.Structure):
    _fields_ = [
        ("numerics", C.c_int32), ("price_kernel", C.c_int32), ("device", C.c_int32),
        ("auto_strict_rows", C.c_int32), ("max_iter", C.c_int64), ("epsilon", C.c_double),
        ("log_capacity", C.c_int64), ("poll_interval", C.c_int32), ("profile", C.c_int32),
        ("col_begin", C.c_int64), ("col_end", C.c_int64), ("rank", C.c_int32),
        ("world", C.c_int32), ("stream", C.c_void_p), ("refactor_interval", C.c_int64),
        ("a_is_block", C.c_int32), ("near_tie_action", C.c_int32),
        ("replicate_matrix", C.c_int32), ("auto_restart_rows", C.c_int32), ("tie_tol", C.c_double),
        ("seven_launches", C.c_int32), ("auto_strict_budget_s", C.c_int32),
        ("shard_rows", C.c_int32), ("reserved0", C.c_int32),
    ]


class Pivot(C.Structure):
    _fields_ = [("kind", C.c_int32), ("reserved", C.c_int32), ("entering", C.c_int64),
                ("leaving", C.c_int64), ("mu", C.c_double)]


class Result(C.Structure):
    _fields_ = [
        ("status", C.c_int32), ("numerics_used", C.c_int32), ("iterations", C.c_int64),
        ("objective", C.c_double),
        ("basis", C.c_void_p), ("nonbasis", C.c_void_p), ("x", C.c_void_p), ("xbar", C.c_void_p),
        ("z", C.c_void_p), ("zbar", C.c_void_p), ("log", C.c_void_p), ("log_cap", C.c_int64),
        ("kernel_ms", C.c_double * K_COUNT), ("kernel_launches", C.c_int64 * K_COUNT),
        ("price_bytes", C.c_double), ("solve_ms", C.c_double), ("max_pivot_error", C.c_double),
        ("near_ties", C.c_int64), ("first_near_tie", C.c_int64), ("min_margin", C.c_double),
        ("margins", C.c_void_p), ("dense_columns", C.c_int64), ("refactors", C.c_int64),
        ("chain_fallbacks", C.c_int64), ("price_pass_used", C.c_int32),
        ("price_rows_copy", C.c_int32), ("state_drift", C.c_double),
    ]


class Model(C.Structure):
    _fields_ = [
        ("nvars", C.c_int64), ("has_lb", C.c_void_p), ("has_ub", C.c_void_p),
        ("lb", C.c_void_p), ("ub", C.c_void_p),
        ("obj_nterms", C.c_int64), ("obj_var", C.c_void_p), ("obj_coef", C.c_void_p),
        ("obj_const", C.c_double),
        ("ncons", C.c_int64), ("con_ptr", C.c_void_p), ("con_var", C.c_void_p),
        ("con_coef", C.c_void_p), ("con_b", C.c_void_p),
    ]


class ModelResult(C.Structure):
    _fields_ = [("status", C.c_int32), ("numerics_used", C.c_int32), ("iterations", C.c_int64),
                ("objective", C.c_double), ("values", C.c_void_p), ("m", C.c_int64),
                ("n", C.c_int64), ("near_ties", C.c_int64), ("first_near_tie", C.c_int64)]


class StdForm(C.Structure):
    _fields_ = [
        ("m", C.c_int64), ("n", C.c_int64), ("n_struct", C.c_int64), ("lda", C.c_int64),
        ("a", C.c_void_p), ("var_col", C.c_void_p), ("c", C.c_void_p), ("constant", C.c_double),
        ("basis", C.c_void_p), ("nonbasis", C.c_void_p), ("x", C.c_void_p), ("z", C.c_void_p),
        ("pos_var", C.c_void_p), ("neg_var", C.c_void_p),
    ]


class Candidate(C.Structure):
    _fields_ = [("ratio", C.c_double), ("pos", C.c_int64), ("y", C.c_double),
                ("ybar", C.c_double), ("dy", C.c_double)]


class DantzigAmdError(RuntimeError):
    """A call into the HIP engine failed (no GPU, bad argument, out of memory)."""


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise DantzigAmdError(
                f"{LIB_PATH} is missing: build the HIP engine first "
                "(python -m dantzig_amd._build). dantzig_amd has no CPU fallback.")
        _lib = C.CDLL(LIB_PATH)
        _lib.dzg_status_str.restype = C.c_char_p
        _lib.dzg_last_error.restype = C.c_char_p
        _lib.dzg_merge_candidates.restype = C.c_int64
        _lib.dzg_solver_destroy.restype = None
        _lib.dzg_opts_default.restype = None
        _lib.dzg_gen_dense_lp.argtypes = [C.c_uint64, C.c_int64, C.c_int64, C.c_void_p, C.c_int64,
                                          C.c_void_p, C.c_void_p]
        _lib.dzg_gen_dense_lp_block.argtypes = [C.c_uint64, C.c_int64, C.c_int64, C.c_int64,
                                                C.c_int64, C.c_void_p, C.c_int64, C.c_void_p,
                                                C.c_void_p]
        _lib.dzg_solver_run.argtypes = [C.c_void_p, C.c_int64]
        _lib.dzg_solver_result.argtypes = [C.c_void_p, C.c_void_p]
        _lib.dzg_solver_destroy.argtypes = [C.c_void_p]
        _lib.dzg_shard_record_doubles.restype = C.c_int64
        _lib.dzg_shard_record_doubles.argtypes = [C.c_void_p]
        _lib.dzg_shard_phase1.argtypes = [C.c_void_p, C.c_void_p]
        _lib.dzg_shard_phase2.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        _lib.dzg_shard_phase3.argtypes = [C.c_void_p, C.c_void_p]
        _lib.dzg_solver_poll.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        _lib.dzg_solver_set_budget.argtypes = [C.c_void_p, C.c_int64]
        _lib.dzg_comm_unique_id.argtypes = [C.c_void_p]
        _lib.dzg_shard_comm_init.argtypes = [C.c_void_p, C.c_void_p]
        _lib.dzg_shard_run.argtypes = [C.c_void_p, C.c_int64]
        _lib.dzg_shard_comm_size.argtypes = [C.c_void_p]
        _lib.dzg_shard_run_lockstep.argtypes = [C.c_void_p, C.c_int32, C.c_int64]
        _lib.dzg_solver_stream.restype = C.c_void_p
        _lib.dzg_solver_stream.argtypes = [C.c_void_p]
        _lib.dzg_solver_refactor.argtypes = [C.c_void_p]
        _lib.dzg_solver_set_profile.argtypes = [C.c_void_p, C.c_int32]
        _lib.dzg_core_solve_full_csc.argtypes = [C.c_int64, C.c_int64, C.c_void_p, C.c_void_p,
                                                 C.c_void_p, C.c_void_p, C.c_double, C.c_void_p,
                                                 C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                                 C.c_void_p]
        _lib.dzg_debug_hold_cus.argtypes = [C.c_int32, C.c_int32, C.c_double]
        _lib.dzg_debug_live_lists.restype = C.c_int64
        _lib.dzg_debug_live_lists.argtypes = [C.c_void_p, C.c_void_p]
        _lib.dzg_solver_upload_columns.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p,
                                                   C.c_int64]
    return _lib


def ptr(a: np.ndarray | None) -> C.c_void_p:
    return C.c_void_p(None if a is None else a.ctypes.data)


def f64(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.float64)


def i64(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.int64)


def status_str(code: int) -> str:
    return lib().dzg_status_str(int(code)).decode()


def check(rc: int, what: str) -> int:
    """Negative return codes are call failures: raise.  Non-negative are solver outcomes."""
    if rc < 0:
        msg = lib().dzg_last_error().decode()
        raise DantzigAmdError(f"{what}: {status_str(rc)} ({msg})")
    return rc


def default_opts(**kw) -> Opts:
    o = Opts()
    lib().dzg_opts_default(C.byref(o))
    for k, v in kw.items():
        if not hasattr(o, k):
            raise TypeError(f"unknown option {k!r}")
        setattr(o, k, v)
    return o


def require_gpu() -> None:
    if lib().dzg_device_count() <= 0:
        raise DantzigAmdError("no HIP device visible: dantzig_amd runs on an MI355X only "
                              "(there is no CPU fallback)")
