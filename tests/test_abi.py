"""The C-ABI library loads on a CPU-only host and exports every entry point that
include/dantzig_amd.h declares; host-only entry points work without a GPU; device entry points
fail loudly (no CPU fallback)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from dantzig_amd import _ffi, core

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    text = open(os.path.join(ROOT, "include", "dantzig_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(dzg_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_what_the_binding_expects():
    names = declared_functions()
    assert len(names) >= 25
    assert set(_ffi.EXPORTS) <= set(names)


@pytest.mark.parametrize("name", declared_functions())
def test_symbol_is_exported(name):
    assert hasattr(_ffi.lib(), name), f"{name} is declared in the header but not exported"


def test_abi_version_and_status_strings():
    lib = _ffi.lib()
    assert lib.dzg_abi_version() == 4
    assert _ffi.status_str(0) == "optimal" and _ffi.status_str(1) == "unbounded"
    assert _ffi.status_str(2) == "infeasible" and _ffi.status_str(-1) == "device_error"
    assert _ffi.status_str(7) == "near_tie"
    o = _ffi.default_opts()
    assert (o.numerics, o.price_kernel, o.auto_strict_rows, o.poll_interval) == (2, 0, 192, 32)
    assert o.epsilon == 1e-12 and o.max_iter == 10_000_000 and o.refactor_interval == 0


def test_struct_layouts_match_the_header(tmp_path):
    """sizeof / offsetof as gcc lays the header's structs out vs the ctypes mirrors: a drifted
    mirror would corrupt arguments silently."""
    import subprocess

    structs = {"dzg_lp": _ffi.Lp, "dzg_opts": _ffi.Opts, "dzg_pivot": _ffi.Pivot,
               "dzg_result": _ffi.Result, "dzg_model": _ffi.Model,
               "dzg_model_result": _ffi.ModelResult, "dzg_stdform": _ffi.StdForm,
               "dzg_candidate": _ffi.Candidate}
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "dantzig_amd.h"',
             'int main(void) {']
    for cname, mirror in structs.items():
        lines.append(f'printf("{cname} %zu\\n", sizeof({cname}));')
        for field, _ in mirror._fields_:
            lines.append(f'printf("{cname}.{field} %zu\\n", offsetof({cname}, {field}));')
    lines += ['return 0; }']
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    got = dict(line.split() for line in subprocess.check_output([str(exe)], text=True).splitlines())
    for cname, mirror in structs.items():
        assert int(got[cname]) == C.sizeof(mirror), cname
        for field, _ in mirror._fields_:
            assert int(got[f"{cname}.{field}"]) == getattr(mirror, field).offset, f"{cname}.{field}"


def test_rust_mirror_in_integration_md_matches_the_header():
    """INTEGRATION.md's `#[repr(C)]` mirrors are hand-written: their array length and field order are
    tied to the header here (the header itself pins DZG_K_COUNT with a static assertion)."""
    import re

    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    assert f"pub const DZG_K_COUNT: usize = {_ffi.K_COUNT};" in text
    header = open(os.path.join(ROOT, "include", "dantzig_amd.h")).read()
    assert "_Static_assert(DZG_K_COUNT == 10" in header and _ffi.K_COUNT == 10
    for rust_name, mirror in (("DzgResult", _ffi.Result), ("DzgOpts", _ffi.Opts)):
        body = text[text.index(f"pub struct {rust_name} {{"):]
        body = body[:body.index("\n}")]
        fields = re.findall(r"pub (\w+):", body)
        assert fields == [name for name, _ in mirror._fields_], rust_name
    # the one call of seam A takes the arguments in the header's order
    sig = header[header.index("int dzg_core_solve_full_csc("):]
    sig = sig[:sig.index(";")]
    c_args = re.findall(r"(\w+)(?:,|\))", sig)
    rust = text[text.index("pub fn dzg_core_solve_full_csc("):]
    rust = rust[:rust.index("-> c_int")]
    assert re.findall(r"(\w+):", rust) == c_args


def test_full_csc_entry_validates_its_arguments_on_the_host():
    """dzg_core_solve_full_csc (Level 1 on the reference's own Simplex fields) checks the CSC before
    anything reaches a device: malformed input is DZG_E_ARG with a reason on any machine; well-formed
    input without a GPU is DZG_E_DEVICE (there is no CPU path)."""
    import ctypes as C

    lib = _ffi.lib()
    i64, f64 = _ffi.i64, _ffi.f64
    # 2 rows, 3 columns: one structural column + the two slacks, as Simplex::new would leave them
    col_ptr, row_idx, val = i64([0, 2, 3, 4]), i64([0, 1, 0, 1]), f64([2.0, 1.0, 1.0, 1.0])
    c, basis, nonbasis = f64([1.0, 0.0, 0.0]), i64([1, 2]), i64([0])
    x, z = f64([4.0, 3.0]), f64([-1.0])
    res = _ffi.Result()

    def call(cp=col_ptr, ri=row_idx, m=2, n=3):
        bb, nn, xx, zz = basis.copy(), nonbasis.copy(), x.copy(), z.copy()   # (in/out arguments)
        return lib.dzg_core_solve_full_csc(m, n, _ffi.ptr(cp), _ffi.ptr(ri), _ffi.ptr(val), _ffi.ptr(c), 0.0,
                                           _ffi.ptr(bb), _ffi.ptr(nn), _ffi.ptr(xx), _ffi.ptr(zz), None,
                                           C.byref(res))

    assert call(cp=i64([0, 2, 1, 4])) == _ffi.E_ARG and b"monotone" in lib.dzg_last_error()
    assert call(ri=i64([1, 0, 0, 1])) == _ffi.E_ARG and b"ascend" in lib.dzg_last_error()
    assert call(ri=i64([0, 2, 0, 1])) == _ffi.E_ARG                      # row index out of range
    assert call(m=4, n=3) == _ffi.E_ARG                                     # n < m
    assert lib.dzg_core_solve_full_csc(2, 3, _ffi.ptr(col_ptr), _ffi.ptr(row_idx), _ffi.ptr(val), _ffi.ptr(c),
                                       0.0, _ffi.ptr(basis), _ffi.ptr(nonbasis), _ffi.ptr(x), _ffi.ptr(z),
                                       None, None) == _ffi.E_ARG            # res is NULL
    if lib.dzg_device_count() == 0:
        assert call() == _ffi.E_DEVICE and b"no CPU path" in lib.dzg_last_error()


def test_generators_are_deterministic_host_code():
    a1, b1, c1 = core.gen_dense_lp(seed=5, m=7, n_struct=11)
    a2, b2, c2 = core.gen_dense_lp(seed=5, m=7, n_struct=11)
    assert np.array_equal(a1, a2) and np.array_equal(b1, b2) and np.array_equal(c1, c2)
    assert a1.shape == (7, 11) and np.abs(a1).max() < 1.0
    # b = A x0 + rb with x0, rb in [0,1): recompute in the same ascending order
    assert np.all(np.isfinite(b1)) and np.all(np.isfinite(c1))
    cp, ri, val, b, c = core.gen_sparse_lp(9, 20, 30, 4)
    assert cp.tolist() == list(range(0, 121, 4)) and len(ri) == 120
    for j in range(30):
        rows = ri[cp[j]:cp[j + 1]]
        assert np.all(np.diff(rows) > 0) and rows.min() >= 0 and rows.max() < 20
    assert np.all(val != 0.0)


def test_device_entry_points_fail_loudly_without_gpu():
    if _ffi.lib().dzg_device_count() > 0:
        pytest.skip("a GPU is visible")
    a, b, c = core.gen_dense_lp(seed=1, m=4, n_struct=6)
    with pytest.raises(_ffi.DantzigAmdError):
        core.solve(core.CoreLP.from_inequality_form(a, b, c))
    with pytest.raises(_ffi.DantzigAmdError):
        core.lu_solve(np.eye(3), np.ones(3))
    with pytest.raises(_ffi.DantzigAmdError):
        core.neg_t_dot(np.eye(3), [0, 1], np.ones(3))


def test_malformed_lps_are_rejected_before_any_device_work():
    a, b, c = core.gen_dense_lp(seed=1, m=4, n_struct=6)
    lp = core.CoreLP.from_inequality_form(a, b, c)
    bad = core.CoreLP(a=lp.a, c=lp.c, basis=lp.basis.copy(), nonbasis=lp.nonbasis.copy(),
                      x=lp.x, z=lp.z)
    bad.basis[0] = bad.nonbasis[0]  # not a partition of 0..n-1 any more
    k = dict(a=np.ascontiguousarray(np.asarray(bad.a).T), c=_ffi.f64(bad.c),
             basis=_ffi.i64(bad.basis), nonbasis=_ffi.i64(bad.nonbasis), x=_ffi.f64(bad.x),
             z=_ffi.f64(bad.z))
    p = _ffi.ptr
    c_lp = _ffi.Lp(4, 10, 6, p(k["a"]), 4, None, p(k["c"]), 0.0, p(k["basis"]), p(k["nonbasis"]),
                   p(k["x"]), p(k["z"]), None, None, None)
    h = C.c_void_p(None)
    rc = _ffi.lib().dzg_solver_create(C.byref(c_lp), None, C.byref(h))
    assert rc == _ffi.E_ARG and not h.value
    assert b"partition" in _ffi.lib().dzg_last_error()
    c_lp.lda = 3  # lda < m
    k["basis"][0] = 6
    assert _ffi.lib().dzg_solver_create(C.byref(c_lp), None, C.byref(h)) == _ffi.E_ARG


def _g1_sequential(seed, m, ns):
    """Generator G1 written out draw by draw: the definition the threaded,
    seekable C++ generator must reproduce bit for bit."""
    mask = (1 << 64) - 1
    state = seed

    def u01():
        nonlocal state
        state = (state + 0x9E3779B97F4A7C15) & mask
        z = state
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & mask
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & mask
        z ^= z >> 31
        return (z >> 11) * 2.0 ** -53

    a = np.empty((m, ns))
    for j in range(ns):
        for i in range(m):
            a[i, j] = 2.0 * u01() - 1.0
    x0 = [u01() for _ in range(ns)]
    y0 = [u01() for _ in range(m)]
    rb = [u01() for _ in range(m)]
    rc = [u01() for _ in range(ns)]
    b = np.zeros(m)
    for j in range(ns):
        for i in range(m):
            b[i] = b[i] + a[i, j] * x0[j]
    b = np.array([b[i] + rb[i] for i in range(m)])
    c = np.empty(ns)
    for j in range(ns):
        acc = 0.0
        for i in range(m):
            acc = acc + a[i, j] * y0[i]
        c[j] = acc - rc[j]
    return a, b, c


@pytest.mark.parametrize("threads", ["1", "3", "16"])
def test_dense_generator_matches_its_sequential_definition(threads, monkeypatch):
    monkeypatch.setenv("DZG_GEN_THREADS", threads)
    m, ns, seed = 13, 29, 77
    a, b, c = core.gen_dense_lp(seed=seed, m=m, n_struct=ns)
    ea, eb, ec = _g1_sequential(seed, m, ns)
    assert np.array_equal(a, ea) and np.array_equal(b, eb) and np.array_equal(c, ec)
    # a column block is that slice of the same LP, with complete b and c
    for begin, end in [(0, ns), (0, 10), (10, 19), (19, 29), (7, 7)]:
        ab, bb, cb = core.gen_dense_lp_block(seed, m, ns, begin, end)
        assert ab.shape == (m, end - begin)
        assert np.array_equal(ab, ea[:, begin:end])
        assert np.array_equal(bb, eb) and np.array_equal(cb, ec)


def _build_c_host(tmp_path):
    """gcc + include/dantzig_amd.h + the shared library: no Python, no torch in between."""
    import subprocess

    exe = tmp_path / "readme_lp"
    libdir = os.path.join(ROOT, "dantzig_amd")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-I",
                           os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "c_host", "readme_lp.c"), "-o", str(exe),
                           "-L", libdir, "-ldantzig_amd", f"-Wl,-rpath,{libdir}"])
    return subprocess.run([str(exe)], capture_output=True, text=True, timeout=300)


def test_plain_c_host_links_and_fails_loudly_without_gpu(tmp_path):
    if _ffi.lib().dzg_device_count() > 0:
        pytest.skip("a GPU is visible: covered by the gpu-marked variant")
    run = _build_c_host(tmp_path)
    assert run.returncode == 3
    assert "abi 4 devices 0" in run.stdout
    assert "device_error" in run.stderr and "no CPU path" in run.stderr


@pytest.mark.gpu
def test_plain_c_host_solves_the_readme_lp(tmp_path):
    """The drop-in boundary used from C: Level 2 on the reference README's LP (expected
    objective -1.0 at (0, 0, 1), README.md:60-73), Level 1 against the oracle."""
    from oracle import oracle as ora

    run = _build_c_host(tmp_path)
    assert run.returncode == 0, run.stderr
    lines = {ln.split()[0]: dict(kv.split("=") for kv in ln.split()[1:])
             for ln in run.stdout.splitlines() if ln.startswith(("level", "fullcsc"))}
    l2 = lines["level2"]
    assert l2["status"] == "optimal"
    assert (float(l2["objective"]), float(l2["x"]), float(l2["y"]), float(l2["z"])) == (-1.0, 0.0, 0.0, 1.0)
    a, b, c = core.gen_dense_lp(seed=7, m=24, n_struct=40)
    want = ora.simplex_solve(ora.stdform_from_dense(a, b, c))
    l1 = lines["level1"]
    assert l1["status"] == want.status == "optimal"
    assert int(l1["iterations"]) == want.iterations
    assert float(l1["objective"]) == want.objective           # STRICT: bit-identical
    k, e, l, _ = want.pivots[0]
    assert l1["first_pivot"] == f"{k}:{e}:{l}"
    # the README LP through dzg_core_solve_full_csc, in the reference's own storage: same optimum,
    # same pivot count as Level 2 took
    fc = lines["fullcsc"]
    assert fc["status"] == "optimal" and fc["iterations"] == l2["iterations"]
    assert (float(fc["objective"]), float(fc["x"]), float(fc["y"]), float(fc["z"])) == (-1.0, 0.0, 0.0, 1.0)


def test_reference_package_name_resolves_to_this_implementation():
    """`import dantzig`, `dantzig.rust`, `from dantzig.model import ...` (the names the reference's
    callers use: python-source/dantzig/model.py:5, optimize.py:4) are this repo's objects."""
    import dantzig
    import dantzig.exceptions
    import dantzig.rust
    import dantzig_amd
    from dantzig import rust as rs
    from dantzig.model import Variable

    assert dantzig.rust is dantzig_amd.rust is rs
    assert Variable is dantzig_amd.Variable is dantzig.Var
    assert dantzig.Minimize is dantzig_amd.Minimize and dantzig.Max is dantzig_amd.Maximize
    assert dantzig.exceptions.UnboundedError is dantzig_amd.exceptions.UnboundedError
    assert set(dantzig.__all__) == {"Variable", "Var", "Minimize", "Min", "Maximize", "Max",
                                    "exceptions"}
    for name in ("Variable", "PyLinExpr", "PyAffExpr", "PyInequality", "PySolution", "solve"):
        assert hasattr(dantzig.rust, name)          # src/lib.rs:29-38


def test_resumed_from_takes_a_result_or_an_archive_of_one():
    """core.resumed_from (host only): the six state arrays -- from a CoreResult-like object or from a
    mapping such as an .npz archive -- replace the LP's start; matrix and objective stay."""
    import io
    import types

    from dantzig_amd import core

    a = np.arange(6, dtype=np.float64).reshape(2, 3)
    lp = core.CoreLP.from_inequality_form(a, np.array([1.0, 2.0]), np.array([1.0, 1.0, 1.0]))
    assert lp.xbar is None and lp.zbar is None  # Simplex::new's ones are the library's default
    state = dict(basis=np.array([0, 4]), nonbasis=np.array([1, 2, 3]), x=np.array([0.5, 0.25]),
                 xbar=np.array([0.1, 1.0]), z=np.array([1.0, 2.0, 3.0]), zbar=np.array([1.0, 0.2, 1.0]))
    for src in (types.SimpleNamespace(**state), state):
        got = core.resumed_from(lp, src)
        for name, want in state.items():
            assert np.array_equal(getattr(got, name), want), name
        assert got.a is lp.a and got.c is lp.c and got.m == 2 and got.n == 5
        assert got.basis.dtype == np.int64 and got.xbar.dtype == np.float64
    buf = io.BytesIO()
    np.savez(buf, **state)
    buf.seek(0)
    got = core.resumed_from(lp, np.load(buf))
    assert np.array_equal(got.zbar, state["zbar"]) and np.array_equal(got.basis, state["basis"])
    assert lp.xbar is None  # the original LP is untouched
