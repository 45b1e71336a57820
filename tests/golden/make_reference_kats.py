"""Writes tests/golden/reference_kats.json.

The numbers below are the known-answer vectors held by the reference's own
tests, transcribed as data (inputs + expected outputs).  Sources, relative to
the reference checkout:
  linalg   : src/linalg.rs:306-446      (exact equality)
  solver   : src/simplex.rs:484-796     (|r-e| <= 1e-12; core MAXIMISES, rows are <=)
  python   : tests/test_optimize.py:4-114, tests/test_exceptions.py:6-16,
             README.md:60-73            (exact == on floats)
Run:  python tests/golden/make_reference_kats.py
"""
import json
import os

NN = {"lb": 0.0, "ub": None}  # Variable::nonneg, src/pyobjs.rs:35-37


def model(vars_, obj, const, rows):
    return {
        "vars": vars_,
        "objective": {"terms": [[i, c] for i, c in obj], "constant": const},
        "constraints": [{"terms": [[i, c] for i, c in terms], "b": b} for terms, b in rows],
    }


def dense_rows(coefs, bs):
    """rows given as dense coefficient lists; zero entries are omitted like the
    reference's Inequality::new call sites omit them."""
    return [([(i, c) for i, c in enumerate(r) if c is not None], b) for r, b in zip(coefs, bs)]


_ = None
solver = [
    dict(name="nonneg_1", src="src/simplex.rs:484-501",
         model=model([NN] * 2, [(0, 4.0), (1, 3.0)], 0.0,
                     dense_rows([[1.0, -1.0], [2.0, -1.0], [_, 1.0]], [1.0, 3.0, 5.0])),
         expect=dict(status="optimal", objective=31.0, values=[4.0, 5.0])),
    dict(name="nonneg_2", src="src/simplex.rs:503-522",
         model=model([NN] * 3, [(0, 5.0), (1, 4.0), (2, 3.0)], 0.0,
                     dense_rows([[2.0, 3.0, 1.0], [4.0, 1.0, 2.0], [3.0, 4.0, 2.0]],
                                [5.0, 11.0, 8.0])),
         expect=dict(status="optimal", objective=13.0, values=[2.0, 0.0, 1.0])),
    dict(name="nonneg_3", src="src/simplex.rs:524-562",
         model=model([NN] * 4, [(0, 300.0), (1, 90.0), (2, 400.0), (3, 150.0)], 0.0,
                     dense_rows([[35000.0, 10000.0, 25000.0, 90000.0], [4.0, 2.0, 7.0, 3.0],
                                 [1.0, 1.0, _, _], [1.0, _, _, _], [_, 1.0, _, _],
                                 [_, _, 1.0, _], [_, _, _, 1.0]],
                                [120000.0, 12.0, 1.0, 1.0, 1.0, 1.0, 1.0])),
         expect=dict(status="optimal", objective=750.0, values=[1.0, 0.0, 1.0, 1.0 / 3.0])),
    dict(name="nonneg_4", src="src/simplex.rs:564-583",
         model=model([NN] * 3, [(0, 10.0), (1, 12.0), (2, 12.0)], 0.0,
                     dense_rows([[1.0, 2.0, 2.0], [2.0, 1.0, 2.0], [2.0, 2.0, 1.0]],
                                [20.0, 20.0, 20.0])),
         expect=dict(status="optimal", objective=136.0, values=[4.0, 4.0, 4.0])),
    dict(name="nonneg_5", src="src/simplex.rs:585-602",
         model=model([NN] * 2, [(0, -1.0), (1, -1.0)], 0.0,
                     dense_rows([[-2.0, -1.0], [-2.0, 4.0], [-1.0, 3.0]], [4.0, -8.0, -7.0])),
         expect=dict(status="optimal", objective=-7.0, values=[7.0, 0.0])),
    dict(name="nonneg_6", src="src/simplex.rs:604-623",
         model=model([NN] * 3, [(0, -10.0), (1, -12.0), (2, -12.0)], 0.0,
                     dense_rows([[-1.0, -2.0, -2.0], [-2.0, -1.0, -2.0], [-2.0, -2.0, -1.0]],
                                [-20.0, -20.0, -20.0])),
         expect=dict(status="optimal", objective=-136.0, values=[4.0, 4.0, 4.0])),
    dict(name="nonneg_8", src="src/simplex.rs:625-642",
         model=model([NN] * 2, [(0, -2.0), (1, 3.0)], 0.0,
                     dense_rows([[-1.0, 1.0], [-1.0, -2.0], [_, 1.0]], [-1.0, -2.0, 1.0])),
         expect=dict(status="optimal", objective=-1.0, values=[2.0, 1.0])),
    dict(name="nonneg_9", src="src/simplex.rs:644-674",
         model=model([NN] * 6, [(1, 2.0), (4, 3.0)], 10.0,
                     dense_rows([[1.0, -1.0, _, 1.0, _, _], [-1.0, 1.0, _, -1.0, _, _],
                                 [_, 3.0, 1.0, _, -1.0, _], [_, -3.0, -1.0, _, 1.0, _],
                                 [_, 1.0, _, 1.0, 2.0, _], [_, -1.0, _, -1.0, -2.0, _],
                                 [_, 2.0, _, _, 1.0, 1.0], [_, -2.0, _, _, -1.0, -1.0]],
                                [4.0, -4.0, 12.0, -12.0, 14.0, -14.0, 13.0, -13.0])),
         expect=dict(status="optimal", objective=33.0, values=[8.0, 4.0, 5.0, 0.0, 5.0, 0.0])),
    dict(name="nonneg_no_constraints", src="src/simplex.rs:676-687",
         model=model([NN], [(0, -3.0)], 2.0, []),
         expect=dict(status="optimal", objective=2.0, values=[0.0])),
    dict(name="variable_constraints", src="src/simplex.rs:689-703",
         model=model([{"lb": 1.0, "ub": 1.0}, {"lb": -3.0, "ub": -1.0}],
                     [(0, 1.0), (1, -1.0)], 5.0, []),
         expect=dict(status="optimal", objective=9.0, values=[1.0, -3.0])),
    dict(name="unbounded_1", src="src/simplex.rs:705-720",
         model=model([NN] * 2, [(0, -1.0), (1, 4.0)], 0.0,
                     dense_rows([[-2.0, -1.0], [-2.0, 4.0], [-1.0, 3.0]], [4.0, -8.0, -7.0])),
         expect=dict(status="unbounded")),
    dict(name="unbounded_2", src="src/simplex.rs:722-734",
         model=model([NN], [(0, 1.0)], 0.0, dense_rows([[-2.0]], [-4.0])),
         expect=dict(status="unbounded")),
    dict(name="unbounded_no_constraints", src="src/simplex.rs:736-747",
         model=model([NN], [(0, 1.0)], 10.0, []),
         expect=dict(status="unbounded")),
    dict(name="infeasible_1", src="src/simplex.rs:749-763",
         model=model([NN] * 2, [(0, 1.0), (1, 1.0)], 0.0,
                     dense_rows([[1.0, _], [_, 5.0]], [-1.0, 0.5])),
         expect=dict(status="infeasible")),
    dict(name="infeasible_2", src="src/simplex.rs:765-778",
         model=model([NN] * 2, [(0, 1.0), (1, -1.0)], 0.0, dense_rows([[1.0, 1.0]], [-1.0])),
         expect=dict(status="infeasible")),
    dict(name="infeasible_3", src="src/simplex.rs:780-796",
         model=model([NN] * 2, [(0, 1.0), (1, 1.0)], 0.0,
                     dense_rows([[1.0, 1.0], [-1.0, -1.0], [1.0, 1.0], [-1.0, -1.0]],
                                [1.0, -1.0, 2.0, -2.0])),
         expect=dict(status="infeasible")),
]

linalg = dict(
    lu_factorization=dict(  # src/linalg.rs:322-345
        a=[[3.0, 17.0, 10.0], [2.0, 4.0, -2.0], [6.0, 18.0, -12.0]], p=[2, 2],
        lu=[6.0, 18.0, -12.0, 1.0 / 3.0, 8.0, 16.0, 1.0 / 2.0, -1.0 / 4.0, 6.0]),
    lu_solve=[  # src/linalg.rs:360-380
        dict(a=[[6.0, 18.0, 3.0], [2.0, 12.0, 1.0], [4.0, 15.0, 3.0]], b=[3.0, 19.0, 0.0],
             x=[-3.0, 3.0, -11.0]),
        dict(a=[[2.0, 0.0, 0.0], [4.0, 1.0, 0.0], [3.0, 0.0, 1.0]], b=[1.0, 2.0, 2.0],
             x=[0.5, 0.0, 0.5]),
    ],
    matrix_roundtrip=dict(a=[[0.0, 1.0], [0.0, 2.0]]),  # :347-358
    csc_from_dense=dict(  # :382-393
        a=[[1.0, 0.0, 2.0], [0.0, 0.0, 3.0], [4.0, 5.0, 6.0]],
        row_idx=[0, 2, 2, 0, 1, 2], col_ptr=[0, 2, 3, 6], data=[1.0, 4.0, 5.0, 2.0, 3.0, 6.0]),
    csc_column=dict(  # :395-406
        a=[[1.0, 0.0, 2.0], [0.0, 0.0, 3.0], [4.0, 5.0, 6.0]],
        columns=[[1.0, 0.0, 4.0], [0.0, 0.0, 5.0], [2.0, 3.0, 6.0]]),
    csc_collect_columns=dict(cols=[1, 2, 0]),  # :408-421 (same matrix as csc_column)
    dense_transpose=dict(a=[[1.0, 2.0], [3.0, 4.0]], t=[[1.0, 3.0], [2.0, 4.0]]),  # :423-433
    neg_transpose_dot=dict(  # :435-446  3x4 matrix 0..12 row-major
        a=[[0.0, 1.0, 2.0, 3.0], [4.0, 5.0, 6.0, 7.0], [8.0, 9.0, 10.0, 11.0]],
        v=[1.0, 2.0, 3.0], out=[-32.0, -38.0, -44.0, -50.0]),
)

# Python-surface KATs.  "vars": name -> constructor spec; expressions are evaluated
# against the modelling surface by the tests (so the lowering of ==, >=, chained
# comparisons and Minimize's negation is exercised exactly as a user would).
python = [
    dict(name="problem_1", src="tests/test_optimize.py:4-11",
         vars={"x": "nonneg", "y": "nonneg"}, sense="min", objective="2 * x - 2 * y",
         constraints=["y == 3"],
         expect=dict(objective=-6.0, values={"x": 0.0, "y": 3.0})),
    dict(name="problem_2", src="tests/test_optimize.py:14-23",
         vars={"x": "nonneg", "y": "nonneg"}, sense="min", objective="2 * x - 2 * y",
         constraints=["y <= 5", "x >= y + 1", "y == 5.0"],
         expect=dict(objective=2.0, values={"x": 6.0, "y": 5.0})),
    dict(name="problem_3", src="tests/test_optimize.py:26-35",
         vars={"x": "nonneg", "y": "nonneg", "z": "nonneg"}, sense="min",
         objective="x + y - z", constraints=["x + y + z <= 1"],
         expect=dict(objective=-1.0, values={"x": 0.0, "y": 0.0, "z": 1.0})),
    dict(name="problem_4", src="tests/test_optimize.py:38-47",
         vars={"x": "nonneg", "y": "nonneg", "z": "nonneg"}, sense="min",
         objective="x + y + z", constraints=["x - y == -2"],
         expect=dict(objective=2.0, values={"x": 0.0, "y": 2.0, "z": 0.0})),
    dict(name="min_max_equivalence_min", src="tests/test_optimize.py:50-59",
         vars={"x": "nonneg", "y": "nonneg"}, sense="min", objective="-x",
         constraints=["x + y <= 1"], expect=dict(objective=-1.0, values={"x": 1.0, "y": 0.0})),
    dict(name="min_max_equivalence_max", src="tests/test_optimize.py:50-59",
         vars={"x": "nonneg", "y": "nonneg"}, sense="max", objective="x",
         constraints=["x + y <= 1"], expect=dict(objective=1.0, values={"x": 1.0, "y": 0.0})),
    dict(name="non_standard_variables", src="tests/test_optimize.py:62-71",
         vars={"x": [-2.0, 2.0], "y": "free", "z": "nonpos"}, sense="min",
         objective="x + y + z", constraints=["y == 4", "-3.0 <= x <= 3.0", "z >= -1"],
         expect=dict(objective=1.0, values={"x": -2.0, "y": 4.0, "z": -1.0})),
    dict(name="inventory_balance", src="tests/test_optimize.py:74-114",
         vars={k: "nonneg" for k in ["x_1", "x_2", "x_3", "z_1", "z_2", "z_3"]}, sense="min",
         objective="sum(p_t * x_t for p_t, x_t in zip([0.5, 3.5, 5.0], [x_1, x_2, x_3]))"
                   " + sum(h_t * z_t for h_t, z_t in zip([1.0, 5.5, 1.5], [z_1, z_2, z_3]))",
         constraints=["x_1 >= 50", "x_2 + z_1 >= 75", "x_3 + z_2 >= 100", "z_1 == x_1 - 50",
                      "z_2 == x_2 + z_1 - 75", "z_3 == x_3 + z_2 - 100"],
         expect=dict(objective=637.5, values={"x_1": 125.0, "x_2": 0.0, "x_3": 100.0})),
    dict(name="readme_quick_start", src="README.md:60-73",
         vars={"x": "nonneg", "y": "nonneg", "z": "nonneg"}, sense="min",
         objective="x + y - z", constraints=["x + y + z == 1"],
         expect=dict(objective=-1.0, values={"x": 0.0, "y": 0.0, "z": 1.0})),
    dict(name="docstring_minimize", src="python-source/dantzig/optimize.py:92-99",
         vars={"x": [1.0, None], "y": [None, 2.0]}, sense="min", objective="x - 5 * y",
         constraints=[], expect=dict(values={"x": 1.0, "y": 2.0})),
    dict(name="docstring_maximize", src="python-source/dantzig/optimize.py:130-137",
         vars={"x": [1.0, None], "y": [None, 2.0]}, sense="max", objective="y - 5 * x",
         constraints=[], expect=dict(values={"x": 1.0, "y": 2.0})),
    dict(name="unbounded_error", src="tests/test_exceptions.py:6-9",
         vars={"x": "nonneg"}, sense="min", objective="-1.0 * x", constraints=[],
         expect=dict(error="UnboundedError")),
    dict(name="infeasible_error", src="tests/test_exceptions.py:12-16",
         vars={"x": "nonneg", "y": "nonneg"}, sense="min", objective="x + y",
         constraints=["x + y == 1", "x + y == 2"], expect=dict(error="InfeasibleError")),
]

if __name__ == "__main__":
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_kats.json")
    with open(out, "w") as f:
        json.dump(dict(linalg=linalg, solver=solver, python=python), f, indent=1)
    print("wrote", out)
