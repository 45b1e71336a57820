// model.cpp -- Level 2 of the C ABI: the standard-form builder and the whole of
// `dantzig.rust.solve` (src/lib.rs:16-27) = Simplex::new + Simplex::solve + PySolution::from.
//
// Builder semantics follow Simplex::new (src/simplex.rs:123-224): every user variable is
// split x = x+ - x- in order of first appearance (objective first, then the rows); a finite
// ub adds the row x+ - x- <= ub and a finite lb the row -x+ + x- <= -lb, ub before lb,
// appended after the user rows; every row gets a slack; variables are numbered by first
// appearance over objective then rows (slack last in its row); slacks start basic with
// x = rhs, everything else nonbasic with z = -c.  Coefficients are scattered by assignment
// (the last duplicate wins, src/linalg.rs:34-36, src/simplex.rs:41-43).
//
// Unlike the reference (COO -> dense m x n row-major -> CSC, src/simplex.rs:62-81) the
// structural block is written straight into the column-major layout the GPU consumes and
// slack columns are never materialised.
#include <cmath>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/dantzig_amd.h"

namespace {

struct Row {
    std::vector<int64_t> id; // internal ids: 2*ord = x+, 2*ord+1 = x-, slack0 + r = slack
    std::vector<double> coef;
    double b = 0.0;
};

struct Built {
    int64_t m = 0, n = 0, ns = 0;
    std::vector<double> a; // column-major m x ns, lda = m (dense mode)
    bool sparse = false;   // large, sparse models: structural block kept CSC, never densified
    std::vector<int64_t> col_ptr;
    std::vector<int32_t> row_idx;
    std::vector<double> val;
    std::vector<int64_t> var_col, basis, nonbasis, pos_var, neg_var;
    std::vector<double> c, x, z;
    double constant = 0.0;
};

bool valid(const dzg_model *md)
{
    if (!md || md->nvars < 0 || md->obj_nterms < 0 || md->ncons < 0) return false;
    if (md->nvars > 0 && (!md->has_lb || !md->has_ub || !md->lb || !md->ub)) return false;
    if (md->obj_nterms > 0 && (!md->obj_var || !md->obj_coef)) return false;
    if (md->ncons > 0 && (!md->con_ptr || !md->con_b)) return false;
    for (int64_t t = 0; t < md->obj_nterms; ++t)
        if (md->obj_var[t] < 0 || md->obj_var[t] >= md->nvars) return false;
    if (md->ncons > 0) {
        if (md->con_ptr[0] != 0) return false;
        for (int64_t r = 0; r < md->ncons; ++r)
            if (md->con_ptr[r + 1] < md->con_ptr[r]) return false;
        const int64_t nt = md->con_ptr[md->ncons];
        if (nt > 0 && (!md->con_var || !md->con_coef)) return false;
        for (int64_t e = 0; e < nt; ++e)
            if (md->con_var[e] < 0 || md->con_var[e] >= md->nvars) return false;
    }
    return true;
}

void build(const dzg_model *md, Built &out, bool allow_sparse)
{
    const int64_t V = md->nvars;
    std::vector<int64_t> ord((size_t)V, -1);
    int64_t nseen = 0;
    std::vector<Row> bound_rows;
    auto see = [&](int64_t u) {
        if (ord[(size_t)u] >= 0) return;
        ord[(size_t)u] = nseen++;
        const int64_t pos = 2 * ord[(size_t)u], neg = pos + 1;
        if (md->has_ub[u]) { // src/simplex.rs:141-144
            Row r;
            r.id = {pos, neg};
            r.coef = {1.0, -1.0};
            r.b = md->ub[u];
            bound_rows.push_back(std::move(r));
        }
        if (md->has_lb[u]) { // :145-148
            Row r;
            r.id = {pos, neg};
            r.coef = {-1.0, 1.0};
            r.b = -md->lb[u];
            bound_rows.push_back(std::move(r));
        }
    };
    for (int64_t t = 0; t < md->obj_nterms; ++t) see(md->obj_var[t]);
    const int64_t nterms = md->ncons ? md->con_ptr[md->ncons] : 0;
    for (int64_t e = 0; e < nterms; ++e) see(md->con_var[e]);

    std::vector<Row> rows((size_t)md->ncons);
    for (int64_t r = 0; r < md->ncons; ++r) {
        Row &row = rows[(size_t)r];
        for (int64_t e = md->con_ptr[r]; e < md->con_ptr[r + 1]; ++e) {
            const int64_t o = ord[(size_t)md->con_var[e]];
            row.id.push_back(2 * o);
            row.coef.push_back(md->con_coef[e]);
            row.id.push_back(2 * o + 1);
            row.coef.push_back(-md->con_coef[e]);
        }
        row.b = md->con_b[r];
    }
    for (Row &r : bound_rows) rows.push_back(std::move(r));

    const int64_t m = (int64_t)rows.size();
    const int64_t ns = 2 * nseen, slack0 = ns, n = ns + m;
    for (int64_t r = 0; r < m; ++r) { // slack injected last, src/simplex.rs:19-31
        rows[(size_t)r].id.push_back(slack0 + r);
        rows[(size_t)r].coef.push_back(1.0);
    }

    // variable index = order of first appearance, src/simplex.rs:168-176
    std::vector<int64_t> index_of((size_t)n, -1), id_of((size_t)n, -1);
    int64_t next = 0;
    auto touch = [&](int64_t id) {
        if (index_of[(size_t)id] < 0) {
            index_of[(size_t)id] = next;
            id_of[(size_t)next++] = id;
        }
    };
    for (int64_t t = 0; t < md->obj_nterms; ++t) {
        const int64_t o = ord[(size_t)md->obj_var[t]];
        touch(2 * o);
        touch(2 * o + 1);
    }
    for (const Row &r : rows)
        for (int64_t id : r.id) touch(id);

    out.m = m;
    out.n = n;
    out.ns = ns;
    out.constant = md->obj_const;
    out.c.assign((size_t)n, 0.0);
    for (int64_t t = 0; t < md->obj_nterms; ++t) { // Objective::new, assignment
        const int64_t o = ord[(size_t)md->obj_var[t]];
        out.c[(size_t)index_of[(size_t)(2 * o)]] = md->obj_coef[t];
        out.c[(size_t)index_of[(size_t)(2 * o + 1)]] = -md->obj_coef[t];
    }
    out.var_col.resize((size_t)n);
    for (int64_t i = 0; i < n; ++i) {
        const int64_t id = id_of[(size_t)i];
        if (id >= slack0) {
            out.var_col[(size_t)i] = -1 - (id - slack0);
            out.basis.push_back(i);
            out.x.push_back(rows[(size_t)(id - slack0)].b);
        } else {
            out.var_col[(size_t)i] = id; // structural column = internal id
            out.nonbasis.push_back(i);
            out.z.push_back(-out.c[(size_t)i]);
        }
    }
    // dense when small or dense; CSC when the m x ns block would be large and mostly zero
    // (the reference densifies to m x n regardless, src/simplex.rs:62-81)
    int64_t nterms_total = 0;
    for (const Row &row : rows) nterms_total += (int64_t)row.id.size() - 1;
    out.sparse = allow_sparse && m * ns >= (int64_t)1 << 22 && nterms_total * 4 < m * ns;
    if (!out.sparse) {
        out.a.assign((size_t)(m * ns > 0 ? m * ns : 1), 0.0);
        for (int64_t r = 0; r < m; ++r) {
            const Row &row = rows[(size_t)r];
            for (size_t e = 0; e + 1 < row.id.size(); ++e) // all but the slack
                out.a[(size_t)(row.id[e] * m + r)] = row.coef[e];
        }
    } else {
        // rows are visited in ascending order, so every column's entries come out row-ascending;
        // a variable repeated inside one row keeps its LAST coefficient (assignment semantics)
        std::vector<int64_t> cnt((size_t)ns + 1, 0);
        std::vector<int64_t> last_row((size_t)ns, -1);
        for (int64_t r = 0; r < m; ++r) {
            const Row &row = rows[(size_t)r];
            for (size_t e = 0; e + 1 < row.id.size(); ++e)
                if (last_row[(size_t)row.id[e]] != r) {
                    last_row[(size_t)row.id[e]] = r;
                    ++cnt[(size_t)row.id[e] + 1];
                }
        }
        for (int64_t j = 0; j < ns; ++j) cnt[(size_t)j + 1] += cnt[(size_t)j];
        out.col_ptr = cnt;
        out.row_idx.assign((size_t)cnt[(size_t)ns] + 1, 0);
        out.val.assign((size_t)cnt[(size_t)ns] + 1, 0.0);
        std::vector<int64_t> fill(cnt.begin(), cnt.end() - 1);
        std::fill(last_row.begin(), last_row.end(), -1);
        std::vector<int64_t> slot_of((size_t)ns, -1);
        for (int64_t r = 0; r < m; ++r) {
            const Row &row = rows[(size_t)r];
            for (size_t e = 0; e + 1 < row.id.size(); ++e) {
                const int64_t j = row.id[e];
                if (last_row[(size_t)j] != r) {
                    last_row[(size_t)j] = r;
                    slot_of[(size_t)j] = fill[(size_t)j]++;
                    out.row_idx[(size_t)slot_of[(size_t)j]] = (int32_t)r;
                }
                out.val[(size_t)slot_of[(size_t)j]] = row.coef[e];
            }
        }
    }
    out.pos_var.assign((size_t)V, -1);
    out.neg_var.assign((size_t)V, -1);
    for (int64_t u = 0; u < V; ++u)
        if (ord[(size_t)u] >= 0) {
            out.pos_var[(size_t)u] = index_of[(size_t)(2 * ord[(size_t)u])];
            out.neg_var[(size_t)u] = index_of[(size_t)(2 * ord[(size_t)u] + 1)];
        }
}

} // namespace

extern "C" int dzg_build_standard_form(const dzg_model *md, dzg_stdform *out)
{
    if (!out || !valid(md)) return DZG_E_ARG;
    Built b;
    build(md, b, false);
    if (!out->a && !out->var_col && !out->c) { // sizing call
        out->m = b.m;
        out->n = b.n;
        out->n_struct = b.ns;
        out->lda = b.m > 0 ? b.m : 1;
        out->constant = b.constant;
        return 0;
    }
    if (out->m != b.m || out->n != b.n || out->n_struct != b.ns || out->lda < b.m) return DZG_E_ARG;
    for (int64_t j = 0; j < b.ns; ++j)
        for (int64_t i = 0; i < b.m; ++i) out->a[j * out->lda + i] = b.a[(size_t)(j * b.m + i)];
    std::memcpy(out->var_col, b.var_col.data(), sizeof(int64_t) * (size_t)b.n);
    std::memcpy(out->c, b.c.data(), sizeof(double) * (size_t)b.n);
    out->constant = b.constant;
    if (b.m) {
        std::memcpy(out->basis, b.basis.data(), sizeof(int64_t) * (size_t)b.m);
        std::memcpy(out->x, b.x.data(), sizeof(double) * (size_t)b.m);
    }
    if (b.n - b.m) {
        std::memcpy(out->nonbasis, b.nonbasis.data(), sizeof(int64_t) * (size_t)(b.n - b.m));
        std::memcpy(out->z, b.z.data(), sizeof(double) * (size_t)(b.n - b.m));
    }
    if (md->nvars) {
        std::memcpy(out->pos_var, b.pos_var.data(), sizeof(int64_t) * (size_t)md->nvars);
        std::memcpy(out->neg_var, b.neg_var.data(), sizeof(int64_t) * (size_t)md->nvars);
    }
    return 0;
}

extern "C" int dzg_model_solve(const dzg_model *md, const dzg_opts *opts, dzg_model_result *res)
{
    if (!res || !valid(md)) return DZG_E_ARG;
    Built b;
    build(md, b, true);
    res->m = b.m;
    res->n = b.n;
    dzg_lp lp;
    std::memset(&lp, 0, sizeof(lp));
    lp.m = b.m;
    lp.n = b.n;
    lp.n_struct = b.ns;
    lp.a = b.sparse ? nullptr : b.a.data();
    lp.lda = b.m > 0 ? b.m : 1;
    if (b.sparse) {
        lp.col_ptr = b.col_ptr.data();
        lp.row_idx = b.row_idx.data();
        lp.val = b.val.data();
    }
    lp.var_col = b.var_col.data();
    lp.c = b.c.data();
    lp.constant = b.constant;
    lp.basis = b.basis.data();
    lp.nonbasis = b.nonbasis.data();
    lp.x = b.x.data();
    lp.z = b.z.data();
    std::vector<int64_t> basis((size_t)(b.m ? b.m : 1));
    std::vector<double> x((size_t)(b.m ? b.m : 1));
    dzg_result r;
    std::memset(&r, 0, sizeof(r));
    r.basis = basis.data();
    r.x = x.data();
    const int rc = dzg_core_solve(&lp, opts, &r); // AUTO: falls back to STRICT if FAST gives up
    res->status = rc < 0 ? rc : r.status;
    res->numerics_used = r.numerics_used;
    res->iterations = r.iterations;
    res->objective = r.objective;
    res->near_ties = r.near_ties;
    res->first_near_tie = r.first_near_tie;
    if (rc < 0) return rc;
    if (res->values) { // Simplex::solution, src/simplex.rs:354-371
        std::vector<int64_t> pos_of((size_t)(b.n ? b.n : 1), -1);
        for (int64_t p = 0; p < b.m; ++p) pos_of[(size_t)basis[(size_t)p]] = p;
        for (int64_t u = 0; u < md->nvars; ++u) {
            if (b.pos_var[(size_t)u] < 0) {
                res->values[u] = 0.0; // unknown variable, src/pyobjs.rs:163-165
                continue;
            }
            const int64_t pp = pos_of[(size_t)b.pos_var[(size_t)u]];
            const int64_t pn = pos_of[(size_t)b.neg_var[(size_t)u]];
            const double pos = pp >= 0 ? x[(size_t)pp] : 0.0;
            const double neg = pn >= 0 ? x[(size_t)pn] : 0.0;
            res->values[u] = pos - neg;
        }
    }
    return r.status;
}
