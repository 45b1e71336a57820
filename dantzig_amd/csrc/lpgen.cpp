// lpgen.cpp -- host utilities of the C ABI that are not on the device path:
// the synthetic LP generator G1 of SURVEY 8(d) and the deterministic max-loc merge used
// by the column-sharded (multi-GPU) exchange.
#include <cmath>
#include <cstdint>
#include <limits>
#include <vector>

#include "../../include/dantzig_amd.h"

namespace {
struct SplitMix64 {
    uint64_t s;
    explicit SplitMix64(uint64_t seed) : s(seed) {}
    inline uint64_t next()
    {
        uint64_t z = (s += 0x9E3779B97F4A7C15ull);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    }
    inline double u01() { return (double)(next() >> 11) * 0x1.0p-53; } // [0,1)
};
} // namespace

// G1: A_ij = 2u-1 drawn column-major, then x0 (n_struct), y0 (m), rb (m), rc (n_struct);
// b = A x0 + rb (primal feasible, mixed-sign), c = A^T y0 - rc (dual feasible => bounded).
// Both products are plain ascending-index loops, one rounding per operation.
extern "C" int dzg_gen_dense_lp(uint64_t seed, int64_t m, int64_t ns, double *a, int64_t lda,
                                double *b, double *c)
{
    if (m <= 0 || ns <= 0 || !a || !b || !c || lda < m) return DZG_E_ARG;
    SplitMix64 g(seed);
    for (int64_t j = 0; j < ns; ++j) {
        double *col = a + j * lda;
        for (int64_t i = 0; i < m; ++i) col[i] = 2.0 * g.u01() - 1.0;
    }
    std::vector<double> x0((size_t)ns), y0((size_t)m), rb((size_t)m), rc((size_t)ns);
    for (auto &v : x0) v = g.u01();
    for (auto &v : y0) v = g.u01();
    for (auto &v : rb) v = g.u01();
    for (auto &v : rc) v = g.u01();
    for (int64_t i = 0; i < m; ++i) b[i] = 0.0;
    for (int64_t j = 0; j < ns; ++j) { // b_i accumulates in ascending j
        const double *col = a + j * lda;
        const double xj = x0[(size_t)j];
        for (int64_t i = 0; i < m; ++i) {
            const double p = col[i] * xj;
            b[i] = b[i] + p;
        }
    }
    for (int64_t i = 0; i < m; ++i) b[i] = b[i] + rb[(size_t)i];
    for (int64_t j = 0; j < ns; ++j) { // c_j accumulates in ascending i
        const double *col = a + j * lda;
        double acc = 0.0;
        for (int64_t i = 0; i < m; ++i) {
            const double p = col[i] * y0[(size_t)i];
            acc = acc + p;
        }
        c[j] = acc - rc[(size_t)j];
    }
    return 0;
}

// G2: sparse columns, `per_col` distinct rows each (rejection sampling, then sorted).
extern "C" int dzg_gen_sparse_lp(uint64_t seed, int64_t m, int64_t ns, int64_t per_col,
                                 int64_t *col_ptr, int32_t *row_idx, double *val, double *b,
                                 double *c)
{
    if (m <= 0 || ns <= 0 || per_col <= 0 || per_col > m || !col_ptr || !row_idx || !val || !b || !c)
        return DZG_E_ARG;
    SplitMix64 g(seed);
    std::vector<char> used((size_t)m, 0);
    std::vector<int32_t> rows((size_t)per_col);
    for (int64_t j = 0; j < ns; ++j) {
        col_ptr[j] = j * per_col;
        for (int64_t e = 0; e < per_col; ++e) {
            int64_t r;
            do {
                r = (int64_t)(g.u01() * (double)m);
                if (r >= m) r = m - 1;
            } while (used[(size_t)r]);
            used[(size_t)r] = 1;
            rows[(size_t)e] = (int32_t)r;
        }
        for (int64_t e = 0; e < per_col; ++e) used[(size_t)rows[(size_t)e]] = 0;
        for (int64_t e = 1; e < per_col; ++e) { // insertion sort, ascending rows
            int32_t key = rows[(size_t)e];
            int64_t f = e - 1;
            while (f >= 0 && rows[(size_t)f] > key) {
                rows[(size_t)(f + 1)] = rows[(size_t)f];
                --f;
            }
            rows[(size_t)(f + 1)] = key;
        }
        for (int64_t e = 0; e < per_col; ++e) {
            double v;
            do v = 2.0 * g.u01() - 1.0; while (v == 0.0);
            row_idx[j * per_col + e] = rows[(size_t)e];
            val[j * per_col + e] = v;
        }
    }
    col_ptr[ns] = ns * per_col;
    std::vector<double> x0((size_t)ns), y0((size_t)m), rb((size_t)m), rc((size_t)ns);
    for (auto &v : x0) v = g.u01();
    for (auto &v : y0) v = g.u01();
    for (auto &v : rb) v = g.u01();
    for (auto &v : rc) v = g.u01();
    for (int64_t i = 0; i < m; ++i) b[i] = 0.0;
    for (int64_t j = 0; j < ns; ++j)
        for (int64_t e = col_ptr[j]; e < col_ptr[j + 1]; ++e) {
            const double p = val[e] * x0[(size_t)j];
            b[row_idx[e]] = b[row_idx[e]] + p;
        }
    for (int64_t i = 0; i < m; ++i) b[i] = b[i] + rb[(size_t)i];
    for (int64_t j = 0; j < ns; ++j) {
        double acc = 0.0;
        for (int64_t e = col_ptr[j]; e < col_ptr[j + 1]; ++e) {
            const double p = val[e] * y0[(size_t)row_idx[e]];
            acc = acc + p;
        }
        c[j] = acc - rc[(size_t)j];
    }
    return 0;
}

// Largest ratio wins, lowest GLOBAL position on ties: the parallel form of the reference's
// sequential "replace only if ratio > best" scan (src/simplex.rs:432-435, :456-459).
extern "C" int64_t dzg_merge_candidates(const dzg_candidate *cands, int64_t count)
{
    int64_t best = -1;
    for (int64_t r = 0; r < count; ++r) {
        const dzg_candidate &c = cands[r];
        if (c.pos == std::numeric_limits<int64_t>::max() || c.pos < 0) continue;
        if (c.ratio != c.ratio) continue;
        if (best < 0 || c.ratio > cands[best].ratio ||
            (c.ratio == cands[best].ratio && c.pos < cands[best].pos))
            best = r;
    }
    return best;
}
