// lpgen.cpp -- host utilities of the C ABI that are not on the device path:
// the synthetic LP generator G1 of SURVEY 8(d) and the deterministic max-loc merge used
// by the column-sharded (multi-GPU) exchange.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <limits>
#include <thread>
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
//
// SplitMix64's state after k draws is seed + k*gamma, so draw k is computable on its own:
// the generator runs on several host threads (columns split for A and c, rows split for b)
// and can produce a COLUMN BLOCK of A without ever materialising the rest -- what a rank of a
// column-sharded solve needs (config 5 is 16 GiB of matrix; a rank holds 1/P of it) -- with
// results bit-identical to the sequential definition above.
namespace {
inline double draw01(uint64_t seed, uint64_t k)
{
    uint64_t z = seed + (k + 1) * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z = z ^ (z >> 31);
    return (double)(z >> 11) * 0x1.0p-53;
}

int gen_threads()
{
    if (const char *e = std::getenv("DZG_GEN_THREADS")) {
        const int t = std::atoi(e);
        if (t >= 1) return t > 64 ? 64 : t;
    }
    const unsigned hw = std::thread::hardware_concurrency();
    return hw == 0 ? 1 : (hw > 16 ? 16 : (int)hw);
}

template <class F> void parallel_ranges(int64_t count, F &&body)
{
    const int nt = (int)std::min<int64_t>(gen_threads(), count > 0 ? count : 1);
    if (nt <= 1) {
        body((int64_t)0, count);
        return;
    }
    std::vector<std::thread> pool;
    for (int t = 0; t < nt; ++t) {
        const int64_t lo = count * t / nt, hi = count * (t + 1) / nt;
        pool.emplace_back([&body, lo, hi] { body(lo, hi); });
    }
    for (auto &th : pool) th.join();
}

// a holds columns [col0, col1) (column j at a + (j - col0) * lda); b has m entries, c has ns.
int gen_dense(uint64_t seed, int64_t m, int64_t ns, int64_t col0, int64_t col1, double *a,
              int64_t lda, double *b, double *c)
{
    const uint64_t base_x0 = (uint64_t)m * (uint64_t)ns, base_y0 = base_x0 + (uint64_t)ns,
                   base_rb = base_y0 + (uint64_t)m, base_rc = base_rb + (uint64_t)m;
    std::vector<double> y0((size_t)m);
    for (int64_t i = 0; i < m; ++i) y0[(size_t)i] = draw01(seed, base_y0 + (uint64_t)i);
    // columns: store the owned ones, c_j = (column j) . y0 - rc_j accumulated in ascending i
    parallel_ranges(ns, [&](int64_t j0, int64_t j1) {
        std::vector<double> tmp((size_t)m);
        for (int64_t j = j0; j < j1; ++j) {
            const bool own = j >= col0 && j < col1;
            double *col = own ? a + (j - col0) * lda : tmp.data();
            const uint64_t k0 = (uint64_t)j * (uint64_t)m;
            for (int64_t i = 0; i < m; ++i) col[i] = 2.0 * draw01(seed, k0 + (uint64_t)i) - 1.0;
            double acc = 0.0;
            for (int64_t i = 0; i < m; ++i) {
                const double p = col[i] * y0[(size_t)i];
                acc = acc + p;
            }
            c[j] = acc - draw01(seed, base_rc + (uint64_t)j);
        }
    });
    // rows: b_i = sum_j A_ij x0_j accumulated in ascending j, then + rb_i
    std::vector<double> x0((size_t)ns);
    for (int64_t j = 0; j < ns; ++j) x0[(size_t)j] = draw01(seed, base_x0 + (uint64_t)j);
    parallel_ranges(m, [&](int64_t i0, int64_t i1) {
        for (int64_t i = i0; i < i1; ++i) b[i] = 0.0;
        for (int64_t j = 0; j < ns; ++j) {
            const double xj = x0[(size_t)j];
            if (j >= col0 && j < col1) {
                const double *col = a + (j - col0) * lda;
                for (int64_t i = i0; i < i1; ++i) {
                    const double p = col[i] * xj;
                    b[i] = b[i] + p;
                }
            } else {
                const uint64_t k0 = (uint64_t)j * (uint64_t)m;
                for (int64_t i = i0; i < i1; ++i) {
                    const double p = (2.0 * draw01(seed, k0 + (uint64_t)i) - 1.0) * xj;
                    b[i] = b[i] + p;
                }
            }
        }
        for (int64_t i = i0; i < i1; ++i) b[i] = b[i] + draw01(seed, base_rb + (uint64_t)i);
    });
    return 0;
}
} // namespace

extern "C" int dzg_gen_dense_lp(uint64_t seed, int64_t m, int64_t ns, double *a, int64_t lda,
                                double *b, double *c)
{
    if (m <= 0 || ns <= 0 || !a || !b || !c || lda < m) return DZG_E_ARG;
    return gen_dense(seed, m, ns, 0, ns, a, lda, b, c);
}

// Same LP, but only columns [col_begin, col_end) of A are produced (b and c are complete).
extern "C" int dzg_gen_dense_lp_block(uint64_t seed, int64_t m, int64_t ns, int64_t col_begin,
                                      int64_t col_end, double *a_block, int64_t lda, double *b,
                                      double *c)
{
    if (m <= 0 || ns <= 0 || !a_block || !b || !c || lda < m || col_begin < 0 || col_end > ns ||
        col_begin > col_end)
        return DZG_E_ARG;
    return gen_dense(seed, m, ns, col_begin, col_end, a_block, lda, b, c);
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
