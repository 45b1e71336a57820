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
