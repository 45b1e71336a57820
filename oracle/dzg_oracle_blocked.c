/*
 * dzg_oracle_blocked.c -- Matrix::factorize (src/linalg.rs:88-128) with the SAME operations on
 * every element in the SAME order as ora_lu_factorize (dzg_oracle.c), rearranged in space and
 * time so that a factorisation of 8192 rows takes seconds instead of minutes on the CPU.
 * TEST INFRASTRUCTURE ONLY (see dzg_oracle.h): it exists to write oracle pivot logs at
 * BASELINE.json's benchmark size (tests/golden/make_oracle_first_pivots.py --blocked); the
 * literal restatement stays the arbiter and tests/test_oracle_kats.py holds the two bit-equal
 * (packed factors and pivot vector) on random, integer, tied and singular matrices.
 *
 * Why the result is bit-identical.  In the reference, element a(i,j) receives, for k = 0, 1, ...
 * in ascending order: the row swap of step k if i is k or p[k] (columns j >= k only), then
 * `a(i,j) -= a(i,k) * a(k,j)` (two roundings) if i > k, j > k and the pivot of step k is not
 * zero.  The operands are final when they are used: a(i,k) is last written at step k (the
 * division by the pivot; later steps touch columns > k only) and row k is final after its own
 * step.  Nothing else orders the work, so steps may be applied to one block of columns at a
 * time (right-looking blocked LU): first the 64 steps of a panel on the panel's own columns,
 * then the same 64 steps, in order, on every block of 32 columns to its right -- independent
 * blocks, one OpenMP task each.  A block is copied into a contiguous buffer that fits the L2
 * cache while its 64 steps run.
 *
 * Build: gcc -O3 -ffp-contract=off -fno-fast-math -fopenmp (no FMA contraction: the product
 * and the subtraction round separately, as in Rust).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define NB 64 /* steps per panel  */
#define JW 32 /* columns per right-hand block */

__attribute__((target_clones("avx2", "default")))
void ora_lu_factorize_blocked(double *a, int64_t n, int64_t *p)
{
    if (n < 2) return;
    double *panel = (double *)malloc(sizeof(double) * (size_t)n * NB);
    unsigned char *pz = (unsigned char *)malloc((size_t)NB);
    for (int64_t kb = 0; kb + 1 < n; kb += NB) {
        const int64_t pe = kb + NB < n ? kb + NB : n;         /* panel columns [kb, pe)   */
        const int64_t ke = pe < n - 1 ? pe : n - 1;           /* steps [kb, ke)           */
        const int64_t pw = pe - kb, rows = n - kb;
        /* panel rows kb..n-1, contiguous: panel[(i - kb) * pw + (j - kb)] */
#pragma omp parallel for schedule(static)
        for (int64_t i = kb; i < n; ++i)
            memcpy(panel + (i - kb) * pw, a + i * n + kb, sizeof(double) * (size_t)pw);
        for (int64_t k = kb; k < ke; ++k) {
            const int64_t kk = k - kb;
            /* :98-105 first row i >= k maximising |a(i,k)|, strict '>' */
            int64_t mu = k;
            double magnitude = fabs(panel[kk * pw + kk]);
            for (int64_t i = k + 1; i < n; ++i) {
                const double v = fabs(panel[(i - kb) * pw + kk]);
                if (v > magnitude) {
                    mu = i;
                    magnitude = v;
                }
            }
            /* :107-113 swap rows k and mu, columns j >= k: the panel's share */
            for (int64_t jj = kk; jj < pw; ++jj) {
                const double t = panel[(mu - kb) * pw + jj];
                panel[(mu - kb) * pw + jj] = panel[kk * pw + jj];
                panel[kk * pw + jj] = t;
            }
            p[k] = mu;
            const double pivot = panel[kk * pw + kk];
            pz[kk] = pivot == 0.0;
            if (pivot != 0.0) { /* :116-125 */
                const double *rk = panel + kk * pw;
#pragma omp parallel for schedule(static) if (rows - kk > 512)
                for (int64_t i = k + 1; i < n; ++i) {
                    double *ri = panel + (i - kb) * pw;
                    ri[kk] /= pivot;
                    const double lik = ri[kk];
                    for (int64_t jj = kk + 1; jj < pw; ++jj) {
                        const double adjustment = lik * rk[jj];
                        ri[jj] -= adjustment;
                    }
                }
            }
        }
        /* The same steps on the columns right of the panel, one block of JW columns at a time.
         * Rows that a swap of this panel touches (positions kb..ke-1 and the pivot rows p[k]:
         * at most 2 NB rows, the set T) change position between steps and take the steps one
         * by one, in order, in a small buffer that holds "the content of position T[t]".  Every
         * other row below the panel keeps its position for all NB steps and needs nothing but
         * the finished rows kb..ke-1: its NB updates are applied in one visit, k ascending,
         * while the row's JW values stay in registers. */
        const int64_t nsteps = ke - kb;
        int64_t *tpos = (int64_t *)malloc(sizeof(int64_t) * (size_t)(2 * NB));
        int64_t nt = 0;
        for (int64_t k = kb; k < ke; ++k) tpos[nt++] = k;
        for (int64_t k = kb; k < ke; ++k) {
            int64_t found = 0;
            for (int64_t t = 0; t < nt; ++t) found |= (tpos[t] == p[k]);
            if (!found) tpos[nt++] = p[k];
        }
        int64_t *tix = (int64_t *)malloc(sizeof(int64_t) * (size_t)NB); /* buffer row of p[k] */
        for (int64_t k = kb; k < ke; ++k)
            for (int64_t t = 0; t < nt; ++t)
                if (tpos[t] == p[k]) tix[k - kb] = t;
        unsigned char *in_t = (unsigned char *)calloc((size_t)rows, 1);
        for (int64_t t = 0; t < nt; ++t) in_t[tpos[t] - kb] = 1;
        const int64_t nblk = (n - pe + JW - 1) / JW;
#pragma omp parallel
        {
            double *blk = (double *)malloc(sizeof(double) * (size_t)(2 * NB) * JW);
#pragma omp for schedule(dynamic, 1)
            for (int64_t bi = 0; bi < nblk; ++bi) {
                const int64_t j0 = pe + bi * JW, jw = (n - j0) < JW ? (n - j0) : JW;
                for (int64_t t = 0; t < nt; ++t)
                    memcpy(blk + t * jw, a + tpos[t] * n + j0, sizeof(double) * (size_t)jw);
                for (int64_t kk = 0; kk < nsteps; ++kk) {
                    double *rk = blk + kk * jw; /* position kb + kk is buffer row kk */
                    if (tix[kk] != kk) {
                        double *rm = blk + tix[kk] * jw;
                        for (int64_t jj = 0; jj < jw; ++jj) {
                            const double t = rm[jj];
                            rm[jj] = rk[jj];
                            rk[jj] = t;
                        }
                    }
                    if (pz[kk]) continue;
                    for (int64_t t = 0; t < nt; ++t) {
                        if (tpos[t] <= kb + kk) continue; /* rows i > k only */
                        const double lik = panel[(tpos[t] - kb) * pw + kk];
                        double *ri = blk + t * jw;
                        for (int64_t jj = 0; jj < jw; ++jj) {
                            const double adjustment = lik * rk[jj];
                            ri[jj] -= adjustment;
                        }
                    }
                }
                for (int64_t t = 0; t < nt; ++t)
                    memcpy(a + tpos[t] * n + j0, blk + t * jw, sizeof(double) * (size_t)jw);
                /* rows outside T: all steps in one visit (blk rows 0..nsteps-1 are final) */
                if (jw == JW) {
                    for (int64_t i = ke; i < n; ++i) {
                        if (in_t[i - kb]) continue;
                        double *ri = a + i * n + j0;
                        const double *li = panel + (i - kb) * pw;
                        double r[JW];
                        for (int64_t jj = 0; jj < JW; ++jj) r[jj] = ri[jj];
                        for (int64_t kk = 0; kk < nsteps; ++kk) {
                            if (pz[kk]) continue;
                            const double lik = li[kk];
                            const double *rk = blk + kk * JW;
                            for (int64_t jj = 0; jj < JW; ++jj) {
                                const double adjustment = lik * rk[jj];
                                r[jj] -= adjustment;
                            }
                        }
                        for (int64_t jj = 0; jj < JW; ++jj) ri[jj] = r[jj];
                    }
                } else {
                    for (int64_t i = ke; i < n; ++i) {
                        if (in_t[i - kb]) continue;
                        double *ri = a + i * n + j0;
                        const double *li = panel + (i - kb) * pw;
                        for (int64_t kk = 0; kk < nsteps; ++kk) {
                            if (pz[kk]) continue;
                            const double lik = li[kk];
                            const double *rk = blk + kk * jw;
                            for (int64_t jj = 0; jj < jw; ++jj) {
                                const double adjustment = lik * rk[jj];
                                ri[jj] -= adjustment;
                            }
                        }
                    }
                }
            }
            free(blk);
        }
        free(tpos);
        free(tix);
        free(in_t);
#pragma omp parallel for schedule(static)
        for (int64_t i = kb; i < n; ++i)
            memcpy(a + i * n + kb, panel + (i - kb) * pw, sizeof(double) * (size_t)pw);
    }
    free(panel);
    free(pz);
}
