// k_strict.hip -- STRICT numerics: the reference's basis solves, operation for operation.
//
//   solve_for_dx  src/simplex.rs:226-229   lu_solve(B.to_dense(),     column(j))
//   solve_for_dz  src/simplex.rs:231-236   lu_solve(B.to_dense().t(), unit(p))
//   factorize     src/linalg.rs:88-128     LINPACK-style in-place LU, partial pivoting,
//                                          swaps on columns k..n only, unpermuted L
//   LU::solve     src/linalg.rs:282-299    forward (swaps interleaved), backward (j ascending)
//
// Every matrix element must see the reference's operations in the reference's order: at step k
// (ascending) one rounded product l(i,k)*u(k,j) and one rounded subtraction, never fused
// (-ffp-contract=off).  WHICH thread applies them, and when, is free -- that freedom is all the
// parallelism there is, and this file takes it in the blocked right-looking shape:
//
//   * panel (64 columns): one kernel per elimination step, restricted to the panel.  The active
//     part of the panel ping-pongs between two buffers, so a step reads only old values and
//     writes only new ones (no read/write race on the two rows that swap); each workgroup
//     re-derives the pivot row from the per-workgroup maxima the previous step left behind
//     (first maximum of |.|, strict '>', src/linalg.rs:98-105).
//   * columns right of the panel are brought up to date once per panel: the panel's row swaps
//     in order, then for every element the 64 updates in ascending k -- an "ordered GEMM" on the
//     vector ALUs (v_mul_f64 + v_add_f64; the matrix cores fuse and would change the bits).
//     Because the reference keeps L unpermuted (swaps touch columns >= k only), the multiplier
//     that met a row at step k sits at the position the row had THEN: ptab[k][p] = position after
//     swap k of the row that ends the panel at position p.
//   * the right-hand side rides along as column n of the matrix: LINPACK's forward elimination
//     (src/linalg.rs:286-291: swap b[k], b[p[k]]; b[i] -= b[k]*a(i,k)) is the same sequence of
//     operations on that column, so the forward substitution needs no pass of its own.
//   * back substitution (src/linalg.rs:292-297) is a true serial chain of n^2/2 dependent
//     subtractions: one wave forms the products of a 64-wide chunk in
//     parallel, parks them in LDS and chains the running difference through them.
//
// The packed factors, the pivot vector and the solution are bit-identical to the reference's
// (tests/test_gpu_parity.py).
#include <cstdlib>

#include "common.h"

#define NB DZG_LU_NB
#define LDP DZG_LU_LDP
typedef double double2_t __attribute__((ext_vector_type(2)));

// W row-major n x ldw.  transposed == 0: W[r][c] = A[r, basis[c]]  (B)
//                       transposed == 1: W[r][c] = A[c, basis[r]]  (B^T)
// Workgroup m copies the right-hand side into column n.
__global__ __launch_bounds__(256) void k_gather_basis(const DzgCtl *ctl, double *__restrict__ W,
                                                      long long ldw, int m,
                                                      const double *__restrict__ A, long long lda,
                                                      const int *__restrict__ basis,
                                                      const int *__restrict__ var_col,
                                                      int transposed,
                                                      const long long *__restrict__ cptr,
                                                      const int *__restrict__ ridx,
                                                      const double *__restrict__ cval,
                                                      const double *__restrict__ rhs)
{
    if (ctl->status != DZG_RUNNING) return;
    const int b = blockIdx.x; // basis position
    if (b == m) {
        for (int i = threadIdx.x; i < m; i += blockDim.x) W[(long long)i * ldw + m] = rhs[i];
        return;
    }
    const int code = var_col[basis[b]];
    if (cptr && code >= 0) { // sparse column: zero this workgroup's row/column of W, then scatter
        for (int i = threadIdx.x; i < m; i += blockDim.x) {
            if (transposed)
                W[(long long)b * ldw + i] = 0.0;
            else
                W[(long long)i * ldw + b] = 0.0;
        }
        __syncthreads();
        for (long long e = cptr[code] + threadIdx.x; e < cptr[code + 1]; e += blockDim.x) {
            const int i = ridx[e];
            if (transposed)
                W[(long long)b * ldw + i] = cval[e];
            else
                W[(long long)i * ldw + b] = cval[e];
        }
        return;
    }
    const double *col = code >= 0 ? A + (long long)code * lda : nullptr;
    const int srow = -1 - code;
    for (int i = threadIdx.x; i < m; i += blockDim.x) {
        const double val = col ? col[i] : (i == srow ? 1.0 : 0.0);
        if (transposed)
            W[(long long)b * ldw + i] = val;
        else
            W[(long long)i * ldw + b] = val;
    }
}

// first maximum of |.|: larger value wins, lower row on ties; NaN never wins (x > NaN is false)
__device__ __forceinline__ DzgCand lu_cand(double value, int row)
{
    DzgCand c;
    c.r = fabs(value);
    c.k = (c.r == c.r) ? row : -1;
    return c;
}

// Start of a panel: copy columns k0..k0+nbw-1 and the right-hand side of rows >= k0 into the
// first panel buffer; leave the per-workgroup maxima of column k0.  RW rows per wave.
template <int RW>
__global__ __launch_bounds__(256) void k_lu_panel_load(const DzgCtl *ctl, int n, int k0, int nbw,
                                                       const double *__restrict__ W, long long ldw,
                                                       double *__restrict__ P,
                                                       double *__restrict__ part_r,
                                                       int *__restrict__ part_k)
{
    if (ctl->status != DZG_RUNNING) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    DzgCand best;
    best.r = 0.0;
    best.k = -1;
#pragma unroll
    for (int r = 0; r < RW; ++r) {
        const int i = k0 + (blockIdx.x * 4 + wave) * RW + r;
        if (i < n) {
            const double *wi = W + (long long)i * ldw;
            double *pi = P + (long long)i * LDP;
            double val = 0.0;
            if (lane < nbw) {
                val = wi[k0 + lane];
                pi[lane] = val;
            }
            if (lane == 0) {
                pi[NB] = wi[n];
                best = dzg_better(best, lu_cand(val, i));
            }
        }
    }
    best = dzg_block_best(best);
    if (threadIdx.x == 0) {
        part_r[blockIdx.x] = best.r;
        part_k[blockIdx.x] = best.k;
    }
}

// Elimination step k inside the panel [k0, k0+nbw).  in -> out (ping-pong); rows >= k.
//   flush: last step of the panel -- the right-hand side (and, at the very end of the matrix,
//   the columns right of k) go back to W, nothing is left in the panel buffers.
// One kernel = one dependent chain (maxima -> pivot row -> update), so everything that does not
// depend on the pivot row is loaded before the reduction: the workgroup's own rows and row k.
template <int RW>
__global__ __launch_bounds__(256) void k_lu_step(const DzgCtl *ctl, int n, int k, int k0, int nbw,
                                                 int flush, const double *__restrict__ in,
                                                 double *__restrict__ out, double *__restrict__ W,
                                                 long long ldw, int *__restrict__ piv,
                                                 int *__restrict__ pz,
                                                 const double *__restrict__ pin_r,
                                                 const int *__restrict__ pin_k, int nparts,
                                                 double *__restrict__ pout_r,
                                                 int *__restrict__ pout_k)
{
    if (ctl->status != DZG_RUNNING) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int ck = k - k0;
    const int lc = lane < nbw ? lane : 0; // clamped panel column: every load is unconditional
    // ---- loads that do not depend on the pivot row
    const double *krow = in + (long long)k * LDP; // row k before the swap
    const double k_c = krow[lc], k_b = krow[NB];
    double own_c[RW], own_b[RW];
    int row[RW];
#pragma unroll
    for (int r = 0; r < RW; ++r) {
        row[r] = k + (blockIdx.x * 4 + wave) * RW + r;
        const double *src = in + (long long)(row[r] < n ? row[r] : k) * LDP;
        own_c[r] = src[lc];
        own_b[r] = src[NB];
    }
    // ---- pivot row: every workgroup reduces the previous step's maxima the same way
    DzgCand best;
    best.r = 0.0;
    best.k = -1;
    for (int g = threadIdx.x; g < nparts; g += blockDim.x) {
        DzgCand c;
        c.r = pin_r[g];
        c.k = pin_k[g];
        best = dzg_better(best, c);
    }
    best = dzg_block_best(best);
    const double akk = dzg_readlane_f64(k_c, ck);
    // `x > NaN` is never true: a NaN at (k,k) keeps mu = k (src/linalg.rs:98-105)
    const int mu = (fabs(akk) != fabs(akk) || best.k < 0) ? k : best.k;
    const double *urow = in + (long long)mu * LDP;
    const double u_c = urow[lc]; // row k after the swap, column k0 + lane
    const double u_b = urow[NB];
    const double pivot = dzg_readlane_f64(u_c, ck);
    const bool zero = !(pivot != 0.0); // `if a(k,k) != 0.0`, src/linalg.rs:117 (NaN is "nonzero")
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        piv[k] = mu;
        pz[k] = zero ? 1 : 0;
    }
    const bool next_col = ck + 1 < nbw; // the next pivot column is still inside this panel
    DzgCand nbest;
    nbest.r = 0.0;
    nbest.k = -1;
#pragma unroll
    for (int r = 0; r < RW; ++r) {
        const int i = row[r];
        if (i >= n) continue;
        double *wi = W + (long long)i * ldw;
        if (i == k) { // the finished row of U (and its right-hand side entry)
            if (lane >= ck && lane < nbw) wi[k0 + lane] = u_c;
            if (lane == 0) wi[n] = u_b;
            continue;
        }
        const double s_c = (i == mu) ? k_c : own_c[r]; // the swap, columns >= k
        const double s_b = (i == mu) ? k_b : own_b[r];
        const double s_k = dzg_readlane_f64(s_c, ck);
        // zero pivot: no scaling, no update of the matrix (src/linalg.rs:117-125); LU::solve
        // still runs b[i] -= b[k] * a(i,k) with the stored entry (src/linalg.rs:288-290)
        const double l = zero ? s_k : dzg_div(s_k, pivot);
        double val = s_c;
        if (!zero) {
            const double adjustment = l * u_c;
            val = s_c - adjustment;
        }
        const double badj = u_b * l;
        const double vb = s_b - badj;
        double *oi = out + (long long)i * LDP;
        if (lane == ck) wi[k] = l;
        if (lane > ck && lane < nbw) {
            if (flush)
                wi[k0 + lane] = val;
            else
                oi[lane] = val;
        }
        if (lane == 0) {
            if (flush)
                wi[n] = vb;
            else
                oi[NB] = vb;
        }
        if (next_col) {
            const double nv = dzg_readlane_f64(val, ck + 1);
            nbest = dzg_better(nbest, lu_cand(nv, i));
        }
    }
    if (next_col && !flush) {
        nbest = dzg_block_best(nbest);
        if (threadIdx.x == 0) {
            pout_r[blockIdx.x] = nbest.r;
            pout_k[blockIdx.x] = nbest.k;
        }
    }
}

// A whole panel in ONE workgroup when its rows fit in LDS ((n - k0) x 66 doubles: every panel of
// a matrix up to ~290 rows, and the last panels of any matrix): the elimination steps are
// separated by workgroup barriers instead of kernel boundaries (~0.5 us instead of ~6 us per
// step).  Same operations in the same order: pivot search, swap on columns >= k, multipliers,
// rank-1 update -- here literally in place, as the reference does it.
__global__ __launch_bounds__(1024) void k_lu_panel_lds(const DzgCtl *ctl, int n, int k0, int nbw,
                                                       int nsteps, double *__restrict__ W,
                                                       long long ldw, int *__restrict__ piv,
                                                       int *__restrict__ pz)
{
    extern __shared__ __attribute__((aligned(16))) double s_pan[]; // [(n - k0)][LDP]
    if (ctl->status != DZG_RUNNING) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
    const int rows = n - k0;
    for (int r = wave; r < rows; r += nwaves) {
        const double *wi = W + (long long)(k0 + r) * ldw;
        if (lane < nbw) s_pan[r * LDP + lane] = wi[k0 + lane];
        if (lane == 0) s_pan[r * LDP + NB] = wi[n];
    }
    __syncthreads();
    for (int s = 0; s < nsteps; ++s) { // step k = k0 + s, local row index s
        DzgCand best;
        best.r = 0.0;
        best.k = -1;
        for (int r = s + threadIdx.x; r < rows; r += blockDim.x)
            best = dzg_better(best, lu_cand(s_pan[r * LDP + s], r));
        best = dzg_block_best(best);
        const double akk = s_pan[s * LDP + s];
        // `x > NaN` is never true: a NaN at (k,k) keeps mu = k (src/linalg.rs:98-105)
        const int mu = (fabs(akk) != fabs(akk) || best.k < 0) ? s : best.k;
        const double pivot = s_pan[mu * LDP + s];
        const bool zero = !(pivot != 0.0);
        if (threadIdx.x == 0) {
            piv[k0 + s] = k0 + mu;
            pz[k0 + s] = zero ? 1 : 0;
        }
        __syncthreads(); // everybody has read the pivot before the rows move
        if (mu != s && wave == 0) { // swap rows k and mu on columns >= k, and the right-hand side
            if (lane >= s && lane < nbw) {
                const double a = s_pan[s * LDP + lane];
                s_pan[s * LDP + lane] = s_pan[mu * LDP + lane];
                s_pan[mu * LDP + lane] = a;
            }
            if (lane == 0) {
                const double a = s_pan[s * LDP + NB];
                s_pan[s * LDP + NB] = s_pan[mu * LDP + NB];
                s_pan[mu * LDP + NB] = a;
            }
        }
        __syncthreads();
        const double u_c = lane < nbw ? s_pan[s * LDP + lane] : 0.0;
        const double u_b = s_pan[s * LDP + NB];
        for (int r = s + 1 + wave; r < rows; r += nwaves) {
            double *row = s_pan + r * LDP;
            const double s_k = row[s];
            // zero pivot: no scaling, no update of the matrix (src/linalg.rs:117-125); LU::solve
            // still runs b[i] -= b[k] * a(i,k) with the stored entry (src/linalg.rs:288-290)
            const double l = zero ? s_k : dzg_div(s_k, pivot);
            if (!zero && lane > s && lane < nbw) {
                const double adjustment = l * u_c;
                row[lane] = row[lane] - adjustment;
            }
            if (lane == 0) {
                const double badj = u_b * l;
                row[NB] = row[NB] - badj;
            }
            __builtin_amdgcn_wave_barrier(); // row[s] is read by every lane before lane s rewrites it
            if (!zero && lane == s) row[s] = l;
        }
        __syncthreads();
    }
    for (int r = wave; r < rows; r += nwaves) {
        double *wi = W + (long long)(k0 + r) * ldw;
        if (lane < nbw) wi[k0 + lane] = s_pan[r * LDP + lane];
        if (lane == 0) wi[n] = s_pan[r * LDP + NB];
    }
}

// ptab[s][p] = position, right after the swap of step k0+s, of the row that ends the panel at
// position p (the multiplier it met at that step sits there: L is unpermuted).
__global__ __launch_bounds__(256) void k_lu_ptab(const DzgCtl *ctl, int n, int k0, int nsteps,
                                                 const int *__restrict__ piv,
                                                 int *__restrict__ ptab)
{
    if (ctl->status != DZG_RUNNING) return;
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    int q = p;
    for (int s = nsteps - 1; s >= 0; --s) {
        ptab[(long long)s * n + p] = q;
        const int k = k0 + s, mu = piv[k];
        if (q == k)
            q = mu;
        else if (q == mu)
            q = k;
    }
}

// Columns right of the panel, block rows k0..k0+63: the panel's swaps in order, then the
// triangular part of the update (row p receives steps k < p, ascending).  A workgroup stages a
// 64 x 64 tile in LDS; the swaps run one thread per column (far rows are read and written 64
// columns at a time); the updates run one wave per 4 columns with lane = row, the pivot-row
// element broadcast by v_readlane.
__global__ __launch_bounds__(256) void k_lu_trail_u(const DzgCtl *ctl, int n, int k0,
                                                    double *__restrict__ W, long long ldw,
                                                    const int *__restrict__ piv,
                                                    const int *__restrict__ pz,
                                                    const int *__restrict__ ptab)
{
    // s_x: first the rows BELOW the panel that its swaps touch (staged so that the 64 swaps run
    // on LDS instead of 64 dependent HBM round trips), afterwards the multipliers
    __shared__ double s_x[NB][NB + 1];
    __shared__ double s_y[NB][NB + 1]; // s_y[r][c] = block row r, tile column c
    __shared__ int s_mu[NB], s_pz[NB], s_slot[NB];
    if (ctl->status != DZG_RUNNING) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int k1 = k0 + NB;
    const int j0 = k1 + blockIdx.x * 64;
    if (tid < NB) {
        s_mu[tid] = piv[k0 + tid];
        s_pz[tid] = pz[k0 + tid];
    }
    __syncthreads();
    if (tid < NB) { // a far row hit by several swaps is staged once: slot = first step that hits it
        int slot = tid;
        const int mu = s_mu[tid];
        for (int t = tid - 1; t >= 0; --t)
            if (s_mu[t] == mu) slot = t;
        s_slot[tid] = slot;
    }
    const bool live = j0 + lane < n;
    const int jc = live ? j0 + lane : n - 1; // clamped: dead columns are read, never stored
    for (int r = wave; r < NB; r += 4) s_y[r][lane] = W[(long long)(k0 + r) * ldw + jc];
    __syncthreads();
    for (int st = wave; st < NB; st += 4) {
        const int mu = s_mu[st];
        if (mu >= k1 && s_slot[st] == st) s_x[st][lane] = W[(long long)mu * ldw + jc];
    }
    __syncthreads();
    if (wave == 0) { // row swaps of the panel, in order (columns >= k), one thread per column
        for (int st = 0; st < NB; ++st) {
            const int mu = s_mu[st];
            if (mu == k0 + st) continue;
            const double a = s_y[st][lane];
            if (mu < k1) {
                s_y[st][lane] = s_y[mu - k0][lane];
                s_y[mu - k0][lane] = a;
            } else {
                const int slot = s_slot[st];
                s_y[st][lane] = s_x[slot][lane];
                s_x[slot][lane] = a;
            }
        }
    }
    __syncthreads();
    if (live)
        for (int st = wave; st < NB; st += 4) {
            const int mu = s_mu[st];
            if (mu >= k1 && s_slot[st] == st) W[(long long)mu * ldw + j0 + lane] = s_x[st][lane];
        }
    __syncthreads();
    for (int e = tid; e < NB * NB; e += 256) { // s_x[r][st] = multiplier of block row r at step st
        const int r = e / NB, st = e % NB;
        s_x[r][st] = r > st ? W[(long long)ptab[(long long)st * n + k0 + r] * ldw + k0 + st] : 0.0;
    }
    __syncthreads();
    for (int c0 = wave * 16; c0 < wave * 16 + 16; c0 += 4) { // lane = block row
        double y0 = s_y[lane][c0], y1 = s_y[lane][c0 + 1], y2 = s_y[lane][c0 + 2],
               y3 = s_y[lane][c0 + 3];
        for (int st = 0; st + 1 < NB; ++st) {
            if (s_pz[st]) continue;
            const double l = s_x[lane][st];
            const double u0 = dzg_readlane_f64(y0, st), u1 = dzg_readlane_f64(y1, st),
                         u2 = dzg_readlane_f64(y2, st), u3 = dzg_readlane_f64(y3, st);
            if (lane > st) {
                const double a0 = l * u0, a1 = l * u1, a2 = l * u2, a3 = l * u3;
                y0 = y0 - a0;
                y1 = y1 - a1;
                y2 = y2 - a2;
                y3 = y3 - a3;
            }
        }
        s_y[lane][c0] = y0;
        s_y[lane][c0 + 1] = y1;
        s_y[lane][c0 + 2] = y2;
        s_y[lane][c0 + 3] = y3;
    }
    __syncthreads();
    if (live)
        for (int r = wave; r < NB; r += 4) W[(long long)(k0 + r) * ldw + j0 + lane] = s_y[r][lane];
}

// Rows and columns right of / below the panel: a(p,j) -= l(p,k) u(k,j) for k = k0..k0+63 in
// ascending order, product and difference rounded separately.  64 x 64 tile per workgroup,
// 4 x 4 elements per thread, both operand tiles staged through LDS.
__global__ __launch_bounds__(256) void k_lu_trail_gemm(const DzgCtl *ctl, int n, int k0,
                                                       double *__restrict__ W, long long ldw,
                                                       const int *__restrict__ pz,
                                                       const int *__restrict__ ptab)
{
    __shared__ double s_l[64][NB + 1]; // [tile row][step]
    __shared__ __attribute__((aligned(16))) double s_u[NB][64]; // [step][tile column]
    __shared__ int s_pz[NB];
    if (ctl->status != DZG_RUNNING) return;
    const int k1 = k0 + NB;
    const int p0 = k1 + blockIdx.y * 64, j0 = k1 + blockIdx.x * 64;
    const int tid = threadIdx.x;
    if (tid < NB) s_pz[tid] = pz[k0 + tid];
    for (int e = tid; e < 64 * NB; e += 256) {
        const int r = e / NB, s = e % NB; // consecutive threads: consecutive steps of one row
        const int p = p0 + r;
        s_l[r][s] = p < n ? W[(long long)ptab[(long long)s * n + p] * ldw + k0 + s] : 0.0;
    }
    for (int e = tid; e < NB * 64; e += 256) {
        const int s = e / 64, c = e % 64;
        const int j = j0 + c;
        s_u[s][c] = j < n ? W[(long long)(k0 + s) * ldw + j] : 0.0;
    }
    const int ty = tid >> 4, tx = tid & 15;
    double c[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const int p = p0 + ty * 4 + a;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int j = j0 + tx * 4 + b;
            c[a][b] = (p < n && j < n) ? W[(long long)p * ldw + j] : 0.0;
        }
    }
    __syncthreads();
    for (int s = 0; s < NB; ++s) {
        if (s_pz[s]) continue;
        double l[4], u[4];
#pragma unroll
        for (int a = 0; a < 4; ++a) l[a] = s_l[ty * 4 + a][s];
#pragma unroll
        for (int b = 0; b < 4; ++b) u[b] = s_u[s][tx * 4 + b];
#pragma unroll
        for (int a = 0; a < 4; ++a) {
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const double adjustment = l[a] * u[b];
                c[a][b] = c[a][b] - adjustment;
            }
        }
    }
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const int p = p0 + ty * 4 + a;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int j = j0 + tx * 4 + b;
            if (p < n && j < n) W[(long long)p * ldw + j] = c[a][b];
        }
    }
}

// Back substitution, src/linalg.rs:292-297: a serial chain of n^2/2 dependent subtractions in
// ascending j.  One wave walks the rows upwards in chunks of 64 columns; for each chunk the 64
// lanes form the products a(i,j)*b[j] in parallel and park them in LDS, every lane then reads all
// 64 back (broadcast reads, independent of the chain) and chains the running difference through
// them -- VGPR operands back to back (v_readlane into one SGPR pair stalls on hazards).
// The matrix elements are the only operand that comes from HBM: the chunk sequence
// (i, j0) = (n-1, n-1), (n-2, n-2), (n-2, n-2+64)?, ... is known in advance, so a second cursor
// runs PF chunks ahead and keeps that many loads in flight (a chunk starts at the diagonal, which
// lane 0 keeps for the final division).  b lives in LDS when it fits.
#define PF 16
template <bool LDS_B>
__global__ __launch_bounds__(64) void k_lu_backsolve(const DzgCtl *ctl, int n,
                                                     const double *__restrict__ W, long long ldw,
                                                     double *x_out)
{
    extern __shared__ __attribute__((aligned(16))) double s_dyn[];
    double2_t *s_p = reinterpret_cast<double2_t *>(s_dyn); // [2][32] product chunks
    if (ctl->status != DZG_RUNNING) return;
    const int lane = threadIdx.x;
    // b: LDS when it fits (LDS_B), else the contiguous output vector itself.  Two instantiations,
    // so that each addresses ONE memory space (a generic pointer would make every access a flat
    // load, which the compiler can only wait for with vmcnt(0) -- and stall the prefetches)
    double *s_b = s_dyn + 128;
#define B_AT(idx) (LDS_B ? s_b[idx] : x_out[idx])
#define B_SET(idx, val)                                                                             \
    do {                                                                                            \
        if (LDS_B)                                                                                  \
            s_b[idx] = (val);                                                                       \
        else                                                                                        \
            x_out[idx] = (val);                                                                     \
    } while (0)
    for (int i = lane; i < n; i += 64) B_SET(i, W[(long long)i * ldw + n]);
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    __builtin_amdgcn_wave_barrier();
    // prefetch cursor (pi, pj) and consumer cursor (ci, cj): same sequence, PF chunks apart.
    // Every global load below is unconditional (addresses clamped into the matrix: the prefetch
    // cursor runs PF chunks past the last row) and the loop body is
    // straight-line, so the compiler can count outstanding loads: s_waitcnt vmcnt(PF-1), not 0.
    int pi = n - 1, pj = n - 1;
    double ring[PF];
#pragma unroll
    for (int u = 0; u < PF; ++u) {
        const int jj = pj + lane;
        ring[u] = W[(long long)(pi > 0 ? pi : 0) * ldw + (jj < 0 ? 0 : (jj < n ? jj : n - 1))];
        pj += 64;
        const bool wrap = pj >= n;
        pi = wrap ? pi - 1 : pi;
        pj = wrap ? pi : pj;
    }
    int ci = n - 1, cj = n - 1, buf = 0;
    double acc = 0.0, diag = 1.0;
    while (ci >= 0) {
#pragma unroll
        for (int u = 0; u < PF; ++u) {
            const double w = ring[u];
            { // refill this slot PF chunks ahead
                const int jj = pj + lane;
                ring[u] = W[(long long)(pi > 0 ? pi : 0) * ldw + (jj < 0 ? 0 : (jj < n ? jj : n - 1))];
                pj += 64;
                const bool wrap = pj >= n;
                pi = wrap ? pi - 1 : pi;
                pj = wrap ? pi : pj;
            }
            const bool alive = ci >= 0; // past the last row the remaining slots are no-ops
            const int ri = alive ? ci : 0;
            const bool first = cj == ci;
            const int j = cj + lane;
            const double bi = B_AT(ri);
            const double d0 = dzg_readlane_f64(w, 0);
            acc = first ? bi : acc;
            diag = first ? d0 : diag;
            // the diagonal and lanes past the end contribute +0.0: acc - (+0.0) == acc
            const double bj = B_AT((alive && j < n) ? j : 0);
            const double prod = w * bj;
            const double p = (alive && j < n && !(first && lane == 0)) ? prod : 0.0;
            // one wave: its LDS operations execute in program order, so the 64 lanes' writes are
            // visible to the reads below without a fence (a fence would also wait for the
            // prefetches: vmcnt(0))
            reinterpret_cast<double *>(s_p + buf * 32)[lane] = p;
            __builtin_amdgcn_wave_barrier();
            const double2_t *q = s_p + buf * 32;
            double2_t v[32]; // all 32 reads in flight before the first add (LDS latency paid once)
#pragma unroll
            for (int l = 0; l < 32; ++l) v[l] = q[l];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int l = 0; l < 32; ++l) {
                acc = acc - v[l].x;
                acc = acc - v[l].y;
            }
            buf ^= 1;
            cj += 64;
            if (alive && cj >= n) { // row finished
                acc = dzg_div(acc, diag);
                if (lane == 0) B_SET(ci, acc);
                if (!LDS_B) __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
                __builtin_amdgcn_wave_barrier();
                --ci;
                cj = ci;
            }
        }
    }
    if (LDS_B)
        for (int i = lane; i < n; i += 64) x_out[i] = s_b[i];
#undef B_AT
#undef B_SET
}

// dynamic LDS the back-substitution kernel may use: 64 KB by default, most of the CU's 160 KB
// if the runtime grants it (right-hand sides up to ~19 000 rows then stay on chip)
static size_t backsolve_lds_limit()
{
    static size_t limit = 0;
    if (!limit) {
        limit = 64 * 1024;
        if (getenv("DZG_LU_SMALL_LDS")) { // test switch: exercise the out-of-LDS path at small n
            limit = 2048;
            return limit;
        }
        const int big = 152 * 1024;
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(k_lu_backsolve<true>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, big) == hipSuccess)
            limit = (size_t)big;
        else
            (void)hipGetLastError();
    }
    return limit;
}

// dynamic LDS of the single-workgroup panel kernel (same policy)
static size_t panel_lds_limit()
{
    static size_t limit = 0;
    if (!limit) {
        limit = 60 * 1024;
        if (getenv("DZG_LU_SMALL_LDS")) { // test switch: multi-workgroup panel steps at small n too
            limit = 1;
            return limit;
        }
        const int big = 152 * 1024;
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(k_lu_panel_lds),
                                hipFuncAttributeMaxDynamicSharedMemorySize, big) == hipSuccess)
            limit = (size_t)big;
        else
            (void)hipGetLastError();
    }
    return limit;
}

// factorise W (columns 0..n-1) with the right-hand side in column n, then back-substitute.
// RW rows per wave in the panel kernels: few for small matrices (more workgroups in flight),
// more for large ones (every workgroup re-reduces all per-workgroup maxima of the previous step).
template <int RW>
static void factorize_and_solve_t(const DzgLu &w, DzgCtl *ctl, double *x_out, hipStream_t st)
{
    const int n = w.n;
    const int rpg = 4 * RW;
    double *P[2] = {w.P0, w.P1};
    for (int k0 = 0; k0 + 1 < n; k0 += NB) {
        const int nbw = (n - k0) < NB ? (n - k0) : NB;
        const int klast = (k0 + nbw - 1) < (n - 2) ? (k0 + nbw - 1) : (n - 2); // last step here
        const int nsteps = klast - k0 + 1;
        const size_t pan_lds = sizeof(double) * (size_t)(n - k0) * LDP;
        if (pan_lds <= panel_lds_limit()) { // the panel's rows fit in one workgroup's LDS
            hipLaunchKernelGGL(k_lu_panel_lds, dim3(1), dim3(1024), pan_lds, st, ctl, n, k0, nbw, nsteps,
                               w.W, w.ldw, w.piv, w.pz);
        } else {
            int nparts = (n - k0 + rpg - 1) / rpg;
            hipLaunchKernelGGL((k_lu_panel_load<RW>), dim3(nparts), dim3(256), 0, st, ctl, n, k0, nbw,
                               w.W, w.ldw, P[0], w.part_r, w.part_k);
            for (int k = k0; k <= klast; ++k) {
                const int par = (k - k0) & 1;
                const int grid = (n - k + rpg - 1) / rpg;
                hipLaunchKernelGGL((k_lu_step<RW>), dim3(grid), dim3(256), 0, st, ctl, n, k, k0, nbw,
                                   k == klast ? 1 : 0, P[par], P[par ^ 1], w.W, w.ldw, w.piv, w.pz,
                                   w.part_r + (size_t)par * w.nparts,
                                   w.part_k + (size_t)par * w.nparts, nparts,
                                   w.part_r + (size_t)(par ^ 1) * w.nparts,
                                   w.part_k + (size_t)(par ^ 1) * w.nparts);
                nparts = grid;
            }
        }
        const int rest = n - (k0 + NB);
        if (rest > 0) { // only full panels have columns to their right (nsteps == NB)
            hipLaunchKernelGGL(k_lu_ptab, dim3((n + 255) / 256), dim3(256), 0, st, ctl, n, k0, nsteps,
                               w.piv, w.ptab);
            hipLaunchKernelGGL(k_lu_trail_u, dim3((rest + 63) / 64), dim3(256), 0, st, ctl, n, k0, w.W,
                               w.ldw, w.piv, w.pz, w.ptab);
            hipLaunchKernelGGL(k_lu_trail_gemm, dim3((rest + 63) / 64, (rest + 63) / 64), dim3(256), 0,
                               st, ctl, n, k0, w.W, w.ldw, w.pz, w.ptab);
        }
    }
    const size_t want = sizeof(double) * (128 + (size_t)n);
    const int use_lds = want <= backsolve_lds_limit();
    const size_t lds = use_lds ? want : sizeof(double) * 128;
    if (use_lds)
        hipLaunchKernelGGL((k_lu_backsolve<true>), dim3(1), dim3(64), lds, st, ctl, n, w.W, w.ldw, x_out);
    else
        hipLaunchKernelGGL((k_lu_backsolve<false>), dim3(1), dim3(64), lds, st, ctl, n, w.W, w.ldw,
                           x_out);
}

static int rows_per_wave(int n) { return n <= 2048 ? 1 : (n <= 4096 ? 2 : 4); }

static void factorize_and_solve(const DzgLu &w, DzgCtl *ctl, double *x_out, hipStream_t st)
{
    switch (rows_per_wave(w.n)) {
    case 1: factorize_and_solve_t<1>(w, ctl, x_out, st); break;
    case 2: factorize_and_solve_t<2>(w, ctl, x_out, st); break;
    default: factorize_and_solve_t<4>(w, ctl, x_out, st); break;
    }
}

void dzg_launch_strict_solve(const DzgDev &d, int transposed, hipStream_t st)
{
    double *vec = transposed ? d.v : d.dx; // right-hand side in, solution out
    hipLaunchKernelGGL(k_gather_basis, dim3(d.m + 1), dim3(256), 0, st, d.ctl, d.lu.W, d.lu.ldw, d.m,
                       d.A, d.lda, d.basis, d.var_col, transposed, d.csc ? d.cptr : nullptr, d.ridx,
                       d.cval, vec);
    factorize_and_solve(d.lu, d.ctl, vec, st);
}

// W holds the matrix (row stride ldw) and the right-hand side in column n on entry; the packed
// factors on exit.  x_out receives the solution.
void dzg_launch_lu_raw(const DzgLu &w, DzgCtl *ctl, double *x_out, hipStream_t st)
{
    factorize_and_solve(w, ctl, x_out, st);
}

void dzg_lu_layout(int n, DzgLu *w)
{
    w->n = n;
    w->ldw = ((long long)n + 1 + 7) / 8 * 8 + 8;
    const int rpg = 4 * rows_per_wave(n);
    w->nparts = (n + rpg - 1) / rpg + 2;
    (void)backsolve_lds_limit(); // set the kernel attributes outside any stream capture
    (void)panel_lds_limit();
}
