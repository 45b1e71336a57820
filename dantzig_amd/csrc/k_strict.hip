// k_strict.hip -- STRICT numerics: the reference's basis solves, operation for operation.
//
//   solve_for_dx  src/simplex.rs:226-229   lu_solve(B.to_dense(),     column(j))
//   solve_for_dz  src/simplex.rs:231-236   lu_solve(B.to_dense().t(), unit(p))
//   factorize     src/linalg.rs:88-128     LINPACK-style in-place LU, partial pivoting,
//                                          swaps on columns k..n only, unpermuted L
//   LU::solve     src/linalg.rs:282-299    forward (swaps interleaved), backward (j ascending)
//
// Parallelism is taken only where it cannot change a single bit: inside one elimination
// step every a(i,j) -= a(i,k)*a(k,j) is independent (product and difference rounded
// separately, -ffp-contract=off), and inside one forward-substitution step every
// b[i] -= b[k]*a(i,k) is independent.  Each element therefore sees its updates in the
// reference's order (ascending k).  Back substitution is a true serial chain
// (SURVEY section 7 "Back-substitution order") and is run by a single wave: the 64 lanes
// form the products a(i,j)*b[j] of a 64-wide chunk in parallel and the running
// difference is then chained through them in ascending j with v_readlane.
// The packed factors and pivot vector that come out are bit-identical to
// Matrix::factorize's, which the parity tests check.
#include "common.h"

// W row-major m x m.  transposed == 0: W[r][c] = A[r, basis[c]]  (B)
//                     transposed == 1: W[r][c] = A[c, basis[r]]  (B^T)
__global__ __launch_bounds__(256) void k_gather_basis(const DzgCtl *ctl, double *__restrict__ W,
                                                      int m, const double *__restrict__ A,
                                                      long long lda, const int *__restrict__ basis,
                                                      const int *__restrict__ var_col,
                                                      int transposed,
                                                      const long long *__restrict__ cptr,
                                                      const int *__restrict__ ridx,
                                                      const double *__restrict__ cval)
{
    if (ctl->status != DZG_RUNNING) return;
    const int b = blockIdx.x; // basis position
    const int code = var_col[basis[b]];
    if (cptr && code >= 0) { // sparse column: zero this workgroup's row/column of W, then scatter
        for (int i = threadIdx.x; i < m; i += blockDim.x) {
            if (transposed)
                W[(long long)b * m + i] = 0.0;
            else
                W[(long long)i * m + b] = 0.0;
        }
        __syncthreads();
        for (long long e = cptr[code] + threadIdx.x; e < cptr[code + 1]; e += blockDim.x) {
            const int i = ridx[e];
            if (transposed)
                W[(long long)b * m + i] = cval[e];
            else
                W[(long long)i * m + b] = cval[e];
        }
        return;
    }
    const double *col = code >= 0 ? A + (long long)code * lda : nullptr;
    const int srow = -1 - code;
    for (int i = threadIdx.x; i < m; i += blockDim.x) {
        const double val = col ? col[i] : (i == srow ? 1.0 : 0.0);
        if (transposed)
            W[(long long)b * m + i] = val;
        else
            W[(long long)i * m + b] = val;
    }
}

// Step k, part 1 (one workgroup): pivot search down column k (first maximum of |.|,
// strict '>', src/linalg.rs:98-105), snapshot of the two rows that swap, multipliers.
__global__ __launch_bounds__(1024) void k_lu_pivot(DzgCtl *ctl, const double *__restrict__ W,
                                                   double *__restrict__ Lt, int *__restrict__ piv,
                                                   double *__restrict__ urow,
                                                   double *__restrict__ krow,
                                                   double *__restrict__ lcol, int n, int k)
{
    if (ctl->status != DZG_RUNNING) return;
    __shared__ int s_mu;
    const double akk = W[(long long)k * n + k];
    DzgCand best;
    best.r = 0.0;
    best.k = -1;
    for (int i = k + threadIdx.x; i < n; i += blockDim.x) {
        DzgCand c;
        c.r = fabs(W[(long long)i * n + k]);
        c.k = i;
        if (c.r == c.r) best = dzg_better(best, c);
    }
    best = dzg_block_best(best);
    if (threadIdx.x == 0) {
        // `x > NaN` is never true: a NaN at (k,k) keeps mu = k
        int mu = (fabs(akk) != fabs(akk) || best.k < 0) ? k : best.k;
        s_mu = mu;
        piv[k] = mu;
        ctl->lu_mu = mu;
    }
    __syncthreads();
    const int mu = s_mu;
    const double pivot = W[(long long)mu * n + k];
    if (threadIdx.x == 0) ctl->lu_pivot_zero = (pivot != 0.0) ? 0 : 1;
    for (int j = k + threadIdx.x; j < n; j += blockDim.x) {
        urow[j] = W[(long long)mu * n + j]; // row k after the swap
        krow[j] = W[(long long)k * n + j];  // goes to row mu
    }
    for (int i = k + 1 + threadIdx.x; i < n; i += blockDim.x) {
        const double src = (i == mu) ? akk : W[(long long)i * n + k];
        const double l = (pivot != 0.0) ? src / pivot : src; // :119, skipped on a zero pivot
        lcol[i] = l;
        Lt[(long long)k * n + i] = l;
    }
}

// Step k, part 2 (grid): swap + scale + rank-1 update of the trailing block,
// src/linalg.rs:107-124.  Every element is read and written by the same thread.
__global__ __launch_bounds__(256) void k_lu_update(const DzgCtl *ctl, double *__restrict__ W,
                                                   const double *__restrict__ urow,
                                                   const double *__restrict__ krow,
                                                   const double *__restrict__ lcol, int n, int k)
{
    if (ctl->status != DZG_RUNNING) return;
    const int mu = ctl->lu_mu;
    const int j = k + blockIdx.x * 64 + (threadIdx.x & 63);
    const int i = k + blockIdx.y * 4 + (threadIdx.x >> 6);
    if (i >= n || j >= n) return;
    if (i == k) {
        W[(long long)k * n + j] = urow[j];
        return;
    }
    if (ctl->lu_pivot_zero) { // no scaling, no update; the swap still happened (mu == k here)
        if (i == mu) W[(long long)i * n + j] = krow[j];
        return;
    }
    const double l = lcol[i];
    if (j == k) {
        W[(long long)i * n + k] = l;
        return;
    }
    const double src = (i == mu) ? krow[j] : W[(long long)i * n + j];
    const double adjustment = l * urow[j];
    W[(long long)i * n + j] = src - adjustment;
}

// LU::solve.  One workgroup; b lives in LDS when it fits (n <= 8192), else in place.
__global__ __launch_bounds__(1024) void k_lu_solve(const DzgCtl *ctl, const double *__restrict__ W,
                                                   const double *__restrict__ Lt,
                                                   const int *__restrict__ piv,
                                                   double *__restrict__ b_glob, int n, int use_lds)
{
    extern __shared__ __attribute__((aligned(16))) double s_b[];
    if (ctl->status != DZG_RUNNING) return;
    double *b = use_lds ? s_b : b_glob;
    const int tid = threadIdx.x;
    if (use_lds) {
        for (int i = tid; i < n; i += blockDim.x) s_b[i] = b_glob[i];
    }
    __syncthreads();
    // forward, src/linalg.rs:286-291
    for (int k = 0; k + 1 < n; ++k) {
        if (tid == 0) {
            const int pk = piv[k];
            const double t = b[k];
            b[k] = b[pk];
            b[pk] = t;
        }
        __syncthreads();
        const double bk = b[k];
        const double *lk = Lt + (long long)k * n;
        for (int i = k + 1 + tid; i < n; i += blockDim.x) {
            const double prod = bk * lk[i];
            b[i] = b[i] - prod;
        }
        __syncthreads();
    }
    // backward, src/linalg.rs:292-297: one wave, serial chain in ascending j
    if (tid < 64) {
        const int lane = tid;
        for (int i = n - 1; i >= 0; --i) {
            const double *wi = W + (long long)i * n;
            double acc = b[i];
            for (int j0 = i + 1; j0 < n; j0 += 64) {
                const int j = j0 + lane;
                // lanes past the end contribute +0.0: acc - (+0.0) == acc for every acc
                const double p = (j < n) ? wi[j] * b[j] : 0.0;
#pragma unroll
                for (int l = 0; l < 64; ++l) acc = acc - dzg_readlane_f64(p, l);
            }
            acc = acc / wi[i];
            if (lane == 0) b[i] = acc;
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        }
        if (use_lds) {
            for (int i = lane; i < n; i += 64) b_glob[i] = s_b[i];
        }
    }
}

static void factorize_and_solve(int n, double *W, double *Lt, int *piv, double *urow, double *krow,
                                double *lcol, DzgCtl *ctl, double *b, hipStream_t st)
{
    for (int k = 0; k + 1 < n; ++k) {
        hipLaunchKernelGGL(k_lu_pivot, dim3(1), dim3(1024), 0, st, ctl, W, Lt, piv, urow, krow, lcol,
                           n, k);
        const int rem = n - k;
        hipLaunchKernelGGL(k_lu_update, dim3((rem + 63) / 64, (rem + 3) / 4), dim3(256), 0, st, ctl,
                           W, urow, krow, lcol, n, k);
    }
    const int use_lds = (size_t)n * sizeof(double) <= 64 * 1024;
    hipLaunchKernelGGL(k_lu_solve, dim3(1), dim3(1024), use_lds ? (size_t)n * sizeof(double) : 0, st,
                       ctl, W, Lt, piv, b, n, use_lds);
}

void dzg_launch_strict_solve(const DzgDev &d, int transposed, hipStream_t st)
{
    hipLaunchKernelGGL(k_gather_basis, dim3(d.m), dim3(256), 0, st, d.ctl, d.lu, d.m, d.A, d.lda,
                       d.basis, d.var_col, transposed, d.csc ? d.cptr : nullptr, d.ridx, d.cval);
    factorize_and_solve(d.m, d.lu, d.lt, d.piv, d.urow, d.krow, d.lcol, d.ctl,
                        transposed ? d.v : d.dx, st);
}

void dzg_launch_lu_raw(int n, double *lu, double *lt, int *piv, double *urow, double *krow,
                       double *lcol, DzgCtl *ctl, double *b, hipStream_t st)
{
    factorize_and_solve(n, lu, lt, piv, urow, krow, lcol, ctl, b, st);
}
