// k_vector.hip -- O(m + q) kernels of the iteration: status / first pivot, ratio test,
// step lengths + bookkeeping, vector updates.  gfx950, wave64.
//
// Reference: src/simplex.rs:274-306 (status), :423-437 (find_first_pivot),
// :439-461 (find_second_pivot), :253-268,:410-421,:464-468 (pivot, safe_divide),
// :239-251 (swap).  Compiled with -ffp-contract=off: every a*b+c below is two roundings,
// like the reference's f64 expressions.
#include "common.h"

// ---------------------------------------------------------------------------------
// find_first_pivot over one side: argmax_k -y_k / ybar_k over ybar_k > 0, as the reference's
// sequential reduce (src/simplex.rs:423-437): the accumulator starts as the FIRST surviving
// element and is replaced only by a strictly larger ratio.  Two consequences: ties go to the
// lowest k, and a NaN ratio on the first surviving element sticks (nothing is > NaN), while a
// NaN further down is skipped.  The reference does reach such states (x = -inf, xbar = +inf
// after a solve through a singular basis: fuzz seed 4251) and its verdict there -- a dual step
// with mu = NaN, hence "infeasible" -- is reproduced.
// ---------------------------------------------------------------------------------
__device__ __forceinline__ DzgCand scan_first(const double *__restrict__ y,
                                              const double *__restrict__ ybar, int len)
{
    DzgCand best, first;
    best.r = 0.0;
    best.k = -1;
    first.r = 0.0;
    first.k = -1;
    for (int k = threadIdx.x; k < len; k += blockDim.x) {
        double yb = ybar[k];
        if (yb > 0.0) {
            DzgCand c;
            c.r = dzg_div(-y[k], yb);
            c.k = k;
            if (c.r == c.r) best = dzg_better(best, c);
            if (first.k < 0) { // this thread's lowest surviving k; max of -k = min of k
                first.r = -(double)k;
                first.k = k;
            }
        }
    }
    first = dzg_block_best(first);
    best = dzg_block_best(best);
    if (first.k >= 0) {
        const double r0 = dzg_div(-y[first.k], ybar[first.k]);
        if (r0 != r0) { // the fold starts on a NaN and never leaves it
            best.r = r0;
            best.k = first.k;
        }
    }
    return best;
}

// find_second_pivot: argmax_k dy_k / (y_k + mu*ybar_k) over ratios > 0 (+inf included).
__device__ __forceinline__ DzgCand scan_second(double mu, const double *__restrict__ y,
                                               const double *__restrict__ ybar,
                                               const double *__restrict__ dy, int len)
{
    DzgCand best;
    best.r = 0.0;
    best.k = -1;
    for (int k = threadIdx.x; k < len; k += blockDim.x) {
        double scaled = mu * ybar[k];
        double den = y[k] + scaled;
        DzgCand c;
        c.r = dzg_div(dy[k], den);
        c.k = k;
        if (c.r > 0.0) best = dzg_better(best, c);
    }
    return dzg_block_best(best);
}

__global__ __launch_bounds__(1024) void k_status(DzgCtl *ctl, const double *x,
                                                 const double *xbar, int m, const double *z,
                                                 const double *zbar, int q, double eps)
{
    if (ctl->status != DZG_RUNNING) return;
    DzgCand cj = scan_first(z, zbar, q);
    DzgCand ci = scan_first(x, xbar, m);
    if (threadIdx.x != 0) return;
    int kind;
    double mu;
    if (cj.k >= 0 && ci.k >= 0) {
        const double primal = ci.r, dual = cj.r; // src/simplex.rs:280-281
        if (primal <= eps && dual <= eps) {
            ctl->status = DZG_OPTIMAL;
            return;
        }
        if (primal < dual) {
            kind = DZG_STEP_PRIMAL;
            mu = dual;
        } else {
            kind = DZG_STEP_DUAL;
            mu = primal;
        }
    } else if (cj.k >= 0) { // :294-298, no optimality test
        kind = DZG_STEP_PRIMAL;
        mu = cj.r;
    } else if (ci.k >= 0) { // :299-303
        kind = DZG_STEP_DUAL;
        mu = ci.r;
    } else {
        ctl->status = DZG_PANIC; // :304
        return;
    }
    if (ctl->iter >= ctl->iter_stop) {
        ctl->status = DZG_ITER_LIMIT;
        return;
    }
    if (m == 0) { // n - 1 underflow in Matrix::factorize: a reference panic path
        ctl->status = DZG_PANIC;
        return;
    }
    ctl->kind = kind;
    ctl->mu = mu;
    if (kind == DZG_STEP_PRIMAL) {
        ctl->enter_pos = cj.k;
        ctl->leave_pos = -1;
    } else {
        ctl->leave_pos = ci.k;
        ctl->enter_pos = -1;
    }
}

// Ratio test on the x side (primal step, picks the leaving position; none = Unbounded)
// or on the z side (dual step, picks the entering position; none = Infeasible).
__global__ __launch_bounds__(1024) void k_ratio(DzgCtl *ctl, int need_kind, const double *y,
                                                const double *ybar, const double *dy, int len)
{
    if (ctl->status != DZG_RUNNING || ctl->kind != need_kind) return;
    DzgCand c = scan_second(ctl->mu, y, ybar, dy, len);
    if (threadIdx.x != 0) return;
    if (need_kind == DZG_STEP_PRIMAL) {
        if (c.k < 0) ctl->status = DZG_UNBOUNDED; // src/simplex.rs:313
        ctl->leave_pos = c.k;
    } else {
        if (c.k < 0) ctl->status = DZG_INFEASIBLE; // :325
        ctl->enter_pos = c.k;
    }
}

// Step lengths, finiteness assert, swap, pivot log.  One thread: O(1) work.
__global__ void k_prepare(DzgCtl *ctl, const double *x, const double *xbar, const double *z,
                          const double *zbar, const double *dx, const double *dz, int *basis,
                          int *nonbasis, const int *var_col, int m, int q, int *log_kind,
                          int *log_enter, int *log_leave, double *log_mu, long long log_cap)
{
    if (ctl->status != DZG_RUNNING) return;
    const int p = ctl->leave_pos, r = ctl->enter_pos;
    int ok = 1;
    const double t = dzg_safe_divide(x[p], dx[p], &ok);
    const double s = dzg_safe_divide(z[r], dz[r], &ok);
    const double tbar = dzg_safe_divide(xbar[p], dx[p], &ok);
    const double sbar = dzg_safe_divide(zbar[r], dz[r], &ok);
    if (!ok) {
        ctl->status = DZG_PANIC; // assert in safe_divide, src/simplex.rs:466
        return;
    }
    ctl->t = t;
    ctl->s = s;
    ctl->tbar = tbar;
    ctl->sbar = sbar;
    const int i = basis[p], j = nonbasis[r];
    const long long it = ctl->iter;
    if (it < log_cap) {
        log_kind[it] = ctl->kind;
        log_enter[it] = j;
        log_leave[it] = i;
        log_mu[it] = ctl->mu;
    }
    // algorithmic bytes of this iteration's pricing pass:
    // 8*m per nonbasic structural column + v + (z, zbar, dz, one ratio pass)
    ctl->price_bytes += 8.0 * (double)m * (double)ctl->nb_struct + 8.0 * (double)m + 32.0 * (double)q;
    // swap, src/simplex.rs:243-247: each variable takes the other's slot
    basis[p] = j;
    nonbasis[r] = i;
    ctl->enter_var = j;
    ctl->leave_var = i;
    ctl->nb_struct += (var_col[i] >= 0 ? 1 : 0) - (var_col[j] >= 0 ? 1 : 0);
    ctl->iter = it + 1;
}

// pivot() x4, src/simplex.rs:262-265 + :410-421: v_k -= step*delta_k, v_pivot = step.
__global__ __launch_bounds__(256) void k_update_vectors(const DzgCtl *ctl, double *x,
                                                        double *xbar, double *z, double *zbar,
                                                        const double *dx, const double *dz,
                                                        int m, int q)
{
    if (ctl->status != DZG_RUNNING) return;
    const int p = ctl->leave_pos, r = ctl->enter_pos;
    const double t = ctl->t, s = ctl->s, tbar = ctl->tbar, sbar = ctl->sbar;
    const int stride = gridDim.x * blockDim.x;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < m; i += stride) {
        const double d = dx[i];
        const double a = t * d, b = tbar * d;
        x[i] = (i == p) ? t : x[i] - a;
        xbar[i] = (i == p) ? tbar : xbar[i] - b;
    }
    for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < q; k += stride) {
        const double d = dz[k];
        const double a = s * d, b = sbar * d;
        z[k] = (k == r) ? s : z[k] - a;
        zbar[k] = (k == r) ? sbar : zbar[k] - b;
    }
}

// constraints.column(j), src/linalg.rs:180-186: dense copy of the entering column.
__global__ __launch_bounds__(256) void k_load_column(const DzgCtl *ctl, int need_kind,
                                                     const double *A, long long lda,
                                                     const int *nonbasis, const int *var_col,
                                                     double *acol, int m, const long long *cptr,
                                                     const int *ridx, const double *cval)
{
    if (ctl->status != DZG_RUNNING) return;
    if (need_kind >= 0 && ctl->kind != need_kind) return;
    const int col = var_col[nonbasis[ctl->enter_pos]];
    if (cptr && col >= 0) { // sparse column (single workgroup): zero, then scatter
        for (int i = threadIdx.x; i < m; i += blockDim.x) acol[i] = 0.0;
        __syncthreads();
        for (long long e = cptr[col] + threadIdx.x; e < cptr[col + 1]; e += blockDim.x)
            acol[ridx[e]] = cval[e];
        return;
    }
    const int stride = gridDim.x * blockDim.x;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < m; i += stride)
        acol[i] = col >= 0 ? A[(long long)col * lda + i] : ((-1 - col) == i ? 1.0 : 0.0);
}

// e = unit(b_key[i]), src/simplex.rs:232-233
__global__ __launch_bounds__(256) void k_unit_rhs(const DzgCtl *ctl, double *v, int m)
{
    if (ctl->status != DZG_RUNNING) return;
    const int p = ctl->leave_pos;
    const int stride = gridDim.x * blockDim.x;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < m; i += stride)
        v[i] = (i == p) ? 1.0 : 0.0;
}

static inline int grid_for(int len, int block, int cap)
{
    int g = (len + block - 1) / block;
    if (g < 1) g = 1;
    return g > cap ? cap : g;
}

void dzg_launch_status(const DzgDev &d, hipStream_t st)
{
    hipLaunchKernelGGL(k_status, dim3(1), dim3(1024), 0, st, d.ctl, d.x, d.xbar, d.m, d.z, d.zbar,
                       d.q, d.eps);
}

void dzg_launch_ratio(const DzgDev &d, int need_kind, hipStream_t st)
{
    if (need_kind == DZG_STEP_PRIMAL)
        hipLaunchKernelGGL(k_ratio, dim3(1), dim3(1024), 0, st, d.ctl, need_kind, d.x, d.xbar, d.dx,
                           d.m);
    else
        hipLaunchKernelGGL(k_ratio, dim3(1), dim3(1024), 0, st, d.ctl, need_kind, d.z, d.zbar, d.dz,
                           d.q);
}

void dzg_launch_prepare(const DzgDev &d, hipStream_t st)
{
    hipLaunchKernelGGL(k_prepare, dim3(1), dim3(1), 0, st, d.ctl, d.x, d.xbar, d.z, d.zbar, d.dx,
                       d.dz, d.basis, d.nonbasis, d.var_col, d.m, d.q, d.log_kind, d.log_enter,
                       d.log_leave, d.log_mu, d.log_cap);
}

void dzg_launch_update_vectors(const DzgDev &d, hipStream_t st)
{
    int len = d.m > d.q ? d.m : d.q;
    hipLaunchKernelGGL(k_update_vectors, dim3(grid_for(len, 256, 1024)), dim3(256), 0, st, d.ctl,
                       d.x, d.xbar, d.z, d.zbar, d.dx, d.dz, d.m, d.q);
}

void dzg_launch_load_column(const DzgDev &d, int need_kind, hipStream_t st)
{
    hipLaunchKernelGGL(k_load_column, dim3(d.csc ? 1 : grid_for(d.m, 256, 256)), dim3(256), 0, st,
                       d.ctl, need_kind, d.A, d.lda, d.nonbasis, d.var_col, d.acol, d.m,
                       d.csc ? d.cptr : nullptr, d.ridx, d.cval);
}

void dzg_launch_unit_rhs(const DzgDev &d, hipStream_t st)
{
    hipLaunchKernelGGL(k_unit_rhs, dim3(grid_for(d.m, 256, 256)), dim3(256), 0, st, d.ctl, d.v, d.m);
}

// ---- single-function entry points for parity tests (device pointers in, host result) ----
__global__ __launch_bounds__(1024) void k_first_pivot_raw(const double *y, const double *ybar,
                                                          int len, int *out)
{
    DzgCand c = scan_first(y, ybar, len);
    if (threadIdx.x == 0) *out = c.k;
}

__global__ __launch_bounds__(1024) void k_second_pivot_raw(double mu, const double *y,
                                                           const double *ybar, const double *dy,
                                                           int len, int *out)
{
    DzgCand c = scan_second(mu, y, ybar, dy, len);
    if (threadIdx.x == 0) *out = c.k;
}

#define RAW_CHECK(e)                         \
    do {                                     \
        if ((e) != hipSuccess) return (int)DZG_E_DEVICE; \
    } while (0)

int dzg_run_first_pivot(int64_t len, const double *y, const double *ybar, int64_t *pos_out)
{
    double *dy_ = nullptr, *dyb = nullptr;
    int *dout = nullptr, h = -1;
    size_t bytes = sizeof(double) * (size_t)(len > 0 ? len : 1);
    RAW_CHECK(hipMalloc(&dy_, bytes));
    RAW_CHECK(hipMalloc(&dyb, bytes));
    RAW_CHECK(hipMalloc(&dout, sizeof(int)));
    if (len > 0) {
        RAW_CHECK(hipMemcpy(dy_, y, bytes, hipMemcpyHostToDevice));
        RAW_CHECK(hipMemcpy(dyb, ybar, bytes, hipMemcpyHostToDevice));
    }
    hipLaunchKernelGGL(k_first_pivot_raw, dim3(1), dim3(1024), 0, 0, dy_, dyb, (int)len, dout);
    RAW_CHECK(hipMemcpy(&h, dout, sizeof(int), hipMemcpyDeviceToHost));
    hipFree(dy_);
    hipFree(dyb);
    hipFree(dout);
    *pos_out = h;
    return 0;
}

int dzg_run_second_pivot(int64_t len, double mu, const double *y, const double *ybar,
                         const double *dy, int64_t *pos_out)
{
    double *d0 = nullptr, *d1 = nullptr, *d2 = nullptr;
    int *dout = nullptr, h = -1;
    size_t bytes = sizeof(double) * (size_t)(len > 0 ? len : 1);
    RAW_CHECK(hipMalloc(&d0, bytes));
    RAW_CHECK(hipMalloc(&d1, bytes));
    RAW_CHECK(hipMalloc(&d2, bytes));
    RAW_CHECK(hipMalloc(&dout, sizeof(int)));
    if (len > 0) {
        RAW_CHECK(hipMemcpy(d0, y, bytes, hipMemcpyHostToDevice));
        RAW_CHECK(hipMemcpy(d1, ybar, bytes, hipMemcpyHostToDevice));
        RAW_CHECK(hipMemcpy(d2, dy, bytes, hipMemcpyHostToDevice));
    }
    hipLaunchKernelGGL(k_second_pivot_raw, dim3(1), dim3(1024), 0, 0, mu, d0, d1, d2, (int)len,
                       dout);
    RAW_CHECK(hipMemcpy(&h, dout, sizeof(int), hipMemcpyDeviceToHost));
    hipFree(d0);
    hipFree(d1);
    hipFree(d2);
    hipFree(dout);
    *pos_out = h;
    return 0;
}
