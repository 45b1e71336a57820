// k_drift.hip -- how far the CARRIED state of a FAST solve has drifted from the state the current
// basis defines, measured at a refactorisation.
//
// FAST numerics carries x, xbar, z, zbar from pivot to pivot (src/simplex.rs:262-265 does the same:
// the reference never recomputes them either, but its rounding is not FAST's).  The health monitor
// (max_pivot_error) watches the INVERSE -- dx_p from FTRAN against -dz_r from BTRAN + pricing --
// and a fresh inverse resets it; the carried vectors keep whatever rounding they have accumulated
// (mu of FAST and STRICT differ by 5e-12 after 1 750 pivots at 8192 rows while the
// monitor reads 8e-14).  Right after a refactorisation the eta file is empty and the inverse is as
// good as it gets, so the state can be recomputed from the data:
//
//      x^ = B^-1 b          xbar^ = B^-1 xbar0          z^_N = N^T (B^-T c_B) - c_N
//
// (b, xbar0: the x and xbar the solve STARTED with, on the slack basis, where x = rhs).  The largest
// relative difference  |x - x^|_inf / max(1, |x^|_inf)  (likewise xbar, z) is reported as
// dzg_result.state_drift and widens the near-tie tolerance: tau = max(tie_tol, 64 max_pivot_error,
// 4 state_drift) -- a decision whose margin is inside what the carried state is known to be off by
// is one the reference may take the other way.  The carried vectors are NOT replaced (the solve
// stays the same solve; replacing them would be a different rounding sequence, not the
// reference's either).
//
// One GPU, dense matrix, solves that start from the slack basis.  B^-1 b is FTRAN's row function
// with b as the column (fast_rows.h), N^T y the column-wise pricing pass with y as v.
#include "common.h"
#include "fast_rows.h"

// gathered copies of b and xbar0 in compact numbering (FTRAN's `ag`), zero-padded to even length
__global__ __launch_bounds__(256) void k_drift_gather(const DzgDev d, const double *__restrict__ b0,
                                                      const double *__restrict__ xb0,
                                                      double *__restrict__ agb, double *__restrict__ agx)
{
    const int k = d.ctl->ncompact, k2 = (k + 1) & ~1;
    for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < k2; c += gridDim.x * blockDim.x) {
        agb[c] = c < k ? b0[d.drow[c]] : 0.0;
        agx[c] = c < k ? xb0[d.drow[c]] : 0.0;
    }
}

// out[4 * block + {0,1,2,3}] = max |x - x^|, max |x^|, max |xbar - xbar^|, max |xbar^| over the block's rows
template <int LPR>
__global__ __launch_bounds__(256) void k_drift_x(const DzgDev d, const double *__restrict__ b0,
                                                 const double *__restrict__ xb0,
                                                 const double *__restrict__ agb,
                                                 const double *__restrict__ agx, double *__restrict__ out)
{
    constexpr int RPW = 64 / LPR;
    __shared__ double s_m[4][4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int sub = lane % LPR, grp = lane / LPR;
    const int wave_global = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int nwaves = (gridDim.x * blockDim.x) >> 6;
    const int k = d.ctl->ncompact, k2 = (k + 1) & ~1, m = d.m;
    double e0 = 0.0, e1 = 0.0, e2 = 0.0, e3 = 0.0;
    for (int i0 = wave_global * RPW; i0 < m; i0 += nwaves * RPW) {
        const int i = i0 + grp;
        // (the eta file is empty right after a refactorisation: neta = 0)
        double xa = fast_gemv_row<LPR>(i, m, k2, 0, d.binv, d.ldb, agb, d.U, d.ldw, d.beta, sub);
        double xb = fast_gemv_row<LPR>(i, m, k2, 0, d.binv, d.ldb, agx, d.U, d.ldw, d.beta, sub);
        if (i < m && sub == 0) {
            const int bc = d.bcode[i];
            if (bc < 0) { // a basic slack: its column of the inverse is the unit vector of this position
                xa += b0[-1 - bc];
                xb += xb0[-1 - bc];
            }
            const double da = fabs(d.x[i] - xa), db = fabs(d.xbar[i] - xb);
            e0 = da > e0 ? da : e0;
            e1 = fabs(xa) > e1 ? fabs(xa) : e1;
            e2 = db > e2 ? db : e2;
            e3 = fabs(xb) > e3 ? fabs(xb) : e3;
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        e0 = fmax(e0, __shfl_xor(e0, off, DZG_WAVE));
        e1 = fmax(e1, __shfl_xor(e1, off, DZG_WAVE));
        e2 = fmax(e2, __shfl_xor(e2, off, DZG_WAVE));
        e3 = fmax(e3, __shfl_xor(e3, off, DZG_WAVE));
    }
    if (lane == 0) {
        s_m[wave][0] = e0;
        s_m[wave][1] = e1;
        s_m[wave][2] = e2;
        s_m[wave][3] = e3;
    }
    __syncthreads();
    if (threadIdx.x < 4) {
        double v = 0.0;
        for (int w = 0; w < 4; ++w) v = fmax(v, s_m[w][threadIdx.x]);
        out[4 * blockIdx.x + threadIdx.x] = v;
    }
}

// y = B^-T c_B on the compact columns, in two deterministic stages (no atomics: the drift feeds
// the tolerance, and a solve must be reproducible):  part[chunk][c] = sum over the chunk's rows p
// of c_B[p] Binv0[p][c];  y[drow[c]] = the chunks' sums in order.  Rows outside R: y = c of the
// slack basic there.
#define DR_CHUNKS 64
__global__ __launch_bounds__(256) void k_drift_y_part(const DzgDev d, const double *__restrict__ cdev,
                                                      double *__restrict__ part)
{
    const int k = d.ctl->ncompact, m = d.m;
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    const int chunk = blockIdx.y;
    const int per = (m + DR_CHUNKS - 1) / DR_CHUNKS;
    const int p0 = chunk * per, p1 = p0 + per < m ? p0 + per : m;
    if (c >= k) return;
    double acc = 0.0;
    for (int p = p0; p < p1; ++p) acc = fma(cdev[d.basis[p]], d.binv[(long long)p * d.ldb + c], acc);
    part[(long long)chunk * d.ldw + c] = acc;
}

__global__ __launch_bounds__(256) void k_drift_y_sum(const DzgDev d, const double *__restrict__ cdev,
                                                     const double *__restrict__ part, double *__restrict__ y)
{
    const int k = d.ctl->ncompact, m = d.m;
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= m + 2) return;
    double v = 0.0;
    if (r < m) {
        const int c = d.dslot[r];
        if (c >= 0 && c < k) // (rows whose slack is basic: k_drift_y_slack, next launch)
            for (int ch = 0; ch < DR_CHUNKS; ++ch) v = v + part[(long long)ch * d.ldw + c];
    }
    y[r] = v; // (the two pads beyond m: zero, the pricing kernel reads pairs)
    (void)cdev;
}

// rows whose slack is basic: y[r] = c of that slack variable (position p holds it: bcode[p] = -1 - r)
__global__ __launch_bounds__(256) void k_drift_y_slack(const DzgDev d, const double *__restrict__ cdev,
                                                       double *__restrict__ y)
{
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= d.m) return;
    const int bc = d.bcode[p];
    if (bc < 0) y[-1 - bc] = cdev[d.basis[p]];
}

// out[2 * block + {0,1}] = max |z - z^|, max |z^|  with  z^_k = -dzy_k - c[nonbasis[k]]  (dzy = -N^T y)
__global__ __launch_bounds__(256) void k_drift_z(const DzgDev d, const double *__restrict__ cdev,
                                                 const double *__restrict__ dzy, double *__restrict__ out)
{
    __shared__ double s_m[4][2];
    double e0 = 0.0, e1 = 0.0;
    for (int kpos = blockIdx.x * blockDim.x + threadIdx.x; kpos < d.q; kpos += gridDim.x * blockDim.x) {
        const double zh = -dzy[kpos] - cdev[d.nonbasis[kpos]];
        const double dd = fabs(d.z[kpos] - zh);
        e0 = dd > e0 ? dd : e0;
        e1 = fabs(zh) > e1 ? fabs(zh) : e1;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        e0 = fmax(e0, __shfl_xor(e0, off, DZG_WAVE));
        e1 = fmax(e1, __shfl_xor(e1, off, DZG_WAVE));
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) {
        s_m[wave][0] = e0;
        s_m[wave][1] = e1;
    }
    __syncthreads();
    if (threadIdx.x < 2) {
        double v = 0.0;
        for (int w = 0; w < 4; ++w) v = fmax(v, s_m[w][threadIdx.x]);
        out[2 * blockIdx.x + threadIdx.x] = v;
    }
}

#define DR_BLOCKS 256

// Enqueues the whole measurement.  scratch: agb, agx [m + 2 each], part [DR_CHUNKS x ldw], y [m + 2],
// dzy [q], out [6 * DR_BLOCKS].  The host reads `out` and reduces (dzg_drift_reduce).
void dzg_launch_drift(const DzgDev &d, const double *b0, const double *xb0, const double *cdev,
                      double *agb, double *agx, double *part, double *y, double *dzy, double *out,
                      int k_bound, hipStream_t st)
{
    const int kb = k_bound > 0 ? k_bound : 1;
    hipLaunchKernelGGL(k_drift_gather, dim3((kb + 1 + 255) / 256), dim3(256), 0, st, d, b0, xb0, agb, agx);
    if (kb > 512)
        hipLaunchKernelGGL((k_drift_x<64>), dim3(DR_BLOCKS), dim3(256), 0, st, d, b0, xb0, agb, agx, out);
    else
        hipLaunchKernelGGL((k_drift_x<16>), dim3(DR_BLOCKS), dim3(256), 0, st, d, b0, xb0, agb, agx, out);
    hipLaunchKernelGGL(k_drift_y_part, dim3((kb + 255) / 256, DR_CHUNKS), dim3(256), 0, st, d, cdev, part);
    hipLaunchKernelGGL(k_drift_y_sum, dim3((d.m + 2 + 255) / 256), dim3(256), 0, st, d, cdev, part, y);
    hipLaunchKernelGGL(k_drift_y_slack, dim3((d.m + 255) / 256), dim3(256), 0, st, d, cdev, y);
    // dzy = -N^T y over every nonbasic position, column-wise (the sums' order does not matter here)
    dzg_launch_price_raw(DZG_PRICE_TREE, d.m, d.lda, d.A, d.nbcode, d.q, y, dzy, st);
    hipLaunchKernelGGL(k_drift_z, dim3(DR_BLOCKS), dim3(256), 0, st, d, cdev, dzy, out + 4 * DR_BLOCKS);
}

int dzg_drift_blocks(void) { return DR_BLOCKS; }
int dzg_drift_chunks(void) { return DR_CHUNKS; }
