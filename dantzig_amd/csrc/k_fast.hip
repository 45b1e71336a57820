// k_fast.hip -- FAST numerics: the basis inverse stays resident in HBM.
//
// The reference refactorises B and B^T from scratch in every iteration
// (src/simplex.rs:228,234 -> src/linalg.rs:88-128), (4/3)m^3 flops per pivot.  On an
// MI355X the same two vectors are obtained from a resident row-major inverse Binv:
//
//   dx = B^-1 a_j          one GEMV over Binv, one wave per row, coalesced 16-B loads
//   v  = B^-T e_p          row p of Binv: a contiguous 8*m-byte read, no solve at all
//   pivot                  Binv <- E * Binv (rank-1): row_p /= dx_p, row_i -= dx_i * row_p
//
// Triangular solves are latency-bound chains of m/nb dependent steps on a GPU; the explicit
// inverse turns both solves into bandwidth-bound streaming, which is what 8 TB/s of HBM
// and 288 GB of capacity are for.  It also row-shards over GPUs (DESIGN.md "Multi-GPU").
// The initial basis is the slack identity (src/simplex.rs:190-201), so Binv starts as a
// permutation matrix and no factorisation is needed to start.
#include "common.h"

typedef double double2_t __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(256) void k_fast_init(double *__restrict__ binv, int m,
                                                   const int *__restrict__ basis,
                                                   const int *__restrict__ var_col)
{
    // B = [e_{r_0} e_{r_1} ...] (position p holds the slack of row r_p)  =>  Binv[p][r_p] = 1
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p < m) {
        const int code = var_col[basis[p]];
        binv[(long long)p * m + (-1 - code)] = 1.0;
    }
}

// dx = Binv * a_j.  One wave per row, rows strided over the grid.
__global__ __launch_bounds__(256) void k_fast_ftran(const DzgCtl *ctl, int need_kind,
                                                    const double *__restrict__ binv, int m,
                                                    const double *__restrict__ A, long long lda,
                                                    const int *__restrict__ nonbasis,
                                                    const int *__restrict__ var_col,
                                                    double *__restrict__ dx)
{
    if (ctl->status != DZG_RUNNING || ctl->kind != need_kind) return;
    const int code = var_col[nonbasis[ctl->enter_pos]];
    const int lane = threadIdx.x & 63;
    const int wave_global = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int nwaves = (gridDim.x * blockDim.x) >> 6;
    if (code < 0) { // entering slack: a_j = e_r, dx = column r of Binv
        const int r = -1 - code;
        for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < m; i += gridDim.x * blockDim.x)
            dx[i] = binv[(long long)i * m + r];
        return;
    }
    const double *a = A + (long long)code * lda;
    const int m2 = m & ~1;
    for (int i = wave_global; i < m; i += nwaves) {
        const double *row = binv + (long long)i * m;
        double a0 = 0.0, a1 = 0.0;
        if ((m & 1) == 0) {
            int c = 2 * lane;
            for (; c + 128 < m2; c += 256) {
                const double2_t r0 = *reinterpret_cast<const double2_t *>(row + c);
                const double2_t r1 = *reinterpret_cast<const double2_t *>(row + c + 128);
                const double2_t x0 = *reinterpret_cast<const double2_t *>(a + c);
                const double2_t x1 = *reinterpret_cast<const double2_t *>(a + c + 128);
                a0 = fma(r0.x, x0.x, a0);
                a1 = fma(r1.x, x1.x, a1);
                a0 = fma(r0.y, x0.y, a0);
                a1 = fma(r1.y, x1.y, a1);
            }
            for (; c < m2; c += 128) {
                const double2_t r0 = *reinterpret_cast<const double2_t *>(row + c);
                const double2_t x0 = *reinterpret_cast<const double2_t *>(a + c);
                a0 = fma(r0.x, x0.x, a0);
                a0 = fma(r0.y, x0.y, a0);
            }
        } else { // odd m: rows are only 8-B aligned
            for (int c = lane; c < m; c += 64) a0 = fma(row[c], a[c], a0);
        }
        double acc = a0 + a1;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, DZG_WAVE);
        if (lane == 0) dx[i] = acc;
    }
}

// v = row leave_pos of Binv
__global__ __launch_bounds__(256) void k_fast_btran(const DzgCtl *ctl,
                                                    const double *__restrict__ binv, int m,
                                                    double *__restrict__ v)
{
    if (ctl->status != DZG_RUNNING) return;
    const double *row = binv + (long long)ctl->leave_pos * m;
    for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < m; c += gridDim.x * blockDim.x)
        v[c] = row[c];
}

// w = v / dx_p : the new row p
__global__ __launch_bounds__(256) void k_fast_newrow(const DzgCtl *ctl, const double *__restrict__ v,
                                                     const double *__restrict__ dx,
                                                     double *__restrict__ w, int m)
{
    if (ctl->status != DZG_RUNNING) return;
    const double dxp = dx[ctl->leave_pos];
    for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < m; c += gridDim.x * blockDim.x)
        w[c] = v[c] / dxp;
}

// Binv[i][:] -= dx_i * w   (i != p),   Binv[p][:] = w.   One workgroup per 4 rows.
__global__ __launch_bounds__(256) void k_fast_rank1(const DzgCtl *ctl, double *__restrict__ binv,
                                                    int m, const double *__restrict__ dx,
                                                    const double *__restrict__ w)
{
    if (ctl->status != DZG_RUNNING) return;
    const int p = ctl->leave_pos;
    const int lane = threadIdx.x & 63;
    const int wave_global = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int nwaves = (gridDim.x * blockDim.x) >> 6;
    for (int i = wave_global; i < m; i += nwaves) {
        double *row = binv + (long long)i * m;
        const double g = dx[i];
        if (i == p) {
            for (int c = lane; c < m; c += 64) row[c] = w[c];
        } else if ((m & 1) == 0) {
            for (int c = 2 * lane; c < m; c += 128) {
                double2_t r = *reinterpret_cast<double2_t *>(row + c);
                const double2_t ww = *reinterpret_cast<const double2_t *>(w + c);
                r.x = fma(-g, ww.x, r.x);
                r.y = fma(-g, ww.y, r.y);
                *reinterpret_cast<double2_t *>(row + c) = r;
            }
        } else {
            for (int c = lane; c < m; c += 64) row[c] = fma(-g, w[c], row[c]);
        }
    }
}

static inline int cap_grid(long long blocks, int cap)
{
    if (blocks < 1) blocks = 1;
    return (int)(blocks > cap ? cap : blocks);
}

void dzg_launch_fast_init(const DzgDev &d, hipStream_t st)
{
    hipMemsetAsync(d.binv, 0, sizeof(double) * (size_t)d.m * (size_t)d.m, st);
    hipLaunchKernelGGL(k_fast_init, dim3((d.m + 255) / 256), dim3(256), 0, st, d.binv, d.m, d.basis,
                       d.var_col);
}

void dzg_launch_fast_ftran(const DzgDev &d, int need_kind, hipStream_t st)
{
    hipLaunchKernelGGL(k_fast_ftran, dim3(cap_grid((d.m + 3) / 4, 2048)), dim3(256), 0, st, d.ctl,
                       need_kind, d.binv, d.m, d.A, d.lda, d.nonbasis, d.var_col, d.dx);
}

void dzg_launch_fast_btran(const DzgDev &d, hipStream_t st)
{
    hipLaunchKernelGGL(k_fast_btran, dim3(cap_grid((d.m + 255) / 256, 256)), dim3(256), 0, st, d.ctl,
                       d.binv, d.m, d.v);
}

void dzg_launch_fast_update(const DzgDev &d, hipStream_t st)
{
    hipLaunchKernelGGL(k_fast_newrow, dim3(cap_grid((d.m + 255) / 256, 256)), dim3(256), 0, st,
                       d.ctl, d.v, d.dx, d.w, d.m);
    hipLaunchKernelGGL(k_fast_rank1, dim3(cap_grid((d.m + 3) / 4, 2048)), dim3(256), 0, st, d.ctl,
                       d.binv, d.m, d.dx, d.w);
}
