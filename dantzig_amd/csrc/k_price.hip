// k_price.hip -- launchers of the pricing kernels (kernels and design notes: k_price_kernels.h)
#include "k_price_kernels.h"

static int resolve(int kernel) { return kernel == DZG_PRICE_WAVE ? DZG_PRICE_WAVE : DZG_PRICE_SEQ; }

static void launch(int kernel, const DzgCtl *ctl, const double *A, long long lda, int m, int q,
                   const int *plist, const int *nonbasis, const int *var_col, const double *v,
                   double *dz, const double *z, const double *zbar, double *rz_r, int *rz_k,
                   double *rz_h, int col0, hipStream_t st)
{
    if (q <= 0) return;
    if (resolve(kernel) == DZG_PRICE_WAVE)
        hipLaunchKernelGGL((k_price_wave2<4>), dim3(DZG_PRICE_WAVE_BLOCKS), dim3(256), 0, st, ctl, A,
                           lda, m, q, plist, nonbasis, var_col, v, dz, z, zbar, rz_r, rz_k, rz_h, col0);
    else
        hipLaunchKernelGGL((k_price_seq2<16>), dim3(DZG_PRICE_SEQ_BLOCKS), dim3(256), 0, st, ctl, A,
                           lda, m, q, plist, nonbasis, var_col, v, dz, z, zbar, rz_r, rz_k, rz_h, col0);
}

// number of per-workgroup ratio partials the chosen kernel leaves in rz_r / rz_k
int dzg_price_partials(int kernel)
{
    if (kernel == DZG_PRICE_CSC_KERNEL) return DZG_PRICE_CSC_BLOCKS;
    return resolve(kernel) == DZG_PRICE_WAVE ? DZG_PRICE_WAVE_BLOCKS : DZG_PRICE_SEQ_BLOCKS;
}

static void launch_csc(const DzgDev &d, const int *plist, const double *z, const double *zbar,
                       double *rz_r, int *rz_k, double *rz_h, hipStream_t st)
{
    if (d.q <= 0) return;
    hipLaunchKernelGGL(k_price_csc, dim3(DZG_PRICE_CSC_BLOCKS), dim3(256), 0, st, d.ctl, d.cptr,
                       d.ridx, d.cval, d.q, plist, d.nonbasis, d.var_col, d.v, d.dz, z, zbar, rz_r,
                       rz_k, rz_h, d.col0);
}

// STRICT numerics: every nonbasic position, no fused ratio test
void dzg_launch_price(const DzgDev &d, int kernel, hipStream_t st)
{
    if (d.csc) {
        launch_csc(d, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, st);
        return;
    }
    launch(kernel, d.ctl, d.A, d.lda, d.m, d.q, nullptr, d.nonbasis, d.var_col, d.v, d.dz, nullptr,
           nullptr, nullptr, nullptr, nullptr, 0, st);
}

// FAST numerics: structural positions from plist, ratio-test partials for the dual step
void dzg_launch_price_fast(const DzgDev &d, int kernel, hipStream_t st)
{
    if (d.csc) {
        launch_csc(d, d.plist, d.z, d.zbar, d.rz_r, d.rz_k, d.rz_h, st);
        return;
    }
    launch(kernel, d.ctl, d.A, d.lda, d.m, d.q, d.plist, d.nonbasis, d.var_col, d.v, d.dz, d.z,
           d.zbar, d.rz_r, d.rz_k, d.rz_h, d.col0, st);
}

void dzg_launch_price_raw(int kernel, int m, long long lda, const double *A, const int *cols,
                          int ncols, const double *v, double *out, hipStream_t st)
{
    launch(kernel, nullptr, A, lda, m, ncols, nullptr, cols, nullptr, v, out, nullptr, nullptr,
           nullptr, nullptr, nullptr, 0, st);
}
