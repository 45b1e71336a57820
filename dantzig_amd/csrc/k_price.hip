// k_price.hip -- launchers of the pricing kernels (kernels: k_price_kernels.h)
#include "k_price_kernels.h"

static void launch(int kernel, const DzgCtl *ctl, const double *A, long long lda, int m, int ncols,
                   const int *nonbasis, const int *var_col, const double *v, double *dz,
                   hipStream_t st)
{
    if (ncols <= 0) return;
    if (kernel == DZG_PRICE_AUTO) kernel = DZG_PRICE_SEQ;
    if (kernel == DZG_PRICE_WAVE) {
        int blocks = (ncols + 3) / 4;
        if (blocks > 2048) blocks = 2048;
        hipLaunchKernelGGL(k_price_wave, dim3(blocks), dim3(256), 0, st, ctl, A, lda, m, ncols,
                           nonbasis, var_col, v, dz);
    } else {
        constexpr int C = 32, TR = 128;
        hipLaunchKernelGGL((k_price_seq<C, TR>), dim3((ncols + C - 1) / C), dim3(256), 0, st, ctl, A,
                           lda, m, ncols, nonbasis, var_col, v, dz);
    }
}

void dzg_launch_price(const DzgDev &d, int kernel, hipStream_t st)
{
    launch(kernel, d.ctl, d.A, d.lda, d.m, d.q, d.nonbasis, d.var_col, d.v, d.dz, st);
}

void dzg_launch_price_raw(int kernel, int m, long long lda, const double *A, const int *cols,
                          int ncols, const double *v, double *out, hipStream_t st)
{
    launch(kernel, nullptr, A, lda, m, ncols, cols, nullptr, v, out, st);
}
