// price_rows_bench.hip -- k_price_rows (row-wise pricing of a dense matrix, csrc/k_price_kernels.h)
// on its own: k random rows of an 8192 x 16384 row-major matrix, the shapes the kernel can take
// (columns per lane, rows per register set, row groups, row stride).  HIP events over 50 launches.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -Wno-unused-result -I include -I dantzig_amd/csrc -o tools/price_rows_bench tools/price_rows_bench.hip
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#include "../dantzig_amd/csrc/k_price_kernels.h"

#define CK(e) do { hipError_t e_ = (e); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

__global__ void k_fill(double *a, size_t n, unsigned long long seed)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned long long z = seed + (i + 1) * 0x9E3779B97F4A7C15ull;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        z ^= z >> 31;
        a[i] = 2.0 * ((double)(z >> 11) * 0x1.0p-53) - 1.0;
    }
}

int main()
{
    const int m = 8192, ns = 16384;
    const long long ldmax = ns + 64;
    double *At, *v, *part;
    int *drow, *bcode;
    DzgCtl *ctl;
    CK(hipMalloc(&At, sizeof(double) * ldmax * m));
    CK(hipMalloc(&v, sizeof(double) * (m + 2)));
    CK(hipMalloc(&part, sizeof(double) * ldmax * 128));
    CK(hipMalloc(&drow, sizeof(int) * m));
    CK(hipMalloc(&bcode, sizeof(int) * m));
    CK(hipMalloc(&ctl, sizeof(DzgCtl)));
    CK(hipMemset(bcode, 0, sizeof(int) * m));
    hipLaunchKernelGGL(k_fill, dim3(2048), dim3(256), 0, 0, At, (size_t)ldmax * m, 1ull);
    hipLaunchKernelGGL(k_fill, dim3(32), dim3(256), 0, 0, v, (size_t)m, 2ull);
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::mt19937 rng(7);
    std::vector<int> rows(m);
    for (int i = 0; i < m; ++i) rows[i] = i;
    std::shuffle(rows.begin(), rows.end(), rng);
    CK(hipMemcpy(drow, rows.data(), sizeof(int) * m, hipMemcpyHostToDevice));
    for (int k : {100, 1044, 4049, 5400}) {
        DzgCtl c;
        memset(&c, 0, sizeof(c));
        c.status = DZG_RUNNING;
        c.ncompact = k;
        c.leave_pos = 0; // bcode[0] = 0: a structural variable leaves, no extra row
        CK(hipMemcpy(ctl, &c, sizeof(c), hipMemcpyHostToDevice));
        const double mb = 8.0 * k * (double)ns / 1e6;
        printf("k = %d rows, %.0f MB\n", k, mb);
        auto time_it = [&](const char *name, auto launch) {
            launch();
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0));
            for (int r = 0; r < 50; ++r) launch();
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms = 0;
            CK(hipEventElapsedTime(&ms, e0, e1));
            const double us = 1e3 * ms / 50;
            printf("  %-64s %8.2f us  %7.0f GB/s\n", name, us, mb / us * 1e3);
        };
        for (long long ldt : {(long long)ns, (long long)ns + 32}) {
            char name[160];
#define VAR(VEC, PIPE, GM)                                                                             \
    snprintf(name, sizeof name, "ldt %lld, %d columns per lane, %2d rows per set, <= %3d groups", ldt, VEC, \
             PIPE, GM);                                                                                \
    time_it(name, [&] {                                                                                \
        hipLaunchKernelGGL((k_price_rows<VEC, PIPE>), dim3((unsigned)((ldt + 256 * VEC - 1) / (256 * VEC)), GM), \
                           dim3(256), 0, 0, ctl, 1 << 30, At, ldt, drow, bcode, v, part); /* v stands in for the compact copy */              \
    });
            VAR(4, 8, 32)
            VAR(4, 8, 64)
            VAR(4, 8, 128)
            VAR(4, 4, 32)
            VAR(4, 4, 64)
            VAR(2, 8, 32)
            VAR(2, 8, 64)
            VAR(2, 16, 32)
            VAR(2, 16, 64)
#undef VAR
        }
    }
    return 0;
}
