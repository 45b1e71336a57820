// price_width_bench.hip -- k_price_tree<CW, DEPTH> when a wave has fewer than 16 columns (deep in a
// solve): which tiles-in-flight depth suits each pass width.  8192 rows, the columns picked at random
// from a 16384-column matrix (as the nonbasic structural columns of a solve are).
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -Wno-unused-result -I include -I dantzig_amd/csrc -o tools/price_width_bench tools/price_width_bench.hip
#include <algorithm>
#include <cstdio>
#include <cstdlib>
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
    const int m = 8192, nsall = 16384;
    const long long lda = m + 272;
    double *A, *v, *dz;
    int *cols;
    CK(hipMalloc(&A, sizeof(double) * lda * nsall));
    CK(hipMalloc(&v, sizeof(double) * (m + 2)));
    CK(hipMalloc(&dz, sizeof(double) * nsall));
    CK(hipMalloc(&cols, sizeof(int) * nsall));
    CK(hipMemset(v, 0, sizeof(double) * (m + 2)));
    hipLaunchKernelGGL(k_fill, dim3(2048), dim3(256), 0, 0, A, (size_t)lda * nsall, 1ull);
    hipLaunchKernelGGL(k_fill, dim3(32), dim3(256), 0, 0, v, (size_t)m, 2ull);
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::mt19937 rng(7);
    for (int ns : {16384, 14300, 12324, 11300, 10200, 9200, 8715, 8200, 7200, 6100, 5100}) {
        std::vector<int> all(nsall);
        for (int i = 0; i < nsall; ++i) all[i] = i;
        std::shuffle(all.begin(), all.end(), rng);
        all.resize(ns);
        CK(hipMemcpy(cols, all.data(), sizeof(int) * ns, hipMemcpyHostToDevice));
        const double gb = 8.0 * m * (double)ns / 1e9;
        printf("%d columns (%.2f per wave), %.0f MB\n", ns, ns / 1024.0, gb * 1e3);
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
            printf("   %-16s %7.1f us  %6.0f GB/s\n", name, us, gb / (us * 1e-6));
        };
#define T(CW, DEP, TP) time_it("tree<" #CW "," #DEP "," #TP ">", [&] { hipLaunchKernelGGL((k_price_tree<CW, DEP, TP>), dim3(256), dim3(256), 0, 0, (const DzgCtl *)nullptr, A, lda, m, ns, (const int *)nullptr, cols, (const int *)nullptr, v, dz, (const double *)nullptr, (const double *)nullptr, (double *)nullptr, (int *)nullptr, (double *)nullptr, 0, (const int *)nullptr); })
        const int pw = (ns + 1023) / 1024;
        if (pw == 16) { T(16, 2, 1); T(8, 2, 2); T(16, 3, 1); }
        if (pw == 14) { T(14, 2, 1); T(14, 3, 1); T(16, 2, 1); T(7, 4, 1); T(7, 2, 2); }
        if (pw == 13) { T(13, 2, 1); T(13, 3, 1); T(16, 2, 1); }
        if (pw == 12) { T(12, 3, 1); T(12, 4, 1); T(12, 2, 1); T(6, 3, 2); T(12, 2, 2); }
        if (pw == 10) { T(10, 3, 1); T(10, 4, 1); T(10, 2, 2); T(5, 4, 2); }
        if (pw == 9) { T(9, 3, 1); T(9, 4, 1); T(9, 5, 1); T(9, 2, 2); T(9, 3, 2); T(16, 2, 1); T(8, 4, 1); }
        if (pw == 8 || pw == 9) { T(8, 4, 1); T(8, 5, 1); T(8, 2, 2); T(8, 3, 2); }
        if (pw == 7 || pw == 8) { T(7, 4, 1); T(7, 5, 1); T(7, 3, 2); }
        if (pw == 6) { T(6, 5, 1); T(6, 6, 1); T(6, 3, 2); T(6, 2, 4); }
        if (pw == 5) { T(5, 6, 1); T(5, 8, 1); T(5, 4, 2); T(5, 2, 4); }
    }
    return 0;
}
