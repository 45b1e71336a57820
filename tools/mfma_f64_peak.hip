// mfma_f64_peak.hip -- on-box peak of v_mfma_f64_16x16x4_f64 (diagnostic tool, not shipped).
// "fp64 peak: not in the local guides; confirm by on-box microbenchmark
// before quoting a fraction".  Every wave runs NACC independent accumulator chains of MFMAs from
// registers (no memory traffic); the grid fills every SIMD with `waves` waves.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_f64_peak.hip -o /tmp/mfma_f64_peak && /tmp/mfma_f64_peak
#include <chrono>
#include <cstdio>
#include <hip/hip_runtime.h>

typedef double double4_t __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ __launch_bounds__(256) void k_mfma(double *out, int iters)
{
    double4_t acc[NACC];
#pragma unroll
    for (int j = 0; j < NACC; ++j) acc[j] = double4_t{0.0, 0.0, 0.0, 0.0};
    double a = 1.0 + 1e-9 * threadIdx.x, b = 1.0 - 1e-9 * threadIdx.x;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < NACC; ++j)
            acc[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[j], 0, 0, 0);
    }
    double s = 0.0;
#pragma unroll
    for (int j = 0; j < NACC; ++j) s += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NACC> static void run(int blocks, const char *what)
{
    double *out;
    (void)hipMalloc(&out, sizeof(double) * 256 * (size_t)blocks);
    const int iters = 20000;
    hipLaunchKernelGGL(k_mfma<NACC>, dim3(blocks), dim3(256), 0, 0, out, 100);
    (void)hipDeviceSynchronize();
    auto t0 = std::chrono::steady_clock::now();
    hipLaunchKernelGGL(k_mfma<NACC>, dim3(blocks), dim3(256), 0, 0, out, iters);
    (void)hipDeviceSynchronize();
    const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    const double flops = 2.0 * 16 * 16 * 4 * (double)NACC * iters * 4.0 * blocks; // 4 waves per block
    printf("%-34s %8.2f TFLOP/s  (%d workgroups x 4 waves, %d independent accumulators, %.1f ms)\n",
           what, flops / s / 1e12, blocks, NACC, s * 1e3);
    (void)hipFree(out);
}

int main()
{
    hipDeviceProp_t p;
    (void)hipGetDeviceProperties(&p, 0);
    printf("%s, %d CUs, %d MHz\n", p.gcnArchName, p.multiProcessorCount, p.clockRate / 1000);
    run<1>(256, "1 wave/SIMD, 1 chain");
    run<4>(256, "1 wave/SIMD, 4 chains");
    run<8>(256, "1 wave/SIMD, 8 chains");
    run<4>(512, "2 waves/SIMD, 4 chains");
    run<8>(512, "2 waves/SIMD, 8 chains");
    run<4>(1024, "4 waves/SIMD, 4 chains");
    return 0;
}
