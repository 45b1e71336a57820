// lat_bench.hip -- dependent-chain latency of fp64 VALU ops on gfx950 (diagnostic tool)
#include <cstdio>
#include <hip/hip_runtime.h>
template <int MODE>
__global__ void k_chain(double *out, const double *in, int iters, long long *cyc)
{
    double acc = in[threadIdx.x], b = in[64 + threadIdx.x], c = in[128 + threadIdx.x];
    double acc2 = in[192 + threadIdx.x];
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 64; ++u) {
            if (MODE == 0) acc = acc + b;                         // dependent add chain
            if (MODE == 1) { double p = c * b; acc = acc + p; c = c + 1e-9; }  // mul feeding add
            if (MODE == 2) { acc = acc + b; acc2 = acc2 + c; }    // two independent chains
            if (MODE == 3) acc = fma(acc, b, c);                  // dependent fma chain
            if (MODE == 4) { acc = acc * b; }                     // dependent mul chain
        }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    out[threadIdx.x] = acc + acc2 + c;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
int main()
{
    double *in, *out; long long *cyc, h;
    hipMalloc(&in, 4096); hipMalloc(&out, 4096); hipMalloc(&cyc, 8);
    double hin[256]; for (int i = 0; i < 256; ++i) hin[i] = 1.0 + i * 1e-6;
    hipMemcpy(in, hin, sizeof(hin), hipMemcpyHostToDevice);
    const int iters = 1000;
    const char *names[] = {"dep add_f64", "mul->add (1 chain)", "2 indep add chains", "dep fma_f64", "dep mul_f64"};
#define RUN(M) hipLaunchKernelGGL(k_chain<M>, dim3(1), dim3(64), 0, 0, out, in, iters, cyc); hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost); printf("%-22s %.2f memtime-ticks per op-group (1 wave, 64 lanes)\n", names[M], (double)h / (iters * 64.0));
    RUN(0) RUN(1) RUN(2) RUN(3) RUN(4)
    // 16 active lanes only
    return 0;
}
