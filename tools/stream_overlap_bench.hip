// stream_overlap_bench.hip -- does a ONE-workgroup, 1024-thread kernel chain on a second (high
// priority) stream make progress while a grid of 256-thread workgroups fills the chip from another
// stream?  (diagnostic tool: decides whether the panel factorisation of the refactorisation, a chain
// of single-CU kernels, can run beside the trailing-matrix GEMM of the previous panel pair.)
//
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/stream_overlap_bench tools/stream_overlap_bench.hip
//   /tmp/stream_overlap_bench
//
// "filler": workgroups of 256 threads, 36 KB of LDS, ~128 VGPRs (the footprint of k_ref_gemm_lds),
// each busy for `spin` clocks; "chain": `steps` dependent launches of one 1024-thread workgroup,
// each busy for ~15 us.  Reported: the chain alone, the filler alone, both together (chain on a
// high-priority stream), and the chain's own span while the filler runs.
#include <chrono>
#include <cstdio>
#include <hip/hip_runtime.h>

#define CHECK(x)                                                                                     \
    do {                                                                                             \
        hipError_t e_ = (x);                                                                         \
        if (e_ != hipSuccess) {                                                                      \
            std::printf("%s: %s\n", #x, hipGetErrorString(e_));                                      \
            return 1;                                                                                \
        }                                                                                            \
    } while (0)

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 8))) void k_filler(
    double *out, long long spin)
{
    __shared__ double s_pad[36 * 128]; // 36 KB
    double acc[48];
#pragma unroll
    for (int i = 0; i < 48; ++i) acc[i] = threadIdx.x + i;
    s_pad[threadIdx.x] = acc[0];
    __syncthreads();
    const long long t0 = __builtin_amdgcn_s_memtime();
    while (__builtin_amdgcn_s_memtime() - t0 < spin) {
#pragma unroll
        for (int i = 0; i < 48; ++i) acc[i] = fma(acc[i], 1.0000001, s_pad[(threadIdx.x + i) & 255]);
    }
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < 48; ++i) s += acc[i];
    if (s == 12345.678) out[blockIdx.x] = s;
}

__global__ __launch_bounds__(1024) void k_chain_step(double *out, long long spin)
{
    double acc[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) acc[i] = threadIdx.x + i;
    const long long t0 = __builtin_amdgcn_s_memtime();
    while (__builtin_amdgcn_s_memtime() - t0 < spin) {
#pragma unroll
        for (int i = 0; i < 32; ++i) acc[i] = fma(acc[i], 1.0000001, 0.5);
        __syncthreads();
    }
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < 32; ++i) s += acc[i];
    if (s == 12345.678) out[threadIdx.x] = s;
}

static double ms_since(std::chrono::steady_clock::time_point t0)
{
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
}

int main()
{
    int lo = 0, hi = 0;
    CHECK(hipDeviceGetStreamPriorityRange(&lo, &hi));
    hipStream_t main_s, chain_s;
    CHECK(hipStreamCreateWithFlags(&main_s, hipStreamNonBlocking));
    CHECK(hipStreamCreateWithPriority(&chain_s, hipStreamNonBlocking, hi));
    double *out;
    CHECK(hipMalloc(&out, sizeof(double) * 65536));
    hipEvent_t c0, c1;
    CHECK(hipEventCreate(&c0));
    CHECK(hipEventCreate(&c1));
    const long long filler_spin = 200;   // 100-MHz clocks: 2 us per filler workgroup
    const long long step_spin = 1500;    // 15 us per chain step
    const int steps = 40, filler_wgs = 60000;
    std::printf("stream priorities: least %d, greatest %d\n", lo, hi);
    for (int rep = 0; rep < 2; ++rep) {
        // chain alone
        CHECK(hipDeviceSynchronize());
        auto t0 = std::chrono::steady_clock::now();
        for (int s = 0; s < steps; ++s) hipLaunchKernelGGL(k_chain_step, dim3(1), dim3(1024), 0, chain_s, out, step_spin);
        CHECK(hipStreamSynchronize(chain_s));
        const double chain_alone = ms_since(t0);
        // filler alone
        t0 = std::chrono::steady_clock::now();
        hipLaunchKernelGGL(k_filler, dim3(filler_wgs), dim3(256), 0, main_s, out, filler_spin);
        CHECK(hipStreamSynchronize(main_s));
        const double filler_alone = ms_since(t0);
        // both
        t0 = std::chrono::steady_clock::now();
        hipLaunchKernelGGL(k_filler, dim3(filler_wgs), dim3(256), 0, main_s, out, filler_spin);
        CHECK(hipEventRecord(c0, chain_s));
        for (int s = 0; s < steps; ++s) hipLaunchKernelGGL(k_chain_step, dim3(1), dim3(1024), 0, chain_s, out, step_spin);
        CHECK(hipEventRecord(c1, chain_s));
        CHECK(hipStreamSynchronize(chain_s));
        const double chain_done = ms_since(t0);
        CHECK(hipStreamSynchronize(main_s));
        const double both = ms_since(t0);
        float chain_span = 0.f;
        CHECK(hipEventElapsedTime(&chain_span, c0, c1));
        std::printf("chain alone %.3f ms (%d steps)   filler alone %.3f ms (%d workgroups)   together %.3f ms "
                    "(chain done after %.3f ms, its own span %.3f ms)\n",
                    chain_alone, steps, filler_alone, filler_wgs, both, chain_done, chain_span);
    }
    return 0;
}
