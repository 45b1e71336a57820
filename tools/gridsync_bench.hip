// gridsync_bench.hip -- cost of a device-wide barrier inside one persistent kernel versus a
// kernel boundary on gfx950 (diagnostic tool; decides whether fusing the latency-bound basis
// kernels of one pivot into a persistent kernel can pay).
#include <chrono>
#include <cstdio>
#include <hip/hip_runtime.h>

// sense-free monotone barrier: every workgroup adds 1, waits until the counter reaches
// (generation+1)*gridDim.  Release/acquire at agent scope so data written before the barrier by
// any XCD is visible after it.
__device__ __forceinline__ void grid_barrier(unsigned *counter, unsigned &gen)
{
    __syncthreads();
    if (threadIdx.x == 0) {
        gen += 1;
        const unsigned target = gen * gridDim.x;
        __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target)
            __builtin_amdgcn_s_sleep(1);
    }
    __syncthreads();
}

__global__ __launch_bounds__(256) void k_persist(unsigned *counter, double *buf, int iters,
                                                 int *errors)
{
    unsigned gen = 0;
    const int nb = gridDim.x;
    int bad = 0;
    for (int it = 0; it < iters; ++it) {
        // every block publishes a value, then reads the value of the block "opposite" to it
        if (threadIdx.x == 0)
            __hip_atomic_store(&buf[blockIdx.x], (double)(it * 1000 + blockIdx.x), __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
        grid_barrier(counter, gen);
        const int other = (blockIdx.x + nb / 2 + 3) % nb;
        const double got = __hip_atomic_load(&buf[other], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (got != (double)(it * 1000 + other)) bad++;
        grid_barrier(counter, gen); // buf may be overwritten only after everybody has read
    }
    if (threadIdx.x == 0 && bad) atomicAdd(errors, bad);
}

// The same exchange WITHOUT release/acquire fences: the published values travel as agent-scope
// relaxed atomic stores / loads (sc1: past the XCD's L2, coherent at the memory side), the counter
// is a relaxed atomic, and program order is kept with s_waitcnt.  An agent-scope release on this
// chip writes back the whole L2 of the XCD (the eight L2s are not coherent with each other), which
// is what the fenced barrier above pays for; data that is published explicitly does not need it.
__device__ __forceinline__ void grid_barrier_relaxed(unsigned *counter, unsigned &gen)
{
    __syncthreads();
    if (threadIdx.x == 0) {
        gen += 1;
        const unsigned target = gen * gridDim.x;
        __builtin_amdgcn_s_waitcnt(0); // my sc1 stores have left
        __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target)
            __builtin_amdgcn_s_sleep(1);
    }
    __syncthreads();
}

__global__ __launch_bounds__(256) void k_persist_relaxed(unsigned *counter, double *buf, int iters,
                                                         int *errors)
{
    unsigned gen = 0;
    const int nb = gridDim.x;
    int bad = 0;
    for (int it = 0; it < iters; ++it) {
        if (threadIdx.x == 0)
            __hip_atomic_store(&buf[blockIdx.x], (double)(it * 1000 + blockIdx.x), __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
        grid_barrier_relaxed(counter, gen);
        const int other = (blockIdx.x + nb / 2 + 3) % nb;
        const double got = __hip_atomic_load(&buf[other], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (got != (double)(it * 1000 + other)) bad++;
        grid_barrier_relaxed(counter, gen);
    }
    if (threadIdx.x == 0 && bad) atomicAdd(errors, bad);
}

// The form csrc/k_chain.hip uses: arrivals counted on eight counters (workgroup index mod 8), the
// last arrival of a class bumps the top counter everybody polls.  64-bit counters, 128 B apart.
__device__ __forceinline__ void grid_barrier_sharded(unsigned long long *bar, unsigned long long &gen)
{
    __syncthreads();
    if (threadIdx.x == 0) {
        gen += 1;
        const unsigned grp = blockIdx.x % 8;
        const unsigned long long members = (gridDim.x - grp + 7) / 8;
        const unsigned long long ngroups = gridDim.x < 8 ? gridDim.x : 8;
        __builtin_amdgcn_s_waitcnt(0);
        const unsigned long long old = __hip_atomic_fetch_add(bar + 16 * (1 + grp), 1ull, __ATOMIC_RELAXED,
                                                              __HIP_MEMORY_SCOPE_AGENT);
        if (old + 1 == gen * members)
            __hip_atomic_fetch_add(bar, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        while (__hip_atomic_load(bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < gen * ngroups)
            __builtin_amdgcn_s_sleep(1);
    }
    __syncthreads();
}

template <int THREADS>
__global__ __launch_bounds__(THREADS) void k_persist_sharded(unsigned long long *bar, double *buf,
                                                             int iters, int *errors)
{
    unsigned long long gen = 0;
    const int nb = gridDim.x;
    int bad = 0;
    for (int it = 0; it < iters; ++it) {
        if (threadIdx.x == 0)
            __hip_atomic_store(&buf[blockIdx.x], (double)(it * 1000 + blockIdx.x), __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
        grid_barrier_sharded(bar, gen);
        const int other = (blockIdx.x + nb / 2 + 3) % nb;
        const double got = __hip_atomic_load(&buf[other], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (got != (double)(it * 1000 + other)) bad++;
        grid_barrier_sharded(bar, gen);
    }
    if (threadIdx.x == 0 && bad) atomicAdd(errors, bad);
}

__global__ __launch_bounds__(256) void k_small(double *buf, int it)
{
    if (threadIdx.x == 0) buf[blockIdx.x] = buf[(blockIdx.x + 7) % gridDim.x] + it;
}

int main()
{
    unsigned *counter;
    double *buf;
    int *errors, herr = 0;
    hipMalloc(&counter, 4);
    hipMalloc(&buf, 4096 * 8);
    hipMalloc(&errors, 4);
    hipMemset(buf, 0, 4096 * 8);
    for (int blocks : {64, 256, 512}) {
        hipMemset(counter, 0, 4);
        hipMemset(errors, 0, 4);
        const int iters = 2000;
        hipLaunchKernelGGL(k_persist, dim3(blocks), dim3(256), 0, 0, counter, buf, 10, errors);
        hipDeviceSynchronize();
        hipMemset(counter, 0, 4);
        auto t0 = std::chrono::steady_clock::now();
        hipLaunchKernelGGL(k_persist, dim3(blocks), dim3(256), 0, 0, counter, buf, iters, errors);
        hipDeviceSynchronize();
        auto t1 = std::chrono::steady_clock::now();
        hipMemcpy(&herr, errors, 4, hipMemcpyDeviceToHost);
        printf("persistent kernel, %3d workgroups: %.2f us per grid barrier (%d stale reads)\n", blocks,
               std::chrono::duration<double, std::micro>(t1 - t0).count() / (2.0 * iters), herr);
    }
    for (int blocks : {64, 256, 512}) {
        hipMemset(counter, 0, 4);
        hipMemset(errors, 0, 4);
        const int iters = 2000;
        hipLaunchKernelGGL(k_persist_relaxed, dim3(blocks), dim3(256), 0, 0, counter, buf, 10, errors);
        hipDeviceSynchronize();
        hipMemset(counter, 0, 4);
        auto t0 = std::chrono::steady_clock::now();
        hipLaunchKernelGGL(k_persist_relaxed, dim3(blocks), dim3(256), 0, 0, counter, buf, iters, errors);
        hipDeviceSynchronize();
        auto t1 = std::chrono::steady_clock::now();
        hipMemcpy(&herr, errors, 4, hipMemcpyDeviceToHost);
        printf("persistent kernel, %3d workgroups, NO fences (sc1 data, relaxed counter): %.2f us per grid barrier (%d stale reads)\n",
               blocks, std::chrono::duration<double, std::micro>(t1 - t0).count() / (2.0 * iters), herr);
    }
    unsigned long long *bar;
    hipMalloc(&bar, 16 * 9 * 8);
    for (int threads : {256, 512}) {
        for (int blocks : {64, 128, 256, 512}) {
            if (threads == 512 && blocks == 512) continue;
            hipMemset(bar, 0, 16 * 9 * 8);
            hipMemset(errors, 0, 4);
            const int iters = 2000;
            auto launch = [&](int n) {
                if (threads == 256)
                    hipLaunchKernelGGL(k_persist_sharded<256>, dim3(blocks), dim3(256), 0, 0, bar, buf, n, errors);
                else
                    hipLaunchKernelGGL(k_persist_sharded<512>, dim3(blocks), dim3(512), 0, 0, bar, buf, n, errors);
            };
            launch(10);
            hipDeviceSynchronize();
            hipMemset(bar, 0, 16 * 9 * 8);
            auto t0 = std::chrono::steady_clock::now();
            launch(iters);
            hipDeviceSynchronize();
            auto t1 = std::chrono::steady_clock::now();
            hipMemcpy(&herr, errors, 4, hipMemcpyDeviceToHost);
            printf("persistent kernel, %3d workgroups of %d, NO fences, eight arrival counters + one top counter: %.2f us per grid barrier (%d stale reads)\n",
                   blocks, threads, std::chrono::duration<double, std::micro>(t1 - t0).count() / (2.0 * iters), herr);
        }
    }
    for (int blocks : {64, 256}) {
        const int iters = 4000;
        for (int i = 0; i < 100; ++i) hipLaunchKernelGGL(k_small, dim3(blocks), dim3(256), 0, 0, buf, i);
        hipDeviceSynchronize();
        auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(k_small, dim3(blocks), dim3(256), 0, 0, buf, i);
        hipDeviceSynchronize();
        auto t1 = std::chrono::steady_clock::now();
        printf("back-to-back tiny kernels, %3d workgroups: %.2f us per kernel boundary\n", blocks,
               std::chrono::duration<double, std::micro>(t1 - t0).count() / iters);
    }
    return 0;
}
