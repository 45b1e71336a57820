// barrier_timeout_test.hip -- the chain kernels' device-wide barrier (csrc/chain_barrier.h) when a
// workgroup never arrives: every other workgroup must give up after its bounded number of polls,
// mark the control block (status DZG_PANIC, bar_timeout) and reach its exit -- no hang.  Also the
// healthy case: 1000 barriers with all workgroups, a value handed round the ring through sc1 stores.
//   hipcc --offload-arch=gfx950 -O3 -I include -I dantzig_amd/csrc -o tools/barrier_timeout_test tools/barrier_timeout_test.hip
#include <chrono>
#include <cstdio>
#include <cstring>

#include "../dantzig_amd/csrc/chain_barrier.h"

__global__ __launch_bounds__(CH_THREADS) void k_test(DzgCtl *ctl, unsigned long long *bar, double *buf,
                                                     int rounds, int absent, int *errors)
{
    unsigned long long gen = ctl->bar_gen;
    int bad = 0;
    for (int it = 0; it < rounds; ++it) {
        if (threadIdx.x == 0) st_sc1(buf + blockIdx.x, (double)(it * 1000 + (int)blockIdx.x));
        if ((int)blockIdx.x == absent) return; // this workgroup leaves without arriving
        if (!chain_barrier(ctl, bar, gen)) return;
        const int other = ((int)blockIdx.x + 1) % (int)gridDim.x;
        if (ld_sc1(buf + other) != (double)(it * 1000 + other)) ++bad;
        if (!chain_barrier(ctl, bar, gen)) return;
    }
    if (threadIdx.x == 0 && bad) atomicAdd(errors, bad);
    if (blockIdx.x == 0 && threadIdx.x == 0) ctl->bar_gen = gen;
}

#define CK(e) do { hipError_t e_ = (e); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

int main()
{
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int grid = prop.multiProcessorCount > 256 ? 256 : prop.multiProcessorCount;
    DzgCtl *ctl;
    unsigned long long *bar;
    double *buf;
    int *errors;
    CK(hipMalloc(&ctl, sizeof(DzgCtl)));
    CK(hipMalloc(&bar, sizeof(unsigned long long) * CH_BAR_WORDS));
    CK(hipMalloc(&buf, sizeof(double) * 1024));
    CK(hipMalloc(&errors, sizeof(int)));
    DzgCtl h;
    int herr = 0;
    for (int absent : {-1, grid / 2}) {
        std::memset(&h, 0, sizeof(h));
        h.status = DZG_RUNNING;
        CK(hipMemcpy(ctl, &h, sizeof(h), hipMemcpyHostToDevice));
        CK(hipMemset(bar, 0, sizeof(unsigned long long) * CH_BAR_WORDS));
        CK(hipMemset(errors, 0, sizeof(int)));
        const int rounds = absent < 0 ? 1000 : 1;
        auto t0 = std::chrono::steady_clock::now();
        hipLaunchKernelGGL(k_test, dim3(grid), dim3(CH_THREADS), 0, 0, ctl, bar, buf, rounds, absent, errors);
        CK(hipDeviceSynchronize());
        auto t1 = std::chrono::steady_clock::now();
        CK(hipMemcpy(&h, ctl, sizeof(h), hipMemcpyDeviceToHost));
        CK(hipMemcpy(&herr, errors, sizeof(int), hipMemcpyDeviceToHost));
        const double ms = std::chrono::duration<double, std::milli>(t1 - t0).count();
        if (absent < 0)
            printf("%d workgroups of %d, all arriving: %d barriers in %.2f ms (%.2f us each), status %d, bar_timeout %d, "
                   "barriers counted %llu, stale reads %d\n",
                   grid, CH_THREADS, 2 * rounds, ms, 1e3 * ms / (2 * rounds), h.status, h.bar_timeout, h.bar_gen, herr);
        else
            printf("%d workgroups, workgroup %d never arrives: the kernel ended by itself after %.0f ms, status %d (DZG_PANIC = %d), "
                   "bar_timeout %d\n",
                   grid, absent, ms, h.status, (int)DZG_PANIC, h.bar_timeout);
    }
    return 0;
}
