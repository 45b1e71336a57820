// chain_barrier.h -- the device-wide barrier of the chain kernels (k_chain.hip), in a header of its
// own so that tools/barrier_timeout_test.hip can exercise exactly this code.
#pragma once
#include "common.h"

#define CH_THREADS 512
#define CH_AGCAP DZG_CHAIN_AGCAP
#define CH_GROUPS 8
#define CH_PAD 16 // counters 128 bytes apart

__device__ __forceinline__ void st_sc1(double *p, double v)
{
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_sc1(int *p, int v)
{
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double ld_sc1(const double *p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ int ld_sc1(const int *p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Device-wide barrier.  Returns false when it timed out (the caller returns at once).
__device__ __forceinline__ bool chain_barrier(DzgCtl *ctl, unsigned long long *bar,
                                              unsigned long long &gen)
{
    __shared__ int s_bar_ok;
    __syncthreads();
    if (threadIdx.x == 0) {
        gen += 1;
        const unsigned grp = blockIdx.x % CH_GROUPS;
        const unsigned long long members = (gridDim.x - grp + CH_GROUPS - 1) / CH_GROUPS;
        const unsigned long long ngroups = gridDim.x < CH_GROUPS ? gridDim.x : CH_GROUPS;
        __builtin_amdgcn_s_waitcnt(0); // this lane's sc1 stores have left the CU
        const unsigned long long old = __hip_atomic_fetch_add(
            bar + (size_t)CH_PAD * (1 + grp), 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (old + 1 == gen * members)
            __hip_atomic_fetch_add(bar, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long target = gen * ngroups;
        int ok = 0;
        for (int spin = 0; spin < (1 << 24); ++spin) { // ~140 ns per poll: gives up after ~2.4 s
            if (__hip_atomic_load(bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= target) {
                ok = 1;
                break;
            }
            __builtin_amdgcn_s_sleep(1);
        }
        if (!ok) { // a workgroup never arrived: give up, loudly (the host turns this into an error)
            ctl->status = DZG_PANIC;
            ctl->bar_timeout = 1;
        }
        s_bar_ok = ok;
    }
    __syncthreads();
    return s_bar_ok != 0;
}

