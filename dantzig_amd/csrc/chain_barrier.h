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

#define CH_BAR_WORDS (CH_PAD * (2 + CH_GROUPS)) // top counter, CH_GROUPS arrival counters, the frozen count
#define CH_POISON (1ull << 62)

// Device-wide barrier.  Returns false when it FAILED (the caller returns at once).
//
// A barrier fails when a workgroup gives up waiting (2^24 polls, ~2.4 s: the launch's workgroups
// were not all resident -- another kernel holds CUs or LDS of this device).  Failure is decided
// CONSISTENTLY: for one barrier instance either every workgroup passes or every workgroup fails,
// whatever the timing, so the state a failed launch leaves is the state before that barrier and
// the host can run the iteration again without barriers (engine.hip, chain_recover).  How: the
// decision is the order of two events on ONE atomic word, the top counter `bar[0]` --
//     C  the increment that completes the instance (count reaches gen * ngroups)
//     P  the first fetch_or of CH_POISON by a lane that gave up
// -- the lane whose fetch_or finds the word unpoisoned publishes the count it found (`frozen`,
// the count at P; increments that land after P are recognisable by the bit in the value they
// return and prove the instance incomplete at P).  Rule: pass iff the word was seen complete and
// unpoisoned, or frozen >= target.  Every later instance fails the same way (the bit stays until
// the host clears the counters), all of its workgroups alike.
//
// Fast path: unchanged -- one relaxed fetch_add per workgroup, one more by the last arrival of
// each residue class, polls of one word; the bit is only looked at when the loop ends.
// INVARIANT (do not break in an edit): everything that crosses a barrier is published by sc1
// stores of the SAME lane that arrives (thread 0, after s_waitcnt(0)), and read with sc1 loads.
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
        const unsigned long long target = gen * ngroups;
        unsigned long long *frozen = bar + (size_t)CH_PAD * (1 + CH_GROUPS);
        __builtin_amdgcn_s_waitcnt(0); // this lane's sc1 stores have left the CU
        const unsigned long long old = __hip_atomic_fetch_add(
            bar + (size_t)CH_PAD * (1 + grp), 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int ok = -1; // -1: undecided, look at the frozen count
        unsigned long long val = 0;
        if (old + 1 == gen * members) {
            val = __hip_atomic_fetch_add(bar, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (val & CH_POISON) ok = 0; // this arrival came after P: the instance was incomplete at P
            val = 0;
        }
        if (ok < 0) {
            int spin = 0;
            for (; spin < (1 << 24); ++spin) { // ~140 ns per poll: gives up after ~2.4 s
                val = __hip_atomic_load(bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (val >= target) break; // (a poisoned word is larger than any target)
                __builtin_amdgcn_s_sleep(1);
            }
            if (spin == (1 << 24)) { // give up: P, unless somebody was first
                val = __hip_atomic_fetch_or(bar, CH_POISON, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (!(val & CH_POISON)) {
                    __hip_atomic_store(frozen, val + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    ok = val >= target ? 1 : 0; // (complete after all, between the last poll and P)
                }
            } else if (!(val & CH_POISON)) {
                ok = 1;
            }
        }
        if (ok < 0) { // poisoned by somebody else: the count at P decides
            unsigned long long f = 0;
            for (int spin = 0; spin < (1 << 22) && f == 0; ++spin) {
                f = __hip_atomic_load(frozen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (f == 0) __builtin_amdgcn_s_sleep(1);
            }
            ok = (f != 0 && f - 1 >= target) ? 1 : 0;
        }
        if (!ok) { // loudly: the host sees bar_timeout and runs the iteration again without barriers
            __hip_atomic_store(&ctl->status, (int)DZG_PANIC, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&ctl->bar_timeout, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        s_bar_ok = ok;
    }
    __syncthreads();
    return s_bar_ok != 0;
}
