// price_bench.hip -- microbenchmark of pricing-kernel variants (diagnostic tool, not shipped).
// usage: price_bench [m] [ns] [reps]
#include <cstdio>
#include <cstdlib>
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

template <int U>
__global__ __launch_bounds__(256) void k_stream(const double *__restrict__ a, size_t n2, double *out)
{
    const double2_t *p = reinterpret_cast<const double2_t *>(a);
    double acc = 0.0;
    size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    for (; i + (U - 1) * stride < n2; i += U * stride) {
        double2_t c[U];
#pragma unroll
        for (int u = 0; u < U; ++u) c[u] = __builtin_nontemporal_load(p + i + u * stride);
#pragma unroll
        for (int u = 0; u < U; ++u) acc += c[u].x + c[u].y;
    }
    if (acc == 1.2345) out[0] = acc;
}

// Experiment (round 2): the tree kernel's stream with TP adjacent 128-row tiles of a column fetched
// back to back (TP KB contiguous per column visit instead of 1 KB), same registers in flight as
// <CW * TP, DEPTH, 1>.  Raw mode only (every position a column), sums in the same order.
template <int CW, int DEPTH, int TP>
__global__ __launch_bounds__(256) void k_tree_tp(const double *__restrict__ A, long long lda, int m, int q,
                                                 const int *__restrict__ cols,
                                                 const double *__restrict__ v, double *__restrict__ dz)
{
    constexpr int TR = 128;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nw = gridDim.x * 4, wg = blockIdx.x * 4 + wave;
    const int base = q / nw, rem = q % nw;
    const int start = wg * base + (wg < rem ? wg : rem), cnt = base + (wg < rem ? 1 : 0);
    const int ntiles = (m + TR - 1) / TR, ngroups = (ntiles + TP - 1) / TP;
    const int lastpair = (int)lda - 2;
    const int lastv = ((m + 1) & ~1) - 2 >= 0 ? ((m + 1) & ~1) - 2 : 0;
    for (int c0 = 0; c0 < cnt; c0 += CW) {
        const int nc = (cnt - c0) < CW ? (cnt - c0) : CW;
        int mycode = -1;
        if (lane < nc) mycode = cols[start + c0 + lane];
        long long off[CW];
        int lastcode = 0;
#pragma unroll
        for (int l = 0; l < CW; ++l) {
            const int code_l = __builtin_amdgcn_readlane(mycode, l);
            if (code_l >= 0) lastcode = code_l;
            off[l] = (long long)(code_l >= 0 ? code_l : lastcode) * lda;
        }
        double2_t rg[DEPTH][TP][CW], vg[DEPTH][TP];
        double acc[CW];
#pragma unroll
        for (int l = 0; l < CW; ++l) acc[l] = 0.0;
        auto fetch = [&](int g, double2_t(&reg)[TP][CW], double2_t(&vreg)[TP]) {
#pragma unroll
            for (int l = 0; l < CW; ++l) // column-major: the TP tiles of a column back to back
#pragma unroll
                for (int u = 0; u < TP; ++u) {
                    const int row = (g * TP + u) * TR + 2 * lane;
                    const int rowc = row < lda ? row : lastpair;
                    reg[u][l] = __builtin_nontemporal_load(reinterpret_cast<const double2_t *>(A + off[l] + rowc));
                }
#pragma unroll
            for (int u = 0; u < TP; ++u) {
                const int row = (g * TP + u) * TR + 2 * lane;
                vreg[u] = *reinterpret_cast<const double2_t *>(v + (row < m ? row : lastv));
            }
        };
        auto consume = [&](int g, const double2_t(&reg)[TP][CW], const double2_t(&vreg)[TP]) {
#pragma unroll
            for (int u = 0; u < TP; ++u) {
                const bool inside = (g * TP + u) * TR + 2 * lane < m;
#pragma unroll
                for (int l = 0; l < CW; ++l) {
                    const double ax = inside ? reg[u][l].x : 0.0, ay = inside ? reg[u][l].y : 0.0;
                    acc[l] = fma(ax, vreg[u].x, acc[l]);
                    acc[l] = fma(ay, vreg[u].y, acc[l]);
                }
            }
        };
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) fetch(d < ngroups ? d : ngroups - 1, rg[d], vg[d]);
        int t = 0;
        for (; t + 2 * DEPTH - 1 < ngroups; t += DEPTH) {
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) {
                consume(t + d, rg[d], vg[d]);
                fetch(t + DEPTH + d, rg[d], vg[d]);
            }
        }
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            if (t + d < ngroups) {
                consume(t + d, rg[d], vg[d]);
                if (t + DEPTH + d < ngroups) fetch(t + DEPTH + d, rg[d], vg[d]);
            }
        }
#pragma unroll
        for (int d = 0; d < DEPTH; ++d)
            if (t + DEPTH + d < ngroups) consume(t + DEPTH + d, rg[d], vg[d]);
        double mine = 0.0;
#pragma unroll
        for (int l = 0; l < CW; ++l) {
            double s = acc[l];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
            if (lane == l) mine = s;
        }
        if (lane < nc && mycode >= 0) dz[start + c0 + lane] = -mine;
    }
}

int main(int argc, char **argv)
{
    int m = argc > 1 ? atoi(argv[1]) : 8192;
    int ns = argc > 2 ? atoi(argv[2]) : 16384;
    int reps = argc > 3 ? atoi(argv[3]) : 20;
    int ldpad = argc > 4 ? atoi(argv[4]) : 0;
    long long lda = (m + 15) / 16 * 16 + ldpad;
    size_t na = (size_t)lda * ns;
    double *A, *v, *dz, *ref;
    int *cols;
    CK(hipMalloc(&A, na * 8)); CK(hipMalloc(&v, (m + 16) * 8)); CK(hipMalloc(&dz, (ns + 4096) * 8)); CK(hipMalloc(&ref, ns * 8));
    CK(hipMalloc(&cols, ns * 4));
    hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, A, na, 1ull);
    CK(hipMemset(v, 0, (m + 16) * 8));
    hipLaunchKernelGGL(k_fill, dim3(64), dim3(256), 0, 0, v, (size_t)m, 2ull);
    std::vector<int> h(ns);
    for (int i = 0; i < ns; ++i) h[i] = i;
    CK(hipMemcpy(cols, h.data(), ns * 4, hipMemcpyHostToDevice));
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const double gbytes = 8.0 * m * (double)ns / 1e9;

    auto time_it = [&](const char *name, auto launch, bool check) {
        launch();
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        for (int r = 0; r < reps; ++r) launch();
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        double us = 1e3 * ms / reps;
        double maxdiff = -1;
        if (check) {
            std::vector<double> a(ns), b(ns);
            CK(hipMemcpy(a.data(), dz, ns * 8, hipMemcpyDeviceToHost));
            CK(hipMemcpy(b.data(), ref, ns * 8, hipMemcpyDeviceToHost));
            maxdiff = 0;
            for (int i = 0; i < ns; ++i) { double d = a[i] - b[i]; if (d < 0) d = -d; if (d > maxdiff) maxdiff = d; }
        }
        printf("%-30s %9.1f us  %7.1f GB/s  (%.1f%% of 8 TB/s)  maxdiff %.2e\n", name, us, gbytes / (us * 1e-6), 100.0 * gbytes / (us * 1e-6) / 8000.0, maxdiff);
        fflush(stdout);
    };
#define ARGS nullptr, A, lda, m, ns, nullptr, cols, nullptr, v
    hipLaunchKernelGGL((k_price_seq2<16>), dim3(256), dim3(256), 0, 0, ARGS, ref, nullptr, nullptr, nullptr, nullptr, nullptr);
    CK(hipDeviceSynchronize());
    time_it("stream U=4 (2048 blk)", [&] { hipLaunchKernelGGL((k_stream<4>), dim3(2048), dim3(256), 0, 0, A, na / 2, dz); }, false);
#define SEQ2(CW, BLK) time_it("seq2<" #CW "> x" #BLK, [&] { hipLaunchKernelGGL((k_price_seq2<CW>), dim3(BLK), dim3(256), 0, 0, ARGS, dz, nullptr, nullptr, nullptr, nullptr, nullptr); }, true)
    SEQ2(16, 256);
#define SEQ2D(CW, DEP, DBG, NT, BLK) time_it("seq2<" #CW "," #DEP "," #DBG "," #NT "> x" #BLK, [&] { hipLaunchKernelGGL((k_price_seq2<CW, DEP, DBG, NT>), dim3(BLK), dim3(256), 0, 0, ARGS, dz, nullptr, nullptr, nullptr, nullptr, nullptr); }, true)
    SEQ2D(16, 3, 0, true, 256); SEQ2D(16, 3, 1, true, 256); SEQ2D(8, 3, 0, true, 512); SEQ2D(8, 4, 0, true, 512); SEQ2D(16, 2, 0, true, 256); SEQ2D(16, 4, 0, true, 256);
    {
        hipLaunchKernelGGL((k_price_seq2<16, 3, 4, true>), dim3(256), dim3(256), 0, 0, ARGS, dz, nullptr, nullptr, nullptr, nullptr, nullptr);
        CK(hipDeviceSynchronize());
        std::vector<double> st(3 * 1024);
        CK(hipMemcpy(st.data(), dz + ns, st.size() * 8, hipMemcpyDeviceToHost));
        double a = 0, b = 0, c = 0;
        for (int w = 0; w < 1024; ++w) { a += st[3 * w]; b += st[3 * w + 1]; c += st[3 * w + 2]; }
        int nt = (m + 127) / 128;
        printf("stamps (avg cycles per tile per wave): park+wait %.0f  fetch %.0f  walk %.0f\n", a / 1024 / nt, b / 1024 / nt, c / 1024 / nt);
    }
#define WAVE2(U, BLK) time_it("wave2<" #U "> x" #BLK, [&] { hipLaunchKernelGGL((k_price_wave2<U>), dim3(BLK), dim3(256), 0, 0, ARGS, dz, nullptr, nullptr, nullptr, nullptr, nullptr); }, true)
    WAVE2(4, 2048); WAVE2(4, 4096);
    // round 2: the register-accumulator kernel in every shape (its results do not depend on it)
#define TREE(CW, DEP, BLK) time_it("tree<" #CW "," #DEP "> x" #BLK, [&] { hipLaunchKernelGGL((k_price_tree<CW, DEP>), dim3(BLK), dim3(256), 0, 0, ARGS, dz, nullptr, nullptr, nullptr, nullptr, nullptr); }, true)
    TREE(16, 2, 256); TREE(16, 3, 256); TREE(16, 4, 256); TREE(8, 4, 256); TREE(8, 4, 512); TREE(8, 6, 512);
    TREE(4, 8, 256); TREE(4, 8, 1024); TREE(4, 4, 1024); TREE(2, 16, 256); TREE(2, 8, 2048); TREE(1, 16, 2048);
    TREE(16, 2, 512); TREE(8, 8, 256);
    // adjacent tiles of a column back to back (2 / 4 KB per column visit)
#define TREETP(CW, DEP, TP, BLK) time_it("tree_tp<" #CW "," #DEP "," #TP "> x" #BLK, [&] { hipLaunchKernelGGL((k_tree_tp<CW, DEP, TP>), dim3(BLK), dim3(256), 0, 0, A, lda, m, ns, cols, v, dz); }, true)
    TREETP(16, 2, 1, 256); TREETP(8, 2, 2, 256); TREETP(8, 2, 2, 512); TREETP(4, 2, 4, 256); TREETP(4, 2, 4, 512); TREETP(4, 2, 4, 1024);
    TREETP(8, 3, 2, 256); TREETP(4, 4, 2, 512); TREETP(2, 2, 8, 1024); TREETP(16, 1, 2, 256);
    return 0;
}
