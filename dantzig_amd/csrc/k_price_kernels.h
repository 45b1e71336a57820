// k_price.hip -- the pricing pass  dz = -N^T v  (src/linalg.rs:199-207 neg_t_dot over
// collect_columns(n), called at src/simplex.rs:235).  This is the HBM-roofline kernel:
// it streams every nonbasic structural column of A once per iteration (8*m bytes each).
//
// Two kernels, same inputs and outputs:
//
//  k_price_seq  one LANE per column.  A C x TR tile (C columns, TR rows, column segments
//               of 1 KiB fetched by fully coalesced 16-B/lane loads) is staged in LDS with
//               a padded, bank-conflict-free column stride, -v for the same rows sits next
//               to it, and lane c then walks its column top to bottom: acc = acc + a*(-v),
//               product and sum rounded separately, rows ascending.  That is exactly the
//               reference's summation order, so dz is BIT-IDENTICAL to neg_t_dot.  The
//               serial chain costs ~TR*12 cycles per tile against ~6-13k cycles of HBM time
//               for the same tile, so the kernel stays bandwidth-bound.
//
//  k_price_wave one WAVE per column, lane-strided partial sums, xor-shuffle tree.  Plain
//               streaming reduction; sums in a different order (not bit-identical).
//
// Column codes: >= 0 structural column index, < 0 unit (slack) column of row -1-code,
// whose "dot product" is 0.0 + 1.0 * -v[row] and costs no matrix bytes.
#pragma once
#include "common.h"

typedef double double2_t __attribute__((ext_vector_type(2)));

__device__ __forceinline__ int price_code(const int *__restrict__ nonbasis,
                                          const int *__restrict__ var_col, int pos)
{
    const int v = nonbasis[pos];
    return var_col ? var_col[v] : v;
}

// ---------------------------------------------------------------------------------
// k_price_seq<C, TR>: 256 threads, C columns per workgroup, tiles of TR rows.
// ---------------------------------------------------------------------------------
template <int C, int TR>
__global__ __launch_bounds__(256) void k_price_seq(const DzgCtl *ctl, const double *__restrict__ A,
                                                   long long lda, int m, int ncols,
                                                   const int *__restrict__ nonbasis,
                                                   const int *__restrict__ var_col,
                                                   const double *__restrict__ v,
                                                   double *__restrict__ dz)
{
    static_assert(TR % 128 == 0, "a wave fetches 128 rows (1 KiB) of one column per load");
    constexpr int PAD = 2;                 // column stride TR+2 doubles: lane c starts 4c banks on
    constexpr int CH = TR / 128;           // 1-KiB chunks per column and tile
    constexpr int NL = C * CH / 4;         // loads per thread and tile (4 waves)
    static_assert(NL >= 1, "tile too small");
    __shared__ __attribute__((aligned(16))) double tile[C][TR + PAD];
    __shared__ __attribute__((aligned(16))) double negv[TR];
    __shared__ int s_code[C];

    if (ctl && ctl->status != DZG_RUNNING) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int pos0 = blockIdx.x * C;
    if (tid < C) s_code[tid] = (pos0 + tid < ncols) ? price_code(nonbasis, var_col, pos0 + tid) : -1;
    __syncthreads();

    // loader geometry: wave-instruction wi covers column wi / CH, rows (wi % CH)*128 + 2*lane
    long long goff[NL]; // element offset of this thread's load inside A, or -1: nothing to fetch
    int lcol[NL], lrow[NL];
#pragma unroll
    for (int l = 0; l < NL; ++l) {
        const int wi = l * 4 + wave;
        lcol[l] = wi / CH;
        lrow[l] = (wi % CH) * 128 + 2 * lane;
        const int code = s_code[lcol[l]];
        goff[l] = (code >= 0) ? (long long)code * lda + lrow[l] : -1;
    }

    const int ntiles = (m + TR - 1) / TR;
    double2_t reg[NL];
    double2_t vreg = {0.0, 0.0};

    auto fetch = [&](int t) {
        const int row0 = t * TR;
#pragma unroll
        for (int l = 0; l < NL; ++l) {
            double2_t val = {0.0, 0.0};
            // lda is a multiple of 16 and rows >= m are zero-padded, so a 16-B load that
            // starts below m never leaves the column
            if (goff[l] >= 0 && row0 + lrow[l] < m)
                val = *reinterpret_cast<const double2_t *>(A + goff[l] + row0);
            reg[l] = val;
        }
        if (tid < TR / 2) {
            const int r = row0 + 2 * tid;
            double2_t vv = {0.0, 0.0};
            if (r < m) vv.x = v[r];
            if (r + 1 < m) vv.y = v[r + 1];
            vreg = vv;
        }
    };

    double acc = 0.0; // Iterator::sum identity
    fetch(0);
    for (int t = 0; t < ntiles; ++t) {
#pragma unroll
        for (int l = 0; l < NL; ++l)
            *reinterpret_cast<double2_t *>(&tile[lcol[l]][lrow[l]]) = reg[l];
        if (tid < TR / 2) {
            double2_t nv = {-vreg.x, -vreg.y};
            *reinterpret_cast<double2_t *>(&negv[2 * tid]) = nv;
        }
        __syncthreads();
        if (t + 1 < ntiles) fetch(t + 1); // in flight while the columns are walked
        if (tid < C) {
            const int rows = min(TR, m - t * TR);
            if (rows == TR) {
#pragma unroll 8
                for (int r = 0; r < TR; r += 2) {
                    const double2_t a = *reinterpret_cast<const double2_t *>(&tile[tid][r]);
                    const double2_t nv = *reinterpret_cast<const double2_t *>(&negv[r]);
                    const double p0 = a.x * nv.x;
                    acc = acc + p0;
                    const double p1 = a.y * nv.y;
                    acc = acc + p1;
                }
            } else {
                for (int r = 0; r < rows; ++r) {
                    const double p = tile[tid][r] * negv[r];
                    acc = acc + p;
                }
            }
        }
        __syncthreads();
    }
    if (tid < C && pos0 + tid < ncols) {
        const int code = s_code[tid];
        if (code < 0) {
            const double p = 1.0 * -v[-1 - code];
            acc = 0.0 + p;
        }
        dz[pos0 + tid] = acc;
    }
}

// ---------------------------------------------------------------------------------
// k_price_wave: one wave per column, 4 x 16-B loads in flight per lane.
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_price_wave(const DzgCtl *ctl, const double *__restrict__ A,
                                                    long long lda, int m, int ncols,
                                                    const int *__restrict__ nonbasis,
                                                    const int *__restrict__ var_col,
                                                    const double *__restrict__ v,
                                                    double *__restrict__ dz)
{
    if (ctl && ctl->status != DZG_RUNNING) return;
    const int lane = threadIdx.x & 63;
    const int wave_global = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int nwaves = (gridDim.x * blockDim.x) >> 6;
    const int m2 = m & ~1;
    for (int pos = wave_global; pos < ncols; pos += nwaves) {
        const int code = price_code(nonbasis, var_col, pos);
        if (code < 0) {
            if (lane == 0) dz[pos] = 0.0 + 1.0 * -v[-1 - code];
            continue;
        }
        const double *col = A + (long long)code * lda;
        double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
        int r = 2 * lane;
        for (; r + 384 < m2; r += 512) {
            const double2_t c0 = *reinterpret_cast<const double2_t *>(col + r);
            const double2_t c1 = *reinterpret_cast<const double2_t *>(col + r + 128);
            const double2_t c2 = *reinterpret_cast<const double2_t *>(col + r + 256);
            const double2_t c3 = *reinterpret_cast<const double2_t *>(col + r + 384);
            const double2_t v0 = *reinterpret_cast<const double2_t *>(v + r);
            const double2_t v1 = *reinterpret_cast<const double2_t *>(v + r + 128);
            const double2_t v2 = *reinterpret_cast<const double2_t *>(v + r + 256);
            const double2_t v3 = *reinterpret_cast<const double2_t *>(v + r + 384);
            a0 = fma(c0.x, v0.x, a0);
            a1 = fma(c1.x, v1.x, a1);
            a2 = fma(c2.x, v2.x, a2);
            a3 = fma(c3.x, v3.x, a3);
            a0 = fma(c0.y, v0.y, a0);
            a1 = fma(c1.y, v1.y, a1);
            a2 = fma(c2.y, v2.y, a2);
            a3 = fma(c3.y, v3.y, a3);
        }
        for (; r < m2; r += 128) {
            const double2_t c0 = *reinterpret_cast<const double2_t *>(col + r);
            const double2_t v0 = *reinterpret_cast<const double2_t *>(v + r);
            a0 = fma(c0.x, v0.x, a0);
            a0 = fma(c0.y, v0.y, a0);
        }
        if (lane == 0 && m2 < m) a1 = fma(col[m2], v[m2], a1);
        double acc = (a0 + a1) + (a2 + a3);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, DZG_WAVE);
        if (lane == 0) dz[pos] = -acc;
    }
}


// ---------------------------------------------------------------------------------
// k_price_wave2<U, VLDS>: one wave per column, U x 16-B loads in flight per lane;
// VLDS stages v in LDS once per workgroup (m*8 bytes of dynamic LDS) so that the L2
// only serves the matrix stream.  Columns are dealt to waves round-robin over the grid.
// ---------------------------------------------------------------------------------
template <int U, bool VLDS>
__global__ __launch_bounds__(1024) void k_price_wave2(const DzgCtl *ctl, const double *__restrict__ A,
                                                      long long lda, int m, int ncols,
                                                      const int *__restrict__ nonbasis,
                                                      const int *__restrict__ var_col,
                                                      const double *__restrict__ v,
                                                      double *__restrict__ dz)
{
    extern __shared__ __attribute__((aligned(16))) double s_v[];
    if (ctl && ctl->status != DZG_RUNNING) return;
    const int lane = threadIdx.x & 63;
    const int wpb = blockDim.x >> 6;
    const int wave_global = blockIdx.x * wpb + (threadIdx.x >> 6);
    const int nwaves = gridDim.x * wpb;
    const int mpad = (m + 1) & ~1;
    if (VLDS) {
        for (int i = threadIdx.x; i < mpad; i += blockDim.x) s_v[i] = (i < m) ? v[i] : 0.0;
        __syncthreads();
    }
    const double *vv = VLDS ? s_v : v; // v has >= 2 zero pad entries past m in global memory
    for (int pos = wave_global; pos < ncols; pos += nwaves) {
        const int code = price_code(nonbasis, var_col, pos);
        if (code < 0) {
            if (lane == 0) dz[pos] = 0.0 + 1.0 * -v[-1 - code];
            continue;
        }
        const double *col = A + (long long)code * lda; // rows m..lda-1 are zero
        double acc[U];
#pragma unroll
        for (int u = 0; u < U; ++u) acc[u] = 0.0;
        int r = 2 * lane;
        for (; r + 128 * (U - 1) < mpad; r += 128 * U) {
            double2_t c[U], w[U];
#pragma unroll
            for (int u = 0; u < U; ++u)
                c[u] = __builtin_nontemporal_load(reinterpret_cast<const double2_t *>(col + r + 128 * u));
#pragma unroll
            for (int u = 0; u < U; ++u) w[u] = *reinterpret_cast<const double2_t *>(vv + r + 128 * u);
#pragma unroll
            for (int u = 0; u < U; ++u) {
                acc[u] = fma(c[u].x, w[u].x, acc[u]);
                acc[u] = fma(c[u].y, w[u].y, acc[u]);
            }
        }
        for (; r < mpad; r += 128) {
            const double2_t c0 = *reinterpret_cast<const double2_t *>(col + r);
            const double2_t w0 = *reinterpret_cast<const double2_t *>(vv + r);
            acc[0] = fma(c0.x, w0.x, acc[0]);
            acc[0] = fma(c0.y, w0.y, acc[0]);
        }
        double s = 0.0;
#pragma unroll
        for (int u = 0; u < U; ++u) s += acc[u];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, DZG_WAVE);
        if (lane == 0) dz[pos] = -s;
    }
}
