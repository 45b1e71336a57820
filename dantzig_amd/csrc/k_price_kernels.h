// k_price_kernels.h -- the pricing pass  dz = -N^T v  (src/linalg.rs:199-207 neg_t_dot over
// collect_columns(n), called at src/simplex.rs:235), fused with the dual-step ratio test
// (find_second_pivot over z, src/simplex.rs:324,439-461).  This is the HBM-roofline kernel:
// it streams every nonbasic structural column of A once per iteration (8*m bytes each).
//
//  k_price_seq2  SEQUENTIAL-ORDER kernel (default).  Each WAVE owns up to 16 columns.  For a
//                tile of 128 rows it fetches 16 x 1 KiB column segments with fully coalesced
//                16-B/lane nontemporal loads (two tiles are kept in flight in registers),
//                parks them in a wave-private, bank-conflict-free LDS tile next to -v for the
//                same rows, and lane c then walks column c top to bottom:
//                    acc = acc + a * (-v)     product and sum rounded separately, rows ascending
//                which is exactly the reference's summation order, so dz is BIT-IDENTICAL to
//                neg_t_dot.  No workgroup barrier anywhere: waves drift freely, the serial
//                chain (~2-3k cycles per tile) hides under the ~6.5k cycles the same tile
//                needs from HBM.
//  k_price_wave2 TREE-ORDER kernel.  One wave per column, lane-strided partial sums with
//                explicit fma, xor-shuffle tree.  A plain streaming reduction, a few percent
//                faster, not bit-identical to the reference's sums.
//
// Work distribution: the engine keeps the list of nonbasic positions that hold a STRUCTURAL
// variable (plist, maintained by the pivot kernel) and the kernels split exactly those
// columns evenly over the waves of a fixed grid, so basic columns and slack positions cost no
// matrix bytes and every CU gets the same share whatever the basis looks like.  Slack
// positions are unit columns: dz = 0.0 + 1.0 * -v[row].
//
// Column codes: >= 0 structural column index, < 0 unit (slack) column of row -1-code.
#pragma once
#include "common.h"

typedef double double2_t __attribute__((ext_vector_type(2)));

__device__ __forceinline__ int price_code(const int *__restrict__ nonbasis,
                                          const int *__restrict__ var_col, int pos)
{
    const int v = nonbasis[pos];
    return var_col ? var_col[v] : v;
}

// ratio-test candidate of one position (src/simplex.rs:449-455): dz / (z + mu*zbar) if > 0
// (FAST numerics only: the candidate carries its competition for the near-tie arbitration, and
// a denominator that is zero up to rounding marks the whole test untrustworthy -- the reference
// may have +inf, which wins, NaN or a negative ratio there)
__device__ __forceinline__ void price_candidate(DzgCand2 &best, double dzk, int pos, double mu,
                                                double tau, const double *__restrict__ z,
                                                const double *__restrict__ zbar)
{
    const double zk = z[pos], scaled = mu * zbar[pos];
    const double den = zk + scaled;
    DzgCand2 c;
    c.r = dzg_div(dzk, den);
    c.k = pos;
    c.h = -__builtin_inf();
    if (c.r > 0.0) best = dzg_better2(best, c);
    if (dzg_noise_zero(den, zk, scaled, tau)) best.h = __builtin_inf();
}

// the same with z[pos], zbar[pos] already in registers
__device__ __forceinline__ void price_candidate_v(DzgCand2 &best, double dzk, int pos, double mu,
                                                  double tau, double zk, double zbk)
{
    const double scaled = mu * zbk;
    const double den = zk + scaled;
    DzgCand2 c;
    c.r = dzg_div(dzk, den);
    c.k = pos;
    c.h = -__builtin_inf();
    if (c.r > 0.0) best = dzg_better2(best, c);
    if (dzg_noise_zero(den, zk, scaled, tau)) best.h = __builtin_inf();
}

// unit columns: every thread of the grid takes positions pos = tid, tid + nthreads, ...
__device__ __forceinline__ void price_slack_positions(DzgCand2 &best, int q,
                                                      const int *__restrict__ nonbasis,
                                                      const int *__restrict__ var_col,
                                                      const double *__restrict__ v,
                                                      double *__restrict__ dz, double mu,
                                                      double tau, const double *__restrict__ z,
                                                      const double *__restrict__ zbar)
{
    const int nthreads = gridDim.x * blockDim.x;
    for (int pos = blockIdx.x * blockDim.x + threadIdx.x; pos < q; pos += nthreads) {
        const int code = price_code(nonbasis, var_col, pos);
        if (code < 0) {
            const double p = 1.0 * -v[-1 - code];
            const double d = 0.0 + p; // Iterator::sum identity + the single stored entry
            dz[pos] = d;
            if (z) price_candidate(best, d, pos, mu, tau, z, zbar);
        }
    }
}

__device__ __forceinline__ void price_publish(DzgCand2 best, double *__restrict__ rz_r,
                                              int *__restrict__ rz_k, double *__restrict__ rz_h)
{
    best = dzg_block_best2(best);
    if (rz_r && threadIdx.x == 0) {
        rz_r[blockIdx.x] = best.r;
        rz_k[blockIdx.x] = best.k;
        rz_h[blockIdx.x] = best.h;
    }
}

// ---------------------------------------------------------------------------------
// k_price_seq2<CW>: 4 waves per workgroup, CW columns per wave and pass, tiles of 128 rows.
// Engine mode: plist != nullptr, count = ctl->nb_struct.  Raw mode (parity tests):
// plist == nullptr, every position 0..q-1 is a column, codes may be negative.
// ---------------------------------------------------------------------------------
template <int CW, int DEPTH = 2, int DBG = 0, bool NT = true>
__global__ __launch_bounds__(256) void k_price_seq2(
    const DzgCtl *ctl, const double *__restrict__ A, long long lda, int m, int q,
    const int *__restrict__ plist, const int *__restrict__ nonbasis,
    const int *__restrict__ var_col, const double *__restrict__ v, double *__restrict__ dz,
    const double *__restrict__ z, const double *__restrict__ zbar, double *__restrict__ rz_r,
    int *__restrict__ rz_k, double *__restrict__ rz_h, int col0 = 0)
{
    constexpr int TR = 128, PAD = 2; // column stride 130 doubles: lane c starts 4c banks on
    __shared__ __attribute__((aligned(16))) double tile[4][CW][TR + PAD];
    if (ctl && ctl->status != DZG_RUNNING) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nw = gridDim.x * 4;
    const int wg = blockIdx.x * 4 + wave;
    const double mu = ctl ? ctl->mu : 0.0, tau = ctl ? ctl->tau : 0.0;
    DzgCand2 best = dzg_cand2_none();
    price_slack_positions(best, q, nonbasis, var_col, v, dz, mu, tau, z, zbar);

    const int count = plist ? (int)ctl->nb_struct : q;
    const int base = count / nw, rem = count % nw;
    const int start = wg * base + (wg < rem ? wg : rem);
    const int cnt = base + (wg < rem ? 1 : 0);
    const int ntiles = (m + TR - 1) / TR;
    double(*mytile)[TR + PAD] = tile[wave];

    for (int c0 = 0; c0 < cnt; c0 += CW) {
        const int nc = (cnt - c0) < CW ? (cnt - c0) : CW;
        int mypos = -1, mycode = -1;
        if (lane < nc) {
            const int idx = start + c0 + lane;
            mypos = plist ? plist[idx] : idx;
            mycode = price_code(nonbasis, var_col, mypos);
        }
        // wave-uniform column offsets (scalar registers).  Every load below is issued
        // unconditionally so that the compiler can count outstanding loads exactly
        // (s_waitcnt vmcnt(N) with N > 0 keeps the second tile in flight); columns past nc and
        // unit columns re-read the wave's last valid column (served by L1/L2) and are ignored.
        int lastcode = col0;
#pragma unroll
        for (int l = 0; l < CW; ++l) {
            const int code_l = __builtin_amdgcn_readlane(mycode, l);
            if (code_l >= 0) lastcode = code_l;
        }
        long long off[CW];
#pragma unroll
        for (int l = 0; l < CW; ++l) {
            const int code_l = __builtin_amdgcn_readlane(mycode, l);
            off[l] = (long long)((code_l >= 0 ? code_l : lastcode) - col0) * lda;
        }
        double2_t rg[DEPTH][CW], vg[DEPTH]; // DEPTH tiles in flight, statically indexed
        double dbg_sink = 0.0;
        long long cyc_park = 0, cyc_fetch = 0, cyc_walk = 0; // DBG == 4 only
        // lda is a multiple of 16 and rows m..lda-1 are zero, v carries 2 zero pads: a 16-B
        // load that starts below lda (resp. m) stays inside its column (resp. v).  Rows past
        // the end are clamped to the last pair and zeroed after the load (no branches).
        const int lastpair = (int)lda - 2;
        const int lastv = ((m + 1) & ~1) - 2 >= 0 ? ((m + 1) & ~1) - 2 : 0;

        // fetch only issues loads (clamped addresses); the masks are applied in park(), at the
        // first use, so nothing here waits on a load
        auto fetch = [&](int t, double2_t(&reg)[CW], double2_t &vreg) {
            const int row = t * TR + 2 * lane;
            const int rowc = row < lda ? row : lastpair;
#pragma unroll
            for (int l = 0; l < CW; ++l) {
                const double2_t *gp = reinterpret_cast<const double2_t *>(A + off[l] + rowc);
                reg[l] = NT ? __builtin_nontemporal_load(gp) : *gp;
            }
            vreg = *reinterpret_cast<const double2_t *>(v + (row < m ? row : lastv));
        };
        // park: lane holds rows (2*lane, 2*lane+1) of every column of the tile, and the same two
        // rows of v.  It forms the PRODUCTS a * (-v) here, with all 64 lanes busy (one rounding,
        // exactly the reference's `val * -v[i]`), and parks the products in LDS; the serial
        // walk then only reads and adds.
        auto park = [&](int t, const double2_t(&reg)[CW], const double2_t &vreg) {
            if (DBG == 2) { // diagnostic: consume the registers, skip LDS
#pragma unroll
                for (int l = 0; l < CW; ++l) dbg_sink += reg[l].x + reg[l].y;
                return;
            }
            const int row = t * TR + 2 * lane;
            const bool inside = row < lda && row < m + 1; // rows >= m hold zeros or padding
            const double nv0 = -vreg.x, nv1 = -vreg.y;    // v carries zero pads past m
            const double2_t zero = {0.0, 0.0};
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int l = 0; l < CW; ++l) {
                // The reference's CSC drops exact zeros (src/linalg.rs:254-270), so neg_t_dot never
                // forms 0 * -v[i]; the dense layout keeps them, and 0 * (-/+inf) or 0 * NaN would
                // put a NaN where the reference skips the entry.  A skipped entry is the
                // addition of +0.0 here: the running sum starts at +0.0 and can never be -0.0
                // (x + -x = +0.0), so adding +0.0 leaves every bit of it.
                double2_t pr;
                pr.x = reg[l].x != 0.0 ? reg[l].x * nv0 : 0.0;
                pr.y = reg[l].y != 0.0 ? reg[l].y * nv1 : 0.0;
                *reinterpret_cast<double2_t *>(&mytile[l][2 * lane]) = inside ? pr : zero;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        };
        double acc = 0.0; // Iterator::sum identity
        // walk: lane c adds the products of column c top to bottom -- the only serial part.
        // LDS reads run one 16-row chunk ahead of the additions.
        const double *colp = mytile[lane & (CW - 1)];
        auto walk = [&](int t) {
            if (DBG != 0 && DBG != 4) return;
            const int row0 = t * TR;
            const int rows = (m - row0) < TR ? (m - row0) : TR;
            if (lane < CW) {
                if (rows == TR) {
                    double2_t buf[2][8];
#pragma unroll
                    for (int j = 0; j < 8; ++j)
                        buf[0][j] = *reinterpret_cast<const double2_t *>(colp + 2 * j);
#pragma unroll
                    for (int c = 0; c < TR / 16; ++c) {
                        if (c + 1 < TR / 16) {
#pragma unroll
                            for (int j = 0; j < 8; ++j)
                                buf[(c + 1) & 1][j] = *reinterpret_cast<const double2_t *>(
                                    colp + 16 * (c + 1) + 2 * j);
                        }
#pragma unroll
                        for (int j = 0; j < 8; ++j) {
                            acc = acc + buf[c & 1][j].x;
                            acc = acc + buf[c & 1][j].y;
                        }
                    }
                } else {
                    for (int r = 0; r < rows; ++r) acc = acc + colp[r];
                }
            }
        };

        // DEPTH tiles in flight; the steady-state loop has no conditional loads
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) fetch(d < ntiles ? d : ntiles - 1, rg[d], vg[d]);
        int t = 0;
        for (; t + 2 * DEPTH - 1 < ntiles; t += DEPTH) {
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) {
                if (DBG == 4) {
                    const long long s0 = __builtin_amdgcn_s_memtime();
                    park(t + d, rg[d], vg[d]);
                    __builtin_amdgcn_s_waitcnt(0xC07F);
                    const long long s1 = __builtin_amdgcn_s_memtime();
                    fetch(t + DEPTH + d, rg[d], vg[d]);
                    const long long s2 = __builtin_amdgcn_s_memtime();
                    walk(t + d);
                    asm volatile("" ::"v"(acc));
                    const long long s3 = __builtin_amdgcn_s_memtime();
                    cyc_park += s1 - s0;
                    cyc_fetch += s2 - s1;
                    cyc_walk += s3 - s2;
                } else {
                    park(t + d, rg[d], vg[d]);
                    fetch(t + DEPTH + d, rg[d], vg[d]);
                    walk(t + d);
                }
            }
        }
        // drain: fewer than 2*DEPTH tiles left
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            if (t + d < ntiles) {
                park(t + d, rg[d], vg[d]);
                if (t + DEPTH + d < ntiles) fetch(t + DEPTH + d, rg[d], vg[d]);
                walk(t + d);
            }
        }
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            if (t + DEPTH + d < ntiles) {
                park(t + DEPTH + d, rg[d], vg[d]);
                walk(t + DEPTH + d);
            }
        }
        if (DBG != 0 && DBG != 4) acc = dbg_sink + mytile[lane & (CW - 1)][lane];
        if (DBG == 4 && lane == 0) { // diagnostic build only: cycles per segment, per wave
            dz[q + 3 * wg + 0] = (double)cyc_park;
            dz[q + 3 * wg + 1] = (double)cyc_fetch;
            dz[q + 3 * wg + 2] = (double)cyc_walk;
        }
        if (lane < nc && mycode >= 0) {
            dz[mypos] = acc;
            if (z) price_candidate(best, acc, mypos, mu, tau, z, zbar);
        }
    }
    price_publish(best, rz_r, rz_k, rz_h);
}

// ---------------------------------------------------------------------------------
// k_price_wave2<U>: one wave per column, U x 16-B nontemporal loads in flight per lane.
// ---------------------------------------------------------------------------------
template <int U>
__global__ __launch_bounds__(256) void k_price_wave2(
    const DzgCtl *ctl, const double *__restrict__ A, long long lda, int m, int q,
    const int *__restrict__ plist, const int *__restrict__ nonbasis,
    const int *__restrict__ var_col, const double *__restrict__ v, double *__restrict__ dz,
    const double *__restrict__ z, const double *__restrict__ zbar, double *__restrict__ rz_r,
    int *__restrict__ rz_k, double *__restrict__ rz_h, int col0 = 0)
{
    if (ctl && ctl->status != DZG_RUNNING) return;
    const int lane = threadIdx.x & 63;
    const int wg = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int nw = (gridDim.x * blockDim.x) >> 6;
    const double mu = ctl ? ctl->mu : 0.0, tau = ctl ? ctl->tau : 0.0;
    DzgCand2 best = dzg_cand2_none();
    price_slack_positions(best, q, nonbasis, var_col, v, dz, mu, tau, z, zbar);
    const int count = plist ? (int)ctl->nb_struct : q;
    const int mpad = (m + 1) & ~1;
    for (int idx = wg; idx < count; idx += nw) {
        const int pos = plist ? plist[idx] : idx;
        const int code = price_code(nonbasis, var_col, pos);
        if (code < 0) continue;
        const double *col = A + (long long)(code - col0) * lda; // rows m..lda-1 are zero
        double acc[U];
#pragma unroll
        for (int u = 0; u < U; ++u) acc[u] = 0.0;
        int r = 2 * lane;
        for (; r + 128 * (U - 1) < mpad; r += 128 * U) {
            double2_t c[U], w[U];
#pragma unroll
            for (int u = 0; u < U; ++u)
                c[u] = __builtin_nontemporal_load(
                    reinterpret_cast<const double2_t *>(col + r + 128 * u));
#pragma unroll
            for (int u = 0; u < U; ++u) w[u] = *reinterpret_cast<const double2_t *>(v + r + 128 * u);
#pragma unroll
            for (int u = 0; u < U; ++u) {
                acc[u] = fma(c[u].x, w[u].x, acc[u]);
                acc[u] = fma(c[u].y, w[u].y, acc[u]);
            }
        }
        for (; r < mpad; r += 128) {
            const double2_t c0 = *reinterpret_cast<const double2_t *>(col + r);
            const double2_t w0 = *reinterpret_cast<const double2_t *>(v + r);
            acc[0] = fma(c0.x, w0.x, acc[0]);
            acc[0] = fma(c0.y, w0.y, acc[0]);
        }
        double s = 0.0;
#pragma unroll
        for (int u = 0; u < U; ++u) s += acc[u];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, DZG_WAVE);
        if (lane == 0) {
            dz[pos] = -s;
            if (z) price_candidate(best, -s, pos, mu, tau, z, zbar);
        }
    }
    price_publish(best, rz_r, rz_k, rz_h);
}

// ---------------------------------------------------------------------------------
// k_price_tree<CW, DEPTH>: the FAST-numerics pricing kernel.  Same streaming skeleton as
// k_price_seq2 -- a wave owns CW columns at a time, 1-KiB fully coalesced nontemporal loads per
// column and 128-row tile, DEPTH tiles in flight in registers, exact s_waitcnt counts -- but the
// sums stay in registers: lane l keeps ONE accumulator per column and adds its two rows of every
// tile with fma, tiles ascending; 64 partial sums per column are folded by an xor-shuffle tree at
// the end.  No LDS, no serial chain, so nothing but HBM bounds it, whatever the column count.
//
// The summation order of a column depends only on m -- not on CW, DEPTH, the grid or which wave
// gets the column -- so dz is bit-identical between a single GPU and any column sharding: the
// sharded solve takes the same pivots as the unsharded one.  (FAST's v comes from the explicit
// inverse and is not bit-identical to the reference's anyway; the reference-order sums of
// k_price_seq2 matter for STRICT numerics only.)
// ---------------------------------------------------------------------------------
template <int CW, int DEPTH, int TP = 1>
__global__ __launch_bounds__(256) void k_price_tree(
    const DzgCtl *ctl, const double *__restrict__ A, long long lda, int m, int q,
    const int *__restrict__ plist, const int *__restrict__ nonbasis,
    const int *__restrict__ var_col, const double *__restrict__ v, double *__restrict__ dz,
    const double *__restrict__ z, const double *__restrict__ zbar, double *__restrict__ rz_r,
    int *__restrict__ rz_k, double *__restrict__ rz_h, int col0 = 0,
    const int *__restrict__ pcode = nullptr, int rows_T = 0, int need_kind = -1)
{
    constexpr int TR = 128;
    if (ctl && ctl->status != DZG_RUNNING) return;
    if (rows_T > 0 && ctl->ncompact < rows_T) return; // the row-wise pass prices this iteration
    // (row-sharded ranks price a dual step before their second exchange and a primal step after it:
    // the host enqueues the pass in both places, the one whose step kind it is not returns)
    if (need_kind >= 0 && ctl->kind != need_kind) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nw = gridDim.x * 4;
    const int wg = blockIdx.x * 4 + wave;
    const double mu = ctl ? ctl->mu : 0.0, tau = ctl ? ctl->tau : 0.0;
    DzgCand2 best = dzg_cand2_none();
    // The unit column of this thread's first position (pos = thread index): its code is fetched
    // here, beside the wave's column list, its v / z / zbar entries beside the first matrix tiles,
    // and the candidate is formed at the very end -- after a kernel boundary each of these is a
    // trip to memory, and taken one behind the other (as price_slack_positions would) they would
    // delay the first byte of the stream by several microseconds.
    const int nthreads = gridDim.x * blockDim.x;
    const int spos = blockIdx.x * blockDim.x + threadIdx.x;
    int scode = 0; // >= 0: not a unit column
    if (spos < q) scode = price_code(nonbasis, var_col, spos);
    double sv = 0.0, sz = 0.0, szb = 0.0;
    bool slack_loaded = false;

    const int count = plist ? (int)ctl->nb_struct : q;
    const int base = count / nw, rem = count % nw;
    const int start = wg * base + (wg < rem ? wg : rem);
    const int cnt = base + (wg < rem ? 1 : 0);
    const int ntiles = (m + TR - 1) / TR;
    const int ngroups = (ntiles + TP - 1) / TP; // TP adjacent tiles of a column are fetched back to back
    const int lastpair = (int)lda - 2;
    const int lastv = ((m + 1) & ~1) - 2 >= 0 ? ((m + 1) & ~1) - 2 : 0;

    for (int c0 = 0; c0 < cnt; c0 += CW) {
        const int nc = (cnt - c0) < CW ? (cnt - c0) : CW;
        int mypos = -1, mycode = -1;
        if (lane < nc) {
            const int idx = start + c0 + lane;
            mypos = plist ? plist[idx] : idx;
            mycode = pcode ? pcode[idx] : price_code(nonbasis, var_col, mypos);
        }
        // wave-uniform column offsets; every load is unconditional (columns past nc and unit
        // columns re-read the wave's last valid column and are ignored), see k_price_seq2
        int lastcode = col0;
#pragma unroll
        for (int l = 0; l < CW; ++l) {
            const int code_l = __builtin_amdgcn_readlane(mycode, l);
            if (code_l >= 0) lastcode = code_l;
        }
        long long off[CW];
#pragma unroll
        for (int l = 0; l < CW; ++l) {
            const int code_l = __builtin_amdgcn_readlane(mycode, l);
            off[l] = (long long)((code_l >= 0 ? code_l : lastcode) - col0) * lda;
        }
        double2_t rg[DEPTH][TP][CW], vg[DEPTH][TP];
        double acc[CW];
#pragma unroll
        for (int l = 0; l < CW; ++l) acc[l] = 0.0;
        // second trip, beside the first tiles: z, zbar of this lane's column; the unit column's data
        double zc = 0.0, zbc = 0.0;
        if (z && lane < nc && mycode >= 0) {
            zc = z[mypos];
            zbc = zbar[mypos];
        }
        if (!slack_loaded) {
            slack_loaded = true;
            if (scode < 0) {
                sv = v[-1 - scode];
                if (z) {
                    sz = z[spos];
                    szb = zbar[spos];
                }
            }
        }

        // column-major: the TP tiles of a column leave back to back (TP KB contiguous per visit of
        // a column: fewer DRAM row activations than 1 KB visits; 1-1.5 % at 8192 rows,
        // profiles/r02_price_microbench_adjacent_tiles.txt); the sums keep their order (tiles
        // ascending per column)
        auto fetch = [&](int g, double2_t(&reg)[TP][CW], double2_t(&vreg)[TP]) {
#pragma unroll
            for (int l = 0; l < CW; ++l)
#pragma unroll
                for (int u = 0; u < TP; ++u) {
                    const int row = (g * TP + u) * TR + 2 * lane;
                    const int rowc = row < lda ? row : lastpair;
                    reg[u][l] = __builtin_nontemporal_load(
                        reinterpret_cast<const double2_t *>(A + off[l] + rowc));
                }
#pragma unroll
            for (int u = 0; u < TP; ++u) {
                const int row = (g * TP + u) * TR + 2 * lane;
                vreg[u] = *reinterpret_cast<const double2_t *>(v + (row < m ? row : lastv));
            }
        };
        // rows >= m: the matrix holds zeros (padding) or, clamped, a repeated pair -- masked
        auto consume = [&](int g, const double2_t(&reg)[TP][CW], const double2_t(&vreg)[TP]) {
#pragma unroll
            for (int u = 0; u < TP; ++u) {
                const bool inside = (g * TP + u) * TR + 2 * lane < m; // v carries zero pads past m
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
        // fold the 64 partial sums of every column; lane l keeps column l's total
        double mine = 0.0;
#pragma unroll
        for (int l = 0; l < CW; ++l) {
            double s = acc[l];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, DZG_WAVE);
            if (lane == l) mine = s;
        }
        if (lane < nc && mycode >= 0) {
            dz[mypos] = -mine;
            if (z) price_candidate_v(best, -mine, mypos, mu, tau, zc, zbc);
        }
    }
    // ---- unit columns (the arithmetic of price_slack_positions)
    if (scode < 0) {
        if (!slack_loaded) { // (a wave without columns)
            sv = v[-1 - scode];
            if (z) {
                sz = z[spos];
                szb = zbar[spos];
            }
        }
        const double p = 1.0 * -sv;
        const double d = 0.0 + p; // Iterator::sum identity + the single stored entry
        dz[spos] = d;
        if (z) price_candidate_v(best, d, spos, mu, tau, sz, szb);
    }
    for (int pos = spos + nthreads; pos < q; pos += nthreads) { // more positions than threads
        const int code = price_code(nonbasis, var_col, pos);
        if (code < 0) {
            const double p = 1.0 * -v[-1 - code];
            const double d = 0.0 + p;
            dz[pos] = d;
            if (z) price_candidate(best, d, pos, mu, tau, z, zbar);
        }
    }
    price_publish(best, rz_r, rz_k, rz_h);
}

// ---------------------------------------------------------------------------------
// k_price_csc: sparse columns (CSC, rows ascending).  One thread per column walks its stored
// entries in order: acc = acc + val * (-v[row]) -- literally the reference's neg_t_dot
// (src/linalg.rs:199-207), so dz is bit-identical.  ~50 entries per column at config 4: the
// whole pass moves 12 bytes per nonzero and is far from any roofline; the gathers of v hit L2.
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_price_csc(
    const DzgCtl *ctl, const long long *__restrict__ cptr, const int *__restrict__ ridx,
    const double *__restrict__ cval, int q, const int *__restrict__ plist,
    const int *__restrict__ nonbasis, const int *__restrict__ var_col,
    const double *__restrict__ v, double *__restrict__ dz, const double *__restrict__ z,
    const double *__restrict__ zbar, double *__restrict__ rz_r, int *__restrict__ rz_k,
    double *__restrict__ rz_h, int col0)
{
    if (ctl && ctl->status != DZG_RUNNING) return;
    const double mu = ctl ? ctl->mu : 0.0, tau = ctl ? ctl->tau : 0.0;
    DzgCand2 best = dzg_cand2_none();
    price_slack_positions(best, q, nonbasis, var_col, v, dz, mu, tau, z, zbar);
    const int count = plist ? (int)ctl->nb_struct : q;
    // 8 lanes share one column: they fetch 8 consecutive stored entries at a time (coalesced
    // 64-B / 32-B segments) and form the 8 products in parallel; the running sum then takes the
    // products one by one in stored (ascending-row) order through shuffles, so the order of
    // additions -- hence every bit of dz -- is the reference's.
    const int lane = threadIdx.x & 63, sub = lane & 7, gbase = lane & ~7;
    const int group = (blockIdx.x * blockDim.x + threadIdx.x) >> 3;
    const int ngroups = (gridDim.x * blockDim.x) >> 3;
    for (int idx0 = 0; idx0 < count; idx0 += ngroups) { // wave-uniform trip count
        const int idx = idx0 + group;
        int pos = -1, code = -1;
        long long e0 = 0, e1 = 0;
        if (idx < count) {
            pos = plist ? plist[idx] : idx;
            code = price_code(nonbasis, var_col, pos);
            if (code >= 0) {
                e0 = cptr[code - col0];
                e1 = cptr[code - col0 + 1];
            }
        }
        long long span = e1 - e0;
#pragma unroll
        for (int off = 32; off >= 8; off >>= 1) { // longest column among the wave's 8 groups
            const long long o = __shfl_xor(span, off, DZG_WAVE);
            span = o > span ? o : span;
        }
        double acc = 0.0; // Iterator::sum identity
        for (long long base = 0; base < span; base += 8) {
            const long long e = e0 + base + sub;
            double p = 0.0;
            if (e < e1) p = cval[e] * -v[ridx[e]];
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const double pt = __shfl(p, gbase + t, DZG_WAVE);
                if (e0 + base + t < e1) acc = acc + pt;
            }
        }
        if (code >= 0 && sub == 0) {
            dz[pos] = acc;
            if (z) price_candidate(best, acc, pos, mu, tau, z, zbar);
        }
    }
    price_publish(best, rz_r, rz_k, rz_h);
}

// ---------------------------------------------------------------------------------
// k_price_csc_tree: the FAST-numerics twin of k_price_csc.  Same 8 lanes per column and the same
// coalesced 8-entry fetches, but every lane keeps its own partial sum (entries sub, sub + 8, ...)
// and the eight partials fold in an xor tree: no 50-long chain of dependent shuffles per column,
// the pass is bound by the gathers of v (L2-resident: 8 bytes used of every 128-byte line that
// travels to a CU -- 718 MB of L2 sectors per launch for 63 MB of matrix, PMC counters in
// profiles/r02_pmc_price_csc_summary.txt) and the 12 bytes per stored entry.  (A variant that staged v through LDS in 16 384-row
// blocks, one 1024-thread workgroup per CU, measured SLOWER -- 42.3 us against 34.1 us per launch
// at config 4, profiles/r02_config4_kernel_stats_lds_pricing_variant.csv: four barrier-separated
// phases with 16 waves per CU hide the latency of the entry stream worse than 32 free waves.)  The
// order of a column's sum depends on nothing but the column (deterministic, shard-independent);
// it is not the reference's order, which only STRICT numerics needs (its v is the reference's v).
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_price_csc_tree(
    const DzgCtl *ctl, const long long *__restrict__ cptr, const int *__restrict__ ridx,
    const double *__restrict__ cval, int q, const int *__restrict__ plist,
    const int *__restrict__ nonbasis, const int *__restrict__ var_col,
    const double *__restrict__ v, double *__restrict__ dz, const double *__restrict__ z,
    const double *__restrict__ zbar, double *__restrict__ rz_r, int *__restrict__ rz_k,
    double *__restrict__ rz_h, int col0, const int *__restrict__ pcode = nullptr)
{
    if (ctl && ctl->status != DZG_RUNNING) return;
    const double mu = ctl ? ctl->mu : 0.0, tau = ctl ? ctl->tau : 0.0;
    DzgCand2 best = dzg_cand2_none();
    // the unit column of this thread's first position: fetched beside the column work, used at the
    // end (see k_price_tree)
    const int nthreads = gridDim.x * blockDim.x;
    const int spos = blockIdx.x * blockDim.x + threadIdx.x;
    int scode = 0;
    if (spos < q) scode = price_code(nonbasis, var_col, spos);
    double sv = 0.0, sz = 0.0, szb = 0.0;
    bool slack_loaded = false;
    const int count = plist ? (int)ctl->nb_struct : q;
    const int sub = threadIdx.x & 7;
    const int group = (blockIdx.x * blockDim.x + threadIdx.x) >> 3;
    const int ngroups = (gridDim.x * blockDim.x) >> 3;
    for (int idx0 = 0; idx0 < count; idx0 += ngroups) { // wave-uniform trip count
        const int idx = idx0 + group;
        int pos = -1, code = -1;
        long long e0 = 0, e1 = 0;
        if (idx < count) {
            pos = plist ? plist[idx] : idx;
            code = pcode ? pcode[idx] : price_code(nonbasis, var_col, pos);
            if (code >= 0) {
                e0 = cptr[code - col0];
                e1 = cptr[code - col0 + 1];
            }
        }
        double zc = 0.0, zbc = 0.0;
        if (z && code >= 0 && sub == 0) {
            zc = z[pos];
            zbc = zbar[pos];
        }
        if (!slack_loaded) {
            slack_loaded = true;
            if (scode < 0) {
                sv = v[-1 - scode];
                if (z) {
                    sz = z[spos];
                    szb = zbar[spos];
                }
            }
        }
        double a0 = 0.0, a1 = 0.0; // two independent chains per lane
        long long e = e0 + sub;
        for (; e + 8 < e1; e += 16) {
            const int r0 = ridx[e], r1 = ridx[e + 8];
            const double c0 = cval[e], c1 = cval[e + 8];
            a0 = fma(c0, v[r0], a0);
            a1 = fma(c1, v[r1], a1);
        }
        if (e < e1) a0 = fma(cval[e], v[ridx[e]], a0);
        double acc = a0 + a1;
#pragma unroll
        for (int off = 4; off > 0; off >>= 1) acc += __shfl_xor(acc, off, DZG_WAVE);
        if (code >= 0 && sub == 0) {
            dz[pos] = -acc;
            if (z) price_candidate_v(best, -acc, pos, mu, tau, zc, zbc);
        }
    }
    // ---- unit columns (the arithmetic of price_slack_positions)
    if (scode < 0) {
        if (!slack_loaded) {
            sv = v[-1 - scode];
            if (z) {
                sz = z[spos];
                szb = zbar[spos];
            }
        }
        const double p = 1.0 * -sv;
        const double d = 0.0 + p; // Iterator::sum identity + the single stored entry
        dz[spos] = d;
        if (z) price_candidate_v(best, d, spos, mu, tau, sz, szb);
    }
    for (int pos = spos + nthreads; pos < q; pos += nthreads) { // more positions than threads
        const int code = price_code(nonbasis, var_col, pos);
        if (code < 0) {
            const double p = 1.0 * -v[-1 - code];
            const double d = 0.0 + p;
            dz[pos] = d;
            if (z) price_candidate(best, d, pos, mu, tau, z, zbar);
        }
    }
    price_publish(best, rz_r, rz_k, rz_h);
}

// ---------------------------------------------------------------------------------
// k_price_csc_rl: FAST pricing of the sparse-basis path (k_sparse.hip) over the LIVE entries only.
// v = row p of B^-1 in row coordinates is zero outside R (the k rows whose slack is nonbasic) and
// the leaving slack's own row, so a column's sum only needs its entries in those rows.  Every
// column keeps them as a list in its own CSC slice (lcnt / lent: k_sp_btran appends the row
// that is about to join R before this launch, k_sp_pivot removes the row that left), and the pass
// walks nnz * k / m entries instead of nnz: 1 in 50 at k = 1 000 of config 4, where the full pass
// (k_price_csc_tree) spent 35 of a pivot's 100 us gathering zeros of v.  Same lane layout as the
// tree kernel (8 lanes per column, two chains per lane, xor tree); the order inside a list is the
// order its rows joined R, a function of the pivot sequence alone (deterministic); it is not the
// reference's order (STRICT and price_kernel = SEQ use k_price_csc).
// work[blockIdx.x] accumulates the entries this workgroup walked (roofline accounting, exact).
// ---------------------------------------------------------------------------------
#define RL_LPC 4 // lanes per column: 2048 x 256 / 4 = 131 072 columns in one pass
__global__ __launch_bounds__(256) void k_price_csc_rl(
    const DzgCtl *ctl, const long long *__restrict__ cptr, const int *__restrict__ lcnt,
    const DzgLiveEntry *__restrict__ lent, int q,
    const int *__restrict__ plist, const int *__restrict__ pcode, const int *__restrict__ nbcode,
    const double *__restrict__ v, double *__restrict__ dz, const double *__restrict__ z,
    const double *__restrict__ zbar, double *__restrict__ rz_r, int *__restrict__ rz_k,
    double *__restrict__ rz_h, unsigned long long *__restrict__ work)
{
    __shared__ unsigned int s_work[4];
    // first trip, side by side: the control block, this thread's unit-column position and the
    // first column of its lane group (plist / pcode hold q entries; those beyond nb_struct are
    // stale and only used once the count has arrived)
    const int nthreads = gridDim.x * blockDim.x;
    const int spos = blockIdx.x * blockDim.x + threadIdx.x;
    const int sub = threadIdx.x & (RL_LPC - 1);
    const int group = spos / RL_LPC;
    const int ngroups = nthreads / RL_LPC;
    int scode = 0, pos0 = -1, code0 = -1;
    if (spos < q) scode = nbcode[spos];
    if (group < q) {
        pos0 = plist[group];
        code0 = pcode[group];
    }
    const int status = ctl->status, count = (int)ctl->nb_struct;
    const double mu = ctl->mu, tau = ctl->tau;
    if (status != DZG_RUNNING) return;
    DzgCand2 best = dzg_cand2_none();
    double sv = 0.0, sz = 0.0, szb = 0.0;
    if (scode < 0) { // second trip, beside the column's descriptors
        sv = v[-1 - scode];
        sz = z[spos];
        szb = zbar[spos];
    }
    unsigned int walked = 0;
    for (int idx0 = 0; idx0 < count; idx0 += ngroups) { // wave-uniform trip count
        const int idx = idx0 + group;
        int pos = -1, code = -1;
        long long e0 = 0, e1 = 0;
        if (idx < count) {
            pos = idx0 == 0 ? pos0 : plist[idx];
            code = idx0 == 0 ? code0 : pcode[idx];
            if (code >= 0) {
                e0 = cptr[code];
                e1 = e0 + lcnt[code];
            }
        }
        double zc = 0.0, zbc = 0.0;
        if (code >= 0 && sub == 0) {
            zc = z[pos];
            zbc = zbar[pos];
            walked += (unsigned int)(e1 - e0);
        }
        double a0 = 0.0, a1 = 0.0; // two independent chains per lane
        long long e = e0 + sub;
        for (; e + RL_LPC < e1; e += 2 * RL_LPC) {
            const DzgLiveEntry n0 = lent[e], n1 = lent[e + RL_LPC];
            a0 = fma(n0.val, v[n0.row], a0);
            a1 = fma(n1.val, v[n1.row], a1);
        }
        if (e < e1) {
            const DzgLiveEntry n0 = lent[e];
            a0 = fma(n0.val, v[n0.row], a0);
        }
        double acc = a0 + a1;
#pragma unroll
        for (int off = RL_LPC / 2; off > 0; off >>= 1) acc += __shfl_xor(acc, off, DZG_WAVE);
        if (code >= 0 && sub == 0) {
            dz[pos] = -acc;
            price_candidate_v(best, -acc, pos, mu, tau, zc, zbc);
        }
    }
    // ---- unit columns (the arithmetic of price_slack_positions)
    if (scode < 0) {
        const double p = 1.0 * -sv;
        const double d = 0.0 + p;
        dz[spos] = d;
        price_candidate_v(best, d, spos, mu, tau, sz, szb);
    }
    for (int pos = spos + nthreads; pos < q; pos += nthreads) { // more positions than threads
        const int code = nbcode[pos];
        if (code < 0) {
            const double p = 1.0 * -v[-1 - code];
            const double d = 0.0 + p;
            dz[pos] = d;
            price_candidate(best, d, pos, mu, tau, z, zbar);
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) walked += __shfl_xor(walked, off, DZG_WAVE);
    if ((threadIdx.x & 63) == 0) s_work[threadIdx.x >> 6] = walked;
    price_publish(best, rz_r, rz_k, rz_h); // (its barriers order s_work)
    if (threadIdx.x == 0) // (an atomic without return: no trip to wait for at the end of the launch)
        __hip_atomic_fetch_add(work + blockIdx.x,
                               (unsigned long long)s_work[0] + s_work[1] + s_work[2] + s_work[3],
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---------------------------------------------------------------------------------
// Row-wise pricing of a DENSE matrix (FAST numerics, price_kernel = AUTO).
//
// v = row p of B^-1 is zero outside R -- the k rows whose slack is nonbasic, the rows of the
// compact columns of the inverse (drow[0..k)) -- and the leaving slack's own row (k_fast_btran,
// k_chain_pre: base 0 and eta rows that are zero there).  Early in a solve k is a small fraction
// of m (k = 26..180 of 8192 rows over the benchmark's timed pivots, 4 050 after 150 000), and the
// column-wise pass spends its 8 m bytes per column multiplying zeros.  With a second, ROW-major
// copy of the matrix (At: m rows of ldt, one more GB at 8192 x 16384 of 288) the same numbers are
//
//      dz_j = - sum_{c < k} v[drow[c]] * At[drow[c]][j]  ( - At[rl][j], the leaving slack's row )
//
// i.e. (k + 1) contiguous rows streamed instead of m rows of every nonbasic column: 8 (k+1) n_s
// bytes against 8 m (n_s - k).  Used while k < T = 0.93 m n_s / (n_s + m) (where the two byte
// counts cross, less the second launch and the kernels' rates); beyond that the column-wise kernel takes over.  The
// rule is evaluated ON THE DEVICE from ctl->ncompact by all three kernels alike (rows, finish,
// columns: the ones that do not apply return at once), so the arithmetic a pivot gets depends on
// the state alone; the host only leaves out launches its bounds on k prove idle.
//
// k_price_rows: grid (column tiles of 256 * VEC, PR_GMAX row groups).  G = min(PR_GMAX,
// ceil((k + 1) / 16)) groups are active; group g adds rows c = g, g + G, ... in ascending order
// (16 rows = 16 x VEC x 8 B per lane in flight) and leaves its partial sums in part[g][j].
// k_price_rows_finish: dz = -(part[0] + part[1] + ...) in group order for the nonbasic structural
// positions, unit columns and the fused dual ratio test exactly as k_price_tree; 256 workgroups
// = the tree kernel's count of ratio partials.  The order of a column's sum depends on k and the
// compact numbering only (not on the grid, the tile width or a column sharding).
// ---------------------------------------------------------------------------------
#define PR_GMAX DZG_PR_GMAX
#define PR_BATCH DZG_PR_BATCH // rows per group at least (G = ceil(rows / PR_BATCH), capped at PR_GMAX)
#define PR_PIPE 8   // rows per register set of the streaming loop
template <int VEC, int PIPE = PR_PIPE>
__global__ __launch_bounds__(256) void k_price_rows(
    const DzgCtl *ctl, int rows_T, const double *__restrict__ At, long long ldt,
    const int *__restrict__ drow, const int *__restrict__ bcode, const double *__restrict__ vc,
    double *__restrict__ part, int need_kind = -1)
{
    typedef double vec_t __attribute__((ext_vector_type(2)));
    __shared__ long long s_off[256];
    __shared__ double s_coef[256];
    const int status = ctl->status, k = ctl->ncompact, lp = ctl->leave_pos;
    if (status != DZG_RUNNING || k >= rows_T) return;
    if (need_kind >= 0 && ctl->kind != need_kind) return; // (see k_price_tree)
    // (the group count follows k alone, so that it -- and with it which rows are this workgroup's --
    // is known after the first trip; the leaving variable's code arrives beside the row list)
    int G = (k + 1 + PR_BATCH - 1) / PR_BATCH;
    G = G > (int)gridDim.y ? (int)gridDim.y : G; // (gridDim.y = the finish kernel's gmax)
    const int g = blockIdx.y;
    if (g >= G) return;
    const int lcode = lp >= 0 ? bcode[lp] : 0; // < 0: a slack leaves, row -1 - lcode carries v = 1
    const int nrows = k + (lcode < 0 ? 1 : 0);
    const long long j0 = ((long long)blockIdx.x * 256 + threadIdx.x) * VEC;
    const bool live = j0 < ldt; // (ldt is a multiple of VEC)
    const double *col = At + (live ? j0 : 0);
    double acc[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) acc[e] = 0.0;
    for (int cbase = g; cbase < nrows; cbase += 256 * G) { // block-uniform
        __syncthreads();
        const int c = cbase + (int)threadIdx.x * G;
        if (c < nrows) { // (BTRAN left v in compact numbering; the leaving slack's own entry is 1)
            const int row = c < k ? drow[c] : -1 - lcode;
            s_off[threadIdx.x] = (long long)row * ldt;
            s_coef[threadIdx.x] = c < k ? vc[c] : 1.0;
        }
        __syncthreads();
        const int left = (nrows - cbase + G - 1) / G;
        const int n = left < 256 ? left : 256;
        // Full batches of PIPE rows, two register sets: the loads of batch b + 1 leave before
        // batch b is added (no branch between a load and its use, so the waits count exactly); the
        // rows are added in ascending order whatever the batching.  Then the guarded tail.
        const int bfull = n / PIPE;
        vec_t ra[PIPE][VEC / 2], rb[PIPE][VEC / 2];
#define PR_LOAD(B, REG)                                                                              \
    _Pragma("unroll") for (int u = 0; u < PIPE; ++u) {                                            \
        const double *src = col + s_off[(B) * PIPE + u];                                          \
        _Pragma("unroll") for (int h = 0; h < VEC / 2; ++h)                                          \
            REG[u][h] = __builtin_nontemporal_load(reinterpret_cast<const vec_t *>(src) + h);        \
    }
#define PR_FMA(B, REG)                                                                               \
    _Pragma("unroll") for (int u = 0; u < PIPE; ++u) {                                            \
        const double cf = s_coef[(B) * PIPE + u];                                                 \
        _Pragma("unroll") for (int h = 0; h < VEC / 2; ++h) {                                        \
            acc[2 * h] = fma(cf, REG[u][h].x, acc[2 * h]);                                           \
            acc[2 * h + 1] = fma(cf, REG[u][h].y, acc[2 * h + 1]);                                   \
        }                                                                                            \
    }
        int bi = 0;
        if (bfull > 0) { PR_LOAD(0, ra) }
        while (bi + 2 < bfull) {
            PR_LOAD(bi + 1, rb)
            PR_FMA(bi, ra)
            PR_LOAD(bi + 2, ra)
            PR_FMA(bi + 1, rb)
            bi += 2;
        }
        if (bfull - bi == 2) {
            PR_LOAD(bi + 1, rb)
            PR_FMA(bi, ra)
            PR_FMA(bi + 1, rb)
        } else if (bfull - bi == 1) {
            PR_FMA(bi, ra)
        }
#undef PR_LOAD
#undef PR_FMA
        {
            const int i0 = bfull * PIPE; // the tail: fewer than PIPE rows
#pragma unroll
            for (int u = 0; u < PIPE; ++u)
                if (i0 + u < n) {
                    const double *src = col + s_off[i0 + u];
#pragma unroll
                    for (int h = 0; h < VEC / 2; ++h)
                        ra[u][h] = __builtin_nontemporal_load(reinterpret_cast<const vec_t *>(src) + h);
                }
#pragma unroll
            for (int u = 0; u < PIPE; ++u)
                if (i0 + u < n) {
                    const double cf = s_coef[i0 + u];
#pragma unroll
                    for (int h = 0; h < VEC / 2; ++h) {
                        acc[2 * h] = fma(cf, ra[u][h].x, acc[2 * h]);
                        acc[2 * h + 1] = fma(cf, ra[u][h].y, acc[2 * h + 1]);
                    }
                }
        }
    }
    if (live) {
        double *dst = part + (long long)g * ldt + j0;
#pragma unroll
        for (int e = 0; e < VEC; ++e) dst[e] = acc[e];
    }
}

// ---------------------------------------------------------------------------------
// k_price_rows_small: the row-wise pass AND its finishing step in one launch, for the first few
// hundred pivots of a solve (k + 1 <= PRS_ROWS for the whole batch: the host's bound, dzg_price_small).
// A workgroup of 512 threads owns a tile of 64 consecutive COLUMNS of the row-major copy (512
// contiguous bytes per row: a wave's load is one coalesced segment); wave w adds the row groups
// g = w, w + 8, ... (group g = rows c = g, g + G, ... ascending, G as in k_price_rows) into LDS,
// wave 0 then adds a column's G partial sums in group order -- exactly k_price_rows +
// k_price_rows_finish's sums, bit for bit -- and, knowing where each column sits among the nonbasic
// positions (cpos, kept by the pivot's books), writes dz and forms the z-side ratio candidates.
// Unit columns and the candidates' per-workgroup reduction as in the finishing launch.  No partial
// sums in memory, no second launch, nothing for k_chain_post to fold: three dependent trips
// (control block | row list, coefficients, positions | rows, z, zbar).
// grid = ceil(ldt / 64) workgroups = its count of ratio partials.
// ---------------------------------------------------------------------------------
// WIDE: the loads of a wave's four groups leave together (G > 8: k >= 128); otherwise one group
// per wave and fewer registers (four workgroups per CU instead of two: config 5 has 1 024 tiles).
#define PRS_TILE 64
#define PRS_ROWS 512
template <bool WIDE>
__global__ __launch_bounds__(512) void k_price_rows_small(
    const DzgCtl *ctl, int rows_T, const double *__restrict__ At, long long ldt, int ncols,
    const int *__restrict__ drow, const int *__restrict__ bcode, const double *__restrict__ vc,
    const int *__restrict__ cpos, int q, const int *__restrict__ nbcode, const double *__restrict__ v,
    double *__restrict__ dz, const double *__restrict__ z, const double *__restrict__ zbar,
    double *__restrict__ rz_r, int *__restrict__ rz_k, double *__restrict__ rz_h)
{
    __shared__ long long s_off[PRS_ROWS];
    __shared__ double s_coef[PRS_ROWS];
    __shared__ double s_part[PR_GMAX][PRS_TILE];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // first trip, side by side: the control block, this thread's unit-column position, this lane's
    // column and where it sits
    const int spos = blockIdx.x * 512 + tid;
    int scode = 0;
    if (spos < q) scode = nbcode[spos];
    const long long j = (long long)blockIdx.x * PRS_TILE + lane;
    int pos = -1;
    if (wave == 0 && j < ncols) pos = cpos[j];
    const int status = ctl->status, k = ctl->ncompact, lp = ctl->leave_pos;
    const double mu = ctl->mu, tau = ctl->tau;
    if (status != DZG_RUNNING || k >= rows_T || k >= PRS_ROWS) return;
    // second trip: the row list and its coefficients; z, zbar of this thread's positions
    const int lcode = lp >= 0 ? bcode[lp] : 0; // < 0: a slack leaves, row -1 - lcode carries v = 1
    if (tid < k) {
        s_off[tid] = (long long)drow[tid] * ldt;
        s_coef[tid] = vc[tid];
    }
    double sv = 0.0, sz = 0.0, szb = 0.0, zc = 0.0, zbc = 0.0;
    if (scode < 0) {
        sv = v[-1 - scode];
        sz = z[spos];
        szb = zbar[spos];
    }
    if (pos >= 0) {
        zc = z[pos];
        zbc = zbar[pos];
    }
    const int nrows = k + (lcode < 0 ? 1 : 0);
    if (tid == k && lcode < 0) {
        s_off[k] = (long long)(-1 - lcode) * ldt;
        s_coef[k] = 1.0;
    }
    int G = (k + 1 + PR_BATCH - 1) / PR_BATCH; // (as k_price_rows)
    G = G > PR_GMAX ? PR_GMAX : G;
    __syncthreads();
    // third trip: the rows of this tile -- eight rows of each of the wave's (up to four) groups in
    // flight per lane, so that a pass of k + 1 <= 512 rows is two trips whatever G is; a group's
    // rows are still added in ascending order into the group's own accumulator
    const double *col = At + (j < ldt ? j : 0);
    if (WIDE) {
        double acc[4] = {0.0, 0.0, 0.0, 0.0};
        const int rpg = (nrows + G - 1) / G; // rows of the fullest group
        for (int r0 = 0; r0 < rpg; r0 += 8) { // block-uniform
            double vv[4][8];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int g = wave + 8 * t; // wave-uniform
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int c = g + (r0 + u) * G;
                    vv[t][u] = (g < G && c < nrows) ? __builtin_nontemporal_load(col + s_off[c]) : 0.0;
                }
            }
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int g = wave + 8 * t;
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int c = g + (r0 + u) * G;
                    if (g < G && c < nrows) acc[t] = fma(s_coef[c], vv[t][u], acc[t]);
                }
            }
        }
#pragma unroll
        for (int t = 0; t < 4; ++t)
            if (wave + 8 * t < G) s_part[wave + 8 * t][lane] = acc[t];
    } else {
        for (int g = wave; g < G; g += 8) { // wave-uniform
            double acc = 0.0;
            for (int c0 = g; c0 < nrows; c0 += 8 * G) {
                double vv[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int c = c0 + u * G;
                    vv[u] = c < nrows ? __builtin_nontemporal_load(col + s_off[c]) : 0.0;
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int c = c0 + u * G;
                    if (c < nrows) acc = fma(s_coef[c], vv[u], acc);
                }
            }
            s_part[g][lane] = acc;
        }
    }
    __syncthreads();
    DzgCand2 best = dzg_cand2_none();
    if (pos >= 0) { // (wave 0) the finishing launch's sum: the groups in order
        double sum = 0.0;
        for (int g = 0; g < G; ++g) sum = sum + s_part[g][lane];
        dz[pos] = -sum;
        price_candidate_v(best, -sum, pos, mu, tau, zc, zbc);
    }
    // ---- unit columns (the arithmetic of price_slack_positions)
    if (scode < 0) {
        const double p = 1.0 * -sv;
        const double d = 0.0 + p;
        dz[spos] = d;
        price_candidate_v(best, d, spos, mu, tau, sz, szb);
    }
    for (int ps = spos + (int)gridDim.x * 512; ps < q; ps += (int)gridDim.x * 512) { // (never: 8 threads per column)
        const int code = nbcode[ps];
        if (code < 0) {
            const double p = 1.0 * -v[-1 - code];
            const double d = 0.0 + p;
            dz[ps] = d;
            price_candidate(best, d, ps, mu, tau, z, zbar);
        }
    }
    price_publish(best, rz_r, rz_k, rz_h);
}

__global__ __launch_bounds__(256) void k_price_rows_finish(
    const DzgCtl *ctl, int rows_T, const double *__restrict__ part, long long ldt, int q,
    const int *__restrict__ plist, const int *__restrict__ pcode, const int *__restrict__ nbcode,
    const int *__restrict__ bcode, int col0, const double *__restrict__ v, double *__restrict__ dz,
    const double *__restrict__ z, const double *__restrict__ zbar, double *__restrict__ rz_r,
    int *__restrict__ rz_k, double *__restrict__ rz_h, int gmax, int need_kind = -1)
{
    // first trip, side by side: the control block, this thread's position (unit columns) and its
    // first list entry (plist / pcode hold q entries; those beyond nb_struct are stale and only
    // used once the count has arrived)
    const int nthreads = gridDim.x * blockDim.x;
    const int spos = blockIdx.x * blockDim.x + threadIdx.x;
    int scode = 0, pos0 = -1, code0 = -1;
    if (spos < q) {
        scode = nbcode[spos];
        pos0 = plist[spos];
        code0 = pcode[spos];
    }
    const int status = ctl->status, k = ctl->ncompact;
    const int count = (int)ctl->nb_struct;
    const double mu = ctl->mu, tau = ctl->tau;
    if (status != DZG_RUNNING || k >= rows_T) return;
    if (need_kind >= 0 && ctl->kind != need_kind) return;
    int G = (k + 1 + PR_BATCH - 1) / PR_BATCH; // (as k_price_rows)
    G = G > gmax ? gmax : G;
    DzgCand2 best = dzg_cand2_none();
    double sv = 0.0, sz = 0.0, szb = 0.0;
    if (scode < 0) {
        sv = v[-1 - scode];
        sz = z[spos];
        szb = zbar[spos];
    }
    for (int idx = spos; idx < count; idx += nthreads) {
        const int pos = idx == spos ? pos0 : plist[idx];
        const int code = idx == spos ? code0 : pcode[idx];
        if (code < 0) continue; // (the list holds structural columns only)
        const double zc = z[pos], zbc = zbar[pos];
        const double *src = part + (code - col0);
        double sum = 0.0;
        int gg = 0;
        for (; gg + 8 <= G; gg += 8) { // eight partials side by side, added in group order
            double t[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) t[e] = src[(long long)(gg + e) * ldt];
#pragma unroll
            for (int e = 0; e < 8; ++e) sum = sum + t[e];
        }
        for (; gg < G; ++gg) sum = sum + src[(long long)gg * ldt];
        dz[pos] = -sum;
        price_candidate_v(best, -sum, pos, mu, tau, zc, zbc);
    }
    // ---- unit columns (the arithmetic of price_slack_positions)
    if (scode < 0) {
        const double p = 1.0 * -sv;
        const double d = 0.0 + p;
        dz[spos] = d;
        price_candidate_v(best, d, spos, mu, tau, sz, szb);
    }
    for (int pos = spos + nthreads; pos < q; pos += nthreads) { // more positions than threads
        const int code = nbcode[pos];
        if (code < 0) {
            const double p = 1.0 * -v[-1 - code];
            const double d = 0.0 + p;
            dz[pos] = d;
            price_candidate(best, d, pos, mu, tau, z, zbar);
        }
    }
    price_publish(best, rz_r, rz_k, rz_h);
}

// column-major m x n (lda) -> row-major m x ldt, 32 x 32 tiles through LDS; columns n..ldt-1 of a
// row are zero.  grid (ceil(ldt / 32), ceil(m / 32)), block (32, 8).
__global__ __launch_bounds__(256) void k_transpose_to_rows(const double *__restrict__ A, long long lda,
                                                           int m, int n, double *__restrict__ At,
                                                           long long ldt)
{
    __shared__ double tile[32][33];
    const int j0 = blockIdx.x * 32, i0 = blockIdx.y * 32;
    for (int jj = threadIdx.y; jj < 32; jj += 8) {
        const int j = j0 + jj, i = i0 + threadIdx.x;
        tile[jj][threadIdx.x] = (j < n && i < m) ? A[(long long)j * lda + i] : 0.0;
    }
    __syncthreads();
    for (int ii = threadIdx.y; ii < 32; ii += 8) {
        const int i = i0 + ii, j = j0 + threadIdx.x;
        if (i < m && j < ldt) At[(long long)i * ldt + j] = tile[threadIdx.x][ii];
    }
}

#define DZG_PRICE_CSC_BLOCKS 2048
#define DZG_PRICE_SEQ_BLOCKS 256   // x 4 waves: one workgroup per CU
#define DZG_PRICE_WAVE_BLOCKS 2048 // x 4 waves
#define DZG_PRICE_TREE_BLOCKS 256  // x 4 waves
