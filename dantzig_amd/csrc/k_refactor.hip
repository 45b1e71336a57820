// k_refactor.hip -- FAST numerics: rebuild the compact basis inverse from scratch on the device.
//
// The reference refactorises in every iteration (Matrix::factorize, src/linalg.rs:88-128, twice
// per pivot).  FAST numerics updates an explicit inverse instead and only REFACTORS now and then
// (opts.refactor_interval, or when the pivot-consistency monitor drifts), to shed the rounding the
// updates have accumulated.  With B = [A_S | E] (k structural basics S, rows R whose slack is
// nonbasic) only the k x k block G = A[R, S] needs factorising:
//
//   1. gather G (row a = dense row drow[a], column b = b-th structural basic position)
//   2. blocked right-looking LU with partial pivoting, panel width NB = 64, panels in PAIRS: the
//      panel itself in sub-panels of 4-8 columns on a compact column-major copy (pivot search =
//      first maximum of |.|, swap, scale, rank-1 inside the sub-panel by one workgroup with the
//      rows in registers; rank-8 update of the rest of the panel by the whole chip)
//        ->  row swaps outside the panel, the 64 x 64 diagonal blocks L11, U11 inverted beside them
//        ->  U12 = L11^-1 A12 as a product  ->  after two panels:
//        trailing update A22 -= [L21a L21b] [U12a; U12b] (rank 128) on the fp64 matrix cores
//        (v_mfma_f64_16x16x4_f64)
//   3. X = G^-1 by blocked forward / backward substitution on the identity, every block product an
//      MFMA GEMM, the diagonal blocks by their inverses.  The forward half keeps X unit lower
//      triangular (columns in pivot order, put back by one permutation at the end): k^3 / 3 flops
//      instead of k^3
//   4. Binv0[p_b, :] = X[b, :] for structural positions, Binv0[p', :] = -A[r', S] * X for the
//      position of the basic slack of row r' (one more MFMA GEMM), eta file emptied.
//
// 2 k^3 + 2 (m-k) k^2 flops, all but the panels in GEMM form.
#include <cstdlib>

#include "common.h"

typedef double double4_t __attribute__((ext_vector_type(4)));

#define NB 64

// ---------------------------------------------------------------------------------
// C[M x N] (+)= alpha * A[M x K] * B[K x N], all row-major.  One wave owns a 16 x 64 strip of C
// (4 MFMA tiles) and walks K in steps of 4.  ZERO_C: C = -A*B (no read of C); otherwise
// C -= A*B.  crow (optional) maps strip rows to rows of C (scatter).
// Lane maps (cdna_hip_programming.md section 3): A[i = l&15][k = l>>4], B[k = l>>4][j = l&15],
// C/D[row = (l>>4) + 4*reg][col = l&15].
// ---------------------------------------------------------------------------------
template <bool ZERO_C>
__global__ __launch_bounds__(256) void k_ref_gemm(int M, int N, int K, const double *__restrict__ A,
                                                  long long lda, const double *__restrict__ B,
                                                  long long ldb, double *__restrict__ C,
                                                  long long ldc, const int *__restrict__ crow,
                                                  int crow0 = 0, int crow1 = 0)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c0 = blockIdx.x * 64;
    const int i0 = (blockIdx.y * 4 + wave) * 16;
    if (c0 >= N || i0 >= M) return;
    // (crow0, crow1) != (0, 0) (a row-sharded basis side): only the rows of C in [crow0, crow1) exist on this
    // device; crow ascends, so a strip whose first and last target lie outside has nothing to do
    const bool ranged = crow && !(crow0 == 0 && crow1 == 0); // (0, 0): every row
    if (ranged) {
        const int first = crow[i0], last = crow[i0 + 15 < M ? i0 + 15 : M - 1];
        if (last < crow0 || first >= crow1) return;
    }
    const int li = lane & 15, lk = lane >> 4;
    double4_t acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int row = i0 + lk + 4 * g, col = c0 + 16 * j + li;
            double cv = 0.0;
            if (!ZERO_C && row < M && col < N)
                cv = C[(long long)(crow ? crow[row] : row) * ldc + col];
            acc[j][g] = cv;
        }
    }
    const int arow = i0 + li;
    const double *ap = A + (long long)(arow < M ? arow : 0) * lda;
    for (int t0 = 0; t0 < K; t0 += 4) {
        const int t = t0 + lk;
        const bool tin = t < K;
        const double a = (arow < M && tin) ? -ap[t] : 0.0;
        const double *bp = B + (long long)(tin ? t : 0) * ldb;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int col = c0 + 16 * j + li;
            const double b = (tin && col < N) ? bp[col] : 0.0;
            acc[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[j], 0, 0, 0);
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int row = i0 + lk + 4 * g, col = c0 + 16 * j + li;
            if (row < M && col < N) {
                const int cr = crow ? crow[row] : row;
                if (!ranged || (cr >= crow0 && cr < crow1)) C[(long long)cr * ldc + col] = acc[j][g];
            }
        }
    }
}

// ---------------------------------------------------------------------------------
// k_ref_gemm_lds: the same product with B staged through LDS.  The four waves of a workgroup own
// four 16 x 64 strips of the SAME 64 columns of C, so they multiply by the same K x 64 panel of B:
// it is fetched once per workgroup, 64 rows of K at a time (32 KB, coalesced 16-byte loads, one
// trip), and every MFMA takes its B operand from LDS; only A (one value per lane and 4 MFMAs, four
// steps per trip) and the C tile itself still come from memory.  In k_ref_gemm every wave loads
// its own copy of B from L2 -- 5 loads per 4 MFMAs, the L1 at its limit: 25 % of the fp64 MFMA
// peak on the rank-64 updates of the factorisation.  Occupancy stays at 4 waves per SIMD (128
// VGPRs, 36 KB of LDS per workgroup): latency is hidden by waves, as there.
// Same lane maps, same order of the K steps: bit-identical results.
// ---------------------------------------------------------------------------------
#define GKC 64  // rows of B per LDS panel
#define GLD 72  // LDS row stride in doubles (k -> k + 1 moves 16 banks on)

typedef double double2r_t __attribute__((ext_vector_type(2)));

template <bool ZERO_C>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 8))) void k_ref_gemm_lds(
    int M, int N, int K, const double *__restrict__ A, long long lda, const double *__restrict__ B,
    long long ldb, double *__restrict__ C, long long ldc, const int *__restrict__ crow,
    int crow0 = 0, int crow1 = 0)
{
    __shared__ __attribute__((aligned(16))) double s_b[GKC][GLD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c0 = blockIdx.x * 64;
    const int i0 = (blockIdx.y * 4 + wave) * 16;
    const int li = lane & 15, lk = lane >> 4;
    // (see k_ref_gemm; the test is on the WORKGROUP's 64 rows: the barriers below stay uniform)
    const bool ranged = crow && !(crow0 == 0 && crow1 == 0); // (0, 0): every row
    if (ranged) {
        const int b0 = blockIdx.y * 64;
        if (b0 < M) {
            const int first = crow[b0], last = crow[b0 + 63 < M ? b0 + 63 : M - 1];
            if (last < crow0 || first >= crow1) return;
        }
    }
    double4_t acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int row = i0 + lk + 4 * g, col = c0 + 16 * j + li;
            double cv = 0.0;
            if (!ZERO_C && row < M && col < N) cv = C[(long long)(crow ? crow[row] : row) * ldc + col];
            acc[j][g] = cv;
        }
    }
    const int arow = i0 + li;
    const double *ap = A + (long long)(arow < M ? arow : M - 1) * lda;
    // staging map of B: thread loads 8 x double2: k = (tid >> 5) + 8 u, columns 2 (tid & 31) ..+1
    const int b_k = tid >> 5, b_c = 2 * (tid & 31);
    for (int k0 = 0; k0 < K; k0 += GKC) {
        double2r_t rb[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int kk = k0 + b_k + 8 * u, c = c0 + b_c;
            const double *p = B + (long long)(kk < K ? kk : K - 1) * ldb;
            double2r_t w;
            w.x = p[c < N ? c : N - 1];
            w.y = p[c + 1 < N ? c + 1 : N - 1];
            if (kk >= K || c >= N) w.x = 0.0;
            if (kk >= K || c + 1 >= N) w.y = 0.0;
            rb[u] = w;
        }
        if (k0 > 0) __syncthreads(); // everybody is done with the previous panel
#pragma unroll
        for (int u = 0; u < 8; ++u) *reinterpret_cast<double2r_t *>(&s_b[b_k + 8 * u][b_c]) = rb[u];
        __syncthreads();
        for (int s0 = 0; s0 < GKC / 4; s0 += 4) { // A: four steps' values per trip
            double av[4];
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const int t = k0 + 4 * (s0 + s) + lk;
                av[s] = ap[t < K ? t : K - 1];
            }
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const bool tin = k0 + 4 * (s0 + s) + lk < K;
                const double a = (arow < M && tin) ? -av[s] : 0.0;
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, s_b[4 * (s0 + s) + lk][16 * j + li], acc[j],
                                                                  0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int row = i0 + lk + 4 * g, col = c0 + 16 * j + li;
            if (row < M && col < N) {
                const int cr = crow ? crow[row] : row;
                if (!ranged || (cr >= crow0 && cr < crow1)) C[(long long)cr * ldc + col] = acc[j][g];
            }
        }
    }
}

// ---------------------------------------------------------------------------------
// lists: structural basic positions in position order (k of them), basic-slack positions
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void k_ref_lists(DzgCtl *ctl, int m,
                                                    const int *__restrict__ basis,
                                                    const int *__restrict__ var_col,
                                                    int *__restrict__ spos, int *__restrict__ scode,
                                                    int *__restrict__ lpos, int *__restrict__ lrow,
                                                    int *__restrict__ counts)
{
    // one workgroup; thread t owns positions [t*chunk, (t+1)*chunk): count, scan, fill
    __shared__ int s_cnt[1024];
    const int tid = threadIdx.x;
    const int chunk = (m + 1023) / 1024;
    const int lo = tid * chunk, hi = (lo + chunk) < m ? (lo + chunk) : m;
    int ns = 0;
    for (int p = lo; p < hi; ++p) ns += var_col[basis[p]] >= 0 ? 1 : 0;
    s_cnt[tid] = ns;
    __syncthreads();
    if (tid == 0) {
        int run = 0;
        for (int t = 0; t < 1024; ++t) {
            const int c = s_cnt[t];
            s_cnt[t] = run;
            run += c;
        }
        counts[0] = run;
        counts[1] = m - run;
    }
    __syncthreads();
    int is = s_cnt[tid];      // structural basics before this chunk
    int il = lo < m ? lo - is : 0; // basic slacks before this chunk
    for (int p = lo; p < hi; ++p) {
        const int code = var_col[basis[p]];
        if (code >= 0) {
            spos[is] = p;
            scode[is] = code;
            ++is;
        } else {
            lpos[il] = p;
            lrow[il] = -1 - code;
            ++il;
        }
    }
    (void)ctl;
}

// G[a][b] = A[drow[a], column of the b-th structural basic]; X = I.  grid (ceil(k/256), k)
// own1 > own0: a PARTITIONED column-sharded rank holds the columns [own0, own1) only; the others
// contribute +0.0 here and arrive through the sum over the ranks that follows (engine.hip: every
// column has exactly one owner, so the sum is that owner's value, exactly)
__global__ __launch_bounds__(256) void k_ref_gather(int k, const double *__restrict__ A,
                                                    long long lda, int col0,
                                                    const int *__restrict__ drow,
                                                    const int *__restrict__ scode,
                                                    double *__restrict__ G, double *__restrict__ X,
                                                    long long ldg, int own0, int own1)
{
    const int a = blockIdx.y;
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= k) return;
    const int code = scode[b];
    const bool have = own1 <= own0 || (code >= own0 && code < own1);
    G[(long long)a * ldg + b] = have ? A[(long long)(code - col0) * lda + drow[a]] : 0.0;
    X[(long long)a * ldg + b] = (a == b) ? 1.0 : 0.0;
}

// Sparse (CSC) input: the same two gathers as scatters.  rowmap[i] = compact index of row i in
// the target (dslot for G, the basic-slack list index for As), < 0: row not wanted.
// grid = k workgroups (one per structural basic column); the targets are zeroed beforehand.
__global__ __launch_bounds__(256) void k_ref_scatter_csc(int k, const long long *__restrict__ cptr,
                                                         const int *__restrict__ ridx,
                                                         const double *__restrict__ cval, int col0,
                                                         const int *__restrict__ scode,
                                                         const int *__restrict__ rowmap,
                                                         double *__restrict__ T, long long ldt,
                                                         int own0, int own1)
{
    const int b = blockIdx.x;
    if (b >= k) return;
    if (own1 > own0 && (scode[b] < own0 || scode[b] >= own1)) return; // (another rank's column)
    const int code = scode[b] - col0;
    for (long long e = cptr[code] + threadIdx.x; e < cptr[code + 1]; e += blockDim.x) {
        const int a = rowmap[ridx[e]];
        if (a >= 0) T[(long long)a * ldt + b] = cval[e];
    }
}

// X = I, G = 0 (k x k);   grid (ceil(k/256), k)
__global__ __launch_bounds__(256) void k_ref_identity(int k, double *__restrict__ G,
                                                      double *__restrict__ X, long long ldg)
{
    const int a = blockIdx.y;
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= k) return;
    G[(long long)a * ldg + b] = 0.0;
    X[(long long)a * ldg + b] = (a == b) ? 1.0 : 0.0;
}

// lslot[row] = index of the row in the basic-slack list, -1 for the other rows
__global__ __launch_bounds__(256) void k_ref_lslot(int m, int nl, const int *__restrict__ lrow,
                                                   int *__restrict__ lslot)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < m) lslot[i] = -1;
}
__global__ __launch_bounds__(256) void k_ref_lslot_fill(int nl, const int *__restrict__ lrow,
                                                        int *__restrict__ lslot)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nl) lslot[lrow[i]] = i;
}

// The 64-column panel is factorised in sub-panels of SPW columns (a second level of blocking:
// an elimination step then sweeps k x SPW values instead of k x 64, and the rest of the panel
// gets one rank-SPW update per sub-panel from the whole chip instead of 64 rank-1 sweeps from one
// workgroup).
//   k_ref_subpanel (one workgroup): unblocked LU with partial pivoting (first maximum of |.|) of
//     columns [c0, c0 + w); then the same workgroup applies the w row swaps to the other columns
//     of the enclosing panel [pj0, pj0 + pnbw) and solves U12' = L11^-1 A12' for the panel
//     columns right of the sub-panel (a w x w unit-lower triangle: one thread per column);
//   k_ref_subupdate (whole chip): A22' -= L21 U12' on rows below the sub-panel, panel columns
//     right of it -- one thread per row, U12' in LDS.
//
// The panel is factorised in a COMPACT COLUMN-MAJOR copy Pn (element (row i, panel column cc) at
// Pn[cc * ldp + i]; k_ref_panel_in / _out transpose it out of and back into G with the whole
// chip).  The sub-panel kernels run on ONE compute unit: in G a row's sub-panel segment is 32-64
// bytes of its own 128-byte line, 4 096-8 192 lines in and as many out through one CU's 64 bytes
// per clock -- some 25 of the 34-38 us a launch took, whatever the elimination steps cost.  In Pn
// consecutive threads hold consecutive rows of a column: every access is a full line.
#define SPW 8
#define PN(i, cc) Pn[(long long)(cc) * ldp + (i)]
__global__ __launch_bounds__(1024) void k_ref_subpanel(int k, int pj0, int pnbw, int c0, int w,
                                                       double *__restrict__ Pn, long long ldp,
                                                       int *__restrict__ piv,
                                                       int *__restrict__ singular)
{
    __shared__ int s_p;
    __shared__ double s_pivrow[SPW];
    const int tid = threadIdx.x;
    const int cc0 = c0 - pj0;
    for (int jj = 0; jj < w; ++jj) {
        const int col = c0 + jj;
        DzgCand best;
        best.r = 0.0;
        best.k = -1;
        for (int i = col + tid; i < k; i += blockDim.x) {
            DzgCand c;
            c.r = fabs(PN(i, cc0 + jj));
            c.k = i;
            if (c.r == c.r) best = dzg_better(best, c);
        }
        best = dzg_block_best(best);
        if (tid == 0) {
            s_p = best.k >= 0 ? best.k : col;
            piv[col] = s_p;
            if (!(best.r > 0.0)) *singular = 1;
        }
        __syncthreads();
        const int p = s_p;
        // swap rows col <-> p inside the sub-panel, keep the pivot row in LDS
        if (tid < w) {
            const double a = PN(col, cc0 + tid), b = PN(p, cc0 + tid);
            PN(col, cc0 + tid) = b;
            PN(p, cc0 + tid) = a;
            s_pivrow[tid] = b;
        }
        __syncthreads();
        const double pv = s_pivrow[jj];
        const double rpv = pv != 0.0 ? 1.0 / pv : 0.0;
        for (int i = col + 1 + tid; i < k; i += blockDim.x) {
            const double l = PN(i, cc0 + jj) * rpv;
            PN(i, cc0 + jj) = l;
            for (int c = jj + 1; c < w; ++c) PN(i, cc0 + c) = fma(-l, s_pivrow[c], PN(i, cc0 + c));
        }
        __syncthreads();
    }
    // the other columns of the enclosing panel: row swaps (in pivot order), then the columns
    // right of the sub-panel take  U12' = L11^-1 A12'
    const int nother = pnbw - w;
    if (tid < nother) {
        const int off = tid < cc0 ? tid : tid + w; // skip the sub-panel's own columns
        double *colp = Pn + (long long)off * ldp;
        for (int jj = 0; jj < w; ++jj) {
            const int r = c0 + jj, p = piv[r];
            if (p != r) {
                const double a = colp[r], b = colp[p];
                colp[r] = b;
                colp[p] = a;
            }
        }
        if (off >= cc0 + w) { // right of the sub-panel
            double y[SPW];
            for (int i = 0; i < w; ++i) y[i] = colp[c0 + i];
            for (int i = 1; i < w; ++i) {
                double acc = y[i];
                for (int j = 0; j < i; ++j) acc = fma(-PN(c0 + i, cc0 + j), y[j], acc);
                y[i] = acc;
            }
            for (int i = 0; i < w; ++i) colp[c0 + i] = y[i];
        }
    }
}

// Pn[cc][i] = G[i][pj0 + cc] for rows [pj0, k) (and back): 64 x 64 tiles through LDS, both sides
// coalesced.  grid ceil((k - pj0) / 64)
template <bool OUT>
__global__ __launch_bounds__(256) void k_ref_panel_copy(int k, int pj0, int pnbw, double *__restrict__ G,
                                                        long long ldg, double *__restrict__ Pn,
                                                        long long ldp)
{
    __shared__ double s_t[NB][NB + 1];
    const int r0 = pj0 + blockIdx.x * NB;
    for (int e = threadIdx.x; e < NB * NB; e += blockDim.x) {
        const int hi = e / NB, lo = e % NB; // lo runs along the contiguous side of the SOURCE
        if (OUT) {
            if (hi < pnbw && r0 + lo < k) s_t[lo][hi] = PN(r0 + lo, hi);
        } else {
            if (r0 + hi < k && lo < pnbw) s_t[hi][lo] = G[(long long)(r0 + hi) * ldg + pj0 + lo];
        }
    }
    __syncthreads();
    for (int e = threadIdx.x; e < NB * NB; e += blockDim.x) {
        const int hi = e / NB, lo = e % NB; // lo along the contiguous side of the TARGET
        if (OUT) {
            if (r0 + hi < k && lo < pnbw) G[(long long)(r0 + hi) * ldg + pj0 + lo] = s_t[hi][lo];
        } else {
            if (hi < pnbw && r0 + lo < k) PN(r0 + lo, hi) = s_t[lo][hi];
        }
    }
}

// The same sub-panel step with the active rows held in REGISTERS: thread t owns rows
// c0 + t, c0 + t + 1024, ... (SPR of them at most, i.e. k - c0 <= 1024 * SPR), loads their W
// values once, runs the w elimination steps on registers -- per step one block-wide max-loc, one
// exchange of the two swapped rows through LDS -- and writes them back once.
//
// The kernel runs on ONE compute unit, 4 waves per SIMD, and a wave issues one instruction every
// four cycles at best: an elimination step costs what its instruction count says.  The first
// version spent some 1 000 instructions per step and wave (3 us): every row of every thread asked
// "am I row col? am I row p? am I below col?" with its own compare-select or exec-mask branch,
// and the max-loc went through 36 ds_bpermute.  Now: the two rows that trade places are handled
// inside wave-uniform branches (the pivot row's wave is read off the scalar p; row col is always
// thread jj's first row), only a thread's FIRST row can lie above the diagonal (its other rows sit
// 1 024 positions further down), rows beyond k are zero rows that eliminate to zero, and the
// max-loc runs on DPP row operations (4 levels inside a row of 16 lanes; 2 ds_bpermute levels
// across rows; the 16 wave winners again as one row of 16).
struct DzgPiv {
    double v; // |a|, or -1: no candidate
    int k;
};
__device__ __forceinline__ DzgPiv dzg_piv_better(DzgPiv a, DzgPiv b)
{
    // the first maximum of |.| in row order: larger value, on equal values the lower row
    const bool bw = b.v > a.v || (b.v == a.v && b.k < a.k);
    DzgPiv r;
    r.v = bw ? b.v : a.v;
    r.k = bw ? b.k : a.k;
    return r;
}
template <int CTRL>
__device__ __forceinline__ DzgPiv dzg_piv_dpp(DzgPiv c)
{
    int lo = __double2loint(c.v), hi = __double2hiint(c.v);
    lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xf, 0xf, false);
    DzgPiv o;
    o.v = __hiloint2double(hi, lo);
    o.k = __builtin_amdgcn_update_dpp(c.k, c.k, CTRL, 0xf, 0xf, false);
    return dzg_piv_better(c, o);
}
// every lane of a row of 16 ends with the row's winner: quad_perm [1,0,3,2], quad_perm [2,3,0,1],
// row_half_mirror, row_mirror (the operation is commutative and idempotent)
__device__ __forceinline__ DzgPiv dzg_piv_row16(DzgPiv c)
{
    c = dzg_piv_dpp<0xB1>(c);
    c = dzg_piv_dpp<0x4E>(c);
    c = dzg_piv_dpp<0x141>(c);
    c = dzg_piv_dpp<0x140>(c);
    return c;
}

template <int SPR, int W> // SPR x W values per thread: 4 x 8 and 8 x 4 stay in registers
__global__ __launch_bounds__(1024) void k_ref_subpanel_reg(int k, int pj0, int pnbw, int c0, int w,
                                                           double *__restrict__ Pn, long long ldp,
                                                           int *__restrict__ piv,
                                                           int *__restrict__ singular)
{
    const int cc0 = c0 - pj0;
    // two workgroup barriers per elimination step: every LDS array the steps use is double-buffered
    // by the parity of the step, so nothing written in step jj + 1 can be something a thread still
    // reads in step jj - 1
    __shared__ double s_rowc[2][W], s_rowp[2][W];
    __shared__ double s_cv[2][16];
    __shared__ int s_ck[2][16];
    __shared__ int s_piv[W];
    __shared__ double s_l11[W][W];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    double a[SPR][W];
#pragma unroll
    for (int r = 0; r < SPR; ++r) {
        const int i = c0 + tid + 1024 * r;
#pragma unroll
        for (int c = 0; c < W; ++c) a[r][c] = (i < k && c < w) ? PN(i, cc0 + c) : 0.0;
    }
#pragma unroll
    for (int jj = 0; jj < W; ++jj) {
        if (jj >= w) break;
        const int col = c0 + jj;
        DzgPiv best;
        best.v = -1.0;
        best.k = -1;
#pragma unroll
        for (int r = 0; r < SPR; ++r) {
            const int i = c0 + tid + 1024 * r;
            const double x = a[r][jj];
            bool ok = i < k && x == x;
            if (r == 0) ok = ok && tid >= jj; // (rows c0 .. col - 1 are done)
            const double v = ok ? fabs(x) : -1.0;
            if (v > best.v) { // rows in ascending order: the first maximum stays
                best.v = v;
                best.k = i;
            }
        }
        const int buf = jj & 1;
        best = dzg_piv_row16(best);
#pragma unroll
        for (int off = 16; off <= 32; off <<= 1) {
            DzgPiv o;
            o.v = __shfl_xor(best.v, off, DZG_WAVE);
            o.k = __shfl_xor(best.k, off, DZG_WAVE);
            best = dzg_piv_better(best, o);
        }
        if (lane == 0) {
            s_cv[buf][wave] = best.v;
            s_ck[buf][wave] = best.k;
        }
        __syncthreads();
        best.v = s_cv[buf][lane & 15];
        best.k = s_ck[buf][lane & 15];
        best = dzg_piv_row16(best); // the 16 wave winners: one row of 16 lanes, in every row
        const int p = __builtin_amdgcn_readfirstlane(best.k >= 0 ? best.k : col);
        if (tid == 0) {
            piv[col] = p;
            s_piv[jj] = p;
            if (!(best.v > 0.0)) *singular = 1;
        }
        // the two rows trade places through LDS: row col is thread jj's first row, row p sits in
        // thread pt, slot ps
        const int pt = (p - c0) & 1023, ps = (p - c0) >> 10;
        if (wave == 0) {
            if (tid == jj)
#pragma unroll
                for (int c = 0; c < W; ++c) s_rowc[buf][c] = a[0][c];
        }
        if (wave == (pt >> 6)) {
            if (tid == pt)
#pragma unroll
                for (int r = 0; r < SPR; ++r)
                    if (r == ps)
#pragma unroll
                        for (int c = 0; c < W; ++c) s_rowp[buf][c] = a[r][c];
        }
        __syncthreads();
        double prow[W];
#pragma unroll
        for (int c = 0; c < W; ++c) prow[c] = s_rowp[buf][c];
        const double pv = prow[jj];
        const double rpv = pv != 0.0 ? 1.0 / pv : 0.0;
        if (wave == 0) {
            if (tid == jj)
#pragma unroll
                for (int c = 0; c < W; ++c) a[0][c] = prow[c];
        }
        if (p != col && wave == (pt >> 6)) {
            if (tid == pt)
#pragma unroll
                for (int r = 0; r < SPR; ++r)
                    if (r == ps)
#pragma unroll
                        for (int c = 0; c < W; ++c) a[r][c] = s_rowc[buf][c];
        }
        // eliminate below the diagonal: a thread's rows 1 .. SPR - 1 always are; rows beyond k
        // are zero rows (never stored, never candidates)
        if (tid > jj) {
            const double l = a[0][jj] * rpv;
            a[0][jj] = l;
#pragma unroll
            for (int c = 0; c < W; ++c)
                if (c > jj) a[0][c] = fma(-l, prow[c], a[0][c]);
        }
#pragma unroll
        for (int r = 1; r < SPR; ++r) {
            const double l = a[r][jj] * rpv;
            a[r][jj] = l;
#pragma unroll
            for (int c = 0; c < W; ++c)
                if (c > jj) a[r][c] = fma(-l, prow[c], a[r][c]);
        }
    }
#pragma unroll
    for (int r = 0; r < SPR; ++r) {
        const int i = c0 + tid + 1024 * r;
        if (i < k) {
#pragma unroll
            for (int c = 0; c < W; ++c)
                if (c < w) PN(i, cc0 + c) = a[r][c];
        }
    }
    if (tid < w) // the sub-panel's own triangle for the tail below (rows c0 .. c0 + w - 1)
#pragma unroll
        for (int c = 0; c < W; ++c) s_l11[tid][c] = a[0][c];
    __syncthreads();
    // the other columns of the enclosing panel: row swaps, then U12' = L11^-1 A12' (as above).
    // The w swaps are applied to REGISTER copies of the (at most 2 w) rows they touch -- rows
    // c0 .. c0 + w - 1 and the pivot rows -- fetched together: taken one swap after the other
    // through memory (two dependent loads and two stores each) this tail was most of the kernel.
    // Pivots and triangle come from LDS: one trip to memory for the values, one back.
    const int nother = pnbw - w;
    if (tid < nother) {
        const int off = tid < cc0 ? tid : tid + w;
        double *colp = Pn + (long long)off * ldp;
        int pr[W];
        double top[W], piv_v[W]; // values at rows c0 + jj and at rows p_jj (before any swap)
#pragma unroll
        for (int jj = 0; jj < W; ++jj) pr[jj] = jj < w ? s_piv[jj] : c0 + jj;
#pragma unroll
        for (int jj = 0; jj < W; ++jj) {
            top[jj] = jj < w ? colp[c0 + jj] : 0.0;
            piv_v[jj] = jj < w ? colp[pr[jj]] : 0.0;
        }
        // replay the swaps on the copies: a row's current value lives in top[] if it is one of the
        // first w rows, else in the piv_v[] slot of the FIRST step that names it
#pragma unroll
        for (int jj = 0; jj < W; ++jj) {
            if (jj >= w) break;
            const int p = pr[jj];
            if (p == c0 + jj) continue;
            if (p < c0 + w) { // both among the first w rows
                const double x0 = top[jj];
#pragma unroll
                for (int t = 0; t < W; ++t)
                    if (t == p - c0) {
                        top[jj] = top[t];
                        top[t] = x0;
                    }
            } else {
                int slot = jj; // the first step whose pivot row is p holds row p's current value
#pragma unroll
                for (int t = W - 1; t >= 0; --t)
                    if (t < jj && pr[t] == p) slot = t;
                const double x0 = top[jj];
#pragma unroll
                for (int t = 0; t < W; ++t)
                    if (t == slot) {
                        top[jj] = piv_v[t];
                        piv_v[t] = x0;
                    }
            }
        }
        // rows below the first w that a swap touched: the slot of the first step naming them
#pragma unroll
        for (int jj = 0; jj < W; ++jj) {
            if (jj >= w || pr[jj] < c0 + w) continue;
            bool first = true;
#pragma unroll
            for (int t = 0; t < W; ++t)
                if (t < jj && pr[t] == pr[jj]) first = false;
            if (first) colp[pr[jj]] = piv_v[jj];
        }
        if (off >= cc0 + w) {
            double y[W];
#pragma unroll
            for (int i = 0; i < W; ++i) y[i] = i < w ? top[i] : 0.0;
#pragma unroll
            for (int i = 1; i < W; ++i) {
                double acc = y[i];
#pragma unroll
                for (int j = 0; j < W; ++j)
                    if (j < i && i < w) acc = fma(-s_l11[i][j], y[j], acc);
                y[i] = acc;
            }
#pragma unroll
            for (int i = 0; i < W; ++i)
                if (i < w) colp[c0 + i] = y[i];
        } else {
#pragma unroll
            for (int i = 0; i < W; ++i)
                if (i < w) colp[c0 + i] = top[i];
        }
    }
}

// rows [c0 + w, k), panel columns [c0 + w, pj0 + pnbw):  A22' -= L21 * U12'.  One thread per row of
// the compact panel and group of 8 columns: its w multipliers in registers, then column after
// column (consecutive threads = consecutive rows of a column: full lines), U12' in LDS.
// grid (ceil((k - c0 - w) / 256), ceil(columns / 8)): one thread per row over ALL columns left 32
// workgroups for 8 192 rows, 17 us of dependent load-FMA-store per call.
__global__ __launch_bounds__(256) void k_ref_subupdate(int k, int pj0, int pnbw, int c0, int w,
                                                       double *__restrict__ Pn, long long ldp)
{
    __shared__ double s_u[SPW][NB];
    const int cc0 = c0 - pj0, ccr = cc0 + w, nright = pnbw - ccr;
    for (int e = threadIdx.x; e < w * nright; e += blockDim.x)
        s_u[e / nright][e % nright] = PN(c0 + e / nright, ccr + e % nright);
    __syncthreads();
    const int i = c0 + w + blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= k) return;
    double l[SPW];
#pragma unroll
    for (int j = 0; j < SPW; ++j) l[j] = j < w ? PN(i, cc0 + j) : 0.0;
    const int cb = blockIdx.y * 8;
    double acc[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) acc[c] = cb + c < nright ? PN(i, ccr + cb + c) : 0.0;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const int cs = cb + c < nright ? cb + c : 0;
#pragma unroll
        for (int j = 0; j < SPW; ++j)
            if (j < w) acc[c] = fma(-l[j], s_u[j][cs], acc[c]);
    }
#pragma unroll
    for (int c = 0; c < 8; ++c)
        if (cb + c < nright) PN(i, ccr + cb + c) = acc[c];
}

// apply the panel's row swaps to every column outside the panel of G and to the columns [0, j0)
// of X.  X is kept in PIVOT ORDER of its columns (stored column c belongs to the original row that
// sits in position c now): a swap of rows r and p then exchanges their entries left of the diagonal
// only and the unit diagonal stays where it is -- what LAPACK does with the L factor itself -- so
// that X stays unit lower triangular all through the forward substitution (k_ref_colperm puts
// the columns back at the end).  one thread per column.
//
// Two more workgroups ride along when the panel is full (tri != nullptr): one wave each inverts the
// panel's diagonal blocks -- L11 (unit lower) and U11 -- column by column in registers, into
// tri[0..4096) and tri[4096..8192) (row-major 64 x 64).  The block substitutions X1 <- L11^-1 X1,
// U12 <- L11^-1 A12 and X1 <- U11^-1 X1 then are products on the matrix cores (k_ref_tri_apply):
// as substitutions, one thread per column, they were 64 dependent steps of LDS-read-then-FMA with
// one wave per CU -- 51-60 us per launch, 20 ms of a refactorisation at k = 8192.  The inversions
// are the same chains, but two waves of them per panel, beside the swaps instead of after them.
__global__ __launch_bounds__(256) void k_ref_swap(int k, int j0, int nbw, double *__restrict__ G,
                                                  double *__restrict__ X, long long ldg,
                                                  const int *__restrict__ piv, double *__restrict__ tri)
{
    const int nswap = (k + j0 + 255) / 256;
    if ((int)blockIdx.x >= nswap) { // (only launched with tri != nullptr and nbw == NB)
        __shared__ __attribute__((aligned(16))) double s_t[NB][NB];
        const bool upper = (int)blockIdx.x == nswap + 1;
        // s_t[j][i] = M[i][j]: column j of the triangle contiguous
        for (int e = threadIdx.x; e < NB * NB; e += blockDim.x)
            s_t[e % NB][e / NB] = G[(long long)(j0 + e / NB) * ldg + j0 + e % NB];
        __syncthreads();
        if (threadIdx.x >= NB) return;
        const int c = threadIdx.x; // column c of the inverse: M y = e_c
        double y[NB];
#pragma unroll
        for (int i = 0; i < NB; ++i) y[i] = i == c ? 1.0 : 0.0;
        if (!upper) {
#pragma unroll
            for (int j = 0; j < NB - 1; ++j) {
                const double yj = y[j];
#pragma unroll
                for (int i = j + 1; i < NB; ++i) y[i] = fma(-s_t[j][i], yj, y[i]);
            }
        } else {
#pragma unroll
            for (int j = NB - 1; j >= 0; --j) {
                const double yj = y[j] / s_t[j][j];
                y[j] = yj;
#pragma unroll
                for (int i = 0; i < j; ++i) y[i] = fma(-s_t[j][i], yj, y[i]);
            }
        }
        double *out = tri + (upper ? NB * NB : 0);
#pragma unroll
        for (int i = 0; i < NB; ++i) out[i * NB + c] = y[i];
        return;
    }
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= k + j0) return;
    double *Mx = c < k ? G : X;
    const int col = c < k ? c : c - k;
    if (c < k && col >= j0 && col < j0 + nbw) return; // the panel swapped itself
    for (int jj = 0; jj < nbw; ++jj) {
        const int r = j0 + jj, p = piv[r];
        if (p != r) {
            const double a = Mx[(long long)r * ldg + col], b = Mx[(long long)p * ldg + col];
            Mx[(long long)r * ldg + col] = b;
            Mx[(long long)p * ldg + col] = a;
        }
    }
}

// The substitutions of a NARROW last panel (fewer than 64 columns; full panels go through
// k_ref_tri_apply): T[j0..j0+nbw, cbeg..cend) <- L11^-1 T[...]  (unit lower L11 = G[j0.., j0..]);
// thread per column
__global__ __launch_bounds__(256) void k_ref_trsm_l(int j0, int nbw, const double *__restrict__ G,
                                                    long long ldg, double *__restrict__ T,
                                                    long long ldt, int cbeg, int cend)
{
    // s_lt[j][i] = L[i][j]
    __shared__ double s_lt[NB][NB];
    for (int e = threadIdx.x; e < nbw * nbw; e += blockDim.x)
        s_lt[e % nbw][e / nbw] = G[(long long)(j0 + e / nbw) * ldg + j0 + e % nbw];
    __syncthreads();
    const int c = cbeg + blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= cend) return;
    double y[NB];
    for (int i = 0; i < nbw; ++i) y[i] = T[(long long)(j0 + i) * ldt + c];
    for (int i = 1; i < nbw; ++i) {
        double acc = y[i];
        for (int j = 0; j < i; ++j) acc = fma(-s_lt[j][i], y[j], acc);
        y[i] = acc;
    }
    for (int i = 0; i < nbw; ++i) T[(long long)(j0 + i) * ldt + c] = y[i];
}

// X[j0..j0+nbw, :) <- U11^-1 X[...]  (upper U11 with diagonal); thread per column
__global__ __launch_bounds__(256) void k_ref_trsm_u(int j0, int nbw, const double *__restrict__ G,
                                                    long long ldg, double *__restrict__ X,
                                                    long long ldx, int ncols)
{
    // s_ut[j][i] = U[i][j]
    __shared__ double s_ut[NB][NB];
    for (int e = threadIdx.x; e < nbw * nbw; e += blockDim.x)
        s_ut[e % nbw][e / nbw] = G[(long long)(j0 + e / nbw) * ldg + j0 + e % nbw];
    __syncthreads();
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= ncols) return;
    double y[NB];
    for (int i = 0; i < nbw; ++i) y[i] = X[(long long)(j0 + i) * ldx + c];
    for (int j = nbw - 1; j >= 0; --j) {
        const double yj = y[j] / s_ut[j][j];
        y[j] = yj;
        for (int i = 0; i < j; ++i) y[i] = fma(-s_ut[j][i], yj, y[i]);
    }
    for (int i = 0; i < nbw; ++i) X[(long long)(j0 + i) * ldx + c] = y[i];
}

// T[row0 .. row0 + 64, cbeg .. cend) <- Minv * T[...] in place, Minv a 64 x 64 block inverse
// (k_ref_swap), on the matrix cores.  A workgroup owns 64 columns: their 64 x 64 tile of T goes into
// LDS whole before anything is written back, so the product may overwrite its own operand.  Lane
// maps as in k_ref_gemm.  grid ceil((cend - cbeg) / 64)
__global__ __launch_bounds__(256) void k_ref_tri_apply(const double *__restrict__ Minv,
                                                       double *__restrict__ T, long long ldt, int row0,
                                                       int cbeg, int cend)
{
    __shared__ __attribute__((aligned(16))) double s_b[NB][GLD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c0 = cbeg + blockIdx.x * 64;
    const int li = lane & 15, lk = lane >> 4;
    const int b_k = tid >> 5, b_c = 2 * (tid & 31);
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int kk = b_k + 8 * u, c = c0 + b_c;
        const double *p = T + (long long)(row0 + kk) * ldt;
        double2r_t w;
        w.x = c < cend ? p[c] : 0.0;
        w.y = c + 1 < cend ? p[c + 1] : 0.0;
        *reinterpret_cast<double2r_t *>(&s_b[kk][b_c]) = w;
    }
    double av[16];
    const double *ap = Minv + (16 * wave + li) * NB;
#pragma unroll
    for (int s = 0; s < 16; ++s) av[s] = ap[4 * s + lk];
    __syncthreads();
    double4_t acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = double4_t{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int s = 0; s < 16; ++s) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
            acc[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[s], s_b[4 * s + lk][16 * j + li], acc[j], 0, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int row = row0 + 16 * wave + lk + 4 * g, col = c0 + 16 * j + li;
            if (col < cend) T[(long long)row * ldt + col] = acc[j][g];
        }
    }
}

// orig[c] = the original row whose unit vector stored column c of X started as: the row swaps of
// the factorisation replayed on the identity list.  One thread: k dependent steps, in LDS while k
// ints fit (use_lds), in global memory beyond.
__global__ __launch_bounds__(256) void k_ref_perm(int k, const int *__restrict__ piv,
                                                  int *__restrict__ orig, int use_lds)
{
    extern __shared__ int s_o[];
    if (use_lds) {
        for (int i = threadIdx.x; i < k; i += blockDim.x) s_o[i] = i;
        __syncthreads();
        if (threadIdx.x == 0) {
            int pn = piv[0];
            for (int c = 0; c < k; ++c) {
                const int p = pn;
                if (c + 1 < k) pn = piv[c + 1]; // (the next pivot is on its way during this step)
                const int a = s_o[c], b = s_o[p];
                s_o[c] = b;
                s_o[p] = a;
            }
        }
        __syncthreads();
        for (int i = threadIdx.x; i < k; i += blockDim.x) orig[i] = s_o[i];
    } else {
        for (int i = threadIdx.x; i < k; i += blockDim.x) orig[i] = i;
        __threadfence();
        __syncthreads();
        if (threadIdx.x == 0) {
            volatile int *o = orig;
            for (int c = 0; c < k; ++c) {
                const int p = piv[c];
                const int a = o[c], b = o[p];
                o[c] = b;
                o[p] = a;
            }
        }
    }
}

// Y[r][orig[c]] = X[r][c]: the columns of the finished inverse back in their own order.
// grid (ceil(k / 256), k)
__global__ __launch_bounds__(256) void k_ref_colperm(int k, const double *__restrict__ X,
                                                     const int *__restrict__ orig,
                                                     double *__restrict__ Y, long long ld)
{
    const int r = blockIdx.y, c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < k) Y[(long long)r * ld + orig[c]] = X[(long long)r * ld + c];
}

// Binv0[spos[b]][a] = X[b][a]   grid (ceil(k/256), k)
__global__ __launch_bounds__(256) void k_ref_scatter(int k, const double *__restrict__ X,
                                                     long long ldx, const int *__restrict__ spos,
                                                     double *__restrict__ binv, long long ldb,
                                                     int row0, int row1)
{
    const int b = blockIdx.y;
    const int a = blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= k) return;
    const int p = spos[b]; // (a row-sharded basis side keeps the rows [row0, row1) of Binv0 only)
    if (p >= row0 && p < row1) binv[(long long)p * ldb + a] = X[(long long)b * ldx + a];
}

// As[i][b] = A[lrow[i], column of the b-th structural basic]   grid (ceil(k/256), nl)
__global__ __launch_bounds__(256) void k_ref_gather_slack(int k, const double *__restrict__ A,
                                                          long long lda, int col0,
                                                          const int *__restrict__ lrow,
                                                          const int *__restrict__ scode,
                                                          double *__restrict__ As, long long ldas,
                                                          int own0, int own1)
{
    const int i = blockIdx.y;
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= k) return;
    const int code = scode[b];
    const bool have = own1 <= own0 || (code >= own0 && code < own1);
    As[(long long)i * ldas + b] = have ? A[(long long)(code - col0) * lda + lrow[i]] : 0.0;
}

__global__ void k_ref_done(DzgCtl *ctl, const int *__restrict__ singular)
{
    ctl->neta = 0;
    // the health monitor restarts with the fresh inverse (the host keeps the lifetime maximum
    // for reporting): a drift that was shed must not trigger a refactorisation per batch
    ctl->max_pivot_err = 0.0;
    if (*singular) ctl->status = DZG_SINGULAR;
}

// ---------------------------------------------------------------------------------
// host sequence.  k and nl are read back by the caller (one sync per refactor).
// ---------------------------------------------------------------------------------
void dzg_launch_refactor_lists(const DzgDev &d, int *spos, int *scode, int *lpos, int *lrow,
                               int *counts, hipStream_t st)
{
    hipLaunchKernelGGL(k_ref_lists, dim3(1), dim3(1024), 0, st, d.ctl, d.m, d.basis, d.var_col, spos,
                       scode, lpos, lrow, counts);
}

static void gemm_sub(int M, int N, int K, const double *A, long long lda, const double *B,
                     long long ldb, double *C, long long ldc, hipStream_t st)
{
    if (M <= 0 || N <= 0 || K <= 0) return;
    // B through LDS (k_ref_gemm_lds) at every shape: the skinny products of the panel pairs too (one
    // trip for the B panel and four for A, where the strip kernel makes sixteen dependent ones);
    // DZG_REF_GEMM_STRIPS keeps the strip kernel for comparison
    if (!std::getenv("DZG_REF_GEMM_STRIPS")) {
        hipLaunchKernelGGL((k_ref_gemm_lds<false>), dim3((N + 63) / 64, (M + 63) / 64), dim3(256), 0, st, M,
                           N, K, A, lda, B, ldb, C, ldc, (const int *)nullptr);
        return;
    }
    hipLaunchKernelGGL((k_ref_gemm<false>), dim3((N + 63) / 64, (M + 63) / 64), dim3(256), 0, st, M, N,
                       K, A, lda, B, ldb, C, ldc, (const int *)nullptr);
}

// The refactorisation in three stages.  A rank of a PARTITIONED column-sharded solve holds its own
// columns only: stage A gathers what it has of G = A[R, S] (zeros for the other ranks' columns),
// stage B factorises and gathers what it has of the basic-slack rows A[L, S]; between the stages
// the host sums the two blocks over the ranks (engine.hip: RCCL all-reduce, or a copy kernel in
// the lockstep harness) -- a column has exactly one owner, so the sum is exact.  Everything else
// is the same computation on every rank.  One GPU / replicated matrix: A, B, C back to back.
static inline void own_range(const DzgDev &d, int &own0, int &own1)
{
    own0 = own1 = 0; // no mask
    if (d.world > 1 && !d.repl) {
        own0 = d.col0;
        own1 = d.col1;
        if (own1 <= own0) { own0 = -2; own1 = -1; } // (a rank without columns: no code matches)
    }
}

void dzg_launch_refactor_a(const DzgDev &d, int k, double *G, double *X, long long ldg, int *scode,
                           int *singular, hipStream_t st)
{
    hipMemsetAsync(singular, 0, sizeof(int), st);
    if (k <= 0) return;
    int own0, own1;
    own_range(d, own0, own1);
    if (d.csc) {
        hipLaunchKernelGGL(k_ref_identity, dim3((k + 255) / 256, k), dim3(256), 0, st, k, G, X, ldg);
        hipLaunchKernelGGL(k_ref_scatter_csc, dim3(k), dim3(256), 0, st, k, d.cptr, d.ridx, d.cval,
                           d.col0, scode, d.dslot, G, ldg, own0, own1);
    } else {
        hipLaunchKernelGGL(k_ref_gather, dim3((k + 255) / 256, k), dim3(256), 0, st, k, d.A, d.lda,
                           d.col0, d.drow, scode, G, X, ldg, own0, own1);
    }
}

// returns the block that stage C multiplies (nl x k, leading dimension ldg; to be summed over the
// ranks of a partitioned solve first), nullptr when there is none
double *dzg_launch_refactor_b(const DzgDev &d, int k, int nl, double *G, double *X, double *Pn, double *Tri,
                              long long ldg,
                              int *piv, int *spos, int *scode, int *lpos, int *lrow, int *lslot,
                              int *singular, hipStream_t st)
{
    double *slack_block = nullptr;
    if (k > 0) {
        // ---- LU of G with the forward substitution of X riding along.
        // Panels are 64 columns wide, but the rank-64 updates they would feed the trailing matrix
        // read and write every element of it (and of X) per 64 columns: two panels a, b are
        // factorised before the trailing matrix sees either -- b needs a's update on its own 64
        // columns (one skinny GEMM) and the row block of b right of it (another) -- and then one
        // rank-128 update does the work of two: half the passes over the trailing matrix.
        const long long ldp = ldg; // the panel's compact column-major copy: >= k rows per column
        auto factor_panel = [&](int j0, int nbw) -> double * {
            hipLaunchKernelGGL((k_ref_panel_copy<false>), dim3((k - j0 + NB - 1) / NB), dim3(256), 0, st, k,
                               j0, nbw, G, ldg, Pn, ldp);
            // sub-panels: 8 columns while the active rows fit 4 per thread, 4 columns up to 8 rows
            // per thread (both all-register), the global-memory kernel beyond 8192 active rows
            for (int c0 = j0; c0 < j0 + nbw;) {
                const int rows = k - c0;
                const int cap = rows <= 1024 * 4 ? 8 : (rows <= 1024 * 8 ? 4 : SPW);
                const int w = (j0 + nbw - c0) < cap ? (j0 + nbw - c0) : cap;
                if (rows <= 1024 * 4)
                    hipLaunchKernelGGL((k_ref_subpanel_reg<4, 8>), dim3(1), dim3(1024), 0, st, k, j0,
                                       nbw, c0, w, Pn, ldp, piv, singular);
                else if (rows <= 1024 * 8)
                    hipLaunchKernelGGL((k_ref_subpanel_reg<8, 4>), dim3(1), dim3(1024), 0, st, k, j0,
                                       nbw, c0, w, Pn, ldp, piv, singular);
                else
                    hipLaunchKernelGGL(k_ref_subpanel, dim3(1), dim3(1024), 0, st, k, j0, nbw, c0, w,
                                       Pn, ldp, piv, singular);
                const int below = k - c0 - w;
                if (below > 0 && c0 + w < j0 + nbw)
                    hipLaunchKernelGGL(k_ref_subupdate,
                                       dim3((below + 255) / 256, (j0 + nbw - c0 - w + 7) / 8), dim3(256), 0,
                                       st, k, j0, nbw, c0, w, Pn, ldp);
                c0 += w;
            }
            hipLaunchKernelGGL((k_ref_panel_copy<true>), dim3((k - j0 + NB - 1) / NB), dim3(256), 0, st, k,
                               j0, nbw, G, ldg, Pn, ldp);
            // full panels: the inverses of L11 and U11 are formed beside the swaps (two more
            // workgroups) and applied as products; a narrower last panel substitutes
            double *tri = nbw == NB ? Tri + (long long)(j0 / NB) * 2 * NB * NB : nullptr;
            hipLaunchKernelGGL(k_ref_swap, dim3((k + j0 + 255) / 256 + (tri ? 2 : 0)), dim3(256), 0, st, k,
                               j0, nbw, G, X, ldg, piv, tri);
            return tri;
        };
        // T[j0 .. j0 + nbw, cbeg .. cend) <- L11^-1 T[...]
        auto solve_l = [&](int j0, int nbw, const double *tri, double *T, int cbeg, int cend) {
            if (cend <= cbeg) return;
            if (tri)
                hipLaunchKernelGGL(k_ref_tri_apply, dim3((cend - cbeg + 63) / 64), dim3(256), 0, st, tri, T,
                                   ldg, j0, cbeg, cend);
            else
                hipLaunchKernelGGL(k_ref_trsm_l, dim3((cend - cbeg + 63) / 64), dim3(64), 0, st, j0,
                                   nbw, G, ldg, T, ldg, cbeg, cend);
        };
        auto at = [&](double *M, int r, int c) { return M + (long long)r * ldg + c; };
        for (int g0 = 0; g0 < k; g0 += 2 * NB) {
            const int gw = (k - g0) < 2 * NB ? (k - g0) : 2 * NB;
            const int na = gw < NB ? gw : NB, nb = gw - na; // panels a = [g0, ja), b = [ja, g0 + gw)
            const int ja = g0 + na, je = g0 + gw;
            const double *tri_a = factor_panel(g0, na);
            solve_l(g0, na, tri_a, G, ja, k); // U12 of a, all columns right of it
            // X stays unit lower triangular in pivot order of its columns (k_ref_swap): the row
            // block of a panel holds something in the columns up to its own diagonal block only
            solve_l(g0, na, tri_a, X, 0, ja);
            if (nb > 0) {
                // b's columns, rows below a:  -= L21a U12a
                gemm_sub(k - ja, nb, na, at(G, ja, g0), ldg, at(G, g0, ja), ldg, at(G, ja, ja), ldg, st);
                const double *tri_b = factor_panel(ja, nb);
                // b's row block right of the pair, and in X:  -= L(b,a) (a's row block)
                gemm_sub(nb, k - je, na, at(G, ja, g0), ldg, at(G, g0, je), ldg, at(G, ja, je), ldg, st);
                gemm_sub(nb, ja, na, at(G, ja, g0), ldg, at(X, g0, 0), ldg, at(X, ja, 0), ldg, st);
                solve_l(ja, nb, tri_b, G, je, k);
                solve_l(ja, nb, tri_b, X, 0, je);
            }
            const int rest = k - je;
            if (rest > 0) {
                // A22 -= [L21a L21b] [U12a; U12b],  X2 -= [L21a L21b] [X1a; X1b] (columns [0, je))
                gemm_sub(rest, rest, gw, at(G, je, g0), ldg, at(G, g0, je), ldg, at(G, je, je), ldg, st);
                gemm_sub(rest, je, gw, at(G, je, g0), ldg, at(X, g0, 0), ldg, at(X, je, 0), ldg, st);
            }
        }
        // ---- backward substitution with U, the same pairs from the last to the first
        auto solve_u = [&](int j0, int nbw) {
            if (nbw == NB)
                hipLaunchKernelGGL(k_ref_tri_apply, dim3((k + 63) / 64), dim3(256), 0, st,
                                   Tri + (long long)(j0 / NB) * 2 * NB * NB + NB * NB, X, ldg, j0, 0, k);
            else
                hipLaunchKernelGGL(k_ref_trsm_u, dim3((k + 63) / 64), dim3(64), 0, st, j0, nbw, G, ldg,
                                   X, ldg, k);
        };
        for (int g0 = ((k - 1) / (2 * NB)) * (2 * NB); g0 >= 0; g0 -= 2 * NB) {
            const int gw = (k - g0) < 2 * NB ? (k - g0) : 2 * NB;
            const int na = gw < NB ? gw : NB, nb = gw - na;
            const int ja = g0 + na;
            if (nb > 0) {
                solve_u(ja, nb);
                // X1a -= U(a,b) X1b
                gemm_sub(na, k, nb, at(G, g0, ja), ldg, at(X, ja, 0), ldg, at(X, g0, 0), ldg, st);
            }
            solve_u(g0, na);
            if (g0 > 0) // X[0..g0) -= U[0..g0, pair] * X1
                gemm_sub(g0, k, gw, at(G, 0, g0), ldg, at(X, g0, 0), ldg, X, ldg, st);
        }
        // ---- the columns of X back from pivot order: G^-1 = U^-1 L^-1 P.  G's factors are done
        // with: the permuted inverse goes there and X becomes the free panel
        {
            static const size_t lds_cap = 150 * 1024; // of the CU's 160 KB
            static const hipError_t attr = hipFuncSetAttribute(
                reinterpret_cast<const void *>(k_ref_perm), hipFuncAttributeMaxDynamicSharedMemorySize,
                (int)lds_cap);
            const size_t need = sizeof(int) * (size_t)k;
            const int use_lds = attr == hipSuccess && need <= lds_cap;
            hipLaunchKernelGGL(k_ref_perm, dim3(1), dim3(256), use_lds ? need : 0, st, k, piv, lslot,
                               use_lds);
            hipLaunchKernelGGL(k_ref_colperm, dim3((k + 255) / 256, k), dim3(256), 0, st, k, X, lslot, G,
                               ldg);
        }
        const double *Xf = G; // the finished inverse
        double *Wk = X;       // free
        if (d.spb) {
            // sparse-basis mode keeps the k x k block only (k_sparse.hip); rows in position order
            dzg_launch_sp_ref_copy(d, k, Xf, ldg, st);
            return nullptr;
        }
        // ---- Binv0 rows of the structural positions
        hipLaunchKernelGGL(k_ref_scatter, dim3((k + 255) / 256, k), dim3(256), 0, st, k, Xf, ldg, spos,
                           d.binv, d.ldb, d.rs ? d.rs_r0 : 0, d.rs ? d.rs_r1 : d.m);
        // ---- Binv0 rows of the basic slacks: -A[r', S] * X   (A[r', S] goes into the free panel)
        if (nl > 0) {
            int own0, own1;
            own_range(d, own0, own1);
            if (d.csc) {
                hipMemsetAsync(Wk, 0, sizeof(double) * (size_t)nl * (size_t)ldg, st);
                hipLaunchKernelGGL(k_ref_lslot, dim3((d.m + 255) / 256), dim3(256), 0, st, d.m, nl, lrow,
                                   lslot);
                hipLaunchKernelGGL(k_ref_lslot_fill, dim3((nl + 255) / 256), dim3(256), 0, st, nl, lrow,
                                   lslot);
                hipLaunchKernelGGL(k_ref_scatter_csc, dim3(k), dim3(256), 0, st, k, d.cptr, d.ridx,
                                   d.cval, d.col0, scode, lslot, Wk, ldg, own0, own1);
            } else {
                hipLaunchKernelGGL(k_ref_gather_slack, dim3((k + 255) / 256, nl), dim3(256), 0, st, k,
                                   d.A, d.lda, d.col0, lrow, scode, Wk, ldg, own0, own1);
            }
            slack_block = Wk;
        }
    }
    return slack_block;
}

void dzg_launch_refactor_c(const DzgDev &d, int k, int nl, double *G, double *X, long long ldg,
                           int *lpos, int *singular, hipStream_t st)
{
    if (k > 0 && nl > 0 && !d.spb) {
        const double *Xf = G; // the finished inverse (stage B)
        double *Wk = X;       // A[L, S]
        // (lpos ascends: a row-sharded rank's rows are a contiguous run of the list, the workgroups
        // outside it return at once)
        const int row0 = d.rs ? d.rs_r0 : 0, row1 = d.rs ? d.rs_r1 : 0;
        if (!std::getenv("DZG_REF_GEMM_STRIPS"))
            hipLaunchKernelGGL((k_ref_gemm_lds<true>), dim3((k + 63) / 64, (nl + 63) / 64), dim3(256), 0, st,
                               nl, k, k, Wk, ldg, Xf, ldg, d.binv, d.ldb, (const int *)lpos, row0, row1);
        else
            hipLaunchKernelGGL((k_ref_gemm<true>), dim3((k + 63) / 64, (nl + 63) / 64), dim3(256), 0, st,
                               nl, k, k, Wk, ldg, Xf, ldg, d.binv, d.ldb, (const int *)lpos, row0, row1);
    }
    hipLaunchKernelGGL(k_ref_done, dim3(1), dim3(1), 0, st, d.ctl, singular);
}

// Lockstep harness (all ranks of a partitioned solve in one process on one GPU): every rank's
// block := the sum of all ranks' blocks, element by element, ranks in order.  bufs[r] = rank r's
// block (device array of device pointers), `count` doubles each.
__global__ __launch_bounds__(256) void k_lockstep_sum(double *const *bufs, int world, long long count)
{
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < count;
         i += (long long)gridDim.x * blockDim.x) {
        double s = 0.0;
        for (int r = 0; r < world; ++r) s += bufs[r][i];
        for (int r = 0; r < world; ++r) bufs[r][i] = s;
    }
}

void dzg_launch_lockstep_sum(double *const *bufs, int world, long long count, hipStream_t st)
{
    if (count <= 0) return;
    long long blocks = (count + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(k_lockstep_sum, dim3((unsigned)blocks), dim3(256), 0, st, bufs, world, count);
}
