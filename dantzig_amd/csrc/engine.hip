// engine.hip -- host orchestration of the simplex iteration and the Level-1 C ABI.
//
// Replaces Simplex::solve (src/simplex.rs:332-343).  The reference recurses once per
// pivot on one CPU thread; here the host only enqueues kernels on one HIP stream.  All
// decisions live in the device control block (DzgCtl), so in FAST numerics the host
// enqueues `poll_interval` iterations back to back and reads the status word once per
// batch.  STRICT numerics needs O(m) launches per iteration anyway and reads the step kind
// after the status kernel to launch only the solves that iteration uses.
#include <dlfcn.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <string>
#include <utility>
#include <algorithm>
#include <vector>

#include "common.h"

static thread_local std::string g_err;

static int fail(int code, const std::string &msg)
{
    g_err = msg;
    return code;
}

#define HIP_OK(expr)                                                                      \
    do {                                                                                  \
        hipError_t e_ = (expr);                                                           \
        if (e_ != hipSuccess)                                                             \
            return fail(DZG_E_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

extern "C" int dzg_abi_version(void) { return DZG_ABI_VERSION; }

extern "C" const char *dzg_last_error(void) { return g_err.c_str(); }

extern "C" const char *dzg_status_str(int s)
{
    switch (s) {
    case DZG_OPTIMAL: return "optimal";
    case DZG_UNBOUNDED: return "unbounded";
    case DZG_INFEASIBLE: return "infeasible";
    case DZG_ITER_LIMIT: return "iter_limit";
    case DZG_SINGULAR: return "singular";
    case DZG_PANIC: return "panic";
    case DZG_RUNNING: return "running";
    case DZG_NEAR_TIE: return "near_tie";
    case DZG_E_DEVICE: return "device_error";
    case DZG_E_ARG: return "bad_argument";
    case DZG_E_NOMEM: return "out_of_memory";
    default: return "unknown";
    }
}

extern "C" int dzg_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

extern "C" void dzg_opts_default(dzg_opts *o)
{
    std::memset(o, 0, sizeof(*o));
    o->numerics = DZG_NUMERICS_AUTO;
    o->price_kernel = DZG_PRICE_AUTO;
    o->device = 0;
    o->auto_strict_rows = 192;
    o->max_iter = 10000000;
    o->epsilon = 1e-12;
    o->log_capacity = -1;
    o->poll_interval = 32;
    o->profile = 0;
    o->world = 1;
    o->near_tie_action = DZG_NEAR_TIE_COUNT;
    o->auto_restart_rows = DZG_AUTO_STRICT_RESTART_ROWS;
    o->tie_tol = 1e-11;
}

struct dzg_solver {
    DzgDev d{};
    dzg_opts opts{};
    int numerics = DZG_NUMERICS_FAST;
    hipStream_t st = nullptr;
    bool own_stream = true;
    DzgCtl *h_ctl = nullptr; // pinned
    std::vector<void *> allocs;
    std::vector<double> c_host;
    double constant = 0.0;
    double solve_ms = 0.0;
    int since_flush = 0; // iterations enqueued since the eta file was last folded into Binv0
    // refactorisation workspace (FAST, opts.refactor_interval != 0)
    double *rfG = nullptr, *rfX = nullptr, *rfPn = nullptr, *rfTri = nullptr; // (rfPn: one 64-column panel,
                                                                 // column-major; rfTri: L11^-1, U11^-1 per panel)
    long long rf_ld = 0, rf_rows = 0; // leading dimension and rows of rfG / rfX as allocated
    int *rf_piv = nullptr, *rf_spos = nullptr, *rf_scode = nullptr, *rf_lpos = nullptr,
        *rf_lrow = nullptr, *rf_counts = nullptr, *rf_lslot = nullptr;
    long long since_refactor = 0;
    bool pending_refactor = false; // a warm start whose basic columns are not all on this device yet
                                   // (partitioned shards; replicate_matrix before the last upload):
                                   // the starting basis is factorised by the first run instead
    bool in_lockstep = false;      // inside dzg_shard_run_lockstep (the harness sums over the ranks)
    // STRICT: the O(m) launches of one basis solve, captured once and replayed (hipGraph)
    hipGraphExec_t g_solve[2] = {nullptr, nullptr}; // [0] B dx = a_j, [1] B^T v = e_p
    bool graphs_tried = false;
    int prof_slot = -1; // phase path: event slot of the iteration being enqueued (-1: none)
    int64_t refactors = 0;
    double *A_alloc = nullptr;  // device matrix as allocated (d.A points at column col0 inside it)
    long long cols_present = 0; // structural columns uploaded so far (replicate_matrix + a_is_block)
    std::vector<std::pair<int64_t, int64_t>> col_blocks; // ... as disjoint intervals [begin, end)
    double drift_trigger = 1e-9; // FAST health: disagreement of the two pivot elements that
                                 // triggers a refactorisation
    double max_err_life = 0.0; // largest ctl->max_pivot_err ever read (the device value restarts
                               // at every refactorisation)
    double state_drift = 0.0;  // carried x_B / z_N against the fresh inverse, at the last refactorisation
    // k_drift.hip: the data the state is recomputed from (solves that start on the slack basis of a
    // dense matrix on one GPU; nullptr otherwise) and scratch
    double *dr_b0 = nullptr, *dr_xb0 = nullptr, *dr_c = nullptr, *dr_agb = nullptr, *dr_agx = nullptr,
           *dr_part = nullptr, *dr_y = nullptr, *dr_dzy = nullptr, *dr_out = nullptr;
    // FAST, dense, one GPU: the three-launch chain (k_chain.hip)
    unsigned long long *chain_bar = nullptr; // barrier counters (cleared only by chain_recover)
    unsigned long long *chain_dbg = nullptr; // DZG_CHAIN_DEBUG=1: phase clocks of workgroup 0
    int chain_grid = 0;                      // one workgroup per CU
    long long chain_kcap = DZG_CHAIN_AGCAP;  // compact width beyond which a batch runs as seven
                                             // launches (DZG_CHAIN_KCAP lowers it: tests)
    bool batch_chain = false;                // the batch in flight runs the chain
    bool chain_fold = true;                  // k_chain_post finishes the row-wise pricing pass itself
                                             // (DZG_CHAIN_NO_FOLD=1 at creation: the separate launch)
    int64_t chain_fallbacks = 0;             // barrier failures recovered from (chain_recover)
    long long chain_retry_iter = 0;          // the chain stays off until this pivot count
    // column sharding
    void *comm = nullptr;                  // ncclComm_t
    double *xsend = nullptr, *xrecv1 = nullptr, *xrecv2 = nullptr;
    // opts.shard_rows (k_rowshard.hip)
    long long xstride_max = 0;             // largest record (room for a row of m entries); d.xstride is
                                           // what the batch in flight sends
    const double *rs_recv1 = nullptr;      // records of exchange 1 (phase 3 of a dual step reads the
                                           // leaving row from them)
    int rs_slice = 0;                      // rows per rank
    double *rs_gsend = nullptr, *rs_grecv = nullptr; // x / xbar slices on their way to every rank
    // profiling
    std::vector<hipEvent_t> ev; // [batch slot][class][2]
    double kernel_ms[DZG_K_COUNT] = {};
    int64_t kernel_launches[DZG_K_COUNT] = {};
};

template <typename T> static int dev_alloc(dzg_solver *s, T **p, size_t count)
{
    void *q = nullptr;
    size_t bytes = sizeof(T) * (count ? count : 1);
    if (hipMalloc(&q, bytes) != hipSuccess)
        return fail(DZG_E_NOMEM, "hipMalloc of " + std::to_string(bytes) + " bytes failed");
    s->allocs.push_back(q);
    *p = static_cast<T *>(q);
    return 0;
}

#define TRY(expr)            \
    do {                     \
        int rc_ = (expr);    \
        if (rc_ != 0) return rc_; \
    } while (0)

static int validate(const dzg_lp *lp, std::string &why)
{
    if (!lp) { why = "lp is NULL"; return 0; }
    if (lp->m < 0 || lp->n < lp->m || lp->n_struct < 0 || lp->n_struct > lp->n) { why = "bad sizes"; return 0; }
    if (lp->n >= (1ll << 31) - 64 || lp->m >= (1ll << 31) - 64) { why = "index range"; return 0; }
    if (lp->n_struct > 0 && lp->a && lp->lda < lp->m) { why = "lda"; return 0; }
    if (lp->n_struct > 0 && !lp->a) {
        if (!lp->col_ptr || lp->col_ptr[0] != 0) { why = "neither a nor col_ptr"; return 0; }
        for (int64_t j = 0; j < lp->n_struct; ++j) {
            if (lp->col_ptr[j + 1] < lp->col_ptr[j]) { why = "col_ptr not monotone"; return 0; }
            for (int64_t e = lp->col_ptr[j]; e < lp->col_ptr[j + 1]; ++e) {
                if (!lp->row_idx || !lp->val || lp->row_idx[e] < 0 || lp->row_idx[e] >= lp->m ||
                    (e > lp->col_ptr[j] && lp->row_idx[e] <= lp->row_idx[e - 1])) {
                    why = "row_idx must ascend strictly inside a column";
                    return 0;
                }
            }
        }
    }
    const int64_t q = lp->n - lp->m;
    if ((lp->m > 0 && (!lp->basis || !lp->x)) || (q > 0 && (!lp->nonbasis || !lp->z)) || (lp->n > 0 && !lp->c)) { why = "NULL state vector"; return 0; }
    if (!lp->var_col && lp->n != lp->n_struct + lp->m) { why = "var_col == NULL needs n == n_struct + m"; return 0; }
    std::vector<char> seen((size_t)lp->n, 0);
    for (int64_t k = 0; k < lp->n; ++k) {
        int64_t v = k < lp->m ? lp->basis[k] : lp->nonbasis[k - lp->m];
        if (v < 0 || v >= lp->n || seen[(size_t)v]) { why = "basis/nonbasis is not a partition of 0..n-1"; return 0; }
        seen[(size_t)v] = 1;
    }
    if (lp->var_col) {
        for (int64_t v = 0; v < lp->n; ++v) {
            int64_t c = lp->var_col[v];
            if (c >= lp->n_struct || c < -lp->m) { why = "var_col out of range"; return 0; }
        }
    }
    return 1;
}

// ---- RCCL, loaded lazily: single-GPU users never need librccl ------------------------
namespace {
struct NcclUniqueId { char internal[128]; };
struct Rccl {
    void *handle = nullptr;
    int (*GetUniqueId)(NcclUniqueId *) = nullptr;
    int (*CommInitRank)(void **, int, NcclUniqueId, int) = nullptr;
    int (*AllGather)(const void *, void *, size_t, int, void *, hipStream_t) = nullptr;
    int (*AllReduce)(const void *, void *, size_t, int, int, void *, hipStream_t) = nullptr;
    int (*CommDestroy)(void *) = nullptr;
    int (*CommCount)(void *, int *) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    bool ok = false;
};
Rccl &rccl()
{
    static Rccl r;
    if (r.handle || r.ok) return r;
    // ROCm's own librccl first, by path: it binds to the same libamdhip64 as this library.  A
    // bare soname could resolve to a copy bundled with a Python package (its own HIP runtime).
    const char *names[] = {"/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so", "librccl.so.1",
                           "librccl.so"};
    for (const char *n : names) {
        r.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL);
        if (r.handle) break;
    }
    if (!r.handle) return r;
    r.GetUniqueId = (int (*)(NcclUniqueId *))dlsym(r.handle, "ncclGetUniqueId");
    r.CommInitRank = (int (*)(void **, int, NcclUniqueId, int))dlsym(r.handle, "ncclCommInitRank");
    r.AllGather = (int (*)(const void *, void *, size_t, int, void *, hipStream_t))dlsym(r.handle, "ncclAllGather");
    r.AllReduce = (int (*)(const void *, void *, size_t, int, int, void *, hipStream_t))dlsym(r.handle, "ncclAllReduce");
    r.CommDestroy = (int (*)(void *))dlsym(r.handle, "ncclCommDestroy");
    r.CommCount = (int (*)(void *, int *))dlsym(r.handle, "ncclCommCount");
    r.GetErrorString = (const char *(*)(int))dlsym(r.handle, "ncclGetErrorString");
    r.ok = r.GetUniqueId && r.CommInitRank && r.AllGather && r.CommDestroy;
    return r;
}
const int kNcclFloat64 = 8;
const int kNcclSum = 0;
} // namespace

static void shard_comm_destroy(dzg_solver *s);
static int shard_buffers(dzg_solver *s);
static int refactor_now(dzg_solver *s);
static int refactor_workspace(dzg_solver *s);
static bool partitioned(const dzg_solver *s);

// DZG_CHAIN_DEBUG=1: where workgroup 0 of the chain kernels spent its time (stderr, at destroy)
static void chain_debug_report(dzg_solver *s)
{
    unsigned long long h[64];
    if (hipMemcpy(h, s->chain_dbg, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess) return;
    static const char *names[4] = {"k_chain_pre  primal", "k_chain_pre  dual  ", "k_chain_post primal",
                                   "k_chain_post dual  "};
    for (int slot = 0; slot < 4; ++slot) {
        const double n = (double)h[16 * slot + 15];
        if (n <= 0) continue;
        std::fprintf(stderr, "[chain] %s x%-8.0f us per stage:", names[slot], n);
        double tot = 0;
        for (int st = 0; st < 15; ++st) {
            if (!h[16 * slot + st]) continue;
            const double us = (double)h[16 * slot + st] / n / 100.0; // 100 MHz clock
            tot += us;
            std::fprintf(stderr, " %d:%.2f", st, us);
        }
        std::fprintf(stderr, "  sum %.2f\n", tot);
    }
}

extern "C" void dzg_solver_destroy(dzg_solver *s)
{
    if (!s) return;
    if (s->chain_dbg) chain_debug_report(s);
    shard_comm_destroy(s);
    for (hipGraphExec_t g : s->g_solve)
        if (g) hipGraphExecDestroy(g);
    for (hipEvent_t e : s->ev) hipEventDestroy(e);
    for (void *p : s->allocs) hipFree(p);
    if (s->h_ctl) hipHostFree(s->h_ctl);
    if (s->st && s->own_stream) hipStreamDestroy(s->st);
    delete s;
}

// columns [c0, c1) of the structural block into their slots of a replicated device matrix;
// -0.0 entries become +0.0 like everywhere else (the reference's CSC drops exact zeros)
static int upload_columns(dzg_solver *s, int64_t c0, int64_t c1, const double *a, int64_t lda)
{
    const DzgDev &d = s->d;
    if (c1 <= c0 || d.m == 0) return 0;
    const int64_t cnt = c1 - c0;
    double *dst = s->A_alloc + (size_t)c0 * (size_t)d.lda;
    bool has_negzero = false;
    for (int64_t j = 0; j < cnt && !has_negzero; ++j)
        for (int64_t i = 0; i < d.m; ++i) {
            const double val = a[j * lda + i];
            if (val == 0.0 && std::signbit(val)) { has_negzero = true; break; }
        }
    if (!has_negzero) {
        HIP_OK(hipMemcpy2DAsync(dst, sizeof(double) * d.lda, a, sizeof(double) * lda,
                                sizeof(double) * d.m, (size_t)cnt, hipMemcpyHostToDevice, s->st));
    } else {
        std::vector<double> tmp((size_t)d.m * (size_t)cnt);
        for (int64_t j = 0; j < cnt; ++j)
            for (int64_t i = 0; i < d.m; ++i) {
                const double val = a[j * lda + i];
                tmp[(size_t)(j * d.m + i)] = (val == 0.0) ? 0.0 : val;
            }
        HIP_OK(hipMemcpy2DAsync(dst, sizeof(double) * d.lda, tmp.data(), sizeof(double) * d.m,
                                sizeof(double) * d.m, (size_t)cnt, hipMemcpyHostToDevice, s->st));
    }
    HIP_OK(hipStreamSynchronize(s->st)); // the caller may reuse its buffer
    s->cols_present += cnt;
    return 0;
}

extern "C" int dzg_solver_upload_columns(dzg_solver *s, int64_t col_begin, int64_t col_end,
                                         const double *a, int64_t lda)
{
    if (!s || !a || !s->d.repl) return fail(DZG_E_ARG, "upload_columns: a solver with replicate_matrix");
    if (col_begin < 0 || col_end > s->d.ns || col_begin > col_end || lda < s->d.m ||
        (col_begin < s->d.col1 && col_end > s->d.col0))
        return fail(DZG_E_ARG, "upload_columns: [col_begin, col_end) must lie outside the rank's own block");
    // every column exactly once: a block handed over twice would leave another one unset (zeros)
    // behind a column count that looks complete
    for (const auto &blk : s->col_blocks)
        if (col_begin < blk.second && col_end > blk.first)
            return fail(DZG_E_ARG, "upload_columns: columns [" + std::to_string(col_begin) + ", " +
                                       std::to_string(col_end) + ") overlap a block that is already present");
    HIP_OK(hipSetDevice(s->opts.device));
    TRY(upload_columns(s, col_begin, col_end, a, lda));
    if (col_end > col_begin) s->col_blocks.emplace_back(col_begin, col_end);
    return 0;
}

extern "C" int dzg_solver_create(const dzg_lp *lp, const dzg_opts *opts_in, dzg_solver **out)
{
    if (!out) return fail(DZG_E_ARG, "out is NULL");
    *out = nullptr;
    std::string why;
    if (!validate(lp, why)) return fail(DZG_E_ARG, "invalid dzg_lp: " + why);
    dzg_opts o;
    if (opts_in) o = *opts_in; else dzg_opts_default(&o);
    if (o.max_iter <= 0) o.max_iter = 10000000;
    if (o.poll_interval <= 0) o.poll_interval = 32;
    if (o.epsilon == 0.0) o.epsilon = 1e-12;
    if (o.auto_strict_rows <= 0) o.auto_strict_rows = 192;
    if (o.tie_tol == 0.0) o.tie_tol = 1e-11;

    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(DZG_E_DEVICE, "no HIP device visible: dantzig_amd has no CPU path");
    if (o.device < 0 || o.device >= ndev) return fail(DZG_E_ARG, "opts.device out of range");
    HIP_OK(hipSetDevice(o.device));

    dzg_solver *s = new dzg_solver();
    s->opts = o;
    struct Guard { dzg_solver *s; ~Guard() { if (s) dzg_solver_destroy(s); } } guard{s};
    if (o.stream) {
        s->st = (hipStream_t)o.stream;
        s->own_stream = false;
    } else {
        HIP_OK(hipStreamCreate(&s->st));
    }
    HIP_OK(hipHostMalloc((void **)&s->h_ctl, sizeof(DzgCtl)));

    DzgDev &d = s->d;
    const int m = (int)lp->m, n = (int)lp->n, ns = (int)lp->n_struct, q = n - m;
    d.m = m; d.n = n; d.ns = ns; d.q = q;
    // 16-B aligned columns; on tall matrices a column stride that is not a multiple of a large
    // power of two spreads the 16 concurrent column streams of a wave over more HBM channels
    // (+1.5-2 % on the pricing pass, profiles/r01_price_microbench_*.txt)
    d.lda = ((long long)m + 15) / 16 * 16 + (m >= 2048 ? 272 : 0);
    if (d.lda == 0) d.lda = 16;
    d.eps = o.epsilon;
    // FTRAN's loads of Binv0's rows (k > 512): nontemporal, eight steps in flight -- 125.4 -> 102.9 us
    // for k_chain_pre at k = 7 700 (profiles/r04_ftran_row_loads_ab.txt); DZG_FTRAN_VARIANT=0: plain loads
    d.ftran_variant = -1;
    if (const char *fv = std::getenv("DZG_FTRAN_VARIANT")) d.ftran_variant = std::atoi(fv);
    {
        const double kk = 384e6 / (8.0 * (double)(lp->m > 0 ? lp->m : 1));
        d.ftran_nt_k = kk > 2e9 ? 2000000000 : (int)kk;
        if (d.ftran_nt_k < 513) d.ftran_nt_k = 513;
    }
    d.fold_k = 0x7fffffff;
    if (const char *fk = std::getenv("DZG_CHAIN_FOLD_K")) d.fold_k = std::atoi(fk);
    d.world = o.world > 1 ? o.world : 1;
    d.rank = d.world > 1 ? o.rank : 0;
    d.col0 = 0;
    d.col1 = ns;
    if (d.world > 1) {
        if (o.rank < 0 || o.rank >= o.world || o.col_begin < 0 || o.col_end > ns ||
            o.col_begin > o.col_end)
            return fail(DZG_E_ARG, "opts.rank / col_begin / col_end");
        d.col0 = (int)o.col_begin;
        d.col1 = (int)o.col_end;
    }
    const int nloc = d.col1 - d.col0; // structural columns this rank prices
    // dense + replicate_matrix: all ns columns are resident, the exchange carries headers only
    d.repl = (d.world > 1 && o.replicate_matrix && lp->a) ? 1 : 0;
    d.xstride = d.repl ? 8 : 8 + ((long long)m + 7) / 8 * 8;
    const double *a_host = !lp->a ? nullptr
                           : (o.a_is_block && d.world > 1) ? lp->a
                                                           : lp->a + (size_t)d.col0 * (size_t)lp->lda;

    d.csc = (ns > 0 && !lp->a) ? 1 : 0;
    // opts.shard_rows: the basis side sharded by rows as well (k_rowshard.hip)
    d.rs = 0;
    d.rs_r0 = 0;
    d.rs_r1 = m;
    d.rs_mcol = 0;
    if (o.shard_rows && d.world > 1) {
        if (d.csc) return fail(DZG_E_ARG, "opts.shard_rows needs a dense matrix");
        if (o.price_kernel != DZG_PRICE_AUTO && o.price_kernel != DZG_PRICE_TREE)
            return fail(DZG_E_ARG, "opts.shard_rows prices with the AUTO / TREE kernels (every rank "
                                   "re-derives dz of the entering column in their summation order)");
        d.rs = 1;
        s->rs_slice = (int)((((long long)m + d.world - 1) / d.world + 15) / 16 * 16);
        if (s->rs_slice < 16) s->rs_slice = 16;
        const long long r0 = (long long)d.rank * s->rs_slice;
        d.rs_r0 = (int)(r0 < m ? r0 : m);
        d.rs_r1 = (int)(r0 + s->rs_slice < m ? r0 + s->rs_slice : m);
        d.rs_mcol = d.repl ? 0 : ((long long)m + 15) / 16 * 16;
        s->xstride_max = DZG_RS_COL + d.rs_mcol + ((long long)m + 2 + 15) / 16 * 16;
        d.xstride = s->xstride_max;
    }
    if (d.csc) {
        // sparse mode: the owned columns stay CSC on the device (12 bytes per nonzero); explicit
        // zeros are dropped like the reference's From<&Matrix> for CscMatrix (src/linalg.rs:261)
        std::vector<long long> cp((size_t)nloc + 1, 0);
        std::vector<int> ri;
        std::vector<double> cv;
        for (int j = 0; j < nloc; ++j) {
            for (int64_t e = lp->col_ptr[d.col0 + j]; e < lp->col_ptr[d.col0 + j + 1]; ++e)
                if (lp->val[e] != 0.0) {
                    ri.push_back((int)lp->row_idx[e]);
                    cv.push_back(lp->val[e]);
                }
            cp[(size_t)j + 1] = (long long)ri.size();
        }
        long long *dcp; int *dri; double *dcv;
        TRY(dev_alloc(s, &dcp, cp.size())); TRY(dev_alloc(s, &dri, ri.size() + 1));
        TRY(dev_alloc(s, &dcv, cv.size() + 1));
        HIP_OK(hipMemcpy(dcp, cp.data(), sizeof(long long) * cp.size(), hipMemcpyHostToDevice));
        if (!ri.empty()) {
            HIP_OK(hipMemcpy(dri, ri.data(), sizeof(int) * ri.size(), hipMemcpyHostToDevice));
            HIP_OK(hipMemcpy(dcv, cv.data(), sizeof(double) * cv.size(), hipMemcpyHostToDevice));
        }
        d.cptr = dcp; d.ridx = dri; d.cval = dcv;
        // One GPU, FAST numerics: the basis lives on its k x k structural block (k_sparse.hip),
        // which needs row access to the matrix as well: a CSR copy, columns ascending in a row.
        const int want = o.numerics == DZG_NUMERICS_AUTO
                             ? (m <= o.auto_strict_rows ? DZG_NUMERICS_STRICT : DZG_NUMERICS_FAST)
                             : o.numerics;
        if (d.world == 1 && want == DZG_NUMERICS_FAST && m <= (1 << 20)) {
            d.spb = 1;
            std::vector<long long> rp((size_t)m + 1, 0);
            for (int r : ri) rp[(size_t)r + 1] += 1;
            for (int r = 0; r < m; ++r) rp[(size_t)r + 1] += rp[(size_t)r];
            std::vector<int> ci(ri.size());
            std::vector<double> rv(ri.size());
            std::vector<long long> fill(rp.begin(), rp.end() - 1);
            for (int j = 0; j < nloc; ++j) // columns ascending => ascending inside every row
                for (long long e = cp[(size_t)j]; e < cp[(size_t)j + 1]; ++e) {
                    const long long at = fill[(size_t)ri[(size_t)e]]++;
                    ci[(size_t)at] = j;
                    rv[(size_t)at] = cv[(size_t)e];
                }
            long long *drp; int *dci; double *drv;
            TRY(dev_alloc(s, &drp, rp.size())); TRY(dev_alloc(s, &dci, ci.size() + 1));
            TRY(dev_alloc(s, &drv, rv.size() + 1));
            HIP_OK(hipMemcpy(drp, rp.data(), sizeof(long long) * rp.size(), hipMemcpyHostToDevice));
            if (!ci.empty()) {
                HIP_OK(hipMemcpy(dci, ci.data(), sizeof(int) * ci.size(), hipMemcpyHostToDevice));
                HIP_OK(hipMemcpy(drv, rv.data(), sizeof(double) * rv.size(), hipMemcpyHostToDevice));
            }
            d.rptr = drp; d.cidx = dci; d.rval = drv;
            TRY(dev_alloc(s, &d.bcnt, (size_t)(m ? m : 1)));
            TRY(dev_alloc(s, &d.bcol, ci.size() + 1));
            TRY(dev_alloc(s, &d.bval, rv.size() + 1));
            // live-entry lists of the columns (k_price_csc_rl); the reference-order kernel and the
            // A/B switch DZG_SP_PRICE_FULL=1 price over every stored entry instead
            const char *full = std::getenv("DZG_SP_PRICE_FULL");
            if (o.price_kernel != DZG_PRICE_SEQ && !(full && full[0] == '1')) {
                TRY(dev_alloc(s, &d.lcnt, (size_t)(ns ? ns : 1)));
                TRY(dev_alloc(s, &d.lent, ri.size() + 1));
                TRY(dev_alloc(s, &d.rl_work, (size_t)DZG_RL_WORK_SLOTS));
                HIP_OK(hipMemsetAsync(d.rl_work, 0, sizeof(unsigned long long) * DZG_RL_WORK_SLOTS, s->st));
            }
        }
    }
    // --- constraint matrix: column-major, zero-padded to lda rows (16-B aligned columns)
    double *A = nullptr;
    const int ndense = d.csc ? 0 : nloc;
    const int nmem = d.csc ? 0 : (d.repl ? ns : nloc);       // columns held in HBM
    const long long mem0 = d.repl ? 0 : d.col0;              // first of them
    double *A_alloc = nullptr;
    TRY(dev_alloc(s, &A_alloc, (size_t)d.lda * (size_t)(nmem ? nmem : 1)));
    HIP_OK(hipMemsetAsync(A_alloc, 0, sizeof(double) * (size_t)d.lda * (size_t)(nmem ? nmem : 1), s->st));
    A = A_alloc + (size_t)(d.col0 - mem0) * (size_t)d.lda;   // column col0: kernels index from it
    s->A_alloc = A_alloc;
    s->cols_present = d.repl ? nloc : ns;
    if (ndense > 0 && m > 0) {
        // the reference's CSC drops exact zeros (src/linalg.rs:261), so -0.0 entries act as +0.0
        bool has_negzero = false;
        for (int64_t j = 0; j < nloc && !has_negzero; ++j)
            for (int64_t i = 0; i < m; ++i) {
                const double val = a_host[j * lp->lda + i];
                if (val == 0.0 && std::signbit(val)) { has_negzero = true; break; }
            }
        if (!has_negzero) {
            HIP_OK(hipMemcpy2DAsync(A, sizeof(double) * d.lda, a_host, sizeof(double) * lp->lda,
                                    sizeof(double) * m, nloc, hipMemcpyHostToDevice, s->st));
        } else {
            std::vector<double> tmp((size_t)m * nloc);
            for (int64_t j = 0; j < nloc; ++j)
                for (int64_t i = 0; i < m; ++i) {
                    const double val = a_host[j * lp->lda + i];
                    tmp[(size_t)(j * m + i)] = (val == 0.0) ? 0.0 : val;
                }
            HIP_OK(hipMemcpy2DAsync(A, sizeof(double) * d.lda, tmp.data(), sizeof(double) * m,
                                    sizeof(double) * m, nloc, hipMemcpyHostToDevice, s->st));
            HIP_OK(hipStreamSynchronize(s->st));
        }
    }
    d.A = A;
    if (d.repl && !o.a_is_block && ns > nloc) {
        // lp->a is the whole matrix: the other ranks' columns come from it right away
        if (d.col0 > 0) {
            TRY(upload_columns(s, 0, d.col0, lp->a, lp->lda));
            s->col_blocks.emplace_back(0, d.col0);
        }
        if (d.col1 < ns) {
            TRY(upload_columns(s, d.col1, ns, lp->a + (size_t)d.col1 * (size_t)lp->lda, lp->lda));
            s->col_blocks.emplace_back(d.col1, ns);
        }
    }

    // --- index maps
    std::vector<int> var_col((size_t)(n ? n : 1));
    for (int v = 0; v < n; ++v)
        var_col[v] = lp->var_col ? (int)lp->var_col[v] : (v < ns ? v : -1 - (v - ns));
    std::vector<int> basis((size_t)(m ? m : 1)), nonbasis((size_t)(q ? q : 1));
    long long nb_struct = 0;
    for (int p = 0; p < m; ++p) basis[p] = (int)lp->basis[p];
    for (int k = 0; k < q; ++k) {
        nonbasis[k] = (int)lp->nonbasis[k];
        if (var_col[nonbasis[k]] >= 0) ++nb_struct;
    }
    int *d_var_col, *d_basis, *d_nonbasis;
    TRY(dev_alloc(s, &d_var_col, (size_t)n));
    TRY(dev_alloc(s, &d_basis, (size_t)m));
    TRY(dev_alloc(s, &d_nonbasis, (size_t)q));
    HIP_OK(hipMemcpyAsync(d_var_col, var_col.data(), sizeof(int) * (size_t)(n ? n : 1), hipMemcpyHostToDevice, s->st));
    HIP_OK(hipMemcpyAsync(d_basis, basis.data(), sizeof(int) * (size_t)(m ? m : 1), hipMemcpyHostToDevice, s->st));
    HIP_OK(hipMemcpyAsync(d_nonbasis, nonbasis.data(), sizeof(int) * (size_t)(q ? q : 1), hipMemcpyHostToDevice, s->st));
    d.var_col = d_var_col; d.basis = d_basis; d.nonbasis = d_nonbasis;

    // --- state vectors: x = rhs, z = -c_N, xbar = zbar = 1 (src/simplex.rs:190-205)
    TRY(dev_alloc(s, &d.x, (size_t)m)); TRY(dev_alloc(s, &d.xbar, (size_t)m));
    TRY(dev_alloc(s, &d.z, (size_t)q)); TRY(dev_alloc(s, &d.zbar, (size_t)q));
    TRY(dev_alloc(s, &d.dx, (size_t)m)); TRY(dev_alloc(s, &d.dz, (size_t)q));
    TRY(dev_alloc(s, &d.v, (size_t)m + 2)); TRY(dev_alloc(s, &d.acol, (size_t)m));
    std::vector<double> ones((size_t)((m > q ? m : q) + 1), 1.0);
    if (m > 0) {
        HIP_OK(hipMemcpyAsync(d.x, lp->x, sizeof(double) * m, hipMemcpyHostToDevice, s->st));
        HIP_OK(hipMemcpyAsync(d.xbar, lp->xbar ? lp->xbar : ones.data(), sizeof(double) * m,
                              hipMemcpyHostToDevice, s->st));
    }
    if (q > 0) {
        HIP_OK(hipMemcpyAsync(d.z, lp->z, sizeof(double) * q, hipMemcpyHostToDevice, s->st));
        HIP_OK(hipMemcpyAsync(d.zbar, lp->zbar ? lp->zbar : ones.data(), sizeof(double) * q,
                              hipMemcpyHostToDevice, s->st));
    }
    HIP_OK(hipMemsetAsync(d.v, 0, sizeof(double) * ((size_t)m + 2), s->st));
    s->c_host.assign(lp->c, lp->c + n);
    s->constant = lp->constant;

    // --- pivot log
    long long cap = o.log_capacity >= 0 ? o.log_capacity : (o.max_iter < (1ll << 22) ? o.max_iter : (1ll << 22));
    d.log_cap = cap;
    TRY(dev_alloc(s, &d.log_kind, (size_t)cap)); TRY(dev_alloc(s, &d.log_enter, (size_t)cap));
    TRY(dev_alloc(s, &d.log_leave, (size_t)cap)); TRY(dev_alloc(s, &d.log_mu, (size_t)cap));
    TRY(dev_alloc(s, &d.log_margin, (size_t)cap));

    // --- numerics
    s->numerics = o.numerics == DZG_NUMERICS_AUTO
                      ? (m <= o.auto_strict_rows ? DZG_NUMERICS_STRICT : DZG_NUMERICS_FAST)
                      : o.numerics;
    if (s->numerics != DZG_NUMERICS_STRICT && s->numerics != DZG_NUMERICS_FAST)
        return fail(DZG_E_ARG, "opts.numerics");
    if (d.world > 1 && s->numerics != DZG_NUMERICS_FAST)
        return fail(DZG_E_ARG, "column sharding needs FAST numerics");
    TRY(dev_alloc(s, &d.ctl, 1));
    DzgCtl c0;
    std::memset(&c0, 0, sizeof(c0));
    c0.status = DZG_RUNNING;
    c0.iter_stop = o.max_iter;
    c0.enter_pos = c0.leave_pos = -1;
    c0.del_ce = c0.del_last = -1;
    c0.rl_listed = -1;
    c0.nb_struct = nb_struct;
    c0.tie_tol = c0.tau = o.tie_tol;
    c0.margin = c0.min_margin = std::numeric_limits<double>::infinity();
    c0.first_near_tie = c0.tie_skip_iter = -1;
    c0.tie_mode = o.near_tie_action == DZG_NEAR_TIE_STOP ? 1 : 0;
    HIP_OK(hipMemcpyAsync(d.ctl, &c0, sizeof(c0), hipMemcpyHostToDevice, s->st));
    *s->h_ctl = c0;

    if (s->numerics == DZG_NUMERICS_FAST) {
        // the slack basis of Simplex::new needs no factorisation (Binv0 starts empty); any other
        // starting basis (warm start) is factorised on the device before the first iteration
        bool slack_basis = true;
        for (int p = 0; p < m; ++p)
            if (var_col[basis[p]] >= 0) slack_basis = false;
        if (!slack_basis) {
            // a column-sharded rank that does not hold every basic column yet (partitioned storage, or
            // replicate_matrix before dzg_solver_upload_columns has run) factorises the starting
            // basis at its first run, where the ranks can exchange their columns' shares
            if (d.world > 1 && !(d.repl && s->cols_present == ns)) s->pending_refactor = true;
            if (o.refactor_interval == 0) o.refactor_interval = -1; // reserve the workspace
            s->opts.refactor_interval = o.refactor_interval;
        }
        const bool needs_initial_refactor = !slack_basis && !s->pending_refactor;
        if (slack_basis && d.world == 1 && !d.csc && m > 0 && q > 0) {
            // the state a refactorisation can be checked against (k_drift.hip): b = the starting x,
            // xbar0 = the starting xbar, c
            TRY(dev_alloc(s, &s->dr_b0, (size_t)m + 2)); TRY(dev_alloc(s, &s->dr_xb0, (size_t)m + 2));
            TRY(dev_alloc(s, &s->dr_c, (size_t)n));
            HIP_OK(hipMemcpyAsync(s->dr_b0, lp->x, sizeof(double) * m, hipMemcpyHostToDevice, s->st));
            HIP_OK(hipMemcpyAsync(s->dr_xb0, lp->xbar ? lp->xbar : ones.data(), sizeof(double) * m,
                                  hipMemcpyHostToDevice, s->st));
            HIP_OK(hipMemcpyAsync(s->dr_c, lp->c, sizeof(double) * n, hipMemcpyHostToDevice, s->st));
        }
        // row stride: not a multiple of a large power of two, so that the first k columns of
        // consecutive rows do not all land on the same HBM channels
        // row stride of the compact inverse: 160 doubles past a multiple of 512 (1 280 bytes past a
        // multiple of 4 KB).  FTRAN streams 2 048 rows at once, one wave each; at k = 7 700 of m = 8 192
        // the launch takes 102.4 us with m + 32, 110.7 with m + 64, 97.7 with m + 160 (or + 288, + 544),
        // 101.0 with m + 1 056 (profiles/r04_ftran_row_loads_ab.txt); DZG_LDB_PAD=<doubles> for A/B
        {
            const long long base = ((long long)m + 15) / 16 * 16;
            long long pad = ((160 - base % 512) % 512 + 512) % 512;
            if (pad < 32) pad += 512;
            if (const char *e = std::getenv("DZG_LDB_PAD")) pad = std::atoi(e) / 16 * 16;
            d.ldb = base + pad;
        }
        d.ldw = ((long long)m + 15) / 16 * 16 + 64;
        {
            // a row-sharded basis side keeps its own rows of Binv0 only; the kernels address rows by
            // their global index, so d.binv points rs_r0 rows BEFORE the allocation (never
            // dereferenced outside [rs_r0, rs_r1): every kernel that touches Binv0 takes the range)
            const size_t rows_kept = d.rs ? (size_t)(d.rs_r1 - d.rs_r0) : (size_t)m;
            double *binv_alloc = nullptr;
            TRY(dev_alloc(s, &binv_alloc, (rows_kept ? rows_kept : 1) * (size_t)d.ldb));
            d.binv = binv_alloc - (d.rs ? (long long)d.rs_r0 * d.ldb : 0);
        }
        TRY(dev_alloc(s, &d.drow, (size_t)m)); TRY(dev_alloc(s, &d.dslot, (size_t)m));
        TRY(dev_alloc(s, &d.U, (size_t)d.ldw * DZG_RMAX));
        TRY(dev_alloc(s, &d.W, (size_t)d.ldw * DZG_RMAX));
        TRY(dev_alloc(s, &d.Wc, (size_t)d.ldw * DZG_RMAX));
        TRY(dev_alloc(s, &d.ag, (size_t)m + 2)); TRY(dev_alloc(s, &d.beta, (size_t)DZG_RMAX));
        TRY(dev_alloc(s, &d.plist, (size_t)q)); TRY(dev_alloc(s, &d.pslot, (size_t)q));
        TRY(dev_alloc(s, &d.bcode, (size_t)m)); TRY(dev_alloc(s, &d.nbcode, (size_t)q));
        TRY(dev_alloc(s, &d.pcode, (size_t)q));
        if (!d.csc) TRY(dev_alloc(s, &d.cpos, (size_t)(d.col1 > d.col0 ? d.col1 - d.col0 : 1)));
        const size_t np = 4096;
        TRY(dev_alloc(s, &d.fpx_r, np)); TRY(dev_alloc(s, &d.fpz_r, np));
        TRY(dev_alloc(s, &d.rx_r, np)); TRY(dev_alloc(s, &d.rz_r, np));
        TRY(dev_alloc(s, &d.fpx_k, np)); TRY(dev_alloc(s, &d.fpz_k, np));
        TRY(dev_alloc(s, &d.rx_k, np)); TRY(dev_alloc(s, &d.rz_k, np));
        TRY(dev_alloc(s, &d.fpx_h, np)); TRY(dev_alloc(s, &d.fpz_h, np));
        TRY(dev_alloc(s, &d.rx_h, np)); TRY(dev_alloc(s, &d.rz_h, np));
        if (o.refactor_interval != 0) TRY(refactor_workspace(s));
        if (d.spb && !o.seven_launches) {
            // the four-launch iteration (k_sp_pre / k_sp_mid, k_sparse.hip) synchronises its phases with
            // device-wide barriers: every workgroup of its grid must be resident at once
            hipDeviceProp_t prop;
            HIP_OK(hipGetDeviceProperties(&prop, o.device));
            const int cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 1;
            const char *off = std::getenv("DZG_SP_FUSED");
            if (!(off && off[0] == '0') && (long long)dzg_sp_fused_resident_per_cu() * cus >= dzg_sp_grid(m)) {
                TRY(dev_alloc(s, &s->chain_bar, (size_t)DZG_CHAIN_BAR_WORDS));
                HIP_OK(hipMemsetAsync(s->chain_bar, 0, sizeof(unsigned long long) * DZG_CHAIN_BAR_WORDS, s->st));
            }
        }
        if (d.spb) {
            TRY(dev_alloc(s, &d.sslot, (size_t)m)); TRY(dev_alloc(s, &d.spos, (size_t)m));
            TRY(dev_alloc(s, &d.bslot, (size_t)(ns ? ns : 1))); TRY(dev_alloc(s, &d.rowpos, (size_t)m));
            TRY(dev_alloc(s, &d.dxs, (size_t)m)); TRY(dev_alloc(s, &d.acol_code, 1));
            HIP_OK(hipMemsetAsync(d.acol, 0, sizeof(double) * (size_t)(m ? m : 1), s->st));
            HIP_OK(hipMemsetAsync(d.dxs, 0, sizeof(double) * (size_t)(m ? m : 1), s->st));
        }
        // dense matrix: one GPU, or a column-sharded rank (replicated or partitioned storage)
        if (!d.csc && !o.seven_launches && !d.rs) { // (a partitioned rank takes the column from the record)
            // the chain's barriers need every workgroup resident at once: one per CU, and the
            // runtime must agree that a workgroup of either kernel fits a CU at all (registers,
            // 136 KB of LDS); otherwise the barrier-free seven launches run
            hipDeviceProp_t prop;
            HIP_OK(hipGetDeviceProperties(&prop, o.device));
            int grid = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 1;
            if (grid > 256) grid = 256; // (k_chain_pre reads <= 256 candidates)
            const int per_cu = dzg_chain_resident_per_cu();
            if (per_cu < 1) grid = 0;
            if (const char *g = std::getenv("DZG_CHAIN_GRID")) { // tests: a device with few CUs
                const int v = std::atoi(g);
                if (v >= 1 && v < grid) grid = v;
            }
            // a thread takes one row and one column of its workgroup's share (512 threads)
            const int gdiv = grid > 0 ? grid : 1;
            const long long rows_per = (((long long)m + gdiv - 1) / gdiv + 3) & ~3ll;
            const long long cols_per = ((long long)q + gdiv - 1) / gdiv;
            if (grid >= 1 && rows_per <= 512 && cols_per <= 512) {
                s->chain_grid = grid;
                if (const char *cap = std::getenv("DZG_CHAIN_KCAP")) {
                    const long long v = std::atoll(cap);
                    if (v >= 0 && v < s->chain_kcap) s->chain_kcap = v;
                }
                if (const char *dbg = std::getenv("DZG_CHAIN_DEBUG")) {
                    if (dbg[0] == '1') {
                        TRY(dev_alloc(s, &s->chain_dbg, (size_t)64));
                        HIP_OK(hipMemsetAsync(s->chain_dbg, 0, sizeof(unsigned long long) * 64, s->st));
                    }
                }
                if (const char *nf = std::getenv("DZG_CHAIN_NO_FOLD")) s->chain_fold = nf[0] != '1';
                TRY(dev_alloc(s, &s->chain_bar, (size_t)DZG_CHAIN_BAR_WORDS));
                HIP_OK(hipMemsetAsync(s->chain_bar, 0, sizeof(unsigned long long) * DZG_CHAIN_BAR_WORDS, s->st));
            }
        }
        // Row-wise pricing (k_price_kernels.h): a row-major copy of the rank's own columns and the
        // partial sums of the row groups.  AUTO pricing only (an explicit kernel choice means that
        // kernel); DZG_PRICE_ROWS=0 is the A/B switch of the tools and tests.
        {
            const char *rows_env = std::getenv("DZG_PRICE_ROWS");
            const int nown = d.col1 - d.col0;
            if (!d.csc && o.price_kernel == DZG_PRICE_AUTO && m > 0 && nown > 0 && ns > 0 &&
                !(rows_env && rows_env[0] == '0')) {
                // (no padding: with a row stride that is not a multiple of 4 KB the pass at k = 4 049 takes
                // 93.9 us instead of 85.8, profiles/r04_ftran_row_loads_ab.txt)
                d.ldt = ((long long)nown + 3) / 4 * 4;
                // (the copy doubles the matrix: a device that cannot hold it prices column-wise)
                void *at_mem = nullptr;
                if (hipMalloc(&at_mem, sizeof(double) * (size_t)m * (size_t)d.ldt) != hipSuccess) {
                    (void)hipGetLastError();
                    at_mem = nullptr;
                }
                double *at = static_cast<double *>(at_mem);
                if (at) {
                    s->allocs.push_back(at_mem);
                    TRY(dev_alloc(s, &d.ppart, (size_t)dzg_price_rows_groups() * (size_t)d.ldt));
                    TRY(dev_alloc(s, &d.vc, (size_t)m));
                    HIP_OK(hipMemsetAsync(d.vc, 0, sizeof(double) * (size_t)m, s->st));
                    dzg_launch_transpose_to_rows(d.A, d.lda, m, nown, at, d.ldt, s->st);
                }
                d.At = at;
                // where 8 (k + 1) n_s bytes of rows meet 8 m (n_s - k) bytes of columns, less the
                // second launch: a function of the problem's shape alone (the same on every rank)
                // (0.93: tools/price_rows_bench.hip -- 0.0205 us per row of 16 384 columns + 6 us for the
                // finishing launch against 0.0099 us per column of 8192 rows: equal at k = 5 150 of 5 461)
                d.rows_T = (int)(0.93 * (double)m * (double)ns / ((double)ns + (double)m));
                if (const char *t = std::getenv("DZG_PRICE_ROWS_T")) d.rows_T = std::atoi(t);
                if (d.rows_T < 1) { d.At = nullptr; d.vc = nullptr; }
            }
        }
        dzg_launch_fast_init(d, s->st);
        if (d.spb) {
            dzg_launch_sp_init(d, 1, s->st);
            dzg_launch_sp_update(d, 1, s->st);
        } else {
            dzg_launch_fast_update(d, 1, s->st); // first-pivot partials of the initial state
        }
        if (needs_initial_refactor) {
            TRY(refactor_now(s));
        }
    } else {
        dzg_lu_layout(m, &d.lu);
        const size_t mm = (size_t)(m ? m : 1);
        TRY(dev_alloc(s, &d.lu.W, mm * (size_t)d.lu.ldw));
        TRY(dev_alloc(s, &d.lu.P0, mm * DZG_LU_LDP)); TRY(dev_alloc(s, &d.lu.P1, mm * DZG_LU_LDP));
        TRY(dev_alloc(s, &d.lu.piv, mm)); TRY(dev_alloc(s, &d.lu.pz, mm));
        TRY(dev_alloc(s, &d.lu.ptab, mm * DZG_LU_NB));
        TRY(dev_alloc(s, &d.lu.part_r, (size_t)2 * d.lu.nparts));
        TRY(dev_alloc(s, &d.lu.part_k, (size_t)2 * d.lu.nparts));
    }
    if (o.profile) {
        s->ev.resize((size_t)o.poll_interval * DZG_K_COUNT * 2);
        // (timing only, read after the stream has been waited for: without the system-scope fence a
        // default event carries -- 5.7 us of idle GPU per record between two 10-us kernels,
        // profiles/r04_chain_phase_clocks.txt)
        for (auto &e : s->ev) HIP_OK(hipEventCreateWithFlags(&e, hipEventDisableSystemFence));
    }
    HIP_OK(hipStreamSynchronize(s->st));
    HIP_OK(hipGetLastError());
    guard.s = nullptr;
    *out = s;
    return 0;
}

// ---- one iteration, enqueued ------------------------------------------------------
namespace {
// opts.profile: the classes to time (low 16 bits; any negative value: all of them) and, in bits
// 16..23, a sampling stride S > 1: only every S-th iteration of a batch is stamped
static inline bool prof_on(const dzg_solver *s, int cls, int slot)
{
    const int p = s->opts.profile;
    if (p == 0 || slot < 0) return false;
    if (p < 0) return true;
    const int stride = (p >> 16) & 0xff;
    // (the last of each S: the first iteration of a batch starts on an empty queue)
    return (p & (1 << cls)) && (stride <= 1 || slot % stride == stride - 1);
}

struct Prof {
    dzg_solver *s;
    int slot;
    void begin(int cls) const
    {
        if (prof_on(s, cls, slot))
            hipEventRecord(s->ev[((size_t)slot * DZG_K_COUNT + cls) * 2], s->st);
    }
    void end(int cls) const
    {
        if (prof_on(s, cls, slot))
            hipEventRecord(s->ev[((size_t)slot * DZG_K_COUNT + cls) * 2 + 1], s->st);
    }
};
} // namespace

static int price_kernel_for(const dzg_solver *s)
{
    // AUTO: FAST numerics streams with the register-accumulator kernel, whose sums depend only
    // on m (bit-identical dz on one GPU and on any column sharding); the reference-order kernel
    // (bit-identical to neg_t_dot) belongs to STRICT, whose v is the reference's v.
    if (s->d.csc) // CSC: the tree-order kernel unless the options insist on the reference's order
        return s->opts.price_kernel == DZG_PRICE_SEQ ? DZG_PRICE_SEQ : DZG_PRICE_CSC_KERNEL;
    if (s->opts.price_kernel != DZG_PRICE_AUTO) return s->opts.price_kernel;
    return DZG_PRICE_TREE;
}

// per-workgroup ratio partials the pricing pass leaves (every CSC kernel uses the same grid)
static int price_partials_for(const dzg_solver *s, int pk)
{
    return dzg_price_partials_dev(s->d, pk);
}

static void enqueue_fast_iteration(dzg_solver *s, int slot)
{
    const DzgDev &d = s->d;
    hipStream_t st = s->st;
    Prof pf{s, slot};
    const int pk = price_kernel_for(s);
    pf.begin(DZG_K_STATUS);
    dzg_launch_fast_select_prep(d, 0, 0, nullptr, st); // status() + FTRAN prep of a primal step
    pf.end(DZG_K_STATUS);
    pf.begin(DZG_K_FTRAN);
    dzg_launch_fast_gemv(d, DZG_STEP_PRIMAL, nullptr, st); // primal step: dx first (+ ratio partials)
    pf.end(DZG_K_FTRAN);
    pf.begin(DZG_K_BTRAN);
    dzg_launch_fast_btran(d, st); // (primal: finishes the ratio test) v = row p of Binv
    pf.end(DZG_K_BTRAN);
    pf.begin(DZG_K_PRICE);
    dzg_launch_price_fast(d, pk, st); // dz (+ dual ratio partials)
    pf.end(DZG_K_PRICE);
    pf.begin(DZG_K_RATIO);
    dzg_launch_fast_select_prep(d, 1, price_partials_for(s, pk), nullptr, st); // dual: ratio + prep
    dzg_launch_fast_gemv(d, DZG_STEP_DUAL, nullptr, st);                    // dual step: dx last
    pf.end(DZG_K_RATIO);
    pf.begin(DZG_K_UPDATE);
    dzg_launch_fast_update(d, 0, st); // (the pivot's books were kept by the launch before)
    pf.end(DZG_K_UPDATE);
    if (++s->since_flush >= DZG_RMAX) {
        pf.begin(DZG_K_BASIS_UPDATE);
        dzg_launch_fast_flush(d, st); // Binv0 -= U Wc on the fp64 matrix cores
        pf.end(DZG_K_BASIS_UPDATE);
        s->since_flush = 0;
    } else {
        pf.begin(DZG_K_BASIS_UPDATE);
        pf.end(DZG_K_BASIS_UPDATE);
    }
}

// Dense matrix on one GPU: the same iteration in three launches (k_chain.hip).
static void enqueue_chain_iteration(dzg_solver *s, int slot)
{
    const DzgDev &d = s->d;
    hipStream_t st = s->st;
    Prof pf{s, slot};
    const int pk = price_kernel_for(s);
    pf.begin(DZG_K_FTRAN);
    dzg_launch_chain_pre(d, s->chain_grid, s->chain_bar, s->chain_dbg, nullptr, st); // status, primal FTRAN + ratio, BTRAN row
    pf.end(DZG_K_FTRAN);
    // k below ~480 for the whole batch: one fused kernel prices row-wise AND finishes (dz, ratio
    // candidates) per column tile -- no partial sums in memory, nothing to fold.  Otherwise, while
    // the batch prices row-wise for certain, k_chain_post finishes the pass itself (FOLD)
    const int small = dzg_price_small(d, pk);
    const int fold = !small && s->chain_fold && dzg_price_rows_certain(d, pk);
    pf.begin(DZG_K_PRICE);
    dzg_launch_price_fast(d, pk, st, -1, fold, small);
    pf.end(DZG_K_PRICE);
    pf.begin(DZG_K_UPDATE);
    dzg_launch_chain_post(d, s->chain_grid, s->chain_bar, s->chain_dbg, 0,
                          small ? dzg_price_small_partials(d)
                                : (fold ? s->chain_grid : price_partials_for(s, pk)),
                          nullptr, st, fold, small);
    pf.end(DZG_K_UPDATE);
    pf.begin(DZG_K_BASIS_UPDATE);
    if (++s->since_flush >= DZG_RMAX) {
        dzg_launch_fast_flush(d, st);
        s->since_flush = 0;
    }
    pf.end(DZG_K_BASIS_UPDATE);
}

// CSC input on one GPU: the sparse-basis kernels (k_sparse.hip).  Same phases as above.
static void enqueue_sparse_iteration(dzg_solver *s, int slot)
{
    const DzgDev &d = s->d;
    hipStream_t st = s->st;
    Prof pf{s, slot};
    if (s->batch_chain) { // four launches: the phases of one side of the pricing pass share a launch
        const int pk = price_kernel_for(s);
        pf.begin(DZG_K_FTRAN);
        dzg_launch_sp_pre(d, s->chain_bar, st); // status, primal FTRAN (both halves) + ratio test, BTRAN row
        pf.end(DZG_K_FTRAN);
        pf.begin(DZG_K_PRICE);
        dzg_launch_price_fast(d, pk, st);
        pf.end(DZG_K_PRICE);
        pf.begin(DZG_K_RATIO);
        dzg_launch_sp_mid(d, s->chain_bar, price_partials_for(s, pk), st); // dual ratio + FTRAN; the books
        pf.end(DZG_K_RATIO);
        pf.begin(DZG_K_UPDATE);
        dzg_launch_sp_update(d, 0, st);
        pf.end(DZG_K_UPDATE);
        pf.begin(DZG_K_BASIS_UPDATE);
        if (++s->since_flush >= DZG_RMAX) {
            dzg_launch_sp_flush(d, st);
            s->since_flush = 0;
        }
        pf.end(DZG_K_BASIS_UPDATE);
        return;
    }
    pf.begin(DZG_K_FTRAN);
    dzg_launch_sp_ftran(d, DZG_STEP_PRIMAL, 0, st); // status() + primal FTRAN (a no-op in a dual step)
    pf.end(DZG_K_FTRAN);
    pf.begin(DZG_K_BTRAN);
    dzg_launch_sp_btran(d, st);
    pf.end(DZG_K_BTRAN);
    pf.begin(DZG_K_PRICE);
    dzg_launch_price_fast(d, price_kernel_for(s), st);
    pf.end(DZG_K_PRICE);
    pf.begin(DZG_K_RATIO);
    dzg_launch_sp_ftran(d, DZG_STEP_DUAL, price_partials_for(s, price_kernel_for(s)), st); // ratio + dual FTRAN
    pf.end(DZG_K_RATIO);
    pf.begin(DZG_K_UPDATE);
    dzg_launch_sp_pivot(d, st);
    dzg_launch_sp_update(d, 0, st);
    pf.end(DZG_K_UPDATE);
    pf.begin(DZG_K_BASIS_UPDATE);
    if (++s->since_flush >= DZG_RMAX) {
        dzg_launch_sp_flush(d, st); // X -= Ub^T Wc on the fp64 matrix cores
        s->since_flush = 0;
    }
    pf.end(DZG_K_BASIS_UPDATE);
}

static void collect_profile(dzg_solver *s, int slots_real)
{
    if (!s->opts.profile) return;
    for (int slot = 0; slot < slots_real; ++slot)
        for (int cls = 0; cls < DZG_K_COUNT; ++cls) {
            if (!prof_on(s, cls, slot)) continue;
            if (s->d.csc && !s->d.spb && s->d.world == 1 && !s->comm && cls != DZG_K_PRICE)
                continue; // the single-GPU record path of a CSC solver only stamps pricing
            if (s->batch_chain && s->d.world == 1 && !s->comm &&
                (cls == DZG_K_STATUS || cls == DZG_K_BTRAN || (cls == DZG_K_RATIO && !s->d.spb)))
                continue; // the chain has no launches of their own for these
            float ms = 0.f;
            size_t base = ((size_t)slot * DZG_K_COUNT + cls) * 2;
            if (hipEventElapsedTime(&ms, s->ev[base], s->ev[base + 1]) == hipSuccess) {
                s->kernel_ms[cls] += ms;
                s->kernel_launches[cls] += 1;
            }
        }
    (void)hipGetLastError(); // a class this numerics mode never stamps (e.g. DZG_K_LU in FAST)
}

static int read_ctl(dzg_solver *s)
{
    HIP_OK(hipMemcpyAsync(s->h_ctl, s->d.ctl, sizeof(DzgCtl), hipMemcpyDeviceToHost, s->st));
    HIP_OK(hipStreamSynchronize(s->st));
    if (s->h_ctl->max_pivot_err > s->max_err_life) s->max_err_life = s->h_ctl->max_pivot_err;
    return 0;
}

// Rebuild Binv0 from scratch for the current basis (k_refactor.hip).  One host sync to learn
// k; everything else is enqueued.
// The refactorisation workspace (two m x m panels and a few index lists) is reserved when the
// options ask for periodic refactorisation, and otherwise on first need: a solver that never
// drifts never pays for it, and one that does can still recover.
static int refactor_workspace(dzg_solver *s)
{
    if (s->rf_piv) return 0;
    const size_t m = (size_t)(s->d.m ? s->d.m : 1);
    s->rf_ld = ((long long)s->d.m + 15) / 16 * 16 + 16;
    TRY(dev_alloc(s, &s->rfPn, (size_t)64 * (size_t)s->rf_ld));
    TRY(dev_alloc(s, &s->rfTri, ((m + 63) / 64) * (size_t)(2 * 64 * 64)));
    TRY(dev_alloc(s, &s->rf_spos, m));
    TRY(dev_alloc(s, &s->rf_scode, m)); TRY(dev_alloc(s, &s->rf_lpos, m));
    TRY(dev_alloc(s, &s->rf_lrow, m)); TRY(dev_alloc(s, &s->rf_counts, 4));
    TRY(dev_alloc(s, &s->rf_lslot, m));
    TRY(dev_alloc(s, &s->rf_piv, m)); // set last: rf_piv != nullptr means "lists reserved"
    return 0;
}

// The two big panels G and X hold k x k (the structural block, its factors, the inverse) and then
// nl x k (the basic-slack rows A[L, S]): max(k, nl) rows of rf_ld doubles each -- half of m x m when
// the basis is half structural, which is what lets eight ranks of config 5 share one device in the
// lockstep harness.  They grow when a later basis needs more rows.
static int refactor_panels(dzg_solver *s, int k, int nl)
{
    const long long need = k > nl ? k : nl;
    if (s->rfG && s->rf_rows >= need) return 0;
    for (double **p : {&s->rfG, &s->rfX})
        if (*p) {
            auto it = std::find(s->allocs.begin(), s->allocs.end(), (void *)*p);
            if (it != s->allocs.end()) s->allocs.erase(it);
            HIP_OK(hipFree(*p));
            *p = nullptr;
        }
    long long rows = need + need / 8 + 64; // (head room: k moves by one per pivot)
    if (rows > s->d.m) rows = s->d.m;
    if (rows < 1) rows = 1;
    double *x = nullptr, *g = nullptr;
    TRY(dev_alloc(s, &x, (size_t)rows * (size_t)s->rf_ld));
    TRY(dev_alloc(s, &g, (size_t)rows * (size_t)s->rf_ld));
    s->rfX = x;
    s->rfG = g;
    s->rf_rows = rows;
    return 0;
}

// A rank of a PARTITIONED column-sharded solve holds its own structural columns only (a
// replicated one holds them all, like a single GPU).
static bool partitioned(const dzg_solver *s) { return s->d.world > 1 && !s->d.repl; }

// Stage A: the basis lists, one host sync to learn k and nl, the rank's share of G = A[R, S].
// *active = false: the solve has ended, nothing to refactorise (every rank alike).
static int refactor_stage_a(dzg_solver *s, int counts[2], bool *active)
{
    TRY(refactor_workspace(s));
    const DzgDev &d = s->d;
    dzg_launch_refactor_lists(d, s->rf_spos, s->rf_scode, s->rf_lpos, s->rf_lrow, s->rf_counts, s->st);
    counts[0] = counts[1] = 0;
    HIP_OK(hipMemcpyAsync(counts, s->rf_counts, sizeof(int) * 2, hipMemcpyDeviceToHost, s->st));
    TRY(read_ctl(s));
    *active = s->h_ctl->status == DZG_RUNNING || s->h_ctl->status == DZG_ITER_LIMIT ||
              s->h_ctl->status == DZG_NEAR_TIE;
    if (!*active) return 0;
    if (counts[0] != s->h_ctl->ncompact)
        return fail(DZG_E_DEVICE, "refactor: structural basics != dense columns");
    TRY(refactor_panels(s, counts[0], counts[1]));
    dzg_launch_refactor_a(d, counts[0], s->rfG, s->rfX, s->rf_ld, s->rf_scode, s->rf_counts + 2, s->st);
    return 0;
}

// Stage B: LU, the inverse, Binv0's rows of the structural positions, the rank's share of A[L, S]
// (*block, nl x rf_ld; nullptr: none).
static int refactor_stage_b(dzg_solver *s, const int counts[2], double **block)
{
    *block = dzg_launch_refactor_b(s->d, counts[0], counts[1], s->rfG, s->rfX, s->rfPn, s->rfTri, s->rf_ld,
                                   s->rf_piv, s->rf_spos, s->rf_scode, s->rf_lpos, s->rf_lrow, s->rf_lslot,
                                   s->rf_counts + 2, s->st);
    return 0;
}

// Stage C: Binv0's rows of the basic slacks, the eta file emptied, the monitor restarted.
static int refactor_stage_c(dzg_solver *s, const int counts[2])
{
    dzg_launch_refactor_c(s->d, counts[0], counts[1], s->rfG, s->rfX, s->rf_ld, s->rf_lpos,
                          s->rf_counts + 2, s->st);
    s->since_flush = 0;
    s->h_ctl->neta = 0; // the refactorisation empties the eta file (k_ref_done)
    s->since_refactor = 0;
    s->refactors += 1;
    s->pending_refactor = false;
    return 0;
}

// sum of a block over the ranks of the RCCL communicator, in place (a column has one owner: exact)
static int shard_sum(dzg_solver *s, double *buf, size_t count)
{
    if (count == 0) return 0;
    Rccl &r = rccl();
    if (!s->comm || !r.ok || !r.AllReduce)
        return fail(DZG_E_ARG, "refactorisation of a partitioned column-sharded solver needs its "
                               "ranks' columns: dzg_shard_comm_init first (or the lockstep loop)");
    if (r.AllReduce(buf, buf, count, kNcclFloat64, kNcclSum, s->comm, s->st) != 0)
        return fail(DZG_E_DEVICE, "ncclAllReduce (basic columns of a refactorisation)");
    return 0;
}

// The carried x, xbar, z against their recomputation from the inverse the refactorisation has just
// built (k_drift.hip); the result widens the near-tie tolerance and is reported as state_drift.
static int measure_drift(dzg_solver *s)
{
    if (!s->dr_b0 || s->d.world != 1 || s->d.spb) return 0;
    DzgDev &d = s->d;
    if (!s->dr_out) {
        TRY(dev_alloc(s, &s->dr_agb, (size_t)d.m + 2)); TRY(dev_alloc(s, &s->dr_agx, (size_t)d.m + 2));
        TRY(dev_alloc(s, &s->dr_part, (size_t)dzg_drift_chunks() * (size_t)d.ldw));
        TRY(dev_alloc(s, &s->dr_y, (size_t)d.m + 2)); TRY(dev_alloc(s, &s->dr_dzy, (size_t)d.q));
        TRY(dev_alloc(s, &s->dr_out, (size_t)6 * dzg_drift_blocks()));
    }
    dzg_launch_drift(d, s->dr_b0, s->dr_xb0, s->dr_c, s->dr_agb, s->dr_agx, s->dr_part, s->dr_y, s->dr_dzy,
                     s->dr_out, (int)s->h_ctl->ncompact, s->st);
    const int nb = dzg_drift_blocks();
    std::vector<double> out((size_t)6 * nb);
    HIP_OK(hipMemcpyAsync(out.data(), s->dr_out, sizeof(double) * out.size(), hipMemcpyDeviceToHost, s->st));
    HIP_OK(hipStreamSynchronize(s->st));
    double e[6] = {0, 0, 0, 0, 0, 0};
    for (int b = 0; b < nb; ++b) {
        for (int j = 0; j < 4; ++j) e[j] = std::max(e[j], out[(size_t)4 * b + j]);
        for (int j = 0; j < 2; ++j) e[4 + j] = std::max(e[4 + j], out[(size_t)4 * nb + 2 * b + j]);
    }
    double drift = 0.0;
    for (int j = 0; j < 6; j += 2) {
        const double rel = e[j] / std::max(1.0, e[j + 1]);
        if (rel == rel && rel > drift) drift = rel;
    }
    s->state_drift = drift;
    DzgCtl *h = s->h_ctl;
    h->drift_tau = 4.0 * drift;
    HIP_OK(hipMemcpyAsync(&d.ctl->drift_tau, &h->drift_tau, sizeof(double), hipMemcpyHostToDevice, s->st));
    if (h->tie_tol >= 0.0) { // (the monitor restarted with the fresh inverse: max_pivot_err = 0)
        h->tau = std::max(h->tie_tol, h->drift_tau);
        HIP_OK(hipMemcpyAsync(&d.ctl->tau, &h->tau, sizeof(double), hipMemcpyHostToDevice, s->st));
    }
    HIP_OK(hipStreamSynchronize(s->st));
    return 0;
}

static int refactor_now(dzg_solver *s)
{
    if (partitioned(s) && s->in_lockstep)
        return fail(DZG_E_ARG, "refactor: the lockstep loop refactorises its ranks together");
    int counts[2];
    bool active = false;
    TRY(refactor_stage_a(s, counts, &active));
    if (!active) return 0;
    // (every rank of a sharded solve gets here at the same pivot count with the same k and nl: the
    // triggers -- interval, health monitor, a pending warm start -- read replicated state only)
    if (partitioned(s)) TRY(shard_sum(s, s->rfG, (size_t)counts[0] * (size_t)s->rf_ld));
    double *block = nullptr;
    TRY(refactor_stage_b(s, counts, &block));
    if (partitioned(s) && block) TRY(shard_sum(s, block, (size_t)counts[1] * (size_t)s->rf_ld));
    TRY(refactor_stage_c(s, counts));
    return measure_drift(s);
}

// All ranks of a lockstep group (one process, one device, one stream) refactorise together; the
// sums over the ranks of a partitioned solve are one kernel each.
static int refactor_lockstep(dzg_solver **sv, int world)
{
    if (!partitioned(sv[0])) {
        for (int r = 0; r < world; ++r) TRY(refactor_now(sv[r]));
        return 0;
    }
    std::vector<int> counts((size_t)2 * world, 0);
    bool active = false;
    for (int r = 0; r < world; ++r) {
        bool a = false;
        TRY(refactor_stage_a(sv[r], &counts[(size_t)2 * r], &a));
        if (r > 0 && (a != active || counts[(size_t)2 * r] != counts[0] || counts[(size_t)2 * r + 1] != counts[1]))
            return fail(DZG_E_DEVICE, "lockstep refactor: ranks diverged");
        active = a;
    }
    if (!active) return 0;
    hipStream_t st = sv[0]->st;
    double **dptrs = nullptr;
    HIP_OK(hipMalloc(&dptrs, sizeof(double *) * (size_t)world));
    struct Free { double **p; ~Free() { hipFree(p); } } free_ptrs{dptrs};
    std::vector<double *> ptrs((size_t)world);
    for (int r = 0; r < world; ++r) ptrs[(size_t)r] = sv[r]->rfG;
    HIP_OK(hipMemcpyAsync(dptrs, ptrs.data(), sizeof(double *) * (size_t)world, hipMemcpyHostToDevice, st));
    dzg_launch_lockstep_sum(dptrs, world, (long long)counts[0] * sv[0]->rf_ld, st);
    HIP_OK(hipStreamSynchronize(st)); // (ptrs is rewritten below)
    bool any_block = false;
    for (int r = 0; r < world; ++r) {
        double *block = nullptr;
        TRY(refactor_stage_b(sv[r], &counts[(size_t)2 * r], &block));
        ptrs[(size_t)r] = block;
        any_block = any_block || block;
    }
    if (any_block) {
        HIP_OK(hipMemcpyAsync(dptrs, ptrs.data(), sizeof(double *) * (size_t)world, hipMemcpyHostToDevice, st));
        dzg_launch_lockstep_sum(dptrs, world, (long long)counts[1] * sv[0]->rf_ld, st);
        HIP_OK(hipStreamSynchronize(st));
    }
    for (int r = 0; r < world; ++r) TRY(refactor_stage_c(sv[r], &counts[(size_t)2 * r]));
    return 0;
}

extern "C" int dzg_solver_refactor(dzg_solver *s)
{
    if (!s || s->numerics != DZG_NUMERICS_FAST) return fail(DZG_E_ARG, "refactor: FAST solver");
    HIP_OK(hipSetDevice(s->opts.device));
    TRY(refactor_now(s));
    HIP_OK(hipStreamSynchronize(s->st));
    HIP_OK(hipGetLastError());
    return 0;
}

// iterations to enqueue before the next status poll: never past the run budget (an iteration
// enqueued beyond it would be a no-op, but it would still cost its launches and exchanges)
static int batch_size(const dzg_solver *s)
{
    const long long remaining = s->h_ctl->iter_stop - s->h_ctl->iter;
    const int poll = s->opts.poll_interval;
    return (int)(remaining < poll ? (remaining < 1 ? 1 : remaining) : poll);
}

// FAST health rule without a refactorisation workspace: the pivot element computed by FTRAN
// and by BTRAN + pricing disagree beyond repair -> stop with DZG_SINGULAR rather than wander.
static int health_stop(dzg_solver *s)
{
    if (s->h_ctl->max_pivot_err > 1e-4) {
        int st = DZG_SINGULAR;
        HIP_OK(hipMemcpy(&s->d.ctl->status, &st, sizeof(int), hipMemcpyHostToDevice));
        s->h_ctl->status = st;
    }
    return 0;
}

// FAST health between batches: the pivot element computed by FTRAN and by BTRAN + pricing must
// agree.  A drift is shed by a refactorisation (workspace reserved on first need; the ranks of a
// partitioned sharded solve exchange their basic columns for it); where that is impossible -- a
// partitioned solver whose host drives the phases itself, out of memory -- or does not help, the
// solve stops with DZG_SINGULAR.  Every rank of a sharded solve sees the same max_pivot_err
// (replicated dx_p, published dz_r) and therefore takes the same action.
enum HealthAction { HEALTH_OK = 0, HEALTH_REFACTOR, HEALTH_GIVE_UP };

static bool can_refactor(const dzg_solver *s)
{
    return !partitioned(s) || s->comm != nullptr || s->in_lockstep;
}

static HealthAction health_decide(dzg_solver *s)
{
    if (!can_refactor(s)) return s->h_ctl->max_pivot_err > 1e-4 ? HEALTH_GIVE_UP : HEALTH_OK;
    if (!(s->h_ctl->max_pivot_err > s->drift_trigger && s->since_refactor > 0)) return HEALTH_OK;
    // a basis whose fresh inverse drifts again at once is ill-conditioned, not stale: accept a
    // larger disagreement instead of refactorising every batch
    const bool fresh = s->refactors > 0 && s->since_refactor <= 2ll * s->opts.poll_interval;
    if (fresh) s->drift_trigger *= 100.0;
    if (fresh && s->h_ctl->max_pivot_err > 1e-4) return HEALTH_GIVE_UP;
    return HEALTH_REFACTOR;
}

static int health_give_up(dzg_solver *s, bool *stop)
{
    if (can_refactor(s)) s->h_ctl->max_pivot_err = 1.0;
    TRY(health_stop(s));
    *stop = s->h_ctl->status != DZG_RUNNING;
    return 0;
}

static int health_check(dzg_solver *s, bool *stop)
{
    *stop = false;
    switch (health_decide(s)) {
    case HEALTH_OK: return 0;
    case HEALTH_REFACTOR: {
        const int rrc = refactor_now(s);
        // (a refactorisation that FAILED as a call -- device error, out of memory -- is that error,
        // not a singular basis)
        if (rrc == DZG_E_DEVICE || rrc == DZG_E_NOMEM) return rrc;
        if (rrc != 0) return health_give_up(s, stop);
        return 0;
    }
    default: return health_give_up(s, stop);
    }
}

// Budget spent: the host ends the run itself instead of enqueueing an iteration that would only
// discover it (a no-op iteration would still advance the host's flush bookkeeping, and the eta
// flush must fall after the same pivots however a solve is cut into runs: results are
// reproducible bit for bit across budgets, poll intervals and near-tie stops).
static int budget_spent(dzg_solver *s, bool *spent)
{
    *spent = s->h_ctl->status == DZG_RUNNING && s->h_ctl->iter >= s->h_ctl->iter_stop;
    if (*spent) {
        int st = DZG_ITER_LIMIT;
        HIP_OK(hipMemcpy(&s->d.ctl->status, &st, sizeof(int), hipMemcpyHostToDevice));
        s->h_ctl->status = st;
    }
    return 0;
}

// A device-wide barrier of the three-launch iteration failed (its workgroups were not all resident:
// another kernel holds CUs or LDS of this device).  chain_barrier fails CONSISTENTLY -- every
// workgroup of the launch alike -- and both chain kernels write nothing but scratch (dx, v, beta,
// candidates, the decision fields of the control block) before their last barrier, and every
// later launch of the batch saw status != RUNNING: the solver state is that of the last completed
// pivot.  Clear the counters, carry on with the barrier-free seven launches (same arithmetic, same
// pivots) and give the chain another chance later; after three failures it stays off.
static int chain_recover(dzg_solver *s)
{
    DzgCtl *h = s->h_ctl;
    s->chain_fallbacks += 1;
    s->chain_retry_iter = s->chain_fallbacks >= 3 ? std::numeric_limits<long long>::max()
                                                  : h->iter + 64ll * s->opts.poll_interval;
    HIP_OK(hipMemsetAsync(s->chain_bar, 0, sizeof(unsigned long long) * DZG_CHAIN_BAR_WORDS, s->st));
    h->status = DZG_RUNNING;
    h->bar_timeout = 0;
    h->bar_gen = 0;
    HIP_OK(hipMemcpyAsync(&s->d.ctl->status, &h->status, sizeof(int), hipMemcpyHostToDevice, s->st));
    HIP_OK(hipMemcpyAsync(&s->d.ctl->bar_timeout, &h->bar_timeout, sizeof(int), hipMemcpyHostToDevice, s->st));
    HIP_OK(hipMemcpyAsync(&s->d.ctl->bar_gen, &h->bar_gen, sizeof(unsigned long long), hipMemcpyHostToDevice, s->st));
    HIP_OK(hipStreamSynchronize(s->st));
    return 0;
}

static int run_fast(dzg_solver *s)
{
    for (;;) {
        bool spent = false;
        TRY(budget_spent(s, &spent));
        if (spent) break;
        if (s->opts.refactor_interval > 0 && s->since_refactor >= s->opts.refactor_interval)
            TRY(refactor_now(s));
        const long long before = s->h_ctl->iter;
        const int batch = batch_size(s);
        s->since_flush = s->h_ctl->neta; // pending etas as the device counts them
        s->d.price_cols_hint = (int)s->h_ctl->nb_struct + batch; // (at most one more per pivot)
        // the eta flush's grid follows k as the host knows it (k grows by at most one per pivot).
        // Only around launches this loop enqueues itself: the phase entry points are also driven
        // from outside (sharded hosts), where nobody refreshes the bound
        const int k_bound = (int)s->h_ctl->ncompact + batch;
        const int k_low = (int)s->h_ctl->ncompact > batch ? (int)s->h_ctl->ncompact - batch : 0;
        if (s->d.spb) {
            s->d.k_hint = k_bound;
            s->batch_chain = s->chain_bar && s->h_ctl->iter >= s->chain_retry_iter;
            for (int b = 0; b < batch; ++b) enqueue_sparse_iteration(s, b);
            s->d.k_hint = 0;
        } else if (s->d.csc) { // sparse input: the record-based phases, exchanging with itself
            TRY(shard_buffers(s));
            const size_t nb = sizeof(double) * (size_t)s->d.xstride;
            for (int b = 0; b < batch; ++b) {
                s->prof_slot = s->opts.profile ? b : -1;
                TRY(dzg_shard_phase1(s, s->xsend));
                HIP_OK(hipMemcpyAsync(s->xrecv1, s->xsend, nb, hipMemcpyDeviceToDevice, s->st));
                TRY(dzg_shard_phase2(s, s->xrecv1, s->xsend));
                HIP_OK(hipMemcpyAsync(s->xrecv2, s->xsend, nb, hipMemcpyDeviceToDevice, s->st));
                TRY(dzg_shard_phase3(s, s->xrecv2));
            }
            s->prof_slot = -1;
        } else {
            // the chain keeps the gathered entering column in LDS: beyond that compact width
            // (k grows by at most one per pivot) the batch runs as seven launches
            s->batch_chain = s->chain_bar && s->h_ctl->iter >= s->chain_retry_iter &&
                             (long long)s->h_ctl->ncompact + batch <= s->chain_kcap;
            s->d.k_hint = k_bound;
            s->d.k_lo_hint = k_low;
            if (s->batch_chain)
                for (int b = 0; b < batch; ++b) enqueue_chain_iteration(s, b);
            else
                for (int b = 0; b < batch; ++b) enqueue_fast_iteration(s, b);
            s->d.k_hint = s->d.k_lo_hint = 0;
        }
        s->since_refactor += batch;
        TRY(read_ctl(s));
        HIP_OK(hipGetLastError());
        if (s->h_ctl->bar_timeout) { // the batch ended at a failed barrier: carry on without barriers
            TRY(chain_recover(s));
            continue;
        }
        collect_profile(s, (int)(s->h_ctl->iter - before));
        if (s->h_ctl->status != DZG_RUNNING) break;
        bool stop = false;
        TRY(health_check(s, &stop));
        if (stop) break;
    }
    return 0;
}

// STRICT basis solve = gather + 2(m-1) elimination launches + one substitution launch, the same
// sequence with the same arguments every time (the kernels read the basis and the control
// block from memory): capture it once per orientation and replay it, so an iteration costs two
// graph launches instead of ~4m kernel launches on the host.
static void strict_solve(dzg_solver *s, int transposed)
{
    const DzgDev &d = s->d;
    hipStream_t st = s->st;
    if (!s->graphs_tried) {
        s->graphs_tried = true;
        const bool off = getenv("DZG_NO_GRAPH") != nullptr; // diagnostic switch
        for (int t = 0; t < 2 && !off; ++t) {
            hipGraph_t graph = nullptr;
            if (hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal) != hipSuccess) break;
            dzg_launch_strict_solve(d, t, st);
            if (hipStreamEndCapture(st, &graph) != hipSuccess || !graph) {
                (void)hipGetLastError();
                break;
            }
            if (hipGraphInstantiate(&s->g_solve[t], graph, nullptr, nullptr, 0) != hipSuccess) {
                s->g_solve[t] = nullptr;
                (void)hipGetLastError();
            }
            hipGraphDestroy(graph);
        }
    }
    if (s->g_solve[transposed] && hipGraphLaunch(s->g_solve[transposed], st) == hipSuccess) return;
    dzg_launch_strict_solve(d, transposed, st); // capture unavailable: plain launches
}

static int run_strict(dzg_solver *s)
{
    const DzgDev &d = s->d;
    hipStream_t st = s->st;
    for (;;) {
        dzg_launch_status(d, st);
        TRY(read_ctl(s));
        if (s->h_ctl->status != DZG_RUNNING) break;
        if (s->h_ctl->kind == DZG_STEP_PRIMAL) { // src/simplex.rs:308-318
            dzg_launch_load_column(d, -1, st);
            hipMemcpyAsync(d.dx, d.acol, sizeof(double) * (size_t)d.m, hipMemcpyDeviceToDevice, st);
            strict_solve(s, 0);
            dzg_launch_ratio(d, DZG_STEP_PRIMAL, st);
            dzg_launch_unit_rhs(d, st); // needs leave_pos: no-op if the ratio test found none
            strict_solve(s, 1);
            dzg_launch_price(d, DZG_PRICE_SEQ, st);
        } else { // :320-330
            dzg_launch_unit_rhs(d, st);
            strict_solve(s, 1);
            dzg_launch_price(d, DZG_PRICE_SEQ, st);
            dzg_launch_ratio(d, DZG_STEP_DUAL, st);
            dzg_launch_load_column(d, -1, st);
            hipMemcpyAsync(d.dx, d.acol, sizeof(double) * (size_t)d.m, hipMemcpyDeviceToDevice, st);
            strict_solve(s, 0);
        }
        dzg_launch_prepare(d, st);
        dzg_launch_update_vectors(d, st);
    }
    HIP_OK(hipGetLastError());
    return 0;
}

static int set_budget(dzg_solver *s, int64_t max_new_iters)
{
    if (s->d.repl && s->cols_present != s->d.ns)
        return fail(DZG_E_ARG, "replicate_matrix: " + std::to_string(s->d.ns - s->cols_present) +
                               " structural columns have not been uploaded (dzg_solver_upload_columns)");
    s->d.k_hint = s->d.k_lo_hint = 0; // (bounds of a batch that ended in an error are void)
    if (s->pending_refactor && !partitioned(s)) TRY(refactor_now(s)); // (replicated: all columns are here now)
    if (s->pending_refactor && !s->comm && !s->in_lockstep)
        return fail(DZG_E_ARG, "a non-slack starting basis of a partitioned column-sharded solver is "
                               "factorised by dzg_shard_run / dzg_shard_run_lockstep (the ranks exchange "
                               "their basic columns): a host that drives the phases itself starts from "
                               "the slack basis or uses replicate_matrix");
    TRY(read_ctl(s));
    DzgCtl *h = s->h_ctl;
    // pending etas as the device counts them: a host that drives the phase entry points itself and
    // enqueued past a stop has a flush counter ahead of the device's (the flush kernels fold a FULL
    // file only), and would otherwise let neta reach DZG_RMAX with no flush pending
    s->since_flush = h->neta;
    if (h->status != DZG_RUNNING && h->status != DZG_ITER_LIMIT && h->status != DZG_NEAR_TIE)
        return h->status;
    long long stop = s->opts.max_iter;
    if (max_new_iters > 0 && h->iter + max_new_iters < stop) stop = h->iter + max_new_iters;
    if (h->status == DZG_ITER_LIMIT && h->iter >= s->opts.max_iter) return h->status;
    if (h->status == DZG_NEAR_TIE) {
        // resuming acknowledges the tie: the pivot in question is decided as FAST sees it
        h->tie_skip_iter = h->iter;
        HIP_OK(hipMemcpyAsync(&s->d.ctl->tie_skip_iter, &h->tie_skip_iter, sizeof(long long),
                              hipMemcpyHostToDevice, s->st));
    }
    h->status = DZG_RUNNING;
    h->iter_stop = stop;
    // only these words change; kernels are idle between runs
    HIP_OK(hipMemcpyAsync(&s->d.ctl->status, &h->status, sizeof(int), hipMemcpyHostToDevice, s->st));
    HIP_OK(hipMemcpyAsync(&s->d.ctl->iter_stop, &h->iter_stop, sizeof(long long), hipMemcpyHostToDevice, s->st));
    HIP_OK(hipStreamSynchronize(s->st));
    return DZG_RUNNING;
}

extern "C" int dzg_solver_set_budget(dzg_solver *s, int64_t max_new_iters)
{
    if (!s) return fail(DZG_E_ARG, "solver is NULL");
    HIP_OK(hipSetDevice(s->opts.device));
    return set_budget(s, max_new_iters);
}

extern "C" int dzg_solver_set_profile(dzg_solver *s, int32_t mask)
{
    if (!s) return fail(DZG_E_ARG, "solver is NULL");
    if (mask != 0 && s->ev.empty())
        return fail(DZG_E_ARG, "kernel timing needs opts.profile != 0 at creation (it reserves the events)");
    s->opts.profile = mask;
    return 0;
}

extern "C" int dzg_solver_poll(dzg_solver *s, int32_t *status, int64_t *iterations)
{
    if (!s) return fail(DZG_E_ARG, "solver is NULL");
    HIP_OK(hipSetDevice(s->opts.device));
    TRY(read_ctl(s));
    HIP_OK(hipGetLastError());
    if (status) *status = s->h_ctl->status;
    if (iterations) *iterations = s->h_ctl->iter;
    return 0;
}

// ---- column sharding: three enqueue-only phases per iteration (dantzig_amd.h) ---------
extern "C" int64_t dzg_shard_record_doubles(const dzg_solver *s) { return s ? s->d.xstride : 0; }

// event stamps of the phase path: slot = s->prof_slot (< 0: none)
static void phase_stamp(dzg_solver *s, int cls, int end)
{
    if (prof_on(s, cls, s->prof_slot))
        hipEventRecord(s->ev[((size_t)s->prof_slot * DZG_K_COUNT + cls) * 2 + end], s->st);
}

extern "C" int dzg_shard_phase1(dzg_solver *s, double *send_dev)
{
    if (!s || !send_dev || s->numerics != DZG_NUMERICS_FAST) return fail(DZG_E_ARG, "phase1");
    phase_stamp(s, DZG_K_STATUS, 0);
    if (s->d.rs)
        dzg_launch_rs_propose(s->d, 0, 0, send_dev, s->st);
    else
        dzg_launch_shard_propose(s->d, 0, 0, send_dev, s->st);
    phase_stamp(s, DZG_K_STATUS, 1);
    return 0;
}

extern "C" int dzg_shard_phase2(dzg_solver *s, const double *recv_dev, double *send_dev)
{
    if (!s || !recv_dev || !send_dev || s->numerics != DZG_NUMERICS_FAST)
        return fail(DZG_E_ARG, "phase2");
    const DzgDev &d = s->d;
    hipStream_t st = s->st;
    const int pk = price_kernel_for(s);
    if (d.rs) { // the basis side is row-sharded: k_rowshard.hip
        s->rs_recv1 = recv_dev;
        phase_stamp(s, DZG_K_FTRAN, 0);
        dzg_launch_rs_select(d, 0, recv_dev, st);                       // merge + status() (+ prep / v)
        dzg_launch_rs_gemv(d, DZG_STEP_PRIMAL, recv_dev, nullptr, st); // primal: FTRAN, own rows
        phase_stamp(s, DZG_K_FTRAN, 1);
        phase_stamp(s, DZG_K_PRICE, 0);
        dzg_launch_price_fast(d, pk, st, DZG_STEP_DUAL);                // dual: pricing, own columns
        phase_stamp(s, DZG_K_PRICE, 1);
        phase_stamp(s, DZG_K_RATIO, 0);
        dzg_launch_rs_propose(d, 1, price_partials_for(s, pk), send_dev, st);
        phase_stamp(s, DZG_K_RATIO, 1);
        return 0;
    }
    phase_stamp(s, DZG_K_FTRAN, 0);
    if (s->batch_chain) { // the three kernels below in one launch, k_chain.hip
        dzg_launch_chain_pre(d, s->chain_grid, s->chain_bar, s->chain_dbg, recv_dev, st);
    } else {
        dzg_launch_fast_select_prep(d, 4, 0, recv_dev, st);    // merge + status() + primal FTRAN prep
        dzg_launch_fast_gemv(d, DZG_STEP_PRIMAL, recv_dev, st);
        dzg_launch_fast_btran(d, st);
    }
    phase_stamp(s, DZG_K_FTRAN, 1);
    phase_stamp(s, DZG_K_PRICE, 0);
    dzg_launch_price_fast(d, pk, st);                      // owned columns only
    phase_stamp(s, DZG_K_PRICE, 1);
    phase_stamp(s, DZG_K_RATIO, 0);
    dzg_launch_shard_propose(d, 1, price_partials_for(s, pk), send_dev, st);
    phase_stamp(s, DZG_K_RATIO, 1);
    return 0;
}

extern "C" int dzg_shard_phase3(dzg_solver *s, const double *recv_dev)
{
    if (!s || !recv_dev || s->numerics != DZG_NUMERICS_FAST) return fail(DZG_E_ARG, "phase3");
    const DzgDev &d = s->d;
    hipStream_t st = s->st;
    phase_stamp(s, DZG_K_UPDATE, 0);
    if (d.rs) {
        if (!s->rs_recv1) return fail(DZG_E_ARG, "phase3 before phase2");
        if (s->rs_recv1 == recv_dev)
            return fail(DZG_E_ARG, "shard_rows: phase 3 still reads the records of exchange 1 (the leaving "
                                   "row of a dual step, the entering column of a primal one): the second "
                                   "exchange needs a receive buffer of its own");
        dzg_launch_rs_select(d, 1, recv_dev, st);                          // merge (+ prep / v)
        dzg_launch_price_fast(d, price_kernel_for(s), st, DZG_STEP_PRIMAL); // primal: pricing
        dzg_launch_rs_gemv(d, DZG_STEP_DUAL, s->rs_recv1, recv_dev, st);   // dual: FTRAN + dx_p
        dzg_launch_rs_books(d, s->rs_recv1, st);
        dzg_launch_fast_update(d, 0, st);
    } else if (s->batch_chain) {
        dzg_launch_chain_post(d, s->chain_grid, s->chain_bar, s->chain_dbg, 0, 0, recv_dev, st);
    } else {
        dzg_launch_fast_select_prep(d, 5, 0, recv_dev, st);    // merge + (dual) FTRAN prep
        dzg_launch_fast_gemv(d, DZG_STEP_DUAL, recv_dev, st); // + the pivot's books
        dzg_launch_fast_update(d, 0, st);
    }
    if (++s->since_flush >= DZG_RMAX) {
        dzg_launch_fast_flush(d, st);
        s->since_flush = 0;
    }
    phase_stamp(s, DZG_K_UPDATE, 1);
    return 0;
}

// Row-sharded ranks (opts.shard_rows): a record carries a row of the compact inverse, k entries
// wide; the batch being enqueued sends what its bound on k needs (k grows by at most one per
// pivot), not the room for m entries the buffers hold.  0: back to the largest record.
static void rs_batch_stride(dzg_solver *s, int k_bound)
{
    if (!s->d.rs) return;
    long long krow = ((long long)s->d.m + 2 + 15) / 16 * 16;
    if (k_bound > 0) {
        const long long need = ((long long)k_bound + 2 + 15) / 16 * 16;
        if (need < krow) krow = need;
    }
    s->d.xstride = DZG_RS_COL + s->d.rs_mcol + krow;
}

// x and xbar live on their rows' owners while a row-sharded solve runs: every rank gets all of them
// when a run returns (RCCL: one all-gather of the two slices)
static int rs_gather_state(dzg_solver *s)
{
    if (!s->d.rs) return 0;
    Rccl &r = rccl();
    dzg_launch_rs_pack(s->d, s->rs_slice, s->rs_gsend, s->st);
    if (r.AllGather(s->rs_gsend, s->rs_grecv, (size_t)2 * s->rs_slice, kNcclFloat64, s->comm, s->st) != 0)
        return fail(DZG_E_DEVICE, "ncclAllGather (x, xbar of a row-sharded solve)");
    dzg_launch_rs_unpack(s->d, s->rs_slice, s->rs_grecv, s->st);
    HIP_OK(hipStreamSynchronize(s->st));
    return 0;
}

static void shard_comm_destroy(dzg_solver *s)
{
    if (s->comm && rccl().ok) rccl().CommDestroy(s->comm);
    s->comm = nullptr;
}

extern "C" void *dzg_solver_stream(dzg_solver *s) { return s ? (void *)s->st : nullptr; }

extern "C" int dzg_comm_unique_id(void *unique_id_128)
{
    if (!unique_id_128) return fail(DZG_E_ARG, "unique_id is NULL");
    Rccl &r = rccl();
    if (!r.ok) return fail(DZG_E_DEVICE, "librccl could not be loaded");
    NcclUniqueId id;
    const int rc = r.GetUniqueId(&id);
    if (rc != 0) return fail(DZG_E_DEVICE, std::string("ncclGetUniqueId: ") + (r.GetErrorString ? r.GetErrorString(rc) : "?"));
    std::memcpy(unique_id_128, &id, sizeof(id));
    return 0;
}

static int shard_buffers(dzg_solver *s)
{
    if (s->xsend) return 0;
    const size_t n = (size_t)(s->xstride_max ? s->xstride_max : s->d.xstride);
    if (s->d.rs) {
        TRY(dev_alloc(s, &s->rs_gsend, (size_t)2 * s->rs_slice));
        TRY(dev_alloc(s, &s->rs_grecv, (size_t)2 * s->rs_slice * (size_t)s->d.world));
    }
    TRY(dev_alloc(s, &s->xsend, n));
    TRY(dev_alloc(s, &s->xrecv1, n * (size_t)s->d.world));
    TRY(dev_alloc(s, &s->xrecv2, n * (size_t)s->d.world));
    HIP_OK(hipMemsetAsync(s->xsend, 0, sizeof(double) * n, s->st));
    HIP_OK(hipMemsetAsync(s->xrecv1, 0, sizeof(double) * n * s->d.world, s->st));
    HIP_OK(hipMemsetAsync(s->xrecv2, 0, sizeof(double) * n * s->d.world, s->st));
    return 0;
}

extern "C" int dzg_shard_comm_init(dzg_solver *s, const void *unique_id_128)
{
    if (!s || !unique_id_128 || s->d.world < 1) return fail(DZG_E_ARG, "comm_init");
    HIP_OK(hipSetDevice(s->opts.device));
    Rccl &r = rccl();
    if (!r.ok) return fail(DZG_E_DEVICE, "librccl could not be loaded");
    NcclUniqueId id;
    std::memcpy(&id, unique_id_128, sizeof(id));
    const int rc = r.CommInitRank(&s->comm, s->d.world, id, s->d.rank);
    if (rc != 0) return fail(DZG_E_DEVICE, std::string("ncclCommInitRank: ") + (r.GetErrorString ? r.GetErrorString(rc) : "?"));
    return shard_buffers(s);
}

extern "C" int dzg_shard_comm_size(dzg_solver *s)
{
    if (!s || !s->comm) return fail(DZG_E_ARG, "dzg_shard_comm_init first");
    Rccl &r = rccl();
    int n = 0;
    if (!r.CommCount || r.CommCount(s->comm, &n) != 0) return fail(DZG_E_DEVICE, "ncclCommCount");
    return n;
}

extern "C" int dzg_shard_run(dzg_solver *s, int64_t max_new_iters)
{
    if (!s || !s->comm) return fail(DZG_E_ARG, "dzg_shard_comm_init first");
    HIP_OK(hipSetDevice(s->opts.device));
    {
        const int rc0 = set_budget(s, max_new_iters);
        if (rc0 != DZG_RUNNING) return rc0;
    }
    Rccl &r = rccl();
    size_t n = (size_t)s->d.xstride;
    auto t0 = std::chrono::steady_clock::now();
    if (s->pending_refactor) TRY(refactor_now(s)); // a warm start: the starting basis, all ranks together
    for (;;) {
        bool spent = false;
        TRY(budget_spent(s, &spent));
        if (spent) break;
        if (s->opts.refactor_interval > 0 && s->since_refactor >= s->opts.refactor_interval)
            TRY(refactor_now(s));
        const long long before = s->h_ctl->iter;
        const int batch = batch_size(s);
        s->since_flush = s->h_ctl->neta;
        s->since_refactor += batch;
        s->batch_chain = s->chain_bar && (long long)s->h_ctl->ncompact + batch <= s->chain_kcap;
        s->d.price_cols_hint = (int)s->h_ctl->nb_struct + batch;
        s->d.k_hint = (int)s->h_ctl->ncompact + batch; // bounds on k for the launches of this batch
        s->d.k_lo_hint = (int)s->h_ctl->ncompact > batch ? (int)s->h_ctl->ncompact - batch : 0;
        rs_batch_stride(s, s->d.k_hint);
        n = (size_t)s->d.xstride;
        for (int b = 0; b < batch; ++b) {
            s->prof_slot = s->opts.profile ? b : -1;
            TRY(dzg_shard_phase1(s, s->xsend));
            phase_stamp(s, DZG_K_XCHG1, 0);
            if (r.AllGather(s->xsend, s->xrecv1, n, kNcclFloat64, s->comm, s->st) != 0)
                return fail(DZG_E_DEVICE, "ncclAllGather (exchange 1)");
            phase_stamp(s, DZG_K_XCHG1, 1);
            TRY(dzg_shard_phase2(s, s->xrecv1, s->xsend));
            phase_stamp(s, DZG_K_XCHG2, 0);
            if (r.AllGather(s->xsend, s->xrecv2, n, kNcclFloat64, s->comm, s->st) != 0)
                return fail(DZG_E_DEVICE, "ncclAllGather (exchange 2)");
            phase_stamp(s, DZG_K_XCHG2, 1);
            TRY(dzg_shard_phase3(s, s->xrecv2));
        }
        s->prof_slot = -1;
        s->d.k_hint = s->d.k_lo_hint = 0;
        rs_batch_stride(s, 0);
        TRY(read_ctl(s));
        HIP_OK(hipGetLastError());
        if (s->h_ctl->bar_timeout)
            return fail(DZG_E_DEVICE, "a device-wide barrier of the fused phase kernels timed out: their "
                                      "workgroups were not all resident; opts.seven_launches = 1 runs "
                                      "without barriers");
        collect_profile(s, (int)(s->h_ctl->iter - before));
        if (s->h_ctl->status != DZG_RUNNING) break;
        bool stop = false;
        TRY(health_check(s, &stop));
        if (stop) break;
    }
    TRY(rs_gather_state(s));
    s->solve_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return s->h_ctl->status;
}

extern "C" int dzg_shard_run_lockstep(dzg_solver **sv, int32_t world, int64_t max_new_iters)
{
    if (!sv || world < 1) return fail(DZG_E_ARG, "lockstep");
    for (int r = 0; r < world; ++r) {
        if (!sv[r] || sv[r]->d.world != world || sv[r]->d.rank != r || sv[r]->st != sv[0]->st ||
            sv[r]->d.rs != sv[0]->d.rs)
            return fail(DZG_E_ARG, "lockstep: solvers must be ranks 0..world-1 on one stream");
        TRY(shard_buffers(sv[r]));
    }
    struct Flag { // (the ranks refactorise together inside this call, and only here)
        dzg_solver **sv; int world;
        ~Flag() { for (int r = 0; r < world; ++r) sv[r]->in_lockstep = false; }
    } flag{sv, world};
    for (int r = 0; r < world; ++r) sv[r]->in_lockstep = true;
    for (int r = 0; r < world; ++r) {
        const int rc0 = set_budget(sv[r], max_new_iters);
        if (rc0 != DZG_RUNNING) return rc0;
    }
    if (sv[0]->pending_refactor) TRY(refactor_lockstep(sv, world)); // a warm start: the starting basis
    hipStream_t st = sv[0]->st;
    // the all-gather of the lockstep harness: ONE kernel copies every rank's record into every
    // rank's receive buffer (world^2 device copies per exchange would cost more than the ranks'
    // own kernels and blur the per-rank compute this harness is profiled for)
    std::vector<double *> ptrs((size_t)3 * world);
    for (int r = 0; r < world; ++r) {
        ptrs[(size_t)r] = sv[r]->xsend;
        ptrs[(size_t)world + r] = sv[r]->xrecv1;
        ptrs[(size_t)2 * world + r] = sv[r]->xrecv2;
    }
    double **dptrs = nullptr;
    HIP_OK(hipMalloc(&dptrs, sizeof(double *) * ptrs.size()));
    struct Free { double **p; ~Free() { hipFree(p); } } free_ptrs{dptrs};
    HIP_OK(hipMemcpy(dptrs, ptrs.data(), sizeof(double *) * ptrs.size(), hipMemcpyHostToDevice));
    auto exchange = [&](bool second) -> int {
        dzg_launch_lockstep_allgather(dptrs, world, second ? 2 : 1, sv[0]->d.xstride, st);
        return 0;
    };
    for (;;) {
        bool spent = false;
        for (int r = 0; r < world; ++r) TRY(budget_spent(sv[r], &spent));
        if (spent) break;
        if (sv[0]->opts.refactor_interval > 0 && sv[0]->since_refactor >= sv[0]->opts.refactor_interval)
            TRY(refactor_lockstep(sv, world));
        const int batch = batch_size(sv[0]);
        for (int r = 0; r < world; ++r) {
            sv[r]->since_flush = sv[r]->h_ctl->neta;
            sv[r]->since_refactor += batch;
            sv[r]->batch_chain =
                sv[r]->chain_bar && (long long)sv[r]->h_ctl->ncompact + batch <= sv[r]->chain_kcap;
            sv[r]->d.price_cols_hint = (int)sv[r]->h_ctl->nb_struct + batch;
            sv[r]->d.k_hint = (int)sv[r]->h_ctl->ncompact + batch;
            sv[r]->d.k_lo_hint = (int)sv[r]->h_ctl->ncompact > batch ? (int)sv[r]->h_ctl->ncompact - batch : 0;
            rs_batch_stride(sv[r], sv[r]->d.k_hint);
        }
        for (int b = 0; b < batch; ++b) {
            for (int r = 0; r < world; ++r) TRY(dzg_shard_phase1(sv[r], sv[r]->xsend));
            TRY(exchange(false));
            for (int r = 0; r < world; ++r) TRY(dzg_shard_phase2(sv[r], sv[r]->xrecv1, sv[r]->xsend));
            TRY(exchange(true));
            for (int r = 0; r < world; ++r) TRY(dzg_shard_phase3(sv[r], sv[r]->xrecv2));
        }
        for (int r = 0; r < world; ++r) {
            sv[r]->d.k_hint = sv[r]->d.k_lo_hint = 0;
            rs_batch_stride(sv[r], 0);
        }
        for (int r = 0; r < world; ++r) TRY(read_ctl(sv[r]));
        HIP_OK(hipGetLastError());
        for (int r = 0; r < world; ++r)
            if (sv[r]->h_ctl->bar_timeout)
                return fail(DZG_E_DEVICE, "lockstep: a device-wide barrier timed out");
        for (int r = 1; r < world; ++r)
            if (sv[r]->h_ctl->status != sv[0]->h_ctl->status || sv[r]->h_ctl->iter != sv[0]->h_ctl->iter)
                return fail(DZG_E_DEVICE, "lockstep: ranks diverged");
        if (sv[0]->h_ctl->status != DZG_RUNNING) break;
        // the health rule reads replicated state: every rank decides alike, the group acts together
        HealthAction act = HEALTH_OK;
        for (int r = 0; r < world; ++r) {
            const HealthAction a = health_decide(sv[r]);
            if (r > 0 && a != act) return fail(DZG_E_DEVICE, "lockstep: ranks diverged (health rule)");
            act = a;
        }
        bool stop = false;
        if (act == HEALTH_REFACTOR) {
            const int rrc = refactor_lockstep(sv, world);
            if (rrc == DZG_E_DEVICE || rrc == DZG_E_NOMEM) return rrc;
            if (rrc != 0) act = HEALTH_GIVE_UP;
        }
        if (act == HEALTH_GIVE_UP)
            for (int r = 0; r < world; ++r) {
                bool one = false;
                TRY(health_give_up(sv[r], &one));
                stop = stop || one;
            }
        if (stop) break;
    }
    if (sv[0]->d.rs) { // x, xbar of every rank's rows into every rank's arrays
        std::vector<double *> xp((size_t)2 * world);
        for (int r = 0; r < world; ++r) {
            xp[(size_t)r] = sv[r]->d.x;
            xp[(size_t)world + r] = sv[r]->d.xbar;
        }
        double **dxp = nullptr;
        HIP_OK(hipMalloc(&dxp, sizeof(double *) * xp.size()));
        struct FreeX { double **p; ~FreeX() { hipFree(p); } } free_xp{dxp};
        HIP_OK(hipMemcpy(dxp, xp.data(), sizeof(double *) * xp.size(), hipMemcpyHostToDevice));
        dzg_launch_rs_lockstep_gather(dxp, world, sv[0]->d.m, sv[0]->rs_slice, st);
        HIP_OK(hipStreamSynchronize(st));
    }
    return sv[0]->h_ctl->status;
}

extern "C" int dzg_solver_run(dzg_solver *s, int64_t max_new_iters)
{
    if (!s) return fail(DZG_E_ARG, "solver is NULL");
    if (s->d.world > 1) return fail(DZG_E_ARG, "a sharded solver is driven through dzg_shard_run");
    HIP_OK(hipSetDevice(s->opts.device));
    {
        const int rc0 = set_budget(s, max_new_iters);
        if (rc0 != DZG_RUNNING) return rc0;
    }
    auto t0 = std::chrono::steady_clock::now();
    int rc = s->numerics == DZG_NUMERICS_FAST ? run_fast(s) : run_strict(s);
    auto t1 = std::chrono::steady_clock::now();
    s->solve_ms += std::chrono::duration<double, std::milli>(t1 - t0).count();
    if (rc != 0) return rc;
    return s->h_ctl->status;
}

// Test hook: the live-entry lists of a sparse-basis solver (k_price_csc_rl's input) against their
// definition -- column j lists exactly its stored entries whose row has a nonbasic slack
// (dslot >= 0), each once, with the stored value.
extern "C" int64_t dzg_debug_live_lists(dzg_solver *s, int64_t *entries)
{
    if (!s) return fail(DZG_E_ARG, "NULL argument");
    const DzgDev &d = s->d;
    if (!d.spb || !d.lcnt) return fail(DZG_E_ARG, "this solver keeps no live-entry lists");
    HIP_OK(hipSetDevice(s->opts.device));
    TRY(read_ctl(s));
    // a run that stopped between BTRAN and the pivot (DZG_NEAR_TIE) has that pivot's leaving slack
    // row listed already (ctl->rl_listed): by definition part of the lists until the pivot executes
    const int pending = s->h_ctl->rl_listed;
    const size_t ns = (size_t)d.ns, m = (size_t)d.m;
    std::vector<long long> cp(ns + 1);
    std::vector<int> dslot(m), lcnt(ns);
    HIP_OK(hipMemcpy(cp.data(), d.cptr, sizeof(long long) * (ns + 1), hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(dslot.data(), d.dslot, sizeof(int) * m, hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(lcnt.data(), d.lcnt, sizeof(int) * ns, hipMemcpyDeviceToHost));
    const size_t nnz = (size_t)cp[ns];
    std::vector<int> ri(nnz + 1);
    std::vector<double> cv(nnz + 1);
    std::vector<DzgLiveEntry> lent(nnz + 1);
    if (nnz) {
        HIP_OK(hipMemcpy(ri.data(), d.ridx, sizeof(int) * nnz, hipMemcpyDeviceToHost));
        HIP_OK(hipMemcpy(cv.data(), d.cval, sizeof(double) * nnz, hipMemcpyDeviceToHost));
        HIP_OK(hipMemcpy(lent.data(), d.lent, sizeof(DzgLiveEntry) * nnz, hipMemcpyDeviceToHost));
    }
    int64_t bad = 0, total = 0;
    std::vector<std::pair<int, double>> want, got;
    for (size_t j = 0; j < ns; ++j) {
        want.clear();
        got.clear();
        for (long long e = cp[j]; e < cp[j + 1]; ++e)
            if (dslot[(size_t)ri[(size_t)e]] >= 0 || ri[(size_t)e] == pending)
                want.emplace_back(ri[(size_t)e], cv[(size_t)e]);
        const long long span = cp[j + 1] - cp[j];
        if (lcnt[j] < 0 || lcnt[j] > span) {
            ++bad;
            continue;
        }
        for (int i = 0; i < lcnt[j]; ++i)
            got.emplace_back(lent[(size_t)cp[j] + (size_t)i].row, lent[(size_t)cp[j] + (size_t)i].val);
        std::sort(got.begin(), got.end());
        std::sort(want.begin(), want.end());
        if (got != want) ++bad;
        total += lcnt[j];
    }
    if (entries) *entries = total;
    return bad;
}

extern "C" int64_t dzg_debug_rl_listed(dzg_solver *s)
{
    if (!s) return fail(DZG_E_ARG, "NULL argument");
    if (!s->d.spb || !s->d.lcnt) return fail(DZG_E_ARG, "this solver keeps no live-entry lists");
    HIP_OK(hipSetDevice(s->opts.device));
    TRY(read_ctl(s));
    return s->h_ctl->rl_listed;
}

extern "C" int dzg_solver_result(dzg_solver *s, dzg_result *res)
{
    if (!s || !res) return fail(DZG_E_ARG, "NULL argument");
    HIP_OK(hipSetDevice(s->opts.device));
    TRY(read_ctl(s));
    const DzgDev &d = s->d;
    const int m = d.m, q = d.q;
    res->status = s->h_ctl->status;
    res->numerics_used = s->numerics;
    res->iterations = s->h_ctl->iter;
    std::vector<int> basis((size_t)(m ? m : 1)), nonbasis((size_t)(q ? q : 1));
    std::vector<double> x((size_t)(m ? m : 1));
    if (m) {
        HIP_OK(hipMemcpy(basis.data(), d.basis, sizeof(int) * m, hipMemcpyDeviceToHost));
        HIP_OK(hipMemcpy(x.data(), d.x, sizeof(double) * m, hipMemcpyDeviceToHost));
    }
    if (q) HIP_OK(hipMemcpy(nonbasis.data(), d.nonbasis, sizeof(int) * q, hipMemcpyDeviceToHost));
    // objective_value, src/simplex.rs:345-352, summed in basis-position order
    double sum = 0.0;
    for (int p = 0; p < m; ++p) {
        const double prod = s->c_host[(size_t)basis[p]] * x[p];
        sum = sum + prod;
    }
    res->objective = s->constant + sum;
    if (res->basis) for (int p = 0; p < m; ++p) res->basis[p] = basis[p];
    if (res->nonbasis) for (int k = 0; k < q; ++k) res->nonbasis[k] = nonbasis[k];
    if (res->x && m) std::memcpy(res->x, x.data(), sizeof(double) * m);
    if (res->xbar && m) HIP_OK(hipMemcpy(res->xbar, d.xbar, sizeof(double) * m, hipMemcpyDeviceToHost));
    if (res->z && q) HIP_OK(hipMemcpy(res->z, d.z, sizeof(double) * q, hipMemcpyDeviceToHost));
    if (res->zbar && q) HIP_OK(hipMemcpy(res->zbar, d.zbar, sizeof(double) * q, hipMemcpyDeviceToHost));
    if (res->log && res->log_cap > 0) {
        long long cnt = res->iterations < d.log_cap ? res->iterations : d.log_cap;
        if (cnt > res->log_cap) cnt = res->log_cap;
        if (cnt > 0) {
            std::vector<int> kind((size_t)cnt), en((size_t)cnt), le((size_t)cnt);
            std::vector<double> mu((size_t)cnt);
            HIP_OK(hipMemcpy(kind.data(), d.log_kind, sizeof(int) * cnt, hipMemcpyDeviceToHost));
            HIP_OK(hipMemcpy(en.data(), d.log_enter, sizeof(int) * cnt, hipMemcpyDeviceToHost));
            HIP_OK(hipMemcpy(le.data(), d.log_leave, sizeof(int) * cnt, hipMemcpyDeviceToHost));
            HIP_OK(hipMemcpy(mu.data(), d.log_mu, sizeof(double) * cnt, hipMemcpyDeviceToHost));
            for (long long i = 0; i < cnt; ++i) {
                res->log[i].kind = kind[(size_t)i];
                res->log[i].reserved = 0;
                res->log[i].entering = en[(size_t)i];
                res->log[i].leaving = le[(size_t)i];
                res->log[i].mu = mu[(size_t)i];
            }
        }
    }
    for (int c = 0; c < DZG_K_COUNT; ++c) {
        res->kernel_ms[c] = s->kernel_ms[c];
        res->kernel_launches[c] = s->kernel_launches[c];
    }
    res->price_bytes = s->h_ctl->price_bytes;
    if (s->d.rl_work) { // live-entry pricing counts the entries it walked itself: 16-byte records
        std::vector<unsigned long long> w(DZG_RL_WORK_SLOTS);
        HIP_OK(hipMemcpyAsync(w.data(), s->d.rl_work, sizeof(unsigned long long) * w.size(),
                              hipMemcpyDeviceToHost, s->st));
        HIP_OK(hipStreamSynchronize(s->st));
        unsigned long long walked = 0;
        for (unsigned long long x : w) walked += x;
        res->price_bytes += 16.0 * (double)walked;
    }
    res->solve_ms = s->solve_ms;
    res->max_pivot_error = s->max_err_life;
    res->near_ties = s->h_ctl->near_ties;
    res->first_near_tie = s->h_ctl->first_near_tie;
    res->min_margin = s->h_ctl->min_margin;
    res->dense_columns = s->numerics == DZG_NUMERICS_FAST ? s->h_ctl->ncompact : 0;
    res->refactors = s->refactors;
    res->chain_fallbacks = s->chain_fallbacks;
    res->price_pass_used = s->numerics == DZG_NUMERICS_FAST && !s->d.csc ? s->h_ctl->price_mask : 0;
    res->price_rows_copy = s->d.At ? 1 : 0;
    res->state_drift = s->state_drift;
    if (res->margins && res->log_cap > 0 && s->numerics == DZG_NUMERICS_FAST) {
        long long cnt = res->iterations < d.log_cap ? res->iterations : d.log_cap;
        if (cnt > res->log_cap) cnt = res->log_cap;
        if (cnt > 0)
            HIP_OK(hipMemcpy(res->margins, d.log_margin, sizeof(double) * cnt, hipMemcpyDeviceToHost));
    }
    return 0;
}

static int solve_once(const dzg_lp *lp, const dzg_opts *o, dzg_result *res)
{
    dzg_solver *s = nullptr;
    int rc = dzg_solver_create(lp, o, &s);
    if (rc != 0) return rc;
    rc = dzg_solver_run(s, 0);
    if (rc < 0) {
        dzg_solver_destroy(s);
        return rc;
    }
    const int rc2 = dzg_solver_result(s, res);
    dzg_solver_destroy(s);
    return rc2 != 0 ? rc2 : rc;
}

extern "C" int dzg_core_solve(const dzg_lp *lp, const dzg_opts *opts, dzg_result *res)
{
    if (!res) return fail(DZG_E_ARG, "res is NULL");
    if (!lp) return fail(DZG_E_ARG, "lp is NULL");
    dzg_opts o;
    if (opts) o = *opts; else dzg_opts_default(&o);
    const bool automatic = o.numerics == DZG_NUMERICS_AUTO;
    const int strict_rows = o.auto_strict_rows > 0 ? o.auto_strict_rows : 192;
    // AUTO above auto_strict_rows = FAST that must prove it followed the reference: up to
    // DZG_AUTO_STRICT_RESTART_ROWS rows it stops at the first decision that is within rounding of
    // a tie (and when it loses its footing: DZG_SINGULAR, DZG_PANIC -- degenerate or badly scaled
    // data), and the LP is solved again from the first pivot in the reference's own arithmetic;
    // the state vectors of a FAST run carry FAST's rounding, so no later point can be handed over
    // bit-exactly.  Above that size STRICT is out of reach and FAST reports what it met.
    const int restart_rows = o.auto_restart_rows != 0 ? o.auto_restart_rows : DZG_AUTO_STRICT_RESTART_ROWS;
    const bool can_restart = automatic && lp->m > strict_rows && lp->m <= restart_rows;
    if (can_restart && o.tie_tol >= 0.0) o.near_tie_action = DZG_NEAR_TIE_STOP;
    int rc = solve_once(lp, &o, res);
    if (rc < 0) return rc;
    if (can_restart && res->numerics_used == DZG_NUMERICS_FAST &&
        (rc == DZG_NEAR_TIE || rc == DZG_SINGULAR || rc == DZG_PANIC)) {
        dzg_opts strict = o;
        strict.numerics = DZG_NUMERICS_STRICT;
        strict.refactor_interval = 0;
        strict.near_tie_action = DZG_NEAR_TIE_COUNT;
        // The re-solve costs 3-57 ms per pivot: it runs against a wall-clock budget (an
        // integer model with many thousands of pivots would otherwise sit here for an hour without
        // a sign of life).  Out of budget: FAST from the first pivot with near ties COUNTED -- the
        // caller sees numerics_used = FAST and near_ties / first_near_tie, i.e. from which pivot on
        // the path is no longer certified to be the reference's.
        const double budget_s = o.auto_strict_budget_s < 0 ? -1.0
                                : (o.auto_strict_budget_s == 0 ? 600.0 : (double)o.auto_strict_budget_s);
        dzg_solver *ss = nullptr;
        rc = dzg_solver_create(lp, &strict, &ss);
        if (rc != 0) return rc;
        const auto t0 = std::chrono::steady_clock::now();
        bool out_of_time = false;
        for (;;) {
            rc = dzg_solver_run(ss, 256);
            // (the solver's own limit: dzg_solver_create normalised max_iter <= 0 in ITS copy only,
            // and a zero-filled dzg_opts from a C host must not read as "limit reached")
            if (rc != DZG_ITER_LIMIT || ss->h_ctl->iter >= ss->opts.max_iter) break;
            const double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            if (budget_s >= 0.0 && el > budget_s) { out_of_time = true; break; }
        }
        if (rc < 0) {
            dzg_solver_destroy(ss);
            return rc;
        }
        if (!out_of_time) {
            const int rc2 = dzg_solver_result(ss, res);
            dzg_solver_destroy(ss);
            return rc2 != 0 ? rc2 : rc;
        }
        dzg_solver_destroy(ss);
        dzg_opts fast = o;
        fast.numerics = DZG_NUMERICS_FAST;
        fast.near_tie_action = DZG_NEAR_TIE_COUNT;
        rc = solve_once(lp, &fast, res);
    }
    return rc;
}

// Level 1 on the fields of the reference's `Simplex` AS THEY ARE (src/simplex.rs:84-112): one
// CscMatrix over all n columns, slack columns included (src/linalg.rs:161-168; usize indices),
// b / n / x / z, the objective's coefficients and constant.  Simplex::solve becomes this one call.
// The unit columns that the device never stores are found here: a column with exactly one stored
// entry, equal to 1.0, is the slack of its row (scanned from the last column down, one per row --
// Simplex::new puts the slacks last; any further unit column of a row stays an ordinary column).
// The remaining columns go to the device dense or as CSC, whichever is smaller.
extern "C" int dzg_core_solve_full_csc(int64_t m, int64_t n, const int64_t *col_ptr,
                                       const int64_t *row_idx, const double *val, const double *c,
                                       double constant, int64_t *basis, int64_t *nonbasis, double *x,
                                       double *z, const dzg_opts *opts, dzg_result *res)
{
    if (!res) return fail(DZG_E_ARG, "res is NULL");
    if (m < 0 || n < m || !col_ptr || (n > 0 && !c) || (m > 0 && (!basis || !x)) ||
        (n > m && (!nonbasis || !z)))
        return fail(DZG_E_ARG, "full_csc: sizes / NULL argument");
    if (n >= (1ll << 31) - 64) return fail(DZG_E_ARG, "full_csc: index range");
    if (col_ptr[0] != 0) return fail(DZG_E_ARG, "full_csc: col_ptr[0] != 0");
    for (int64_t j = 0; j < n; ++j) {
        if (col_ptr[j + 1] < col_ptr[j]) return fail(DZG_E_ARG, "full_csc: col_ptr not monotone");
        for (int64_t e = col_ptr[j]; e < col_ptr[j + 1]; ++e)
            if (!row_idx || !val || row_idx[e] < 0 || row_idx[e] >= m ||
                (e > col_ptr[j] && row_idx[e] <= row_idx[e - 1]))
                return fail(DZG_E_ARG, "full_csc: row_idx must ascend strictly inside a column");
    }
    std::vector<int64_t> var_col((size_t)(n ? n : 1), 0);
    std::vector<char> row_has_slack((size_t)(m ? m : 1), 0), is_slack((size_t)(n ? n : 1), 0);
    for (int64_t j = n - 1; j >= 0; --j) {
        const int64_t e = col_ptr[j];
        if (col_ptr[j + 1] - e == 1 && val[e] == 1.0 && !row_has_slack[(size_t)row_idx[e]]) {
            row_has_slack[(size_t)row_idx[e]] = 1;
            is_slack[(size_t)j] = 1;
            var_col[(size_t)j] = -1 - row_idx[e];
        }
    }
    int64_t ns = 0, nnz = 0;
    for (int64_t j = 0; j < n; ++j)
        if (!is_slack[(size_t)j]) {
            var_col[(size_t)j] = ns++;
            for (int64_t e = col_ptr[j]; e < col_ptr[j + 1]; ++e) nnz += val[e] != 0.0;
        }
    dzg_lp lp;
    std::memset(&lp, 0, sizeof(lp));
    lp.m = m; lp.n = n; lp.n_struct = ns;
    lp.var_col = var_col.data();
    lp.c = c; lp.constant = constant;
    lp.basis = basis; lp.nonbasis = nonbasis; lp.x = x; lp.z = z;
    // dense (8 m ns bytes) or CSC (12 bytes per entry + the CSR copy of the sparse-basis path)
    const bool dense = ns > 0 && (double)nnz * 24.0 >= (double)m * (double)ns * 8.0;
    std::vector<double> a, sval;
    std::vector<int64_t> scp;
    std::vector<int32_t> sri;
    if (dense) {
        a.assign((size_t)m * (size_t)ns, 0.0);
        for (int64_t j = 0; j < n; ++j)
            if (!is_slack[(size_t)j])
                for (int64_t e = col_ptr[j]; e < col_ptr[j + 1]; ++e)
                    a[(size_t)var_col[(size_t)j] * (size_t)m + (size_t)row_idx[e]] = val[e];
        lp.a = a.data();
        lp.lda = m;
    } else if (ns > 0) {
        scp.assign((size_t)ns + 1, 0);
        sri.reserve((size_t)nnz);
        sval.reserve((size_t)nnz);
        for (int64_t j = 0; j < n; ++j)
            if (!is_slack[(size_t)j]) {
                for (int64_t e = col_ptr[j]; e < col_ptr[j + 1]; ++e)
                    if (val[e] != 0.0) { // (the reference's CSC holds no explicit zeros, src/linalg.rs:261)
                        sri.push_back((int32_t)row_idx[e]);
                        sval.push_back(val[e]);
                    }
                scp[(size_t)var_col[(size_t)j] + 1] = (int64_t)sri.size();
            }
        if (sri.empty()) { sri.push_back(0); sval.push_back(0.0); }
        lp.col_ptr = scp.data();
        lp.row_idx = sri.data();
        lp.val = sval.data();
    }
    // the final state lands in the caller's b / n / x / z, as Simplex::solve leaves it in `self`
    const int64_t q = n - m;
    std::vector<int64_t> ob, on;
    std::vector<double> ox, oz;
    dzg_result r = *res;
    if (!r.basis) { ob.resize((size_t)(m ? m : 1)); r.basis = ob.data(); }
    if (!r.nonbasis) { on.resize((size_t)(q ? q : 1)); r.nonbasis = on.data(); }
    if (!r.x) { ox.resize((size_t)(m ? m : 1)); r.x = ox.data(); }
    if (!r.z) { oz.resize((size_t)(q ? q : 1)); r.z = oz.data(); }
    const int rc = dzg_core_solve(&lp, opts, &r);
    if (rc < 0) return rc;
    for (int64_t p = 0; p < m; ++p) { basis[p] = r.basis[p]; x[p] = r.x[p]; }
    for (int64_t k = 0; k < q; ++k) { nonbasis[k] = r.nonbasis[k]; z[k] = r.z[k]; }
    if (!res->basis) r.basis = nullptr;
    if (!res->nonbasis) r.nonbasis = nullptr;
    if (!res->x) r.x = nullptr;
    if (!res->z) r.z = nullptr;
    *res = r;
    return rc;
}

// ---- single-function entry points for parity tests -------------------------------
extern "C" int dzg_kernel_lu_solve(int64_t n, const double *a, const double *b, double *x_out,
                                   double *lu_out, int64_t *p_out, int32_t device)
{
    if (n <= 0 || !a || !b || !x_out) return fail(DZG_E_ARG, "bad argument");
    if (dzg_device_count() <= 0) return fail(DZG_E_DEVICE, "no HIP device visible");
    HIP_OK(hipSetDevice(device));
    DzgLu w{};
    dzg_lu_layout((int)n, &w);
    double *xo;
    DzgCtl *ctl;
    HIP_OK(hipMalloc(&w.W, sizeof(double) * (size_t)n * (size_t)w.ldw));
    HIP_OK(hipMalloc(&w.P0, sizeof(double) * (size_t)n * DZG_LU_LDP));
    HIP_OK(hipMalloc(&w.P1, sizeof(double) * (size_t)n * DZG_LU_LDP));
    HIP_OK(hipMalloc(&w.piv, sizeof(int) * n));
    HIP_OK(hipMalloc(&w.pz, sizeof(int) * n));
    HIP_OK(hipMalloc(&w.ptab, sizeof(int) * (size_t)n * DZG_LU_NB));
    HIP_OK(hipMalloc(&w.part_r, sizeof(double) * 2 * (size_t)w.nparts));
    HIP_OK(hipMalloc(&w.part_k, sizeof(int) * 2 * (size_t)w.nparts));
    HIP_OK(hipMalloc(&xo, sizeof(double) * n));
    HIP_OK(hipMalloc(&ctl, sizeof(DzgCtl)));
    DzgCtl c0;
    std::memset(&c0, 0, sizeof(c0));
    c0.status = DZG_RUNNING;
    HIP_OK(hipMemcpy(ctl, &c0, sizeof(c0), hipMemcpyHostToDevice));
    const size_t pitch = sizeof(double) * (size_t)w.ldw, rowb = sizeof(double) * (size_t)n;
    HIP_OK(hipMemcpy2D(w.W, pitch, a, rowb, rowb, (size_t)n, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy2D(w.W + n, pitch, b, sizeof(double), sizeof(double), (size_t)n,
                       hipMemcpyHostToDevice)); // the right-hand side is column n
    HIP_OK(hipMemset(w.piv, 0, sizeof(int) * n));
    dzg_launch_lu_raw(w, ctl, xo, 0);
    HIP_OK(hipDeviceSynchronize());
    HIP_OK(hipGetLastError());
    HIP_OK(hipMemcpy(x_out, xo, sizeof(double) * n, hipMemcpyDeviceToHost));
    if (lu_out) HIP_OK(hipMemcpy2D(lu_out, rowb, w.W, pitch, rowb, (size_t)n, hipMemcpyDeviceToHost));
    if (p_out && n > 1) {
        std::vector<int> p((size_t)n);
        HIP_OK(hipMemcpy(p.data(), w.piv, sizeof(int) * n, hipMemcpyDeviceToHost));
        for (int64_t k = 0; k + 1 < n; ++k) p_out[k] = p[(size_t)k];
    }
    hipFree(w.W); hipFree(w.P0); hipFree(w.P1); hipFree(w.piv); hipFree(w.pz); hipFree(w.ptab);
    hipFree(w.part_r); hipFree(w.part_k); hipFree(xo); hipFree(ctl);
    return 0;
}

extern "C" int dzg_kernel_neg_t_dot(int64_t m, int64_t n_struct, const double *a, int64_t lda,
                                    const int64_t *cols, int64_t ncols, const double *v,
                                    double *out, int32_t kernel, int32_t device)
{
    if (m <= 0 || n_struct < 0 || ncols < 0 || !v || !out || (ncols > 0 && !cols) ||
        (n_struct > 0 && (!a || lda < m)))
        return fail(DZG_E_ARG, "bad argument");
    if (dzg_device_count() <= 0) return fail(DZG_E_DEVICE, "no HIP device visible");
    HIP_OK(hipSetDevice(device));
    const long long ldd = (m + 15) / 16 * 16;
    const size_t abytes = sizeof(double) * (size_t)ldd * (size_t)(n_struct ? n_struct : 1);
    double *dA, *dv, *dout;
    int *dcols;
    HIP_OK(hipMalloc(&dA, abytes));
    HIP_OK(hipMalloc(&dv, sizeof(double) * ((size_t)m + 2)));
    HIP_OK(hipMalloc(&dout, sizeof(double) * (size_t)(ncols ? ncols : 1)));
    HIP_OK(hipMalloc(&dcols, sizeof(int) * (size_t)(ncols ? ncols : 1)));
    HIP_OK(hipMemset(dA, 0, abytes));
    HIP_OK(hipMemset(dv, 0, sizeof(double) * ((size_t)m + 2)));
    if (n_struct > 0)
        HIP_OK(hipMemcpy2D(dA, sizeof(double) * ldd, a, sizeof(double) * lda, sizeof(double) * m,
                           n_struct, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(dv, v, sizeof(double) * m, hipMemcpyHostToDevice));
    std::vector<int> c32((size_t)(ncols ? ncols : 1));
    for (int64_t k = 0; k < ncols; ++k) {
        if (cols[k] >= n_struct || cols[k] < -m) return fail(DZG_E_ARG, "cols out of range");
        c32[(size_t)k] = (int)cols[k];
    }
    HIP_OK(hipMemcpy(dcols, c32.data(), sizeof(int) * (size_t)(ncols ? ncols : 1), hipMemcpyHostToDevice));
    dzg_launch_price_raw(kernel, (int)m, ldd, dA, dcols, (int)ncols, dv, dout, 0);
    HIP_OK(hipDeviceSynchronize());
    HIP_OK(hipGetLastError());
    if (ncols > 0) HIP_OK(hipMemcpy(out, dout, sizeof(double) * ncols, hipMemcpyDeviceToHost));
    hipFree(dA); hipFree(dv); hipFree(dout); hipFree(dcols);
    return 0;
}

extern "C" int dzg_kernel_neg_t_dot_csc(int64_t m, int64_t n_struct, const int64_t *col_ptr,
                                        const int32_t *row_idx, const double *val,
                                        const int64_t *cols, int64_t ncols, const double *v,
                                        double *out, int32_t device)
{
    if (m <= 0 || n_struct < 0 || ncols < 0 || !v || !out || (ncols > 0 && !cols) || !col_ptr ||
        col_ptr[0] != 0 || (n_struct > 0 && col_ptr[n_struct] > 0 && (!row_idx || !val)))
        return fail(DZG_E_ARG, "bad argument");
    if (dzg_device_count() <= 0) return fail(DZG_E_DEVICE, "no HIP device visible");
    HIP_OK(hipSetDevice(device));
    const size_t nnz = (size_t)col_ptr[n_struct];
    for (int64_t j = 0; j < n_struct; ++j)
        for (int64_t e = col_ptr[j]; e < col_ptr[j + 1]; ++e)
            if (row_idx[e] < 0 || row_idx[e] >= m) return fail(DZG_E_ARG, "row_idx out of range");
    std::vector<long long> cp((size_t)n_struct + 1);
    for (int64_t j = 0; j <= n_struct; ++j) cp[(size_t)j] = col_ptr[j];
    std::vector<int> c32((size_t)(ncols ? ncols : 1));
    for (int64_t k = 0; k < ncols; ++k) {
        if (cols[k] >= n_struct || cols[k] < -m) return fail(DZG_E_ARG, "cols out of range");
        c32[(size_t)k] = (int)cols[k];
    }
    long long *dcp; int *dri, *dcols; double *dcv, *dv, *dout;
    HIP_OK(hipMalloc(&dcp, sizeof(long long) * cp.size()));
    HIP_OK(hipMalloc(&dri, sizeof(int) * (nnz + 1)));
    HIP_OK(hipMalloc(&dcv, sizeof(double) * (nnz + 1)));
    HIP_OK(hipMalloc(&dv, sizeof(double) * ((size_t)m + 2)));
    HIP_OK(hipMalloc(&dout, sizeof(double) * c32.size()));
    HIP_OK(hipMalloc(&dcols, sizeof(int) * c32.size()));
    HIP_OK(hipMemcpy(dcp, cp.data(), sizeof(long long) * cp.size(), hipMemcpyHostToDevice));
    if (nnz) {
        HIP_OK(hipMemcpy(dri, row_idx, sizeof(int) * nnz, hipMemcpyHostToDevice));
        HIP_OK(hipMemcpy(dcv, val, sizeof(double) * nnz, hipMemcpyHostToDevice));
    }
    HIP_OK(hipMemset(dv, 0, sizeof(double) * ((size_t)m + 2)));
    HIP_OK(hipMemcpy(dv, v, sizeof(double) * m, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(dcols, c32.data(), sizeof(int) * c32.size(), hipMemcpyHostToDevice));
    dzg_launch_price_csc_raw(dcp, dri, dcv, dcols, (int)ncols, dv, dout, 0);
    HIP_OK(hipDeviceSynchronize());
    HIP_OK(hipGetLastError());
    if (ncols > 0) HIP_OK(hipMemcpy(out, dout, sizeof(double) * ncols, hipMemcpyDeviceToHost));
    hipFree(dcp); hipFree(dri); hipFree(dcv); hipFree(dv); hipFree(dout); hipFree(dcols);
    return 0;
}

extern "C" int dzg_kernel_first_pivot(int64_t len, const double *y, const double *ybar,
                                      int64_t *pos_out, int32_t device)
{
    if (len < 0 || !pos_out) return fail(DZG_E_ARG, "bad argument");
    if (dzg_device_count() <= 0) return fail(DZG_E_DEVICE, "no HIP device visible");
    HIP_OK(hipSetDevice(device));
    return dzg_run_first_pivot(len, y, ybar, pos_out);
}

extern "C" int dzg_kernel_second_pivot(int64_t len, double mu, const double *y, const double *ybar,
                                       const double *dy, int64_t *pos_out, int32_t device)
{
    if (len < 0 || !pos_out) return fail(DZG_E_ARG, "bad argument");
    if (dzg_device_count() <= 0) return fail(DZG_E_DEVICE, "no HIP device visible");
    HIP_OK(hipSetDevice(device));
    return dzg_run_second_pivot(len, mu, y, ybar, dy, pos_out);
}
