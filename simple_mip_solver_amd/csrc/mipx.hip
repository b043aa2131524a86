// mipx.hip -- host side of the C ABI declared in include/mipx.h (HIP runtime only: no ML framework,
// no BLAS).  One context = one GPU + one stream.  No CPU fallback exists: every entry point
// that computes requires a live HIP device.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "../../include/mipx.h"
#include "lp_kernel.hip.h"
#include "lp_kernel_big.hip.h"
#include "lp_kernel_root.hip.h"

struct mipx_ctx {
    int device = -1;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    hipEvent_t k0 = nullptr, k1 = nullptr;   // around the LP launch of the last host-buffer call (mipx_last_kernel_ms)
    float last_kernel_ms = -1.f;
    std::string err;
    // staging of the host-buffer cut entry points, grown on demand (a hipMalloc / hipFree pair per
    // call cost more than the kernels between them)
    void *scratch = nullptr;
    size_t scratch_bytes = 0;
};

struct mipx_problem {
    mipx_ctx *ctx = nullptr;
    int m = 0, n = 0;
    double *dA = nullptr, *db = nullptr, *dc = nullptr;
    // staging for the host-pointer entry point (grown on demand)
    void *scratch = nullptr;
    size_t scratch_bytes = 0;
    // debug dump buffers (device), enabled by mipx_debug_enable
    double *dbg_T = nullptr, *dbg_vec = nullptr;
    int32_t *dbg_idx = nullptr;
    // K1b (tableau streamed from HBM): one m x n slab per concurrently resident workgroup
    double *big_scratch = nullptr;
    int big_slabs = 0;
    double *big_scratch2 = nullptr;   // launches beside the main stream's (the engine's probes) stream their own slabs
    int big_slabs2 = 0;
    int big_rows = 0;                 // rows per slab (m, or m + cut rows once a launch carried them)
    char *root_state = nullptr;       // K1c (one cold LP over the chip): its buffers, allocated on first use
    size_t root_state_bytes = 0;
    // anchor tableau (mipx_problem_set_anchor): warm starts refactor from it
    double *anchor_T = nullptr, *anchor_vec = nullptr;
    int32_t *anchor_idx = nullptr;
    bool anchor_on = false;
};

namespace {

int fail(mipx_ctx *ctx, int code, const char *what, hipError_t e = hipSuccess) {
    if (ctx) {
        ctx->err = what;
        if (e != hipSuccess) {
            ctx->err += ": ";
            ctx->err += hipGetErrorString(e);
        }
    }
    return code;
}

#define HIP_TRY(ctx, call)                                        \
    do {                                                          \
        hipError_t e_ = (call);                                   \
        if (e_ != hipSuccess) return fail((ctx), MIPX_EHIP, #call, e_); \
    } while (0)

struct KernelCfg {
    int mp, np, threads;
    const char *name;
    void (*launch)(const mipx::LpArgs &, int grid, hipStream_t);
};

template <int NW, int R, int C, int MP>
void launch_cfg(const mipx::LpArgs &a, int grid, hipStream_t st) {
    // the dive variant carries the pass loop (a few % slower per node LP): only when asked for
    // the cut-row variant reads its rows through the per-node cut lists: only when a launch has any
    if (a.ncut) hipLaunchKernelGGL((mipx::lp_dual_simplex<NW, R, C, MP, false, true>), dim3(grid), dim3(64 * (NW + 1)), 0, st, a);
    else if (a.dive) hipLaunchKernelGGL((mipx::lp_dual_simplex<NW, R, C, MP, true>), dim3(grid), dim3(64 * (NW + 1)), 0, st, a);
    else hipLaunchKernelGGL((mipx::lp_dual_simplex<NW, R, C, MP, false>), dim3(grid), dim3(64 * (NW + 1)), 0, st, a);
}

// ordered by on-chip footprint; the first that fits (m <= mp, n <= np) is used
const KernelCfg kCfgs[] = {
    // <tableau waves, rows per thread, columns per thread> (+ one control wave); a tableau wave
    // is 4 row groups x 16 column lanes: rows <= 4 * waves * R, columns <= 16 * C
    {32, 64, 128, "lp_dual_simplex<1,8,4>", launch_cfg<1, 8, 4, 32>},
    {64, 128, 256, "lp_dual_simplex<3,6,8>", launch_cfg<3, 6, 8, 64>},
    {128, 256, 512, "lp_dual_simplex<7,5,16>", launch_cfg<7, 5, 16, 128>},
    {192, 256, 512, "lp_dual_simplex<7,7,16>", launch_cfg<7, 7, 16, 192>},
};

const KernelCfg *pick_cfg(int m, int n) {
    for (const KernelCfg &c : kCfgs)
        if (m <= c.mp && n <= c.np) return &c;
    return nullptr;
}

size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

bool big_fits(int m, int n) { return m >= 1 && m <= mipx::kBigMaxM && n <= mipx::kBigMaxN; }
bool shape_supported(int m, int n) { return pick_cfg(m, n) != nullptr || big_fits(m, n); }

// K1c: one cold node LP (no warm-start basis) above the register tiles, its rows dealt out to G workgroups, one
// launch per pivot (lp_kernel_root.hip.h).  The host queues the pivots in chunks and looks at the status word
// between them.  Returns MIPX_OK with *done = false where it does not apply.
int launch_root_coop(mipx_problem *p, mipx::LpArgs &a, hipStream_t stream, bool *done) {
    *done = false;
    mipx_ctx *ctx = p->ctx;
    const int m = p->m, n = p->n;
    // (LpArgs::cold: no basis and no cut row, whatever vstat_in / ncut point to; one node: a dump of "every node" is node 0's)
    if (a.batch != 1 || ((a.vstat_in != nullptr || a.ncut != nullptr) && !a.cold) || a.refactor_only || m < 64 || n > mipx::kBigMaxN ||
        a.A_stride != 0 || (getenv("MIPX_NO_COOP_ROOT") && atoi(getenv("MIPX_NO_COOP_ROOT"))))
        return MIPX_OK;
    int32_t src = 0;
    if (a.slot) {
        HIP_TRY(ctx, hipMemcpyAsync(&src, a.slot, 4, hipMemcpyDeviceToHost, stream));
        HIP_TRY(ctx, hipStreamSynchronize(stream));
    }
    // at most 256 workgroups (one candidate per thread of a workgroup), RPB = 1, 2, 4 or 8 rows of the tableau in
    // the registers of each
    int gmax = getenv("MIPX_ROOT_WG") ? atoi(getenv("MIPX_ROOT_WG")) : 256;
    gmax = gmax < 32 ? 32 : gmax > 256 ? 256 : gmax;
    int RPB = 1;
    while (RPB < 8 && (m + RPB - 1) / RPB > gmax) RPB *= 2;
    const int G = (m + RPB - 1) / RPB;
    if (G > 256) return MIPX_OK;
    const size_t nn = (size_t)n, mm = (size_t)m;
    size_t off = 0;
    auto carve = [&](size_t bytes) { size_t o = off; off = align_up(off + bytes, 256); return o; };
    const size_t o_T = carve(mm * nn * 8), o_b0 = carve(mm * 8), o_ba = carve(mm * 8), o_bb = carve(mm * 8),
                 o_w = carve(mm * 8), o_rl = carve(mm * 8), o_ru = carve(mm * 8), o_bv = carve(mm * 4), o_d = carve(2 * nn * 8), o_va = carve(2 * nn * 8),
                 o_vb = carve(2 * nn * 8), o_nv = carve(2 * nn * 4), o_sd = carve(2 * nn * 4),
                 o_ck = carve(2 * (size_t)G * sizeof(mipx::RootKey)), o_cr = carve(2 * (size_t)G * sizeof(mipx::RootRow)), o_row = carve(2 * (size_t)G * nn * 8),
                 o_ctl = carve(16 * 4);
    if (p->root_state_bytes < off) {   // (the size depends on the number of workgroups: MIPX_ROOT_WG may have changed)
        if (p->root_state) (void)hipFree(p->root_state);
        p->root_state = nullptr; p->root_state_bytes = 0;
        HIP_TRY(ctx, hipMalloc((void **)&p->root_state, off));
        p->root_state_bytes = off;
    }
    char *base = p->root_state;
    mipx::RootState S;
    S.m = m; S.n = n; S.G = G; S.RPB = RPB;
    S.T = (double *)(base + o_T); S.beta0 = (double *)(base + o_b0); S.ba = (double *)(base + o_ba);
    S.bb = (double *)(base + o_bb); S.wgt = (double *)(base + o_w); S.bvar = (int *)(base + o_bv);
    S.rlo = (double *)(base + o_rl); S.rup = (double *)(base + o_ru);
    S.d = (double *)(base + o_d); S.va = (double *)(base + o_va); S.vb = (double *)(base + o_vb);
    S.nvar = (int *)(base + o_nv); S.side = (int *)(base + o_sd);
    S.ckey = (mipx::RootKey *)(base + o_ck); S.crow = (mipx::RootRow *)(base + o_cr); S.rowbuf = (double *)(base + o_row); S.ctl = (int *)(base + o_ctl);
    S.lo = a.l + (size_t)src * nn; S.up = a.u + (size_t)src * nn;
    S.max_iter = a.max_iter; S.cap = 100 * (m + n) + 1000;
    hipLaunchKernelGGL(mipx::lp_root_init, dim3(G), dim3(mipx::kRootNT), 2 * nn * 8, stream, S, a.A, a.b, a.c);
    HIP_TRY(ctx, hipGetLastError());
    int par = 0;
    const int chunk = 192;
    void (*pivot)(mipx::RootState, int) = RPB == 1 ? mipx::lp_root_pivot<1> : RPB == 2 ? mipx::lp_root_pivot<2>
                                           : RPB == 4 ? mipx::lp_root_pivot<4> : mipx::lp_root_pivot<8>;
    const long limit = (long)S.cap + 8;
    for (long done_pivots = 0; done_pivots <= limit; done_pivots += chunk) {
        for (int k = 0; k < chunk; k++, par ^= 1)
            hipLaunchKernelGGL(pivot, dim3(G), dim3(mipx::kRootNT), 0, stream, S, par);
        HIP_TRY(ctx, hipGetLastError());
        int32_t st = -1;
        HIP_TRY(ctx, hipMemcpyAsync(&st, S.ctl + 8 * par, 4, hipMemcpyDeviceToHost, stream));
        HIP_TRY(ctx, hipStreamSynchronize(stream));
        if (st == -2) {   // a verdict is due on values that carry the running updates of many pivots: afresh from the tableau first
            hipLaunchKernelGGL(mipx::lp_root_values, dim3(G), dim3(mipx::kRootNT), 2 * nn * 8, stream, S, par);
            HIP_TRY(ctx, hipGetLastError());
            par ^= 1;
        }
        if (st >= 0) break;
    }
    hipLaunchKernelGGL(mipx::lp_root_final, dim3(1), dim3(mipx::kRootNT), nn * 8, stream, S, par, a, (size_t)0);
    HIP_TRY(ctx, hipGetLastError());
    *done = true;
    return MIPX_OK;
}

// Launch K1 (register-resident tableau) or, above its tiles, K1b (tableau streamed from HBM).
// m_rows: the largest row count of any node of the launch when nodes carry cut rows (a.ncut), so
// that the tile covers it (results do not depend on the tile); -1: the problem's own m.
int launch_lp_any(mipx_problem *p, mipx::LpArgs &a, int batch, hipStream_t stream = nullptr, int m_rows = -1) {
    mipx_ctx *ctx = p->ctx;
    if (!stream) stream = ctx->stream;
    if (m_rows < 0) m_rows = p->m;
    // (a launch with cut rows above the register tiles goes to K1b like any other shape above them)
    if (const KernelCfg *cfg = pick_cfg(m_rows, p->n)) {
#ifdef MIPX_KPROF
        // profiling build: per-section cycle totals of wave 0, summed over the launch
        static unsigned long long *d_prof = nullptr;
        if (!d_prof) HIP_TRY(ctx, hipMalloc((void **)&d_prof, 32 * 8));
        HIP_TRY(ctx, hipMemsetAsync(d_prof, 0, 32 * 8, stream));
        a.prof = d_prof;
        a.prof_wave = getenv("MIPX_KPROF_WAVE") ? atoi(getenv("MIPX_KPROF_WAVE")) : 0;
        cfg->launch(a, batch, stream);
        HIP_TRY(ctx, hipStreamSynchronize(stream));
        unsigned long long h[32];
        HIP_TRY(ctx, hipMemcpy(h, d_prof, sizeof h, hipMemcpyDeviceToHost));
        if (getenv("MIPX_KPROF_PRINT")) {
            const double it = h[16] ? (double)h[16] : 1.0, rf = h[17] ? (double)h[17] : 1.0;
            const double lps = h[18] ? (double)h[18] : 1.0;
            fprintf(stderr, "[kprof] %s batch %d wave %d  lps %llu iters %llu refactor pivots %llu\n"
                    "  simplex cycles/iter: m1 %.0f m3 %.0f m2 %.0f m4 %.0f m5 %.0f m0 %.0f  (sum %.0f)\n"
                    "  refactor cycles/pivot: m9 %.0f m10 %.0f m12 %.0f m13 %.0f m14 %.0f  (sum %.0f)\n"
                    "  per LP: setup %.0f m8/iter %.0f m11/iter %.0f value-init %.0f tail %.0f\n",
                    cfg->name, batch, a.prof_wave, h[18], h[16], h[17], h[1] / it, h[3] / it, h[2] / it,
                    h[4] / it, h[5] / it, h[0] / it, (h[0] + h[1] + h[2] + h[3] + h[4] + h[5]) / it,
                    h[9] / rf, h[10] / rf, h[12] / rf, h[13] / rf, h[14] / rf,
                    (h[9] + h[10] + h[12] + h[13] + h[14]) / rf, h[15] / lps, h[8] / it, h[11] / it,
                    h[7] / lps, h[6] / lps);
#ifdef MIPX_KPROF_RT
            fprintf(stderr, "  ratio test cycles/iter: wait at B %.0f pass1 %.0f argmin %.0f pass2 %.0f argmax %.0f tail %.0f  (m3 = the barrier C)\n",
                    h[8] / it, h[9] / it, h[10] / it, h[12] / it, h[13] / it, h[14] / it);
#endif
        }
        return MIPX_OK;
#else
        a.prof = nullptr;
        a.prof_wave = 0;
        cfg->launch(a, batch, stream);
        HIP_TRY(ctx, hipGetLastError());
        return MIPX_OK;
#endif
    }
    {   // one cold LP: over the whole chip instead of one CU
        bool done = false;
        const int rrc = launch_root_coop(p, a, stream, &done);
        if (rrc || done) return rrc;
    }
    // rows a node can have: the shared ones, or (cut rows) the rows allotted per node
    const int mcap = a.ncut ? a.mstride : p->m;
    if (mcap < m_rows || !big_fits(mcap, p->n)) return fail(ctx, MIPX_ETOOBIG, "(m,n) exceeds every LP kernel");
    // one tableau slab per workgroup; a launch on another stream than the context's runs beside the main
    // one and has slabs of its own (fewer: those launches are the engine's strong-branching probes)
    const bool side = stream != ctx->stream;
    const int cap = side ? 256 : 1024;              // 1024 x 4 MiB = 4 GiB at 1024 x 512
    const int slabs = batch < cap ? batch : cap;
    double *&scratch = side ? p->big_scratch2 : p->big_scratch;
    int &have = side ? p->big_slabs2 : p->big_slabs;
    if (mcap > p->big_rows) {   // (slabs of mcap rows: what is there is too small)
        have = 0;
        if (!side) p->big_slabs2 = 0; else p->big_slabs = 0;
        p->big_rows = mcap;
    }
    if (slabs > have) {
        if (side) HIP_TRY(ctx, hipStreamSynchronize(stream));   // (an earlier side launch may still use the old slabs)
        if (scratch) (void)hipFree(scratch);
        scratch = nullptr;
        have = 0;
        HIP_TRY(ctx, hipMalloc((void **)&scratch, (size_t)slabs * p->big_rows * p->n * sizeof(double)));
        have = slabs;
    }
    const size_t lds = mipx::big_lds_bytes(mcap, p->n);
    HIP_TRY(ctx, hipFuncSetAttribute((const void *)mipx::lp_dual_simplex_big,
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(mipx::lp_dual_simplex_big, dim3(slabs), dim3(mipx::kBigNT), lds, stream,
                       a, scratch);
    HIP_TRY(ctx, hipGetLastError());
    return MIPX_OK;
}

}  // namespace

#include "cut_kernels.hip.h"
#include "comm.hip.h"
#include "tree_engine.hip.h"

extern "C" {

int mipx_abi_version(void) { return 1; }

int mipx_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int mipx_ctx_create(int device, mipx_ctx **out) {
    if (!out) return MIPX_EINVAL;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return MIPX_ENODEV;
    if (device < 0 || device >= ndev) return MIPX_ENODEV;
    mipx_ctx *ctx = new (std::nothrow) mipx_ctx();
    if (!ctx) return MIPX_ENOMEM;
    ctx->device = device;
    if (hipSetDevice(device) != hipSuccess ||
        hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreate(&ctx->ev0) != hipSuccess || hipEventCreate(&ctx->ev1) != hipSuccess) {
        delete ctx;
        return MIPX_EHIP;
    }
    *out = ctx;
    return MIPX_OK;
}

void mipx_ctx_destroy(mipx_ctx *ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    if (ctx->ev0) (void)hipEventDestroy(ctx->ev0);
    if (ctx->ev1) (void)hipEventDestroy(ctx->ev1);
    if (ctx->k0) (void)hipEventDestroy(ctx->k0);
    if (ctx->k1) (void)hipEventDestroy(ctx->k1);
    if (ctx->scratch) (void)hipFree(ctx->scratch);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

const char *mipx_last_error(const mipx_ctx *ctx) { return ctx ? ctx->err.c_str() : "null context"; }

int mipx_ctx_sync(mipx_ctx *ctx) {
    if (!ctx) return MIPX_EINVAL;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return MIPX_OK;
}

int mipx_problem_create(mipx_ctx *ctx, int m, int n, const double *A, const double *b,
                        const double *c, mipx_problem **out) {
    if (!ctx || !out || m < 0 || n <= 0 || !c || (m > 0 && (!A || !b)))
        return fail(ctx, MIPX_EINVAL, "mipx_problem_create: bad argument");
    *out = nullptr;
    if (!shape_supported(m, n)) return fail(ctx, MIPX_ETOOBIG, "mipx_problem_create: (m,n) exceeds every LP kernel");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    mipx_problem *p = new (std::nothrow) mipx_problem();
    if (!p) return fail(ctx, MIPX_ENOMEM, "mipx_problem_create: host alloc");
    p->ctx = ctx;
    p->m = m;
    p->n = n;
    const size_t am = (size_t)(m > 0 ? m : 1);
    hipError_t e;
    if ((e = hipMalloc(&p->dA, am * n * sizeof(double))) != hipSuccess ||
        (e = hipMalloc(&p->db, am * sizeof(double))) != hipSuccess ||
        (e = hipMalloc(&p->dc, (size_t)n * sizeof(double))) != hipSuccess) {
        mipx_problem_destroy(p);
        return fail(ctx, MIPX_EHIP, "mipx_problem_create: hipMalloc", e);
    }
    if (m > 0) {
        HIP_TRY(ctx, hipMemcpyAsync(p->dA, A, (size_t)m * n * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
        HIP_TRY(ctx, hipMemcpyAsync(p->db, b, (size_t)m * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    }
    HIP_TRY(ctx, hipMemcpyAsync(p->dc, c, (size_t)n * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    *out = p;
    return MIPX_OK;
}

void mipx_problem_destroy(mipx_problem *p) {
    if (!p) return;
    if (p->ctx) (void)hipSetDevice(p->ctx->device);
    if (p->dA) (void)hipFree(p->dA);
    if (p->db) (void)hipFree(p->db);
    if (p->dc) (void)hipFree(p->dc);
    if (p->scratch) (void)hipFree(p->scratch);
    if (p->dbg_T) (void)hipFree(p->dbg_T);
    if (p->dbg_vec) (void)hipFree(p->dbg_vec);
    if (p->dbg_idx) (void)hipFree(p->dbg_idx);
    if (p->big_scratch) (void)hipFree(p->big_scratch);
    if (p->big_scratch2) (void)hipFree(p->big_scratch2);
    if (p->root_state) (void)hipFree(p->root_state);
    if (p->anchor_T) (void)hipFree(p->anchor_T);
    if (p->anchor_vec) (void)hipFree(p->anchor_vec);
    if (p->anchor_idx) (void)hipFree(p->anchor_idx);
    delete p;
}

int mipx_lp_solve_batch_dev(mipx_problem *p, int batch, const double *l, const double *u,
                            const int8_t *vstat_in, int max_iter, int32_t *status, double *obj,
                            double *x, double *y, int8_t *vstat_out, int32_t *iters,
                            int32_t *npivots) {
    if (!p) return MIPX_EINVAL;
    mipx_ctx *ctx = p->ctx;
    if (batch < 0 || (batch > 0 && (!l || !u))) return fail(ctx, MIPX_EINVAL, "mipx_lp_solve_batch_dev: bad argument");
    if (batch == 0) return MIPX_OK;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    mipx::LpArgs a;
    a.m = p->m; a.n = p->n;
    a.A = p->dA; a.b = p->db; a.c = p->dc;
    a.A_stride = a.b_stride = a.c_stride = 0;
    a.l = l; a.u = u; a.vstat_in = vstat_in; a.slot = nullptr; a.max_iter = max_iter;
    a.anchor_T = p->anchor_on ? p->anchor_T : nullptr;
    a.anchor_vec = p->anchor_on ? p->anchor_vec : nullptr;
    a.anchor_idx = p->anchor_on ? p->anchor_idx : nullptr;
    a.refactor_only = 0;
    a.status = status; a.obj = obj; a.x = x; a.y = y; a.vstat_out = vstat_out;
    a.iters = iters; a.npivots = npivots; a.batch = batch;
    a.dbg_T = p->dbg_T; a.dbg_vec = p->dbg_vec; a.dbg_idx = p->dbg_idx; a.dbg_all = 0;
    return launch_lp_any(p, a, batch);
}

int mipx_lp_solve_batch(mipx_problem *p, int batch, const double *l, const double *u,
                        const int8_t *vstat_in, int max_iter, int32_t *status, double *obj,
                        double *x, double *y, int8_t *vstat_out, int32_t *iters,
                        int32_t *npivots) {
    if (!p) return MIPX_EINVAL;
    mipx_ctx *ctx = p->ctx;
    if (batch < 0 || (batch > 0 && (!l || !u))) return fail(ctx, MIPX_EINVAL, "mipx_lp_solve_batch: bad argument");
    if (batch == 0) return MIPX_OK;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t B = (size_t)batch, n = (size_t)p->n, m = (size_t)p->m, nv = n + m;
    // carve one staging allocation
    size_t off = 0;
    auto carve = [&](size_t bytes) { size_t o = off; off = align_up(off + bytes, 256); return o; };
    const size_t o_l = carve(B * n * 8), o_u = carve(B * n * 8), o_vin = carve(B * nv),
                 o_st = carve(B * 4), o_obj = carve(B * 8), o_x = carve(B * n * 8),
                 o_y = carve(B * (m ? m : 1) * 8), o_vout = carve(B * nv), o_it = carve(B * 4),
                 o_np = carve(B * 4);
    if (off > p->scratch_bytes) {
        if (p->scratch) (void)hipFree(p->scratch);
        p->scratch = nullptr;
        p->scratch_bytes = 0;
        HIP_TRY(ctx, hipMalloc(&p->scratch, off));
        p->scratch_bytes = off;
    }
    char *base = (char *)p->scratch;
    hipStream_t st = ctx->stream;
    HIP_TRY(ctx, hipMemcpyAsync(base + o_l, l, B * n * 8, hipMemcpyHostToDevice, st));
    HIP_TRY(ctx, hipMemcpyAsync(base + o_u, u, B * n * 8, hipMemcpyHostToDevice, st));
    if (vstat_in) HIP_TRY(ctx, hipMemcpyAsync(base + o_vin, vstat_in, B * nv, hipMemcpyHostToDevice, st));
    int rc = mipx_lp_solve_batch_dev(
        p, batch, (const double *)(base + o_l), (const double *)(base + o_u),
        vstat_in ? (const int8_t *)(base + o_vin) : nullptr, max_iter, (int32_t *)(base + o_st),
        (double *)(base + o_obj), (double *)(base + o_x), (double *)(base + o_y),
        (int8_t *)(base + o_vout), (int32_t *)(base + o_it), (int32_t *)(base + o_np));
    if (rc) return rc;
    if (status) HIP_TRY(ctx, hipMemcpyAsync(status, base + o_st, B * 4, hipMemcpyDeviceToHost, st));
    if (obj) HIP_TRY(ctx, hipMemcpyAsync(obj, base + o_obj, B * 8, hipMemcpyDeviceToHost, st));
    if (x) HIP_TRY(ctx, hipMemcpyAsync(x, base + o_x, B * n * 8, hipMemcpyDeviceToHost, st));
    if (y && m) HIP_TRY(ctx, hipMemcpyAsync(y, base + o_y, B * m * 8, hipMemcpyDeviceToHost, st));
    if (vstat_out) HIP_TRY(ctx, hipMemcpyAsync(vstat_out, base + o_vout, B * nv, hipMemcpyDeviceToHost, st));
    if (iters) HIP_TRY(ctx, hipMemcpyAsync(iters, base + o_it, B * 4, hipMemcpyDeviceToHost, st));
    if (npivots) HIP_TRY(ctx, hipMemcpyAsync(npivots, base + o_np, B * 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, hipStreamSynchronize(st));
    return MIPX_OK;
}

int mipx_lp_solve_batch_cuts(mipx_problem *p, int batch, const double *l, const double *u,
                             const int8_t *vstat_in, int ncuts_total, const double *cut_pi,
                             const double *cut_pi0, int kc, const int32_t *ncut,
                             const int32_t *cut_ids, int max_iter, int32_t *status, double *obj,
                             double *x, double *y, int8_t *vstat_out, int32_t *iters,
                             int32_t *npivots) {
    if (!p) return MIPX_EINVAL;
    mipx_ctx *ctx = p->ctx;
    if (batch < 0 || kc < 1 || kc > 64 || ncuts_total < 0 || (ncuts_total && (!cut_pi || !cut_pi0)) ||
        (batch > 0 && (!l || !u || !ncut || !cut_ids)))
        return fail(ctx, MIPX_EINVAL, "mipx_lp_solve_batch_cuts: bad argument");
    if (batch == 0) return MIPX_OK;
    int maxc = 0;
    for (int k = 0; k < batch; k++) {
        if (ncut[k] < 0 || ncut[k] > kc) return fail(ctx, MIPX_EINVAL, "mipx_lp_solve_batch_cuts: ncut out of range");
        for (int i = 0; i < ncut[k]; i++)
            if (cut_ids[(size_t)k * kc + i] < 0 || cut_ids[(size_t)k * kc + i] >= ncuts_total)
                return fail(ctx, MIPX_EINVAL, "mipx_lp_solve_batch_cuts: cut id out of range");
        if (ncut[k] > maxc) maxc = ncut[k];
    }
    // (a problem of the register tiles stays on them with its cut rows -- one pricing rule for all its LPs --
    // or is refused; above the tiles the streamed kernel takes the cut rows too)
    if (pick_cfg(p->m, p->n) ? !pick_cfg(p->m + maxc, p->n) : !big_fits(p->m + kc, p->n))
        return fail(ctx, MIPX_ETOOBIG, "mipx_lp_solve_batch_cuts: m + cuts exceeds the LP kernels of this shape");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t B = (size_t)batch, n = (size_t)p->n, M = (size_t)p->m + kc, nvs = n + M, NC = (size_t)(ncuts_total ? ncuts_total : 1);
    size_t off = 0;
    auto carve = [&](size_t bytes) { size_t o = off; off = align_up(off + bytes, 256); return o; };
    const size_t o_l = carve(B * n * 8), o_u = carve(B * n * 8), o_vin = carve(B * nvs), o_cp = carve(NC * n * 8),
                 o_c0 = carve(NC * 8), o_nc = carve(B * 4), o_id = carve(B * kc * 4), o_st = carve(B * 4),
                 o_obj = carve(B * 8), o_x = carve(B * n * 8), o_y = carve(B * M * 8), o_vout = carve(B * nvs),
                 o_it = carve(B * 4), o_np = carve(B * 4);
    char *base = nullptr;
    HIP_TRY(ctx, hipMalloc((void **)&base, off));
    hipStream_t st = ctx->stream;
    auto run = [&]() -> int {
        HIP_TRY(ctx, hipMemcpyAsync(base + o_l, l, B * n * 8, hipMemcpyHostToDevice, st));
        HIP_TRY(ctx, hipMemcpyAsync(base + o_u, u, B * n * 8, hipMemcpyHostToDevice, st));
        if (vstat_in) HIP_TRY(ctx, hipMemcpyAsync(base + o_vin, vstat_in, B * nvs, hipMemcpyHostToDevice, st));
        if (ncuts_total) {
            HIP_TRY(ctx, hipMemcpyAsync(base + o_cp, cut_pi, (size_t)ncuts_total * n * 8, hipMemcpyHostToDevice, st));
            HIP_TRY(ctx, hipMemcpyAsync(base + o_c0, cut_pi0, (size_t)ncuts_total * 8, hipMemcpyHostToDevice, st));
        }
        HIP_TRY(ctx, hipMemcpyAsync(base + o_nc, ncut, B * 4, hipMemcpyHostToDevice, st));
        HIP_TRY(ctx, hipMemcpyAsync(base + o_id, cut_ids, B * kc * 4, hipMemcpyHostToDevice, st));
        mipx::LpArgs a;
        a.m = p->m; a.n = p->n;
        a.A = p->dA; a.b = p->db; a.c = p->dc;
        a.A_stride = a.b_stride = a.c_stride = 0;
        a.l = (const double *)(base + o_l); a.u = (const double *)(base + o_u);
        a.vstat_in = vstat_in ? (const int8_t *)(base + o_vin) : nullptr;
        a.slot = nullptr; a.max_iter = max_iter;
        a.anchor_T = nullptr; a.anchor_vec = nullptr; a.anchor_idx = nullptr; a.refactor_only = 0;
        a.status = (int32_t *)(base + o_st); a.obj = (double *)(base + o_obj); a.x = (double *)(base + o_x);
        a.y = (double *)(base + o_y); a.vstat_out = (int8_t *)(base + o_vout);
        a.iters = (int32_t *)(base + o_it); a.npivots = (int32_t *)(base + o_np); a.batch = batch;
        a.dbg_T = nullptr; a.dbg_vec = nullptr; a.dbg_idx = nullptr; a.dbg_all = 0;
        a.ncut = (const int32_t *)(base + o_nc); a.cut_ids = (const int32_t *)(base + o_id);
        a.cut_pi = (const double *)(base + o_cp); a.cut_pi0 = (const double *)(base + o_c0);
        a.cut_stride = kc; a.mstride = (int)M; a.vstat_by_node = 0; a.active = nullptr;
        const int rc = launch_lp_any(p, a, batch, nullptr, p->m + maxc);
        if (rc) return rc;
        if (status) HIP_TRY(ctx, hipMemcpyAsync(status, base + o_st, B * 4, hipMemcpyDeviceToHost, st));
        if (obj) HIP_TRY(ctx, hipMemcpyAsync(obj, base + o_obj, B * 8, hipMemcpyDeviceToHost, st));
        if (x) HIP_TRY(ctx, hipMemcpyAsync(x, base + o_x, B * n * 8, hipMemcpyDeviceToHost, st));
        if (y) HIP_TRY(ctx, hipMemcpyAsync(y, base + o_y, B * M * 8, hipMemcpyDeviceToHost, st));
        if (vstat_out) HIP_TRY(ctx, hipMemcpyAsync(vstat_out, base + o_vout, B * nvs, hipMemcpyDeviceToHost, st));
        if (iters) HIP_TRY(ctx, hipMemcpyAsync(iters, base + o_it, B * 4, hipMemcpyDeviceToHost, st));
        if (npivots) HIP_TRY(ctx, hipMemcpyAsync(npivots, base + o_np, B * 4, hipMemcpyDeviceToHost, st));
        HIP_TRY(ctx, hipStreamSynchronize(st));
        return MIPX_OK;
    };
    const int rc = run();
    (void)hipStreamSynchronize(st);
    (void)hipFree(base);
    return rc;
}

int mipx_lp_dive_batch(mipx_problem *p, int batch, const double *l, const double *u,
                       const int8_t *vstat_in, int max_iter, int rule, const int32_t *int_idx,
                       int n_int, const double *cost_l, const double *cost_r,
                       const uint8_t *has_entry, double cutoff, int32_t *status, double *obj,
                       double *x, int8_t *vstat_out, int32_t *iters, int32_t *npivots,
                       int32_t *dive_var, int32_t *dive_dir, double *dive_val) {
    return mipx_lp_plunge_batch(p, batch, 1, l, u, vstat_in, max_iter, rule, int_idx, n_int, cost_l, cost_r,
                                has_entry, cutoff, status, obj, x, vstat_out, iters, npivots, dive_var, dive_dir,
                                dive_val);
}

int mipx_lp_plunge_batch(mipx_problem *p, int batch, int depth, const double *l, const double *u,
                         const int8_t *vstat_in, int max_iter, int rule, const int32_t *int_idx,
                         int n_int, const double *cost_l, const double *cost_r,
                         const uint8_t *has_entry, double cutoff, int32_t *status, double *obj,
                         double *x, int8_t *vstat_out, int32_t *iters, int32_t *npivots,
                         int32_t *dive_var, int32_t *dive_dir, double *dive_val) {
    if (!p) return MIPX_EINVAL;
    mipx_ctx *ctx = p->ctx;
    if (batch < 0 || n_int < 0 || depth < 1 || depth > 8 || (rule != 0 && rule != 1) ||
        (batch > 0 && (!l || !u || !status || !dive_var || !dive_dir || !dive_val)) ||
        (n_int > 0 && (!int_idx || (rule == 1 && (!cost_l || !cost_r || !has_entry)))))
        return fail(ctx, MIPX_EINVAL, "mipx_lp_plunge_batch: bad argument");
    if (batch == 0) return MIPX_OK;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t B = (size_t)batch, n = (size_t)p->n, m = (size_t)p->m, nv = n + m, ni = (size_t)(n_int ? n_int : 1);
    const size_t LB = ((size_t)depth + 1) * B, DB = (size_t)depth * B;   // output rows, decisions
    size_t off = 0;
    auto carve = [&](size_t bytes) { size_t o = off; off = align_up(off + bytes, 256); return o; };
    const size_t o_l = carve(B * n * 8), o_u = carve(B * n * 8), o_vin = carve(B * nv),
                 o_st = carve(LB * 4), o_obj = carve(LB * 8), o_x = carve(LB * n * 8),
                 o_vout = carve(LB * nv), o_it = carve(LB * 4), o_np = carve(LB * 4),
                 o_dv = carve(DB * 4), o_dd = carve(DB * 4), o_dx = carve(DB * 8), o_ii = carve(ni * 4),
                 o_cl = carve(n * 8), o_cr = carve(n * 8), o_he = carve(n);
    char *base = nullptr;
    HIP_TRY(ctx, hipMalloc((void **)&base, off));
    hipStream_t st = ctx->stream;
    auto run = [&]() -> int {
        HIP_TRY(ctx, hipMemsetAsync(base, 0, off, st));
        HIP_TRY(ctx, hipMemsetAsync(base + o_st + B * 4, 0xff, DB * 4, st));  // no child: status -1
        HIP_TRY(ctx, hipMemsetAsync(base + o_dv, 0xff, DB * 4, st));          // no dive: -1
        HIP_TRY(ctx, hipMemcpyAsync(base + o_l, l, B * n * 8, hipMemcpyHostToDevice, st));
        HIP_TRY(ctx, hipMemcpyAsync(base + o_u, u, B * n * 8, hipMemcpyHostToDevice, st));
        if (vstat_in) HIP_TRY(ctx, hipMemcpyAsync(base + o_vin, vstat_in, B * nv, hipMemcpyHostToDevice, st));
        if (n_int) HIP_TRY(ctx, hipMemcpyAsync(base + o_ii, int_idx, (size_t)n_int * 4, hipMemcpyHostToDevice, st));
        if (cost_l) HIP_TRY(ctx, hipMemcpyAsync(base + o_cl, cost_l, n * 8, hipMemcpyHostToDevice, st));
        if (cost_r) HIP_TRY(ctx, hipMemcpyAsync(base + o_cr, cost_r, n * 8, hipMemcpyHostToDevice, st));
        if (has_entry) HIP_TRY(ctx, hipMemcpyAsync(base + o_he, has_entry, n, hipMemcpyHostToDevice, st));
        mipx::LpArgs a;
        a.m = p->m; a.n = p->n;
        a.A = p->dA; a.b = p->db; a.c = p->dc;
        a.A_stride = a.b_stride = a.c_stride = 0;
        a.l = (const double *)(base + o_l); a.u = (const double *)(base + o_u);
        a.vstat_in = vstat_in ? (const int8_t *)(base + o_vin) : nullptr;
        a.slot = nullptr; a.max_iter = max_iter;
        a.anchor_T = p->anchor_on ? p->anchor_T : nullptr;
        a.anchor_vec = p->anchor_on ? p->anchor_vec : nullptr;
        a.anchor_idx = p->anchor_on ? p->anchor_idx : nullptr;
        a.refactor_only = 0;
        a.status = (int32_t *)(base + o_st); a.obj = (double *)(base + o_obj); a.x = (double *)(base + o_x);
        a.y = nullptr; a.vstat_out = (int8_t *)(base + o_vout);
        a.iters = (int32_t *)(base + o_it); a.npivots = (int32_t *)(base + o_np); a.batch = batch;
        a.dbg_T = nullptr; a.dbg_vec = nullptr; a.dbg_idx = nullptr; a.dbg_all = 0;
        a.dive = depth; a.dive_off = batch; a.rule = rule; a.n_int = n_int;
        a.int_idx = (const int32_t *)(base + o_ii);
        a.cost_l = (const double *)(base + o_cl); a.cost_r = (const double *)(base + o_cr);
        a.has_entry = (const uint8_t *)(base + o_he);
        a.dive_cutoff = cutoff;
        a.dive_var = (int32_t *)(base + o_dv); a.dive_dir = (int32_t *)(base + o_dd);
        a.dive_val = (double *)(base + o_dx);
        const int rc = launch_lp_any(p, a, batch);
        if (rc) return rc;
        HIP_TRY(ctx, hipMemcpyAsync(status, base + o_st, LB * 4, hipMemcpyDeviceToHost, st));
        if (obj) HIP_TRY(ctx, hipMemcpyAsync(obj, base + o_obj, LB * 8, hipMemcpyDeviceToHost, st));
        if (x) HIP_TRY(ctx, hipMemcpyAsync(x, base + o_x, LB * n * 8, hipMemcpyDeviceToHost, st));
        if (vstat_out) HIP_TRY(ctx, hipMemcpyAsync(vstat_out, base + o_vout, LB * nv, hipMemcpyDeviceToHost, st));
        if (iters) HIP_TRY(ctx, hipMemcpyAsync(iters, base + o_it, LB * 4, hipMemcpyDeviceToHost, st));
        if (npivots) HIP_TRY(ctx, hipMemcpyAsync(npivots, base + o_np, LB * 4, hipMemcpyDeviceToHost, st));
        HIP_TRY(ctx, hipMemcpyAsync(dive_var, base + o_dv, DB * 4, hipMemcpyDeviceToHost, st));
        HIP_TRY(ctx, hipMemcpyAsync(dive_dir, base + o_dd, DB * 4, hipMemcpyDeviceToHost, st));
        HIP_TRY(ctx, hipMemcpyAsync(dive_val, base + o_dx, DB * 8, hipMemcpyDeviceToHost, st));
        HIP_TRY(ctx, hipStreamSynchronize(st));
        return MIPX_OK;
    };
    const int rc = run();
    (void)hipStreamSynchronize(st);
    (void)hipFree(base);
    return rc;
}

int mipx_problem_set_anchor(mipx_problem *p, const int8_t *vstat) {
    if (!p) return MIPX_EINVAL;
    mipx_ctx *ctx = p->ctx;
    if (!vstat) { p->anchor_on = false; return MIPX_OK; }
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t m = p->m ? p->m : 1, n = p->n, nv = n + p->m;
    if (!p->anchor_T) {
        HIP_TRY(ctx, hipMalloc((void **)&p->anchor_T, m * n * 8));
        HIP_TRY(ctx, hipMalloc((void **)&p->anchor_vec, (n + 3 * m) * 8));
        HIP_TRY(ctx, hipMalloc((void **)&p->anchor_idx, (2 * n + m) * 4));
    }
    double *zeros = nullptr;
    int8_t *dv = nullptr;
    HIP_TRY(ctx, hipMalloc((void **)&zeros, n * 8));
    HIP_TRY(ctx, hipMalloc((void **)&dv, nv));
    HIP_TRY(ctx, hipMemset(zeros, 0, n * 8));
    HIP_TRY(ctx, hipMemcpy(dv, vstat, nv, hipMemcpyHostToDevice));
    mipx::LpArgs a;
    a.m = p->m; a.n = p->n;
    a.A = p->dA; a.b = p->db; a.c = p->dc;
    a.A_stride = a.b_stride = a.c_stride = 0;
    a.l = zeros; a.u = zeros; a.vstat_in = dv; a.slot = nullptr; a.max_iter = 0;
    a.anchor_T = nullptr; a.anchor_vec = nullptr; a.anchor_idx = nullptr; a.refactor_only = 1;
    a.status = nullptr; a.obj = nullptr; a.x = nullptr; a.y = nullptr; a.vstat_out = nullptr;
    a.iters = nullptr; a.npivots = nullptr; a.batch = 1;
    a.dbg_T = p->anchor_T; a.dbg_vec = p->anchor_vec; a.dbg_idx = p->anchor_idx; a.dbg_all = 0;
    const int lrc = launch_lp_any(p, a, 1);
    hipError_t e = lrc == MIPX_OK ? hipStreamSynchronize(ctx->stream) : hipSuccess;
    (void)hipFree(zeros);
    (void)hipFree(dv);
    if (lrc != MIPX_OK) return lrc;
    if (e != hipSuccess) return fail(ctx, MIPX_EHIP, "mipx_problem_set_anchor", e);
    p->anchor_on = true;
    return MIPX_OK;
}

int mipx_debug_enable(mipx_problem *p) {
    if (!p) return MIPX_EINVAL;
    mipx_ctx *ctx = p->ctx;
    if (p->dbg_T) return MIPX_OK;
    const size_t m = p->m ? p->m : 1, n = p->n;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipMalloc(&p->dbg_T, m * n * 8));
    HIP_TRY(ctx, hipMalloc(&p->dbg_vec, (n + 3 * m) * 8));
    HIP_TRY(ctx, hipMalloc(&p->dbg_idx, (2 * n + m) * 4));
    return MIPX_OK;
}

int mipx_debug_read(mipx_problem *p, double *T, double *vec, int32_t *idx) {
    if (!p || !p->dbg_T) return MIPX_EINVAL;
    mipx_ctx *ctx = p->ctx;
    const size_t m = p->m, n = p->n;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (T) HIP_TRY(ctx, hipMemcpy(T, p->dbg_T, m * n * 8, hipMemcpyDeviceToHost));
    if (vec) HIP_TRY(ctx, hipMemcpy(vec, p->dbg_vec, (n + 3 * m) * 8, hipMemcpyDeviceToHost));
    if (idx) HIP_TRY(ctx, hipMemcpy(idx, p->dbg_idx, (2 * n + m) * 4, hipMemcpyDeviceToHost));
    return MIPX_OK;
}

int mipx_lp_solve_multi(mipx_ctx *ctx, int m, int n, int batch, const double *A, const double *b,
                        const double *c, const double *l, const double *u, int max_iter,
                        int32_t *status, double *obj, double *x, int8_t *vstat_out, int32_t *iters,
                        int32_t *npivots) {
    if (!ctx || batch < 0 || m < 0 || n <= 0 || (batch && (!A || !b || !c || !l || !u)))
        return fail(ctx, MIPX_EINVAL, "mipx_lp_solve_multi: bad argument");
    if (batch == 0) return MIPX_OK;
    const KernelCfg *cfg = pick_cfg(m, n);
    if (!cfg) return fail(ctx, MIPX_ETOOBIG, "mipx_lp_solve_multi: (m,n) too big");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t B = (size_t)batch, nn = (size_t)n, mm = (size_t)(m ? m : 1), nv = (size_t)n + m;
    size_t off = 0;
    auto carve = [&](size_t bytes) { size_t o = off; off = align_up(off + bytes, 256); return o; };
    const size_t o_A = carve(B * mm * nn * 8), o_b = carve(B * mm * 8), o_c = carve(B * nn * 8),
                 o_l = carve(B * nn * 8), o_u = carve(B * nn * 8), o_st = carve(B * 4),
                 o_obj = carve(B * 8), o_x = carve(B * nn * 8), o_v = carve(B * nv),
                 o_it = carve(B * 4), o_np = carve(B * 4);
    char *base = nullptr;
    HIP_TRY(ctx, hipMalloc((void **)&base, off));
    hipStream_t st = ctx->stream;
    int rc = MIPX_OK;
    auto up = [&](size_t o, const void *src, size_t bytes) {
        if (rc == MIPX_OK && hipMemcpyAsync(base + o, src, bytes, hipMemcpyHostToDevice, st) != hipSuccess)
            rc = fail(ctx, MIPX_EHIP, "mipx_lp_solve_multi: upload");
    };
    up(o_A, A, B * (size_t)m * nn * 8); up(o_b, b, B * (size_t)m * 8); up(o_c, c, B * nn * 8);
    up(o_l, l, B * nn * 8); up(o_u, u, B * nn * 8);
    if (rc == MIPX_OK) {
        mipx::LpArgs a;
        a.m = m; a.n = n;
        a.A = (const double *)(base + o_A); a.b = (const double *)(base + o_b);
        a.c = (const double *)(base + o_c);
        a.A_stride = (size_t)m * nn; a.b_stride = (size_t)m; a.c_stride = nn;
        a.l = (const double *)(base + o_l); a.u = (const double *)(base + o_u);
        a.vstat_in = nullptr; a.slot = nullptr; a.max_iter = max_iter;
    a.anchor_T = nullptr; a.anchor_vec = nullptr; a.anchor_idx = nullptr; a.refactor_only = 0;
        a.status = (int32_t *)(base + o_st); a.obj = (double *)(base + o_obj);
        a.x = (double *)(base + o_x); a.y = nullptr; a.vstat_out = (int8_t *)(base + o_v);
        a.iters = (int32_t *)(base + o_it); a.npivots = (int32_t *)(base + o_np); a.batch = batch;
        a.dbg_T = nullptr; a.dbg_vec = nullptr; a.dbg_idx = nullptr; a.dbg_all = 0;
        if (!ctx->k0 && (hipEventCreate(&ctx->k0) != hipSuccess || hipEventCreate(&ctx->k1) != hipSuccess))
            rc = fail(ctx, MIPX_EHIP, "mipx_lp_solve_multi: events");
        if (rc == MIPX_OK) (void)hipEventRecord(ctx->k0, st);
        cfg->launch(a, batch, st);
        if (hipGetLastError() != hipSuccess) rc = fail(ctx, MIPX_EHIP, "mipx_lp_solve_multi: launch");
        if (rc == MIPX_OK) (void)hipEventRecord(ctx->k1, st);
    }
    auto down = [&](void *dst, size_t o, size_t bytes) {
        if (dst && rc == MIPX_OK && hipMemcpyAsync(dst, base + o, bytes, hipMemcpyDeviceToHost, st) != hipSuccess)
            rc = fail(ctx, MIPX_EHIP, "mipx_lp_solve_multi: download");
    };
    down(status, o_st, B * 4); down(obj, o_obj, B * 8); down(x, o_x, B * nn * 8);
    down(vstat_out, o_v, B * nv); down(iters, o_it, B * 4); down(npivots, o_np, B * 4);
    if (hipStreamSynchronize(st) != hipSuccess && rc == MIPX_OK) rc = fail(ctx, MIPX_EHIP, "mipx_lp_solve_multi: sync");
    ctx->last_kernel_ms = -1.f;
    if (rc == MIPX_OK && hipEventElapsedTime(&ctx->last_kernel_ms, ctx->k0, ctx->k1) != hipSuccess) ctx->last_kernel_ms = -1.f;
    (void)hipFree(base);
    return rc;
}

int mipx_gomory_batch(mipx_problem *p, int batch, const double *l, const double *u,
                      const int8_t *vstat, const double *x, const uint8_t *is_int, double max_term,
                      int32_t *ncuts, int32_t *row_idx, double *pi, double *pi0, double *safe_pi,
                      double *safe_pi0) {
    if (!p) return MIPX_EINVAL;
    mipx_ctx *ctx = p->ctx;
    if (batch < 0 || (batch && (!l || !u || !vstat || !x || !is_int || !ncuts)))
        return fail(ctx, MIPX_EINVAL, "mipx_gomory_batch: bad argument");
    if (batch == 0) return MIPX_OK;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t B = (size_t)batch, n = (size_t)p->n, m = (size_t)(p->m ? p->m : 1), nv = n + p->m;
    size_t off = 0;
    auto carve = [&](size_t bytes) { size_t o = off; off = align_up(off + bytes, 256); return o; };
    const size_t o_l = carve(B * n * 8), o_u = carve(B * n * 8), o_v = carve(B * nv),
                 o_x = carve(B * n * 8), o_int = carve(n), o_T = carve(B * m * n * 8),
                 o_vec = carve(B * (n + 3 * m) * 8), o_idx = carve(B * (2 * n + m) * 4),
                 o_nc = carve(B * 4), o_ri = carve(B * m * 4), o_pi = carve(B * m * n * 8),
                 o_p0 = carve(B * m * 8), o_sp = carve(B * m * n * 8), o_s0 = carve(B * m * 8);
    if (off > ctx->scratch_bytes) {
        if (ctx->scratch) (void)hipFree(ctx->scratch);
        ctx->scratch = nullptr;
        ctx->scratch_bytes = 0;
        HIP_TRY(ctx, hipMalloc(&ctx->scratch, off));
        ctx->scratch_bytes = off;
    }
    char *base = (char *)ctx->scratch;
    hipStream_t st = ctx->stream;
    int rc = MIPX_OK;
    auto up = [&](size_t o, const void *src, size_t bytes) {
        if (rc == MIPX_OK && hipMemcpyAsync(base + o, src, bytes, hipMemcpyHostToDevice, st) != hipSuccess)
            rc = fail(ctx, MIPX_EHIP, "mipx_gomory_batch: upload");
    };
    up(o_l, l, B * n * 8); up(o_u, u, B * n * 8); up(o_v, vstat, B * nv); up(o_x, x, B * n * 8);
    up(o_int, is_int, n);
    if (rc == MIPX_OK) {
        // K1 from the given (optimal) basis: refactorises and leaves T / bvar / nvar in HBM
        mipx::LpArgs a;
        a.m = p->m; a.n = p->n;
        a.A = p->dA; a.b = p->db; a.c = p->dc;
        a.A_stride = a.b_stride = a.c_stride = 0;
        a.l = (const double *)(base + o_l); a.u = (const double *)(base + o_u);
        a.vstat_in = (const int8_t *)(base + o_v); a.slot = nullptr; a.max_iter = 0;
        a.anchor_T = nullptr; a.anchor_vec = nullptr; a.anchor_idx = nullptr; a.refactor_only = 0;
        a.status = nullptr; a.obj = nullptr; a.x = nullptr; a.y = nullptr; a.vstat_out = nullptr;
        a.iters = nullptr; a.npivots = nullptr; a.batch = batch;
        a.dbg_T = (double *)(base + o_T); a.dbg_vec = (double *)(base + o_vec);
        a.dbg_idx = (int32_t *)(base + o_idx); a.dbg_all = 1;
        rc = launch_lp_any(p, a, batch);
    }
    if (rc == MIPX_OK) {
        mipx::GomoryArgs g;
        g.m = p->m; g.n = p->n; g.batch = batch;
        g.A = p->dA; g.b = p->db;
        g.T = (const double *)(base + o_T); g.idx = (const int32_t *)(base + o_idx);
        g.x = (const double *)(base + o_x); g.is_int = (const uint8_t *)(base + o_int);
        g.max_term = max_term;
        g.ncuts = (int32_t *)(base + o_nc); g.row_idx = (int32_t *)(base + o_ri);
        g.pi = (double *)(base + o_pi); g.pi0 = (double *)(base + o_p0);
        g.safe_pi = (double *)(base + o_sp); g.safe_pi0 = (double *)(base + o_s0);
        g.chunks = batch >= 256 ? 1 : (batch >= 32 ? 4 : 16);  // (enough workgroups to use the GPU either way)
        g.group = mipx::gomory_group(p->n, p->m, (long)batch * g.chunks);
        g.mfma = (getenv("MIPX_K2_MFMA") && atoi(getenv("MIPX_K2_MFMA"))) ? 1 : 0;   // (experiment: cut_kernels.hip.h)
        const size_t lds = mipx::gomory_lds_bytes(p->n, p->m, g.group);
        if (lds > 64 * 1024 && hipFuncSetAttribute((const void *)mipx::gomory_cuts<256>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            rc = fail(ctx, MIPX_EHIP, "mipx_gomory_batch: LDS size");
        hipLaunchKernelGGL((mipx::gomory_cuts<256>), dim3(batch * g.chunks), dim3(256), lds, st, g);
        if (hipGetLastError() != hipSuccess) rc = fail(ctx, MIPX_EHIP, "mipx_gomory_batch: launch");
    }
    auto down = [&](void *dst, size_t o, size_t bytes) {
        if (dst && rc == MIPX_OK && hipMemcpyAsync(dst, base + o, bytes, hipMemcpyDeviceToHost, st) != hipSuccess)
            rc = fail(ctx, MIPX_EHIP, "mipx_gomory_batch: download");
    };
    down(ncuts, o_nc, B * 4); down(row_idx, o_ri, B * m * 4); down(pi, o_pi, B * m * n * 8);
    down(pi0, o_p0, B * m * 8); down(safe_pi, o_sp, B * m * n * 8); down(safe_pi0, o_s0, B * m * 8);
    if (hipStreamSynchronize(st) != hipSuccess && rc == MIPX_OK) rc = fail(ctx, MIPX_EHIP, "mipx_gomory_batch: sync");
    return rc;
}

int mipx_cut_select_batch(mipx_ctx *ctx, int n, int batch, int kmax, const int32_t *npool,
                          const double *pi, const double *pi0, const double *x,
                          int max_nonzero_coefs, double min_cut_depth, double cos_parallel,
                          double max_abs_coef, int32_t *nadded, int32_t *added, int32_t *terminator,
                          double *depth) {
    if (!ctx || n <= 0 || n > 1024 || batch < 0 || kmax < 1 ||
        (batch && (!npool || !pi || !pi0 || !x || !nadded || !added || !terminator)))
        return fail(ctx, MIPX_EINVAL, "mipx_cut_select_batch: bad argument");
    if (batch == 0) return MIPX_OK;
    for (int k = 0; k < batch; k++)
        if (npool[k] < 0 || npool[k] > kmax) return fail(ctx, MIPX_EINVAL, "mipx_cut_select_batch: npool out of range");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t B = (size_t)batch, nn = (size_t)n, K = (size_t)kmax;
    size_t off = 0;
    auto carve = [&](size_t bytes) { size_t o = off; off = align_up(off + bytes, 256); return o; };
    const size_t o_np = carve(B * 4), o_pi = carve(B * K * nn * 8), o_p0 = carve(B * K * 8),
                 o_x = carve(B * nn * 8), o_na = carve(B * 4), o_ad = carve(B * K * 4),
                 o_te = carve(B * 4), o_de = carve(B * K * 8);
    if (off > ctx->scratch_bytes) {
        if (ctx->scratch) (void)hipFree(ctx->scratch);
        ctx->scratch = nullptr;
        ctx->scratch_bytes = 0;
        HIP_TRY(ctx, hipMalloc(&ctx->scratch, off));
        ctx->scratch_bytes = off;
    }
    char *base = (char *)ctx->scratch;
    hipStream_t st = ctx->stream;
    int rc = MIPX_OK;
    auto up = [&](size_t o, const void *src, size_t bytes) {
        if (rc == MIPX_OK && hipMemcpyAsync(base + o, src, bytes, hipMemcpyHostToDevice, st) != hipSuccess)
            rc = fail(ctx, MIPX_EHIP, "mipx_cut_select_batch: upload");
    };
    up(o_np, npool, B * 4); up(o_pi, pi, B * K * nn * 8); up(o_p0, pi0, B * K * 8); up(o_x, x, B * nn * 8);
    if (rc == MIPX_OK) {
        mipx::SelectArgs g;
        g.n = n; g.batch = batch; g.kmax = kmax;
        g.npool = (const int32_t *)(base + o_np); g.pi = (const double *)(base + o_pi);
        g.pi0 = (const double *)(base + o_p0); g.x = (const double *)(base + o_x);
        g.max_nonzero_coefs = max_nonzero_coefs; g.min_cut_depth = min_cut_depth;
        g.cos_parallel = cos_parallel; g.max_abs_coef = max_abs_coef;
        g.nadded = (int32_t *)(base + o_na); g.added = (int32_t *)(base + o_ad);
        g.terminator = (int32_t *)(base + o_te); g.depth = (double *)(base + o_de);
        const size_t lds = K * (3 * 8 + 2 * 4) + 16 + 64;
        hipLaunchKernelGGL(mipx::select_cuts, dim3(batch), dim3(256), lds, st, g);
        if (hipGetLastError() != hipSuccess) rc = fail(ctx, MIPX_EHIP, "mipx_cut_select_batch: launch");
    }
    auto down = [&](void *dst, size_t o, size_t bytes) {
        if (dst && rc == MIPX_OK && hipMemcpyAsync(dst, base + o, bytes, hipMemcpyDeviceToHost, st) != hipSuccess)
            rc = fail(ctx, MIPX_EHIP, "mipx_cut_select_batch: download");
    };
    down(nadded, o_na, B * 4); down(added, o_ad, B * K * 4); down(terminator, o_te, B * 4);
    down(depth, o_de, B * K * 8);
    if (hipStreamSynchronize(st) != hipSuccess && rc == MIPX_OK) rc = fail(ctx, MIPX_EHIP, "mipx_cut_select_batch: sync");
    return rc;
}

int mipx_safe_cut_batch(mipx_ctx *ctx, int n, int batch, const double *pi, const double *pi0,
                        int estimate, int make_integer, double max_term, double *safe_pi,
                        double *safe_pi0, double *num, double *den, double *scaled_pi,
                        double *scaled_pi0, int32_t *nonzero) {
    if (!ctx || n <= 0 || batch < 0 || (estimate != 1 && estimate != 2) || !(max_term > 0) ||
        (batch && (!pi || !pi0 || !safe_pi || !safe_pi0)) || ((num == nullptr) != (den == nullptr)))
        return fail(ctx, MIPX_EINVAL, "mipx_safe_cut_batch: bad argument");
    if (batch == 0) return MIPX_OK;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t B = (size_t)batch, nn = (size_t)n;
    size_t off = 0;
    auto carve = [&](size_t bytes) { size_t o = off; off = align_up(off + bytes, 256); return o; };
    const size_t o_pi = carve(B * nn * 8), o_p0 = carve(B * 8), o_sp = carve(B * nn * 8), o_s0 = carve(B * 8),
                 o_nu = carve(B * (nn + 1) * 8), o_de = carve(B * (nn + 1) * 8), o_cp = carve(B * nn * 8),
                 o_c0 = carve(B * 8), o_nz = carve(B * 4);
    if (off > ctx->scratch_bytes) {
        if (ctx->scratch) (void)hipFree(ctx->scratch);
        ctx->scratch = nullptr;
        ctx->scratch_bytes = 0;
        HIP_TRY(ctx, hipMalloc(&ctx->scratch, off));
        ctx->scratch_bytes = off;
    }
    char *base = (char *)ctx->scratch;
    hipStream_t st = ctx->stream;
    HIP_TRY(ctx, hipMemcpyAsync(base + o_pi, pi, B * nn * 8, hipMemcpyHostToDevice, st));
    HIP_TRY(ctx, hipMemcpyAsync(base + o_p0, pi0, B * 8, hipMemcpyHostToDevice, st));
    mipx::SafeCutArgs g;
    g.n = n; g.batch = batch;
    g.pi = (const double *)(base + o_pi); g.pi0 = (const double *)(base + o_p0);
    g.estimate = estimate; g.make_integer = make_integer ? 1 : 0; g.max_term = max_term;
    g.safe_pi = (double *)(base + o_sp); g.safe_pi0 = (double *)(base + o_s0);
    g.num = (double *)(base + o_nu); g.den = (double *)(base + o_de);
    g.scaled_pi = (double *)(base + o_cp); g.scaled_pi0 = (double *)(base + o_c0);
    g.nonzero = (int32_t *)(base + o_nz);
    hipLaunchKernelGGL(mipx::safe_cut_batch, dim3(batch), dim3(256), 0, st, g);
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipMemcpyAsync(safe_pi, base + o_sp, B * nn * 8, hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, hipMemcpyAsync(safe_pi0, base + o_s0, B * 8, hipMemcpyDeviceToHost, st));
    if (num) {
        HIP_TRY(ctx, hipMemcpyAsync(num, base + o_nu, B * (nn + 1) * 8, hipMemcpyDeviceToHost, st));
        HIP_TRY(ctx, hipMemcpyAsync(den, base + o_de, B * (nn + 1) * 8, hipMemcpyDeviceToHost, st));
    }
    if (scaled_pi) HIP_TRY(ctx, hipMemcpyAsync(scaled_pi, base + o_cp, B * nn * 8, hipMemcpyDeviceToHost, st));
    if (scaled_pi0) HIP_TRY(ctx, hipMemcpyAsync(scaled_pi0, base + o_c0, B * 8, hipMemcpyDeviceToHost, st));
    if (nonzero) HIP_TRY(ctx, hipMemcpyAsync(nonzero, base + o_nz, B * 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, hipStreamSynchronize(st));
    return MIPX_OK;
}

int mipx_get_fraction_batch(mipx_ctx *ctx, int count, const double *x, const double *max_term,
                            const int32_t *estimate, double *num, double *den) {
    if (!ctx || count < 0 || (count && (!x || !max_term || !estimate || !num || !den)))
        return fail(ctx, MIPX_EINVAL, "mipx_get_fraction_batch: bad argument");
    if (count == 0) return MIPX_OK;
    for (int i = 0; i < count; i++)
        if (estimate[i] < 0 || estimate[i] > 2 || !(max_term[i] > 0))
            return fail(ctx, MIPX_EINVAL, "mipx_get_fraction_batch: estimate must be 0..2, max_term positive");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t N = (size_t)count;
    size_t off = 0;
    auto carve = [&](size_t bytes) { size_t o = off; off = align_up(off + bytes, 256); return o; };
    const size_t o_x = carve(N * 8), o_mt = carve(N * 8), o_es = carve(N * 4), o_nu = carve(N * 8), o_de = carve(N * 8);
    if (off > ctx->scratch_bytes) {
        if (ctx->scratch) (void)hipFree(ctx->scratch);
        ctx->scratch = nullptr;
        ctx->scratch_bytes = 0;
        HIP_TRY(ctx, hipMalloc(&ctx->scratch, off));
        ctx->scratch_bytes = off;
    }
    char *base = (char *)ctx->scratch;
    hipStream_t st = ctx->stream;
    HIP_TRY(ctx, hipMemcpyAsync(base + o_x, x, N * 8, hipMemcpyHostToDevice, st));
    HIP_TRY(ctx, hipMemcpyAsync(base + o_mt, max_term, N * 8, hipMemcpyHostToDevice, st));
    HIP_TRY(ctx, hipMemcpyAsync(base + o_es, estimate, N * 4, hipMemcpyHostToDevice, st));
    mipx::FractionArgs g;
    g.count = count;
    g.x = (const double *)(base + o_x); g.max_term = (const double *)(base + o_mt);
    g.estimate = (const int32_t *)(base + o_es);
    g.num = (double *)(base + o_nu); g.den = (double *)(base + o_de);
    hipLaunchKernelGGL(mipx::get_fraction_batch, dim3((count + 255) / 256), dim3(256), 0, st, g);
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipMemcpyAsync(num, base + o_nu, N * 8, hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, hipMemcpyAsync(den, base + o_de, N * 8, hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, hipStreamSynchronize(st));
    return MIPX_OK;
}

int mipx_dev_alloc(mipx_ctx *ctx, size_t bytes, void **dptr) {
    if (!ctx || !dptr) return MIPX_EINVAL;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipMalloc(dptr, bytes ? bytes : 1));
    return MIPX_OK;
}

int mipx_dev_free(mipx_ctx *ctx, void *dptr) {
    if (!ctx) return MIPX_EINVAL;
    if (!dptr) return MIPX_OK;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipFree(dptr));
    return MIPX_OK;
}

int mipx_memcpy_h2d(mipx_ctx *ctx, void *dst, const void *src, size_t bytes) {
    if (!ctx || (bytes && (!dst || !src))) return MIPX_EINVAL;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return MIPX_OK;
}

int mipx_memcpy_d2h(mipx_ctx *ctx, void *dst, const void *src, size_t bytes) {
    if (!ctx || (bytes && (!dst || !src))) return MIPX_EINVAL;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return MIPX_OK;
}

int mipx_timer_start(mipx_ctx *ctx) {
    if (!ctx) return MIPX_EINVAL;
    HIP_TRY(ctx, hipEventRecord(ctx->ev0, ctx->stream));
    return MIPX_OK;
}

int mipx_timer_stop(mipx_ctx *ctx, float *ms) {
    if (!ctx || !ms) return MIPX_EINVAL;
    HIP_TRY(ctx, hipEventRecord(ctx->ev1, ctx->stream));
    HIP_TRY(ctx, hipEventSynchronize(ctx->ev1));
    HIP_TRY(ctx, hipEventElapsedTime(ms, ctx->ev0, ctx->ev1));
    return MIPX_OK;
}

int mipx_last_kernel_ms(mipx_ctx *ctx, float *ms) {
    if (!ctx || !ms) return MIPX_EINVAL;
    *ms = ctx->last_kernel_ms;
    return MIPX_OK;
}

int mipx_kernel_name(int m, int n, char *buf, size_t buflen) {
    const KernelCfg *cfg = pick_cfg(m, n);
    if (!cfg && !big_fits(m, n)) return MIPX_ETOOBIG;
    if (!buf || buflen == 0) return MIPX_EINVAL;
    std::snprintf(buf, buflen, "%s", cfg ? cfg->name : "lp_dual_simplex_big");
    return MIPX_OK;
}

}  // extern "C"
