// lp_kernel_root.hip.h -- K1c: ONE cold node LP spread over the chip.
//
// K1b solves a node LP with one workgroup: right for a frontier of a thousand nodes, but the cold root of a
// 1024 x 512 tree is one LP of ~3 800 pivots, each streaming the 4 MiB tableau through ONE CU (0.42 s while
// 255 CUs idle -- the serial head of every C5 ramp-up).  Here the rows of the tableau are dealt out to G
// workgroups and every pivot is one kernel launch; the launch boundary is the device-wide barrier (no
// cooperative launch, no spinning: nothing here can hang).  Per pivot every workgroup
//   A. reduces the G leaving-row candidates the previous launch left (its own arithmetic, identical everywhere);
//   B. takes the winner's tableau row and border values (published with the candidate: nobody reads a row
//      another workgroup is about to overwrite);
//   C. runs the Harris ratio test on that row REDUNDANTLY (the column borders are read-only in a launch:
//      double-buffered by launch parity);
//   D. updates ITS rows of the tableau and their borders; workgroup w also updates columns j = w (mod G) of
//      the column borders into the other buffer;
//   E. leaves its best leaving-row candidate (+ that row) for the next launch.
// Same arithmetic per element and the same selection rules (lexicographic: level, priced violation, payload)
// as K1b / the oracle -- dual Devex pricing, Harris two-pass ratio test, Bland after m + n degenerate steps,
// fold-in-half sums, the symbolic component cleared below its tolerance at every update -- so the results are
// bit-identical to K1b's for the same LP (tests/test_lp_kernel_gpu.py).  A verdict of a solve that has carried
// symbolic values is taken on values worked out afresh from the tableau, as in K1b: the pivot launch leaves the
// status word -2 and the host puts lp_root_values in between.
// Cold starts only (no warm-start basis, no cut rows, no dive): the engine uses it for the root of shapes above
// the register tiles; mipx_lp_solve_batch for a single node without a basis.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "lp_kernel.hip.h"

namespace mipx {

constexpr int kRootNT = 256;      // threads per workgroup: thread t holds columns t + 256 k (n <= 1024), and one candidate of <= 256

struct RootKey {                  // a workgroup's best leaving row: what the choice between workgroups compares
    double viol;                  // priced violation (Bland: 0)
    int level, pay;               // level 0: no violated row here; pay = var << 16 | (sigma < 0) << 15 | row
    int bad, pad;                 // any of its rows holds a symbolic value (status 2 at the end)
};

struct RootRow {                  // ... and what the pivot on it needs (read for the winner only)
    double b0, ba, bb, w;         // border values of the row
    double lo, up;                // bounds of its basic variable
    int bvar, pad;
};

struct RootState {
    int m, n, G, RPB;             // G workgroups of RPB rows each
    double *T;                    // m x n
    double *beta0, *ba, *bb, *wgt;
    double *rlo, *rup;            // bounds of the rows' basic variables (slacks: 0, inf)
    int *bvar;
    // column borders, double-buffered by launch parity: [2][n]
    double *d, *va, *vb;
    int *nvar, *side;
    RootKey *ckey;                // [2][G]: the candidates
    RootRow *crow;                // [2][G]
    double *rowbuf;               // [2][G][n]: the candidate rows
    int *ctl;                     // [2][8]: status (-1 running) | iters | npiv | degen | parity of the final column borders
    const double *lo, *up;        // n (the node's bounds)
    int max_iter, cap;
};

__device__ __forceinline__ bool root_better(int la, double va, int pa, int lb, double vb, int pb) {
    // is (la, va, pa) a better leaving candidate than (lb, vb, pb): level, then priced violation, then payload
    return la > lb || (la == lb && la > 0 && (va > vb || (va == vb && pa < pb)));
}

// the candidate of rows [i0, i1) out of global memory (launch 0, and after the values were worked out afresh):
// one wave scans them, lane 0 publishes
__device__ inline void root_first_candidate(const RootState &S, int i0, int i1, bool bland, int lane, RootKey *okey,
                                            RootRow *orow, double *rowout) {
    const double INF = __builtin_huge_val();
    int blevel = 0, bp = kNoCand, bad = 0;
    double bk = -INF;
    for (int i = i0 + lane; i < i1; i += 64) {
        const int v = S.bvar[i];
        const double lo = S.rlo[i], up = S.rup[i];
        const double a = S.ba[i], bM = S.bb[i];
        bad |= bM > kBTol;
        int level = 0, sg = 0;
        double viol = 0.0;
        if (bM < -kBTol) { level = 2; viol = -bM; sg = 1; }
        else if (bM > kBTol) {
            if (!isinf(up)) { level = 2; viol = bM; sg = -1; }
            else if (bM > 1.0 + kBTol) { level = 2; viol = bM - 1.0; sg = -1; }
            else if (bM >= 1.0 - kBTol && a > kPTol) { level = 1; viol = a; sg = -1; }
        } else {
            if (a < lo - kPTol) { level = 1; viol = lo - a; sg = 1; }
            else if (!isinf(up) && a > up + kPTol) { level = 1; viol = a - up; sg = -1; }
        }
        if (level > 0) viol = viol * viol / S.wgt[i];   // dual Devex pricing
        if (bland && level > 0) { level = 1; viol = 0.0; }
        const int pay = (v << 16) | (sg < 0 ? 0x8000 : 0) | i;
        if (root_better(level, viol, pay, blevel, bk, bp)) { blevel = level; bk = viol; bp = pay; }
    }
    for (int h = 32; h >= 1; h >>= 1) {
        const int ol = __shfl_down(blevel, h, 64), op = __shfl_down(bp, h, 64);
        const double ok = __shfl_down(bk, h, 64);
        bad |= __shfl_down(bad, h, 64);
        if (root_better(ol, ok, op, blevel, bk, bp)) { blevel = ol; bk = ok; bp = op; }
    }
    blevel = __shfl(blevel, 0, 64); bp = __shfl(bp, 0, 64); bk = __shfl(bk, 0, 64);
    if (lane == 0) {
        RootKey c;
        c.viol = blevel > 0 ? bk : -INF; c.level = blevel; c.pay = blevel > 0 ? bp : kNoCand; c.bad = bad; c.pad = 0;
        *okey = c;
        if (blevel > 0) {
            const int r = bp & 0x7fff;
            RootRow q;
            q.b0 = S.beta0[r]; q.ba = S.ba[r]; q.bb = S.bb[r]; q.w = S.wgt[r]; q.bvar = S.bvar[r];
            q.lo = S.rlo[r]; q.up = S.rup[r]; q.pad = 0;
            *orow = q;
        }
    }
    if (blevel > 0) {
        const double *src = S.T + (size_t)(bp & 0x7fff) * S.n;
        for (int j = lane; j < S.n; j += 64) rowout[j] = src[j];
    }
}

// launch 0: T = -A, slack basis, nonbasic sides and values from the signs of d = c, beta = beta0 - T v, weights 1
__global__ __launch_bounds__(kRootNT) void lp_root_init(RootState S, const double *A, const double *b, const double *c) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, w = blockIdx.x;
    const int m = S.m, n = S.n;
    double *s_va = (double *)smem_raw, *s_vb = s_va + n;
    int n2 = 64;
    while (n2 < n) n2 <<= 1;
    const int PER = n2 / 64;
    // column borders (every workgroup works them out for its beta; the columns j = w (mod G) go to buffer 0)
    int anyb = 0;
    for (int j = tid; j < n; j += kRootNT) {
        const double lo = S.lo[j], up = S.up[j], dj = c[j];
        int side;
        if (lo == up) side = 0;
        else if (dj < -kDTol) side = isinf(up) ? 2 : 1;
        else if (dj > kDTol) side = 0;
        else side = 0;                       // (cold start: no warm-start code says "at upper")
        const double va = side == 0 ? lo : side == 1 ? up : 0.0, vb = side == 2 ? 1.0 : 0.0;
        s_va[j] = va; s_vb[j] = vb;
        anyb |= side == 2;
        if (j % S.G == w) { S.d[j] = dj; S.nvar[j] = j; S.side[j] = side; S.va[j] = va; S.vb[j] = vb; }
    }
    const int sym = __syncthreads_or(anyb);   // (the solve starts with symbolic values: see K1b on what that means for a verdict)
    const int i0 = min(w * S.RPB, m), i1 = min(i0 + S.RPB, m);
    for (int i = i0 + wave; i < i1; i += kRootNT / 64) {
        double *Ti = S.T + (size_t)i * n;
        const double *Ai = A + (size_t)i * n;
        double pa[16], pb[16];
#pragma unroll
        for (int k = 0; k < 16; k++) {
            const int j = lane + 64 * k;
            const bool in = k < PER && j < n;
            const double t = in ? -Ai[j] : 0.0;
            if (in) Ti[j] = t;
            pa[k] = in ? t * s_va[j] : 0.0;
            pb[k] = in ? t * s_vb[j] : 0.0;
        }
        for (int h = PER / 2; h >= 1; h >>= 1) {
#pragma unroll
            for (int k = 0; k < 8; k++)
                if (k < h) { pa[k] = pa[k] + pa[k + h]; pb[k] = pb[k] + pb[k + h]; }
        }
        double sa = pa[0], sb = pb[0];
#pragma unroll
        for (int h = 32; h >= 1; h >>= 1) {
            sa = sa + __shfl_down(sa, h, 64);
            sb = sb + __shfl_down(sb, h, 64);
        }
        if (lane == 0) {
            const double b0 = -b[i];
            S.beta0[i] = b0; S.ba[i] = b0 - sa; S.bb[i] = snap_m(0.0 - sb); S.wgt[i] = 1.0; S.bvar[i] = n + i;
            S.rlo[i] = 0.0; S.rup[i] = __builtin_huge_val();
        }
    }
    __syncthreads();   // (the rows' new borders are this workgroup's own writes: the barrier orders them for wave 0)
    if (wave == 0) root_first_candidate(S, i0, i1, false, lane, S.ckey + w, S.crow + w, S.rowbuf + (size_t)w * n);
    if (w == 0 && tid == 0) {
        S.ctl[0] = -1; S.ctl[1] = 0; S.ctl[2] = 0; S.ctl[3] = 0; S.ctl[4] = 0; S.ctl[5] = sym; S.ctl[6] = 1; S.ctl[7] = 0;
    }
}

// the values beta = beta0 - T v of every row afresh from the tableau (the host launches this when the status word
// says -2: a verdict was due on values that thousands of running updates had touched -- the rule of K1b's phase
// 3).  Same folds as launch 0; the candidates, the column borders and the control words go to parity par ^ 1 like
// a pivot launch's.
__global__ __launch_bounds__(kRootNT) void lp_root_values(RootState S, int par) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, w = blockIdx.x;
    const int m = S.m, n = S.n;
    double *s_va = (double *)smem_raw, *s_vb = s_va + n;
    const int *ctl = S.ctl + 8 * par;
    int *ctl_o = S.ctl + 8 * (par ^ 1);
    const int cpar = ctl[4] & 1;   // the column borders of the current state
    const int iters = ctl[1], npiv = ctl[2], degen = ctl[3], sym = ctl[5];
    int n2 = 64;
    while (n2 < n) n2 <<= 1;
    const int PER = n2 / 64;
    for (int j = tid; j < n; j += kRootNT) {
        const double va = S.va[(size_t)cpar * n + j], vb = S.vb[(size_t)cpar * n + j];
        s_va[j] = va; s_vb[j] = vb;
        if (cpar == par && j % S.G == w) {
            const size_t a = (size_t)par * n + j, o = (size_t)(par ^ 1) * n + j;
            S.d[o] = S.d[a]; S.nvar[o] = S.nvar[a]; S.side[o] = S.side[a]; S.va[o] = va; S.vb[o] = vb;
        }
    }
    __syncthreads();
    const int i0 = min(w * S.RPB, m), i1 = min(i0 + S.RPB, m);
    for (int i = i0 + wave; i < i1; i += kRootNT / 64) {
        const double *Ti = S.T + (size_t)i * n;
        double pa[16], pb[16];
#pragma unroll
        for (int k = 0; k < 16; k++) {
            const int j = lane + 64 * k;
            const bool in = k < PER && j < n;
            const double t = in ? Ti[j] : 0.0;
            pa[k] = in ? t * s_va[j] : 0.0;
            pb[k] = in ? t * s_vb[j] : 0.0;
        }
        for (int h = PER / 2; h >= 1; h >>= 1) {
#pragma unroll
            for (int k = 0; k < 8; k++)
                if (k < h) { pa[k] = pa[k] + pa[k + h]; pb[k] = pb[k] + pb[k + h]; }
        }
        double sa = pa[0], sb = pb[0];
#pragma unroll
        for (int h = 32; h >= 1; h >>= 1) {
            sa = sa + __shfl_down(sa, h, 64);
            sb = sb + __shfl_down(sb, h, 64);
        }
        if (lane == 0) { S.ba[i] = S.beta0[i] - sa; S.bb[i] = snap_m(0.0 - sb); }
    }
    __syncthreads();
    if (wave == 0)
        root_first_candidate(S, i0, i1, degen > m + n, lane, S.ckey + (size_t)(par ^ 1) * S.G + w, S.crow + (size_t)(par ^ 1) * S.G + w,
                             S.rowbuf + ((size_t)(par ^ 1) * S.G + w) * n);
    if (w == 0 && tid == 0) {
        ctl_o[0] = -1; ctl_o[1] = iters; ctl_o[2] = npiv; ctl_o[3] = degen; ctl_o[4] = par ^ 1; ctl_o[5] = sym; ctl_o[6] = 1; ctl_o[7] = 0;
    }
}

// the best of four (key, payload) pairs the waves left in LDS, by `keep`'s order (larger key, then smaller payload)
__device__ __forceinline__ void root_combine4(const double *k4, const int *p4, double &bk, int &bp) {
    bk = -__builtin_huge_val();
    bp = kNoCand;
#pragma unroll
    for (int q = 0; q < kRootNT / 64; q++) keep(bk, bp, k4[q], p4[q], p4[q] != kNoCand);
}

// one pivot; `par` = parity of the buffers this launch READS.  A launch is a chain of dependent memory round
// trips, so everything that does not depend on the leaving row is fetched at once, into REGISTERS: thread t
// holds columns t + 256 k of this workgroup's RPB rows of the tableau and of the column borders, lanes < RPB
// of every wave hold the rows' borders.  What remains is: the candidates' keys -> the winner's row -> arithmetic
// -> stores.  The selections are the same arg-min / arg-max as K1b's (larger key, then smaller payload:
// order-independent), taken thread -> wave -> workgroup.
template <int RPB>
__global__ __launch_bounds__(kRootNT) void lp_root_pivot(RootState S, int par) {
    constexpr int NWV = kRootNT / 64, K = 4;   // n <= 1024
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, w = blockIdx.x;
    const int m = S.m, n = S.n, G = S.G;
    const double INF = __builtin_huge_val();
    const int *ctl = S.ctl + 8 * par;
    int *ctl_o = S.ctl + 8 * (par ^ 1);
    __shared__ double s_rk[NWV];  // the waves' partial selections
    __shared__ int s_rp[NWV], s_rl[NWV], s_rs[NWV], s_rb[NWV];
    __shared__ int s_ci[4];
    __shared__ double s_cd[8], s_ac[RPB];
    const double *d = S.d + (size_t)par * n, *va = S.va + (size_t)par * n, *vb = S.vb + (size_t)par * n;
    const int *nvar = S.nvar + (size_t)par * n, *side = S.side + (size_t)par * n;
    double *d_o = S.d + (size_t)(par ^ 1) * n, *va_o = S.va + (size_t)(par ^ 1) * n, *vb_o = S.vb + (size_t)(par ^ 1) * n;
    int *nvar_o = S.nvar + (size_t)(par ^ 1) * n, *side_o = S.side + (size_t)(par ^ 1) * n;
    // ---- everything that can be asked for now --------------------------------------------------------------
    RootKey ck;
    ck.viol = -INF; ck.level = 0; ck.pay = kNoCand; ck.bad = 0;
    if (tid < G) ck = S.ckey[(size_t)par * G + tid];
    const int st = ctl[0], iters = ctl[1], npiv = ctl[2], degen = ctl[3], sym = ctl[5], fresh = ctl[6];
    const int i0 = min(w * RPB, m), nr = min(RPB, m - i0);
    double t_[RPB][K];
#pragma unroll
    for (int li = 0; li < RPB; li++)
#pragma unroll
        for (int k = 0; k < K; k++) {
            const int j = tid + kRootNT * k;
            t_[li][k] = (li < nr && j < n) ? S.T[(size_t)(i0 + li) * n + j] : 0.0;
        }
    double rb0 = 0.0, rba = 0.0, rbb = 0.0, rwg = 1.0, rlo = 0.0, rup = 0.0;   // lane < nr of every wave: row i0 + lane's borders
    int rbv = 0;
    if (lane < nr) {
        const int i = i0 + lane;
        rb0 = S.beta0[i]; rba = S.ba[i]; rbb = S.bb[i]; rwg = S.wgt[i]; rlo = S.rlo[i]; rup = S.rup[i]; rbv = S.bvar[i];
    }
    double cd_[K], clo_[K], cup_[K], cva_[K], cvb_[K];
    int cnv_[K], csd_[K];
#pragma unroll
    for (int k = 0; k < K; k++) {
        const int j = tid + kRootNT * k;
        const bool in = j < n;
        cnv_[k] = in ? nvar[j] : 0;
        csd_[k] = in ? side[j] : 0;
        cd_[k] = in ? d[j] : 0.0;
        cva_[k] = in ? va[j] : 0.0;
        cvb_[k] = in ? vb[j] : 0.0;
    }
#pragma unroll
    for (int k = 0; k < K; k++) {
        const int v = cnv_[k];
        clo_[k] = v < n ? S.lo[v] : 0.0;
        cup_[k] = v < n ? S.up[v] : INF;
    }
    if (st != -1) {               // finished in an earlier launch (or waiting for fresh values, -2): hand the word on
        if (w == 0 && tid < 8) ctl_o[tid] = ctl[tid];
        return;
    }
    const bool bland = degen > m + n;
    // ---- A. the leaving row: the best of the workgroups' candidates (thread g holds workgroup g's) ------------
    int blevel = ck.level, bp = ck.pay, bad = ck.bad, src = tid;
    double bk = ck.viol;
    for (int h = 32; h >= 1; h >>= 1) {
        const int ol = __shfl_down(blevel, h, 64), op = __shfl_down(bp, h, 64), os = __shfl_down(src, h, 64);
        const double ok = __shfl_down(bk, h, 64);
        bad |= __shfl_down(bad, h, 64);
        if (root_better(ol, ok, op, blevel, bk, bp)) { blevel = ol; bk = ok; bp = op; src = os; }
    }
    if (lane == 0) { s_rl[wave] = blevel; s_rk[wave] = bk; s_rp[wave] = bp; s_rs[wave] = src; s_rb[wave] = bad; }
    __syncthreads();
    blevel = 0; bp = kNoCand; bk = -INF; bad = 0; src = 0;
#pragma unroll
    for (int q = 0; q < NWV; q++) {
        bad |= s_rb[q];
        if (root_better(s_rl[q], s_rk[q], s_rp[q], blevel, bk, bp)) { blevel = s_rl[q]; bk = s_rk[q]; bp = s_rp[q]; src = s_rs[q]; }
    }
    int cmd = 0;
    if (blevel == 0) {
        int b2 = bad;
#pragma unroll
        for (int k = 0; k < K; k++) b2 |= (tid + kRootNT * k < n) & (csd_[k] == 2);
        cmd = __syncthreads_or(b2) ? 3 : 1;
    } else if ((S.max_iter > 0 && iters >= S.max_iter) || iters >= S.cap) {
        cmd = 4;
    }
    if (cmd) {   // (ctl[4]: the parity of the column borders that hold the final state; -2: fresh values first)
        const bool want = cmd != 4 && sym && !fresh;
        if (w == 0 && tid == 0) {
            ctl_o[0] = want ? -2 : cmd == 1 ? 0 : cmd == 3 ? 2 : 3;
            ctl_o[1] = iters; ctl_o[2] = npiv; ctl_o[3] = degen; ctl_o[4] = par; ctl_o[5] = sym; ctl_o[6] = fresh; ctl_o[7] = 0;
        }
        return;
    }
    src = __builtin_amdgcn_readfirstlane(src);
    const int r = bp & 0x7fff;
    const int sigma = (bp & 0x8000) ? -1 : 1;
    // ---- B. the row -----------------------------------------------------------------------------------------
    const double *rowsrc = S.rowbuf + ((size_t)par * G + src) * n;
    double rv_[K];
#pragma unroll
    for (int k = 0; k < K; k++) rv_[k] = tid + kRootNT * k < n ? rowsrc[tid + kRootNT * k] : 0.0;
    const RootRow rc = S.crow[(size_t)par * G + src];
    // ---- C. Harris ratio test on row r (every workgroup, identically) -------------------------------------
    double key_[K], aabs_[K], dje_[K];
    double k1 = -INF;
    int p1 = kNoCand;
#pragma unroll
    for (int k = 0; k < K; k++) {
        const int j = tid + kRootNT * k;
        const double a = sigma * rv_[k];
        const bool elig = j < n && clo_[k] != cup_[k] && (csd_[k] == 0 ? (a < -kPivTol) : (a > kPivTol));
        const double dj = csd_[k] == 0 ? fmax(cd_[k], 0.0) : fmax(-cd_[k], 0.0);
        const double aa = fabs(a);
        key_[k] = elig ? (bland ? dj / aa : (dj + kDTol) / aa) : INF;
        aabs_[k] = elig ? aa : -1.0;
        dje_[k] = dj;
        keep(k1, p1, -key_[k], (cnv_[k] << 16) | j, elig);
    }
    double wk;
    int wp = wave_argmax(k1, p1, wk);
    __syncthreads();   // (s_rk / s_rp of stage A have been read)
    if (lane == 0) { s_rk[wave] = wk; s_rp[wave] = wp; }
    __syncthreads();
    double nthmax;
    int w1;
    root_combine4(s_rk, s_rp, nthmax, w1);
    int qq = -1;
    if (w1 != kNoCand && bland) {
        qq = w1 & 0xffff;
    } else if (w1 != kNoCand) {
        const double thmax = -nthmax;
        const int jmin = w1 & 0xffff;
        double k2 = -INF;
        int p2 = kNoCand;
#pragma unroll
        for (int k = 0; k < K; k++) {
            const int j = tid + kRootNT * k;
            const bool ok = aabs_[k] >= 0.0 && (j == jmin || !(dje_[k] > thmax * aabs_[k]));
            keep(k2, p2, aabs_[k], (cnv_[k] << 16) | j, ok);
        }
        wp = wave_argmax(k2, p2, wk);
        __syncthreads();
        if (lane == 0) { s_rk[wave] = wk; s_rp[wave] = wp; }
        __syncthreads();
        double amax;
        int w2;
        root_combine4(s_rk, s_rp, amax, w2);
        qq = w2 & 0xffff;
    }
    // the entering column's data: in the registers of the thread that holds column q
    if (qq >= 0) {
#pragma unroll
        for (int k = 0; k < K; k++)
            if (tid + kRootNT * k == qq) {
                s_cd[0] = 1.0 / rv_[k]; s_cd[1] = cd_[k]; s_cd[2] = clo_[k]; s_cd[3] = cup_[k];
                s_cd[4] = cva_[k]; s_cd[5] = cvb_[k];
                s_ci[2] = cnv_[k];
                s_ci[1] = dje_[k] <= kDTol ? degen + 1 : 0;
#pragma unroll
                for (int li = 0; li < RPB; li++) s_ac[li] = t_[li][k];   // the pivot column's entries of this workgroup's rows
            }
    }
    __syncthreads();
    const int q = qq;
    if (q < 0) {   // no entering column: primal infeasible (on fresh values)
        if (w == 0 && tid == 0) {
            ctl_o[0] = (sym && !fresh) ? -2 : 1;
            ctl_o[1] = iters; ctl_o[2] = npiv; ctl_o[3] = degen; ctl_o[4] = par; ctl_o[5] = sym; ctl_o[6] = fresh; ctl_o[7] = 0;
        }
        return;
    }
    const int degen_n = s_ci[1];
    const double pinv = s_cd[0];
    const int lv = rc.bvar;
    const double llo = rc.lo, lup = rc.up;
    double la, lb;
    int newside;
    if (sigma > 0) { la = llo; lb = 0.0; newside = 0; }
    else if (!isinf(lup)) { la = lup; lb = 0.0; newside = 1; }
    else { la = 0.0; lb = 1.0; newside = 2; }
    const double dq = s_cd[1], b0r = rc.b0, bar = rc.ba, bbr = rc.bb, vaq = s_cd[4], vbq = s_cd[5], wr = rc.w;
    const int nq = s_ci[2];
    const double elo = s_cd[2], eup = s_cd[3];   // bounds of the entering variable (it becomes row r's basic variable)
    // ---- D. this workgroup's rows, their borders; its columns of the column borders ---------------------------
    const double rhon = b0r * pinv;
    const double ta = (bar - la) * pinv, tb = (bbr - lb) * pinv;
#pragma unroll
    for (int li = 0; li < RPB; li++) {
        const double a = s_ac[li];
        const bool isr = i0 + li == r;
#pragma unroll
        for (int k = 0; k < K; k++) {
            const int j = tid + kRootNT * k;
            const double rj = rv_[k] * pinv;
            const double o = isr ? (j == q ? pinv : rj) : (j == q ? -a * pinv : fma(-a, rj, t_[li][k]));
            t_[li][k] = o;
            if (li < nr && j < n) S.T[(size_t)(i0 + li) * n + j] = o;
        }
    }
    if (lane < nr) {   // the rows' borders (every wave keeps them; wave 0 stores).  Devex weights with the values, as K1b
        const int i = i0 + lane;
        const double a = s_ac[lane];
        if (i == r) {
            rb0 = rhon; rba = vaq + ta; rbb = snap_m(vbq + tb);
            const double wn = (wr * pinv) * pinv;
            rwg = wn < 1.0 ? 1.0 : wn;
            rbv = nq; rlo = elo; rup = eup;
            if (wave == 0) { S.bvar[i] = nq; S.rlo[i] = elo; S.rup[i] = eup; }
        } else {
            rb0 = fma(-a, rhon, rb0);
            rba = fma(-a, ta, rba); rbb = snap_m(fma(-a, tb, rbb));
            const double ratio = a * pinv;
            const double wn = (ratio * ratio) * wr;
            rwg = wn > rwg ? wn : rwg;
        }
        if (wave == 0) { S.beta0[i] = rb0; S.ba[i] = rba; S.bb[i] = rbb; S.wgt[i] = rwg; }
    }
    // columns j = w (mod G) of the column borders, out of the registers of the threads that hold them
#pragma unroll
    for (int k = 0; k < K; k++) {
        const int j = tid + kRootNT * k;
        if (j < n && j % G == w) {
            if (j == q) {
                d_o[j] = -dq * pinv; side_o[j] = newside; va_o[j] = la; vb_o[j] = lb; nvar_o[j] = lv;
            } else {
                d_o[j] = fma(-dq, rv_[k] * pinv, cd_[k]); side_o[j] = csd_[k]; va_o[j] = cva_[k]; vb_o[j] = cvb_[k]; nvar_o[j] = cnv_[k];
            }
        }
    }
    if (w == 0 && tid == 0) {
        ctl_o[0] = -1; ctl_o[1] = iters + 1; ctl_o[2] = npiv + 1; ctl_o[3] = degen_n; ctl_o[4] = par ^ 1;
        ctl_o[5] = sym | (lb != 0.0); ctl_o[6] = 0; ctl_o[7] = 0;
    }
    // ---- E. the candidate for the next launch (every wave works it out; wave 0 publishes) ------------------------
    const bool bland_n = degen_n > m + n;
    int cl = 0, cp = kNoCand, cb = 0;
    double cv = -INF;
    if (lane < nr) {
        cb = rbb > kBTol;
        int level = 0, sg = 0;
        double viol = 0.0;
        if (rbb < -kBTol) { level = 2; viol = -rbb; sg = 1; }
        else if (rbb > kBTol) {
            if (!isinf(rup)) { level = 2; viol = rbb; sg = -1; }
            else if (rbb > 1.0 + kBTol) { level = 2; viol = rbb - 1.0; sg = -1; }
            else if (rbb >= 1.0 - kBTol && rba > kPTol) { level = 1; viol = rba; sg = -1; }
        } else {
            if (rba < rlo - kPTol) { level = 1; viol = rlo - rba; sg = 1; }
            else if (!isinf(rup) && rba > rup + kPTol) { level = 1; viol = rba - rup; sg = -1; }
        }
        if (level > 0) viol = viol * viol / rwg;   // dual Devex pricing
        if (bland_n && level > 0) { level = 1; viol = 0.0; }
        if (level > 0) { cl = level; cv = viol; cp = (rbv << 16) | (sg < 0 ? 0x8000 : 0) | (i0 + lane); }
    }
    int cl0 = cl, cp0 = cp;
    double cv0 = cv;
#pragma unroll
    for (int h = RPB / 2; h >= 1; h >>= 1) {
        const int ol = __shfl_down(cl0, h, 64), op = __shfl_down(cp0, h, 64);
        const double ok = __shfl_down(cv0, h, 64);
        cb |= __shfl_down(cb, h, 64);
        if (root_better(ol, ok, op, cl0, cv0, cp0)) { cl0 = ol; cv0 = ok; cp0 = op; }
    }
    cl0 = __shfl(cl0, 0, 64); cp0 = __shfl(cp0, 0, 64); cv0 = __shfl(cv0, 0, 64); cb = __shfl(cb, 0, 64);
    const int lw = cl0 > 0 ? (cp0 & 0x7fff) - i0 : 0;   // the winner's lane
    if (wave == 0 && lane == lw) {
        RootKey c;
        c.viol = cl0 > 0 ? cv0 : -INF; c.level = cl0; c.pay = cl0 > 0 ? cp0 : kNoCand; c.bad = cb; c.pad = 0;
        S.ckey[(size_t)(par ^ 1) * G + w] = c;
        RootRow o;
        o.b0 = rb0; o.ba = rba; o.bb = rbb; o.w = rwg; o.lo = rlo; o.up = rup; o.bvar = rbv; o.pad = 0;
        S.crow[(size_t)(par ^ 1) * G + w] = o;
    }
    if (cl0 > 0) {
        double *rowout = S.rowbuf + ((size_t)(par ^ 1) * G + w) * n;
#pragma unroll
        for (int k = 0; k < K; k++) {
            const int j = tid + kRootNT * k;
            double v = t_[0][k];
#pragma unroll
            for (int li = 1; li < RPB; li++) v = li == lw ? t_[li][k] : v;
            if (j < n) rowout[j] = v;
        }
    }
}

// outputs, exactly as K1b writes them (one workgroup); par = parity of the final buffers
__global__ __launch_bounds__(kRootNT) void lp_root_final(RootState S, int par, LpArgs g, size_t onode) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int m = S.m, n = S.n, nv = n + m;
    const double INF = __builtin_huge_val();
    double *s_x = (double *)smem_raw;
    const int *ctl = S.ctl + 8 * par;
    const int status = ctl[0] < 0 ? 3 : ctl[0];
    const int cpar = ctl[4] & 1;   // (the column borders of the final state: empty launches after the last pivot flip `par`, not these)
    const double *d = S.d + (size_t)cpar * n, *va = S.va + (size_t)cpar * n;
    const int *nvar = S.nvar + (size_t)cpar * n, *side = S.side + (size_t)cpar * n;
    int n2 = 64;
    while (n2 < n) n2 <<= 1;
    const int PER = n2 / 64;
    for (int j = tid; j < n; j += kRootNT) {
        const int v = nvar[j];
        if (v < n) s_x[v] = side[j] == 2 ? kMReport : va[j];
    }
    for (int i = tid; i < m; i += kRootNT) {
        const int v = S.bvar[i];
        if (v < n) s_x[v] = fma(S.bb[i], kMReport, S.ba[i]);
    }
    __syncthreads();
    if (g.x) for (int j = tid; j < n; j += kRootNT) g.x[onode * n + j] = s_x[j];
    if (g.y) {
        for (int i = tid; i < m; i += kRootNT) g.y[onode * m + i] = 0.0;
        __syncthreads();
        for (int j = tid; j < n; j += kRootNT)
            if (nvar[j] >= n) g.y[onode * m + (nvar[j] - n)] = d[j];
    }
    if (g.vstat_out) {
        int8_t *vo = g.vstat_out + onode * nv;
        for (int i = tid; i < m; i += kRootNT) vo[S.bvar[i]] = 1;
        for (int j = tid; j < n; j += kRootNT) vo[nvar[j]] = side[j] ? 2 : 3;
    }
    if (g.dbg_T) {
        for (size_t e = tid; e < (size_t)m * n; e += kRootNT) g.dbg_T[e] = S.T[e];
        for (int j = tid; j < n; j += kRootNT) { g.dbg_vec[j] = d[j]; g.dbg_idx[j] = nvar[j]; g.dbg_idx[n + m + j] = side[j]; }
        for (int i = tid; i < m; i += kRootNT) {
            g.dbg_vec[n + i] = S.beta0[i]; g.dbg_vec[n + m + i] = S.ba[i]; g.dbg_vec[n + 2 * m + i] = S.bb[i];
            g.dbg_idx[n + i] = S.bvar[i];
        }
    }
    if (g.dive && tid < g.dive) {   // no plunge below a root solved this way: "no child" at every level
        g.status[onode + (size_t)(tid + 1) * (size_t)g.dive_off] = -1;
        g.dive_var[(size_t)tid * (size_t)g.dive_off + onode] = -1;
    }
    if (g.zero16 != nullptr && tid < 4) g.zero16[tid] = 0;
    if (tid < 64) {
        double p[16];
#pragma unroll
        for (int k = 0; k < 16; k++) {
            const int j = lane + 64 * k;
            p[k] = (k < PER && j < n) ? g.c[j] * s_x[j] : 0.0;
        }
        for (int h = PER / 2; h >= 1; h >>= 1) {
#pragma unroll
            for (int k = 0; k < 8; k++)
                if (k < h) p[k] = p[k] + p[k + h];
        }
        double sum = p[0];
#pragma unroll
        for (int h = 32; h >= 1; h >>= 1) sum = sum + __shfl_down(sum, h, 64);
        if (tid == 0) {
            if (g.obj) g.obj[onode] = status == 1 ? INF : sum;
            if (g.status) g.status[onode] = status;
            if (g.iters) g.iters[onode] = ctl[1];
            if (g.npivots) g.npivots[onode] = ctl[2];
        }
    }
}

}  // namespace mipx
