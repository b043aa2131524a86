// lp_kernel.hip.h -- K1: batched bounded dual simplex, one workgroup per node LP, gfx950.
//
// Replaces what the reference asks Clp to do in BaseNode._bound_lp / _strong_branch
// (simple_mip_solver/nodes/base_node.py:273, :645-646).  Not a translation of anything: the
// reference has no kernel.  Design:
//
//   * The condensed simplex tableau T (m x n, f64) lives in VGPRs for the whole solve: thread
//     (bi, bj) of a TBI x TBJ thread grid owns T[bi + TBI*ii][bj + TBJ*jj], ii < R, jj < C.
//     256x128 -> 16x32 threads x 8x8 doubles = 256 KiB of registers on one CU; it never touches
//     HBM again after the initial coalesced read of A.
//   * The interleaved ownership makes every LDS access of the per-pivot vectors (pivot row rho,
//     pivot column alpha) conflict-free (consecutive lanes -> consecutive 8-byte words, or a
//     half-wave broadcast), and lets work scale with ceil(m/TBI), ceil(n/TBJ).
//   * Borders (beta0, reduced costs d, basic values a + b*M, bounds, basis lists) live in LDS.
//   * Row/column extraction uses wave-uniform (SGPR) local indices so register arrays are only
//     ever indexed statically (no scratch).
//   * Selections (leaving row, Harris ratio test) are wavefront-wide butterfly reductions done
//     redundantly by every wave, so they need no extra barrier to publish the winner.
//   * Arithmetic is IEEE f64 with explicit fma and true division, compiled with
//     -ffp-contract=off, following the canonical operation order documented in
//     oracle/mipx_oracle.c so results are bit-identical to the CPU oracle.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mipx {

constexpr double kPTol = 1e-7;
constexpr double kDTol = 1e-7;
constexpr double kPivTol = 1e-9;
constexpr double kBTol = 1e-9;
constexpr double kMReport = 1e10;  // stands for the symbolic bound M when reporting an unbounded x

struct LpArgs {
    int m, n;
    const double *A, *b, *c;       // shared by the batch (strides 0) or one problem per node
    size_t A_stride, b_stride, c_stride;  // elements between consecutive nodes' A / b / c
    const double *l, *u;           // batch x n
    const int8_t *vstat_in;        // batch x (n+m) or nullptr
    const int32_t *slot;           // optional: node k reads l/u/vstat_in at row slot[k] (node pool)
    // optional anchor: tableau state of some basis of the same rows (layout of the dump: T m x n,
    // vec = [d (n) | beta0 (m) | ..], idx = [nvar (n) | bvar (m) | ..]); warm starts refactor from it
    // instead of from the slack basis (fewer pivots when the bases are close, e.g. the root's)
    const double *anchor_T, *anchor_vec;
    const int32_t *anchor_idx;
    int refactor_only;             // stop after the refactorisation (used to build an anchor)
    int max_iter;
    int32_t *status;
    double *obj;
    double *x;                     // batch x n
    double *y;                     // batch x m
    int8_t *vstat_out;             // batch x (n+m)
    int32_t *iters;
    int32_t *npivots;
    int batch;
    // optional debug dump of the final tableau state of node 0 (nullptr in production)
    double *dbg_T;      // m x n row-major
    double *dbg_vec;    // [d (n) | beta0 (m) | ba (m) | bb (m)]
    int32_t *dbg_idx;   // [nvar (n) | bvar (m) | side (n)]
    int dbg_all;        // 0: node 0 only; 1: every node k at offsets k*m*n, k*(n+3m), k*(2n+m)
};

// ---- wavefront-wide reductions on DPP (row_shr prefix-doubling inside each row of 16 lanes,
// then row_bcast:15 / row_bcast:31 across rows; the total lands in lane 63 and is read back with
// v_readlane, i.e. the result is wave-uniform).  A few dozen low-latency VALU instructions
// instead of 6 rounds of ds_bpermute.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_i32(int v) {
    return __builtin_amdgcn_update_dpp(v, v, CTRL, ROW_MASK, 0xf, false);
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_f64(double v) {
    const int lo = dpp_i32<CTRL, ROW_MASK>(__double2loint(v));
    const int hi = dpp_i32<CTRL, ROW_MASK>(__double2hiint(v));
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_max_f64(double v) {
    v = fmax(v, dpp_f64<0x111, 0xf>(v));  // row_shr:1
    v = fmax(v, dpp_f64<0x112, 0xf>(v));  // row_shr:2
    v = fmax(v, dpp_f64<0x114, 0xf>(v));  // row_shr:4
    v = fmax(v, dpp_f64<0x118, 0xf>(v));  // row_shr:8
    v = fmax(v, dpp_f64<0x142, 0xa>(v));  // row_bcast:15 -> rows 1,3
    v = fmax(v, dpp_f64<0x143, 0xc>(v));  // row_bcast:31 -> rows 2,3
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), 63);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), 63);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ int wave_min_i32(int v) {
    v = min(v, dpp_i32<0x111, 0xf>(v));
    v = min(v, dpp_i32<0x112, 0xf>(v));
    v = min(v, dpp_i32<0x114, 0xf>(v));
    v = min(v, dpp_i32<0x118, 0xf>(v));
    v = min(v, dpp_i32<0x142, 0xa>(v));
    v = min(v, dpp_i32<0x143, 0xc>(v));
    return __builtin_amdgcn_readlane(v, 63);
}
__device__ __forceinline__ int wave_max_i32(int v) {
    v = max(v, dpp_i32<0x111, 0xf>(v));
    v = max(v, dpp_i32<0x112, 0xf>(v));
    v = max(v, dpp_i32<0x114, 0xf>(v));
    v = max(v, dpp_i32<0x118, 0xf>(v));
    v = max(v, dpp_i32<0x142, 0xa>(v));
    v = max(v, dpp_i32<0x143, 0xc>(v));
    return __builtin_amdgcn_readlane(v, 63);
}
constexpr int kNoCand = 0x7fffffff;
// lane-local "keep the better candidate": larger key wins, ties go to the smaller payload
__device__ __forceinline__ void keep(double &bk, int &bp, double k, int p, bool valid) {
    const bool b = valid && (k > bk || (k == bk && p < bp));
    bk = b ? k : bk;
    bp = b ? p : bp;
}
// wave-wide argmax of (key, then smallest payload); kNoCand if no lane holds a candidate
__device__ __forceinline__ int wave_argmax(double key, int payload, double &kmax) {
    kmax = wave_max_f64(key);
    return wave_min_i32((payload != kNoCand && key == kmax) ? payload : kNoCand);
}

template <int MP, int NP>
struct Smem {
    double row[NP];    // extracted pivot row T[r][.]
    double alpha[MP];  // extracted pivot column T[.][q]
    double beta0[MP];
    double ba[MP];
    double bb[MP];
    double d[NP];
    double va[NP];
    double vb[NP];
    double lo[NP];     // structural bounds by variable index
    double up[NP];
    double key[NP];    // scratch: ratio keys / x assembly
    double aabs[NP];
    double dje[NP];
    int bvar[MP];
    int nvar[NP];
    int side[NP];      // 0 lower, 1 upper, 2 fake upper
    int wlist[NP];     // columns of the variables the warm start wants basic, ascending variable
    int nw;
    int ci[4];         // control words published by the control waves
    int pos[NP + MP];  // column of each variable in the starting tableau, -1 if basic
    int8_t entered[MP];// rows pivoted by the refactorisation
    double cd[2];
    int8_t wantb[NP + MP];
    int8_t atup[NP + MP];
};

template <int TBI, int TBJ, int R, int C>
__global__ __launch_bounds__(TBI *TBJ) void lp_dual_simplex(LpArgs g) {
    constexpr int NT = TBI * TBJ;
    constexpr int MP = TBI * R;
    constexpr int NP = TBJ * C;
    static_assert((NP & (NP - 1)) == 0, "padded column count must be a power of two");
    static_assert(NT % 64 == 0, "whole waves only");
    constexpr int PI = (MP + 63) / 64;  // rows per lane in a wave-wide scan
    constexpr int PJ = (NP + 63) / 64;  // columns per lane in a wave-wide scan
    __shared__ Smem<MP, NP> s;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int bi = tid / TBJ;
    const int bj = tid % TBJ;
    const int m = g.m, n = g.n;
    const int nv = n + m;
    const double INF = __builtin_huge_val();

    // one workgroup per node LP (no grid-stride loop: a loop here makes the compiler hoist every
    // per-element predicate and address of the setup across the whole solve -> register spills)
    const int node = blockIdx.x;
    if (node >= g.batch) return;
    {
        double T[R][C];
        const size_t src = g.slot ? (size_t)g.slot[node] : (size_t)node;
        const double *gA = g.A + (size_t)node * g.A_stride;
        const double *gb = g.b + (size_t)node * g.b_stride;
        const double *gc = g.c + (size_t)node * g.c_stride;
        const double *lk = g.l + src * n;
        const double *uk = g.u + src * n;
        const int8_t *vin = g.vstat_in ? g.vstat_in + src * nv : nullptr;

        // ---- 0. T = -A, beta0 = -b, d = c, slack basis (or the anchor's tableau state) -------
        const bool anchored = g.anchor_T != nullptr && vin != nullptr;
        const double sgn = anchored ? 1.0 : -1.0;
#pragma unroll
        for (int ii = 0; ii < R; ii++) {
            const int i = bi + TBI * ii;
            const double *arow = (anchored ? g.anchor_T : gA) + (size_t)(i < m ? i : 0) * n + bj;
#pragma unroll
            for (int jj = 0; jj < C; jj++) {
                const int j = bj + TBJ * jj;
                T[ii][jj] = (i < m && j < n) ? sgn * arow[TBJ * jj] : 0.0;
            }
        }
#pragma unroll 1
        for (int i = tid; i < MP; i += NT) {
            s.beta0[i] = i < m ? (anchored ? g.anchor_vec[n + i] : -gb[i]) : 0.0;
            s.bvar[i] = i < m ? (anchored ? g.anchor_idx[n + i] : n + i) : -1;
            s.ba[i] = 0.0;
            s.bb[i] = 0.0;
            s.entered[i] = 0;
        }
#pragma unroll 1
        for (int j = tid; j < NP; j += NT) {
            s.d[j] = j < n ? (anchored ? g.anchor_vec[j] : gc[j]) : 0.0;
            s.nvar[j] = j < n ? (anchored ? g.anchor_idx[j] : j) : -1;
            s.lo[j] = j < n ? lk[j] : 0.0;
            s.up[j] = j < n ? uk[j] : 0.0;
            s.side[j] = 0;
            s.va[j] = 0.0;
            s.vb[j] = 0.0;
        }
#pragma unroll 1
        for (int v = tid; v < NP + MP; v += NT) {
            int8_t st = (vin && v < nv) ? vin[v] : (int8_t)0;
            s.wantb[v] = st == 1;
            s.atup[v] = st == 2;
            s.pos[v] = -1;
        }
        __syncthreads();
        for (int j = tid; j < n; j += NT) s.pos[s.nvar[j]] = j;
        __syncthreads();
        if (tid < 64) {  // columns of the variables to pivot in, in ascending variable order
            int cnt = 0;
            for (int base = 0; base < nv; base += 64) {
                const int v = base + lane;
                const bool w = v < nv && s.wantb[v] && s.pos[v] >= 0;
                const unsigned long long mask = __ballot(w);
                if (w) s.wlist[cnt + __popcll(mask & ((1ull << lane) - 1ull))] = s.pos[v];
                cnt += __popcll(mask);
            }
            if (lane == 0) s.nw = cnt;
        }
        __syncthreads();
        const int nw = __builtin_amdgcn_readfirstlane(s.nw);

        int npiv = 0, iters = 0, status = -1;
        int phase = vin ? 0 : 1;  // 0 refactor, 1 value initialisation, 2 dual simplex
        int w = 0;
        const int cap = 100 * (m + n) + 1000;
        // Role split: the selections (pivot row / ratio test) are done by the "control" waves only
        // -- the first wave on each SIMD -- and published through LDS; their sibling waves on the
        // same SIMDs skip that work instead of repeating it (it would just double the SIMD's
        // instruction stream).  All waves take part in the tableau update.
        constexpr int CT = NT < 256 ? NT : 256;
        const bool ctl = tid < CT;
        int degen = 0;  // consecutive degenerate steps (control waves); > m+n -> Bland's rule

        for (;;) {
            int r = 0, q = 0, sigma = 1, newside = 0;
            double la = 0.0, lb = 0.0, pinv = 0.0;

            if (phase == 0) {
                // ---- 1. refactor: pivot the next wanted structural into the basis -----------
                if (w >= nw) {
                    if (g.refactor_only) { status = 3; break; }
                    phase = 1;
                    continue;
                }
                q = __builtin_amdgcn_readfirstlane(s.wlist[w]);
                w++;
                {   // column q -> s.alpha
                    const int qb = q % TBJ, ql = q / TBJ;
                    if (bj == qb) {
#pragma unroll
                        for (int jj = 0; jj < C; jj++)
                            if (jj == ql) {
#pragma unroll
                                for (int ii = 0; ii < R; ii++) s.alpha[bi + TBI * ii] = T[ii][jj];
                            }
                    }
                }
                __syncthreads();
                if (ctl) {
                    double k1 = -INF, k2 = -INF;   // preferred rows / any slack row
                    int p1 = kNoCand, p2 = kNoCand;
#pragma unroll
                    for (int kk = 0; kk < PI; kk++) {
                        const int i = lane + 64 * kk;
                        if (i < m) {
                            const bool wanted = s.wantb[s.bvar[i]];
                            const double a = fabs(s.alpha[i]);
                            const bool ok = a > kPivTol;
                            keep(k1, p1, a, i, ok && !wanted);
                            keep(k2, p2, a, i, ok && wanted && !s.entered[i]);
                        }
                    }
                    double km;
                    int rr = wave_argmax(k1, p1, km);
                    if (rr == kNoCand) rr = wave_argmax(k2, p2, km);
                    if (tid == 0) {
                        s.ci[0] = rr == kNoCand ? -1 : rr;
                        if (rr != kNoCand) s.cd[0] = 1.0 / s.alpha[rr];  // 1/p, p = T[r][q]
                    }
                }
                __syncthreads();
                r = __builtin_amdgcn_readfirstlane(s.ci[0]);
                if (r < 0) continue;  // singular: stays nonbasic
                pinv = s.cd[0];
                {   // row r -> s.row
                    const int rb = r % TBI, rl = r / TBI;
                    if (bi == rb) {
#pragma unroll
                        for (int ii = 0; ii < R; ii++)
                            if (ii == rl) {
#pragma unroll
                                for (int jj = 0; jj < C; jj++) s.row[bj + TBJ * jj] = T[ii][jj];
                            }
                    }
                }
            } else if (phase == 1) {
                // ---- 2. nonbasic sides, basic values ----------------------------------------
#pragma unroll 1
                for (int j = tid; j < n; j += NT) {
                    const int v = s.nvar[j];
                    const double lo = v < n ? s.lo[v] : 0.0, up = v < n ? s.up[v] : INF;
                    const double dj = s.d[j];
                    int side;
                    if (lo == up) side = 0;
                    else if (dj < -kDTol) side = isinf(up) ? 2 : 1;
                    else if (dj > kDTol) side = 0;
                    else side = (s.atup[v] && !isinf(up)) ? 1 : 0;
                    s.side[j] = side;
                    s.va[j] = side == 0 ? lo : side == 1 ? up : 0.0;
                    s.vb[j] = side == 2 ? 1.0 : 0.0;
                }
                __syncthreads();
                {
                    double va[C], vb[C];
#pragma unroll
                    for (int jj = 0; jj < C; jj++) {
                        va[jj] = s.va[bj + TBJ * jj];
                        vb[jj] = s.vb[bj + TBJ * jj];
                    }
#pragma unroll
                    for (int ii = 0; ii < R; ii++) {
                        double pa[C], pb[C];
#pragma unroll
                        for (int jj = 0; jj < C; jj++) {
                            pa[jj] = T[ii][jj] * va[jj];
                            pb[jj] = T[ii][jj] * vb[jj];
                        }
#pragma unroll
                        for (int h = C / 2; h >= 1; h >>= 1) {
#pragma unroll
                            for (int jj = 0; jj < h; jj++) {
                                pa[jj] = pa[jj] + pa[jj + h];
                                pb[jj] = pb[jj] + pb[jj + h];
                            }
                        }
                        double sa = pa[0], sb = pb[0];
#pragma unroll
                        for (int h = TBJ / 2; h >= 1; h >>= 1) {
                            sa = sa + __shfl_down(sa, h, TBJ);
                            sb = sb + __shfl_down(sb, h, TBJ);
                        }
                        const int i = bi + TBI * ii;
                        if (bj == 0 && i < m) {
                            s.ba[i] = s.beta0[i] - sa;
                            s.bb[i] = 0.0 - sb;
                        }
                        // keep the rows' temporaries from being interleaved (register pressure)
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
                __syncthreads();
                phase = 2;
                continue;
            } else {
                // ---- 3. dual simplex iteration ----------------------------------------------
                // (a) leaving row, by the control waves
                const bool bland = degen > m + n;
                if (ctl) {
                    int blevel = 0, bp = kNoCand;
                    double bk = -INF;
#pragma unroll
                    for (int kk = 0; kk < PI; kk++) {
                        const int i = lane + 64 * kk;
                        if (i < m) {
                            const int v = s.bvar[i];
                            const double lo = v < n ? s.lo[v] : 0.0, up = v < n ? s.up[v] : INF;
                            const double a = s.ba[i], bM = s.bb[i];
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
                            if (bland && level > 0) { level = 1; viol = 0.0; }  // lowest variable wins
                            // payload: variable index (tie-break), direction flag, row
                            const int pay = (v << 16) | (sg < 0 ? 0x8000 : 0) | i;
                            const bool up_lvl = level > blevel;
                            const bool same = level == blevel && level > 0 &&
                                              (viol > bk || (viol == bk && pay < bp));
                            if (up_lvl || same) { blevel = level; bk = viol; bp = pay; }
                        }
                    }
                    const int lvl = wave_max_i32(blevel);
                    int cmd = 0, win = 0;
                    if (lvl == 0) {
                        int bad = 0;
                        for (int i = lane; i < m; i += 64) bad |= s.bb[i] > kBTol;
                        for (int j = lane; j < n; j += 64) bad |= s.side[j] == 2;
                        cmd = __any(bad) ? 3 : 1;  // exit: unbounded / optimal
                    } else if ((g.max_iter > 0 && iters >= g.max_iter) || iters >= cap) {
                        cmd = 4;                   // exit: iteration limit
                    } else {
                        double km;
                        win = wave_argmax(blevel == lvl ? bk : -INF, blevel == lvl ? bp : kNoCand, km);
                    }
                    if (tid == 0) { s.ci[0] = cmd; s.ci[1] = win; }
                }
                __syncthreads();
                {
                    const int cmd = __builtin_amdgcn_readfirstlane(s.ci[0]);
                    if (cmd) { status = cmd == 1 ? 0 : cmd == 3 ? 2 : 3; break; }
                    const int win = __builtin_amdgcn_readfirstlane(s.ci[1]);
                    r = win & 0x7fff;
                    sigma = (win & 0x8000) ? -1 : 1;
                }
                {   // (b) row r -> s.row
                    const int rb = r % TBI, rl = r / TBI;
                    if (bi == rb) {
#pragma unroll
                        for (int ii = 0; ii < R; ii++)
                            if (ii == rl) {
#pragma unroll
                                for (int jj = 0; jj < C; jj++) s.row[bj + TBJ * jj] = T[ii][jj];
                            }
                    }
                }
                __syncthreads();
                // (c) Harris ratio test on row r, by the control waves
                if (ctl) {
#pragma unroll 1
                    for (int j = tid; j < n; j += CT) {
                        const int v = s.nvar[j];
                        const double lo = v < n ? s.lo[v] : 0.0, up = v < n ? s.up[v] : INF;
                        const double a = sigma * s.row[j];
                        const int sd = s.side[j];
                        const bool elig = lo != up && (sd == 0 ? (a < -kPivTol) : (a > kPivTol));
                        const double dj = sd == 0 ? fmax(s.d[j], 0.0) : fmax(-s.d[j], 0.0);
                        const double aa = fabs(a);
                        s.key[j] = elig ? (bland ? dj / aa : (dj + kDTol) / aa) : INF;
                        s.aabs[j] = elig ? aa : -1.0;
                        s.dje[j] = dj;
                    }
                }
                if (CT > 64) __syncthreads();  // a single control wave needs no barrier here
                if (ctl) {
                    double k1 = -INF;
                    int p1 = kNoCand;
#pragma unroll
                    for (int kk = 0; kk < PJ; kk++) {
                        const int j = lane + 64 * kk;
                        if (j < n) keep(k1, p1, -s.key[j], (s.nvar[j] << 16) | j, s.aabs[j] >= 0.0);
                    }
                    double nthmax;
                    const int w1 = wave_argmax(k1, p1, nthmax);
                    int qq = -1;
                    if (w1 != kNoCand && bland) {
                        qq = w1 & 0xffff;  // textbook ratio test, ties -> lowest variable index
                    } else if (w1 != kNoCand) {
                        const double thmax = -nthmax;
                        const int jmin = w1 & 0xffff;
                        double k2 = -INF;
                        int p2 = kNoCand;
#pragma unroll
                        for (int kk = 0; kk < PJ; kk++) {
                            const int j = lane + 64 * kk;
                            if (j < n) {
                                const double aa = s.aabs[j];
                                const bool ok = aa >= 0.0 && (j == jmin || !(s.dje[j] > thmax * aa));
                                keep(k2, p2, aa, (s.nvar[j] << 16) | j, ok);
                            }
                        }
                        double amax;
                        qq = wave_argmax(k2, p2, amax) & 0xffff;
                    }
                    if (qq >= 0) degen = s.dje[qq] <= kDTol ? degen + 1 : 0;
                    if (tid == 0) {
                        s.ci[0] = qq;
                        if (qq >= 0) s.cd[0] = 1.0 / s.row[qq];
                    }
                }
                __syncthreads();
                q = __builtin_amdgcn_readfirstlane(s.ci[0]);
                if (q < 0) { status = 1; break; }  // no entering column: primal infeasible
                pinv = s.cd[0];
                {   // (d) column q -> s.alpha
                    const int qb = q % TBJ, ql = q / TBJ;
                    if (bj == qb) {
#pragma unroll
                        for (int jj = 0; jj < C; jj++)
                            if (jj == ql) {
#pragma unroll
                                for (int ii = 0; ii < R; ii++) s.alpha[bi + TBI * ii] = T[ii][jj];
                            }
                    }
                }
                const int lv = s.bvar[r];
                const double lo = lv < n ? s.lo[lv] : 0.0, up = lv < n ? s.up[lv] : INF;
                if (sigma > 0) { la = lo; lb = 0.0; newside = 0; }
                else if (!isinf(up)) { la = up; lb = 0.0; newside = 1; }
                else { la = 0.0; lb = 1.0; newside = 2; }
            }

            // scalars every thread needs for the border updates: read BEFORE the barrier,
            // written (by their owner threads) only after it
            const double dq = s.d[q], b0r = s.beta0[r];
            const double bar = s.ba[r], bbr = s.bb[r], vaq = s.va[q], vbq = s.vb[q];
            __syncthreads();

            // ---- rank-1 update of the register tableau and the borders ----------------------
            {
                const bool vals = phase == 2;
                const int rb = r % TBI, rl = r / TBI;
                const int qb = q % TBJ, ql = q / TBJ;
                double al[R], rh[C];
#pragma unroll
                for (int ii = 0; ii < R; ii++) al[ii] = s.alpha[bi + TBI * ii];
#pragma unroll
                for (int jj = 0; jj < C; jj++) rh[jj] = s.row[bj + TBJ * jj] * pinv;
                // row r and column q come out of this as junk and are overwritten just below
#pragma unroll
                for (int ii = 0; ii < R; ii++) {
#pragma unroll
                    for (int jj = 0; jj < C; jj++) T[ii][jj] = fma(-al[ii], rh[jj], T[ii][jj]);
                }
                if (bj == qb) {  // column q <- -alpha * (1/p)
#pragma unroll
                    for (int jj = 0; jj < C; jj++)
                        if (jj == ql) {
#pragma unroll
                            for (int ii = 0; ii < R; ii++) T[ii][jj] = -al[ii] * pinv;
                        }
                }
                if (bi == rb) {  // row r <- row * (1/p), and 1/p at the pivot position
#pragma unroll
                    for (int ii = 0; ii < R; ii++)
                        if (ii == rl) {
#pragma unroll
                            for (int jj = 0; jj < C; jj++)
                                T[ii][jj] = (bj == qb && jj == ql) ? pinv : rh[jj];
                        }
                }
                const double rhon = b0r * pinv;
                const double ta = (bar - la) * pinv, tb = (bbr - lb) * pinv;
#pragma unroll 1
                for (int i = tid; i < m; i += NT) {
                    const double a = s.alpha[i];
                    if (i == r) {
                        s.beta0[i] = rhon;
                        if (vals) { s.ba[i] = vaq + ta; s.bb[i] = vbq + tb; }
                    } else {
                        s.beta0[i] = fma(-a, rhon, s.beta0[i]);
                        if (vals) {
                            s.ba[i] = fma(-a, ta, s.ba[i]);
                            s.bb[i] = fma(-a, tb, s.bb[i]);
                        }
                    }
                }
#pragma unroll 1
                for (int j = tid; j < n; j += NT) {
                    if (j == q) {
                        s.d[j] = -dq * pinv;
                        if (vals) { s.side[j] = newside; s.va[j] = la; s.vb[j] = lb; }
                    } else {
                        s.d[j] = fma(-dq, s.row[j] * pinv, s.d[j]);
                    }
                }
                if (tid == NT - 1) {
                    const int tmp = s.bvar[r];
                    s.bvar[r] = s.nvar[q];
                    s.nvar[q] = tmp;
                    if (!vals) s.entered[r] = 1;
                }
            }
            __syncthreads();
            npiv++;
            if (phase == 2) iters++;
        }

        // ---- 4. outputs --------------------------------------------------------------------
        // assemble x by variable index in s.key
        for (int j = tid; j < n; j += NT) {
            const int v = s.nvar[j];
            if (v < n) s.key[v] = s.side[j] == 2 ? kMReport : s.va[j];
        }
        for (int i = tid; i < m; i += NT) {
            const int v = s.bvar[i];
            if (v < n) s.key[v] = fma(s.bb[i], kMReport, s.ba[i]);
        }
        for (int j = n + tid; j < NP; j += NT) s.key[j] = 0.0;
        __syncthreads();
        if (g.x)
            for (int j = tid; j < n; j += NT) g.x[(size_t)node * n + j] = s.key[j];
        if (g.y) {
            for (int i = tid; i < m; i += NT) g.y[(size_t)node * m + i] = 0.0;
            __syncthreads();
            for (int j = tid; j < n; j += NT)
                if (s.nvar[j] >= n) g.y[(size_t)node * m + (s.nvar[j] - n)] = s.d[j];
        }
        if (g.vstat_out) {
            int8_t *vo = g.vstat_out + (size_t)node * nv;
            for (int i = tid; i < m; i += NT) vo[s.bvar[i]] = 1;
            for (int j = tid; j < n; j += NT) vo[s.nvar[j]] = s.side[j] ? 2 : 3;
        }
        if (g.dbg_T && (node == 0 || g.dbg_all)) {
            const size_t k = g.dbg_all ? (size_t)node : 0;
            double *dT = g.dbg_T + k * (size_t)m * n;
            double *dvec = g.dbg_vec + k * (size_t)(n + 3 * m);
            int32_t *didx = g.dbg_idx + k * (size_t)(2 * n + m);
#pragma unroll
            for (int ii = 0; ii < R; ii++) {
#pragma unroll
                for (int jj = 0; jj < C; jj++) {
                    const int i = bi + TBI * ii, j = bj + TBJ * jj;
                    if (i < m && j < n) dT[(size_t)i * n + j] = T[ii][jj];
                }
            }
            for (int j = tid; j < n; j += NT) {
                dvec[j] = s.d[j];
                didx[j] = s.nvar[j];
                didx[n + m + j] = s.side[j];
            }
            for (int i = tid; i < m; i += NT) {
                dvec[n + i] = s.beta0[i];
                dvec[n + m + i] = s.ba[i];
                dvec[n + 2 * m + i] = s.bb[i];
                didx[n + i] = s.bvar[i];
            }
        }
        if (tid < 64) {
            // obj = fold-in-half sum of c_j x_j over the padded power-of-two length
            constexpr int PER = NP / 64;
            double p[PER];
#pragma unroll
            for (int k = 0; k < PER; k++) {
                const int j = lane + 64 * k;
                p[k] = j < n ? gc[j] * s.key[j] : 0.0;
            }
#pragma unroll
            for (int h = PER / 2; h >= 1; h >>= 1) {
#pragma unroll
                for (int k = 0; k < h; k++) p[k] = p[k] + p[k + h];
            }
            double sum = p[0];
#pragma unroll
            for (int h = 32; h >= 1; h >>= 1) sum = sum + __shfl_down(sum, h, 64);
            if (tid == 0) {
                if (g.obj) g.obj[node] = status == 1 ? INF : sum;
                if (g.status) g.status[node] = status;
                if (g.iters) g.iters[node] = iters;
                if (g.npivots) g.npivots[node] = npiv;
            }
        }
        __syncthreads();
    }
}

}  // namespace mipx
