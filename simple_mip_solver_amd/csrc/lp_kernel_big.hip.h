// lp_kernel_big.hip.h -- K1b: the same canonical dual simplex as K1 for node LPs whose tableau
// does not fit the register file of one CU (above 256 x 192; BASELINE config C5 is 1024 x 512 =
// 4 MiB of tableau).  One workgroup of 1024 threads per LP; the tableau lives in a per-workgroup
// HBM scratch slab and is STREAMED once per pivot (one coalesced read + one write of every row):
// this is the regime SURVEY.md section 8(d) prices, 2*8*(m+1)(n+m+1) bytes per pivot, and the
// kernel is bound by HBM bandwidth.  Borders and the pivot row/column live in LDS (dynamic,
// ~100 KiB at 1024 x 512).  Same arithmetic and reductions as K1; the leaving row is priced with dual
// Devex weights (w_i from the pivot column alone: the inner products of K1's steepest edge would take
// another pass over the tableau).  Results are bit-identical to the oracle's for these shapes.
// Cut rows (LpArgs::ncut, as in K1's CUTS variant): node k has m + ncut[k] rows, the extra ones read from
// the cut store; slabs, LDS borders and the strided arrays are sized for mstride rows.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "lp_kernel.hip.h"

namespace mipx {

// Threads per workgroup.  One workgroup per CU either way (the LDS borders take ~100 KiB); with 512 threads
// a lane has 256 registers instead of 128 -- the two rows of 16-byte pairs a wave keeps in flight, the
// scaled pivot row and the fold arrays fit with 356 B of scratch instead of 816 -- and the kernel is 15 %
// faster at 1024 x 512 (74.6 -> 85.5 k LP/s; 256 threads: no scratch at all but too few loads in flight,
// 72.4 k; four rows in flight at 512 threads spill again, 76.2 k).  The arithmetic is per wave and per row:
// the results do not depend on it.
#ifndef MIPX_BIG_NT
#define MIPX_BIG_NT 512
#endif
constexpr int kBigNT = MIPX_BIG_NT;
constexpr int kBigMaxN = 1024;  // 16 elements per lane in the wave folds
constexpr int kBigMaxM = 1024;

__host__ __device__ inline size_t big_lds_bytes(int m, int n) {
    const size_t dbl = (size_t)n * 10 + (size_t)m * 5 + 8;   // row d va vb lo up key aabs dje x | alpha beta0 ba bb wgt | cd
    const size_t i32 = (size_t)n * 3 + (size_t)m + 8 + (size_t)(n + m);  // nvar side wlist | bvar | nw ci | pos
    const size_t i8 = 2 * (size_t)(n + m) + (size_t)m;       // wantb atup | entered
    return dbl * 8 + i32 * 4 + i8 + 64;
}

typedef double d2v __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(kBigNT) void lp_dual_simplex_big(LpArgs g, double *scratch) {
    constexpr int NT = kBigNT, NW = NT / 64, CT = 256;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m0 = g.m, n = g.n;
    const bool cuts = g.ncut != nullptr;
    const int mcap = cuts ? g.mstride : m0;   // rows a node can have: slab, LDS borders and strided arrays are sized for it
    const double INF = __builtin_huge_val();
    int n2 = 64;
    while (n2 < n) n2 <<= 1;
    const int PER = n2 / 64;            // <= 16
    // ---- LDS carve ----------------------------------------------------------------------------
    double *s_row = (double *)smem_raw;
    double *s_d = s_row + n, *s_va = s_d + n, *s_vb = s_va + n, *s_lo = s_vb + n, *s_up = s_lo + n;
    double *s_key = s_up + n, *s_aabs = s_key + n, *s_dje = s_aabs + n, *s_x = s_dje + n;
    double *s_alpha = s_x + n, *s_beta0 = s_alpha + mcap, *s_ba = s_beta0 + mcap, *s_bb = s_ba + mcap;
    double *s_wgt = s_bb + mcap;        // dual Devex weights of the rows (reference framework: 1 at the start of a node LP)
    double *s_cd = s_wgt + mcap;
    int *s_nvar = (int *)(s_cd + 8), *s_side = s_nvar + n, *s_wlist = s_side + n, *s_bvar = s_wlist + n;
    int *s_ci = s_bvar + mcap;          // [0..3] control words, [4] nw
    int *s_pos = s_ci + 8;              // column of each variable in the starting tableau, -1 if basic
    int8_t *s_wantb = (int8_t *)(s_pos + (n + mcap)), *s_atup = s_wantb + (n + mcap), *s_entered = s_atup + (n + mcap);
    double *T = scratch + (size_t)blockIdx.x * (size_t)mcap * n;

    for (int node = blockIdx.x; node < g.batch; node += gridDim.x) {
        if (g.active != nullptr && g.active[node] == 0) continue;   // (uniform over the workgroup)
        // rows: the m0 shared ones, then this node's cuts (rows of the cut store)
        const int kcut = cuts ? __builtin_amdgcn_readfirstlane(g.ncut[node]) : 0;
        const int m = m0 + kcut, nv = n + m;
        const int32_t *cids = cuts ? g.cut_ids + (size_t)node * g.cut_stride : nullptr;
        const int PIr = (m + 63) / 64;      // rows per lane in a wave scan
        const size_t src = g.slot ? (size_t)g.slot[node] : (size_t)node;
        const double *gA = g.A + (size_t)node * g.A_stride;
        const double *gb = g.b + (size_t)node * g.b_stride;
        const double *gc = g.c + (size_t)node * g.c_stride;
        const double *lk = g.l + src * n, *uk = g.u + src * n;
        const int8_t *vin = g.vstat_in ? g.vstat_in + ((cuts && g.vstat_by_node) ? (size_t)node : src) * (size_t)(n + mcap) : nullptr;
        // ---- 0. T = -A, beta0 = -b, d = c, slack basis (or the anchor's tableau state) -------
        // (the anchor may come from the engine's table: entry anchor_sel[node], as in K1)
        const int asel = (g.anchor_sel != nullptr && vin != nullptr) ? __builtin_amdgcn_readfirstlane(g.anchor_sel[node]) : -1;
        const double *aT = asel >= 0 ? g.atab_T + (size_t)asel * ((size_t)m * n) : g.anchor_T;
        const double *avec = asel >= 0 ? g.atab_vec + (size_t)asel * (size_t)(n + 3 * m) : g.anchor_vec;
        const int32_t *aidx = asel >= 0 ? g.atab_idx + (size_t)asel * (size_t)(2 * n + m) : g.anchor_idx;
        const bool anchored = aT != nullptr && vin != nullptr && kcut == 0;   // (an anchor has the shared rows only)
        if (anchored) {
            for (size_t e = tid; e < (size_t)m * n; e += NT) T[e] = aT[e];
        } else {
            for (size_t e = tid; e < (size_t)m0 * n; e += NT) T[e] = -gA[e];
            for (int i = m0; i < m; i++) {
                const double *cp = g.cut_pi + (size_t)cids[i - m0] * n;
                for (int j = tid; j < n; j += NT) T[(size_t)i * n + j] = -cp[j];
            }
        }
        for (int i = tid; i < m; i += NT) {
            s_beta0[i] = anchored ? avec[n + i] : (i < m0 ? -gb[i] : -g.cut_pi0[cids[i - m0]]);
            s_bvar[i] = anchored ? aidx[n + i] : n + i;
            s_ba[i] = 0.0; s_bb[i] = 0.0; s_entered[i] = 0;
        }
        for (int j = tid; j < n; j += NT) {
            s_d[j] = anchored ? avec[j] : gc[j];
            s_nvar[j] = anchored ? aidx[j] : j;
            s_lo[j] = lk[j]; s_up[j] = uk[j];
            s_side[j] = 0; s_va[j] = 0.0; s_vb[j] = 0.0;
        }
        for (int v = tid; v < nv; v += NT) {
            const int8_t st = vin ? vin[v] : (int8_t)0;
            s_wantb[v] = st == 1; s_atup[v] = st == 2;
            s_pos[v] = -1;
        }
        __syncthreads();
        for (int j = tid; j < n; j += NT) s_pos[s_nvar[j]] = j;
        __syncthreads();
        if (tid < 64) {  // columns of the variables to pivot in, in ascending variable order
            int cnt = 0;
            for (int base = 0; base < nv; base += 64) {
                const int v = base + lane;
                const bool w = v < nv && s_wantb[v] && s_pos[v] >= 0;
                const unsigned long long mask = __ballot(w);
                if (w) s_wlist[cnt + __popcll(mask & ((1ull << lane) - 1ull))] = s_pos[v];
                cnt += __popcll(mask);
            }
            if (lane == 0) s_ci[4] = cnt;
        }
        __syncthreads();
        const int nw = __builtin_amdgcn_readfirstlane(s_ci[4]);
        int npiv = 0, iters = 0, status = -1, phase = vin ? 0 : 1, w = 0, degen = 0;
        // a verdict (optimal / unbounded / infeasible) of a solve that has carried symbolic values a + b M is only
        // taken on values worked out afresh from the tableau (phase 3): the running updates of thousands of pivots
        // are not trusted with it.  An LP with finite bounds never carries one: nothing changes for it.
        int sym = 0, fresh = 1;
        const int cap = 100 * (m + n) + 1000;
        const bool ctl = tid < CT;
        // in-place dive (LpArgs::dive, like K1): pass 0 the node, pass 1 one child on the same slab
        if (g.dive && g.dive_preset && tid < g.dive) {
            g.status[(size_t)node + (size_t)(tid + 1) * (size_t)g.dive_off] = -1;
            g.dive_var[(size_t)tid * (size_t)g.dive_off + node] = -1;
        }
        if (g.zero16 != nullptr && node == 0 && tid < 4) g.zero16[tid] = 0;
        int pass = 0;
        size_t onode = (size_t)node;
        for (;;) {  // passes

        for (;;) {
            int r = 0, q = 0, sigma = 1, newside = 0;
            double la = 0.0, lb = 0.0, pinv = 0.0;
            if (phase == 0) {
                if (w >= nw) {
                    if (g.refactor_only) { status = 3; break; }
                    phase = 1;
                    continue;
                }
                q = __builtin_amdgcn_readfirstlane(s_wlist[w]);
                w++;
                for (int i = tid; i < m; i += NT) s_alpha[i] = T[(size_t)i * n + q];
                __syncthreads();
                if (ctl) {
                    double k1 = -INF, k2 = -INF;
                    int p1 = kNoCand, p2 = kNoCand;
#pragma unroll 1
                    for (int kk = 0; kk < PIr; kk++) {
                        const int i = lane + 64 * kk;
                        if (i < m) {  // rows whose basic variable is not wanted; else wanted, not yet pivoted in
                            const bool wanted = s_wantb[s_bvar[i]];
                            const double a = fabs(s_alpha[i]);
                            const bool ok = a > kPivTol;
                            keep(k1, p1, a, i, ok && !wanted);
                            keep(k2, p2, a, i, ok && wanted && !s_entered[i]);
                        }
                    }
                    double km;
                    int rr = wave_argmax(k1, p1, km);
                    if (rr == kNoCand) rr = wave_argmax(k2, p2, km);
                    if (tid == 0) {
                        s_ci[0] = rr == kNoCand ? -1 : rr;
                        if (rr != kNoCand) s_cd[0] = 1.0 / s_alpha[rr];
                    }
                }
                __syncthreads();
                r = __builtin_amdgcn_readfirstlane(s_ci[0]);
                if (r < 0) continue;
                pinv = s_cd[0];
                for (int j = tid; j < n; j += NT) s_row[j] = T[(size_t)r * n + j];
            } else if (phase == 1 || phase == 3) {
                int anyb = 0;
                if (phase == 1)
                for (int j = tid; j < n; j += NT) {
                    const int v = s_nvar[j];
                    const double lo = v < n ? s_lo[v] : 0.0, up = v < n ? s_up[v] : INF;
                    const double dj = s_d[j];
                    int side;
                    if (lo == up) side = 0;
                    else if (dj < -kDTol) side = isinf(up) ? 2 : 1;
                    else if (dj > kDTol) side = 0;
                    else side = (s_atup[v] && !isinf(up)) ? 1 : 0;
                    s_side[j] = side;
                    s_va[j] = side == 0 ? lo : side == 1 ? up : 0.0;
                    s_vb[j] = side == 2 ? 1.0 : 0.0;
                    anyb |= side == 2;
                }
                if (phase == 1) sym = __syncthreads_or(anyb); else __syncthreads();
                // beta = beta0 - T v  (fold-in-half tree over n2; lane holds j = lane + 64 k)
                for (int i = wave; i < m; i += NW) {
                    const double *Ti = T + (size_t)i * n;
                    double pa[16], pb[16];
#pragma unroll
                    for (int k = 0; k < 16; k++) {
                        const int j = lane + 64 * k;
                        const double t = (k < PER && j < n) ? Ti[j] : 0.0;
                        pa[k] = (k < PER && j < n) ? t * s_va[j] : 0.0;
                        pb[k] = (k < PER && j < n) ? t * s_vb[j] : 0.0;
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
                    if (lane == 0) { s_ba[i] = s_beta0[i] - sa; s_bb[i] = snap_m(0.0 - sb); }
                }
                if (phase == 1)
                    for (int i = tid; i < m; i += NT) s_wgt[i] = 1.0;   // Devex: a fresh reference framework
                __syncthreads();
                phase = 2;
                fresh = 1;
                continue;
            } else {
                const bool bland = degen > m + n;
                if (ctl) {
                    int blevel = 0, bp = kNoCand;
                    double bk = -INF;
#pragma unroll 1
                    for (int kk = 0; kk < PIr; kk++) {
                        const int i = lane + 64 * kk;
                        if (i < m) {
                            const int v = s_bvar[i];
                            const double lo = v < n ? s_lo[v] : 0.0, up = v < n ? s_up[v] : INF;
                            const double a = s_ba[i], bM = s_bb[i];
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
                            if (level > 0) viol = viol * viol / s_wgt[i];   // dual Devex pricing
                            if (bland && level > 0) { level = 1; viol = 0.0; }
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
                        for (int i = lane; i < m; i += 64) bad |= s_bb[i] > kBTol;
                        for (int j = lane; j < n; j += 64) bad |= s_side[j] == 2;
                        cmd = __any(bad) ? 3 : 1;
                    } else if ((g.max_iter > 0 && iters >= g.max_iter) || iters >= cap) {
                        cmd = 4;
                    } else {
                        double km;
                        win = wave_argmax(blevel == lvl ? bk : -INF, blevel == lvl ? bp : kNoCand, km);
                    }
                    if (tid == 0) { s_ci[0] = cmd; s_ci[1] = win; }
                }
                __syncthreads();
                {
                    const int cmd = __builtin_amdgcn_readfirstlane(s_ci[0]);
                    if (cmd && cmd != 4 && sym && !fresh) { phase = 3; continue; }
                    if (cmd) { status = cmd == 1 ? 0 : cmd == 3 ? 2 : 3; break; }
                    const int win = __builtin_amdgcn_readfirstlane(s_ci[1]);
                    r = win & 0x7fff;
                    sigma = (win & 0x8000) ? -1 : 1;
                }
                for (int j = tid; j < n; j += NT) s_row[j] = T[(size_t)r * n + j];
                __syncthreads();
                if (ctl) {
#pragma unroll 1
                    for (int j = tid; j < n; j += CT) {
                        const int v = s_nvar[j];
                        const double lo = v < n ? s_lo[v] : 0.0, up = v < n ? s_up[v] : INF;
                        const double a = sigma * s_row[j];
                        const int sd = s_side[j];
                        const bool elig = lo != up && (sd == 0 ? (a < -kPivTol) : (a > kPivTol));
                        const double dj = sd == 0 ? fmax(s_d[j], 0.0) : fmax(-s_d[j], 0.0);
                        const double aa = fabs(a);
                        s_key[j] = elig ? (bland ? dj / aa : (dj + kDTol) / aa) : INF;
                        s_aabs[j] = elig ? aa : -1.0;
                        s_dje[j] = dj;
                    }
                }
                __syncthreads();
                if (ctl) {
                    double k1 = -INF;
                    int p1 = kNoCand;
#pragma unroll 1
                    for (int kk = 0; kk < PER; kk++) {
                        const int j = lane + 64 * kk;
                        if (j < n) keep(k1, p1, -s_key[j], (s_nvar[j] << 16) | j, s_aabs[j] >= 0.0);
                    }
                    double nthmax;
                    const int w1 = wave_argmax(k1, p1, nthmax);
                    int qq = -1;
                    if (w1 != kNoCand && bland) {
                        qq = w1 & 0xffff;
                    } else if (w1 != kNoCand) {
                        const double thmax = -nthmax;
                        const int jmin = w1 & 0xffff;
                        double k2 = -INF;
                        int p2 = kNoCand;
#pragma unroll 1
                        for (int kk = 0; kk < PER; kk++) {
                            const int j = lane + 64 * kk;
                            if (j < n) {
                                const double aa = s_aabs[j];
                                const bool ok = aa >= 0.0 && (j == jmin || !(s_dje[j] > thmax * aa));
                                keep(k2, p2, aa, (s_nvar[j] << 16) | j, ok);
                            }
                        }
                        double amax;
                        qq = wave_argmax(k2, p2, amax) & 0xffff;
                    }
                    if (qq >= 0) degen = s_dje[qq] <= kDTol ? degen + 1 : 0;
                    if (tid == 0) {
                        s_ci[0] = qq;
                        if (qq >= 0) s_cd[0] = 1.0 / s_row[qq];
                    }
                }
                __syncthreads();
                q = __builtin_amdgcn_readfirstlane(s_ci[0]);
                if (q < 0 && sym && !fresh) { phase = 3; continue; }
                if (q < 0) { status = 1; break; }
                pinv = s_cd[0];
                for (int i = tid; i < m; i += NT) s_alpha[i] = T[(size_t)i * n + q];
                const int lv = s_bvar[r];
                const double lo = lv < n ? s_lo[lv] : 0.0, up = lv < n ? s_up[lv] : INF;
                if (sigma > 0) { la = lo; lb = 0.0; newside = 0; }
                else if (!isinf(up)) { la = up; lb = 0.0; newside = 1; }
                else { la = 0.0; lb = 1.0; newside = 2; sym = 1; }
                fresh = 0;
            }
            const double dq = s_d[q], b0r = s_beta0[r];
            const double bar = s_ba[r], bbr = s_bb[r], vaq = s_va[q], vbq = s_vb[q];
            const double wr = s_wgt[r];
            __syncthreads();
            {   // ---- stream the tableau: T_ij <- fma(-alpha_i, rho_j, T_ij) ---------------------
                const bool vals = phase == 2;
                if ((n & 1) == 0) {
                    // adjacent column pairs: one 16-byte load / store per lane (1 KiB per wave
                    // instruction), two rows of a wave in flight at a time
                    d2v rj[8];
#pragma unroll
                    for (int k = 0; k < 8; k++) {
                        const int j = 2 * lane + 128 * k;
                        rj[k].x = (2 * k < PER && j < n) ? s_row[j] * pinv : 0.0;
                        rj[k].y = (2 * k < PER && j + 1 < n) ? s_row[j + 1] * pinv : 0.0;
                    }
                    const int qk = q >> 7, ql = (q & 127) >> 1, qe = q & 1;
                    for (int i0 = wave; i0 < m; i0 += 2 * NW) {
                        const int i1 = i0 + NW;
                        d2v *T0 = reinterpret_cast<d2v *>(T + (size_t)i0 * n);
                        d2v *T1 = reinterpret_cast<d2v *>(T + (size_t)(i1 < m ? i1 : i0) * n);
                        const double a0 = s_alpha[i0], a1 = s_alpha[i1 < m ? i1 : i0];
                        d2v v0[8], v1[8];
#pragma unroll
                        for (int k = 0; k < 8; k++) {
                            const int jp = lane + 64 * k;  // pair index
                            const bool in = 2 * k < PER && 2 * jp < n;
                            v0[k] = in ? __builtin_nontemporal_load(T0 + jp) : d2v{0.0, 0.0};
                            v1[k] = (in && i1 < m) ? __builtin_nontemporal_load(T1 + jp) : d2v{0.0, 0.0};
                        }
#pragma unroll
                        for (int k = 0; k < 8; k++) {
                            const int jp = lane + 64 * k;
                            const bool in = 2 * k < PER && 2 * jp < n;
                            const bool qh = k == qk && lane == ql;  // this pair holds column q
                            d2v o0, o1;
                            o0.x = i0 == r ? rj[k].x : fma(-a0, rj[k].x, v0[k].x);
                            o0.y = i0 == r ? rj[k].y : fma(-a0, rj[k].y, v0[k].y);
                            o1.x = i1 == r ? rj[k].x : fma(-a1, rj[k].x, v1[k].x);
                            o1.y = i1 == r ? rj[k].y : fma(-a1, rj[k].y, v1[k].y);
                            if (qh) {
                                const double c0 = i0 == r ? pinv : -a0 * pinv, c1 = i1 == r ? pinv : -a1 * pinv;
                                if (qe == 0) { o0.x = c0; o1.x = c1; } else { o0.y = c0; o1.y = c1; }
                            }
                            if (in) __builtin_nontemporal_store(o0, T0 + jp);
                            if (in && i1 < m) __builtin_nontemporal_store(o1, T1 + jp);
                        }
                    }
                } else {
                double rj[16];
#pragma unroll
                for (int k = 0; k < 16; k++) {
                    const int j = lane + 64 * k;
                    rj[k] = (k < PER && j < n) ? s_row[j] * pinv : 0.0;
                }
                for (int i = wave; i < m; i += NW) {
                    double *Ti = T + (size_t)i * n;
                    const double a = s_alpha[i];
                    if (i == r) {
#pragma unroll
                        for (int k = 0; k < 16; k++) {
                            const int j = lane + 64 * k;
                            if (k < PER && j < n) Ti[j] = j == q ? pinv : rj[k];
                        }
                    } else {
#pragma unroll
                        for (int k = 0; k < 16; k++) {
                            const int j = lane + 64 * k;
                            if (k < PER && j < n) Ti[j] = j == q ? -a * pinv : fma(-a, rj[k], Ti[j]);
                        }
                    }
                }
                }
                const double rhon = b0r * pinv;
                const double ta = (bar - la) * pinv, tb = (bbr - lb) * pinv;
                for (int i = tid; i < m; i += NT) {
                    const double a = s_alpha[i];
                    // (Devex weights with the values: w_i <- max(w_i, (alpha_i / p)^2 w_r), w_r <- max(w_r / p^2, 1))
                    if (i == r) {
                        s_beta0[i] = rhon;
                        if (vals) {
                            s_ba[i] = vaq + ta; s_bb[i] = snap_m(vbq + tb);
                            const double w = (wr * pinv) * pinv;
                            s_wgt[i] = w < 1.0 ? 1.0 : w;
                        }
                    } else {
                        s_beta0[i] = fma(-a, rhon, s_beta0[i]);
                        if (vals) {
                            s_ba[i] = fma(-a, ta, s_ba[i]); s_bb[i] = snap_m(fma(-a, tb, s_bb[i]));
                            const double ratio = a * pinv;
                            const double w = (ratio * ratio) * wr;
                            s_wgt[i] = w > s_wgt[i] ? w : s_wgt[i];
                        }
                    }
                }
                for (int j = tid; j < n; j += NT) {
                    if (j == q) {
                        s_d[j] = -dq * pinv;
                        if (vals) { s_side[j] = newside; s_va[j] = la; s_vb[j] = lb; }
                    } else {
                        s_d[j] = fma(-dq, s_row[j] * pinv, s_d[j]);
                    }
                }
                if (tid == NT - 1) {
                    const int tmp = s_bvar[r]; s_bvar[r] = s_nvar[q]; s_nvar[q] = tmp;
                    if (!vals) s_entered[r] = 1;
                }
            }
            __syncthreads();
            npiv++;
            if (phase == 2) iters++;
        }

        // ---- outputs -----------------------------------------------------------------------------
        for (int j = tid; j < n; j += NT) {
            const int v = s_nvar[j];
            if (v < n) s_x[v] = s_side[j] == 2 ? kMReport : s_va[j];
        }
        for (int i = tid; i < m; i += NT) {
            const int v = s_bvar[i];
            if (v < n) s_x[v] = fma(s_bb[i], kMReport, s_ba[i]);
        }
        __syncthreads();
        if (g.x) for (int j = tid; j < n; j += NT) g.x[onode * n + j] = s_x[j];
        if (g.y) {
            for (int i = tid; i < m; i += NT) g.y[onode * mcap + i] = 0.0;
            __syncthreads();
            for (int j = tid; j < n; j += NT)
                if (s_nvar[j] >= n) g.y[onode * mcap + (s_nvar[j] - n)] = s_d[j];
        }
        if (g.vstat_out) {
            int8_t *vo = g.vstat_out + onode * (size_t)(n + mcap);
            for (int i = tid; i < m; i += NT) vo[s_bvar[i]] = 1;
            for (int j = tid; j < n; j += NT) vo[s_nvar[j]] = s_side[j] ? 2 : 3;
        }
        if (g.dbg_T && (node == 0 || g.dbg_all)) {
            const size_t k = g.dbg_all ? (size_t)node : 0;
            double *dT = g.dbg_T + k * (size_t)mcap * n;
            double *dvec = g.dbg_vec + k * (size_t)(n + 3 * mcap);
            int32_t *didx = g.dbg_idx + k * (size_t)(2 * n + mcap);
            for (size_t e = tid; e < (size_t)m * n; e += NT) dT[e] = T[e];
            for (int j = tid; j < n; j += NT) { dvec[j] = s_d[j]; didx[j] = s_nvar[j]; didx[n + m + j] = s_side[j]; }
            for (int i = tid; i < m; i += NT) {
                dvec[n + i] = s_beta0[i]; dvec[n + m + i] = s_ba[i]; dvec[n + 2 * m + i] = s_bb[i];
                didx[n + i] = s_bvar[i];
            }
        }
        if (tid < 64) {
            double p[16];
#pragma unroll
            for (int k = 0; k < 16; k++) {
                const int j = lane + 64 * k;
                p[k] = (k < PER && j < n) ? gc[j] * s_x[j] : 0.0;
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
                if (g.iters) g.iters[onode] = iters;
                if (g.npivots) g.npivots[onode] = npiv;
            }
            // the dive: K4's branching rule on the solution in s_x (wave 0), as in K1
            if (pass < g.dive) {
                int code = -1, ddir = 0;
                double dbound = 0.0;
                const double objv = __shfl(sum, 0, 64);
                if (status == 0 && objv < g.dive_cutoff) {
                    double bk = -INF;
                    int bp = kNoCand, nprobe = 0;
                    for (int base = 0; base < g.n_int; base += 64) {
                        const int k = base + lane;
                        bool need_probe = false;
                        if (k < g.n_int) {
                            const int i = g.int_idx[k];
                            const double v = s_x[i];
                            const double fl = floor(v), ce = ceil(v);
                            const double dist = fmin(v - fl, ce - v);
                            const bool frac = dist > kVarEps;
                            if (g.rule == 0) {
                                keep(bk, bp, dist, k, frac);
                            } else if (frac) {
                                if (g.has_entry[i]) keep(bk, bp, fmin(g.cost_r[i] * (ce - v), g.cost_l[i] * (v - fl)), k, true);
                                else need_probe = true;
                            }
                        }
                        nprobe += __popcll(__ballot(need_probe));
                    }
                    double km;
                    const int win = wave_argmax(bk, bp, km);
                    if (win != kNoCand && nprobe == 0) {
                        const int dvar = g.int_idx[win];
                        const double v = s_x[dvar];
                        const double fl = floor(v), ce = ceil(v);
                        if (g.rule == 0) ddir = (v - fl <= ce - v) ? 0 : 1;
                        else ddir = (g.cost_l[dvar] * (v - fl) <= g.cost_r[dvar] * (ce - v)) ? 0 : 1;
                        dbound = ddir == 0 ? fl : ce;
                        bool mine = false;  // a bound change in place needs the variable basic
                        for (int i = lane; i < m; i += 64) mine |= s_bvar[i] == dvar;
                        if (__any(mine)) {
                            code = dvar;
                            if (lane == 0) {
                                const size_t di = (size_t)pass * (size_t)g.dive_off + node;
                                g.dive_var[di] = dvar;
                                g.dive_dir[di] = ddir;
                                g.dive_val[di] = v;
                                if (ddir == 0) s_up[dvar] = dbound;
                                else s_lo[dvar] = dbound;
                            }
                        }
                    }
                }
                if (lane == 0) s_ci[0] = code;
            }
        }
        __syncthreads();
        if (pass >= g.dive) break;
        if (__builtin_amdgcn_readfirstlane(s_ci[0]) < 0) break;
        pass++;
        onode += (size_t)g.dive_off;
        npiv = 0; iters = 0; degen = 0; status = -1; phase = 2;  // (straight back into the iterations)
        }  // passes
    }
}

}  // namespace mipx
