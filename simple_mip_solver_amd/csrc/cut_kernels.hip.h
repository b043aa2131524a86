// cut_kernels.hip.h -- per-node cutting-plane kernels that run alongside K1:
//   K2 gomory_cuts: Gomory mixed-integer cuts from the optimal tableau with the slacks substituted
//      out (reference nodes/base_node.py:468-511) and their numerically safe rounding to ratios of
//      small integers by continued fractions (utils/floating_point.py:40-167, as called at
//      base_node.py:381 with estimate='over').
//   K3 select_cuts: depth ordering, size / ratio / parallelism filters (base_node.py:387-466).
// The tableau is the one K1 leaves behind (T, bvar, nvar dumped to HBM); one workgroup per node.
// Arithmetic follows the canonical order of oracle/mipx_oracle.c (no contraction).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "lp_kernel.hip.h"

namespace mipx {

constexpr double kGoodEps = 1e-2;    // good_coefficient_approximation_epsilon (tolerance.py:5)
constexpr double kExactEps = 1e-14;  // exact_coefficient_approximation_epsilon (tolerance.py:8)
constexpr double kVarEpsCut = 1e-4;  // variable_epsilon (tolerance.py:2)
enum { kEstNone = 0, kEstOver = 1, kEstUnder = 2 };

// utils/floating_point.py:106-167 (numerators/denominators as doubles holding integers)
__device__ inline void get_fraction_dev(double x, double max_term, int estimate, double &n_out,
                                        double &d_out) {
    if (fabs(x) > max_term) {
        n_out = estimate == kEstOver ? ceil(x) : estimate == kEstUnder ? floor(x) : rint(x);
        d_out = 1.0;
        return;
    }
    // only the last three convergents are ever needed
    double n2 = 0.0, d2 = 1.0;   // h_{k-2}
    double n1 = 1.0, d1 = 0.0;   // h_{k-1}
    double n0 = 0.0, d0 = 0.0;   // h_k
    double n3 = 0.0, d3 = 0.0;   // h_{k-3} (for the parity step-back)
    int last = -1;               // index of h_k
    bool exact = false;
    double value = x;
    for (int it = 0; it < 68; it++) {
        const double whole = floor(value);
        n3 = n2; d3 = d2;
        n0 = whole * n1 + n2;
        d0 = whole * d1 + d2;
        last++;
        const bool over_limit = n0 > max_term || d0 > max_term;
        const double rem = value - whole;
        if (over_limit || rem == 0.0 || it == 67) {
            exact = !over_limit && rem == 0.0;
            break;
        }
        n2 = n1; d2 = d1;
        n1 = n0; d1 = d0;
        value = 1.0 / rem;
    }
    // at the break: h_last = (n0,d0), h_{last-1} = (n1,d1), h_{last-2} = (n2,d2) ... but n2/n1 were
    // not shifted on the breaking iteration, so: prev = (n1,d1), prev-1 = (n2,d2)
    (void)n3; (void)d3;
    if (exact) { n_out = n0; d_out = d0; return; }
    const int prev = last - 1;
    int pick;  // 0 -> prev, 1 -> prev-1
    if (estimate == kEstNone) pick = 0;
    else if (estimate == kEstOver) {
        if (prev % 2 != 0) pick = 0;
        else if (prev >= 1) pick = 1;
        else { n_out = ceil(x); d_out = 1.0; return; }
    } else {
        pick = (prev % 2 == 0) ? 0 : 1;
    }
    n_out = pick == 0 ? n1 : n2;
    d_out = pick == 0 ? d1 : d2;
}

// one coefficient of numerically_safe_cut (utils/floating_point.py:77-92): the directed estimate,
// replaced by the undirected one where that is exact to 1e-14 and the directed one is > 1 % off
__device__ inline void safe_coef_dev(double coef, double max_term, int estimate, double &nn, double &dd) {
    get_fraction_dev(coef, max_term, estimate, nn, dd);
    if (coef != 0.0 && fabs(1.0 - ((nn / dd) / coef)) > kGoodEps) {
        double n2, d2;
        get_fraction_dev(coef, max_term, kEstNone, n2, d2);
        if (fabs(n2 / d2 - coef) < kExactEps) { nn = n2; dd = d2; }
    }
}

// ---- the rounding stage on its own (mipx_safe_cut_batch / mipx_get_fraction_batch): the same device
// functions K2 rounds with, on supplied cuts, so that they can be pinned against the reference's
// vectors (tests/golden/floating_point.json) in isolation ------------------------------------------
struct SafeCutArgs {
    int n, batch;
    const double *pi, *pi0;        // batch x n, batch
    int estimate;                  // kEstOver / kEstUnder (the right-hand side gets the other one)
    int make_integer;              // multiply through by the lcm of the denominators
    double max_term;
    double *safe_pi, *safe_pi0;    // batch x n, batch
    double *num, *den;             // batch x (n + 1): per coefficient, then the right-hand side (may be null)
    double *scaled_pi, *scaled_pi0;  // scale_cut's output (may be null)
    int32_t *nonzero;              // batch: 0 where pi == 0 (the cut is returned unchanged)
};

__global__ __launch_bounds__(256) void safe_cut_batch(SafeCutArgs g) {
    __shared__ double red[8];
    __shared__ double dens[256];
    __shared__ long long lcm_s;
    const int k = blockIdx.x;
    if (k >= g.batch) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, n = g.n;
    const double INF = __builtin_huge_val();
    const double *pi = g.pi + (size_t)k * n;
    double *out = g.safe_pi + (size_t)k * n;
    // scale = min_j |1 / pi_j| (scale_cut, utils/floating_point.py:33-35; zeros give inf)
    double smin = INF;
    bool any = false;
    for (int j = tid; j < n; j += 256) {
        const double c = pi[j];
        any |= c != 0.0;
        smin = fmin(smin, fabs(1.0 / c));
    }
    smin = -wave_max_f64(-smin);
    const int anyw = __any(any);
    if (lane == 0) { red[wave] = smin; red[4 + wave] = anyw; }
    __syncthreads();
    double scale = INF;
    bool nonzero = false;
    for (int w = 0; w < 4; w++) { scale = fmin(scale, red[w]); nonzero |= red[4 + w] != 0.0; }
    if (tid == 0 && g.nonzero) g.nonzero[k] = nonzero ? 1 : 0;
    if (!nonzero) {
        for (int j = tid; j < n; j += 256) {
            out[j] = pi[j];
            if (g.num) { g.num[(size_t)k * (n + 1) + j] = pi[j]; g.den[(size_t)k * (n + 1) + j] = 1.0; }
            if (g.scaled_pi) g.scaled_pi[(size_t)k * n + j] = pi[j];
        }
        if (tid == 0) {
            g.safe_pi0[k] = g.pi0[k];
            if (g.num) { g.num[(size_t)k * (n + 1) + n] = g.pi0[k]; g.den[(size_t)k * (n + 1) + n] = 1.0; }
            if (g.scaled_pi0) g.scaled_pi0[k] = g.pi0[k];
        }
        return;
    }
    // every coefficient; with make_integer the quotients wait for the lcm of the denominators
    if (tid == 0) lcm_s = 1;
    __syncthreads();
    for (int base = 0; base < n; base += 256) {
        const int j = base + tid;
        double nn = 0.0, dd = 1.0;
        if (j < n) {
            const double coef = pi[j] * scale;
            safe_coef_dev(coef, g.max_term, g.estimate, nn, dd);
            if (g.scaled_pi) g.scaled_pi[(size_t)k * n + j] = coef;
            if (g.num) { g.num[(size_t)k * (n + 1) + j] = nn; g.den[(size_t)k * (n + 1) + j] = dd; }
            if (!g.make_integer) out[j] = nn / dd;
        }
        if (g.make_integer) {  // np.lcm.reduce over the denominators, in order (int64), by one thread
            dens[tid] = j < n ? dd : 1.0;
            __syncthreads();
            if (tid == 0) {
                long long l = lcm_s;
                for (int q = 0; q < 256 && base + q < n; q++) {
                    const long long a = l < 0 ? -l : l, b = (long long)fabs(dens[q]);
                    long long x = a, y = b;
                    while (y != 0) { const long long t = x % y; x = y; y = t; }
                    l = x == 0 ? 0 : (a / x) * b;
                }
                lcm_s = l;
            }
            __syncthreads();
        }
    }
    __syncthreads();
    const long long lcm = lcm_s;
    if (g.make_integer) {
        for (int j = tid; j < n; j += 256) {
            double nn, dd;
            safe_coef_dev(pi[j] * scale, g.max_term, g.estimate, nn, dd);
            // (lcm * n) / d as numpy does it: an int64 product, then a true division
            out[j] = (double)(lcm * (long long)nn) / dd;
        }
    }
    if (tid == 0) {
        const double s0 = g.pi0[k] * scale;
        if (g.scaled_pi0) g.scaled_pi0[k] = s0;
        double n0, d0;
        // (the reference passes no max_term for the right-hand side: its default, tolerance.max_term)
        get_fraction_dev(g.make_integer ? s0 * (double)lcm : s0, 1e3,
                         g.estimate == kEstOver ? kEstUnder : kEstOver, n0, d0);
        g.safe_pi0[k] = n0 / d0;
        if (g.num) { g.num[(size_t)k * (n + 1) + n] = n0; g.den[(size_t)k * (n + 1) + n] = d0; }
    }
}

struct FractionArgs {
    int count;
    const double *x, *max_term;
    const int32_t *estimate;
    double *num, *den;
};

__global__ __launch_bounds__(256) void get_fraction_batch(FractionArgs g) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= g.count) return;
    double nn, dd;
    get_fraction_dev(g.x[i], g.max_term[i], g.estimate[i], nn, dd);
    g.num[i] = nn;
    g.den[i] = dd;
}

constexpr int kGomoryGroup = 8;
// dynamic LDS of gomory_cuts for n columns, ms allotted rows and `group` cuts at a time
inline size_t gomory_lds_bytes(int n, int ms, int group) {
    return ((size_t)group * (n + ms) + ms + 128) * 8 + (3 * (size_t)ms + n) * 4 + 64;
}
// the largest group whose staging fits the LDS a workgroup may take: 48 KiB where several workgroups should
// share a CU; with no more workgroups than CUs (few nodes of a large shape: 1024 x 512 stages 12.5 KiB per cut)
// most of the CU's 160 KiB -- a group of 8 walks the rows of A 4 x less often than a group of 2
inline int gomory_group(int n, int ms, long workgroups = 1 << 30) {
    const size_t cap = workgroups <= 256 ? 144 * 1024 : workgroups <= 512 ? 72 * 1024 : 48 * 1024;
    int grp = kGomoryGroup;
    while (grp > 1 && gomory_lds_bytes(n, ms, grp) > cap) grp--;
    return grp;
}

struct GomoryArgs {
    int m, n, batch;
    const double *A, *b;          // shared rows (m x n), rhs
    const double *T;              // batch x m x n   (dump of K1)
    const int32_t *idx;           // batch x (2n+m): [nvar (n) | bvar (m) | side (n)]
    const double *x;              // batch x n  solution (already clipped at 0)
    const uint8_t *is_int;        // n
    double max_term;              // tolerance.max_term
    int32_t *ncuts;               // batch
    int32_t *row_idx;             // batch x m : rank of the generating basic variable
    double *pi, *pi0;             // batch x m x n, batch x m : raw GMI cuts  pi.x >= pi0
    double *safe_pi, *safe_pi0;   // rounded ('over' coefficients, 'under' right-hand side)
    int chunks = 1;               // workgroups per node: cut c of a node is worked out by workgroup c % chunks
    int group = 1;                // cuts a workgroup substitutes the slacks of at once (<= kGomoryGroup; LDS: gomory_lds_bytes)
    // EXPERIMENT (MIPX_K2_MFMA=1, never the default): the slack substitution pi + A' pi_s of a group as
    // v_mfma_f64_16x16x4_f64 tiles (cuts x columns, reduced over rows four at a time, fused) instead of the
    // reference's row-by-row multiply-then-add.  Another summation order: the raw coefficients differ in
    // their last bits from the canonical ones (tests/test_cut_kernels_gpu.py reports by how much).
    int mfma = 0;
    // ---- frontier engine with cut rounds (all optional) ------------------------------------------
    // per-node cut rows, as in LpArgs: node k has m + ncut[k] rows, row m + i = cut cut_ids[k * cut_stride + i]
    const int32_t *ncut = nullptr, *cut_ids = nullptr;
    const double *cut_pi = nullptr, *cut_pi0 = nullptr;
    int cut_stride = 0;
    int mstride = 0;              // rows allotted per node in T / idx / row_idx (0: m)
    const int8_t *vstat = nullptr;  // batch x (n + mstride): no cuts unless exactly m + ncut entries are basic
                                    //   (the reference's tableau is None then, base_node.py:518-519)
    int clip_x = 0;               // read x as max(x, 0) (base_node.py:310)
    const int32_t *active = nullptr;  // skip nodes with active[k] == 0
    // a node's rounded cuts go to its slab of the engine's cut pool instead of safe_pi: cut c to row
    // slab_n[k] + c of the slab_rows rows that start at row k * slab_rows of slab_pi / slab_pi0
    double *slab_pi = nullptr, *slab_pi0 = nullptr;
    const int32_t *slab_n = nullptr;
    int slab_rows = 0;
};

template <int NT>
__global__ __launch_bounds__(NT) void gomory_cuts(GomoryArgs g) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    // a node's cuts are independent of each other: with few nodes in the batch (the per-node Python
    // path: one) `chunks` workgroups share them, every one walking the same row order
    const int node = blockIdx.x / g.chunks, chunk = blockIdx.x % g.chunks;
    if (node >= g.batch) return;
    if (g.active != nullptr && g.active[node] == 0) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m0 = g.m, n = g.n;
    const int kcut = g.ncut ? g.ncut[node] : 0;
    const int m = m0 + kcut;                       // this node's rows: the shared ones, then its cuts
    const int ms = g.mstride ? g.mstride : m0;     // rows allotted per node in the strided arrays
    const int32_t *cids = g.ncut ? g.cut_ids + (size_t)node * g.cut_stride : nullptr;
    const double INF = __builtin_huge_val();
    // dynamic LDS carve (sized for ms rows): group x (pi_var[n] | ps[ms]) | cut_f0[ms] | red[128] |
    // order[ms] | bvar[ms] | nvar[n] | cut_rank[ms]      (gomory_lds_bytes)
    double *pi_var = (double *)smem_raw;
    double *cut_f0 = pi_var + (size_t)g.group * (n + ms);
    double *red = cut_f0 + ms;
    int *order = (int *)(red + 128);  // order[rank] = tableau row
    int *bvar_s = order + ms;
    int *nvar_s = bvar_s + ms;
    int *cut_rank = nvar_s + n;
    const int32_t *idx = g.idx + (size_t)node * (2 * n + ms);
    const double *T = g.T + (size_t)node * ms * n;
    const double *x = g.x + (size_t)node * n;
    for (int j = tid; j < n; j += NT) nvar_s[j] = idx[j];
    for (int i = tid; i < m; i += NT) bvar_s[i] = idx[n + i];
    if (g.vstat != nullptr) {  // a basis that is not square gives no tableau, hence no cuts
        const int8_t *vs = g.vstat + (size_t)node * (n + ms);
        int cnt = 0;
        for (int v = tid; v < n + m; v += NT) cnt += vs[v] == 1;
        for (int h = 32; h >= 1; h >>= 1) cnt += __shfl_down(cnt, h, 64);
        if (lane == 0) red[wave] = (double)cnt;
        __syncthreads();
        int total = 0;
        for (int w = 0; w < NT / 64; w++) total += (int)red[w];
        __syncthreads();
        if (total != m) {
            if (tid == 0 && chunk == 0) g.ncuts[node] = 0;
            return;
        }
    }
    __syncthreads();
    // rank of each basic variable among the basics (row order of inv(A_B) in the reference)
    for (int i = tid; i < m; i += NT) {
        const int v = bvar_s[i];
        int rank = 0;
        for (int k = 0; k < m; k++) rank += bvar_s[k] < v;
        order[rank] = i;
    }
    __syncthreads();
    // ---- which cuts there are (uniform over the workgroup), and which are this workgroup's ---------
    // rows whose basic variable is an integer structural with a fractional value (f0 in [0.01, 0.99])
    // (one rank per thread, compacted in rank order: a ballot per wave, the waves' counts through LDS)
    int ncuts = 0;
    for (int base = 0; base < m; base += NT) {
        const int rank = base + tid;
        bool gen = false;
        double f0 = 0.0;
        if (rank < m) {
            const int r = order[rank];
            const int v = bvar_s[r];
            gen = v < n && g.is_int[v < n ? v : 0];
            if (gen) {
                const double xv = g.clip_x ? fmax(x[v], 0.0) : x[v];
                const double fl = floor(xv), ce = ceil(xv);
                gen = fmin(xv - fl, ce - xv) > kVarEpsCut;
                f0 = xv - fl;
                if (f0 < kGoodEps || f0 + kGoodEps > 1.0) gen = false;
            }
        }
        const unsigned long long bal = __ballot(gen);
        if (lane == 0) red[wave] = (double)__popcll(bal);
        __syncthreads();
        int before = 0, total = 0;
        for (int w = 0; w < NT / 64; w++) {
            const int c = (int)red[w];
            before += w < wave ? c : 0;
            total += c;
        }
        if (gen) {
            const int pos = ncuts + before + __popcll(bal & ((1ull << lane) - 1ull));
            cut_rank[pos] = rank;
            cut_f0[pos] = f0;
        }
        ncuts += total;
        __syncthreads();
    }
    // ---- this workgroup's cuts, GRP at a time: the slack substitution of a group reads every row of
    // A once for all of them (one cut at a time re-read the 256 KiB of A from L2 per cut: the kernel
    // was bound by that).  Per cut the arithmetic and its order are unchanged.
    const int GRP = g.group;
    for (int c0 = chunk; c0 < ncuts; c0 += g.chunks * GRP) {
        int gc = 0;   // cuts in this group: c0, c0 + chunks, ...
        while (gc < GRP && c0 + gc * g.chunks < ncuts) gc++;
        for (int q = 0; q < GRP; q++) {   // (every slot of the group: the substitution below runs over all of them)
            double *pv = pi_var + (size_t)q * (n + ms), *psq = pv + n;
            for (int j = tid; j < n; j += NT) pv[j] = 0.0;
            for (int i = tid; i < m; i += NT) psq[i] = 0.0;
        }
        __syncthreads();
        for (int q = 0; q < gc; q++) {
            const int cq = c0 + q * g.chunks;
            const int r = order[cut_rank[cq]];
            const double f0 = cut_f0[cq];
            double *pv = pi_var + (size_t)q * (n + ms), *psq = pv + n;
            const double *Tr = T + (size_t)r * n;
            for (int j = tid; j < n; j += NT) {
                const int var = nvar_s[j];
                const double a = Tr[j];
                const double cont = a > 0.0 ? a / f0 : -a / (1.0 - f0);
                if (var < n) {
                    double val = cont;
                    if (g.is_int[var]) {
                        const double f = a - floor(a);
                        val = f <= f0 ? f / f0 : (1.0 - f) / (1.0 - f0);
                    }
                    pv[var] = val;
                } else {
                    psq[var - n] = cont;
                }
            }
        }
        __syncthreads();
        if (g.mfma) {
            // one 16 x 16 tile = 16 cuts (the group's, zero-padded) x 16 columns per wave and step of four rows:
            // A operand ps[cut = lane & 15][row = 4 kb + (lane >> 4)] from LDS, B operand A[row][col0 + (lane & 15)]
            // from L2, D: column lane & 15, cuts (lane >> 4) + 4 r
            typedef double d4v __attribute__((ext_vector_type(4)));
            const int cq_ = lane & 15, kq_ = lane >> 4;
            for (int tile = wave; tile * 16 < n; tile += NT / 64) {
                const int var = tile * 16 + cq_;
                d4v acc = {0.0, 0.0, 0.0, 0.0};
                for (int i0 = 0; i0 < m; i0 += 4) {
                    const int i = i0 + kq_;
                    const double a_op = (cq_ < gc && i < m) ? pi_var[(size_t)cq_ * (n + ms) + n + i] : 0.0;
                    double b_op = 0.0;
                    if (i < m && var < n) b_op = i < m0 ? g.A[(size_t)i * n + var] : g.cut_pi[(size_t)cids[i - m0] * n + var];
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a_op, b_op, acc, 0, 0, 0);
                }
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int q = kq_ + 4 * r;
                    if (q < gc && var < n) {
                        double *pv = pi_var + (size_t)q * (n + ms);
                        const double coef = pv[var] + acc[r];
                        if (g.pi) g.pi[((size_t)node * ms + (c0 + q * g.chunks)) * n + var] = coef;
                        pv[var] = coef;
                    }
                }
            }
        } else
        // coefs = pi + A' ps, accumulated row by row (the order of the reference's sparse product);
        // the node's cut rows follow the shared ones
        for (int var = tid; var < n; var += NT) {
            double acc[kGomoryGroup];
#pragma unroll
            for (int q = 0; q < kGomoryGroup; q++) acc[q] = 0.0;
            int i = 0;
            if (GRP == kGomoryGroup && ((n + ms) & 1) == 0 && (n & 1) == 0) {
                // a full group: every slot unconditionally (the empty ones hold zeros and are never read back), the
                // multipliers of 8 rows x 8 cuts fetched as 16-byte LDS reads BEFORE the first multiply -- a test and
                // an LDS wait per term made this loop 5 x longer than its multiply-adds.  Per cut and column the
                // same terms in the same order.
                typedef double d2s __attribute__((ext_vector_type(2)));
                for (; i + 8 <= m0; i += 8) {
                    double a[8];
#pragma unroll
                    for (int k = 0; k < 8; k++) a[k] = g.A[(size_t)(i + k) * n + var];
                    d2s p[kGomoryGroup][4];
#pragma unroll
                    for (int q = 0; q < kGomoryGroup; q++) {
                        const d2s *pp = reinterpret_cast<const d2s *>(pi_var + (size_t)q * (n + ms) + n + i);
#pragma unroll
                        for (int h = 0; h < 4; h++) p[q][h] = pp[h];
                    }
#pragma unroll
                    for (int k = 0; k < 8; k++) {
#pragma unroll
                        for (int q = 0; q < kGomoryGroup; q++)
                            acc[q] = acc[q] + a[k] * ((k & 1) ? p[q][k >> 1].y : p[q][k >> 1].x);
                    }
                }
            }
            for (; i + 8 <= m0; i += 8) {   // (the loads of 8 rows are issued before the first is consumed)
                double a[8];
#pragma unroll
                for (int k = 0; k < 8; k++) a[k] = g.A[(size_t)(i + k) * n + var];
#pragma unroll
                for (int k = 0; k < 8; k++) {
#pragma unroll
                    for (int q = 0; q < kGomoryGroup; q++)
                        if (q < gc) acc[q] = acc[q] + a[k] * pi_var[(size_t)q * (n + ms) + n + i + k];
                }
            }
            for (; i < m; i++) {
                const double a = i < m0 ? g.A[(size_t)i * n + var] : g.cut_pi[(size_t)cids[i - m0] * n + var];
#pragma unroll
                for (int q = 0; q < kGomoryGroup; q++)
                    if (q < gc) acc[q] = acc[q] + a * pi_var[(size_t)q * (n + ms) + n + i];
            }
#pragma unroll
            for (int q = 0; q < kGomoryGroup; q++) {
                if (q < gc) {
                    double *pv = pi_var + (size_t)q * (n + ms);
                    const double coef = pv[var] + acc[q];
                    if (g.pi) g.pi[((size_t)node * ms + (c0 + q * g.chunks)) * n + var] = coef;
                    pv[var] = coef;   // keep for the rounding below
                }
            }
        }
        __syncthreads();
        // ---- the group's right-hand sides, scales and rounding: per cut the arithmetic of the reference,
        // but the cuts of a group go through each stage together (three barriers per group, not per cut)
        constexpr int NW = NT / 64;
        double *rhs_s = red;               // [kGomoryGroup]
        double *smin_s = red + kGomoryGroup;                 // [kGomoryGroup][NW]
        double *any_s = red + kGomoryGroup * (1 + NW);        // [kGomoryGroup][NW]
        // rhs = 1 + ps . b with a fold-in-half tree over the next power of two: wave w takes cuts w, w + NW, ..
        for (int q = wave; q < gc; q += NW) {
            const int cq = c0 + q * g.chunks;
            const double *psq = pi_var + (size_t)q * (n + ms) + n;
            int m2 = 1;
            while (m2 < m) m2 <<= 1;
            double part = 0.0;
            auto rhs_of = [&](int j) { return j < m0 ? g.b[j] : g.cut_pi0[cids[j - m0]]; };
            if (m2 <= 64) {
                part = lane < m ? psq[lane] * rhs_of(lane) : 0.0;
                for (int h = m2 / 2; h >= 1; h >>= 1) part = part + __shfl_down(part, h, 64);
            } else {
                // lane holds elements lane + 64*k: fold the k levels in registers, then across lanes
                double e[16];
                const int per = m2 / 64;  // <= 16 for m <= 1024
                for (int k = 0; k < 16; k++) {
                    const int j = lane + 64 * k;
                    e[k] = (k < per && j < m) ? psq[j] * rhs_of(j) : 0.0;
                }
                for (int h = per / 2; h >= 1; h >>= 1)
                    for (int k = 0; k < h; k++) e[k] = e[k] + e[k + h];
                part = e[0];
                for (int h = 32; h >= 1; h >>= 1) part = part + __shfl_down(part, h, 64);
            }
            if (lane == 0) {
                const double rhs = 1.0 + part;
                if (g.pi0) g.pi0[(size_t)node * ms + cq] = rhs;
                if (g.row_idx) g.row_idx[(size_t)node * ms + cq] = cut_rank[cq];
                rhs_s[q] = rhs;
            }
        }
        // scale = min_j |1 / coef_j| (min is order independent): the waves' minima through LDS
        for (int q = 0; q < gc; q++) {
            const double *pv = pi_var + (size_t)q * (n + ms);
            double smin = INF;
            bool any = false;
            for (int var = tid; var < n; var += NT) {
                const double c = pv[var];
                any |= c != 0.0;
                smin = fmin(smin, fabs(1.0 / c));
            }
            smin = -wave_max_f64(-smin);
            const int anyw = __any(any);
            if (lane == 0) { smin_s[q * NW + wave] = smin; any_s[q * NW + wave] = anyw; }
        }
        __syncthreads();
        // numerically safe rounding (estimate 'over'; rhs 'under'), cut after cut without a barrier
        for (int q = 0; q < gc; q++) {
            const int cq = c0 + q * g.chunks;
            const double *pv = pi_var + (size_t)q * (n + ms);
            double scale = INF;
            bool nonzero = false;
            for (int wv = 0; wv < NW; wv++) { scale = fmin(scale, smin_s[q * NW + wv]); nonzero |= any_s[q * NW + wv] != 0.0; }
            // where the rounded cut goes: row cq of the node's block of safe_pi, or (engine) the next
            // free rows of the node's pool slab; a full slab drops the cut
            double *out_sp = nullptr, *out_s0 = nullptr;
            if (g.slab_pi != nullptr) {
                const int row = g.slab_n[node] + cq;
                if (row < g.slab_rows) {
                    out_sp = g.slab_pi + ((size_t)node * g.slab_rows + row) * n;
                    out_s0 = g.slab_pi0 + (size_t)node * g.slab_rows + row;
                }
            } else {
                out_sp = g.safe_pi + ((size_t)node * ms + cq) * n;
                out_s0 = g.safe_pi0 + (size_t)node * ms + cq;
            }
            if (out_sp != nullptr) {   // (uniform)
                if (!nonzero) {
                    for (int var = tid; var < n; var += NT) out_sp[var] = pv[var];
                    if (tid == q) *out_s0 = rhs_s[q];
                } else {
                    for (int var = tid; var < n; var += NT) {
                        const double coef = pv[var] * scale;
                        double nn, dd;
                        safe_coef_dev(coef, g.max_term, kEstOver, nn, dd);
                        out_sp[var] = nn / dd;
                    }
                    if (tid == q) {   // (the right-hand sides of the group's cuts on different threads)
                        double n0, d0;
                        get_fraction_dev(rhs_s[q] * scale, 1e3, kEstUnder, n0, d0);
                        *out_s0 = n0 / d0;
                    }
                }
            }
        }
        __syncthreads();
    }
    // (engine: the new slab rows join the node's pool in pool_append, after every workgroup of the
    // node is done with slab_n)
    if (tid == 0 && chunk == 0) g.ncuts[node] = ncuts;
}

}  // namespace mipx

namespace mipx {

// ---- K3: cut selection (reference base_node.py:387-466) -----------------------------------------
// One workgroup (256 threads) per node.  Per pool cut: support count, euclidean depth of the
// violation, norm and largest |coefficient| (fold-in-half sums over the padded power of two);
// then the reference's greedy pass in ascending depth (stable), with the parallelism test done as
// cos(angle) > cos(parallel_cut_tolerance) instead of acos (same predicate, no libm dependence).
struct SelectArgs {
    int n, batch, kmax;           // kmax: pool rows allotted per node
    const int32_t *npool;         // batch: cuts in each node's pool
    const double *pi, *pi0;       // batch x kmax x n, batch x kmax
    const double *x;              // batch x n
    int max_nonzero_coefs;
    double min_cut_depth, cos_parallel, max_abs_coef;  // max_relative_cut_term_ratio * max_term
    int32_t *nadded;              // batch
    int32_t *added;               // batch x kmax: pool positions in the order they are added
    int32_t *terminator;          // batch: 0 none, 1 'no cuts', 2 'no improving cuts', 3 'no sufficient cuts'
    double *depth;                // batch x kmax (NaN-free: +inf for cuts that are not candidates); may be null
    // frontier engine (optional): pool entry k of a node is row pool_list[node * kmax + k] of the
    // node's kmax-row slab of pi / pi0 (slab rows are never moved; the list keeps the pool's order)
    const int32_t *pool_list = nullptr;
    int clip_x = 0;               // read x as max(x, 0) (base_node.py:310)
    const int32_t *active = nullptr;  // skip nodes with active[k] == 0
};

// fold-in-half sum of f(j) over j < n2 (power of two, >= 64) by one wave; result in every lane
// (PER = n2 / 64 is a compile-time constant: with a run-time trip count the 16-entry array lived in
// scratch memory and every fold was a round trip through it)
template <int PER>
__device__ __forceinline__ double wave_fold_regs(double (&e)[PER]) {
#pragma unroll
    for (int h = PER / 2; h >= 1; h >>= 1) {
#pragma unroll
        for (int k = 0; k < h; k++) e[k] = e[k] + e[k + h];
    }
    double s = e[0];
    for (int h = 32; h >= 1; h >>= 1) s = s + __shfl_down(s, h, 64);
    return __shfl(s, 0, 64);
}

// per-cut statistics of one pool cut by one wave: support, largest |coefficient|, pi . x, pi . pi
template <int PER>
__device__ __forceinline__ void cut_stats(const double *pk, const double *x, int clip_x, int n, int lane,
                                          int &sup, double &mx, double &dot, double &sq) {
    double v[PER], t[PER], q[PER];
    sup = 0;
    mx = 0.0;
#pragma unroll
    for (int k = 0; k < PER; k++) {
        const int j = lane + 64 * k;
        v[k] = j < n ? pk[j] : 0.0;
        const double xv = j < n ? (clip_x ? fmax(x[j], 0.0) : x[j]) : 0.0;
        sup += (v[k] > kGoodEps) + (v[k] < -kGoodEps);
        mx = fmax(mx, fabs(v[k]));
        t[k] = j < n ? v[k] * xv : 0.0;
        q[k] = j < n ? v[k] * v[k] : 0.0;
    }
    for (int h = 32; h >= 1; h >>= 1) sup += __shfl_down(sup, h, 64);
    sup = __shfl(sup, 0, 64);
    mx = wave_max_f64(mx);
    dot = wave_fold_regs<PER>(t);
    sq = wave_fold_regs<PER>(q);
}

template <int PER>
__device__ __forceinline__ double cut_dot(const double *pk, const double *pa, int n, int lane) {
    double t[PER];
#pragma unroll
    for (int k = 0; k < PER; k++) {
        const int j = lane + 64 * k;
        t[k] = j < n ? pk[j] * pa[j] : 0.0;
    }
    return wave_fold_regs<PER>(t);
}
#define MIPX_PER_DISPATCH(per_, call_)     \
    switch (per_) {                        \
    case 1: { constexpr int PER = 1; call_; break; }   \
    case 2: { constexpr int PER = 2; call_; break; }   \
    case 4: { constexpr int PER = 4; call_; break; }   \
    case 8: { constexpr int PER = 8; call_; break; }   \
    default: { constexpr int PER = 16; call_; break; } \
    }

__global__ __launch_bounds__(256) void select_cuts(SelectArgs g) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int node = blockIdx.x;
    if (node >= g.batch) return;
    if (g.active != nullptr && g.active[node] == 0) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = g.n, K = g.npool[node];
    const double INF = __builtin_huge_val();
    const int32_t *plist = g.pool_list ? g.pool_list + (size_t)node * g.kmax : nullptr;
    auto row_of = [&](int k) { return plist ? plist[k] : k; };
    double *dep = (double *)smem_raw;   // K
    double *nrm = dep + g.kmax;         // K
    double *mab = nrm + g.kmax;         // K
    int *ord = (int *)(mab + g.kmax);   // K: candidates in ascending depth (stable)
    int *add = ord + g.kmax;            // K
    int *cnt = add + g.kmax;            // [ncand, nadded]
    int n2 = 64;
    while (n2 < n) n2 <<= 1;
    const double *P = g.pi + (size_t)node * g.kmax * n;
    const double *x = g.x + (size_t)node * n;
    // per-cut statistics, one wave per cut
    for (int k = wave; k < K; k += 4) {
        const double *pk = P + (size_t)row_of(k) * n;
        int sup;
        double mx, dot, sq;
        MIPX_PER_DISPATCH(n2 >> 6, (cut_stats<PER>(pk, x, g.clip_x, n, lane, sup, mx, dot, sq)));
        if (lane == 0) {
            const double nr = sqrt(sq);
            nrm[k] = nr;
            mab[k] = mx;
            dep[k] = (sup > 0 && sup <= g.max_nonzero_coefs) ? (dot - g.pi0[(size_t)node * g.kmax + row_of(k)]) / nr : INF;
        }
    }
    __syncthreads();
    // stable ascending order of the candidates (rank = #smaller + #equal-before)
    if (tid == 0) { cnt[0] = 0; cnt[1] = 0; }
    __syncthreads();
    for (int k = tid; k < K; k += 256) {
        if (dep[k] == INF) continue;
        int rank = 0;
        for (int q = 0; q < K; q++)
            if (dep[q] != INF && (dep[q] < dep[k] || (dep[q] == dep[k] && q < k))) rank++;
        ord[rank] = k;
        atomicAdd(&cnt[0], 1);
    }
    __syncthreads();
    const int ncand = cnt[0];
    int term = 0;
    if (ncand == 0) term = 1;
    else if (dep[ord[0]] >= 0.0) term = 2;
    else if (dep[ord[0]] >= -g.min_cut_depth) term = 3;
    // greedy pass: sequential over candidates, each test parallel over the cuts already added
    int nadd = 0;
    for (int c = 0; c < ncand; c++) {
        const int k = ord[c];
        if (dep[k] >= -g.min_cut_depth) break;
        if (mab[k] > g.max_abs_coef) continue;
        const double *pk = P + (size_t)row_of(k) * n;
        // dot products with the added cuts: one wave per added cut
        __syncthreads();
        if (tid == 0) cnt[1] = 0;
        __syncthreads();
        for (int a = wave; a < nadd; a += 4) {
            const double *pa = P + (size_t)row_of(add[a]) * n;
            double dot;
            MIPX_PER_DISPATCH(n2 >> 6, (dot = cut_dot<PER>(pk, pa, n, lane)));
            if (lane == 0) {
                double cs = dot / (nrm[k] * nrm[add[a]]);
                cs = fmin(1.0, fmax(-1.0, cs));
                if (cs > g.cos_parallel) atomicAdd(&cnt[1], 1);
            }
        }
        __syncthreads();
        if (cnt[1] == 0) {
            if (tid == 0) add[nadd] = k;
            nadd++;
        }
    }
    __syncthreads();
    for (int k = tid; k < g.kmax; k += 256) {
        if (g.depth) g.depth[(size_t)node * g.kmax + k] = k < K ? dep[k] : INF;
        if (k < nadd) g.added[(size_t)node * g.kmax + k] = add[k];
    }
    if (tid == 0) { g.nadded[node] = nadd; g.terminator[node] = term; }
}
#undef MIPX_PER_DISPATCH

}  // namespace mipx
