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
};

template <int NT>
__global__ __launch_bounds__(NT) void gomory_cuts(GomoryArgs g) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int node = blockIdx.x;
    if (node >= g.batch) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m = g.m, n = g.n;
    const double INF = __builtin_huge_val();
    // dynamic LDS carve: pi_var[n] | ps[m] | red[64] | order[m] | basic flag not needed
    double *pi_var = (double *)smem_raw;
    double *ps = pi_var + n;
    double *red = ps + m;
    int *order = (int *)(red + 64);   // order[rank] = tableau row
    int *bvar_s = order + m;
    int *nvar_s = bvar_s + m;
    const int32_t *idx = g.idx + (size_t)node * (2 * n + m);
    const double *T = g.T + (size_t)node * m * n;
    const double *x = g.x + (size_t)node * n;
    for (int j = tid; j < n; j += NT) nvar_s[j] = idx[j];
    for (int i = tid; i < m; i += NT) bvar_s[i] = idx[n + i];
    __syncthreads();
    // rank of each basic variable among the basics (row order of inv(A_B) in the reference)
    for (int i = tid; i < m; i += NT) {
        const int v = bvar_s[i];
        int rank = 0;
        for (int k = 0; k < m; k++) rank += bvar_s[k] < v;
        order[rank] = i;
    }
    __syncthreads();
    int ncuts = 0;
    for (int rank = 0; rank < m; rank++) {
        const int r = order[rank];
        const int v = bvar_s[r];
        bool gen = v < n && g.is_int[v < n ? v : 0];
        double f0 = 0.0;
        if (gen) {
            const double xv = x[v];
            const double fl = floor(xv), ce = ceil(xv);
            gen = fmin(xv - fl, ce - xv) > kVarEpsCut;
            f0 = xv - fl;
            if (f0 < kGoodEps || f0 + kGoodEps > 1.0) gen = false;
        }
        if (!gen) continue;  // uniform: depends on shared data only
        for (int j = tid; j < n; j += NT) pi_var[j] = 0.0;
        for (int i = tid; i < m; i += NT) ps[i] = 0.0;
        __syncthreads();
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
                pi_var[var] = val;
            } else {
                ps[var - n] = cont;
            }
        }
        __syncthreads();
        // coefs = pi + A' ps, accumulated row by row (the order of the reference's sparse product)
        double *out_pi = g.pi + ((size_t)node * m + ncuts) * n;
        for (int var = tid; var < n; var += NT) {
            double acc = 0.0;
            for (int i = 0; i < m; i++) acc = acc + g.A[(size_t)i * n + var] * ps[i];
            const double coef = pi_var[var] + acc;
            out_pi[var] = coef;
            pi_var[var] = coef;   // keep for the rounding below
        }
        // rhs = 1 + ps . b with a fold-in-half tree over the next power of two (wave 0)
        if (wave == 0) {
            int m2 = 1;
            while (m2 < m) m2 <<= 1;
            // each lane folds its strided elements first (j and j + m2/2 ... down to 64 lanes)
            double part = 0.0;
            if (m2 <= 64) {
                part = lane < m ? ps[lane] * g.b[lane] : 0.0;
                for (int h = m2 / 2; h >= 1; h >>= 1) part = part + __shfl_down(part, h, 64);
            } else {
                // lane holds elements lane + 64*k: fold the k levels in registers, then across lanes
                double e[8];
                const int per = m2 / 64;  // <= 8 for m <= 512
                for (int k = 0; k < 8; k++) {
                    const int j = lane + 64 * k;
                    e[k] = (k < per && j < m) ? ps[j] * g.b[j] : 0.0;
                }
                for (int h = per / 2; h >= 1; h >>= 1)
                    for (int k = 0; k < h; k++) e[k] = e[k] + e[k + h];
                part = e[0];
                for (int h = 32; h >= 1; h >>= 1) part = part + __shfl_down(part, h, 64);
            }
            if (lane == 0) {
                const double rhs = 1.0 + part;
                g.pi0[(size_t)node * m + ncuts] = rhs;
                g.row_idx[(size_t)node * m + ncuts] = rank;
                red[0] = rhs;
            }
        }
        __syncthreads();
        // ---- numerically safe rounding (estimate 'over'; rhs 'under') -----------------------
        // scale = min_j |1 / coef_j|
        double smin = INF;
        bool any = false;
        for (int var = tid; var < n; var += NT) {
            const double c = pi_var[var];
            any |= c != 0.0;
            smin = fmin(smin, fabs(1.0 / c));
        }
        // block reduction (min is order independent)
        smin = -wave_max_f64(-smin);
        const int anyw = __any(any);
        if (lane == 0) { red[1 + wave] = smin; red[33 + wave] = anyw; }
        __syncthreads();
        double scale = INF;
        bool nonzero = false;
        for (int wv = 0; wv < NT / 64; wv++) { scale = fmin(scale, red[1 + wv]); nonzero |= red[33 + wv] != 0.0; }
        double *out_sp = g.safe_pi + ((size_t)node * m + ncuts) * n;
        if (!nonzero) {
            for (int var = tid; var < n; var += NT) out_sp[var] = pi_var[var];
            if (tid == 0) g.safe_pi0[(size_t)node * m + ncuts] = red[0];
        } else {
            for (int var = tid; var < n; var += NT) {
                const double coef = pi_var[var] * scale;
                double nn, dd;
                get_fraction_dev(coef, g.max_term, kEstOver, nn, dd);
                if (coef != 0.0 && fabs(1.0 - ((nn / dd) / coef)) > kGoodEps) {
                    double n2, d2;
                    get_fraction_dev(coef, g.max_term, kEstNone, n2, d2);
                    if (fabs(n2 / d2 - coef) < kExactEps) { nn = n2; dd = d2; }
                }
                out_sp[var] = nn / dd;
            }
            if (tid == 0) {
                double n0, d0;
                get_fraction_dev(red[0] * scale, 1e3, kEstUnder, n0, d0);
                g.safe_pi0[(size_t)node * m + ncuts] = n0 / d0;
            }
        }
        __syncthreads();
        ncuts++;
    }
    if (tid == 0) g.ncuts[node] = ncuts;
}

}  // namespace mipx
