// tree_kernels.hip.h -- per-node kernels that run alongside K1 in the frontier engine:
//   K4 branch_score: integrality test, most-fractional / pseudo-cost branching index, list of
//      indices that still need strong-branching initialisation
//      (reference: nodes/base_node.py:281-283, :544-562; nodes/branch/pseudo_cost.py:57-59, :118-133)
//   K5 make_children: child node records = parent's bounds with one bound moved + parent's optimal
//      basis as warm start (reference: nodes/base_node.py:592-608)
// One wave per node for K4 (wavefront-wide DPP reductions), one workgroup per child for K5.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "lp_kernel.hip.h"

namespace mipx {

struct ScoreArgs {
    int n, n_int, batch, rule;      // rule 0: most fractional, 1: pseudo cost
    const int32_t *int_idx;         // n_int
    const double *x;                // batch x n
    const int32_t *status;          // batch (Clp codes)
    const double *cost_l, *cost_r;  // n (pseudo-cost table by variable)
    const uint8_t *has_entry;       // n
    int32_t *branch_idx;            // batch: variable to branch on or -1
    double *branch_val;             // batch: x[branch_idx] (b_val of the children)
    int32_t *mip_feasible;          // batch
    int32_t *n_probe;               // batch
    int32_t *probe_list;            // batch x n_int (ascending position in int_idx)
    // compact copy of the probe requests for the host (read with the step's single D2H): one
    // entry per (node, unprobed fractional variable), a node's entries contiguous and ascending;
    // ask_count may exceed ask_cap (then nothing past the cap is written and the host falls back
    // to probe_list).  ask == nullptr: not wanted (re-scoring).
    int32_t *ask_count;
    int32_t ask_cap;
    int32_t ask_nodes;              // only nodes < ask_nodes file requests
    struct Ask { int32_t node, k; double x; } *ask;
};

__global__ __launch_bounds__(64) void branch_score(ScoreArgs g) {
    const int node = blockIdx.x;
    if (node >= g.batch) return;
    const int lane = threadIdx.x;
    const double INF = __builtin_huge_val();
    const int st = g.status[node];
    if (!(st == 0 || st == 2)) {  // not lp_feasible: nothing to score
        if (lane == 0) { g.branch_idx[node] = -1; g.mip_feasible[node] = 0; g.n_probe[node] = 0; }
        return;
    }
    const double *x = g.x + (size_t)node * g.n;
    double worst = 0.0;           // max |round(x) - x|
    double bk = -INF;             // best key
    int bp = kNoCand;             // payload: position in int_idx (ties -> earliest)
    int nprobe = 0;
    for (int base = 0; base < g.n_int; base += 64) {
        const int k = base + lane;
        bool need_probe = false;
        if (k < g.n_int) {
            const int i = g.int_idx[k];
            const double v = x[i];
            const double fl = floor(v), ce = ceil(v);
            const double dist = fmin(v - fl, ce - v);
            worst = fmax(worst, fabs(rint(v) - v));
            const bool frac = dist > kVarEps;
            if (g.rule == 0) {
                keep(bk, bp, dist, k, frac);
            } else if (frac) {
                if (g.has_entry[i]) {
                    const double sc = fmin(g.cost_r[i] * (ce - v), g.cost_l[i] * (v - fl));
                    keep(bk, bp, sc, k, true);
                } else {
                    need_probe = true;
                }
            }
        }
        const unsigned long long mask = __ballot(need_probe);
        if (need_probe)
            g.probe_list[(size_t)node * g.n_int + nprobe + __popcll(mask & ((1ull << lane) - 1ull))] = k;
        nprobe += __popcll(mask);
    }
    const double wmax = wave_max_f64(worst);
    double km;
    const int win = wave_argmax(bk, bp, km);
    if (lane == 0) {
        g.mip_feasible[node] = wmax <= kVarEps;
        const int bvar = win == kNoCand ? -1 : g.int_idx[win];
        g.branch_idx[node] = bvar;
        g.branch_val[node] = bvar < 0 ? 0.0 : x[bvar];
        g.n_probe[node] = nprobe;
    }
    if (g.ask != nullptr && nprobe > 0 && node < g.ask_nodes) {  // wave-uniform; rare once the table has filled
        int off = 0;
        if (lane == 0) off = atomicAdd(g.ask_count, nprobe);
        off = __shfl(off, 0);
        if (off + nprobe <= g.ask_cap) {
            for (int base = 0; base < g.n_int; base += 64) {
                const int k = base + lane;
                bool need_probe = false;
                double v = 0.0;
                if (k < g.n_int) {
                    const int i = g.int_idx[k];
                    v = x[i];
                    need_probe = fmin(v - floor(v), ceil(v) - v) > kVarEps && !g.has_entry[i];
                }
                const unsigned long long mask = __ballot(need_probe);
                if (need_probe) {
                    ScoreArgs::Ask &e = g.ask[off + __popcll(mask & ((1ull << lane) - 1ull))];
                    e.node = node; e.k = k; e.x = v;
                }
                off += __popcll(mask);
            }
        }
    }
}

struct ChildArgs {
    int n, m, count;               // count = number of (parent, variable) pairs
    const double *src_l, *src_u;   // parent pool
    const int32_t *parent_slot;    // count: row of the parent in the source pool
    const int32_t *parent_pos;     // count: position of the parent in the dense batch outputs
    const int32_t *var;            // count: branching variable
    const double *x;               // batch x n (dense outputs of K1)
    const int8_t *vstat;           // batch x (n+m) (dense outputs of K1): warm start
    double *dst_l, *dst_u;         // destination pool
    int8_t *dst_v;
    const int32_t *child_slot;     // 2*count: left (x <= floor) then right (x >= ceil)
};

__global__ __launch_bounds__(256) void make_children(ChildArgs g) {
    const int c = blockIdx.x;  // child number: pair = c / 2, direction = c % 2
    if (c >= 2 * g.count) return;
    const int pair = c >> 1, right = c & 1;
    const int n = g.n, nv = g.n + g.m;
    const size_t ps = (size_t)g.parent_slot[pair], pos = (size_t)g.parent_pos[pair];
    const size_t ds = (size_t)g.child_slot[c];
    const int j = g.var[pair];
    const double xv = g.x[pos * n + j];
    for (int k = threadIdx.x; k < n; k += blockDim.x) {
        double lo = g.src_l[ps * n + k], up = g.src_u[ps * n + k];
        if (k == j) {
            if (right) lo = ceil(xv);
            else up = floor(xv);
        }
        g.dst_l[ds * n + k] = lo;
        g.dst_u[ds * n + k] = up;
    }
    for (int k = threadIdx.x; k < nv; k += blockDim.x) g.dst_v[ds * nv + k] = g.vstat[pos * nv + k];
}

}  // namespace mipx
