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
    // cut rounds (optional): rows allotted per node in vstat / dst_v (0: m), and the parent's cut list
    // (dense, by parent_pos) copied to the child (by child_slot): base_node.py:602-606 copies every row
    int mstride = 0, kc = 0;
    const int32_t *src_ncut = nullptr, *src_ids = nullptr;
    int32_t *dst_ncut = nullptr, *dst_ids = nullptr;
};

__global__ __launch_bounds__(256) void make_children(ChildArgs g) {
    const int c = blockIdx.x;  // child number: pair = c / 2, direction = c % 2
    if (c >= 2 * g.count) return;
    const int pair = c >> 1, right = c & 1;
    const int n = g.n;
    const size_t ps = (size_t)g.parent_slot[pair], pos = (size_t)g.parent_pos[pair];
    const int kcut = g.src_ncut ? g.src_ncut[pos] : 0;
    const int nv = g.n + g.m + kcut;                             // entries of the parent's basis
    const size_t vs = (size_t)(g.n + (g.mstride ? g.mstride : g.m));  // entries allotted per node
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
    for (int k = threadIdx.x; k < nv; k += blockDim.x) g.dst_v[ds * vs + k] = g.vstat[pos * vs + k];
    if (g.src_ncut != nullptr) {
        for (int k = threadIdx.x; k < kcut; k += blockDim.x) g.dst_ids[ds * g.kc + k] = g.src_ids[pos * g.kc + k];
        if (threadIdx.x == 0) g.dst_ncut[ds] = kcut;
    }
}

// ---- node migration between ranks (multi-GPU): pool rows <-> one contiguous message -----------------
struct PackArgs {
    int n, nvs, count;             // nvs: basis entries allotted per node (n + rows)
    size_t rowbytes;               // 16 n + nvs rounded up to 8
    const int32_t *slot;           // count pool rows
    double *pool_l, *pool_u;
    int8_t *pool_v;
    char *msg;
};
__global__ __launch_bounds__(256) void pack_nodes(PackArgs g) {
    const int k = blockIdx.x;
    if (k >= g.count) return;
    const size_t s = (size_t)g.slot[k];
    double *row = (double *)(g.msg + (size_t)k * g.rowbytes);
    for (int j = threadIdx.x; j < g.n; j += 256) {
        row[j] = g.pool_l[s * g.n + j];
        row[g.n + j] = g.pool_u[s * g.n + j];
    }
    int8_t *v = (int8_t *)(row + 2 * g.n);
    for (int j = threadIdx.x; j < g.nvs; j += 256) v[j] = g.pool_v[s * g.nvs + j];
}
__global__ __launch_bounds__(256) void unpack_nodes(PackArgs g) {
    const int k = blockIdx.x;
    if (k >= g.count) return;
    const size_t s = (size_t)g.slot[k];
    const double *row = (const double *)(g.msg + (size_t)k * g.rowbytes);
    for (int j = threadIdx.x; j < g.n; j += 256) {
        g.pool_l[s * g.n + j] = row[j];
        g.pool_u[s * g.n + j] = row[g.n + j];
    }
    const int8_t *v = (const int8_t *)(row + 2 * g.n);
    for (int j = threadIdx.x; j < g.nvs; j += 256) g.pool_v[s * g.nvs + j] = v[j];
}

// bounds and basis of pool rows without cut rows as dense arrays over the shared rows only (the re-anchoring
// launch of a tree with cut rounds: its pool keeps n + mstride basis codes per node)
struct PlainGatherArgs {
    int n, m, nvs_pool, count;
    const int32_t *slot;
    const double *pool_l, *pool_u;
    const int8_t *pool_v;
    double *l, *u;      // count x n
    int8_t *v;          // count x (n + m)
};
__global__ __launch_bounds__(256) void gather_plain_nodes(PlainGatherArgs g) {
    const int k = blockIdx.x;
    if (k >= g.count) return;
    const size_t s = (size_t)g.slot[k];
    for (int j = threadIdx.x; j < g.n; j += 256) {
        g.l[(size_t)k * g.n + j] = g.pool_l[s * g.n + j];
        g.u[(size_t)k * g.n + j] = g.pool_u[s * g.n + j];
    }
    for (int j = threadIdx.x; j < g.n + g.m; j += 256) g.v[(size_t)k * (g.n + g.m) + j] = g.pool_v[s * g.nvs_pool + j];
}

// ---- cut rounds inside the frontier engine (reference base_node.py:137-230, :292-341) ------------
// Per node of the batch a working copy of its cut list (ids into the engine's cut store, at most 64)
// and a few counters; the kernels below are the host-free parts of _base_bound's loop: who is still
// generating, the slack-cut removal, joining new cuts to the pool, applying K3's selection.
// State arrays are int32[batch] each, laid out field-major in one allocation.
enum CutField { CF_ROUNDS = 0, CF_IT_CREATED, CF_N_CREATED, CF_IT_ADDED, CF_N_ADDED, CF_IT_REMOVED,
                CF_N_REMOVED, CF_NCUT_OUT, CF_STALLED, CF_POOL_N, CF_SLAB_N, CF_DROPPED, CF_FIELDS };
constexpr int kMaxNodeCuts = 64;   // cut rows a node can carry (one wave handles a node's list)

struct CutGatherArgs {
    int batch, kc;
    const int32_t *slot;                     // pool row of each node of the batch
    const int32_t *pool_ncut, *pool_ids;     // node pool
    int32_t *ncut, *ids;                     // working lists (dense)
    int32_t *state;                          // CF_FIELDS x batch, zeroed here
    int32_t *counters;                       // [n_active, n_changed, max_ncut, -]: max_ncut initialised here
};
__global__ __launch_bounds__(64) void cut_gather_state(CutGatherArgs g) {
    const int k = blockIdx.x, lane = threadIdx.x;
    if (k >= g.batch) return;
    const size_t s = (size_t)g.slot[k];
    const int nc = g.pool_ncut[s];
    if (lane < nc) g.ids[(size_t)k * g.kc + lane] = g.pool_ids[s * g.kc + lane];
    if (lane == 0) {
        g.ncut[k] = nc;
        atomicMax(&g.counters[2], nc);
    }
    if (lane < CF_FIELDS) g.state[(size_t)lane * g.batch + k] = 0;
}

struct CutRoundArgs {
    int n, m0, mstride, kc, batch, round, max_rounds;
    double progress_tol, max_dual_bound;
    const int32_t *status;         // dense node outputs of K1 / K4
    const double *obj;
    const int32_t *mipf;
    const double *y;               // batch x mstride row duals
    int8_t *vstat;                 // batch x (n + mstride): the nodes' current bases
    int32_t *ncut, *ids;           // working cut lists
    int32_t *state;                // CF_FIELDS x batch
    double *obj_before;            // batch
    int32_t *active;               // batch: in/out (still generating this round)
    int32_t *resolve;              // batch: out, 1 where a cut was removed (the LP changed)
    int32_t *counters;             // [n_active, n_changed, max_ncut, n_need_tab]
    // The LP launch before this round may have dumped the tableau it ended with (have_dump): K2 reads
    // that, except where rows were just removed -- need_tab marks the nodes whose tableau has to be
    // worked out by a launch of its own (all generating nodes without have_dump).
    int have_dump = 0;
    int32_t *need_tab = nullptr;   // batch: out
};
// One wave per node.  round > 0: the stall test of the round just finished (base_node.py:320-324).
// Then base_node.py:196-203's loop condition, and for the nodes that go on: clip (left to K2 / K3),
// count the round, remember the objective, drop the cuts whose dual is exactly 0 (:326-341).
__global__ __launch_bounds__(64) void cut_round_begin(CutRoundArgs g) {
    const int k = blockIdx.x, lane = threadIdx.x;
    if (k >= g.batch) return;
    const size_t B = (size_t)g.batch;
    const int st = g.status[k];
    const bool lp_feasible = st == 0 || st == 2;
    const double obj = lp_feasible ? g.obj[k] : __builtin_huge_val();
    int stalled = g.state[CF_STALLED * B + k];
    if (g.round > 0 && g.active[k]) {
        const double before = g.obj_before[k];
        if (fabs(before - obj) / fabs(before) < g.progress_tol) stalled = 1;   // (0/0: not stalled, as numpy)
    }
    const int rounds = g.state[CF_ROUNDS * B + k];
    const bool act = lp_feasible && !g.mipf[k] && !stalled && rounds < g.max_rounds && obj < g.max_dual_bound;
    int nc = g.ncut[k];
    int nrem = 0;
    if (act) {
        int8_t *vs = g.vstat + (size_t)k * (g.n + g.mstride) + g.n + g.m0;
        int32_t *ids = g.ids + (size_t)k * g.kc;
        const bool mine = lane < nc;
        const bool gone = mine && g.y[(size_t)k * g.mstride + g.m0 + lane] == 0.0;
        const int id = mine ? ids[lane] : 0;
        const int8_t code = mine ? vs[lane] : (int8_t)0;
        const unsigned long long keep = __ballot(mine && !gone);
        nrem = nc - __popcll(keep);
        __builtin_amdgcn_wave_barrier();
        if (mine && !gone) {
            const int pos = __popcll(keep & ((1ull << lane) - 1ull));
            ids[pos] = id;
            vs[pos] = code;
        }
        nc -= nrem;
    }
    if (lane == 0) {
        g.state[CF_STALLED * B + k] = stalled;
        g.active[k] = act ? 1 : 0;
        g.resolve[k] = (act && nrem > 0) ? 1 : 0;
        const int tab = (act && (nrem > 0 || !g.have_dump)) ? 1 : 0;
        if (g.need_tab) g.need_tab[k] = tab;
        if (tab) atomicAdd(&g.counters[3], 1);
        if (act) {
            g.state[CF_ROUNDS * B + k] = rounds + 1;
            g.obj_before[k] = obj;
            g.ncut[k] = nc;
            if (nrem > 0) {
                g.state[CF_IT_REMOVED * B + k] += 1;
                g.state[CF_N_REMOVED * B + k] += nrem;
            }
            atomicAdd(&g.counters[0], 1);
        }
    }
}

struct PoolAppendArgs {
    int batch, slab_rows;
    const int32_t *active, *k2_ncuts;
    int32_t *state, *pool_list;
};
// the cuts K2 just wrote to the node's slab join its pool, in row order (cut_pool = {**old, **new},
// base_node.py:306) and are counted as created (:383)
__global__ __launch_bounds__(64) void pool_append(PoolAppendArgs g) {
    const int k = blockIdx.x * 64 + threadIdx.x;
    if (k >= g.batch || !g.active[k]) return;
    const size_t B = (size_t)g.batch;
    const int made = g.k2_ncuts[k];
    const int base = g.state[CF_SLAB_N * B + k];
    const int kept = min(made, g.slab_rows - base);
    const int pn = g.state[CF_POOL_N * B + k];
    int32_t *pl = g.pool_list + (size_t)k * g.slab_rows;
    for (int c = 0; c < kept; c++) pl[pn + c] = base + c;
    g.state[CF_POOL_N * B + k] = pn + kept;
    g.state[CF_SLAB_N * B + k] = base + kept;
    if (made > kept) g.state[CF_DROPPED * B + k] += made - kept;
    if (made > 0) {
        g.state[CF_IT_CREATED * B + k] += 1;
        g.state[CF_N_CREATED * B + k] += made;
    }
}

struct CutApplyArgs {
    int n, m0, mstride, kc, batch, slab_rows;
    const int32_t *active;
    const int32_t *k3_nadded, *k3_added;     // batch, batch x slab_rows (pool positions, in order)
    const double *slab_pi, *slab_pi0;
    int32_t *pool_list;
    int8_t *vstat;
    int32_t *ncut, *ids, *state;
    double *store_pi, *store_pi0;            // the engine's cut store (append-only)
    int32_t *store_count;
    int store_cap;
    int32_t *resolve;                        // in/out: 1 where the node's LP changed this round
    int32_t *counters;                       // [n_active, n_changed, max_ncut, -]
};
// K3's selection, applied (base_node.py:456-463): every selected cut becomes a row of the node's LP --
// a new entry of the cut store, appended to the node's list, its slack basic -- and leaves the pool.
__global__ __launch_bounds__(256) void cut_round_apply(CutApplyArgs g) {
    __shared__ int id_s;
    const int k = blockIdx.x, tid = threadIdx.x;
    if (k >= g.batch || !g.active[k]) return;
    const size_t B = (size_t)g.batch;
    const int nadd = g.k3_nadded[k];
    int32_t *pl = g.pool_list + (size_t)k * g.slab_rows;
    const int32_t *sel = g.k3_added + (size_t)k * g.slab_rows;
    int nc = g.ncut[k];
    int taken = 0;
    for (int a = 0; a < nadd; a++) {
        const int row = pl[sel[a]];
        __syncthreads();
        if (tid == 0) id_s = nc < g.kc ? atomicAdd(g.store_count, 1) : -1;
        __syncthreads();
        const int id = id_s;
        if (id < 0 || id >= g.store_cap) continue;   // the node's list or the store is full: the cut is dropped
        const double *src = g.slab_pi + ((size_t)k * g.slab_rows + row) * g.n;
        double *dst = g.store_pi + (size_t)id * g.n;
        for (int j = tid; j < g.n; j += 256) dst[j] = src[j];
        if (tid == 0) {
            g.store_pi0[id] = g.slab_pi0[(size_t)k * g.slab_rows + row];
            g.ids[(size_t)k * g.kc + nc] = id;
            g.vstat[(size_t)k * (g.n + g.mstride) + g.n + g.m0 + nc] = 1;   // a new row enters with its slack basic
        }
        nc++;
        taken++;
    }
    __syncthreads();
    if (tid == 0) {
        // the pool keeps its order without the selected entries (del self.cut_pool[idx])
        const int pn = g.state[CF_POOL_N * B + k];
        for (int a = 0; a < nadd; a++) pl[sel[a]] = -1;
        int w = 0;
        for (int q = 0; q < pn; q++)
            if (pl[q] >= 0) pl[w++] = pl[q];
        g.state[CF_POOL_N * B + k] = w;
        g.ncut[k] = nc;
        g.state[CF_NCUT_OUT * B + k] = nc;
        if (nadd > 0) {
            g.state[CF_IT_ADDED * B + k] += 1;
            g.state[CF_N_ADDED * B + k] += nadd;
        }
        if (nadd > taken) g.state[CF_DROPPED * B + k] += nadd - taken;
        const int changed = g.resolve[k] | (taken > 0 ? 1 : 0);
        g.resolve[k] = changed;
        if (changed) atomicAdd(&g.counters[1], 1);
        atomicMax(&g.counters[2], nc);
    }
}

}  // namespace mipx
