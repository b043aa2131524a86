// finish_kernels.hip.h -- the host half of a frontier step, on the device.
//
// After K1 (node LPs + plunge) and K4 (scoring) of a step the reference's _evaluate_node bookkeeping
// (algorithms/branch_and_bound.py:243-266), the child records (nodes/base_node.py:592-608) and the
// pseudo-cost updates (nodes/branch/pseudo_cost.py:38-43, :68-100) used to be a host loop over every
// evaluated node (54 ns each: 4-5 ms per 73 000-node step beside a 6.4 ms kernel).  Here they are
// four small launches on the step's own stream; the host reads back one summary, a COMPACT list of the
// new open nodes (key + pool row: what its priority queue needs) and the list of pool rows that are
// free again:
//   finish_candidates / finish_prefix_min  the incumbent value the host loop would hold when it reaches
//                  each chain (it lowers the value as it walks the batch): an exclusive prefix minimum;
//   finish_decide  one thread per chain (a node of the batch and the children solved in place below
//                  it): which nodes branch, which children are queued, the incumbent candidate, the
//                  closed leaves' bound, counters -- the decisions of tree_finish's `evaluate`;
//   finish_scan    one workgroup: exclusive scans of the per-chain counts (offsets into the compact
//                  lists, in chain order: deterministic), reductions of the rest, the incumbent;
//   finish_write   one workgroup per chain: the parent's bounds are read ONCE into registers and
//                  carried down the chain; every QUEUED child record (bounds + the basis its parent's
//                  LP ended with) is written to its pool row -- the child solved in place needs no
//                  record at all (K5 wrote and re-read one); open entries, free rows, samples;
//   pc_apply       one wave per (variable, direction): walks the step's sample list in order and
//                  applies the running-mean recurrence exactly as the host did, sample by sample --
//                  samples of different table entries are independent, so the table is bit-identical
//                  to the sequential update in reference order.
// A step that filed strong-branching requests (K4's ask counter > 0) is left to the host path: the
// kernels see the counter and do nothing.
// Pool rows for the children are handed out by the host BEFORE the launch (chain k, level p,
// direction d owns budget[k][2 p + d]): no device allocator, the same rows in every run.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "lp_kernel.hip.h"

namespace mipx {

struct OpenEntry {       // one new open node, as the host's queue and node table need it
    double key;          // its inherited bound (the parent's LP objective)
    double bval;
    int32_t slot, depth, bidx_dir, anchor;   // bidx_dir = 2 * branching variable + direction
};
static_assert(sizeof(OpenEntry) == 32, "OpenEntry layout");

struct PcSample {        // one pseudo-cost update (pseudo_cost.py:68-100), in the order they are applied
    int32_t var_dir;     // 2 * variable + direction (0 left, 1 right)
    int32_t status;      // LP status of the child the sample comes from
    double obj, bound, vc;   // its objective, its parent's bound, the variable change
};
static_assert(sizeof(PcSample) == 32, "PcSample layout");

struct FinishSummary {
    int32_t host_path;           // 1: probe requests were filed -- the host path finishes the step
    int32_t n_open, n_dead, n_samples;
    int32_t unbounded, incumbent_pos;   // output position of the step's best improving integral node, -1: none
    int32_t n_deferred, pad0;           // nodes that filed probe requests: left to the host, after everything else
    int64_t evaluated, dives, pivots;
    double closed_min, primal;   // lowest value among the step's closed leaves; the incumbent value after the step
    double primal_before;        // ... and before it: what the step's decisions compared against
    int64_t pad[6];
};
static_assert(sizeof(FinishSummary) == 128, "FinishSummary layout");

struct FinishArgs {
    int n, m, B, dive, rule;
    // outputs of K1 / K4, position level * B + k
    const int32_t *status, *bidx, *mipf, *nprobe, *npiv, *dvar, *ddir;
    const double *obj, *bval, *dval;
    const int32_t *ask_count;
    int ask_cap;                 // more requests than the compact list holds (the ramp-up): the whole step is the host's
    const int8_t *vout;          // (dive + 1) * B x (n + m): the basis every LP ended with
    // the batch: pool rows, the parents' records (dense by batch position), the rows its children may take
    const int32_t *slot;         // B
    const int32_t *par_i;        // 4 x B: b_idx | b_dir | depth | anchor
    const double *par_d;         // 2 x B: dual_bound | b_val
    const int32_t *budget;       // B x 2 (1 + dive)
    double *pool_l, *pool_u;
    int8_t *pool_v;
    double *primal;              // the incumbent value on the device (in / out)
    // per-chain scratch
    int32_t *c_info;             // B: branchings | took the last dive << 8
    int32_t *c_cnt;              // 3 x B: open | dead | samples -> their exclusive prefix sums
    int32_t *c_eval;             // 2 x B: evaluated levels | pivots
    double *c_val;               // 2 x B: closed-leaf minimum | incumbent candidate (inf: none)
    int32_t *c_flag;             // B: unbounded | candidate level << 8
    double *c_run;               // B: the chain's integral leaf value (finish_candidates) -> the incumbent value the
                                 // host loop would hold when it reaches chain k (finish_prefix_min)
    // compact outputs
    FinishSummary *sum;
    OpenEntry *open;
    int32_t *dead;
    PcSample *samples;
    int32_t *sample_keys;        // var_dir of every sample again, densely: what pc_apply's 2 n waves all scan
};

__device__ __forceinline__ bool lp_feasible_code(int st) { return st == 0 || st == 2; }

// the node at pos was followed in place by a child at pos + B that counts (tree_finish's `dived`)
__device__ __forceinline__ bool chain_dived(const FinishArgs &g, int pos) {
    return pos < g.dive * g.B && g.dvar[pos] >= 0 && g.status[pos + g.B] >= 0 && g.nprobe[pos + g.B] == 0;
}

__device__ __forceinline__ double variable_change(double v, int dir) { return dir == 0 ? v - floor(v) : ceil(v) - v; }

// The samples chain k contributes, in the host's order: the branch that made the node, then one per
// dive child that counts (tree_finish section 3).  out == nullptr: count only.
__device__ inline int chain_samples(const FinishArgs &g, int k, double primal, PcSample *out) {
    int cnt = 0;
    if (g.rule != 1 || !lp_feasible_code(g.status[k])) return 0;
    const int pb = g.par_i[k];
    if (pb >= 0) {
        if (out) {
            const int dir = g.par_i[g.B + k];
            PcSample s;
            s.var_dir = 2 * pb + dir; s.status = g.status[k]; s.obj = g.obj[k]; s.bound = g.par_d[k];
            s.vc = variable_change(g.par_d[g.B + k], dir);
            out[cnt] = s;
        }
        cnt++;
    }
    for (int pos = k; chain_dived(g, pos) && g.obj[pos] < primal && !g.mipf[pos] && lp_feasible_code(g.status[pos]); pos += g.B) {
        const int cp = pos + g.B;
        if (!lp_feasible_code(g.status[cp])) break;
        if (out) {
            PcSample s;
            s.var_dir = 2 * g.dvar[pos] + g.ddir[pos]; s.status = g.status[cp]; s.obj = g.obj[cp]; s.bound = g.obj[pos];
            s.vc = variable_change(g.dval[pos], g.ddir[pos]);
            out[cnt] = s;
        }
        cnt++;
    }
    return cnt;
}

// The host loop lowers the incumbent value as it walks the batch: chain k is judged against the best
// integral node of the chains before it.  A chain holds at most one integral node -- where its plunge
// ended -- and whether an earlier chain was cut short by a still earlier incumbent does not matter (LP
// values do not decrease down a chain: a node behind a pruned one is no better than what pruned it).  So
// the value chain k sees is min(value at the start of the step, integral leaves of chains < k): one
// exclusive prefix minimum.
__global__ __launch_bounds__(256) void finish_candidates(FinishArgs g) {
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (*g.ask_count > g.ask_cap) {
        if (k == 0) g.sum->host_path = 1;
        return;
    }
    if (k == 0) { g.sum->host_path = 0; g.sum->primal_before = *g.primal; }
    if (k >= g.B) return;
    double cand = __builtin_huge_val();
    // (a node that filed probe requests is the host's: it comes after every other chain of the step)
    for (int pos = k; g.nprobe[k] == 0; pos += g.B) {
        if (!lp_feasible_code(g.status[pos])) break;
        if (g.mipf[pos]) { cand = g.obj[pos]; break; }
        const bool take_dive = chain_dived(g, pos);
        if (!take_dive || g.dvar[pos] < 0) break;
    }
    g.c_run[k] = cand;
}

__global__ __launch_bounds__(1024) void finish_prefix_min(FinishArgs g) {
    __shared__ double s_min[1024];
    if (g.sum->host_path) return;
    const int t = threadIdx.x, B = g.B;
    const int C = (B + 1023) / 1024;
    const int k0 = min(t * C, B), k1 = min(k0 + C, B);
    const double INF = __builtin_huge_val();
    double loc = INF;
    for (int k = k0; k < k1; k++) loc = fmin(loc, g.c_run[k]);
    s_min[t] = loc;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {       // inclusive prefix minimum of the partial minima
        const double o = t >= d ? s_min[t - d] : INF;
        __syncthreads();
        s_min[t] = fmin(s_min[t], o);
        __syncthreads();
    }
    double run = fmin(g.sum->primal_before, t > 0 ? s_min[t - 1] : INF);
    for (int k = k0; k < k1; k++) {
        const double c = g.c_run[k];
        g.c_run[k] = run;
        run = fmin(run, c);
    }
}

__global__ __launch_bounds__(256) void finish_decide(FinishArgs g) {
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (g.sum->host_path || k >= g.B) return;
    const double primal = g.c_run[k];
    const double INF = __builtin_huge_val();
    const int per = 2 * (1 + g.dive);
    int nbranch = 0, nopen = 0, took_last = 0, evaluated = 0, pivots = 0, unbounded = 0, cand_level = 0;
    double closed = INF, cand = INF;
    const bool deferred = g.nprobe[k] > 0;
    for (int level = 0, pos = k; !deferred; level++, pos += g.B) {
        const int st = g.status[pos];
        const bool feas = lp_feasible_code(st);
        evaluated++;
        pivots += g.npiv[pos];
        unbounded |= st == 2;
        bool branched = false, go_on = false;
        if (feas && g.obj[pos] < primal) {
            const bool take_dive = chain_dived(g, pos);
            const int bvar = take_dive ? g.dvar[pos] : g.bidx[pos];
            if (g.mipf[pos]) {
                cand = g.obj[pos];      // (ends the chain: one candidate per chain at most)
                cand_level = level;
            } else if (bvar >= 0) {
                branched = true;
                nbranch++;
                nopen += take_dive ? 1 : 2;
                go_on = take_dive;
                took_last = take_dive ? 1 : 0;
            }
        }
        if (!branched) closed = fmin(closed, feas ? g.obj[pos] : INF);
        if (!go_on) break;
    }
    g.c_info[k] = nbranch | (took_last << 8) | (deferred ? 1 << 9 : 0);
    g.c_cnt[k] = nopen;
    g.c_cnt[g.B + k] = per - nopen;
    // (section 3 runs before the walk: the step's starting value)
    g.c_cnt[2 * g.B + k] = deferred ? 0 : chain_samples(g, k, g.sum->primal_before, nullptr);
    g.c_eval[k] = evaluated;
    g.c_eval[g.B + k] = pivots;
    g.c_val[k] = closed;
    g.c_val[g.B + k] = cand;
    g.c_flag[k] = unbounded | (cand_level << 8);
}

// One workgroup of 1024 threads: thread t owns the chains [t C, (t + 1) C).
__global__ __launch_bounds__(1024) void finish_scan(FinishArgs g) {
    __shared__ int s_cnt[3][1024];
    __shared__ long long s_ev[2][1024];
    __shared__ double s_val[2][1024];
    __shared__ int s_ord[1024], s_unb[1024];
    if (g.sum->host_path) return;
    const int t = threadIdx.x, B = g.B;
    const int C = (B + 1023) / 1024;
    const int k0 = min(t * C, B), k1 = min(k0 + C, B);
    const double INF = __builtin_huge_val();
    int loc[3] = {0, 0, 0};
    long long ev = 0, pv = 0;
    double closed = INF, cand = INF;
    int ord = 0x7fffffff, unb = 0;
    for (int k = k0; k < k1; k++) {
        unb += (g.c_info[k] >> 9 & 1) << 1;    // (bit 0: unbounded flag; above it: the count of deferred nodes)
        for (int f = 0; f < 3; f++) loc[f] += g.c_cnt[f * B + k];
        ev += g.c_eval[k];
        pv += g.c_eval[B + k];
        closed = fmin(closed, g.c_val[k]);
        const double c = g.c_val[B + k];
        if (c < cand) { cand = c; ord = k; }     // (ascending k: the first chain that holds the minimum stays)
        unb |= g.c_flag[k] & 1;
    }
    for (int f = 0; f < 3; f++) s_cnt[f][t] = loc[f];
    s_ev[0][t] = ev; s_ev[1][t] = pv;
    s_val[0][t] = closed; s_val[1][t] = cand;
    s_ord[t] = ord; s_unb[t] = unb;
    __syncthreads();
    // inclusive scan of the 1024 partial counts (Hillis-Steele), tree reductions of the rest
    for (int d = 1; d < 1024; d <<= 1) {
        int add[3] = {0, 0, 0};
        if (t >= d)
            for (int f = 0; f < 3; f++) add[f] = s_cnt[f][t - d];
        __syncthreads();
        for (int f = 0; f < 3; f++) s_cnt[f][t] += add[f];
        __syncthreads();
    }
    for (int d = 512; d >= 1; d >>= 1) {
        if (t < d) {
            s_ev[0][t] += s_ev[0][t + d];
            s_ev[1][t] += s_ev[1][t + d];
            s_val[0][t] = fmin(s_val[0][t], s_val[0][t + d]);
            const double a = s_val[1][t], b = s_val[1][t + d];
            if (b < a || (b == a && s_ord[t + d] < s_ord[t])) { s_val[1][t] = b; s_ord[t] = s_ord[t + d]; }
            s_unb[t] = ((s_unb[t] | s_unb[t + d]) & 1) | ((s_unb[t] >> 1) + (s_unb[t + d] >> 1)) << 1;
        }
        __syncthreads();
    }
    int run[3];
    for (int f = 0; f < 3; f++) run[f] = s_cnt[f][t] - loc[f];     // exclusive prefix of this thread's chains
    for (int k = k0; k < k1; k++)
        for (int f = 0; f < 3; f++) {
            const int c = g.c_cnt[f * B + k];
            g.c_cnt[f * B + k] = run[f];
            run[f] += c;
        }
    if (t == 0) {
        FinishSummary *s = g.sum;
        s->n_open = s_cnt[0][1023]; s->n_dead = s_cnt[1][1023]; s->n_samples = s_cnt[2][1023];
        s->evaluated = s_ev[0][0]; s->pivots = s_ev[1][0];
        s->n_deferred = s_unb[0] >> 1;
        s->dives = s_ev[0][0] - (B - s->n_deferred);
        s->closed_min = s_val[0][0];
        s->unbounded = s_unb[0] & 1;
        const double best = s_val[1][0], old = *g.primal;
        if (best < old) {
            const int kk = s_ord[0];
            s->incumbent_pos = (g.c_flag[kk] >> 8) * B + kk;
            *g.primal = best;
            s->primal = best;
        } else {
            s->incumbent_pos = -1;
            s->primal = old;
        }
    }
}

// One workgroup per chain.  NE = elements of l / u a thread carries (n <= 256 NE).
template <int NE>
__global__ __launch_bounds__(256) void finish_write(FinishArgs g) {
    if (g.sum->host_path) return;
    const int k = blockIdx.x, tid = threadIdx.x;
    if (k >= g.B) return;
    const int n = g.n, nv = g.n + g.m, B = g.B, per = 2 * (1 + g.dive);
    const int info = g.c_info[k];
    const int nbranch = info & 0xff, took_last = (info >> 8) & 1;
    if (nbranch > 0) {
        double lreg[NE], ureg[NE];
        const size_t ps = (size_t)g.slot[k];
#pragma unroll
        for (int e = 0; e < NE; e++) {
            const int j = tid + 256 * e;
            lreg[e] = j < n ? g.pool_l[ps * n + j] : 0.0;
            ureg[e] = j < n ? g.pool_u[ps * n + j] : 0.0;
        }
        const int depth0 = g.par_i[2 * B + k], anchor = g.par_i[3 * B + k];
        int off = g.c_cnt[k];
        for (int p = 0; p < nbranch; p++) {
            const int pos = p * B + k;
            const bool take_dive = p < nbranch - 1 || took_last;
            const int bvar = take_dive ? g.dvar[pos] : g.bidx[pos];
            const double xv = take_dive ? g.dval[pos] : g.bval[pos];
            const int dd = take_dive ? g.ddir[pos] : -1;
            const double lo_r = ceil(xv), up_l = floor(xv);
            for (int dir = 0; dir < 2; dir++) {
                if (dir == dd) continue;      // solved in place: it never needs a record
                const size_t cs = (size_t)g.budget[(size_t)k * per + 2 * p + dir];
#pragma unroll
                for (int e = 0; e < NE; e++) {
                    const int j = tid + 256 * e;
                    if (j < n) {
                        g.pool_l[cs * n + j] = (j == bvar && dir == 1) ? lo_r : lreg[e];
                        g.pool_u[cs * n + j] = (j == bvar && dir == 0) ? up_l : ureg[e];
                    }
                }
                const int8_t *src = g.vout + (size_t)pos * nv;
                int8_t *dst = g.pool_v + cs * nv;
                if ((nv & 3) == 0) {
                    for (int q = tid; q < nv / 4; q += 256) ((int32_t *)dst)[q] = ((const int32_t *)src)[q];
                } else {
                    for (int q = tid; q < nv; q += 256) dst[q] = src[q];
                }
                if (tid == 0) {
                    OpenEntry e;
                    e.key = g.obj[pos]; e.bval = xv; e.slot = (int32_t)cs; e.depth = depth0 + p + 1;
                    e.bidx_dir = 2 * bvar + dir; e.anchor = anchor;
                    g.open[off] = e;
                }
                off++;
            }
            if (take_dive) {   // the chain goes on below the child solved in place
#pragma unroll
                for (int e = 0; e < NE; e++) {
                    const int j = tid + 256 * e;
                    if (j == bvar) {
                        if (dd == 1) lreg[e] = lo_r;
                        else ureg[e] = up_l;
                    }
                }
            }
        }
    }
    if (tid == 0) {
        // rows of the chain's budget that hold no open node: free again
        int off = g.c_cnt[B + k];
        for (int p = 0; p <= g.dive; p++) {
            const bool branched = p < nbranch;
            const bool take_dive = branched && (p < nbranch - 1 || took_last);
            const int dd = take_dive ? g.ddir[p * B + k] : -1;
            for (int dir = 0; dir < 2; dir++)
                if (!branched || dir == dd) g.dead[off++] = g.budget[(size_t)k * per + 2 * p + dir];
        }
    }
}

// the samples, written by the thread that owns the chain at the chain's offset (the same walk as in
// finish_decide, against the incumbent value the step's decisions compared with)
__global__ __launch_bounds__(256) void finish_samples(FinishArgs g) {
    if (g.sum->host_path || g.rule != 1) return;
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= g.B) return;
    if (g.nprobe[k] > 0) return;
    const int off = g.c_cnt[2 * g.B + k];
    const int cnt = chain_samples(g, k, g.sum->primal_before, g.samples + off);
    for (int q = 0; q < cnt; q++) g.sample_keys[off + q] = g.samples[off + q].var_dir;
}

struct PcApplyArgs {
    int n;
    const FinishSummary *sum;    // count = sum->n_samples unless count >= 0
    int count;
    const PcSample *samples;
    const int32_t *keys;         // samples[i].var_dir, densely (4 bytes per sample instead of a 32-byte stride)
    double *cost_l, *cost_r;
    uint8_t *has;
    int32_t *times;              // [left | right]
    double *own;                 // [sum_l | sum_r | times_l | times_r]
};
// One wave per table entry (variable, direction); pseudo_cost.py:68-100 sample by sample, in list order.
__global__ __launch_bounds__(64) void pc_apply(PcApplyArgs a) {
    if (a.count < 0 && a.sum->host_path) return;
    const int cnt = a.count >= 0 ? a.count : a.sum->n_samples;
    if (cnt == 0) return;
    const int vd = blockIdx.x, var = vd >> 1, dir = vd & 1, lane = threadIdx.x, n = a.n;
    double cost = dir ? a.cost_r[var] : a.cost_l[var];
    int times = a.times[dir * n + var];
    double own_sum = a.own[dir * n + var], own_t = a.own[(2 + dir) * n + var];
    bool any = false;
    constexpr int UN = 16;   // loads in flight per lane: the walk is bound by their latency
    for (int base0 = 0; base0 < cnt; base0 += 64 * UN) {
        int key[UN];
#pragma unroll
        for (int c = 0; c < UN; c++) {
            const int i = base0 + 64 * c + lane;
            key[c] = i < cnt ? a.keys[i] : -1;
        }
#pragma unroll
        for (int c = 0; c < UN; c++) {
            const int base = base0 + 64 * c;
            unsigned long long mask = __ballot(key[c] == vd);
            if (mask == 0ull) continue;
            // every lane fetches its own sample of the chunk (one coalesced load); the matches are then read
            // out of the lanes in order -- a load per match would put its latency into the serial recurrence
            // (a frequently branched variable has thousands of samples in a step)
            const int i = base + lane;
            PcSample mine;
            mine.var_dir = 0; mine.status = 1; mine.obj = 0.0; mine.bound = 0.0; mine.vc = 1.0;
            if (i < cnt) mine = a.samples[i];
            while (mask) {
                const int j = __ffsll((long long)mask) - 1;
                mask &= mask - 1ull;
                const int st = __builtin_amdgcn_readlane(mine.status, j);
                if (st == 0 || st == 3) {
                    const double obj = readlane_f64(mine.obj, j), bound = readlane_f64(mine.bound, j), vc = readlane_f64(mine.vc, j);
                    double bc = obj - bound;
                    if (bc < 0) bc = 0;
                    cost = (cost * (double)times + bc / vc) / (double)(times + 1);
                    own_sum += bc / vc;
                } else {
                    own_sum += cost;
                }
                times += 1;
                own_t += 1.0;
                any = true;
            }
        }
    }
    if (any && lane == 0) {
        if (dir) a.cost_r[var] = cost;
        else a.cost_l[var] = cost;
        a.times[dir * n + var] = times;
        a.own[dir * n + var] = own_sum;
        a.own[(2 + dir) * n + var] = own_t;
        a.has[var] = 1;
    }
}

// An exchange between ranks: what the OTHER ranks sampled since the last exchange joins the table in sum
// form -- mean <- (mean * times + sum) / (times + count) -- without touching this rank's own running
// recurrence (its samples of the steps in flight are in the table already).
struct PcMergeArgs {
    int n;
    const double *delta;         // [sum_l | sum_r | times_l | times_r], n each
    double *cost_l, *cost_r;
    uint8_t *has;
    int32_t *times;
};
__global__ __launch_bounds__(256) void pc_merge(PcMergeArgs a) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= 2 * a.n) return;
    const int dir = e / a.n, var = e - dir * a.n;
    const double ds = a.delta[dir * a.n + var], dt = a.delta[(2 + dir) * a.n + var];
    if (!(dt > 0.0)) return;
    double *cost = dir ? a.cost_r : a.cost_l;
    const int t0 = a.times[dir * a.n + var];
    cost[var] = (cost[var] * (double)t0 + ds) / ((double)t0 + dt);
    a.times[dir * a.n + var] = t0 + (int)dt;
    a.has[var] = 1;
}

// the host lowered the incumbent (an exchange, mipx_tree_set_primal_bound, a step finished by the host)
__global__ void primal_lower(double *primal, double value) {
    if (value < *primal) *primal = value;
}

}  // namespace mipx
