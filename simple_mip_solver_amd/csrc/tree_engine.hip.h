// tree_engine.hip.h -- native frontier engine: the BranchAndBound.solve() loop of the reference
// (simple_mip_solver/algorithms/branch_and_bound.py:215-266) with the stock node classes
// (BaseNode / PseudoCostBranchNode, best-first or depth-first), processing a BATCH of open nodes
// per step on the GPU.  Node records (bounds + warm-start basis) live in a device-resident pool;
// the host keeps only the priority queue and 48 bytes of bookkeeping per node.
//
// frontier_batch = 1 reproduces the reference's node order exactly (the queue is a re-statement of
// CPython's heapq, which queue.PriorityQueue uses), including pseudo-cost table evolution; larger
// batches evaluate the B best open nodes per step against the table as of the start of the step.
// Included by mipx.hip (needs mipx_ctx, mipx_problem, pick_cfg, HIP_TRY, fail).
#pragma once
#include <algorithm>
#include <chrono>
#include <cmath>
#include <limits>
#include <memory>
#include <queue>

#include "tree_kernels.hip.h"
#include "finish_kernels.hip.h"

struct NodeRec {
    double key;          // queue key: dual bound (best first) or -depth (depth first)
    double dual_bound;   // bound inherited from the parent (parent's LP objective)
    double b_val;
    int32_t slot;        // row in the device pool, -1 once released
    int32_t depth;
    int32_t b_idx;       // variable branched on to create this node (-1 root)
    int32_t b_dir;       // 0 left (x <= floor), 1 right (x >= ceil)
    int32_t anchor;      // entry of the anchor table its warm start refactors from (-1: the root's)
    int32_t born;        // step that created the node (MIPX_TREE_PROFILE: age histogram)
    int32_t ncut;        // cut rows the node carries (cut rounds: inherited from its parent)
    // (no default initialisers: a fresh block of the node table must not be touched page by page)
};

// The node table: every node ever created, indexed by id.  Blocks of 2^18 records instead of one
// vector: growing never copies (a 5 M-node table is 240 MB -- one reallocation stalled the step
// loop for 59 ms).
struct NodeTable {
    static constexpr int kShift = 18;
    static constexpr size_t kMask = ((size_t)1 << kShift) - 1;
    std::vector<std::unique_ptr<NodeRec[]>> blocks;
    size_t n = 0;
    NodeRec &operator[](size_t i) { return blocks[i >> kShift][i & kMask]; }
    const NodeRec &operator[](size_t i) const { return blocks[i >> kShift][i & kMask]; }
    void push_back(const NodeRec &r) {
        if ((n >> kShift) == blocks.size()) blocks.emplace_back(new NodeRec[(size_t)1 << kShift]);
        (*this)[n++] = r;
    }
    size_t size() const { return n; }
};

// CPython's heapq on node ids (Lib/heapq.py heappush/heappop/_siftdown/_siftup), so that ties are
// broken exactly as queue.PriorityQueue breaks them in the reference.
struct PyHeap {
    struct Item { double key; int64_t id; };
    std::vector<Item> h;   // keys inline: sifting never touches the node table
    static bool lt(const Item &a, const Item &b) { return a.key < b.key; }
    void siftdown(size_t startpos, size_t pos) {
        const Item item = h[pos];
        while (pos > startpos) {
            const size_t parent = (pos - 1) >> 1;
            if (lt(item, h[parent])) { h[pos] = h[parent]; pos = parent; continue; }
            break;
        }
        h[pos] = item;
    }
    void siftup(size_t pos) {
        const size_t end = h.size(), start = pos;
        const Item item = h[pos];
        size_t child = 2 * pos + 1;
        while (child < end) {
            const size_t right = child + 1;
            if (right < end && !lt(h[child], h[right])) child = right;
            h[pos] = h[child];
            pos = child;
            child = 2 * pos + 1;
        }
        h[pos] = item;
        siftdown(start, pos);
    }
    void push(double key, int64_t id) { h.push_back({key, id}); siftdown(0, h.size() - 1); }
    int64_t pop() {
        const Item last = h.back();
        h.pop_back();
        if (h.empty()) return last.id;
        const Item top = h[0];
        h[0] = last;
        siftup(0);
        return top.id;
    }
    bool empty() const { return h.empty(); }
    size_t size() const { return h.size(); }
};

// Open list of the batched best-first search (frontier batches > 1): a step takes the B smallest
// keys at once, so per-node heap order is not needed.  Keys are bucketed linearly above the first
// (smallest: a child's bound is never below its parent's) key; a step empties whole buckets in
// key order and splits the last one exactly with nth_element.  O(1) pushes, O(B) per step, where
// the binary heap paid ~0.1 us per push and pop in cache misses.
struct BucketQueue {
    struct Item { double key; int64_t id; };
    static constexpr size_t kMaxBuckets = (size_t)1 << 22;
    std::vector<std::vector<Item>> tab;
    double k0 = 0.0, inv_width = 0.0;
    size_t cur = 0, count = 0;
    bool started = false;
    size_t index(double key) const {
        const double r = (key - k0) * inv_width;
        if (!(r > 0.0)) return 0;
        return r >= (double)(kMaxBuckets - 1) ? kMaxBuckets - 1 : (size_t)r;
    }
    void push(double key, int64_t id) {
        if (!started && std::isfinite(key)) {  // (the root's inherited bound is -inf: bucket 0)
            started = true;
            k0 = key;
            inv_width = 4096.0 / std::fmax(std::fabs(key), 1.0);  // bucket width: 2^-12 of |first key|
        }
        const size_t i = started ? index(key) : 0;
        if (i >= tab.size()) tab.resize(i + 1 + (i >> 3));
        tab[i].push_back({key, id});
        if (count == 0 || i < cur) cur = i;
        count++;
    }
    // A step's pushes in one go, in the given order (the queue ends up exactly as after the pushes one by
    // one): a push lands in one of 10^5 buckets -- the header of its vector and the cache line of its tail
    // are two misses -- so the headers are prefetched 16 items ahead and the tails 8 ahead.
    void push_many(const Item *it, size_t cnt) {
        if (cnt == 0) return;
        if (!started) {   // (the scale comes from the first finite key: let the one-by-one path set it)
            for (size_t k = 0; k < cnt; k++) push(it[k].key, it[k].id);
            return;
        }
        size_t hi = 0;
        for (size_t k = 0; k < cnt; k++) hi = std::max(hi, index(it[k].key));
        if (hi >= tab.size()) tab.resize(hi + 1 + (hi >> 3));
        constexpr size_t A = 16, Bd = 8;
        for (size_t k = 0; k < cnt + A; k++) {
            if (k < cnt) __builtin_prefetch(&tab[index(it[k].key)], 1);
            if (k >= A - Bd && k - (A - Bd) < cnt) {
                const std::vector<Item> &v = tab[index(it[k - (A - Bd)].key)];
                if (!v.empty()) __builtin_prefetch(v.data() + v.size(), 1);
            }
            if (k >= A) {
                const Item &e = it[k - A];
                const size_t i = index(e.key);
                tab[i].push_back(e);
                if (count == 0 || i < cur) cur = i;
                count++;
            }
        }
    }
    bool empty() const { return count == 0; }
    size_t size() const { return count; }
    void settle() { while (cur < tab.size() && tab[cur].empty()) cur++; }
    double min_key() {
        settle();
        double k = std::numeric_limits<double>::infinity();
        for (const Item &it : tab[cur]) k = std::fmin(k, it.key);
        return k;
    }
    // the `want` smallest items (ties: smaller id first) are appended to out
    void pop_batch(size_t want, std::vector<Item> &out) {
        while (want > 0 && count > 0) {
            settle();
            std::vector<Item> &v = tab[cur];
            if (v.size() <= want) {
                out.insert(out.end(), v.begin(), v.end());
                want -= v.size();
                count -= v.size();
                v.clear();
            } else if (v.size() > 8 * want && tab.size() * 16 < kMaxBuckets && refinements < max_refinements) {
                // a deep search piles its open nodes just above the dual bound: buckets sized for
                // the first key end up holding 10^5..10^6 of them and every step pays nth_element
                // and an erase over all of those (2.7 ms per step at 5 M open nodes) -> finer buckets
                refine();
            } else {
                auto less = [](const Item &a, const Item &b) { return a.key < b.key || (a.key == b.key && a.id < b.id); };
                std::nth_element(v.begin(), v.begin() + (std::ptrdiff_t)want, v.end(), less);
                out.insert(out.end(), v.begin(), v.begin() + (std::ptrdiff_t)want);
                v.erase(v.begin(), v.begin() + (std::ptrdiff_t)want);
                count -= want;
                want = 0;
            }
        }
    }
    int refinements = 0;
    // (two refinements = 256 x finer than the first scale: beyond that the pushes of a step scatter
    // over 10^5 bucket tails and the bookkeeping pays in cache misses what the pop saves)
    int max_refinements = std::getenv("MIPX_BQ_REFINE") ? std::atoi(std::getenv("MIPX_BQ_REFINE")) : 2;
    void refine() {  // 16 x finer buckets, same order of the items inside the new buckets' sources
        std::vector<std::vector<Item>> old;
        old.swap(tab);
        inv_width *= 16.0;
        count = 0;
        cur = 0;
        refinements++;
        for (const auto &v : old)
            for (const Item &it : v) push(it.key, it.id);
    }
    // every item, in bucket order (for peek / sharding)
    void items(std::vector<Item> &out) const {
        for (size_t i = cur; i < tab.size(); i++) out.insert(out.end(), tab[i].begin(), tab[i].end());
    }
    void clear() {
        for (auto &v : tab) v.clear();
        count = 0;
        cur = 0;
    }
    // (the bucket scale -- k0, width, refinements -- survives a clear: keep_shard refills the queue)
};

// Per-step device outputs and host staging, double-buffered so that the host bookkeeping of step
// k overlaps the node-LP kernel of step k+1 (frontier batches > 1).
struct StepBuf {
    int32_t *d_slot = nullptr, *d_status = nullptr, *d_iters = nullptr, *d_npiv = nullptr,
            *d_bidx = nullptr, *d_mipf = nullptr, *d_nprobe = nullptr, *d_plist = nullptr;
    int32_t *d_ask_count = nullptr;          // [count | pad] then kAskCap entries, inside d_pack
    mipx::ScoreArgs::Ask *d_ask = nullptr;
    size_t ask_off = 0;
    double *d_obj = nullptr, *d_x = nullptr, *d_bval = nullptr, *d_dval = nullptr;
    int32_t *d_dvar = nullptr, *d_ddir = nullptr;   // in-place dives of the step (per parent)
    int8_t *d_vout = nullptr;
    // what the host reads back every step, packed so that ONE copy into pinned memory fetches it:
    // [obj | bval] (2 * max_batch f64 each: the batch, then its dive children) [dive_val]
    // (max_batch f64), then [status | bidx | mipf | nprobe | npiv] (2 * max_batch i32 each)
    // [dive_var | dive_dir] (max_batch i32 each), then the probe requests
    char *d_pack = nullptr, *h_pack = nullptr;
    size_t pack_bytes = 0;
    int32_t *h_slot = nullptr;  // pinned staging of the batch's pool rows
    hipEvent_t e0 = nullptr, e1 = nullptr, done = nullptr;
    hipEvent_t k2a = nullptr, k2b = nullptr, k3b = nullptr;   // cut rounds: around K2 and K3 of a round
    std::vector<int64_t> ids;
    std::vector<NodeRec> recs;  // the batch's node records, copied at pop time (one random access per node and step)
    std::vector<int32_t> slots;  // staging kept alive
    // branching parents by dive level (0: the batch, p: the p-th dive children): output position,
    // record row, variable, and the two child rows each
    struct Branchings { std::vector<int32_t> pos, slot, var, child; };
    std::vector<Branchings> br;
    std::vector<int32_t> dive_slots;  // record rows of the dive children (freed with the step)
    // cut rounds (mipx_tree_create_ex with cut parameters): per node of the batch the working cut
    // list, the row duals, the loop's state and the pool of candidate cuts (a slab of slab_rows rows)
    int32_t *w_ncut = nullptr, *w_ids = nullptr, *cs_state = nullptr, *cs_active = nullptr,
            *cs_resolve = nullptr, *cs_counters = nullptr, *k2_ncuts = nullptr, *k3_nadded = nullptr,
            *k3_added = nullptr, *k3_term = nullptr, *pool_list = nullptr, *dump_idx = nullptr,
            *cs_need_tab = nullptr;
    double *d_y = nullptr, *cs_before = nullptr, *slab_pi = nullptr, *slab_pi0 = nullptr,
           *dump_T = nullptr, *dump_vec = nullptr;
    int32_t *h_cs = nullptr;   // pinned: [counters (4) | state fields 0..6 + w_ncut (8 x max_batch)]
    int dive = 0;  // this step was launched with the in-place dive: children in a row per node
    bool scored_once = false;  // K4 ran on this step (the first run's request counter was zeroed by K1)
    // the step finished on the device (finish_kernels.hip.h): the parents' records and the pool rows the
    // children may take go up with the batch, a summary + compact lists come back
    bool fast = false;
    int tabv = 0;                                // the table version this step's finish writes
    char *d_par = nullptr, *h_par = nullptr;     // [par_d (2 MB f64) | par_i (4 MB i32) | budget (per MB i32)]
    int32_t *c_info = nullptr, *c_cnt = nullptr, *c_eval = nullptr, *c_flag = nullptr;
    double *c_val = nullptr;
    mipx::FinishSummary *d_sum = nullptr;
    mipx::OpenEntry *d_open = nullptr;
    int32_t *d_dead = nullptr;
    mipx::PcSample *d_samples = nullptr, *h_samples = nullptr;   // h_samples: pinned staging of a host-finished step's samples
    int32_t *d_skeys = nullptr;                  // the samples' table entries, densely (pc_apply scans them)
    char *h_fin = nullptr;                       // pinned: [summary | table block | open entries | dead rows]
    std::vector<int32_t> budget;
    int B = 0;
    bool in_flight = false;
    double inflight_min = std::numeric_limits<double>::infinity();   // lowest inherited bound of the batch (exchange record)
};

struct mipx_tree {
    mipx_problem *prob = nullptr;
    mipx_ctx *ctx = nullptr;
    int n = 0, m = 0, n_int = 0;
    int rule = 0, search = 0, sb_iters = 5, max_batch = 1;
    int64_t capacity = 0;
    std::vector<int32_t> int_idx;
    // device pool + per-step buffers
    double *pool_l = nullptr, *pool_u = nullptr;
    int8_t *pool_v = nullptr;
    double *atab_T = nullptr, *atab_vec = nullptr;   // mipx_tree_reanchor: one anchor per re-anchored node
    int32_t *atab_idx = nullptr;
    int64_t atab_count = 0;
    int32_t *d_int_idx = nullptr, *d_pairs = nullptr, *d_pairs2 = nullptr;
    double *d_cost_l = nullptr, *d_cost_r = nullptr, *d_cost_l2 = nullptr, *d_cost_r2 = nullptr;
    uint8_t *d_has = nullptr, *d_has2 = nullptr;
    char *h_tab = nullptr;  // pinned mirror of [cost_l | cost_r | has_entry]: d_cost_l .. d_has are one allocation
    // the table block on the device: [cost_l | cost_r | has | pad | times_l | times_r | own sums (4 n)]; with the
    // device finish the device holds the table (every sample reaches it through pc_apply, in stream order) and
    // the host vectors are the snapshot read back with each step
    size_t tab_off2 = 0, tab_bytes = 0, tab_stride = 0;
    // Device finish: the table is a ring of kTabV versions.  Step k's finish (on stf, beside the node LPs of
    // step k + 1) copies the version at the tail and applies its samples to the copy; a launch reads the
    // version of the last step the HOST has finished -- complete by then, never written again while a
    // kernel in flight reads it, and the same version in every run.
    int tab_tail = 0, tab_host = 0;
    // Samples of a host-finished node / an exchange's merge are applied LATE, to the version at the tail of stf's
    // queue at that time: ev_tab[v] / tab_late[v] say that version v got such an update and when it is complete.
    // Only a launch that reads that very version waits for it (in the pipeline a launch reads an older one).
    hipEvent_t ev_tab[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    bool tab_late[8] = {false, false, false, false, false, false, false, false};
    double *h_delta = nullptr, *d_delta = nullptr;   // exchange: the other ranks' new samples (4 staging slots)
    int delta_turn = 0;
    std::vector<double> pc_others_prev;
    double *d_primal = nullptr;
    double primal_sent = std::numeric_limits<double>::infinity();   // what the device knows of the host's incumbent value
    bool fast_ok = false;           // steps are finished on the device where they file no probe request
    bool times_dirty = false;       // the host replaced the table: times go up with it
    size_t samples_cap = 0;
    std::vector<mipx::PcSample> pend_samples;   // a host-finished step's samples, on their way to the device table
    hipStream_t st2 = nullptr;  // strong-branching probes + re-scoring run beside the step in flight
    hipStream_t st3 = nullptr;  // children records of step k are written beside the node LPs of step k+1
    hipStream_t stf = nullptr;  // device finish: the finish kernels of step k and the table's version chain, beside the node LPs of step k+1
    hipEvent_t ev_child = nullptr;
    bool child_pending = false;
    bool cold_launch = false;       // the next launch_lp is the root's first solve (LpArgs::cold)
    int32_t *h_pairs = nullptr; // pinned staging of the branching lists
    char *h_pres = nullptr;     // pinned mirror of the probe results [pp_obj | pp_status]
    StepBuf buf[3];   // a ring: up to three steps in flight (mipx_tree_solve)
    bool table_dirty = false, pipeline = true;
    int dive = 0;           // mipx_tree_set_dive: dive children in a row per node (0: off)
    bool pool_exhausted = false;
    int64_t age_hist[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // popped nodes by age in steps (1 = created by the previous step)
    int64_t depth_sum = 0;
    int64_t dives = 0;      // dive children evaluated in place
    mipx_tree_hook hook = nullptr;
    void *hook_user = nullptr;
    int hook_every = 0;
    // probe pool (strong branching)
    int64_t probe_cap = 0;
    double *pp_l = nullptr, *pp_u = nullptr, *pp_obj = nullptr;
    int8_t *pp_v = nullptr;
    int32_t *pp_status = nullptr;
    // host state
    NodeTable nodes;
    std::vector<int32_t> free_slots;
    PyHeap heap;        // exact mode (max_batch == 1) and depth-first search
    BucketQueue bq;     // batched best-first search
    bool use_bq = false;
    std::vector<BucketQueue::Item> popped;
    std::vector<BucketQueue::Item> pend;   // a step's pushes, handed to the bucket queue in one go (tree_finish)
    std::priority_queue<std::pair<double, int64_t>, std::vector<std::pair<double, int64_t>>,
                        std::greater<std::pair<double, int64_t>>> open_bounds;  // lazy, for DFS
    std::vector<uint8_t> is_open;
    double closed_min = std::numeric_limits<double>::infinity();
    double primal = std::numeric_limits<double>::infinity();
    std::vector<double> best_x;
    bool have_x = false, unbounded = false, started = false;
    int status = 0;  // 0 unsolved, 1 optimal, 2 infeasible, 3 unbounded, 4 stopped
    int64_t evaluated = 0, lps = 0, probes = 0, pivots = 0, steps = 0, cut_resolves = 0;
    double solve_seconds = 0.0, kernel_ms = 0.0, k2_ms = 0.0, k3_ms = 0.0;
    std::vector<double> cost_l, cost_r;
    std::vector<int32_t> times_l, times_r;
    std::vector<uint8_t> has_entry;
    // trace of evaluated nodes: id, lp status, branch variable, objective
    std::vector<int64_t> tr_id;
    std::vector<int32_t> tr_status, tr_bidx;
    std::vector<double> tr_obj;
    std::vector<int32_t> tr_cuts;   // cut rounds: 8 per evaluated node (rounds, the six GMIC counters, cut rows it ends with)
    bool trace = false;
    bool anchor_mode = false, anchor_set = false;
    // multi-GPU exchange (mipx_tree_set_comm)
    mipx_comm *comm = nullptr;
    int x_every = 0;
    bool x_done = false;             // the ranks agreed to stop (set by an applied exchange)
    bool x_fatal = false;            // ... because a rank failed (reason 5)
    bool child_recorded = false;     // ev_child has been recorded at least once
    int fault_step = 0;              // MIPX_FAULT_STEP (tests): the step whose host half fails
    int x_stop_flag = 0;             // this rank hit one of its own limits (3: it failed)
    int64_t x_rounds = 0;            // exchanges applied
    int x_batch = 1;                 // frontier batch of the running solve (what "cannot fill a batch" means)
    double x_mip_gap = 0.0;
    std::vector<double> x_rec;       // this rank's record
    std::vector<double> pc_base, pc_own, pc_others;   // [sum_l | sum_r | times_l | times_r], n each
    int64_t ramp[4] = {0, 0, 0, 0};  // evaluated, lps, probes, pivots at sharding time (replicated ramp-up)
    int64_t g_counts[5] = {0, 0, 0, 0, 0};   // global evaluated, lps, probes, pivots, open
    double g_dual = -std::numeric_limits<double>::infinity();
    int g_inc_rank = -1;
    int64_t nodes_sent = 0, nodes_received = 0;
    // cut rounds
    bool cuts = false;
    mipx_cut_params cp{};
    int kc = 0;             // cut rows a node can carry
    int mrows = 0;          // rows allotted per node: m + kc (m without cut rounds)
    int slab_rows = 0;      // candidate cuts a node can hold while it is being bounded
    double *store_pi = nullptr, *store_pi0 = nullptr;   // the cut store (append-only)
    int32_t *store_count = nullptr;
    int64_t store_cap = 0;
    int32_t *pool_ncut = nullptr, *pool_ids = nullptr;  // node pool: the nodes' cut lists
    int32_t *pp_ncut = nullptr, *pp_ids = nullptr;      // probe pool
    uint8_t *d_is_int = nullptr;
    int64_t cut_totals[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // rounds, it/n created, it/n added, it/n removed, dropped
    double phase_ms[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // MIPX_TREE_PROFILE=1: host-side breakdown
    double probe_ms[4] = {0, 0, 0, 0};  // probes phase: requests | enqueue | wait | results
};

namespace {

template <typename T>
int dmalloc(mipx_ctx *ctx, T **p, size_t count) {
    HIP_TRY(ctx, hipMalloc((void **)p, (count ? count : 1) * sizeof(T)));
    return MIPX_OK;
}

double tree_open_min(mipx_tree *t) {
    const double inf = std::numeric_limits<double>::infinity();
    if (t->use_bq) return t->bq.empty() ? inf : t->bq.min_key();
    if (t->search == 0) return t->heap.empty() ? inf : t->heap.h[0].key;
    while (!t->open_bounds.empty() && !t->is_open[t->open_bounds.top().second]) t->open_bounds.pop();
    return t->open_bounds.empty() ? inf : t->open_bounds.top().first;
}

double tree_dual_bound(mipx_tree *t) { return std::fmin(tree_open_min(t), t->closed_min); }   // (of this rank's shard)

// reference current_gap (branch_and_bound.py:203-213); -1 encodes None
double tree_gap(mipx_tree *t) {
    const double inf = std::numeric_limits<double>::infinity();
    const double p = t->primal;
    if (p == inf) return -1.0;  // (before the dual bound: that one scans a bucket of the open list)
    const double d = t->comm ? t->g_dual : tree_dual_bound(t);   // (sharded: the ranks' MIN, as of the last exchange)
    if (p == 0 && d == 0) return 0.0;
    if (p == 0) return inf;
    return std::fabs(p - d) / std::fabs(p);
}

bool tree_queue_empty(const mipx_tree *t) { return t->use_bq ? t->bq.empty() : t->heap.empty(); }

// open nodes in queue order (heap array order / bucket order)
void tree_queue_ids(const mipx_tree *t, std::vector<int64_t> &out) {
    if (t->use_bq) {
        std::vector<BucketQueue::Item> items;
        t->bq.items(items);
        for (const auto &it : items) out.push_back(it.id);
    } else {
        for (const auto &it : t->heap.h) out.push_back(it.id);
    }
}

#ifdef MIPX_HOSTPROF
#include <x86intrin.h>
static unsigned long long g_hp_push = 0, g_hp_rec = 0, g_hp_n = 0;
#endif
void tree_push(mipx_tree *t, int64_t id) {
#ifdef MIPX_HOSTPROF
    const unsigned long long t0_ = __rdtsc();
    struct Acc { unsigned long long t0; ~Acc() { g_hp_push += __rdtsc() - t0; g_hp_n++; } } acc_{t0_};
#endif
    if (t->use_bq) t->bq.push(t->nodes[id].key, id);
    else t->heap.push(t->nodes[id].key, id);
    if (t->search != 0) {  // depth first: the dual bound comes from a lazy heap over the open nodes
        if ((size_t)id >= t->is_open.size()) t->is_open.resize(id + 1, 0);
        t->is_open[id] = 1;
        t->open_bounds.push({t->nodes[id].dual_bound, id});
    }
}

constexpr int kTabV = 8;   // table versions (device finish): one per step in flight and a few to spare
struct TabPtr { double *cl, *cr; uint8_t *has; int32_t *times; double *own; };
TabPtr tab_at(const mipx_tree *t, int v) {
    char *base = (char *)t->d_cost_l + (size_t)v * t->tab_stride;
    const size_t n = (size_t)t->n;
    TabPtr p;
    p.cl = (double *)base; p.cr = (double *)(base + 8 * n); p.has = (uint8_t *)(base + 16 * n);
    p.times = (int32_t *)(base + t->tab_off2); p.own = (double *)(base + t->tab_off2 + 8 * n);
    return p;
}

// what a launch over nodes with cut rows adds to launch_lp (cut rounds only)
struct CutLaunch {
    const int32_t *ncut = nullptr, *ids = nullptr;   // the nodes' cut lists, by batch position
    int vstat_by_node = 0;                           // the warm-start bases are dense too (l, u through slot)
    const int32_t *active = nullptr;                 // skip mask
    double *y = nullptr;                             // row duals out (batch x mrows)
    int m_rows = -1;                                 // largest row count in the launch (tile choice)
    double *dT = nullptr, *dvec = nullptr;           // dump every node's final tableau (K2's input)
    int32_t *didx = nullptr;
    bool no_anchor = false;
};

int launch_lp(mipx_tree *t, int batch, const double *l, const double *u, const int8_t *v,
              const int32_t *slot, int max_iter, int32_t *status, double *obj, double *x,
              int8_t *vout, int32_t *iters, int32_t *npiv, hipStream_t stream = nullptr,
              const StepBuf *dive = nullptr, const int32_t *asel = nullptr, const CutLaunch *cl = nullptr) {
    mipx::LpArgs a;
    if (dive) {  // in-place dive: K4's rule inside K1, level p's children at positions p * batch ..
        a.dive = dive->dive; a.dive_off = batch; a.rule = t->rule; a.n_int = t->n_int;
        const TabPtr tb = tab_at(t, t->fast_ok ? t->tab_host : 0);
        a.int_idx = t->d_int_idx; a.cost_l = tb.cl; a.cost_r = tb.cr; a.has_entry = tb.has;
        a.dive_cutoff = t->primal;
        a.dive_var = dive->d_dvar; a.dive_dir = dive->d_ddir; a.dive_val = dive->d_dval;
        a.dive_preset = 1;           // the kernel itself marks "no child / no dive" first
        a.zero16 = dive->d_ask_count; // and zeroes K4's request counter
    }
    a.m = t->m; a.n = t->n;
    a.A = t->prob->dA; a.b = t->prob->db; a.c = t->prob->dc;
    a.A_stride = a.b_stride = a.c_stride = 0;
    a.l = l; a.u = u; a.vstat_in = v; a.slot = slot; a.max_iter = max_iter;
    a.cold = t->cold_launch ? 1 : 0;   // (the root's step: its pool row holds no basis)
    t->cold_launch = false;
    a.anchor_T = t->prob->anchor_on ? t->prob->anchor_T : nullptr;
    a.anchor_vec = t->prob->anchor_on ? t->prob->anchor_vec : nullptr;
    a.anchor_idx = t->prob->anchor_on ? t->prob->anchor_idx : nullptr;
    if (asel != nullptr && t->atab_T != nullptr) {  // node LPs: the anchor their record names
        a.anchor_sel = asel; a.atab_T = t->atab_T; a.atab_vec = t->atab_vec; a.atab_idx = t->atab_idx;
    }
    a.refactor_only = 0;
    a.status = status; a.obj = obj; a.x = x; a.y = nullptr; a.vstat_out = vout;
    a.iters = iters; a.npivots = npiv; a.batch = batch;
    a.dbg_T = nullptr; a.dbg_vec = nullptr; a.dbg_idx = nullptr; a.dbg_all = 0;
    int m_rows = -1;
    if (cl != nullptr) {
        a.ncut = cl->ncut; a.cut_ids = cl->ids; a.cut_stride = t->kc;
        a.cut_pi = t->store_pi; a.cut_pi0 = t->store_pi0;
        a.mstride = t->mrows; a.vstat_by_node = cl->vstat_by_node; a.active = cl->active;
        a.y = cl->y;
        if (cl->dT) { a.dbg_T = cl->dT; a.dbg_vec = cl->dvec; a.dbg_idx = cl->didx; a.dbg_all = 1; }
        if (cl->no_anchor) { a.anchor_T = nullptr; a.anchor_vec = nullptr; a.anchor_idx = nullptr; }
        m_rows = cl->m_rows;
    }
    return launch_lp_any(t->prob, a, batch, stream, m_rows);
}

// Device -> host copies of the step loop go through the side stream, never the null stream: a
// null-stream copy shares a hardware queue with whatever the runtime mapped there, and in a
// process that also ran an ML framework's streams and RCCL that was the main stream with a 2 ms node-LP launch queued
// (measured: ~1 ms per synchronous hipMemcpy, 18 ms per 20 steps).
int tree_d2h(mipx_tree *t, void *dst, const void *src, size_t bytes) {
    HIP_TRY(t->ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, t->st2));
    HIP_TRY(t->ctx, hipStreamSynchronize(t->st2));
    return MIPX_OK;
}

constexpr int kAskCap = 2048;  // probe requests per step carried in the packed read-back
constexpr int kMaxDive = 8;    // dive children in a row per node (buffers are sized for it)

// What the host reads back every step, packed so that ONE copy into pinned memory fetches it, laid
// out for L = 1 + dive output levels of max_batch positions each:
//   [obj | bval] (L * MB f64 each) [dive_val] ((L - 1 or 1) * MB f64)
//   [status | bidx | mipf | nprobe | npiv] (L * MB i32 each) [dive_var | dive_dir] ((L - 1 or 1) * MB each)
//   then the probe requests
void layout_pack(mipx_tree *t, StepBuf &S, int levels) {
    const size_t MB = (size_t)t->max_batch, OB = (size_t)levels * MB, DB = (size_t)(levels > 1 ? levels - 1 : 1) * MB;
    S.d_obj = (double *)S.d_pack; S.d_bval = S.d_obj + OB; S.d_dval = S.d_bval + OB;
    S.d_status = (int32_t *)(S.d_dval + DB); S.d_bidx = S.d_status + OB; S.d_mipf = S.d_bidx + OB;
    S.d_nprobe = S.d_mipf + OB; S.d_npiv = S.d_nprobe + OB;
    S.d_dvar = S.d_npiv + OB; S.d_ddir = S.d_dvar + DB;
    S.ask_off = (OB * (2 * 8 + 5 * 4) + DB * (8 + 2 * 4) + 15) / 16 * 16;
    S.pack_bytes = S.ask_off + 16 + (size_t)kAskCap * sizeof(mipx::ScoreArgs::Ask);
    S.d_ask_count = (int32_t *)((char *)S.d_pack + S.ask_off);
    S.d_ask = (mipx::ScoreArgs::Ask *)((char *)S.d_pack + S.ask_off + 16);
}

int launch_score(mipx_tree *t, StepBuf &S, int batch, bool side = false, bool no_ask = false, int side_stream = -1) {
    const bool on_st2 = side_stream < 0 ? side : side_stream == 1;   // (default: the side tables go with the side stream)
    hipStream_t where = side_stream == 2 ? t->stf : on_st2 ? t->st2 : t->ctx->stream;   // (2: the finish stream)
    mipx::ScoreArgs s;
    s.n = t->n; s.n_int = t->n_int; s.batch = batch; s.rule = t->rule;
    s.int_idx = t->d_int_idx; s.x = S.d_x; s.status = S.d_status;
    const TabPtr tb = tab_at(t, t->fast_ok ? t->tab_host : 0);
    s.cost_l = side ? t->d_cost_l2 : tb.cl; s.cost_r = side ? t->d_cost_r2 : tb.cr;
    s.has_entry = side ? t->d_has2 : tb.has;
    s.branch_idx = S.d_bidx; s.branch_val = S.d_bval; s.mip_feasible = S.d_mipf;
    s.n_probe = S.d_nprobe;
    s.probe_list = S.d_plist;
    s.ask_count = S.d_ask_count; s.ask_cap = kAskCap; s.ask = (side || no_ask) ? nullptr : S.d_ask;
    s.ask_nodes = S.B;  // dive children (positions >= B) are never probed in their own step
    if (!side && !no_ask && !(S.dive && batch == (S.dive + 1) * S.B && !S.scored_once))
        HIP_TRY(t->ctx, hipMemsetAsync(S.d_ask_count, 0, 16, side_stream == 2 ? t->stf : t->ctx->stream));
    if (!no_ask) S.scored_once = true;
    hipLaunchKernelGGL(mipx::branch_score, dim3(batch), dim3(64), 0, where, s);
    HIP_TRY(t->ctx, hipGetLastError());
    return MIPX_OK;
}

// pseudo_cost.py:68-100
void pc_update(mipx_tree *t, int var, int dir, int lp_status, double objective, double dual_bound,
               double variable_change) {
    double &cost = dir ? t->cost_r[var] : t->cost_l[var];
    int32_t &times = dir ? t->times_r[var] : t->times_l[var];
    const size_t n = (size_t)t->n;
    if (lp_status == 0 || lp_status == 3) {
        double bc = objective - dual_bound;
        if (bc < 0) bc = 0;
        cost = (cost * (double)times + bc / variable_change) / (double)(times + 1);
        if (t->comm) t->pc_own[(dir ? n : 0) + (size_t)var] += bc / variable_change;   // this rank's samples, in sum form
    } else if (t->comm) {
        // a sample that counts without a cost leaves the mean where it is (pseudo_cost.py:97-100): in
        // sum form it weighs in with the current mean, so that sum / times reproduces the recurrence
        t->pc_own[(dir ? n : 0) + (size_t)var] += cost;
    }
    times += 1;
    if (t->comm) t->pc_own[(dir ? 3 * n : 2 * n) + (size_t)var] += 1.0;
    t->has_entry[var] = 1;
    if (t->fast_ok) {   // the device holds the table: the sample follows (pc_apply, tree_finish)
        mipx::PcSample sm;
        sm.var_dir = 2 * var + dir; sm.status = lp_status; sm.obj = objective; sm.bound = dual_bound; sm.vc = variable_change;
        t->pend_samples.push_back(sm);
    }
}

// the kernels that finish a step on the device, queued behind its scoring
int launch_finish(mipx_tree *t, StepBuf &S) {
    mipx_ctx *ctx = t->ctx;
    hipStream_t st = t->stf;   // (behind the step's scoring, queued there by tree_launch)
    const int prev = t->tab_tail;
    S.tabv = (prev + 1) % kTabV;
    t->tab_tail = S.tabv;
    t->tab_late[S.tabv] = false;   // (written afresh: a copy of the tail, which holds every late update queued so far)
    if (t->rule == 1)
        HIP_TRY(ctx, hipMemcpyAsync(tab_at(t, S.tabv).cl, tab_at(t, prev).cl, t->tab_bytes, hipMemcpyDeviceToDevice, st));
    const int B = S.B, L = t->dive + 1;
    const size_t MB = (size_t)t->max_batch;
    mipx::FinishArgs g;
    g.n = t->n; g.m = t->m; g.B = B; g.dive = S.dive; g.rule = t->rule;
    g.status = S.d_status; g.bidx = S.d_bidx; g.mipf = S.d_mipf; g.nprobe = S.d_nprobe; g.npiv = S.d_npiv;
    g.dvar = S.d_dvar; g.ddir = S.d_ddir; g.obj = S.d_obj; g.bval = S.d_bval; g.dval = S.d_dval;
    g.ask_count = S.d_ask_count; g.ask_cap = kAskCap; g.vout = S.d_vout;
    g.slot = S.d_slot;
    g.par_d = (const double *)S.d_par;
    g.par_i = (const int32_t *)(S.d_par + 2 * (size_t)B * 8);
    g.budget = g.par_i + 4 * (size_t)B;
    g.pool_l = t->pool_l; g.pool_u = t->pool_u; g.pool_v = t->pool_v; g.primal = t->d_primal;
    g.c_info = S.c_info; g.c_cnt = S.c_cnt; g.c_eval = S.c_eval; g.c_val = S.c_val; g.c_flag = S.c_flag;
    g.sum = S.d_sum; g.open = S.d_open; g.dead = S.d_dead; g.samples = S.d_samples; g.sample_keys = S.d_skeys;
    (void)L;
    g.c_run = S.c_val + 2 * MB;
    hipLaunchKernelGGL(mipx::finish_candidates, dim3((B + 255) / 256), dim3(256), 0, st, g);
    hipLaunchKernelGGL(mipx::finish_prefix_min, dim3(1), dim3(1024), 0, st, g);
    hipLaunchKernelGGL(mipx::finish_decide, dim3((B + 255) / 256), dim3(256), 0, st, g);
    hipLaunchKernelGGL(mipx::finish_scan, dim3(1), dim3(1024), 0, st, g);
    if (t->n <= 256) hipLaunchKernelGGL((mipx::finish_write<1>), dim3(B), dim3(256), 0, st, g);
    else if (t->n <= 512) hipLaunchKernelGGL((mipx::finish_write<2>), dim3(B), dim3(256), 0, st, g);
    else hipLaunchKernelGGL((mipx::finish_write<4>), dim3(B), dim3(256), 0, st, g);
    if (t->rule == 1) {
        hipLaunchKernelGGL(mipx::finish_samples, dim3((B + 255) / 256), dim3(256), 0, st, g);
        mipx::PcApplyArgs a;
        const TabPtr tb = tab_at(t, S.tabv);
        a.n = t->n; a.sum = S.d_sum; a.count = -1; a.samples = S.d_samples; a.keys = S.d_skeys;
        a.cost_l = tb.cl; a.cost_r = tb.cr; a.has = tb.has; a.times = tb.times; a.own = tb.own;
        hipLaunchKernelGGL(mipx::pc_apply, dim3(2 * t->n), dim3(64), 0, st, a);
    }
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipEventRecord(S.done, st));
    return MIPX_OK;
}

// The host replaced the table (mipx_tree_set_pseudo_costs): a new version at the tail, read from now on.
// Rare and blocking: nothing else orders a host-side replacement against the steps in flight.
int table_replace(mipx_tree *t) {
    mipx_ctx *ctx = t->ctx;
    const size_t n = (size_t)t->n;
    HIP_TRY(ctx, hipStreamSynchronize(t->stf));
    const int prev = t->tab_tail, w = (prev + 1) % kTabV;
    const TabPtr src = tab_at(t, prev), dst = tab_at(t, w);
    HIP_TRY(ctx, hipMemcpy(dst.cl, src.cl, t->tab_bytes, hipMemcpyDeviceToDevice));
    HIP_TRY(ctx, hipMemcpy(dst.cl, t->cost_l.data(), n * 8, hipMemcpyHostToDevice));
    HIP_TRY(ctx, hipMemcpy(dst.cr, t->cost_r.data(), n * 8, hipMemcpyHostToDevice));
    HIP_TRY(ctx, hipMemcpy(dst.has, t->has_entry.data(), n, hipMemcpyHostToDevice));
    HIP_TRY(ctx, hipMemcpy(dst.times, t->times_l.data(), n * 4, hipMemcpyHostToDevice));
    HIP_TRY(ctx, hipMemcpy(dst.times + n, t->times_r.data(), n * 4, hipMemcpyHostToDevice));
    t->tab_tail = w;
    t->tab_host = w;
    t->table_dirty = false;
    t->times_dirty = false;
    return MIPX_OK;
}

// First half of a step: pop the batch and enqueue its node LPs + scoring (no host wait).
int tree_launch(mipx_tree *t, StepBuf &S, int want) {
    mipx_ctx *ctx = t->ctx;
    hipStream_t st = ctx->stream;
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto tp = now();
    // 1. pop the batch (a node whose inherited bound cannot beat the incumbent is closed unevaluated)
    std::vector<int64_t> &ids = S.ids;
    std::vector<int32_t> &slots = S.slots;
    ids.clear();
    slots.clear();
    S.recs.clear();
    S.B = 0;
    S.in_flight = false;
    S.inflight_min = std::numeric_limits<double>::infinity();
    auto take = [&](int64_t id) {
        if (t->search != 0) t->is_open[id] = 0;
        NodeRec &nd = t->nodes[id];
        const int32_t slot = nd.slot;
        nd.slot = -1;  // (the row itself is released when the step is finished)
        if (!(nd.dual_bound < t->primal)) {
            t->closed_min = std::fmin(t->closed_min, nd.dual_bound);
            t->free_slots.push_back(slot);
            return;
        }
        ids.push_back(id);
        slots.push_back(slot);
        S.recs.push_back(nd);
        S.inflight_min = std::fmin(S.inflight_min, nd.dual_bound);
        { const int64_t age = t->steps + 1 - nd.born; t->age_hist[age < 1 ? 0 : age > 7 ? 7 : age]++; t->depth_sum += nd.depth; }
    };
    if (t->use_bq) {
        while ((int)ids.size() < want && !t->bq.empty()) {
            t->popped.clear();
            t->bq.pop_batch((size_t)want - ids.size(), t->popped);
            const size_t np = t->popped.size();
            for (size_t i = 0; i < np; i++) {  // the node table is far bigger than the caches
                if (i + 16 < np) __builtin_prefetch(&t->nodes[t->popped[i + 16].id], 1);
                take(t->popped[i].id);
            }
        }
    } else {
        while ((int)ids.size() < want && !tree_queue_empty(t)) take(t->heap.pop());
    }
    const int B = (int)ids.size();
    if (B == 0) return MIPX_OK;
    S.B = B;
    S.in_flight = true;
    t->steps++;
    t->phase_ms[0] += std::chrono::duration<double, std::milli>(now() - tp).count();
    if (t->fast_ok && t->table_dirty) {
        const int trc = table_replace(t);
        if (trc) return trc;
    }
    if (t->fast_ok && t->tab_late[t->tab_host]) {   // late samples / a merge on their way into the version read below
        HIP_TRY(ctx, hipStreamWaitEvent(st, t->ev_tab[t->tab_host], 0));
        t->tab_late[t->tab_host] = false;
    }
    if (t->table_dirty) {  // pseudo-cost table as of the last finished step
        const size_t n = t->n;
        // (one copy from the pinned mirror; the previous upload has long been consumed: two steps ago)
        char *ht = t->h_tab + (size_t)(t->steps & 3) * 17 * n;  // (four mirrors in turn: three steps can be in flight)
        std::memcpy(ht, t->cost_l.data(), n * 8);
        std::memcpy(ht + n * 8, t->cost_r.data(), n * 8);
        std::memcpy(ht + 2 * n * 8, t->has_entry.data(), n);
        HIP_TRY(ctx, hipMemcpyAsync(t->d_cost_l, ht, 17 * n, hipMemcpyHostToDevice, st));
        t->table_dirty = false;
    }
    S.fast = t->fast_ok && !t->trace;
    if (t->fast_ok && t->rule == 1 && std::getenv("MIPX_DEBUG_TABLE")) {
        // debugging aid: the version the launch reads against the host's snapshot (equal when nothing is in flight)
        (void)hipStreamSynchronize(t->stf);
        std::vector<char> blk(t->tab_bytes);
        (void)hipMemcpy(blk.data(), tab_at(t, t->tab_host).cl, t->tab_bytes, hipMemcpyDeviceToHost);
        const size_t n = (size_t)t->n;
        const double *cl = (const double *)blk.data(), *cr = cl + n;
        const uint8_t *has = (const uint8_t *)(blk.data() + 16 * n);
        const int32_t *tl = (const int32_t *)(blk.data() + t->tab_off2), *tr = tl + n;
        int bad = 0;
        for (size_t j = 0; j < n; j++)
            if (cl[j] != t->cost_l[j] || cr[j] != t->cost_r[j] || has[j] != t->has_entry[j] || tl[j] != t->times_l[j] || tr[j] != t->times_r[j]) {
                if (bad++ < 4)
                    std::fprintf(stderr, "[mipx table] step %lld var %zu: device (%.17g %.17g has %d times %d %d) host (%.17g %.17g has %d times %d %d)\n",
                                 (long long)t->steps, j, cl[j], cr[j], has[j], tl[j], tr[j], t->cost_l[j], t->cost_r[j], t->has_entry[j], t->times_l[j], t->times_r[j]);
            }
        if (bad) std::fprintf(stderr, "[mipx table] step %lld: %d entries differ (version %d, tail %d)\n", (long long)t->steps, bad, t->tab_host, t->tab_tail);
    }
    if (S.fast) {
        if (t->primal < t->primal_sent) {   // the host knows a better incumbent than the device (exchange, host-finished step)
            hipLaunchKernelGGL(mipx::primal_lower, dim3(1), dim3(1), 0, t->stf, t->d_primal, t->primal);
            t->primal_sent = t->primal;
        }
        // the parents' records and the pool rows the children may take: chain k, level p, direction d owns
        // budget[k][2 p + d] (the claim batch_size() left room for, handed out up front)
        const size_t per = 2 * (1 + (size_t)t->dive), need = per * (size_t)B;
        S.budget.assign(t->free_slots.end() - (std::ptrdiff_t)need, t->free_slots.end());
        t->free_slots.resize(t->free_slots.size() - need);
        double *pd = (double *)S.h_par;
        int32_t *pi = (int32_t *)(S.h_par + 2 * (size_t)B * 8);
        for (int k = 0; k < B; k++) {
            const NodeRec &nd = S.recs[(size_t)k];
            pd[k] = nd.dual_bound; pd[B + k] = nd.b_val;
            pi[k] = nd.b_idx; pi[B + k] = nd.b_dir; pi[2 * B + k] = nd.depth; pi[3 * B + k] = nd.anchor;
        }
        std::memcpy(pi + 4 * (size_t)B, S.budget.data(), need * 4);
        HIP_TRY(ctx, hipMemcpyAsync(S.d_par, S.h_par, 2 * (size_t)B * 8 + (4 * (size_t)B + need) * 4, hipMemcpyHostToDevice, st));
    }
    std::memcpy(S.h_slot, slots.data(), (size_t)B * 4);
    for (int k = 0; k < B; k++) S.h_slot[B + k] = S.recs[(size_t)k].anchor;  // [pool rows | anchor-table entries]
    HIP_TRY(ctx, hipMemcpyAsync(S.d_slot, S.h_slot, (size_t)B * 8, hipMemcpyHostToDevice, st));
    // 2. LP relaxations + scoring (after the children records of the last finished step)
    if (t->child_pending) {
        HIP_TRY(ctx, hipStreamWaitEvent(st, t->ev_child, 0));
        t->child_pending = false;
    }
    S.dive = t->dive;  // (register tiles and the HBM-streaming kernel alike)
    S.scored_once = false;
    int rc = MIPX_OK;
    if (t->cuts) {
        // working copies of the nodes' cut lists, the loop state zeroed; then the node LPs over their
        // m + ncut rows (row duals kept: the slack-cut removal reads them) and the integrality test
        HIP_TRY(ctx, hipMemsetAsync(S.cs_counters, 0, 16, st));
        mipx::CutGatherArgs ga;
        ga.batch = B; ga.kc = t->kc; ga.slot = S.d_slot;
        ga.pool_ncut = t->pool_ncut; ga.pool_ids = t->pool_ids;
        ga.ncut = S.w_ncut; ga.ids = S.w_ids; ga.state = S.cs_state; ga.counters = S.cs_counters;
        hipLaunchKernelGGL(mipx::cut_gather_state, dim3(B), dim3(64), 0, st, ga);
        HIP_TRY(ctx, hipGetLastError());
        int maxc = 0;
        for (int k = 0; k < B; k++) maxc = std::max(maxc, (int)S.recs[(size_t)k].ncut);
        CutLaunch cl;
        cl.ncut = S.w_ncut; cl.ids = S.w_ids; cl.y = S.d_y; cl.m_rows = t->m + maxc;
        if (t->cp.exact_tableau == 0) {   // the tableau the solve ends with is the first round's (K2's input)
            cl.dT = S.dump_T; cl.dvec = S.dump_vec; cl.didx = S.dump_idx;
        }
        HIP_TRY(ctx, hipEventRecord(S.e0, st));
        t->cold_launch = B == 1 && t->nodes.size() == 1 && S.recs[0].depth == 0 && S.recs[0].b_idx == -1 && S.recs[0].ncut == 0 &&
                         !t->prob->anchor_on;   // (the root alone, never solved, no cut yet: one cold LP)
        rc = launch_lp(t, B, t->pool_l, t->pool_u, t->pool_v, S.d_slot, 0, S.d_status, S.d_obj, S.d_x,
                       S.d_vout, S.d_iters, S.d_npiv, nullptr, nullptr, S.d_slot + B, &cl);
        if (rc) return rc;
        HIP_TRY(ctx, hipEventRecord(S.e1, st));
        if ((rc = launch_score(t, S, B, false, true))) return rc;
        HIP_TRY(ctx, hipEventRecord(S.done, st));
        return MIPX_OK;
    }
    HIP_TRY(ctx, hipEventRecord(S.e0, st));
    // the root alone, never solved: one cold LP (above the register tiles it is spread over the chip, K1c)
    t->cold_launch = B == 1 && t->nodes.size() == 1 && S.recs[0].depth == 0 && S.recs[0].b_idx == -1 && !t->prob->anchor_on;
    rc = launch_lp(t, B, t->pool_l, t->pool_u, t->pool_v, S.d_slot, 0, S.d_status, S.d_obj,
                   S.d_x, S.d_vout, S.d_iters, S.d_npiv, nullptr, S.dive ? &S : nullptr, S.d_slot + B);
    if (rc) return rc;
    HIP_TRY(ctx, hipEventRecord(S.e1, st));
    if (S.fast) {
        // scoring and finish beside the node LPs of the next step: stf waits for this step's node LPs, the main
        // stream goes straight on to the next launch
        HIP_TRY(ctx, hipStreamWaitEvent(t->stf, S.e1, 0));
        if ((rc = launch_score(t, S, (S.dive + 1) * B, false, false, 2))) return rc;
        return launch_finish(t, S);   // (records S.done behind the finish, on stf)
    }
    if ((rc = launch_score(t, S, (S.dive + 1) * B))) return rc;
    HIP_TRY(ctx, hipEventRecord(S.done, st));
    return MIPX_OK;
}

// The cut loop of BaseNode._base_bound (base_node.py:196-203) for the whole batch at once: round r
// of every node that is still generating runs together.  Per round two small read-backs (how many
// nodes go on; how many LPs changed and the largest row count), everything else stays on the device:
//   cut_round_begin   stall test of the last round, loop condition, slack-cut removal   (:292-341)
//   K1                the tableau of the basis on the remaining rows                     (:513-526)
//   K2 + pool_append  GMI cuts, safely rounded, into the node's pool                     (:365-385, :468-511)
//   K3                selection                                                          (:387-454)
//   cut_round_apply   selected cuts -> cut store, node list, basis; pool minus selected  (:456-463)
//   K1 + K4           re-solve of the nodes whose rows changed, integrality test         (:319)
// A node whose rows did not change is not re-solved (the reference re-solves the unchanged LP and
// gets the same objective: it stalls either way).
int tree_cut_rounds(mipx_tree *t, StepBuf &S) {
    mipx_ctx *ctx = t->ctx;
    hipStream_t st = ctx->stream;
    const int B = S.B, n = t->n;
    const size_t MB = (size_t)t->max_batch;
    int maxc = 0;
    for (int k = 0; k < B; k++) maxc = std::max(maxc, (int)S.recs[(size_t)k].ncut);
    // Without exact_tableau every LP launch of the loop dumps the tableau it ends with, and K2 reads that:
    // a launch of its own for the tableau only where a round starts by removing rows.  (exact_tableau:
    // the tableau is refactored from the slack basis like the per-node path's, always by its own launch.)
    const bool fused = t->cp.exact_tableau == 0;
    for (int round = 0;; round++) {
        HIP_TRY(ctx, hipMemsetAsync(S.cs_counters, 0, 8, st));   // n_active, n_changed (max_ncut stays)
        HIP_TRY(ctx, hipMemsetAsync(S.cs_counters + 3, 0, 4, st));   // n_need_tab
        mipx::CutRoundArgs ra;
        ra.n = n; ra.m0 = t->m; ra.mstride = t->mrows; ra.kc = t->kc; ra.batch = B; ra.round = round;
        ra.max_rounds = t->cp.max_cut_generation_iterations;
        ra.progress_tol = t->cp.cutting_plane_progress_tolerance; ra.max_dual_bound = t->cp.max_dual_bound;
        ra.status = S.d_status; ra.obj = S.d_obj; ra.mipf = S.d_mipf; ra.y = S.d_y; ra.vstat = S.d_vout;
        ra.ncut = S.w_ncut; ra.ids = S.w_ids; ra.state = S.cs_state; ra.obj_before = S.cs_before;
        ra.active = S.cs_active; ra.resolve = S.cs_resolve; ra.counters = S.cs_counters;
        ra.have_dump = fused ? 1 : 0; ra.need_tab = S.cs_need_tab;
        hipLaunchKernelGGL(mipx::cut_round_begin, dim3(B), dim3(64), 0, st, ra);
        HIP_TRY(ctx, hipGetLastError());
        HIP_TRY(ctx, hipMemcpyAsync(S.h_cs, S.cs_counters, 16, hipMemcpyDeviceToHost, st));
        HIP_TRY(ctx, hipStreamSynchronize(st));
        if (S.h_cs[0] == 0) break;   // nobody generates any more
        // the tableau of every generating node's basis (refactorisation from the slack basis -- or,
        // without exact_tableau, from the root's anchor where the node has no cut rows -- then zero
        // iterations: the basis is optimal), dumped for K2
        int rc = MIPX_OK;
        if (S.h_cs[3] > 0) {
            CutLaunch cl;
            cl.ncut = S.w_ncut; cl.ids = S.w_ids; cl.vstat_by_node = 1; cl.active = S.cs_need_tab;
            cl.m_rows = t->m + maxc; cl.dT = S.dump_T; cl.dvec = S.dump_vec; cl.didx = S.dump_idx;
            cl.no_anchor = t->cp.exact_tableau != 0;
            rc = launch_lp(t, B, t->pool_l, t->pool_u, S.d_vout, S.d_slot, 0, nullptr, nullptr, nullptr,
                           nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, &cl);
            if (rc) return rc;
        }
        mipx::GomoryArgs ga;
        ga.m = t->m; ga.n = n; ga.batch = B;
        ga.A = t->prob->dA; ga.b = t->prob->db;
        ga.T = S.dump_T; ga.idx = S.dump_idx; ga.x = S.d_x; ga.is_int = t->d_is_int;
        ga.max_term = t->cp.max_term;
        ga.ncuts = S.k2_ncuts; ga.row_idx = nullptr; ga.pi = nullptr; ga.pi0 = nullptr;
        ga.safe_pi = nullptr; ga.safe_pi0 = nullptr;
        ga.chunks = B >= 256 ? 1 : (B >= 32 ? 4 : 16);
        ga.ncut = S.w_ncut; ga.cut_ids = S.w_ids; ga.cut_pi = t->store_pi; ga.cut_pi0 = t->store_pi0;
        ga.cut_stride = t->kc; ga.mstride = t->mrows; ga.vstat = S.d_vout; ga.clip_x = 1;
        ga.active = S.cs_active;
        ga.slab_pi = S.slab_pi; ga.slab_pi0 = S.slab_pi0;
        ga.slab_n = S.cs_state + (size_t)mipx::CF_SLAB_N * B; ga.slab_rows = t->slab_rows;
        ga.group = mipx::gomory_group(n, t->mrows, (long)B * ga.chunks);
        ga.mfma = (std::getenv("MIPX_K2_MFMA") && std::atoi(std::getenv("MIPX_K2_MFMA"))) ? 1 : 0;   // (experiment: cut_kernels.hip.h)
        const size_t lds2 = mipx::gomory_lds_bytes(n, t->mrows, ga.group);
        HIP_TRY(ctx, hipEventRecord(S.k2a, st));
        if (lds2 > 64 * 1024)
            HIP_TRY(ctx, hipFuncSetAttribute((const void *)mipx::gomory_cuts<256>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2));
        hipLaunchKernelGGL((mipx::gomory_cuts<256>), dim3(B * ga.chunks), dim3(256), lds2, st, ga);
        HIP_TRY(ctx, hipGetLastError());
        HIP_TRY(ctx, hipEventRecord(S.k2b, st));
        mipx::PoolAppendArgs pa;
        pa.batch = B; pa.slab_rows = t->slab_rows; pa.active = S.cs_active; pa.k2_ncuts = S.k2_ncuts;
        pa.state = S.cs_state; pa.pool_list = S.pool_list;
        hipLaunchKernelGGL(mipx::pool_append, dim3((B + 63) / 64), dim3(64), 0, st, pa);
        HIP_TRY(ctx, hipGetLastError());
        mipx::SelectArgs sa;
        sa.n = n; sa.batch = B; sa.kmax = t->slab_rows;
        sa.npool = S.cs_state + (size_t)mipx::CF_POOL_N * B; sa.pi = S.slab_pi; sa.pi0 = S.slab_pi0; sa.x = S.d_x;
        sa.max_nonzero_coefs = t->cp.max_nonzero_coefs; sa.min_cut_depth = t->cp.min_cut_depth;
        sa.cos_parallel = t->cp.cos_parallel; sa.max_abs_coef = t->cp.max_abs_coef;
        sa.nadded = S.k3_nadded; sa.added = S.k3_added; sa.terminator = S.k3_term; sa.depth = nullptr;
        sa.pool_list = S.pool_list; sa.clip_x = 1; sa.active = S.cs_active;
        const size_t lds3 = (size_t)t->slab_rows * (3 * 8 + 2 * 4) + 16 + 64;
        hipLaunchKernelGGL(mipx::select_cuts, dim3(B), dim3(256), lds3, st, sa);
        HIP_TRY(ctx, hipGetLastError());
        HIP_TRY(ctx, hipEventRecord(S.k3b, st));
        mipx::CutApplyArgs aa;
        aa.n = n; aa.m0 = t->m; aa.mstride = t->mrows; aa.kc = t->kc; aa.batch = B; aa.slab_rows = t->slab_rows;
        aa.active = S.cs_active; aa.k3_nadded = S.k3_nadded; aa.k3_added = S.k3_added;
        aa.slab_pi = S.slab_pi; aa.slab_pi0 = S.slab_pi0; aa.pool_list = S.pool_list;
        aa.vstat = S.d_vout; aa.ncut = S.w_ncut; aa.ids = S.w_ids; aa.state = S.cs_state;
        aa.store_pi = t->store_pi; aa.store_pi0 = t->store_pi0; aa.store_count = t->store_count;
        aa.store_cap = (int)t->store_cap;
        aa.resolve = S.cs_resolve; aa.counters = S.cs_counters;
        hipLaunchKernelGGL(mipx::cut_round_apply, dim3(B), dim3(256), 0, st, aa);
        HIP_TRY(ctx, hipGetLastError());
        HIP_TRY(ctx, hipMemcpyAsync(S.h_cs, S.cs_counters, 16, hipMemcpyDeviceToHost, st));
        HIP_TRY(ctx, hipStreamSynchronize(st));
        {   // (the stream is idle: the round's kernel times for the roofline entries of the cut configs)
            float a = 0.f, b = 0.f;
            if (hipEventElapsedTime(&a, S.k2a, S.k2b) == hipSuccess) t->k2_ms += a;
            if (hipEventElapsedTime(&b, S.k2b, S.k3b) == hipSuccess) t->k3_ms += b;   // (pool_append + K3)
        }
        const int changed = S.h_cs[1];
        maxc = std::max(maxc, (int)S.h_cs[2]);
        if (changed > 0) {   // re-solve where rows came or went, warm from the node's own basis (:319)
            CutLaunch rl;
            rl.ncut = S.w_ncut; rl.ids = S.w_ids; rl.vstat_by_node = 1; rl.active = S.cs_resolve;
            rl.y = S.d_y; rl.m_rows = t->m + maxc;
            if (fused) { rl.dT = S.dump_T; rl.dvec = S.dump_vec; rl.didx = S.dump_idx; }
            rl.no_anchor = true;   // (a node that changed rows has, or just had, cut rows)
            rc = launch_lp(t, B, t->pool_l, t->pool_u, S.d_vout, S.d_slot, 0, S.d_status, S.d_obj, S.d_x,
                           S.d_vout, S.d_iters, S.d_npiv, nullptr, nullptr, nullptr, &rl);
            if (rc) return rc;
            if ((rc = launch_score(t, S, B, false, true))) return rc;
            t->lps += changed;
            t->cut_resolves += changed;
        }
    }
    // per node: rounds and the six GMIC counters, its final number of cut rows
    // (fields 0..6 lie behind one another on the device; with a full batch also on the host: one copy)
    if ((size_t)B == MB) {
        HIP_TRY(ctx, hipMemcpyAsync(S.h_cs + 4, S.cs_state, 7 * (size_t)B * 4, hipMemcpyDeviceToHost, st));
    } else {
        for (int f = 0; f < 7; f++)
            HIP_TRY(ctx, hipMemcpyAsync(S.h_cs + 4 + (size_t)f * MB, S.cs_state + (size_t)f * B, (size_t)B * 4,
                                        hipMemcpyDeviceToHost, st));
    }
    HIP_TRY(ctx, hipMemcpyAsync(S.h_cs + 4 + 7 * MB, S.w_ncut, (size_t)B * 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, hipMemcpyAsync(S.h_cs + 4 + 8 * MB, S.cs_state + (size_t)mipx::CF_DROPPED * B, (size_t)B * 4,
                                hipMemcpyDeviceToHost, st));
    // the final scoring of the batch (branching variable, strong-branching requests)
    int rc = launch_score(t, S, B);
    if (rc) return rc;
    HIP_TRY(ctx, hipEventRecord(S.done, st));
    HIP_TRY(ctx, hipStreamSynchronize(st));
    for (int k = 0; k < B; k++) {
        for (int f = 0; f < 7; f++) t->cut_totals[f] += S.h_cs[4 + (size_t)f * MB + k];
        t->cut_totals[7] += S.h_cs[4 + 8 * MB + k];
    }
    return MIPX_OK;
}

// The table block as the device holds it -> the host's vectors (a snapshot: x_fill_record, the API, the
// re-scoring of a host-finished step read them).
void table_snapshot(mipx_tree *t, const char *blk) {
    const size_t n = (size_t)t->n;
    std::memcpy(t->cost_l.data(), blk, n * 8);
    std::memcpy(t->cost_r.data(), blk + n * 8, n * 8);
    std::memcpy(t->has_entry.data(), blk + 16 * n, n);
    std::memcpy(t->times_l.data(), blk + t->tab_off2, n * 4);
    std::memcpy(t->times_r.data(), blk + t->tab_off2 + n * 4, n * 4);
    if (t->comm) std::memcpy(t->pc_own.data(), blk + t->tab_off2 + 8 * n, 4 * n * 8);
}

// Second half of a step the device finished (finish_kernels.hip.h): counters, the new open nodes into the
// node table and the queue, the free rows back, the incumbent, the table snapshot.
int tree_finish_fast(mipx_tree *t, StepBuf &S, const mipx::FinishSummary &sum, std::vector<int> &deferred) {
    mipx_ctx *ctx = t->ctx;
    const int B = S.B, n = t->n;
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto tp = now();
    deferred.clear();
    t->lps += (B - sum.n_deferred) + sum.dives;
    t->dives += sum.dives;
    t->evaluated += sum.evaluated;
    t->pivots += sum.pivots;
    t->closed_min = std::fmin(t->closed_min, sum.closed_min);
    if (sum.unbounded) t->unbounded = true;
    const size_t per = 2 * (1 + (size_t)t->dive);
    if (sum.n_open < 0 || (size_t)sum.n_open > per * (size_t)B || sum.n_dead < 0 || (size_t)sum.n_dead > per * (size_t)B ||
        (size_t)sum.n_open + (size_t)sum.n_dead != per * (size_t)B)
        return fail(ctx, MIPX_EHIP, "tree: the device finish returned an inconsistent summary");
    char *hp = S.h_fin + 128;
    const bool tab = t->rule == 1;
    mipx::OpenEntry *open = (mipx::OpenEntry *)(hp + (t->tab_bytes + 31) / 32 * 32);
    int32_t *dead = (int32_t *)(open + per * (size_t)t->max_batch);
    if (tab) HIP_TRY(ctx, hipMemcpyAsync(hp, tab_at(t, S.tabv).cl, t->tab_bytes, hipMemcpyDeviceToHost, t->st2));
    t->tab_host = S.tabv;   // launches from now on read the table as of this step
    if (sum.n_open > 0)
        HIP_TRY(ctx, hipMemcpyAsync(open, S.d_open, (size_t)sum.n_open * sizeof(mipx::OpenEntry), hipMemcpyDeviceToHost, t->st2));
    if (sum.n_dead > 0)
        HIP_TRY(ctx, hipMemcpyAsync(dead, S.d_dead, (size_t)sum.n_dead * 4, hipMemcpyDeviceToHost, t->st2));
    if (sum.n_deferred > 0)   // the nodes that filed probe requests: everything the host loop reads per node
        HIP_TRY(ctx, hipMemcpyAsync(S.h_pack, S.d_pack, S.pack_bytes, hipMemcpyDeviceToHost, t->st2));
    HIP_TRY(ctx, hipStreamSynchronize(t->st2));
    if (sum.n_deferred > 0) {
        const int32_t asked = *(const int32_t *)(S.h_pack + S.ask_off);
        if (asked < 1 || asked > kAskCap) return fail(ctx, MIPX_EHIP, "tree: the device finish deferred nodes without a request list");
        const mipx::ScoreArgs::Ask *ask = (const mipx::ScoreArgs::Ask *)(S.h_pack + S.ask_off + 16);
        for (int32_t q = 0; q < asked; q++) deferred.push_back(ask[q].node);
        std::sort(deferred.begin(), deferred.end());
        deferred.erase(std::unique(deferred.begin(), deferred.end()), deferred.end());
        if ((int)deferred.size() != sum.n_deferred) return fail(ctx, MIPX_EHIP, "tree: deferred nodes and request list disagree");
    }
    t->phase_ms[1] += std::chrono::duration<double, std::milli>(now() - tp).count(); tp = now();
    if (tab) table_snapshot(t, hp);
    // the new open nodes, in the order the host loop created them (chain by chain, level by level, left
    // then right)
    const bool defer_push = t->use_bq && t->search == 0;
    t->pend.clear();
    for (int q = 0; q < sum.n_open; q++) {
        const mipx::OpenEntry &e = open[q];
        NodeRec c;
        c.dual_bound = e.key;
        c.depth = e.depth;
        c.key = t->search == 0 ? e.key : -(double)e.depth;
        c.b_idx = e.bidx_dir >> 1; c.b_dir = e.bidx_dir & 1; c.b_val = e.bval;
        c.born = (int32_t)t->steps; c.anchor = e.anchor; c.ncut = 0; c.slot = e.slot;
        t->nodes.push_back(c);
        const int64_t cid = (int64_t)t->nodes.size() - 1;
        if (defer_push) t->pend.push_back({c.key, cid});
        else tree_push(t, cid);
    }
    if (defer_push) t->bq.push_many(t->pend.data(), t->pend.size());
    // rows free again: the budget rows that hold no open node, the batch's own rows
    t->free_slots.insert(t->free_slots.end(), dead, dead + sum.n_dead);
    if (deferred.empty()) {
        t->free_slots.insert(t->free_slots.end(), S.slots.begin(), S.slots.begin() + B);
    } else {    // (a deferred node's row feeds its children: the host part releases it)
        size_t d = 0;
        for (int k = 0; k < B; k++) {
            if (d < deferred.size() && deferred[d] == k) { d++; continue; }
            t->free_slots.push_back(S.slots[(size_t)k]);
        }
    }
    S.budget.clear();
    t->phase_ms[3] += std::chrono::duration<double, std::milli>(now() - tp).count(); tp = now();
    t->primal_sent = std::fmin(t->primal_sent, sum.primal);
    if (sum.incumbent_pos >= 0 && sum.primal < t->primal) {
        t->primal = sum.primal;
        int rc = tree_d2h(t, t->best_x.data(), S.d_x + (size_t)sum.incumbent_pos * n, (size_t)n * 8);
        if (rc) return rc;
        t->have_x = true;
    }
    // anchor mode: the root's optimal tableau (a root that needed no probe comes this way)
    if (t->anchor_mode && !t->anchor_set && S.ids[0] == 0 && (deferred.empty() || deferred[0] != 0)) {
        int32_t st0 = -1;
        int rc = tree_d2h(t, &st0, S.d_status, 4);
        if (rc) return rc;
        if (st0 == 0) {
            std::vector<int8_t> rootv((size_t)t->n + t->m);
            HIP_TRY(ctx, hipMemcpy(rootv.data(), S.d_vout, rootv.size(), hipMemcpyDeviceToHost));
            const int arc = mipx_problem_set_anchor(t->prob, rootv.data());
            if (arc) return arc;
            t->anchor_set = true;
        }
    }
    t->phase_ms[4] += std::chrono::duration<double, std::milli>(now() - tp).count();
    return MIPX_OK;
}

// Second half: wait for that batch only, then the reference's bookkeeping and the children.
int tree_finish(mipx_tree *t, StepBuf &S, bool overlapped) {
    mipx_ctx *ctx = t->ctx;
    const int n = t->n, nv = t->n + t->m;
    const double inf = std::numeric_limits<double>::infinity();
    hipStream_t st = ctx->stream;
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto ms_since = [&](std::chrono::steady_clock::time_point t0) {
        return std::chrono::duration<double, std::milli>(now() - t0).count();
    };
    auto tp = now();
    const int B = S.B;
    if (!S.in_flight || B == 0) return MIPX_OK;
    S.in_flight = false;
    const std::vector<int64_t> &ids = S.ids;
    const std::vector<int32_t> &slots = S.slots;
    int rc = MIPX_OK;
    if (t->cuts && (rc = tree_cut_rounds(t, S))) return rc;
    HIP_TRY(ctx, hipEventSynchronize(S.done));
    {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, S.e0, S.e1) == hipSuccess) t->kernel_ms += ms;
    }
    // The nodes this function finishes, in order.  The whole batch: the nodes without probe requests first,
    // then those with (each group in batch order) -- the order in which the device finish and the host part
    // below share a step, so that both ways end with the same table, ids and queue.  Device finish: only
    // the nodes it deferred (they filed probe requests).
    std::vector<int> todo;
    bool partial = false;
    if (S.fast) {
        mipx::FinishSummary *hs = (mipx::FinishSummary *)S.h_fin;
        if ((rc = tree_d2h(t, hs, S.d_sum, sizeof(mipx::FinishSummary)))) return rc;
        if (!hs->host_path) {
            if ((rc = tree_finish_fast(t, S, *hs, todo))) return rc;
            if (todo.empty()) return MIPX_OK;
            partial = true;    // (the pack came with the lists)
        } else {
            // more probe requests than the compact list holds: the host finishes the whole step; the rows
            // handed out for its children come back
            t->free_slots.insert(t->free_slots.end(), S.budget.begin(), S.budget.end());
            S.budget.clear();
        }
    }
    // one copy (pinned destination) for everything the host reads per node
    if (!partial && (rc = tree_d2h(t, S.h_pack, S.d_pack, S.pack_bytes))) return rc;
    const int L = t->dive + 1;   // (the pack is laid out for the tree's dive depth: layout_pack)
    const size_t MB = (size_t)t->max_batch, OB = (size_t)L * MB, DB = (size_t)(L > 1 ? L - 1 : 1) * MB;
    double *obj = (double *)S.h_pack, *bval = obj + OB, *dval = bval + OB;
    int32_t *status = (int32_t *)(dval + DB), *bidx = status + OB, *mipf = bidx + OB, *nprobe = mipf + OB,
            *npiv = nprobe + OB, *dvar = npiv + OB, *ddir = dvar + DB;
    const int NB = (S.dive + 1) * B;  // output positions in use: the batch, then its dive children level by level
    // The node at position pos (level pos / B) was followed in place by a dive child at pos + B: it
    // counts when its LP was solved and it needs no strong-branching initialisation of its own (else
    // it is dropped and queued like any other child).  The decision taken after pos is dvar[pos].
    auto dived = [&](int pos) {
        return pos < S.dive * B && dvar[pos] >= 0 && status[pos + B] >= 0 && nprobe[pos + B] == 0;
    };
    if (!partial) {
        for (int k = 0; k < B; k++)
            if (nprobe[k] == 0) todo.push_back(k);
        for (int k = 0; k < B; k++)
            if (nprobe[k] != 0) todo.push_back(k);
    }
    t->lps += (int64_t)todo.size();
    for (int k : todo) t->pivots += npiv[k];
    t->phase_ms[1] += ms_since(tp); tp = now();

    // 3. pseudo costs: strong-branch initialisation + the update for the branch that made the node
    if (t->rule == 1) {
        int64_t total = 0;
        for (int k : todo) total += nprobe[k];
        std::vector<int32_t> plist, pair_pos, pair_var, pair_slot, child_slot;
        std::vector<double> xrow;  // x of the probed variables
        std::vector<int32_t> pst;
        std::vector<double> pobj;
        // a step in flight on the main stream: probes and re-scoring go to the side stream (K1b streams
        // slabs of its own there)
        const bool use_side = overlapped;
        hipStream_t ps = use_side ? t->st2 : st;
        if (total > 0) {
            if (2 * total > t->probe_cap) return fail(ctx, MIPX_ENOMEM, "tree: probe pool exhausted");
            // The probe requests came with the packed read-back (K4 writes one compact entry per
            // request); only a step with more than kAskCap of them -- the ramp-up -- reads the
            // per-node lists and solutions in bulk.
            const int32_t asked = *(const int32_t *)(S.h_pack + S.ask_off);
            if (asked == total && asked <= kAskCap) {
                using Ask = mipx::ScoreArgs::Ask;
                const Ask *ask = (const Ask *)(S.h_pack + S.ask_off + 16);
                std::vector<Ask> es(ask, ask + asked);
                std::sort(es.begin(), es.end(), [](const Ask &a, const Ask &b) {
                    return a.node < b.node || (a.node == b.node && a.k < b.k);
                });
                for (const Ask &e : es) {
                    pair_pos.push_back(e.node);
                    pair_slot.push_back(slots[e.node]);
                    pair_var.push_back(t->int_idx[e.k]);
                    xrow.push_back(e.x);
                }
            } else {
                std::vector<double> xall((size_t)B * n);
                plist.resize((size_t)B * t->n_int);
                HIP_TRY(ctx, hipMemcpy(plist.data(), S.d_plist, plist.size() * 4, hipMemcpyDeviceToHost));
                HIP_TRY(ctx, hipMemcpy(xall.data(), S.d_x, xall.size() * 8, hipMemcpyDeviceToHost));
                for (int k = 0; k < B; k++) {
                    const double *xk = xall.data() + (size_t)k * n;
                    for (int e = 0; e < nprobe[k]; e++) {
                        const int var = t->int_idx[plist[(size_t)k * t->n_int + e]];
                        pair_pos.push_back(k);
                        pair_slot.push_back(slots[k]);
                        pair_var.push_back(var);
                        xrow.push_back(xk[var]);
                    }
                }
            }
            auto tq = now();
            t->probe_ms[0] += ms_since(tp);
            const int P = (int)pair_pos.size();
            child_slot.resize(2 * (size_t)P);
            for (int c = 0; c < 2 * P; c++) child_slot[c] = c;
            // d_pairs layout: [parent_slot | parent_pos | var | child_slot(2P)]
            // (on the side stream when a step is in flight: the probes run beside its node LPs)
            int32_t *dp = use_side ? t->d_pairs2 : t->d_pairs;
            HIP_TRY(ctx, hipMemcpyAsync(dp, pair_slot.data(), (size_t)P * 4, hipMemcpyHostToDevice, ps));
            HIP_TRY(ctx, hipMemcpyAsync(dp + P, pair_pos.data(), (size_t)P * 4, hipMemcpyHostToDevice, ps));
            HIP_TRY(ctx, hipMemcpyAsync(dp + 2 * P, pair_var.data(), (size_t)P * 4, hipMemcpyHostToDevice, ps));
            HIP_TRY(ctx, hipMemcpyAsync(dp + 3 * P, child_slot.data(), (size_t)P * 8, hipMemcpyHostToDevice, ps));
            mipx::ChildArgs ca;
            ca.n = n; ca.m = t->m; ca.count = P;
            ca.src_l = t->pool_l; ca.src_u = t->pool_u;
            ca.parent_slot = dp; ca.parent_pos = dp + P; ca.var = dp + 2 * P;
            ca.x = S.d_x; ca.vstat = S.d_vout;
            ca.dst_l = t->pp_l; ca.dst_u = t->pp_u; ca.dst_v = t->pp_v;
            ca.child_slot = dp + 3 * P;
            CutLaunch pcl;
            if (t->cuts) {  // a probe is a child: it has its parent's rows, cuts included (base_node.py:602-606)
                ca.mstride = t->mrows; ca.kc = t->kc;
                ca.src_ncut = S.w_ncut; ca.src_ids = S.w_ids; ca.dst_ncut = t->pp_ncut; ca.dst_ids = t->pp_ids;
                int maxc = 0;
                for (int k = 0; k < B; k++) maxc = std::max(maxc, (int)S.h_cs[4 + 7 * MB + k]);
                pcl.ncut = t->pp_ncut; pcl.ids = t->pp_ids; pcl.m_rows = t->m + maxc;
            }
            hipLaunchKernelGGL(mipx::make_children, dim3(2 * P), dim3(256), 0, ps, ca);
            HIP_TRY(ctx, hipGetLastError());
            // truncated dual simplex on every probe (base_node.py:645-646)
            rc = launch_lp(t, 2 * P, t->pp_l, t->pp_u, t->pp_v, nullptr, t->sb_iters, t->pp_status,
                           t->pp_obj, nullptr, nullptr, nullptr, nullptr, ps, nullptr, nullptr,
                           t->cuts ? &pcl : nullptr);
            if (rc) return rc;
            pst.resize(2 * (size_t)P);
            pobj.resize(2 * (size_t)P);
            t->probe_ms[1] += ms_since(tq); tq = now();
            HIP_TRY(ctx, hipStreamSynchronize(ps));
            t->probe_ms[2] += ms_since(tq); tq = now();
            HIP_TRY(ctx, hipMemcpyAsync(t->h_pres, t->pp_obj, pobj.size() * 8, hipMemcpyDeviceToHost, t->st2));
            if ((rc = tree_d2h(t, t->h_pres + (size_t)t->probe_cap * 8, t->pp_status, pst.size() * 4))) return rc;
            std::memcpy(pobj.data(), t->h_pres, pobj.size() * 8);
            std::memcpy(pst.data(), t->h_pres + (size_t)t->probe_cap * 8, pst.size() * 4);
            t->probe_ms[3] += ms_since(tq);
            t->probes += 2 * P;
        }
        t->phase_ms[5] += ms_since(tp); tp = now();
        // table updates in the reference's order: node by node; per node its probes (ascending
        // integer index, left then right), then its own branch unless just initialised
        size_t e = 0;
        bool changed = false;
        for (int k : todo) {   // (the requests are sorted by node: the nodes with requests come in that order)
            const bool lp_feasible = status[k] == 0 || status[k] == 2;
            const NodeRec &nd = S.recs[k];
            bool own_probed = false;
            if (lp_feasible) {
                for (int q = 0; q < nprobe[k]; q++, e++) {
                    const int var = pair_var[e];
                    const double bv = xrow[e];
                    pc_update(t, var, 0, pst[2 * e], pobj[2 * e], obj[k], bv - std::floor(bv));
                    pc_update(t, var, 1, pst[2 * e + 1], pobj[2 * e + 1], obj[k], std::ceil(bv) - bv);
                    if (var == nd.b_idx) own_probed = true;
                    changed = true;
                }
                if (nd.b_idx >= 0 && !own_probed) {
                    // variable_change: b_val - u[b_idx] (left) or l[b_idx] - b_val (right)
                    const double vc = nd.b_dir == 0 ? nd.b_val - std::floor(nd.b_val)
                                                    : std::ceil(nd.b_val) - nd.b_val;
                    pc_update(t, nd.b_idx, nd.b_dir, status[k], obj[k], nd.dual_bound, vc);
                    changed = true;
                }
                // the dive child: the update for the branch that made it -- like any node only if its
                // own LP is feasible (pseudo_cost.py:42-43), and only where step 4 will accept the
                // dive (the reference never creates that child under a pruned or integral parent)
                // (a plunge: level by level, each child under the node before it)
                for (int pos = k; dived(pos) && obj[pos] < t->primal && !mipf[pos] &&
                                  (status[pos] == 0 || status[pos] == 2); pos += B) {
                    const int cp = pos + B;
                    if (!(status[cp] == 0 || status[cp] == 2)) break;
                    const double vc = ddir[pos] == 0 ? dval[pos] - std::floor(dval[pos]) : std::ceil(dval[pos]) - dval[pos];
                    pc_update(t, dvar[pos], ddir[pos], status[cp], obj[cp], obj[pos], vc);
                    changed = true;
                }
            } else {
                e += 0;
            }
        }
        if (changed && !t->fast_ok) t->table_dirty = true;   // (device finish: the samples go to the device table below)
        // re-score with the updated table: always in the sequential mode (the reference branches
        // with the table its own node just updated); when steps overlap, only if probes created
        // entries that the first scoring had to leave out (the wait covers the step in flight)
        t->phase_ms[6] += ms_since(tp); tp = now();
        if (t->fast_ok && !t->pend_samples.empty()) {
            // the device holds the table: this step's samples reach it in the order they were applied here
            const size_t cnt = t->pend_samples.size();
            if (cnt > t->samples_cap) return fail(ctx, MIPX_ENOMEM, "tree: more pseudo-cost samples than the staging holds");
            std::memcpy(S.h_samples, t->pend_samples.data(), cnt * sizeof(mipx::PcSample));
            int32_t *hk = (int32_t *)(S.h_samples + t->samples_cap);   // (the keys again, densely, behind the samples)
            for (size_t q = 0; q < cnt; q++) hk[q] = t->pend_samples[q].var_dir;
            t->pend_samples.clear();
            HIP_TRY(ctx, hipMemcpyAsync(S.d_skeys, hk, cnt * 4, hipMemcpyHostToDevice, t->stf));
            // (onto the version at the tail of stf's queue: every later version is copied from it)
            HIP_TRY(ctx, hipMemcpyAsync(S.d_samples, S.h_samples, cnt * sizeof(mipx::PcSample), hipMemcpyHostToDevice, t->stf));
            const TabPtr tb = tab_at(t, t->tab_tail);
            mipx::PcApplyArgs pa;
            pa.n = n; pa.sum = nullptr; pa.count = (int)cnt; pa.samples = S.d_samples; pa.keys = S.d_skeys;
            pa.cost_l = tb.cl; pa.cost_r = tb.cr; pa.has = tb.has; pa.times = tb.times; pa.own = tb.own;
            hipLaunchKernelGGL(mipx::pc_apply, dim3(2 * n), dim3(64), 0, t->stf, pa);
            HIP_TRY(ctx, hipGetLastError());
            HIP_TRY(ctx, hipEventRecord(t->ev_tab[t->tab_tail], t->stf));
            t->tab_late[t->tab_tail] = true;
        }
        if (t->fast_ok) t->tab_host = S.fast ? S.tabv : t->tab_tail;
        if (changed && (!overlapped || total > 0)) {
            // (device finish: the re-scoring reads copies of the host's snapshot + this step's updates, the device
            // table itself only ever changes through pc_apply)
            const bool side_tab = use_side || t->fast_ok;
            double *cl = side_tab ? t->d_cost_l2 : t->d_cost_l, *cr = side_tab ? t->d_cost_r2 : t->d_cost_r;
            uint8_t *ch = side_tab ? t->d_has2 : t->d_has;
            HIP_TRY(ctx, hipMemcpyAsync(cl, t->cost_l.data(), (size_t)n * 8, hipMemcpyHostToDevice, ps));
            HIP_TRY(ctx, hipMemcpyAsync(cr, t->cost_r.data(), (size_t)n * 8, hipMemcpyHostToDevice, ps));
            HIP_TRY(ctx, hipMemcpyAsync(ch, t->has_entry.data(), (size_t)n, hipMemcpyHostToDevice, ps));
            if (!side_tab) t->table_dirty = false;
            if ((rc = launch_score(t, S, NB, side_tab, false, use_side ? 1 : 0))) return rc;
            HIP_TRY(ctx, hipStreamSynchronize(ps));
            if (!overlapped) {   // sequential mode: every node branches with the table its step just updated
                if ((rc = tree_d2h(t, bidx, S.d_bidx, (size_t)NB * 4))) return rc;
                if ((rc = tree_d2h(t, bval, S.d_bval, (size_t)NB * 8))) return rc;
            } else {
                // batches: only the nodes whose probes created the entries their first scoring had to leave out
                // take the new choice; a node that asked for nothing branches as it was scored at the launch
                // (what the device finish does with it, before the host ever sees the step)
                std::vector<int32_t> nb((size_t)B);
                std::vector<double> nvv((size_t)B);
                if ((rc = tree_d2h(t, nb.data(), S.d_bidx, (size_t)B * 4))) return rc;
                if ((rc = tree_d2h(t, nvv.data(), S.d_bval, (size_t)B * 8))) return rc;
                for (int k : todo)
                    if (nprobe[k] > 0) { bidx[k] = nb[(size_t)k]; bval[k] = nvv[(size_t)k]; }
            }
        }
    }

    t->phase_ms[2] += ms_since(tp); tp = now();
    // 4. the reference's _evaluate_node bookkeeping, node by node (a dive child right after its
    //    parent: it was solved in the same workgroup)
    std::vector<int32_t> &dive_slots = S.dive_slots;
    if ((int)S.br.size() < L) S.br.resize((size_t)L);
    for (auto &bl : S.br) { bl.pos.clear(); bl.slot.clear(); bl.var.clear(); bl.child.clear(); }
    dive_slots.clear();
    int incumbent_pos = -1;
    // One evaluated node at output position pos.  level 0: a node of the batch (pool row
    // `slot`); level 1: a dive child.  Returns the id of the child that was solved in place by
    // the dive (to be evaluated next), or -1.
    // (best first on the bucket queue: the pushes of the step are collected and queued together below)
    const bool defer_push = t->use_bq && t->search == 0;
    t->pend.clear();
    auto evaluate = [&](int64_t id, int pos, int32_t slot, int level, int depth, int32_t anchor, int &err) -> int64_t {
        t->evaluated++;
        const bool lp_feasible = status[pos] == 0 || status[pos] == 2;
        if (status[pos] == 2) t->unbounded = true;
        int branched_on = -1;
        int64_t dive_child = -1;
        double leaf_value = lp_feasible ? obj[pos] : inf;
        if (lp_feasible && obj[pos] < t->primal) {
            const bool take_dive = dived(pos);
            const int bvar = take_dive ? dvar[pos] : bidx[pos];  // a dive has already branched
            if (mipf[pos]) {
                t->primal = obj[pos];
                incumbent_pos = pos;
            } else if (bvar >= 0) {
                if (t->free_slots.size() < 2) {
                    err = fail(ctx, MIPX_ENOMEM, "tree: node pool exhausted");
                    return -1;
                }
                branched_on = bvar;
                const double xv = take_dive ? dval[pos] : bval[pos];
                for (int dir = 0; dir < 2; dir++) {
                    NodeRec c;
                    c.dual_bound = obj[pos];
                    c.depth = depth + 1;
                    c.key = t->search == 0 ? c.dual_bound : -(double)c.depth;
                    c.b_idx = branched_on; c.b_dir = dir; c.b_val = xv;
                    c.born = (int32_t)t->steps;
                    c.anchor = anchor;
                    c.ncut = t->cuts ? S.h_cs[4 + 7 * MB + pos] : 0;   // the rows of its parent, cuts included
                    c.slot = t->free_slots.back();
                    t->free_slots.pop_back();
                    S.br[(size_t)level].child.push_back(c.slot);
#ifdef MIPX_HOSTPROF
                    const unsigned long long r0_ = __rdtsc();
#endif
                    t->nodes.push_back(c);
#ifdef MIPX_HOSTPROF
                    g_hp_rec += __rdtsc() - r0_;
#endif
                    const int64_t cid = (int64_t)t->nodes.size() - 1;
                    if (take_dive && dir == ddir[pos]) {
                        dive_child = cid;  // already solved: never enters the queue
                    } else if (defer_push) {
                        t->pend.push_back({c.key, cid});
                    } else {
                        tree_push(t, cid);
                    }
                }
                S.br[(size_t)level].pos.push_back(pos);
                S.br[(size_t)level].slot.push_back(slot);
                S.br[(size_t)level].var.push_back(branched_on);
                leaf_value = inf;  // no longer a leaf
            }
        }
        if (branched_on < 0) t->closed_min = std::fmin(t->closed_min, leaf_value);
        if (t->trace) {
            t->tr_id.push_back(id); t->tr_status.push_back(status[pos]);
            t->tr_bidx.push_back(branched_on); t->tr_obj.push_back(obj[pos]);
            if (t->cuts)   // (the node's cut-round counters, in the order of the trace)
                for (int f = 0; f < 8; f++) t->tr_cuts.push_back(S.h_cs[4 + (size_t)f * MB + pos]);
        }
        return dive_child;
    };
    for (int k : todo) {
        int err = MIPX_OK;
        // (a chain reads ten result arrays at every level: more streams than the hardware prefetcher follows)
        if (!partial && (k & 7) == 0 && k + 40 < B) {
            for (int lv = 0; lv <= S.dive; lv++) {
                const int pp = lv * B + k + 32;
                __builtin_prefetch(&status[pp]); __builtin_prefetch(&obj[pp]); __builtin_prefetch(&mipf[pp]);
                __builtin_prefetch(&bidx[pp]); __builtin_prefetch(&bval[pp]); __builtin_prefetch(&nprobe[pp]);
                __builtin_prefetch(&npiv[pp]);
                if (lv < S.dive) { __builtin_prefetch(&dvar[pp]); __builtin_prefetch(&ddir[pp]); __builtin_prefetch(&dval[pp]); }
            }
        }
        int64_t cid = evaluate(ids[k], k, slots[k], 0, S.recs[k].depth, S.recs[k].anchor, err);
        if (err) { if (defer_push) t->bq.push_many(t->pend.data(), t->pend.size()); return err; }
        for (int level = 1; cid >= 0; level++) {   // the plunge: every dive child right after its parent
            const int32_t cslot = t->nodes[cid].slot;
            const int pos = level * B + k;
            t->lps++;
            t->dives++;
            t->pivots += npiv[pos];
            const int64_t next = evaluate(cid, pos, cslot, level, S.recs[k].depth + level, S.recs[k].anchor, err);
            if (err) { if (defer_push) t->bq.push_many(t->pend.data(), t->pend.size()); return err; }
            dive_slots.push_back(cslot);  // its record row feeds its own children below
            t->nodes[cid].slot = -1;
            cid = next;
        }
    }
    if (defer_push) t->bq.push_many(t->pend.data(), t->pend.size());
    if (incumbent_pos >= 0) {
        // the last improving node of the batch holds the incumbent
        if ((rc = tree_d2h(t, t->best_x.data(), S.d_x + (size_t)incumbent_pos * n, (size_t)n * 8))) return rc;
        t->have_x = true;
    }
    t->phase_ms[3] += ms_since(tp); tp = now();
    // anchor mode: the refactorisations of every later node start from the root's optimal tableau
    // (cut rounds: only a root that kept no cut rows has a basis of the shared rows alone)
    if (t->anchor_mode && !t->anchor_set && ids[0] == 0 && status[0] == 0 && (!partial || todo[0] == 0) &&
        !(t->cuts && S.h_cs[4 + 7 * MB] != 0)) {
        std::vector<int8_t> rootv(nv);
        HIP_TRY(ctx, hipMemcpy(rootv.data(), S.d_vout, (size_t)nv, hipMemcpyDeviceToHost));
        const int arc = mipx_problem_set_anchor(t->prob, rootv.data());
        if (arc) return arc;
        t->anchor_set = true;
    }
    // 5. children records on the device, then release the evaluated nodes' rows
    const int P = (int)S.br[0].pos.size();
    if (P > 0) {
        // when steps overlap this runs on its own stream, beside the node LPs of the step in flight
        // (it writes fresh pool rows only); the next launch waits for it
        hipStream_t cs = overlapped ? t->st3 : st;
        if (overlapped) HIP_TRY(ctx, hipStreamSynchronize(t->st3));  // h_pairs / d_pairs free again
        auto children = [&](int cnt, const std::vector<int32_t> &cslot, const std::vector<int32_t> &cpos,
                            const std::vector<int32_t> &cvar, const std::vector<int32_t> &cchild,
                            int32_t *hp, int32_t *dp) -> int {
            std::memcpy(hp, cslot.data(), (size_t)cnt * 4);
            std::memcpy(hp + cnt, cpos.data(), (size_t)cnt * 4);
            std::memcpy(hp + 2 * cnt, cvar.data(), (size_t)cnt * 4);
            std::memcpy(hp + 3 * cnt, cchild.data(), (size_t)cnt * 8);
            HIP_TRY(ctx, hipMemcpyAsync(dp, hp, (size_t)cnt * 20, hipMemcpyHostToDevice, cs));
            mipx::ChildArgs ca;
            ca.n = n; ca.m = t->m; ca.count = cnt;
            ca.src_l = t->pool_l; ca.src_u = t->pool_u;
            ca.parent_slot = dp; ca.parent_pos = dp + cnt; ca.var = dp + 2 * cnt;
            ca.x = S.d_x; ca.vstat = S.d_vout;
            ca.dst_l = t->pool_l; ca.dst_u = t->pool_u; ca.dst_v = t->pool_v;
            ca.child_slot = dp + 3 * cnt;
            if (t->cuts) {
                ca.mstride = t->mrows; ca.kc = t->kc;
                ca.src_ncut = S.w_ncut; ca.src_ids = S.w_ids; ca.dst_ncut = t->pool_ncut; ca.dst_ids = t->pool_ids;
            }
            hipLaunchKernelGGL(mipx::make_children, dim3(2 * cnt), dim3(256), 0, cs, ca);
            HIP_TRY(ctx, hipGetLastError());
            return MIPX_OK;
        };
        // level by level on one stream: the record of a dive child exists before its children are derived
        const size_t part = 5 * (size_t)t->max_batch;  // staging per level
        for (int level = 0; level < L; level++) {
            const auto &bl = S.br[(size_t)level];
            if (bl.pos.empty()) break;   // (no branching at this level: none below it either)
            if ((rc = children((int)bl.pos.size(), bl.slot, bl.pos, bl.var, bl.child, t->h_pairs + (size_t)level * part,
                               t->d_pairs + (size_t)level * part)))
                return rc;
        }
        if (overlapped) {
            HIP_TRY(ctx, hipEventRecord(t->ev_child, t->st3));
            t->child_pending = true;
            t->child_recorded = true;
        } else {
            HIP_TRY(ctx, hipStreamSynchronize(st));
        }
    }
    for (int32_t sl : dive_slots) t->free_slots.push_back(sl);
    for (int k : todo) t->free_slots.push_back(slots[k]);
    (void)nv;
    t->phase_ms[4] += ms_since(tp);
    return MIPX_OK;
}

// ---- the exchange between the ranks of one search (mipx_tree_set_comm) --------------------------------
constexpr int kRecHead = 16;         // doubles in front of a record's solution and pseudo-cost samples
constexpr int64_t kMaxMigrate = 4096;  // node records per donation

size_t x_rec_len(const mipx_tree *t) { return (size_t)kRecHead + 5 * (size_t)t->n; }

int64_t tree_open_count(const mipx_tree *t) { return (int64_t)(t->use_bq ? t->bq.size() : t->heap.size()); }

// this rank's state as one record: [primal, dual, open + in flight, stop flag, evaluated, lps, probes,
// pivots (since sharding), has_x, round, frontier batch, ...] [x (n)] [own pseudo-cost samples (4 n)]
void x_fill_record(mipx_tree *t) {
    const size_t n = (size_t)t->n;
    std::vector<double> &r = t->x_rec;
    r.assign(x_rec_len(t), 0.0);
    int64_t inflight = 0;
    // the nodes popped into the steps in flight are neither queued nor closed: their bounds (the lowest
    // of the shard under best first) count, or the shard would report a dual bound it has not proved
    double inflight_min = std::numeric_limits<double>::infinity();
    for (const StepBuf &S : t->buf)
        if (S.in_flight) { inflight += S.B; inflight_min = std::fmin(inflight_min, S.inflight_min); }
    r[0] = t->primal;
    r[1] = std::fmin(tree_dual_bound(t), inflight_min);
    r[2] = (double)(tree_open_count(t) + inflight);
    r[3] = (double)t->x_stop_flag;
    r[4] = (double)(t->evaluated - t->ramp[0]);
    r[5] = (double)(t->lps - t->ramp[1]);
    r[6] = (double)(t->probes - t->ramp[2]);
    r[7] = (double)(t->pivots - t->ramp[3]);
    r[8] = t->have_x ? 1.0 : 0.0;
    r[9] = (double)t->x_rounds;
    r[10] = (double)t->x_batch;
    // pool rows this rank can take from a donor: what is free beyond the children the steps in flight
    // (up to three) and the steps until the record is applied may still claim
    {
        const int64_t per = 2 * (1 + (int64_t)t->dive) * (int64_t)t->max_batch;
        const int64_t room = (int64_t)t->free_slots.size() - (3 + (int64_t)t->x_every) * per;
        r[11] = (double)(room > 0 ? room : 0);
    }
    r[12] = inflight_min;
    if (t->have_x) std::memcpy(r.data() + kRecHead, t->best_x.data(), n * 8);
    std::memcpy(r.data() + kRecHead + n, t->pc_own.data(), 4 * n * 8);
}

// A donation: `amount` node records from rank `from` to rank `to` (decided from the same records on
// every rank).  The donor gives every second of its best 2 * amount open nodes, so that both keep
// good ones; fewer if it has consumed them since it posted (the message has the planned size, a
// count in front says how many records are real).
struct MigLayout { size_t rowbytes, meta_off, bytes; };
MigLayout mig_layout(const mipx_tree *t, int64_t amount) {
    const size_t n = (size_t)t->n, nvs = n + (size_t)t->mrows;
    MigLayout L;
    L.rowbytes = 16 * n + (nvs + 7) / 8 * 8;
    L.meta_off = (size_t)amount * L.rowbytes;
    L.bytes = L.meta_off + (1 + 5 * (size_t)amount) * 8;
    return L;
}
mipx::PackArgs mig_args(mipx_tree *t, const MigLayout &L, char *msg, int32_t *d_slots) {
    mipx::PackArgs pa;
    pa.n = t->n; pa.nvs = t->n + t->mrows; pa.rowbytes = L.rowbytes; pa.slot = d_slots;
    pa.pool_l = t->pool_l; pa.pool_u = t->pool_u; pa.pool_v = t->pool_v; pa.msg = msg;
    pa.count = 0;
    return pa;
}

// donor half: up to `amount` records leave the queue and are packed into msg (device memory)
int mig_pack(mipx_tree *t, const MigLayout &L, char *msg, int32_t *d_slots, int64_t amount, std::vector<int32_t> &slots,
             int64_t *count) {
    mipx_ctx *ctx = t->ctx;
    std::vector<double> meta(1 + 5 * (size_t)amount, 0.0);
    mipx::PackArgs pa = mig_args(t, L, msg, d_slots);
    // the child records of the last finished step may still be in the making on st3: the donor must not
    // pack rows before they are written
    if (t->child_recorded) HIP_TRY(ctx, hipStreamWaitEvent(t->st2, t->ev_child, 0));
    const int64_t have = tree_open_count(t);
    const int64_t give = std::max<int64_t>(0, std::min<int64_t>(amount, (have - t->x_batch) / 2));
    std::vector<int64_t> ids;
    if (t->use_bq) {
        t->popped.clear();
        t->bq.pop_batch((size_t)(2 * give), t->popped);
        for (const auto &it : t->popped) ids.push_back(it.id);
    } else {
        for (int64_t k = 0; k < 2 * give && !t->heap.empty(); k++) ids.push_back(t->heap.pop());
    }
    int64_t cnt = 0;
    for (size_t k = 0; k < ids.size(); k++) {
        NodeRec &nd = t->nodes[ids[k]];
        if (t->search != 0) t->is_open[ids[k]] = 0;
        if ((k & 1) == 0 || cnt >= give) { tree_push(t, ids[k]); continue; }   // kept
        double *mrec = meta.data() + 1 + 5 * cnt;
        mrec[0] = nd.dual_bound; mrec[1] = nd.b_val; mrec[2] = (double)nd.depth;
        mrec[3] = (double)nd.b_idx; mrec[4] = (double)nd.b_dir;
        slots.push_back(nd.slot);
        nd.slot = -1;
        cnt++;
    }
    meta[0] = (double)cnt;
    if (cnt > 0) {
        HIP_TRY(ctx, hipMemcpyAsync(d_slots, slots.data(), (size_t)cnt * 4, hipMemcpyHostToDevice, t->st2));
        pa.count = (int)cnt;
        hipLaunchKernelGGL(mipx::pack_nodes, dim3((unsigned)cnt), dim3(256), 0, t->st2, pa);
        HIP_TRY(ctx, hipGetLastError());
    }
    HIP_TRY(ctx, hipMemcpyAsync(msg + L.meta_off, meta.data(), meta.size() * 8, hipMemcpyHostToDevice, t->st2));
    HIP_TRY(ctx, hipStreamSynchronize(t->st2));
    *count = cnt;
    return MIPX_OK;
}

// receiver half: the records of msg (device memory) get pool rows and join the queue
int mig_unpack(mipx_tree *t, const MigLayout &L, char *msg, int32_t *d_slots, int64_t amount, int64_t *count) {
    mipx_ctx *ctx = t->ctx;
    std::vector<double> meta(1 + 5 * (size_t)amount, 0.0);
    std::vector<int32_t> slots;
    mipx::PackArgs pa = mig_args(t, L, msg, d_slots);
    HIP_TRY(ctx, hipMemcpy(meta.data(), msg + L.meta_off, meta.size() * 8, hipMemcpyDeviceToHost));
    const int64_t cnt = (int64_t)meta[0];
    if (cnt < 0 || cnt > amount) return fail(ctx, MIPX_EHIP, "tree: corrupt migration message");
    // (the amount was capped by the room this rank reported in its record -- x_decide -- which left the
    // claims of the steps in flight aside; batch_size() keeps later steps within what is left)
    if ((int64_t)t->free_slots.size() < cnt)
        return fail(ctx, MIPX_ENOMEM, "tree: node pool too small for the migrated nodes (raise pool_capacity)");
    // the parents' rows freed by the last finished step may still be read by its make_children on st3
    if (t->child_recorded) HIP_TRY(ctx, hipStreamWaitEvent(t->st2, t->ev_child, 0));
    for (int64_t k = 0; k < cnt; k++) {
        const double *mrec = meta.data() + 1 + 5 * k;
        NodeRec nd;
        nd.dual_bound = mrec[0]; nd.b_val = mrec[1]; nd.depth = (int32_t)mrec[2];
        nd.b_idx = (int32_t)mrec[3]; nd.b_dir = (int32_t)mrec[4];
        nd.key = t->search == 0 ? nd.dual_bound : -(double)nd.depth;
        nd.anchor = -1; nd.born = (int32_t)t->steps; nd.ncut = 0;
        nd.slot = t->free_slots.back();
        t->free_slots.pop_back();
        slots.push_back(nd.slot);
        t->nodes.push_back(nd);
        tree_push(t, (int64_t)t->nodes.size() - 1);
    }
    if (cnt > 0) {
        HIP_TRY(ctx, hipMemcpyAsync(d_slots, slots.data(), (size_t)cnt * 4, hipMemcpyHostToDevice, t->st2));
        pa.count = (int)cnt;
        hipLaunchKernelGGL(mipx::unpack_nodes, dim3((unsigned)cnt), dim3(256), 0, t->st2, pa);
        HIP_TRY(ctx, hipGetLastError());
        HIP_TRY(ctx, hipStreamSynchronize(t->st2));
    }
    *count = cnt;
    return MIPX_OK;
}

int x_migrate(mipx_tree *t, int from, int to, int64_t amount) {
    mipx_comm *c = t->comm;
    const int me = c->rank;
    if (me != from && me != to) return MIPX_OK;
    const MigLayout L = mig_layout(t, amount);
    char *msg = nullptr;
    int rc = comm_msg_buffer(c, L.bytes + (size_t)amount * 4, &msg);   // (+ the slot list of the pack kernels)
    if (rc) return rc;
    int32_t *d_slots = (int32_t *)(msg + L.bytes);
    int64_t cnt = 0;
    if (me == from) {
        std::vector<int32_t> slots;
        if ((rc = mig_pack(t, L, msg, d_slots, amount, slots, &cnt))) return rc;
        if ((rc = comm_send_dev(c, to, msg, L.bytes))) return rc;
        for (int32_t sl : slots) t->free_slots.push_back(sl);
        t->nodes_sent += cnt;
    } else {
        if ((rc = comm_recv_dev(c, from, msg, L.bytes))) return rc;
        if ((rc = mig_unpack(t, L, msg, d_slots, amount, &cnt))) return rc;
        t->nodes_received += cnt;
    }
    return MIPX_OK;
}

// What every rank concludes from one gathered set of records -- a pure function of the records, so
// that all ranks conclude the same (exposed as mipx_exchange_decide for the CPU tests).
void x_decide(int W, int n, const double *records, double mip_gap, bool allow_migration,
              mipx_exchange_decision *out) {
    const size_t len = (size_t)kRecHead + 5 * (size_t)n;
    const double inf = std::numeric_limits<double>::infinity();
    auto rec = [&](int r) { return records + (size_t)r * len; };
    // incumbent: the best value of any rank, and the lowest rank that holds a solution for it
    double best = inf, dual = inf;
    int who = -1;
    int64_t sums[5] = {0, 0, 0, 0, 0};
    // stop flags: 1 = a limit that ends the search for everybody (node_limit, max_seconds, unbounded,
    // a full pool); 2 = this rank has done the steps it was asked for (max_steps is a per-rank quota:
    // the others finish theirs)
    bool any_stop = false, all_finished = true, any_fatal = false;
    for (int r = 0; r < W; r++) {
        const double *q = rec(r);
        if (q[0] < best) best = q[0];
        dual = std::fmin(dual, q[1]);
        sums[0] += (int64_t)q[4]; sums[1] += (int64_t)q[5]; sums[2] += (int64_t)q[6]; sums[3] += (int64_t)q[7];
        sums[4] += (int64_t)q[2];
        any_stop |= q[3] == 1.0;
        any_fatal |= q[3] == 3.0;
        all_finished &= q[3] == 2.0 || q[2] == 0.0;
    }
    for (int r = 0; r < W && who < 0; r++)
        if (rec(r)[0] == best && rec(r)[8] != 0.0 && best < inf) who = r;
    double gap = -1.0;   // reference current_gap (branch_and_bound.py:203-213); -1 encodes None
    if (best < inf) {
        if (best == 0 && dual == 0) gap = 0.0;
        else if (best == 0) gap = inf;
        else gap = std::fabs(best - dual) / std::fabs(best);
    }
    out->primal = best; out->dual = dual; out->gap = gap; out->incumbent_rank = who;
    for (int k = 0; k < 4; k++) out->sums[k] = sums[k];
    out->open_nodes = sums[4];
    out->reason = any_fatal ? 5 : sums[4] == 0 ? 1 : any_stop ? 2 : (gap >= 0 && gap <= mip_gap) ? 3 : all_finished ? 4 : 0;
    out->done = out->reason != 0;
    out->n_moves = 0;
    // migration: a rank that cannot fill a batch gets half the surplus of the fullest rank
    if (!out->done && allow_migration && W > 1) {
        std::vector<int64_t> open((size_t)W), low((size_t)W), room((size_t)W);
        for (int r = 0; r < W; r++) {
            open[(size_t)r] = (int64_t)rec(r)[2]; low[(size_t)r] = std::max<int64_t>(1, (int64_t)rec(r)[10]);
            room[(size_t)r] = (int64_t)rec(r)[11];
        }
        for (int d = 0; d < W && out->n_moves < 64; d++) {
            if (open[(size_t)d] >= low[(size_t)d]) continue;
            int src = 0;
            for (int r = 1; r < W; r++)
                if (open[(size_t)r] > open[(size_t)src]) src = r;
            if (src == d || open[(size_t)src] < 2 * low[(size_t)src]) continue;
            // (never more than the receiver said it has room for: it cannot refuse once the donor has sent)
            const int64_t amount = std::min<int64_t>(std::min<int64_t>((open[(size_t)src] - open[(size_t)d]) / 2, kMaxMigrate),
                                                     room[(size_t)d]);
            if (amount <= 0) continue;
            room[(size_t)d] -= amount;
            int32_t *mv = out->moves + 3 * out->n_moves++;
            mv[0] = src; mv[1] = d; mv[2] = (int32_t)amount;
            open[(size_t)src] -= amount;
            open[(size_t)d] += amount;
        }
    }
}

// Apply one gathered set of records (identical on every rank).  last: the closing exchange -- merge
// only, no decisions.
int x_apply(mipx_tree *t, const char *gathered, bool last) {
    const mipx_comm *c = t->comm;
    const int W = c->world, me = c->rank;
    const size_t n = (size_t)t->n, len = x_rec_len(t);
    const double inf = std::numeric_limits<double>::infinity();
    auto rec = [&](int r) { return (const double *)(gathered + (size_t)r * len * 8); };
    mipx_exchange_decision D;
    x_decide(W, t->n, (const double *)gathered, t->x_mip_gap, !t->cuts, &D);
    const double best = D.primal;
    const int who = D.incumbent_rank;
    if (best < t->primal || (best == t->primal && best < inf && !t->have_x && who >= 0)) {
        t->primal = best;
        if (who >= 0) {
            std::memcpy(t->best_x.data(), rec(who) + kRecHead, n * 8);
            t->have_x = true;
        }
    }
    if (t->primal == best) t->g_inc_rank = (t->primal < inf && who >= 0) ? who : (t->have_x && t->primal < inf ? me : -1);
    else t->g_inc_rank = t->have_x ? me : -1;   // (this rank found a better one since it posted)
    t->g_dual = D.dual;
    for (int k = 0; k < 4; k++) t->g_counts[k] = t->ramp[k] + D.sums[k];
    t->g_counts[4] = D.open_nodes;
    // pseudo costs: what all ranks agreed on at sharding time + every other rank's samples as of its
    // record + this rank's own samples as of now
    if (t->rule == 1) {
        std::fill(t->pc_others.begin(), t->pc_others.end(), 0.0);
        for (int r = 0; r < W; r++) {
            if (r == me) continue;
            const double *o = rec(r) + kRecHead + n;
            for (size_t j = 0; j < 4 * n; j++) t->pc_others[j] += o[j];
        }
    }
    if (t->rule == 1 && t->fast_ok) {
        // Device finish: the table lives on the device and holds this rank's samples up to the steps in flight.
        // What the OTHER ranks sampled since the last exchange joins it in sum form (pc_merge, queued on stf
        // behind the finishes already there); the host's snapshot is merged the same way.
        if (t->pc_others_prev.size() != 4 * n) t->pc_others_prev.assign(4 * n, 0.0);
        double *hd = t->h_delta + (size_t)(t->delta_turn & 3) * 4 * n;
        bool any = false;
        for (size_t j = 0; j < 4 * n; j++) {
            hd[j] = t->pc_others[j] - t->pc_others_prev[j];
            any |= hd[j] != 0.0;
        }
        t->pc_others_prev = t->pc_others;
        if (any) {
            for (size_t e = 0; e < 2 * n; e++) {
                const size_t dir = e / n, var = e - dir * n;
                const double ds = hd[dir * n + var], dt = hd[(2 + dir) * n + var];
                if (!(dt > 0.0)) continue;
                double &cost = dir ? t->cost_r[var] : t->cost_l[var];
                int32_t &times = dir ? t->times_r[var] : t->times_l[var];
                cost = (cost * (double)times + ds) / ((double)times + dt);
                times += (int32_t)dt;
                t->has_entry[var] = 1;
            }
            double *dd = t->d_delta + (size_t)(t->delta_turn & 3) * 4 * n;
            t->delta_turn++;
            mipx_ctx *ctx = t->ctx;
            HIP_TRY(ctx, hipMemcpyAsync(dd, hd, 4 * n * 8, hipMemcpyHostToDevice, t->stf));
            const TabPtr tb = tab_at(t, t->tab_tail);
            mipx::PcMergeArgs ma;
            ma.n = t->n; ma.delta = dd; ma.cost_l = tb.cl; ma.cost_r = tb.cr; ma.has = tb.has; ma.times = tb.times;
            hipLaunchKernelGGL(mipx::pc_merge, dim3((2 * t->n + 255) / 256), dim3(256), 0, t->stf, ma);
            HIP_TRY(ctx, hipGetLastError());
            HIP_TRY(ctx, hipEventRecord(t->ev_tab[t->tab_tail], t->stf));
            t->tab_late[t->tab_tail] = true;
        }
    } else if (t->rule == 1) {
        for (size_t j = 0; j < n; j++) {
            const double tl = t->pc_base[2 * n + j] + t->pc_others[2 * n + j] + t->pc_own[2 * n + j];
            const double tr = t->pc_base[3 * n + j] + t->pc_others[3 * n + j] + t->pc_own[3 * n + j];
            const double sl = t->pc_base[j] + t->pc_others[j] + t->pc_own[j];
            const double sr = t->pc_base[n + j] + t->pc_others[n + j] + t->pc_own[n + j];
            t->times_l[j] = (int32_t)tl; t->times_r[j] = (int32_t)tr;
            t->cost_l[j] = tl > 0 ? sl / tl : 0.0;
            t->cost_r[j] = tr > 0 ? sr / tr : 0.0;
            t->has_entry[j] = (tl > 0 || tr > 0) ? 1 : 0;
        }
        t->table_dirty = true;
        t->times_dirty = true;
    }
    t->x_rounds++;
    if (last) return MIPX_OK;
    if (D.done) {   // every rank idle, a rank's limit, the global gap, or a rank that failed: the same on every rank
        t->x_done = true;
        t->x_fatal = D.reason == 5;
        return MIPX_OK;
    }
    for (int k = 0; k < D.n_moves; k++) {
        const int rc = x_migrate(t, D.moves[3 * k], D.moves[3 * k + 1], D.moves[3 * k + 2]);
        if (rc) return rc;
    }
    return MIPX_OK;
}

// One call at an exchange point.  Collects the all-gather posted last time (by then long finished on
// a busy rank) and posts the next; a rank with nothing to do (blocking) waits for that one as well.
int x_tick(mipx_tree *t, bool blocking) {
    mipx_comm *c = t->comm;
    const char *g = nullptr;
    int rc;
    if (c->pending) {
        if ((rc = comm_collect(c, &g))) return rc;
        if ((rc = x_apply(t, g, false))) return rc;
        if (t->x_done) return MIPX_OK;
    }
    x_fill_record(t);
    if ((rc = comm_post(c, t->x_rec.data(), x_rec_len(t) * 8))) return rc;
    if (blocking) {
        if ((rc = comm_collect(c, &g))) return rc;
        if ((rc = x_apply(t, g, false))) return rc;
    }
    return MIPX_OK;
}

// The closing exchange after the ranks agreed to stop: what each found since its last record.
int x_close(mipx_tree *t) {
    mipx_comm *c = t->comm;
    const char *g = nullptr;
    int rc;
    x_fill_record(t);
    if ((rc = comm_post(c, t->x_rec.data(), x_rec_len(t) * 8))) return rc;
    if ((rc = comm_collect(c, &g))) return rc;
    return x_apply(t, g, true);
}

// This rank is about to return an error from the step loop: tell the others, or they would wait for it
// in the next all-gather for ever.  The sequence of all-gathers stays aligned: what was posted is
// collected and applied, a record with the fatal flag follows (every rank that applies it stops, reason
// 5), then the closing exchange the others run.  Best effort: an error in here is dropped (the rank is
// failing already); a failure inside a migration's send / recv cannot be announced this way.
void x_fail(mipx_tree *t) {
    mipx_comm *c = t->comm;
    const char *g = nullptr;
    t->x_stop_flag = 3;
    if (c->pending) {
        if (comm_collect(c, &g)) return;
        if (x_apply(t, g, false)) return;
    }
    if (!t->x_done) {
        x_fill_record(t);
        if (comm_post(c, t->x_rec.data(), x_rec_len(t) * 8)) return;
        if (comm_collect(c, &g)) return;
        if (x_apply(t, g, false)) return;
    }
    if (t->x_done) (void)x_close(t);
}

}  // namespace

extern "C" {

int mipx_tree_create(mipx_problem *p, const int32_t *int_idx, int n_int, const double *l,
                     const double *u, int branch_rule, int search_rule, int strong_branch_iters,
                     int max_batch, int64_t pool_capacity, mipx_tree **out) {
    return mipx_tree_create_ex(p, int_idx, n_int, l, u, branch_rule, search_rule, strong_branch_iters,
                               max_batch, pool_capacity, nullptr, out);
}

int mipx_tree_create_ex(mipx_problem *p, const int32_t *int_idx, int n_int, const double *l,
                        const double *u, int branch_rule, int search_rule, int strong_branch_iters,
                        int max_batch, int64_t pool_capacity, const mipx_cut_params *cuts,
                        mipx_tree **out) {
    if (!p || !out || n_int < 0 || (n_int && !int_idx) || !l || !u || max_batch < 1 ||
        branch_rule < 0 || branch_rule > 1 || search_rule < 0 || search_rule > 1 ||
        strong_branch_iters < 1)
        return fail(p ? p->ctx : nullptr, MIPX_EINVAL, "mipx_tree_create: bad argument");
    mipx_ctx *ctx = p->ctx;
    *out = nullptr;
    int kc = 0;
    if (cuts != nullptr) {
        kc = cuts->max_cuts_per_node > 0 ? cuts->max_cuts_per_node : mipx::kMaxNodeCuts;
        if (kc > mipx::kMaxNodeCuts || cuts->max_cut_generation_iterations < 1 || cuts->max_nonzero_coefs < 1 ||
            !(cuts->cutting_plane_progress_tolerance > 0) || !(cuts->min_cut_depth > 0) || !(cuts->max_term > 0) ||
            cuts->store_capacity < 0)
            return fail(ctx, MIPX_EINVAL, "mipx_tree_create_ex: bad cut parameter");
        if (pick_cfg(p->m, p->n) != nullptr) {
            // register tiles: the rows of a node must fit one; keep as many cut rows as the largest tile takes
            // (all launches of the tree then price alike: the tiles' steepest edge)
            while (kc > 0 && pick_cfg(p->m + kc, p->n) == nullptr) kc--;
        } else {
            // above them every launch streams its tableau (K1b): the slabs take the cut rows too
            while (kc > 0 && !big_fits(p->m + kc, p->n)) kc--;
        }
        if (kc == 0)
            return fail(ctx, MIPX_ETOOBIG, "mipx_tree_create_ex: no room for a cut row (m + cuts <= 192 rows on the register "
                                            "tiles, <= 1024 above them)");
    }
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    mipx_tree *t = new (std::nothrow) mipx_tree();
    if (!t) return fail(ctx, MIPX_ENOMEM, "mipx_tree_create: host alloc");
    t->prob = p; t->ctx = ctx; t->n = p->n; t->m = p->m; t->n_int = n_int;
    t->rule = branch_rule; t->search = search_rule; t->sb_iters = strong_branch_iters;
    t->max_batch = max_batch;
    t->use_bq = max_batch > 1 && search_rule == 0;
    t->capacity = pool_capacity > 2 * (int64_t)max_batch + 2 ? pool_capacity : 2 * (int64_t)max_batch + 2;
    t->int_idx.assign(int_idx, int_idx + n_int);
    for (int i = 0; i < n_int; i++)
        if (int_idx[i] < 0 || int_idx[i] >= t->n) { delete t; return fail(ctx, MIPX_EINVAL, "mipx_tree_create: integer index out of range"); }
    t->cuts = cuts != nullptr;
    t->kc = kc;
    t->mrows = t->m + kc;
    if (t->cuts) {
        t->cp = *cuts;
        t->store_cap = cuts->store_capacity > 0 ? cuts->store_capacity : ((int64_t)1 << 20);
        if (t->store_cap > 0x7fffffff) t->store_cap = 0x7fffffff;
        // candidate cuts of one node while it is bounded: every round adds at most one per row; K3
        // keeps three doubles and two ints per candidate in LDS (64 KiB)
        int64_t rows = (int64_t)cuts->max_cut_generation_iterations * t->mrows;
        if (rows > 2040) rows = 2040;
        if (rows < t->mrows) rows = t->mrows;
        t->slab_rows = (int)rows;
        t->pipeline = false;   // (the cut loop of a step reads back per round: steps do not overlap)
    }
    const size_t n = t->n, nv = (size_t)t->n + t->mrows, cap = (size_t)t->capacity, B = (size_t)max_batch;
    t->probe_cap = 2 * (int64_t)B * (n_int ? n_int : 1);
    if (t->probe_cap > (int64_t)1 << 18) t->probe_cap = (int64_t)1 << 18;  // 2^18 probe records (~1.2 GB at 256x128)
    const size_t pc = (size_t)t->probe_cap;
    int rc = 0;
    rc |= dmalloc(ctx, &t->pool_l, cap * n); rc |= dmalloc(ctx, &t->pool_u, cap * n);
    rc |= dmalloc(ctx, &t->pool_v, cap * nv);
    rc |= dmalloc(ctx, &t->d_int_idx, (size_t)n_int);
    const size_t LC = (size_t)kMaxDive + 1;   // output levels the buffers are sized for
    rc |= dmalloc(ctx, &t->d_pairs, 5 * (pc / 2 > LC * B ? pc / 2 : LC * B));   // (one part per dive level)
    rc |= dmalloc(ctx, &t->d_pairs2, 5 * (pc / 2 > LC * B ? pc / 2 : LC * B));
    rc |= dmalloc(ctx, &t->d_cost_l2, n); rc |= dmalloc(ctx, &t->d_cost_r2, n); rc |= dmalloc(ctx, &t->d_has2, n);
    // The side streams carry short, latency-critical work (probes, re-scoring, child records) that
    // must overtake the 2 ms node-LP launch queued on the main stream.  HIP multiplexes the streams
    // of one priority onto a few hardware queues, so in a process that owns more streams (a framework +
    // RCCL in a multi-GPU rank) a normal-priority side stream can land behind the main stream's
    // queue; high-priority streams come from their own queue pool.
    int prio_least = 0, prio_greatest = 0;
    if (hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest) != hipSuccess) prio_greatest = 0;
    if (hipStreamCreateWithPriority(&t->st2, hipStreamNonBlocking, prio_greatest) != hipSuccess) rc |= MIPX_EHIP;
    if (hipStreamCreateWithPriority(&t->stf, hipStreamNonBlocking, prio_greatest) != hipSuccess) rc |= MIPX_EHIP;
    if (hipStreamCreateWithPriority(&t->st3, hipStreamNonBlocking, prio_greatest) != hipSuccess ||
        hipEventCreateWithFlags(&t->ev_child, hipEventDisableTiming) != hipSuccess ||
        hipHostMalloc((void **)&t->h_pairs, 5 * ((size_t)kMaxDive + 1) * B * 4, hipHostMallocDefault) != hipSuccess) rc |= MIPX_EHIP;
    if (t->cuts) {
        const size_t M = (size_t)t->mrows, K = (size_t)t->kc, SR = (size_t)t->slab_rows;
        rc |= dmalloc(ctx, &t->store_pi, (size_t)t->store_cap * n); rc |= dmalloc(ctx, &t->store_pi0, (size_t)t->store_cap);
        rc |= dmalloc(ctx, &t->store_count, 4);
        rc |= dmalloc(ctx, &t->pool_ncut, cap); rc |= dmalloc(ctx, &t->pool_ids, cap * K);
        rc |= dmalloc(ctx, &t->pp_ncut, pc); rc |= dmalloc(ctx, &t->pp_ids, pc * K);
        rc |= dmalloc(ctx, &t->d_is_int, n);
        for (StepBuf &S : t->buf) {
            rc |= dmalloc(ctx, &S.w_ncut, B); rc |= dmalloc(ctx, &S.w_ids, B * K);
            rc |= dmalloc(ctx, &S.cs_state, (size_t)mipx::CF_FIELDS * B);
            rc |= dmalloc(ctx, &S.cs_active, B); rc |= dmalloc(ctx, &S.cs_resolve, B);
            rc |= dmalloc(ctx, &S.cs_need_tab, B);
            rc |= dmalloc(ctx, &S.cs_counters, 4); rc |= dmalloc(ctx, &S.k2_ncuts, B);
            rc |= dmalloc(ctx, &S.k3_nadded, B); rc |= dmalloc(ctx, &S.k3_added, B * SR);
            rc |= dmalloc(ctx, &S.k3_term, B); rc |= dmalloc(ctx, &S.pool_list, B * SR);
            rc |= dmalloc(ctx, &S.d_y, B * M); rc |= dmalloc(ctx, &S.cs_before, B);
            rc |= dmalloc(ctx, &S.slab_pi, B * SR * n); rc |= dmalloc(ctx, &S.slab_pi0, B * SR);
            rc |= dmalloc(ctx, &S.dump_T, B * M * n); rc |= dmalloc(ctx, &S.dump_vec, B * (n + 3 * M));
            rc |= dmalloc(ctx, &S.dump_idx, B * (2 * n + M));
            if (hipHostMalloc((void **)&S.h_cs, (4 + 9 * B) * 4, hipHostMallocDefault) != hipSuccess) rc |= MIPX_EHIP;
            if (!t->pipeline) break;   // steps do not overlap: one buffer set is in use
        }
    }
    for (StepBuf &S : t->buf) {
        rc |= dmalloc(ctx, &S.d_slot, 2 * B);   // [pool rows | anchor-table entries] of the batch
        // per-node outputs have one row per output level: the batch, then its dive children level by
        // level (allocated for the deepest plunge, laid out for the depth in use: layout_pack)
        rc |= dmalloc(ctx, &S.d_iters, LC * B);
        const size_t pack_cap = (LC * B * (2 * 8 + 5 * 4) + LC * B * (8 + 2 * 4) + 15) / 16 * 16 + 16 +
                                (size_t)kAskCap * sizeof(mipx::ScoreArgs::Ask);
        rc |= dmalloc(ctx, &S.d_pack, pack_cap);
        if (hipHostMalloc((void **)&S.h_pack, pack_cap, hipHostMallocDefault) != hipSuccess ||
            hipHostMalloc((void **)&S.h_slot, 2 * B * 4, hipHostMallocDefault) != hipSuccess) rc |= MIPX_EHIP;
        if (S.d_pack) layout_pack(t, S, 1);
        rc |= dmalloc(ctx, &S.d_plist, LC * B * (size_t)(n_int ? n_int : 1));
        rc |= dmalloc(ctx, &S.d_x, LC * B * n);
        rc |= dmalloc(ctx, &S.d_vout, LC * B * nv);
        if (hipEventCreate(&S.e0) != hipSuccess || hipEventCreate(&S.e1) != hipSuccess ||
            hipEventCreate(&S.done) != hipSuccess) rc |= MIPX_EHIP;
        if (t->cuts && (hipEventCreate(&S.k2a) != hipSuccess || hipEventCreate(&S.k2b) != hipSuccess ||
                        hipEventCreate(&S.k3b) != hipSuccess)) rc |= MIPX_EHIP;
    }
    // Steps are finished on the device (finish_kernels.hip.h) in the batched modes without cut rounds;
    // MIPX_HOST_FINISH=1 keeps the host loop (A/B runs).  The exact mode (max_batch = 1) reproduces the
    // reference's node order on the host.
    t->fast_ok = max_batch > 1 && !t->cuts && !(std::getenv("MIPX_HOST_FINISH") && std::atoi(std::getenv("MIPX_HOST_FINISH")) != 0);
    {   // [cost_l | cost_r | has_entry] in one allocation: one upload per step; behind them (device finish)
        // [times_l | times_r | own sums], and kTabV versions of the whole block
        char *tab = nullptr;
        t->tab_off2 = (17 * n + 7) / 8 * 8;
        t->tab_bytes = t->tab_off2 + 8 * n + 32 * n;
        t->tab_stride = (t->tab_bytes + 255) / 256 * 256;
        rc |= dmalloc(ctx, &tab, t->tab_stride * (t->fast_ok ? (size_t)kTabV : 1));
        t->d_cost_l = (double *)tab;
        t->d_cost_r = tab ? (double *)(tab + 8 * n) : nullptr;
        t->d_has = tab ? (uint8_t *)(tab + 16 * n) : nullptr;
        if (hipHostMalloc((void **)&t->h_tab, 4 * 17 * n + 4 * 8 * n, hipHostMallocDefault) != hipSuccess) rc |= MIPX_EHIP;
        rc |= dmalloc(ctx, &t->d_primal, 1);
    }
    if (t->fast_ok) {
        const size_t per = 2 * LC;
        rc |= dmalloc(ctx, &t->d_delta, 4 * 4 * n);
        if (hipHostMalloc((void **)&t->h_delta, 4 * 4 * n * 8, hipHostMallocDefault) != hipSuccess) rc |= MIPX_EHIP;
        static_assert(kTabV == 8, "ev_tab / tab_late are sized for the version ring");
        for (int v = 0; v < kTabV; v++)
            if (hipEventCreateWithFlags(&t->ev_tab[v], hipEventDisableTiming) != hipSuccess) rc |= MIPX_EHIP;
        t->samples_cap = (size_t)t->probe_cap + LC * B;
        for (StepBuf &S : t->buf) {
            const size_t par_bytes = 2 * B * 8 + (4 * B + per * B) * 4;
            rc |= dmalloc(ctx, &S.d_par, par_bytes);
            rc |= dmalloc(ctx, &S.c_info, B); rc |= dmalloc(ctx, &S.c_cnt, 3 * B); rc |= dmalloc(ctx, &S.c_eval, 2 * B);
            rc |= dmalloc(ctx, &S.c_flag, B); rc |= dmalloc(ctx, &S.c_val, 3 * B);
            rc |= dmalloc(ctx, &S.d_sum, 1);
            rc |= dmalloc(ctx, &S.d_open, per * B); rc |= dmalloc(ctx, &S.d_dead, per * B);
            rc |= dmalloc(ctx, &S.d_samples, t->samples_cap);
            rc |= dmalloc(ctx, &S.d_skeys, t->samples_cap);
            const size_t fin_bytes = 128 + (t->tab_bytes + 31) / 32 * 32 + per * B * (sizeof(mipx::OpenEntry) + 4);
            if (hipHostMalloc((void **)&S.h_par, par_bytes, hipHostMallocDefault) != hipSuccess ||
                hipHostMalloc((void **)&S.h_fin, fin_bytes, hipHostMallocDefault) != hipSuccess ||
                hipHostMalloc((void **)&S.h_samples, t->samples_cap * (sizeof(mipx::PcSample) + 4), hipHostMallocDefault) != hipSuccess)
                rc |= MIPX_EHIP;
        }
    }
    if (branch_rule == 1) {
        rc |= dmalloc(ctx, &t->pp_l, pc * n); rc |= dmalloc(ctx, &t->pp_u, pc * n);
        rc |= dmalloc(ctx, &t->pp_v, pc * nv); rc |= dmalloc(ctx, &t->pp_obj, pc);
        rc |= dmalloc(ctx, &t->pp_status, pc);
        if (hipHostMalloc((void **)&t->h_pres, pc * 12, hipHostMallocDefault) != hipSuccess) rc |= MIPX_EHIP;
    }
    if (rc) { mipx_tree_destroy(t); return MIPX_EHIP; }
    t->cost_l.assign(n, 0.0); t->cost_r.assign(n, 0.0);
    t->times_l.assign(n, 0); t->times_r.assign(n, 0);
    t->has_entry.assign(n, 0);
    t->best_x.assign(n, 0.0);
    HIP_TRY(ctx, hipMemcpy(t->d_int_idx, int_idx, (size_t)n_int * 4, hipMemcpyHostToDevice));
    HIP_TRY(ctx, hipMemset(t->d_cost_l, 0, t->tab_stride * (t->fast_ok ? (size_t)kTabV : 1)));   // costs, entries, times, own sums
    {
        const double inf_ = std::numeric_limits<double>::infinity();
        HIP_TRY(ctx, hipMemcpy(t->d_primal, &inf_, 8, hipMemcpyHostToDevice));
    }
    // root record in slot 0: cold start (all status codes 0 -> slack basis, sides from d_j)
    HIP_TRY(ctx, hipMemcpy(t->pool_l, l, n * 8, hipMemcpyHostToDevice));
    HIP_TRY(ctx, hipMemcpy(t->pool_u, u, n * 8, hipMemcpyHostToDevice));
    HIP_TRY(ctx, hipMemset(t->pool_v, 0, nv));
    if (t->cuts) {
        std::vector<uint8_t> is_int(n, 0);
        for (int i = 0; i < n_int; i++) is_int[(size_t)int_idx[i]] = 1;
        HIP_TRY(ctx, hipMemcpy(t->d_is_int, is_int.data(), n, hipMemcpyHostToDevice));
        HIP_TRY(ctx, hipMemset(t->store_count, 0, 4));
        HIP_TRY(ctx, hipMemset(t->pool_ncut, 0, 4));   // the root carries no cut
    }
    t->free_slots.reserve(cap);
    for (int64_t s = (int64_t)cap - 1; s >= 1; s--) t->free_slots.push_back((int32_t)s);
    NodeRec root;
    root.dual_bound = -std::numeric_limits<double>::infinity();
    root.depth = 0; root.key = search_rule == 0 ? root.dual_bound : 0.0;
    root.b_idx = -1; root.b_dir = 0; root.b_val = 0.0; root.slot = 0; root.born = 0; root.anchor = -1;
    root.ncut = 0;
    t->nodes.push_back(root);
    *out = t;
    return MIPX_OK;
}

void mipx_tree_destroy(mipx_tree *t) {
    if (!t) return;
    if (std::getenv("MIPX_TREE_PROFILE"))
        std::fprintf(stderr, "[mipx_tree] steps %lld  pop %.1f ms  lp+score+d2h %.1f ms  pseudo-cost %.1f ms  "
                     "bookkeeping %.1f ms  children %.1f ms  (lp kernel %.1f ms)\n", (long long)t->steps,
                     t->phase_ms[0], t->phase_ms[1], t->phase_ms[2], t->phase_ms[3], t->phase_ms[4], t->kernel_ms);
    if (t->ctx) (void)hipSetDevice(t->ctx->device);
    if (t->ctx && t->ctx->stream) (void)hipStreamSynchronize(t->ctx->stream);
    if (t->st2) { (void)hipStreamSynchronize(t->st2); (void)hipStreamDestroy(t->st2); }
    if (t->st3) { (void)hipStreamSynchronize(t->st3); (void)hipStreamDestroy(t->st3); }
    if (t->stf) { (void)hipStreamSynchronize(t->stf); (void)hipStreamDestroy(t->stf); }
    if (t->ev_child) (void)hipEventDestroy(t->ev_child);
    if (t->h_pairs) (void)hipHostFree(t->h_pairs);
    if (t->h_pres) (void)hipHostFree(t->h_pres);
    if (t->h_tab) (void)hipHostFree(t->h_tab);
    void *cptrs[] = {t->store_pi, t->store_pi0, t->store_count, t->pool_ncut, t->pool_ids, t->pp_ncut, t->pp_ids, t->d_is_int};
    for (void *q : cptrs)
        if (q) (void)hipFree(q);
    for (StepBuf &S : t->buf) {
        void *sp[] = {S.w_ncut, S.w_ids, S.cs_state, S.cs_active, S.cs_resolve, S.cs_counters, S.k2_ncuts, S.k3_nadded,
                      S.k3_added, S.k3_term, S.pool_list, S.d_y, S.cs_before, S.slab_pi, S.slab_pi0, S.dump_T, S.dump_vec,
                      S.dump_idx, S.cs_need_tab};
        for (void *q : sp)
            if (q) (void)hipFree(q);
        if (S.h_cs) (void)hipHostFree(S.h_cs);
    }
    void *ptrs[] = {t->atab_T, t->atab_vec, t->atab_idx, t->pool_l, t->pool_u, t->pool_v, t->d_int_idx, t->d_pairs, t->d_pairs2, t->d_cost_l,
                    t->d_cost_l2, t->d_cost_r2, t->d_has2, t->pp_l, t->pp_u, t->pp_v, t->pp_obj, t->pp_status};
    for (void *q : ptrs)
        if (q) (void)hipFree(q);
    if (t->d_primal) (void)hipFree(t->d_primal);
    if (t->d_delta) (void)hipFree(t->d_delta);
    if (t->h_delta) (void)hipHostFree(t->h_delta);
    for (hipEvent_t e : t->ev_tab)
        if (e) (void)hipEventDestroy(e);
    for (StepBuf &S : t->buf) {
        void *fp[] = {S.d_par, S.c_info, S.c_cnt, S.c_eval, S.c_flag, S.c_val, S.d_sum, S.d_open, S.d_dead, S.d_samples, S.d_skeys};
        for (void *q : fp)
            if (q) (void)hipFree(q);
        if (S.h_par) (void)hipHostFree(S.h_par);
        if (S.h_fin) (void)hipHostFree(S.h_fin);
        if (S.h_samples) (void)hipHostFree(S.h_samples);
    }
    for (StepBuf &S : t->buf) {
        void *sp[] = {S.d_slot, S.d_iters, S.d_pack, S.d_plist, S.d_x, S.d_vout};
        for (void *q : sp)
            if (q) (void)hipFree(q);
        if (S.h_pack) (void)hipHostFree(S.h_pack);
        if (S.h_slot) (void)hipHostFree(S.h_slot);
        if (S.e0) (void)hipEventDestroy(S.e0);
        if (S.e1) (void)hipEventDestroy(S.e1);
        if (S.done) (void)hipEventDestroy(S.done);
        if (S.k2a) (void)hipEventDestroy(S.k2a);
        if (S.k2b) (void)hipEventDestroy(S.k2b);
        if (S.k3b) (void)hipEventDestroy(S.k3b);
    }
    delete t;
}

int mipx_tree_set_anchor_mode(mipx_tree *t, int on) {
    if (!t) return MIPX_EINVAL;
    t->anchor_mode = on != 0;
    return MIPX_OK;
}

int mipx_tree_set_trace(mipx_tree *t, int on) {
    if (!t) return MIPX_EINVAL;
    t->trace = on != 0;
    return MIPX_OK;
}

int mipx_tree_set_primal_bound(mipx_tree *t, double bound) {
    if (!t) return MIPX_EINVAL;
    t->primal = bound;
    return MIPX_OK;
}

/*
 * Run (or continue) the search; mirrors BranchAndBound.solve (branch_and_bound.py:215-241).
 * node_limit <= 0 and max_seconds <= 0 and max_steps <= 0 mean "no limit".
 */
int mipx_tree_solve(mipx_tree *t, int64_t node_limit, double mip_gap, double max_seconds,
                    int frontier_batch, int64_t max_steps, mipx_tree_stats *out) {
    if (!t || frontier_batch < 1 || frontier_batch > t->max_batch)
        return fail(t ? t->ctx : nullptr, MIPX_EINVAL, "mipx_tree_solve: bad argument");
    mipx_ctx *ctx = t->ctx;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const auto t0 = std::chrono::steady_clock::now();
    const double inf = std::numeric_limits<double>::infinity();
    double ph0[8], pr0[4];
    for (int k = 0; k < 8; k++) ph0[k] = t->phase_ms[k];
    for (int k = 0; k < 4; k++) pr0[k] = t->probe_ms[k];
    const double k0 = t->kernel_ms;
    if (!t->started) {
        t->started = true;
        tree_push(t, 0);
    }
    int64_t steps = 0, hooked_at = 0, xchg_at = 0;
    bool hook_stop = false;
    if (t->comm) {
        t->x_mip_gap = mip_gap; t->x_batch = frontier_batch; t->x_stop_flag = 0; t->x_done = false; t->x_fatal = false;
    }
    t->fault_step = std::getenv("MIPX_FAULT_STEP") ? std::atoi(std::getenv("MIPX_FAULT_STEP")) : 0;
    // an error of this rank's own step loop: the peers are told before it returns (x_fail)
    auto bail = [&](int rc) {
        if (t->comm && !t->x_done) x_fail(t);
        return rc;
    };
    // With frontier batches > 1 the host half of step k (bookkeeping, children) overlaps the GPU
    // halves of steps k+1 and k+2, whose batches are popped before the children of step k exist: a ring
    // of three step buffers.  Two steps queued ahead rather than one absorb a slow host half (the host
    // half is about as long as the node-LP kernel; with one step ahead every hiccup of the host idled
    // the GPU).
    const bool overlap = t->pipeline && frontier_batch > 1;
    const int NBUF = overlap ? 3 : 1;
    int head = 0, next = 0, nfl = 0;   // oldest step in flight, next free buffer, steps in flight
    auto inflight_nodes = [&]() {
        int64_t s = 0;
        for (int k = 0; k < nfl; k++) s += t->buf[(head + k) % NBUF].B;
        return s;
    };
    // this rank's own limits: 1 = one that ends the search, 2 = its step quota is done, 0 = none
    auto limit_kind = [&](int64_t inflight) {
        if (t->unbounded || hook_stop || t->pool_exhausted) return 1;
        if (node_limit > 0 && t->evaluated + inflight >= node_limit) return 1;
        const double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        if (max_seconds > 0 && el > max_seconds) return 1;
        if (max_steps > 0 && steps >= max_steps) return 2;
        return 0;
    };
    auto limit_now = [&](int64_t inflight) { return limit_kind(inflight) != 0; };
    auto stop_now = [&](int64_t inflight) {
        if (limit_now(inflight)) return true;
        if (t->comm) return t->x_done;   // (the gap is the ranks' joint decision: x_apply)
        const double gap = tree_gap(t);
        return gap >= 0 && gap <= mip_gap;
    };
    auto batch_size = [&](int64_t inflight) {
        int64_t want = frontier_batch;
        if (node_limit > 0 && node_limit - t->evaluated - inflight < want)
            want = node_limit - t->evaluated - inflight;
        // every evaluated node may need pool rows for two children per output level (the child solved
        // in place branches too); the steps in flight have the same claim.  When the pool cannot take
        // a single node's children the search stops (status 4, stats.pool_exhausted).
        const int64_t per = 2 * (1 + (int64_t)t->dive);
        // (device finish: the steps in flight took their rows out of the free list when they were launched)
        const int64_t room = ((int64_t)t->free_slots.size() - (t->fast_ok ? 0 : per * inflight)) / per;
        if (room < want) {
            want = room > 0 ? room : 0;
            if (want == 0 && inflight == 0) {
                t->pool_exhausted = true;
                (void)fail(ctx, MIPX_ENOMEM, "tree: node pool exhausted (search stopped; raise pool_capacity)");
            }
        }
        return (int)want;
    };
    for (;;) {
    for (;;) {
        // queue steps ahead: the first unconditionally (an empty batch launches nothing), the others only
        // with nodes to take
        while (nfl < NBUF && !tree_queue_empty(t)) {
            const int64_t fl = inflight_nodes();
            if (stop_now(fl)) break;
            const int want = batch_size(fl);
            if (nfl > 0 && want <= 0) break;
            StepBuf &N = t->buf[next];
            int rc = tree_launch(t, N, want);
            if (rc) return bail(rc);
            if (!N.in_flight) break;
            steps++;
            nfl++;
            next = (next + 1) % NBUF;
        }
        if (nfl == 0) break;
        if (t->hook && steps > 0 && steps % t->hook_every == 0 && steps != hooked_at) {
            hooked_at = steps;  // the GPU is busy with the queued steps while the ranks exchange
            if (t->hook(t->hook_user)) hook_stop = true;
        }
        if (t->comm && steps > 0 && steps % t->x_every == 0 && steps != xchg_at) {
            xchg_at = steps;  // collect the exchange posted x_every steps ago, post the next
            const int xrc = x_tick(t, false);
            if (xrc) return xrc;
        }
        int rc = tree_finish(t, t->buf[head], overlap);
        if (rc == MIPX_OK && t->fault_step > 0 && t->steps >= t->fault_step)
            rc = fail(ctx, MIPX_EHIP, "tree: injected fault (MIPX_FAULT_STEP)");
        if (rc) return bail(rc);
        head = (head + 1) % NBUF;
        nfl--;
    }
    if (!t->comm || t->x_done) break;
    // Nothing in flight: out of open nodes, or at one of this rank's limits.  The other ranks may
    // still work: wait for them in the exchange (it blocks until every rank has posted), where open
    // nodes may arrive from a fuller rank or the joint decision to stop is taken.
    t->x_stop_flag = limit_kind(0);
    const int xrc = x_tick(t, true);
    if (xrc) return xrc;
    if (t->x_done) break;
    }
    HIP_TRY(ctx, hipStreamSynchronize(t->st3));
    HIP_TRY(ctx, hipStreamSynchronize(t->stf));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (t->comm) {
        const int xrc = x_close(t);
        if (xrc) return xrc;
    }
    t->solve_seconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (std::getenv("MIPX_TREE_PROFILE")) {
        const double wall = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        std::fprintf(stderr, "[mipx_tree] call: %lld steps in %.2f ms | pop %.2f  wait+d2h %.2f  pseudo-cost %.2f  "
                     "bookkeeping %.2f  children %.2f | lp kernel %.2f | pc: probes %.2f updates %.2f rescoring %.2f\n", (long long)steps, wall,
                     t->phase_ms[0] - ph0[0], t->phase_ms[1] - ph0[1], t->phase_ms[2] - ph0[2] + t->phase_ms[5] - ph0[5] + t->phase_ms[6] - ph0[6],
                     t->phase_ms[3] - ph0[3], t->phase_ms[4] - ph0[4], t->kernel_ms - k0,
                     t->phase_ms[5] - ph0[5], t->phase_ms[6] - ph0[6], t->phase_ms[2] - ph0[2]);
#ifdef MIPX_HOSTPROF
        std::fprintf(stderr, "[mipx_tree]   hostprof: %llu queue pushes %.1f ms (%.0f cycles each), node records %.1f ms\n",
                     g_hp_n, g_hp_push / 2.0e6, g_hp_n ? (double)g_hp_push / g_hp_n : 0.0, g_hp_rec / 2.0e6);
        g_hp_push = g_hp_rec = g_hp_n = 0;
#endif
        std::fprintf(stderr, "[mipx_tree]   popped so far by age in steps: 1:%lld 2:%lld 3:%lld 4:%lld 5:%lld 6:%lld 7+:%lld  mean depth %.1f\n",
                     (long long)t->age_hist[1], (long long)t->age_hist[2], (long long)t->age_hist[3], (long long)t->age_hist[4],
                     (long long)t->age_hist[5], (long long)t->age_hist[6], (long long)t->age_hist[7],
                     (double)t->depth_sum / (double)std::max<int64_t>(1, t->age_hist[1] + t->age_hist[2] + t->age_hist[3] + t->age_hist[4] + t->age_hist[5] + t->age_hist[6] + t->age_hist[7] + t->age_hist[0]));
        std::fprintf(stderr, "[mipx_tree]   probes: read-back %.2f  enqueue %.2f  wait %.2f  results %.2f\n",
                     t->probe_ms[0] - pr0[0], t->probe_ms[1] - pr0[1], t->probe_ms[2] - pr0[2], t->probe_ms[3] - pr0[3]);
    }
    const double gap = tree_gap(t);   // (with a communicator: of the global bounds)
    const bool nothing_open = t->comm ? t->g_counts[4] == 0 : tree_queue_empty(t);
    if (t->unbounded) t->status = 3;
    else if (nothing_open && t->primal == inf) t->status = 2;
    else if (t->primal < inf && gap >= 0 && gap <= mip_gap) t->status = 1;
    else t->status = 4;
    if (out) mipx_tree_get_stats(t, out);
    if (hook_stop) return fail(ctx, MIPX_EHOOK, "mipx_tree_solve: stopped by the step hook");
    if (t->comm && t->x_fatal) return fail(ctx, MIPX_EPEER, "mipx_tree_solve: another rank failed; the search was stopped");
    return MIPX_OK;
}

/* Re-anchoring: the first `max_nodes` open nodes in queue order each get an anchor of their own --
 * the tableau of their warm-start basis, built by one refactor-only launch from the current
 * anchors -- and their descendants inherit it (their bases stay a few pivots away from it, where
 * the root's tableau drifts further away with every level).  Every other open node goes back to
 * the problem's single anchor.  288 GB of HBM hold a million 256 x 128 anchors; 8192 take 2.1 GB. */
int mipx_tree_reanchor(mipx_tree *t, int64_t max_nodes) {
    if (!t || max_nodes < 1) return MIPX_EINVAL;
    mipx_ctx *ctx = t->ctx;
    // (with cut rounds a root that kept cut rows leaves no root anchor: the launch below then refactors
    // from the slack basis)
    if (!t->anchor_mode || (!t->anchor_set && !(t->cuts && t->evaluated > 0)))
        return fail(ctx, MIPX_EINVAL, "mipx_tree_reanchor: needs the anchor mode and a solved root");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(t->st3));
    HIP_TRY(ctx, hipStreamSynchronize(t->st2));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    std::vector<int64_t> order;
    tree_queue_ids(t, order);
    if (t->cuts) {   // an anchor is a tableau of the shared rows: only nodes that carry no cut row get one
        std::stable_partition(order.begin(), order.end(), [&](int64_t id) { return t->nodes[id].ncut == 0; });
        int64_t plain = 0;
        for (int64_t id : order) plain += t->nodes[id].ncut == 0;
        max_nodes = std::min<int64_t>(max_nodes, plain);
    }
    const int64_t K = std::min<int64_t>(max_nodes, (int64_t)order.size());
    if (K == 0) {
        for (int64_t id : order) t->nodes[id].anchor = -1;
        return MIPX_OK;
    }
    const size_t m = t->m, n = t->n;
    double *nT = nullptr, *nvec = nullptr;
    int32_t *nidx = nullptr, *d_sl = nullptr;
    HIP_TRY(ctx, hipMalloc((void **)&nT, (size_t)K * m * n * 8));
    HIP_TRY(ctx, hipMalloc((void **)&nvec, (size_t)K * (n + 3 * m) * 8));
    HIP_TRY(ctx, hipMalloc((void **)&nidx, (size_t)K * (2 * n + m) * 4));
    HIP_TRY(ctx, hipMalloc((void **)&d_sl, (size_t)K * 8));
    std::vector<int32_t> sl(2 * (size_t)K);  // [pool rows | the anchors the nodes have now]
    for (int64_t k = 0; k < K; k++) {
        sl[(size_t)k] = t->nodes[order[(size_t)k]].slot;
        sl[(size_t)(K + k)] = t->nodes[order[(size_t)k]].anchor;
    }
    hipStream_t st = ctx->stream;
    HIP_TRY(ctx, hipMemcpyAsync(d_sl, sl.data(), (size_t)K * 8, hipMemcpyHostToDevice, st));
    // refactor-only solves of the K nodes (from the anchors they have now), every final tableau dumped
    mipx::LpArgs a;
    a.m = t->m; a.n = t->n;
    a.A = t->prob->dA; a.b = t->prob->db; a.c = t->prob->dc;
    a.A_stride = a.b_stride = a.c_stride = 0;
    a.l = t->pool_l; a.u = t->pool_u; a.vstat_in = t->pool_v; a.slot = d_sl; a.max_iter = 0;
    double *gl = nullptr, *gu = nullptr;
    int8_t *gv = nullptr;
    if (t->cuts) {   // (the pool's basis rows are n + mrows wide: dense copies over the shared rows)
        HIP_TRY(ctx, hipMalloc((void **)&gl, (size_t)K * n * 8));
        HIP_TRY(ctx, hipMalloc((void **)&gu, (size_t)K * n * 8));
        HIP_TRY(ctx, hipMalloc((void **)&gv, (size_t)K * (n + m)));
        mipx::PlainGatherArgs ga;
        ga.n = (int)n; ga.m = (int)m; ga.nvs_pool = t->n + t->mrows; ga.count = (int)K; ga.slot = d_sl;
        ga.pool_l = t->pool_l; ga.pool_u = t->pool_u; ga.pool_v = t->pool_v; ga.l = gl; ga.u = gu; ga.v = gv;
        hipLaunchKernelGGL(mipx::gather_plain_nodes, dim3((unsigned)K), dim3(256), 0, st, ga);
        a.l = gl; a.u = gu; a.vstat_in = gv; a.slot = nullptr;
    }
    a.anchor_T = t->prob->anchor_on ? t->prob->anchor_T : nullptr;
    a.anchor_vec = t->prob->anchor_on ? t->prob->anchor_vec : nullptr;
    a.anchor_idx = t->prob->anchor_on ? t->prob->anchor_idx : nullptr;
    if (t->atab_T != nullptr) { a.anchor_sel = d_sl + K; a.atab_T = t->atab_T; a.atab_vec = t->atab_vec; a.atab_idx = t->atab_idx; }
    a.refactor_only = 1;
    a.status = nullptr; a.obj = nullptr; a.x = nullptr; a.y = nullptr; a.vstat_out = nullptr;
    a.iters = nullptr; a.npivots = nullptr; a.batch = (int)K;
    a.dbg_T = nT; a.dbg_vec = nvec; a.dbg_idx = nidx; a.dbg_all = 1;
    int rc = launch_lp_any(t->prob, a, (int)K, st);
    if (rc == MIPX_OK && hipStreamSynchronize(st) != hipSuccess) rc = fail(ctx, MIPX_EHIP, "mipx_tree_reanchor: launch failed");
    (void)hipFree(d_sl);
    if (gl) { (void)hipFree(gl); (void)hipFree(gu); (void)hipFree(gv); }
    if (rc != MIPX_OK) { (void)hipFree(nT); (void)hipFree(nvec); (void)hipFree(nidx); return rc; }
    if (t->atab_T) { (void)hipFree(t->atab_T); (void)hipFree(t->atab_vec); (void)hipFree(t->atab_idx); }
    t->atab_T = nT; t->atab_vec = nvec; t->atab_idx = nidx; t->atab_count = K;
    // the old table is gone: every open node back to the root's anchor, then the K new entries
    // (the host's node records are the authority; a step uploads its batch's entries with the rows)
    for (size_t pos = 0; pos < order.size(); pos++) t->nodes[order[pos]].anchor = pos < (size_t)K ? (int32_t)pos : -1;
    return MIPX_OK;
}

int mipx_tree_set_dive(mipx_tree *t, int on) {
    if (!t) return MIPX_EINVAL;
    if (on && t->max_batch == 1)
        return fail(t->ctx, MIPX_EINVAL, "mipx_tree_set_dive: the exact mode (max_batch = 1) reproduces the "
                                         "reference's node order and cannot dive");
    if (on && t->cuts)
        return fail(t->ctx, MIPX_EINVAL, "mipx_tree_set_dive: a dive child would be solved before its parent's cut "
                                         "rounds; not available with cut rounds");
    if (on < 0 || on > kMaxDive) return fail(t->ctx, MIPX_EINVAL, "mipx_tree_set_dive: depth out of range (0..8)");
    for (const StepBuf &S : t->buf)
        if (S.in_flight) return fail(t->ctx, MIPX_EINVAL, "mipx_tree_set_dive: a step is in flight");
    t->dive = on;
    for (StepBuf &S : t->buf) layout_pack(t, S, 1 + on);
    return MIPX_OK;
}

int mipx_tree_set_step_hook(mipx_tree *t, mipx_tree_hook fn, void *user, int every_steps) {
    if (!t || (fn && every_steps < 1)) return MIPX_EINVAL;
    t->hook = fn;
    t->hook_user = user;
    t->hook_every = fn ? every_steps : 0;
    return MIPX_OK;
}

int mipx_tree_get_stats(mipx_tree *t, mipx_tree_stats *out) {
    if (!t || !out) return MIPX_EINVAL;
    out->evaluated_nodes = t->evaluated;
    out->lp_solved = t->lps;
    out->probes_solved = t->probes;
    out->pivots = t->pivots;
    out->open_nodes = (int64_t)(t->use_bq ? t->bq.size() : t->heap.size());
    out->created_nodes = (int64_t)t->nodes.size();
    out->steps = t->steps;
    out->primal_bound = t->primal;
    out->dual_bound = t->comm ? t->g_dual : tree_dual_bound(t);
    out->gap = tree_gap(t);
    out->solve_seconds = t->solve_seconds;
    out->kernel_ms = t->kernel_ms;
    out->status = t->status;
    out->has_solution = t->have_x ? 1 : 0;
    out->dives = t->dives;
    out->pool_exhausted = t->pool_exhausted ? 1 : 0;
    out->reserved = 0;
    return MIPX_OK;
}

int mipx_exchange_record_len(int n) { return n > 0 ? kRecHead + 5 * n : MIPX_EINVAL; }

int mipx_tree_exchange_record(mipx_tree *t, double *record) {
    if (!t || !record) return MIPX_EINVAL;
    if (!t->comm) return fail(t->ctx, MIPX_EINVAL, "mipx_tree_exchange_record: no communicator attached");
    x_fill_record(t);
    std::memcpy(record, t->x_rec.data(), x_rec_len(t) * 8);
    return MIPX_OK;
}

int mipx_exchange_decide(int world, int n, const double *records, double mip_gap, int allow_migration,
                         mipx_exchange_decision *out) {
    if (world < 1 || n < 1 || !records || !out) return MIPX_EINVAL;
    x_decide(world, n, records, mip_gap, allow_migration != 0, out);
    return MIPX_OK;
}

int mipx_tree_set_comm(mipx_tree *t, mipx_comm *c, int every_steps) {
    if (!t || (c && every_steps < 1)) return MIPX_EINVAL;
    if (c && c->ctx != t->ctx) return fail(t->ctx, MIPX_EINVAL, "mipx_tree_set_comm: communicator of another context");
    t->comm = c;
    t->x_every = c ? every_steps : 0;
    if (!c) return MIPX_OK;
    const size_t n = (size_t)t->n;
    // what every rank holds at sharding time (replicated ramp-up) is counted once
    t->ramp[0] = t->evaluated; t->ramp[1] = t->lps; t->ramp[2] = t->probes; t->ramp[3] = t->pivots;
    t->pc_base.assign(4 * n, 0.0);
    t->pc_own.assign(4 * n, 0.0);
    t->pc_others.assign(4 * n, 0.0);
    for (size_t j = 0; j < n; j++) {
        t->pc_base[j] = t->cost_l[j] * (double)t->times_l[j];
        t->pc_base[n + j] = t->cost_r[j] * (double)t->times_r[j];
        t->pc_base[2 * n + j] = (double)t->times_l[j];
        t->pc_base[3 * n + j] = (double)t->times_r[j];
    }
    if (t->fast_ok) {   // (nothing is in flight between two solves: the tail version is the one every later one copies)
        HIP_TRY(t->ctx, hipStreamSynchronize(t->ctx->stream));
        HIP_TRY(t->ctx, hipStreamSynchronize(t->stf));
        HIP_TRY(t->ctx, hipMemset(tab_at(t, t->tab_tail).own, 0, 4 * n * 8));
        t->pc_others_prev.assign(4 * n, 0.0);
    }
    t->g_dual = tree_dual_bound(t);
    for (int k = 0; k < 4; k++) t->g_counts[k] = t->ramp[k];
    t->g_counts[4] = tree_open_count(t);
    t->x_rounds = 0;
    t->x_done = false;
    return MIPX_OK;
}

/* Test hook: a donation of up to `amount` open nodes from this rank to ITSELF through the communicator's
 * point-to-point path -- pack kernel, ncclSend + ncclRecv to the own rank inside one ncclGroupStart /
 * ncclGroupEnd (custom transport: a device copy), unpack kernel -- so that the wrappers of a migration
 * run on RCCL on a one-GPU box.  The nodes come back under new ids in fresh pool rows; the search is
 * otherwise unchanged.  Returns the number of records moved (>= 0) or an error. */
int64_t mipx_tree_migrate_self(mipx_tree *t, int64_t amount) {
    if (!t || amount < 1) return MIPX_EINVAL;
    if (!t->comm) return fail(t->ctx, MIPX_EINVAL, "mipx_tree_migrate_self: no communicator attached");
    if (t->cuts) return fail(t->ctx, MIPX_EINVAL, "mipx_tree_migrate_self: not with cut rounds");
    for (const StepBuf &S : t->buf)
        if (S.in_flight) return fail(t->ctx, MIPX_EINVAL, "mipx_tree_migrate_self: a step is in flight");
    mipx_comm *c = t->comm;
    mipx_ctx *ctx = t->ctx;
    if (amount > kMaxMigrate) amount = kMaxMigrate;
    const MigLayout L = mig_layout(t, amount);
    const size_t half = (L.bytes + 255) / 256 * 256;
    char *msg = nullptr;
    int rc = comm_msg_buffer(c, 2 * half + (size_t)amount * 4, &msg);
    if (rc) return rc;
    int32_t *d_slots = (int32_t *)(msg + 2 * half);
    std::vector<int32_t> slots;
    int64_t sent = 0, got = 0;
    const int keep_batch = t->x_batch;
    t->x_batch = 0;   // (a test may move every open node)
    rc = mig_pack(t, L, msg, d_slots, amount, slots, &sent);
    t->x_batch = keep_batch;
    if (rc) return rc;
    if ((rc = comm_sendrecv_self(c, msg, msg + half, L.bytes))) return rc;
    for (int32_t sl : slots) t->free_slots.push_back(sl);
    if ((rc = mig_unpack(t, L, msg + half, d_slots, amount, &got))) return rc;
    if (got != sent) return fail(ctx, MIPX_EHIP, "mipx_tree_migrate_self: the count did not survive the round trip");
    t->nodes_sent += sent;
    t->nodes_received += got;
    return got;
}

int mipx_tree_global_stats(mipx_tree *t, mipx_tree_global_stats_t *out) {
    if (!t || !out) return MIPX_EINVAL;
    out->primal_bound = t->primal;
    out->dual_bound = t->comm ? t->g_dual : tree_dual_bound(t);
    out->gap = tree_gap(t);
    if (t->comm) {
        out->evaluated_nodes = t->g_counts[0]; out->lp_solved = t->g_counts[1];
        out->probes_solved = t->g_counts[2]; out->pivots = t->g_counts[3]; out->open_nodes = t->g_counts[4];
    } else {
        out->evaluated_nodes = t->evaluated; out->lp_solved = t->lps; out->probes_solved = t->probes;
        out->pivots = t->pivots; out->open_nodes = tree_open_count(t);
    }
    out->exchanges = t->x_rounds;
    out->nodes_sent = t->nodes_sent;
    out->nodes_received = t->nodes_received;
    out->world = t->comm ? t->comm->world : 1;
    out->incumbent_rank = t->comm ? t->g_inc_rank : (t->have_x ? 0 : -1);
    return MIPX_OK;
}

int mipx_tree_kernel_ms(mipx_tree *t, double out[4]) {
    if (!t || !out) return MIPX_EINVAL;
    out[0] = t->kernel_ms; out[1] = t->k2_ms; out[2] = t->k3_ms; out[3] = 0.0;
    return MIPX_OK;
}

int mipx_tree_cut_stats(mipx_tree *t, int64_t out[8]) {
    if (!t || !out) return MIPX_EINVAL;
    for (int k = 0; k < 8; k++) out[k] = t->cut_totals[k];
    return MIPX_OK;
}

int mipx_tree_solution(mipx_tree *t, double *x) {
    if (!t || !x) return MIPX_EINVAL;
    if (!t->have_x) return fail(t->ctx, MIPX_EINVAL, "mipx_tree_solution: no incumbent");
    std::memcpy(x, t->best_x.data(), (size_t)t->n * 8);
    return MIPX_OK;
}

int mipx_tree_pseudo_costs(mipx_tree *t, double *cost_l, double *cost_r, int32_t *times_l,
                           int32_t *times_r) {
    if (!t || !cost_l || !cost_r || !times_l || !times_r) return MIPX_EINVAL;
    std::memcpy(cost_l, t->cost_l.data(), (size_t)t->n * 8);
    std::memcpy(cost_r, t->cost_r.data(), (size_t)t->n * 8);
    std::memcpy(times_l, t->times_l.data(), (size_t)t->n * 4);
    std::memcpy(times_r, t->times_r.data(), (size_t)t->n * 4);
    return MIPX_OK;
}

/* Replace the pseudo-cost table (multi-GPU: the ranks all-reduce their updates -- the table is a
 * running mean, hence sum-decomposable -- and install the merged table). */
int mipx_tree_set_pseudo_costs(mipx_tree *t, const double *cost_l, const double *cost_r,
                               const int32_t *times_l, const int32_t *times_r) {
    if (!t || !cost_l || !cost_r || !times_l || !times_r) return MIPX_EINVAL;
    std::memcpy(t->cost_l.data(), cost_l, (size_t)t->n * 8);
    std::memcpy(t->cost_r.data(), cost_r, (size_t)t->n * 8);
    std::memcpy(t->times_l.data(), times_l, (size_t)t->n * 4);
    std::memcpy(t->times_r.data(), times_r, (size_t)t->n * 4);
    for (int j = 0; j < t->n; j++) t->has_entry[j] = (times_l[j] > 0 || times_r[j] > 0) ? 1 : 0;
    t->table_dirty = true;
    t->times_dirty = true;
    return MIPX_OK;
}

/* Copy the records (bounds + warm-start basis) of up to max_nodes open nodes, in queue-array
 * order, to host buffers without removing them; returns how many were copied. */
int64_t mipx_tree_peek_open(mipx_tree *t, int64_t max_nodes, double *l, double *u, int8_t *vstat,
                            double *dual_bound) {
    if (!t || max_nodes < 0) return MIPX_EINVAL;
    mipx_ctx *ctx = t->ctx;
    const size_t n = t->n, nv = t->n + t->m, nvs = (size_t)t->n + t->mrows;   // (cut rows of the basis are not copied)
    int64_t k = 0;
    if (hipStreamSynchronize(ctx->stream) != hipSuccess) return MIPX_EHIP;
    std::vector<int64_t> order;
    tree_queue_ids(t, order);
    for (size_t pos = 0; pos < order.size() && k < max_nodes; pos++, k++) {
        const NodeRec &nd = t->nodes[order[pos]];
        const size_t s = (size_t)nd.slot;
        if (l && hipMemcpy(l + k * n, t->pool_l + s * n, n * 8, hipMemcpyDeviceToHost) != hipSuccess) return MIPX_EHIP;
        if (u && hipMemcpy(u + k * n, t->pool_u + s * n, n * 8, hipMemcpyDeviceToHost) != hipSuccess) return MIPX_EHIP;
        if (vstat && hipMemcpy(vstat + k * nv, t->pool_v + s * nvs, nv, hipMemcpyDeviceToHost) != hipSuccess) return MIPX_EHIP;
        if (dual_bound) dual_bound[k] = nd.dual_bound;
    }
    return k;
}

/* Anchor-table entries of the open nodes, in the order of mipx_tree_peek_open (-1: the single anchor). */
int64_t mipx_tree_peek_anchors(mipx_tree *t, int64_t max_nodes, int32_t *anchor) {
    if (!t || max_nodes < 0 || !anchor) return MIPX_EINVAL;
    std::vector<int64_t> order;
    tree_queue_ids(t, order);
    int64_t k = 0;
    for (size_t pos = 0; pos < order.size() && k < max_nodes; pos++, k++)
        anchor[k] = t->atab_T ? t->nodes[order[pos]].anchor : -1;
    return k;
}

/* The anchor table of mipx_tree_reanchor: returns the number of entries; copies them to HOST buffers
 * where given (T: count x m x n, vec: count x (n + 3m), idx: count x (2n + m)). */
int64_t mipx_tree_anchor_table(mipx_tree *t, double *T, double *vec, int32_t *idx) {
    if (!t) return MIPX_EINVAL;
    const size_t K = (size_t)t->atab_count, m = t->m, n = t->n;
    if (K == 0 || !t->atab_T) return 0;
    if (hipStreamSynchronize(t->ctx->stream) != hipSuccess) return MIPX_EHIP;
    if (T && hipMemcpy(T, t->atab_T, K * m * n * 8, hipMemcpyDeviceToHost) != hipSuccess) return MIPX_EHIP;
    if (vec && hipMemcpy(vec, t->atab_vec, K * (n + 3 * m) * 8, hipMemcpyDeviceToHost) != hipSuccess) return MIPX_EHIP;
    if (idx && hipMemcpy(idx, t->atab_idx, K * (2 * n + m) * 4, hipMemcpyDeviceToHost) != hipSuccess) return MIPX_EHIP;
    return (int64_t)K;
}

/* Multi-GPU sharding: after a replicated, deterministic ramp-up every rank
 * keeps the open nodes whose position in the queue array is congruent to its rank and drops the
 * others (they live on the other ranks).  The dual bound of the whole tree is then the MIN over
 * ranks of mipx_tree_stats.dual_bound. */
int mipx_tree_keep_shard(mipx_tree *t, int rank, int world) {
    if (!t || world < 1 || rank < 0 || rank >= world) return MIPX_EINVAL;
    std::vector<int64_t> old;
    tree_queue_ids(t, old);
    t->heap.h.clear();
    t->bq.clear();
    while (!t->open_bounds.empty()) t->open_bounds.pop();
    for (size_t pos = 0; pos < old.size(); pos++) {
        const int64_t id = old[pos];
        if (t->search != 0) t->is_open[id] = 0;
        if ((int)(pos % (size_t)world) == rank) {
            tree_push(t, id);
        } else {
            t->free_slots.push_back(t->nodes[id].slot);
            t->nodes[id].slot = -1;
        }
    }
    if (rank != 0) t->closed_min = std::numeric_limits<double>::infinity();  // counted once, on rank 0
    return MIPX_OK;
}

int64_t mipx_tree_trace_cuts(mipx_tree *t, int64_t capacity, int32_t *counters) {
    if (!t) return MIPX_EINVAL;
    const int64_t have = (int64_t)t->tr_cuts.size() / 8;
    const int64_t k = have < capacity ? have : capacity;
    if (counters && k > 0) std::memcpy(counters, t->tr_cuts.data(), (size_t)k * 8 * 4);
    return have;
}

/* Test hooks of the cut rounds: the open nodes' ids, cut lists (ids into the cut store) and the basis
 * codes of their cut rows, in the order of mipx_tree_peek_open; and the cut store itself. */
int64_t mipx_tree_peek_cuts(mipx_tree *t, int64_t max_nodes, int64_t *node_id, int32_t *ncut, int32_t *cut_ids,
                            int8_t *cut_vstat) {
    if (!t || max_nodes < 0) return MIPX_EINVAL;
    mipx_ctx *ctx = t->ctx;
    if (hipStreamSynchronize(t->st3) != hipSuccess || hipStreamSynchronize(ctx->stream) != hipSuccess) return MIPX_EHIP;
    std::vector<int64_t> order;
    tree_queue_ids(t, order);
    const size_t n = t->n, m = t->m, nvs = n + (size_t)t->mrows, K = (size_t)t->kc;
    int64_t k = 0;
    for (size_t pos = 0; pos < order.size() && k < max_nodes; pos++, k++) {
        const NodeRec &nd = t->nodes[order[pos]];
        const size_t s = (size_t)nd.slot;
        if (node_id) node_id[k] = order[pos];
        if (ncut) ncut[k] = t->cuts ? nd.ncut : 0;
        if (t->cuts && K > 0) {
            if (cut_ids && hipMemcpy(cut_ids + k * K, t->pool_ids + s * K, K * 4, hipMemcpyDeviceToHost) != hipSuccess) return MIPX_EHIP;
            if (cut_vstat && hipMemcpy(cut_vstat + k * K, t->pool_v + s * nvs + n + m, K, hipMemcpyDeviceToHost) != hipSuccess) return MIPX_EHIP;
        }
    }
    return k;
}

int64_t mipx_tree_cut_store(mipx_tree *t, int64_t capacity, double *pi, double *pi0) {
    if (!t) return MIPX_EINVAL;
    if (!t->cuts) return 0;
    if (hipStreamSynchronize(t->ctx->stream) != hipSuccess) return MIPX_EHIP;
    int32_t cnt = 0;
    if (hipMemcpy(&cnt, t->store_count, 4, hipMemcpyDeviceToHost) != hipSuccess) return MIPX_EHIP;
    int64_t have = cnt < t->store_cap ? cnt : t->store_cap;
    const int64_t k = have < capacity ? have : capacity;
    if (pi && k > 0 && hipMemcpy(pi, t->store_pi, (size_t)k * t->n * 8, hipMemcpyDeviceToHost) != hipSuccess) return MIPX_EHIP;
    if (pi0 && k > 0 && hipMemcpy(pi0, t->store_pi0, (size_t)k * 8, hipMemcpyDeviceToHost) != hipSuccess) return MIPX_EHIP;
    return have;
}

int mipx_tree_cut_rows_per_node(const mipx_tree *t) { return t ? t->kc : MIPX_EINVAL; }

int64_t mipx_tree_trace(mipx_tree *t, int64_t capacity, int64_t *node_id, int32_t *lp_status,
                        int32_t *branch_var, double *objective) {
    if (!t) return MIPX_EINVAL;
    const int64_t k = (int64_t)t->tr_id.size() < capacity ? (int64_t)t->tr_id.size() : capacity;
    for (int64_t i = 0; i < k; i++) {
        if (node_id) node_id[i] = t->tr_id[i];
        if (lp_status) lp_status[i] = t->tr_status[i];
        if (branch_var) branch_var[i] = t->tr_bidx[i];
        if (objective) objective[i] = t->tr_obj[i];
    }
    return (int64_t)t->tr_id.size();
}

}  // extern "C"
