/*
 * mipx.h -- C ABI of the MI355X-native branch-and-bound node engine (libmipx.so).
 *
 * The reference (spkelle2/simple_mip_solver) has no FFI: its node hot path calls COIN-OR Clp
 * through the cylp Python wrapper.  Each entry point below replaces one such call site; the
 * Python mirror of the reference's Node/BranchAndBound plugin surface in
 * simple_mip_solver_amd/ binds these through ctypes (see INTEGRATION.md for the stub a
 * maintainer of the reference would add).
 *
 * Conventions
 *   - Problem form on the hot path (simple_mip_solver/nodes/base_node.py:111-112,
 *     algorithms/base_algorithm.py:48-61):  min c'x  s.t.  A x >= b,  l <= x <= u,  l >= 0.
 *   - A is dense row-major f64 (m x n).  u may be +inf.
 *   - Basis/status codes are Clp's, as read by the reference (base_node.py:530):
 *     1 basic, 2 at upper, 3 at lower (anything else is treated as at lower).
 *     Layout: n structural codes then m row (slack) codes -- the concatenation of the pair
 *     returned by CyClpSimplex.getBasisStatus() (base_node.py:589).
 *   - LP status codes are Clp's (base_node.py:274-275, pseudo_cost.py:86):
 *     0 optimal, 1 primal infeasible, 2 dual infeasible/unbounded, 3 iteration limit.
 *   - Every function returns 0 on success or a negative MIPX_E* code; nothing throws across
 *     the ABI; mipx_last_error() returns a description of the last failure on that context.
 *   - Host buffers stay owned by the caller; the library keeps no host pointer past return.
 *   - A context is bound to one GPU and one HIP stream and is not thread-safe.
 *   - There is NO CPU fallback: without a usable gfx950 device mipx_ctx_create fails.
 */
#ifndef MIPX_H
#define MIPX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MIPX_OK 0
#define MIPX_EINVAL -1    /* bad argument */
#define MIPX_ENODEV -2    /* no usable HIP device */
#define MIPX_EHIP -3      /* HIP runtime error (see mipx_last_error) */
#define MIPX_ETOOBIG -4   /* (m, n) exceeds what the on-chip tableau kernels support */
#define MIPX_ENOMEM -5
#define MIPX_EHOOK -6     /* a step hook asked mipx_tree_solve to stop */
#define MIPX_EPEER -7     /* multi-GPU: another rank failed, the joint search was stopped */

typedef struct mipx_ctx mipx_ctx;
typedef struct mipx_problem mipx_problem;

/* ABI version of this header (bumped on any signature change). */
int mipx_abi_version(void);

/* Number of visible HIP devices (0 if none / runtime unavailable). */
int mipx_device_count(void);

/* Create a context on HIP device `device` with its own stream. */
int mipx_ctx_create(int device, mipx_ctx **out);
void mipx_ctx_destroy(mipx_ctx *ctx);
const char *mipx_last_error(const mipx_ctx *ctx);
int mipx_ctx_sync(mipx_ctx *ctx);

/*
 * Upload the data shared by every node of one tree: replaces the reference keeping (A,b,c)
 * inside every node's CyClpSimplex (base_node.py:74, rebuilt per child at :592-607).
 */
int mipx_problem_create(mipx_ctx *ctx, int m, int n, const double *A_rowmajor, const double *b,
                        const double *c, mipx_problem **out);
void mipx_problem_destroy(mipx_problem *p);
/* Node LPs with the in-place dive (the frontier engine's throughput option, exposed for parity
 * tests): where a node LP ends optimal, fractional and with objective < cutoff, and the branching
 * rule (0 most fractional, base_node.py:544-562; 1 pseudo costs, branch/pseudo_cost.py:118-133,
 * only when every fractional variable has an entry) picks a basic variable, the workgroup moves
 * one bound of it (towards the side the rule expects to cost less) and continues the dual simplex
 * on the tableau it holds: the child LP of base_node.py:592-608 without a reload or a
 * refactorisation.  HOST buffers.  status, obj, x, vstat_out, iters, npivots have 2 * batch rows:
 * the nodes, then their children (status -1 where no dive happened); dive_var (-1: none),
 * dive_dir (0 left: x <= floor, 1 right: x >= ceil), dive_val (the value branched on): batch. */
int mipx_lp_dive_batch(mipx_problem *p, int batch, const double *l, const double *u,
                       const int8_t *vstat_in, int max_iter, int rule, const int32_t *int_idx,
                       int n_int, const double *cost_l, const double *cost_r,
                       const uint8_t *has_entry, double cutoff, int32_t *status, double *obj,
                       double *x, int8_t *vstat_out, int32_t *iters, int32_t *npivots,
                       int32_t *dive_var, int32_t *dive_dir, double *dive_val);
/* The same with up to `depth` (1..8) dive children IN A ROW on one tableau -- a plunge: after each
 * level's LP the rule branches again where it can.  status, obj, x, vstat_out, iters, npivots have
 * (depth + 1) * batch rows: the nodes, then their first dive children, then the second, ... (status -1
 * where a level was not reached); dive_var / dive_dir / dive_val depth * batch entries: the decision
 * taken after level p's LP at [p * batch + node]. */
int mipx_lp_plunge_batch(mipx_problem *p, int batch, int depth, const double *l, const double *u,
                         const int8_t *vstat_in, int max_iter, int rule, const int32_t *int_idx,
                         int n_int, const double *cost_l, const double *cost_r,
                         const uint8_t *has_entry, double cutoff, int32_t *status, double *obj,
                         double *x, int8_t *vstat_out, int32_t *iters, int32_t *npivots,
                         int32_t *dive_var, int32_t *dive_dir, double *dive_val);
/*
 * Optional: make warm starts refactor from the tableau of the basis `vstat` (n+m Clp codes, e.g.
 * the root's optimal basis) instead of from the slack basis.  The number of refactorisation pivots
 * of a node then equals the distance between its basis and the anchor's, not the number of its
 * basic structurals.  NULL switches it off.  Results stay within rounding of the unanchored solve
 * (a different, equally valid pivot sequence); only the register-resident tiles support it.
 */
int mipx_problem_set_anchor(mipx_problem *p, const int8_t *vstat);

/*
 * Batched node LP relaxation: replaces `self.lp.dual()` + the status/objective/solution reads
 * of BaseNode._bound_lp (base_node.py:273-280) for `batch` nodes at once, and the truncated
 * solves of BaseNode._strong_branch (base_node.py:645-646) when max_iter > 0.
 *
 *   l, u        batch x n   per-node variable bounds (base_node.py:595-600)
 *   vstat_in    batch x (n+m) warm-start basis (base_node.py:608 setBasisStatus) or NULL = cold
 *   max_iter    <= 0: run to termination; > 0: lp.maxNumIteration (base_node.py:645)
 *   status      batch       Clp status code
 *   obj         batch       c'x (+inf if status 1; for status 2 the symbolic bound M is reported as 1e10)
 *   x           batch x n   primalVariableSolution
 *   y           batch x m   dualConstraintSolution (row duals; 0 where the row's slack is basic)
 *   vstat_out   batch x (n+m) getBasisStatus
 *   iters       batch       dual simplex iterations (CyClpSimplex.iteration)
 *   npivots     batch       tableau pivots incl. the warm-start refactorisation
 * Any output pointer may be NULL.  All pointers are HOST pointers.
 * A single node without a basis above the register tiles (a cold root of up to 1024 x 1024) is one LP of
 * thousands of pivots: it is spread over up to 256 workgroups, one launch per pivot (csrc/lp_kernel_root.hip.h)
 * -- the same result bit for bit as the one-workgroup kernel, about ten times sooner at 1024 x 512.
 */
int mipx_lp_solve_batch(mipx_problem *p, int batch, const double *l, const double *u,
                        const int8_t *vstat_in, int max_iter, int32_t *status, double *obj,
                        double *x, double *y, int8_t *vstat_out, int32_t *iters,
                        int32_t *npivots);

/*
 * The same for nodes that carry cut rows (the kernel variant the frontier engine's cut rounds run;
 * exposed for the parity tests): node k solves  min c'x, A x >= b, cut rows cut_ids[k*kc + i] (i <
 * ncut[k]) of the store (cut_pi: ncuts_total x n, cut_pi0), l_k <= x <= u_k -- base_node.py:459-460
 * appends a selected cut as one more `>=` row.  Results are those of mipx_lp_solve_batch on the
 * problem with the node's rows materialised.  vstat_in / vstat_out have n + m + kc entries per node
 * (structurals, shared rows, cut rows), y has m + kc.  kc <= 64, m + kc <= 192, n <= 256.
 * HOST pointers.
 */
int mipx_lp_solve_batch_cuts(mipx_problem *p, int batch, const double *l, const double *u,
                             const int8_t *vstat_in, int ncuts_total, const double *cut_pi,
                             const double *cut_pi0, int kc, const int32_t *ncut,
                             const int32_t *cut_ids, int max_iter, int32_t *status, double *obj,
                             double *x, double *y, int8_t *vstat_out, int32_t *iters,
                             int32_t *npivots);

/*
 * Device-resident variant (inputs and outputs already in HBM; nothing crosses PCIe).
 * Pointers are DEVICE pointers obtained from mipx_dev_alloc.  Asynchronous on the context
 * stream; call mipx_ctx_sync before reading results back.
 */
int mipx_lp_solve_batch_dev(mipx_problem *p, int batch, const double *l, const double *u,
                            const int8_t *vstat_in, int max_iter, int32_t *status, double *obj,
                            double *x, double *y, int8_t *vstat_out, int32_t *iters,
                            int32_t *npivots);

/*
 * Independent problems, one per LP (BASELINE config C2: a batch of root relaxations of different
 * random instances): A is batch x m x n, b batch x m, c batch x n, l/u batch x n; cold start.
 * Replaces a Python loop of `BaseNode(model_k.lp, ...).bound()` over models (base_node.py:259-286).
 */
int mipx_lp_solve_multi(mipx_ctx *ctx, int m, int n, int batch, const double *A, const double *b,
                        const double *c, const double *l, const double *u, int max_iter,
                        int32_t *status, double *obj, double *x, int8_t *vstat_out, int32_t *iters,
                        int32_t *npivots);

/*
 * Batched Gomory mixed-integer cuts with numerically safe rounding: replaces
 * BaseNode.tableau + _find_gomory_cuts (base_node.py:468-530) and the numerically_safe_cut call of
 * _generate_cuts (base_node.py:381, utils/floating_point.py:40-167) for `batch` solved nodes.
 *   l, u, vstat   the nodes' bounds and OPTIMAL basis (getBasisStatus after the solve)
 *   x             batch x n  the nodes' solutions clipped at 0 (base_node.py:310)
 *   is_int        n bytes, 1 for integer variables
 *   max_term      utils/tolerance.py max_term (1e3)
 *   ncuts         batch        number of cuts generated per node
 *   row_idx       batch x m    tableau row of each cut = rank of its basic variable (the suffix
 *                              of the reference's cut name cut_gomory_<node>_<round>_<row>)
 *   pi, pi0       batch x m x n, batch x m   raw cuts  pi.x >= pi0
 *   safe_pi(0)    same shapes: the rounded outer approximation ('over' estimate)
 * All pointers are HOST pointers; outputs other than ncuts may be NULL.
 */
int mipx_gomory_batch(mipx_problem *p, int batch, const double *l, const double *u,
                      const int8_t *vstat, const double *x, const uint8_t *is_int, double max_term,
                      int32_t *ncuts, int32_t *row_idx, double *pi, double *pi0, double *safe_pi,
                      double *safe_pi0);

/*
 * Batched cut selection: replaces the arithmetic of BaseNode._select_cuts (base_node.py:414-452).
 *   npool       batch            cuts in each node's pool (<= kmax)
 *   pi, pi0     batch x kmax x n, batch x kmax   the pools (rows beyond npool ignored)
 *   x           batch x n        the nodes' LP solutions
 *   max_nonzero_coefs, min_cut_depth               as the reference's keyword arguments
 *   cos_parallel    cos(radians(parallel_cut_tolerance)): a cut closer than that to an added one is skipped
 *   max_abs_coef    max_relative_cut_term_ratio * node.max_term
 *   nadded, added   batch, batch x kmax   pool positions in the order the cuts are added to the LP
 *   terminator      batch   0 none, 1 'no cuts', 2 'no improving cuts', 3 'no sufficient cuts'
 *   depth           batch x kmax (may be NULL)  euclidean depth per pool cut, +inf if not a candidate
 * HOST pointers.
 */
int mipx_cut_select_batch(mipx_ctx *ctx, int n, int batch, int kmax, const int32_t *npool,
                          const double *pi, const double *pi0, const double *x,
                          int max_nonzero_coefs, double min_cut_depth, double cos_parallel,
                          double max_abs_coef, int32_t *nadded, int32_t *added, int32_t *terminator,
                          double *depth);

/*
 * The rounding stage of the cut path on its own: utils/floating_point.py numerically_safe_cut
 * (:40-103) -- scale_cut (:11-37), then per coefficient get_fraction (:106-167) with the
 * good / exact approximation rule (:77-92) -- for `batch` supplied cuts of n coefficients.  The
 * same device functions mipx_gomory_batch rounds with.
 *   estimate       1 'over' (pi.x >= pi0 cuts: coefficients over-, right-hand side under-estimated),
 *                  2 'under'
 *   make_integer   0 / 1 (multiply through by the lcm of the denominators, :94-101)
 *   max_term       largest numerator / denominator of a coefficient (tolerance.max_term)
 *   safe_pi(0)     batch x n, batch
 *   num, den       batch x (n+1) (may be NULL): numerator / denominator chosen per coefficient, then
 *                  of the right-hand side -- integers held in doubles (exact: they are <= max_term,
 *                  or a rounded |x| > max_term)
 *   scaled_pi(0)   batch x n, batch (may be NULL): scale_cut's output
 *   nonzero        batch (may be NULL): 0 where pi == 0 (the cut is returned unchanged)
 * HOST pointers.
 */
int mipx_safe_cut_batch(mipx_ctx *ctx, int n, int batch, const double *pi, const double *pi0,
                        int estimate, int make_integer, double max_term, double *safe_pi,
                        double *safe_pi0, double *num, double *den, double *scaled_pi,
                        double *scaled_pi0, int32_t *nonzero);
/* get_fraction (utils/floating_point.py:106-167) for `count` numbers: x, max_term, estimate
 * (0 none, 1 over, 2 under) per item -> numerator, denominator (integers held in doubles). */
int mipx_get_fraction_batch(mipx_ctx *ctx, int count, const double *x, const double *max_term,
                            const int32_t *estimate, double *num, double *den);

/* Device memory owned by the library, for the device-resident entry points. */
int mipx_dev_alloc(mipx_ctx *ctx, size_t bytes, void **dptr);
int mipx_dev_free(mipx_ctx *ctx, void *dptr);
int mipx_memcpy_h2d(mipx_ctx *ctx, void *dst_dev, const void *src_host, size_t bytes);
int mipx_memcpy_d2h(mipx_ctx *ctx, void *dst_host, const void *src_dev, size_t bytes);

/*
 * HIP-event timer on the context stream (the stream the kernels are launched on), used by
 * bench.py for the roofline figure.  stop returns elapsed milliseconds since start.
 */
int mipx_timer_start(mipx_ctx *ctx);
int mipx_timer_stop(mipx_ctx *ctx, float *ms);

/*
 * Test hook: after mipx_debug_enable(p) every solve also dumps the final simplex state of node 0
 * of the batch: T (m x n row-major), vec = [d (n) | beta0 (m) | ba (m) | bb (m)],
 * idx = [nvar (n) | bvar (m) | side (n)].  Used by the tableau-parity tests only.
 */
int mipx_debug_enable(mipx_problem *p);
int mipx_debug_read(mipx_problem *p, double *T, double *vec, int32_t *idx);

/* ------------------------------------------------------------------------------------------
 * Native frontier engine: BranchAndBound.solve() (algorithms/branch_and_bound.py:215-266) for the
 * stock node classes, a batch of open nodes per step, node records resident in HBM.
 * ---------------------------------------------------------------------------------------- */
typedef struct mipx_tree mipx_tree;

typedef struct mipx_tree_stats {
    int64_t evaluated_nodes;  /* BranchAndBound.evaluated_nodes */
    int64_t lp_solved;        /* node LP relaxations solved to termination (_bound_lp calls) */
    int64_t probes_solved;    /* truncated strong-branching LPs */
    int64_t pivots;           /* tableau pivots of the node LPs (refactor + simplex) */
    int64_t open_nodes;
    int64_t created_nodes;    /* == next_node_idx */
    int64_t steps;            /* frontier batches processed */
    double primal_bound;      /* +inf while no incumbent */
    double dual_bound;
    double gap;               /* current_gap; -1 encodes None (no incumbent) */
    double solve_seconds;     /* accumulated wall time inside mipx_tree_solve */
    double kernel_ms;         /* accumulated HIP-event time of the node LP kernel */
    int32_t status;           /* 0 unsolved, 1 optimal, 2 infeasible, 3 unbounded,
                                 4 stopped on iterations or time */
    int32_t has_solution;
    int64_t dives;            /* of lp_solved / evaluated_nodes: children solved in place by the dive */
    int32_t pool_exhausted;   /* 1: the search stopped (status 4) because pool_capacity is used up */
    int32_t reserved;
} mipx_tree_stats;

/*
 * branch_rule: 0 most fractional (BaseNode.branch, base_node.py:532-562),
 *              1 pseudo cost with strong-branching initialisation (pseudo_cost.py:22-133)
 * search_rule: 0 best first (BaseNode.__lt__, base_node.py:677-681), 1 depth first
 *              (nodes/search/depth_first.py:24-28)
 * l, u: root bounds; int_idx: integerIndices.  pool_capacity: node records kept in HBM.
 */
int mipx_tree_create(mipx_problem *p, const int32_t *int_idx, int n_int, const double *l,
                     const double *u, int branch_rule, int search_rule, int strong_branch_iters,
                     int max_batch, int64_t pool_capacity, mipx_tree **out);
/*
 * The same with Gomory cut rounds inside the engine (BASELINE config C4): every evaluated node runs
 * BaseNode._base_bound's loop (base_node.py:137-230) -- LP, then while fractional and progressing at
 * most max_cut_generation_iterations rounds of: drop cuts with zero dual (:326-341), GMI cuts from
 * the tableau of its basis with safe rounding (:365-385, :468-511), selection (:387-466), re-solve,
 * stall test (:320-324).  Cut rows live in an append-only store in HBM (n + 1 doubles per cut); a
 * node record carries the ids of its rows (at most max_cuts_per_node <= 64), children and
 * strong-branching probes inherit them with the basis (base_node.py:602-608).  The LP of a node with
 * k cuts is exactly the (m + k)-row LP: same arithmetic as mipx_lp_solve_batch on the materialised
 * rows.  Register-tile shapes only: m + max_cuts_per_node <= 192, n <= 256.  cuts == NULL: no cut
 * rounds (mipx_tree_create).  The in-place dive and mipx_tree_reanchor are not available with cuts.
 */
typedef struct mipx_cut_params {
    int32_t max_cut_generation_iterations; /* tolerance.max_cut_generation_iterations (10) */
    int32_t max_nonzero_coefs;             /* _select_cuts keyword (base_node.py:387) */
    int32_t max_cuts_per_node;             /* <= 64; 0 = 64.  A node whose list is full adds no more (counted as dropped) */
    int32_t exact_tableau;                 /* 1: the GMI tableau is refactored from the slack basis like the
                                              per-node path (node-for-node parity); 0: anchors allowed */
    double cutting_plane_progress_tolerance; /* base_node.py:292 (1e-4) */
    double min_cut_depth;                  /* _select_cuts keyword (1e-8) */
    double cos_parallel;                   /* cos(radians(parallel_cut_tolerance)) */
    double max_abs_coef;                   /* max_relative_cut_term_ratio * node.max_term (max |A|) */
    double max_term;                       /* tolerance.max_term (1e3): largest numerator / denominator */
    double max_dual_bound;                 /* _base_bound keyword (+inf) */
    int64_t store_capacity;                /* cuts the store can hold; 0 = 1 << 20 */
} mipx_cut_params;
int mipx_tree_create_ex(mipx_problem *p, const int32_t *int_idx, int n_int, const double *l,
                        const double *u, int branch_rule, int search_rule, int strong_branch_iters,
                        int max_batch, int64_t pool_capacity, const mipx_cut_params *cuts,
                        mipx_tree **out);
/* Running totals of the cut loop over every evaluated node, in the order BaseNode._base_bound returns
 * them (base_node.py:218-226): total_cut_generation_iterations, total_iterations_gmic_created,
 * total_number_gmic_created, total_iterations_gmic_added, total_number_gmic_added,
 * total_iterations_gmic_removed, total_number_gmic_removed; out[7] = cuts dropped for lack of room
 * (a full node list, pool slab or store: 0 unless a capacity was set too small). */
int mipx_tree_cut_stats(mipx_tree *t, int64_t out[8]);
void mipx_tree_destroy(mipx_tree *t);
/* Run or continue the search (re-entrant like BranchAndBound.solve).  Limits <= 0 mean none.
 * frontier_batch = 1 reproduces the reference's node order exactly. */
int mipx_tree_solve(mipx_tree *t, int64_t node_limit, double mip_gap, double max_seconds,
                    int frontier_batch, int64_t max_steps, mipx_tree_stats *out);
int mipx_tree_get_stats(mipx_tree *t, mipx_tree_stats *out);
int mipx_tree_solution(mipx_tree *t, double *x);
int mipx_tree_set_primal_bound(mipx_tree *t, double bound); /* initial_primal_bound / exchange */
/* Throughput option: once the root is solved, anchor every later refactorisation at its tableau. */
int mipx_tree_set_anchor_mode(mipx_tree *t, int on);
int mipx_tree_pseudo_costs(mipx_tree *t, double *cost_l, double *cost_r, int32_t *times_l,
                           int32_t *times_r);
/* Install a pseudo-cost table (n entries each): the merged table of a multi-GPU exchange
 * (PseudoCostBranchNode.pseudo_costs shared through _kwargs, branch/pseudo_cost.py:38-43). */
int mipx_tree_set_pseudo_costs(mipx_tree *t, const double *cost_l, const double *cost_r,
                               const int32_t *times_l, const int32_t *times_r);
/* Re-anchoring (anchor mode, register-tile shapes): the first max_nodes open nodes in queue order
 * each get an anchor of their own -- the tableau of their warm-start basis -- which their
 * descendants inherit, so that warm starts refactor over the few pivots that separate a node from
 * that ancestor instead of the many that separate it from the root.  All other open nodes return
 * to the root's anchor; an earlier table is released.  max_nodes anchors take
 * max_nodes * 8 * m * n bytes of HBM (8192 at 256 x 128: 2.1 GB).  No reference counterpart. */
int mipx_tree_reanchor(mipx_tree *t, int64_t max_nodes);
/* For the CPU baseline and the parity tests: the anchor-table entry of every open node, in the
 * order of mipx_tree_peek_open (-1: the root's anchor), and the table itself copied to HOST buffers
 * (T: count x m x n, vec: count x (n + 3m), idx: count x (2n + m); any may be NULL).  Both return
 * a count. */
int64_t mipx_tree_peek_anchors(mipx_tree *t, int64_t max_nodes, int32_t *anchor);
int64_t mipx_tree_anchor_table(mipx_tree *t, double *T, double *vec, int32_t *idx);
/* Throughput option for frontier batches > 1: the workgroup that solved a node branches in place
 * when the rule can decide without probes (see mipx_lp_dive_batch) and solves the child on the
 * tableau it holds -- and that child's child, up to `depth` (0 off, 1..8) in a row (a plunge: see
 * mipx_lp_plunge_batch); they are evaluated in the same step, their siblings are queued as usual.
 * A best-first search with a plunge of that depth: same optimum, different node order.  Not
 * available with max_batch = 1 (the reference's exact order) or with cut rounds. */
int mipx_tree_set_dive(mipx_tree *t, int depth);
/* Step hook: `fn(user)` is called on the calling thread every `every_steps` frontier steps of
 * mipx_tree_solve, after the next step's kernels are queued and before the host waits for the
 * current one -- the slot in which a multi-GPU rank runs its incumbent / bound / pseudo-cost
 * all-reduce without draining the launch pipeline.  Inside the hook the tree may be
 * read (get_stats, pseudo_costs) and mipx_tree_set_primal_bound / mipx_tree_set_pseudo_costs may be
 * called; a nonzero return ends the solve with MIPX_EHOOK.  fn = NULL removes the hook.  The
 * reference has no counterpart (single process). */
typedef int (*mipx_tree_hook)(void *user);
int mipx_tree_set_step_hook(mipx_tree *t, mipx_tree_hook fn, void *user, int every_steps);
/* Copy the records of up to max_nodes open nodes (queue-array order) to HOST buffers without
 * removing them: l, u (max_nodes x n), vstat (max_nodes x (n+m)), dual_bound (max_nodes); any may
 * be NULL.  Returns the number copied (used by bench.py to time the CPU oracle on the very LPs the
 * GPU is about to solve). */
int64_t mipx_tree_peek_open(mipx_tree *t, int64_t max_nodes, double *l, double *u, int8_t *vstat,
                            double *dual_bound);
/* Multi-GPU: keep only this rank's share of the open nodes after a replicated ramp-up. */
int mipx_tree_keep_shard(mipx_tree *t, int rank, int world);
/* Test hook: record (node id, LP status, branching variable or -1, objective) per evaluated
 * node; mipx_tree_trace returns the number recorded and copies at most `capacity` of them. */
int mipx_tree_set_trace(mipx_tree *t, int on);
int64_t mipx_tree_trace(mipx_tree *t, int64_t capacity, int64_t *node_id, int32_t *lp_status,
                        int32_t *branch_var, double *objective);

/* Device time by kernel over the tree's life, HIP events on the stream of the launches (ms): [0] K1 node LPs
 * (mipx_tree_stats.kernel_ms), [1] K2 gomory_cuts, [2] pool_append + K3 select_cuts, [3] reserved.  [1], [2]:
 * trees with cut rounds only.  For the roofline entries of bench.py. */
int mipx_tree_kernel_ms(mipx_tree *t, double out[4]);
/* Test hooks of the cut rounds (mipx_tree_create_ex with cut parameters).  mipx_tree_trace_cuts: with
 * the trace on, 8 ints per evaluated node in trace order -- cut rounds, iterations / number of GMICs
 * created, added, removed (the per-node increments of BaseNode._base_bound's totals,
 * base_node.py:215-222), the cut rows the node ends with; returns the number of nodes recorded.
 * mipx_tree_peek_cuts: for the open nodes in the order of mipx_tree_peek_open their ids, cut counts, cut
 * lists (max_nodes x mipx_tree_cut_rows_per_node ids into the cut store) and the basis codes of their cut
 * rows (same shape); any may be NULL.  mipx_tree_cut_store: the cuts added so far (pi: count x n, pi0),
 * returns the count. */
int64_t mipx_tree_trace_cuts(mipx_tree *t, int64_t capacity, int32_t *counters);
int64_t mipx_tree_peek_cuts(mipx_tree *t, int64_t max_nodes, int64_t *node_id, int32_t *ncut, int32_t *cut_ids,
                            int8_t *cut_vstat);
int64_t mipx_tree_cut_store(mipx_tree *t, int64_t capacity, double *pi, double *pi0);
int mipx_tree_cut_rows_per_node(const mipx_tree *t);

/* ------------------------------------------------------------------------------------------
 * Multi-GPU: one process per GPU, open nodes sharded, a best-first queue per rank.
 * The reference is single-process (branch_and_bound.py:215-241): no counterpart.  The communicator
 * binds RCCL (xGMI) directly -- librccl.so is loaded on first use, nothing else is needed.
 * ---------------------------------------------------------------------------------------- */
typedef struct mipx_comm mipx_comm;
/* Rank 0 makes the id (ncclGetUniqueId), the launcher hands its 128 bytes to every rank. */
int mipx_comm_unique_id(char id[128]);
int mipx_comm_create_rccl(mipx_ctx *ctx, const char id[128], int rank, int world, mipx_comm **out);
/* The same protocol over caller-supplied host-buffer primitives (0 = ok): the CPU tests and the
 * one-GPU rehearsal plug a gloo process group in here (RCCL refuses two ranks on one device). */
typedef struct mipx_comm_ops {
    int (*allgather)(void *user, const void *send, void *recv, size_t bytes_per_rank);
    int (*send)(void *user, int peer, const void *buf, size_t bytes);
    int (*recv)(void *user, int peer, void *buf, size_t bytes);
} mipx_comm_ops;
int mipx_comm_create_custom(mipx_ctx *ctx, int rank, int world, const mipx_comm_ops *ops, void *user,
                            mipx_comm **out);
void mipx_comm_destroy(mipx_comm *c);
int mipx_comm_rank(const mipx_comm *c);
int mipx_comm_size(const mipx_comm *c);
/* HOST buffers; blocking.  (The engine's own exchange posts its all-gather and collects it later.) */
int mipx_comm_allgather(mipx_comm *c, const void *send, void *recv, size_t bytes_per_rank);
int mipx_comm_barrier(mipx_comm *c);

/*
 * Attach a communicator to a tree that has run the replicated ramp-up and kept its shard
 * (mipx_tree_keep_shard).  From then on mipx_tree_solve is a COLLECTIVE call: every rank must call it.
 * Every `every_steps` frontier steps -- after the next step's kernels are queued -- a rank collects
 * the all-gather it posted last time and posts the next: one record per rank with its incumbent
 * value and solution, the dual bound of its shard, its open-node count, its stop flag, its counters
 * and its pseudo-cost samples (a running mean is sum-decomposable, pseudo_cost.py:97-98).  Applied:
 *   - the best incumbent of any rank becomes everybody's (value AND solution: the reference's
 *     contract is objective_value and solution, branch_and_bound.py:236-241);
 *   - the merged pseudo-cost table;
 *   - termination, decided identically by all ranks from the same records: every rank idle (no open
 *     node anywhere), the global gap |primal - min dual| / |primal| <= mip_gap, any rank's node_limit /
 *     max_seconds (they count per rank), or every rank done with its max_steps (a per-rank quota: a
 *     rank that has done its steps waits in the exchange while the others finish theirs);
 *   - migration: a rank that cannot fill a batch gets half the surplus of the fullest rank (node
 *     records move by ncclSend / ncclRecv; not with cut rounds).
 * A rank without open nodes blocks in the exchange until work or the end arrives.  After the solve
 * mipx_tree_get_stats / mipx_tree_solution report the GLOBAL incumbent on every rank;
 * mipx_tree_global_stats adds the summed counters.
 */
int mipx_tree_set_comm(mipx_tree *t, mipx_comm *c, int every_steps);
/* What every rank concludes from one gathered set of records -- a pure function of the records (no
 * GPU involved), exposed so that the decision logic can be tested across real processes on the CPU.
 * A record is mipx_exchange_record_len(n) doubles: [0] incumbent value, [1] dual bound of the shard,
 * [2] open nodes (queued + in flight), [3] stop flag, [4..7] evaluated / LPs / probes / pivots since
 * sharding, [8] 1 if the rank holds a solution for [0], [9] exchange number, [10] its frontier batch,
 * [11] pool rows it can take from a donor, [12] the lowest bound among its nodes in flight (inf: none;
 * [1] includes it: a shard's dual bound covers queued, in-flight and closed nodes),
 * [16..16+n) the solution, then 4 n pseudo-cost samples (sum_l, sum_r, times_l, times_r).
 * [3] stop flag: 1 a limit that ends the search, 2 the rank's step quota is done, 3 the rank failed
 * (it is returning an error: every other rank stops and returns MIPX_EPEER).
 * reason: 0 go on, 1 no open node anywhere, 2 a rank's limit, 3 global gap <= mip_gap, 4 every rank has
 * done its steps or run dry, 5 a rank failed.
 * moves: n_moves triples (from rank, to rank, node records), in the order they are carried out; a move
 * never exceeds the room [11] its receiver reported. */
typedef struct mipx_exchange_decision {
    double primal, dual, gap;
    int64_t sums[4], open_nodes;
    int32_t incumbent_rank, done, reason, n_moves;
    int32_t moves[3 * 64];
} mipx_exchange_decision;
int mipx_exchange_record_len(int n);
int mipx_exchange_decide(int world, int n, const double *records, double mip_gap, int allow_migration,
                         mipx_exchange_decision *out);
/* Test hook: up to `amount` open nodes travel from this rank to itself through the communicator's
 * point-to-point path (pack kernel -> ncclSend + ncclRecv to the own rank in one ncclGroupStart / End ->
 * unpack kernel; custom transport: a device copy), i.e. what a migration does between two ranks.
 * Returns the number of records moved.  Not with cut rounds, not with a step in flight. */
int64_t mipx_tree_migrate_self(mipx_tree *t, int64_t amount);
/* Test hook: the record this rank would post right now (mipx_exchange_record_len(n) doubles; callable
 * from a step hook, i.e. with steps in flight). */
int mipx_tree_exchange_record(mipx_tree *t, double *record);
typedef struct mipx_tree_global_stats_t {
    double primal_bound, dual_bound, gap;   /* gap: -1 encodes None */
    int64_t evaluated_nodes;                /* ramp-up (replicated: counted once) + every rank's since sharding */
    int64_t lp_solved, probes_solved, pivots, open_nodes;
    int64_t exchanges, nodes_sent, nodes_received;   /* of this rank */
    int32_t world, incumbent_rank;          /* rank whose solution everybody holds (-1: none) */
} mipx_tree_global_stats_t;
int mipx_tree_global_stats(mipx_tree *t, mipx_tree_global_stats_t *out);

/* Device time of the LP kernel launch inside the last mipx_lp_solve_multi call on this context (HIP events on
 * its stream, ms; -1 if none): the rate with the inputs resident in HBM, beside the call's wall time which
 * includes the PCIe copies of its host buffers. */
int mipx_last_kernel_ms(mipx_ctx *ctx, float *ms);
/* Name of the kernel instantiation that (m, n) dispatches to, e.g. "lp_dual_simplex<7,5,16>". */
int mipx_kernel_name(int m, int n, char *buf, size_t buflen);

#ifdef __cplusplus
}
#endif
#endif /* MIPX_H */
