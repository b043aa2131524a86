/*
 * mipx_oracle.c -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE).
 *
 * A plain-C restatement of the per-node hot path of spkelle2/simple_mip_solver:
 *   - the node LP relaxation that the reference delegates to Clp through
 *     `self.lp.dual()` (simple_mip_solver/nodes/base_node.py:259-286, :273) and the
 *     truncated strong-branching solves (base_node.py:629-647),
 *   - most-fractional index selection (base_node.py:544-562),
 *   - pseudo-cost bookkeeping and scoring (nodes/branch/pseudo_cost.py:68-133),
 *   - tableau / Gomory mixed-integer cuts (base_node.py:468-530),
 *   - numerically safe cut rounding (utils/floating_point.py:11-167),
 *   - cut selection (base_node.py:387-466).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library.  The product (simple_mip_solver_amd/) never links, imports or calls it.
 *
 * PARITY STATUS.  The simplex arithmetic of the reference lives in third-party COIN-OR
 * Clp reached via cylp (un-vendored, version unpinned: environment.yml:7,15-16), which is
 * absent from this image.  The LP part of this oracle is therefore a restatement of the
 * textbook bounded dual simplex that `lp.dual()` stands for, pinned by the reference's own
 * known-answer tests at the Clp boundary (test_base_node.py:394-437, :681-684, :711-755,
 * test_branch_and_bound.py:48-322; see tests/test_oracle_known_answers.py) and by
 * independent HiGHS optima (tests/golden/example_models_optima.json).  The pure-Python
 * arithmetic (Gomory, safe cuts, selection, branching, pseudo-costs) is pinned by golden
 * vectors generated from the reference's own code (tests/golden/make_golden.py).
 *
 * CANONICAL ALGORITHM ("mipx dual simplex", also followed bit-for-bit by the HIP kernel):
 *   variables 0..n-1 structural (l <= x <= u, u may be +inf), n..n+m-1 slacks s = Ax - b >= 0.
 *   condensed tableau  x_B + T x_N = beta0,  reduced costs d_N, carried through pivots.
 *   0. T = -A, beta0 = -b, d = c, basis = all slacks.
 *   1. refactor: every structural marked basic in the warm-start status is pivoted in
 *      (increasing index) on the row of largest |T_iq| among rows whose basic slack is marked
 *      nonbasic (fallback: any slack row); ties -> lowest row.
 *   2. nonbasic bound choice: d_j < -DTOL -> upper (a symbolic "fake" upper M if u = inf),
 *      d_j > DTOL -> lower, else keep the warm-start side.  Basic values are kept as
 *      two-component numbers a + b*M; beta = beta0 - T v_N with a fold-in-half summation tree.
 *   3. dual simplex: leaving row = largest bound violation (M-level beats real level, ties ->
 *      lowest variable index); Harris two-pass ratio test (ties -> lowest variable index);
 *      after more than m+n consecutive degenerate steps (entering d_j <= DTOL) Bland's rule takes
 *      over until a non-degenerate step (anti-cycling);
 *      rank-1 tableau update with explicit fma; the pivot row is scaled by the reciprocal
 *      1/p (one division per pivot), as the kernel does.
 *   3'. pricing (which violated row leaves).  Pricing 0 (the HBM-streaming kernel K1b): the largest
 *      violation, as above.  Pricing 1 (the register-tile kernel K1: every shape with m <= 192 and
 *      n <= 256): dual steepest edge -- the largest viol^2 / w_i, w_i the squared norm of row i of
 *      the full tableau [I | T] (>= 1).  w is set after step 2 (w_i = 1 + sum_j T_ij^2, fold-in-half),
 *      carried through pivots by the exact recurrence for the row operations of a pivot on (r, q):
 *      tau_i = sum_j T_ij T_rj (before the pivot, fold-in-half), ratio_i = T_iq * (1/p),
 *      w_i <- max(1, fma(ratio_i, fma(ratio_i, w_r, -2 tau_i), w_i)), w_r <- max(1, (w_r (1/p)) (1/p)),
 *      and recomputed from the tableau after every 64th iteration since they were last exact (the
 *      recurrence drifts over long cold solves).  Bland's rule ignores the weights.  The weights
 *      survive an in-place dive (same tableau).  On the node LPs of the 256 x 128 benchmark tree this
 *      takes 20 % fewer iterations than the largest-violation rule, on its root 10 x fewer.
 *   4. status 0 optimal / 1 primal infeasible / 2 unbounded (optimum depends on M) /
 *      3 iteration limit -- the Clp codes the reference reads (base_node.py:274-275,
 *      pseudo_cost.py:86).
 *   Options the frontier engine uses, restated here for the parity tests:
 *   A. anchored refactorisation: step 0 may start from the stored tableau of another basis of the
 *      same rows (one for all: mipx_oracle_set_anchor; or one per node from a table:
 *      mipx_oracle_lp_solve_dive_batch) instead of the slack basis; step 1 then pivots in every
 *      variable wanted basic that is nonbasic there (ascending), on the row of largest |T_iq| among
 *      rows whose variable is not wanted basic (fallback: wanted, not pivoted in by this
 *      refactorisation).
 *   B. in-place dive (mipx_oracle_lp_solve_dive): after an optimal, fractional node LP below the
 *      cutoff the branching rule (most fractional, base_node.py:544-562; or pseudo costs,
 *      pseudo_cost.py:118-133, only if every fractional variable has an entry; ties -> earliest
 *      in the integer list) picks a variable; if it is basic one of its bounds moves (the child of
 *      base_node.py:592-608: left u_j = floor(x_j) if cost_l (x_j - floor) <= cost_r (ceil - x_j),
 *      else right l_j = ceil(x_j); most fractional: the nearer side, ties left) and step 3
 *      continues on the tableau at hand with fresh iteration / anti-cycling counters.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define MIPX_PTOL 1e-7   /* primal feasibility tolerance (Clp default primalTolerance) */
#define MIPX_DTOL 1e-7   /* dual feasibility tolerance (Clp default dualTolerance)     */
#define MIPX_PIVTOL 1e-9 /* smallest acceptable |pivot|                                 */
#define MIPX_BTOL 1e-9   /* zero test on the M component                                */
#define MIPX_MREPORT 1e10 /* value substituted for the symbolic bound M when reporting x of an
                             unbounded LP (Clp reports its artificial dual bound likewise) */

#define ST_BASIC 1
#define ST_UPPER 2
#define ST_LOWER 3

/* optional dump of the final simplex state (set by tests through mipx_oracle_set_dump) */
/* The M component of a value a + b M is a combination of tableau entries with 0/1 weights: what is left of it
 * below MIPX_BTOL after an update is rounding noise, and is taken out at once -- left in, it piles up over
 * thousands of pivots until the zero tests see a symbolic violation that is not there (and the report would
 * multiply it by MIPX_MREPORT).  A no-op on an LP with finite bounds: every b is exactly 0 there. */
static inline double snap_m(double v) { return fabs(v) <= 1e-9 ? 0.0 : v; }
static double *g_dump_T = 0, *g_dump_vec = 0;
static int32_t *g_dump_idx = 0;
void mipx_oracle_set_dump(double *T, double *vec, int32_t *idx) {
    g_dump_T = T; g_dump_vec = vec; g_dump_idx = idx;
}

/* optional anchor: a tableau state (T, d, beta0, nvar, bvar -- the layout of the dump) of some
 * basis of the same rows from which the warm-start refactorisation starts instead of the slack
 * basis; set by tests through mipx_oracle_set_anchor, cleared with NULLs */
static const double *g_anchor_T = 0, *g_anchor_vec = 0;
static const int32_t *g_anchor_idx = 0;
static int g_refactor_only = 0;
/* pricing rule of the node LPs: 0 largest violation, 1 dual steepest edge, 2 dual Devex; -1 = what the GPU
 * path runs for the shape (1 where the register-tile kernel takes it: m <= 192 and n <= 256 with the tile
 * table of csrc/mipx.hip; else 2: the HBM-streaming kernel cannot afford the extra tableau pass the
 * steepest-edge inner products would take, Devex weights need the pivot column only) */
static int g_pricing = -1;
void mipx_oracle_set_pricing(int pricing) { g_pricing = pricing; }
static int pricing_for(int m, int n) {
    if (g_pricing >= 0) return g_pricing;
    return ((m <= 32 && n <= 64) || (m <= 64 && n <= 128) || (m <= 128 && n <= 256) || (m <= 192 && n <= 256)) ? 1 : 2;
}
#define MIPX_DSE_REFRESH 64
void mipx_oracle_set_anchor(const double *T, const double *vec, const int32_t *idx) {
    g_anchor_T = T; g_anchor_vec = vec; g_anchor_idx = idx;
}
void mipx_oracle_set_refactor_only(int on) { g_refactor_only = on; }

static int next_pow2(int n) { int p = 1; while (p < n) p <<= 1; return p; }

/* fold-in-half summation tree over a power-of-two length buffer (destroys p) */
static double fold_sum(double *p, int n2) {
    for (int h = n2 / 2; h >= 1; h /= 2)
        for (int j = 0; j < h; j++) p[j] = p[j] + p[j + h];
    return p[0];
}

typedef struct {
    int m, n;
    double *T;      /* m x n */
    double *beta0;  /* m */
    double *d;      /* n */
    int *bvar;      /* m : variable basic in row i */
    int *nvar;      /* n : variable nonbasic in column j */
} tab_t;

/* condensed-tableau pivot on (r,q); also transforms beta0 and d */
static void tab_pivot(tab_t *t, int r, int q) {
    const int m = t->m, n = t->n;
    double *T = t->T;
    const double p = T[(size_t)r * n + q];
    const double pinv = 1.0 / p;
    /* rho_j = T_rj * (1/p), alpha_i = T_iq */
    double *rho = (double *)malloc(sizeof(double) * (size_t)(n + 1));
    for (int j = 0; j < n; j++) rho[j] = T[(size_t)r * n + j] * pinv;
    rho[n] = t->beta0[r] * pinv;
    for (int i = 0; i < m; i++) {
        if (i == r) continue;
        const double a = T[(size_t)i * n + q];
        double *Ti = T + (size_t)i * n;
        for (int j = 0; j < n; j++)
            if (j != q) Ti[j] = fma(-a, rho[j], Ti[j]);
        Ti[q] = -a * pinv;
        t->beta0[i] = fma(-a, rho[n], t->beta0[i]);
    }
    {
        const double a = t->d[q];
        for (int j = 0; j < n; j++)
            if (j != q) t->d[j] = fma(-a, rho[j], t->d[j]);
        t->d[q] = -a * pinv;
    }
    for (int j = 0; j < n; j++) T[(size_t)r * n + j] = rho[j];
    T[(size_t)r * n + q] = pinv;
    t->beta0[r] = rho[n];
    int tmp = t->bvar[r]; t->bvar[r] = t->nvar[q]; t->nvar[q] = tmp;
    free(rho);
}

/*
 * Solve  min c'x  s.t.  A x >= b,  l <= x <= u   (A row-major m x n, u may be +inf, l finite).
 * vstat_in: n+m Clp status codes (1 basic, 2 at upper, 3 at lower, anything else = at lower)
 *           or NULL for a cold start from the slack basis.
 * max_iter <= 0: no limit other than the internal cap.
 * Outputs: status (0/1/2/3), obj (c.x; +inf if infeasible; M := 1e10 if unbounded), x[n],
 *          y[m] row duals, dj[n] reduced costs of structurals (0 for basic), vstat_out[n+m],
 *          iters (dual simplex iterations), npivots (refactor + simplex pivots).
 * Any output pointer may be NULL.
 */
/* In-place dive of the frontier engine (lp_kernel.hip.h, LpArgs::dive): when the node LP ends
 * optimal, fractional and below the cutoff, and the branching rule (K4's: most fractional, or
 * pseudo costs with every fractional variable initialised) picks a basic variable, one of its
 * bounds moves (towards the side the rule expects to cost less) and the dual simplex goes on from
 * the tableau at hand; the child's results go to the second set of outputs.  dive_var = -1: no dive. */
typedef struct {
    int32_t rule, n_int;
    const int32_t *int_idx;
    const double *cost_l, *cost_r;
    const uint8_t *has_entry;
    double cutoff;
    int32_t *status; double *obj; double *x; int8_t *vstat; int32_t *iters; int32_t *npivots;
    int32_t *dive_var, *dive_dir; double *dive_val;
    /* multi-level dive: up to `depth` children in a row on the same tableau (0 reads as 1).  Level p
     * (1-based) writes its outputs level_stride nodes after level p - 1: status / obj / iters / npivots
     * at [+ (p-1) * level_stride], x at [+ (p-1) * level_stride * n], vstat likewise with n + m; the
     * branching decision taken after level p's LP goes to dive_var/dir/val[p * level_stride]. */
    int32_t depth;
    int64_t level_stride;
} mipx_dive_t;

static int lp_solve_impl(int m, int n, const double *A, const double *b, const double *c,
                         const double *l_in, const double *u_in, const int8_t *vstat_in, int max_iter,
                         int32_t *status_out, double *obj_out, double *x_out, double *y_out,
                         double *dj_out, int8_t *vstat_out, int32_t *iters_out,
                         int32_t *npivots_out, const mipx_dive_t *dv, const double *aT,
                         const double *avec, const int32_t *aidx) {
    if (m < 0 || n <= 0) return -1;
    if (!aT) { aT = g_anchor_T; avec = g_anchor_vec; aidx = g_anchor_idx; }  /* (set_anchor: one for all) */
    /* the dive moves a bound: work on copies */
    double *l = (double *)malloc(sizeof(double) * (size_t)n);
    double *u = (double *)malloc(sizeof(double) * (size_t)n);
    memcpy(l, l_in, sizeof(double) * (size_t)n);
    memcpy(u, u_in, sizeof(double) * (size_t)n);
    int pass = 0;   /* dive level being solved: 0 the node, p its p-th child in a row */
    if (dv && dv->dive_var)
        for (int p = 0; p < (dv->depth > 0 ? dv->depth : 1); p++) dv->dive_var[(size_t)p * (size_t)dv->level_stride] = -1;
    tab_t t;
    t.m = m; t.n = n;
    t.T = (double *)malloc(sizeof(double) * (size_t)(m > 0 ? m : 1) * n);
    t.beta0 = (double *)malloc(sizeof(double) * (size_t)(m + 1));
    t.d = (double *)malloc(sizeof(double) * (size_t)n);
    t.bvar = (int *)malloc(sizeof(int) * (size_t)(m + 1));
    t.nvar = (int *)malloc(sizeof(int) * (size_t)n);
    const int nv = n + m;
    int8_t *atup = (int8_t *)calloc((size_t)nv, 1);       /* warm-start side */
    int8_t *wantb = (int8_t *)calloc((size_t)nv, 1);      /* warm-start basic flag */
    int8_t *nb_up = (int8_t *)calloc((size_t)n, 1);       /* per column: 0 lower, 1 upper, 2 fake upper */
    double *ba = (double *)malloc(sizeof(double) * (size_t)(m + 1));
    double *bb = (double *)malloc(sizeof(double) * (size_t)(m + 1));
    const int n2 = next_pow2(n);
    double *buf = (double *)malloc(sizeof(double) * (size_t)n2);
    double *va = (double *)malloc(sizeof(double) * (size_t)n);
    double *vb = (double *)malloc(sizeof(double) * (size_t)n);

    if (aT && vstat_in) {
        memcpy(t.T, aT, sizeof(double) * (size_t)m * n);
        for (int j = 0; j < n; j++) { t.d[j] = avec[j]; t.nvar[j] = aidx[j]; }
        for (int i = 0; i < m; i++) { t.beta0[i] = avec[n + i]; t.bvar[i] = aidx[n + i]; }
    } else {
        for (int i = 0; i < m; i++) {
            for (int j = 0; j < n; j++) t.T[(size_t)i * n + j] = -A[(size_t)i * n + j];
            t.beta0[i] = -b[i];
            t.bvar[i] = n + i;
        }
        for (int j = 0; j < n; j++) { t.d[j] = c[j]; t.nvar[j] = j; }
    }
    int npiv = 0;

    /* 1. refactor to the warm-start basis */
    if (vstat_in) {
        for (int v = 0; v < nv; v++) {
            wantb[v] = (vstat_in[v] == ST_BASIC);
            atup[v] = (vstat_in[v] == ST_UPPER);
        }
        /* every variable wanted basic that is nonbasic in the starting tableau enters, in
         * ascending variable order; it replaces the basic variable of the row with the largest
         * |T_iq| among rows whose variable is not wanted basic (fallback: wanted but not entered by
         * this refactorisation); ties -> lowest row */
        int *pos = (int *)malloc(sizeof(int) * (size_t)nv);
        int8_t *entered = (int8_t *)calloc((size_t)(m + 1), 1);
        for (int v = 0; v < nv; v++) pos[v] = -1;
        for (int j = 0; j < n; j++) pos[t.nvar[j]] = j;
        for (int v = 0; v < nv; v++) {
            if (!wantb[v] || pos[v] < 0) continue;
            const int q = pos[v];
            int best = -1; double bestv = MIPX_PIVTOL;
            for (int i = 0; i < m; i++) {
                if (wantb[t.bvar[i]]) continue;
                double a = fabs(t.T[(size_t)i * n + q]);
                if (a > bestv) { bestv = a; best = i; }
            }
            if (best < 0) {
                for (int i = 0; i < m; i++) {
                    if (!wantb[t.bvar[i]] || entered[i]) continue;
                    double a = fabs(t.T[(size_t)i * n + q]);
                    if (a > bestv) { bestv = a; best = i; }
                }
            }
            if (best < 0) continue; /* singular: stays nonbasic */
            tab_pivot(&t, best, q);
            entered[best] = 1;
            npiv++;
        }
        free(pos); free(entered);
    }

#define VLO(v) ((v) < n ? l[(v)] : 0.0)
#define VUP(v) ((v) < n ? u[(v)] : INFINITY)

    /* 2. nonbasic sides and basic values */
    for (int j = 0; j < n; j++) {
        int v = t.nvar[j];
        double lo = VLO(v), up = VUP(v);
        int side;
        if (lo == up) side = 0;
        else if (t.d[j] < -MIPX_DTOL) side = isinf(up) ? 2 : 1;
        else if (t.d[j] > MIPX_DTOL) side = 0;
        else side = (atup[v] && !isinf(up)) ? 1 : 0;
        nb_up[j] = (int8_t)side;
        va[j] = side == 0 ? lo : side == 1 ? up : 0.0;
        vb[j] = side == 2 ? 1.0 : 0.0;
    }
    for (int i = 0; i < m; i++) {
        const double *Ti = t.T + (size_t)i * n;
        for (int j = 0; j < n; j++) buf[j] = Ti[j] * va[j];
        for (int j = n; j < n2; j++) buf[j] = 0.0;
        ba[i] = t.beta0[i] - fold_sum(buf, n2);
        for (int j = 0; j < n; j++) buf[j] = Ti[j] * vb[j];
        for (int j = n; j < n2; j++) buf[j] = 0.0;
        bb[i] = snap_m(0.0 - fold_sum(buf, n2));
    }

    /* 3'. dual steepest edge weights (pricing 1): squared norms of the rows of [I | T] */
    const int dse = pricing_for(m, n) == 1;
    const int devex = pricing_for(m, n) == 2;
    double *wgt = (double *)malloc(sizeof(double) * (size_t)(m + 1));
    double *tau = (double *)malloc(sizeof(double) * (size_t)(m + 1));
    int wage = 0; /* iterations since the weights were exact */
    /* 3''. Devex (pricing 2): reference weights 1 at the start of a node LP (a dive goes on with them) */
    if (devex) for (int i = 0; i < m; i++) wgt[i] = 1.0;
    if (dse && !g_refactor_only) {
        for (int i = 0; i < m; i++) {
            const double *Ti = t.T + (size_t)i * n;
            for (int j = 0; j < n; j++) buf[j] = Ti[j] * Ti[j];
            for (int j = n; j < n2; j++) buf[j] = 0.0;
            wgt[i] = 1.0 + fold_sum(buf, n2);
        }
    }

#define VALUES_FROM_TABLEAU() do { \
    for (int i_ = 0; i_ < m; i_++) { \
        const double *Ti_ = t.T + (size_t)i_ * n; \
        for (int j = 0; j < n; j++) buf[j] = Ti_[j] * va[j]; \
        for (int j = n; j < n2; j++) buf[j] = 0.0; \
        ba[i_] = t.beta0[i_] - fold_sum(buf, n2); \
        for (int j = 0; j < n; j++) buf[j] = Ti_[j] * vb[j]; \
        for (int j = n; j < n2; j++) buf[j] = 0.0; \
        bb[i_] = snap_m(0.0 - fold_sum(buf, n2)); \
    } } while (0)
    /* Above the register tiles (the shapes of K1b / K1c: long cold solves) a verdict -- optimal, unbounded,
     * infeasible -- of a solve that has carried symbolic values is only taken on values worked out afresh from
     * the tableau: the running updates of thousands of pivots are not trusted with it.  (An LP with finite
     * bounds never carries one: nothing changes for it.) */
    const int refresh = devex;
    int fresh = 1, sym = 0;
    for (int j = 0; j < n; j++) if (vb[j] != 0.0) sym = 1;
    /* 3. dual simplex */
    int iters = 0, status = -1;
    int degen = 0; /* consecutive degenerate steps; > m+n switches to Bland's rule (anti-cycling) */
    const int cap = 100 * (m + n) + 1000;
    if (g_refactor_only) { status = 3; goto done; }
next_pass:
    iters = 0; status = -1; degen = 0;
    for (;;) {
        const int bland = degen > m + n;
        /* (a) leaving row */
        int r = -1, rlevel = 0, rvar = 0, sigma = 0; double rviol = 0.0;
        for (int i = 0; i < m; i++) {
            int v = t.bvar[i];
            double lo = VLO(v), up = VUP(v);
            double a = ba[i], bM = bb[i];
            int level = 0, sg = 0; double viol = 0.0;
            if (bM < -MIPX_BTOL) { level = 2; viol = -bM; sg = +1; }
            else if (bM > MIPX_BTOL) {
                if (!isinf(up)) { level = 2; viol = bM; sg = -1; }
                else if (bM > 1.0 + MIPX_BTOL) { level = 2; viol = bM - 1.0; sg = -1; }
                else if (bM >= 1.0 - MIPX_BTOL && a > MIPX_PTOL) { level = 1; viol = a; sg = -1; }
            } else {
                if (a < lo - MIPX_PTOL) { level = 1; viol = lo - a; sg = +1; }
                else if (!isinf(up) && a > up + MIPX_PTOL) { level = 1; viol = a - up; sg = -1; }
            }
            if (level == 0) continue;
            if (dse || devex) viol = viol * viol / wgt[i];
            if (bland) { level = 1; viol = 0.0; } /* Bland: lowest variable index among violated */
            int better = 0;
            if (r < 0) better = 1;
            else if (level != rlevel) better = level > rlevel;
            else if (viol != rviol) better = viol > rviol;
            else better = v < rvar;
            if (better) { r = i; rlevel = level; rviol = viol; rvar = v; sigma = sg; }
        }
        if (r < 0 && refresh && sym && !fresh) { VALUES_FROM_TABLEAU(); fresh = 1; continue; }
        if (r < 0) {
            status = 0;
            for (int i = 0; i < m; i++) if (bb[i] > MIPX_BTOL) status = 2;
            for (int j = 0; j < n; j++) if (nb_up[j] == 2) status = 2;
            break;
        }
        if ((max_iter > 0 && iters >= max_iter) || iters >= cap) { status = 3; break; }
        /* (b) ratio test, Harris two pass */
        const double *Tr = t.T + (size_t)r * n;
        double thmax = INFINITY; int jmin = -1, jminvar = 0;
        for (int j = 0; j < n; j++) {
            int v = t.nvar[j];
            if (VLO(v) == VUP(v)) continue;
            double a = sigma * Tr[j];
            int elig = nb_up[j] == 0 ? (a < -MIPX_PIVTOL) : (a > MIPX_PIVTOL);
            if (!elig) continue;
            double dj = nb_up[j] == 0 ? fmax(t.d[j], 0.0) : fmax(-t.d[j], 0.0);
            double ratio = bland ? dj / fabs(a) : (dj + MIPX_DTOL) / fabs(a);
            if (jmin < 0 || ratio < thmax || (ratio == thmax && v < jminvar)) {
                thmax = ratio; jmin = j; jminvar = v;
            }
        }
        if (jmin < 0 && refresh && sym && !fresh) { VALUES_FROM_TABLEAU(); fresh = 1; continue; }
        if (jmin < 0) { status = 1; break; }
        /* pass 2: largest |a| among columns with dj <= thmax*|a| (the pass-1 argmin always
         * qualifies); ties -> lowest variable index */
        int q = -1, qvar = 0; double qabs = 0.0;
        for (int j = 0; j < n && !bland; j++) {
            int v = t.nvar[j];
            if (VLO(v) == VUP(v)) continue;
            double a = sigma * Tr[j];
            int elig = nb_up[j] == 0 ? (a < -MIPX_PIVTOL) : (a > MIPX_PIVTOL);
            if (!elig) continue;
            double dj = nb_up[j] == 0 ? fmax(t.d[j], 0.0) : fmax(-t.d[j], 0.0);
            double aa = fabs(a);
            if (j != jmin && dj > thmax * aa) continue;
            if (q < 0 || aa > qabs || (aa == qabs && v < qvar)) { q = j; qabs = aa; qvar = v; }
        }
        if (bland) q = jmin; /* textbook ratio test, ties -> lowest variable index */
        {
            const double djq = nb_up[q] == 0 ? fmax(t.d[q], 0.0) : fmax(-t.d[q], 0.0);
            degen = djq <= MIPX_DTOL ? degen + 1 : 0;
        }
        /* (c) value update + pivot */
        {
            int lv = t.bvar[r];
            double lo = VLO(lv), up = VUP(lv);
            double la, lb; int newside;
            if (sigma > 0) { la = lo; lb = 0.0; newside = 0; }
            else if (!isinf(up)) { la = up; lb = 0.0; newside = 1; }
            else { la = 0.0; lb = 1.0; newside = 2; }
            const double p = Tr[q];
            const double pinv = 1.0 / p;
            if (devex) { /* w_i <- max(w_i, (alpha_i / p)^2 w_r), w_r <- max(w_r / p^2, 1) */
                const double wr = wgt[r];
                for (int i = 0; i < m; i++) {
                    if (i == r) continue;
                    const double ratio = t.T[(size_t)i * n + q] * pinv;
                    const double w = (ratio * ratio) * wr;
                    wgt[i] = w > wgt[i] ? w : wgt[i];
                }
                const double w = (wr * pinv) * pinv;
                wgt[r] = w < 1.0 ? 1.0 : w;
            }
            if (dse) { /* the weights after the row operations of this pivot */
                for (int i = 0; i < m; i++) {
                    const double *Ti = t.T + (size_t)i * n;
                    for (int j = 0; j < n; j++) buf[j] = Ti[j] * Tr[j];
                    for (int j = n; j < n2; j++) buf[j] = 0.0;
                    tau[i] = fold_sum(buf, n2);
                }
                const double wr = wgt[r];
                for (int i = 0; i < m; i++) {
                    if (i == r) continue;
                    const double ratio = t.T[(size_t)i * n + q] * pinv;
                    const double w = fma(ratio, fma(ratio, wr, -2.0 * tau[i]), wgt[i]);
                    wgt[i] = w < 1.0 ? 1.0 : w;
                }
                const double w = (wr * pinv) * pinv;
                wgt[r] = w < 1.0 ? 1.0 : w;
            }
            const double ta = (ba[r] - la) * pinv, tb = (bb[r] - lb) * pinv;
            for (int i = 0; i < m; i++) {
                if (i == r) continue;
                double al = t.T[(size_t)i * n + q];
                ba[i] = fma(-al, ta, ba[i]);
                bb[i] = snap_m(fma(-al, tb, bb[i]));
            }
            ba[r] = va[q] + ta;
            bb[r] = snap_m(vb[q] + tb);
            tab_pivot(&t, r, q);
            nb_up[q] = (int8_t)newside;
            va[q] = la; vb[q] = lb;
            iters++; npiv++; fresh = 0;
            if (lb != 0.0) sym = 1;
            if (dse && ++wage == MIPX_DSE_REFRESH) { /* exact again, from the tableau after the pivot */
                wage = 0;
                for (int i = 0; i < m; i++) {
                    const double *Ti = t.T + (size_t)i * n;
                    for (int j = 0; j < n; j++) buf[j] = Ti[j] * Ti[j];
                    for (int j = n; j < n2; j++) buf[j] = 0.0;
                    wgt[i] = 1.0 + fold_sum(buf, n2);
                }
            }
        }
    }

done:
    if (g_dump_T) {
        memcpy(g_dump_T, t.T, sizeof(double) * (size_t)m * n);
        for (int j = 0; j < n; j++) {
            g_dump_vec[j] = t.d[j]; g_dump_idx[j] = t.nvar[j]; g_dump_idx[n + m + j] = nb_up[j];
        }
        for (int i = 0; i < m; i++) {
            g_dump_vec[n + i] = t.beta0[i]; g_dump_vec[n + m + i] = ba[i];
            g_dump_vec[n + 2 * m + i] = bb[i]; g_dump_idx[n + i] = t.bvar[i];
        }
    }
    /* 4. outputs */
    double dive_obj = INFINITY;
    double *dive_x = NULL;
    if (x_out || obj_out || dv) {
        double *x = (double *)malloc(sizeof(double) * (size_t)n);
        for (int j = 0; j < n; j++) {
            int v = t.nvar[j];
            if (v < n) x[v] = nb_up[j] == 2 ? MIPX_MREPORT : va[j];
        }
        for (int i = 0; i < m; i++) {
            int v = t.bvar[i];
            if (v < n) x[v] = fma(snap_m(bb[i]), MIPX_MREPORT, ba[i]);
        }
        if (status == 1) dive_obj = INFINITY;
        else {
            for (int j = 0; j < n; j++) buf[j] = c[j] * x[j];
            for (int j = n; j < n2; j++) buf[j] = 0.0;
            dive_obj = fold_sum(buf, n2);
        }
        if (obj_out) *obj_out = dive_obj;
        if (x_out) memcpy(x_out, x, sizeof(double) * (size_t)n);
        dive_x = x;
    }
    if (y_out) {
        for (int i = 0; i < m; i++) y_out[i] = 0.0;
        for (int j = 0; j < n; j++) if (t.nvar[j] >= n) y_out[t.nvar[j] - n] = t.d[j];
    }
    if (dj_out) {
        for (int j = 0; j < n; j++) dj_out[j] = 0.0;
        for (int j = 0; j < n; j++) if (t.nvar[j] < n) dj_out[t.nvar[j]] = t.d[j];
    }
    if (vstat_out) {
        for (int i = 0; i < m; i++) vstat_out[t.bvar[i]] = ST_BASIC;
        for (int j = 0; j < n; j++) vstat_out[t.nvar[j]] = nb_up[j] ? ST_UPPER : ST_LOWER;
    }
    if (status_out) *status_out = status;
    if (iters_out) *iters_out = iters;
    if (npivots_out) *npivots_out = npiv;

    /* 5. dive: K4's rule on x, then one bound of the chosen (basic) variable moves */
    if (dv && pass < (dv->depth > 0 ? dv->depth : 1) && !g_refactor_only && status == 0 && dive_obj < dv->cutoff) {
        const double *x = dive_x;
        int win = -1, need_probe = 0; double bk = -INFINITY;
        for (int k = 0; k < dv->n_int; k++) {
            const int i = dv->int_idx[k];
            const double v = x[i], fl = floor(v), ce = ceil(v);
            const double dist = fmin(v - fl, ce - v);
            if (!(dist > 1e-4)) continue; /* variable_epsilon */
            double key;
            if (dv->rule == 0) key = dist;
            else if (dv->has_entry[i]) key = fmin(dv->cost_r[i] * (ce - v), dv->cost_l[i] * (v - fl));
            else { need_probe = 1; continue; }
            if (win < 0 || key > bk) { win = k; bk = key; } /* ties -> earliest in int_idx */
        }
        if (win >= 0 && !need_probe) {
            const int var = dv->int_idx[win];
            const double v = x[var], fl = floor(v), ce = ceil(v);
            int dir;
            if (dv->rule == 0) dir = (v - fl <= ce - v) ? 0 : 1;
            else dir = (dv->cost_l[var] * (v - fl) <= dv->cost_r[var] * (ce - v)) ? 0 : 1;
            int basic = 0;
            for (int i = 0; i < m; i++) if (t.bvar[i] == var) basic = 1;
            if (basic) {
                const size_t ls = (size_t)dv->level_stride, lv = (size_t)pass;   /* this decision: level `pass` */
                if (dir == 0) u[var] = fl; else l[var] = ce;
                if (dv->dive_var) dv->dive_var[lv * ls] = var;
                if (dv->dive_dir) dv->dive_dir[lv * ls] = dir;
                if (dv->dive_val) dv->dive_val[lv * ls] = v;
                pass++;
                status_out = dv->status + lv * ls; obj_out = dv->obj ? dv->obj + lv * ls : NULL;
                x_out = dv->x ? dv->x + lv * ls * (size_t)n : NULL;
                vstat_out = dv->vstat ? dv->vstat + lv * ls * (size_t)(n + m) : NULL;
                iters_out = dv->iters ? dv->iters + lv * ls : NULL;
                npivots_out = dv->npivots ? dv->npivots + lv * ls : NULL;
                y_out = NULL; dj_out = NULL;
                npiv = 0;
                free(dive_x);
                goto next_pass;
            }
        }
    }
    free(dive_x);
    free(l); free(u);

    free(t.T); free(t.beta0); free(t.d); free(t.bvar); free(t.nvar);
    free(atup); free(wantb); free(nb_up); free(ba); free(bb); free(buf); free(va); free(vb);
    free(wgt); free(tau);
    return 0;
}

int mipx_oracle_lp_solve(int m, int n, const double *A, const double *b, const double *c,
                         const double *l, const double *u, const int8_t *vstat_in, int max_iter,
                         int32_t *status_out, double *obj_out, double *x_out, double *y_out,
                         double *dj_out, int8_t *vstat_out, int32_t *iters_out,
                         int32_t *npivots_out) {
    return lp_solve_impl(m, n, A, b, c, l, u, vstat_in, max_iter, status_out, obj_out, x_out, y_out, dj_out,
                         vstat_out, iters_out, npivots_out, NULL, NULL, NULL, NULL);
}

/* one node LP with the in-place dive (outputs of the child through dv) */
int mipx_oracle_lp_solve_dive(int m, int n, const double *A, const double *b, const double *c,
                              const double *l, const double *u, const int8_t *vstat_in, int max_iter,
                              int32_t *status_out, double *obj_out, double *x_out,
                              int8_t *vstat_out, int32_t *iters_out, int32_t *npivots_out,
                              const mipx_dive_t *dv) {
    return lp_solve_impl(m, n, A, b, c, l, u, vstat_in, max_iter, status_out, obj_out, x_out, NULL, NULL,
                         vstat_out, iters_out, npivots_out, dv, NULL, NULL, NULL);
}

/* batch with the dive: every output array has (depth + 1) * batch rows (the nodes, then their first
 * dive children, then the second ... ; the caller presets status[batch..] = -1), dive_var / dir / val
 * depth * batch entries (the decision after level p at [p * batch + k]) */
int mipx_oracle_lp_solve_dive_batch(int m, int n, const double *A, const double *b, const double *c,
                                    int batch, const double *l, const double *u,
                                    const int8_t *vstat_in, int max_iter, int rule, int n_int,
                                    const int32_t *int_idx, const double *cost_l, const double *cost_r,
                                    const uint8_t *has_entry, double cutoff, int32_t *status,
                                    double *obj, double *x, int8_t *vstat_out, int32_t *iters,
                                    int32_t *npivots, int32_t *dive_var, int32_t *dive_dir,
                                    double *dive_val, const double *atab_T, const double *atab_vec,
                                    const int32_t *atab_idx, const int32_t *anchor_sel, int depth) {
    /* optional anchor table (mipx_tree_reanchor): node k starts from entry anchor_sel[k], entries
     * m*n / n+3m / 2n+m apart; -1 or no table: the anchor set by mipx_oracle_set_anchor */
    const size_t nv = (size_t)n + m;
    for (int k = 0; k < batch; k++) {
        const size_t ck = (size_t)batch + k;
        mipx_dive_t dv;
        dv.rule = rule; dv.n_int = n_int; dv.int_idx = int_idx;
        dv.cost_l = cost_l; dv.cost_r = cost_r; dv.has_entry = has_entry; dv.cutoff = cutoff;
        dv.status = status + ck; dv.obj = obj ? obj + ck : NULL; dv.x = x ? x + ck * n : NULL;
        dv.vstat = vstat_out ? vstat_out + ck * nv : NULL;
        dv.iters = iters ? iters + ck : NULL; dv.npivots = npivots ? npivots + ck : NULL;
        dv.dive_var = dive_var + k; dv.dive_dir = dive_dir + k; dv.dive_val = dive_val + k;
        dv.depth = depth > 0 ? depth : 1; dv.level_stride = batch;
        int rc = lp_solve_impl(m, n, A, b, c, l + (size_t)k * n, u + (size_t)k * n,
                               vstat_in ? vstat_in + (size_t)k * nv : NULL, max_iter, status + k,
                               obj ? obj + k : NULL, x ? x + (size_t)k * n : NULL, NULL, NULL,
                               vstat_out ? vstat_out + (size_t)k * nv : NULL, iters ? iters + k : NULL,
                               npivots ? npivots + k : NULL, rule >= 0 ? &dv : NULL,
                               (atab_T && anchor_sel && anchor_sel[k] >= 0) ? atab_T + (size_t)anchor_sel[k] * m * n : NULL,
                               (atab_T && anchor_sel && anchor_sel[k] >= 0) ? atab_vec + (size_t)anchor_sel[k] * (n + 3 * (size_t)m) : NULL,
                               (atab_T && anchor_sel && anchor_sel[k] >= 0) ? atab_idx + (size_t)anchor_sel[k] * (2 * (size_t)n + m) : NULL);
        if (rc) return rc;
    }
    return 0;
}

/* batch of node LPs sharing (A,b,c): l,u are batch x n, vstat batch x (n+m) (or NULL) */
int mipx_oracle_lp_solve_batch(int m, int n, const double *A, const double *b, const double *c,
                               int batch, const double *l, const double *u,
                               const int8_t *vstat_in, int max_iter, int32_t *status, double *obj,
                               double *x, double *y, int8_t *vstat_out, int32_t *iters,
                               int32_t *npivots) {
    for (int k = 0; k < batch; k++) {
        int rc = mipx_oracle_lp_solve(
            m, n, A, b, c, l + (size_t)k * n, u + (size_t)k * n,
            vstat_in ? vstat_in + (size_t)k * (n + m) : NULL, max_iter, status ? status + k : NULL,
            obj ? obj + k : NULL, x ? x + (size_t)k * n : NULL, y ? y + (size_t)k * m : NULL, NULL,
            vstat_out ? vstat_out + (size_t)k * (n + m) : NULL, iters ? iters + k : NULL,
            npivots ? npivots + k : NULL);
        if (rc) return rc;
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * Branching: base_node.py:544-562 (_most_fractional_index), :649-666
 * ---------------------------------------------------------------------------------------- */
#define VARIABLE_EPSILON 1e-4 /* utils/tolerance.py:2 */

/* returns index of most fractional integer variable or -1 (reference: None) */
int mipx_oracle_most_fractional(int n_int, const int32_t *int_idx, const double *x) {
    int furthest = -1; double fd = VARIABLE_EPSILON;
    for (int k = 0; k < n_int; k++) {
        double v = x[int_idx[k]];
        double dist = fmin(v - floor(v), ceil(v) - v);
        if (dist > fd) { fd = dist; furthest = int_idx[k]; }
    }
    return furthest;
}

/* mip_feasible test of base_node.py:281-283 */
int mipx_oracle_mip_feasible(int n_int, const int32_t *int_idx, const double *x) {
    double worst = 0.0;
    for (int k = 0; k < n_int; k++) {
        double v = x[int_idx[k]];
        double e = fabs(rint(v) - v); /* np.round = half-to-even = rint in default mode */
        if (e > worst) worst = e;
    }
    return worst <= VARIABLE_EPSILON;
}

/*
 * pseudo_cost.py:118-133: score_i = min(cost_right*(ceil-x), cost_left*(x-floor)) over fractional
 * integer i; argmax with ties -> earliest in int_idx order (stable descending sort).
 * cost arrays are indexed by variable (length n); returns -1 if no fractional variable.
 */
int mipx_oracle_best_pseudo_cost(int n_int, const int32_t *int_idx, const double *x,
                                 const double *cost_left, const double *cost_right) {
    int best = -1; double bs = 0.0;
    for (int k = 0; k < n_int; k++) {
        int i = int_idx[k];
        double v = x[i];
        double fl = floor(v), ce = ceil(v);
        if (!(fmin(v - fl, ce - v) > VARIABLE_EPSILON)) continue;
        double s = fmin(cost_right[i] * (ce - v), cost_left[i] * (v - fl));
        if (best < 0 || s > bs) { best = i; bs = s; }
    }
    return best;
}

/* pseudo_cost.py:68-100 running mean update; status = LP status of the (probe or own) node */
void mipx_oracle_pseudo_cost_update(double *cost, int32_t *times, int status, double objective,
                                    double dual_bound, double variable_change) {
    if (status == 0 || status == 3) {
        double bc = objective - dual_bound;
        if (bc < 0) bc = 0;
        *cost = (*cost * (double)*times + bc / variable_change) / (double)(*times + 1);
    }
    *times += 1;
}

/* ------------------------------------------------------------------------------------------
 * Cut rounding: utils/floating_point.py (get_fraction :106-167, scale_cut :11-37,
 * numerically_safe_cut :40-103 with make_integer=False, the form used at base_node.py:381).
 * Numerators / denominators are carried as doubles holding integers (exact below 2^53; anything
 * larger has already exceeded max_term and only ends the expansion).
 * ---------------------------------------------------------------------------------------- */
#define EST_NONE 0
#define EST_OVER 1
#define EST_UNDER 2

void mipx_oracle_get_fraction(double x, double max_term, int estimate, double *n_out, double *d_out) {
    if (fabs(x) > max_term) {
        *n_out = estimate == EST_OVER ? ceil(x) : estimate == EST_UNDER ? floor(x) : rint(x);
        *d_out = 1.0;
        return;
    }
    /* convergents h_k / k_k; index 0,1 hold h_{-2}, h_{-1} */
    double hn[72], hd[72];
    hn[0] = 0.0; hd[0] = 1.0; hn[1] = 1.0; hd[1] = 0.0;
    int cnt = 2, exact = 0;
    double value = x;
    for (;;) {
        const double whole = floor(value);
        hn[cnt] = whole * hn[cnt - 1] + hn[cnt - 2];
        hd[cnt] = whole * hd[cnt - 1] + hd[cnt - 2];
        cnt++;
        if (hn[cnt - 1] > max_term || hd[cnt - 1] > max_term || cnt >= 70) break;
        const double rem = value - whole;
        if (rem == 0.0) { exact = 1; break; }
        value = 1.0 / rem;
    }
    const int last = cnt - 3; /* index (0-based, convergent numbering) of the final convergent */
    int pick;
    if (exact) pick = last;
    else {
        const int prev = last - 1;
        if (estimate == EST_NONE) pick = prev;
        else if (estimate == EST_OVER) {
            /* Python: (i-1) % 2 truthy -> prev (also for prev = -1), else prev-1 if >= 0 else ceil */
            if (prev % 2 != 0) pick = prev;
            else if (prev >= 1) pick = prev - 1;
            else { *n_out = ceil(x); *d_out = 1.0; return; }
        } else {
            pick = (prev % 2 == 0) ? prev : prev - 1;
        }
    }
    *n_out = hn[pick + 2];
    *d_out = hd[pick + 2];
}

/* pi (n coefficients), pi0 -> safe_pi, safe_pi0; estimate EST_OVER / EST_UNDER.
 * Returns 0 if pi is all zero (cut returned unchanged), 1 otherwise. */
int mipx_oracle_safe_cut(int n, const double *pi, double pi0, int estimate, double max_term,
                         double *safe_pi, double *safe_pi0) {
    double scale = INFINITY;
    int any = 0;
    for (int j = 0; j < n; j++) {
        if (pi[j] != 0.0) any = 1;
        const double s = fabs(1.0 / pi[j]);
        if (s < scale) scale = s;
    }
    if (!any) {
        for (int j = 0; j < n; j++) safe_pi[j] = pi[j];
        *safe_pi0 = pi0;
        return 0;
    }
    for (int j = 0; j < n; j++) {
        const double coef = pi[j] * scale;
        double nn, dd;
        mipx_oracle_get_fraction(coef, max_term, estimate, &nn, &dd);
        if (coef != 0.0 && fabs(1.0 - ((nn / dd) / coef)) > 1e-2) {
            double n2, d2;
            mipx_oracle_get_fraction(coef, max_term, EST_NONE, &n2, &d2);
            if (fabs(n2 / d2 - coef) < 1e-14) { nn = n2; dd = d2; }
        }
        safe_pi[j] = nn / dd;
    }
    double n0, d0;
    mipx_oracle_get_fraction(pi0 * scale, 1e3 /* the reference passes no max_term here */,
                             estimate == EST_OVER ? EST_UNDER : EST_OVER, &n0, &d0);
    *safe_pi0 = n0 / d0;
    return 1;
}

/* The same with the chosen numerators / denominators reported (num, den: n + 1 entries, the
 * right-hand side last; may be NULL) and the make_integer form of floating_point.py:94-101
 * (np.lcm.reduce over the denominators as int64; coefficients (lcm * n) / d; the right-hand side
 * is scaled by the lcm before it is rounded).  Mirrors mipx_safe_cut_batch of the product. */
static void safe_coef(double coef, double max_term, int estimate, double *nn, double *dd) {
    mipx_oracle_get_fraction(coef, max_term, estimate, nn, dd);
    if (coef != 0.0 && fabs(1.0 - ((*nn / *dd) / coef)) > 1e-2) {
        double n2, d2;
        mipx_oracle_get_fraction(coef, max_term, EST_NONE, &n2, &d2);
        if (fabs(n2 / d2 - coef) < 1e-14) { *nn = n2; *dd = d2; }
    }
}
int mipx_oracle_safe_cut_ex(int n, const double *pi, double pi0, int estimate, double max_term,
                            int make_integer, double *safe_pi, double *safe_pi0, double *num,
                            double *den) {
    double scale = INFINITY;
    int any = 0;
    for (int j = 0; j < n; j++) {
        if (pi[j] != 0.0) any = 1;
        const double s = fabs(1.0 / pi[j]);
        if (s < scale) scale = s;
    }
    if (!any) {
        for (int j = 0; j < n; j++) { safe_pi[j] = pi[j]; if (num) { num[j] = pi[j]; den[j] = 1.0; } }
        *safe_pi0 = pi0;
        if (num) { num[n] = pi0; den[n] = 1.0; }
        return 0;
    }
    long long lcm = 1;
    for (int j = 0; j < n; j++) {
        double nn, dd;
        safe_coef(pi[j] * scale, max_term, estimate, &nn, &dd);
        if (num) { num[j] = nn; den[j] = dd; }
        safe_pi[j] = nn / dd;
        if (make_integer) {
            const long long a = lcm < 0 ? -lcm : lcm, b = (long long)fabs(dd);
            long long x = a, y = b;
            while (y != 0) { const long long t = x % y; x = y; y = t; }
            lcm = x == 0 ? 0 : (a / x) * b;
        }
    }
    if (make_integer) {
        for (int j = 0; j < n; j++) {
            double nn, dd;
            safe_coef(pi[j] * scale, max_term, estimate, &nn, &dd);
            safe_pi[j] = (double)(lcm * (long long)nn) / dd;
        }
    }
    double n0, d0;
    const double s0 = pi0 * scale;
    mipx_oracle_get_fraction(make_integer ? s0 * (double)lcm : s0, 1e3,
                             estimate == EST_OVER ? EST_UNDER : EST_OVER, &n0, &d0);
    *safe_pi0 = n0 / d0;
    if (num) { num[n] = n0; den[n] = d0; }
    return 1;
}

/* ------------------------------------------------------------------------------------------
 * Gomory mixed-integer cuts from the condensed tableau (base_node.py:468-511): for every row whose
 * basic variable is an integer structural with fractional value (f0 in [0.01, 0.99]), in the
 * order of ascending basic variable index (= row order of inv(A_B) [A | -I] in the reference),
 *   integer nonbasic j:    f = a - floor(a);  f <= f0 ? f/f0 : (1-f)/(1-f0)
 *   continuous / slack:    a > 0 ? a/f0 : -a/(1-f0)
 * then the slacks s = Ax - b are substituted out: coefs = pi + A' pi_s (accumulated row by row),
 * rhs = 1 + pi_s . b (fold-in-half tree), and the cut is rounded by mipx_oracle_safe_cut('over').
 * T is the m x n condensed tableau of the basis, bvar/nvar its row/column variables.
 * Returns the number of cuts; row_idx[k] is the rank of the generating basic variable.
 * ---------------------------------------------------------------------------------------- */
int mipx_oracle_gomory(int m, int n, const double *A, const double *b, const double *T,
                       const int32_t *bvar, const int32_t *nvar, const double *x,
                       const uint8_t *is_int, double max_term, int32_t *row_idx, double *pi,
                       double *pi0, double *safe_pi, double *safe_pi0) {
    int *order = (int *)malloc(sizeof(int) * (size_t)(m + 1));
    for (int i = 0; i < m; i++) {
        int rank = 0;
        for (int k = 0; k < m; k++) rank += bvar[k] < bvar[i];
        order[rank] = i;
    }
    double *pv = (double *)malloc(sizeof(double) * (size_t)n);
    double *ps = (double *)malloc(sizeof(double) * (size_t)(m + 1));
    int m2 = 1;
    while (m2 < m) m2 <<= 1;
    double *buf = (double *)malloc(sizeof(double) * (size_t)m2);
    int ncuts = 0;
    for (int rank = 0; rank < m; rank++) {
        const int r = order[rank], v = bvar[r];
        if (v >= n || !is_int[v]) continue;
        const double xv = x[v], fl = floor(xv), ce = ceil(xv);
        if (!(fmin(xv - fl, ce - xv) > VARIABLE_EPSILON)) continue;
        const double f0 = xv - fl;
        if (f0 < 1e-2 || f0 + 1e-2 > 1.0) continue;
        for (int j = 0; j < n; j++) pv[j] = 0.0;
        for (int i = 0; i < m; i++) ps[i] = 0.0;
        for (int j = 0; j < n; j++) {
            const int var = nvar[j];
            const double a = T[(size_t)r * n + j];
            const double cont = a > 0.0 ? a / f0 : -a / (1.0 - f0);
            if (var < n) {
                double val = cont;
                if (is_int[var]) {
                    const double f = a - floor(a);
                    val = f <= f0 ? f / f0 : (1.0 - f) / (1.0 - f0);
                }
                pv[var] = val;
            } else {
                ps[var - n] = cont;
            }
        }
        double *out = pi + (size_t)ncuts * n;
        for (int var = 0; var < n; var++) {
            double acc = 0.0;
            for (int i = 0; i < m; i++) acc = acc + A[(size_t)i * n + var] * ps[i];
            out[var] = pv[var] + acc;
        }
        for (int i = 0; i < m; i++) buf[i] = ps[i] * b[i];
        for (int i = m; i < m2; i++) buf[i] = 0.0;
        pi0[ncuts] = 1.0 + fold_sum(buf, m2);
        row_idx[ncuts] = rank;
        mipx_oracle_safe_cut(n, out, pi0[ncuts], EST_OVER, max_term, safe_pi + (size_t)ncuts * n,
                             safe_pi0 + ncuts);
        ncuts++;
    }
    free(order); free(pv); free(ps); free(buf);
    return ncuts;
}

/* ------------------------------------------------------------------------------------------
 * Cut selection (base_node.py:387-466): candidates = pool cuts with 0 < support <= max_nonzero
 * (support counts |pi_j| > 1e-2); depth = (pi.x - pi0)/||pi|| with fold-in-half sums; greedy in
 * ascending depth (stable): stop at depth >= -min_cut_depth, skip max|pi| > max_abs_coef, skip
 * if cos(angle to an added cut) > cos_parallel.  terminator: 0 none, 1 'no cuts',
 * 2 'no improving cuts', 3 'no sufficient cuts'.  Returns the number added.
 * ---------------------------------------------------------------------------------------- */
static double fold_dot(const double *a, const double *b, int n, int n2, double *buf) {
    for (int j = 0; j < n; j++) buf[j] = a[j] * b[j];
    for (int j = n; j < n2; j++) buf[j] = 0.0;
    return fold_sum(buf, n2);
}

int mipx_oracle_select_cuts(int n, int K, const double *pi, const double *pi0, const double *x,
                            int max_nonzero_coefs, double min_cut_depth, double cos_parallel,
                            double max_abs_coef, int32_t *added, int32_t *terminator, double *depth) {
    int n2 = 64;
    while (n2 < n) n2 <<= 1;
    double *buf = (double *)malloc(sizeof(double) * (size_t)n2);
    double *nrm = (double *)malloc(sizeof(double) * (size_t)(K + 1));
    double *mab = (double *)malloc(sizeof(double) * (size_t)(K + 1));
    int *ord = (int *)malloc(sizeof(int) * (size_t)(K + 1));
    int ncand = 0;
    for (int k = 0; k < K; k++) {
        const double *pk = pi + (size_t)k * n;
        int sup = 0; double mx = 0.0;
        for (int j = 0; j < n; j++) { sup += (pk[j] > 1e-2) + (pk[j] < -1e-2); mx = fmax(mx, fabs(pk[j])); }
        const double dot = fold_dot(pk, x, n, n2, buf);
        const double sq = fold_dot(pk, pk, n, n2, buf);
        nrm[k] = sqrt(sq); mab[k] = mx;
        depth[k] = (sup > 0 && sup <= max_nonzero_coefs) ? (dot - pi0[k]) / nrm[k] : INFINITY;
    }
    for (int k = 0; k < K; k++) {
        if (isinf(depth[k]) && depth[k] > 0) continue;
        int rank = 0;
        for (int q = 0; q < K; q++)
            if (!(isinf(depth[q]) && depth[q] > 0) && (depth[q] < depth[k] || (depth[q] == depth[k] && q < k))) rank++;
        ord[rank] = k; ncand++;
    }
    *terminator = 0;
    if (ncand == 0) *terminator = 1;
    else if (depth[ord[0]] >= 0.0) *terminator = 2;
    else if (depth[ord[0]] >= -min_cut_depth) *terminator = 3;
    int nadd = 0;
    for (int c = 0; c < ncand; c++) {
        const int k = ord[c];
        if (depth[k] >= -min_cut_depth) break;
        if (mab[k] > max_abs_coef) continue;
        int parallel = 0;
        for (int a = 0; a < nadd && !parallel; a++) {
            double cs = fold_dot(pi + (size_t)k * n, pi + (size_t)added[a] * n, n, n2, buf) / (nrm[k] * nrm[added[a]]);
            cs = fmin(1.0, fmax(-1.0, cs));
            if (cs > cos_parallel) parallel = 1;
        }
        if (!parallel) added[nadd++] = k;
    }
    free(buf); free(nrm); free(mab); free(ord);
    return nadd;
}
