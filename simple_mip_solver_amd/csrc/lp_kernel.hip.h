// lp_kernel.hip.h -- K1: batched bounded dual simplex, one workgroup per node LP, gfx950.
//
// Replaces what the reference asks Clp to do in BaseNode._bound_lp / _strong_branch
// (simple_mip_solver/nodes/base_node.py:273, :645-646).  Not a translation of anything: the
// reference has no kernel.  Design:
//
//   * The condensed simplex tableau T (m x n, f64) lives in VGPRs for the whole solve.  A tableau
//     wave is 4 row groups x 16 column lanes: lane (rg, cl) of wave w owns rows g + NG*ii of its
//     group g = 4w + rg and the adjacent column pairs 32*pp + 2*cl + {0,1}.  256x128 -> 7 waves x
//     64 lanes x 5x16 doubles of registers on one CU; it never touches HBM again after the initial
//     read.  A row then sits in 16 lanes x 16 registers of one wave and a column in 4 lanes x 5
//     registers of every wave: moving either through LDS takes few store instructions (an LDS
//     store costs the CU-wide pipe the same whatever its exec mask).
//   * Wave 0 is the control wave: it holds no tableau rows but both borders in registers (reduced
//     costs and nonbasic variable/side per column; beta0, basic values a + b*M, basic variable and
//     its bounds per row) and runs every selection -- leaving row, Harris ratio test, the
//     refactorisation's pivot rows.  The selections are chains of dependent instructions on one
//     wave; the tableau waves' rank-1 update is throughput work.  Splitting them lets the next
//     leaving row be chosen while the tableau waves are still in the fma sweep.
//   * A tableau wave gets its part of the pivot column with v_readlane (into SGPRs, which feed the
//     fma sweep as scalar operands) and hands a copy to the control wave through LDS and a
//     sequence counter (no barrier); rows travel through LDS between two barriers.
//   * Register arrays are only ever indexed statically: dynamic rows/columns are reached through
//     wave-uniform (SGPR) select chains (no scratch).
//   * Arithmetic is IEEE f64 with explicit fma and true division, compiled with
//     -ffp-contract=off, following the canonical operation order documented in
//     oracle/mipx_oracle.c so results are bit-identical to the CPU oracle.
#pragma once
#include <type_traits>
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
    // optional table of anchors (frontier engine): node k of the batch starts from entry
    // anchor_sel[k] of the table (same layout per entry, entries m*n / n+3m / 2n+m apart), or from
    // the single anchor above where the entry is -1.  Indexed by batch position like `slot`, so that
    // both loads leave together.  Register kernels only.
    const int32_t *anchor_sel = nullptr;
    const double *atab_T = nullptr, *atab_vec = nullptr;
    const int32_t *atab_idx = nullptr;
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
    unsigned long long *prof = nullptr;  // MIPX_KPROF builds only: per-section cycle totals of one wave
    int prof_wave = 0;
    // optional dive (frontier engine): when the node LP ends optimal, fractional and below the
    // cutoff, and the branching rule can decide without strong-branching probes, the workgroup
    // branches in place -- one bound of the (basic) branching variable moves -- and goes on with the
    // dual simplex on the tableau it holds in registers: the child costs its few iterations, not a
    // tableau load and a refactorisation.  The child's outputs go to position node + dive_off of
    // the same output arrays; dive_var[node] (preset to -1 by the caller) says whether it happened.
    // dive = D > 1: up to D children in a row (a plunge): level p's outputs at node + p * dive_off, the
    // branching decision taken after level p's LP at dive_var / dir / val[p * dive_off + node].
    int dive = 0, dive_off = 0;
    int rule = 0;                   // 0 most fractional, 1 pseudo cost (K4's rules)
    int n_int = 0;
    const int32_t *int_idx = nullptr;
    const double *cost_l = nullptr, *cost_r = nullptr;
    const uint8_t *has_entry = nullptr;
    double dive_cutoff = 0.0;       // dive only below this objective (the incumbent at launch)
    int32_t *dive_var = nullptr, *dive_dir = nullptr;
    double *dive_val = nullptr;
    int dive_preset = 0;            // 1: the kernel presets status[node + dive_off] = dive_var[node] = -1
    int32_t *zero16 = nullptr;      // optional: 4 words zeroed by workgroup 0 (K4's request counter)
    // optional per-node cut rows (frontier engine with cut rounds; CUTS instantiation only): node k
    // has ncut[k] rows appended to the m shared ones, row m + i being cut cut_ids[k * cut_stride + i]
    // of the cut store (cut_pi: n doubles per cut, cut_pi0: its right-hand side; pi.x >= pi0 like
    // every row, base_node.py:459-460).  The node's LP then is exactly the LP of its m + ncut rows:
    // same arithmetic as a problem with those rows materialised.  mstride = rows allotted per node
    // in vstat_in / vstat_out (n + mstride entries per node), y and the dbg_* dumps.  Indexed by
    // batch position; with vstat_by_node the warm-start basis is too (l, u still through slot).
    const int32_t *ncut = nullptr, *cut_ids = nullptr;
    const double *cut_pi = nullptr, *cut_pi0 = nullptr;
    int cut_stride = 0, mstride = 0, vstat_by_node = 0;
    const int32_t *active = nullptr;  // optional: node k is skipped (nothing read or written) where active[k] == 0
    int cold = 0;                     // the caller vouches that vstat_in holds no basis (all codes 0) and the nodes no cut row: a cold start (K1c may take it)
};

constexpr double kVarEps = 1e-4;  // utils/tolerance.py:2 variable_epsilon
constexpr int kDseRefresh = 64;   // iterations after which the steepest-edge weights are recomputed from the tableau

#ifdef MIPX_KPROF
#define KPROF_MARK(k)                                  \
    do {                                               \
        const unsigned long long t_ = clock64();       \
        if (tid == 64 * g.prof_wave) s.prof[k] += t_ - tprev; \
        tprev = t_;                                    \
    } while (0)
#else
#define KPROF_MARK(k) do { } while (0)
#endif
// MIPX_KPROF_OUT: the slots of the refactorisation marks time the output section instead
#if defined(MIPX_KPROF_RT) || defined(MIPX_KPROF_SETUP)   /* ... or the stages of the ratio test (per iteration), or of the set-up */
#define KPROF_REF_MARK(k) do { } while (0)
#define KPROF_OUT_MARK(k) do { } while (0)
#elif defined(MIPX_KPROF_OUT)
#define KPROF_REF_MARK(k) do { } while (0)
#define KPROF_OUT_MARK(k) KPROF_MARK(k)
#else
#define KPROF_REF_MARK(k) KPROF_MARK(k)
#define KPROF_OUT_MARK(k) do { } while (0)
#endif
#ifdef MIPX_KPROF_RT
#define KPROF_RT_MARK(k) KPROF_MARK(k)
#define KPROF_RT_PIN_D(x) asm volatile("" ::"v"(x) : "memory")   /* the value is worked out before the mark */
#define KPROF_RT_PIN_I(x) asm volatile("" ::"s"(x) : "memory")
#else
#define KPROF_RT_MARK(k) do { } while (0)
#define KPROF_RT_PIN_D(x) do { } while (0)
#define KPROF_RT_PIN_I(x) do { } while (0)
#endif
#ifdef MIPX_KPROF_SETUP
#define KPROF_SETUP_MARK(k) KPROF_MARK(k)
#else
#define KPROF_SETUP_MARK(k) do { } while (0)
#endif

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
// The M component of a value a + b M is a combination of tableau entries with 0/1 weights: what an update
// leaves of it below kBTol is rounding noise and is taken out at once.  Left in, it piles up over thousands
// of pivots until the zero tests see a symbolic violation that is not there, and the report multiplies it by
// kMReport.  A no-op on an LP with finite bounds (every b is exactly 0 there).
__device__ __forceinline__ double snap_m(double v) { return fabs(v) <= kBTol ? 0.0 : v; }

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

// ---- leaner selections for K1: every key it ranks is a non-negative double (violations, |a|,
// ratios) and those order like their bit patterns, so the wave-wide extremum is two u32
// reductions (high words, then low words among the lanes that hold the winning high word), each
// six DPP-fused VOP2 instructions.  s_nop 1 covers the VALU-write -> DPP-read hazard, which the
// compiler does not track through inline asm.  All 64 lanes must be active.
__device__ __forceinline__ unsigned wave_max_u32(unsigned v) {
    asm("s_nop 1\n\tv_max_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_max_u32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_max_u32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_max_u32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_max_u32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
        "s_nop 1\n\tv_max_u32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
        "s_nop 0"
        : "+v"(v));
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}
__device__ __forceinline__ unsigned wave_min_u32(unsigned v) {
    asm("s_nop 1\n\tv_min_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_min_u32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_min_u32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_min_u32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_min_u32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
        "s_nop 1\n\tv_min_u32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
        "s_nop 0"
        : "+v"(v));
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}
// lane-local "keep the better candidate", branch-free (ties go to the smaller payload)
__device__ __forceinline__ void keep_max(double &bk, int &bp, double k, int p, bool valid) {
    const bool b = valid & ((k > bk) | ((k == bk) & (p < bp)));
    bk = b ? k : bk;
    bp = b ? p : bp;
}
__device__ __forceinline__ void keep_min(double &bk, int &bp, double k, int p, bool valid) {
    const bool b = valid & ((k < bk) | ((k == bk) & (p < bp)));
    bk = b ? k : bk;
    bp = b ? p : bp;
}
// the lane(s) holding the extremum: one ballot; a single hit (the usual case) is read back with
// one v_readlane, ties go to the smallest payload
__device__ __forceinline__ int wave_pick(bool hit, int payload) {
    const unsigned long long mask = __ballot(hit);
    if (mask == 0ull) return kNoCand;
    if ((mask & (mask - 1ull)) == 0ull)
        return __builtin_amdgcn_readlane(payload, __ffsll((long long)mask) - 1);
    return (int)wave_min_u32(hit ? (unsigned)payload : (unsigned)kNoCand);
}
// wave-wide argmax / argmin over keys >= +0 (lanes without a candidate carry payload kNoCand).
// The low words are only reduced when the high word does not single out one lane.
__device__ __forceinline__ int wave_argmax_pos(double key, int payload, double &kmax) {
    const bool valid = payload != kNoCand;
    const unsigned hi = valid ? (unsigned)__double2hiint(key) : 0u;
    const unsigned lo = valid ? (unsigned)__double2loint(key) : 0u;
    const unsigned hm = wave_max_u32(hi);
    const unsigned long long m1 = __ballot(valid & (hi == hm));
    if (m1 == 0ull) { kmax = 0.0; return kNoCand; }
    if ((m1 & (m1 - 1ull)) == 0ull) {
        const int l = __ffsll((long long)m1) - 1;
        kmax = __hiloint2double((int)hm, __builtin_amdgcn_readlane((int)lo, l));
        return __builtin_amdgcn_readlane(payload, l);
    }
    const unsigned lm = wave_max_u32(hi == hm ? lo : 0u);
    kmax = __hiloint2double((int)hm, (int)lm);
    return wave_pick(valid & (hi == hm) & (lo == lm), payload);
}
__device__ __forceinline__ int wave_argmin_pos(double key, int payload, double &kmin) {
    const bool valid = payload != kNoCand;
    const unsigned hi = valid ? (unsigned)__double2hiint(key) : 0x7ff00000u;
    const unsigned lo = valid ? (unsigned)__double2loint(key) : 0u;
    const unsigned hm = wave_min_u32(hi);
    const unsigned long long m1 = __ballot(valid & (hi == hm));
    if (m1 == 0ull) { kmin = __builtin_huge_val(); return kNoCand; }
    if ((m1 & (m1 - 1ull)) == 0ull) {
        const int l = __ffsll((long long)m1) - 1;
        kmin = __hiloint2double((int)hm, __builtin_amdgcn_readlane((int)lo, l));
        return __builtin_amdgcn_readlane(payload, l);
    }
    const unsigned lm = wave_min_u32(hi == hm ? lo : 0xffffffffu);
    kmin = __hiloint2double((int)hm, (int)lm);
    return wave_pick(valid & (hi == hm) & (lo == lm), payload);
}
// a wave-uniform double moved to scalar registers
__device__ __forceinline__ double uniform_f64(double v) {
    const int lo = __builtin_amdgcn_readfirstlane(__double2loint(v));
    const int hi = __builtin_amdgcn_readfirstlane(__double2hiint(v));
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double readlane_f64(double v, int l) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), l);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
    return __hiloint2double(hi, lo);
}
// value of lane ^ 16 / lane ^ 32
__device__ __forceinline__ double swz16_f64(double v) {
    const int lo = __builtin_amdgcn_ds_swizzle(__double2loint(v), 0x401f);  // bit mode: and 0x1f, xor 0x10
    const int hi = __builtin_amdgcn_ds_swizzle(__double2hiint(v), 0x401f);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double xor32_f64(double v) { return __shfl_xor(v, 32, 64); }
// value of lane ^ 8 / ^ 4 / ^ 2 / ^ 1 on DPP (no LDS traffic)
__device__ __forceinline__ double xor8_f64(double v) { return dpp_f64<0x128, 0xf>(v); }  // row_ror:8
__device__ __forceinline__ double xor4_f64(double v) {
    // banks 0,2 (lanes 0-3, 8-11 of a row) read lane + 4, banks 1,3 read lane - 4
    int lo = __builtin_amdgcn_update_dpp(__double2loint(v), __double2loint(v), 0x104, 0xf, 0x5, false);
    int hi = __builtin_amdgcn_update_dpp(__double2hiint(v), __double2hiint(v), 0x104, 0xf, 0x5, false);
    lo = __builtin_amdgcn_update_dpp(lo, __double2loint(v), 0x114, 0xf, 0xa, false);
    hi = __builtin_amdgcn_update_dpp(hi, __double2hiint(v), 0x114, 0xf, 0xa, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double xor2_f64(double v) { return dpp_f64<0x4e, 0xf>(v); }  // quad_perm [2,3,0,1]
__device__ __forceinline__ double xor1_f64(double v) { return dpp_f64<0xb1, 0xf>(v); }  // quad_perm [1,0,3,2]
// Row sums over the 64 lanes of a wave for RP rows at once (RP = 16 or 32 partial sums per lane in
// arr_[]): a reduce-scatter whose levels pair (lane, lane ^ h), h = 32, 16, 8, ... -- the canonical
// fold-in-half tree -- so that each exchange moves half as many values as the one before.  The sum
// of row ROWSUM_ROW(lane) ends in arr_[0] of every lane.
#define MIPX_ROWSUM_STEP(arr_, h_, dist_, xch_)                                             \
    do {                                                                                    \
        const bool up_ = (lane & (dist_)) != 0;                                             \
        _Pragma("unroll") for (int k_ = 0; k_ < (h_); k_++) {                               \
            const double send_ = up_ ? arr_[k_] : arr_[k_ + (h_)];                          \
            const double keep_ = up_ ? arr_[k_ + (h_)] : arr_[k_];                          \
            arr_[k_] = keep_ + xch_(send_);                                                 \
        }                                                                                   \
    } while (0)
#define MIPX_ROWSUMS(arr_, rp_)                                                             \
    do {                                                                                    \
        if ((rp_) == 32) {                                                                  \
            MIPX_ROWSUM_STEP(arr_, 16, 32, xor32_f64);                                      \
            MIPX_ROWSUM_STEP(arr_, 8, 16, swz16_f64);                                       \
            MIPX_ROWSUM_STEP(arr_, 4, 8, xor8_f64);                                         \
            MIPX_ROWSUM_STEP(arr_, 2, 4, xor4_f64);                                         \
            MIPX_ROWSUM_STEP(arr_, 1, 2, xor2_f64);                                         \
            arr_[0] = arr_[0] + xor1_f64(arr_[0]);                                          \
        } else {                                                                            \
            MIPX_ROWSUM_STEP(arr_, 8, 32, xor32_f64);                                       \
            MIPX_ROWSUM_STEP(arr_, 4, 16, swz16_f64);                                       \
            MIPX_ROWSUM_STEP(arr_, 2, 8, xor8_f64);                                         \
            MIPX_ROWSUM_STEP(arr_, 1, 4, xor4_f64);                                         \
            arr_[0] = arr_[0] + xor2_f64(arr_[0]);                                          \
            arr_[0] = arr_[0] + xor1_f64(arr_[0]);                                          \
        }                                                                                   \
    } while (0)
// which row (local index ii) a lane holds after MIPX_ROWSUMS
__device__ __forceinline__ int rowsum_row(int lane, int rp) {
    const int r16 = ((lane & 32) ? 8 : 0) | ((lane & 16) ? 4 : 0) | ((lane & 8) ? 2 : 0) | ((lane & 4) ? 1 : 0);
    return rp == 32 ? 2 * r16 + ((lane & 2) ? 1 : 0) : r16;
}

// mailboxes (one 16-byte LDS word each, so a reader needs a single ds_read_b128)
struct alignas(16) MailA {  // control wave -> tableau waves: the pivot row
    int win;     // refactor: leaving row or -1.  simplex: row | 0x8000 if sigma = -1 | cmd << 16
    int lvmeta;  // leaving variable << 3 | fixed << 2 | side it goes to
    double x;    // refactor: 1/p.  simplex: la (value a-part the leaving variable goes to)
};
struct alignas(16) MailB {  // control wave -> tableau waves: the pivot column
    int q;       // entering column or -1
    int ev;      // entering variable
    double pinv;
};

template <int NW, int R, int C, int MP>
struct Smem {
    static constexpr int NP = 16 * C;
    alignas(16) double row[NP];  // pivot row T[r][.]
    double alpha[4 * NW * R];    // pivot column T[.][q]
    double lo[NP];      // structural bounds by variable index
    double up[NP];
    double va[NP];      // nonbasic values a + b*M by column
    double vb[NP];
    double d[NP];       // staging of the borders at setup / output (they live in the control wave's
    double key[NP];     //   registers in between); key: x assembly
    double cvec[NP];    // objective coefficients (the objective is re-summed from x at the end)
    double beta0[MP];
    double ba[4 * NW * R];
    double bb[4 * NW * R];
    double wgt[4 * NW * R];   // dual steepest edge: exact row weights (set-up, refresh) on their way to the control wave
    double tau[4 * NW * R];   //   and tau_i = sum_j T_ij T_rj of the current pivot row
    MailA mbA;
    MailB mbB;
    int bvar[MP];
    int meta[NP];       // nonbasic variable << 3 | fixed << 2 | side (0 lower, 1 upper, 2 fake upper)
    int nvar[NP];       // output staging
    int side[NP];
    int wlist[NP];      // columns of the variables the warm start wants basic, ascending variable
    int nw;
    int nfake0;
    double objv;        // objective of the LP just solved (worked out by tableau wave 0 for the control wave)
    int dive_code;      // branching variable of the in-place dive, -1: none
    int seq;            // pivot column parts published so far (one count per tableau wave and column)
    int pos[NP + MP];   // column of each variable in the starting tableau, -1 if basic
    int8_t wantb[NP + MP];
    int8_t atup[NP + MP];
#ifdef MIPX_KPROF
    unsigned long long prof[16];
#endif
};
// wave-uniform reads of the mailboxes: one ds_read_b128 each, fields moved to scalar registers
__device__ __forceinline__ void read_mail(const MailA &mb, int &win, int &lvmeta, double &x) {
    const int4 v = *reinterpret_cast<const int4 *>(&mb);
    win = __builtin_amdgcn_readfirstlane(v.x);
    lvmeta = __builtin_amdgcn_readfirstlane(v.y);
    x = __hiloint2double(__builtin_amdgcn_readfirstlane(v.w), __builtin_amdgcn_readfirstlane(v.z));
}
__device__ __forceinline__ void read_mail(const MailB &mb, int &q, int &ev, double &pinv) {
    const int4 v = *reinterpret_cast<const int4 *>(&mb);
    q = __builtin_amdgcn_readfirstlane(v.x);
    ev = __builtin_amdgcn_readfirstlane(v.y);
    pinv = __hiloint2double(__builtin_amdgcn_readfirstlane(v.w), __builtin_amdgcn_readfirstlane(v.z));
}

// ---- building blocks of the kernel body.  Macros, not lambdas: the register tableau T must be
// seen as plain local arrays with static indices from the first optimisation pass on, or it is
// demoted to scratch memory.
// element k (wave-uniform) of a short register array: a select chain
#define MIPX_PICK(dst_, arr_, n_, k_)                                                       \
    do {                                                                                    \
        dst_ = arr_[0];                                                                     \
        _Pragma("unroll") for (int t_ = 1; t_ < n_; t_++) dst_ = (k_) == t_ ? arr_[t_] : dst_; \
    } while (0)
template <int C>
struct RowArr {
    double v[C];
    __device__ __forceinline__ double &operator[](int j) { return v[j]; }
    __device__ __forceinline__ const double &operator[](int j) const { return v[j]; }
};
// row / column of a tableau lane's element (ii, jj)
#define MIPX_ROW(ii_) (grp + NG * (ii_))
#define MIPX_COL(jj_) (32 * ((jj_) >> 1) + 2 * cl + ((jj_)&1))
// tableau wave: the four lanes that hold column q (one per row group) store the wave's part of it
// in s.alpha, for the control wave and for the wave itself: every lane reads its rows back
// (a v_readlane per value would cost ~10 cycles each), after one count on s.seq
#define MIPX_PUBLISH_COL(q_)                                                                \
    do {                                                                                    \
        const int qjj_ = 2 * ((q_) >> 5) + ((q_)&1);                                        \
        if (cl == (((q_)&31) >> 1)) {                                                       \
            if constexpr (kVecT) {                                                          \
                _Pragma("unroll") for (int ii = 0; ii < R; ii++) s.alpha[MIPX_ROW(ii)] = T[ii][qjj_]; \
            } else {                                                                        \
                _Pragma("unroll") for (int jj = 0; jj < C; jj++) if (jj == qjj_) {          \
                    _Pragma("unroll") for (int ii = 0; ii < R; ii++) s.alpha[MIPX_ROW(ii)] = T[ii][jj]; \
                }                                                                           \
            }                                                                               \
        }                                                                                   \
        /* the count first: the control wave is waiting for it.  (The release orders the stores */ \
        /* before it; same wave, LDS executes in order, so the reads below see them too.)       */ \
        if (lane == 0) __hip_atomic_fetch_add(&s.seq, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP); \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");                              \
        __builtin_amdgcn_wave_barrier();                                                    \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");                              \
        _Pragma("unroll") for (int ii = 0; ii < R; ii++) al[ii] = s.alpha[MIPX_ROW(ii)];    \
    } while (0)
// the control wave waits until all NW parts of the current pivot column are in s.alpha
#define MIPX_AWAIT_COL(target_)                                                             \
    do {                                                                                    \
        while (__hip_atomic_load(&s.seq, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < (target_)) \
            __builtin_amdgcn_s_sleep(1);                                                    \
    } while (0)
// T[r][.] -> s.row (by the 16 lanes of the row group that holds row r)
#define MIPX_EXTRACT_ROW(r_)                                                                \
    do {                                                                                    \
        const int rl_ = (r_) / NG;                                                          \
        if (grp == (r_) % NG) {                                                             \
            _Pragma("unroll") for (int ii = 0; ii < R; ii++) if (ii == rl_) {               \
                _Pragma("unroll") for (int jj = 0; jj < C; jj++) s.row[MIPX_COL(jj)] = T[ii][jj]; \
            }                                                                               \
        }                                                                                   \
    } while (0)
// rank-1 update of the register tableau for the pivot on (r, q): this lane's part of column q
// is in al[] (MIPX_PUBLISH_COL), row r in s.row.  Row r and column q come out of the fma sweep as
// junk and are overwritten right after it.
#define MIPX_ROW_LDS(jj_) s.row[MIPX_COL(jj_)]
#define MIPX_ROW_REG(jj_) rw[jj_]
#define MIPX_UPDATE_T(r_, q_, pinv_, rowsrc_)                                               \
    do {                                                                                    \
        const int rg_ = (r_) % NG, rl_ = (r_) / NG;                                         \
        const int qcl_ = ((q_)&31) >> 1, qjj_ = 2 * ((q_) >> 5) + ((q_)&1);                 \
        double rh[C];                                                                       \
        _Pragma("unroll") for (int jj = 0; jj < C; jj++) rh[jj] = rowsrc_(jj) * (pinv_);    \
        _Pragma("unroll") for (int ii = 0; ii < R; ii++) {                                  \
            _Pragma("unroll") for (int jj = 0; jj < C; jj++)                                \
                T[ii][jj] = fma(-al[ii], rh[jj], T[ii][jj]);                                \
        }                                                                                   \
        if (cl == qcl_) { /* column q <- -alpha * (1/p) */                                  \
            if constexpr (kVecT) {                                                          \
                _Pragma("unroll") for (int ii = 0; ii < R; ii++) T[ii][qjj_] = -al[ii] * (pinv_); \
            } else {                                                                        \
                _Pragma("unroll") for (int jj = 0; jj < C; jj++) if (jj == qjj_) {          \
                    _Pragma("unroll") for (int ii = 0; ii < R; ii++) T[ii][jj] = -al[ii] * (pinv_); \
                }                                                                           \
            }                                                                               \
        }                                                                                   \
        if (grp == rg_) { /* row r <- row * (1/p), and 1/p at the pivot position */         \
            _Pragma("unroll") for (int ii = 0; ii < R; ii++) if (ii == rl_) {               \
                _Pragma("unroll") for (int jj = 0; jj < C; jj++)                            \
                    T[ii][jj] = (cl == qcl_ && jj == qjj_) ? (pinv_) : rh[jj];              \
            }                                                                               \
        }                                                                                   \
    } while (0)
// Refactorisation: the next pivot column qn is known in advance, so its values AFTER the pivot on
// (r, q) are worked out first (same expressions as the sweep will use: fma(-al, rh, T), rh in row r)
// and handed to the control wave at once; its choice of the next pivot row then overlaps the
// sweep.  T is not touched here.
#define MIPX_PUBLISH_NEXT_COL(r_, pinv_, qn_)                                               \
    do {                                                                                    \
        const int rg_ = (r_) % NG, rl_ = (r_) / NG;                                         \
        const int qnjj_ = 2 * ((qn_) >> 5) + ((qn_)&1);                                     \
        const double rhn_ = s.row[qn_] * (pinv_);                                           \
        if (cl == (((qn_)&31) >> 1)) {                                                      \
            if constexpr (kVecT) {                                                          \
                _Pragma("unroll") for (int ii = 0; ii < R; ii++) {                          \
                    const double v_ = fma(-al[ii], rhn_, T[ii][qnjj_]);                     \
                    s.alpha[MIPX_ROW(ii)] = (grp == rg_ && ii == rl_) ? rhn_ : v_;          \
                }                                                                           \
            } else {                                                                        \
                _Pragma("unroll") for (int jj = 0; jj < C; jj++) if (jj == qnjj_) {         \
                    _Pragma("unroll") for (int ii = 0; ii < R; ii++) {                      \
                        const double v_ = fma(-al[ii], rhn_, T[ii][jj]);                    \
                        s.alpha[MIPX_ROW(ii)] = (grp == rg_ && ii == rl_) ? rhn_ : v_;      \
                    }                                                                       \
                }                                                                           \
            }                                                                               \
        }                                                                                   \
        if (lane == 0) __hip_atomic_fetch_add(&s.seq, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP); \
    } while (0)
// ... and after the sweep every lane picks its rows of that column up for the next pivot
#define MIPX_READ_COL()                                                                     \
    do {                                                                                    \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");                              \
        __builtin_amdgcn_wave_barrier();                                                    \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");                              \
        _Pragma("unroll") for (int ii = 0; ii < R; ii++) al[ii] = s.alpha[MIPX_ROW(ii)];    \
    } while (0)
// control wave: d after the pivot on (r, q), and the variable that takes over column q
#define MIPX_UPDATE_COLS(q_, pinv_, lvmeta_)                                                \
    do {                                                                                    \
        double dq_;                                                                         \
        MIPX_PICK(dq_, cD, PJ, (q_) >> 6);                                                  \
        dq_ = readlane_f64(dq_, (q_)&63);                                                   \
        _Pragma("unroll") for (int kk = 0; kk < PJ; kk++) {                                 \
            const int j = lane + 64 * kk;                                                   \
            const double rho_ = s.row[j] * (pinv_);                                         \
            const double upd_ = fma(-dq_, rho_, cD[kk]);                                    \
            cD[kk] = j == (q_) ? -dq_ * (pinv_) : upd_;                                     \
            cM[kk] = j == (q_) ? (lvmeta_) : cM[kk];                                        \
        }                                                                                   \
    } while (0)

// control wave: what the ratio test needs of every column and does not depend on the pivot row
#define MIPX_PREP_COLS()                                                                    \
    do {                                                                                    \
        const double tol_ = bland ? 0.0 : kDTol; /* Bland: the textbook ratio dj / |a| */   \
        _Pragma("unroll") for (int kk = 0; kk < PJ; kk++) {                                 \
            const int sd_ = cM[kk] & 3;                                                     \
            dje[kk] = sd_ == 0 ? fmax(cD[kk], 0.0) : fmax(-cD[kk], 0.0);                    \
            cN[kk] = dje[kk] + tol_;                                                        \
            cS[kk] = sd_ == 0 ? 0x80000000u : 0u; /* at lower: eligible entries are negative */ \
            cF[kk] = cM[kk] & 4;                                                            \
            cP[kk] = ((cM[kk] >> 3) << 16) | (lane + 64 * kk);                              \
            /* worked out HERE (the compiler would sink it to the uses behind the barrier) */ \
            asm volatile("" : "+v"(dje[kk]), "+v"(cN[kk]), "+v"(cS[kk]), "+v"(cF[kk]), "+v"(cP[kk])); \
        }                                                                                   \
    } while (0)

// Tableau waves: out_[ii] = fold-in-half sum over the padded columns of term_(ii, jj), valid at the
// lanes with cl == 0.  Column j = 32*pp + 2*cl + e, so the levels of the canonical tree are pp (in the
// thread, per e), then the column lanes cl + 8, 4, 2, 1 (DPP inside the 16-lane row group), then e.
#define MIPX_ROW_FOLD(term_, out_)                                                          \
    do {                                                                                    \
        double f0_[R], f1_[R];                                                              \
        _Pragma("unroll") for (int ii = 0; ii < R; ii++) {                                  \
            double t_[C];                                                                   \
            _Pragma("unroll") for (int jj = 0; jj < C; jj++) t_[jj] = term_(ii, jj);        \
            _Pragma("unroll") for (int h = C / 4; h >= 1; h >>= 1) {                        \
                _Pragma("unroll") for (int k = 0; k < 2 * h; k++) t_[k] = t_[k] + t_[k + 2 * h]; \
            }                                                                               \
            f0_[ii] = t_[0];                                                                \
            f1_[ii] = t_[1];                                                                \
        }                                                                                   \
        _Pragma("unroll") for (int ii = 0; ii < R; ii++) {                                  \
            f0_[ii] = f0_[ii] + dpp_f64<0x108, 0xf>(f0_[ii]);                               \
            f1_[ii] = f1_[ii] + dpp_f64<0x108, 0xf>(f1_[ii]);                               \
            f0_[ii] = f0_[ii] + dpp_f64<0x104, 0xf>(f0_[ii]);                               \
            f1_[ii] = f1_[ii] + dpp_f64<0x104, 0xf>(f1_[ii]);                               \
            f0_[ii] = f0_[ii] + dpp_f64<0x102, 0xf>(f0_[ii]);                               \
            f1_[ii] = f1_[ii] + dpp_f64<0x102, 0xf>(f1_[ii]);                               \
            f0_[ii] = f0_[ii] + dpp_f64<0x101, 0xf>(f0_[ii]);                               \
            f1_[ii] = f1_[ii] + dpp_f64<0x101, 0xf>(f1_[ii]);                               \
            out_[ii] = f0_[ii] + f1_[ii];                                                   \
        }                                                                                   \
    } while (0)
#define MIPX_TERM_SQ(ii_, jj_) (T[ii_][jj_] * T[ii_][jj_])
#define MIPX_TERM_ROW(ii_, jj_) (T[ii_][jj_] * rw[jj_])
// exact steepest-edge weights of this wave's rows: 1 + sum_j T_ij^2 (the row of [I | T])
#define MIPX_EXACT_WEIGHTS()                                                                \
    do {                                                                                    \
        double wq_[R];                                                                      \
        MIPX_ROW_FOLD(MIPX_TERM_SQ, wq_);                                                   \
        if (cl == 0) {                                                                      \
            _Pragma("unroll") for (int ii = 0; ii < R; ii++) s.wgt[MIPX_ROW(ii)] = 1.0 + wq_[ii]; \
        }                                                                                   \
    } while (0)

// (a) the control wave's choice of the leaving row, or of the end of the solve, published for everybody:
// largest violation (violations of the symbolic bound M first), ties -> lowest variable index
#define MIPX_LEAVE_SELECT()                                                                  \
    do {                                                                                     \
        int cmd = 0, win = 0;                                                                    \
        if (!symU) {                                                                             \
            /* the usual case, no symbolic-M part anywhere (symU, a scalar, says so: no ballot): */  \
            /* plain bound violations; no candidate in any lane = the end of the solve */        \
            int bp = kNoCand;                                                                    \
            double bk = -1.0;                                                                    \
_Pragma("unroll")                                                                                \
            for (int kk = 0; kk < PI; kk++) {                                                    \
                const int i = lane + 64 * kk;                                                    \
                const double lo = rLo[kk], up = rUp[kk], a = rBa[kk];                            \
                const bool lowv = a < lo - kPTol;                                                \
                const bool upv = !lowv & (a > up + kPTol); /* up = +inf never fires */           \
                const double v0_ = lowv ? lo - a : a - up; /* (dual steepest edge: viol^2 / w) */\
                const double viol = bland ? 0.0 : v0_ * v0_ / rW[kk];                            \
                const int pay = ((rM[kk] >> 2) << 16) | (lowv ? 0 : 0x8000) | i;                 \
                keep_max(bk, bp, viol, pay, (i < m) & (lowv | upv));                             \
            }                                                                                    \
            double km;                                                                           \
            win = wave_argmax_pos(bk, bp, km);                                                   \
            if (win == kNoCand) cmd = nfk > 0 ? 3 : 1;                                           \
            else if ((g.max_iter > 0 && iters >= g.max_iter) || iters >= cap) cmd = 4;           \
        } else {                                                                                 \
            int blevel = 0, bp = kNoCand;                                                        \
            double bk = -1.0;                                                                    \
            bool anybb = false;                                                                  \
_Pragma("unroll")                                                                                \
            for (int kk = 0; kk < PI; kk++) {                                                    \
                const int i = lane + 64 * kk;                                                    \
                if (i < m) {                                                                     \
                    const int v = rM[kk] >> 2;                                                   \
                    const double lo = rLo[kk], up = rUp[kk];                                     \
                    const double a = rBa[kk], bM = rBb[kk];                                      \
                    int level = 0, sg = 0;                                                       \
                    double viol = 0.0;                                                           \
                    if (bM < -kBTol) { level = 2; viol = -bM; sg = 1; }                          \
                    else if (bM > kBTol) {                                                       \
                        if (!isinf(up)) { level = 2; viol = bM; sg = -1; }                       \
                        else if (bM > 1.0 + kBTol) { level = 2; viol = bM - 1.0; sg = -1; }      \
                        else if (bM >= 1.0 - kBTol && a > kPTol) { level = 1; viol = a; sg = -1; }\
                    } else {                                                                     \
                        if (a < lo - kPTol) { level = 1; viol = lo - a; sg = 1; }                \
                        else if (!isinf(up) && a > up + kPTol) { level = 1; viol = a - up; sg = -1; }\
                    }                                                                            \
                    viol = viol * viol / rW[kk];                                                 \
                    if (bland && level > 0) { level = 1; viol = 0.0; }                           \
                    anybb |= bM > kBTol;                                                         \
                    const int pay = (v << 16) | (sg < 0 ? 0x8000 : 0) | i;                       \
                    const bool up_lvl = level > blevel;                                          \
                    const bool same = level == blevel && level > 0 &&                            \
                                      (viol > bk || (viol == bk && pay < bp));                   \
                    if (up_lvl || same) { blevel = level; bk = viol; bp = pay; }                 \
                }                                                                                \
            }                                                                                    \
            const int lvl = __ballot(blevel == 2) ? 2 : (__ballot(blevel == 1) ? 1 : 0);         \
            if (lvl == 0) {                                                                      \
                cmd = (__ballot(anybb) || nfk > 0) ? 3 : 1;                                      \
            } else if ((g.max_iter > 0 && iters >= g.max_iter) || iters >= cap) {                \
                cmd = 4;                                                                         \
            } else {                                                                             \
                double km;                                                                       \
                win = wave_argmax_pos(bk, blevel == lvl ? bp : kNoCand, km);                     \
            }                                                                                    \
        }                                                                                        \
        /* the choice goes out at once; the border values of row r follow after barrier A */      \
        sel_win = cmd == 0 ? (win & 0xffff) : (cmd << 16);                                       \
        if (lane == 0) s.mbA.win = sel_win;                                                      \
    } while (0)
// ... and, after barrier A (the tableau waves are busy handing row r over): what the control wave itself
// needs of the chosen row -- its border values, weight, the bound its variable leaves at
#define MIPX_LEAVE_FETCH()                                                                   \
    do {                                                                                     \
        if (!(sel_win >> 16)) {                                                                  \
            const int rr = sel_win & 0x7fff, rl = rr & 63, rk = rr >> 6;                         \
            double t0, t1, t2, t3, t4;                                                           \
            int tm;                                                                              \
            MIPX_PICK(t0, rLo, PI, rk);                                                          \
            MIPX_PICK(t1, rUp, PI, rk);                                                          \
            MIPX_PICK(t2, rB0, PI, rk);                                                          \
            MIPX_PICK(t3, rBa, PI, rk);                                                          \
            MIPX_PICK(t4, rBb, PI, rk);                                                          \
            MIPX_PICK(tm, rM, PI, rk);                                                           \
            const double lo = readlane_f64(t0, rl), up = readlane_f64(t1, rl);                   \
            sel_b0 = readlane_f64(t2, rl);                                                       \
            sel_ba = readlane_f64(t3, rl);                                                       \
            sel_bb = readlane_f64(t4, rl);                                                       \
            { double t5; MIPX_PICK(t5, rW, PI, rk); sel_w = readlane_f64(t5, rl); }              \
            const int lvv = __builtin_amdgcn_readlane(tm, rl) >> 2;                              \
            double la_;                                                                          \
            int newside;                                                                         \
            if (!(sel_win & 0x8000)) { la_ = lo; newside = 0; }                                  \
            else if (!isinf(up)) { la_ = up; newside = 1; }                                      \
            else { la_ = 0.0; newside = 2; }                                                     \
            sel_lvmeta = (lvv << 3) | newside | (lo == up ? 4 : 0);                              \
            sel_la = la_;                                                                        \
            /* (worked out HERE: the compiler would sink it to the uses behind the barriers) */  \
            asm volatile("" : "+s"(sel_b0), "+s"(sel_ba), "+s"(sel_bb), "+s"(sel_w), "+s"(sel_la), "+s"(sel_lvmeta)); \
        }                                                                                        \
    } while (0)

// Roles: wave 0 is the control wave (borders in registers, every selection); waves 1..NW are the
// tableau waves (whole rows of T in registers, rank-1 updates).  They meet at three barriers per
// simplex iteration (row chosen / row published / column chosen) and two per refactorisation
// pivot; the pivot column goes to the control wave through s.alpha and the s.seq counter.
// The body is instantiated twice, once per role (CTL): with the role a compile-time constant the
// control wave's copy never sees the register tableau (T is dead there), so the selections get
// the registers the tableau would otherwise pin across them, and the tableau waves' copy carries
// none of the control wave's borders.  Same source, same barriers.
template <int NW, int R, int C, int MP, bool DIVE, bool CTL, bool CUTS>
__device__ __forceinline__ void lp_dual_simplex_role(const LpArgs &g, Smem<NW, R, C, MP> &s) {
    static_assert(!(DIVE && CUTS), "the in-place dive runs before a node's cut rounds: not combined");
    constexpr int NT = 64 * (NW + 1);
    constexpr int NG = 4 * NW;          // row groups of the workgroup (4 per tableau wave)
    constexpr int NP = 16 * C;          // padded columns: 16 column lanes x C columns each
    static_assert((NP & (NP - 1)) == 0 && NP % 64 == 0 && C % 2 == 0, "power-of-two columns in adjacent pairs");
    static_assert(MP <= NG * R, "the tableau waves must cover every row");
    constexpr int PI = (MP + 63) / 64;  // rows per lane of the control wave
    constexpr int PJ = NP / 64;         // columns per lane of the control wave

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = CTL ? 0 : __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr bool ctl = CTL;
    const int tw = wave - 1;  // tableau wave index
    const int cl = lane & 15;                 // tableau lanes: column lane,
    const int grp = 4 * tw + (lane >> 4);     //   row group (rows grp + NG*ii)
    const int n = g.n;
    const double INF = __builtin_huge_val();

    // one workgroup per node LP (no grid-stride loop: a loop here makes the compiler hoist every
    // per-element predicate and address of the setup across the whole solve -> register spills)
    const int node = blockIdx.x;
    if (node >= g.batch) return;
    if (CUTS && g.active != nullptr && g.active[node] == 0) return;  // (uniform over the workgroup)
    // rows: the m shared ones, then this node's cuts
    const int m0 = g.m;
    const int kcut = CUTS ? __builtin_amdgcn_readfirstlane(g.ncut[node]) : 0;
    const int m = m0 + kcut;
    const int nv = n + m;
    const int mstr = CUTS ? g.mstride : m;   // rows allotted per node in the strided arrays
    const int32_t *cids = CUTS ? g.cut_ids + (size_t)node * g.cut_stride : nullptr;
#ifdef MIPX_KPROF
    if (tid < 16) s.prof[tid] = 0;
    unsigned long long tprev = clock64();
#endif
    // A lane's part of a tableau row is a C-vector where the register file has room for the tuples:
    // the entry of a wave-uniform column then comes out by GPR indexing (s_set_gpr_idx) instead of a
    // ladder of C uniform branches.  Only where the ladder is long and the tuples fit: the 192-row tile
    // keeps plain arrays (it spills less with them), the small tiles their 4- and 8-way ladders.
    constexpr bool kVecT = C >= 16 && R * C <= 96;
    typedef double TVec __attribute__((ext_vector_type(C)));
    typename std::conditional<kVecT, TVec, RowArr<C>>::type T[R];  // tableau waves: T[ii][jj] = tableau[grp + NG*ii][32*(jj/2) + 2*cl + jj%2]
    double al[R];    // tableau waves: their part of the current pivot column
    // control wave: column border (column j = lane + 64*kk) and row border (row i = lane + 64*kk)
    double cD[PJ];   // reduced cost
    int cM[PJ];      // nonbasic variable << 3 | fixed << 2 | side
    double rB0[PI], rBa[PI], rBb[PI], rLo[PI], rUp[PI];
    double rW[PI];   // dual steepest edge weight of the row: squared norm of its row of [I | T]
    int rM[PI];      // basic variable << 2 | pivoted by the refactorisation << 1 | wanted basic
    const size_t src = g.slot ? (size_t)g.slot[node] : (size_t)node;
    const double *gA = g.A + (size_t)node * g.A_stride;
    const double *gb = g.b + (size_t)node * g.b_stride;
    const double *gc = g.c + (size_t)node * g.c_stride;
    const double *lk = g.l + src * n;
    const double *uk = g.u + src * n;
    const int8_t *vin = g.vstat_in ? g.vstat_in + ((CUTS && g.vstat_by_node) ? (size_t)node : src) * (size_t)(n + mstr) : nullptr;

    // ---- 0. T = -A, beta0 = -b, d = c, slack basis (or the anchor's tableau state) -----------
    const int asel = (g.anchor_sel != nullptr && g.vstat_in != nullptr) ? __builtin_amdgcn_readfirstlane(g.anchor_sel[node]) : -1;
    const double *aT = asel >= 0 ? g.atab_T + (size_t)asel * ((size_t)m * n) : g.anchor_T;
    const double *avec = asel >= 0 ? g.atab_vec + (size_t)asel * (size_t)(n + 3 * m) : g.anchor_vec;
    const int32_t *aidx = asel >= 0 ? g.atab_idx + (size_t)asel * (size_t)(2 * n + m) : g.anchor_idx;
    const bool anchored = aT != nullptr && vin != nullptr && kcut == 0;  // (an anchor has the shared rows only)
    const double sgn = anchored ? 1.0 : -1.0;
#pragma unroll
    for (int ii = 0; ii < R; ii++) {
        al[ii] = 0.0;
#pragma unroll
        for (int jj = 0; jj < C; jj++) T[ii][jj] = 0.0;
    }
    if (!ctl) {
        // The tableau first: its 256 KiB stream through the L1 while the border loads below wait
        // for HBM.  Every load is issued unconditionally from a clamped address (512 contiguous
        // bytes per wave instruction); sign and padding are fixed where the values are first used
        const double *tsrc = anchored ? aT : gA;
        // row r of the node's LP: a shared row, or (CUTS) a row of the cut store
        const double *rowp[R];
#pragma unroll
        for (int ii = 0; ii < R; ii++) {
            const int r = min(MIPX_ROW(ii), m > 0 ? m - 1 : 0);
            if (CUTS && r >= m0) rowp[ii] = g.cut_pi + (size_t)cids[r - m0] * n;
            else rowp[ii] = tsrc + (size_t)r * n;
        }
        if ((n & 1) == 0 && n >= 2) {  // adjacent column pairs as one 16-byte load
            int poff[C / 2];
#pragma unroll
            for (int pp = 0; pp < C / 2; pp++) poff[pp] = min(32 * pp + 2 * cl, n - 2);
#pragma unroll
            for (int ii = 0; ii < R; ii++) {
                const double *arow = rowp[ii];
#pragma unroll
                for (int pp = 0; pp < C / 2; pp++) {
                    const double2 v = m > 0 ? *reinterpret_cast<const double2 *>(arow + poff[pp]) : double2{0.0, 0.0};
                    T[ii][2 * pp] = v.x;
                    T[ii][2 * pp + 1] = v.y;
                }
            }
        } else {
#pragma unroll
            for (int ii = 0; ii < R; ii++) {
                const double *arow = rowp[ii];
#pragma unroll
                for (int jj = 0; jj < C; jj++) T[ii][jj] = m > 0 ? arow[min(MIPX_COL(jj), n - 1)] : 0.0;
            }
        }
    }
    {
        // borders: every global load of the stage is issued before the first one is consumed (one
        // memory latency, not one per array); clamped addresses, the padding is fixed afterwards
        static_assert(NT >= NP + MP, "one staging element per thread");
        const int ic = min(tid, m > 0 ? m - 1 : 0), jc = min(tid, n - 1), vc = min(tid, nv - 1);
        double g_b0;
        if (CUTS && ic >= m0) g_b0 = g.cut_pi0[cids[ic - m0]];
        else g_b0 = anchored ? avec[n + ic] : (m > 0 ? gb[ic] : 0.0);
        const int g_bv = anchored ? aidx[n + ic] : n + ic;
        const double g_d = anchored ? avec[jc] : gc[jc];
        const int g_nv = anchored ? aidx[jc] : jc;
        const double g_lo = lk[jc], g_up = uk[jc];
        const double g_c = gc[jc];
        const int8_t g_st = vin ? vin[vc] : (int8_t)0;
        if (tid < MP) {
            s.beta0[tid] = tid < m ? (anchored ? g_b0 : -g_b0) : 0.0;
            s.bvar[tid] = tid < m ? g_bv : -1;
        }
        if (tid < NP) {
            s.d[tid] = tid < n ? g_d : 0.0;
            s.nvar[tid] = tid < n ? g_nv : -1;
            s.lo[tid] = tid < n ? g_lo : 0.0;
            s.up[tid] = tid < n ? g_up : 0.0;
            s.va[tid] = 0.0;
            s.vb[tid] = 0.0;
            s.cvec[tid] = tid < n ? g_c : 0.0;
        }
        if (tid < NP + MP) {
            const int8_t st = tid < nv ? g_st : (int8_t)0;
            s.wantb[tid] = st == 1;
            s.atup[tid] = st == 2;
            s.pos[tid] = -1;
        }
    }
    // the dive's branching rule (control wave): its candidates (n_int <= n <= NP: PJ per lane) and
    // their pseudo-cost entries are fetched now, so that the rule finds them in registers
    int ci[PJ];
    bool cv[PJ], che[PJ];
    double ccl[PJ], ccr[PJ];
#pragma unroll
    for (int kk = 0; kk < PJ; kk++) { ci[kk] = 0; cv[kk] = false; che[kk] = true; ccl[kk] = 0.0; ccr[kk] = 0.0; }
    if (DIVE && ctl && g.dive) {
#pragma unroll
        for (int kk = 0; kk < PJ; kk++) {
            const int k = lane + 64 * kk;
            cv[kk] = k < g.n_int;
            ci[kk] = g.int_idx[cv[kk] ? k : 0];
        }
        if (g.rule != 0) {
#pragma unroll
            for (int kk = 0; kk < PJ; kk++) {
                che[kk] = g.has_entry[ci[kk]] != 0;
                ccl[kk] = g.cost_l[ci[kk]];
                ccr[kk] = g.cost_r[ci[kk]];
            }
        }
    }
    if (tid == 0) s.seq = 0;
    if (DIVE && g.dive_preset && tid < g.dive) {   // level tid + 1 has no LP, level tid no decision (yet)
        g.status[(size_t)node + (size_t)(tid + 1) * (size_t)g.dive_off] = -1;
        g.dive_var[(size_t)tid * (size_t)g.dive_off + node] = -1;
    }
    if (g.zero16 != nullptr && node == 0 && tid < 4) g.zero16[tid] = 0;
    KPROF_SETUP_MARK(9);
    __syncthreads();
    KPROF_SETUP_MARK(10);
    for (int j = tid; j < NP; j += NT) {
        const int v = s.nvar[j];
        if (j < n) s.pos[v] = j;
        const bool fix = j < n && v < n && s.lo[v < n && v >= 0 ? v : 0] == s.up[v < n && v >= 0 ? v : 0];
        s.meta[j] = (v << 3) | (fix ? 4 : 0);
    }
    __syncthreads();
    KPROF_SETUP_MARK(12);
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
    // the control wave takes the borders into registers
#pragma unroll
    for (int kk = 0; kk < PJ; kk++) {
        cD[kk] = s.d[lane + 64 * kk];
        cM[kk] = s.meta[lane + 64 * kk];
    }
#pragma unroll
    for (int kk = 0; kk < PI; kk++) {
        const int i = lane + 64 * kk;
        const int ic = i < MP ? i : 0;
        const int v = i < MP ? s.bvar[ic] : -1;
        const bool st = v >= 0 && v < n;
        rB0[kk] = i < MP ? s.beta0[ic] : 0.0;
        rBa[kk] = 0.0;
        rBb[kk] = 0.0;
        rW[kk] = 1.0;
        rLo[kk] = st ? s.lo[st ? v : 0] : 0.0;
        rUp[kk] = st ? s.up[st ? v : 0] : INF;
        rM[kk] = (v << 2) | ((v >= 0 && s.wantb[v < 0 ? 0 : v]) ? 1 : 0);
    }
    KPROF_SETUP_MARK(13);
    __syncthreads();
    const int nw = __builtin_amdgcn_readfirstlane(s.nw);
    if (!ctl) {  // the tableau loads land here: sign, and zeros in the padding
#pragma unroll
        for (int ii = 0; ii < R; ii++) {
#pragma unroll
            for (int jj = 0; jj < C; jj++)
                T[ii][jj] = (MIPX_ROW(ii) < m && MIPX_COL(jj) < n) ? sgn * T[ii][jj] : 0.0;
        }
    }

    int npiv = 0, iters = 0, status = -1;
    const int cap = 100 * (m + n) + 1000;
    int cols = 0;  // pivot columns published so far (the control wave waits for NW * cols counts)
    KPROF_MARK(15);

    // ---- 1. refactor: pivot the wanted variables into the basis, ascending variable -----------
    // Per pivot: the tableau waves publish their parts of column q (after their update of the
    // previous pivot); the control wave picks the row; barrier; the wave holding row r publishes
    // it; barrier; rank-1 update / reduced costs.
    if (vin) {
        if (ctl) {
            for (int w = 0; w < nw; w++) {
                const int q = __builtin_amdgcn_readfirstlane(s.wlist[w]);
                cols++;
                MIPX_AWAIT_COL(NW * cols);
                // leaving row: largest |T_iq| among rows whose basic variable is not wanted
                double k1 = -1.0, k2 = -1.0;  // (fallback: wanted, but not pivoted in yet)
                int p1 = kNoCand, p2 = kNoCand;
                double rc[PI], av[PI];
#pragma unroll
                for (int kk = 0; kk < PI; kk++) {
                    const int i = lane + 64 * kk;
                    av[kk] = i < MP ? s.alpha[i < MP ? i : 0] : 0.0;
                    const double a = fabs(av[kk]);
                    const bool ok = i < m && a > kPivTol;
                    const bool wanted = rM[kk] & 1, ent = rM[kk] & 2;
                    rc[kk] = 1.0 / av[kk];  // 1/p of every candidate, off the selection's critical path
                    keep_max(k1, p1, a, i, ok & !wanted);
                    keep_max(k2, p2, a, i, ok & wanted & !ent);
                }
                double km;
                int rr = wave_argmax_pos(k1, p1, km);
                if (rr == kNoCand) rr = wave_argmax_pos(k2, p2, km);
                double pinv = 0.0;
                int lvmeta = 0;
                if (rr != kNoCand) {
                    double rcv, b0r;
                    int rmv;
                    MIPX_PICK(rcv, rc, PI, rr >> 6);
                    MIPX_PICK(rmv, rM, PI, rr >> 6);
                    MIPX_PICK(b0r, rB0, PI, rr >> 6);
                    pinv = readlane_f64(rcv, rr & 63);
                    const int lv = __builtin_amdgcn_readlane(rmv, rr & 63) >> 2;
                    const bool fix = lv < n && s.lo[lv < n ? lv : 0] == s.up[lv < n ? lv : 0];
                    lvmeta = (lv << 3) | (fix ? 4 : 0);
                    if (lane == 0) {
                        MailA mb;
                        mb.win = rr;
                        mb.lvmeta = lvmeta;
                        mb.x = pinv;
                        s.mbA = mb;
                    }
                    // beta0 and the basis list, right away
                    int cmq;
                    MIPX_PICK(cmq, cM, PJ, q >> 6);
                    const int ev = __builtin_amdgcn_readlane(cmq, q & 63) >> 3;
                    const double rhon = readlane_f64(b0r, rr & 63) * pinv;
                    const double elo = ev < n ? s.lo[ev < n ? ev : 0] : 0.0;
                    const double eup = ev < n ? s.up[ev < n ? ev : 0] : INF;
                    const int em = (ev << 2) | 2 | (s.wantb[ev] ? 1 : 0);
#pragma unroll
                    for (int kk = 0; kk < PI; kk++) {
                        const bool pr = lane + 64 * kk == rr;
                        const double upd = fma(-av[kk], rhon, rB0[kk]);
                        rB0[kk] = pr ? rhon : upd;
                        rM[kk] = pr ? em : rM[kk];
                        rLo[kk] = pr ? elo : rLo[kk];
                        rUp[kk] = pr ? eup : rUp[kk];
                    }
                    npiv++;
                } else if (lane == 0) {
                    s.mbA.win = -1;  // singular: the variable stays nonbasic
                }
                KPROF_REF_MARK(9);
                __syncthreads();  // A: row chosen
                __syncthreads();  // B: row published
                KPROF_REF_MARK(10);
                if (rr != kNoCand) MIPX_UPDATE_COLS(q, pinv, lvmeta);
                KPROF_REF_MARK(12);
            }
        } else {
            if (nw > 0) MIPX_PUBLISH_COL(__builtin_amdgcn_readfirstlane(s.wlist[0]));
            for (int w = 0; w < nw; w++) {
                const int q = __builtin_amdgcn_readfirstlane(s.wlist[w]);
                __syncthreads();  // A
                int r, lvmeta;
                double pinv;
                read_mail(s.mbA, r, lvmeta, pinv);
                if (r >= 0) MIPX_EXTRACT_ROW(r);
                __syncthreads();  // B
                KPROF_REF_MARK(10);
                const int qn = w + 1 < nw ? __builtin_amdgcn_readfirstlane(s.wlist[w + 1 < nw ? w + 1 : 0]) : -1;
                if (r >= 0 && qn >= 0) MIPX_PUBLISH_NEXT_COL(r, pinv, qn);  // out before the sweep
                KPROF_REF_MARK(14);
                if (r >= 0) MIPX_UPDATE_T(r, q, pinv, MIPX_ROW_LDS);
                if (qn >= 0) {
                    if (r >= 0) MIPX_READ_COL();
                    else MIPX_PUBLISH_COL(qn);  // (singular column: nothing changed, plain publish)
                }
                KPROF_REF_MARK(13);
            }
        }
    }

    const bool solve = !(vin && g.refactor_only);
    int nfake = 0;  // nonbasic columns at the symbolic bound M (control wave)
    if (solve) {
        // ---- 2. nonbasic sides and values (control wave), then the basic values ---------------
        if (ctl) {
#pragma unroll
            for (int kk = 0; kk < PJ; kk++) {
                const int j = lane + 64 * kk;
                int side = 0;
                if (j < n) {
                    const int v = cM[kk] >> 3;
                    const double lo = v < n ? s.lo[v < n ? v : 0] : 0.0;
                    const double up = v < n ? s.up[v < n ? v : 0] : INF;
                    const double dj = cD[kk];
                    if (lo == up) side = 0;
                    else if (dj < -kDTol) side = isinf(up) ? 2 : 1;
                    else if (dj > kDTol) side = 0;
                    else side = (s.atup[v] && !isinf(up)) ? 1 : 0;
                    s.va[j] = side == 0 ? lo : side == 1 ? up : 0.0;
                    s.vb[j] = side == 2 ? 1.0 : 0.0;
                }
                cM[kk] = (cM[kk] & ~3) | side;
                nfake += __popcll(__ballot(side == 2));
            }
            if (lane == 0) s.nfake0 = nfake;
        }
        __syncthreads();
        if (!ctl) {
            const int nf0 = __builtin_amdgcn_readfirstlane(s.nfake0);
            // row sums over the padded columns with the canonical fold-in-half tree: column
            // j = 32*pp + 2*cl + e, so the levels are pp (in the thread, per e), then the column
            // lanes cl + 8, 4, 2, 1 (DPP inside the 16-lane row group), then e last
            double va[C];
#pragma unroll
            for (int jj = 0; jj < C; jj++) va[jj] = s.va[MIPX_COL(jj)];
            double s0[R], s1[R];
#pragma unroll
            for (int ii = 0; ii < R; ii++) {
                double t[C];
#pragma unroll
                for (int jj = 0; jj < C; jj++) t[jj] = T[ii][jj] * va[jj];
#pragma unroll
                for (int h = C / 4; h >= 1; h >>= 1) {  // pairs pp and pp + h, both columns of the pair
#pragma unroll
                    for (int k = 0; k < 2 * h; k++) t[k] = t[k] + t[k + 2 * h];
                }
                s0[ii] = t[0];
                s1[ii] = t[1];
            }
#pragma unroll
            for (int ii = 0; ii < R; ii++) {
                s0[ii] = s0[ii] + dpp_f64<0x108, 0xf>(s0[ii]);  // row_shl:8
                s1[ii] = s1[ii] + dpp_f64<0x108, 0xf>(s1[ii]);
                s0[ii] = s0[ii] + dpp_f64<0x104, 0xf>(s0[ii]);  // row_shl:4
                s1[ii] = s1[ii] + dpp_f64<0x104, 0xf>(s1[ii]);
                s0[ii] = s0[ii] + dpp_f64<0x102, 0xf>(s0[ii]);  // row_shl:2
                s1[ii] = s1[ii] + dpp_f64<0x102, 0xf>(s1[ii]);
                s0[ii] = s0[ii] + dpp_f64<0x101, 0xf>(s0[ii]);  // row_shl:1
                s1[ii] = s1[ii] + dpp_f64<0x101, 0xf>(s1[ii]);
            }
            if (cl == 0) {
#pragma unroll
                for (int ii = 0; ii < R; ii++) {
                    s.ba[MIPX_ROW(ii)] = s0[ii] + s1[ii];
                    s.bb[MIPX_ROW(ii)] = 0.0;
                }
            }
            if (nf0 != 0) {  // the M parts: all zero unless some nonbasic sits at the symbolic bound
#pragma unroll
                for (int jj = 0; jj < C; jj++) va[jj] = s.vb[MIPX_COL(jj)];
#pragma unroll
                for (int ii = 0; ii < R; ii++) {
                    double t[C];
#pragma unroll
                    for (int jj = 0; jj < C; jj++) t[jj] = T[ii][jj] * va[jj];
#pragma unroll
                    for (int h = C / 4; h >= 1; h >>= 1) {
#pragma unroll
                        for (int k = 0; k < 2 * h; k++) t[k] = t[k] + t[k + 2 * h];
                    }
                    s0[ii] = t[0];
                    s1[ii] = t[1];
                }
#pragma unroll
                for (int ii = 0; ii < R; ii++) {
                    s0[ii] = s0[ii] + dpp_f64<0x108, 0xf>(s0[ii]);
                    s1[ii] = s1[ii] + dpp_f64<0x108, 0xf>(s1[ii]);
                    s0[ii] = s0[ii] + dpp_f64<0x104, 0xf>(s0[ii]);
                    s1[ii] = s1[ii] + dpp_f64<0x104, 0xf>(s1[ii]);
                    s0[ii] = s0[ii] + dpp_f64<0x102, 0xf>(s0[ii]);
                    s1[ii] = s1[ii] + dpp_f64<0x102, 0xf>(s1[ii]);
                    s0[ii] = s0[ii] + dpp_f64<0x101, 0xf>(s0[ii]);
                    s1[ii] = s1[ii] + dpp_f64<0x101, 0xf>(s1[ii]);
                }
                if (cl == 0) {
#pragma unroll
                    for (int ii = 0; ii < R; ii++) s.bb[MIPX_ROW(ii)] = snap_m(0.0 - (s0[ii] + s1[ii]));
                }
            }
            MIPX_EXACT_WEIGHTS();   // the steepest-edge weights the solve starts with
        }
        __syncthreads();
        KPROF_MARK(7);
    }

    // pass 0: the node; pass 1: its child after an in-place dive (g.dive)
    int pass = 0;
    asm volatile("" : "+s"(pass));  // opaque: keeps the compiler from peeling the first pass (two copies of the sweep)
    size_t onode = (size_t)node;
    bool symU = false;           // control wave: some basic value may carry a symbolic part (MIPX_LEAVE_SELECT)
    int dvar = -1, ddir = 0;     // control wave: the dive's branching variable, direction,
    double dbound = 0.0;         //   and the bound that moves (floor / ceil of its value)
    int wage = 0;                // iterations since the steepest-edge weights were exact (both roles count)
#pragma clang loop unroll(disable)
    for (;;) {
    if (solve) {
        // ---- 3. dual simplex ------------------------------------------------------------------
        // Control wave: leaving row r | A | (row r arrives) | B | ratio test: column q | C | reduced
        // costs, wait for column q, basic values, next leaving row | A ...
        // Tableau waves: | A | the wave holding row r publishes it | B | | C | column parts out,
        // rank-1 update | A ...
        if (ctl) {
            bool bland = false;
            int degen = 0;  // consecutive degenerate steps; > m+n -> Bland's rule
            const int nfk0 = nfake;
            int nfk = nfk0;
            // may a basic value carry a symbolic part?  Only once a nonbasic variable has sat at the symbolic bound
            // (then it stays true for the solve: the general selection gives the same answer where none does)
            if (pass == 0) symU = nfake != 0;
            double dje[PJ], cN[PJ];  // per column, for the ratio test: max(+-d, 0), that + tol,
            unsigned cS[PJ];         //   the sign mask of an eligible entry,
            int cP[PJ];              //   the payload (variable << 16 | column)
            int cF[PJ];              //   and "fixed: never enters" (nonzero)
            double sel_b0 = 0.0, sel_ba = 0.0, sel_bb = 0.0;  // border values of row r
            double sel_w = 1.0;                               //   and its steepest-edge weight
            int sel_win = 0, sel_lvmeta = 0;
            double sel_la = 0.0;
            if (pass == 0) {
#pragma unroll
                for (int kk = 0; kk < PI; kk++) {
                    const int i = lane + 64 * kk;
                    rBa[kk] = i < MP ? rB0[kk] - s.ba[i < MP ? i : 0] : 0.0;
                    rBb[kk] = i < MP ? s.bb[i < MP ? i : 0] : 0.0;
                    rW[kk] = i < MP ? s.wgt[i < MP ? i : 0] : 1.0;
                }
            } else {  // the dive: one bound of the (basic) branching variable moves
#pragma unroll
                for (int kk = 0; kk < PI; kk++) {
                    const bool hit = (rM[kk] >> 2) == dvar;
                    rUp[kk] = (hit && ddir == 0) ? dbound : rUp[kk];
                    rLo[kk] = (hit && ddir != 0) ? dbound : rLo[kk];
                }
                if (lane == 0) {
                    if (ddir == 0) s.up[dvar] = dbound;
                    else s.lo[dvar] = dbound;
                }
            }
            // Where the hand-overs of the large tiles leave the control wave idle, work that does not
            // depend on the row / column just chosen is placed in their shadow; the small tiles' hand-overs
            // are too short for that (same arithmetic either way).
            constexpr bool kShadow = NW >= 7;
            MIPX_LEAVE_SELECT();
            if constexpr (!kShadow) MIPX_LEAVE_FETCH();
            __syncthreads();  // A
            for (;;) {
                if (sel_win >> 16) { const int cmd = sel_win >> 16; status = cmd == 1 ? 0 : cmd == 3 ? 2 : 3; break; }
                const int r = sel_win & 0x7fff;
                const unsigned sflip = (sel_win & 0x8000) ? 0x80000000u : 0u;  // sigma = -1
                if constexpr (kShadow) {  // (while the tableau waves hand row r over)
                    MIPX_LEAVE_FETCH();
                    MIPX_PREP_COLS();
                }
                KPROF_MARK(1);
                __syncthreads();  // B: row r is in s.row
                KPROF_RT_MARK(8);
                if constexpr (!kShadow) MIPX_PREP_COLS();
                // (c) Harris ratio test on row r
                double aa[PJ];
                bool el[PJ];
                double k1 = INF;
                int p1 = kNoCand;
                // (per column, prepared while the last pivot's column was on its way: cS the sign that
                // makes an eligible entry positive, cN = max(+-d, 0) + tol, cP the payload, cF "fixed")
#pragma unroll
                for (int kk = 0; kk < PJ; kk++) {
                    const double rv = s.row[lane + 64 * kk];
                    const double a = __hiloint2double(__double2hiint(rv) ^ (sflip ^ cS[kk]), __double2loint(rv));
                    el[kk] = (cF[kk] == 0) & (a > kPivTol);
                    aa[kk] = fabs(a);
                }
                double key[PJ];  // the PJ divisions are independent chains: issued side by side
#pragma unroll
                for (int kk = 0; kk < PJ; kk++) key[kk] = cN[kk] / aa[kk];  // unconditionally: no divergent branch
#pragma unroll
                for (int kk = 0; kk < PJ; kk++) keep_min(k1, p1, key[kk], cP[kk], el[kk]);
                KPROF_RT_PIN_D(k1); KPROF_RT_PIN_D(p1);
                KPROF_RT_MARK(9);
                double thmax;
                const int w1 = wave_argmin_pos(k1, p1, thmax);
                KPROF_RT_PIN_I(w1); KPROF_RT_PIN_D(thmax);
                KPROF_RT_MARK(10);
                int qq = -1;
                if (w1 != kNoCand && bland) {
                    qq = w1 & 0xffff;  // ties -> lowest variable index
                } else if (w1 != kNoCand) {
                    const int jmin = w1 & 0xffff;
                    double k2 = -1.0;
                    int p2 = kNoCand;
#pragma unroll
                    for (int kk = 0; kk < PJ; kk++) {
                        const int j = lane + 64 * kk;
                        const bool ok = el[kk] & ((j == jmin) | !(dje[kk] > thmax * aa[kk]));
                        keep_max(k2, p2, aa[kk], cP[kk], ok);
                    }
                    KPROF_RT_PIN_D(k2); KPROF_RT_PIN_D(p2);
                    KPROF_RT_MARK(12);
                    double amax;
                    qq = wave_argmax_pos(k2, p2, amax) & 0xffff;
                    KPROF_RT_PIN_I(qq);
                    KPROF_RT_MARK(13);
                }
                if (lane == 0) s.mbB.q = qq;  // (the tableau waves want q and nothing else)
                KPROF_RT_PIN_I(qq);
                KPROF_RT_MARK(14);
                KPROF_MARK(3);
                __syncthreads();  // C: column chosen
                if (qq < 0) { status = 1; break; }  // no entering column: primal infeasible
                const int q = qq;
                // the bookkeeping of the choice, while the tableau waves get column q out
                const int lvmeta = sel_lvmeta;
                const double la = sel_la, lb = (lvmeta & 3) == 2 ? 1.0 : 0.0;
                int ev;
                double vaq, vbq;
                {
                    const int ql = qq & 63, qk = qq >> 6;
                    double t0;
                    int tm;
                    MIPX_PICK(t0, dje, PJ, qk);
                    MIPX_PICK(tm, cM, PJ, qk);
                    const double djq = readlane_f64(t0, ql);
                    const int cm = __builtin_amdgcn_readlane(tm, ql);
                    degen = djq <= kDTol ? degen + 1 : 0;
                    bland = degen > m + n;
                    nfk += ((lvmeta & 3) == 2 ? 1 : 0) - ((cm & 3) == 2 ? 1 : 0);
                    symU |= ((lvmeta & 3) == 2) | ((cm & 3) == 2);
                    ev = cm >> 3;
                    vaq = uniform_f64(s.va[qq]);
                    vbq = uniform_f64(s.vb[qq]);
                }
                const double pinv = 1.0 / uniform_f64(s.row[q]);  // 1/p: every wave works it out for itself
                iters++;
                npiv++;
                cols++;
                MIPX_UPDATE_COLS(q, pinv, lvmeta);
                // (worked out HERE, while column q is on its way: the compiler would sink it behind the wait)
#pragma unroll
                for (int kk = 0; kk < PJ; kk++) asm volatile("" : "+v"(cD[kk]), "+v"(cM[kk]));
                if (lane == 0) { s.va[q] = la; s.vb[q] = lb; }
                KPROF_MARK(2);
                {   // basic values after the pivot on (r, q), then the next leaving row
                    double rhon = sel_b0 * pinv;
                    double ta = (sel_ba - la) * pinv, tb = (sel_bb - lb) * pinv;
                    const double elo = ev < n ? s.lo[ev < n ? ev : 0] : 0.0;
                    const double eup = ev < n ? s.up[ev < n ? ev : 0] : INF;
                    // steepest-edge weights after the row operations of this pivot (tau came before C)
                    const double wr = sel_w;
                    double wrn = (wr * pinv) * pinv;
                    wrn = wrn < 1.0 ? 1.0 : wrn;
                    double tauv[PI];
#pragma unroll
                    for (int kk = 0; kk < PI; kk++) tauv[kk] = lane + 64 * kk < MP ? s.tau[lane + 64 * kk < MP ? lane + 64 * kk : 0] : 0.0;
                    asm volatile("" : "+v"(rhon), "+v"(ta), "+v"(tb), "+v"(wrn));
                    MIPX_AWAIT_COL(NW * cols);
                    KPROF_MARK(4);
#pragma unroll
                    for (int kk = 0; kk < PI; kk++) {
                        const int i = lane + 64 * kk;
                        const double a = i < MP ? s.alpha[i < MP ? i : 0] : 0.0;
                        const bool pr = i == r;
                        const double u0 = fma(-a, rhon, rB0[kk]), u1 = fma(-a, ta, rBa[kk]), u2 = fma(-a, tb, rBb[kk]);
                        const double ratio = a * pinv;
                        double wn = fma(ratio, fma(ratio, wr, -2.0 * tauv[kk]), rW[kk]);
                        wn = wn < 1.0 ? 1.0 : wn;
                        rB0[kk] = pr ? rhon : u0;
                        rBa[kk] = pr ? vaq + ta : u1;
                        rBb[kk] = snap_m(pr ? vbq + tb : u2);
                        rW[kk] = pr ? wrn : wn;
                        rM[kk] = pr ? (ev << 2) : rM[kk];
                        rLo[kk] = pr ? elo : rLo[kk];
                        rUp[kk] = pr ? eup : rUp[kk];
                    }
                    if (++wage == kDseRefresh) {  // exact weights from the tableau after this pivot (A')
                        wage = 0;
                        __syncthreads();
#pragma unroll
                        for (int kk = 0; kk < PI; kk++) rW[kk] = lane + 64 * kk < MP ? s.wgt[lane + 64 * kk < MP ? lane + 64 * kk : 0] : 1.0;
                    }
                    MIPX_LEAVE_SELECT();
                    if constexpr (!kShadow) MIPX_LEAVE_FETCH();
                }
                KPROF_MARK(5);
                __syncthreads();  // A: row chosen
                KPROF_MARK(0);
            }
            (void)nfk0;
            nfake = nfk;
        } else {
            __syncthreads();  // A
            for (;;) {
                int win, lvmeta;
                double la;
                read_mail(s.mbA, win, lvmeta, la);
                if (win >> 16) break;
                const int r = win & 0x7fff;
                MIPX_EXTRACT_ROW(r);  // (b)
                KPROF_MARK(1);
                __syncthreads();  // B
                // while the control wave runs the ratio test: tau_i = sum_j T_ij T_rj for the steepest-edge
                // weights (the pivot row stays in registers for the sweep)
                double rw[C];
#pragma unroll
                for (int jj = 0; jj < C; jj++) rw[jj] = s.row[MIPX_COL(jj)];
                {
                    double tq[R];
                    MIPX_ROW_FOLD(MIPX_TERM_ROW, tq);
                    if (cl == 0) {
#pragma unroll
                        for (int ii = 0; ii < R; ii++) s.tau[MIPX_ROW(ii)] = tq[ii];
                    }
                }
                __syncthreads();  // C
                KPROF_MARK(3);
                int q, ev;
                double pinv;
                read_mail(s.mbB, q, ev, pinv);
                if (q < 0) break;
                KPROF_MARK(8);
                MIPX_PUBLISH_COL(q);  // (d)
                pinv = 1.0 / uniform_f64(s.row[q]);  // 1/p: every wave works it out for itself
                KPROF_MARK(4);
                MIPX_UPDATE_T(r, q, pinv, MIPX_ROW_REG);
                if (++wage == kDseRefresh) {  // the weights afresh from the tableau after this pivot
                    wage = 0;
                    MIPX_EXACT_WEIGHTS();
                    __syncthreads();  // A'
                }
                KPROF_MARK(5);
                __syncthreads();  // A
                KPROF_MARK(0);
            }
        }
    } else {
        status = 3;
    }

    // Sections 4 and 5 sit inside the pass loop: without the opaque copies below the compiler hoists
    // their loop-invariant indices, predicates and addresses above the whole solve (290 spilled
    // SGPRs, 164 spilled VGPRs).
    int o_tid = tid, o_lane = lane, o_n = n, o_m = m;
    asm volatile("" : "+v"(o_tid), "+v"(o_lane), "+s"(o_n), "+s"(o_m));
    {
    const int tid = o_tid, lane = o_lane, n = o_n, m = o_m;
    const int ms = CUTS ? g.mstride : o_m;  // rows allotted per node in y / vstat_out / the dumps
    // ---- 4. outputs ---------------------------------------------------------------------------
#ifdef MIPX_KPROF
    KPROF_MARK(6);
    __syncthreads();
    if (g.prof && tid == 0) {
        for (int k = 0; k < 16; k++) { atomicAdd(&g.prof[k], s.prof[k]); s.prof[k] = 0; }  // (per pass)
        atomicAdd(&g.prof[16], (unsigned long long)iters);
        atomicAdd(&g.prof[17], (unsigned long long)(npiv - iters));
        atomicAdd(&g.prof[18], 1ull);
    }
    __syncthreads();
    tprev = clock64();   // (the dump above is the profiler's own cost)
#endif
    // The engine's steps want neither row duals nor a tableau dump: then everything below is the control
    // wave's alone -- it holds both borders in registers -- and the tableau waves go straight to the barrier
    // behind which the dive's decision is known.  (The general path hands the borders over through LDS and
    // shares the loops out: three more barriers, each with a serial stretch before it.)  Same arithmetic,
    // same order: x by variable index, the objective's fold-in-half sum, K4's rule.
    if (g.y == nullptr && g.dbg_T == nullptr) {
    const bool more = DIVE && pass < g.dive && solve;  // (DIVE = false: the pass loop folds away)
    if (ctl) {
        constexpr int PER = NP / 64;
        static_assert(PER == PJ, "one column per lane and slot");
#pragma unroll
        for (int kk = 0; kk < PJ; kk++) {
            const int j = lane + 64 * kk;
            const int v = cM[kk] >> 3, sd = cM[kk] & 3;
            if (j >= n) s.key[j] = 0.0;   // (the padding of the fold; columns >= n hold no variable)
            if (j < n && v < n) s.key[v] = sd == 2 ? kMReport : s.va[j];
        }
#pragma unroll
        for (int kk = 0; kk < PI; kk++) {
            const int i = lane + 64 * kk;
            const int v = rM[kk] >> 2;
            if (i < m && v < n) s.key[v] = fma(rBb[kk], kMReport, rBa[kk]);
        }
        if (g.vstat_out) {   // the basis, straight from the registers
            int8_t *vo = g.vstat_out + onode * (size_t)(n + ms);
#pragma unroll
            for (int kk = 0; kk < PI; kk++)
                if (lane + 64 * kk < m) vo[rM[kk] >> 2] = 1;
#pragma unroll
            for (int kk = 0; kk < PJ; kk++)
                if (lane + 64 * kk < n) vo[cM[kk] >> 3] = (cM[kk] & 3) ? 2 : 3;
        }
        // (one wave: its LDS writes above are in order before the reads below)
        double p[PER];
#pragma unroll
        for (int k = 0; k < PER; k++) {
            const int j = lane + 64 * k;
            const double xj = s.key[j];
            if (g.x && j < n) g.x[onode * n + j] = xj;
            p[k] = j < n ? s.cvec[j] * xj : 0.0;
        }
#pragma unroll
        for (int h = PER / 2; h >= 1; h >>= 1) {
#pragma unroll
            for (int k = 0; k < h; k++) p[k] = p[k] + p[k + h];
        }
        double sum = p[0];
        sum = sum + xor32_f64(sum);
        sum = sum + swz16_f64(sum);
        sum = sum + xor8_f64(sum);
        sum = sum + xor4_f64(sum);
        sum = sum + xor2_f64(sum);
        sum = sum + xor1_f64(sum);
        const double objv = uniform_f64(sum);
        int code = -1;
        double dval = 0.0;
        if (more && status == 0 && objv < g.dive_cutoff) {
            double bk = -1.0;
            int bp = kNoCand, nprobe = 0;
#pragma unroll
            for (int kk = 0; kk < PJ; kk++) {
                const int k = lane + 64 * kk;
                const double v = s.key[ci[kk]];
                const double fl = floor(v), ce = ceil(v);
                const double dist = fmin(v - fl, ce - v);
                const bool frac = cv[kk] && dist > kVarEps;
                const double key = g.rule == 0 ? dist : fmin(ccr[kk] * (ce - v), ccl[kk] * (v - fl));
                keep_max(bk, bp, key, k, frac && che[kk]);
                nprobe += __popcll(__ballot(frac && !che[kk]));
            }
            double km;
            const int win = wave_argmax_pos(bk, bp, km);
            if (win != kNoCand && nprobe == 0) {
                const int wl = win & 63, wk = win >> 6;
                int t_i;
                double t_l, t_r;
                MIPX_PICK(t_i, ci, PJ, wk);
                MIPX_PICK(t_l, ccl, PJ, wk);
                MIPX_PICK(t_r, ccr, PJ, wk);
                dvar = __builtin_amdgcn_readlane(t_i, wl);
                const double wcl = readlane_f64(t_l, wl), wcr = readlane_f64(t_r, wl);
                const double v = uniform_f64(s.key[dvar]);
                const double fl = floor(v), ce = ceil(v);
                if (g.rule == 0) ddir = (v - fl <= ce - v) ? 0 : 1;
                else ddir = (wcl * (v - fl) <= wcr * (ce - v)) ? 0 : 1;
                dbound = ddir == 0 ? fl : ce;
                dval = v;
                bool mine = false;  // a bound change in place needs the variable basic
#pragma unroll
                for (int kk = 0; kk < PI; kk++) mine |= (lane + 64 * kk < m) && (rM[kk] >> 2) == dvar;
                if (__ballot(mine) != 0ull) code = dvar;
            }
        }
        if (lane == 0) {
            if (g.obj) g.obj[onode] = status == 1 ? INF : objv;
            if (g.status) g.status[onode] = status;
            if (g.iters) g.iters[onode] = iters;
            if (g.npivots) g.npivots[onode] = npiv;
            if (code >= 0) {
                const size_t di = (size_t)pass * (size_t)g.dive_off + node;
                g.dive_var[di] = dvar;
                g.dive_dir[di] = ddir;
                g.dive_val[di] = dval;
            }
            s.dive_code = code;
        }
    }
    if (!more) break;
    KPROF_OUT_MARK(14);
    __syncthreads();
    } else {
    if (ctl) {
        // the borders go to LDS for the output loops; x by variable index (s.key) straight from the registers
#pragma unroll
        for (int kk = 0; kk < PJ; kk++) {
            const int j = lane + 64 * kk;
            const int v = cM[kk] >> 3, sd = cM[kk] & 3;
            s.d[j] = cD[kk];
            s.nvar[j] = v;
            s.side[j] = sd;
            if (j < n && v < n) s.key[v] = sd == 2 ? kMReport : s.va[j];
        }
#pragma unroll
        for (int kk = 0; kk < PI; kk++) {
            const int i = lane + 64 * kk;
            if (i < MP) {
                const int v = rM[kk] >> 2;
                s.beta0[i] = rB0[kk];
                s.ba[i] = rBa[kk];
                s.bb[i] = rBb[kk];
                s.bvar[i] = v;
                if (i < m && v < n) s.key[v] = fma(rBb[kk], kMReport, rBa[kk]);
            }
        }
    }
    for (int j = n + tid; j < NP; j += NT) s.key[j] = 0.0;
    __syncthreads();
    KPROF_OUT_MARK(9);
    if (g.x)
        for (int j = tid; j < n; j += NT) g.x[onode * n + j] = s.key[j];
    if (g.y) {
        for (int i = tid; i < m; i += NT) g.y[onode * ms + i] = 0.0;
        __syncthreads();
        for (int j = tid; j < n; j += NT)
            if (s.nvar[j] >= n) g.y[onode * ms + (s.nvar[j] - n)] = s.d[j];
    }
    if (g.vstat_out) {
        int8_t *vo = g.vstat_out + onode * (size_t)(n + ms);
        for (int i = tid; i < m; i += NT) vo[s.bvar[i]] = 1;
        for (int j = tid; j < n; j += NT) vo[s.nvar[j]] = s.side[j] ? 2 : 3;
    }
    if (g.dbg_T && (node == 0 || g.dbg_all)) {
        const size_t k = g.dbg_all ? (size_t)node : 0;
        double *dT = g.dbg_T + k * (size_t)ms * n;
        double *dvec = g.dbg_vec + k * (size_t)(n + 3 * ms);
        int32_t *didx = g.dbg_idx + k * (size_t)(2 * n + ms);
#pragma unroll
        for (int ii = 0; ii < R; ii++) {
#pragma unroll
            for (int jj = 0; jj < C; jj++) {
                const int i = MIPX_ROW(ii), j = MIPX_COL(jj);
                if (!ctl && i < m && j < n) dT[(size_t)i * n + j] = T[ii][jj];
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
    KPROF_OUT_MARK(10);
    if (!ctl && tw == 0) {
        // obj = fold-in-half sum of c_j x_j over the padded power-of-two length
        constexpr int PER = NP / 64;
        double p[PER];
#pragma unroll
        for (int k = 0; k < PER; k++) {
            const int j = lane + 64 * k;
            p[k] = j < n ? s.cvec[j] * s.key[j] : 0.0;
        }
#pragma unroll
        for (int h = PER / 2; h >= 1; h >>= 1) {
#pragma unroll
            for (int k = 0; k < h; k++) p[k] = p[k] + p[k + h];
        }
        // lane i < h adds lane i + h = i ^ h: the fold-in-half pairs, without an LDS round trip below 32
        double sum = p[0];
        sum = sum + xor32_f64(sum);
        sum = sum + swz16_f64(sum);
        sum = sum + xor8_f64(sum);
        sum = sum + xor4_f64(sum);
        sum = sum + xor2_f64(sum);
        sum = sum + xor1_f64(sum);
        if (lane == 0) s.objv = sum;
    }
    // ---- 5. dive: K4's branching rule on the solution in s.key, by the control wave, meanwhile; whether
    // the node may dive at all (optimal, below the cutoff) is settled behind the barrier ----------------
    const bool more = DIVE && pass < g.dive && solve;  // (DIVE = false: the pass loop folds away)
    int code = -1;
    double dval = 0.0;
    if (ctl && more && status == 0) {
        double bk = -1.0;
        int bp = kNoCand, nprobe = 0;
        // (candidates and their table entries were loaded at the start of the kernel)
#pragma unroll
        for (int kk = 0; kk < PJ; kk++) {
            const int k = lane + 64 * kk;
            const double v = s.key[ci[kk]];
            const double fl = floor(v), ce = ceil(v);
            const double dist = fmin(v - fl, ce - v);
            const bool frac = cv[kk] && dist > kVarEps;
            const double key = g.rule == 0 ? dist : fmin(ccr[kk] * (ce - v), ccl[kk] * (v - fl));
            keep_max(bk, bp, key, k, frac && che[kk]);
            nprobe += __popcll(__ballot(frac && !che[kk]));
        }
        // (every key is >= +0: a distance to an integer, or a product of non-negative costs and distances)
        double km;
        const int win = wave_argmax_pos(bk, bp, km);
        if (win != kNoCand && nprobe == 0) {
            // the winner's variable, value and costs sit in lane win % 64, slot win / 64
            const int wl = win & 63, wk = win >> 6;
            int t_i;
            double t_l, t_r;
            MIPX_PICK(t_i, ci, PJ, wk);
            MIPX_PICK(t_l, ccl, PJ, wk);
            MIPX_PICK(t_r, ccr, PJ, wk);
            dvar = __builtin_amdgcn_readlane(t_i, wl);
            const double wcl = readlane_f64(t_l, wl), wcr = readlane_f64(t_r, wl);
            const double v = uniform_f64(s.key[dvar]);
            const double fl = floor(v), ce = ceil(v);
            // towards the side the rule expects to cost less (most fractional: the nearer one)
            if (g.rule == 0) ddir = (v - fl <= ce - v) ? 0 : 1;
            else ddir = (wcl * (v - fl) <= wcr * (ce - v)) ? 0 : 1;
            dbound = ddir == 0 ? fl : ce;
            dval = v;
            bool mine = false;  // a bound change in place needs the variable basic
#pragma unroll
            for (int kk = 0; kk < PI; kk++) mine |= (lane + 64 * kk < m) && (rM[kk] >> 2) == dvar;
            if (__ballot(mine) != 0ull) code = dvar;
        }
    }
    KPROF_OUT_MARK(12);
    __syncthreads();
    KPROF_OUT_MARK(13);
    if (ctl) {
        const double objv = uniform_f64(s.objv);
        if (!(status == 0 && objv < g.dive_cutoff)) code = -1;
        if (lane == 0) {
            if (g.obj) g.obj[onode] = status == 1 ? INF : objv;
            if (g.status) g.status[onode] = status;
            if (g.iters) g.iters[onode] = iters;
            if (g.npivots) g.npivots[onode] = npiv;
            if (code >= 0) {
                const size_t di = (size_t)pass * (size_t)g.dive_off + node;
                g.dive_var[di] = dvar;
                g.dive_dir[di] = ddir;
                g.dive_val[di] = dval;
            }
            s.dive_code = code;
        }
    }
    if (!more) break;
    KPROF_OUT_MARK(14);
    __syncthreads();
    }
    }
    if (__builtin_amdgcn_readfirstlane(s.dive_code) < 0) break;
    KPROF_MARK(11);  // outputs of the node + the branching rule
    pass++;
    onode += (size_t)g.dive_off;
    iters = 0;
    npiv = 0;
    }
}

template <int NW, int R, int C, int MP, bool DIVE = false, bool CUTS = false>
__global__ __launch_bounds__(64 * (NW + 1)) void lp_dual_simplex(LpArgs g) {
    __shared__ Smem<NW, R, C, MP> s;
    if (threadIdx.x < 64) lp_dual_simplex_role<NW, R, C, MP, DIVE, true, CUTS>(g, s);   // wave 0: control
    else lp_dual_simplex_role<NW, R, C, MP, DIVE, false, CUTS>(g, s);                   // tableau waves
}

#undef MIPX_PUBLISH_COL
#undef MIPX_PUBLISH_NEXT_COL
#undef MIPX_READ_COL
#undef MIPX_AWAIT_COL
#undef MIPX_EXTRACT_ROW
#undef MIPX_UPDATE_T
#undef MIPX_UPDATE_COLS
#undef MIPX_PREP_COLS
#undef MIPX_PICK
#undef MIPX_ROWSUMS
#undef MIPX_ROWSUM_STEP
#undef MIPX_ROW
#undef MIPX_COL
#undef MIPX_LEAVE_SELECT
#undef MIPX_LEAVE_FETCH
#undef MIPX_ROW_FOLD
#undef MIPX_TERM_SQ
#undef MIPX_TERM_ROW
#undef MIPX_EXACT_WEIGHTS
#undef MIPX_ROW_LDS
#undef MIPX_ROW_REG

}  // namespace mipx
