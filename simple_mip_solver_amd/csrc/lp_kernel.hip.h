// lp_kernel.hip.h -- K1: batched bounded dual simplex, one workgroup per node LP, gfx950.
//
// Replaces what the reference asks Clp to do in BaseNode._bound_lp / _strong_branch
// (simple_mip_solver/nodes/base_node.py:273, :645-646).  Not a translation of anything: the
// reference has no kernel.  Design:
//
//   * The condensed simplex tableau T (m x n, f64) lives in VGPRs for the whole solve: thread
//     (bi, bj) of a TBI x TBJ thread grid owns T[bi + TBI*ii][bj + TBJ*jj], ii < R, jj < C.
//     256x128 -> 16x32 threads x 8x8 doubles = 256 KiB of registers on one CU; it never touches
//     HBM again after the initial coalesced read of A.
//   * The interleaved ownership makes every LDS access of the per-pivot vectors (pivot row rho,
//     pivot column alpha) conflict-free (consecutive lanes -> consecutive 8-byte words, or a
//     half-wave broadcast), and lets work scale with ceil(m/TBI), ceil(n/TBJ).
//   * Bounds by variable index and the nonbasic values live in LDS.
//   * Row/column extraction uses wave-uniform (SGPR) local indices so register arrays are only
//     ever indexed statically (no scratch).
//   * The borders live in the registers of two owner waves (a column wave and a row wave, on
//     different SIMDs); selections (leaving row, Harris ratio test) are single-wave DPP
//     reductions by the owner, published through LDS mailbox words.  The next leaving row is
//     picked while the other waves are still in the rank-1 update.
//   * Arithmetic is IEEE f64 with explicit fma and true division, compiled with
//     -ffp-contract=off, following the canonical operation order documented in
//     oracle/mipx_oracle.c so results are bit-identical to the CPU oracle.
#pragma once
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
    unsigned long long *prof;  // MIPX_KPROF builds only: per-section cycle totals of wave 0
};

#ifdef MIPX_KPROF
#define KPROF_MARK(k)                                  \
    do {                                               \
        const unsigned long long t_ = clock64();       \
        if (tid == 0) s.prof[k] += t_ - tprev;         \
        tprev = t_;                                    \
    } while (0)
#else
#define KPROF_MARK(k) do { } while (0)
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
// wave-wide argmax / argmin over keys >= +0 (lanes without a candidate carry payload kNoCand)
__device__ __forceinline__ int wave_argmax_pos(double key, int payload, double &kmax) {
    const bool valid = payload != kNoCand;
    const unsigned hi = valid ? (unsigned)__double2hiint(key) : 0u;
    const unsigned lo = valid ? (unsigned)__double2loint(key) : 0u;
    const unsigned hm = wave_max_u32(hi);
    const unsigned lm = wave_max_u32(hi == hm ? lo : 0u);
    kmax = __hiloint2double((int)hm, (int)lm);
    return wave_pick(valid & (hi == hm) & (lo == lm), payload);
}
__device__ __forceinline__ int wave_argmin_pos(double key, int payload, double &kmin) {
    const bool valid = payload != kNoCand;
    const unsigned hi = valid ? (unsigned)__double2hiint(key) : 0x7ff00000u;
    const unsigned lo = valid ? (unsigned)__double2loint(key) : 0u;
    const unsigned hm = wave_min_u32(hi);
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
// value of lane ^ 16 (ds_swizzle bit mode: and 0x1f, xor 0x10)
__device__ __forceinline__ double swz16_f64(double v) {
    const int lo = __builtin_amdgcn_ds_swizzle(__double2loint(v), 0x401f);
    const int hi = __builtin_amdgcn_ds_swizzle(__double2hiint(v), 0x401f);
    return __hiloint2double(hi, lo);
}
// fold-in-half sum over the TBJ lanes that share a tableau row (valid in the lane with bj == 0):
// the pairing (lane, lane + h), h = TBJ/2 .. 1, is the canonical summation tree
template <int TBJ>
__device__ __forceinline__ double fold_bj(double v) {
    static_assert(TBJ == 16 || TBJ == 32, "row groups of 16 or 32 lanes");
    if (TBJ == 32) v = v + swz16_f64(v);
    v = v + dpp_f64<0x108, 0xf>(v);  // row_shl:8
    v = v + dpp_f64<0x104, 0xf>(v);  // row_shl:4
    v = v + dpp_f64<0x102, 0xf>(v);  // row_shl:2
    v = v + dpp_f64<0x101, 0xf>(v);  // row_shl:1
    return v;
}

template <int MP, int NP>
struct Smem {
    double row[NP];       // extracted pivot row T[r][.]
    double alpha[2][MP];  // extracted pivot column T[.][q] (double-buffered: the refactorisation
                          // publishes the next column while the current one is still being read)
    double lo[NP];        // structural bounds by variable index
    double up[NP];
    double va[NP];        // nonbasic values a + b*M by column
    double vb[NP];
    double d[NP];         // d, beta0, ba, bb, the basis lists and sides are staged here at setup and
    double key[NP];       //   for the outputs; in between they live in their owners' registers
    double beta0[MP];
    double ba[MP];
    double bb[MP];
    int bvar[MP];
    int nvar[NP];
    int side[NP];         // 0 lower, 1 upper, 2 fake upper
    int wlist[NP];        // columns of the variables the warm start wants basic, ascending variable
    int nw;
    int ci[8];            // mailboxes: row wave / column wave -> everybody
    double cd[4];
    int pos[NP + MP];     // column of each variable in the starting tableau, -1 if basic
    int8_t wantb[NP + MP];
    int8_t atup[NP + MP];
#ifdef MIPX_KPROF
    unsigned long long prof[16];
#endif
};
// mailbox words
enum { kCmd = 0, kRow = 1, kLv = 2, kNewSide = 3, kCol = 4, kEv = 5, kFake = 6, kBland = 7 };
enum { kPinv = 0, kLa = 1, kLb = 2 };

// ---- building blocks of the kernel body.  Macros, not lambdas: the register tableau T must be
// seen as plain local arrays with static indices from the first optimisation pass on, or it is
// demoted to scratch memory.
// T[.][q] -> s.alpha[buf]
#define MIPX_EXTRACT_COL(q_, buf_)                                                          \
    do {                                                                                    \
        const int qb_ = (q_) % TBJ, ql_ = (q_) / TBJ;                                       \
        if (bj == qb_) {                                                                    \
            _Pragma("unroll") for (int jj = 0; jj < C; jj++) if (jj == ql_) {               \
                _Pragma("unroll") for (int ii = 0; ii < R; ii++)                            \
                    s.alpha[buf_][bi + TBI * ii] = T[ii][jj];                               \
            }                                                                               \
        }                                                                                   \
    } while (0)
// T[r][.] -> s.row
#define MIPX_EXTRACT_ROW(r_)                                                                \
    do {                                                                                    \
        const int rb_ = (r_) % TBI, rl_ = (r_) / TBI;                                       \
        if (bi == rb_) {                                                                    \
            _Pragma("unroll") for (int ii = 0; ii < R; ii++) if (ii == rl_) {               \
                _Pragma("unroll") for (int jj = 0; jj < C; jj++)                            \
                    s.row[bj + TBJ * jj] = T[ii][jj];                                       \
            }                                                                               \
        }                                                                                   \
    } while (0)
// rank-1 update of the register tableau; row r is in s.row, column q in s.alpha[buf].  Row r and
// column q come out of the fma sweep as junk and are overwritten right after it.
#define MIPX_UPDATE_T(r_, q_, pinv_, buf_)                                                  \
    do {                                                                                    \
        const int rb_ = (r_) % TBI, rl_ = (r_) / TBI;                                       \
        const int qb_ = (q_) % TBJ, ql_ = (q_) / TBJ;                                       \
        double al[R], rh[C];                                                                \
        _Pragma("unroll") for (int ii = 0; ii < R; ii++) al[ii] = s.alpha[buf_][bi + TBI * ii]; \
        _Pragma("unroll") for (int jj = 0; jj < C; jj++) rh[jj] = s.row[bj + TBJ * jj] * (pinv_); \
        _Pragma("unroll") for (int ii = 0; ii < R; ii++) {                                  \
            _Pragma("unroll") for (int jj = 0; jj < C; jj++)                                \
                T[ii][jj] = fma(-al[ii], rh[jj], T[ii][jj]);                                \
        }                                                                                   \
        if (bj == qb_) { /* column q <- -alpha * (1/p) */                                   \
            _Pragma("unroll") for (int jj = 0; jj < C; jj++) if (jj == ql_) {               \
                _Pragma("unroll") for (int ii = 0; ii < R; ii++) T[ii][jj] = -al[ii] * (pinv_); \
            }                                                                               \
        }                                                                                   \
        if (bi == rb_) { /* row r <- row * (1/p), and 1/p at the pivot position */          \
            _Pragma("unroll") for (int ii = 0; ii < R; ii++) if (ii == rl_) {               \
                _Pragma("unroll") for (int jj = 0; jj < C; jj++)                            \
                    T[ii][jj] = (bj == qb_ && jj == ql_) ? (pinv_) : rh[jj];                \
            }                                                                               \
        }                                                                                   \
    } while (0)
// element k (wave-uniform) of a short register array: a select chain, no scratch
#define MIPX_PICK(dst_, arr_, n_, k_)                                                       \
    do {                                                                                    \
        dst_ = arr_[0];                                                                     \
        _Pragma("unroll") for (int t_ = 1; t_ < n_; t_++) dst_ = (k_) == t_ ? arr_[t_] : dst_; \
    } while (0)
// the column wave's half of a pivot: d, and the variable that takes over column q
#define MIPX_UPDATE_COLS(q_, pinv_, lv_, meta_)                                             \
    do {                                                                                    \
        double dq_;                                                                         \
        MIPX_PICK(dq_, cD, PJ, (q_) >> 6);                                                  \
        dq_ = readlane_f64(dq_, (q_)&63);                                                   \
        _Pragma("unroll") for (int kk = 0; kk < PJ; kk++) {                                 \
            const int j = lane + 64 * kk;                                                   \
            const double rho_ = s.row[j] * (pinv_);                                         \
            const double upd_ = fma(-dq_, rho_, cD[kk]);                                    \
            cD[kk] = j == (q_) ? -dq_ * (pinv_) : upd_;                                     \
            cM[kk] = j == (q_) ? (((lv_) << 3) | (meta_)) : cM[kk];                         \
        }                                                                                   \
    } while (0)

// (a) the row wave's choice of the leaving row, or of the end of the solve, published for everybody:
// largest violation (violations of the symbolic bound M first), ties -> lowest variable index
#define MIPX_LEAVE_SELECT()                                                                  \
    do {                                                                                     \
        int blevel = 0, bp = kNoCand;                                                            \
        double bk = -1.0;                                                                        \
        bool anybb = false, anym = false;                                                        \
_Pragma("unroll")                                                                                \
        for (int kk = 0; kk < PI; kk++) anym |= (lane + 64 * kk < m) & (fabs(rBb[kk]) > kBTol);  \
        if (!__ballot(anym)) {                                                                   \
            /* the usual case, no symbolic-M part anywhere: plain bound violations */            \
_Pragma("unroll")                                                                                \
            for (int kk = 0; kk < PI; kk++) {                                                    \
                const int i = lane + 64 * kk;                                                    \
                const double lo = rLo[kk], up = rUp[kk], a = rBa[kk];                            \
                const bool lowv = a < lo - kPTol;                                                \
                const bool upv = !lowv & (a > up + kPTol); /* up = +inf never fires */           \
                const double viol = bland ? 0.0 : (lowv ? lo - a : a - up);                      \
                const int pay = ((rM[kk] >> 2) << 16) | (lowv ? 0 : 0x8000) | i;                 \
                keep_max(bk, bp, viol, pay, (i < m) & (lowv | upv));                             \
            }                                                                                    \
            blevel = bp != kNoCand ? 1 : 0;                                                      \
        } else {                                                                                 \
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
                    if (bland && level > 0) { level = 1; viol = 0.0; }                           \
                    anybb |= bM > kBTol;                                                         \
                    const int pay = (v << 16) | (sg < 0 ? 0x8000 : 0) | i;                       \
                    const bool up_lvl = level > blevel;                                          \
                    const bool same = level == blevel && level > 0 &&                            \
                                      (viol > bk || (viol == bk && pay < bp));                   \
                    if (up_lvl || same) { blevel = level; bk = viol; bp = pay; }                 \
                }                                                                                \
            }                                                                                    \
        }                                                                                        \
        const int lvl = __ballot(blevel == 2) ? 2 : (__ballot(blevel == 1) ? 1 : 0);             \
        int cmd = 0, win = 0;                                                                    \
        if (lvl == 0) {                                                                          \
            cmd = (__ballot(anybb) || nfk > 0) ? 3 : 1;                                          \
        } else if ((g.max_iter > 0 && iters >= g.max_iter) || iters >= cap) {                    \
            cmd = 4;                                                                             \
        } else {                                                                                 \
            double km;                                                                           \
            win = wave_argmax_pos(bk, blevel == lvl ? bp : kNoCand, km);                         \
        }                                                                                        \
        if (cmd == 0) {                                                                          \
            const int rr = win & 0x7fff, rl = rr & 63, rk = rr >> 6;                             \
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
            const int lvv = __builtin_amdgcn_readlane(tm, rl) >> 2;                              \
            double la_, lb_;                                                                     \
            int newside;                                                                         \
            if (!(win & 0x8000)) { la_ = lo; lb_ = 0.0; newside = 0; }                           \
            else if (!isinf(up)) { la_ = up; lb_ = 0.0; newside = 1; }                           \
            else { la_ = 0.0; lb_ = 1.0; newside = 2; }                                          \
            if (lane == 0) {                                                                     \
                s.ci[kRow] = win;                                                                \
                s.ci[kLv] = lvv;                                                                 \
                s.ci[kNewSide] = newside | (lo == up ? 4 : 0);                                   \
                s.cd[kLa] = la_;                                                                 \
                s.cd[kLb] = lb_;                                                                 \
            }                                                                                    \
        }                                                                                        \
        if (lane == 0) s.ci[kCmd] = cmd;                                                         \
    } while (0)

// Roles.  Every wave holds a slab of the tableau and takes part in the rank-1 update.  On top:
//   * the column wave (wave 0) owns the column border -- reduced cost d_j, nonbasic variable, side,
//     fixed flag -- in registers, NP/64 columns per lane, and runs the ratio test;
//   * the row wave (wave 1; wave 0 on a single-wave tile) owns the row border -- beta0, the basic
//     values a + b*M, the basic variable and its bounds -- and picks the leaving row.
// They sit on different SIMDs and talk through LDS mailbox words; each hand-over costs one barrier.
template <int TBI, int TBJ, int R, int C>
__global__ __launch_bounds__(TBI *TBJ) void lp_dual_simplex(LpArgs g) {
    constexpr int NT = TBI * TBJ;
    constexpr int MP = TBI * R;
    constexpr int NP = TBJ * C;
    static_assert((NP & (NP - 1)) == 0, "padded column count must be a power of two");
    static_assert(NT % 64 == 0 && NP % 64 == 0, "whole waves");
    constexpr int PI = (MP + 63) / 64;   // rows per lane of the row wave
    constexpr int PJ = NP / 64;          // columns per lane of the column wave
    constexpr int RW = NT > 64 ? 1 : 0;  // the row wave
    __shared__ Smem<MP, NP> s;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool isC = wave == 0, isR = wave == RW;
    const int bi = tid / TBJ;
    const int bj = tid % TBJ;
    const int m = g.m, n = g.n;
    const int nv = n + m;
    const double INF = __builtin_huge_val();

    // one workgroup per node LP (no grid-stride loop: a loop here makes the compiler hoist every
    // per-element predicate and address of the setup across the whole solve -> register spills)
    const int node = blockIdx.x;
    if (node >= g.batch) return;
#ifdef MIPX_KPROF
    if (tid < 16) s.prof[tid] = 0;
    unsigned long long tprev = clock64();
#endif
    double T[R][C];
    // border registers of the owner waves
    double cD[PJ];  // reduced cost
    int cM[PJ];     // nonbasic variable << 3 | fixed << 2 | side
    double rB0[PI], rBa[PI], rBb[PI], rLo[PI], rUp[PI];
    int rM[PI];     // basic variable << 2 | pivoted by the refactorisation << 1 | wanted basic
    const size_t src = g.slot ? (size_t)g.slot[node] : (size_t)node;
    const double *gA = g.A + (size_t)node * g.A_stride;
    const double *gb = g.b + (size_t)node * g.b_stride;
    const double *gc = g.c + (size_t)node * g.c_stride;
    const double *lk = g.l + src * n;
    const double *uk = g.u + src * n;
    const int8_t *vin = g.vstat_in ? g.vstat_in + src * nv : nullptr;

    // ---- 0. T = -A, beta0 = -b, d = c, slack basis (or the anchor's tableau state) -----------
    const bool anchored = g.anchor_T != nullptr && vin != nullptr;
    const double sgn = anchored ? 1.0 : -1.0;
    {
        // every load is issued unconditionally from a clamped address (64 independent requests in
        // flight); the padding is zeroed afterwards
        const double *tsrc = anchored ? g.anchor_T : gA;
        int joff[C];
#pragma unroll
        for (int jj = 0; jj < C; jj++) joff[jj] = min(bj + TBJ * jj, n - 1);
#pragma unroll
        for (int ii = 0; ii < R; ii++) {
            const double *arow = tsrc + (size_t)min(bi + TBI * ii, m > 0 ? m - 1 : 0) * n;
#pragma unroll
            for (int jj = 0; jj < C; jj++) T[ii][jj] = m > 0 ? arow[joff[jj]] : 0.0;
        }
#pragma unroll
        for (int ii = 0; ii < R; ii++) {
#pragma unroll
            for (int jj = 0; jj < C; jj++)
                T[ii][jj] = (bi + TBI * ii < m && bj + TBJ * jj < n) ? sgn * T[ii][jj] : 0.0;
        }
    }
#pragma unroll 1
    for (int i = tid; i < MP; i += NT) {
        s.beta0[i] = i < m ? (anchored ? g.anchor_vec[n + i] : -gb[i]) : 0.0;
        s.bvar[i] = i < m ? (anchored ? g.anchor_idx[n + i] : n + i) : -1;
    }
#pragma unroll 1
    for (int j = tid; j < NP; j += NT) {
        s.d[j] = j < n ? (anchored ? g.anchor_vec[j] : gc[j]) : 0.0;
        s.nvar[j] = j < n ? (anchored ? g.anchor_idx[j] : j) : -1;
        s.lo[j] = j < n ? lk[j] : 0.0;
        s.up[j] = j < n ? uk[j] : 0.0;
        s.va[j] = 0.0;
        s.vb[j] = 0.0;
    }
#pragma unroll 1
    for (int v = tid; v < NP + MP; v += NT) {
        int8_t st = (vin && v < nv) ? vin[v] : (int8_t)0;
        s.wantb[v] = st == 1;
        s.atup[v] = st == 2;
        s.pos[v] = -1;
    }
    __syncthreads();
    for (int j = tid; j < n; j += NT) s.pos[s.nvar[j]] = j;
    __syncthreads();
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
    // the owners take their borders into registers
#pragma unroll
    for (int kk = 0; kk < PJ; kk++) {
        const int j = lane + 64 * kk;
        const int v = s.nvar[j];
        cD[kk] = s.d[j];
        const bool fix = v >= 0 && v < n && s.lo[v < 0 ? 0 : (v < n ? v : 0)] == s.up[v < 0 ? 0 : (v < n ? v : 0)];
        cM[kk] = (v << 3) | (fix ? 4 : 0);
    }
#pragma unroll
    for (int kk = 0; kk < PI; kk++) {
        const int i = lane + 64 * kk;
        const int v = i < MP ? s.bvar[i] : -1;
        const bool st = v >= 0 && v < n;
        rB0[kk] = i < MP ? s.beta0[i] : 0.0;
        rBa[kk] = 0.0;
        rBb[kk] = 0.0;
        rLo[kk] = st ? s.lo[st ? v : 0] : 0.0;
        rUp[kk] = st ? s.up[st ? v : 0] : INF;
        rM[kk] = (v << 2) | ((v >= 0 && s.wantb[v < 0 ? 0 : v]) ? 1 : 0);
    }
    __syncthreads();
    const int nw = __builtin_amdgcn_readfirstlane(s.nw);

    int npiv = 0, iters = 0, status = -1;
    const int cap = 100 * (m + n) + 1000;
    int ab = 0;     // alpha buffer in use
    int degen = 0;  // consecutive degenerate steps (column wave); > m+n -> Bland's rule
    int nfake = 0;  // nonbasic columns at the symbolic bound M (column wave)
    KPROF_MARK(15);

    // ---- 1. refactor: pivot the wanted variables into the basis, ascending variable -----------
    if (vin) {
        if (nw > 0) MIPX_EXTRACT_COL(__builtin_amdgcn_readfirstlane(s.wlist[0]), 0);
        __syncthreads();
        KPROF_MARK(8);
        for (int w = 0; w < nw;) {
            const int q = __builtin_amdgcn_readfirstlane(s.wlist[w]);
            if (isC) {  // the entering variable, for the row wave
                int cm;
                MIPX_PICK(cm, cM, PJ, q >> 6);
                const int ev = __builtin_amdgcn_readlane(cm, q & 63) >> 3;
                if (lane == 0) s.ci[kEv] = ev;
            }
            if (isR) {  // leaving row: largest |T_iq| among rows whose basic variable is not wanted
                double k1 = -1.0, k2 = -1.0;  // (fallback: wanted, but not pivoted in yet)
                int p1 = kNoCand, p2 = kNoCand;
                double rc[PI];
#pragma unroll
                for (int kk = 0; kk < PI; kk++) {
                    const int i = lane + 64 * kk;
                    const double av = i < MP ? s.alpha[ab][i < MP ? i : 0] : 0.0;
                    const double a = fabs(av);
                    const bool ok = i < m && a > kPivTol;
                    const bool wanted = rM[kk] & 1, ent = rM[kk] & 2;
                    rc[kk] = 1.0 / av;  // 1/p of every candidate, off the selection's critical path
                    keep_max(k1, p1, a, i, ok & !wanted);
                    keep_max(k2, p2, a, i, ok & wanted & !ent);
                }
                double km;
                int rr = wave_argmax_pos(k1, p1, km);
                if (rr == kNoCand) rr = wave_argmax_pos(k2, p2, km);
                if (rr != kNoCand) {
                    double rcv;
                    int rmv;
                    MIPX_PICK(rcv, rc, PI, rr >> 6);
                    MIPX_PICK(rmv, rM, PI, rr >> 6);
                    const double pinv = readlane_f64(rcv, rr & 63);
                    const int lv = __builtin_amdgcn_readlane(rmv, rr & 63) >> 2;
                    const bool fix = lv < n && s.lo[lv < n ? lv : 0] == s.up[lv < n ? lv : 0];
                    if (lane == 0) {
                        s.ci[kRow] = rr;
                        s.ci[kLv] = lv;
                        s.ci[kNewSide] = fix ? 4 : 0;
                        s.cd[kPinv] = pinv;
                    }
                } else if (lane == 0) {
                    s.ci[kRow] = -1;
                }
            }
            __syncthreads();
            KPROF_MARK(9);
            const int r = __builtin_amdgcn_readfirstlane(s.ci[kRow]);
            w++;
            const int qn = w < nw ? __builtin_amdgcn_readfirstlane(s.wlist[w < nw ? w : 0]) : -1;
            // (a singular column -- no usable pivot row -- stays nonbasic: both halves are skipped)
            const double pinv = uniform_f64(s.cd[kPinv]);
            const int lv = __builtin_amdgcn_readfirstlane(s.ci[kLv]);
            const int meta = __builtin_amdgcn_readfirstlane(s.ci[kNewSide]);
            const int ev = __builtin_amdgcn_readfirstlane(s.ci[kEv]);
            if (r >= 0) MIPX_EXTRACT_ROW(r);
            __syncthreads();
            KPROF_MARK(10);
            if (r >= 0) {
                if (isR) {
                    double b0r;
                    MIPX_PICK(b0r, rB0, PI, r >> 6);
                    const double rhon = readlane_f64(b0r, r & 63) * pinv;
                    const double elo = ev < n ? s.lo[ev < n ? ev : 0] : 0.0;
                    const double eup = ev < n ? s.up[ev < n ? ev : 0] : INF;
                    const int em = (ev << 2) | 2 | (s.wantb[ev] ? 1 : 0);
#pragma unroll
                    for (int kk = 0; kk < PI; kk++) {
                        const int i = lane + 64 * kk;
                        const double a = i < MP ? s.alpha[ab][i < MP ? i : 0] : 0.0;
                        const bool pr = i == r;
                        rB0[kk] = pr ? rhon : fma(-a, rhon, rB0[kk]);
                        rM[kk] = pr ? em : rM[kk];
                        rLo[kk] = pr ? elo : rLo[kk];
                        rUp[kk] = pr ? eup : rUp[kk];
                    }
                }
                if (isC) MIPX_UPDATE_COLS(q, pinv, lv, meta);
                MIPX_UPDATE_T(r, q, pinv, ab);
                npiv++;
            }
            if (qn >= 0) MIPX_EXTRACT_COL(qn, ab ^ 1);  // next column, from registers already updated
            ab ^= 1;
            __syncthreads();
            KPROF_MARK(11);
        }
    }

    if (!(vin && g.refactor_only)) {
        // ---- 2. nonbasic sides and values (column wave), then the basic values ----------------
        if (isC) {
            int fakes = 0;
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
                fakes += __popcll(__ballot(side == 2));
            }
            nfake = fakes;
            if (lane == 0) s.ci[kFake] = fakes;
        }
        __syncthreads();
        int nfk = __builtin_amdgcn_readfirstlane(s.ci[kFake]);
        {
            double va[C], vb[C];
#pragma unroll
            for (int jj = 0; jj < C; jj++) {
                va[jj] = s.va[bj + TBJ * jj];
                vb[jj] = s.vb[bj + TBJ * jj];
            }
#pragma unroll
            for (int ii = 0; ii < R; ii++) {
                double pa[C];
#pragma unroll
                for (int jj = 0; jj < C; jj++) pa[jj] = T[ii][jj] * va[jj];
#pragma unroll
                for (int h = C / 2; h >= 1; h >>= 1) {
#pragma unroll
                    for (int jj = 0; jj < h; jj++) pa[jj] = pa[jj] + pa[jj + h];
                }
                const double sa = fold_bj<TBJ>(pa[0]);
                double sb = 0.0;
                if (nfk != 0) {  // the M parts: all zero unless some nonbasic sits at the symbolic bound
                    double pb[C];
#pragma unroll
                    for (int jj = 0; jj < C; jj++) pb[jj] = T[ii][jj] * vb[jj];
#pragma unroll
                    for (int h = C / 2; h >= 1; h >>= 1) {
#pragma unroll
                        for (int jj = 0; jj < h; jj++) pb[jj] = pb[jj] + pb[jj + h];
                    }
                    sb = fold_bj<TBJ>(pb[0]);
                }
                if (bj == 0) {
                    s.ba[bi + TBI * ii] = sa;
                    s.bb[bi + TBI * ii] = 0.0 - sb;
                }
            }
        }
        __syncthreads();
        if (isR) {
#pragma unroll
            for (int kk = 0; kk < PI; kk++) {
                const int i = lane + 64 * kk;
                rBa[kk] = i < MP ? rB0[kk] - s.ba[i < MP ? i : 0] : 0.0;
                rBb[kk] = i < MP ? s.bb[i < MP ? i : 0] : 0.0;
            }
        }
        KPROF_MARK(7);

        // ---- 3. dual simplex ------------------------------------------------------------------
        // The row wave picks the leaving row; row r is published; the column wave runs the ratio
        // test; column q is published; then the pivot: borders by their owners -- the row wave goes
        // straight on to the next leaving row -- and the tableau by everybody.
        bool bland = false;
        int r = 0, q = 0, sigma = 1, lv = 0, meta = 0, ev = 0;
        double la = 0.0, lb = 0.0, pinv = 0.0, vaq = 0.0, vbq = 0.0;
        double sel_b0 = 0.0, sel_ba = 0.0, sel_bb = 0.0;  // border values of row r (row wave)
        if (isR) MIPX_LEAVE_SELECT();
        __syncthreads();
        for (;;) {
            {
                const int cmd = __builtin_amdgcn_readfirstlane(s.ci[kCmd]);
                if (cmd) { status = cmd == 1 ? 0 : cmd == 3 ? 2 : 3; break; }
                const int win = __builtin_amdgcn_readfirstlane(s.ci[kRow]);
                r = win & 0x7fff;
                sigma = (win & 0x8000) ? -1 : 1;
                lv = __builtin_amdgcn_readfirstlane(s.ci[kLv]);
                meta = __builtin_amdgcn_readfirstlane(s.ci[kNewSide]);
                la = uniform_f64(s.cd[kLa]);
                lb = uniform_f64(s.cd[kLb]);
            }
            MIPX_EXTRACT_ROW(r);  // (b)
            __syncthreads();
            KPROF_MARK(1);
            if (isC) {  // (c) Harris ratio test on row r
                double aa[PJ], dje[PJ], rc[PJ];
                bool el[PJ];
                double k1 = INF;
                int p1 = kNoCand;
                const double tol = bland ? 0.0 : kDTol;  // Bland: the textbook ratio dj / |a|
#pragma unroll
                for (int kk = 0; kk < PJ; kk++) {
                    const int j = lane + 64 * kk;
                    const double rv = s.row[j];
                    const double a = sigma < 0 ? -rv : rv;
                    const int sd = cM[kk] & 3;
                    el[kk] = ((cM[kk] & 4) == 0) & (sd == 0 ? (a < -kPivTol) : (a > kPivTol));
                    dje[kk] = sd == 0 ? fmax(cD[kk], 0.0) : fmax(-cD[kk], 0.0);
                    aa[kk] = fabs(a);
                    const double key = (dje[kk] + tol) / aa[kk];  // unconditionally: no divergent branch
                    rc[kk] = 1.0 / rv;                            // 1/p of every candidate, likewise
                    keep_min(k1, p1, key, ((cM[kk] >> 3) << 16) | j, el[kk]);
                }
                double thmax;
                const int w1 = wave_argmin_pos(k1, p1, thmax);
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
                        keep_max(k2, p2, aa[kk], ((cM[kk] >> 3) << 16) | j, ok);
                    }
                    double amax;
                    qq = wave_argmax_pos(k2, p2, amax) & 0xffff;
                }
                if (qq >= 0) {
                    const int ql = qq & 63, qk = qq >> 6;
                    double t0, t1;
                    int tm;
                    MIPX_PICK(t0, dje, PJ, qk);
                    MIPX_PICK(t1, rc, PJ, qk);
                    MIPX_PICK(tm, cM, PJ, qk);
                    const double djq = readlane_f64(t0, ql);
                    degen = djq <= kDTol ? degen + 1 : 0;
                    const int cm = __builtin_amdgcn_readlane(tm, ql);
                    nfake += ((meta & 3) == 2 ? 1 : 0) - ((cm & 3) == 2 ? 1 : 0);
                    const double pv = readlane_f64(t1, ql);
                    if (lane == 0) {
                        s.ci[kEv] = cm >> 3;
                        s.ci[kFake] = nfake;
                        s.ci[kBland] = degen > m + n;
                        s.cd[kPinv] = pv;
                    }
                }
                if (lane == 0) s.ci[kCol] = qq;
            }
            __syncthreads();
            KPROF_MARK(3);
            q = __builtin_amdgcn_readfirstlane(s.ci[kCol]);
            if (q < 0) { status = 1; break; }  // no entering column: primal infeasible
            pinv = uniform_f64(s.cd[kPinv]);
            ev = __builtin_amdgcn_readfirstlane(s.ci[kEv]);
            nfk = __builtin_amdgcn_readfirstlane(s.ci[kFake]);
            bland = __builtin_amdgcn_readfirstlane(s.ci[kBland]) != 0;
            vaq = uniform_f64(s.va[q]);
            vbq = uniform_f64(s.vb[q]);
            MIPX_EXTRACT_COL(q, ab);  // (d)
            __syncthreads();
            KPROF_MARK(4);
            iters++;
            npiv++;
            if (isR) {  // basic values after the pivot on (r, q), then the next leaving row
                const double rhon = sel_b0 * pinv;
                const double ta = (sel_ba - la) * pinv, tb = (sel_bb - lb) * pinv;
                const double elo = ev < n ? s.lo[ev < n ? ev : 0] : 0.0;
                const double eup = ev < n ? s.up[ev < n ? ev : 0] : INF;
#pragma unroll
                for (int kk = 0; kk < PI; kk++) {
                    const int i = lane + 64 * kk;
                    const double a = i < MP ? s.alpha[ab][i < MP ? i : 0] : 0.0;
                    const bool pr = i == r;
                    rB0[kk] = pr ? rhon : fma(-a, rhon, rB0[kk]);
                    rBa[kk] = pr ? vaq + ta : fma(-a, ta, rBa[kk]);
                    rBb[kk] = pr ? vbq + tb : fma(-a, tb, rBb[kk]);
                    rM[kk] = pr ? (ev << 2) : rM[kk];
                    rLo[kk] = pr ? elo : rLo[kk];
                    rUp[kk] = pr ? eup : rUp[kk];
                }
                MIPX_LEAVE_SELECT();
            }
            if (isC) {
                MIPX_UPDATE_COLS(q, pinv, lv, meta);
                if (lane == 0) { s.va[q] = la; s.vb[q] = lb; }
            }
            MIPX_UPDATE_T(r, q, pinv, ab);
            __syncthreads();
            KPROF_MARK(0);
        }
    } else {
        status = 3;
    }


    // ---- 4. outputs ---------------------------------------------------------------------------
#ifdef MIPX_KPROF
    KPROF_MARK(6);
    if (g.prof && tid == 0) {
        for (int k = 0; k < 12; k++) atomicAdd(&g.prof[k], s.prof[k]);
        atomicAdd(&g.prof[12], (unsigned long long)iters);
        atomicAdd(&g.prof[13], (unsigned long long)(npiv - iters));
        atomicAdd(&g.prof[14], 1ull);
        atomicAdd(&g.prof[15], s.prof[15]);
    }
#endif
    if (isC) {
#pragma unroll
        for (int kk = 0; kk < PJ; kk++) {
            const int j = lane + 64 * kk;
            s.d[j] = cD[kk];
            s.nvar[j] = cM[kk] >> 3;
            s.side[j] = cM[kk] & 3;
        }
    }
    if (isR) {
#pragma unroll
        for (int kk = 0; kk < PI; kk++) {
            const int i = lane + 64 * kk;
            if (i < MP) {
                s.beta0[i] = rB0[kk];
                s.ba[i] = rBa[kk];
                s.bb[i] = rBb[kk];
                s.bvar[i] = rM[kk] >> 2;
            }
        }
    }
    __syncthreads();
    // assemble x by variable index in s.key
    for (int j = tid; j < n; j += NT) {
        const int v = s.nvar[j];
        if (v < n) s.key[v] = s.side[j] == 2 ? kMReport : s.va[j];
    }
    for (int i = tid; i < m; i += NT) {
        const int v = s.bvar[i];
        if (v < n) s.key[v] = fma(s.bb[i], kMReport, s.ba[i]);
    }
    for (int j = n + tid; j < NP; j += NT) s.key[j] = 0.0;
    __syncthreads();
    if (g.x)
        for (int j = tid; j < n; j += NT) g.x[(size_t)node * n + j] = s.key[j];
    if (g.y) {
        for (int i = tid; i < m; i += NT) g.y[(size_t)node * m + i] = 0.0;
        __syncthreads();
        for (int j = tid; j < n; j += NT)
            if (s.nvar[j] >= n) g.y[(size_t)node * m + (s.nvar[j] - n)] = s.d[j];
    }
    if (g.vstat_out) {
        int8_t *vo = g.vstat_out + (size_t)node * nv;
        for (int i = tid; i < m; i += NT) vo[s.bvar[i]] = 1;
        for (int j = tid; j < n; j += NT) vo[s.nvar[j]] = s.side[j] ? 2 : 3;
    }
    if (g.dbg_T && (node == 0 || g.dbg_all)) {
        const size_t k = g.dbg_all ? (size_t)node : 0;
        double *dT = g.dbg_T + k * (size_t)m * n;
        double *dvec = g.dbg_vec + k * (size_t)(n + 3 * m);
        int32_t *didx = g.dbg_idx + k * (size_t)(2 * n + m);
#pragma unroll
        for (int ii = 0; ii < R; ii++) {
#pragma unroll
            for (int jj = 0; jj < C; jj++) {
                const int i = bi + TBI * ii, j = bj + TBJ * jj;
                if (i < m && j < n) dT[(size_t)i * n + j] = T[ii][jj];
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
    if (tid < 64) {
        // obj = fold-in-half sum of c_j x_j over the padded power-of-two length
        constexpr int PER = NP / 64;
        double p[PER];
#pragma unroll
        for (int k = 0; k < PER; k++) {
            const int j = lane + 64 * k;
            p[k] = j < n ? gc[j] * s.key[j] : 0.0;
        }
#pragma unroll
        for (int h = PER / 2; h >= 1; h >>= 1) {
#pragma unroll
            for (int k = 0; k < h; k++) p[k] = p[k] + p[k + h];
        }
        double sum = p[0];
#pragma unroll
        for (int h = 32; h >= 1; h >>= 1) sum = sum + __shfl_down(sum, h, 64);
        if (tid == 0) {
            if (g.obj) g.obj[node] = status == 1 ? INF : sum;
            if (g.status) g.status[node] = status;
            if (g.iters) g.iters[node] = iters;
            if (g.npivots) g.npivots[node] = npiv;
        }
    }
    __syncthreads();
}

#undef MIPX_EXTRACT_COL
#undef MIPX_EXTRACT_ROW
#undef MIPX_UPDATE_T
#undef MIPX_UPDATE_COLS
#undef MIPX_PICK
#undef MIPX_LEAVE_SELECT

}  // namespace mipx
