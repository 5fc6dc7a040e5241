// K4 / K5 — batched dot-product scorer fused with the in-batch sampled-softmax loss and its
// gradient (SURVEY.md §2.2 K4/K5, §8a a3+a4; tfrs.tasks.Retrieval semantics, Appendix A).
// Probabilities / coefficients never reach HBM.  The exact-f32 TRAINING entry keeps the raw dot products [nq][nc] in the
// workspace between its two passes (MODE_FUSED_S writes, MODE_BWD_S reads: pass 2 skips GEMM1); every other entry keeps nothing.
//
// One kernel serves the forward statistics pass and both gradient passes.  It sees a STATIONARY
// matrix R [n_r, D] (one 32-row fragment per wave, held in registers for the whole launch) and a
// STREAMED matrix K [n_c, D] (32-row tiles, double-buffered in LDS, shared by the 4 waves):
//
//   GEMM1   X[c][r]  = sum_d K[c][d] * R[r][d]          v_mfma_f32_32x32x2_f32, D/2 per tile
//           (c on accumulator registers, r on lanes: every softmax reduction over c is lane-local)
//   FWD     online (max, sum-exp2) over c per lane; positive logit captured in-tile
//   BWD     coef[c][r] = exp2(X*c1 + a_c + a_r) * s_c*s_r  - [c == r+diag] s_c*s_r
//   GEMM2   G^T[d][r] += sum_c K[c][d] * coef[c][r]      the accumulator IS the B operand (no LDS
//           transpose): lane half h of register `reg` holds row acc_row(reg,h) = k of that step.
//
//   pass        R        K        a_r          s_r        a_c          s_c        diag
//   FWD  (lse)  q        c        -            -          -log2 p_c    -          +off
//   BWD dq      q        c        -lse2        w/T*g      -log2 p_c    1          +off
//   BWD dc      c        q        -log2 p_r    1          -lse2        w/T*g      -off
//
// f32-input MFMA: exact f32 products (guide §3 'FP32-input MFMA'); peak 157.3 TF on MI355X.
// Per (32 r x 32 c) tile and wave: D/2 + D/2 MFMAs (BWD), 16 v_exp_f32 per lane, 16+16(+8) LDS reads.
// The c range is split over the grid (blockIdx % nsplit; blocks b and b+8 share an XCD, so one
// XCD's L2 keeps re-serving the same K range); partial results go to per-split slabs that a small
// kernel sums in split order (bitwise reproducible; no float atomics).
#include "common.h"
#include <cstdlib>

namespace {

using tt::f32x4;
using tt::f32x16;
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

constexpr float kLog2e = 1.44269504088896340736f;
constexpr float kLn2 = 0.69314718055994530942f;
constexpr float kNegBig = -1.0e30f;   // log2-domain stand-in for -inf (tfrs uses finfo.min/100)

// MODE_FUSED_S: MODE_FUSED that also writes the raw dot products X[q][c] to ScoreArgs::S; MODE_BWD_S: MODE_BWD that reads
// them back instead of recomputing GEMM1 (the training entry's pass 2: half the matrix-pipe work)
enum { MODE_FWD = 0, MODE_BWD = 1, MODE_FUSED = 2, MODE_RANK = 3, MODE_FUSED_S = 4, MODE_BWD_S = 5 };
#ifndef TT_BX3_ABL
#define TT_BX3_ABL 0          // timing-only ablation hooks of the bf16x3 kernel (wrong results when non-zero; scratch/abl_bx3.sh)
#endif
#ifndef TT_BX3_STAGGER
// bf16x3, 8 waves: 1 = barrier between GEMM1 and the epilogue + 3-buffer LDS ring (see the tile loop), so that the two waves
// of a SIMD may drift apart.  r02 measurements at B = 8192, D = 128 (FUSED / BWD pass, us): lock-step (0) 147 / 151;
// this form 163 / 160; a first form (waves 4-7 run GEMM2 one tile late, coefficients kept across the barrier) 164 / 175 with
// 24 spilled VGPRs.  The ablation says why lock-step still wins: GEMM1 (69 us) and GEMM2 (39 us) already run at the
// bf16 matrix pipe's pace at the clock it holds (~1.6 GHz), and what is left (~60 us) is the epilogue + staging VALU,
// which neither form managed to slide under the partner wave's MFMAs at a 256-VGPR budget.
#define TT_BX3_STAGGER 0
#endif
#ifndef TT_PK_EPILOGUE
// 1: v_pk_fma_f32 / v_pk_add_f32 pairs in the softmax epilogue; 0: scalar f32 ops.  A/B on one box, alternating runs
// (profiles/r02_ab_epilogue_pk_vs_scalar.txt): FUSED 273.1 vs 272.4 us, BWD 271.3 vs 271.7 us, step 0.7024 vs 0.7019 ms —
// a tie; the scalar form is the default (MI355X_MICROARCH.md: packed f32 VALU beside MFMAs is at best neutral).
#define TT_PK_EPILOGUE 0
#endif
constexpr float kRescaleThr = 8.0f;   // FUSED: rescale the accumulators only when a row max grows by > 2^8 (p stays <= 256)

struct ScoreArgs {
  const float* R;
  const float* K;
  int64_t n_r, n_c;
  int64_t diag;             // positive pair: c == r + diag
  float c1;                 // log2(e) / temperature
  const float* a_r;         // [n_r] additive, log2 domain (nullable = 0)
  const float* s_r;         // [n_r] multiplicative (nullable = 1)
  const float* a_c;         // [n_c]
  const float* s_c;         // [n_c]
  const int64_t* id_r;      // accidental-hit ids (HAS_IDS only)
  const int64_t* id_c;
  int nsplit;
  int64_t c_per_split;      // multiple of 32
  float* part_m;            // FWD [nsplit][n_r]
  float* part_l;            // FWD [nsplit][n_r]
  float* pos2;              // FWD [n_r] positive logit (log2 domain)
  float* slab;              // BWD [nsplit][n_r][D]
  const float* h_r;         // optional [n_r]: hard-negative threshold per row   (element kept iff t >= h or positive)
  const float* h_c;         // optional [n_c]: hard-negative threshold per column (same domain as the masked value t)
  const int64_t* pos_idx;   // optional [n_r]: explicit positive column per row (else r + diag)
  int32_t* part_cnt;        // RANK [nsplit][n_r]: columns scoring strictly above the row's threshold a_r
  float* S;                 // FUSED_S (out) / BWD_S (in): raw dot products in 32 x 32 blocks (see store_S)
  int64_t ldS;              // number of query tiles = ceil(nq / 32)
};

// PREC = 0: exact f32 products (v_mfma_f32_32x32x2_f32).  PREC = 1: "bf16x3" — every f32 operand is split into three
// bf16 pieces (hi + mid + lo, the two residuals exact in f32) and the products run on v_mfma_f32_32x32x16_bf16 with f32
// accumulation: GEMM1 (the logits) keeps the 6 products down to 2^-24 relative (hi*hi, hi*mid, mid*hi, hi*lo, lo*hi,
// mid*mid), GEMM2 (the gradient sums, judged at 1e-4 of their maximum) the 3 down to 2^-16 (hi*hi, hi*mid, mid*hi).
// 72 bf16 MFMAs of 32 cycles per 32x32 tile and wave instead of 128 f32 MFMAs of 64: 0.28x the matrix-pipe time.
template <int D, int PREC>
struct Geo {
  static constexpr int LS = D + 4;             // f32: LDS row stride (floats): bank-conflict-free b128 reads
  static constexpr int NG = D / 8;             // f32 GEMM1 k-groups of 8 (4 per lane half)
  static constexpr int NB = D / 32;            // GEMM2 d-blocks (f32: d = NB*lane_row + b; bf16x3: d = 32*b + acc row)
  static constexpr int KS = D / 16;            // bf16x3 GEMM1 k-steps
  static constexpr int HALF_B = 32 * 256;      // bf16x3: bytes of one [32 rows][128 cols] bf16 sub-image
  static constexpr int PIECE_B = (D / 128) * HALF_B;
  static constexpr int TILE_F = PREC == 0 ? 32 * LS : 3 * PIECE_B / 4;
  static constexpr int BUF_F = TILE_F + 160;   // + a_c[32] + s_c[32] + id_c[32] (int64) + h_c[32]
  static constexpr int LDS_BYTES = 2 * BUF_F * 4;      // 4-wave schedule: 2 buffers
};

// x = hi + mid + lo, each a bf16 (round to nearest even); x - hi and (x - hi) - mid are exact in f32
struct Bf3 {
  __bf16 hi, mid, lo;
};
__device__ __forceinline__ Bf3 split3(float x) {
  Bf3 o;
  o.hi = (__bf16)x;
  const float r1 = x - (float)o.hi;
  o.mid = (__bf16)r1;
  o.lo = (__bf16)(r1 - (float)o.mid);
  return o;
}

// Byte offset of 16-byte chunk `ch` (8 bf16) of row `row` in a [32][128] bf16 sub-image with plain 256-byte rows and
// the chunk index XOR-swizzled by the row (cdna_hip_programming.md T10, image (b)): conflict-free both for the row
// reads (ds_read_b128, GEMM1's A operand) and for the transposing reads (ds_read_b64_tr_b16, GEMM2's A operand = K^T).
__device__ __forceinline__ int img_off(int row, int ch) { return 256 * row + 16 * (ch ^ (((row & 3) << 2) | ((row >> 2) & 3))); }

#ifndef TT_BWDS_WAVES
#define TT_BWDS_WAVES 2       // min waves per SIMD of the BWD_S pass at D <= 128 (3 = 170 VGPRs)
#endif
#ifndef TT_S_PREFETCH
#define TT_S_PREFETCH 2       // BWD_S: tiles of stored dot products in flight ahead of the one being worked on (1 or 2)
#endif
#ifndef TT_ABL_BWDS_NOSTAGE
#define TT_ABL_BWDS_NOSTAGE 0
#endif
#ifndef TT_LOOP_LAMBDA
#define TT_LOOP_LAMBDA 0
#endif
#ifndef TT_TILES_PER_BARRIER
#define TT_TILES_PER_BARRIER 2
#endif
#ifndef TT_TILES_PER_BARRIER_BWDS
#define TT_TILES_PER_BARRIER_BWDS TT_TILES_PER_BARRIER      // the dc pass (one 8-wave workgroup per CU: room for a ring of 8 buffers)
#endif
template <int D, int MODE, int PREC>
constexpr int tiles_per_barrier() {
  if (PREC == 0 && D <= 128 && MODE == MODE_BWD_S) return TT_TILES_PER_BARRIER_BWDS;
  return (PREC == 0 && D <= 128 && (MODE == MODE_BWD || MODE == MODE_FUSED || MODE == MODE_FUSED_S))
             ? TT_TILES_PER_BARRIER : 1;
}

// bf16x3 at dim 256, passes with GEMM1 AND GEMM2: a PAIR of waves shares 32 stationary rows and splits the embedding
// dimension - each wave holds the three bf16 pieces of R[r][128 hw .. 128 hw + 127] (96 VGPRs instead of 192), multiplies
// the tile's matching k-half, the two partial dot-product blocks meet through LDS (4 KB per wave, one extra barrier), both
// waves run the (identical) softmax on the sum, and each accumulates its own 128 columns of the gradient block (64 AGPRs
// instead of 128).  r02's one-wave-per-32-rows form needed ~640 registers of the 512 and spilled 416-556 B per lane.
template <int D, int MODE, int PREC>
constexpr bool split_d() {
  return PREC == 1 && D == 256 && (MODE == MODE_BWD || MODE == MODE_FUSED || MODE == MODE_FUSED_S);
}
// exact f32 at dim 256, online-softmax passes WITH accidental-hit ids / hard negatives: the stationary fragment (128 VGPRs) +
// the gradient block (128) + the tile in flight (32) + the id / threshold state no longer fit 512 registers (r03: 40 / 56 /
// 132 B of scratch per lane).  The LAST k-groups of the fragment - 5 with ids, 8 with thresholds, 10 with both - live in LDS, one
// 16-byte slot per lane, written once and read back by the same lane (one ds_read_b128 per group and tile): 20 / 40 VGPRs less.
template <int D, int MODE, bool HAS_IDS, bool HAS_HN, int PREC>
constexpr int rf_lds_groups() {
  if (PREC == 0 && D == 256 && (MODE == MODE_FUSED || MODE == MODE_FUSED_S) && (HAS_IDS || HAS_HN))
    return (HAS_IDS && HAS_HN) ? 10 : (HAS_HN ? 8 : 5);
  if (PREC == 0 && D == 128 && MODE == MODE_BWD && HAS_IDS && HAS_HN) return 2;      // (20 B of scratch at 2 waves per SIMD)
  return 0;
}
template <int D, int MODE, int PREC, int WAVES>
constexpr int rows_per_wg() { return (split_d<D, MODE, PREC>() ? WAVES / 2 : WAVES) * 32; }

template <int D, int MODE, bool HAS_IDS, bool HAS_HN, int WAVES, int PREC>
__global__ __launch_bounds__(WAVES * 64, (WAVES == 8 ? 2 : (D <= 128 ? ((MODE == MODE_BWD_S && PREC == 0) ? TT_BWDS_WAVES : 2) : 1))) void score_kernel(ScoreArgs p) {   // (min waves per SIMD)
  constexpr bool IS_FUSED = MODE == MODE_FUSED || MODE == MODE_FUSED_S;     // online softmax + dq
  constexpr bool IS_BWD = MODE == MODE_BWD || MODE == MODE_BWD_S;            // gradient pass with given row statistics
  constexpr bool FROM_S = MODE == MODE_BWD_S, TO_S = MODE == MODE_FUSED_S;
  using G_ = Geo<D, PREC>;
  constexpr int LS = G_::LS, NG = G_::NG, NB = G_::NB, TILE_F = G_::TILE_F, BUF_F = G_::BUF_F;
  constexpr int HALF_B = G_::HALF_B, PIECE_B = G_::PIECE_B;
  constexpr bool SPLIT = split_d<D, MODE, PREC>();
  constexpr int KS = SPLIT ? G_::KS / 2 : G_::KS;     // GEMM1 k-steps of THIS wave
  constexpr int NBW = SPLIT ? NB / 2 : NB;            // GEMM2 d-blocks of THIS wave
  static_assert(PREC == 0 || D % 128 == 0, "bf16x3: dim 128 / 256 only (whole [32][128] bf16 images)");
  constexpr int ROW4 = D / 4;                       // float4 per K row
  constexpr int THREADS = WAVES * 64;
  constexpr int NV = (32 * ROW4 + THREADS - 1) / THREADS;   // staged float4 per thread
  extern __shared__ __attribute__((aligned(16))) float smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int h = lane >> 5;
  const int ln = lane & 31;

  // Everything that steers a loop with a workgroup barrier in it is made PROVABLY wave-uniform (readfirstlane): the split,
  // the row block and, below, the tile count.  They depend on blockIdx and kernel arguments only, but hipcc divides on the
  // vector ALU; whether the quotient comes back to an SGPR is its choice, and a trip count left in VGPRs turns the tile
  // loop's exit test into a vector compare with the barrier inside a loop the compiler treats as divergent.
  // tests/isa_audit/audit_barriers.py checks the built ISA: in all instantiations every barrier loop closes on scalar branches.
  const int split = __builtin_amdgcn_readfirstlane((int)(blockIdx.x % (unsigned)p.nsplit));
  const int64_t rblk = __builtin_amdgcn_readfirstlane((int)(blockIdx.x / (unsigned)p.nsplit));
  const int hw = SPLIT ? (wave & 1) : 0;             // SPLIT: the half of the embedding dimension this wave owns
  const int64_t r0w = rblk * rows_per_wg<D, MODE, PREC, WAVES>() + (SPLIT ? wave >> 1 : wave) * 32;   // first row of this wave
  const int64_t r = r0w + ln;                       // this lane's row (both halves)
  const bool r_ok = r < p.n_r;
  const int64_t c_begin = (int64_t)split * p.c_per_split;
  int64_t c_end = c_begin + p.c_per_split;
  if (c_end > p.n_c) c_end = p.n_c;
  const int ntiles = __builtin_amdgcn_readfirstlane(c_end > c_begin ? (int)((c_end - c_begin + 31) >> 5) : 0);

  // ---- stationary fragment (B operand of GEMM1), in registers for the whole launch ----
  //   f32:     rf[g][s]     = R[r][8g + 4h + s]
  //   bf16x3:  rp[q][ks][j] = piece q of R[r][16 ks + 8h + j]
  // (bf16x3 with 8-wave workgroups: the lo piece — one product per k-step — lives in LDS, [ks][lane] x 16 B per wave,
  // written once: 32 VGPRs less, which is what keeps the gradient kernels free of scratch spills)
  constexpr bool RLO_LDS = PREC == 1 && WAVES == 8;
  constexpr bool STAG = PREC == 1 && WAVES == 8 && TT_BX3_STAGGER && !FROM_S && !TO_S;     // staggered wave halves + 3-buffer LDS ring (see the tile loop)
  // exact-f32 gradient passes: TPB tiles between workgroup barriers (ring of 2*TPB LDS buffers, the prefetch runs TPB tiles
  // ahead); everything else: one tile per barrier, two buffers
  constexpr int TPB = tiles_per_barrier<D, MODE, PREC>();
  // Two LDS assumptions of the wave-pair (SPLIT) path, pinned: its exchange buffer `xch` sits where the 8-wave kernels keep the
  // lo fragments (`rlo`) - never both in one kernel; and exchange() has ONE barrier between a pair's write and its partner's
  // read, so the next tile's write is only safe behind the per-tile barrier that TPB == 1 gives.
  static_assert(!(SPLIT && RLO_LDS), "wave-pair exchange buffer and the LDS lo fragments share one LDS region");
  static_assert(!SPLIT || TPB == 1, "the wave-pair exchange relies on one workgroup barrier per tile");
  constexpr int NBUF = STAG ? 3 : 2 * TPB;
  constexpr int RFL = rf_lds_groups<D, MODE, HAS_IDS, HAS_HN, PREC>();      // k-groups of the stationary fragment kept in LDS
  constexpr int RFR = PREC == 0 ? NG - RFL : 1;                             // ... and in registers
  f32x4 rf[RFR];
  f32x4* rfl = reinterpret_cast<f32x4*>(smem + NBUF * BUF_F) + (wave * (RFL > 0 ? RFL : 1)) * 64 + lane;   // [group][lane], this wave's
  bf16x8 rp[PREC == 0 ? 1 : (RLO_LDS ? 2 : 3)][PREC == 0 ? 1 : KS];
  bf16x8* rlo = reinterpret_cast<bf16x8*>(smem + NBUF * BUF_F) + (wave * KS) * 64 + lane;
  if constexpr (FROM_S) {
    // pass 2 of the training entry: the dot products come back from HBM, no stationary fragment, no GEMM1
  } else if constexpr (PREC == 0) {
    const f32x4* R4 = reinterpret_cast<const f32x4*>(p.R + (r_ok ? r : 0) * D) + h;
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      const f32x4 v = r_ok ? R4[2 * g] : f32x4{0.f, 0.f, 0.f, 0.f};
      if constexpr (RFL > 0) {
        if (g >= RFR) rfl[(g - RFR) * 64] = v; else rf[g < RFR ? g : 0] = v;   // (read back by this lane only: no barrier needed)
      } else {
        rf[g] = v;
      }
    }
  } else {
    const f32x4* R4 = reinterpret_cast<const f32x4*>(p.R + (r_ok ? r : 0) * D) + 2 * h + 32 * hw;
    const float live = r_ok ? 1.f : 0.f;            // rows past n_r read row 0 and are zeroed (all loads unconditional)
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const f32x4 x0 = R4[4 * ks] * live;
      const f32x4 x1 = R4[4 * ks + 1] * live;
      bf16x8 lo8;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const Bf3 u = split3(x0[j]), v = split3(x1[j]);
        rp[0][ks][j] = u.hi; rp[1][ks][j] = u.mid; lo8[j] = u.lo;
        rp[0][ks][4 + j] = v.hi; rp[1][ks][4 + j] = v.mid; lo8[4 + j] = v.lo;
      }
      if constexpr (RLO_LDS) rlo[ks * 64] = lo8;        // read back by this lane only: no barrier needed
      else rp[2][ks] = lo8;
    }
  }
  const float ar = ((IS_BWD || MODE == MODE_RANK) && p.a_r != nullptr && r_ok) ? p.a_r[r] : 0.f;
  const float sr = (IS_BWD && p.s_r != nullptr && r_ok) ? p.s_r[r] : 1.f;
  int64_t idr = 0;
  if constexpr (HAS_IDS) idr = r_ok ? p.id_r[r] : (int64_t)-1;
  const int64_t cpos = p.pos_idx != nullptr ? (r_ok ? p.pos_idx[r] : (int64_t)-1) : r + p.diag;   // positive column

  // ---- staging (global -> regs -> LDS, one tile ahead).  Thread `tid` moves float4 number
  // tid + 256*j of the tile (row = f / ROW4, col4 = f % ROW4): 32-bit offsets from one running pointer.
  f32x4 st[NV];
  float st_a = 0.f, st_s = 0.f, st_h = -3.0e38f;
  constexpr bool hn = HAS_HN;                                        // hard-negative mining compiled in
  const float hr = (HAS_HN && p.h_r != nullptr && r_ok) ? p.h_r[r] : -3.0e38f;
  const float* hcp = (HAS_HN && p.h_c != nullptr) ? p.h_c + c_begin + tid : nullptr;
  int64_t st_id = -2;
  constexpr int RPJ = THREADS / ROW4 > 0 ? THREADS / ROW4 : 1;   // tile rows between a thread's consecutive float4s
  const int st_row = tid / ROW4, st_col4 = tid % ROW4;
  const f32x4* kp = reinterpret_cast<const f32x4*>(p.K) + c_begin * ROW4 + tid;   // tile 0
  const float* acp = p.a_c != nullptr ? p.a_c + c_begin + tid : nullptr;
  const float* scp = p.s_c != nullptr ? p.s_c + c_begin + tid : nullptr;
  const int64_t* idp = nullptr;
  if constexpr (HAS_IDS) idp = p.id_c + c_begin + tid;
  const int ncols = (int)(c_end - c_begin);         // columns of this split (<= c_per_split)

  auto load_tile = [&](int t) {
    const int nvalid = ncols - 32 * t;              // valid rows of tile t (may exceed 32)
    const f32x4* src = kp + (int64_t)t * (32 * ROW4);
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      const int row = st_row + j * RPJ;
      st[j] = (row < 32 && row < nvalid) ? src[THREADS * j] : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    if (tid < 32) {
      const bool ok = tid < nvalid;
      st_a = ok ? (acp != nullptr ? acp[32 * t] : 0.f) : kNegBig;
      if constexpr (HAS_HN) st_h = (ok && hcp != nullptr) ? hcp[32 * t] : -3.0e38f;
      st_s = ok ? (scp != nullptr ? scp[32 * t] : 1.f) : 0.f;
      if constexpr (HAS_IDS) st_id = ok ? idp[32 * t] : (int64_t)-2;
    }
  };
  auto store_tile = [&](int buf) {
    float* T = smem + buf * BUF_F;
    if constexpr (PREC == 0) {
      float* dst = T + st_row * LS + st_col4 * 4;
#pragma unroll
      for (int j = 0; j < NV; ++j)
        if (st_row + j * RPJ < 32) *reinterpret_cast<f32x4*>(dst + j * RPJ * LS) = st[j];
    } else {
      // three bf16 images of the tile (hi / mid / lo pieces), each float4 -> 4 bf16 = one ds_write_b64 per piece
      char* img = reinterpret_cast<char*>(T);
      const int d0 = 4 * st_col4;
      const int sub = (d0 >> 7) * HALF_B + 8 * ((d0 >> 2) & 1), ch = (d0 & 127) >> 3;
#pragma unroll
      for (int j = 0; j < NV; ++j) {
        const int row = st_row + j * RPJ;
        if (row < 32) {
          bf16x4 q0, q1, q2;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            if constexpr (TT_BX3_ABL & 1) { q0[e] = (__bf16)st[j][e]; q1[e] = q0[e]; q2[e] = q0[e]; continue; }
            const Bf3 u = split3(st[j][e]);
            q0[e] = u.hi; q1[e] = u.mid; q2[e] = u.lo;
          }
          char* dst = img + sub + img_off(row, ch);
          *reinterpret_cast<bf16x4*>(dst) = q0;
          *reinterpret_cast<bf16x4*>(dst + PIECE_B) = q1;
          *reinterpret_cast<bf16x4*>(dst + 2 * PIECE_B) = q2;
        }
      }
    }
    if (tid < 32) {
      T[TILE_F + tid] = st_a;
      T[TILE_F + 32 + tid] = st_s;
      if constexpr (HAS_HN) T[TILE_F + 128 + tid] = st_h;
      if constexpr (HAS_IDS) reinterpret_cast<int64_t*>(T + TILE_F + 64)[tid] = st_id;
    }
  };

  // ---- the raw dot products through HBM (training entry: pass 1 writes them, pass 2 reads them back) ----
  // Stored as 32 x 32 BLOCKS of 4 KB, block (candidate tile, query tile) at [ctile * ldS + qtile] (ldS = query tiles), inside a
  // block [query][candidate]: the block one wave produces in pass 1 is the block one wave consumes in pass 2, and both
  // move it as one contiguous 4 KB burst (row-major [nq][nc] made every 128-byte line of a tile a separate DRAM page).
  // FUSED_S (R = q, K = c): lane = query row, accumulator register reg = candidate acc_row(reg, h): registers 4g .. 4g+3
  //   are 4 consecutive candidates -> one 16-byte store per g.
  // BWD_S (R = c, K = q): lane = candidate, register = query row: for a fixed register the 32 lanes of a half read 32
  //   consecutive candidates of one query row -> coalesced 4-byte loads.
  // (r02 alternatives, cfg3, pass 1 / pass 2 us against 287 / 174 for row-major: transposed row-major 290 / 170;
  // nontemporal stores 387 / 175 - the pieces of a line are no longer merged in L2; nontemporal loads 287 / 175.)
  auto store_S = [&](int t, const f32x16& X) {
    if (!r_ok) return;
    const int64_t ctile = (c_begin >> 5) + t;
    float* blk = p.S + (ctile * p.ldS + (r0w >> 5)) * 1024 + ln * 32 + 4 * h;
#pragma unroll
    for (int g = 0; g < 4; ++g)
      if (!SPLIT || (g >> 1) == hw)       // SPLIT: both waves of the pair hold the same block, each stores half of it
        *reinterpret_cast<f32x4*>(blk + 8 * g) = f32x4{X[4 * g], X[4 * g + 1], X[4 * g + 2], X[4 * g + 3]};
  };
  // SPLIT: the partial dot products of the two k-halves meet in LDS.  a + b == b + a bit for bit, so both waves of a pair go
  // on with the same block (their softmax statistics are duplicates, their gradient columns disjoint).
  float* xch = smem + NBUF * BUF_F;
  auto exchange = [&](f32x16& X) {
    f32x4* mine = reinterpret_cast<f32x4*>(xch + wave * 1024) + lane;
    const f32x4* other = reinterpret_cast<const f32x4*>(xch + (wave ^ 1) * 1024) + lane;
#pragma unroll
    for (int g = 0; g < 4; ++g) mine[64 * g] = f32x4{X[4 * g], X[4 * g + 1], X[4 * g + 2], X[4 * g + 3]};
    __syncthreads();
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const f32x4 o = other[64 * g];
#pragma unroll
      for (int i = 0; i < 4; ++i) X[4 * g + i] += o[i];
    }
  };
  auto load_S = [&](int t, f32x16& X) {
    const int64_t q0 = c_begin + 32 * (int64_t)t;
    const int nq_left = (int)(c_end - q0);            // valid query rows of this tile (may exceed 32)
    const float* blk = p.S + ((r0w >> 5) * p.ldS + (q0 >> 5)) * 1024 + 4 * h * 32 + ln;
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
      const bool ok = r_ok && (tt::acc_row(reg, 0) + 4 * h) < nq_left;
      X[reg] = ok ? blk[tt::acc_row(reg, 0) * 32] : 0.f;
    }
  };

  // ---- per-lane state ----
  float run_m = kNegBig, run_l = 0.f, pos = 0.f;
  int cnt = 0;
  bool have_pos = false;
  f32x16 G[NBW];
  if constexpr (IS_BWD || IS_FUSED) {
#pragma unroll
    for (int b = 0; b < NBW; ++b)
#pragma unroll
      for (int i = 0; i < 16; ++i) G[b][i] = 0.f;
  }

  // (r02: starting the second workgroup of every CU 1-4 us late, so that the two waves of a SIMD run half a tile apart, only
  // added the delay: 174.9 -> 175.5 .. 177.7 us for the dc pass.  Ablation of that pass: GEMM2 alone 112 us = the matrix pipe's
  // time; everything else - tile staging 25, dot-product loads 6-19, epilogue 9, loop skeleton 23 - adds to it instead of
  // hiding under it, with or without a phase shift between the waves.)
#pragma unroll
  for (int t0 = 0; t0 < TPB; ++t0)
    if (t0 < ntiles) {
      load_tile(t0);
      store_tile(t0);
    }
  __syncthreads();
  // ---- GEMM1: X[c][r] = sum_d K[c][d] R[r][d] ----
  auto gemm1 = [&](const float* T) -> f32x16 {
    f32x16 X;
#pragma unroll
    for (int i = 0; i < 16; ++i) X[i] = 0.f;
    if constexpr (PREC == 1) {
      // A operand: lane (row c = ln, half h) reads k = 16 ks + 8h .. +7 of every piece: one ds_read_b128 per piece.
      // Software-pipelined by hand: the three fragments of k-step ks+1 are in flight under the six MFMAs of k-step ks
      // (left to the compiler the MFMAs sat behind each read's LDS latency: 2.3x the pure MFMA time, r02 ablation).
      const char* img = reinterpret_cast<const char*>(T);
      constexpr int KSN = (TT_BX3_ABL & 2) ? 1 : KS;
      auto frag = [&](int ks, bf16x8& f_hi, bf16x8& f_mid, bf16x8& f_lo, bf16x8& r_lo) {
        const int off = (SPLIT ? hw : (ks >> 3)) * HALF_B + img_off(ln, 2 * (ks & 7) + h);
        f_hi = *reinterpret_cast<const bf16x8*>(img + off);
        f_mid = *reinterpret_cast<const bf16x8*>(img + PIECE_B + off);
        f_lo = *reinterpret_cast<const bf16x8*>(img + 2 * PIECE_B + off);
        if constexpr (RLO_LDS) r_lo = rlo[ks * 64];
      };
      bf16x8 a_hi, a_mid, a_lo, b_lo = {};
      frag(0, a_hi, a_mid, a_lo, b_lo);
#pragma unroll
      for (int ks = 0; ks < KSN; ++ks) {
        bf16x8 n_hi = a_hi, n_mid = a_mid, n_lo = a_lo, nb_lo = b_lo;
        if (ks + 1 < KSN) frag(ks + 1, n_hi, n_mid, n_lo, nb_lo);
        if constexpr (!RLO_LDS) b_lo = rp[2][ks];
        X = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_lo, rp[0][ks], X, 0, 0, 0);     // smallest terms first
        X = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_hi, b_lo, X, 0, 0, 0);
        X = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_mid, rp[1][ks], X, 0, 0, 0);
        X = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_mid, rp[0][ks], X, 0, 0, 0);
        X = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_hi, rp[1][ks], X, 0, 0, 0);
        X = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_hi, rp[0][ks], X, 0, 0, 0);
        a_hi = n_hi; a_mid = n_mid; a_lo = n_lo; b_lo = nb_lo;
        __builtin_amdgcn_sched_barrier(0);
      }
    } else {
      const float* arow = T + ln * LS + 4 * h;
      // one ds_read_b128 ahead of the MFMAs that use it; sched_barrier keeps hipcc from hoisting all
      // NG reads (4*NG VGPRs) to the loop top
      f32x4 a_cur = *reinterpret_cast<const f32x4*>(arow);
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        f32x4 a_nxt = a_cur;
        if (g + 1 < NG) a_nxt = *reinterpret_cast<const f32x4*>(arow + 8 * (g + 1));
        f32x4 rg;
        if constexpr (RFL > 0) { if (g >= RFR) rg = rfl[(g - RFR) * 64]; else rg = rf[g < RFR ? g : 0]; } else rg = rf[g];
        X = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[0], rg[0], X, 0, 0, 0);
        X = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[1], rg[1], X, 0, 0, 0);
        X = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[2], rg[2], X, 0, 0, 0);
        X = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[3], rg[3], X, 0, 0, 0);
        a_cur = a_nxt;
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    return X;
  };

  // ---- per-tile softmax math on the accumulator: statistics (FWD/RANK/FUSED) and the GEMM2 coefficients (BWD/FUSED) ----
  auto epilogue = [&](const float* T, int t, const f32x16& X, float (&coef)[16]) {
    const int64_t c0 = c_begin + 32 * (int64_t)t;
    // ---- per-column terms of this lane's 16 accumulator rows: c = c0 + 8*q + 4*h + i ----
    float ac[16], sc[16];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 va = *reinterpret_cast<const f32x4*>(T + TILE_F + 8 * q + 4 * h);
#pragma unroll
      for (int i = 0; i < 4; ++i) ac[4 * q + i] = va[i];
      if constexpr (IS_BWD) {
        const f32x4 vs = *reinterpret_cast<const f32x4*>(T + TILE_F + 32 + 8 * q + 4 * h);
#pragma unroll
        for (int i = 0; i < 4; ++i) sc[4 * q + i] = vs[i];
      }
    }
    float hth[16];                                   // per-element hard-negative threshold: max(row's, column's)
    if constexpr (hn) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 vh = *reinterpret_cast<const f32x4*>(T + TILE_F + 128 + 8 * q + 4 * h);
#pragma unroll
        for (int i = 0; i < 4; ++i) hth[4 * q + i] = fmaxf(vh[i], hr);
      }
    }
    // wave-uniform: does this tile hold the positive of any row of this wave?
    bool diag_tile;
    if (p.pos_idx != nullptr) {
      const int64_t dd0 = cpos - c0;
      diag_tile = __any(dd0 >= 0 && dd0 < 32) != 0;
    } else {
      diag_tile = (c0 < r0w + p.diag + 32) && (c0 + 32 > r0w + p.diag);
    }
    // tile-local row (minus the lane half's +4h) of this lane's positive, or a value no row matches
    int dloc = -100;
    if (diag_tile) {
      const int64_t dd = cpos - c0;
      dloc = (dd >= 0 && dd < 32) ? (int)dd - 4 * h : -100;
    }
    bool dup[16];
    if constexpr (HAS_IDS) {
      const int64_t* idc = reinterpret_cast<const int64_t*>(T + TILE_F + 64) + 4 * h;
#pragma unroll
      for (int reg = 0; reg < 16; ++reg)
        dup[reg] = (idc[tt::acc_row(reg, 0)] == idr) && (tt::acc_row(reg, 0) != dloc);
    }

    if constexpr (MODE == MODE_RANK) {
      // rank of the positive = number of OTHER columns scoring strictly above the row's threshold (its own logit)
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const float v = __builtin_fmaf(X[reg], p.c1, ac[reg]);
        bool above = v > ar && tt::acc_row(reg, 0) != dloc;
        if constexpr (HAS_IDS) above = above && !dup[reg];          // an accidental hit is not a competitor (tfrs: its logit is -inf)
        cnt += above ? 1 : 0;
      }
    } else if constexpr (MODE == MODE_FWD) {
      float t2[16];
      float mx = kNegBig;
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        float v = __builtin_fmaf(X[reg], p.c1, ac[reg]);
        if constexpr (HAS_IDS) v = dup[reg] ? kNegBig : v;
        if constexpr (hn) v = (v < hth[reg] && tt::acc_row(reg, 0) != dloc) ? kNegBig : v;
        t2[reg] = v;
        mx = fmaxf(mx, v);
      }
      if (diag_tile) {
#pragma unroll
        for (int reg = 0; reg < 16; ++reg)
          if (tt::acc_row(reg, 0) == dloc) { pos = t2[reg]; have_pos = true; }
      }
      const float m_new = fmaxf(run_m, mx);
      float sum = 0.f;
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) sum += __builtin_amdgcn_exp2f(t2[reg] - m_new);
      run_l = run_l * __builtin_amdgcn_exp2f(run_m - m_new) + sum;
      run_m = m_new;
    } else {
      if constexpr (IS_BWD) {
        // the fma and the add as packed f32 pairs (v_pk_fma_f32 / v_pk_add_f32: same roundings, half the VALU
        // issue slots — VALU cycles are not hidden behind f32 MFMAs)
        float tvs[16];
        if constexpr (TT_PK_EPILOGUE) {
#pragma unroll
          for (int q = 0; q < 8; ++q) {
            const f32x2 x2 = f32x2{X[2 * q], X[2 * q + 1]}, a2 = f32x2{ac[2 * q], ac[2 * q + 1]};
            const f32x2 t2 = __builtin_elementwise_fma(x2, f32x2{p.c1, p.c1}, a2) + f32x2{ar, ar};
            tvs[2 * q] = t2[0];
            tvs[2 * q + 1] = t2[1];
          }
        } else {
#pragma unroll
          for (int reg = 0; reg < 16; ++reg) tvs[reg] = __builtin_fmaf(X[reg], p.c1, ac[reg]) + ar;
        }
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
          const float w = sc[reg] * sr;
          float tv = tvs[reg];
          if constexpr (hn) tv = (tv < hth[reg] && tt::acc_row(reg, 0) != dloc) ? kNegBig : tv;
          float e = __builtin_amdgcn_exp2f(tv) * w;
          if constexpr (HAS_IDS) e = dup[reg] ? 0.f : e;
          coef[reg] = e;
        }
        if (diag_tile) {
#pragma unroll
          for (int reg = 0; reg < 16; ++reg)
            if (tt::acc_row(reg, 0) == dloc) coef[reg] -= sc[reg] * sr;
        }
      } else {
        // FUSED (flash-style): online softmax over c; coef = exp2(t - m_row) un-normalised, the accumulators
        // carry sum_c p*K[c] at scale m_row.  Both lane halves of a row share ONE running max (GEMM2 sums
        // over both halves' c).  Lazy rescale (threshold kRescaleThr): wave-uniform branch.
        float mx = kNegBig;
        float vs[16];
        if constexpr (TT_PK_EPILOGUE) {
#pragma unroll
          for (int q = 0; q < 8; ++q) {      // packed f32 pairs: v_pk_fma_f32
            const f32x2 x2 = f32x2{X[2 * q], X[2 * q + 1]}, a2 = f32x2{ac[2 * q], ac[2 * q + 1]};
            const f32x2 t2 = __builtin_elementwise_fma(x2, f32x2{p.c1, p.c1}, a2);
            vs[2 * q] = t2[0];
            vs[2 * q + 1] = t2[1];
          }
        } else {
#pragma unroll
          for (int reg = 0; reg < 16; ++reg) vs[reg] = __builtin_fmaf(X[reg], p.c1, ac[reg]);
        }
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
          float v = vs[reg];
          if constexpr (HAS_IDS) v = dup[reg] ? kNegBig : v;
          if constexpr (hn) v = (v < hth[reg] && tt::acc_row(reg, 0) != dloc) ? kNegBig : v;
          coef[reg] = v;
          mx = fmaxf(mx, v);
        }
        if (diag_tile) {
#pragma unroll
          for (int reg = 0; reg < 16; ++reg)
            if (tt::acc_row(reg, 0) == dloc) { pos = coef[reg]; have_pos = true; }
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        if (__any(mx - run_m > kRescaleThr)) {
          const float m_new = fmaxf(run_m, mx);
          const float alpha = __builtin_amdgcn_exp2f(run_m - m_new);
          run_l *= alpha;
#pragma unroll
          for (int b = 0; b < NBW; ++b)
#pragma unroll
            for (int i = 0; i < 16; ++i) G[b][i] *= alpha;
          run_m = m_new;
        }
        float sum = 0.f;
        if constexpr (TT_PK_EPILOGUE) {
#pragma unroll
          for (int q = 0; q < 8; ++q) {      // the subtraction as packed pairs (v_pk_add_f32 with negated operand)
            const f32x2 d2 = f32x2{coef[2 * q], coef[2 * q + 1]} - f32x2{run_m, run_m};
            coef[2 * q] = __builtin_amdgcn_exp2f(d2[0]);
            coef[2 * q + 1] = __builtin_amdgcn_exp2f(d2[1]);
          }
          f32x2 s2 = f32x2{coef[0], coef[1]};           // 8 packed adds + 1 instead of 16 sequential ones
#pragma unroll
          for (int q = 1; q < 8; ++q) s2 += f32x2{coef[2 * q], coef[2 * q + 1]};
          sum = s2[0] + s2[1];
        } else {                                        // the same additions (pairs, then the two lanes), scalar ops
          float s0 = 0.f, s1 = 0.f;
#pragma unroll
          for (int q = 0; q < 8; ++q) {
            coef[2 * q] = __builtin_amdgcn_exp2f(coef[2 * q] - run_m);
            coef[2 * q + 1] = __builtin_amdgcn_exp2f(coef[2 * q + 1] - run_m);
            s0 = q == 0 ? coef[0] : s0 + coef[2 * q];
            s1 = q == 0 ? coef[1] : s1 + coef[2 * q + 1];
          }
          sum = s0 + s1;
        }
        run_l += sum;
      }
    }
  };

  auto gemm2 = [&](const float* T, const float (&coef)[16]) {
    if constexpr (PREC == 1) {
      // ---- GEMM2 (bf16x3): G^T[d = 32 b + acc row][r] += sum_c K[c][d] * coef[c][r].  The accumulator's registers
      // 8s .. 8s+7 ARE the B fragment of k-step s (element j of lane half h = tile row c = 16s + 8(j>>2) + 4h + (j&3));
      // the A fragment K^T[d][those c] comes out of the SAME row-major images through transposing reads
      // (ds_read_b64_tr_b16: a 16-lane group fetches 4 rows x 16 columns and gets them column-major). ----
      bf16x8 c_hi[2], c_mid[2];
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const __bf16 hi = (__bf16)coef[i];
        c_hi[i >> 3][i & 7] = hi;
        c_mid[i >> 3][i & 7] = (TT_BX3_ABL & 8) ? hi : (__bf16)(coef[i] - (float)hi);
      }
      const char* img = reinterpret_cast<const char*>(T);
      const int g16 = (lane >> 4) & 1, q = (lane >> 2) & 3, pp = lane & 3;
      // flat step i = 2 b + s; the four transposing reads of step i+1 are in flight under the three MFMAs of step i
      constexpr int NSTEP = 2 * ((TT_BX3_ABL & 4) ? 1 : NBW);
      auto frag = [&](int i, bf16x8& k_hi, bf16x8& k_mid) {
        const int b = i >> 1, s = i & 1;
        const int sub = (SPLIT ? hw : ((32 * b) >> 7)) * HALF_B + 8 * (pp & 1);     // (SPLIT: b counts inside the wave's own 128 columns)
        const int ch = (((32 * b) & 127) >> 3) + 2 * g16 + (pp >> 1);
#pragma unroll
        for (int half = 0; half < 2; ++half) {
          const int row = 16 * s + 8 * half + 4 * h + q;
          const int off = sub + img_off(row, ch);
          const s16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(img + off));
          const s16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(img + PIECE_B + off));
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            k_hi[4 * half + e] = __builtin_bit_cast(__bf16, (short)v0[e]);
            k_mid[4 * half + e] = __builtin_bit_cast(__bf16, (short)v1[e]);
          }
        }
      };
      bf16x8 k_hi, k_mid;
      frag(0, k_hi, k_mid);
#pragma unroll
      for (int i = 0; i < NSTEP; ++i) {
        bf16x8 n_hi = k_hi, n_mid = k_mid;
        if (i + 1 < NSTEP) frag(i + 1, n_hi, n_mid);
        const int b = i >> 1, s = i & 1;
        G[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k_mid, c_hi[s], G[b], 0, 0, 0);
        G[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k_hi, c_mid[s], G[b], 0, 0, 0);
        G[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k_hi, c_hi[s], G[b], 0, 0, 0);
        k_hi = n_hi; k_mid = n_mid;
        __builtin_amdgcn_sched_barrier(0);
      }
      return;
    } else {
      // ---- GEMM2: G^T[d = NB*i + b][r] += K[c(reg,h)][d] * coef[reg] ----
      const float* kbase = T + 4 * h * LS + NB * ln;
      f32x4 kc0, kc1;
      auto read_k = [&](int reg, f32x4& k0, f32x4& k1) {
        const float* krow = kbase + tt::acc_row(reg, 0) * LS;
        if constexpr (NB == 8) {
          k0 = *reinterpret_cast<const f32x4*>(krow);
          k1 = *reinterpret_cast<const f32x4*>(krow + 4);
        } else if constexpr (NB == 4) {
          k0 = *reinterpret_cast<const f32x4*>(krow);
        } else if constexpr (NB == 2) {
          const float2 k2 = *reinterpret_cast<const float2*>(krow);
          k0[0] = k2.x; k0[1] = k2.y;
        } else {
          k0[0] = krow[0];
        }
      };
      read_k(0, kc0, kc1);
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        f32x4 kn0 = kc0, kn1 = kc1;
        if (reg + 1 < 16) read_k(reg + 1, kn0, kn1);
#pragma unroll
        for (int b = 0; b < (NB < 4 ? NB : 4); ++b)
          G[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(kc0[b], coef[reg], G[b], 0, 0, 0);
        if constexpr (NB == 8) {
#pragma unroll
          for (int b = 0; b < 4; ++b)
            G[4 + b] = __builtin_amdgcn_mfma_f32_32x32x2f32(kc1[b], coef[reg], G[4 + b], 0, 0, 0);
        }
        kc0 = kn0; kc1 = kn1;
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  };

  if constexpr (STAG) {
    // ---- bf16x3, 8 waves: ONE barrier per tile, placed between [GEMM1(t), LDS store of tile t+1] and [epilogue(t),
    // GEMM2(t)], with a 3-buffer ring.  A wave that passes barrier t knows tile t+1 is complete, so after its GEMM2(t) it
    // runs straight into GEMM1(t+1); it only waits at barrier t+1 for the others to finish iteration t.  The two waves of
    // a SIMD therefore drift up to an epilogue + GEMM2 apart: one wave's softmax VALU runs under the other's MFMAs instead
    // of both sitting in the same phase (bf16 MFMAs leave 3/4 of the SIMD's issue slots free; MI355X_MICROARCH.md "Two
    // waves per SIMD").  Buffer t is overwritten by the store of iteration t+2, which is behind barrier t+1, i.e. after
    // every wave has finished GEMM2(t).  Same code and same arithmetic order for every wave. ----
    if (ntiles > 1) load_tile(1);
    for (int t = 0; t < ntiles; ++t) {
      const float* T = smem + (t % 3) * BUF_F;
      const f32x16 X = gemm1(T);
      if (t + 1 < ntiles) store_tile((t + 1) % 3);
      if (t + 2 < ntiles) load_tile(t + 2);
      __syncthreads();
      float coef[16];
      epilogue(T, t, X, coef);
      gemm2(T, coef);
    }
  } else if constexpr (FROM_S && TT_S_PREFETCH == 2) {
    // ---- BWD_S: epilogue -> GEMM2 per tile on the stored dot products; those of tiles t+1 AND t+2 are in flight while tile
    // t is worked on (two register sets, the body instantiated for each: f32 170 -> 168 us, bf16x3 96 -> 89 us) ----
    f32x16 xa, xb;
#pragma unroll
    for (int i = 0; i < 16; ++i) { xa[i] = 0.f; xb[i] = 0.f; }
    if (ntiles > 0) load_S(0, xa);
    if (ntiles > 1) load_S(1, xb);
    auto tile_step = [&](int t, f32x16& xs_t) {
      if constexpr (PREC == 1) { if (t + 1 < ntiles) load_tile(t + 1); }
      const float* T = smem + (t % NBUF) * BUF_F;
      const f32x16 X = xs_t;
      float coef[16];
      epilogue(T, t, X, coef);
#if TT_ABL_BWDS_NOSTAGE      // (timing-only ablation, results wrong: no K-tile staging and no barrier in the dc pass)
      if (t + 2 < ntiles) load_S(t + 2, xs_t);
      gemm2(T, coef);
#else
      if constexpr (PREC == 0) { if (t + TPB < ntiles) load_tile(t + TPB); }
      if (t + 2 < ntiles) load_S(t + 2, xs_t);
      gemm2(T, coef);
      if (t + TPB < ntiles) store_tile((t + TPB) % NBUF);
      if ((t % TPB) == TPB - 1) __syncthreads();
#endif
    };
    for (int t = 0; t < ntiles; t += 2) {
      tile_step(t, xa);
      if (t + 1 < ntiles) tile_step(t + 1, xb);
    }
  } else {
    // ---- every wave runs GEMM1 -> epilogue -> GEMM2 per tile; 2 LDS buffers, one barrier per tile ----
    // History (r02 -> r03).  In r02 a build of this loop with its body in a lambda called twice per iteration (the form the
    // BWD_S branch above uses) gave, for score_kernel<256, FWD>, wrong values in about half the rows, different from run to
    // run; the note left here said its loop "exit test is a vector compare".  That build was never committed.  r03 rebuilt
    // the form from the description (TT_LOOP_LAMBDA=1, below) with the same compiler: its ISA has SCALAR loop control
    // (s_cmp / s_cbranch_scc on the tile count), the `s_waitcnt vmcnt(0)` in front of every staged tile's ds_write and
    // `s_waitcnt lgkmcnt(0)` in front of every s_barrier exactly as the plain loop, and it passes the parity and determinism
    // tests on the GPU three runs out of three (profiles/r03_loop_lambda_audit.txt) - so the lambda form as such is not the
    // cause.  What the r02 note does pin down is a trip count held in VGPRs: results that change from run to run need waves
    // that disagree about how often they reach the barrier, and the only way the tile loop gets there is a per-lane copy of
    // `ntiles` / `t` (the quotient blockIdx / nsplit is computed on the vector ALU) that the compiler no longer proves uniform
    // once the loop body is large enough to spill or re-materialise it (D = 256: 240+ VGPRs).  The fix is by construction:
    // split, row block and ntiles go through readfirstlane (top of the kernel), so in EVERY loop form the compiler sees
    // scalar control, and tests/isa_audit/audit_barriers.py verifies on the built ISA - for all 112 instantiations, both loop forms -
    // that each barrier loop closes and exits on scalar branches and that no barrier can be skipped under a lane mask (the
    // script's negative control, a barrier loop on a per-lane trip count, is flagged).  The shipping BWD_S loop above
    // satisfies the same condition.  tests: test_retrieval_baseline_configs[1024-256], test_retrieval_is_deterministic
    // (dims 128 and 256, every form, rank pass), test_retrieval_odd_and_short_tile_counts.
    f32x16 xs;
#pragma unroll
    for (int i = 0; i < 16; ++i) xs[i] = 0.f;
    if constexpr (FROM_S) { if (ntiles > 0) load_S(0, xs); }
#if TT_LOOP_LAMBDA
    auto tile_step = [&](int t) {
      if constexpr (MODE == MODE_FWD || MODE == MODE_RANK || PREC == 1) { if (t + 1 < ntiles) load_tile(t + 1); }
      const float* T = smem + (t % NBUF) * BUF_F;
      f32x16 X;
      if constexpr (FROM_S) X = xs; else X = gemm1(T);
      if constexpr (SPLIT) exchange(X);
      if constexpr (TO_S) store_S(t, X);
      float coef[16];
      epilogue(T, t, X, coef);
      if constexpr (IS_BWD || IS_FUSED) {
        if constexpr (PREC == 0) { if (t + TPB < ntiles) load_tile(t + TPB); }
        if constexpr (FROM_S) { if (t + 1 < ntiles) load_S(t + 1, xs); }
        gemm2(T, coef);
      }
      if (t + TPB < ntiles) store_tile((t + TPB) % NBUF);
      if ((t % TPB) == TPB - 1) __syncthreads();
    };
    for (int t = 0; t < ntiles; t += 2) {
      tile_step(t);
      if (t + 1 < ntiles) tile_step(t + 1);
    }
#else
    for (int t = 0; t < ntiles; ++t) {
      // FWD/RANK: prefetch the next tile at the top.  BWD/FUSED: registers are tight (rf + G + X + coef), so the
      // prefetch is issued just before GEMM2, whose 16*NB MFMAs (>= 1.7 us at D=128) cover its latency.
      // bf16x3: a tile is ~1 us of MFMAs, less than a global round trip under load: the prefetch goes to the top too
      if constexpr (MODE == MODE_FWD || MODE == MODE_RANK || PREC == 1) { if (t + 1 < ntiles) load_tile(t + 1); }
      const float* T = smem + (t % NBUF) * BUF_F;
      f32x16 X;
      if constexpr (FROM_S) X = xs; else X = gemm1(T);
      if constexpr (SPLIT) exchange(X);
      if constexpr (TO_S) store_S(t, X);
      float coef[16];
      epilogue(T, t, X, coef);
      if constexpr (IS_BWD || IS_FUSED) {
        if constexpr (PREC == 0) { if (t + TPB < ntiles) load_tile(t + TPB); }
        if constexpr (FROM_S) { if (t + 1 < ntiles) load_S(t + 1, xs); }     // next tile's dot products, under GEMM2
        gemm2(T, coef);
      }
      if (t + TPB < ntiles) store_tile((t + TPB) % NBUF);
      if ((t % TPB) == TPB - 1) __syncthreads();       // (uniform) tiles of the next group are complete, this group's buffers free
    }
#endif
  }

  // ---- epilogue ----
  if constexpr (IS_FUSED) {
    const float L = run_l + __shfl_xor(run_l, 32);      // both halves share run_m
    if (r_ok && hw == 0) {
      if (h == 0) {
        p.part_m[(int64_t)split * p.n_r + r] = run_m;
        p.part_l[(int64_t)split * p.n_r + r] = L;
      }
      if (have_pos) p.pos2[r] = pos;
    }
  }
  if constexpr (MODE == MODE_RANK) {
    const int total = cnt + __shfl_xor(cnt, 32);
    if (r_ok && h == 0) p.part_cnt[(int64_t)split * p.n_r + r] = total;
  } else if constexpr (MODE == MODE_FWD) {
    const float om = __shfl_xor(run_m, 32);
    const float ol = __shfl_xor(run_l, 32);
    const float M = fmaxf(run_m, om);
    const float L = run_l * __builtin_amdgcn_exp2f(run_m - M) + ol * __builtin_amdgcn_exp2f(om - M);
    if (r_ok) {
      if (h == 0) {
        p.part_m[(int64_t)split * p.n_r + r] = M;
        p.part_l[(int64_t)split * p.n_r + r] = L;
      }
      if (have_pos) p.pos2[r] = pos;
    }
  } else {
    if (r_ok) {
      float* out = p.slab + ((int64_t)split * p.n_r + r) * D + 128 * hw;
      if constexpr (PREC == 1) {          // G[b][reg] = G^T[d = 32 b + acc row(reg, h)][r]: registers 4g .. 4g+3 are 4 consecutive d
#pragma unroll
        for (int b = 0; b < NBW; ++b)
#pragma unroll
          for (int g = 0; g < 4; ++g)
            *reinterpret_cast<f32x4*>(out + 32 * b + 8 * g + 4 * h) = f32x4{G[b][4 * g], G[b][4 * g + 1], G[b][4 * g + 2], G[b][4 * g + 3]};
      } else
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const int i = tt::acc_row(reg, 0) + 4 * h;
        if constexpr (NB == 8) {
          *reinterpret_cast<f32x4*>(out + 8 * i) = f32x4{G[0][reg], G[1][reg], G[2][reg], G[3][reg]};
          *reinterpret_cast<f32x4*>(out + 8 * i + 4) = f32x4{G[4][reg], G[5][reg], G[6][reg], G[7][reg]};
        } else if constexpr (NB == 4) {
          *reinterpret_cast<f32x4*>(out + 4 * i) = f32x4{G[0][reg], G[1][reg], G[2][reg], G[3][reg]};
        } else if constexpr (NB == 2) {
          *reinterpret_cast<float2*>(out + 2 * i) = float2{G[0][reg], G[1][reg]};
        } else {
          out[i] = G[0][reg];
        }
      }
    }
  }
}

// ---- small kernels around the main pass -------------------------------------------------------

// -log2(clip(p, 1e-6, 1))  (tfrs SamplingProbablityCorrection, log2 domain)
__global__ __launch_bounds__(256) void prob_bias_kernel(const float* __restrict__ prob, float* __restrict__ out, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) out[i] = -__log2f(fminf(fmaxf(prob[i], 1e-6f), 1.0f));
}

// a = -lse*log2e ; s = w * inv_t * gscale
__global__ __launch_bounds__(256) void bwd_prep_kernel(const float* __restrict__ lse, const float* __restrict__ w,
                                                       float* __restrict__ a, float* __restrict__ s, int64_t n,
                                                       float scale) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) {
    a[i] = -lse[i] * kLog2e;
    s[i] = (w != nullptr ? w[i] : 1.0f) * scale;
  }
}

// lse / per-row loss from the per-split (max, sum) pairs (the total is summed by sum_rows_kernel in a fixed order)
__global__ __launch_bounds__(256) void fwd_combine_kernel(const float* __restrict__ part_m, const float* __restrict__ part_l,
                                                          const float* __restrict__ pos2, const float* __restrict__ w,
                                                          int64_t n_r, int nsplit, float* __restrict__ lse,
                                                          float* __restrict__ per_row) {
  const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (r >= n_r) return;
  float M = kNegBig;
  for (int s = 0; s < nsplit; ++s) M = fmaxf(M, part_m[(int64_t)s * n_r + r]);
  float L = 0.f;
  for (int s = 0; s < nsplit; ++s)
    L += part_l[(int64_t)s * n_r + r] * __builtin_amdgcn_exp2f(part_m[(int64_t)s * n_r + r] - M);
  const float lse2 = M + __log2f(L);
  lse[r] = lse2 * kLn2;
  per_row[r] = (lse2 - pos2[r]) * kLn2 * (w != nullptr ? w[r] : 1.0f);
}

// FUSED pass 1 epilogue: per-split (max, sum, G) -> lse, per-row loss, the per-row terms of the dc pass, and
//   dq[r] = (w_r/T*g) * ( sum_s G_s[r]*2^(m_s-M) / L  -  c[r+diag] ).
// dim/4 lanes per row (8 .. 64: one float4 column each - no idle half rows at dim 64, no second column trip at dim 256),
// 256 >> lpr_log2 rows per block.  Splits are taken 8 at a time, every chunk's loads issued together from clamped (valid)
// split indices; additions in split order.  (Through r02: 32 lanes per row whatever the dim, and the splits past the 8th -
// B <= 4096 runs 16 or more - one dependent load after the other: 11.9 us at cfg2.)
__global__ __launch_bounds__(256) void fused_combine_kernel(const float* __restrict__ part_m, const float* __restrict__ part_l,
                                                            const float* __restrict__ pos2, const float* __restrict__ w,
                                                            const f32x4* __restrict__ slab, const f32x4* __restrict__ cpos,
                                                            int64_t n_r, int d4, int lpr_log2, int nsplit, float scale,
                                                            float* __restrict__ lse, float* __restrict__ per_row,
                                                            float* __restrict__ aq, float* __restrict__ sq,
                                                            f32x4* __restrict__ dq) {
  const int rpb = 256 >> lpr_log2;
  const int64_t row = (int64_t)blockIdx.x * rpb + (threadIdx.x >> lpr_log2);
  const int lane = threadIdx.x & ((1 << lpr_log2) - 1);
  if (row >= n_r) return;
  constexpr int U = 8;
  const int c = lane < d4 ? lane : d4 - 1;           // (lpr == d4 for the supported dims; clamped all the same)
  const int last = nsplit - 1;
  // The slab rows of the first 8 splits do not depend on the row statistics: their loads are issued first, the (max, sum)
  // passes over part_m / part_l run underneath them.
  f32x4 v0[U];
  const f32x4 cp = cpos[row * d4 + c];
#pragma unroll
  for (int u = 0; u < U; ++u) v0[u] = slab[((int64_t)(u < last ? u : last) * n_r + row) * d4 + c];
  const float p2 = pos2[row];
  const float wr = (w != nullptr ? w[row] : 1.0f);
  float pm0[U];                                      // the first chunk's maxima stay in registers for the sum below
  float M = kNegBig;
  for (int s0 = 0; s0 < nsplit; s0 += U) {
    float pm[U];
#pragma unroll
    for (int u = 0; u < U; ++u) pm[u] = part_m[(int64_t)(s0 + u < last ? s0 + u : last) * n_r + row];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      M = fmaxf(M, pm[u]);                           // (a clamped duplicate of the last split changes no maximum)
      if (s0 == 0) pm0[u] = pm[u];
    }
  }
  float L = 0.f;
  for (int s0 = 0; s0 < nsplit; s0 += U) {
    float pm[U], pl[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t o = (int64_t)(s0 + u < last ? s0 + u : last) * n_r + row;
      pm[u] = part_m[o];
      pl[u] = part_l[o];
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (s0 + u < nsplit) L += pl[u] * __builtin_amdgcn_exp2f(pm[u] - M);
  }
  const float lse2 = M + __log2f(L);
  const float sr = wr * scale;
  if (lane == 0) {
    lse[row] = lse2 * kLn2;
    per_row[row] = (lse2 - p2) * kLn2 * wr;
    aq[row] = -lse2;
    sq[row] = sr;
  }
  const float inv_l = 1.0f / L;
  f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int u = 0; u < U; ++u)
    if (u < nsplit) acc += v0[u] * __builtin_amdgcn_exp2f(pm0[u] - M);
  for (int s0 = U; s0 < nsplit; s0 += U) {           // splits beyond the first 8, in order, 8 loads at a time
    f32x4 v[U];
    float pm[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t sidx = s0 + u < last ? s0 + u : last;
      v[u] = slab[(sidx * n_r + row) * d4 + c];
      pm[u] = part_m[sidx * n_r + row];
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (s0 + u < nsplit) acc += v[u] * __builtin_amdgcn_exp2f(pm[u] - M);
  }
  if (lane < d4) dq[row * d4 + lane] = (acc * inv_l - cp) * sr;
}

// thr[r] = c1 * <q_r, c_pos(r)> + bias[pos(r)]  (log2 domain): the positive's logit, 32 lanes per row
__global__ __launch_bounds__(256) void pos_logit_kernel(const f32x4* __restrict__ q, const f32x4* __restrict__ c,
                                                        const int64_t* __restrict__ pos_idx, const float* __restrict__ bias,
                                                        int64_t nq, int64_t nc, int d4, float c1, float* __restrict__ thr,
                                                        int64_t diag) {
  const int64_t row = (int64_t)blockIdx.x * 8 + (threadIdx.x >> 5);
  const int lane = threadIdx.x & 31;
  if (row >= nq) return;
  const int64_t pc = pos_idx != nullptr ? pos_idx[row] : row + diag;
  float acc = 0.f;
  if (pc >= 0 && pc < nc) {
    for (int k = lane; k < d4; k += 32) {
      const f32x4 a = q[row * d4 + k], b = c[pc * d4 + k];
      acc += a[0] * b[0] + a[1] * b[1] + a[2] * b[2] + a[3] * b[3];
    }
  }
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
  if (lane == 0) thr[row] = (pc >= 0 && pc < nc) ? __builtin_fmaf(acc, c1, bias != nullptr ? bias[pc] : 0.f) : 3.0e38f;
}

__global__ __launch_bounds__(256) void rank_combine_kernel(const int32_t* __restrict__ part_cnt, int64_t n_r, int nsplit,
                                                           int32_t* __restrict__ rank) {
  const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (r >= n_r) return;
  int tot = 0;
  for (int s = 0; s < nsplit; ++s) tot += part_cnt[(int64_t)s * n_r + r];
  rank[r] = tot;
}

// ---- hard-negative mining (tfrs.layers.loss.HardNegativeMining): per query keep the positive and the k highest-scoring
// negatives.  The kernels only need a per-row THRESHOLD; it is found on a materialised row of raw dot products
// (scratch [nq][nc], written by the tiled GEMM) with a 4-pass MSB-first radix select (8-bit digits, LDS histogram) of
// the k-th largest negative logit, then the largest value below it; thr = their midpoint, so the later comparisons
// inside the fused kernels cannot flip on rounding.  Ties AT the k-th value are all kept.  One workgroup per row.
__device__ __forceinline__ uint32_t f2key(float v) {
  const uint32_t u = __float_as_uint(v);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);       // monotone: larger float -> larger key
}
__device__ __forceinline__ float key2f(uint32_t k) {
  return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}

__global__ __launch_bounds__(256) void hardneg_select_kernel(const float* __restrict__ S, int64_t nc, float c1,
                                                            const float* __restrict__ bias2, const int64_t* __restrict__ ids,
                                                            int64_t diag, int k, float* __restrict__ thr) {
  __shared__ int hist[256];
  __shared__ uint32_t sh_prefix, sh_mask, sh_lower;
  __shared__ int sh_remaining, sh_valid;
  const int tid = threadIdx.x;
  const int64_t row = blockIdx.x;
  const int64_t pos = row + diag;
  const float* srow = S + row * nc;
  const int64_t idpos = ids != nullptr ? ids[pos] : 0;
  auto valid = [&](int64_t j) { return j != pos && !(ids != nullptr && ids[j] == idpos); };
  auto keyof = [&](int64_t j) { return f2key(__builtin_fmaf(srow[j], c1, bias2 != nullptr ? bias2[j] : 0.f)); };
  if (tid == 0) { sh_prefix = 0; sh_mask = 0; sh_remaining = k; sh_valid = 0; sh_lower = 0; }
  __syncthreads();
  int nv = 0;
  for (int64_t j = tid; j < nc; j += 256) nv += valid(j) ? 1 : 0;
  atomicAdd(&sh_valid, nv);
  __syncthreads();
  if (k >= sh_valid) {                                       // fewer negatives than k: keep everything
    if (tid == 0) thr[row] = -3.0e38f;
    return;
  }
  for (int pass = 0; pass < 4; ++pass) {
    const int shift = 24 - 8 * pass;
    hist[tid] = 0;
    __syncthreads();
    const uint32_t prefix = sh_prefix, mask = sh_mask;
    for (int64_t j = tid; j < nc; j += 256) {
      if (!valid(j)) continue;
      const uint32_t key = keyof(j);
      if ((key & mask) == prefix) atomicAdd(&hist[(key >> shift) & 255u], 1);
    }
    __syncthreads();
    if (tid == 0) {
      int cum = 0, rem = sh_remaining;
      for (int b = 255; b >= 0; --b) {
        if (cum + hist[b] >= rem) {
          sh_prefix = prefix | ((uint32_t)b << shift);
          sh_mask = mask | (255u << shift);
          sh_remaining = rem - cum;
          break;
        }
        cum += hist[b];
      }
    }
    __syncthreads();
  }
  const uint32_t key_k = sh_prefix;                          // exact key of the k-th largest negative
  uint32_t lower = 0;
  for (int64_t j = tid; j < nc; j += 256) {
    if (!valid(j)) continue;
    const uint32_t key = keyof(j);
    if (key < key_k && key > lower) lower = key;
  }
  atomicMax(&sh_lower, lower);
  __syncthreads();
  if (tid == 0) {
    const float vk = key2f(key_k);
    const float vn = sh_lower != 0 ? key2f(sh_lower) : vk - 1.0f;
    thr[row] = 0.5f * vk + 0.5f * vn;
  }
}

// hq = thr + aq : the threshold in the domain of the BWD passes (logit2 - lse2)
__global__ __launch_bounds__(256) void hn_shift_kernel(const float* __restrict__ thr, const float* __restrict__ aq,
                                                       float* __restrict__ hq, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) hq[i] = thr[i] + aq[i];
}

// loss = sum_r per_row[r], fixed order (one workgroup)
__global__ __launch_bounds__(1024) void sum_rows_kernel(const float* __restrict__ per_row, int64_t n, float* __restrict__ loss) {
  __shared__ float red[1024];
  float acc = 0.f;
  for (int64_t r = threadIdx.x; r < n; r += 1024) acc += per_row[r];
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int s = 512; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) loss[0] = red[0];
}

// out[i] = sum_s slab[s][i], s ascending (float4).  With per_row != NULL the launch has one EXTRA workgroup (the last)
// that computes loss = sum_r per_row[r] in exactly sum_rows_kernel's order (1024 strided partial sums, then the
// binary tree), so the fused training form needs no separate single-workgroup launch for the scalar.
__global__ __launch_bounds__(256) void reduce_slabs_kernel(const f32x4* __restrict__ slab, f32x4* __restrict__ out,
                                                           int64_t n4, int nsplit, const float* __restrict__ per_row,
                                                           int64_t n_rows, float* __restrict__ loss) {
  if (per_row != nullptr && blockIdx.x == gridDim.x - 1) {
    __shared__ float red[256];
    const int t = threadIdx.x;
    float a[4] = {0.f, 0.f, 0.f, 0.f};
    for (int64_t r0 = t; r0 < n_rows; r0 += 8 * 1024) {        // 32 loads in flight per thread, adds in row order
      float v[8][4];
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int64_t r = r0 + 1024 * i + 256 * j;
          v[i][j] = r < n_rows ? per_row[r] : 0.f;
        }
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (r0 + 1024 * i + 256 * j < n_rows) a[j] += v[i][j];
    }
    red[t] = (a[0] + a[2]) + (a[1] + a[3]);          // tree steps s = 512 and s = 256
    __syncthreads();
    for (int s2 = 128; s2 > 0; s2 >>= 1) {
      if (t < s2) red[t] += red[t + s2];
      __syncthreads();
    }
    if (t == 0) loss[0] = red[0];
    return;
  }
  const int64_t nblk = per_row != nullptr ? gridDim.x - 1 : gridDim.x;
  const int64_t stride = nblk * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
    f32x4 a = slab[i];
    for (int s = 1; s < nsplit; ++s) a += slab[(int64_t)s * n4 + i];
    out[i] = a;
  }
}

// ---- host side --------------------------------------------------------------------------------

#ifndef TT_SCORE_WAVES
#define TT_SCORE_WAVES 4      // waves (32-row fragments) per workgroup of the exact-f32 passes
#endif
#ifndef TT_SCORE_WGS
#define TT_SCORE_WGS 512      // workgroups a pass aims for (two 4-wave workgroups per CU)
#endif
#ifndef TT_BWDS_WGS
#define TT_BWDS_WGS 512       // workgroups the BWD_S pass (dc from the stored dot products) aims for
#endif
int choose_nsplit_pow2(int64_t n_r, int64_t n_c, int target_wgs = TT_SCORE_WGS) {
  const int64_t nrb = (n_r + 32 * TT_SCORE_WAVES - 1) / (32 * TT_SCORE_WAVES);
  int ns = 1;
  // 512 workgroups = two per CU measured best at B = 8192 (256: 291 us, 512: 272 us, 1024: 283 us, 2048: 294 us per launch)
  while (nrb * ns < target_wgs && (int64_t)ns * 2 * 64 <= n_c && ns < 64) ns *= 2;
  return ns;
}

// r04: the split count by a model of the launch instead of "double until 512 workgroups".  A scorer workgroup runs for the WHOLE
// launch (its row block against 1/ns of the columns), and `slots` of them are resident at once (two 4-wave or one 8-wave
// workgroup per CU): n_r = 8192 gives 64 row blocks x 8 splits = exactly 512 - and n_r = 8200 gave 65 x 8 = 520, eight workgroups
// that ran a second round alone: batch 8200 took 0.879 ms against 0.558 for 8192 (pass 1 289 -> 420 us, pass 2 173 -> 318;
// 4100 against 4096: 0.318 vs 0.214 ms).  cost(ns) = rounds(ns) / ns x (1 + 0.005 ns): the rounds the grid needs, each 1/ns of the
// columns long, plus the measured price of more, shorter workgroups (1024 instead of 512 at 8192: +4 %).  The minimum over ns =
// 1..64 is the old choice for every power-of-two shape (checked over n_r, n_c in 64 .. 262144: scratch/r04_nsplit_model.py) and e.g.
// 15 splits (975 workgroups, two nearly full rounds) for 8200.  Any ns is valid: the columns are cut at multiples of 32.
int choose_nsplit(int64_t n_r, int64_t n_c, int rows_per_wg = 32 * TT_SCORE_WAVES, int slots = TT_SCORE_WGS) {
  const int64_t nrb = (n_r + rows_per_wg - 1) / rows_per_wg;
  int best = 1;
  double best_cost = 0.0;
  for (int ns = 1; ns <= 64; ++ns) {
    if (ns > 1 && (int64_t)ns * 64 > n_c) break;
    const int64_t rounds = (nrb * ns + slots - 1) / slots;
    const double cost = (double)rounds / ns * (1.0 + 0.005 * ns);
    if (ns == 1 || cost < best_cost - 1e-12) { best = ns; best_cost = cost; }
  }
  return best;
}

int64_t align_up(int64_t x, int64_t a) { return (x + a - 1) / a * a; }

struct WsLayout {
  int ns_q, ns_c;           // splits for the passes whose stationary side is q / c
  int ns_cs;                // ... for the dc pass that reads the stored dot products (BWD_S)
  int64_t off_bias, off_aq, off_sq, off_hq, off_pm, off_pl, off_pos, off_slab, off_S, total_no_S, total;
};

WsLayout ws_layout(int64_t nq, int64_t nc, int32_t dim) {
  WsLayout w{};
  if (dim <= 128 && std::getenv("TT_NSPLIT_POW2") == nullptr) {
    w.ns_q = choose_nsplit(nq, nc);
    w.ns_c = choose_nsplit(nc, nq);
    w.ns_cs = choose_nsplit(nc, nq, 256, 256);     // the dc pass from the stored dot products: 8-wave workgroups, one per CU
    // (experiment hooks: TT_NSPLIT_Q / TT_NSPLIT_CS force the split counts of pass 1 / the dc pass)
    if (const char* e = std::getenv("TT_NSPLIT_Q")) { const int v = std::atoi(e); if (v >= 1 && v <= 64 && (int64_t)v * 64 <= nc) w.ns_q = v; }
    if (const char* e = std::getenv("TT_NSPLIT_CS")) { const int v = std::atoi(e); if (v >= 1 && v <= 64 && (int64_t)v * 64 <= nq) w.ns_cs = v; }
  } else {                                         // (dim 256: other workgroup shapes per kernel form; not re-measured - r03's rule)
    w.ns_q = choose_nsplit_pow2(nq, nc);
    w.ns_c = choose_nsplit_pow2(nc, nq);
    w.ns_cs = choose_nsplit_pow2(nc, nq, TT_BWDS_WGS);
  }
  int64_t o = 0;
  w.off_bias = o; o = align_up(o + nc * 4, 256);
  w.off_aq = o;   o = align_up(o + nq * 4, 256);
  w.off_sq = o;   o = align_up(o + nq * 4, 256);
  w.off_hq = o;   o = align_up(o + nq * 4, 256);
  w.off_pm = o;   o = align_up(o + (int64_t)w.ns_q * nq * 4, 256);
  w.off_pl = o;   o = align_up(o + (int64_t)w.ns_q * nq * 4, 256);
  w.off_pos = o;  o = align_up(o + nq * 4, 256);
  w.off_slab = o;
  const int64_t slab_q = (int64_t)w.ns_q * nq * dim * 4, slab_c = (int64_t)(w.ns_c > w.ns_cs ? w.ns_c : w.ns_cs) * nc * dim * 4;
  o = align_up(o + (slab_q > slab_c ? slab_q : slab_c), 256);
  w.total_no_S = o;
  // the training entry keeps the raw dot products [nq][nc] between its two passes (pass 2 reads them back instead of
  // recomputing them: half its matrix-pipe work for 8 bytes of hidden HBM traffic per logit)
  w.off_S = o;
  o = align_up(o + ((nq + 31) / 32) * ((nc + 31) / 32) * 4096, 256);
  w.total = o;
  return w;
}

// 4 waves (128 rows) per workgroup, two workgroups per CU.  An 8-wave variant whose two wave groups run one phase apart
// (GEMM1 | GEMM2, epilogue | GEMM1, GEMM2 | epilogue, a barrier per phase) was built and measured: no gain (277 vs 271 us).
// Ablation on MI355X (B = 8192, D = 128; us per launch): all 272 | no epilogue math 257 | no tile staging 270 |
// no GEMM1 171 | no GEMM2 171 | neither GEMM 58 | nothing but the loop skeleton 34.  So the two GEMMs together cost
// 220 us = 156 TF, the f32 MFMA peak; what is left is ~34 us of fixed cost (prologue, per-tile barriers, 33.5 MB of
// slab stores, drain) and ~20 us around the softmax epilogue that the partner wave's MFMAs do not hide (f32-input
// MFMA runs at the f32 vector rate; a 25 % cut of the epilogue's VALU instructions changed nothing measurable, so the
// cost is in the GEMM1 -> epilogue -> GEMM2 dependency hand-offs rather than in VALU throughput).
template <int D, int MODE, int PREC = 0>
// waves per workgroup of the exact-f32 training passes at D <= 128 (r02 A/B at cfg3: the dc pass with 8 waves = 256 stationary
// rows per workgroup, one workgroup per CU, half the tile staging per MFMA: 169.0 -> 165.8 us; pass 1 with 8: 276.8 -> 277.6)
#ifndef TT_BWDS_WG_WAVES
#define TT_BWDS_WG_WAVES 8
#endif
#ifndef TT_FUSEDS_WG_WAVES
#define TT_FUSEDS_WG_WAVES 4
#endif
constexpr int waves_for() {
  return (PREC == 1 && D == 128) ? 8 : (PREC == 1 ? 4 : ((MODE == MODE_BWD_S && D <= 128) ? TT_BWDS_WG_WAVES
                                                         : ((MODE == MODE_FUSED_S && D <= 128) ? TT_FUSEDS_WG_WAVES : TT_SCORE_WAVES)));
}     // bf16x3 at dim 128: 256-row workgroups, one per CU

template <int D, int MODE, int PREC = 0>
int launch_score(const ScoreArgs& a_in, bool has_ids, hipStream_t stream) {
  constexpr int W = waves_for<D, MODE, PREC>();
  constexpr int RPW = rows_per_wg<D, MODE, PREC, W>();
  const int64_t nrb = (a_in.n_r + RPW - 1) / RPW;
  const int64_t blocks = nrb * a_in.nsplit;
  // bf16x3 with 8 waves: a third tile buffer (staggered wave halves) + the R_lo fragments of the 8 waves
  const int lds_base = Geo<D, PREC>::LDS_BYTES * tiles_per_barrier<D, MODE, PREC>() +
                  ((PREC == 1 && W == 8) ? (TT_BX3_STAGGER ? Geo<D, PREC>::BUF_F * 4 : 0) + W * Geo<D, PREC>::KS * 64 * 16 : 0) +
                  (split_d<D, MODE, PREC>() ? W * 4096 : 0);          // + the pair's dot-product exchange
  const bool has_hn = (a_in.h_r != nullptr) || (a_in.h_c != nullptr);
  const ScoreArgs& a = a_in;
  auto go = [&](auto kern, int rfl_groups = 0) -> int {
    const int lds = lds_base + W * rfl_groups * 64 * 16;        // + the LDS-resident k-groups of the stationary fragment
    if (lds > 64 * 1024) {   // above the 64 KiB default the limit must be raised (cheap, idempotent)
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess)
        return tt::fail(TT_ERR_LAUNCH, "hipFuncSetAttribute(LDS %d) failed", lds);
    }
    constexpr const char* tag = MODE == MODE_FWD ? "score_fwd" : ((MODE == MODE_BWD || MODE == MODE_BWD_S) ? "score_bwd"
                                : ((MODE == MODE_FUSED || MODE == MODE_FUSED_S) ? "score_fused" : "score_rank"));
    tt::launch(tag, kern, dim3((unsigned)blocks), dim3(W * 64), (unsigned)lds, stream, a);
    return tt::check_launch(tag);
  };
  if (has_ids && has_hn) return go(score_kernel<D, MODE, true, true, W, PREC>, rf_lds_groups<D, MODE, true, true, PREC>());
  if (has_ids) return go(score_kernel<D, MODE, true, false, W, PREC>, rf_lds_groups<D, MODE, true, false, PREC>());
  if (has_hn) return go(score_kernel<D, MODE, false, true, W, PREC>, rf_lds_groups<D, MODE, false, true, PREC>());
  return go(score_kernel<D, MODE, false, false, W, PREC>);
}

// bf16x3 passes at the dims whose tiles are whole [32][128] bf16 images
template <int MODE>
int dispatch_score_bx3(int32_t dim, const ScoreArgs& a, bool has_ids, hipStream_t stream) {
  switch (dim) {
    case 128: return launch_score<128, MODE, 1>(a, has_ids, stream);
    case 256: return launch_score<256, MODE, 1>(a, has_ids, stream);
    default: return tt::fail(TT_ERR_UNSUPPORTED, "retrieval (bf16x3): dim %d not in {128,256}", dim);
  }
}

// the recomputing bf16x3 gradient passes are reachable at dim 256 only (TT_BX3_RECOMPUTE256, the A/B of keeping the dot products)
template <int MODE>
int dispatch_score_bx3_256(int32_t dim, const ScoreArgs& a, bool has_ids, hipStream_t stream) {
  if (dim != 256) return tt::fail(TT_ERR_UNSUPPORTED, "retrieval (bf16x3, recomputing form): dim %d != 256", dim);
  return launch_score<256, MODE, 1>(a, has_ids, stream);
}

template <int MODE>
int dispatch_score(int32_t dim, const ScoreArgs& a, bool has_ids, hipStream_t stream) {
  switch (dim) {
    case 32: return launch_score<32, MODE>(a, has_ids, stream);
    case 64: return launch_score<64, MODE>(a, has_ids, stream);
    case 128: return launch_score<128, MODE>(a, has_ids, stream);
    case 256: return launch_score<256, MODE>(a, has_ids, stream);
    default: return tt::fail(TT_ERR_UNSUPPORTED, "retrieval: dim %d not in {32,64,128,256}", dim);
  }
}

int check_common(const char* fn, const float* q, const float* c, int64_t nq, int64_t nc, int32_t dim,
                 int64_t diag_offset, const void* ws, int64_t ws_bytes, bool need_S = false) {
  TT_REQUIRE(q && c && ws, "%s: null pointer", fn);
  TT_REQUIRE(nq > 0 && nc > 0, "%s: nq and nc must be positive", fn);
  TT_REQUIRE(diag_offset >= 0 && nq + diag_offset <= nc, "%s: need 0 <= diag_offset and nq + diag_offset <= nc", fn);
  TT_REQUIRE(dim == 32 || dim == 64 || dim == 128 || dim == 256, "%s: dim %d not in {32,64,128,256}", fn, dim);
  TT_REQUIRE(tt::aligned16(q) && tt::aligned16(c), "%s: q/c must be 16-byte aligned", fn);
  TT_REQUIRE((reinterpret_cast<uintptr_t>(ws) & 255u) == 0, "%s: workspace must be 256-byte aligned", fn);
  const WsLayout wl = ws_layout(nq, nc, dim);
  const int64_t need = need_S ? wl.total : wl.total_no_S;
  if (ws_bytes < need)
    return tt::fail(TT_ERR_WORKSPACE, "%s: workspace %lld < %lld bytes", fn, (long long)ws_bytes, (long long)need);
  return TT_OK;
}

}  // namespace

// How many ways a pass of the scorer cuts its streamed side (a host query: what sizes the workspace, and what a test can pin):
// pass 0 = stationary q (loss + dq, forward, rank), 1 = stationary c (dc, recomputing the products), 2 = dc from the stored products.
extern "C" int32_t tt_retrieval_num_splits(int64_t nq, int64_t nc, int32_t dim, int32_t pass) {
  if (nq <= 0 || nc <= 0 || dim <= 0 || pass < 0 || pass > 2) return 0;
  const WsLayout w = ws_layout(nq, nc, dim);
  return pass == 0 ? w.ns_q : (pass == 1 ? w.ns_c : w.ns_cs);
}

extern "C" int64_t tt_retrieval_workspace_bytes(int64_t nq, int64_t nc, int32_t dim) {
  if (nq <= 0 || nc <= 0 || dim <= 0) return 0;
  return ws_layout(nq, nc, dim).total;
}

// the forward-only (validation) and separate-backward entries need no logit buffer: everything in front of it
extern "C" int64_t tt_retrieval_fwd_workspace_bytes(int64_t nq, int64_t nc, int32_t dim) {
  if (nq <= 0 || nc <= 0 || dim <= 0) return 0;
  return ws_layout(nq, nc, dim).total_no_S;
}

// the rank (metric) pass uses the regions in front of the gradient slabs only: bias, threshold, per-split counts
extern "C" int64_t tt_retrieval_rank_workspace_bytes(int64_t nq, int64_t nc, int32_t dim) {
  if (nq <= 0 || nc <= 0 || dim <= 0) return 0;
  return ws_layout(nq, nc, dim).off_pl;
}

static int retrieval_fwd(int prec, const float* q, const float* c, int64_t nq, int64_t nc, int32_t dim,
                         int64_t diag_offset, float inv_temperature, const float* sample_weight,
                         const float* cand_prob, const int64_t* cand_ids, const float* hard_thr,
                         void* workspace, int64_t workspace_bytes, float* lse, float* per_row, float* loss,
                         tt_stream_t stream_) {
  int rc = check_common("tt_retrieval_fwd_f32", q, c, nq, nc, dim, diag_offset, workspace, workspace_bytes);
  if (rc != TT_OK) return rc;
  TT_REQUIRE(lse && per_row && loss, "tt_retrieval_fwd_f32: null output pointer");
  hipStream_t stream = tt::as_stream(stream_);
  const WsLayout w = ws_layout(nq, nc, dim);
  char* ws = static_cast<char*>(workspace);
  float* bias = reinterpret_cast<float*>(ws + w.off_bias);
  if (cand_prob != nullptr) {
    hipLaunchKernelGGL(prob_bias_kernel, dim3((unsigned)((nc + 255) / 256)), dim3(256), 0, stream, cand_prob, bias, nc);
    if ((rc = tt::check_launch("prob_bias")) != TT_OK) return rc;
  }
  ScoreArgs a{};
  a.R = q; a.K = c; a.n_r = nq; a.n_c = nc; a.diag = diag_offset;
  a.c1 = kLog2e * inv_temperature;
  a.a_c = cand_prob != nullptr ? bias : nullptr;
  a.id_r = cand_ids != nullptr ? cand_ids + diag_offset : nullptr;
  a.id_c = cand_ids;
  a.h_r = hard_thr;
  a.nsplit = w.ns_q;
  a.c_per_split = align_up((nc + a.nsplit - 1) / a.nsplit, 32);
  a.part_m = reinterpret_cast<float*>(ws + w.off_pm);
  a.part_l = reinterpret_cast<float*>(ws + w.off_pl);
  a.pos2 = reinterpret_cast<float*>(ws + w.off_pos);
  rc = prec == 1 ? dispatch_score_bx3<MODE_FWD>(dim, a, cand_ids != nullptr, stream) : dispatch_score<MODE_FWD>(dim, a, cand_ids != nullptr, stream);
  if (rc != TT_OK) return rc;
  tt::ProfScope prof("score_aux", stream);
  hipLaunchKernelGGL(fwd_combine_kernel, dim3((unsigned)((nq + 255) / 256)), dim3(256), 0, stream, a.part_m, a.part_l, a.pos2,
                     sample_weight, nq, a.nsplit, lse, per_row);
  if ((rc = tt::check_launch("fwd_combine")) != TT_OK) return rc;
  hipLaunchKernelGGL(sum_rows_kernel, dim3(1), dim3(1024), 0, stream, per_row, nq, loss);
  return tt::check_launch("sum_rows");
}

extern "C" int tt_retrieval_fwd_f32(const float* q, const float* c, int64_t nq, int64_t nc, int32_t dim,
                                    int64_t diag_offset, float inv_temperature, const float* sample_weight,
                                    const float* cand_prob, const int64_t* cand_ids, const float* hard_thr,
                                    void* workspace, int64_t workspace_bytes, float* lse, float* per_row, float* loss,
                                    tt_stream_t stream) {
  return retrieval_fwd(0, q, c, nq, nc, dim, diag_offset, inv_temperature, sample_weight, cand_prob, cand_ids, hard_thr, workspace,
                       workspace_bytes, lse, per_row, loss, stream);
}

// The validation pass with the logits' products on the bf16 matrix cores (the 6-product split of GEMM1: 2^-24 relative,
// f32 accumulation, f32 softmax): the forward pass is pure GEMM1, so the 6-product form is the whole kernel.  dim in {128, 256}.
extern "C" int tt_retrieval_fwd_bf16x3_f32(const float* q, const float* c, int64_t nq, int64_t nc, int32_t dim,
                                           int64_t diag_offset, float inv_temperature, const float* sample_weight,
                                           const float* cand_prob, const int64_t* cand_ids, const float* hard_thr,
                                           void* workspace, int64_t workspace_bytes, float* lse, float* per_row, float* loss,
                                           tt_stream_t stream) {
  if (dim != 128 && dim != 256) return tt::fail(TT_ERR_UNSUPPORTED, "tt_retrieval_fwd_bf16x3_f32: dim %d not in {128,256}", dim);
  return retrieval_fwd(1, q, c, nq, nc, dim, diag_offset, inv_temperature, sample_weight, cand_prob, cand_ids, hard_thr, workspace,
                       workspace_bytes, lse, per_row, loss, stream);
}

extern "C" int tt_retrieval_bwd_f32(const float* q, const float* c, int64_t nq, int64_t nc, int32_t dim,
                                    int64_t diag_offset, float inv_temperature, const float* sample_weight,
                                    const float* cand_prob, const int64_t* cand_ids, const float* hard_thr,
                                    const float* lse, float grad_scale, void* workspace, int64_t workspace_bytes,
                                    float* dq, float* dc, tt_stream_t stream_) {
  int rc = check_common("tt_retrieval_bwd_f32", q, c, nq, nc, dim, diag_offset, workspace, workspace_bytes);
  if (rc != TT_OK) return rc;
  TT_REQUIRE(lse && dq && dc, "tt_retrieval_bwd_f32: null lse/dq/dc");
  TT_REQUIRE(tt::aligned16(dq) && tt::aligned16(dc), "tt_retrieval_bwd_f32: dq/dc must be 16-byte aligned");
  hipStream_t stream = tt::as_stream(stream_);
  const WsLayout w = ws_layout(nq, nc, dim);
  char* ws = static_cast<char*>(workspace);
  float* bias = reinterpret_cast<float*>(ws + w.off_bias);
  float* aq = reinterpret_cast<float*>(ws + w.off_aq);
  float* sq = reinterpret_cast<float*>(ws + w.off_sq);
  float* slab = reinterpret_cast<float*>(ws + w.off_slab);
  if (cand_prob != nullptr) {
    hipLaunchKernelGGL(prob_bias_kernel, dim3((unsigned)((nc + 255) / 256)), dim3(256), 0, stream, cand_prob, bias, nc);
    if ((rc = tt::check_launch("prob_bias")) != TT_OK) return rc;
  }
  hipLaunchKernelGGL(bwd_prep_kernel, dim3((unsigned)((nq + 255) / 256)), dim3(256), 0, stream, lse, sample_weight, aq,
                     sq, nq, inv_temperature * grad_scale);
  if ((rc = tt::check_launch("bwd_prep")) != TT_OK) return rc;
  const float* biasp = cand_prob != nullptr ? bias : nullptr;
  float* hq = nullptr;
  if (hard_thr != nullptr) {
    hq = reinterpret_cast<float*>(ws + w.off_hq);
    hipLaunchKernelGGL(hn_shift_kernel, dim3((unsigned)((nq + 255) / 256)), dim3(256), 0, stream, hard_thr, aq, hq, nq);
    if ((rc = tt::check_launch("hn_shift")) != TT_OK) return rc;
  }

  // dq: stationary q, stream c
  {
    ScoreArgs a{};
    a.R = q; a.K = c; a.n_r = nq; a.n_c = nc; a.diag = diag_offset;
    a.c1 = kLog2e * inv_temperature;
    a.a_r = aq; a.s_r = sq; a.a_c = biasp; a.s_c = nullptr;
    a.h_r = hq;
    a.id_r = cand_ids != nullptr ? cand_ids + diag_offset : nullptr;
    a.id_c = cand_ids;
    a.nsplit = w.ns_q;
    a.c_per_split = align_up((nc + a.nsplit - 1) / a.nsplit, 32);
    a.slab = slab;
    if ((rc = dispatch_score<MODE_BWD>(dim, a, cand_ids != nullptr, stream)) != TT_OK) return rc;
    const int64_t n4 = nq * dim / 4;
    const int64_t blocks = (n4 + 255) / 256 < 2048 ? (n4 + 255) / 256 : 2048;
    tt::launch("score_aux", reduce_slabs_kernel, dim3((unsigned)blocks), dim3(256), 0, stream,
               reinterpret_cast<const f32x4*>(slab), reinterpret_cast<f32x4*>(dq), n4, a.nsplit, (const float*)nullptr, (int64_t)0, (float*)nullptr);
    if ((rc = tt::check_launch("reduce_slabs(dq)")) != TT_OK) return rc;
  }
  // dc: stationary c, stream q.  Candidates beyond nq + diag_offset have no positive: diag never matches.
  {
    ScoreArgs a{};
    a.R = c; a.K = q; a.n_r = nc; a.n_c = nq; a.diag = -diag_offset;
    a.c1 = kLog2e * inv_temperature;
    a.a_r = biasp; a.s_r = nullptr; a.a_c = aq; a.s_c = sq;
    a.h_c = hq;
    a.id_r = cand_ids;
    a.id_c = cand_ids != nullptr ? cand_ids + diag_offset : nullptr;
    a.nsplit = w.ns_c;
    a.c_per_split = align_up((nq + a.nsplit - 1) / a.nsplit, 32);
    a.slab = slab;
    if ((rc = dispatch_score<MODE_BWD>(dim, a, cand_ids != nullptr, stream)) != TT_OK) return rc;
    const int64_t n4 = nc * dim / 4;
    const int64_t blocks = (n4 + 255) / 256 < 2048 ? (n4 + 255) / 256 : 2048;
    tt::launch("score_aux", reduce_slabs_kernel, dim3((unsigned)blocks), dim3(256), 0, stream,
               reinterpret_cast<const f32x4*>(slab), reinterpret_cast<f32x4*>(dc), n4, a.nsplit, (const float*)nullptr, (int64_t)0, (float*)nullptr);
    if ((rc = tt::check_launch("reduce_slabs(dc)")) != TT_OK) return rc;
  }
  return TT_OK;
}

// Fused training entry: loss AND both gradients in two passes (8*B^2*D executed FLOPs instead of 10):
//   pass 1 (R = q, K = c, MODE_FUSED): online softmax + sum_c p*c  -> lse, per-row loss, dq
//   pass 2 (R = c, K = q, MODE_BWD)  : recompute with the final lse -> dc
static int retrieval_fwd_bwd(int prec, const float* q, const float* c, int64_t nq, int64_t nc, int32_t dim,
                             int64_t diag_offset, float inv_temperature, const float* sample_weight,
                             const float* cand_prob, const int64_t* cand_ids, const float* hard_thr,
                             float grad_scale, void* workspace, int64_t workspace_bytes, float* lse,
                             float* per_row, float* loss, float* dq, float* dc, tt_stream_t stream_) {
  // Both precisions keep pass 1's dot products for pass 2 (the workspace includes the buffer).  bf16x3 at dim 256 recomputed them
  // until r03 (TT_BX3_RECOMPUTE256=1 still does, for A/B): with the wave-pair kernels, B 32768: pass 1 / pass 2 = 6.10 / 5.87 ms
  // recomputing, 6.45 / 3.51 ms keeping them (exact f32: 8.77 / 5.17 ms) - profiles/r03_score_f32_vs_bf16x3.jsonl.
  static const bool recompute256 = [] { const char* e = getenv("TT_BX3_RECOMPUTE256"); return e != nullptr && e[0] == '1'; }();
  const bool bx3_keep = prec == 1 && (dim == 128 || !recompute256);
  int rc = check_common("tt_retrieval_fwd_bwd_f32", q, c, nq, nc, dim, diag_offset, workspace, workspace_bytes, true);
  if (rc != TT_OK) return rc;
  TT_REQUIRE(lse && per_row && loss && dq && dc, "tt_retrieval_fwd_bwd_f32: null output pointer");
  TT_REQUIRE(tt::aligned16(dq) && tt::aligned16(dc), "tt_retrieval_fwd_bwd_f32: dq/dc must be 16-byte aligned");
  hipStream_t stream = tt::as_stream(stream_);
  const WsLayout w = ws_layout(nq, nc, dim);
  char* ws = static_cast<char*>(workspace);
  float* bias = reinterpret_cast<float*>(ws + w.off_bias);
  float* aq = reinterpret_cast<float*>(ws + w.off_aq);
  float* sq = reinterpret_cast<float*>(ws + w.off_sq);
  float* slab = reinterpret_cast<float*>(ws + w.off_slab);
  float* smat = reinterpret_cast<float*>(ws + w.off_S);       // [nq][nc] raw dot products, pass 1 -> pass 2
  if (cand_prob != nullptr) {
    hipLaunchKernelGGL(prob_bias_kernel, dim3((unsigned)((nc + 255) / 256)), dim3(256), 0, stream, cand_prob, bias, nc);
    if ((rc = tt::check_launch("prob_bias")) != TT_OK) return rc;
  }
  const float* biasp = cand_prob != nullptr ? bias : nullptr;
  {
    ScoreArgs a{};
    a.R = q; a.K = c; a.n_r = nq; a.n_c = nc; a.diag = diag_offset;
    a.c1 = kLog2e * inv_temperature;
    a.a_c = biasp;
    a.h_r = hard_thr;
    a.S = smat; a.ldS = (nq + 31) / 32;
    a.id_r = cand_ids != nullptr ? cand_ids + diag_offset : nullptr;
    a.id_c = cand_ids;
    a.nsplit = w.ns_q;
    a.c_per_split = align_up((nc + a.nsplit - 1) / a.nsplit, 32);
    a.part_m = reinterpret_cast<float*>(ws + w.off_pm);
    a.part_l = reinterpret_cast<float*>(ws + w.off_pl);
    a.pos2 = reinterpret_cast<float*>(ws + w.off_pos);
    a.slab = slab;
    // (bf16x3 at dim 128 keeps the dot products too: 134 + 89 us against 123 + 124 us recomputing - with the row-major buffer
    // it had been 200 + 127 us, the blocked layout is what makes it pay.)
    rc = prec == 1 ? (bx3_keep ? dispatch_score_bx3<MODE_FUSED_S>(dim, a, cand_ids != nullptr, stream)
                               : dispatch_score_bx3_256<MODE_FUSED>(dim, a, cand_ids != nullptr, stream))
                   : dispatch_score<MODE_FUSED_S>(dim, a, cand_ids != nullptr, stream);
    if (rc != TT_OK) return rc;
    {
      int lpr_log2 = 3;                              // dim/4 lanes per row: 8 (dim 32) .. 64 (dim 256)
      while ((1 << lpr_log2) < dim / 4) ++lpr_log2;
      const int rpb = 256 >> lpr_log2;
      tt::launch("score_aux", fused_combine_kernel, dim3((unsigned)((nq + rpb - 1) / rpb)), dim3(256), 0, stream, a.part_m, a.part_l, a.pos2,
                         sample_weight, reinterpret_cast<const f32x4*>(slab),
                         reinterpret_cast<const f32x4*>(c + diag_offset * dim), nq, dim / 4, lpr_log2, a.nsplit,
                         inv_temperature * grad_scale, lse, per_row, aq, sq, reinterpret_cast<f32x4*>(dq));
      if ((rc = tt::check_launch("fused_combine")) != TT_OK) return rc;
      // loss = sum(per_row): by one extra workgroup of the dc slab reduction below
    }
  }
  float* hq = nullptr;
  if (hard_thr != nullptr) {
    hq = reinterpret_cast<float*>(ws + w.off_hq);
    hipLaunchKernelGGL(hn_shift_kernel, dim3((unsigned)((nq + 255) / 256)), dim3(256), 0, stream, hard_thr, aq, hq, nq);
    if ((rc = tt::check_launch("hn_shift")) != TT_OK) return rc;
  }
  {
    ScoreArgs a{};
    a.R = c; a.K = q; a.n_r = nc; a.n_c = nq; a.diag = -diag_offset;
    a.c1 = kLog2e * inv_temperature;
    a.a_r = biasp; a.s_r = nullptr; a.a_c = aq; a.s_c = sq;
    a.h_c = hq;
    a.id_r = cand_ids;
    a.id_c = cand_ids != nullptr ? cand_ids + diag_offset : nullptr;
    a.nsplit = (prec == 1 && !bx3_keep) ? w.ns_c : w.ns_cs;
    a.c_per_split = align_up((nq + a.nsplit - 1) / a.nsplit, 32);
    a.slab = slab;
    a.S = smat; a.ldS = (nq + 31) / 32;
    rc = prec == 1 ? (bx3_keep ? dispatch_score_bx3<MODE_BWD_S>(dim, a, cand_ids != nullptr, stream)
                                 : dispatch_score_bx3_256<MODE_BWD>(dim, a, cand_ids != nullptr, stream))
                   : dispatch_score<MODE_BWD_S>(dim, a, cand_ids != nullptr, stream);
    if (rc != TT_OK) return rc;
    const int64_t n4 = nc * dim / 4;
    const int64_t blocks = (n4 + 255) / 256 < 2048 ? (n4 + 255) / 256 : 2048;
    tt::launch("score_aux", reduce_slabs_kernel, dim3((unsigned)blocks + 1), dim3(256), 0, stream,
                       reinterpret_cast<const f32x4*>(slab), reinterpret_cast<f32x4*>(dc), n4, a.nsplit, per_row, nq, loss);
    if ((rc = tt::check_launch("reduce_slabs(dc)")) != TT_OK) return rc;
  }
  return TT_OK;
}

extern "C" int tt_retrieval_fwd_bwd_f32(const float* q, const float* c, int64_t nq, int64_t nc, int32_t dim,
                                        int64_t diag_offset, float inv_temperature, const float* sample_weight,
                                        const float* cand_prob, const int64_t* cand_ids, const float* hard_thr,
                                        float grad_scale, void* workspace, int64_t workspace_bytes, float* lse,
                                        float* per_row, float* loss, float* dq, float* dc, tt_stream_t stream) {
  return retrieval_fwd_bwd(0, q, c, nq, nc, dim, diag_offset, inv_temperature, sample_weight, cand_prob, cand_ids, hard_thr,
                           grad_scale, workspace, workspace_bytes, lse, per_row, loss, dq, dc, stream);
}

// The same two passes with every matrix product on bf16 MFMA through the hi/mid/lo split of the f32 operands
// (f32-emulated: 6 products for the logits, 3 for the gradient sums; f32 accumulation, f32 softmax).  dim in {128, 256}.
extern "C" int tt_retrieval_fwd_bwd_bf16x3_f32(const float* q, const float* c, int64_t nq, int64_t nc, int32_t dim,
                                               int64_t diag_offset, float inv_temperature, const float* sample_weight,
                                               const float* cand_prob, const int64_t* cand_ids, const float* hard_thr,
                                               float grad_scale, void* workspace, int64_t workspace_bytes, float* lse,
                                               float* per_row, float* loss, float* dq, float* dc, tt_stream_t stream) {
  if (dim != 128 && dim != 256) return tt::fail(TT_ERR_UNSUPPORTED, "tt_retrieval_fwd_bwd_bf16x3_f32: dim %d not in {128,256}", dim);
  return retrieval_fwd_bwd(1, q, c, nq, nc, dim, diag_offset, inv_temperature, sample_weight, cand_prob, cand_ids, hard_thr,
                           grad_scale, workspace, workspace_bytes, lse, per_row, loss, dq, dc, stream);
}

// Retrieval metric support (configs/data_config.yaml:71 top_k_eval): rank of each query's true
// candidate among ALL nc candidates = number of other candidates with a strictly larger logit.  One fused pass over
// the [nq, nc] logits (never materialised); Recall@K / NDCG@K follow from rank < K on the host side.
static int retrieval_rank(int prec, const float* q, const float* c, int64_t nq, int64_t nc, int32_t dim,
                          float inv_temperature, const float* cand_prob, const int64_t* pos_index,
                          void* workspace, int64_t workspace_bytes, int32_t* rank, tt_stream_t stream_) {
  // (not check_common: a rank pass has no diagonal, so nq > nc is fine, and it needs only the front of the workspace)
  int rc;
  TT_REQUIRE(q && c && workspace && pos_index && rank, "tt_retrieval_rank_f32: null pointer");
  TT_REQUIRE(nq > 0 && nc > 0, "tt_retrieval_rank_f32: nq and nc must be positive");
  TT_REQUIRE(dim == 32 || dim == 64 || dim == 128 || dim == 256, "tt_retrieval_rank_f32: dim %d not in {32,64,128,256}", dim);
  TT_REQUIRE(tt::aligned16(q) && tt::aligned16(c), "tt_retrieval_rank_f32: q/c must be 16-byte aligned");
  TT_REQUIRE((reinterpret_cast<uintptr_t>(workspace) & 255u) == 0, "tt_retrieval_rank_f32: workspace must be 256-byte aligned");
  const WsLayout w = ws_layout(nq, nc, dim);
  if (workspace_bytes < w.off_pl)
    return tt::fail(TT_ERR_WORKSPACE, "tt_retrieval_rank_f32: workspace %lld < %lld bytes", (long long)workspace_bytes,
                    (long long)w.off_pl);
  hipStream_t stream = tt::as_stream(stream_);
  char* ws = static_cast<char*>(workspace);
  float* bias = reinterpret_cast<float*>(ws + w.off_bias);
  float* thr = reinterpret_cast<float*>(ws + w.off_aq);
  int32_t* part_cnt = reinterpret_cast<int32_t*>(ws + w.off_pm);
  const float c1 = kLog2e * inv_temperature;
  if (cand_prob != nullptr) {
    hipLaunchKernelGGL(prob_bias_kernel, dim3((unsigned)((nc + 255) / 256)), dim3(256), 0, stream, cand_prob, bias, nc);
    if ((rc = tt::check_launch("prob_bias")) != TT_OK) return rc;
  }
  const float* biasp = cand_prob != nullptr ? bias : nullptr;
  hipLaunchKernelGGL(pos_logit_kernel, dim3((unsigned)((nq + 7) / 8)), dim3(256), 0, stream, reinterpret_cast<const f32x4*>(q),
                     reinterpret_cast<const f32x4*>(c), pos_index, biasp, nq, nc, dim / 4, c1, thr, (int64_t)0);
  if ((rc = tt::check_launch("pos_logit")) != TT_OK) return rc;
  ScoreArgs a{};
  a.R = q; a.K = c; a.n_r = nq; a.n_c = nc; a.diag = 0;
  a.c1 = c1;
  a.a_r = thr; a.a_c = biasp;
  a.pos_idx = pos_index;
  a.nsplit = w.ns_q;
  a.c_per_split = align_up((nc + a.nsplit - 1) / a.nsplit, 32);
  a.part_cnt = part_cnt;
  rc = prec == 1 ? dispatch_score_bx3<MODE_RANK>(dim, a, false, stream) : dispatch_score<MODE_RANK>(dim, a, false, stream);
  if (rc != TT_OK) return rc;
  hipLaunchKernelGGL(rank_combine_kernel, dim3((unsigned)((nq + 255) / 256)), dim3(256), 0, stream, part_cnt, nq, a.nsplit, rank);
  return tt::check_launch("rank_combine");
}

extern "C" int tt_retrieval_rank_f32(const float* q, const float* c, int64_t nq, int64_t nc, int32_t dim,
                                     float inv_temperature, const float* cand_prob, const int64_t* pos_index,
                                     void* workspace, int64_t workspace_bytes, int32_t* rank, tt_stream_t stream) {
  return retrieval_rank(0, q, c, nq, nc, dim, inv_temperature, cand_prob, pos_index, workspace, workspace_bytes, rank, stream);
}

// Top-K over a 10M-100M-row corpus is pure GEMM1 as well: the same rank pass with the 6-product bf16 split.  The positive's
// threshold logit stays an exact f32 dot product (pos_logit_kernel); a competitor within ~2^-22 relative of it may fall on
// either side, exactly as between two exact-f32 evaluation orders (the tests pin ranks between f64 +-1e-5 bounds).
extern "C" int tt_retrieval_rank_bf16x3_f32(const float* q, const float* c, int64_t nq, int64_t nc, int32_t dim,
                                            float inv_temperature, const float* cand_prob, const int64_t* pos_index,
                                            void* workspace, int64_t workspace_bytes, int32_t* rank, tt_stream_t stream) {
  if (dim != 128 && dim != 256) return tt::fail(TT_ERR_UNSUPPORTED, "tt_retrieval_rank_bf16x3_f32: dim %d not in {128,256}", dim);
  return retrieval_rank(1, q, c, nq, nc, dim, inv_temperature, cand_prob, pos_index, workspace, workspace_bytes, rank, stream);
}

// In-batch rank (tfrs.tasks.Retrieval(batch_metrics=...): Keras top-k categorical accuracy over the IN-BATCH score matrix):
// rank[i] = number of in-batch candidates j != i + diag_offset whose logit - after temperature, sampling-probability
// correction and accidental-hit removal, exactly the scores the loss sees - is strictly above the positive's.  Top-k
// accuracy = mean(rank < k): no [nq, nc] score matrix is materialised, it is the metric pass above run on the batch.
extern "C" int tt_retrieval_batch_rank_f32(const float* q, const float* c, int64_t nq, int64_t nc, int32_t dim, int64_t diag_offset,
                                           float inv_temperature, const float* cand_prob, const int64_t* cand_ids,
                                           void* workspace, int64_t workspace_bytes, int32_t* rank, tt_stream_t stream_) {
  int rc;
  TT_REQUIRE(q && c && workspace && rank, "tt_retrieval_batch_rank_f32: null pointer");
  TT_REQUIRE(nq > 0 && nc > 0, "tt_retrieval_batch_rank_f32: nq and nc must be positive");
  TT_REQUIRE(diag_offset >= 0 && nq + diag_offset <= nc, "tt_retrieval_batch_rank_f32: need 0 <= diag_offset and nq + diag_offset <= nc");
  TT_REQUIRE(dim == 32 || dim == 64 || dim == 128 || dim == 256, "tt_retrieval_batch_rank_f32: dim %d not in {32,64,128,256}", dim);
  TT_REQUIRE(tt::aligned16(q) && tt::aligned16(c), "tt_retrieval_batch_rank_f32: q/c must be 16-byte aligned");
  TT_REQUIRE((reinterpret_cast<uintptr_t>(workspace) & 255u) == 0, "tt_retrieval_batch_rank_f32: workspace must be 256-byte aligned");
  const WsLayout w = ws_layout(nq, nc, dim);
  if (workspace_bytes < w.off_pl)
    return tt::fail(TT_ERR_WORKSPACE, "tt_retrieval_batch_rank_f32: workspace %lld < %lld bytes", (long long)workspace_bytes,
                    (long long)w.off_pl);
  hipStream_t stream = tt::as_stream(stream_);
  char* ws = static_cast<char*>(workspace);
  float* bias = reinterpret_cast<float*>(ws + w.off_bias);
  float* thr = reinterpret_cast<float*>(ws + w.off_aq);
  int32_t* part_cnt = reinterpret_cast<int32_t*>(ws + w.off_pm);
  const float c1 = kLog2e * inv_temperature;
  if (cand_prob != nullptr) {
    hipLaunchKernelGGL(prob_bias_kernel, dim3((unsigned)((nc + 255) / 256)), dim3(256), 0, stream, cand_prob, bias, nc);
    if ((rc = tt::check_launch("prob_bias")) != TT_OK) return rc;
  }
  const float* biasp = cand_prob != nullptr ? bias : nullptr;
  hipLaunchKernelGGL(pos_logit_kernel, dim3((unsigned)((nq + 7) / 8)), dim3(256), 0, stream, reinterpret_cast<const f32x4*>(q),
                     reinterpret_cast<const f32x4*>(c), static_cast<const int64_t*>(nullptr), biasp, nq, nc, dim / 4, c1, thr, diag_offset);
  if ((rc = tt::check_launch("pos_logit")) != TT_OK) return rc;
  ScoreArgs a{};
  a.R = q; a.K = c; a.n_r = nq; a.n_c = nc; a.diag = diag_offset;
  a.c1 = c1;
  a.a_r = thr; a.a_c = biasp;
  a.id_r = cand_ids != nullptr ? cand_ids + diag_offset : nullptr;
  a.id_c = cand_ids;
  a.nsplit = w.ns_q;
  a.c_per_split = align_up((nc + a.nsplit - 1) / a.nsplit, 32);
  a.part_cnt = part_cnt;
  if ((rc = dispatch_score<MODE_RANK>(dim, a, cand_ids != nullptr, stream)) != TT_OK) return rc;
  hipLaunchKernelGGL(rank_combine_kernel, dim3((unsigned)((nq + 255) / 256)), dim3(256), 0, stream, part_cnt, nq, a.nsplit, rank);
  return tt::check_launch("rank_combine");
}

// Hard-negative thresholds (tfrs.tasks.Retrieval(num_hard_negatives=k)): thr[i] separates the k highest-scoring negatives
// of query i (after temperature, sampling-probability correction and accidental-hit removal) from the rest; pass it as
// `hard_thr` to tt_retrieval_{fwd,bwd,fwd_bwd}_f32.  scratch: nq*nc floats (the only place logits are materialised).
extern "C" int tt_retrieval_hard_negative_thresholds_f32(const float* q, const float* c, int64_t nq, int64_t nc, int32_t dim,
                                                         int64_t diag_offset, float inv_temperature, const float* cand_prob,
                                                         const int64_t* cand_ids, int32_t num_hard_negatives, void* workspace,
                                                         int64_t workspace_bytes, float* scratch, int64_t scratch_bytes,
                                                         float* thr, tt_stream_t stream_) {
  int rc = check_common("tt_retrieval_hard_negative_thresholds_f32", q, c, nq, nc, dim, diag_offset, workspace, workspace_bytes);
  if (rc != TT_OK) return rc;
  TT_REQUIRE(num_hard_negatives >= 1, "tt_retrieval_hard_negative_thresholds_f32: num_hard_negatives must be >= 1");
  TT_REQUIRE(scratch && thr, "tt_retrieval_hard_negative_thresholds_f32: null scratch/thr");
  if (scratch_bytes < nq * nc * 4)
    return tt::fail(TT_ERR_WORKSPACE, "tt_retrieval_hard_negative_thresholds_f32: scratch %lld < %lld bytes",
                    (long long)scratch_bytes, (long long)(nq * nc * 4));
  hipStream_t stream = tt::as_stream(stream_);
  const WsLayout w = ws_layout(nq, nc, dim);
  float* bias = reinterpret_cast<float*>(static_cast<char*>(workspace) + w.off_bias);
  if (cand_prob != nullptr) {
    hipLaunchKernelGGL(prob_bias_kernel, dim3((unsigned)((nc + 255) / 256)), dim3(256), 0, stream, cand_prob, bias, nc);
    if ((rc = tt::check_launch("prob_bias")) != TT_OK) return rc;
  }
  if ((rc = tt::gemm_nt(q, c, scratch, nq, nc, dim, stream)) != TT_OK) return rc;
  tt::launch("score_aux", hardneg_select_kernel, dim3((unsigned)nq), dim3(256), 0, stream, scratch, nc, kLog2e * inv_temperature,
                     cand_prob != nullptr ? bias : nullptr, cand_ids, diag_offset, num_hard_negatives, thr);
  return tt::check_launch("hardneg_select");
}
