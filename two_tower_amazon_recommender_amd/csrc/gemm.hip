// K3 — MLP tower layers: y = act(x@w+b), dx = dz@w^T (* relu mask),
// dw = x^T@dz (split-K slabs), db = colsum(dz).  f32-input MFMA (v_mfma_f32_32x32x2_f32, exact f32
// products, peak 157.3 TF).  One tiled kernel, three operand-orientation instantiations:
//
//   fwd  NN : A = x  [m][k] k-contiguous      B = w  [k][n] n-contiguous
//   dx   NT : A = dz [m][k=n_out] k-contig.   B = w  [n=k_in][k=n_out] k-contiguous
//   dw   TN : A = x  [k=batch][m=k_in]        B = dz [k=batch][n] (both "m-contiguous"), split over batch
//
// Measured (MI355X, m 8192, k/n 128/256): fwd 11.5 us per call back to back vs 17 us for torch.mm (hipBLASLt f32).
// Block tile 64x64x32, 4 waves (2x2), one 32x32 accumulator per wave: the tower GEMMs are skinny
// (n = 128..512) so small tiles keep >= 256 workgroups in flight.  Tiles go global -> registers ->
// LDS through a register ring that keeps PF = 4 k-tiles of loads in flight (LDS double-buffered, one barrier
// per k-tile).  MFMA k order inside a step is permuted
// (k = 8g + 4*lanehalf + s) so a k-contiguous operand is read with one ds_read_b128 per 4 MFMAs
// from rows padded to BK+4 floats (conflict-free); an m-contiguous operand is read with ds_read_b32.
// The embedding lookup of a tower's first layer is fused into the A loader (fwd and dW): see GemmArgs::a_ids.
#include "common.h"
#include "dense_update_body.h"
#include <cstdlib>

namespace {

using tt::f32x4;
using tt::f32x16;

#ifndef TT_GEMM_BK
#define TT_GEMM_BK 32                     // k-tile depth (A/B hook: 64 halves the barrier rounds, doubles LDS and the register ring)
#endif
constexpr int BM = 64, BN = 64, BK = TT_GEMM_BK;
// k-tiles of global loads in flight per thread.  r02 sweep of the four cfg3 tower launches (us): 1: 96.3, 2: 95.3, 3: 98.8,
// 4 (3 for the forward orientation, the r01 setting): 98.2 - with four resident workgroups per CU the other waves cover a
// tile's latency; the smaller ring leaves registers.
#ifndef TT_GEMM_PF
#define TT_GEMM_PF 2
#endif
#ifndef TT_GEMM_WIDE_STORE
#define TT_GEMM_WIDE_STORE 1  // the output tile goes through LDS and leaves as 16-byte stores (0: 4-byte stores from the accumulators)
#endif
#ifndef TT_GEMM_PF_FWD
#define TT_GEMM_PF_FWD 2
#endif
constexpr int PF_MAX = TT_GEMM_PF;       // k-tiles of global loads kept in flight per thread (register ring); the mixed-orientation
                                         // forward kernel runs with 3 (it spills 22-48 VGPRs at 4 under the 4-waves/SIMD bound)
constexpr int NST = BM * BK / 4 / 256;   // staged float4 per thread and operand (= 2)
constexpr int LS_KC = BK + 4;            // [row][k] stride
constexpr int LS_MC = BM + 4;            // [k][row] stride
constexpr int TILE_F = (BM * LS_KC > BK * LS_MC) ? BM * LS_KC : BK * LS_MC;   // floats per operand tile
constexpr int kMaxGatherK = 1024;        // fused-gather dW tiles: ids of one split's batch rows staged in LDS (2 KB up to 256 rows
                                         // per split = batch <= 8192: 4 workgroups per CU still fit; 8 KB up to 1024)

#ifdef TT_GEMM_STAMPS
__device__ unsigned long long g_stamps[8192 * 4];
#define STAMP(i) do { if (threadIdx.x == 0) g_stamps[(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z)) % 8192 * 4 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define STAMP(i) do {} while (0)
#endif

struct GemmArgs {
  const float* A;
  const float* B;
  float* C;
  int64_t M;            // rows of C
  int64_t N;            // cols of C
  int64_t K;            // reduction length
  int64_t lda, ldb, ldc;
  const float* bias;    // [N] or null
  int relu;
  const float* mask_src;   // [M][ldc] or null: C *= (mask_src > 0)
  const uint32_t* mask_bits;   // [M][N/32] or null: the same mask as sign bits (N % 32 == 0); takes precedence over mask_src
  uint32_t* relu_bits;     // fwd, [M][N/32] or null: written beside C, bit = (C > 0)
  float mask_scale;        // multiplies the kept entries of the masked epilogue (1/(1-p) under dropout, else 1)
  uint32_t drop_p24;       // fwd dropout: element dropped iff 24-bit hash < drop_p24 (0 = no dropout)
  float drop_scale;        // 1/(1-p)
  uint64_t drop_key;       // counter-based stream key
  uint64_t drop_offset;    // counter of element (0,0); element (m,n) uses drop_offset + m*N + n
  int64_t k_per_split;     // multiple of BK
  int64_t slab_stride;     // C offset per blockIdx.z
  float* db_slabs;         // TN only: [splits][N] column sums of B
  // K1 fused into the loader (north_star: "LDS-staged embedding rows"): logical A row r is row a_ids[r] of the TABLE
  // A points to (+ row a_ids2[r] of table A2, the hashed-category feature) — the [batch, dim] tower input is never
  // written to or re-read from HBM.  Ids outside [0, a_rows) give a zero row (-1 = padding: silent; else *oob_flag = 1).
  const int64_t* a_ids;
  int64_t a_rows;
  const float* A2;
  const int64_t* a_ids2;
  int64_t a_rows2;
  int32_t* oob_flag;
  // row-range id lists for the optimizer launch of the same step (tt_id_buckets; forward GEMM of layer 0 with the fused lookup):
  // the by == 0 tile of every row block appends its 64 rows' ids, as csrc/tower.hip's fused kernel does
  uint32_t* bk_counts; uint64_t* bk_pairs;
  uint32_t bk_width, bk_magic, bk_groups, bk_cap, bk_gen;
};

__device__ __forceinline__ f32x4 ldg4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }

// ---- operand staging: global -> registers (ring) -> LDS ----
//   KC  operand is k-contiguous   [row][k]: thread f = tid + 256 j moves float4 (row = f / 8, k = 4 (f % 8)); LDS [row][k]
//   MC  operand is row-contiguous [k][row]: thread f moves float4 (k = f / 16, rows 4 (f % 16) ..);           LDS [k][row]
// (A transposing store that gives MC operands the [row][k] layout too — ds_read_b128 fragments everywhere — was built
// and measured in r02: 118 vs 114 us for the four tower launches.  The kernel is not LDS-read-bound; the 16-row global
// load pattern the transpose needs costs more than the wider reads save.)
template <bool KC>
__device__ __forceinline__ void load_operand(f32x4 (&st)[NST], const float* __restrict__ base, int64_t ld, int64_t row0,
                                             int64_t nrows, int64_t k0, int64_t kend, int tid) {
#pragma unroll
  for (int j = 0; j < NST; ++j) {
    const int f = tid + 256 * j;
    if constexpr (KC) {
      const int row = f / (BK / 4), k4 = f % (BK / 4);
      const int64_t r = row0 + row, k = k0 + 4 * k4;
      st[j] = (r < nrows && k < kend) ? ldg4(base + r * ld + k) : f32x4{0.f, 0.f, 0.f, 0.f};
    } else {
      const int kk = f >> 4, m4 = f & 15;
      const int64_t k = k0 + kk, r = row0 + 4 * m4;
      st[j] = (k < kend && r < nrows) ? ldg4(base + k * ld + r) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
  }
}

// KC gather (forward layer 0): the thread's two tile rows are table rows; off[j] = element offset of the row or -1
__device__ __forceinline__ void load_rows_kc(f32x4 (&st)[NST], const float* __restrict__ t1, const int64_t (&off1)[NST],
                                             const float* __restrict__ t2, const int64_t (&off2)[NST], int64_t k0, int64_t kend,
                                             int tid) {
#pragma unroll
  for (int j = 0; j < NST; ++j) {
    const int64_t k = k0 + 4 * ((tid + 256 * j) % (BK / 4));
    f32x4 v = (off1[j] >= 0 && k < kend) ? ldg4(t1 + off1[j] + k) : f32x4{0.f, 0.f, 0.f, 0.f};
    if (t2 != nullptr) {
      const f32x4 c = (off2[j] >= 0 && k < kend) ? ldg4(t2 + off2[j] + k) : f32x4{0.f, 0.f, 0.f, 0.f};
      v = v + c;                                            // one f32 add per element (the oracle's row + category row)
    }
    st[j] = v;
  }
}

// MC gather (dW of layer 0: A = x^T, x[b][:] = table row ids[b]): the tile's k rows are batch rows; their table rows
// come from the ids staged in LDS (ids1 / ids2: int32 row index, -1 = none)
__device__ __forceinline__ void load_rows_mc(f32x4 (&st)[NST], const float* __restrict__ t1, const int32_t* ids1,
                                             const float* __restrict__ t2, const int32_t* ids2, int64_t ld, int64_t row0,
                                             int64_t nrows, int kloc0 /* k0 - kbeg */, int klen, int tid) {
#pragma unroll
  for (int j = 0; j < NST; ++j) {
    const int f = tid + 256 * j;
    const int kl = kloc0 + (f >> 4);
    const int64_t r = row0 + 4 * (f & 15);
    f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
    if (kl < klen && r < nrows) {
      const int i1 = ids1[kl];
      if (i1 >= 0) v = ldg4(t1 + (int64_t)i1 * ld + r);
      if (t2 != nullptr) {
        const int i2 = ids2[kl];
        if (i2 >= 0) v = v + ldg4(t2 + (int64_t)i2 * ld + r);
      }
    }
    st[j] = v;
  }
}

template <bool KC>
__device__ __forceinline__ void store_operand(const f32x4 (&st)[NST], float* T, int tid) {
#pragma unroll
  for (int j = 0; j < NST; ++j) {
    const int f = tid + 256 * j;
    if constexpr (KC) {
      const int row = f / (BK / 4), k4 = f % (BK / 4);
      *reinterpret_cast<f32x4*>(T + row * LS_KC + 4 * k4) = st[j];
    } else {
      const int kk = f >> 4, m4 = f & 15;
      *reinterpret_cast<f32x4*>(T + kk * LS_MC + 4 * m4) = st[j];
    }
  }
}

// the 4 operand values of k-group g (k = 8g + 4h + s, s = 0..3) for tile row `row`
template <bool KC>
__device__ __forceinline__ f32x4 read_operand(const float* T, int row, int g, int h) {
  if constexpr (KC) {
    return *reinterpret_cast<const f32x4*>(T + row * LS_KC + 8 * g + 4 * h);
  } else {
    const float* p = T + (8 * g + 4 * h) * LS_MC + row;
    return f32x4{p[0], p[LS_MC], p[2 * LS_MC], p[3 * LS_MC]};
  }
}

// table row of logical row r for the fused gather: element offset id * ld, or -1 (zero row); flags ids out of range
__device__ __forceinline__ int64_t gather_row(const int64_t* __restrict__ ids, int64_t r, int64_t n, int64_t rows, int64_t ld,
                                              int32_t* oob_flag, bool flag_it) {
  if (r >= n) return -1;
  const int64_t id = ids[r];
  if (id >= 0 && id < rows) return id * ld;
  if (flag_it && oob_flag != nullptr && id != -1) atomicOr(oob_flag, 1);
  return -1;
}

// up to two independent problems of identical shape per launch (the user and the item tower's layer l):
// blockIdx.z = problem * splits + split
struct GemmBatch {
  GemmArgs a[2];
  int splits;
};

// ---- dW tiles WITHOUT LDS operand tiles (r04) ------------------------------------------------------------------------------
// dW[k][n] = sum_b x[b][k] * dz[b][n] over the batch rows of one split.  Both operands are ROW-contiguous in the reduction index
// (x is [b][k], dz is [b][n]), which is exactly v_mfma_f32_32x32x2_f32's operand layout: lane l supplies A[row l%32][k-slot l/32]
// and B[k-slot l/32][col l%32] - one batch row per lane half, 32 consecutive k (n) across the lanes.  So a wave feeds its MFMAs
// straight from global memory: an 8-byte load per lane and operand (half-wave = 256 contiguous bytes of one batch row) gives a
// lane the values of output rows 2 ln, 2 ln + 1 (columns 2 ln, 2 ln + 1): four MFMAs per two batch rows cover a 64 x 64 tile as
// 2 x 2 INTERLEAVED 32 x 32 blocks (block (i, j) = rows 2r + i, columns 2c + j).  No staging, no transpose, no workgroup barrier
// in the reduction loop; the four waves of the workgroup split the split's batch rows four ways and meet once, through LDS, in a
// fixed order ((w0 + w2) + (w1 + w3)).  The LDS-tile form this replaces ran a chain of load -> ds_write -> barrier -> 16 MFMAs per
// 32 batch rows: r04 stamps, layer 1 of cfg3: the dW workgroups' MFMA loops took 11.8-17.6 us for 3.8 us of MFMA time per wave,
// and the dx workgroups of the same launch finished their second tile only when those had retired.
// Needs M % 64 == 0 and N % 64 == 0 (every tower layer of the BASELINE configs but cfg1's 32 x 32); TT_DW_DIRECT=0 keeps the tiles.
#ifndef TT_DW_DIRECT
#define TT_DW_DIRECT 1
#endif
#ifndef TT_DW_PF_DEEP
#define TT_DW_PF_DEEP 8          // ring depth where no lookup is fused into the rows.  16 fits the registers but ties (towers 87.6-87.8
#endif                           // vs 87.8-88.1 us, r04) and would give dense and looked-up rows different eligibility (splits of 128 vs 64 rows)
#ifndef TT_DX_PRIO
#define TT_DX_PRIO 2
#endif
#ifndef TT_DW_PF
#define TT_DW_PF 8               // iterations (2 batch rows each) of operand loads in flight per wave
#endif
typedef float f32x2 __attribute__((ext_vector_type(2)));
// TWO: a second table's row is summed into every gathered row (the hashed category feature); its ring is paid for with a
// shallower one (4 rounds in flight instead of 8: the lookup kernels have no registers to spare at 4 workgroups per CU).
template <int GK, bool TWO, int PFD = TT_DW_PF>
__device__ __forceinline__ void dw_tile_direct(const GemmArgs& p, const int zsplit, const int bx, const int by, float* smem,
                                               int32_t* gids) {
  constexpr bool GATHER = GK != 0;
  static_assert(GATHER || !TWO, "a second table only with the fused lookup");
  constexpr int PF = TWO ? PFD / 2 : PFD;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int h = lane >> 5, ln = lane & 31;
  const int64_t m0 = (int64_t)bx * BM;        // first output row (k of the layer)
  const int64_t n0 = (int64_t)by * BN;        // first output column
  const int64_t kbeg = (int64_t)zsplit * p.k_per_split;
  int64_t kend = kbeg + p.k_per_split;
  if (kend > p.K) kend = p.K;
  const int klen = kend > kbeg ? (int)(kend - kbeg) : 0;
  if constexpr (GATHER) {
    for (int i = tid; i < klen; i += 256) {
      const int64_t i1 = p.a_ids[kbeg + i];
      gids[i] = (i1 >= 0 && i1 < p.a_rows) ? (int32_t)i1 : -1;
      if constexpr (TWO) {
        const int64_t i2 = p.a_ids2[kbeg + i];
        gids[GK + i] = (i2 >= 0 && i2 < p.a_rows2) ? (int32_t)i2 : -1;
      }
    }
    __syncthreads();
  }
  STAMP(0);
  // this wave's batch rows: a quarter of the split, in pairs (lane half h takes row 2 it + h of the quarter).  The direct form
  // only takes splits whose length is a multiple of 64 rows (the caller checks): every wave then has a whole number of ring
  // rounds, every row exists - no clamping, no zero masks, and the loop body is ONE basic block (with a per-slot `if (it < nit)`
  // hipcc waits `vmcnt(0)` at the loop header, i.e. for the loads it issued a moment ago; "load, then zero if past the end"
  // puts a select - and a full wait - right behind every load: both seen in the r04 ISA).
  const int q = klen >> 2;                                     // rows per wave
  const int nit = q >> 1;                                      // a multiple of PF
  const uint32_t r0h = (uint32_t)(wave * q + h);               // this lane half's first row, relative to kbeg
  const uint32_t lda = (uint32_t)p.lda, ldb = (uint32_t)p.ldb;
  const float* Bb = p.B + kbeg * p.ldb + n0;                   // (uniform bases: the lane's part is a 32-bit element offset)
  const float* Ab = GATHER ? p.A + m0 : p.A + kbeg * p.lda + m0;
  f32x2 ra[PF], rb[PF], ra2[TWO ? PF : 1];
  uint32_t a1m = 0u, a2m = 0u;                                 // GATHER: bit u = ring slot u holds a real table row (of table 1 / 2)
  auto load_raw = [&](int u, int it) {
    const uint32_t rl = r0h + 2u * (uint32_t)it;
    rb[u] = *reinterpret_cast<const f32x2*>(Bb + (rl * ldb + 2u * (uint32_t)ln));
    if constexpr (GATHER) {
      const int i1 = gids[rl];
      const uint32_t bit = 1u << u;
      a1m = i1 >= 0 ? (a1m | bit) : (a1m & ~bit);
      ra[u] = *reinterpret_cast<const f32x2*>(Ab + ((uint64_t)(uint32_t)(i1 >= 0 ? i1 : 0) * lda + 2u * (uint32_t)ln));
      if constexpr (TWO) {
        const int i2 = gids[GK + rl];
        a2m = i2 >= 0 ? (a2m | bit) : (a2m & ~bit);
        ra2[u] = *reinterpret_cast<const f32x2*>(p.A2 + m0 + ((uint64_t)(uint32_t)(i2 >= 0 ? i2 : 0) * lda + 2u * (uint32_t)ln));
      }
    } else {
      ra[u] = *reinterpret_cast<const f32x2*>(Ab + (rl * lda + 2u * (uint32_t)ln));
    }
  };
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  float cs0 = 0.f, cs1 = 0.f;
#pragma unroll
  for (int u = 0; u < PF; ++u) load_raw(u, u);
  for (int t0 = 0; t0 < nit; t0 += PF) {
    const bool more = t0 + PF < nit;                           // (uniform) the last round re-reads its own rows: never consumed
#pragma unroll
    for (int u = 0; u < PF; ++u) {
      f32x2 a = ra[u];
      const f32x2 b = rb[u];
      if constexpr (GATHER) {
        if (!((a1m >> u) & 1u)) a = f32x2{0.f, 0.f};          // a padding / out-of-range id: zero row
        if constexpr (TWO) {
          if ((a2m >> u) & 1u) a = a + ra2[u];                 // one f32 add per element (the oracle's row + category row)
        }
      }
      load_raw(u, more ? t0 + u + PF : t0 + u);
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[0], b[0], acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[0], b[1], acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[1], b[0], acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[1], b[1], acc[1][1], 0, 0, 0);
      cs0 = __fadd_rn(cs0, b[0]);                              // db: column sums of dz ride along (used by the bx == 0 tiles)
      cs1 = __fadd_rn(cs1, b[1]);
    }
  }
  STAMP(1);
  STAMP(2);
  // the two lane halves hold different batch rows of the same columns
  cs0 = __fadd_rn(cs0, __shfl_xor(cs0, 32));
  cs1 = __fadd_rn(cs1, __shfl_xor(cs1, 32));
  // ---- the four waves' partial tiles meet in LDS: (w0 + w2) + (w1 + w3), 66 values per lane ([reg][lane]: conflict-free) ----
  constexpr int SLOT = 66 * 64;
  auto put = [&](float* S) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) S[((i * 2 + j) * 16 + e) * 64 + lane] = acc[i][j][e];
    S[64 * 64 + lane] = cs0;
    S[65 * 64 + lane] = cs1;
  };
  auto add = [&](const float* S) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][j][e] = __fadd_rn(acc[i][j][e], S[((i * 2 + j) * 16 + e) * 64 + lane]);
    cs0 = __fadd_rn(cs0, S[64 * 64 + lane]);
    cs1 = __fadd_rn(cs1, S[65 * 64 + lane]);
  };
  if (wave >= 2) put(smem + (wave - 2) * SLOT);
  __syncthreads();
  if (wave < 2) add(smem + wave * SLOT);
  __syncthreads();
  if (wave == 1) put(smem);
  __syncthreads();
  if (wave == 0) add(smem);
  __syncthreads();                                             // (everyone is done reading the slots)
  // ---- out: wave 0's registers -> [64][64 + 4] floats in LDS -> 16 bytes per lane, all four waves storing ----
  constexpr int LSO = BN + 4;
  if (wave == 0) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = 2 * tt::acc_row(e, h) + i;
        *reinterpret_cast<f32x2*>(smem + row * LSO + 2 * ln) = f32x2{acc[i][0][e], acc[i][1][e]};
      }
  }
  __syncthreads();
  float* C = p.C + (int64_t)zsplit * p.slab_stride;
#pragma unroll
  for (int j = 0; j < BM * BN / 4 / 256; ++j) {
    const int f = tid + 256 * j;
    const int row = f / (BN / 4), c4 = f % (BN / 4);
    *reinterpret_cast<f32x4*>(C + (m0 + row) * p.ldc + n0 + 4 * c4) = *reinterpret_cast<const f32x4*>(smem + row * LSO + 4 * c4);
  }
  if (bx == 0 && wave == 0 && lane < 32)
    *reinterpret_cast<f32x2*>(p.db_slabs + (int64_t)zsplit * p.N + n0 + 2 * ln) = f32x2{cs0, cs1};
#ifdef TT_GEMM_STAMPS
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  STAMP(3);
#endif
}

// one 64x64 output tile (bx, by) of problem p, k-range of split zsplit.  GATHER: A is an embedding table read through
// p.a_ids (KC: the forward GEMM of layer 0; MC: its dW GEMM, ids staged in `gids`).
//
// r02 experiments on this structure, both measured on the four cfg3 tower launches and dropped:
//  * every operand staged [row][k] (transposing ds_write_b64 for the row-contiguous ones, all fragments ds_read_b128):
//    118 vs 114 us — the kernel is not LDS-read-bound, and the 16-rows-per-instruction global load pattern costs more;
//  * a persistent workgroup walking several tiles as one flat k-tile sequence (register ring across tile boundaries,
//    epilogue stores under the next tile's MFMAs), 1-4 workgroups per CU: 125-196 us vs 114 — the cursor state costs
//    ~50 VGPRs (149-176 -> 3 waves/SIMD instead of 4) and these 4-8 k-tile GEMMs want occupancy (memory-level
//    parallelism across many small workgroups) more than an in-workgroup pipeline.
template <bool A_KC, bool B_KC, bool COLSUM, int GK, bool DROP = false>
__device__ __forceinline__ void gemm_tile(const GemmArgs& p, const int zsplit, const int bx, const int by, float* smem,
                                          int32_t* gids) {
#if TT_DW_DIRECT
  if constexpr (!A_KC && !B_KC && COLSUM) {
    // (workgroup-uniform) the direct form takes whole 64 x 64 tiles and 8-byte row pieces
    // (rows straight from a dense activation - GK == 0 -: twice the ring depth, for splits that are whole rounds of it - 4 MFMAs
    // per ring slot are 0.12 us of matrix-pipe time against a 1-2 us load round trip; only ONE depth per kernel: with both
    // instantiated beside the dx tiles the kernel needs more than the 128 VGPRs that 4 workgroups per CU leave)
    constexpr int RND = 8 * (GK == 0 ? TT_DW_PF_DEEP : TT_DW_PF);       // batch rows of one ring round of the four waves
    if ((p.M & 63) == 0 && (p.N & 63) == 0 && ((p.lda | p.ldb | p.ldc) & 3) == 0 && p.db_slabs != nullptr &&
        ((p.K | p.k_per_split) & (RND - 1)) == 0 && p.k_per_split * (p.lda > p.ldb ? p.lda : p.ldb) < ((int64_t)1 << 31)) {
      if constexpr (GK != 0) {
        if (p.A2 != nullptr) { dw_tile_direct<GK, true>(p, zsplit, bx, by, smem, gids); return; }
      }
      dw_tile_direct<GK, false, (GK == 0 ? TT_DW_PF_DEEP : TT_DW_PF)>(p, zsplit, bx, by, smem, gids);
      return;
    }
  }
#endif
  constexpr int PF = (A_KC && !B_KC) ? TT_GEMM_PF_FWD : PF_MAX;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int h = lane >> 5, ln = lane & 31;
  const int wm = wave >> 1, wn = wave & 1;
  const int64_t m0 = (int64_t)bx * BM;
  const int64_t n0 = (int64_t)by * BN;
  const int64_t kbeg = (int64_t)zsplit * p.k_per_split;
  int64_t kend = kbeg + p.k_per_split;
  if (kend > p.K) kend = p.K;
  const int nk = kend > kbeg ? (int)((kend - kbeg + BK - 1) / BK) : 0;
  constexpr bool GATHER = GK != 0;    // GK: capacity of the LDS id stage (MC form); any non-zero value selects the KC form

  // ---- fused gather: resolve the table rows once per tile ----
  int64_t off1[NST] = {}, off2[NST] = {};
  const int klen = (int)(kend - kbeg);
  if constexpr (GATHER && A_KC) {
#pragma unroll
    for (int j = 0; j < NST; ++j) {
      const int64_t r = m0 + (tid + 256 * j) / (BK / 4);
      const bool first = by == 0 && ((tid + 256 * j) % (BK / 4)) == 0;     // one flagging thread per row
      off1[j] = gather_row(p.a_ids, r, p.M, p.a_rows, p.lda, p.oob_flag, first);
      off2[j] = p.A2 != nullptr ? gather_row(p.a_ids2, r, p.M, p.a_rows2, p.lda, p.oob_flag, first) : -1;
    }
  }
  if constexpr (GATHER && !A_KC) {
    for (int i = tid; i < klen; i += 256) {
      const int64_t i1 = p.a_ids[kbeg + i];
      gids[i] = (i1 >= 0 && i1 < p.a_rows) ? (int32_t)i1 : -1;
      if (p.A2 != nullptr) {
        const int64_t i2 = p.a_ids2[kbeg + i];
        gids[GK + i] = (i2 >= 0 && i2 < p.a_rows2) ? (int32_t)i2 : -1;
      }
    }
    __syncthreads();
  }
  // r04: the row-range id lists (forward of layer 0 with the fused lookup; the by == 0 tile of each row block, wave 0 alone: one
  // load of the 64 ids - an L1 hit -, ONE returning atomic instruction for the slots, ONE 8-byte store instruction at the end)
  [[maybe_unused]] uint32_t bslot = 0u, bgrp = 0u, blk = 0u;
  [[maybe_unused]] bool bemit = false;
  [[maybe_unused]] const int64_t brow = m0 + lane;
  if constexpr (GATHER && A_KC) {
    if (p.bk_pairs != nullptr && by == 0 && wave == 0) {                       // (wave-uniform)
      const int64_t bid = p.a_ids[brow < p.M ? brow : m0];
      bemit = brow < p.M && bid >= 0 && bid < p.a_rows;
      if (bemit) {
        const uint32_t key = (uint32_t)bid;
        uint32_t q = __umulhi(key, p.bk_magic);
        q -= (q * p.bk_width > key) ? 1u : 0u;
        bgrp = q < p.bk_groups ? q : p.bk_groups - 1u;
        blk = key - bgrp * p.bk_width;
        bslot = __hip_atomic_fetch_add(p.bk_counts + (size_t)bgrp * tt::kBucketCountStride, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  }
  auto load_a = [&](f32x4 (&st)[NST], int t) {
    const int64_t k0 = kbeg + (int64_t)t * BK;
    if constexpr (GATHER && A_KC) load_rows_kc(st, p.A, off1, p.A2, off2, k0, kend, tid);
    else if constexpr (GATHER) load_rows_mc(st, p.A, gids, p.A2, gids + GK, p.lda, m0, p.M, t * BK, klen, tid);
    else load_operand<A_KC>(st, p.A, p.lda, m0, p.M, k0, kend, tid);
  };

  // Register ring: the loads of PF k-tiles are in flight at any time (a dependent round trip through L2/HBM costs
  // ~1 us under load, one k-tile is only 16 MFMAs per wave), so the wait at step t is for a load issued PF steps ago.
  f32x4 sa[PF][NST], sb[PF][NST];
  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  float colsum = 0.f;
  STAMP(0);
  // dx epilogue mask (ReLU output of the previous layer): its 16 values per lane do not depend on the GEMM, so they are
  // requested BEFORE the operand tiles and folded into 16 bits as soon as they land (the operand loads issued after them
  // stay in flight): loading them after the MFMAs made the store phase of the masked dx tiles 9.3 us against 3.7-3.9 us
  // for the other kinds (r02 per-workgroup stamps)
  float mk[A_KC && B_KC ? 16 : 1];
  uint32_t mbits = 0xffffu, mword = 0u;
  const bool bitmask = A_KC && B_KC && p.mask_bits != nullptr;
  const bool masked = A_KC && B_KC && p.mask_src != nullptr && !bitmask;
  if constexpr (A_KC && B_KC) {
    if (bitmask) {          // sign bits: lane l < 32 fetches the 32-column word of tile row l of this wave's 32x32 block
      const int64_t m = m0 + wm * 32 + ln;
      const int64_t nw = (n0 + wn * 32) >> 5;
      mword = (m < p.M && nw < (p.N >> 5)) ? p.mask_bits[m * (p.N >> 5) + nw] : 0u;
    }
    if (masked) {
      const int64_t nn = n0 + wn * 32 + ln;
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const int64_t m = m0 + wm * 32 + tt::acc_row(reg, h);
        mk[reg] = (nn < p.N && m < p.M) ? p.mask_src[m * p.ldc + nn] : 0.f;
      }
    }
  }

#pragma unroll
  for (int u = 0; u < PF; ++u) {
    if (u < nk) {
      load_a(sa[u], u);
      load_operand<B_KC>(sb[u], p.B, p.ldb, n0, p.N, kbeg + (int64_t)u * BK, kend, tid);
    }
  }

  if constexpr (A_KC && B_KC) {
    if (masked) {
      mbits = 0u;
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) mbits |= (mk[reg] > 0.f ? 1u : 0u) << reg;
    }
    if (bitmask) {          // register `reg` of lane half h is tile row acc_row(reg, h), column ln
      mbits = 0u;
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const uint32_t w0 = (uint32_t)__builtin_amdgcn_readlane((int)mword, tt::acc_row(reg, 0));
        const uint32_t w1 = (uint32_t)__builtin_amdgcn_readlane((int)mword, tt::acc_row(reg, 1));
        mbits |= (((h ? w1 : w0) >> ln) & 1u) << reg;
      }
    }
  }

  for (int t0 = 0; t0 < nk; t0 += PF) {
#pragma unroll
    for (int u = 0; u < PF; ++u) {
      const int t = t0 + u;
      if (t < nk) {                                   // workgroup-uniform
        float* TA = smem + (t & 1) * 2 * TILE_F;
        float* TB = TA + TILE_F;
        store_operand<A_KC>(sa[u], TA, tid);          // waits only for ring slot u
        store_operand<B_KC>(sb[u], TB, tid);
        if (t + PF < nk) {
          load_a(sa[u], t + PF);
          load_operand<B_KC>(sb[u], p.B, p.ldb, n0, p.N, kbeg + (int64_t)(t + PF) * BK, kend, tid);
        }
        __syncthreads();                              // tile t visible; everyone is done with the other buffer
        if (t == 0) STAMP(1);
#pragma unroll
        for (int g = 0; g < BK / 8; ++g) {
          const f32x4 a4 = read_operand<A_KC>(TA, wm * 32 + ln, g, h);
          const f32x4 b4 = read_operand<B_KC>(TB, wn * 32 + ln, g, h);
#pragma unroll
          for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[s], b4[s], acc, 0, 0, 0);
        }
        if constexpr (COLSUM) {
          // db: column sums of the dz tile, once per n-tile (m-tile 0 only); rows beyond kend are zero-filled
          if (bx == 0 && tid < BN) {
#pragma unroll 8
            for (int kk = 0; kk < BK; ++kk) colsum += TB[kk * LS_MC + tid];
          }
        }
      }
    }
  }

  STAMP(2);
  float* C = p.C + (int64_t)zsplit * p.slab_stride;
  const int64_t n = n0 + wn * 32 + ln;
  uint32_t posbits = 0u;                        // fwd: (C > 0) of this lane's 16 elements
#if TT_GEMM_WIDE_STORE
  // The output tile leaves through LDS: accumulators -> [64][64 + 4] floats in the (now idle) operand buffers -> 16 bytes
  // per lane, a wave-instruction storing four whole 256-byte row pieces.  Stored straight from the accumulators a wave
  // needs 16 instructions of 4 bytes per lane (two 128-byte pieces each) and the epilogue is store-ISSUE-bound: in the fused
  // tower kernel the same change took the output stores from 3.7 to 1.1 us (r03 stamps, csrc/tower.hip).
  constexpr int LSO = BN + 4;
  // (rows of 16-byte pieces need N and ldc to be multiples of 4 floats: true for every tower layer - k % 4 == 0 and n % 4 == 0
  // are API requirements - but not for the hard-negative search's [nq][nc] scratch matrix, which keeps the 4-byte stores)
  const bool wide = ((p.ldc | p.N) & 3) == 0 && (reinterpret_cast<uintptr_t>(C) & 15u) == 0;
  if (wide) __syncthreads();                    // every wave has read its last fragments of the operand tiles
#else
  constexpr bool wide = false;
  constexpr int LSO = 1;
#endif
  if (n < p.N) {
    const float bias = p.bias != nullptr ? p.bias[n] : 0.f;
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
      const int64_t m = m0 + wm * 32 + tt::acc_row(reg, h);
      if (m < p.M) {
        float v = acc[reg] + bias;
        if (p.relu) v = fmaxf(v, 0.f);
        if constexpr (DROP) {        // inverted dropout on the activation (Keras Dropout after the Dense); compiled in
                                     // only for dropout launches: its 64-bit hash costs the plain forward kernel registers
          const uint64_t hsh = tt::splitmix(p.drop_key + p.drop_offset + (uint64_t)m * (uint64_t)p.N + (uint64_t)n);
          v = ((uint32_t)(hsh >> 40) < p.drop_p24) ? 0.f : v * p.drop_scale;
        }
        if constexpr (A_KC && B_KC) {
          if (masked || bitmask) v = ((mbits >> reg) & 1u) ? v * p.mask_scale : 0.f;
        } else {
          if (p.mask_src != nullptr) v = p.mask_src[m * p.ldc + n] > 0.f ? v * p.mask_scale : 0.f;
        }
        if (wide) smem[(wm * 32 + tt::acc_row(reg, h)) * LSO + wn * 32 + ln] = v;
        else C[m * p.ldc + n] = v;
        if constexpr (A_KC && !B_KC) posbits |= (v > 0.f ? 1u : 0u) << reg;
      }
    }
  }
#if TT_GEMM_WIDE_STORE
  if (wide) __syncthreads();
#pragma unroll
  for (int j = 0; wide && j < BM * BN / 4 / 256; ++j) {
    const int f = tid + 256 * j;
    const int row = f / (BN / 4), c4 = f % (BN / 4);
    const int64_t m = m0 + row, nn = n0 + 4 * c4;
    if (m < p.M && nn < p.N) *reinterpret_cast<f32x4*>(C + m * p.ldc + nn) = *reinterpret_cast<const f32x4*>(smem + row * LSO + 4 * c4);
  }
#endif
  if constexpr (A_KC && !B_KC) {
    if (p.relu_bits != nullptr) {      // (workgroup-uniform) one ballot per register: low half = tile row acc_row(reg, 0), high = (reg, 1)
      uint32_t myword = 0u;
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const uint64_t b = __builtin_amdgcn_ballot_w64(((posbits >> reg) & 1u) != 0u);
        tt::writelane(myword, (uint32_t)b, tt::acc_row(reg, 0));           // (see csrc/tower.hip)
        tt::writelane(myword, (uint32_t)(b >> 32), tt::acc_row(reg, 1));
      }
      const int64_t m = m0 + wm * 32 + lane;
      const int64_t nw = (n0 + wn * 32) >> 5;
      if (lane < 32 && m < p.M && nw < (p.N >> 5)) p.relu_bits[m * (p.N >> 5) + nw] = myword;
    }
  }
  if constexpr (COLSUM) {
    if (bx == 0 && tid < BN && n0 + tid < p.N) p.db_slabs[(int64_t)zsplit * p.N + n0 + tid] = colsum;
  }
  if constexpr (GATHER && A_KC) {
    if (bemit && bslot < p.bk_cap)                       // (a full list keeps counting: the optimizer falls back to its scan)
      p.bk_pairs[(size_t)bgrp * p.bk_cap + bslot] = (uint64_t)blk | ((uint64_t)((uint32_t)brow & 0xffffu) << 32) | ((uint64_t)(p.bk_gen & 0xffffu) << 48);
  }
#ifdef TT_GEMM_STAMPS
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  STAMP(3);
#endif
}

constexpr int kGidsInts(int gk, bool mc) { return (gk != 0 && mc) ? 2 * gk : 1; }

// __launch_bounds__(256, 4): 4 waves per SIMD = 4 workgroups per CU = every tile of a tower launch resident at once.  Without
// the bound the forward kernels took 122-126 VGPRs + 16 AGPRs = 3 waves/SIMD: 768 of layer 0's 1024 workgroups started, the
// rest ~10 us later (per-workgroup s_memrealtime stamps, r02: start spread 14 us, launch 22.8 us).
template <bool A_KC, bool B_KC, bool COLSUM, int GK, bool DROP>
__global__ __launch_bounds__(256, (GK > 256 ? 3 : 4)) void gemm_kernel(GemmBatch pb) {     // (the 8 KB id stage: 3 workgroups per CU by LDS)
  __shared__ __attribute__((aligned(16))) float smem[2 * 2 * TILE_F];
  __shared__ int32_t gids[kGidsInts(GK, !A_KC)];
  gemm_tile<A_KC, B_KC, COLSUM, GK, DROP>(pb.a[blockIdx.z / pb.splits], blockIdx.z % pb.splits, blockIdx.x, blockIdx.y, smem, gids);
}

// dx AND dw+db of one layer (of both towers) in ONE launch: the two kinds of tiles are independent given dz, so
// one launch ramp is saved and the output stores of one kind run under the MFMAs of the other.  Flat workgroup
// index -> (kind, problem, split, x, y); dw tiles are TN and split over the batch, dx tiles NT.
struct BwdBatch {
  GemmArgs ax[2], aw[2];
  int nprob, splits;
  int dx_gm, dx_gn;      // dx tiles per problem: dx_gm x dx_gn
  int dw_gm, dw_gn;      // dw tiles per problem and split
  int dw_first;          // 1: the dw tiles take the first workgroup indices (when their k-range is the longer one)
  int dx_pair;           // 1: a dx workgroup computes TWO column tiles (by, by + dx_gn/2) of its row block, one after the other
  // riders (r04): the dense parameter update of ANOTHER layer's segments - the layer above, whose gradient slabs the previous
  // backward launch completed - as the first `rider_blocks` workgroups of this launch (tt_dense_bwd_batched_update_f32)
  int rider_blocks, rider_segs, rider_opt;
  int rider_first[TT_MAX_DENSE_SEGS + 1];
  float rider_lr, rider_eps;
};

// ---- two layers' backward passes in ONE launch (r04): the lower layer's tiles wait for the rows of dz they read ----
// Layer l's dx IS layer l-1's dz.  As two launches the second starts when the last workgroup of the first has retired - the
// platform's 4 us between two kernels plus the first one's tail and the second one's ramp (r04 stamps: the last dx tiles of a
// backward launch end at 25-28 us of 29, most workgroups are gone by 20).  In one launch the upper layer's workgroups take the
// first indices; a dx workgroup of the upper layer, once its tiles are stored, RELEASES a counter of its 64-row block (agent
// scope: the XCDs' L2s are not coherent with each other - the release writes this L2's dirty lines back, the acquire below
// invalidates the reader's), and a lower-layer tile starts by waiting until the row blocks it reads have all their column
// tiles, then ACQUIRES.  Nothing else changes: same tiles, same order of every sum, results identical to the two launches.
// No deadlock: a waiting workgroup has a higher index than every workgroup it waits for, and workgroups are dispatched in
// index order (per XCD: round-robin) - when a consumer runs, its producers are running or done.  The wait is bounded all the
// same (kDepSpinLimit polls, then the error word is set and the tile goes on with what is there: a wrong result the caller
// sees in `err`, never a hung GPU).  The last consumer of a row block zeroes its two counters for the next launch.
struct DepFlags {
  uint32_t* ready;       // [nprob][row_blocks]  producers that have released the row block
  uint32_t* done;        // [nprob][row_blocks]  consumers that have passed their wait
  int32_t* err;          // set when a wait ran out
  int row_blocks;        // rows / 64
  uint32_t producers;    // releases per row block = dx workgroups per row block of the upper layer
  uint32_t consumers;    // waits per row block = dx workgroups per row block + dW tiles per batch split of the lower layer
};
constexpr int kDepSpinLimit = 1 << 22;

__device__ __forceinline__ void dep_release(const DepFlags& d, int prob, int row_block) {
  __syncthreads();                                           // every wave's stores of the tile(s) are issued and counted
  if (threadIdx.x == 0)
#ifdef TT_DEP_NOREL
    __hip_atomic_fetch_add(d.ready + prob * d.row_blocks + row_block, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#else
    __hip_atomic_fetch_add(d.ready + prob * d.row_blocks + row_block, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
#endif
}

__device__ __forceinline__ void dep_wait(const DepFlags& d, int prob, int first_block, int count) {
  if ((int)threadIdx.x < count) {
    const int i = prob * d.row_blocks + first_block + (int)threadIdx.x;
    int polls = 0;
    while (__hip_atomic_load(d.ready + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < d.producers) {
      __builtin_amdgcn_s_sleep(4);
      if (++polls > kDepSpinLimit) { atomicOr(d.err, 1); break; }
    }
    const uint32_t seen = atomicAdd(d.done + i, 1u);
    if (seen == d.consumers - 1u) {                          // everybody who reads this row block has seen it complete
      __hip_atomic_store(d.done + i, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(d.ready + i, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
#ifndef TT_DEP_NOACQ
  if (threadIdx.x < 64u) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
#endif
  __syncthreads();
}

// one workgroup of a layer's backward launch: flat index b -> (kind, problem, split, x, y).  ROLE 0: plain; 1: the upper layer
// of a fused pair (its dx workgroups release their row block); 2: the lower layer (every tile waits for its rows of dz).
template <int GK, int ROLE>
__device__ __forceinline__ void bwd_block(const BwdBatch& pb, int b, float* smem, int32_t* gids, const DepFlags& dep) {
  const int per_split = pb.dw_gm * pb.dw_gn;
  const int per_w = per_split * pb.splits;
  const int per_x = pb.dx_gm * (pb.dx_pair ? pb.dx_gn / 2 : pb.dx_gn);     // dx WORKGROUPS per problem
  const int n_dw = pb.nprob * per_w, n_dx = pb.nprob * per_x;
  // longest tiles first: workgroups are dispatched in index order, so the short tiles fill the tail
  const bool is_dw = pb.dw_first ? b < n_dw : b >= n_dx;
  if (is_dw) {
    if (!pb.dw_first) b -= n_dx;
    // XCD-aware order: the tiles of one (problem, batch split) group read the SAME x / dz slices (A by the column
    // tiles, B by the row tiles).  Workgroups are dealt round-robin over the 8 XCDs (private L2 each), so in plain index
    // order a group's tiles land on 8 different L2s and every slice is fetched up to 4x (PMC r02: 102-114 MB per
    // backward launch against ~45 MB algorithmic).  Here workgroup w takes tile k = (w/8) % T of group (w/8/T)*8 + w%8:
    // a group's T tiles run back to back on ONE XCD.  (Speed only; any placement computes the same tiles.)
    const int groups = pb.nprob * pb.splits;
    int grp = b / per_split, k = b % per_split;
    if (groups % 8 == 0 && (pb.dw_first || n_dx % 8 == 0)) {
      const int j = b >> 3;
      k = j % per_split;
      grp = (j / per_split) * 8 + (b & 7);
    }
    const int prob = grp / pb.splits, split = grp % pb.splits;
    if constexpr (ROLE == 2) {
      const int per = (int)(pb.aw[0].k_per_split / BM);      // row blocks of one batch split (the host checks: a whole number)
      dep_wait(dep, prob, split * per, per);
    }
    gemm_tile<false, false, true, GK>(pb.aw[prob], split, k % pb.dw_gm, k / pb.dw_gm, smem, gids);
  } else {
    if (pb.dw_first) b -= n_dw;
    const int prob = b / per_x;
    b -= prob * per_x;
    // (r04) the dx tiles - short k loops between a load round trip and a store drain, two of them per workgroup - go FIRST at
    // the matrix pipe; the dW tiles (since r04 one long barrier-free MFMA stream per wave) fill what they leave.  At equal
    // priority the oldest wave wins: the dW waves, dispatched first, held the pipe and the dx workgroups' second tiles ran
    // alone at the end of the launch, latency-bound (r04 stamps: second dx tiles from 19-24 us to 27-32 us of a 32 us launch).
    __builtin_amdgcn_s_setprio(TT_DX_PRIO);
    if constexpr (ROLE == 2) dep_wait(dep, prob, b % pb.dx_gm, 1);
    gemm_tile<true, true, false, 0>(pb.ax[prob], 0, b % pb.dx_gm, b / pb.dx_gm, smem, gids);
    if (pb.dx_pair) {
      __syncthreads();                               // every wave is done with the first tile's LDS buffers
      gemm_tile<true, true, false, 0>(pb.ax[prob], 0, b % pb.dx_gm, b / pb.dx_gm + pb.dx_gn / 2, smem, gids);
    }
    if constexpr (ROLE == 1) dep_release(dep, prob, b % pb.dx_gm);
  }
}

template <int GK>
__global__ __launch_bounds__(256, (GK > 256 ? 3 : 4)) void gemm_bwd_kernel(BwdBatch pb, tt::SegTable riders) {
  __shared__ __attribute__((aligned(16))) float smem[2 * 2 * TILE_F];
  __shared__ int32_t gids[kGidsInts(GK, true)];
  if ((int)blockIdx.x < pb.rider_blocks) {               // (workgroup-uniform)
    const int d = (int)blockIdx.x;
    int si = 0;
    while (si + 1 < pb.rider_segs && d >= pb.rider_first[si + 1]) ++si;
    const int nb = pb.rider_first[si + 1] - pb.rider_first[si];
    if (pb.rider_opt == TT_OPT_SGD) tt::dense_update_body<TT_OPT_SGD, 256, 16>(riders.seg[si], d - pb.rider_first[si], nb, 1, pb.rider_lr, pb.rider_eps);
    else tt::dense_update_body<TT_OPT_ADAGRAD, 256, 16>(riders.seg[si], d - pb.rider_first[si], nb, 1, pb.rider_lr, pb.rider_eps);
    return;
  }
  const DepFlags none{};
  bwd_block<GK, 0>(pb, (int)blockIdx.x - pb.rider_blocks, smem, gids, none);
}

struct Bwd2Batch {
  BwdBatch up, lo;
  int n_up;              // workgroups of the upper layer (they take the first indices)
  DepFlags dep;
};

template <int GK>
__global__ __launch_bounds__(256, (GK > 256 ? 3 : 4)) void tower_bwd2_kernel(Bwd2Batch pb) {
  __shared__ __attribute__((aligned(16))) float smem[2 * 2 * TILE_F];
  __shared__ int32_t gids[kGidsInts(GK, true)];
  const int b = (int)blockIdx.x;
  if (b < pb.n_up) bwd_block<0, 1>(pb.up, b, smem, gids, pb.dep);
  else bwd_block<GK, 2>(pb.lo, b - pb.n_up, smem, gids, pb.dep);
}

template <bool A_KC, bool B_KC, bool COLSUM, int GK = 0, bool DROP = false>
int launch(const GemmArgs* probs, int nprob, int splits, hipStream_t stream, const char* what, const char* tag) {
  const GemmArgs& a0 = probs[0];
  const int64_t gm = (a0.M + BM - 1) / BM, gn = (a0.N + BN - 1) / BN;
  TT_REQUIRE(nprob >= 1 && nprob <= 2, "%s: 1 or 2 problems per launch", what);
  TT_REQUIRE(gm <= 0x7fffffff && gn <= 65535 && (int64_t)splits * nprob <= 65535, "%s: grid too large", what);
  GemmBatch pb{};
  for (int i = 0; i < nprob; ++i) pb.a[i] = probs[i];
  pb.splits = splits;
  tt::launch(tag, (gemm_kernel<A_KC, B_KC, COLSUM, GK, DROP>), dim3((unsigned)gm, (unsigned)gn, (unsigned)(splits * nprob)), dim3(256), 0,
                     stream, pb);
  return tt::check_launch(what);
}

}  // namespace

// the fused embedding lookup of a layer-0 problem: A becomes the table, rows come through the ids
static int set_lookup(GemmArgs& g, const tt_dense_lookup& lk, const float* x, int64_t m, const char* fn) {
  if (lk.ids == nullptr) {
    TT_REQUIRE(x != nullptr, "%s: null input x (and no lookup)", fn);
    return TT_OK;
  }
  TT_REQUIRE(lk.table != nullptr && lk.table_rows > 0 && lk.table_rows <= 0x7fffffff, "%s: lookup needs a table of 1..2^31-1 rows", fn);
  TT_REQUIRE(tt::aligned16(lk.table) && tt::aligned16(lk.table2), "%s: lookup tables must be 16-byte aligned", fn);
  TT_REQUIRE((lk.table2 == nullptr) == (lk.ids2 == nullptr), "%s: lookup.table2 and lookup.ids2 go together", fn);
  TT_REQUIRE(lk.table2 == nullptr || (lk.table2_rows > 0 && lk.table2_rows <= 0x7fffffff), "%s: bad lookup.table2_rows", fn);
  TT_REQUIRE(m <= 32 * (int64_t)kMaxGatherK, "%s: the fused lookup supports m <= %d", fn, 32 * kMaxGatherK);
  g.A = lk.table; g.a_ids = lk.ids; g.a_rows = lk.table_rows;
  g.A2 = lk.table2; g.a_ids2 = lk.ids2; g.a_rows2 = lk.table2_rows;
  g.oob_flag = lk.oob_flag;
  return TT_OK;
}

static uint64_t dropout_stream_key(uint64_t seed, uint64_t tensor_id) {
  return tt::splitmix_host(tt::splitmix_host(seed) ^ (tensor_id * 0xD6E8FEB86659FD93ull));
}

// c[m][n] = sum_k a[m][k] * b[n][k]   (both operands k-contiguous; used by the hard-negative threshold search)
namespace tt {
int gemm_nt(const float* a, const float* b, float* c, int64_t m, int64_t n, int64_t k, hipStream_t stream) {
  GemmArgs g{};
  g.A = a; g.B = b; g.C = c; g.M = m; g.N = n; g.K = k; g.lda = k; g.ldb = k; g.ldc = n;
  g.mask_scale = 1.f; g.k_per_split = (k + BK - 1) / BK * BK;
  return launch<true, true, false>(&g, 1, 1, stream, "gemm_nt", "score_aux");
}
}  // namespace tt

extern "C" int tt_dense_fwd_batched_f32(const tt_dense_fwd_args* probs, int32_t n_probs, int64_t m, int32_t k, int32_t n,
                                        int32_t relu, float drop_rate, uint64_t seed, uint64_t counter_offset,
                                        tt_stream_t stream) {
  TT_REQUIRE(probs != nullptr && n_probs >= 1 && n_probs <= 2, "tt_dense_fwd_batched_f32: 1 or 2 problems");
  TT_REQUIRE(drop_rate >= 0.f && drop_rate < 1.f, "tt_dense_fwd_batched_f32: drop_rate must be in [0,1)");
  TT_REQUIRE(m > 0 && k > 0 && n > 0 && k % 4 == 0 && n % 4 == 0, "tt_dense_fwd_f32: need m>0, k%%4==0, n%%4==0 (m=%lld k=%d n=%d)",
             (long long)m, k, n);
  GemmArgs a[2] = {};
  const bool gather = probs[0].lookup.ids != nullptr;
  for (int i = 0; i < n_probs; ++i) {
    const tt_dense_fwd_args& q = probs[i];
    TT_REQUIRE((q.lookup.ids != nullptr) == gather, "tt_dense_fwd_batched_f32: the lookup must be given for all problems or for none");
    TT_REQUIRE(q.w && q.y, "tt_dense_fwd_f32: null pointer");
    TT_REQUIRE(tt::aligned16(q.x) && tt::aligned16(q.w) && tt::aligned16(q.y), "tt_dense_fwd_f32: pointers must be 16-byte aligned");
    a[i].A = q.x; a[i].B = q.w; a[i].C = q.y; a[i].M = m; a[i].N = n; a[i].K = k; a[i].lda = k; a[i].ldb = n; a[i].ldc = n;
    a[i].bias = q.b; a[i].relu = relu; a[i].mask_scale = 1.f; a[i].k_per_split = (k + BK - 1) / BK * BK;
    TT_REQUIRE(q.relu_bits == nullptr || n % 32 == 0, "tt_dense_fwd_f32: relu_bits need n %% 32 == 0 (n=%d)", n);
    a[i].relu_bits = q.relu_bits;
    int rc = set_lookup(a[i], q.lookup, q.x, m, "tt_dense_fwd_f32");
    if (rc != TT_OK) return rc;
    const tt_id_buckets& bk = q.lookup.buckets;
    if (q.lookup.ids != nullptr && bk.pairs != nullptr) {
      TT_REQUIRE(bk.counts != nullptr && bk.groups >= 1 && bk.width >= 1u && bk.cap >= 1 && m <= 65536,
                 "tt_dense_fwd_f32: lookup.buckets: counts / groups / width / cap must be set (and m <= 65536)");
      a[i].bk_counts = bk.counts; a[i].bk_pairs = bk.pairs; a[i].bk_width = bk.width; a[i].bk_groups = (uint32_t)bk.groups;
      a[i].bk_cap = (uint32_t)bk.cap; a[i].bk_gen = bk.gen;
      a[i].bk_magic = (uint32_t)((((uint64_t)1 << 32) / bk.width) + 1u);
    }
    if (drop_rate > 0.f) {
      a[i].drop_p24 = (uint32_t)((double)drop_rate * 16777216.0 + 0.5);
      a[i].drop_scale = 1.0f / (1.0f - drop_rate);
      a[i].drop_key = dropout_stream_key(seed, q.dropout_tensor_id);
      a[i].drop_offset = counter_offset;
    }
  }
  const bool drop = drop_rate > 0.f;
  if (gather && drop) return launch<true, false, false, 1, true>(a, n_probs, 1, tt::as_stream(stream), "tt_dense_fwd_f32(lookup)", "dense_fwd");
  if (gather) return launch<true, false, false, 1, false>(a, n_probs, 1, tt::as_stream(stream), "tt_dense_fwd_f32(lookup)", "dense_fwd");
  if (drop) return launch<true, false, false, 0, true>(a, n_probs, 1, tt::as_stream(stream), "tt_dense_fwd_f32", "dense_fwd");
  return launch<true, false, false>(a, n_probs, 1, tt::as_stream(stream), "tt_dense_fwd_f32", "dense_fwd");
}

extern "C" int tt_dense_fwd_f32(const float* x, const float* w, const float* b, float* y, int64_t m, int32_t k,
                                int32_t n, int32_t relu, tt_stream_t stream) {
  const tt_dense_fwd_args q{x, w, b, y, 0, {}, nullptr};
  return tt_dense_fwd_batched_f32(&q, 1, m, k, n, relu, 0.f, 0, 0, stream);
}

extern "C" int tt_dense_fwd_dropout_f32(const float* x, const float* w, const float* b, float* y, int64_t m, int32_t k,
                                        int32_t n, int32_t relu, float drop_rate, uint64_t seed, uint64_t tensor_id,
                                        uint64_t counter_offset, tt_stream_t stream) {
  const tt_dense_fwd_args q{x, w, b, y, tensor_id, {}, nullptr};
  return tt_dense_fwd_batched_f32(&q, 1, m, k, n, relu, drop_rate, seed, counter_offset, stream);
}

#ifndef TT_DW_MAX_SLABS
#define TT_DW_MAX_SLABS 32
#endif
// Batch rows per dW slab (the slab count is capped at TT_DW_MAX_SLABS).  64: batches below 8192 get more, shorter slabs - the
// dW tiles' k loop is a chain of load -> barrier -> MFMA rounds, 256 rows of it were 10 of gemm_bwd's 14 us at B = 256
// (r03 A/B, step time with 256 -> 64: cfg1 51.0 -> 43.8 us, cfg2 124.9 -> 119.6 us, cfg3 / cfg4 unchanged: capped at 32 slabs)
#ifndef TT_DW_SLAB_ROWS
#define TT_DW_SLAB_ROWS 64
#endif
extern "C" int32_t tt_dense_bwd_num_slabs(int64_t m) {
  int64_t s = (m + TT_DW_SLAB_ROWS - 1) / TT_DW_SLAB_ROWS;
  if (s < 1) s = 1;
  if (s > TT_DW_MAX_SLABS) s = TT_DW_MAX_SLABS;
  return (int32_t)s;
}

namespace {
int dense_bwd_batched(const tt_dense_bwd_args* probs, int32_t n_probs, float dx_scale, int64_t m, int32_t k, int32_t n,
                      const tt_dense_seg* rsegs, int32_t n_rsegs, int32_t ropt, float rlr, float reps, bool* riders_done, tt_stream_t stream_);
}
extern "C" int tt_dense_bwd_batched_f32(const tt_dense_bwd_args* probs, int32_t n_probs, float dx_scale, int64_t m, int32_t k,
                                        int32_t n, tt_stream_t stream_) {
  return dense_bwd_batched(probs, n_probs, dx_scale, m, k, n, nullptr, 0, 0, 0.f, 0.f, nullptr, stream_);
}

extern "C" int tt_dense_bwd_batched_update_f32(const tt_dense_bwd_args* probs, int32_t n_probs, float dx_scale, int64_t m, int32_t k,
                                               int32_t n, const tt_dense_seg* segs, int32_t n_segs, int32_t opt, float lr, float eps,
                                               tt_stream_t stream_) {
  TT_REQUIRE(segs != nullptr && n_segs >= 1 && n_segs <= TT_MAX_DENSE_SEGS, "tt_dense_bwd_batched_update_f32: 1..%d dense segments", TT_MAX_DENSE_SEGS);
  TT_REQUIRE(opt == TT_OPT_SGD || opt == TT_OPT_ADAGRAD, "tt_dense_bwd_batched_update_f32: unknown optimizer %d", opt);
  for (int i = 0; i < n_segs; ++i) {
    const tt_dense_seg& s = segs[i];
    TT_REQUIRE(s.count > 0 && s.n_slabs >= 1 && s.grad_slabs != nullptr && s.param != nullptr, "tt_dense_bwd_batched_update_f32: segment %d: bad count/slabs/param", i);
    TT_REQUIRE(opt == TT_OPT_SGD || s.accum != nullptr, "tt_dense_bwd_batched_update_f32: segment %d: Adagrad needs accum", i);
    for (int j = 0; j < n_probs; ++j)
      TT_REQUIRE(s.grad_slabs != probs[j].dw_slabs && s.grad_slabs != probs[j].db_slabs && s.param != probs[j].w,
                 "tt_dense_bwd_batched_update_f32: segment %d belongs to THIS layer (its slabs are written / its weights read by this launch)", i);
  }
  bool done = false;
  int rc = dense_bwd_batched(probs, n_probs, dx_scale, m, k, n, segs, n_segs, opt, lr, eps, &done, stream_);
  if (rc != TT_OK || done) return rc;
  return tt_dense_update_f32(segs, n_segs, opt, 1, lr, eps, stream_);      // (dx only / dw only forms: the update as its own launch)
}

namespace {
// the dx (NT) and dW (TN, split over the batch) problems of one layer's backward pass
int build_bwd_args(const tt_dense_bwd_args* probs, int32_t n_probs, float dx_scale, int64_t m, int32_t k, int32_t n, GemmArgs (&ax)[2],
                   GemmArgs (&aw)[2], int& splits, bool& want_dx, bool& want_dw, bool& gather) {
  TT_REQUIRE(probs != nullptr && n_probs >= 1 && n_probs <= 2, "tt_dense_bwd_batched_f32: 1 or 2 problems");
  TT_REQUIRE(m > 0 && k > 0 && n > 0 && k % 4 == 0 && n % 4 == 0, "tt_dense_bwd_f32: need m>0, k%%4==0, n%%4==0 (m=%lld k=%d n=%d)",
             (long long)m, k, n);
  splits = tt_dense_bwd_num_slabs(m);
  want_dx = probs[0].dx != nullptr;
  want_dw = probs[0].dw_slabs != nullptr;
  gather = probs[0].lookup.ids != nullptr;
  for (int i = 0; i < n_probs; ++i) {
    const tt_dense_bwd_args& q = probs[i];
    TT_REQUIRE((q.lookup.ids != nullptr) == gather, "tt_dense_bwd_batched_f32: the lookup must be given for all problems or for none");
    TT_REQUIRE(q.w && q.dz, "tt_dense_bwd_f32: null pointer");
    TT_REQUIRE((q.dx != nullptr) == want_dx, "tt_dense_bwd_batched_f32: dx must be given for all problems or for none");
    TT_REQUIRE((q.dw_slabs != nullptr) == want_dw && (q.db_slabs != nullptr) == want_dw,
               "tt_dense_bwd_batched_f32: dw_slabs and db_slabs must be given together, for all problems or for none");
    TT_REQUIRE(want_dx || want_dw, "tt_dense_bwd_f32: nothing to compute (dx and dw_slabs are both NULL)");
    TT_REQUIRE(tt::aligned16(q.x) && tt::aligned16(q.w) && tt::aligned16(q.dz) && (q.dw_slabs == nullptr || tt::aligned16(q.dw_slabs)) &&
                   (q.dx == nullptr || tt::aligned16(q.dx)),
               "tt_dense_bwd_f32: pointers must be 16-byte aligned");
    // dx[m][k] = sum_n dz[m][n] * w[k][n]
    ax[i].A = q.dz; ax[i].B = q.w; ax[i].C = q.dx; ax[i].M = m; ax[i].N = k; ax[i].K = n; ax[i].lda = n; ax[i].ldb = n; ax[i].ldc = k;
    ax[i].mask_src = q.dx_relu_src; ax[i].mask_scale = dx_scale; ax[i].k_per_split = (n + BK - 1) / BK * BK;
    TT_REQUIRE(q.dx_relu_bits == nullptr || k % 32 == 0, "tt_dense_bwd_f32: dx_relu_bits need k %% 32 == 0 (k=%d)", k);
    ax[i].mask_bits = q.dx_relu_bits;
    // dw[k][n] = sum_b x[b][k] * dz[b][n], split over the batch into slabs; db rides along
    aw[i].A = q.x; aw[i].B = q.dz; aw[i].C = q.dw_slabs; aw[i].M = k; aw[i].N = n; aw[i].K = m; aw[i].lda = k; aw[i].ldb = n; aw[i].ldc = n;
    aw[i].k_per_split = ((m + splits - 1) / splits + BK - 1) / BK * BK;
    aw[i].slab_stride = (int64_t)k * n;
    aw[i].db_slabs = q.db_slabs;
    aw[i].mask_scale = 1.f;
    int rcl = set_lookup(aw[i], q.lookup, q.x, m, "tt_dense_bwd_f32");
    if (rcl != TT_OK) return rcl;
    aw[i].oob_flag = nullptr;            // the forward pass has flagged bad ids already
    TT_REQUIRE(!gather || aw[i].k_per_split <= kMaxGatherK, "tt_dense_bwd_f32: fused lookup: batch split too long");
  }
  return TT_OK;
}

// one layer's launch description (both kinds of tiles wanted)
void fill_bwd_batch(BwdBatch& pb, const GemmArgs (&ax)[2], const GemmArgs (&aw)[2], int n_probs, int splits) {
  for (int i = 0; i < n_probs; ++i) { pb.ax[i] = ax[i]; pb.aw[i] = aw[i]; }
  pb.nprob = n_probs; pb.splits = splits;
  pb.dx_gm = (int)((ax[0].M + BM - 1) / BM); pb.dx_gn = (int)((ax[0].N + BN - 1) / BN);
  pb.dw_gm = (int)((aw[0].M + BM - 1) / BM); pb.dw_gn = (int)((aw[0].N + BN - 1) / BN);
  pb.dw_first = aw[0].k_per_split > ax[0].k_per_split;     // k-tiles per tile: batch/splits rows vs n columns
  // 4 workgroups per CU are resident (LDS, VGPRs): 1024 on the chip.  When the launch has more AND a dx tile is at most
  // half as long as a dw tile (layer 1 of cfg3: 1024 dx tiles of 4 k-tiles + 512 dw tiles of 8), the extra workgroups start
  // when the first ones retire and the launch runs 1.5 rounds (r02 stamps: 512 workgroups started ~20 us late, end 39.7 us,
  // median workgroup end 22.8 us).  Pairing two dx tiles per workgroup makes every workgroup equally long and all resident.
  const int64_t dx_tiles = (int64_t)pb.dx_gm * pb.dx_gn, dw_tiles = (int64_t)pb.dw_gm * pb.dw_gn * splits;
  pb.dx_pair = (n_probs * (dx_tiles + dw_tiles) > 1024 && pb.dx_gn % 2 == 0 && 2 * ax[0].k_per_split <= aw[0].k_per_split) ? 1 : 0;
  if (const char* e = std::getenv("TT_GEMM_DX_PAIR")) pb.dx_pair = (std::atoi(e) != 0 && pb.dx_gn % 2 == 0) ? 1 : 0;
}
int64_t bwd_batch_blocks(const BwdBatch& pb) {
  const int64_t dx_tiles = (int64_t)pb.dx_gm * pb.dx_gn, dw_tiles = (int64_t)pb.dw_gm * pb.dw_gn * pb.splits;
  return (int64_t)pb.nprob * (dx_tiles / (pb.dx_pair ? 2 : 1) + dw_tiles);
}

int dense_bwd_batched(const tt_dense_bwd_args* probs, int32_t n_probs, float dx_scale, int64_t m, int32_t k, int32_t n,
                      const tt_dense_seg* rsegs, int32_t n_rsegs, int32_t ropt, float rlr, float reps, bool* riders_done, tt_stream_t stream_) {
  hipStream_t stream = tt::as_stream(stream_);
  GemmArgs ax[2] = {}, aw[2] = {};
  int splits = 1;
  bool want_dx = false, want_dw = false, gather = false;
  int rc = build_bwd_args(probs, n_probs, dx_scale, m, k, n, ax, aw, splits, want_dx, want_dw, gather);
  if (rc != TT_OK) return rc;
  if (want_dx && want_dw) {
    BwdBatch pb{};
    fill_bwd_batch(pb, ax, aw, n_probs, splits);
    tt::SegTable riders{};
    if (rsegs != nullptr) {                                  // one thread per 4 elements, at most 32 blocks per segment
      for (int i = 0; i < n_rsegs; ++i) {
        int64_t nb = (rsegs[i].count / 4 + 255) / 256;
        if (nb < 1) nb = 1;
        if (nb > 32) nb = 32;
        pb.rider_first[i + 1] = pb.rider_first[i] + (int)nb;
        riders.seg[i] = rsegs[i];
      }
      pb.rider_segs = n_rsegs; pb.rider_blocks = pb.rider_first[n_rsegs]; pb.rider_opt = ropt; pb.rider_lr = rlr; pb.rider_eps = reps;
      *riders_done = true;
    }
    const int64_t blocks = bwd_batch_blocks(pb) + pb.rider_blocks;
    TT_REQUIRE(blocks <= 0x7fffffff && (ax[0].M + BM - 1) / BM <= 0x3fffffff, "tt_dense_bwd_f32: grid too large");
    if (gather && aw[0].k_per_split <= 256) tt::launch("dense_bwd", gemm_bwd_kernel<256>, dim3((unsigned)blocks), dim3(256), 0, stream, pb, riders);
    else if (gather) tt::launch("dense_bwd", gemm_bwd_kernel<kMaxGatherK>, dim3((unsigned)blocks), dim3(256), 0, stream, pb, riders);
    else tt::launch("dense_bwd", gemm_bwd_kernel<0>, dim3((unsigned)blocks), dim3(256), 0, stream, pb, riders);
    return tt::check_launch("tt_dense_bwd_f32(dx+dw)");
  }
  if (want_dx && (rc = launch<true, true, false>(ax, n_probs, 1, stream, "tt_dense_bwd_f32(dx)", "dense_bwd_dx")) != TT_OK) return rc;
  if (!want_dw) return TT_OK;
  if (gather && aw[0].k_per_split <= 256)
    return launch<false, false, true, 256>(aw, n_probs, splits, stream, "tt_dense_bwd_f32(dw, lookup)", "dense_bwd_dw");
  if (gather) return launch<false, false, true, kMaxGatherK>(aw, n_probs, splits, stream, "tt_dense_bwd_f32(dw, lookup)", "dense_bwd_dw");
  return launch<false, false, true>(aw, n_probs, splits, stream, "tt_dense_bwd_f32(dw)", "dense_bwd_dw");
}
}  // namespace

// ---- two layers' backward passes in one launch (tower_bwd2_kernel above) ----
namespace {
int64_t bwd2_row_blocks(int64_t m) { return m / BM; }
}
extern "C" int32_t tt_tower_bwd2_supported(int64_t m, int32_t k0, int32_t k1, int32_t n) {
  if (m <= 0 || k0 <= 0 || k1 <= 0 || n <= 0 || k0 % 4 || k1 % 4 || n % 4 || m % BM) return 0;
  const int splits = tt_dense_bwd_num_slabs(m);
  const int64_t kps = ((m + splits - 1) / splits + BK - 1) / BK * BK;
  if (kps % BM != 0 || kps * splits != m) return 0;          // a batch split = a whole number of 64-row blocks
  if (kps / BM > 64) return 0;                               // (one lane of wave 0 polls each of a split's row blocks)
  return 1;
}

extern "C" int64_t tt_tower_bwd2_workspace_bytes(int64_t m) {
  if (m <= 0) return 256;
  return (4 * bwd2_row_blocks(m) * 4 + 256 + 255) / 256 * 256;   // ready + done for two problems, the error word; zeroed once
}

extern "C" int tt_tower_bwd2_batched_f32(const tt_dense_bwd_args* upper, const tt_dense_bwd_args* lower, int32_t n_probs,
                                         float dx_scale_upper, float dx_scale_lower, int64_t m, int32_t k0, int32_t k1, int32_t n,
                                         void* workspace, tt_stream_t stream_) {
  TT_REQUIRE(upper != nullptr && lower != nullptr && workspace != nullptr, "tt_tower_bwd2_batched_f32: null pointer");
  TT_REQUIRE((reinterpret_cast<uintptr_t>(workspace) & 255u) == 0, "tt_tower_bwd2_batched_f32: workspace must be 256-byte aligned");
  if (!tt_tower_bwd2_supported(m, k0, k1, n))
    return tt::fail(TT_ERR_UNSUPPORTED, "tt_tower_bwd2_batched_f32: shape (m %lld, %d -> %d -> %d) not supported (tt_tower_bwd2_supported)",
                    (long long)m, k0, k1, n);
  hipStream_t stream = tt::as_stream(stream_);
  GemmArgs axu[2] = {}, awu[2] = {}, axl[2] = {}, awl[2] = {};
  int su = 1, sl = 1;
  bool dxu = false, dwu = false, gu = false, dxl = false, dwl = false, gl = false;
  int rc = build_bwd_args(upper, n_probs, dx_scale_upper, m, k1, n, axu, awu, su, dxu, dwu, gu);
  if (rc != TT_OK) return rc;
  rc = build_bwd_args(lower, n_probs, dx_scale_lower, m, k0, k1, axl, awl, sl, dxl, dwl, gl);
  if (rc != TT_OK) return rc;
  TT_REQUIRE(dxu && dwu && dxl && dwl, "tt_tower_bwd2_batched_f32: both layers need dx and dw_slabs");
  TT_REQUIRE(!gu, "tt_tower_bwd2_batched_f32: only the lower layer can carry the fused lookup");
  for (int i = 0; i < n_probs; ++i)
    TT_REQUIRE(upper[i].dx == lower[i].dz, "tt_tower_bwd2_batched_f32: problem %d: the upper layer's dx must BE the lower layer's dz", i);
  Bwd2Batch pb{};
  fill_bwd_batch(pb.up, axu, awu, n_probs, su);
  fill_bwd_batch(pb.lo, axl, awl, n_probs, sl);
  if (const char* e = std::getenv("TT_BWD2_UP_DXFIRST")) { if (std::atoi(e) != 0) pb.up.dw_first = 0; }
  if (const char* e = std::getenv("TT_BWD2_UP_PAIR")) pb.up.dx_pair = (std::atoi(e) != 0 && pb.up.dx_gn % 2 == 0) ? 1 : 0;
  const int64_t n_up = bwd_batch_blocks(pb.up), n_lo = bwd_batch_blocks(pb.lo);
  TT_REQUIRE(n_up + n_lo <= 0x7fffffff, "tt_tower_bwd2_batched_f32: grid too large");
  pb.n_up = (int)n_up;
  const int64_t rb = bwd2_row_blocks(m);
  uint32_t* w32 = static_cast<uint32_t*>(workspace);
  pb.dep.ready = w32;
  pb.dep.done = w32 + 2 * rb;
  pb.dep.err = reinterpret_cast<int32_t*>(w32 + 4 * rb);
  pb.dep.row_blocks = (int)rb;
  pb.dep.producers = (uint32_t)(pb.up.dx_pair ? pb.up.dx_gn / 2 : pb.up.dx_gn);
  pb.dep.consumers = (uint32_t)((pb.lo.dx_pair ? pb.lo.dx_gn / 2 : pb.lo.dx_gn) + pb.lo.dw_gm * pb.lo.dw_gn);
  const unsigned blocks = (unsigned)(n_up + n_lo);
  if (gl && awl[0].k_per_split <= 256) tt::launch("dense_bwd", tower_bwd2_kernel<256>, dim3(blocks), dim3(256), 0, stream, pb);
  else if (gl) tt::launch("dense_bwd", tower_bwd2_kernel<kMaxGatherK>, dim3(blocks), dim3(256), 0, stream, pb);
  else tt::launch("dense_bwd", tower_bwd2_kernel<0>, dim3(blocks), dim3(256), 0, stream, pb);
  return tt::check_launch("tt_tower_bwd2_batched_f32");
}

extern "C" int tt_dense_bwd_f32(const float* x, const float* w, const float* dz, float* dx, const float* dx_relu_src,
                                float* dw_slabs, float* db_slabs, int64_t m, int32_t k, int32_t n, tt_stream_t stream_) {
  return tt_dense_bwd_scaled_f32(x, w, dz, dx, dx_relu_src, 1.0f, dw_slabs, db_slabs, m, k, n, stream_);
}

extern "C" int tt_dense_bwd_scaled_f32(const float* x, const float* w, const float* dz, float* dx, const float* dx_relu_src,
                                       float dx_scale, float* dw_slabs, float* db_slabs, int64_t m, int32_t k, int32_t n,
                                       tt_stream_t stream_) {
  const tt_dense_bwd_args q{x, w, dz, dx, dx_relu_src, dw_slabs, db_slabs, {}, nullptr};
  return tt_dense_bwd_batched_f32(&q, 1, dx_scale, m, k, n, stream_);
}

#ifdef TT_GEMM_STAMPS
extern "C" int tt_debug_gemm_stamps(unsigned long long* host_out, int n) {
  return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_stamps), sizeof(unsigned long long) * (size_t)n) == hipSuccess ? 0 : 2;
}
#endif
