// K3, fused — the forward pass of a TWO-layer tower in one launch (the `model:` block's
// *_tower_dims, /root/reference/configs/data_config.yaml:56-57):   h = relu(x W0 + b0) [dropout],   y = h W1 + b1.
//
// Why: as two launches (csrc/gemm.hip) every workgroup of a launch is in the same phase at the same time - operand loads,
// then 4-8 k-tiles of f32 MFMAs, then 16.8 MB of stores - and the phases add (r02: 19.3 + 18.2 us at cfg3 for 13.7 us of MFMA
// time); the hidden activation goes out to HBM and comes straight back.  Here a workgroup owns 32 batch rows of one tower for
// BOTH layers: the [32, H] hidden tile stays in LDS (33 KB at H = 256, in the place of the input tile it was computed from),
// every wave stages the weight columns it owns through a private LDS buffer (see the kernel), h's global stores (the backward
// pass needs h and its sign bits) drain under layer 1's MFMAs.  ~67 KB of LDS: 512 workgroups at cfg3 (8192 rows x 2 towers
// / 32) = two per CU, all resident, every wave free of the others between the three barriers.
// (r02's first fused form took a 64-row block with LDS weight tiles: 134 KB, ONE workgroup per CU, 46.7 vs 38.6 us.)
//
// Arithmetic: the same v_mfma_f32_32x32x2_f32 products in the same k order as gemm.hip (k = 8g + 4*lanehalf + s inside
// a group of 8, groups ascending), the same bias / ReLU / dropout / sign-bit epilogue: results are bit-identical to
// tt_dense_fwd_batched_f32 called twice (tests/test_gpu_parity.py::test_fused_tower_forward_is_bit_identical_to_two_layers).
// The embedding lookup (a1) is fused into the input tile exactly as there (tt_dense_lookup).
#include "common.h"
#include <cstdlib>
#include <type_traits>

namespace {

using tt::f32x4;
using tt::f32x16;

constexpr int RB = 32;            // batch rows per workgroup

struct Tower2Args {
  const float* x;                 // [M, K0] or NULL with a lookup
  const float* w0; const float* b0; float* h; uint32_t* h_bits;     // [K0, H], [H], [M, H], [M, H/32] (optional)
  const float* w1; const float* b1; float* y;                        // [H, N1], [N1], [M, N1]
  // fused lookup (tt_dense_lookup): row r of x is row ids[r] of table (+ row ids2[r] of table2)
  const float* table; const int64_t* ids; int64_t table_rows;
  const float* table2; const int64_t* ids2; int64_t table2_rows;
  int32_t* oob_flag;
  uint64_t drop_key;              // counter stream of this tower's hidden-layer dropout mask
  // row-range id lists for the optimizer launch of the same step (tt_id_buckets): every in-range id of `ids` is appended to the
  // list of the row range it falls in - (id - g*width) | position << 32 | generation << 48 at pairs[g*cap + slot], slot drawn
  // from counts[g] - so that launch's sorting workgroups read ~64 entries each instead of all the batch's ids
  uint32_t* bk_counts; uint64_t* bk_pairs;
  uint32_t bk_width, bk_magic, bk_groups, bk_cap, bk_gen;
};
struct Tower2Batch {
  Tower2Args a[2];
  int64_t M;
  int K0, H, N1;
  int blocks_per_prob;
  uint32_t drop_p24; float drop_scale; uint64_t drop_offset;
};

__device__ __forceinline__ f32x4 ldg4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }

#ifdef TT_TOWER_STAMPS      // per-workgroup phase stamps (debug builds: scratch/tower_stamps.py)
__device__ unsigned long long g_tstamps[1024 * 8];
#define TSTAMP(i) do { if (threadIdx.x == 0) g_tstamps[(blockIdx.x % 1024) * 8 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define TSTAMP(i) do {} while (0)
#endif

// HB = 32-column blocks of h per wave (H = 128 * HB), NB = 32-column blocks of y per wave (N1 = 128 * NB).
//
// Operands.  A (the 32 input rows, then the 32 hidden rows) is shared by the four waves: LDS, [row][k], rows padded by 4
// floats, one ds_read_b128 per 4 MFMAs.  B (the weights) is NOT shared inside a workgroup - wave w owns output columns
// [w*H/4, (w+1)*H/4) of layer 0 and [w*N1/4, ...) of layer 1 - so every wave stages ITS OWN weight sub-tiles, 1024 floats
// each ([16 k][64 cols] for layer 0 at H = 256, [32 k][32 cols] for layer 1 at N1 = 128): four 16-byte global loads per lane
// (whole 128- / 256-byte row segments), a register ring TT_TOWER_PF tiles deep, a private double buffer in LDS, fragments
// read back with ds_read_b32.  A wave's LDS operations execute in order, so write -> read of its private buffer needs no
// workgroup barrier: two barriers per workgroup in front of the MFMA loops (input tile ready, hidden tile ready) and two
// around the output tile's trip through LDS.
// r03 forms measured at cfg3 (8192 rows x 2 towers, 128 -> 256 -> 128; two launches of gemm.hip: 19.4 + 16.9 us):
//   weight k-tiles shared through LDS, one workgroup barrier per k-tile (18 barriers, 16 MFMAs per wave between them): 33.7 us,
//     the same with one or two tiles of loads in flight - it waits at the barriers, not for the loads;
//   no LDS weights, every lane loading its own fragment with global_load_dword (one 4-byte load per MFMA): 34.2 us, the same at
//     2 or 4 workgroups per CU - 32 B/clk/CU of 4-byte requests is what the vector memory pipe delivers;
//   this form (private per-wave LDS buffer + register ring, 16-byte output stores from LDS, ids first): 29.6 us - the steps in
//     between and the stamps are in profiles/r03_tower_forms.txt.
#ifndef TT_TOWER_PF
#define TT_TOWER_PF 2             // weight sub-tiles of loads in flight per wave (register ring)
#endif
constexpr int WT_F = 1024;        // floats per weight sub-tile of a wave
template <int HB, int NB, bool DROP>
__global__ __launch_bounds__(256, 2) void tower_fwd2_kernel(Tower2Batch pb) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int H = 128 * HB, N1 = 128 * NB;
  constexpr int LH = H + 4;                                  // hidden tile [32][H]: row stride
  constexpr int C1 = H / 4, C2 = N1 / 4;                     // columns per wave in layer 0 / layer 1
  constexpr int KT1 = WT_F / C1, KT2 = WT_F / C2;            // k rows per sub-tile (H = 256: 16; N1 = 128: 32)
  constexpr int L1 = C1 + 4, L2 = C2 + 4;                    // sub-tile row strides in LDS
  constexpr int WB_F = (KT1 * L1 > KT2 * L2) ? KT1 * L1 : KT2 * L2;   // floats per private buffer
  constexpr int PF = TT_TOWER_PF;
  const int K0 = pb.K0;
  const int LX = K0 + 4;                                     // input tile [32][K0]: row stride
  float* XH = smem;                                          // hidden tile [32][H + 4] (later: the output tile on its way out)
  float* XT = smem + RB * LH;                                // input tile  [32][K0 + 4]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hh = lane >> 5, ln = lane & 31;
  float* WP = smem + RB * LH + RB * LX + wave * WB_F;        // this wave's weight buffer (ONE: its LDS operations execute in order,
                                                             // so tile t+1 is written behind the last fragment read of tile t)
  const int prob = blockIdx.x / pb.blocks_per_prob;
  const Tower2Args& p = pb.a[prob];
  const int64_t m0 = (int64_t)(blockIdx.x - prob * pb.blocks_per_prob) * RB;

  // sub-tile t of this wave: k rows [t*KT, (t+1)*KT) x its CW columns; lane i moves float4 number i + 64 j (row f / (CW/4))
  f32x4 st[PF][4];
  auto load_w = [&](f32x4 (&r)[4], const float* __restrict__ w, int ncols, int cw, int kt, int t) {
    const int c4n = cw / 4;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int f = lane + 64 * j;
      const int kk = f / c4n, c4 = f - kk * c4n;
      r[j] = ldg4(w + (int64_t)(t * kt + kk) * ncols + wave * cw + 4 * c4);
    }
  };
  auto store_w = [&](const f32x4 (&r)[4], float* T, int cw, int ld) {
    const int c4n = cw / 4;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int f = lane + 64 * j;
      const int kk = f / c4n, c4 = f - kk * c4n;
      *reinterpret_cast<f32x4*>(T + kk * ld + 4 * c4) = r[j];
    }
  };
  const int nt1 = K0 / KT1;
  TSTAMP(0);
  // ---- the input tile: 32 rows x K0, through the ids when the lookup is fused.  Four float4 per thread and round, in phases:
  // the round's ids, then its rows, then the LDS stores.  Every load of a phase is UNCONDITIONAL, from a clamped (valid)
  // address: written as `if (live) { id = ids[r]; if (in range) ... }` per float4, hipcc put each id load in its own branch
  // with `s_waitcnt vmcnt(0)` behind it - four dependent round trips in front of the rows (r03 ISA audit,
  // tests/isa_audit/audit_serial_loads.py).  Rows that do not exist / ids out of range read row 0 and are zeroed afterwards.
  // The ids of round 0 are the FIRST loads of the kernel - they head its longest dependency chain (ids -> rows -> LDS ->
  // barrier -> first MFMA); the bias and weight prefetches are issued behind them, the rows behind those. ----
  const int c4n = K0 / 4;                                    // float4 per row
  const int total = RB * c4n;
  const bool has_ids = p.ids != nullptr, has2 = p.table2 != nullptr;        // (uniform)
  const int64_t* idp1 = has_ids ? p.ids : reinterpret_cast<const int64_t*>(p.x);
  const int64_t* idp2 = has2 ? p.ids2 : idp1;
  int64_t rr[4], id1[4], id2[4];
  int c4s[4];
  bool live[4];
  auto request_ids = [&](int f0) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int f = f0 + tid + 256 * j;
      const int row = f / c4n;
      c4s[j] = f - row * c4n;
      const int64_t r = m0 + row;
      live[j] = f < total && r < pb.M;
      rr[j] = live[j] ? r : m0;                              // (m0 < M: the workgroup has at least one row)
    }
    // no branch, no "0 unless loaded" merge (hipcc resolves that with register copies behind an s_waitcnt right after the
    // loads): without a lookup the same 8 bytes are read from x itself (in bounds: 8 r < 4 M K0) and never looked at
#pragma unroll
    for (int j = 0; j < 4; ++j) id1[j] = idp1[rr[j]];
#pragma unroll
    for (int j = 0; j < 4; ++j) id2[j] = idp2[rr[j]];
  };
  // row-range id lists (r04, tt_id_buckets): the 32 ids of this workgroup's rows are appended to the lists of the row ranges they
  // fall in by lanes 0..31 of WAVE 0 alone - one load of the ids (again: an L1 hit), ONE returning atomic instruction for the
  // slots, issued behind the round's row loads (memory operations return in order: in front of them the rows would wait for the
  // slots), ONE 8-byte store instruction for the entries at the very end of the kernel.  One counter per 256-byte line
  // (kBucketCountStride): the first version packed a table's ~220 counters into 7 lines and the forward launch took 25 us
  // longer - device-scope atomics on one LINE serialise (~11 ns each), whatever the word.  (The second version let the lane that
  // holds float4 0 of a row do it: 2 active lanes x 4 atomic + 4 store instructions in every wave, +1 us on the launch.)
  const bool lists = has_ids && p.bk_pairs != nullptr && wave == 0;           // (wave-uniform)
  uint32_t bslot = 0u, bgrp = 0u, blk = 0u;
  bool emit = false;
  int64_t bid = -1;
  const int64_t brow = m0 + ln;
  if (lists) bid = p.ids[brow < pb.M ? brow : m0];
  auto draw_slot = [&]() {
    if (lists) {
      emit = lane < 32 && brow < pb.M && bid >= 0 && bid < p.table_rows;
      if (emit) {
        const uint32_t key = (uint32_t)bid;
        uint32_t q = __umulhi(key, p.bk_magic);
        q -= (q * p.bk_width > key) ? 1u : 0u;
        bgrp = q < p.bk_groups ? q : p.bk_groups - 1u;
        blk = key - bgrp * p.bk_width;
        bslot = __hip_atomic_fetch_add(p.bk_counts + (size_t)bgrp * tt::kBucketCountStride, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  };
  auto fetch_rows = [&](int f0) {
    bool ok1[4], ok2[4];
    f32x4 v[4], v2[4];
    const float* src = has_ids ? p.table : p.x;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      ok1[j] = live[j] && (!has_ids || (id1[j] >= 0 && id1[j] < p.table_rows));
      ok2[j] = live[j] && has2 && id2[j] >= 0 && id2[j] < p.table2_rows;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int64_t row1 = has_ids ? (ok1[j] ? id1[j] : 0) : rr[j];
      v[j] = ldg4(src + row1 * K0 + 4 * c4s[j]);
    }
    if (f0 == 0) draw_slot();
    if (has2) {
#pragma unroll
      for (int j = 0; j < 4; ++j) v2[j] = ldg4(p.table2 + (ok2[j] ? id2[j] : 0) * K0 + 4 * c4s[j]);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const f32x4 zero = f32x4{0.f, 0.f, 0.f, 0.f};
      f32x4 x = ok1[j] ? v[j] : zero;
      if (has2) x = ok2[j] ? x + v2[j] : x;                                  // one f32 add per element, only where the second row exists
      const int f = f0 + tid + 256 * j;
      if (f < total) *reinterpret_cast<f32x4*>(XT + (f / c4n) * LX + 4 * c4s[j]) = x;
    }
    if (has_ids && p.oob_flag != nullptr) {                                  // an id outside [0, rows) other than the -1 padding is reported
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (live[j] && c4s[j] == 0 &&
            ((!ok1[j] && id1[j] != -1) || (has2 && !ok2[j] && id2[j] != -1)))
          atomicOr(p.oob_flag, 1);
    }
  };
  request_ids(0);
  float bias0[HB], bias1[NB];                                // requested up front: a load behind the k loop would be exposed
  // (unconditional as well: without a bias the same index is read from the weight matrix - in bounds - and the value dropped;
  // a branch here ends the basic block, and hipcc then parks the wait for the ids in front of it)
  const float* b0p = p.b0 != nullptr ? p.b0 : p.w0;
  const float* b1p = p.b1 != nullptr ? p.b1 : p.w1;
#pragma unroll
  for (int b = 0; b < HB; ++b) bias0[b] = b0p[wave * C1 + 32 * b + ln];
#pragma unroll
  for (int b = 0; b < NB; ++b) bias1[b] = b1p[wave * C2 + 32 * b + ln];
  // (every ring load is UNCONDITIONAL - past the end it re-reads the last sub-tile, an L1 hit nobody consumes: with an `if`
  // around it hipcc loads into temporaries and copies them into the ring behind an `s_waitcnt vmcnt`)
#pragma unroll
  for (int u = 0; u < PF; ++u) load_w(st[u], p.w0, H, C1, KT1, u < nt1 ? u : nt1 - 1);
  __builtin_amdgcn_sched_barrier(0);                         // nothing that needs the ids moves above the prefetches
  fetch_rows(0);
#pragma unroll
  for (int b = 0; b < HB; ++b) bias0[b] = p.b0 != nullptr ? bias0[b] : 0.f;
#pragma unroll
  for (int b = 0; b < NB; ++b) bias1[b] = p.b1 != nullptr ? bias1[b] : 0.f;
  for (int f0 = 4 * 256; f0 < total; f0 += 4 * 256) {        // (K0 > 128: further rounds)
    request_ids(f0);
    fetch_rows(f0);
  }
  __syncthreads();                                           // the input tile is complete
  TSTAMP(1);

  // ================= layer 0: h[32, H] = x[32, K0] @ W0[K0, H] =================
  f32x16 acc[HB > NB ? HB : NB];
#pragma unroll
  for (int b = 0; b < HB; ++b)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[b][i] = 0.f;
  for (int t0 = 0; t0 < nt1; t0 += PF) {
#pragma unroll
    for (int u = 0; u < PF; ++u) {
      const int t = t0 + u;
      if (t < nt1) {                                         // workgroup-uniform
        float* T = WP;
        store_w(st[u], T, C1, L1);                           // (waits for ring slot u only)
        load_w(st[u], p.w0, H, C1, KT1, t + PF < nt1 ? t + PF : nt1 - 1);
        asm volatile("" ::: "memory");                       // the wave reads back what its own lanes have just written: program order
#pragma unroll
        for (int g = 0; g < KT1 / 8; ++g) {
          const f32x4 a4 = *reinterpret_cast<const f32x4*>(XT + ln * LX + t * KT1 + 8 * g + 4 * hh);
#pragma unroll
          for (int b = 0; b < HB; ++b) {
            const float* bp = T + (8 * g + 4 * hh) * L1 + 32 * b + ln;
            const f32x4 b4 = f32x4{bp[0], bp[L1], bp[2 * L1], bp[3 * L1]};
#pragma unroll
            for (int s = 0; s < 4; ++s) acc[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[s], b4[s], acc[b], 0, 0, 0);
          }
        }
        asm volatile("" ::: "memory");
      }
    }
  }
  TSTAMP(2);
  // layer 1's first weight sub-tiles are requested now: they land under the epilogue below
  constexpr int nt2 = H / KT2;
#pragma unroll
  for (int u = 0; u < PF; ++u) load_w(st[u], p.w1, N1, C2, KT2, u < nt2 ? u : nt2 - 1);

  // ---- epilogue of layer 0: bias, ReLU, dropout; h to LDS (layer 1's A operand), to HBM (the backward pass), sign bits ----
#pragma unroll
  for (int b = 0; b < HB; ++b) {
    const int col = wave * C1 + 32 * b + ln;
    const float bias = bias0[b];
    // sign bits: `v > 0` of register reg over the wave IS the two words (low half = tile row acc_row(reg, 0), high half =
    // acc_row(reg, 1), bit = column) - the compare's lane mask goes straight into its lane with v_writelane.  (As
    // `posbits |= ...; ... if (lane == k) myword = ballot` each of the 32 words cost a compare, selects and hazard nops: ~400 of
    // this epilogue's ~670 instructions per wave, r03 ISA.  Rows past M get bits too; their words are never stored.)
    uint32_t myword = 0u;
    auto rows = [&](auto with_bits) {                        // (two copies of the loop: the test is uniform, not per register)
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const int row = tt::acc_row(reg, hh);
        [[maybe_unused]] const int64_t m = m0 + row;
        float v = fmaxf(acc[b][reg] + bias, 0.f);
        if constexpr (DROP) {
          const uint64_t hsh = tt::splitmix(p.drop_key + pb.drop_offset + (uint64_t)m * (uint64_t)H + (uint64_t)col);
          v = ((uint32_t)(hsh >> 40) < pb.drop_p24) ? 0.f : v * pb.drop_scale;
        }
        XH[row * LH + col] = v;
        if constexpr (decltype(with_bits)::value) {
          const uint64_t bal = __builtin_amdgcn_ballot_w64(v > 0.f);
          tt::writelane(myword, (uint32_t)bal, tt::acc_row(reg, 0));
          tt::writelane(myword, (uint32_t)(bal >> 32), tt::acc_row(reg, 1));
        }
      }
    };
    if (p.h_bits != nullptr) {
      rows(std::true_type{});
      const int64_t mw = m0 + lane;
      if (lane < 32 && mw < pb.M) p.h_bits[mw * (H / 32) + (wave * C1 + 32 * b) / 32] = myword;
    } else {
      rows(std::false_type{});
    }
  }
  TSTAMP(7);
  __syncthreads();                                           // the hidden tile is complete
  // h to HBM (the backward pass reads it) FROM THE LDS TILE, 16 bytes per lane: a wave-instruction stores four whole 1 KB rows.
  // (Stored from the accumulators - 4 bytes per lane, two 128-byte row pieces per instruction, 32 instructions per wave -
  // this epilogue took 6.2 us of the launch's 28.6, the 16 stores of y another 3.7: store-ISSUE-bound, r03 stamps.)
  {
    constexpr int c4n = H / 4;
#pragma unroll
    for (int j = 0; j < RB * c4n / 256; ++j) {
      const int f = tid + 256 * j;
      const int row = f / c4n, c4 = f - row * c4n;
      const int64_t m = m0 + row;
      if (m < pb.M) *reinterpret_cast<f32x4*>(p.h + m * H + 4 * c4) = *reinterpret_cast<const f32x4*>(XH + row * LH + 4 * c4);
    }
  }
  TSTAMP(3);

  // ================= layer 1: y[32, N1] = h[32, H] @ W1[H, N1] =================
#pragma unroll
  for (int b = 0; b < NB; ++b)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[b][i] = 0.f;
  for (int t0 = 0; t0 < nt2; t0 += PF) {
#pragma unroll
    for (int u = 0; u < PF; ++u) {
      const int t = t0 + u;
      if (t < nt2) {
        float* T = WP;
        store_w(st[u], T, C2, L2);
        load_w(st[u], p.w1, N1, C2, KT2, t + PF < nt2 ? t + PF : nt2 - 1);
        asm volatile("" ::: "memory");
#pragma unroll
        for (int g = 0; g < KT2 / 8; ++g) {
          const f32x4 a4 = *reinterpret_cast<const f32x4*>(XH + ln * LH + t * KT2 + 8 * g + 4 * hh);
#pragma unroll
          for (int b = 0; b < NB; ++b) {
            const float* bp = T + (8 * g + 4 * hh) * L2 + 32 * b + ln;
            const f32x4 b4 = f32x4{bp[0], bp[L2], bp[2 * L2], bp[3 * L2]};
#pragma unroll
            for (int s = 0; s < 4; ++s) acc[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[s], b4[s], acc[b], 0, 0, 0);
          }
        }
        asm volatile("" ::: "memory");
      }
    }
  }
  TSTAMP(4);
  // y: accumulators -> the (now dead) hidden tile's place in LDS -> 16-byte stores, as for h
  constexpr int LY = N1 + 4;
  __syncthreads();                                           // every wave has read its last hidden-tile fragment
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    const int col = wave * C2 + 32 * b + ln;
    const float bias = bias1[b];
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) XH[tt::acc_row(reg, hh) * LY + col] = acc[b][reg] + bias;
  }
  __syncthreads();
  {
    constexpr int c4n = N1 / 4;
#pragma unroll
    for (int j = 0; j < RB * c4n / 256; ++j) {
      const int f = tid + 256 * j;
      const int row = f / c4n, c4 = f - row * c4n;
      const int64_t m = m0 + row;
      if (m < pb.M) *reinterpret_cast<f32x4*>(p.y + m * N1 + 4 * c4) = *reinterpret_cast<const f32x4*>(XH + row * LY + 4 * c4);
    }
  }
  // the list entries leave at the very END of the kernel: in front of the first barrier its fence would wait for their
  // acknowledgement (a memory round trip on the kernel's longest dependency chain), and right behind it hipcc still parks a
  // `s_waitcnt vmcnt(0)` in front of the k loop (r04 ISA); here nothing follows that could wait for them
  if (lists && emit && bslot < p.bk_cap)                       // (a full list keeps counting: the optimizer falls back to its scan)
    p.bk_pairs[(size_t)bgrp * p.bk_cap + bslot] = (uint64_t)blk | ((uint64_t)((uint32_t)brow & 0xffffu) << 32) | ((uint64_t)(p.bk_gen & 0xffffu) << 48);
  TSTAMP(5);
#ifdef TT_TOWER_STAMPS
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  TSTAMP(6);
#endif
}

uint64_t dropout_stream_key(uint64_t seed, uint64_t tensor_id) {       // (the same key derivation as csrc/gemm.hip)
  return tt::splitmix_host(tt::splitmix_host(seed) ^ (tensor_id * 0xD6E8FEB86659FD93ull));
}

}  // namespace

extern "C" int32_t tt_tower_fwd2_supported(int64_t m, int32_t k0, int32_t h, int32_t n1) {
  return (m > 0 && k0 >= 32 && k0 % 32 == 0 && k0 <= 512 && (h == 128 || h == 256) && (n1 == 128 || n1 == 256)) ? 1 : 0;
}

extern "C" int tt_tower_fwd2_batched_f32(const tt_dense_fwd_args* layer0, const tt_dense_fwd_args* layer1, int32_t n_probs, int64_t m,
                                         int32_t k0, int32_t h, int32_t n1, float drop_rate, uint64_t seed, uint64_t counter_offset,
                                         tt_stream_t stream_) {
  TT_REQUIRE(layer0 != nullptr && layer1 != nullptr && n_probs >= 1 && n_probs <= 2, "tt_tower_fwd2_batched_f32: 1 or 2 towers");
  TT_REQUIRE(drop_rate >= 0.f && drop_rate < 1.f, "tt_tower_fwd2_batched_f32: drop_rate must be in [0,1)");
  if (!tt_tower_fwd2_supported(m, k0, h, n1))
    return tt::fail(TT_ERR_UNSUPPORTED, "tt_tower_fwd2_batched_f32: shapes m=%lld k0=%d h=%d n1=%d (need k0 %% 32 == 0, k0 <= 512, h and n1 in {128, 256})",
                    (long long)m, k0, h, n1);
  Tower2Batch pb{};
  pb.M = m; pb.K0 = k0; pb.H = h; pb.N1 = n1;
  pb.blocks_per_prob = (int)((m + RB - 1) / RB);
  TT_REQUIRE((int64_t)pb.blocks_per_prob * n_probs <= 0x7fffffff, "tt_tower_fwd2_batched_f32: grid too large");
  const bool gather = layer0[0].lookup.ids != nullptr;
  for (int i = 0; i < n_probs; ++i) {
    const tt_dense_fwd_args& a = layer0[i];
    const tt_dense_fwd_args& b = layer1[i];
    TT_REQUIRE((a.lookup.ids != nullptr) == gather, "tt_tower_fwd2_batched_f32: the lookup must be given for all towers or for none");
    TT_REQUIRE(a.w && a.y && b.w && b.y, "tt_tower_fwd2_batched_f32: null pointer");
    TT_REQUIRE(b.x == nullptr || b.x == a.y, "tt_tower_fwd2_batched_f32: layer 1's input must be layer 0's output");
    TT_REQUIRE(b.lookup.ids == nullptr, "tt_tower_fwd2_batched_f32: only layer 0 takes a lookup");
    TT_REQUIRE(tt::aligned16(a.x) && tt::aligned16(a.w) && tt::aligned16(a.y) && tt::aligned16(b.w) && tt::aligned16(b.y),
               "tt_tower_fwd2_batched_f32: pointers must be 16-byte aligned");
    Tower2Args& t = pb.a[i];
    t.x = a.x; t.w0 = a.w; t.b0 = a.b; t.h = a.y; t.h_bits = a.relu_bits;
    t.w1 = b.w; t.b1 = b.b; t.y = b.y;
    if (gather) {
      const tt_dense_lookup& lk = a.lookup;
      TT_REQUIRE(lk.table != nullptr && lk.table_rows > 0, "tt_tower_fwd2_batched_f32: lookup needs a table");
      TT_REQUIRE(tt::aligned16(lk.table) && tt::aligned16(lk.table2), "tt_tower_fwd2_batched_f32: lookup tables must be 16-byte aligned");
      TT_REQUIRE((lk.table2 == nullptr) == (lk.ids2 == nullptr), "tt_tower_fwd2_batched_f32: lookup.table2 and lookup.ids2 go together");
      t.table = lk.table; t.ids = lk.ids; t.table_rows = lk.table_rows;
      t.table2 = lk.table2; t.ids2 = lk.ids2; t.table2_rows = lk.table2_rows; t.oob_flag = lk.oob_flag;
      const tt_id_buckets& bk = lk.buckets;
      if (bk.pairs != nullptr) {
        TT_REQUIRE(bk.counts != nullptr && bk.groups >= 1 && bk.width >= 1u && bk.cap >= 1 && m <= 65536,
                   "tt_tower_fwd2_batched_f32: lookup.buckets: counts / groups / width / cap must be set (and m <= 65536)");
        t.bk_counts = bk.counts; t.bk_pairs = bk.pairs; t.bk_width = bk.width; t.bk_groups = (uint32_t)bk.groups;
        t.bk_cap = (uint32_t)bk.cap; t.bk_gen = bk.gen;
        t.bk_magic = (uint32_t)((((uint64_t)1 << 32) / bk.width) + 1u);
      }
    } else {
      TT_REQUIRE(a.x != nullptr, "tt_tower_fwd2_batched_f32: null input x (and no lookup)");
    }
    if (drop_rate > 0.f) t.drop_key = dropout_stream_key(seed, a.dropout_tensor_id);
  }
  const bool drop = drop_rate > 0.f;
  if (drop) {
    pb.drop_p24 = (uint32_t)((double)drop_rate * 16777216.0 + 0.5);
    pb.drop_scale = 1.0f / (1.0f - drop_rate);
    pb.drop_offset = counter_offset;
  }
  const int c1 = h / 4, c2 = n1 / 4;
  const int wb1 = (1024 / c1) * (c1 + 4), wb2 = (1024 / c2) * (c2 + 4);
  // hidden tile + input tile + one private weight buffer per wave (cfg3: 33.3 + 16.9 + 18.4 = 68.6 KB: two workgroups per CU)
  int lds = (RB * (h + 4) + RB * (k0 + 4) + 4 * (wb1 > wb2 ? wb1 : wb2)) * 4;
  if (const char* e = std::getenv("TT_TOWER_LDS_KB")) {        // experiment: a larger request caps the workgroups per CU
    const int want = std::atoi(e) * 1024;
    if (want > lds && want <= 160 * 1024) lds = want;
  }
  hipStream_t stream = tt::as_stream(stream_);
  const unsigned blocks = (unsigned)(pb.blocks_per_prob * n_probs);
  auto go = [&](auto kern) -> int {
    if (lds > 64 * 1024 &&
        hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess)
      return tt::fail(TT_ERR_LAUNCH, "tt_tower_fwd2_batched_f32: hipFuncSetAttribute(LDS %d) failed", lds);
    tt::launch("dense_fwd", kern, dim3(blocks), dim3(256), lds, stream, pb);
    return tt::check_launch("tt_tower_fwd2_batched_f32");
  };
  const int hb = h / 128, nb = n1 / 128;
  if (hb == 2 && nb == 1) return drop ? go(tower_fwd2_kernel<2, 1, true>) : go(tower_fwd2_kernel<2, 1, false>);
  if (hb == 1 && nb == 1) return drop ? go(tower_fwd2_kernel<1, 1, true>) : go(tower_fwd2_kernel<1, 1, false>);
  if (hb == 2 && nb == 2) return drop ? go(tower_fwd2_kernel<2, 2, true>) : go(tower_fwd2_kernel<2, 2, false>);
  return drop ? go(tower_fwd2_kernel<1, 2, true>) : go(tower_fwd2_kernel<1, 2, false>);
}

#ifdef TT_TOWER_STAMPS
extern "C" int tt_debug_tower_stamps(unsigned long long* host_out, int n) {
  return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_tstamps), sizeof(unsigned long long) * (size_t)n) == hipSuccess ? 0 : 2;
}
#endif
