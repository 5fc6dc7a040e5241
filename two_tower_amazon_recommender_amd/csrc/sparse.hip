// K2 — fused sparse optimizer on embedding rows.
//
// plan  : stable LSD radix sort of (id, position) by id — csrc/sort.hip (one launch, LDS-resident; only the ids are
//         needed, so it runs before / beside the forward pass).
// apply : one group of dim/4 lanes per sorted slot.  A slot that starts a run of equal ids walks the
//         run, summing the gradient rows in ascending position order (the IndexedSlices
//         de-duplication Keras performs before the update; bitwise reproducible), then
//         read-modify-writes the table row (and the Adagrad accumulator row) once.  Runs are cut
//         into pieces at global multiples of 64 sorted slots so that a hot id is summed by many lane
//         groups in parallel (pieces sequential inside; the LAST piece of a run to finish — an arrival ticket per
//         run — adds the pieces in index order and applies the update: order fixed, no second launch);
//         a run inside one 64-slot block is a plain sequential sum, bit-equal to np.add.at.
//         No float atomics: Adagrad is non-linear in g, so duplicates MUST be summed first.
// HBM-bound.  Algorithmic bytes per distinct row: 4*dim (grad) + 4*dim (row read) + 4*dim (row write)
// [+ 8*dim accumulator read/write for Adagrad] + 12 (sorted id + position).
#include "common.h"
#include "dense_update_body.h"
#include "part_sort.h"
#include <cstdlib>
#include <cstring>

namespace {

using tt::f32x4;

constexpr int kPiece = 64;   // sorted slots per piece: runs are cut at global multiples of 64 slots

constexpr int kMaxSparseTables = 3;   // user, item, hashed category
struct ApplyArgs {
  float* table[kMaxSparseTables];
  float* accum[kMaxSparseTables];
  const float* grads[kMaxSparseTables];
  const int64_t* sorted_ids[kMaxSparseTables];
  const int32_t* order[kMaxSparseTables];
  int64_t rows[kMaxSparseTables];
  // per-table piece workspace (tt_sparse_apply_workspace_bytes): sums of the pieces of runs that cross a 64-slot boundary
  float* p_sum[kMaxSparseTables];      // [nblk][dim]  first piece of a run that continues past its block (head in block j)
  float* s_sum[kMaxSparseTables];      // [nblk][dim]  piece starting exactly at slot 64*j
  int32_t* p_flag[kMaxSparseTables];   // [nblk]       arrival ticket of the deferred run whose head is in block j (0 between launches)
  // n_ids > tt::kPartSortMaxIds only (the fused optimizer's long-list form, part_sort_global): 2 x n words, n keys, n positions
  uint64_t* big_pairs[kMaxSparseTables];
  uint32_t* big_keys[kMaxSparseTables];
  uint16_t* big_pos[kMaxSparseTables];
};

// NT: the updated rows leave with nontemporal stores.  Only the fused optimizer launch's rank-free path uses them (fast_apply:
// the end-of-kernel write-back has less left to do, cfg3 launch 15.1 -> 14.8 us in r03's A/B); for the large-list kernel
// the same stores cost 5 % (1M ids: 351 -> 370 us SGD/U, 512 -> 544 us Adagrad/U), so everything else stores normally.
template <int OPT, bool NT = false>
__device__ __forceinline__ void update_store(f32x4* __restrict__ table, f32x4* __restrict__ accum, int64_t off, f32x4 w, f32x4 acc,
                                             const f32x4& g, float lr, float eps) {
  if constexpr (OPT == TT_OPT_SGD) {
#pragma unroll
    for (int e = 0; e < 4; ++e) w[e] = __fsub_rn(w[e], __fmul_rn(lr, g[e]));
  } else {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      acc[e] = __fadd_rn(acc[e], __fmul_rn(g[e], g[e]));
      const float den = sqrtf(__fadd_rn(acc[e], eps));
      w[e] = __fsub_rn(w[e], __fdiv_rn(__fmul_rn(lr, g[e]), den));
    }
    if constexpr (NT) __builtin_nontemporal_store(acc, accum + off); else accum[off] = acc;
  }
  if constexpr (NT) __builtin_nontemporal_store(w, table + off); else table[off] = w;
}

template <int OPT>
__device__ __forceinline__ void update_row(f32x4* __restrict__ table, f32x4* __restrict__ accum, int64_t off, const f32x4& g,
                                           float lr, float eps) {
  const f32x4 w = table[off];
  f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
  if constexpr (OPT != TT_OPT_SGD) acc = accum[off];
  update_store<OPT>(table, accum, off, w, acc, g, lr, eps);
}

// Piece sums travel between lane groups of DIFFERENT workgroups inside one launch.  They are stored WRITE-THROUGH
// (sc1: two 8-byte agent-scope relaxed stores per float4) and read back with sc1 loads behind an agent-scope acquire, so
// the producer needs no release fence.  (r02 used plain stores + fence(release, agent) = `buffer_wbl2 sc1`, which writes
// back EVERY dirty line of the XCD's L2 - in a kernel whose whole job is to dirty L2 with table rows.  One fence per piece
// of a multi-piece run: ~1,600 of them at 1M uniform ids, ~100x more with power-law ids; the launch went from 350 to 690 us
// (U) and 634 to 1,992 us (Z) against r01's two-launch form - VERDICT r02.  TT_SPARSE_RELEASE_FENCE=1 rebuilds that form
// for the A/B in profiles/r03_sparse_apply_ab.jsonl.)
#ifndef TT_SPARSE_RELEASE_FENCE
#define TT_SPARSE_RELEASE_FENCE 0
#endif
__device__ __forceinline__ void store_piece(f32x4* p, const f32x4& v) {
#if TT_SPARSE_RELEASE_FENCE
  *p = v;
#else
  unsigned long long* q = reinterpret_cast<unsigned long long*>(p);
  const unsigned long long lo = ((unsigned long long)__float_as_uint(v[1]) << 32) | __float_as_uint(v[0]);
  const unsigned long long hi = ((unsigned long long)__float_as_uint(v[3]) << 32) | __float_as_uint(v[2]);
  __hip_atomic_store(q, lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);          // global_store_dwordx2 ... sc1
  __hip_atomic_store(q + 1, hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
}
__device__ __forceinline__ f32x4 load_piece(const f32x4* p) {
#if TT_SPARSE_RELEASE_FENCE
  return *p;
#else
  unsigned long long* q = reinterpret_cast<unsigned long long*>(const_cast<f32x4*>(p));
  const unsigned long long lo = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // global_load_dwordx2 ... sc1
  const unsigned long long hi = __hip_atomic_load(q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return f32x4{__uint_as_float((uint32_t)lo), __uint_as_float((uint32_t)(lo >> 32)), __uint_as_float((uint32_t)hi),
               __uint_as_float((uint32_t)(hi >> 32))};
#endif
}

// A piece of a run that spans several 64-slot blocks (rare): find the run's extent, publish, take a ticket; the last
// arriver adds the pieces in index order and applies the update.  (TT_SPARSE_TAIL_NOINLINE=1 keeps it out of line - an
// A/B hook: as a real call it costs the callers 32-76 bytes of scratch for the registers saved around it.)
#ifndef TT_SPARSE_TAIL_NOINLINE
#define TT_SPARSE_TAIL_NOINLINE 0
#endif
#if TT_SPARSE_TAIL_NOINLINE
#define TT_TAIL_ATTR __attribute__((noinline))
#else
#define TT_TAIL_ATTR __forceinline__
#endif
template <int OPT>
__device__ TT_TAIL_ATTR void finish_run_piece(f32x4* __restrict__ table, f32x4* __restrict__ accum, const int64_t* __restrict__ sid,
                                              const f32x4* P, const f32x4* S, int32_t* p_flag, const int64_t k, const int64_t e,
                                              const int64_t pend, const int64_t id, const bool run_head, const bool continues,
                                              const int dim4, const int lpr_log2, const int64_t n_ids, const float lr, const float eps) {
  const int lpr = 1 << lpr_log2;
  const int l = threadIdx.x & (lpr - 1);
  int64_t first = k;                              // first slot of the run (lower bound of id in the sorted ids)
  if (!run_head) {
    int64_t lo = 0, hi = k;                       // sid[k] == id, and the slot before a boundary piece holds id too
    while (lo < hi) {
      const int64_t mid = (lo + hi) >> 1;
      if (sid[mid] < id) lo = mid + 1; else hi = mid;
    }
    first = lo;
  }
  int64_t last = e;                               // one past the last slot of the run
  if (continues) {
    int64_t lo = pend + 1, hi = n_ids;
    while (lo < hi) {
      const int64_t mid = (lo + hi) >> 1;
      if (sid[mid] <= id) lo = mid + 1; else hi = mid;      // sentinel / valid ids ascend; nothing sorts below 0
    }
    last = lo;
  }
  const int64_t jh = first / kPiece, jl = (last - 1) / kPiece;
  const int npieces = (int)(jl - jh + 1);
  // publish: this lane group's write-through stores (all in this wave) have left the CU before the ticket is drawn
#if TT_SPARSE_RELEASE_FENCE
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
#endif
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  int ticket = 0;
  // (the ticket stays a RELAXED add behind the `s_waitcnt vmcnt(0)` above: the piece sums left as write-through sc1 stores and
  // have been acknowledged.  r04 tried the textbook form - __hip_atomic_fetch_add(..., __ATOMIC_RELEASE, agent), ADVICE r03 -:
  // hipcc puts the release in front of it as `buffer_wbl2 sc1`, the write-back of every dirty L2 line of the XCD that r03
  // had removed, and the 1M-id lines went straight back to r02's: 347 / 519 / 522 / 644 us -> 694 / 909 / 2018 / 2229 us
  // (profiles/r04_sparse_ticket_ab.jsonl).  gfx942 / gfx950 only: the library builds for nothing else.)
  if (l == 0) ticket = atomicAdd(&p_flag[jh], 1);
  ticket = __shfl(ticket, (int)(threadIdx.x & 63u & ~(unsigned)(lpr - 1)));
  if (ticket != npieces - 1) return;
  // last arriver: every piece of the run is published
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (l == 0) p_flag[jh] = 0;                     // leave the workspace zeroed for the next call
  // (r04: FOUR piece sums requested before the first is added - the adds stay in index order, the loads do not depend on
  // them.  One sc1 load per round trip, as hipcc schedules the plain loop, made the last arriver of a long run a chain of
  // dependent L2 reads: the hottest id of a 1M-id power-law batch is 18,000 slots = 281 pieces = ~250 us of a 522 us launch.
  // Four, not eight: they reuse the registers of the walk's four gradient rows; more would cost the kernel a wave of occupancy.)
  for (int c = l; c < dim4; c += lpr) {
    f32x4 g = load_piece(P + jh * dim4 + c);
    int64_t m = jh + 1;
    while (m + 3 <= jl) {
      f32x4 s[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) s[u] = load_piece(S + (m + u) * dim4 + c);
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int q = 0; q < 4; ++q) g[q] = __fadd_rn(g[q], s[u][q]);
      m += 4;
    }
    for (; m <= jl; ++m) {
      const f32x4 s = load_piece(S + m * dim4 + c);
#pragma unroll
      for (int q = 0; q < 4; ++q) g[q] = __fadd_rn(g[q], s[q]);
    }
    update_row<OPT>(table, accum, id * dim4 + c, g, lr, eps);
  }
}

// Summation order (the oracle's dedup_sum restates it): a run of equal ids occupies consecutive sorted slots; it is cut
// into PIECES at global multiples of 64 slots; every piece is summed sequentially in slot (= ascending position) order,
// then the pieces are added in order.  A run inside one 64-slot block — the common case — is a plain sequential sum
// (bit-equal to np.add.at).  A hot id repeated thousands of times is summed by many lane groups in parallel.
//
// A run that crosses a block boundary is finished INSIDE this launch: every piece stores its sum write-through (p_sum /
// s_sum of its block), drains its stores (s_waitcnt vmcnt(0)) and takes a ticket on the run's counter p_flag[head block];
// the piece that draws the last ticket acquires, adds the pieces in INDEX order (never arrival order: bitwise
// reproducible) and applies the update.  A block holds at most one run head that continues past its end, so the counter is
// unambiguous; the last arriver leaves it at 0 for the next launch.
template <int OPT, bool AHEAD = true>
__device__ __forceinline__ void sparse_apply_body(const ApplyArgs& a, const int t, const int64_t bx, int dim4, int lpr_log2,
                                                  int64_t n_ids, float lr, float eps) {
  f32x4* __restrict__ table = reinterpret_cast<f32x4*>(a.table[t]);
  f32x4* __restrict__ accum = reinterpret_cast<f32x4*>(a.accum[t]);
  const f32x4* __restrict__ grads = reinterpret_cast<const f32x4*>(a.grads[t]);
  const int64_t* __restrict__ sid = a.sorted_ids[t];
  const int32_t* __restrict__ order = a.order[t];
  const int64_t rows = a.rows[t];

  const int lpr = 1 << lpr_log2;
  const int groups = 256 >> lpr_log2;
  const int64_t k = bx * groups + (threadIdx.x >> lpr_log2);   // sorted slot
  const int l = threadIdx.x & (lpr - 1);
  if (k >= n_ids) return;
  // One round trip for everything the slot's position gives: its id, both neighbours, its batch position - then ONE more for
  // the gradient row together with the table (accumulator) row.  (Through r02 five dependent ones: id and left neighbour,
  // right neighbour, batch position, gradient row, table row; the 1M-id Adagrad line was 11 % behind r01h.)
  const int64_t id = sid[k];
  const int64_t id_left = sid[k > 0 ? k - 1 : 0], id_right = sid[k + 1 < n_ids ? k + 1 : n_ids - 1];
  const int32_t ord0 = order[k];
  const bool run_head = (k == 0) || (id_left != id);
  const bool boundary = (k % kPiece) == 0;
  if (!run_head && !boundary) return;             // inside a piece
  if (id < 0 || id >= rows) return;               // out-of-range / padding ids (the plan's sentinel) are skipped
  const int lc = l < dim4 ? l : dim4 - 1;
  // (SGD: the table row with the gradient row.  Adagrad has no registers left for two more rows at 8 waves per SIMD - with
  // them it spills 20-72 B per lane and the 1M-id lines get 6 % slower, at 6 waves they are level - so its rows are read
  // where they are needed; r03 A/B on one box, SGD U/Z 371/544 -> 354/523 us.)
  // (AHEAD = false: inside optimizer_kernel, whose dense half needs the registers too)
  constexpr bool ROW_AHEAD = AHEAD && OPT == TT_OPT_SGD;
  const f32x4 g_first = grads[(int64_t)ord0 * dim4 + lc];
  f32x4 w_first = f32x4{0.f, 0.f, 0.f, 0.f};
  if constexpr (ROW_AHEAD) w_first = table[id * dim4 + lc];
  int64_t pend = (k / kPiece + 1) * kPiece;       // this piece ends at the next 64-slot boundary at the latest
  if (pend > n_ids) pend = n_ids;
  int64_t e = k + 1;                              // end of this piece: known from the right neighbour for the usual run of length 1,
  if (e < pend && id_right == id) {               // else a binary search (sorted ids): <= 6 dependent loads per piece
    int64_t lo = e + 1, hi = pend;
    while (lo < hi) {
      const int64_t mid = (lo + hi) >> 1;
      if (sid[mid] == id) lo = mid + 1; else hi = mid;
    }
    e = lo;
  }
  const bool continues = (e == pend) && (pend < n_ids) && (sid[pend] == id);
  const bool whole = run_head && !continues;      // the usual case: the whole run is this piece
  const int64_t blk = k / kPiece;

  for (int c = l; c < dim4; c += lpr) {
    f32x4 g = c == l ? g_first : grads[(int64_t)ord0 * dim4 + c];
    // sequential walk over [k, e); four independent row loads in flight (more would cost a wave of occupancy,
    // which the random single-row case — almost every slot — needs more)
    int64_t j = k + 1;
    while (j + 3 < e) {
      f32x4 r[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) r[u] = grads[(int64_t)order[j + u] * dim4 + c];
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int q = 0; q < 4; ++q) g[q] = __fadd_rn(g[q], r[u][q]);
      j += 4;
    }
    while (j < e) {
      const f32x4 g1 = grads[(int64_t)order[j] * dim4 + c];
#pragma unroll
      for (int e = 0; e < 4; ++e) g[e] = __fadd_rn(g[e], g1[e]);
      ++j;
    }
    if (whole) {                                                          // whole run summed: fused update
      if (ROW_AHEAD && c == l) update_store<OPT>(table, accum, id * dim4 + c, w_first, w_first, g, lr, eps);
      else update_row<OPT>(table, accum, id * dim4 + c, g, lr, eps);
    } else if (run_head) {
      store_piece(reinterpret_cast<f32x4*>(a.p_sum[t]) + blk * dim4 + c, g);   // first piece of a long run
    } else {
      store_piece(reinterpret_cast<f32x4*>(a.s_sum[t]) + blk * dim4 + c, g);   // later piece (starts at slot 64*blk)
    }
  }
  if (whole) return;
  finish_run_piece<OPT>(table, accum, sid, reinterpret_cast<const f32x4*>(a.p_sum[t]), reinterpret_cast<const f32x4*>(a.s_sum[t]),
                        a.p_flag[t], k, e, pend, id, run_head, continues, dim4, lpr_log2, n_ids, lr, eps);
}

// TT_OPT_PREFETCH=1: the fused optimizer touches every line of a row the moment the scan finds its id (optimizer_ids_kernel).
// Built, measured three ways on cfg3 and left OFF: the rows do arrive earlier (apply phase 4.9 -> 3.6-3.8 us between the
// stamps) but issuing the touches costs the scan more than that - a load instruction occupies the address path for a whole
// wave however few lanes are active, and the per-id broadcast / address arithmetic runs on every wave: launch 12.6 us ->
// 14.1 (per lane and line, inside the classification loop: its in-order vmcnt waits then also wait for the touches),
// 13.7 (same, after the loop), 13.25 us (one instruction per id, 12 lanes = the row's lines); =2, one 4-byte touch of the
// table row per id to start its address translation early: 12.8-13.0 -> 13.4 us (the touch returns no sooner than the row
// loads issued 1 us later: the rows' 4.9 us is a burst of 17 MB of random 512-byte rows draining, not a page walk).
// profiles/r03_optimizer_ab.txt
#ifndef TT_OPT_PREFETCH
#define TT_OPT_PREFETCH 0
#endif
#ifndef TT_APPLY_WAVES
#define TT_APPLY_WAVES 8      // min waves per SIMD of the large-list kernels (8: 64 VGPRs, 6: 80)
#endif
template <int OPT>
__global__ __launch_bounds__(256, TT_APPLY_WAVES) void sparse_apply_kernel(ApplyArgs a, int dim4, int lpr_log2, int64_t n_ids, float lr,
                                                           float eps) {
  sparse_apply_body<OPT>(a, blockIdx.y, blockIdx.x, dim4, lpr_log2, n_ids, lr, eps);
}

// The whole optimizer step of the train step in ONE launch: blockIdx.y < n_tables -> the fused sparse update of table y,
// else dense segment y - n_tables (csrc/dense_update_body.h).  The two halves are independent and both memory-bound; as
// two launches they cost 7.5 + 9.0 us plus a launch boundary per step.
template <int OPT>
__global__ __launch_bounds__(256, TT_APPLY_WAVES) void optimizer_kernel(ApplyArgs a, int n_tables, int dim4, int lpr_log2, int64_t n_ids,
                                                         int64_t sparse_blocks, tt::SegTable tbl, int dense_blocks, float lr,
                                                         float eps) {
  if ((int)blockIdx.y < n_tables) {
    if ((int64_t)blockIdx.x < sparse_blocks) sparse_apply_body<OPT, false>(a, blockIdx.y, blockIdx.x, dim4, lpr_log2, n_ids, lr, eps);
  } else if ((int)blockIdx.x < dense_blocks) {
    tt::dense_update_body<OPT>(tbl.seg[blockIdx.y - n_tables], blockIdx.x, dense_blocks, 1, lr, eps);
  }
}

// ---- the optimizer step FROM THE RAW IDS: sort + apply in one workgroup (no plan launch, no sorted ids in HBM) --------
// Workgroup g of table t sorts the ids of its row range in LDS (csrc/part_sort.h) — that is: it knows the position
// `offset` of its first key in the table's sorted list and holds its m (local key, batch position) pairs in sorted
// order — and applies the update to exactly those rows itself: equal ids share a row range, so every run of the sorted
// list lies inside ONE workgroup and the arrival tickets of sparse_apply_body are not needed.  Summation order is the
// same as there, bit for bit: a run is cut into pieces at GLOBAL multiples of 64 sorted slots (offset + i), each piece is
// summed sequentially, the pieces are added in index order (phase B, after a workgroup barrier; piece sums travel
// through the same p_sum / s_sum workspace).
struct FusedTables {
  tt::PartTable part[kMaxSparseTables];
  int32_t cap;
  int32_t first[kMaxSparseTables + 1];        // flat workgroup index of table t's group 0; first[n_tables] = first dense block
  int32_t seg_first[TT_MAX_DENSE_SEGS + 1];   // flat index (from first[n_tables]) of segment s's block 0
  // row-range id lists filled by this step's forward lookup (tt_id_buckets, include/twotower_hip.h); bk_pairs[t] NULL = scan
  uint32_t* bk_counts[kMaxSparseTables];
  const uint64_t* bk_pairs[kMaxSparseTables];
  int32_t bk_cap;
  uint32_t bk_gen;
};

// Rows requested per lane group before any of them is finished (one memory round trip for the usual 2-3 slots per group)
constexpr int kRowsAhead = 4;
struct RowsAhead {
  f32x4 g[kRowsAhead], w[kRowsAhead], a[kRowsAhead];   // first float4 chunk (c = l) of: the pair's gradient row, its table row, its accumulator row
};

// The rank-by-counting path: called when the workgroup's unordered (key, position) list is complete, BEFORE it is ranked.
// Lane group grp requests the gradient row and the table (+ accumulator) row of the pairs grp, grp + ngroups, ... - their
// addresses need the key and the batch position only, not the rank -, so the rows' HBM round trip runs under the ranking
// (~2 us of LDS work) instead of after it.  (r02: requested after the ranking; per-workgroup stamps 5.5 -> 9.3 us for the
// apply phase, of which one HBM round trip.)
template <int OPT>
__device__ __forceinline__ void request_rows(const ApplyArgs& a, const int t, const uint32_t* K, const uint16_t* P, const uint32_t m,
                                             const uint32_t base_key, int dim4, int lpr_log2, RowsAhead& ra) {
  const f32x4* __restrict__ table = reinterpret_cast<const f32x4*>(a.table[t]);
  const f32x4* __restrict__ accum = reinterpret_cast<const f32x4*>(a.accum[t]);
  const f32x4* __restrict__ grads = reinterpret_cast<const f32x4*>(a.grads[t]);
  const int64_t rows = a.rows[t];
  const uint32_t ngroups = 1024u >> lpr_log2;
  const uint32_t grp = threadIdx.x >> lpr_log2;
  const int l = threadIdx.x & ((1 << lpr_log2) - 1);
#pragma unroll
  for (int r = 0; r < kRowsAhead; ++r) {
    const uint32_t i = grp + (uint32_t)r * ngroups;
    // (no zero-initialisation: only slots that were requested are consumed.  With one, hipcc merged "zero or loaded" through
    // a register copy placed right behind the load - an `s_waitcnt vmcnt` per slot inside this function, i.e. one HBM round
    // trip paid HERE: 2.1 us between the stamps around it instead of ~0.2, r03 call 3.)
    if (i < m && l < dim4) {
      const int64_t id = (int64_t)base_key + K[i];
      if (id < rows) {                                  // (the out-of-range sentinel has no row)
        ra.g[r] = grads[(int64_t)P[i] * dim4 + l];
        ra.w[r] = table[id * dim4 + l];
        if constexpr (OPT != TT_OPT_SGD) ra.a[r] = accum[id * dim4 + l];
      }
    }
  }
}

// The usual row range WITHOUT the ranking on the critical path (r03).  Which sorted slot a pair lands in matters only
// for runs of three or more equal ids (the order and the 64-slot pieces of their sums); a row touched ONCE is updated from
// its one gradient row whatever its slot, and for a row touched TWICE the sum is fadd(first, second) for either piece cut
// (one piece: g0 then += g1; two pieces: P = g0, S = g1, P + S).  So each lane group, for each of its <= kRowsAhead pairs of
// the UNORDERED list: requests the pair's gradient row and table (+ accumulator) row; while they are in flight, compares the
// pair's key with the whole list (lane l takes entries l, l + lpr, ...; one ballot per compare, the group's bits counted)
// to get n_eq = entries with this key, n_before = those at a smaller batch position, and the index of another entry with
// the key; then
//   n_eq == 1                      update the row from the requested registers
//   n_eq == 2, n_before == 0       fetch the partner's gradient row (L2: its own lane group has just requested it), add, update
//   n_eq == 2, n_before == 1       nothing (the partner does it)
//   n_eq >= 3                      left to the ranked path: bit r of the return value, *s_slow = 1
// No workgroup barrier, no LDS write-back, no dependence on the ranks: a workgroup without a triple (every workgroup, for
// uniform ids over >= 1M rows) is done when its rows have landed.  Needs dim4 <= lpr <= 32 and m <= kRowsAhead * ngroups.
template <int OPT>
__device__ __forceinline__ uint32_t fast_apply(const ApplyArgs& a, const int t, const uint32_t* K, const uint16_t* P, const uint32_t m,
                                               const uint32_t base_key, int dim4, int lpr_log2, float lr, float eps, int* s_slow) {
  f32x4* __restrict__ table = reinterpret_cast<f32x4*>(a.table[t]);
  f32x4* __restrict__ accum = reinterpret_cast<f32x4*>(a.accum[t]);
  const f32x4* __restrict__ grads = reinterpret_cast<const f32x4*>(a.grads[t]);
  const int64_t rows = a.rows[t];
  const uint32_t lpr = 1u << lpr_log2;
  const uint32_t ngroups = 1024u >> lpr_log2;
  const uint32_t grp = threadIdx.x >> lpr_log2;
  const uint32_t l = threadIdx.x & (lpr - 1u);
  const uint32_t gshift = (threadIdx.x & 63u) & ~(lpr - 1u);           // first lane of this lane group within its wave
  const uint32_t gmask = lpr >= 32u ? 0xffffffffu : ((1u << lpr) - 1u);
  constexpr int RP = kRowsAhead;
  uint32_t key[RP], pos[RP], n_eq[RP], n_bf[RP], jo[RP];
  bool live[RP];
  f32x4 g[RP], w[RP], ac[RP] = {};       // (ac: read only by Adagrad; zeroed so the SGD instantiation passes no indeterminate value)
  // Every lane ALWAYS loads (no branch around the requests: with one, hipcc merged "loaded or not" through register copies
  // placed right behind the loads, i.e. an `s_waitcnt vmcnt` - a whole HBM round trip - inside the request phase).  A lane
  // group without a pair r (i >= m), a pair with the out-of-range sentinel and the lanes past the row's end re-read pair
  // 0's rows / the row's last chunk: L1 hits, discarded.
  const uint32_t lc = l < (uint32_t)dim4 ? l : (uint32_t)dim4 - 1u;
#pragma unroll
  for (int r = 0; r < RP; ++r) {
    const uint32_t i = grp + (uint32_t)r * ngroups;
    const uint32_t ic = i < m ? i : 0u;
    const uint32_t kk = K[ic];
    pos[r] = P[ic];
    const int64_t id = (int64_t)base_key + kk;
    live[r] = i < m && id < rows;                                        // (the out-of-range sentinel has no row)
    key[r] = live[r] ? kk : 0xffffffffu;                                 // (no list entry has the all-ones local key)
    n_eq[r] = 0u; n_bf[r] = 0u; jo[r] = 0u;
    const int64_t off = (id < rows ? id : rows - 1) * dim4 + lc;
    g[r] = grads[(int64_t)pos[r] * dim4 + lc];
    w[r] = table[off];
    if constexpr (OPT != TT_OPT_SGD) ac[r] = accum[off];
  }
  // ---- the pair's key against the whole list, under the row loads ----
  for (uint32_t j0 = 0u; j0 < m; j0 += lpr) {
    const uint32_t j = j0 + l;
    const uint32_t kj = j < m ? K[j] : 0xfffffffeu;
    const uint32_t pj = j < m ? (uint32_t)P[j] : 0u;
#pragma unroll
    for (int r = 0; r < RP; ++r) {
      if (!__builtin_amdgcn_ballot_w64(live[r])) continue;               // (wave-uniform: no lane group of this wave has a pair r)
      const bool eq = kj == key[r];
      const uint32_t eb = (uint32_t)(__builtin_amdgcn_ballot_w64(eq) >> gshift) & gmask;
      const uint32_t bb = (uint32_t)(__builtin_amdgcn_ballot_w64(eq && pj < pos[r]) >> gshift) & gmask;
      n_eq[r] += (uint32_t)__popc(eb);
      n_bf[r] += (uint32_t)__popc(bb);
      const uint32_t self = grp + (uint32_t)r * ngroups - j0;            // this pair's own bit, when it is in this slice
      const uint32_t others = self < lpr ? (eb & ~(1u << self)) : eb;
      if (others != 0u) jo[r] = j0 + (uint32_t)__ffs((int)others) - 1u;
    }
  }
  // What each pair needs is known now, before any row has landed: the verdict on the ranked path (s_slow) is published and
  // the workgroup barrier taken HERE, after the evenly long compare loop - not behind the rows' HBM latency, where the
  // barrier would also cost the spread between the first and the last wave's data (1.3 us between the stamps, r03 call 7).
  // After it a wave only waits for its own rows, stores and - unless some id of the range occurs three times - leaves.
  uint32_t slow = 0u, put = 0u;
  uint32_t ppos[RP];                                                      // batch position of a pair's partner
#pragma unroll
  for (int r = 0; r < RP; ++r) {
    ppos[r] = 0u;
    if (!live[r]) continue;
    if (n_eq[r] >= 3u) { slow |= 1u << r; continue; }
    if (n_eq[r] == 2u && n_bf[r] != 0u) continue;                         // second of a pair: its partner sums and updates
    put |= 1u << r;
    if (n_eq[r] == 2u) ppos[r] = P[jo[r]];                                // (the list is read for the last time)
  }
  if (slow != 0u && l == 0u) *s_slow = 1;
  tt::lds_barrier();
  // two loops: first every pair's gradient is completed (a partner row fetched and added where there is one), then all the
  // rows are stored - a store issued between two pairs would sit in front of the next pair's `s_waitcnt vmcnt`
#pragma unroll
  for (int r = 0; r < RP; ++r) {
    if (((put >> r) & 1u) && n_eq[r] == 2u && l < (uint32_t)dim4) {
      const f32x4 g2 = grads[(int64_t)ppos[r] * dim4 + l];
#pragma unroll
      for (int q = 0; q < 4; ++q) g[r][q] = __fadd_rn(g[r][q], g2[q]);
    }
  }
#pragma unroll
  for (int r = 0; r < RP; ++r)
    if (((put >> r) & 1u) && l < (uint32_t)dim4)
      update_store<OPT, true>(table, accum, ((int64_t)base_key + key[r]) * dim4 + l, w[r], ac[r], g[r], lr, eps);
  return slow;
}

template <int OPT, int DBITS, bool BY_LIST, bool USE_RA>
__device__ __forceinline__ void apply_from_lds(const ApplyArgs& a, const int t, const uint32_t* K, const uint16_t* P, const uint16_t* R,
                                               const uint32_t m, const uint32_t offset, const uint32_t base_key, int dim4, int lpr_log2,
                                               float lr, float eps, int* s_multi, const RowsAhead& ra, const uint32_t todo) {
  f32x4* __restrict__ table = reinterpret_cast<f32x4*>(a.table[t]);
  f32x4* __restrict__ accum = reinterpret_cast<f32x4*>(a.accum[t]);
  const f32x4* __restrict__ grads = reinterpret_cast<const f32x4*>(a.grads[t]);
  const int64_t rows = a.rows[t];
  const int lpr = 1 << lpr_log2;
  const uint32_t ngroups = 1024u >> lpr_log2;
  const uint32_t grp = threadIdx.x >> lpr_log2;
  const int l = threadIdx.x & (lpr - 1);

  // what slot i has to do: 0 nothing (inside a piece / skipped id), else the end e of its piece
  auto piece = [&](uint32_t i, uint32_t& e, bool& run_head, bool& continues) -> bool {
    if (i >= m) return false;
    const uint32_t key = K[i];
    run_head = (i == 0u) || (K[i - 1u] != key);               // runs never cross a workgroup: slot 0 always starts one
    const uint32_t kg = offset + i;
    if (!run_head && (kg % kPiece) != 0u) return false;
    if ((int64_t)base_key + key >= rows) return false;        // the out-of-range sentinel is skipped
    uint32_t iend = i + (kPiece - kg % kPiece);
    if (iend > m) iend = m;
    e = i + 1u;
    while (e < iend && K[e] == key) ++e;
    continues = (e == iend) && (iend < m) && (K[iend] == key);
    return true;
  };

  // ---- phase A: every piece is summed; whole runs (the usual case) are applied at once.  The first gradient row AND the
  // table (accumulator) row of a lane group's first kRowsAhead slots have been requested together: either per unordered
  // pair before the ranking (BY_LIST: the pair i of the list sits in sorted slot R[i]) or, for a radix-sorted hot range,
  // here per sorted slot grp, grp + ngroups, ... ----
  bool multi = false;
  auto finish = [&](uint32_t i, uint32_t e, bool head, bool cont, bool pre, const f32x4& g0, const f32x4& w0, const f32x4& a0) {
    const int64_t id = (int64_t)base_key + K[i];
    const bool whole = head && !cont;
    const int64_t blk = (int64_t)(offset + i) / kPiece;
    for (int c = l; c < dim4; c += lpr) {
      const bool first = pre && c == l;
      f32x4 g = first ? g0 : grads[(int64_t)P[i] * dim4 + c];
      uint32_t j = i + 1u;
      while (j + 3u < e) {
        f32x4 r[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) r[u] = grads[(int64_t)P[j + u] * dim4 + c];
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int q = 0; q < 4; ++q) g[q] = __fadd_rn(g[q], r[u][q]);
        j += 4u;
      }
      while (j < e) {
        const f32x4 g1 = grads[(int64_t)P[j] * dim4 + c];
#pragma unroll
        for (int q = 0; q < 4; ++q) g[q] = __fadd_rn(g[q], g1[q]);
        ++j;
      }
      if (whole) {
        if (first) update_store<OPT>(table, accum, id * dim4 + c, w0, a0, g, lr, eps);
        else update_row<OPT>(table, accum, id * dim4 + c, g, lr, eps);
      } else if (head) {
        reinterpret_cast<f32x4*>(a.p_sum[t])[blk * dim4 + c] = g;
      } else {
        reinterpret_cast<f32x4*>(a.s_sum[t])[blk * dim4 + c] = g;
      }
    }
    multi = multi || !whole;
  };
  constexpr int RP = kRowsAhead;
  if constexpr (BY_LIST) {
    // every sorted slot is the rank of exactly one pair of the list: the lane group that requested pair i's rows finishes
    // slot R[i].  The usual slot - a run of ONE id, the row fits the lane group - is updated straight from the requested
    // registers; anything else (duplicates, rows wider than the lane group, pairs beyond kRowsAhead) takes the general walk,
    // which loads for itself (one instance of it: the requested registers are dead by then).
    const f32x4 z = f32x4{0.f, 0.f, 0.f, 0.f};
    uint32_t slow = todo;                  // !USE_RA: after fast_apply - only the pairs it left (bit r of todo), nothing requested ahead
    if constexpr (USE_RA) {
      slow = 0u;
#pragma unroll
      for (int r = 0; r < RP; ++r) {
        const uint32_t i = grp + (uint32_t)r * ngroups;
        if (i < m) {
          const uint32_t sl = R[i];
          uint32_t e = 0u;
          bool head = false, cont = false;
          if (piece(sl, e, head, cont)) {
            if (head && !cont && e == sl + 1u && dim4 <= lpr) {
              if (l < dim4) update_store<OPT>(table, accum, ((int64_t)base_key + K[sl]) * dim4 + l, ra.w[r], ra.a[r], ra.g[r], lr, eps);
            } else {
              slow |= 1u << r;
            }
          }
        }
      }
    }
    for (uint32_t i = grp; i < m; i += ngroups) {
      const uint32_t r = (i - grp) / ngroups;
      if (r < (uint32_t)RP && !((slow >> r) & 1u)) continue;
      const uint32_t sl = R[i];
      uint32_t e = 0u;
      bool head = false, cont = false;
      if (piece(sl, e, head, cont)) finish(sl, e, head, cont, false, z, z, z);
    }
  } else {
    f32x4 g0[RP], w0[RP], a0[RP];
    uint32_t ee[RP];
    bool act[RP], hd[RP], ct[RP];
#pragma unroll
    for (int r = 0; r < RP; ++r) {
      const uint32_t i = grp + (uint32_t)r * ngroups;
      ee[r] = 0u; hd[r] = false; ct[r] = false;
      g0[r] = f32x4{0.f, 0.f, 0.f, 0.f}; w0[r] = g0[r]; a0[r] = g0[r];
      act[r] = piece(i, ee[r], hd[r], ct[r]);
      if (act[r] && l < dim4) {
        g0[r] = grads[(int64_t)P[i] * dim4 + l];
        if (hd[r] && !ct[r]) {
          const int64_t off = ((int64_t)base_key + K[i]) * dim4 + l;
          w0[r] = table[off];
          if constexpr (OPT != TT_OPT_SGD) a0[r] = accum[off];
        }
      }
    }
#pragma unroll
    for (int r = 0; r < RP; ++r)
      if (act[r]) finish(grp + (uint32_t)r * ngroups, ee[r], hd[r], ct[r], true, g0[r], w0[r], a0[r]);
    for (uint32_t i = grp + RP * ngroups; i < m; i += ngroups) {     // (a hot range: more than RP slots per lane group)
      uint32_t e = 0u;
      bool head = false, cont = false;
      const f32x4 z = f32x4{0.f, 0.f, 0.f, 0.f};
      if (piece(i, e, head, cont)) finish(i, e, head, cont, false, z, z, z);
    }
  }
  if (multi && l == 0) *s_multi = 1;
  // ---- phase B (only when some run of this workgroup has several pieces): the run's head adds the pieces in index order ----
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __syncthreads();
  if (*s_multi == 0) return;
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  for (uint32_t h = grp; h < m; h += ngroups) {
    uint32_t he = 0u;
    bool hh = false, hc = false;
    if (!piece(h, he, hh, hc) || !hh || !hc) continue;        // heads of multi-piece runs only
    const uint32_t key = K[h];
    uint32_t last = he;                                       // one past the last slot of the run
    while (last < m && K[last] == key) ++last;
    // after fast_apply only runs of three or more ids came through phase A: a PAIR that straddles a 64-slot boundary was
    // finished there and has no piece sums in the workspace
    if constexpr (BY_LIST && !USE_RA) { if (last - h < 3u) continue; }
    const int64_t id = (int64_t)base_key + key;
    const int64_t jh = (int64_t)(offset + h) / kPiece, jl = (int64_t)(offset + last - 1u) / kPiece;
    const f32x4* Ps = reinterpret_cast<const f32x4*>(a.p_sum[t]);
    const f32x4* Ss = reinterpret_cast<const f32x4*>(a.s_sum[t]);
    for (int c = l; c < dim4; c += lpr) {
      f32x4 g = Ps[jh * dim4 + c];
      int64_t mb = jh + 1;
      while (mb + 3 <= jl) {                                    // four piece sums in flight, added in index order
        f32x4 sv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) sv[u] = Ss[(mb + u) * dim4 + c];
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int q = 0; q < 4; ++q) g[q] = __fadd_rn(g[q], sv[u][q]);
        mb += 4;
      }
      for (; mb <= jl; ++mb) {
        const f32x4 sv = Ss[mb * dim4 + c];
#pragma unroll
        for (int q = 0; q < 4; ++q) g[q] = __fadd_rn(g[q], sv[q]);
      }
      update_row<OPT>(table, accum, id * dim4 + c, g, lr, eps);
    }
  }
}

template <int OPT, int DBITS, int JMAX>
__global__ __launch_bounds__(1024) void optimizer_ids_kernel(ApplyArgs a, FusedTables ft, int n_tables, int dim4, int lpr_log2,
                                                              tt::SegTable tbl, int dense_blocks, int n_dense, float lr, float eps) {
  extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
  __shared__ int s_multi, s_slow;
  __shared__ uint32_t s_below;
  // flat grid: exactly the dense blocks each segment needs FIRST (a sorting workgroup fills a CU - 16 waves at up to 128
  // VGPRs - so dense blocks dispatched behind 256 of them would only start when those retire: r02 stamps, 10 us late),
  // then the sorting workgroups of table 0, 1, (2); the host keeps the total at one workgroup per CU
  // (r03: everything a sorting workgroup needs from the kernel arguments sits at STATIC offsets - the dense block count is an
  // argument of its own, the table is picked by compares and the three descriptors by selects - so the scalar loads leave in
  // one batch.  Indexed by values that were themselves loaded (seg_first[n_segs], first[ti + 1] in a loop, part[ti]) they
  // formed a chain of three dependent scalar-cache misses in front of the first id load.)
  const int b = (int)blockIdx.x - n_dense;
  if (b >= 0) {
    static_assert(kMaxSparseTables == 3, "table pick below");
    const int ti = (n_tables > 2 && b >= ft.first[2]) ? 2 : ((n_tables > 1 && b >= ft.first[1]) ? 1 : 0);
    tt::PartTable t{};
    {
      // all three descriptors' fields into SGPRs first (the empty asm pins the VALUES: left to itself the compiler selects the
      // field's ADDRESS by ti and loads through it - a second dependent round trip), then scalar selects
      const int64_t *i0 = ft.part[0].ids, *i1 = ft.part[1].ids, *i2 = ft.part[2].ids;
      int64_t r0 = ft.part[0].num_rows, r1 = ft.part[1].num_rows, r2 = ft.part[2].num_rows;
      int32_t n0 = ft.part[0].n, n1 = ft.part[1].n, n2 = ft.part[2].n;
      int32_t g0 = ft.part[0].groups, g1 = ft.part[1].groups, g2 = ft.part[2].groups;
      uint32_t w0 = ft.part[0].width, w1 = ft.part[1].width, w2 = ft.part[2].width;
      uint32_t m0 = ft.part[0].magic, m1 = ft.part[1].magic, m2 = ft.part[2].magic;
      uint32_t s0 = ft.part[0].sentinel, s1 = ft.part[1].sentinel, s2 = ft.part[2].sentinel;
      asm volatile("" : "+s"(i0), "+s"(i1), "+s"(i2), "+s"(r0), "+s"(r1), "+s"(r2), "+s"(n0), "+s"(n1), "+s"(n2), "+s"(g0), "+s"(g1),
                        "+s"(g2), "+s"(w0), "+s"(w1), "+s"(w2), "+s"(m0), "+s"(m1), "+s"(m2), "+s"(s0), "+s"(s1), "+s"(s2));
      t.ids = ti == 2 ? i2 : (ti == 1 ? i1 : i0);
      t.num_rows = ti == 2 ? r2 : (ti == 1 ? r1 : r0);
      t.n = ti == 2 ? n2 : (ti == 1 ? n1 : n0);
      t.groups = ti == 2 ? g2 : (ti == 1 ? g1 : g0);
      t.width = ti == 2 ? w2 : (ti == 1 ? w1 : w0);
      t.magic = ti == 2 ? m2 : (ti == 1 ? m1 : m0);
      t.sentinel = ti == 2 ? s2 : (ti == 1 ? s1 : s0);
    }
    if (threadIdx.x == 0) { s_multi = 0; s_slow = 0; }
    uint32_t offset, base_key;
    const int g = b - (ti == 2 ? ft.first[2] : (ti == 1 ? ft.first[1] : 0));
    // ---- r04: the range's ids from the list this step's FORWARD LOOKUP appended them to (tt_id_buckets) - one round trip for
    // the count and the <= bk_cap entries instead of reading and classifying all n ids of the table (r03 stamps: ids landed
    // 1.4 us, classified 2.4, appended 3.0, barrier 3.5 -> entries landed, barrier).  The list is in NO particular order
    // (arrival order of the forward pass's atomics): fast_apply needs none, and the ranked path ranks by (key, position).
    // An entry of another generation (a forward pass whose optimizer step never ran) becomes the all-ones key, which every
    // path skips and the ranking sorts last.  A list that overflowed (count > cap: a hot range) falls back to the scan. ----
    uint32_t* reset_count = nullptr;                                     // an overflowed list: reset behind the scan's first barrier
    {
      uint32_t* c0 = ft.bk_counts[0]; uint32_t* c1 = ft.bk_counts[1]; uint32_t* c2 = ft.bk_counts[2];
      const uint64_t *p0 = ft.bk_pairs[0], *p1 = ft.bk_pairs[1], *p2 = ft.bk_pairs[2];
      asm volatile("" : "+s"(c0), "+s"(c1), "+s"(c2), "+s"(p0), "+s"(p1), "+s"(p2));
      uint32_t* bc = ti == 2 ? c2 : (ti == 1 ? c1 : c0);
      const uint64_t* bp = ti == 2 ? p2 : (ti == 1 ? p1 : p0);
      if (bp != nullptr) {                                               // (workgroup-uniform)
        const uint32_t bcap = (uint32_t)ft.bk_cap, tid = threadIdx.x;
        const __attribute__((address_space(1))) uint32_t* bcg = (const __attribute__((address_space(1))) uint32_t*)(bc + (size_t)g * tt::kBucketCountStride);
        const __attribute__((address_space(1))) uint64_t* bpg = (const __attribute__((address_space(1))) uint64_t*)(bp + (size_t)g * bcap);
        const uint32_t cnt_v = *bcg;
        const uint64_t pr = bpg[tid < bcap ? tid : bcap - 1u];           // unconditional, clamped: one batch of loads
        const uint32_t cnt = (uint32_t)__builtin_amdgcn_readfirstlane((int)cnt_v);
        SSTAMP(0);
        if (cnt == 0u) return;                                           // (uniform: nobody writes the counter before the barrier below)
        reset_count = bc + (size_t)g * tt::kBucketCountStride;
        if (cnt <= bcap) {
          uint32_t* Kw = tt::part_keys(smem);
          uint16_t* Pw = tt::part_poss<DBITS>(smem, ft.cap);
          if (tid < cnt) {
            const bool fresh = (uint32_t)(pr >> 48) == (ft.bk_gen & 0xffffu);
            Kw[tid] = fresh ? (uint32_t)pr : 0xffffffffu;
            Pw[tid] = (uint16_t)(pr >> 32);
          }
          __syncthreads();
          if (tid == 0) *reset_count = 0u;                               // every wave has read it: zeroed for the next step's forward pass
          SSTAMP(2);
          const uint32_t m = cnt;
          base_key = (uint32_t)g * t.width;
          const uint32_t todo = fast_apply<OPT>(a, ti, Kw, Pw, m, base_key, dim4, lpr_log2, lr, eps, &s_slow);
          SSTAMP(3);
          if (s_slow == 0) { SSTAMP(5); SSTAMP(6); return; }
          // some id of the range occurs three times or more: the ranked path needs the range's position in the table's sorted
          // list (the 64-slot pieces of the sums are cut at GLOBAL slots) = the ids below the range, counted from the ids
          offset = tt::part_count_below<JMAX>(t, base_key, &s_below);
          tt::part_rank_small<DBITS, false>(t, ft.cap, smem, m, offset, base_key);
          const RowsAhead none_l{};
          apply_from_lds<OPT, DBITS, true, false>(a, ti, Kw, Pw, tt::part_ranks(smem, ft.cap), m, offset, base_key, dim4, lpr_log2, lr, eps,
                                                  &s_multi, none_l, todo);
          SSTAMP(6);
          return;
        }
      }
    }
    tt::PartScan<JMAX> sc;
    // (TT_OPT_PREFETCH, off - see the top of the file: the rows' trip from HBM started at the scan.  A wave that finds an id of
    // this workgroup's range touches every 128-byte line of that id's table (accumulator) row and of its gradient row with a
    // 4-byte load nobody reads; inline asm, the destination is one scratch VGPR kept reserved until the apply phase has
    // waited for its own, later loads.)
    uint32_t sink = 0u;
#if TT_OPT_PREFETCH == 2
    // form 4: ONE 4-byte touch per id, by the lane that holds it, of the first line of its TABLE (and accumulator) row only - the
    // gradient rows are a contiguous [B, D] array, a handful of pages.  What it buys is the address translation: random rows of a
    // multi-GB table miss the TLBs, and the page walk (several dependent memory reads) is most of the rows' 4.9 us round trip.
    const char* pf_tab = reinterpret_cast<const char*>(a.table[ti]);
    const char* pf_acc = reinterpret_cast<const char*>(a.accum[ti]);
    const int row_bytes = dim4 * 16;
    const int pf_lane = (int)(threadIdx.x & 63u);
    auto prefetch = [&](uint64_t mask, uint32_t key, uint32_t) {
      if ((mask >> pf_lane) & 1ull) {
        const char* tr = pf_tab + (int64_t)key * row_bytes;
        asm volatile("global_load_dword %0, %1, off" : "+v"(sink) : "v"(tr) : "memory");
        if constexpr (OPT != TT_OPT_SGD) {
          const char* ar = pf_acc + (int64_t)key * row_bytes;
          asm volatile("global_load_dword %0, %1, off" : "+v"(sink) : "v"(ar) : "memory");
        }
      }
    };
    const uint32_t m = tt::part_scan_append<DBITS, JMAX, true>(t, g, ft.cap, smem, sc, offset, base_key, prefetch);
#elif TT_OPT_PREFETCH
    const char* pf_tab = reinterpret_cast<const char*>(a.table[ti]);
    const char* pf_acc = reinterpret_cast<const char*>(a.accum[ti]);
    const char* pf_grd = reinterpret_cast<const char*>(a.grads[ti]);
    const int row_bytes = dim4 * 16;
    // ONE load instruction per id of the range: its key and position are broadcast from the lane that holds them, lane 4 s + i
    // of the wave touches line i of row s (s = 0 table, 1 gradient, 2 accumulator).  (A load instruction occupies the address
    // path for a whole wave however few lanes are active - one instruction per lane and line, the first form of this
    // prefetch, cost more than the rows' earlier arrival saved: launch 12.6 -> 13.7 us.)
    const int lines = row_bytes >> 7;                                   // 128-byte lines per row (dim 128: 4)
    const int pf_lane = (int)(threadIdx.x & 63u);
    const int pf_div = lines > 0 ? lines : 1;
    const int pf_seg = pf_lane / pf_div;
    // this lane's row base + line offset, chosen once (plain selects: indexed by pf_seg the three pointers became a scratch array)
    const char* pf_base = pf_tab;
    if (pf_seg == 1) pf_base = pf_grd;
    if (pf_seg == 2) pf_base = pf_acc;
    pf_base += 128 * (pf_lane % pf_div);
    const bool pf_by_pos = pf_seg == 1;
    const bool pf_on = lines > 0 && pf_seg < (OPT != TT_OPT_SGD ? 3 : 2);
    auto prefetch = [&](uint64_t mask, uint32_t key, uint32_t pos) {
      while (mask != 0ull) {
        const int b = __builtin_ctzll(mask);
        mask &= mask - 1ull;
        const uint32_t k = (uint32_t)__builtin_amdgcn_readlane((int)key, b), p = (uint32_t)__builtin_amdgcn_readlane((int)pos, b);
        const char* addr = pf_base + (int64_t)(pf_by_pos ? p : k) * row_bytes;
        if (pf_on) asm volatile("global_load_dword %0, %1, off" : "+v"(sink) : "v"(addr) : "memory");
      }
    };
    const uint32_t m = tt::part_scan_append<DBITS, JMAX, true>(t, g, ft.cap, smem, sc, offset, base_key, prefetch);
#else
    const uint32_t m = tt::part_scan_append<DBITS, JMAX, true>(t, g, ft.cap, smem, sc, offset, base_key);
#endif
    if (reset_count != nullptr && threadIdx.x == 0) *reset_count = 0u;
    if (m == 0u) return;
    const uint32_t* K = tt::part_keys(smem);
    const uint16_t* P = tt::part_poss<DBITS>(smem, ft.cap);
    const RowsAhead none{};
    if (m <= tt::kPartRankMax) {
      if (lpr_log2 <= 5 && dim4 <= (1 << lpr_log2) && m <= (uint32_t)kRowsAhead * (1024u >> lpr_log2)) {
        // the usual row range: rows touched once or twice are finished without ranks (fast_apply); the ranking runs only
        // if some id of the range occurs three times or more, and then only those pairs go through it
        const uint32_t todo = fast_apply<OPT>(a, ti, K, P, m, base_key, dim4, lpr_log2, lr, eps, &s_slow);
        SSTAMP(3);                         // (fast_apply holds the workgroup barrier: s_slow is final, the unordered list read)
        if (s_slow == 0) { SSTAMP(5); SSTAMP(6); asm volatile("" :: "v"(sink)); return; }
        tt::part_rank_small<DBITS, false>(t, ft.cap, smem, m, offset, base_key);
        apply_from_lds<OPT, DBITS, true, false>(a, ti, K, P, tt::part_ranks(smem, ft.cap), m, offset, base_key, dim4, lpr_log2, lr, eps,
                                                &s_multi, none, todo);
      } else {
        // a longer list (or rows wider than a lane group): rows requested per unordered pair, THEN the ranking
        RowsAhead ra;
        request_rows<OPT>(a, ti, K, P, m, base_key, dim4, lpr_log2, ra);
        SSTAMP(3);
        tt::part_rank_small<DBITS, false>(t, ft.cap, smem, m, offset, base_key);
        apply_from_lds<OPT, DBITS, true, true>(a, ti, K, P, tt::part_ranks(smem, ft.cap), m, offset, base_key, dim4, lpr_log2, lr, eps,
                                               &s_multi, ra, 0u);
      }
    } else {
      tt::part_sort_hot<DBITS, JMAX, false>(t, g, ft.cap, smem, sc, m, offset, base_key);
      apply_from_lds<OPT, DBITS, false, false>(a, ti, K, P, nullptr, m, offset, base_key, dim4, lpr_log2, lr, eps, &s_multi, none, 0u);
    }
    SSTAMP(6);
    asm volatile("" :: "v"(sink));                 // (the scratch VGPR of the prefetch loads stays reserved to here)
  } else {
    const int d = (int)blockIdx.x;
    int si = 0;
    while (si + 1 < dense_blocks && d >= ft.seg_first[si + 1]) ++si;        // (dense_blocks = number of segments here)
    SSTAMP(0);
    tt::dense_update_body<OPT, 1024, 16>(tbl.seg[si], d - ft.seg_first[si], ft.seg_first[si + 1] - ft.seg_first[si], 1, lr, eps);
    SSTAMP(6);
  }
}

// ---- the same launch for lists of 16,385 .. 65,536 ids per table (r04; cfg5 un-sharded: 32,768) ----
// r03 sent these through tt_sparse_plan (chunk sorts + merge: 70-82 us at cfg5) and the separate apply kernel (85 us).  Here
// the sorting workgroups keep their 16384-slot LDS list and scan the ids in chunks (csrc/part_sort.h, the long-list section);
// the row ranges are the same ~120 per table, so the usual range holds n/120 ~ 270 ids: rows requested per unordered pair,
// ranked by counting, applied - the short-list kernel's general path.  A hot range (> 512 ids) is radix-sorted in LDS from the
// unordered list; a range with more ids than the LDS list holds is sorted in the apply workspace's global scratch and applied
// from there (correct for any batch, fast for none that has such a range: a trainer watches tt_id_range_load and takes the
// plan + apply path while its batches are that skewed).  No row-range id lists here.  Measured (2 x 32768 ids, dim 256,
// Adagrad, uniform ids): 93 us against 113 us for plan (53) + apply (60).
template <int OPT, int DBITS>
__global__ __launch_bounds__(1024) void optimizer_ids_big_kernel(ApplyArgs a, FusedTables ft, int n_tables, int dim4, int lpr_log2,
                                                                  tt::SegTable tbl, int dense_blocks, int n_dense, float lr, float eps) {
  extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
  __shared__ int s_multi;
  const int b = (int)blockIdx.x - n_dense;
  if (b >= 0) {
    const int ti = (n_tables > 2 && b >= ft.first[2]) ? 2 : ((n_tables > 1 && b >= ft.first[1]) ? 1 : 0);
    const tt::PartTable t = ti == 2 ? ft.part[2] : (ti == 1 ? ft.part[1] : ft.part[0]);
    const int g = b - (ti == 2 ? ft.first[2] : (ti == 1 ? ft.first[1] : 0));
    if (threadIdx.x == 0) s_multi = 0;
    uint32_t offset, base_key, diff;
    const uint32_t span = tt::part_range_span(t, g, base_key);
    const uint32_t m = tt::part_scan_append_big<DBITS, true>(t, base_key, span, ft.cap, smem, offset, diff);
    if (m == 0u) return;
    const RowsAhead none{};
    if (m > (uint32_t)ft.cap) {
      uint64_t* gp = (ti == 2 ? a.big_pairs[2] : (ti == 1 ? a.big_pairs[1] : a.big_pairs[0])) + offset;
      uint32_t* kg = (ti == 2 ? a.big_keys[2] : (ti == 1 ? a.big_keys[1] : a.big_keys[0])) + offset;
      uint16_t* pg = (ti == 2 ? a.big_pos[2] : (ti == 1 ? a.big_pos[1] : a.big_pos[0])) + offset;
      const uint64_t* srt = tt::part_sort_global<true>(t, base_key, span, smem, gp, gp + t.n, m, diff);
      for (uint32_t e = threadIdx.x; e < m; e += 1024u) {
        const uint64_t v = srt[e];
        kg[e] = (uint32_t)(v >> 16);
        pg[e] = (uint16_t)(v & 0xffffu);
      }
      __syncthreads();
      apply_from_lds<OPT, DBITS, false, false>(a, ti, kg, pg, nullptr, m, offset, base_key, dim4, lpr_log2, lr, eps, &s_multi, none, 0u);
      return;
    }
    const uint32_t* K = tt::part_keys(smem);
    const uint16_t* P = tt::part_poss<DBITS>(smem, ft.cap);
    if (m <= tt::kPartRankMax) {
      RowsAhead ra;
      request_rows<OPT>(a, ti, K, P, m, base_key, dim4, lpr_log2, ra);
      tt::part_rank_small<DBITS, false>(t, ft.cap, smem, m, offset, base_key);
      apply_from_lds<OPT, DBITS, true, true>(a, ti, K, P, tt::part_ranks(smem, ft.cap), m, offset, base_key, dim4, lpr_log2, lr, eps,
                                             &s_multi, ra, 0u);
    } else {
      tt::part_sort_hot_unordered<DBITS, false>(t, span, ft.cap, smem, m, offset, base_key, diff);
      apply_from_lds<OPT, DBITS, false, false>(a, ti, K, P, nullptr, m, offset, base_key, dim4, lpr_log2, lr, eps, &s_multi, none, 0u);
    }
    SSTAMP(6);
  } else {
    const int d = (int)blockIdx.x;
    int si = 0;
    while (si + 1 < dense_blocks && d >= ft.seg_first[si + 1]) ++si;
    tt::dense_update_body<OPT, 1024, 16>(tbl.seg[si], d - ft.seg_first[si], ft.seg_first[si + 1] - ft.seg_first[si], 1, lr, eps);
  }
}

int64_t align_up(int64_t x, int64_t a);

struct PieceWs {
  int64_t nblk, off_p, off_s, off_flag, off_pairs, off_keys, off_pos, total;
};
PieceWs piece_ws(int64_t n_ids, int32_t dim) {
  PieceWs w{};
  w.nblk = (n_ids + kPiece - 1) / kPiece;
  w.off_flag = 0;
  w.off_p = align_up(w.nblk * 4, 256);
  w.off_s = w.off_p + align_up(w.nblk * (int64_t)dim * 4, 256);
  w.total = w.off_s + align_up(w.nblk * (int64_t)dim * 4, 256);
  w.off_pairs = w.off_keys = w.off_pos = 0;
  if (n_ids > tt::kPartSortMaxIds && n_ids <= tt::kPartSortBigMaxIds) {       // (scratch of a degenerate batch's global sort)
    w.off_pairs = w.total;
    w.off_keys = w.off_pairs + align_up(2 * n_ids * 8, 256);
    w.off_pos = w.off_keys + align_up(n_ids * 4, 256);
    w.total = w.off_pos + align_up(n_ids * 2, 256);
  }
  return w;
}

int prepare_ws(ApplyArgs& a, void* const* ws, int n_tables, int32_t dim, int64_t n_ids, const char* what) {
  const PieceWs w = piece_ws(n_ids, dim);
  for (int t = 0; t < n_tables; ++t) {
    TT_REQUIRE(ws[t] != nullptr && (reinterpret_cast<uintptr_t>(ws[t]) & 255u) == 0, "%s: apply workspace must be non-null, 256-byte aligned", what);
    char* base = static_cast<char*>(ws[t]);
    a.p_flag[t] = reinterpret_cast<int32_t*>(base + w.off_flag);
    a.p_sum[t] = reinterpret_cast<float*>(base + w.off_p);
    a.s_sum[t] = reinterpret_cast<float*>(base + w.off_s);
    if (w.off_pairs != 0) {
      a.big_pairs[t] = reinterpret_cast<uint64_t*>(base + w.off_pairs);
      a.big_keys[t] = reinterpret_cast<uint32_t*>(base + w.off_keys);
      a.big_pos[t] = reinterpret_cast<uint16_t*>(base + w.off_pos);
    }
  }
  return TT_OK;
}

int launch_apply(int opt, ApplyArgs a, void* const ws[2], int n_tables, int32_t dim, int64_t n_ids, float lr, float eps,
                 hipStream_t stream, const char* what) {
  if (n_ids == 0) return TT_OK;
  const int dim4 = dim / 4;
  int lpr_log2 = 0;
  while ((1 << lpr_log2) < dim4 && lpr_log2 < 6) ++lpr_log2;
  const int groups = 256 >> lpr_log2;
  const int64_t blocks = (n_ids + groups - 1) / groups;
  TT_REQUIRE(blocks <= 0x7fffffff, "%s: n_ids too large", what);
  const PieceWs w = piece_ws(n_ids, dim);
  for (int t = 0; t < n_tables; ++t) {
    TT_REQUIRE(ws[t] != nullptr && (reinterpret_cast<uintptr_t>(ws[t]) & 255u) == 0, "%s: apply workspace must be non-null, 256-byte aligned", what);
    char* base = static_cast<char*>(ws[t]);
    a.p_flag[t] = reinterpret_cast<int32_t*>(base + w.off_flag);
    a.p_sum[t] = reinterpret_cast<float*>(base + w.off_p);
    a.s_sum[t] = reinterpret_cast<float*>(base + w.off_s);
  }
  if (opt == TT_OPT_SGD)
    tt::launch("sparse_apply", sparse_apply_kernel<TT_OPT_SGD>, dim3((unsigned)blocks, n_tables), dim3(256), 0, stream, a, dim4,
                       lpr_log2, n_ids, lr, eps);
  else
    tt::launch("sparse_apply", sparse_apply_kernel<TT_OPT_ADAGRAD>, dim3((unsigned)blocks, n_tables), dim3(256), 0, stream, a,
                       dim4, lpr_log2, n_ids, lr, eps);
  return tt::check_launch(what);
}

int64_t align_up(int64_t x, int64_t a) { return (x + a - 1) / a * a; }

}  // namespace

extern "C" int64_t tt_sparse_apply_workspace_bytes(int64_t n_ids, int32_t dim) {
  if (n_ids <= 0 || dim <= 0) return 256;
  return piece_ws(n_ids, dim).total;
}

extern "C" int tt_sparse_sgd_f32(float* table, int64_t num_rows, int32_t dim, const float* grads, const int64_t* sorted_ids,
                                 const int32_t* order, int64_t n_ids, float lr, void* apply_ws, tt_stream_t stream) {
  TT_REQUIRE(n_ids >= 0 && num_rows > 0 && dim > 0 && dim % 4 == 0, "tt_sparse_sgd_f32: bad n_ids/num_rows/dim");
  TT_REQUIRE(n_ids == 0 || (table && grads && sorted_ids && order), "tt_sparse_sgd_f32: null pointer");
  TT_REQUIRE(tt::aligned16(table) && tt::aligned16(grads), "tt_sparse_sgd_f32: table/grads must be 16-byte aligned");
  ApplyArgs a{};
  a.table[0] = table; a.grads[0] = grads; a.sorted_ids[0] = sorted_ids; a.order[0] = order; a.rows[0] = num_rows;
  void* const ws[2] = {apply_ws, nullptr};
  return launch_apply(TT_OPT_SGD, a, ws, 1, dim, n_ids, lr, 0.f, tt::as_stream(stream), "tt_sparse_sgd_f32");
}

extern "C" int tt_sparse_adagrad_f32(float* table, float* accum, int64_t num_rows, int32_t dim, const float* grads,
                                     const int64_t* sorted_ids, const int32_t* order, int64_t n_ids, float lr, float eps,
                                     void* apply_ws, tt_stream_t stream) {
  TT_REQUIRE(n_ids >= 0 && num_rows > 0 && dim > 0 && dim % 4 == 0, "tt_sparse_adagrad_f32: bad n_ids/num_rows/dim");
  TT_REQUIRE(n_ids == 0 || (table && accum && grads && sorted_ids && order), "tt_sparse_adagrad_f32: null pointer");
  TT_REQUIRE(tt::aligned16(table) && tt::aligned16(accum) && tt::aligned16(grads),
             "tt_sparse_adagrad_f32: table/accum/grads must be 16-byte aligned");
  ApplyArgs a{};
  a.table[0] = table; a.accum[0] = accum; a.grads[0] = grads; a.sorted_ids[0] = sorted_ids; a.order[0] = order;
  a.rows[0] = num_rows;
  void* const ws[2] = {apply_ws, nullptr};
  return launch_apply(TT_OPT_ADAGRAD, a, ws, 1, dim, n_ids, lr, eps, tt::as_stream(stream), "tt_sparse_adagrad_f32");
}

extern "C" int tt_sparse_update2_f32(int32_t opt, float* table_a, float* accum_a, int64_t rows_a, const float* grads_a,
                                     const int64_t* sorted_ids_a, const int32_t* order_a, float* table_b, float* accum_b,
                                     int64_t rows_b, const float* grads_b, const int64_t* sorted_ids_b,
                                     const int32_t* order_b, int32_t dim, int64_t n_ids, float lr, float eps,
                                     void* apply_ws_a, void* apply_ws_b, tt_stream_t stream) {
  TT_REQUIRE(opt == TT_OPT_SGD || opt == TT_OPT_ADAGRAD, "tt_sparse_update2_f32: unknown optimizer %d", opt);
  TT_REQUIRE(n_ids >= 0 && rows_a > 0 && rows_b > 0 && dim > 0 && dim % 4 == 0, "tt_sparse_update2_f32: bad sizes");
  TT_REQUIRE(n_ids == 0 || (table_a && grads_a && sorted_ids_a && order_a && table_b && grads_b && sorted_ids_b && order_b),
             "tt_sparse_update2_f32: null pointer");
  TT_REQUIRE(opt == TT_OPT_SGD || (accum_a && accum_b), "tt_sparse_update2_f32: Adagrad needs accumulators");
  TT_REQUIRE(tt::aligned16(table_a) && tt::aligned16(table_b) && tt::aligned16(grads_a) && tt::aligned16(grads_b) &&
                 tt::aligned16(accum_a) && tt::aligned16(accum_b),
             "tt_sparse_update2_f32: pointers must be 16-byte aligned");
  ApplyArgs a{};
  a.table[0] = table_a; a.accum[0] = accum_a; a.grads[0] = grads_a; a.sorted_ids[0] = sorted_ids_a; a.order[0] = order_a; a.rows[0] = rows_a;
  a.table[1] = table_b; a.accum[1] = accum_b; a.grads[1] = grads_b; a.sorted_ids[1] = sorted_ids_b; a.order[1] = order_b; a.rows[1] = rows_b;
  void* const ws[2] = {apply_ws_a, apply_ws_b};
  return launch_apply(opt, a, ws, 2, dim, n_ids, lr, eps, tt::as_stream(stream), "tt_sparse_update2_f32");
}

// ---- the train step's whole optimizer in one launch ---------------------------------------------------------------
extern "C" int tt_optimizer_step_f32(int32_t opt, const tt_sparse_table* tables, int32_t n_tables, int32_t dim, int64_t n_ids,
                                     const tt_dense_seg* segs, int32_t n_segs, float lr, float eps, tt_stream_t stream_) {
  TT_REQUIRE(opt == TT_OPT_SGD || opt == TT_OPT_ADAGRAD, "tt_optimizer_step_f32: unknown optimizer %d", opt);
  TT_REQUIRE(tables != nullptr && n_tables >= 1 && n_tables <= kMaxSparseTables, "tt_optimizer_step_f32: 1..%d sparse tables", kMaxSparseTables);
  TT_REQUIRE(segs != nullptr && n_segs >= 1 && n_segs <= TT_MAX_DENSE_SEGS, "tt_optimizer_step_f32: 1..%d dense segments", TT_MAX_DENSE_SEGS);
  TT_REQUIRE(n_ids > 0 && dim > 0 && dim % 4 == 0, "tt_optimizer_step_f32: bad n_ids/dim");
  ApplyArgs a{};
  void* ws[kMaxSparseTables] = {};
  for (int t = 0; t < n_tables; ++t) {
    const tt_sparse_table& s = tables[t];
    TT_REQUIRE(s.table && s.grads && s.sorted_ids && s.order && s.rows > 0, "tt_optimizer_step_f32: table %d: null pointer / bad rows", t);
    TT_REQUIRE(opt == TT_OPT_SGD || s.accum != nullptr, "tt_optimizer_step_f32: table %d: Adagrad needs the accumulator", t);
    TT_REQUIRE(tt::aligned16(s.table) && tt::aligned16(s.grads) && tt::aligned16(s.accum), "tt_optimizer_step_f32: table %d: pointers must be 16-byte aligned", t);
    a.table[t] = s.table; a.accum[t] = s.accum; a.grads[t] = s.grads; a.sorted_ids[t] = s.sorted_ids; a.order[t] = s.order;
    a.rows[t] = s.rows;
    ws[t] = s.apply_ws;
  }
  int rc = prepare_ws(a, ws, n_tables, dim, n_ids, "tt_optimizer_step_f32");
  if (rc != TT_OK) return rc;
  tt::SegTable tbl{};
  int64_t max_count = 0;
  for (int i = 0; i < n_segs; ++i) {
    const tt_dense_seg& s = segs[i];
    TT_REQUIRE(s.count > 0 && s.n_slabs >= 1 && s.grad_slabs != nullptr && s.param != nullptr, "tt_optimizer_step_f32: segment %d: bad count/slabs/param", i);
    TT_REQUIRE(opt == TT_OPT_SGD || s.accum != nullptr, "tt_optimizer_step_f32: segment %d: Adagrad needs accum", i);
    tbl.seg[i] = s;
    if (s.count > max_count) max_count = s.count;
  }
  const int dim4 = dim / 4;
  int lpr_log2 = 0;
  while ((1 << lpr_log2) < dim4 && lpr_log2 < 6) ++lpr_log2;
  const int groups = 256 >> lpr_log2;
  const int64_t sparse_blocks = (n_ids + groups - 1) / groups;
  int64_t dense_blocks = (max_count + 255) / 256;
  if (dense_blocks > 512) dense_blocks = 512;
  const int64_t gx = sparse_blocks > dense_blocks ? sparse_blocks : dense_blocks;
  TT_REQUIRE(gx <= 0x7fffffff, "tt_optimizer_step_f32: n_ids too large");
  hipStream_t stream = tt::as_stream(stream_);
  if (opt == TT_OPT_SGD)
    tt::launch("optimizer", optimizer_kernel<TT_OPT_SGD>, dim3((unsigned)gx, (unsigned)(n_tables + n_segs)), dim3(256), 0, stream, a,
                       n_tables, dim4, lpr_log2, n_ids, sparse_blocks, tbl, (int)dense_blocks, lr, eps);
  else
    tt::launch("optimizer", optimizer_kernel<TT_OPT_ADAGRAD>, dim3((unsigned)gx, (unsigned)(n_tables + n_segs)), dim3(256), 0, stream, a,
                       n_tables, dim4, lpr_log2, n_ids, sparse_blocks, tbl, (int)dense_blocks, lr, eps);
  return tt::check_launch("tt_optimizer_step_f32");
}

// The row ranges of the fused launch: exactly the dense blocks each segment needs, then ~64 ids per sorting workgroup and no
// more workgroups than fit beside the dense blocks at one per CU.  Shared by the launcher and tt_optimizer_ids_geometry (the
// forward lookup that fills the row-range id lists must cut the tables the same way).
namespace {
struct IdsGeometry {
  int32_t seg_first[TT_MAX_DENSE_SEGS + 1];
  int32_t groups[kMaxSparseTables];
  uint32_t width[kMaxSparseTables];
};
void ids_geometry(const int64_t* rows, int n_tables, int64_t n_ids, const tt_dense_seg* segs, int n_segs, IdsGeometry& ge) {
  ge.seg_first[0] = 0;
  for (int i = 0; i < n_segs; ++i) {                      // one thread per 4 elements, at most 16 blocks per segment
    int64_t nb = (segs[i].count / 4 + 1023) / 1024;
    if (nb < 1) nb = 1;
    if (nb > 16) nb = 16;
    ge.seg_first[i + 1] = ge.seg_first[i] + (int32_t)nb;
  }
  int64_t group_cap = (256 - ge.seg_first[n_segs]) / n_tables;
  if (group_cap < 16) group_cap = 16;
  for (int t = 0; t < n_tables; ++t) {
    int64_t g = (n_ids + 63) / 64;
    if (g > group_cap) g = group_cap;              // (one table - a sharded owner's combined shard - takes all the CUs the dense blocks leave)
    if (g > rows[t] / 2) g = rows[t] / 2;
    ge.groups[t] = g < 1 ? 1 : (int32_t)g;
    ge.width[t] = (uint32_t)((rows[t] + ge.groups[t] - 1) / ge.groups[t]);
    if (ge.width[t] < 2u) ge.width[t] = 2u;
  }
}
int lanes_per_row_log2(int dim4) {
  int l = 0;
  while ((1 << l) < dim4 && l < 6) ++l;
  return l;
}
// entries per row-range id list: what fast_apply finishes without ranks (kRowsAhead pairs per lane group), at most 256
int bucket_cap(int32_t dim, int64_t n_ids) {
  const int l = lanes_per_row_log2(dim / 4);
  if (l > 5 || n_ids > tt::kPartSortMaxIds) return 0;
  const int c = kRowsAhead * (1024 >> l);
  return c < 256 ? c : 256;
}
using tt::kBucketGroupsMax;
}  // namespace

extern "C" int32_t tt_optimizer_ids_max_ids(void) { return tt::kPartSortBigMaxIds; }

extern "C" int64_t tt_id_buckets_workspace_bytes(void) {
  return (int64_t)kBucketGroupsMax * tt::kBucketCountStride * 4 + (int64_t)kBucketGroupsMax * 256 * 8;      // counters (a line each) + lists
}

extern "C" int tt_optimizer_ids_geometry(const int64_t* table_rows, int32_t n_tables, int32_t dim, int64_t n_ids,
                                         const tt_dense_seg* segs, int32_t n_segs, int32_t* groups, uint32_t* width, int32_t* cap) {
  TT_REQUIRE(table_rows && groups && width && cap, "tt_optimizer_ids_geometry: null pointer");
  TT_REQUIRE(n_tables >= 1 && n_tables <= kMaxSparseTables, "tt_optimizer_ids_geometry: 1..%d sparse tables", kMaxSparseTables);
  TT_REQUIRE(segs != nullptr && n_segs >= 1 && n_segs <= TT_MAX_DENSE_SEGS, "tt_optimizer_ids_geometry: 1..%d dense segments", TT_MAX_DENSE_SEGS);
  TT_REQUIRE(n_ids > 0 && dim > 0 && dim % 4 == 0, "tt_optimizer_ids_geometry: bad n_ids/dim");
  for (int t = 0; t < n_tables; ++t) TT_REQUIRE(table_rows[t] > 0, "tt_optimizer_ids_geometry: table %d: bad rows", t);
  IdsGeometry ge{};
  ids_geometry(table_rows, n_tables, n_ids, segs, n_segs, ge);
  for (int t = 0; t < n_tables; ++t) { groups[t] = ge.groups[t]; width[t] = ge.width[t]; }
  *cap = bucket_cap(dim, n_ids);
  return TT_OK;
}

namespace {
// ---- how unevenly a batch's ids fall on the row ranges (a trainer's lagged skew probe) ----
// The fused launch gives every row range to ONE workgroup.  Ids that are uniform over the rows put n / groups ~ 64-270 ids
// in each; ids that crowd a few ranges (a vocabulary in order of frequency: 30 % of a power-law batch lands in the first of
// 119 ranges) put thousands into one workgroup, and that workgroup is the launch: cfg3 with ids ~ rows * u^4, 12 -> 169 us,
// step 0.56 -> 0.75 ms (r04).  The plan + apply path spreads the SORTED list over all CUs whatever the ids (45 us there).
// One workgroup per table counts the ids of each range in LDS and writes the largest count to out_max[t] (device memory): a
// trainer runs this every few dozen steps, copies the word to pinned host memory without waiting, and picks the path of the
// following steps from it (trainer.py: the one-launch form while no range holds more than kPartRankMax ids).
struct RangeLoadArgs {
  const int64_t* ids[kMaxSparseTables];
  int64_t rows[kMaxSparseTables];
  int32_t groups[kMaxSparseTables];
  uint32_t width[kMaxSparseTables];
  int32_t n;
  int32_t* out_max;
};

__global__ __launch_bounds__(1024) void range_load_kernel(RangeLoadArgs a) {
  __shared__ uint32_t cnt[tt::kBucketGroupsMax];
  __shared__ uint32_t best;
  const int t = blockIdx.x, tid = threadIdx.x;
  const int groups = a.groups[t];
  const uint32_t width = a.width[t];
  for (int i = tid; i < groups; i += 1024) cnt[i] = 0u;
  if (tid == 0) best = 0u;
  __syncthreads();
  for (int i = tid; i < a.n; i += 1024) {
    const int64_t id = a.ids[t][i];
    if (id >= 0 && id < a.rows[t]) {
      uint32_t g = (uint32_t)id / width;
      if (g >= (uint32_t)groups) g = (uint32_t)groups - 1u;
      atomicAdd(&cnt[g], 1u);
    }
  }
  __syncthreads();
  uint32_t m = 0u;
  for (int i = tid; i < groups; i += 1024) m = cnt[i] > m ? cnt[i] : m;
  if (m != 0u) atomicMax(&best, m);
  __syncthreads();
  if (tid == 0) a.out_max[t] = (int32_t)best;
}
}  // namespace

extern "C" int tt_id_range_load(const int64_t* const* ids, const int64_t* table_rows, int32_t n_tables, int32_t dim, int64_t n_ids,
                                const tt_dense_seg* segs, int32_t n_segs, int32_t* out_max, tt_stream_t stream_) {
  TT_REQUIRE(ids && table_rows && out_max, "tt_id_range_load: null pointer");
  TT_REQUIRE(n_tables >= 1 && n_tables <= kMaxSparseTables, "tt_id_range_load: 1..%d sparse tables", kMaxSparseTables);
  TT_REQUIRE(segs != nullptr && n_segs >= 1 && n_segs <= TT_MAX_DENSE_SEGS, "tt_id_range_load: 1..%d dense segments", TT_MAX_DENSE_SEGS);
  TT_REQUIRE(n_ids > 0 && n_ids <= tt::kPartSortBigMaxIds && dim > 0 && dim % 4 == 0, "tt_id_range_load: bad n_ids/dim");
  IdsGeometry ge{};
  ids_geometry(table_rows, n_tables, n_ids, segs, n_segs, ge);
  RangeLoadArgs a{};
  for (int t = 0; t < n_tables; ++t) {
    TT_REQUIRE(ids[t] != nullptr && table_rows[t] > 0 && table_rows[t] <= 0x7fffffff, "tt_id_range_load: table %d: null ids / bad rows", t);
    TT_REQUIRE(ge.groups[t] <= kBucketGroupsMax, "tt_id_range_load: table %d: %d row ranges", t, ge.groups[t]);
    a.ids[t] = ids[t]; a.rows[t] = table_rows[t]; a.groups[t] = ge.groups[t]; a.width[t] = ge.width[t];
  }
  a.n = (int32_t)n_ids;
  a.out_max = out_max;
  tt::launch("range_load", range_load_kernel, dim3((unsigned)n_tables), dim3(1024), 0, tt::as_stream(stream_), a);
  return tt::check_launch("tt_id_range_load");
}

// The same optimizer step from the RAW ids: no tt_sparse_plan launch, no sorted ids / positions in HBM.  One launch: the
// sorting workgroups of every table (n/64 per table, csrc/part_sort.h) apply the update to the rows of their own key
// range, the dense segments are updated beside them.  n_ids <= tt_optimizer_ids_max_ids() (65536; beyond 16384 the long-list
// kernel, which needs the larger tt_sparse_apply_workspace_bytes of such a list); same results, bit for bit, as
// tt_sparse_plan_batched + tt_optimizer_step_f32.
extern "C" int tt_optimizer_step_ids_f32(int32_t opt, const tt_sparse_table_ids* tables, int32_t n_tables, int32_t dim, int64_t n_ids,
                                         const tt_dense_seg* segs, int32_t n_segs, float lr, float eps, tt_stream_t stream_) {
  TT_REQUIRE(opt == TT_OPT_SGD || opt == TT_OPT_ADAGRAD, "tt_optimizer_step_ids_f32: unknown optimizer %d", opt);
  TT_REQUIRE(tables != nullptr && n_tables >= 1 && n_tables <= kMaxSparseTables, "tt_optimizer_step_ids_f32: 1..%d sparse tables", kMaxSparseTables);
  TT_REQUIRE(segs != nullptr && n_segs >= 1 && n_segs <= TT_MAX_DENSE_SEGS, "tt_optimizer_step_ids_f32: 1..%d dense segments", TT_MAX_DENSE_SEGS);
  TT_REQUIRE(n_ids > 0 && dim > 0 && dim % 4 == 0, "tt_optimizer_step_ids_f32: bad n_ids/dim");
  if (n_ids > tt::kPartSortBigMaxIds)
    return tt::fail(TT_ERR_UNSUPPORTED, "tt_optimizer_step_ids_f32: n_ids %lld > %d (use tt_sparse_plan_batched + tt_optimizer_step_f32)",
                    (long long)n_ids, tt::kPartSortBigMaxIds);
  const bool big = n_ids > tt::kPartSortMaxIds;              // 16,385 .. 65,536 ids: optimizer_ids_big_kernel
  ApplyArgs a{};
  FusedTables ft{};
  void* ws[kMaxSparseTables] = {};
  int max_groups = 0, max_lbits = 1;
  IdsGeometry ge{};
  {
    int64_t rows[kMaxSparseTables] = {1, 1, 1};
    for (int t = 0; t < n_tables; ++t) {
      TT_REQUIRE(tables[t].rows > 0, "tt_optimizer_step_ids_f32: table %d: null pointer / bad rows", t);
      rows[t] = tables[t].rows;
    }
    ids_geometry(rows, n_tables, n_ids, segs, n_segs, ge);
  }
  for (int i = 0; i <= n_segs; ++i) ft.seg_first[i] = ge.seg_first[i];
  ft.bk_cap = bucket_cap(dim, n_ids);
  for (int t = 0; t < n_tables; ++t) {
    const tt_sparse_table_ids& s = tables[t];
    TT_REQUIRE(s.table && s.grads && s.ids && s.rows > 0, "tt_optimizer_step_ids_f32: table %d: null pointer / bad rows", t);
    TT_REQUIRE(s.rows <= 0x7fffffff, "tt_optimizer_step_ids_f32: table %d: rows must fit in 31 bits", t);
    TT_REQUIRE(opt == TT_OPT_SGD || s.accum != nullptr, "tt_optimizer_step_ids_f32: table %d: Adagrad needs the accumulator", t);
    TT_REQUIRE(tt::aligned16(s.table) && tt::aligned16(s.grads) && tt::aligned16(s.accum), "tt_optimizer_step_ids_f32: table %d: pointers must be 16-byte aligned", t);
    a.table[t] = s.table; a.accum[t] = s.accum; a.grads[t] = s.grads; a.rows[t] = s.rows;
    ws[t] = s.apply_ws;
    int bits = 1;
    while (bits < 63 && ((int64_t)1 << bits) < s.rows + 1) ++bits;
    tt::PartTable& p = ft.part[t];
    p.ids = s.ids; p.sorted_ids = nullptr; p.order = nullptr; p.num_rows = s.rows; p.n = (int32_t)n_ids;
    p.groups = ge.groups[t];
    p.width = ge.width[t];
    p.magic = (uint32_t)((((uint64_t)1 << 32) / p.width) + 1u);
    p.sentinel = (uint32_t)(((uint64_t)1 << bits) - 1);
    int lb = 1;
    while (((int64_t)1 << lb) < (int64_t)p.width) ++lb;
    if (lb > max_lbits) max_lbits = lb;
    if (p.groups > max_groups) max_groups = p.groups;
    ft.first[t + 1] = ft.first[t] + p.groups;
    // the row-range id lists of this step's forward lookup: taken only when they were cut exactly like this launch's ranges
    const tt_id_buckets& bk = s.buckets;
    if (bk.pairs != nullptr) {
      TT_REQUIRE(bk.counts != nullptr, "tt_optimizer_step_ids_f32: table %d: buckets.pairs without buckets.counts", t);
      TT_REQUIRE(ft.bk_cap > 0 && bk.groups == p.groups && bk.width == p.width && bk.cap == ft.bk_cap && p.groups <= kBucketGroupsMax,
                 "tt_optimizer_step_ids_f32: table %d: id buckets (groups %d width %u cap %d) do not match this step's row ranges "
                 "(groups %d width %u cap %d: tt_optimizer_ids_geometry)", t, bk.groups, bk.width, bk.cap, p.groups, p.width, ft.bk_cap);
      TT_REQUIRE(t == 0 || tables[0].buckets.pairs == nullptr || tables[0].buckets.gen == bk.gen,
                 "tt_optimizer_step_ids_f32: the tables' id buckets must carry one generation");
      ft.bk_counts[t] = bk.counts; ft.bk_pairs[t] = bk.pairs; ft.bk_gen = bk.gen;
    }
  }
  int rc = prepare_ws(a, ws, n_tables, dim, n_ids, "tt_optimizer_step_ids_f32");
  if (rc != TT_OK) return rc;
  tt::SegTable tbl{};
  int64_t max_count = 0;
  for (int i = 0; i < n_segs; ++i) {
    const tt_dense_seg& s = segs[i];
    TT_REQUIRE(s.count > 0 && s.n_slabs >= 1 && s.grad_slabs != nullptr && s.param != nullptr, "tt_optimizer_step_ids_f32: segment %d: bad count/slabs/param", i);
    TT_REQUIRE(opt == TT_OPT_SGD || s.accum != nullptr, "tt_optimizer_step_ids_f32: segment %d: Adagrad needs accum", i);
    tbl.seg[i] = s;
    if (s.count > max_count) max_count = s.count;
  }
  const int dim4 = dim / 4;
  int lpr_log2 = 0;
  while ((1 << lpr_log2) < dim4 && lpr_log2 < 6) ++lpr_log2;
  (void)max_count; (void)max_groups;
  const int64_t gx = (int64_t)ft.first[n_tables] + ft.seg_first[n_segs];
  ft.cap = big ? tt::kPartSortMaxIds : (int32_t)((n_ids + 1023) / 1024 * 1024);
  const bool nine = (max_lbits + 8) / 9 < (max_lbits + 7) / 8;     // digits of the hot-range radix passes (csrc/sort.hip)
  const bool small = n_ids <= 8 * 1024;
  const int lds = tt::part_sort_lds_bytes(ft.cap, nine ? 512 : 256);
  hipStream_t stream = tt::as_stream(stream_);
  auto go = [&](auto kern) -> int {
    if (lds > 64 * 1024 &&
        hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess)
      return tt::fail(TT_ERR_LAUNCH, "tt_optimizer_step_ids_f32: hipFuncSetAttribute(LDS %d) failed", lds);
    tt::launch("optimizer", kern, dim3((unsigned)gx), dim3(1024), lds, stream, a, ft, n_tables, dim4, lpr_log2, tbl, n_segs, (int)ft.seg_first[n_segs], lr, eps);
    return tt::check_launch("tt_optimizer_step_ids_f32");
  };
  if (big) {
    if (opt == TT_OPT_SGD) return nine ? go(optimizer_ids_big_kernel<TT_OPT_SGD, 9>) : go(optimizer_ids_big_kernel<TT_OPT_SGD, 8>);
    return nine ? go(optimizer_ids_big_kernel<TT_OPT_ADAGRAD, 9>) : go(optimizer_ids_big_kernel<TT_OPT_ADAGRAD, 8>);
  }
  if (opt == TT_OPT_SGD) {
    if (small) return nine ? go(optimizer_ids_kernel<TT_OPT_SGD, 9, 8>) : go(optimizer_ids_kernel<TT_OPT_SGD, 8, 8>);
    return nine ? go(optimizer_ids_kernel<TT_OPT_SGD, 9, 16>) : go(optimizer_ids_kernel<TT_OPT_SGD, 8, 16>);
  }
  if (small) return nine ? go(optimizer_ids_kernel<TT_OPT_ADAGRAD, 9, 8>) : go(optimizer_ids_kernel<TT_OPT_ADAGRAD, 8, 8>);
  return nine ? go(optimizer_ids_kernel<TT_OPT_ADAGRAD, 9, 16>) : go(optimizer_ids_kernel<TT_OPT_ADAGRAD, 8, 16>);
}

#ifdef TT_SORT_STAMPS
extern "C" int tt_debug_opt_stamps(unsigned long long* host_out, int n) {
  return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(tt::g_sort_stamps), sizeof(unsigned long long) * (size_t)n) == hipSuccess ? 0 : 2;
}
#endif
