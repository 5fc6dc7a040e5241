// Key-range partitioned LDS sort of one id list (csrc/sort.hip: tt_sparse_plan; csrc/sparse.hip: the fused optimizer).
#pragma once
#include "common.h"

namespace tt {

constexpr int kPartSortMaxIds = 16384;   // ids one workgroup can hold in LDS (positions are u16)
constexpr int kPartSortBigMaxIds = 65536;   // the fused optimizer's long-list form (part_scan_append_big ...): positions still fit u16

template <int DBITS>
__device__ __forceinline__ uint64_t match_any(uint32_t d) {
  // per bit: the lane's bit as an all-ones / all-zeros word (v_bfe_i32), one v_cmp (= the ballot), and for each half
  // of the mask  peers &= ~(ballot ^ word)  (v_xnor + v_and): 6 VALU instructions per bit
  uint32_t lo = ~0u, hi = ~0u;
#pragma unroll
  for (int b = 0; b < DBITS; ++b) {
    const int32_t word = ((int32_t)(d << (31 - b))) >> 31;
    const uint64_t m = __builtin_amdgcn_ballot_w64(word != 0);
    lo &= ~((uint32_t)m ^ (uint32_t)word);
    hi &= ~((uint32_t)(m >> 32) ^ (uint32_t)word);
  }
  return ((uint64_t)hi << 32) | lo;
}


// ---- G workgroups per table, NO inter-workgroup synchronisation ----
// One workgroup sorting 8192 ids takes 21-33 us (VALU-bound by the ballots).  Here workgroup g of a table owns the key
// range [g * width, (g+1) * width) of the rows (width = ceil(rows / G); the last group also takes the out-of-range
// sentinel): it scans ALL n ids (L2-resident, 64-128 KB), counts the keys below its range (= its offset in the sorted
// output) and compacts its own keys into LDS in position order (ballots + a scan of the per-(load, wave) counts), then
// sorts only those - n/G ~ 128 of them, ranked by counting in one step; a hot range (> 512 keys) by LSD radix passes over
// its local key bits - and writes them at its offset.  Every workgroup derives everything it needs from the ids
// themselves.  Skew only unbalances the work: one hot range is at worst the single-workgroup sort again (capacity = the
// whole list).  n <= 16384 (positions as u16, one workgroup's LDS can hold every id).
struct PartTable {
  const int64_t* ids;
  int64_t* sorted_ids;
  int32_t* order;
  int64_t num_rows;
  int32_t n, groups;
  uint32_t width, magic;     // key range of a group = width ids (the last group also takes the sentinel); magic = 2^32 / width + 1
  uint32_t sentinel;
};

#ifdef TT_SORT_STAMPS
static __device__ unsigned long long g_sort_stamps[1024 * 8];   // (one copy per translation unit: debug builds only)
#define SSTAMP(i) do { if (threadIdx.x == 0) tt::g_sort_stamps[((blockIdx.x + gridDim.x * blockIdx.y) % 1024) * 8 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define SSTAMP(i) do {} while (0)
#endif


// key range of a (clamped) key: key / width by multiply-high + one correction, capped to the last group (which also
// takes the sentinel 2^bits - 1 of the out-of-range ids); ids past n carry 0xffffffff and belong to no group
__device__ __forceinline__ uint32_t bucket_of(uint32_t key, const PartTable& t) {
  if (key == 0xffffffffu) return 0xffffffffu;
  uint32_t q = __umulhi(key, t.magic);
  q -= (q * t.width > key) ? 1u : 0u;
  return q < (uint32_t)t.groups ? q : (uint32_t)t.groups - 1u;
}


template <int DBITS, int RMAX, bool TO_GLOBAL>
__device__ __forceinline__ void part_local_sort(const PartTable& t, uint32_t* keys, uint16_t* poss, uint32_t* cnt, uint32_t* wtot,
                                                uint32_t m, uint32_t offset, uint32_t base_key, int npass, int ppass = 0,
                                                uint32_t diff = 0xffffffffu) {
  // ppass > 0 (a list in NO particular order: part_sort_hot_unordered): the first ppass passes run over the digits of the batch
  // POSITION, the rest over the key's - the result is ordered by (key, position) as if the list had started in position order.
  // diff: bits in which the keys of the list differ at all (default: unknown = all).  A key pass over a digit that is the same
  // in every key is skipped - a range that is ONE id thousands of times (what makes a data-cut range hot) takes the position
  // passes only; the padding's position is all-ones for that (it ties with position 65535 at most, behind which it started).
  constexpr int RADIX = 1 << DBITS;
  constexpr int W = 16, T = 1024;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  // element e = (w * rounds + r) * 64 + lane
  const int rounds = (int)((m + T - 1) / T);
  uint32_t key[RMAX], pos[RMAX];
#pragma unroll
  for (int r = 0; r < RMAX; ++r) {
    key[r] = 0xffffffffu; pos[r] = ppass > 0 ? 0xffffffffu : 0u;   // (padding: last under the position passes too)
    if (r < rounds) {
      const uint32_t e = (uint32_t)((w * rounds + r) * 64 + lane);
      if (e < m) { key[r] = keys[e]; pos[r] = poss[e]; }
    }
  }
  uint32_t* mycnt = cnt + w * RADIX;
  int plast = npass - 1;                                    // the last pass that runs (TO_GLOBAL: it writes the output)
  while (plast >= ppass && ((diff >> ((plast - ppass) * DBITS)) & (RADIX - 1)) == 0u) --plast;
  if (plast < 0) plast = 0;                                 // (a list of equal keys in position order: one pass, over digit 0)
  for (int p = 0; p <= plast; ++p) {
    const bool by_pos = p < ppass;
    const int shift = (by_pos ? p : p - ppass) * DBITS;
    if (!by_pos && p < plast && ((diff >> shift) & (RADIX - 1)) == 0u) continue;
    __syncthreads();                                        // the loads above / of the previous pass are done
    for (int j = lane; j < RADIX; j += 64) mycnt[j] = 0u;
    // SLIM (the 16-round instantiation: a hot row range of more than 6144 keys in a list of up to 16384): one element at a
    // time from digit to slot - the LDS atomic's round trip is paid per element instead of once per pass, and the digit is
    // recomputed where it is needed - so that only key / position / slot arrays are live across the pass: 48 VGPRs instead
    // of 96 (with the three-loop form this instantiation spilled 92 B per lane inside the 128-VGPR budget of a 1024-thread
    // workgroup).  The shorter instantiations keep the three loops: their atomics overlap.
    constexpr bool SLIM = RMAX > 8;
    auto digit = [&](int r) -> uint32_t { return ((by_pos ? pos[r] : key[r]) >> shift) & (RADIX - 1); };
    uint32_t dg[SLIM ? 1 : RMAX], rk[SLIM ? 1 : RMAX], lead[SLIM ? 1 : RMAX], old[RMAX];
    if constexpr (SLIM) {
#pragma unroll
      for (int r = 0; r < RMAX; ++r) {
        old[r] = 0u;
        if (r < rounds) {
          const uint32_t d = digit(r);
          const uint64_t peers = match_any<DBITS>(d);
          const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(peers >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)peers, 0u));
          uint32_t o = (uint32_t)__popcll(peers);
          if (rank == 0u) o = atomicAdd(&mycnt[d], o);
          old[r] = (uint32_t)__shfl((int)o, (int)((uint32_t)__ffsll((unsigned long long)peers) - 1u)) + rank;
        }
      }
    } else {
#pragma unroll
      for (int r = 0; r < RMAX; ++r) {
        if (r < rounds) {
          dg[r] = digit(r);
          const uint64_t peers = match_any<DBITS>(dg[r]);
          rk[r] = __builtin_amdgcn_mbcnt_hi((uint32_t)(peers >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)peers, 0u));
          lead[r] = (uint32_t)__ffsll((unsigned long long)peers) - 1u;
          old[r] = (uint32_t)__popcll(peers);
        }
      }
#pragma unroll
      for (int r = 0; r < RMAX; ++r)
        if (r < rounds && rk[r] == 0u) old[r] = atomicAdd(&mycnt[dg[r]], old[r]);
#pragma unroll
      for (int r = 0; r < RMAX; ++r)
        if (r < rounds) old[r] = (uint32_t)__shfl((int)old[r], (int)lead[r]) + rk[r];
    }
    auto bucket = [&](int r) -> uint32_t { if constexpr (SLIM) return mycnt[digit(r)]; else return mycnt[dg[r]]; };
    __syncthreads();
    uint32_t v[W], total = 0u;
    if (tid < RADIX) {
#pragma unroll
      for (int ww = 0; ww < W; ++ww) v[ww] = cnt[ww * RADIX + tid];
#pragma unroll
      for (int ww = 0; ww < W; ++ww) {
        const uint32_t x = v[ww];
        v[ww] = total;
        total += x;
      }
    }
    uint32_t inc2 = total;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const uint32_t up = (uint32_t)__shfl_up((int)inc2, o);
      if (lane >= o) inc2 += up;
    }
    if (lane == 63) wtot[w] = inc2;
    __syncthreads();
    if (tid < RADIX) {
      uint32_t base = inc2 - total;
      for (int ww = 0; ww < w; ++ww) base += wtot[ww];
#pragma unroll
      for (int ww = 0; ww < W; ++ww) cnt[ww * RADIX + tid] = base + v[ww];
    }
    __syncthreads();
    if (p < plast || !TO_GLOBAL) {              // (!TO_GLOBAL: the last pass too leaves the sorted pairs in LDS)
#pragma unroll
      for (int r = 0; r < RMAX; ++r)
        if (r < rounds) {
          const uint32_t dst = bucket(r) + old[r];
          keys[dst] = key[r];
          poss[dst] = (uint16_t)pos[r];
        }
      __syncthreads();
#pragma unroll
      for (int r = 0; r < RMAX; ++r)
        if (r < rounds) {
          const uint32_t e = (uint32_t)((w * rounds + r) * 64 + lane);
          key[r] = keys[e];            // slots >= m hold the padding (all-ones key, sorted last by every pass)
          pos[r] = poss[e];
        }
    } else {
#pragma unroll
      for (int r = 0; r < RMAX; ++r)
        if (r < rounds) {
          const uint32_t dst = bucket(r) + old[r];
          if (dst < m) {
            t.sorted_ids[offset + dst] = (int64_t)(key[r] + base_key);
            t.order[offset + dst] = (int32_t)pos[r];
          }
        }
    }
  }
}


// LDS layout of one sorting workgroup (uint32 words from `smem`; part_sort_lds_bytes): keys [cap + 64] | counters
// [16][RADIX] | per-(load, wave) counts [16][16] | 2 x 16 scan partials | 16 counters | positions u16 [cap]
__host__ __device__ inline int part_sort_lds_bytes(int cap, int radix) { return (cap + 64 + 16 * radix + 16 * 16 + 48) * 4 + cap * 2; }
__device__ __forceinline__ uint32_t* part_keys(uint32_t* smem) { return smem; }
// !TO_GLOBAL, m <= 512 (ranked by counting): rank of the i-th pair of the UNORDERED list the workgroup appended (u16 [512],
// in the radix counters' area, which that path does not use) - lets a caller that requested rows per unordered pair find
// the pair's sorted slot afterwards
__device__ __forceinline__ uint16_t* part_ranks(uint32_t* smem, int cap) { return reinterpret_cast<uint16_t*>(smem + cap + 64); }
constexpr uint32_t kPartRankMax = 512u;                     // lists up to this length are ranked by counting
template <int DBITS>
__device__ __forceinline__ uint16_t* part_poss(uint32_t* smem, int cap) {
  return reinterpret_cast<uint16_t*>(smem + cap + 64 + 16 * (1 << DBITS) + 16 * 16 + 48);
}

// The body of one sorting workgroup (1024 threads), group g of table t, in three pieces (part_sort_body composes them; the
// fused optimizer, csrc/sparse.hip, calls them itself so that it can request rows between the scan and the ranking):
//   part_scan_append   every id of the table is read and clamped, the keys of the group's row range are appended to the LDS
//                      list (unordered); returns m = their number (workgroup-uniform), offset = keys sorting before the group
//   part_rank_small    m <= kPartRankMax: stable rank by counting
//   part_sort_hot      a hot range: the list is rebuilt in position order and LSD-radix-sorted
// TO_GLOBAL: the sorted (id, position) pairs go to t.sorted_ids / t.order at the group's offset.  !TO_GLOBAL: they stay in
// LDS - part_keys(smem)[0..m) local keys ascending (id = base_key + key), part_poss(smem, cap)[0..m) their batch positions
// (and, ranked lists, part_ranks(smem, cap)[i] = the sorted slot of the i-th appended pair) - and every thread returns
// after a barrier.  JMAX: ids per thread of the scan (8: lists <= 8192, 16: <= 16384).
template <int JMAX>
struct PartScan {
  uint32_t kj[JMAX];       // this thread's clamped keys (0xffffffff past n)
  uint64_t mm[JMAX];       // per load: which lanes of the wave hold a key of this group's range
};

// DROP_OOR: an id outside [0, num_rows) - the exchange's -1 padding, a corrupt id - belongs to NO range (the fused optimizer:
// such an id has no row to update, and a half-empty fixed-capacity list would otherwise pile its padding onto the last
// group as one hot "sentinel" range).  !DROP_OOR: it takes the sentinel key and sorts last (the plan's sorted_ids hold all n).
// on_mine(mask, key, position): called by ALL lanes of a wave for every load j, once every id of the thread has been consumed;
// mask (wave-uniform) = the lanes whose id j falls in this group's range, key / position = the calling lane's own (the
// fused optimizer starts the rows' trip from HBM there, two phases before it can consume them).
struct NoHook {
  __device__ __forceinline__ void operator()(uint64_t, uint32_t, uint32_t) const {}
};
template <int DBITS, int JMAX, bool DROP_OOR = false, typename Hook = NoHook>
__device__ __forceinline__ uint32_t part_scan_append(const PartTable& t, const int g, const int cap, uint32_t* smem, PartScan<JMAX>& sc,
                                                     uint32_t& offset, uint32_t& base_key, Hook on_mine = Hook()) {
  constexpr int RADIX = 1 << DBITS;
  constexpr int W = 16, T = 1024;
  uint32_t* keys = smem;                                    // [cap]
  uint32_t* cnt = keys + cap + 64;                          // [W][RADIX]   (keys: cap + 64 slots of padding)
  uint32_t* cjw = cnt + W * RADIX;                          // [JMAX][W] own-range keys per (load, wave) -> exclusive bases
  uint32_t* wtot = cjw + 16 * W;                            // [16] scan partials (cjw sized for JMAX = 16: one layout)
  uint32_t* wbel = wtot + 16;                               // [16] keys below the range, per wave
  uint32_t* ctr = wbel + 16;                                // [16] ctr[0] = keys of this range appended so far, ctr[1] = keys below
  uint16_t* poss = reinterpret_cast<uint16_t*>(ctr + 16);   // [cap]
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  // the descriptor's fields as values (read through the reference inside the loops below they stayed scalar LOADS, which the
  // compiler placed - and waited for - between the id loads)
  // (global address space spelled out: a pointer that went through the caller's SGPR pin is a generic one, and flat loads
  // count against lgkmcnt as well - every LDS wait would wait for them)
  const __attribute__((address_space(1))) int64_t* ids = (const __attribute__((address_space(1))) int64_t*)t.ids;
  const int64_t num_rows = t.num_rows;
  const int n = t.n, J = (n + T - 1) / T;
  const uint32_t sentinel = t.sentinel, width = t.width;
  const int groups = t.groups;
  SSTAMP(0);
  if (n <= 0) { offset = 0u; base_key = 0u; return 0u; }   // (uniform; the host never launches an empty table)

  // ---- scan: every id of the table; clamp; which range ----
  // ALL of this thread's ids are requested before any is looked at: unconditional loads from clamped addresses.  (r03 ISA
  // audit: written as `cond ? ids[i] : ...` with the range checks in the same statement, each load sat in its own branch
  // with `s_waitcnt vmcnt(0)` behind it - 8 dependent L2 round trips, the 3.2 us "ids landed" stamp of every workgroup.)
  int64_t raw[JMAX];
  const int last = n - 1;
#pragma unroll
  for (int j = 0; j < JMAX; ++j) {
    const int i = j * T + tid;
    raw[j] = ids[i < last ? i : last];
  }
#pragma unroll
  for (int j = 0; j < JMAX; ++j) {
    const int i = j * T + tid;
    const int64_t id = raw[j];
    const uint32_t k = (id >= 0 && id < num_rows) ? (uint32_t)id : (DROP_OOR ? 0xffffffffu : sentinel);   // (num_rows < 2^31: no key is all-ones)
    sc.kj[j] = i < n ? k : 0xffffffffu;                      // past n: no range
  }
  if (tid == 0) { ctr[0] = 0u; ctr[1] = 0u; }
  __syncthreads();                                          // counters zeroed; the ids have landed (the barrier's fence waits for them)
  SSTAMP(7);
  // ---- group of every id; own-range masks (wave-uniform); the wave appends its own keys to the LDS list at a base drawn
  // from one LDS counter - in NO particular order across waves: the usual group (<= 512 keys) is ranked by (key, position)
  // below, which needs no order.  The per-(load, wave) counts are kept for the ordered path of a hot range. ----
  // (r03: a key is classified against THIS group's range [lo, lo + span) directly - one subtract and two compares per id -
  // instead of computing its group number first (multiply-high, multiply, compare, subtract, min: bucket_of, kept for the
  // routing kernels).  The last group's span runs up to the out-of-range sentinel; the all-ones key of a slot past n is in
  // no range and not below any.  Every workgroup classifies all n ids, so these instructions ARE the scan's VALU time.)
  base_key = (uint32_t)g * width;
  const uint32_t span = (g == groups - 1) ? (base_key <= sentinel ? sentinel - base_key + 1u : 0u) : width;
  uint32_t below = 0u, wave_mine = 0u;
#pragma unroll
  for (int j = 0; j < JMAX; ++j) {
    sc.mm[j] = 0ull;
    if (j < J) {
      sc.mm[j] = __builtin_amdgcn_ballot_w64(sc.kj[j] - base_key < span);
      below += (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(sc.kj[j] < base_key));
      const uint32_t c1 = (uint32_t)__popcll(sc.mm[j]);
      wave_mine += c1;
      if (lane == 0) cjw[j * W + w] = c1;
    }
  }
  // (the hook runs when EVERY id of the thread has been consumed: its loads are invisible to the compiler's counter model -
  // inline asm - and the vmcnt(N) it places in front of id j+1 would otherwise wait for the hook's loads of id j too:
  // memory operations retire in order.  Measured that way: classification 0.8 -> 3 us.)
#pragma unroll
  for (int j = 0; j < JMAX; ++j)
    if (j < J && sc.mm[j] != 0ull) on_mine(sc.mm[j], sc.kj[j], (uint32_t)(j * T + tid));
  SSTAMP(4);
  uint32_t wbase = 0u;
  if (lane == 0) {
    wbel[w] = below;
    wbase = atomicAdd(&ctr[0], wave_mine);
    atomicAdd(&ctr[1], below);
  }
  wbase = (uint32_t)__builtin_amdgcn_readfirstlane((int)wbase);
#pragma unroll
  for (int j = 0; j < JMAX; ++j) {
    if (j < J) {
      if ((sc.mm[j] >> lane) & 1ull) {
        const uint32_t dst = wbase + __builtin_amdgcn_mbcnt_hi((uint32_t)(sc.mm[j] >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)sc.mm[j], 0u));
        keys[dst] = sc.kj[j] - base_key;
        poss[dst] = (uint16_t)(j * T + tid);
      }
      wbase += (uint32_t)__popcll(sc.mm[j]);
    }
  }
  SSTAMP(1);
  __syncthreads();
  SSTAMP(2);
  // (readfirstlane: the counts are workgroup-uniform, but read from LDS the compiler takes them for per-lane values and
  // every branch on m for a divergent one - both sides of it then run back to back under masks, and the registers of the
  // hot path (kj, mm: 24-48 VGPRs) stay allocated through the ranking path and the caller's apply)
  const uint32_t m = (uint32_t)__builtin_amdgcn_readfirstlane((int)ctr[0]);
  offset = (uint32_t)__builtin_amdgcn_readfirstlane((int)ctr[1]);
  return m;
}

// The number of the table's ids that sort below `base_key` (= the first sorted slot of the row range that starts there), counted
// from the ids by the whole workgroup: what part_scan_append returns as `offset`, for a workgroup that got its own keys from
// the forward pass's row-range list (csrc/sparse.hip) and finds it needs the ranked path after all.  Out-of-range ids sort last.
template <int JMAX>
__device__ __forceinline__ uint32_t part_count_below(const PartTable& t, const uint32_t base_key, uint32_t* s_word) {
  constexpr int T = 1024;
  const int tid = threadIdx.x, lane = tid & 63;
  const __attribute__((address_space(1))) int64_t* ids = (const __attribute__((address_space(1))) int64_t*)t.ids;
  const int64_t num_rows = t.num_rows;
  const int n = t.n, last = n - 1;
  int64_t raw[JMAX];
#pragma unroll
  for (int j = 0; j < JMAX; ++j) {
    const int i = j * T + tid;
    raw[j] = ids[i < last ? i : last];
  }
  if (tid == 0) *s_word = 0u;
  uint32_t below = 0u;
#pragma unroll
  for (int j = 0; j < JMAX; ++j) {
    const int i = j * T + tid;
    const bool b = i < n && raw[j] >= 0 && raw[j] < num_rows && (uint32_t)raw[j] < base_key;
    below += (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(b));
  }
  __syncthreads();
  if (lane == 0 && below != 0u) atomicAdd(s_word, below);
  __syncthreads();
  return (uint32_t)__builtin_amdgcn_readfirstlane((int)*s_word);
}

// Workgroup barrier for LDS traffic only: waits for this wave's LDS operations, NOT for its global loads in flight
// (__syncthreads() is a fence + barrier: `s_waitcnt vmcnt(0)` first, which would park the rows the fused optimizer has
// just requested in front of the ranking instead of letting them land under it).
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// ---- the usual case (n / groups ~ 64-128 keys): rank by counting.  P = 1024 / pow2(m) threads per element e, each
// compares a 1/P slice of the list: rank = #(smaller keys) + #(equal keys at a smaller batch position) = the stable
// rank, whatever order the list is in. ----
template <int DBITS, bool TO_GLOBAL>
__device__ __forceinline__ void part_rank_small(const PartTable& t, const int cap, uint32_t* smem, const uint32_t m, const uint32_t offset,
                                                const uint32_t base_key) {
  constexpr int T = 1024;
  uint32_t* keys = smem;
  uint16_t* poss = part_poss<DBITS>(smem, cap);
  const int tid = threadIdx.x;
  uint32_t lp = 4;                                        // log2 P: 16 threads per element up to 64 keys ... 2 up to 512
  while ((T >> lp) < m) --lp;
  const uint32_t e = (uint32_t)tid >> lp, sub = (uint32_t)tid & ((1u << lp) - 1u), P = 1u << lp;
  const uint32_t ee = e < m ? e : 0u;
  const uint32_t key = keys[ee], pos = poss[ee];
  uint32_t c = 0u;
  // 4 pairs per iteration (one ds_read_b128 + one ds_read_b64); slots past m (inside the LDS arrays) are masked out
  for (uint32_t j = 4u * sub; j < m; j += 4u * P) {
    const uint4 k4 = *reinterpret_cast<const uint4*>(keys + j);
    const uint2 p2 = *reinterpret_cast<const uint2*>(poss + j);
    const uint32_t p0 = p2.x & 0xffffu, p1 = p2.x >> 16, pq2 = p2.y & 0xffffu, p3 = p2.y >> 16;
    c += (k4.x < key || (k4.x == key && p0 < pos)) ? 1u : 0u;
    c += (j + 1u < m && (k4.y < key || (k4.y == key && p1 < pos))) ? 1u : 0u;
    c += (j + 2u < m && (k4.z < key || (k4.z == key && pq2 < pos))) ? 1u : 0u;
    c += (j + 3u < m && (k4.w < key || (k4.w == key && p3 < pos))) ? 1u : 0u;
  }
  for (uint32_t o = 1u; o < P; o <<= 1) c += (uint32_t)__shfl_xor((int)c, (int)o);
  if constexpr (TO_GLOBAL) {
    if (sub == 0u && e < m) {
      t.sorted_ids[offset + c] = (int64_t)(key + base_key);
      t.order[offset + c] = (int32_t)pos;
    }
  } else {
    lds_barrier();                                        // every thread has read the pairs it compares with
    if (sub == 0u && e < m) { keys[c] = key; poss[c] = (uint16_t)pos; part_ranks(smem, cap)[e] = (uint16_t)c; }
    lds_barrier();
  }
  SSTAMP(5);
}

// ---- a hot key range (> kPartRankMax keys: skewed ids): the list is rebuilt IN POSITION ORDER (exclusive scan of the
// per-(load, wave) counts, j major / wave minor) - the LSD radix passes are stable with respect to it ----
template <int DBITS, int JMAX, bool TO_GLOBAL>
__device__ __forceinline__ void part_sort_hot(const PartTable& t, const int g, const int cap, uint32_t* smem, const PartScan<JMAX>& sc,
                                              const uint32_t m, const uint32_t offset, const uint32_t base_key) {
  constexpr int RADIX = 1 << DBITS;
  constexpr int W = 16, T = 1024;
  uint32_t* keys = smem;
  uint32_t* cnt = keys + cap + 64;
  uint32_t* cjw = cnt + W * RADIX;
  uint32_t* wtot = cjw + 16 * W;
  uint16_t* poss = part_poss<DBITS>(smem, cap);
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int n = t.n, J = (n + T - 1) / T;
  __syncthreads();                                          // everyone has read ctr; the unordered list may be overwritten
  {
    uint32_t val = (tid < J * W) ? cjw[tid] : 0u, incl = val;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const uint32_t up = (uint32_t)__shfl_up((int)incl, o);
      if (lane >= o) incl += up;
    }
    if (lane == 63) wtot[w] = incl;
    __syncthreads();
    uint32_t pre = 0u;
    for (int ww = 0; ww < w; ++ww) pre += wtot[ww];
    __syncthreads();                                        // everyone has read cjw / wtot
    if (tid < J * W) cjw[tid] = pre + incl - val;
    __syncthreads();
#pragma unroll
    for (int j = 0; j < JMAX; ++j) {
      if (j < J && ((sc.mm[j] >> lane) & 1ull)) {
        const uint32_t dst = cjw[j * W + w] + __builtin_amdgcn_mbcnt_hi((uint32_t)(sc.mm[j] >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)sc.mm[j], 0u));
        keys[dst] = sc.kj[j] - base_key;
        poss[dst] = (uint16_t)(j * T + tid);
      }
    }
    if (tid < 64) keys[m + (uint32_t)tid] = 0xffffffffu;        // padding slots sort last in every radix pass
    __syncthreads();
  }
  SSTAMP(4);
  // ---- LSD radix sort of the m compacted (local key, position) pairs; register arrays by rounds needed; passes by the
  // bits of this group's largest local key (the last group's is the sentinel's) ----
  const uint32_t local_max = g == t.groups - 1 ? t.sentinel - base_key : t.width - 1u;
  const int lbits = 32 - __builtin_clz(local_max | 1u);
  const int npass = (lbits + DBITS - 1) / DBITS;
  if constexpr (JMAX <= 8) {
    if (m <= 2u * T) part_local_sort<DBITS, 2, TO_GLOBAL>(t, keys, poss, cnt, wtot, m, offset, base_key, npass);
    else part_local_sort<DBITS, 8, TO_GLOBAL>(t, keys, poss, cnt, wtot, m, offset, base_key, npass);
  } else {
    if (m <= 2u * T) part_local_sort<DBITS, 2, TO_GLOBAL>(t, keys, poss, cnt, wtot, m, offset, base_key, npass);
    else if (m <= 6u * T) part_local_sort<DBITS, 6, TO_GLOBAL>(t, keys, poss, cnt, wtot, m, offset, base_key, npass);
    else part_local_sort<DBITS, 16, TO_GLOBAL>(t, keys, poss, cnt, wtot, m, offset, base_key, npass);
  }
}

template <int DBITS, int JMAX, bool TO_GLOBAL>
__device__ __forceinline__ uint32_t part_sort_body(const PartTable& t, const int g, const int cap, uint32_t* smem, uint32_t& offset,
                                                   uint32_t& base_key) {
  PartScan<JMAX> sc;
  const uint32_t m = part_scan_append<DBITS, JMAX>(t, g, cap, smem, sc, offset, base_key);
  if (m == 0u) return 0u;
  if (m <= kPartRankMax) part_rank_small<DBITS, TO_GLOBAL>(t, cap, smem, m, offset, base_key);
  else part_sort_hot<DBITS, JMAX, TO_GLOBAL>(t, g, cap, smem, sc, m, offset, base_key);
  return m;
}

// ---- lists of more than kPartSortMaxIds ids (the fused optimizer; n <= kPartSortBigMaxIds: positions are still u16) ----
// The LDS list keeps its 16384 slots and the ids are scanned in chunks of 16 x 1024; nothing of the scan is kept in registers
// across chunks: a hot range (> kPartRankMax ids) is sorted from the UNORDERED list (position passes first,
// part_sort_hot_unordered), and a range with more ids than the LDS list holds in a global scratch (part_sort_global).
// (Tried in r04 and dropped: ranges cut BY THE DATA - every workgroup sorts the same 512 sampled ids and takes two quantiles of
// the sample as its range.  It evens out a power-law batch (optimizer launch 1124 -> 258-322 us at 2 x 32768 ids, dim 256),
// but 4 samples per range leave ranges of 2-3x the mean, and the slowest workgroup is the launch: uniform ids 93 -> 128-155 us;
// as a PLAN it took 58-85 us against 53 us for r03's chunk sorts + merge - ranking 256-1024 ids by counting is 17-68 us of VALU
// time on one CU.  What handles skew instead is the trainer's choice of path from a lagged load probe, tt_id_range_load.)
__device__ __forceinline__ uint32_t part_range_span(const PartTable& t, const int g, uint32_t& base_key) {
  base_key = (uint32_t)g * t.width;
  return (g == t.groups - 1) ? (base_key <= t.sentinel ? t.sentinel - base_key + 1u : 0u) : t.width;
}

// Scan + unordered append of the range [base_key, base_key + span).  Returns m = the range's ids (may exceed cap: then only
// the first cap appended pairs are in LDS and the caller must take the global path), offset = ids sorting below the range.
// DROP_OOR as in part_scan_append.
// diff = the bits in which the range's local keys differ at all (0: the range is ONE id; the radix paths skip those digits).
template <int DBITS, bool DROP_OOR>
__device__ __forceinline__ uint32_t part_scan_append_big(const PartTable& t, const uint32_t base_key, const uint32_t span, const int cap,
                                                         uint32_t* smem, uint32_t& offset, uint32_t& diff) {
  constexpr int RADIX = 1 << DBITS;
  constexpr int W = 16, T = 1024, JC = 16;
  uint32_t* keys = smem;
  uint32_t* ctr = keys + cap + 64 + W * RADIX + 16 * W + 32;
  uint16_t* poss = reinterpret_cast<uint16_t*>(ctr + 16);
  const int tid = threadIdx.x, lane = tid & 63;
  const __attribute__((address_space(1))) int64_t* ids = (const __attribute__((address_space(1))) int64_t*)t.ids;
  const int64_t num_rows = t.num_rows;
  const int n = t.n, last = n - 1;
  const uint32_t oor = DROP_OOR ? 0xffffffffu : t.sentinel;
  if (tid == 0) { ctr[0] = 0u; ctr[1] = 0u; ctr[2] = 0u; ctr[3] = 0xffffffffu; }
  __syncthreads();
  uint32_t k_or = 0u, k_and = 0xffffffffu;
  for (int c0 = 0; c0 < n; c0 += JC * T) {                  // (workgroup-uniform)
    int64_t raw[JC];
#pragma unroll
    for (int j = 0; j < JC; ++j) {
      const int i = c0 + j * T + tid;
      raw[j] = ids[i < last ? i : last];
    }
    // (the own-range ballots are taken twice - once for the counts, once for the slots - instead of being kept: 16 wave masks
    // are 32 SGPRs across the LDS atomic's round trip, and the kernel around this already spills SGPRs to VGPR lanes)
    uint32_t kj[JC];
    uint32_t below = 0u, wave_mine = 0u;
#pragma unroll
    for (int j = 0; j < JC; ++j) {
      const int i = c0 + j * T + tid;
      const int64_t id = raw[j];
      const uint32_t k = (id >= 0 && id < num_rows) ? (uint32_t)id : oor;
      kj[j] = i < n ? k : 0xffffffffu;                      // past n: no range (span never reaches the all-ones key)
      wave_mine += (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(kj[j] - base_key < span));
      below += (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(kj[j] < base_key));
    }
    uint32_t wbase = 0u;
    if (lane == 0) {
      wbase = atomicAdd(&ctr[0], wave_mine);
      atomicAdd(&ctr[1], below);
    }
    wbase = (uint32_t)__builtin_amdgcn_readfirstlane((int)wbase);
#pragma unroll
    for (int j = 0; j < JC; ++j) {
      const bool mine = kj[j] - base_key < span;
      const uint64_t mk = __builtin_amdgcn_ballot_w64(mine);
      if (mine) {
        const uint32_t dst = wbase + __builtin_amdgcn_mbcnt_hi((uint32_t)(mk >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mk, 0u));
        const uint32_t lk = kj[j] - base_key;
        k_or |= lk;
        k_and &= lk;
        if (dst < (uint32_t)cap) {
          keys[dst] = lk;
          poss[dst] = (uint16_t)(c0 + j * T + tid);
        }
      }
      wbase += (uint32_t)__popcll(mk);
    }
  }
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    k_or |= (uint32_t)__shfl_xor((int)k_or, o);
    k_and &= (uint32_t)__shfl_xor((int)k_and, o);
  }
  if (lane == 0) { atomicOr(&ctr[2], k_or); atomicAnd(&ctr[3], k_and); }
  __syncthreads();
  SSTAMP(2);
  const uint32_t m = (uint32_t)__builtin_amdgcn_readfirstlane((int)ctr[0]);
  offset = (uint32_t)__builtin_amdgcn_readfirstlane((int)ctr[1]);
  diff = (uint32_t)__builtin_amdgcn_readfirstlane((int)(ctr[2] & ~ctr[3]));
  return m;
}

// A hot range of the long-list form (kPartRankMax < m <= cap): LSD radix passes over the position digits, then the key digits.
template <int DBITS, bool TO_GLOBAL>
__device__ __forceinline__ void part_sort_hot_unordered(const PartTable& t, const uint32_t span, const int cap, uint32_t* smem,
                                                        const uint32_t m, const uint32_t offset, const uint32_t base_key, const uint32_t diff) {
  constexpr int RADIX = 1 << DBITS;
  constexpr int W = 16, T = 1024;
  uint32_t* keys = smem;
  uint32_t* cnt = keys + cap + 64;
  uint32_t* wtot = cnt + W * RADIX + 16 * W;
  uint16_t* poss = part_poss<DBITS>(smem, cap);
  __syncthreads();                                          // everyone has read the counters
  const int lbits = 32 - __builtin_clz((span - 1u) | 1u);
  const int pbits = 32 - __builtin_clz((uint32_t)(t.n - 1) | 1u);
  const int ppass = (pbits + DBITS - 1) / DBITS, npass = ppass + (lbits + DBITS - 1) / DBITS;
  if (m <= 2u * T) part_local_sort<DBITS, 2, TO_GLOBAL>(t, keys, poss, cnt, wtot, m, offset, base_key, npass, ppass, diff);
  else if (m <= 6u * T) part_local_sort<DBITS, 6, TO_GLOBAL>(t, keys, poss, cnt, wtot, m, offset, base_key, npass, ppass, diff);
  else part_local_sort<DBITS, 16, TO_GLOBAL>(t, keys, poss, cnt, wtot, m, offset, base_key, npass, ppass, diff);
}

// A range with more ids than the LDS list holds (m > cap: one id repeated more than 16384 times): its (local key << 16 |
// position) words are appended to a global scratch and sorted there by this ONE workgroup - LSD radix, 4-bit digits, every
// thread owns a contiguous block of the list and one column of the [16][1024] LDS histogram (count, scan in digit-major /
// thread-minor order, stable scatter).  L2-resident scratch, ~10 passes of two reads and one write per element: a hundred
// microseconds and more, on a path only a degenerate batch takes.  g0 / g1: m words each, private to this range (the callers
// pass scratch + offset).  Returns the buffer that holds the sorted words; ends with a barrier.
template <bool DROP_OOR>
__device__ __forceinline__ uint64_t* part_sort_global(const PartTable& t, const uint32_t base_key, const uint32_t span, uint32_t* smem,
                                                      uint64_t* g0, uint64_t* g1, const uint32_t m, const uint32_t diff) {
  constexpr int T = 1024, JC = 16;
  uint32_t* hist = smem;                                    // [16][T] (64 KB = the key list's 16384 slots)
  uint32_t* wtot = smem + 16 * T;                           // [16]  (the padding behind the key list)
  uint32_t* ctr = smem + 16 * T + 32;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const __attribute__((address_space(1))) int64_t* ids = (const __attribute__((address_space(1))) int64_t*)t.ids;
  const int64_t num_rows = t.num_rows;
  const int n = t.n, last = n - 1;
  const uint32_t oor = DROP_OOR ? 0xffffffffu : t.sentinel;
  __syncthreads();                                          // the LDS list is dead
  if (tid == 0) ctr[0] = 0u;
  __syncthreads();
  for (int c0 = 0; c0 < n; c0 += JC * T) {
    int64_t raw[JC];
#pragma unroll
    for (int j = 0; j < JC; ++j) {
      const int i = c0 + j * T + tid;
      raw[j] = ids[i < last ? i : last];
    }
#pragma unroll
    for (int j = 0; j < JC; ++j) {
      const int i = c0 + j * T + tid;
      const int64_t id = raw[j];
      const uint32_t k = i < n ? ((id >= 0 && id < num_rows) ? (uint32_t)id : oor) : 0xffffffffu;
      const bool mine = k - base_key < span;
      const uint64_t mk = __builtin_amdgcn_ballot_w64(mine);
      uint32_t wbase = 0u;
      if (lane == 0 && mk != 0ull) wbase = atomicAdd(&ctr[0], (uint32_t)__popcll(mk));
      wbase = (uint32_t)__builtin_amdgcn_readfirstlane((int)wbase);
      if (mine) {
        const uint32_t dst = wbase + __builtin_amdgcn_mbcnt_hi((uint32_t)(mk >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mk, 0u));
        g0[dst] = ((uint64_t)(k - base_key) << 16) | (uint64_t)(uint32_t)i;
      }
    }
  }
  const int lbits = 32 - __builtin_clz((span - 1u) | 1u);
  const int pbits = 32 - __builtin_clz((uint32_t)(n - 1) | 1u);
  const uint32_t E = (m + T - 1) / T;                       // elements per thread (<= 64)
  const uint32_t lo = (uint32_t)tid * E < m ? (uint32_t)tid * E : m, hi = lo + E < m ? lo + E : m;
  uint64_t* src = g0;
  uint64_t* dst = g1;
  const int ppass = (pbits + 3) / 4, npass = ppass + (lbits + 3) / 4;
  constexpr int U = 8;                                      // words a thread loads before it touches its histogram column
  for (int p = 0; p < npass; ++p) {
    const int shift = p < ppass ? 4 * p : 16 + 4 * (p - ppass);
    if (p >= ppass && ((diff >> (4 * (p - ppass))) & 15u) == 0u) continue;      // a digit every key of the range shares
#pragma unroll
    for (int d = 0; d < 16; ++d) hist[d * T + tid] = 0u;
    __syncthreads();                                        // the appended / scattered words of every thread are visible
    for (uint32_t e0 = lo; e0 < hi; e0 += U) {
      uint64_t v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) v[u] = src[e0 + u < hi ? e0 + u : hi - 1u];
#pragma unroll
      for (int u = 0; u < U; ++u)
        if (e0 + u < hi) hist[((uint32_t)(v[u] >> shift) & 15u) * T + tid] += 1u;       // (this thread's own column)
    }
    __syncthreads();
    // exclusive scan of the 16 * T counts in index order (digit major, thread minor): thread q takes indices [16 q, 16 q + 16)
    uint32_t x[16], total = 0u;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const uint32_t v = hist[16 * tid + k];
      x[k] = total;
      total += v;
    }
    uint32_t incl = total;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const uint32_t up = (uint32_t)__shfl_up((int)incl, o);
      if (lane >= o) incl += up;
    }
    if (lane == 63) wtot[w] = incl;
    __syncthreads();
    uint32_t base = incl - total;
    for (int ww = 0; ww < w; ++ww) base += wtot[ww];
#pragma unroll
    for (int k = 0; k < 16; ++k) hist[16 * tid + k] = base + x[k];
    __syncthreads();
    for (uint32_t e0 = lo; e0 < hi; e0 += U) {
      uint64_t v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) v[u] = src[e0 + u < hi ? e0 + u : hi - 1u];
#pragma unroll
      for (int u = 0; u < U; ++u)
        if (e0 + u < hi) {
          const uint32_t d = (uint32_t)(v[u] >> shift) & 15u;
          const uint32_t at = hist[d * T + tid];
          hist[d * T + tid] = at + 1u;
          dst[at] = v[u];
        }
    }
    uint64_t* sw = src; src = dst; dst = sw;
  }
  __syncthreads();
  return src;
}

}  // namespace tt
