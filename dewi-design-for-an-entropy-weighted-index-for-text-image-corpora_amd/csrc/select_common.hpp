// Workgroup-level selection primitives shared by the select / merge kernels and the batched
// bf16 path: LDS scratch, bitonic sort, rank sort and an exact MSB radix select (gfx950 only).
#pragma once
#include "common.hpp"
#include "launch.hpp"

namespace dewi {

struct SelectShared {
  uint64_t sel[kMaxSortCandidates];   // selected candidate keys, then sorted descending
  uint64_t sel2[kMaxSortCandidates];  // re-rank keys
  uint32_t val[kMaxSortCandidates];   // payload carried through the first sort (merge path)
  uint32_t hist[256];
  uint32_t wave_tot[4];
  uint32_t pick_digit, pick_above, pick_count, total;
  uint32_t count;
  uint32_t n_contrib;   // sorted-list route: lists that can hold a key at or above the bound
  uint64_t bound;
};

// Descending bitonic sort of p2 (power of two) keys in LDS, optional 32-bit payload.
template <bool WITH_VAL>
__device__ inline void bitonic_sort_desc(uint64_t* key, uint32_t* val, int p2) {
  const int tid = static_cast<int>(threadIdx.x), nt = static_cast<int>(blockDim.x);
  for (int size = 2; size <= p2; size <<= 1) {
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      __syncthreads();
      for (int t = tid; t < (p2 >> 1); t += nt) {
        const int lo = 2 * t - (t & (stride - 1));
        const int hi = lo + stride;
        const bool first_half = (lo & size) == 0;
        const uint64_t a = key[lo], b = key[hi];
        if (first_half ? (a < b) : (a > b)) {
          key[lo] = b;
          key[hi] = a;
          if constexpr (WITH_VAL) {
            const uint32_t va = val[lo];
            val[lo] = val[hi];
            val[hi] = va;
          }
        }
      }
    }
  }
  __syncthreads();
}

// Descending sort by ranking: thread t counts the keys that beat src[t] (broadcast LDS reads) and
// drops it at that position of dst.  O(n^2 / threads) work but a single barrier, which beats the
// log^2(n) barriers of the bitonic network for the few dozen keys the usual query ends with.
// Equal keys (only the empty key can repeat) are ordered by position.  src != dst.
constexpr int kRankSortMax = 256;

// Threads per key for the ranking loops: 1 up to 64 keys (the usual batch-1 query: one thread per
// key, no cross-lane traffic), otherwise the largest power of two such that every key still gets
// its own group of lanes (<= 64).  A group's lanes split the n comparisons and add their partial
// ranks with xor shuffles.
__device__ __forceinline__ int rank_split(int n, int nt) {
  if (n <= 64) return 1;
  int s = 1;
  while (s < 64 && 2 * s * n <= nt) s <<= 1;
  return s;
}

// UNIQUE_PADDED: the caller guarantees distinct keys and kKeyEmpty in src[n .. n+63], so a group's
// lanes run the same number of unmasked iterations of one 64-bit compare each.
template <bool UNIQUE_PADDED = false>
__device__ inline void rank_sort_desc(const uint64_t* src, uint64_t* dst, int n) {
  const int tid = static_cast<int>(threadIdx.x), nt = static_cast<int>(blockDim.x);
  const int S = rank_split(n, nt);
  if (S == 1) {
    for (int t = tid; t < n; t += nt) {
      const uint64_t mine = src[t];
      int rank = 0;
#pragma unroll 4
      for (int j = 0; j < n; ++j) {
        const uint64_t o = src[j];
        rank += (o > mine || (o == mine && j < t)) ? 1 : 0;
      }
      dst[rank] = mine;
    }
  } else {
    const int part = tid & (S - 1);
    for (int t = tid / S; t < n; t += nt / S) {   // the S lanes of a group share t (same trip count)
      const uint64_t mine = src[t];
      int rank = 0;
      const int iters = (n + S - 1) / S;   // UNIFORM trip count (a per-lane bound makes hipcc wait after
#pragma unroll 8                           // every LDS read)
      for (int i = 0; i < iters; ++i) {
        const int j = part + i * S;
        if constexpr (UNIQUE_PADDED) {
          rank += src[j] > mine ? 1 : 0;   // padding keys are 0: never greater
        } else {
          const uint64_t o = src[j < n ? j : n - 1];
          rank += (j < n && (o > mine || (o == mine && j < t))) ? 1 : 0;
        }
      }
      for (int m = 1; m < S; m <<= 1) rank += __shfl_xor(rank, m, kWave);
      if (part == 0) dst[rank] = mine;
    }
  }
  __syncthreads();
}

// Where a query's candidate keys live.  ArrayKeys: one dense array (kKeyEmpty entries are skipped).
// SegmentKeys: the batched matrix-core scan's output — `n_seg` per-workgroup segments of capacity
// `cap`, of which only the first count[seg] entries were written.
struct ArrayKeys {
  const uint64_t* p;
  int64_t n;
  // f(key) for every non-empty key, the work split over the nt threads of the workgroup
  template <class F>
  __device__ __forceinline__ void for_each(int tid, int nt, F f) const {
    for (int64_t i = tid; i < n; i += nt) {
      const uint64_t k = p[i];
      if (k != kKeyEmpty) f(k);
    }
  }
};
struct SegmentKeys {
  const uint64_t* p;       // this query's first segment
  const uint32_t* count;   // this query's first count
  int64_t seg_stride;      // keys between consecutive segments of the same query
  int64_t count_stride;    // counts between consecutive segments of the same query
  int n_seg, cap;
  bool raw;                // entries are (row << 32 | fp32 score bits) records: build the key on read
  // Threads take (segment, lane-in-group) pairs: 4 consecutive threads share a segment, so the few
  // valid records at the head of each of the hundreds of segments are read without touching the
  // empty tails.
  template <class F>
  __device__ __forceinline__ void for_each(int tid, int nt, F f) const {
    constexpr int kGroup = 4;
    const int sub = tid % kGroup;
    for (int seg = tid / kGroup; seg < n_seg; seg += nt / kGroup) {
      uint32_t c = count[seg * count_stride];
      c = c < static_cast<uint32_t>(cap) ? c : static_cast<uint32_t>(cap);
      const uint64_t* sp = p + seg * seg_stride;
      for (uint32_t j = sub; j < c; j += kGroup) {
        const uint64_t v = sp[j];
        const uint64_t k = raw ? make_key(__uint_as_float(static_cast<uint32_t>(v)), static_cast<uint32_t>(v >> 32)) : v;
        if (k != kKeyEmpty) f(k);
      }
    }
  }
};

// Threshold T such that exactly `kth` of the non-empty keys are >= T (keys are unique).  If fewer
// than `kth` non-empty keys exist, returns 1 (every non-empty key).  All threads return the same
// value.  MSB-first radix select, 8 bits per pass, early exit once a whole bin is taken.
template <class Keys>
__device__ inline uint64_t block_kth_largest(const Keys& keys, uint32_t kth, SelectShared& sh) {
  const int tid = static_cast<int>(threadIdx.x), nt = static_cast<int>(blockDim.x);
  const int lane = tid & 63, wave = tid >> 6;
  uint64_t prefix = 0;
  uint32_t remaining = kth;
  for (int shift = 56; shift >= 0; shift -= 8) {
    if (tid < 256) sh.hist[tid] = 0;
    __syncthreads();
    keys.for_each(tid, nt, [&](uint64_t k) {
      if (shift == 56 || (k >> (shift + 8)) == prefix) atomicAdd(&sh.hist[static_cast<uint32_t>(k >> shift) & 0xFFu], 1u);
    });
    __syncthreads();
    uint32_t h = 0, sfx = 0;
    if (tid < 256) {  // waves 0..3, all lanes active
      h = sh.hist[tid];
      sfx = h;
      for (int off = 1; off < 64; off <<= 1) {
        const uint32_t o = __shfl_down(sfx, off, kWave);
        if (lane + off < 64) sfx += o;
      }
      if (lane == 0) sh.wave_tot[wave] = sfx;
    }
    __syncthreads();
    if (tid < 256) {
      uint32_t above = 0;
      for (int w = wave + 1; w < 4; ++w) above += sh.wave_tot[w];
      const uint32_t incl = sfx + above;  // keys (under this prefix) with digit >= tid
      const uint32_t excl = incl - h;     // ... with digit > tid
      if (tid == 0) sh.total = incl;
      if (incl >= remaining && excl < remaining) {
        sh.pick_digit = static_cast<uint32_t>(tid);
        sh.pick_above = excl;
        sh.pick_count = h;
      }
    }
    __syncthreads();
    if (sh.total < remaining) return 1ull;  // only possible on the first pass: fewer keys than kth
    prefix = (prefix << 8) | sh.pick_digit;
    remaining -= sh.pick_above;
    const bool whole_bin = sh.pick_count == remaining;
    __syncthreads();  // pick_* are rewritten by the next pass
    if (whole_bin) return prefix << shift;
  }
  return prefix;
}

// ---------------------------------------------------------------------------------------------
// Exact k-th largest of a dense LDS array of unique 64-bit keys (the survivors of the batched
// matrix-core scan: a few thousand keys whose scores all lie in a narrow band above the query's
// threshold).  The byte-aligned select above spends its first passes on bits every key shares (and
// serialises on one histogram bin while doing so); here the common prefix of min and max is
// skipped and the remaining bits are taken 11 at a time: two passes for the usual query.
// ---------------------------------------------------------------------------------------------
struct WideRadixShared {
  uint32_t hist[2048];
  uint32_t wave_tot[kSelectThreads / kWave];
  uint32_t pick[4];          // [0] digit, [1] keys above it, [3] keys in it
  uint64_t red[2][kSelectThreads / kWave];
  uint64_t small[64];
};
constexpr uint32_t kSmallBin = 64;

__device__ inline uint64_t block_kth_largest_lds(const uint64_t* keys, int n, uint32_t kth, WideRadixShared& ws) {
  const int tid = static_cast<int>(threadIdx.x), nt = static_cast<int>(blockDim.x);
  const int lane = tid & 63, wave = tid >> 6, n_waves = nt >> 6;
  if (static_cast<uint32_t>(n) < kth) return 1ull;   // fewer keys than wanted: every key qualifies
  uint64_t lo = ~0ull, hi = 0ull;
  for (int i = tid; i < n; i += nt) {
    const uint64_t k = keys[i];
    lo = k < lo ? k : lo;
    hi = k > hi ? k : hi;
  }
  lo = wave_min_u64(lo);
  hi = wave_max_u64(hi);
  if (lane == 0) {
    ws.red[0][wave] = lo;
    ws.red[1][wave] = hi;
  }
  __syncthreads();
  for (int w = 0; w < n_waves; ++w) {
    const uint64_t a = ws.red[0][w], b = ws.red[1][w];
    lo = a < lo ? a : lo;
    hi = b > hi ? b : hi;
  }
  if (lo == hi) return lo;   // n == 1 (keys are unique)
  int top = 64 - __builtin_clzll(lo ^ hi);   // undecided low bits
  uint64_t prefix = top == 64 ? 0ull : hi >> top;
  uint32_t remaining = kth;
  while (top > 0) {
    const int bits = top < 11 ? top : 11;
    const int shift = top - bits;
    const uint32_t mask = (1u << bits) - 1u;
    for (int b = tid; b < 2048; b += nt) ws.hist[b] = 0;
    __syncthreads();
    for (int i = tid; i < n; i += nt) {
      const uint64_t k = keys[i];
      if (top == 64 || (k >> top) == prefix) atomicAdd(&ws.hist[static_cast<uint32_t>(k >> shift) & mask], 1u);
    }
    __syncthreads();
    // suffix sums: thread t owns bins 2t and 2t+1 (nt >= 1024 covers all 2048; fewer threads loop)
    uint32_t carry = 0;   // keys in bins above the chunk being scanned
    for (int base = 2048 - 2 * nt; ; base -= 2 * nt) {
      const int b0 = base + 2 * tid;
      const uint32_t h0 = b0 >= 0 ? ws.hist[b0] : 0u, h1 = b0 + 1 >= 0 ? ws.hist[b0 + 1] : 0u;
      uint32_t sfx = h0 + h1;
      for (int off = 1; off < 64; off <<= 1) {
        const uint32_t o = __shfl_down(sfx, off, kWave);
        if (lane + off < 64) sfx += o;
      }
      if (lane == 0) ws.wave_tot[wave] = sfx;
      __syncthreads();
      uint32_t above = carry;
      for (int w = wave + 1; w < n_waves; ++w) above += ws.wave_tot[w];
      const uint32_t incl1 = sfx + above - h0;  // keys with digit >= b0+1
      const uint32_t incl0 = sfx + above;       // keys with digit >= b0
      if (incl1 >= remaining && incl1 - h1 < remaining) {
        ws.pick[0] = static_cast<uint32_t>(b0 + 1);
        ws.pick[1] = incl1 - h1;
        ws.pick[3] = h1;
      } else if (incl0 >= remaining && incl1 < remaining) {
        ws.pick[0] = static_cast<uint32_t>(b0);
        ws.pick[1] = incl1;
        ws.pick[3] = h0;
      }
      uint32_t chunk_total = 0;
      for (int w = 0; w < n_waves; ++w) chunk_total += ws.wave_tot[w];
      carry += chunk_total;
      __syncthreads();
      if (carry >= remaining || base <= 0) break;
    }
    prefix = (prefix << bits) | ws.pick[0];
    remaining -= ws.pick[1];
    const uint32_t in_bin = ws.pick[3];
    __syncthreads();   // pick[] is rewritten below / by the next pass
    if (in_bin == remaining) return prefix << shift;   // the whole bin is taken
    top = shift;
    if (in_bin <= kSmallBin) {
      // A handful of keys left under this prefix: list them and rank them against each other
      // instead of running further 2048-bin passes.
      if (tid == 0) ws.pick[2] = 0;
      __syncthreads();
      for (int i = tid; i < n; i += nt) {
        const uint64_t k = keys[i];
        if ((k >> top) == prefix) ws.small[atomicAdd(&ws.pick[2], 1u)] = k;
      }
      __syncthreads();
      if (tid < static_cast<int>(in_bin)) {
        const uint64_t mine = ws.small[tid];
        uint32_t greater = 0;
        for (uint32_t j = 0; j < in_bin; ++j) greater += ws.small[j] > mine ? 1u : 0u;
        if (greater + 1 == remaining) ws.red[0][0] = mine;   // keys are unique: exactly one writer
      }
      __syncthreads();
      return ws.red[0][0];
    }
  }
  return prefix;
}

}  // namespace dewi
