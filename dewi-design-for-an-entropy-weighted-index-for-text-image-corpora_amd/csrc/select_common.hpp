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
__device__ inline void rank_sort_desc(const uint64_t* src, uint64_t* dst, int n) {
  const int tid = static_cast<int>(threadIdx.x), nt = static_cast<int>(blockDim.x);
  for (int t = tid; t < n; t += nt) {
    const uint64_t mine = src[t];
    int rank = 0;
    for (int j = 0; j < n; ++j) {
      const uint64_t o = src[j];
      rank += (o > mine || (o == mine && j < t)) ? 1 : 0;
    }
    dst[rank] = mine;
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

}  // namespace dewi
