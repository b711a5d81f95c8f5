// Pieces shared by the fp32 and bf16 corpus-scan kernels: 16-byte loads, the per-wave candidate
// list and the workgroup-level merge (gfx950 only).
#pragma once
#include "common.hpp"
#include "launch.hpp"

namespace dewi {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kMaxSlots = kMaxListCandidates / kWave;  // up to 4 key registers per lane per query

template <bool NT>
__device__ __forceinline__ f32x4 load_x4(const f32x4* p) {
  if constexpr (NT) return __builtin_nontemporal_load(p);
  return *p;
}

// ---------------------------------------------------------------------------------------------
// Per-wave top-c list.  Position p = slot*64 + lane is active when p < c.  `thr` is the smallest
// active key (the entry a better candidate replaces), `thr_s` its score for the cheap test.
// ---------------------------------------------------------------------------------------------
template <int kSlots>
struct WaveList {
  uint64_t key[kSlots];
  uint64_t thr;
  float thr_s;

  __device__ __forceinline__ void init(int c, int lane) {
#pragma unroll
    for (int s = 0; s < kSlots; ++s) key[s] = (s * kWave + lane) < c ? kKeyEmpty : kKeyInactive;
    thr = kKeyEmpty;
    thr_s = -__builtin_inff();
  }
  // `score` and `row` are wave-uniform.
  __device__ __forceinline__ void offer(float score, uint32_t row, int lane) {
    if (score < thr_s) return;  // common case; false for NaN so NaN rows reach the exact test
    const uint64_t k = make_key(score, row);
    if (k <= thr) return;
    bool placed = false;
#pragma unroll
    for (int s = 0; s < kSlots; ++s) {
      const unsigned long long m = __ballot(key[s] == thr);
      if (!placed && m != 0ull) {
        if (lane == __ffsll(m) - 1) key[s] = k;
        placed = true;
      }
    }
    uint64_t local = key[0];
#pragma unroll
    for (int s = 1; s < kSlots; ++s) local = key[s] < local ? key[s] : local;
    thr = wave_min_u64(local);
    thr_s = thr == kKeyEmpty ? -__builtin_inff() : key_score(thr);
  }
  __device__ __forceinline__ void store(uint64_t* dst, int c, int lane) const {
#pragma unroll
    for (int s = 0; s < kSlots; ++s) {
      const int p = s * kWave + lane;
      if (p < c) dst[p] = key[s];
    }
  }
};

// Block-level merge of the per-wave lists (single-slot lists, c <= 64) into ONE list per workgroup,
// sorted descending.  The workgroup's c-th best key is at least every wave's own c-th best (`thr`),
// so only keys at or above the largest of the eight `thr` can survive: usually c..2c of the 8c keys.
// Survivors are compacted into LDS (ballot prefix per wave) and each ranks itself against the
// others (broadcast reads) — ~30 iterations instead of 8c = 160; this tail was 6 us of a 62 us scan
// at 125 K rows per GPU.  Empty keys never survive; missing positions are written as empty.
struct MergeShared {
  uint64_t key[kScanThreads];
  uint64_t thr[kScanThreads / kWave];
  uint32_t count;
};

__device__ __forceinline__ void block_merge_store(const WaveList<1>& lst, MergeShared& sh, uint64_t* __restrict__ dst,
                                                  int c, int lane, int wave_in_block) {
  constexpr int kWavesPerBlock = kScanThreads / kWave;
  __syncthreads();  // sh may still be read by the previous query's merge
  if (lane == 0) sh.thr[wave_in_block] = lst.thr;
  if (threadIdx.x == 0) sh.count = 0;
  __syncthreads();
  uint64_t bound = sh.thr[0];
#pragma unroll
  for (int w = 1; w < kWavesPerBlock; ++w) bound = sh.thr[w] > bound ? sh.thr[w] : bound;
  const uint64_t mine = lst.key[0];
  const bool keep = lane < c && mine != kKeyEmpty && mine >= bound;
  const unsigned long long mask = __ballot(keep);
  uint32_t base = 0;
  if (lane == 0 && mask != 0ull) base = atomicAdd(&sh.count, static_cast<uint32_t>(__popcll(mask)));
  base = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(base)));
  if (keep) sh.key[base + __popcll(mask & ((1ull << lane) - 1ull))] = mine;   // <= 8c <= kScanThreads entries
  __syncthreads();
  const int total = static_cast<int>(sh.count);
  const int i = static_cast<int>(threadIdx.x);
  if (i < total) {
    const uint64_t k = sh.key[i];
    int rank = 0;
    for (int j = 0; j < total; ++j) rank += sh.key[j] > k ? 1 : 0;   // survivors are real keys: all distinct
    if (rank < c) dst[rank] = k;
  } else if (i < c) {
    dst[i] = kKeyEmpty;  // fewer than c rows were seen by this workgroup
  }
}

}  // namespace dewi
