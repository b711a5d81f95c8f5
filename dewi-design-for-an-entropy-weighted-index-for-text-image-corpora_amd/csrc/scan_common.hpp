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

// Block-level merge of the per-wave lists (single-slot lists, c <= 64): every wave drops its
// c keys into LDS, each thread ranks one key against all of them (broadcast reads), and the c best
// leave the kernel already sorted descending.  Empty keys tie at 0 and are ordered by position.
__device__ __forceinline__ void block_merge_store(const WaveList<1>& lst, uint64_t* __restrict__ sh,
                                                  uint64_t* __restrict__ dst, int c, int lane, int wave_in_block) {
  constexpr int kWavesPerBlock = kScanThreads / kWave;
  __syncthreads();  // sh may still be read by the previous query's merge
  if (lane < c) sh[wave_in_block * c + lane] = lst.key[0];
  __syncthreads();
  const int total = kWavesPerBlock * c;  // <= blockDim because c <= 64
  const int i = static_cast<int>(threadIdx.x);
  if (i < total) {
    const uint64_t mine = sh[i];
    int rank = 0;
    for (int j = 0; j < total; ++j) {
      const uint64_t o = sh[j];
      rank += (o > mine || (o == mine && j < i)) ? 1 : 0;
    }
    if (rank < c) dst[rank] = mine;
  }
}

}  // namespace dewi
