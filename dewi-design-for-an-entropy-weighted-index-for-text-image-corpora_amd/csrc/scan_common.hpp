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

// ---- fp32 rows: 16-byte units of 4 columns; x y z w in this order (the exact re-scoring of select_rerank.hip repeats it) ----
template <int SPACE>
__device__ __forceinline__ float accum4(f32x4 e, f32x4 q, float acc) {
  if constexpr (SPACE == DEWI_SPACE_COSINE) {
    acc = __builtin_fmaf(e.x, q.x, acc);
    acc = __builtin_fmaf(e.y, q.y, acc);
    acc = __builtin_fmaf(e.z, q.z, acc);
    acc = __builtin_fmaf(e.w, q.w, acc);
  } else {
    float d;
    d = e.x - q.x; acc = __builtin_fmaf(d, d, acc);
    d = e.y - q.y; acc = __builtin_fmaf(d, d, acc);
    d = e.z - q.z; acc = __builtin_fmaf(d, d, acc);
    d = e.w - q.w; acc = __builtin_fmaf(d, d, acc);
  }
  return acc;
}

// ---- bf16 rows: 16-byte units of 8 columns (knn_scan_bf16.hip, knn_scan_any.hpp) ----
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <bool NT>
__device__ __forceinline__ u32x4 load_u4(const u32x4* p) {
  if constexpr (NT) return __builtin_nontemporal_load(p);
  return *p;
}

// fp32 -> nearest-even bf16, returned as the fp32 value it represents (NaN stays NaN).
__device__ __forceinline__ float round_to_bf16(float f) {
  if (f != f) return f;
  const uint32_t u = __float_as_uint(f);
  return __uint_as_float((u + 0x7FFFu + ((u >> 16) & 1u)) & 0xFFFF0000u);
}

typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

// Cosine: 8 bf16 x bf16 products accumulated in fp32 with v_dot2c_f32_bf16 — the corpus dwords and the
// packed query dwords are both (element 2i | element 2i+1 << 16), so no unpacking at all: 4 VALU
// instructions per 16 bytes instead of 16 (shift, mask, 2 FMAs per dword).
__device__ __forceinline__ float dot8_packed(u32x4 e, const uint32_t (&q)[4], float acc) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    // copy the lane to a scalar first: __builtin_bit_cast applied directly to a vector subscript
    // (e[i]) made hipcc 7.2 use element 0 for every i
    const uint32_t ew = e[i], qw = q[i];
    acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, ew), __builtin_bit_cast(bf16x2, qw), acc, false);
  }
  return acc;
}

template <int SPACE>
__device__ __forceinline__ float dot8(u32x4 e, const float (&q)[8], float acc) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float lo = __uint_as_float(e[i] << 16);
    const float hi = __uint_as_float(e[i] & 0xFFFF0000u);
    if constexpr (SPACE == DEWI_SPACE_COSINE) {
      acc = __builtin_fmaf(lo, q[2 * i], acc);
      acc = __builtin_fmaf(hi, q[2 * i + 1], acc);
    } else {
      const float d0 = lo - q[2 * i], d1 = hi - q[2 * i + 1];
      acc = __builtin_fmaf(d0, d0, acc);
      acc = __builtin_fmaf(d1, d1, acc);
    }
  }
  return acc;
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

// Repair launches: call f(q) for every query whose flag is set.  One vector load + one ballot per 64 queries (a loop of
// scalar loads, one per query, cost a 256-query batch ~25 us of latency in every workgroup even with nothing to repair).
template <class F>
__device__ __forceinline__ void for_each_flagged(const uint32_t* __restrict__ flags, int n_queries, F f) {
  const int lane = lane_id();
  for (int base = 0; base < n_queries; base += kWave) {
    const int q = base + lane;
    const uint32_t v = q < n_queries ? flags[q] : 0u;
    unsigned long long m = __ballot(v != 0u);
    while (m != 0ull) {   // wave-uniform
      const int j = __ffsll(m) - 1;
      m &= m - 1ull;
      f(base + j);
    }
  }
}

}  // namespace dewi
