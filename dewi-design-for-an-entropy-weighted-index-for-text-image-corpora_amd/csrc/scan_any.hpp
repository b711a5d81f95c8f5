// Row scan for ANY embedding width whose rows are whole 16-byte units (fp32: dim % 4 == 0, bf16: dim % 8 == 0), gfx950.
//
// Same contract as knn_scan.hip (steps 1-3 of ExactIndex.search, reference src/dewi/backends.py:420-444 — one BLAS call
// for every dim there, `np.dot(self._embeddings, query)` :431-433): normalise the query, score every row, keep the best c
// keys per wave in registers.  knn_scan.hip / knn_scan_bf16.hip keep the widths they were tuned on (dim = 256 U);
// everything else used to fall to scan_generic_*, which re-read the queries from global memory inside the inner loop and
// kept one short row in flight per wave (0.24-0.44 of HBM at 3 GB, profiles/r04/dims_before).  Two kernels replace it:
//
//   scan_rows_any        rows of 33 .. 64 U units.  One row per wave step, lane l holds the units l + 64 u of the row and
//                        the matching query fragments IN REGISTERS; units past the end of the row are predicated off (EXEC
//                        mask: no bytes move for them, the registers read as zero).  R rows are issued back to back so that
//                        a wave keeps ~3-4 KiB in flight whatever the width.
//   scan_short_rows_any  rows of 1 .. 32 units: P = 2^log2p lanes share a row, a wave-instruction loads 64 / P CONSECUTIVE
//                        rows (one contiguous span), R such loads are in flight; the P partial sums are folded on the DPP
//                        crossbar (quad_perm, row_half_mirror, row_mirror, row_bcast15) and the lane that ends up with a row's
//                        score offers it to the wave's list — a ballot finds the few rows worth an offer.
//
// Roofline: HBM.  Algorithmic bytes per launch = n_rows * dim * sizeof(elem) + n_queries * dim * 4.
#pragma once
#include "scan_common.hpp"

namespace dewi {

// ---------------------------------------------------------------------------------------------
// The query values that face one 16-byte unit of a row
// ---------------------------------------------------------------------------------------------
template <int ELEM, int SPACE>
struct UnitFrag;

template <int SPACE>
struct UnitFrag<0, SPACE> {   // fp32 rows: 4 columns per unit
  f32x4 q;
  __device__ __forceinline__ void load(const float* __restrict__ qrow, int unit, bool active) {
    q = f32x4{0.f, 0.f, 0.f, 0.f};
    if (active) q = reinterpret_cast<const f32x4*>(qrow)[unit];
  }
  __device__ __forceinline__ double sumsq() const { return square_f64(q.x) + square_f64(q.y) + square_f64(q.z) + square_f64(q.w); }
  __device__ __forceinline__ void scale(float norm) {
    q.x = __fdiv_rn(q.x, norm);
    q.y = __fdiv_rn(q.y, norm);
    q.z = __fdiv_rn(q.z, norm);
    q.w = __fdiv_rn(q.w, norm);
  }
  __device__ __forceinline__ void finish() {}
  __device__ __forceinline__ float dot(u32x4 e, float acc) const { return accum4<SPACE>(__builtin_bit_cast(f32x4, e), q, acc); }
};

template <int SPACE>
struct UnitFrag<1, SPACE> {   // bf16 rows: 8 columns per unit; the prepared query is rounded to bf16 (knn_scan_bf16.hip)
  float f[8];
  uint32_t p[4];              // cosine: packed pairs for v_dot2c_f32_bf16
  __device__ __forceinline__ void load(const float* __restrict__ qrow, int unit, bool active) {
    f32x4 a = {0.f, 0.f, 0.f, 0.f}, b = {0.f, 0.f, 0.f, 0.f};
    if (active) {
      a = reinterpret_cast<const f32x4*>(qrow)[2 * unit];
      b = reinterpret_cast<const f32x4*>(qrow)[2 * unit + 1];
    }
    f[0] = a.x; f[1] = a.y; f[2] = a.z; f[3] = a.w;
    f[4] = b.x; f[5] = b.y; f[6] = b.z; f[7] = b.w;
  }
  __device__ __forceinline__ double sumsq() const {
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += square_f64(f[i]);
    return s;
  }
  __device__ __forceinline__ void scale(float norm) {
#pragma unroll
    for (int i = 0; i < 8; ++i) f[i] = __fdiv_rn(f[i], norm);
  }
  __device__ __forceinline__ void finish() {
#pragma unroll
    for (int i = 0; i < 8; ++i) f[i] = round_to_bf16(f[i]);
    if constexpr (SPACE == DEWI_SPACE_COSINE) {
#pragma unroll
      for (int i = 0; i < 4; ++i) p[i] = (__float_as_uint(f[2 * i]) >> 16) | (__float_as_uint(f[2 * i + 1]) & 0xFFFF0000u);
    }
  }
  __device__ __forceinline__ float dot(u32x4 e, float acc) const {
    if constexpr (SPACE == DEWI_SPACE_COSINE) return dot8_packed(e, p, acc);
    else return dot8<SPACE>(e, f, acc);
  }
};

// Rows a wave issues before its first reduction.  Measured on MI355X at 3 GB (scripts/sweep_any.sh, profiles/r04): rows
// that are not whole KiB want ~5-6 KiB in flight per wave — more than the 3 KiB of the dim = 768 kernel, because the
// reduction and the list offer come round more often per byte and a wave shares its first and last cache line with its
// neighbours.  TB/s by (row bytes, rows in flight): 640 B: 8 rows 6.47, 12 rows 6.92; 768 B: 8 -> 6.99, 12 -> 6.87;
// 1040 B: 3 -> 5.81, 4 -> 6.45; 1280 B: 3 -> 6.67, 4 -> 7.07; 1536 B: 2 -> 6.33, 3 -> 7.07, 4 -> 6.95; 2064 B: 1 -> 4.94,
// 2 -> 6.69; 2800 B: 1 -> 6.25, 2 -> 6.97.  So up to three choices per units-per-lane count, picked by the row's width.
#ifndef DEWI_ANY_RSHORT
#define DEWI_ANY_RSHORT 8
#endif
constexpr int kAnyLevels = 3;
constexpr int any_level(int units) {            // 0: the narrowest rows of a units-per-lane count ... 2: the widest
  const int u = (units + 63) / 64;
  if (u == 1) return units <= 40 ? 0 : 1;                        // 528-640 B : 656-1024 B
  if (u == 2) return units <= 72 ? 0 : (units <= 92 ? 1 : 2);    // 1040-1152 B : 1168-1472 B : 1488-2048 B (1408 B: 4 rows 6.98 TB/s,
                                                                 // 3 rows 6.86; 1424 B: 6.97 / 6.92; 1536 B: 6.95 / 7.07)
  if (u == 4) return units <= 224 ? 0 : 1;                       // 3088-3584 B : 3600-4096 B
  return 1;
}
constexpr int any_rows(int u, int nq, int level) {
#ifdef DEWI_ANY_NQ4_OLD
  if (nq >= 4) {                     // four queries behind every row: the arithmetic hides more of the latency
    if (u == 1) return level == 0 ? 6 : 4;
    if (u <= 3) return 2;
    return 1;
  }
#endif
  if (u == 1) return level == 0 ? 12 : 8;
  if (u == 2) return level == 0 ? 6 : (level == 1 ? 4 : 3);
  if (u == 3) return 2;
  if (u == 4) return level == 0 ? 2 : 1;
  return 1;
}
// most queries one corpus pass serves besides 1 (launch_any_long instantiates exactly these)
constexpr int any_nq_max(int elem_bytes, int u_pad) {
  if (elem_bytes == 2) return u_pad <= 4 ? 4 : (u_pad <= 8 ? 2 : 1);
  return u_pad <= 8 ? 4 : 2;
}
constexpr int kAnyShortRows = DEWI_ANY_RSHORT;   // loads (of 64 / P rows each) a wave of the short-row kernel keeps in flight

// ---------------------------------------------------------------------------------------------
// 33 .. 64 U units per row
// ---------------------------------------------------------------------------------------------
template <int ELEM, int U, int R, int NQ, int SPACE, int S>
__device__ __forceinline__ void scan_rows_any_body(const u32x4* __restrict__ E, int64_t n_rows, int units,
                                                   const float* __restrict__ Q, int n_candidates,
                                                   uint64_t* __restrict__ keys, int64_t keys_per_query, MergeShared& merge_buf) {
  constexpr int kCols = ELEM ? 8 : 4;
  constexpr bool DENSE = S == 0;
  const int lane = lane_id();
  const int wave_in_block = static_cast<int>(threadIdx.x) >> 6;
  const int64_t gwave = static_cast<int64_t>(blockIdx.x) * (kScanThreads / kWave) + wave_in_block;
  const int64_t n_waves = static_cast<int64_t>(gridDim.x) * (kScanThreads / kWave);
  const int dim = units * kCols;

  bool act[U];
#pragma unroll
  for (int u = 0; u < U; ++u) act[u] = lane + 64 * u < units;

  // Query fragments, normalised here for cosine (reference backends.py:420-424) with the float64-summed norm every
  // kernel uses (common.hpp wave_query_norm): a query has one prepared form whatever kernel serves it.
  UnitFrag<ELEM, SPACE> qf[NQ][U];
#pragma unroll
  for (int qi = 0; qi < NQ; ++qi) {
    const float* qrow = Q + static_cast<int64_t>(qi) * dim;
    double ss = 0.0;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      qf[qi][u].load(qrow, lane + 64 * u, act[u]);
      ss += qf[qi][u].sumsq();
    }
    if constexpr (SPACE == DEWI_SPACE_COSINE) {
      const float norm = wave_query_norm(ss);
      if (norm > 0.f) {
#pragma unroll
        for (int u = 0; u < U; ++u) qf[qi][u].scale(norm);
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) qf[qi][u].finish();
  }

  WaveList<DENSE ? 1 : S> lst[DENSE ? 1 : NQ];
  if constexpr (!DENSE) {
#pragma unroll
    for (int qi = 0; qi < NQ; ++qi) lst[qi].init(n_candidates, lane);
  }

  auto fetch = [&](u32x4(&v)[U], int64_t row) {
    const u32x4* p = E + row * units + lane;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      v[u] = u32x4{0u, 0u, 0u, 0u};
      if (act[u]) v[u] = load_u4<true>(p + 64 * u);
    }
  };
  auto consume = [&](const u32x4(&v)[U], int64_t row) {
#pragma unroll
    for (int qi = 0; qi < NQ; ++qi) {
      float acc = 0.f;
#pragma unroll
      for (int u = 0; u < U; ++u) acc = qf[qi][u].dot(v[u], acc);
      float s = wave_sum_f32(acc);
      if constexpr (SPACE == DEWI_SPACE_L2) s = -s;
      if constexpr (DENSE) {
        if (lane == 0) keys[qi * keys_per_query + row] = make_key(s, static_cast<uint32_t>(row));
      } else {
        lst[qi].offer(s, static_cast<uint32_t>(row), lane);
      }
    }
  };

  const int64_t n_groups = n_rows / R;
  for (int64_t g = gwave; g < n_groups; g += n_waves) {
    u32x4 v[R][U];
#pragma unroll
    for (int r = 0; r < R; ++r) fetch(v[r], g * R + r);
#pragma unroll
    for (int r = 0; r < R; ++r) consume(v[r], g * R + r);
  }
  for (int64_t row = n_groups * R + gwave; row < n_rows; row += n_waves) {   // fewer than R rows left
    u32x4 v[U];
    fetch(v, row);
    consume(v, row);
  }

  if constexpr (S == 1) {
#pragma unroll
    for (int qi = 0; qi < NQ; ++qi)
      block_merge_store(lst[qi], merge_buf, keys + qi * keys_per_query + static_cast<int64_t>(blockIdx.x) * n_candidates,
                        n_candidates, lane, wave_in_block);
  } else if constexpr (!DENSE) {
#pragma unroll
    for (int qi = 0; qi < NQ; ++qi)
      lst[qi].store(keys + qi * keys_per_query + gwave * n_candidates, n_candidates, lane);
  }
}

template <int ELEM, int U, int R, int NQ, int SPACE, int S>
__global__ __launch_bounds__(kScanThreads) void scan_rows_any(const u32x4* __restrict__ E, int64_t n_rows, int units,
                                                              const float* __restrict__ Q, int n_candidates,
                                                              uint64_t* __restrict__ keys, int64_t keys_per_query) {
  __shared__ MergeShared merge_buf;
  scan_rows_any_body<ELEM, U, R, NQ, SPACE, S>(E, n_rows, units, Q, n_candidates, keys, keys_per_query, merge_buf);
}

// REPAIR form: every flagged query of a batch, one corpus pass each, in one launch (knn_scan.hip scan_rows_f32_flagged)
template <int ELEM, int U, int R, int SPACE, int S>
__global__ __launch_bounds__(kScanThreads) void scan_rows_any_flagged(const u32x4* __restrict__ E, int64_t n_rows, int units,
                                                                      const float* __restrict__ Q, int n_candidates,
                                                                      uint64_t* __restrict__ keys, int64_t keys_per_query,
                                                                      const uint32_t* __restrict__ flags, int n_queries) {
  __shared__ MergeShared merge_buf;
  const int dim = units * (ELEM ? 8 : 4);
  for_each_flagged(flags, n_queries, [&](int q) {
    scan_rows_any_body<ELEM, U, R, 1, SPACE, S>(E, n_rows, units, Q + static_cast<int64_t>(q) * dim, n_candidates,
                                                keys + static_cast<int64_t>(q) * keys_per_query, keys_per_query, merge_buf);
  });
}

// ---------------------------------------------------------------------------------------------
// 1 .. 32 units per row: P = 2^log2p lanes per row
// ---------------------------------------------------------------------------------------------
// Sum over each aligned group of 2^log2p lanes, in a fixed order.  Valid in (at least) the LAST lane of every group:
// up to 16 lanes every lane of the group holds the sum; the 32-lane step (row_bcast15) leaves it in DPP rows 1 and 3.
__device__ __forceinline__ float group_sum_f32(float v, int log2p) {
#define DEWI_STEP(CTRL, MASK) v = v + __int_as_float(dpp_i32<CTRL, MASK>(0, __float_as_int(v)));
  if (log2p >= 1) { DEWI_STEP(0xB1, 0xF) }    // quad_perm [1,0,3,2]
  if (log2p >= 2) { DEWI_STEP(0x4E, 0xF) }    // quad_perm [2,3,0,1]
  if (log2p >= 3) { DEWI_STEP(0x141, 0xF) }   // row_half_mirror: the other quad of the 8
  if (log2p >= 4) { DEWI_STEP(0x140, 0xF) }   // row_mirror: the other 8 of the 16
  if (log2p >= 5) { DEWI_STEP(0x142, 0xA) }   // row_bcast15 into rows 1, 3: the other 16 of the 32
#undef DEWI_STEP
  return v;
}

template <int ELEM, int R, int NQ, int SPACE, int S>
__device__ __forceinline__ void scan_short_rows_any_body(const u32x4* __restrict__ E, int64_t n_rows, int units, int log2p,
                                                         const float* __restrict__ Q, int n_candidates,
                                                         uint64_t* __restrict__ keys, int64_t keys_per_query,
                                                         MergeShared& merge_buf) {
  constexpr int kCols = ELEM ? 8 : 4;
  constexpr bool DENSE = S == 0;
  const int lane = lane_id();
  const int wave_in_block = static_cast<int>(threadIdx.x) >> 6;
  const int64_t gwave = static_cast<int64_t>(blockIdx.x) * (kScanThreads / kWave) + wave_in_block;
  const int64_t n_waves = static_cast<int64_t>(gridDim.x) * (kScanThreads / kWave);
  const int dim = units * kCols;
  const int group = 1 << log2p;           // lanes per row
  const int rows_per_load = kWave >> log2p;
  const int sub = lane >> log2p;          // which row of the load
  const int pos = lane & (group - 1);     // which unit of the row
  const bool act = pos < units;
  const bool holder = pos == group - 1;   // the lane group_sum_f32 leaves the row's score in

  UnitFrag<ELEM, SPACE> qf[NQ];
#pragma unroll
  for (int qi = 0; qi < NQ; ++qi) {
    const float* qrow = Q + static_cast<int64_t>(qi) * dim;
    qf[qi].load(qrow, pos, act);
    if constexpr (SPACE == DEWI_SPACE_COSINE) {
      // every lane group holds a copy of the query: the first one alone feeds the norm
      const float norm = wave_query_norm(sub == 0 ? qf[qi].sumsq() : 0.0);
      if (norm > 0.f) qf[qi].scale(norm);
    }
    qf[qi].finish();
  }

  WaveList<DENSE ? 1 : S> lst[DENSE ? 1 : NQ];
  if constexpr (!DENSE) {
#pragma unroll
    for (int qi = 0; qi < NQ; ++qi) lst[qi].init(n_candidates, lane);
  }

  const int64_t rows_per_step = static_cast<int64_t>(rows_per_load) * R;
  const int64_t n_steps = (n_rows + rows_per_step - 1) / rows_per_step;
  for (int64_t st = gwave; st < n_steps; st += n_waves) {
    const int64_t row_first = st * rows_per_step + sub;
    u32x4 v[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int64_t row = row_first + static_cast<int64_t>(r) * rows_per_load;
      v[r] = u32x4{0u, 0u, 0u, 0u};
      if (act && row < n_rows) v[r] = load_u4<true>(E + row * units + pos);
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int64_t row = row_first + static_cast<int64_t>(r) * rows_per_load;
      const bool mine = holder && row < n_rows;
#pragma unroll
      for (int qi = 0; qi < NQ; ++qi) {
        float s = group_sum_f32(qf[qi].dot(v[r], 0.f), log2p);
        if constexpr (SPACE == DEWI_SPACE_L2) s = -s;
        if constexpr (DENSE) {
          if (mine) keys[qi * keys_per_query + row] = make_key(s, static_cast<uint32_t>(row));
        } else {
          // rows worth an offer: not below the list's current worst (NaN rows pass, as in WaveList::offer)
          unsigned long long m = __ballot(mine && !(s < lst[qi].thr_s));
          while (m != 0ull) {
            const int src = __ffsll(m) - 1;
            m &= m - 1ull;
            const float sc = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(s), src));
            const uint32_t rw = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(static_cast<uint32_t>(row)), src));
            lst[qi].offer(sc, rw, lane);
          }
        }
      }
    }
  }

  if constexpr (S == 1) {
#pragma unroll
    for (int qi = 0; qi < NQ; ++qi)
      block_merge_store(lst[qi], merge_buf, keys + qi * keys_per_query + static_cast<int64_t>(blockIdx.x) * n_candidates,
                        n_candidates, lane, wave_in_block);
  } else if constexpr (!DENSE) {
#pragma unroll
    for (int qi = 0; qi < NQ; ++qi)
      lst[qi].store(keys + qi * keys_per_query + gwave * n_candidates, n_candidates, lane);
  }
}

template <int ELEM, int R, int NQ, int SPACE, int S>
__global__ __launch_bounds__(kScanThreads) void scan_short_rows_any(const u32x4* __restrict__ E, int64_t n_rows, int units,
                                                                    int log2p, const float* __restrict__ Q, int n_candidates,
                                                                    uint64_t* __restrict__ keys, int64_t keys_per_query) {
  __shared__ MergeShared merge_buf;
  scan_short_rows_any_body<ELEM, R, NQ, SPACE, S>(E, n_rows, units, log2p, Q, n_candidates, keys, keys_per_query, merge_buf);
}

template <int ELEM, int R, int SPACE, int S>
__global__ __launch_bounds__(kScanThreads) void scan_short_rows_any_flagged(const u32x4* __restrict__ E, int64_t n_rows, int units,
                                                                            int log2p, const float* __restrict__ Q,
                                                                            int n_candidates, uint64_t* __restrict__ keys,
                                                                            int64_t keys_per_query,
                                                                            const uint32_t* __restrict__ flags, int n_queries) {
  __shared__ MergeShared merge_buf;
  const int dim = units * (ELEM ? 8 : 4);
  for_each_flagged(flags, n_queries, [&](int q) {
    scan_short_rows_any_body<ELEM, R, 1, SPACE, S>(E, n_rows, units, log2p, Q + static_cast<int64_t>(q) * dim, n_candidates,
                                                   keys + static_cast<int64_t>(q) * keys_per_query, keys_per_query, merge_buf);
  });
}

// ---------------------------------------------------------------------------------------------
// dispatch (one translation unit per element type instantiates it: knn_scan_any_f32.hip, knn_scan_any_bf16.hip)
// ---------------------------------------------------------------------------------------------
template <int ELEM, int NQ, int SPACE, int S>
static hipError_t launch_any_long(const ScanPlan& plan, const u32x4* E, int64_t n_rows, const float* Q, int c, uint64_t* keys,
                                  hipStream_t stream) {
#define DEWI_ANY_LAUNCH(UU, RR)                                                                                          \
  hipLaunchKernelGGL((scan_rows_any<ELEM, UU, RR, NQ, SPACE, S>), dim3(plan.blocks), dim3(kScanThreads), 0, stream, E, n_rows, \
                     plan.units, Q, c, keys, plan.keys_per_query);                                                      \
  return hipGetLastError();
#define DEWI_ANY_CASE(UU)                                                                                       \
  case UU:                                                                                                      \
    if constexpr (any_rows(UU, NQ, 0) != any_rows(UU, NQ, 1)) {                                                 \
      if (plan.level == 0) { DEWI_ANY_LAUNCH(UU, any_rows(UU, NQ, 0)) }                                         \
    }                                                                                                           \
    if constexpr (any_rows(UU, NQ, 2) != any_rows(UU, NQ, 1)) {                                                 \
      if (plan.level == 2) { DEWI_ANY_LAUNCH(UU, any_rows(UU, NQ, 2)) }                                         \
    }                                                                                                           \
    DEWI_ANY_LAUNCH(UU, any_rows(UU, NQ, 1))
  // Queries per pass by row width (the fragments of NQ queries must fit the registers next to the rows in flight): an fp32
  // unit keeps 4 registers per query, a bf16 unit 4 (cosine, packed pairs) or 8 (l2) — any_nq_max() is the same table
  constexpr bool kNarrow = ELEM == 0 ? (NQ == 1 || NQ == 4) : true;     // units per lane 1 .. 4 (bf16) / 1 .. 8 (fp32)
  constexpr bool kMiddle = ELEM == 0 ? (NQ == 1 || NQ == 4) : (NQ == 1 || NQ == 2);   // 5 .. 8
  constexpr bool kWide = ELEM == 0 ? (NQ == 1 || NQ == 2) : NQ == 1;    // 10 .. 16
  if constexpr (kNarrow && NQ != 2) {
    switch (plan.u_pad) {
      DEWI_ANY_CASE(1)
      DEWI_ANY_CASE(2)
      DEWI_ANY_CASE(3)
      DEWI_ANY_CASE(4)
      default: break;
    }
  }
  if constexpr (kMiddle) {
    switch (plan.u_pad) {
      DEWI_ANY_CASE(5)
      DEWI_ANY_CASE(6)
      DEWI_ANY_CASE(8)
      default: break;
    }
  }
  if constexpr (kWide) {
    switch (plan.u_pad) {
      DEWI_ANY_CASE(10)
      DEWI_ANY_CASE(12)
      DEWI_ANY_CASE(16)
      default: break;
    }
  }
#undef DEWI_ANY_LAUNCH
#undef DEWI_ANY_CASE
  return hipErrorInvalidValue;
}

template <int ELEM, int NQ, int SPACE, int S>
static hipError_t launch_any_kind(const ScanPlan& plan, const u32x4* E, int64_t n_rows, const float* Q, int c, uint64_t* keys,
                                  hipStream_t stream) {
  if (plan.kind == kScanAnyShort) {
    if constexpr (NQ == 2) {
      return hipErrorInvalidValue;
    } else {
      hipLaunchKernelGGL((scan_short_rows_any<ELEM, kAnyShortRows, NQ, SPACE, S>), dim3(plan.blocks), dim3(kScanThreads), 0, stream,
                         E, n_rows, plan.units, plan.log2p, Q, c, keys, plan.keys_per_query);
      return hipGetLastError();
    }
  }
  return launch_any_long<ELEM, NQ, SPACE, S>(plan, E, n_rows, Q, c, keys, stream);
}

template <int ELEM>
static hipError_t launch_scan_any_impl(const ScanPlan& plan, const void* d_E, int64_t n_rows, int dim, const float* d_q_raw, int q0,
                                       int nq, int n_candidates, int space, uint64_t* d_keys, hipStream_t stream) {
  const u32x4* E = static_cast<const u32x4*>(d_E);
  const float* Q = d_q_raw + static_cast<int64_t>(q0) * dim;
  uint64_t* keys = d_keys + static_cast<int64_t>(q0) * plan.keys_per_query;
#define DEWI_ANY_S(NQ, SPACE)                                                                                        \
  switch (plan.slots) {                                                                                              \
    case 0: return launch_any_kind<ELEM, NQ, SPACE, 0>(plan, E, n_rows, Q, n_candidates, keys, stream);              \
    case 1: return launch_any_kind<ELEM, NQ, SPACE, 1>(plan, E, n_rows, Q, n_candidates, keys, stream);              \
    default: return launch_any_kind<ELEM, NQ, SPACE, kMaxSlots>(plan, E, n_rows, Q, n_candidates, keys, stream);     \
  }
#define DEWI_ANY_Q(NQ)                 \
  if (space == DEWI_SPACE_COSINE) {    \
    DEWI_ANY_S(NQ, DEWI_SPACE_COSINE)  \
  } else {                             \
    DEWI_ANY_S(NQ, DEWI_SPACE_L2)      \
  }
  if (nq == 1) {
    DEWI_ANY_Q(1)
  } else if (nq == 2) {
    DEWI_ANY_Q(2)
  } else if (nq == 4) {
    DEWI_ANY_Q(4)
  }
#undef DEWI_ANY_Q
#undef DEWI_ANY_S
  return hipErrorInvalidValue;
}

// flagged (repair) launches: the middle rows-in-flight level, rows of at most 512 units (see scan_flagged_supported)
template <int ELEM, int SPACE, int S>
static hipError_t launch_any_flagged_s(const ScanPlan& plan, const u32x4* E, int64_t n_rows, const float* Q, int n_queries, int c,
                                       uint64_t* keys, const uint32_t* flags, hipStream_t stream) {
  if (plan.kind == kScanAnyShort) {
    hipLaunchKernelGGL((scan_short_rows_any_flagged<ELEM, kAnyShortRows, SPACE, S>), dim3(plan.blocks), dim3(kScanThreads), 0, stream,
                       E, n_rows, plan.units, plan.log2p, Q, c, keys, plan.keys_per_query, flags, n_queries);
    return hipGetLastError();
  }
#define DEWI_ANY_FLAGGED(UU)                                                                                                     \
  case UU:                                                                                                                       \
    hipLaunchKernelGGL((scan_rows_any_flagged<ELEM, UU, any_rows(UU, 1, 1), SPACE, S>), dim3(plan.blocks), dim3(kScanThreads), 0,  \
                       stream, E, n_rows, plan.units, Q, c, keys, plan.keys_per_query, flags, n_queries);                         \
    return hipGetLastError();
  switch (plan.u_pad) {
    DEWI_ANY_FLAGGED(1)
    DEWI_ANY_FLAGGED(2)
    DEWI_ANY_FLAGGED(3)
    DEWI_ANY_FLAGGED(4)
    DEWI_ANY_FLAGGED(5)
    DEWI_ANY_FLAGGED(6)
    DEWI_ANY_FLAGGED(8)
    default: break;
  }
#undef DEWI_ANY_FLAGGED
  return hipErrorInvalidValue;
}

template <int ELEM>
static hipError_t launch_scan_any_flagged_impl(const ScanPlan& plan, const void* d_E, int64_t n_rows, const float* d_q_raw,
                                               int n_queries, int n_candidates, int space, uint64_t* d_keys,
                                               const uint32_t* d_flags, hipStream_t stream) {
  const u32x4* E = static_cast<const u32x4*>(d_E);
#define DEWI_ANY_FS(SPACE)                                                                                                         \
  switch (plan.slots) {                                                                                                            \
    case 0: return launch_any_flagged_s<ELEM, SPACE, 0>(plan, E, n_rows, d_q_raw, n_queries, n_candidates, d_keys, d_flags, stream); \
    case 1: return launch_any_flagged_s<ELEM, SPACE, 1>(plan, E, n_rows, d_q_raw, n_queries, n_candidates, d_keys, d_flags, stream); \
    default: return launch_any_flagged_s<ELEM, SPACE, kMaxSlots>(plan, E, n_rows, d_q_raw, n_queries, n_candidates, d_keys, d_flags, stream); \
  }
  if (space == DEWI_SPACE_COSINE) {
    DEWI_ANY_FS(DEWI_SPACE_COSINE)
  } else {
    DEWI_ANY_FS(DEWI_SPACE_L2)
  }
#undef DEWI_ANY_FS
}

}  // namespace dewi
